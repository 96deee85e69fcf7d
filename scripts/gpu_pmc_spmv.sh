#!/bin/bash
# FETCH_SIZE pass over one V-cycle of a configuration; per grid size of the SpMV kernels: launches and mean fetched bytes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${CFG:-cfg6}
O=gpurun_out/pmc_$CFG
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --config $CFG --no-cpu-baseline --steps 1 --warmup 0 > $O/fetch.json 2> $O/fetch.err
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/fetch/**/*_counter_collection.csv", recursive=True)[0]
t = pd.read_csv(f)
t = t[t["Kernel_Name"].str.contains("spmv")]
t["name"] = t["Kernel_Name"].str.slice(0, 60)
g = t.groupby(["name", "Grid_Size", "Dispatch_Id"])["Counter_Value"].sum().reset_index()
out = g.groupby(["name", "Grid_Size"])["Counter_Value"].agg(["count", "mean"])
out["fetch_GB_corrected"] = out["mean"] * 2 * 1024 / 1e9
print(out.sort_values("mean").tail(8).to_string())
PY
rm -rf $O/fetch
