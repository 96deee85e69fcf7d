#!/bin/bash
# PMC traffic passes (separate --pmc runs) of configs 4 and 5; the leading probe launches of the factorisation are skipped
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void patch_apply_kernel" $O/pmc_patch_apply_cfg4.json "r02 ($STAMP) patch_apply_kernel, the 60 launches of the first V-cycle" 60 - 3
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void bsr_spmv_flat_kernel" $O/pmc_bsr_spmv_cfg4.json "r02 ($STAMP) bsr_spmv_flat_kernel, finest-level launches (grid 28487168... selected by size)" 0 -
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_fetch.json 2> $O/pmc5_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_write.json 2> $O/pmc5_write.err
python scripts/pmc_summary.py $O/pmc5_fetch $O/pmc5_write "void cond_apply_kernel" $O/pmc_patch_apply_cfg5.json "r02 ($STAMP) cond_apply_kernel (condensed macro-star factors), the 40 launches of the first V-cycle" 40 - 2
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/pmc_fetch/**/*_counter_collection.csv", recursive=True)[0]
t = pd.read_csv(f)
t = t[t["Kernel_Name"].str.startswith("void bsr_spmv_flat_kernel")]
print(t.groupby("Grid_Size")["Counter_Value"].agg(["count", "mean"]).sort_values("mean").tail(6))
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc5_fetch $O/pmc5_write
cat $O/pmc_patch_apply_cfg4.json $O/pmc_patch_apply_cfg5.json
