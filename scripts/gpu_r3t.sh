#!/bin/bash
# kernel traces of config 5 with the three-launch condensed apply under several settings: where the time of an apply goes
# usage: gpu_r3t.sh "ENV1=a ENV2=b" "ENV1=c" ...   (one traced run per argument; "-" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $O
cd /tmp
i=0
for v in "$@"; do
  i=$((i+1))
  [ "$v" = "-" ] && v=""
  for kv in $v; do export "$kv"; done
  rocprofv3 --kernel-trace --output-format csv -d $O/trace$i -o cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench$i.json 2> $O/bench$i.err
  for kv in $v; do unset "${kv%%=*}"; done
  echo "== variant $i: '$v'"
  python3 - $O/trace$i $O/bench$i.json <<'PY'
import glob, json, sys, pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
t = pd.read_csv(f)
t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
t["name"] = t["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
c = t[t["name"].str.contains("cond_front|cond_back|cond_sigma|cond_apply|cond_g")]
g = c.groupby(["name", "Grid_Size_X", "VGPR_Count"])["dur_us"].agg(["count", "mean", "min", "max"]).round(1)
print(g.to_string())
d = json.load(open(sys.argv[2]))
print("ms/cycle without events", round(d["ms_per_step_without_events"], 2), " finest apply us", round(d["roofline"]["finest_level_avg_launch_us"], 1))
PY
done
