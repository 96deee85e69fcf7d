#!/bin/bash
# End-of-round artefacts for profiles/ (round 5).  Outputs: gpurun_out/r05f/.  PART: refresh | headline | pmc | configs | suite.
# ALFI_COMMIT (the commit the snapshot was taken from) goes into the PMC summaries.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
PART=${PART:-refresh}
if [ "$PART" = refresh ]; then
  # the operator refresh / residual kernels of a Newton step at config-4 size: device times, kernel stats, HBM traffic
  timeout 900 python scripts/refresh_time.py cfg4 --supg 0.05 > $O/r05_refresh_time_cfg4.txt 2> $O/refresh.err
  cat $O/r05_refresh_time_cfg4.txt
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refresh -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 --supg 0.05 --reps 2 > $O/prof_refresh.out 2> $O/prof_refresh.err
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_refresh_fetch -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 --supg 0.05 --reps 1 > $O/pmc_refresh_fetch.out 2> $O/pmc_refresh_fetch.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_refresh_write -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 --supg 0.05 --reps 1 > $O/pmc_refresh_write.out 2> $O/pmc_refresh_write.err
  cd $GRAFT_REPO_ROOT
  find $O/prof_refresh -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_refresh_cfg4_kernel_stats.csv
  python - $O/prof_refresh <<'PY' > $O/r05_refresh_cfg4_by_level.txt
import glob, sys, pandas as pd
t = pd.read_csv(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])
t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
t["name"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
c = t[t["name"].str.contains("element_cell|element_gather|supg_matrix|supg_residual|cell_vector_gather|bc_code")]
print("config 4, operator refresh / residual kernels by grid size (the largest grid of each kernel = the finest level), durations in us")
print(c.groupby(["name", "Grid_Size_X", "VGPR_Count"])["dur_us"].agg(["count", "mean", "min", "max"]).round(1).to_string())
PY
  cat $O/r05_refresh_cfg4_by_level.txt
  for K in "void (anonymous namespace)::element_cell_kernel<3, 14, 0>" "void (anonymous namespace)::element_gather_kernel<3>" "void (anonymous namespace)::supg_matrix_kernel<3, 14>" "void (anonymous namespace)::element_cell_kernel<3, 14, 1>" "void (anonymous namespace)::supg_residual_cell_kernel<3, 14>"; do
    N=$(echo "$K" | sed 's/.*:://; s/[<>, ]/_/g; s/__*/_/g; s/_$//')
    python scripts/pmc_summary.py $O/pmc_refresh_fetch $O/pmc_refresh_write "$K" $O/r05_pmc_${N}_cfg4.json "r05 ($STAMP) $K on config 4's finest level (largest grid), scripts/refresh_time.py cfg4 --supg 0.05" 0 max
  done
  rm -rf $O/prof_refresh $O/pmc_refresh_fetch $O/pmc_refresh_write
  ls $O
fi
