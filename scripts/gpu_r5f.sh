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
  # the operator refresh / residual kernels of a Newton step at config-4 size: device times, kernel stats, HBM traffic.  Two
  # passes: without SUPG (the element blocks in the interleaved layout) and with it (cell-contiguous layout, shared with the SUPG kernel)
  timeout 900 python scripts/refresh_time.py cfg4 > $O/r05_refresh_time_cfg4.txt 2> $O/refresh.err
  timeout 900 python scripts/refresh_time.py cfg4 --supg 0.05 >> $O/r05_refresh_time_cfg4.txt 2>> $O/refresh.err
  grep -v amdgpu $O/r05_refresh_time_cfg4.txt
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refresh -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 --supg 0.05 --reps 2 > $O/prof_refresh.out 2> $O/prof_refresh.err
  for V in plain supg; do
    if [ $V = supg ]; then A="--supg 0.05"; else A=""; fi
    timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_refresh_fetch_$V -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 $A --reps 1 > $O/pmc_refresh_fetch.out 2> $O/pmc_refresh_fetch.err
    timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_refresh_write_$V -- python3 $GRAFT_REPO_ROOT/scripts/refresh_time.py cfg4 $A --reps 1 > $O/pmc_refresh_write.out 2> $O/pmc_refresh_write.err
  done
  cd $GRAFT_REPO_ROOT
  find $O/prof_refresh -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_refresh_cfg4_kernel_stats.csv
  python - $O/prof_refresh <<'PY' > $O/r05_refresh_cfg4_by_level.txt
import glob, sys, pandas as pd
t = pd.read_csv(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])
t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
t["name"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
c = t[t["name"].str.contains("element_cell|element_gather|supg_matrix|supg_residual|cell_vector_gather|bc_code")]
print("config 4, operator refresh / residual kernels by grid size (the largest grid of each kernel = the finest level), durations in us")
print("(scripts/refresh_time.py cfg4 --supg 0.05: the launches of the plain refresh -- interleaved element blocks -- and of the stabilised one mix in the element / gather rows)")
print(c.groupby(["name", "Grid_Size_X", "VGPR_Count"])["dur_us"].agg(["count", "mean", "min", "max"]).round(1).to_string())
PY
  cat $O/r05_refresh_cfg4_by_level.txt
  for K in "void (anonymous namespace)::element_cell_kernel<3, 14, 0>" "void (anonymous namespace)::element_gather_kernel<3>" "void (anonymous namespace)::element_cell_kernel<3, 14, 1>"; do
    N=$(echo "$K" | sed 's/.*:://; s/[<>, ]/_/g; s/__*/_/g; s/_$//')
    python scripts/pmc_summary.py $O/pmc_refresh_fetch_plain $O/pmc_refresh_write_plain "$K" $O/r05_pmc_${N}_cfg4.json "r05 ($STAMP) $K on config 4's finest level (largest grid), scripts/refresh_time.py cfg4 (no SUPG: interleaved element blocks)" 0 max
  done
  for K in "void (anonymous namespace)::supg_matrix_kernel<3, 14>" "void (anonymous namespace)::supg_residual_cell_kernel<3, 14>"; do
    N=$(echo "$K" | sed 's/.*:://; s/[<>, ]/_/g; s/__*/_/g; s/_$//')
    python scripts/pmc_summary.py $O/pmc_refresh_fetch_supg $O/pmc_refresh_write_supg "$K" $O/r05_pmc_${N}_cfg4.json "r05 ($STAMP) $K on config 4's finest level (largest grid), scripts/refresh_time.py cfg4 --supg 0.05 (cell-contiguous element blocks)" 0 max
  done
  rm -rf $O/prof_refresh $O/pmc_refresh_fetch_* $O/pmc_refresh_write_*
  ls $O
fi
if [ "$PART" = headline ]; then
  python bench.py --steps 20 --warmup 5 > $O/r05_bench_cfg4.json 2> $O/bench_cfg4.err
  python bench.py > $O/r05_bench_cfg4_default_flags.json 2> $O/bench_cfg4_default.err
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r05_bench_cfg4_under_rocprof.json 2> $O/prof_cfg4.err
  cd $GRAFT_REPO_ROOT
  find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_cfg4_kernel_stats.csv
  python - $O/prof_cfg4 $O/r05_bench_cfg4_under_rocprof.json <<'PY' > $O/r05_cfg4_patch_apply_by_level.txt
import glob, json, sys, pandas as pd
t = pd.read_csv(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])
t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
a = t[t["Kernel_Name"].str.contains("patch_apply_kernel")]
print("rocprofv3 --kernel-trace of `bench.py --no-cpu-baseline --steps 20 --warmup 5`: patch_apply_kernel launches by grid size (largest = finest level)")
print(a.groupby("Grid_Size_X")["dur_us"].agg(["count", "mean", "min", "max"]).round(1).to_string())
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]
print("the same run's bench line (HIP events on the library's stream, timed cycles only): finest level %.1f us per launch, all smoothed levels %.1f us over %d launches"
      % (r["finest_level_avg_launch_us"], r["avg_launch_us"], r["launches"]))
PY
  cat $O/r05_cfg4_patch_apply_by_level.txt
  rm -rf $O/prof_cfg4
  head -c 700 $O/r05_bench_cfg4.json; echo
  head -c 400 $O/r05_bench_cfg4_default_flags.json; echo
fi
if [ "$PART" = pmc ]; then
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void patch_apply_kernel" $O/pmc_patch_apply_cfg4.json "r05 ($STAMP) patch_apply_kernel, the 60 launches of the first V-cycle" 60 - 3
  python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void bsr_spmv_dedup_kernel" $O/pmc_bsr_spmv_cfg4.json "r05 ($STAMP) bsr_spmv_dedup_kernel, the launches on the finest level (largest grid) of the run" 0 max
  rm -rf $O/pmc_fetch $O/pmc_write
fi
if [ "$PART" = configs ]; then
  for C in cfg2 cfg3 cfg6; do
    python bench.py --config $C --steps 20 --warmup 5 > $O/r05_bench_$C.json 2> $O/bench_$C.err
  done
  python bench.py --config cfg5 --steps 10 --warmup 3 > $O/r05_bench_cfg5.json 2> $O/bench_cfg5.err
  python bench.py --config cfg5L --no-cpu-baseline --steps 10 --warmup 3 > $O/r05_bench_cfg5L.json 2> $O/bench_cfg5L.err
  for C in cfg2 cfg3 cfg5 cfg5L cfg6; do head -c 300 $O/r05_bench_$C.json; echo; done
fi
if [ "$PART" = suite ]; then
  timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/r05_pytest_gpu.txt 2>&1
  echo "pytest exit $?" >> $O/r05_pytest_gpu.txt
  tail -n 32 $O/r05_pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.txt 2>&1; tail -n 2 $O/r05_smoke.txt
fi
if [ "$PART" = newton ]; then
  python scripts/newton_step_time.py cfg4 --re 10 100 1000 > $O/r05_newton_cfg4_device_assembly.txt 2>&1
  grep -v amdgpu.ids $O/r05_newton_cfg4_device_assembly.txt | tail -n 4
  python scripts/newton_step_time.py cfg4 --re 10 100 1000 --supg 0.05 > $O/r05_newton_cfg4_supg_device_assembly.txt 2>&1
  grep -v amdgpu.ids $O/r05_newton_cfg4_supg_device_assembly.txt | tail -n 4
  timeout 1500 python scripts/dist_newton_time.py cfg4 --ranks 4 --re 10 100 1000 > $O/r05_dist_newton_cfg4_4ranks_mock_sharedgpu_functional.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $O/r05_dist_newton_cfg4_4ranks_mock_sharedgpu_functional.txt | tail -n 6
  python scripts/mult_time.py cfg4 > $O/r05_mult_cfg4.txt 2>&1; tail -n 2 $O/r05_mult_cfg4.txt
  python scripts/mult_time.py cfg5 > $O/r05_mult_cfg5_macro_stars.txt 2>&1; tail -n 2 $O/r05_mult_cfg5_macro_stars.txt
  ALFI_MULT_PERSISTENT=0 python scripts/mult_time.py cfg5 > $O/r05_mult_cfg5_macro_stars_per_wavefront.txt 2>&1; tail -n 2 $O/r05_mult_cfg5_macro_stars_per_wavefront.txt
fi
if [ "$PART" = setup ]; then
  python scripts/generation_profile.py cfg4 > $O/r05_generation_profile_cfg4.txt 2>&1; grep "^total" $O/r05_generation_profile_cfg4.txt
  python scripts/setup_profile.py cfg4 --lines 60 > $O/r05_setup_profile_cfg4.txt 2>&1; grep "set-up" $O/r05_setup_profile_cfg4.txt
  python scripts/setup_profile.py cfg4 --supg 0.05 --lines 30 > $O/r05_setup_profile_cfg4_supg.txt 2>&1; grep "set-up" $O/r05_setup_profile_cfg4_supg.txt
fi
ls $O | tail -40
