#!/bin/bash
# End-of-round artefacts for profiles/ (round 4).  Outputs: gpurun_out/r04f/.  PART selects what to run: headline | pmc | configs |
# cfg5 | cfg5L | dist | setup | suite.  setup needs the timing builds libalfi_hip_invtiming.so / libalfi_hip_mtiming.so
# (-DALFI_INVERT_TIMING / -DALFI_MULT_TIMING).  ALFI_COMMIT (the commit the snapshot was taken from: the GPU box has no .git) goes into the
# PMC summaries.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
PART=${PART:-headline}
if [ "$PART" = headline ]; then
  python bench.py --steps 20 --warmup 5 > $O/r04_bench_cfg4.json 2> $O/bench_cfg4.err
  python bench.py --steps 20 --warmup 5 --restriction --no-cpu-baseline > $O/r04_bench_cfg4_restriction.json 2> $O/bench_cfg4_restriction.err
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r04_bench_cfg4_under_rocprof.json 2> $O/prof_cfg4.err
  cd $GRAFT_REPO_ROOT
  find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_cfg4_kernel_stats.csv
  # the dominant kernel by grid size (= by level): the kernel-stats average mixes the levels in the proportions of the WHOLE run
  # (warm-up, probe, F-cycle, the cycles of the other restriction setting), the bench line's events cover the timed cycles only
  python - $O/prof_cfg4 $O/r04_bench_cfg4_under_rocprof.json <<'PY' > $O/r04_cfg4_patch_apply_by_level.txt
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
  cat $O/r04_cfg4_patch_apply_by_level.txt
  rm -rf $O/prof_cfg4
  head -c 600 $O/r04_bench_cfg4.json; echo
  head -8 $O/r04_cfg4_kernel_stats.csv
fi
if [ "$PART" = pmc ]; then
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void patch_apply_kernel" $O/pmc_patch_apply_cfg4.json "r04 end of round ($STAMP) patch_apply_kernel, the 60 launches of the first V-cycle" 60 - 3
  python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void bsr_spmv_dedup_kernel" $O/pmc_bsr_spmv_cfg4.json "r04 end of round ($STAMP) bsr_spmv_dedup_kernel, the launches on the finest level (largest grid) of the run" 0 max
  rm -rf $O/pmc_fetch $O/pmc_write
fi
if [ "$PART" = configs ]; then
  for C in cfg2 cfg3 cfg6; do
    python bench.py --config $C --steps 20 --warmup 5 > $O/r04_bench_$C.json 2> $O/bench_$C.err
  done
  python bench.py --config cfg5 --steps 10 --warmup 3 > $O/r04_bench_cfg5.json 2> $O/bench_cfg5.err
  for CFG in cfg2 cfg3; do
    cd /tmp
    rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace_$CFG -o run -- python3 $GRAFT_REPO_ROOT/scripts/trace_cycle.py --config $CFG --cycles 6 > $O/trace_$CFG.out 2> $O/trace_$CFG.err
    cd $GRAFT_REPO_ROOT
    DB=$(find $O/trace_$CFG -name "*.db" | head -1)
    python scripts/timeline_summary.py $DB 4 > $O/r04_timeline_$CFG.txt 2>&1
    rm -rf $O/trace_$CFG
  done
  for C in cfg2 cfg3 cfg5 cfg6; do head -c 300 $O/r04_bench_$C.json; echo; done
fi
if [ "$PART" = cfg5 ]; then
  # condensed apply (three launches): the bench line, a kernel trace, HBM traffic
  python bench.py --config cfg5 --steps 10 --warmup 3 > $O/r04_bench_cfg5.json 2> $O/bench_cfg5.err
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 10 --warmup 3 > $O/r04_bench_cfg5_under_rocprof.json 2> $O/prof_cfg5.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_fetch.json 2> $O/pmc5_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_write.json 2> $O/pmc5_write.err
  cd $GRAFT_REPO_ROOT
  find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_cfg5_kernel_stats.csv
  python - $O/prof_cfg5 <<'PY' > $O/r04_cond_apply_trace_cfg5.txt
import glob, sys, pandas as pd
t = pd.read_csv(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])
t["dur_us"] = (t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3
t["name"] = t["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
c = t[t["name"].str.contains("cond_front|cond_back|cond_sigma|cond_apply|cond_gfront|cond_gback|cond_gsigma")]
print("config 5, condensed apply: launches by kernel and grid (the larger grid of each kernel = the finest level), durations in us")
print(c.groupby(["name", "Grid_Size_X", "VGPR_Count"])["dur_us"].agg(["count", "mean", "min", "max"]).round(1).to_string())
PY
  cat $O/r04_cond_apply_trace_cfg5.txt
  python scripts/pmc_summary.py $O/pmc5_fetch $O/pmc5_write "void cond_front_kernel+void cond_sigma_kernel+void cond_back_kernel|void cond_gfront_kernel+void cond_gsigma_kernel+void cond_gback_kernel" $O/pmc_patch_apply_cfg5.json "r04 end of round ($STAMP) one apply of the condensed macro-star factors = three launches, counters of the three added: cond_front + cond_sigma + cond_back on the finest level, cond_gfront + cond_gsigma + cond_gback on level 1; all applies of the run"
  rm -rf $O/prof_cfg5 $O/pmc5_fetch $O/pmc5_write
  head -c 400 $O/r04_bench_cfg5.json; echo
fi
if [ "$PART" = cfg5L ]; then
  python bench.py --config cfg5L --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_bench_cfg5L.json 2> $O/bench_cfg5L.err
  head -c 400 $O/r04_bench_cfg5L.json; echo
  ALFI_TEST_CFG5L=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg5L" > $O/r04_pytest_cfg5L.txt 2>&1
  tail -3 $O/r04_pytest_cfg5L.txt
fi
if [ "$PART" = dist ]; then
  MOCK=$(python -c "from tests.mock_rccl.build import build; print(build())")
  for N in 4 6; do
    ALFI_DIST_BACKEND=gloo ALFI_DIST_TRANSPORT=rccl ALFI_RCCL_LIB=$MOCK ALFI_BENCH_TIMEOUT_S=1500 python bench.py --gpus $N --steps 2 --warmup 1 > $O/r04_bench_dist${N}_native_transport_mock_sharedgpu_functional.json 2> $O/bench_dist$N.err
    tail -3 $O/bench_dist$N.err
    head -c 500 $O/r04_bench_dist${N}_native_transport_mock_sharedgpu_functional.json; echo
  done
fi
if [ "$PART" = suite ]; then
  timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/r04_pytest_gpu.txt 2>&1
  echo "pytest exit $?" >> $O/r04_pytest_gpu.txt
  tail -n 32 $O/r04_pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.txt 2>&1; tail -n 2 $O/r04_smoke.txt
fi
if [ "$PART" = setup ]; then
  python scripts/factor_time.py cfg4 > $O/factor_cfg4_mfma.txt 2>&1
  ALFI_INVERT_MFMA=0 python scripts/factor_time.py cfg4 > $O/factor_cfg4_reg.txt 2>&1
  tail -n 1 $O/factor_cfg4_mfma.txt $O/factor_cfg4_reg.txt
  # the inversion kernel on config 4's finest level: duration (kernel trace) and MFMA pipe busy (its own counter pass)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_factor -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4 > $O/prof_factor.out 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_factor -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4 > $O/pmc_factor.out 2>&1
  cd $GRAFT_REPO_ROOT
  {
    echo "# round 4 end of round ($STAMP), commit $ALFI_COMMIT: scripts/factor_time.py cfg4 (185 193 patches of <= 153 dofs, 4 factorisations)"
    echo "# kernel durations (rocprofv3 --kernel-trace --stats), then MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"
    find $O/prof_factor -name "*kernel_stats.csv" | head -1 | xargs -I{} head -6 {}
    python scripts/mfma_busy_summary.py $O/pmc_factor $O/r04_mfma_busy_patch_invert_cfg4.json "round 4 end of round ($STAMP), commit $ALFI_COMMIT: factor_time.py cfg4"
  } > $O/r04_mfma_busy_patch_invert_cfg4.txt 2>&1
  cat $O/r04_mfma_busy_patch_invert_cfg4.txt
  rm -rf $O/prof_factor $O/pmc_factor
  ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_invtiming.so python scripts/invert_phases.py cfg4s > $O/r04_invert_phases.txt 2>&1; grep -v amdgpu.ids $O/r04_invert_phases.txt | tail -n 18
  python scripts/newton_step_time.py cfg4 --re 10 100 1000 > $O/r04_newton_cfg4_device_assembly.txt 2>&1
  tail -n 4 $O/r04_newton_cfg4_device_assembly.txt
  timeout 1500 python scripts/dist_newton_time.py cfg4 --ranks 4 --re 10 100 1000 > $O/r04_dist_newton_cfg4_4ranks_mock_sharedgpu_functional.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $O/r04_dist_newton_cfg4_4ranks_mock_sharedgpu_functional.txt | tail -n 6
  python scripts/mult_time.py cfg4 > $O/r04_mult_cfg4.txt 2>&1; tail -n 2 $O/r04_mult_cfg4.txt
  ALFI_MULT_PERSISTENT=0 python scripts/mult_time.py cfg4 > $O/r04_mult_cfg4_per_wavefront.txt 2>&1; tail -n 2 $O/r04_mult_cfg4_per_wavefront.txt
  ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_mtiming.so python scripts/mult_stamps.py cfg4 > $O/r04_mult_stamps_cfg4.txt 2>&1; grep -v amdgpu.ids $O/r04_mult_stamps_cfg4.txt | tail -n 12
  python scripts/apply_time.py cfg4 > $O/r04_apply_time_cfg4.txt 2>&1; grep level $O/r04_apply_time_cfg4.txt
  python scripts/apply_time.py cfg3 > $O/r04_apply_time_cfg3.txt 2>&1; grep level $O/r04_apply_time_cfg3.txt
fi
ls -la $O | tail -30
