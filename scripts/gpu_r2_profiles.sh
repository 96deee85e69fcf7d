#!/bin/bash
# Round-2 artefacts for profiles/: bench lines, rocprofv3 kernel stats, PMC traffic passes (separate --pmc runs, as the guide
# prescribes), the forced 1-rank exchange overhead and a functional 4-rank record.  Outputs under gpurun_out/r02/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
# --- config 4: the headline line (with cpu_baseline on the full hierarchy), kernel stats, PMC
python bench.py > $O/r02_bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 bench.py --no-cpu-baseline > $O/r02_bench_cfg4_under_rocprof.json 2> $O/prof_cfg4.err
find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_cfg4_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void patch_apply_kernel" $O/pmc_patch_apply_cfg4.json "r02 ($STAMP) patch_apply_kernel, V-cycle launches" 60
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void bsr_spmv_flat_kernel" $O/pmc_bsr_spmv_cfg4.json "r02 ($STAMP) bsr_spmv_flat_kernel, V-cycle launches" 79
# --- config 5: condensed factors
python bench.py --config cfg5 --no-cpu-baseline > $O/r02_bench_cfg5.json 2> $O/bench_cfg5.err
ALFI_CONDENSE=0 python bench.py --config cfg5 --no-cpu-baseline > $O/r02_bench_cfg5_dense_inverses.json 2> $O/bench_cfg5d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 bench.py --config cfg5 --no-cpu-baseline > $O/r02_bench_cfg5_under_rocprof.json 2> $O/prof_cfg5.err
find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_cfg5_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_fetch.json 2> $O/pmc5_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_write.json 2> $O/pmc5_write.err
python scripts/pmc_summary.py $O/pmc5_fetch $O/pmc5_write "void cond_apply_kernel" $O/pmc_patch_apply_cfg5.json "r02 ($STAMP) cond_apply_kernel (condensed macro-star factors), all launches of the run"
# --- configs 2 and 3: bench lines without events + kernel trace summaries (durations and idle gaps per kernel and grid)
for cfg in cfg2 cfg3; do
  ALFI_BENCH_PROF=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline > $O/r02_bench_$cfg.json 2> $O/bench_$cfg.err
  ( cd /tmp && ALFI_BENCH_PROF=0 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_$cfg -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_$cfg.err )
  python scripts/rocpd_summary.py $(find $O/prof_$cfg -name "*results.db" | head -1) 10 > $O/r02_kernel_trace_$cfg.txt 2>&1
done
# --- exchange points: 1-rank RCCL group, every exchange point forced on (fixed cost of the two transports)
{
python bench.py --config cfg4t --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg4t plain                      ms/cycle %.3f' % d['ms_per_step'])"
for tp in rccl callback; do
ALFI_DIST_TRANSPORT=$tp ALFI_DIST_FORCE=1 ALFI_DIST_MIN_DOFS=1000 ALFI_DIST_OVERLAP=0 python bench.py --config cfg4t --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg4t forced, transport=%-9s ms/cycle %.3f' % ('$tp', d['ms_per_step']))"
done
} > $O/r02_exchange_overhead_1rank.txt 2>&1
# --- functional record of the driver's command shape at config 4: 4 ranks on the box's ONE GPU over gloo (never a perf number)
ALFI_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 4 --steps 2 --warmup 1 > $O/r02_bench_dist4_gloo_sharedgpu_functional.json 2> $O/dist4.err
echo "dist4 exit $?"
rm -rf $O/prof_cfg4 $O/prof_cfg5 $O/pmc_fetch $O/pmc_write $O/pmc5_fetch $O/pmc5_write $O/prof_cfg2 $O/prof_cfg3
ls -la $O
head -c 600 $O/r02_bench_cfg4.json; echo; cat $O/r02_exchange_overhead_1rank.txt; head -3 $O/r02_kernel_trace_cfg2.txt
