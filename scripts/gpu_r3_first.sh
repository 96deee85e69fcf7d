#!/bin/bash
# round 3, first GPU pass: new dist / watchdog tests, a cfg2 + cfg3 timeline (ONE-cycle attribution), hipGraph replay re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -k "watchdog or overlap_threshold or native_transport or bench" > $O/pytest_dist.log 2>&1
echo "pytest exit $?" >> $O/pytest_dist.log
for CFG in cfg2 cfg3; do
  cd /tmp
  rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace_$CFG -o run -- python3 $GRAFT_REPO_ROOT/scripts/trace_cycle.py --config $CFG --cycles 6 > $O/trace_$CFG.out 2> $O/trace_$CFG.err
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/trace_$CFG -name "*.db" | head -1)
  python scripts/timeline_summary.py $DB 4 > $O/timeline_$CFG.txt 2>&1
  rm -rf $O/trace_$CFG
  ALFI_BENCH_PROF=0 python bench.py --config $CFG --steps 20 --warmup 3 --graph --no-cpu-baseline > $O/bench_${CFG}_graph.json 2> $O/bench_${CFG}_graph.err
done
python bench.py --config cfg4 --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -5 $O/pytest_dist.log
head -30 $O/timeline_cfg2.txt
cat $O/trace_cfg2.out
