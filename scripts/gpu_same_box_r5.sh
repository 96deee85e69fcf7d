#!/bin/bash
# Round 5 against the end of round 4 on ONE box: the current tree and a checkout of e10525d under .r03tree/ (git-ignored; `git archive
# e10525d | tar -x -C .r03tree` + its own build), alternating.  V-cycles of configs 4, 5, 2 (events off), then the Newton solver at
# config-4 size (set-up, seconds per Newton step, pieces) and the bench's host generation.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/same_box
mkdir -p $O
{
for C in cfg4 cfg5 cfg2; do
for i in 1 2; do
  for T in r05 r04; do
    D=$GRAFT_REPO_ROOT; [ $T = r04 ] && D=$GRAFT_REPO_ROOT/.r03tree
    (cd $D && ALFI_BENCH_PROF=0 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --config $C 2> /dev/null) | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$C $T: %.3f ms per V-cycle, %.3f V-cycles/s, rel. residual %.6e, host generation %.1f s' % (d['ms_per_step'], d['value'], d.get('rel_residual_after_timed_cycles', float('nan')), d['setup_s']['host_generation']))"
  done
done
done
for T in r05 r04; do
  D=$GRAFT_REPO_ROOT; [ $T = r04 ] && D=$GRAFT_REPO_ROOT/.r03tree
  echo "== $T: scripts/newton_step_time.py cfg4 --re 10 100 1000"
  (cd $D && python scripts/newton_step_time.py cfg4 --re 10 100 1000 2>&1 | grep -v amdgpu | tail -n 4)
  echo "== $T: scripts/newton_step_time.py cfg4 --re 10 100 1000 --supg 0.05"
  (cd $D && python scripts/newton_step_time.py cfg4 --re 10 100 1000 --supg 0.05 2>&1 | grep -v amdgpu | tail -n 4)
done
} | tee $O/same_box_r04_vs_r05.txt
