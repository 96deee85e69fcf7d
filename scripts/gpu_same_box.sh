#!/bin/bash
# same-box comparison of two trees: the current one and another checkout placed under .r03tree/ (git-ignored; e.g. the end of
# round 3: `git archive 2c8e274 | tar -x -C .r03tree` + its own build).  bench.py of each tree, alternating, events off.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/same_box
mkdir -p $O
for C in cfg4 cfg3 cfg2; do
for i in 1 2; do
  for T in new old; do
    D=$GRAFT_REPO_ROOT; [ $T = old ] && D=$GRAFT_REPO_ROOT/.r03tree
    (cd $D && ALFI_BENCH_PROF=0 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --config $C 2> /dev/null) | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$C $T: %.3f ms per V-cycle, %.3f V-cycles/s, rel. residual %.6e' % (d['ms_per_step'], d['value'], d.get('rel_residual_after_timed_cycles', float('nan'))))"
  done
done
done | tee $O/same_box.txt
