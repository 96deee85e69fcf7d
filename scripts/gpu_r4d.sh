#!/bin/bash
# round 4: look-ahead inversion variant F (bitwise check with a deterministic host assembly), multiplicative sweep with more
# bytes in flight, partitioned Newton at config-4 size over 4 mock ranks, the trimmed GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $O
for la in 1 0; do
  OMP_NUM_THREADS=1 ALFI_HOST_THREADS=1 ALFI_INVERT_LA=$la python scripts/factor_time.py cfg4s $O/apply_la$la.npy > $O/factor_cfg4s_la$la.txt 2>&1
  tail -n 2 $O/factor_cfg4s_la$la.txt
done
python - <<PY
import numpy as np
a, b = np.load("$O/apply_la1.npy"), np.load("$O/apply_la0.npy")
d = np.abs(a - b)
print("look-ahead vs round-3 kernel (single-threaded host assembly):", "bitwise equal" if np.array_equal(a, b) else "DIFFER: %d entries, max rel %.3e" % ((d > 0).sum(), d.max() / np.abs(b).max()))
PY
cd /tmp
for la in 1 0; do
  ALFI_INVERT_LA=$la rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_la$la -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4s > $O/trace_la$la.out 2>&1
  f=$(find $O/trace_la$la -name "*kernel_stats.csv" | head -1)
  echo "== LA=$la"; grep -i "invert" "$f" | cut -c1-160
  rm -rf $O/trace_la$la
done
cd $GRAFT_REPO_ROOT
python scripts/mult_time.py cfg4s $O/mult_cfg4s.npy > $O/mult_cfg4s.txt 2>&1; tail -n 2 $O/mult_cfg4s.txt
python scripts/mult_time.py cfg4 > $O/mult_cfg4.txt 2>&1; tail -n 2 $O/mult_cfg4.txt
python scripts/mult_time.py cfg2 > $O/mult_cfg2.txt 2>&1; tail -n 2 $O/mult_cfg2.txt
timeout 1500 python scripts/dist_newton_time.py cfg4 --ranks 4 --re 10 100 > $O/dist_newton_cfg4_4ranks.txt 2>&1
tail -n 8 $O/dist_newton_cfg4_4ranks.txt
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -30 $O/pytest.log
