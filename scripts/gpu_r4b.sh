#!/bin/bash
# round 4: look-ahead MFMA inversion A/B (time under the kernel trace, bitwise comparison), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $O
for la in 1 0; do
  ALFI_INVERT_LA=$la python scripts/factor_time.py cfg4s $O/apply_la$la.npy > $O/factor_cfg4s_la$la.txt 2>&1
  tail -n 2 $O/factor_cfg4s_la$la.txt
done
python - <<PY
import numpy as np
a, b = np.load("$O/apply_la1.npy"), np.load("$O/apply_la0.npy")
print("look-ahead vs round-3 kernel: bitwise equal" if np.array_equal(a, b) else "DIFFER: max rel %.3e" % (np.abs(a - b).max() / np.abs(b).max()))
PY
cd /tmp
for la in 1 0; do
  ALFI_INVERT_LA=$la rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_la$la -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4 > $O/trace_la$la.out 2>&1
  f=$(find $O/trace_la$la -name "*kernel_stats.csv" | head -1)
  echo "== LA=$la"; head -8 "$f" | cut -c1-200
  cp "$f" $O/kernel_stats_factor_cfg4_la$la.csv
  rm -rf $O/trace_la$la
done
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -q -m gpu --durations=25 > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -45 $O/pytest.log
