#!/bin/bash
# round 4: where do the cycles of the MFMA inversion kernels go (PMC), run-to-run determinism of both, new tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $O
for la in 1 0; do for rep in a b; do
  ALFI_INVERT_LA=$la python scripts/factor_time.py cfg4s $O/apply_la${la}${rep}.npy > $O/factor_cfg4s_la${la}${rep}.txt 2>&1
done; done
python - <<PY
import numpy as np
L = {k: np.load("$O/apply_%s.npy" % k) for k in ("la1a", "la1b", "la0a", "la0b")}
for a, b in (("la1a", "la1b"), ("la0a", "la0b"), ("la1a", "la0a")):
    d = np.abs(L[a] - L[b])
    print(a, b, "bitwise equal" if np.array_equal(L[a], L[b]) else "DIFFER: %d of %d entries, max rel %.3e" % ((d > 0).sum(), d.size, d.max() / np.abs(L[b]).max()))
PY
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
for la in 1 0; do
  ALFI_INVERT_LA=$la rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1_la$la -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4s > $O/pmc1_la$la.out 2>&1
  ALFI_INVERT_LA=$la rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc2_la$la -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4s > $O/pmc2_la$la.out 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for tag in ("pmc1_la1", "pmc1_la0", "pmc2_la1", "pmc2_la0"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "invert" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    with open("$O/%s_summary.txt" % tag, "w") as out:
        for k, v in agg.items():
            for c, x in sorted(v.items()):
                line = "%s %-50s %-28s %.6e per dispatch (%d dispatches)" % (tag, k, c, x / cnt[(k, c)], cnt[(k, c)])
                print(line); out.write(line + "\n")
PY
tail -3 $O/pmc2_la1.out
rm -rf $O/pmc1_la1 $O/pmc1_la0 $O/pmc2_la1 $O/pmc2_la0
timeout 1500 python -m pytest tests/test_gpu_condensed.py tests/test_frontend.py tests/test_gpu_assemble.py tests/test_gpu_newton.py tests/test_gpu_sv.py -q -m gpu -x --durations=10 -k "not bfs3d" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -25 $O/pytest.log
