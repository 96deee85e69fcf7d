#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
python scripts/factor_time.py cfg4s > $O/factor_cfg4s_mfma.txt 2>&1
ALFI_INVERT_MFMA=0 python scripts/factor_time.py cfg4s > $O/factor_cfg4s_reg.txt 2>&1
for f in $O/factor_*.txt; do echo $f; tail -n 1 $f; done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_patch_check.py tests/test_gpu_env_variants.py tests/test_gpu_dist.py tests/test_gpu_assemble.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
# MFMA-busy of the inversion kernel (separate --pmc pass, kernel-trace only)
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_invert -- python3 $GRAFT_REPO_ROOT/scripts/factor_time.py cfg4s > $O/pmc_invert.out 2> $O/pmc_invert.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_invert/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k] += 1
with open("$O/mfma_busy_invert.txt", "w") as out:
    for k, v in agg.items():
        if "invert" in k or "gather_dense" in k or "check" in k:
            line = "%-72s %s" % (k, {c: x for c, x in v.items()})
            print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_invert
python scripts/newton_step_time.py cfg4s > $O/newton_cfg4s_device.txt 2>&1
python scripts/newton_step_time.py cfg4s --host > $O/newton_cfg4s_host.txt 2>&1
python scripts/newton_step_time.py cfg4 --re 10 100 1000 > $O/newton_cfg4_device.txt 2>&1
tail -4 $O/newton_*.txt
