#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_assemble.py tests/test_gpu_newton.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log; tail -12 $O/pytest.log | cut -c1-300
ALFI_SUPG_SCRATCH_MB=1 timeout 1800 python -m pytest tests/test_gpu_assemble.py -x -q -m gpu -k supg > $O/pytest_batches.log 2>&1
echo "pytest exit $?" >> $O/pytest_batches.log; tail -4 $O/pytest_batches.log | cut -c1-300
python scripts/newton_step_time.py cfg4s --supg 0.05 > $O/newton_cfg4s_supg_device.txt 2>&1
python scripts/newton_step_time.py cfg4s --supg 0.05 --host > $O/newton_cfg4s_supg_host.txt 2>&1
python scripts/newton_step_time.py cfg4 --supg 0.05 --re 10 100 1000 > $O/newton_cfg4_supg_device.txt 2>&1
for f in $O/newton_*.txt; do echo "== $f"; grep -v amdgpu $f | tail -n 4; done
