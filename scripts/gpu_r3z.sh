#!/bin/bash
# end of round 3, final code: whole GPU suite, smoke, then the artefacts of gpu_r3_final.sh (headline, pmc, configs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03f
timeout 1300 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03f/r03_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r03f/r03_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
PART=headline bash scripts/gpu_r3_final.sh
PART=pmc bash scripts/gpu_r3_final.sh
PART=configs bash scripts/gpu_r3_final.sh
