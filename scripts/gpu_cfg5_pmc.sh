#!/bin/bash
# HBM traffic counters of the config-5 patch apply (big_apply_kernel): two separate --pmc passes as the guide prescribes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc5_fetch gpurun_out/pmc5_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc5_fetch -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc5_fetch.json 2> gpurun_out/pmc5_fetch.err
echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc5_write -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc5_write.json 2> gpurun_out/pmc5_write.err
echo "write exit $?"
python scripts/pmc_summary.py gpurun_out/pmc5_fetch gpurun_out/pmc5_write "void big_apply_kernel" gpurun_out/pmc_patch_apply_cfg5.json "big_apply_kernel, the smoother launches of the V-cycle" 40 903680,310272
