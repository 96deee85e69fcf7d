#!/bin/bash
# MFMA-busy counters of the setup kernels (separate --pmc passes, --kernel-trace only): config 5 (condensed factors: group
# inverses, Schur complements, blocked Gauss-Jordan + Newton-Schulz products), config 6 (multifrontal coarse factorisation),
# config 4 (patch inversion on the matrix cores, coarse factorisation)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
for C in cfg5 cfg6 cfg4; do
  cd /tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_$C.json 2> $O/pmc_$C.err
  cd $GRAFT_REPO_ROOT
  python scripts/mfma_busy_summary.py $O/pmc_$C $O/r03_mfma_busy_setup_$C.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --config $C --no-cpu-baseline --steps 1 --warmup 0 (round 3)"
  rm -rf $O/pmc_$C
done
