#!/bin/bash
# End-of-round artefacts for profiles/ (config 4 and 5 with the final code): headline bench line with cpu_baseline on the full
# hierarchy, rocprofv3 kernel stats of the same command, PMC traffic passes (separate --pmc runs).  Outputs: gpurun_out/r02f/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
python bench.py > $O/r02_bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 bench.py --no-cpu-baseline > $O/r02_bench_cfg4_under_rocprof.json 2> $O/prof_cfg4.err
find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_cfg4_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.json 2> $O/pmc_write.err
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void patch_apply_kernel" $O/pmc_patch_apply_cfg4.json "r02 end of round ($STAMP) patch_apply_kernel, the 60 launches of the first V-cycle" 60 - 3
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write "void bsr_spmv_flat_kernel" $O/pmc_bsr_spmv_cfg4.json "r02 end of round ($STAMP) bsr_spmv_flat_kernel, all launches of the run" 0 -
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 bench.py --config cfg5 --no-cpu-baseline > $O/r02_bench_cfg5_under_rocprof.json 2> $O/prof_cfg5.err
find $O/prof_cfg5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_cfg5_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_fetch.json 2> $O/pmc5_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_write.json 2> $O/pmc5_write.err
python scripts/pmc_summary.py $O/pmc5_fetch $O/pmc5_write "void cond_apply_kernel" $O/pmc_patch_apply_cfg5.json "r02 end of round ($STAMP) cond_apply_kernel (condensed macro-star factors), the 40 launches of the first V-cycle" 40 - 2
rm -rf $O/prof_cfg4 $O/prof_cfg5 $O/pmc_fetch $O/pmc_write $O/pmc5_fetch $O/pmc5_write
ls -la $O
head -c 700 $O/r02_bench_cfg4.json; echo
cat $O/pmc_patch_apply_cfg4.json $O/pmc_patch_apply_cfg5.json
grep -E "patch_apply_kernel|cond_apply_kernel|bsr_spmv_flat" $O/r02_cfg4_kernel_stats.csv $O/r02_cfg5_kernel_stats.csv | head -8
