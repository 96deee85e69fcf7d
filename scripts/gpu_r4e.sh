#!/bin/bash
# round 4: multiplicative sweep A/B over the loads-in-flight parameters (alt builds of kernels_patch.hip), additive apply with 16
# loads in flight on config 3, single-GPU Newton counts for the comparison with the 4-rank run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $O
for v in u8r1 u32r1 u8r4 u16r2; do
  OMP_NUM_THREADS=1 ALFI_HOST_THREADS=1 ALFI_MULT_PERSISTENT=0 ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so python scripts/mult_time.py cfg4s $O/mult_cfg4s_$v.npy > $O/mult_cfg4s_$v.txt 2>&1
  echo "$v: $(tail -n 2 $O/mult_cfg4s_$v.txt | head -1)"
done
python - <<PY
import numpy as np
ref = np.load("$O/mult_cfg4s_u8r1.npy")
for v in ("u32r1", "u8r4", "u16r2"):
    a = np.load("$O/mult_cfg4s_%s.npy" % v)
    print(v, "bitwise equal to u8r1" if np.array_equal(a, ref) else "DIFFERS %.3e" % (np.abs(a - ref).max() / np.abs(ref).max()))
PY
for v in u8r1 u32r1 u8r4; do
  for pm in 1 0; do
    ALFI_MULT_PERSISTENT=$pm ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so python scripts/mult_time.py cfg4 $O/mult_cfg4_${v}_p$pm.npy > $O/mult_cfg4_${v}_p$pm.txt 2>&1
    echo "$v persistent=$pm: $(tail -n 2 $O/mult_cfg4_${v}_p$pm.txt | head -1)"
  done
done
grep -h checksum $O/mult_cfg4_*_p*.txt
rm -f $O/mult_cfg4_*.npy
timeout 900 python -m pytest tests/test_gpu_env_variants.py tests/test_frontend.py -q -m gpu -x -k "persistent or multiplicative" > $O/pytest_mult.log 2>&1; tail -n 5 $O/pytest_mult.log
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3 --config cfg3"
ALFI_BENCH_PROF=0 $B > $O/cfg3_u8.json 2> $O/cfg3_u8.err
ALFI_BENCH_PROF=0 ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_au16.so $B > $O/cfg3_u16.json 2> $O/cfg3_u16.err
ALFI_BENCH_PROF=0 $B > $O/cfg3_u8b.json 2> $O/cfg3_u8b.err
ALFI_BENCH_PROF=0 ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_au16.so $B > $O/cfg3_u16b.json 2> $O/cfg3_u16b.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/cfg3_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["ms_per_step"], d.get("rel_residual_after_timed_cycles"))
    except Exception as e:
        print(f, "FAILED", e)
PY
python scripts/newton_step_time.py cfg4 --re 10 100 > $O/newton_cfg4_1gpu.txt 2>&1; tail -n 3 $O/newton_cfg4_1gpu.txt
