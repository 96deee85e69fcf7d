"""The few switches the library reads from the environment (once per process) -- the test hook that sends small hierarchies
through the kernels of the large levels, the choice of the inversion kernel, the scratch budget of the blocked factorisation,
the tolerance of the residual probe -- must all give the cycle.  -m gpu; one subprocess per setting.  (Round 3 had 31 settings
here: the measured losers they selected were removed from the library in round 4.)"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env,tol", [({}, 1e-5),
                                     ({"ALFI_BIG_SCRATCH_MB": "64"}, 1e-5),       # many small factorisation batches
                                     # the size thresholds of the large levels lowered to zero: chunked condensed apply
                                     # on every level, nnz-balanced SpMV + fix-up launch, de-duplicated x gathers
                                     ({"ALFI_TEST_LARGE_PATHS": "1"}, 1e-5),
                                     # EVERY condensed patch flagged by the residual probe: group inverses and Schur
                                     # complements formed again by pivoted LU, in place (kernels_check.hip: cond_repair)
                                     ({"ALFI_PATCH_CHECK_TOL": "1e-14", "ALFI_PATCH_CHECK_FAIL": "1e-6"}, 1e-5)])
def test_switch(env, tol):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_env_variant_worker.py")],
                         env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=600)
    m = re.search(r"RELERR (\S+)", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-2000:] + out.stderr[-3000:]
    assert float(m.group(1)) < tol, (env, m.group(1))


@pytest.mark.parametrize("env", [{},                                                  # defaults: four-launch fused iteration on the small levels
                                 # the kernels of the large levels on these small ones: general launch chain, nnz-balanced
                                 # SpMV with the fix-up launch and the de-duplicated x gathers (strips of 64 per XCD)
                                 {"ALFI_TEST_LARGE_PATHS": "1"},
                                 {"ALFI_INVERT_MFMA": "0"}, {"ALFI_INVERT_MFMA": "2"},   # register / matrix-core inversion for every size
                                 # EVERY patch flagged by the residual probe and re-inverted by the pivoted LU repair
                                 # (kernels_check.hip), 14-dof and 153-dof patches alike
                                 {"ALFI_PATCH_CHECK_TOL": "1e-14", "ALFI_PATCH_CHECK_FAIL": "1e-6"}])
def test_smoother_paths(env):
    """Every implementation of the level smoother (alfi/solver.py:313-328) and of the level product behind it gives the
    oracle's FGMRES iterate (1e-7) and cycles (1e-5)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_env_pkp0_worker.py")],
                         env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=900)
    ms, mc = re.search(r"SMOOTH (\S+)", out.stdout), re.search(r"CYCLE (\S+)", out.stdout)
    assert out.returncode == 0 and ms and mc, out.stdout[-2000:] + out.stderr[-3000:]
    assert float(ms.group(1)) < 1e-7 and float(mc.group(1)) < 1e-5, (env, ms.group(1), mc.group(1))


def test_persistent_sweep_equals_the_launch_per_wavefront_schedule(tmp_path):
    """the multiplicative sweep as ONE launch of a resident grid that walks the wavefront schedule with
    per-item dependency counters (patch_mult_persistent_kernel; alfi/solver.py:322-335, relaxation.py:139-150) gives, bit for
    bit, what one launch per dependency wavefront gives (ALFI_MULT_PERSISTENT=0: the schedule of rounds 1-3) -- 2-D with two
    '|'-separated sort orders (every patch twice per sweep), 3-D, and the macro stars of the 3-D Scott-Vogelius pair (more than
    64 nodes per patch: big_mult_persistent_kernel, round 5), symmetrised, each apply repeated three times."""
    import numpy as np
    res = {}
    for mode in ("1", "0"):
        f = str(tmp_path / ("sweep%s.npz" % mode))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_mult_schedule_worker.py"), f],
                             # (one host thread: the generator's parallel assembly sums in a run-dependent order)
                             env=dict(os.environ, ALFI_MULT_PERSISTENT=mode, OMP_NUM_THREADS="1", ALFI_HOST_THREADS="1"),
                             cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
        res[mode] = np.load(f)
    for k in ("2d", "3d", "sv3d"):
        assert int(res["1"][k + "_waves"]) > 4
        assert np.array_equal(res["1"][k], res["0"][k]), (k, np.abs(res["1"][k] - res["0"][k]).max())
