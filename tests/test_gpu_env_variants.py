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
