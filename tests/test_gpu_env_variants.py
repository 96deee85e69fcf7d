"""The library's tuning switches (environment variables read once per process) select alternative kernels or skip
refinement steps; each must still give the cycle -- exactly where it only changes the memory path, to the stated accuracy
where it drops a refinement.  -m gpu; one subprocess per switch."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env,tol", [({}, 1e-5),
                                     ({"ALFI_NT": "0"}, 1e-5),                    # plain instead of nontemporal loads
                                     ({"ALFI_SPMV": "legacy"}, 1e-5),             # row-per-wave SpMV, (nnzb, bs, bs) values
                                     ({"ALFI_BIG_SPLIT": "1"}, 1e-5),             # one workgroup per large patch
                                     ({"ALFI_BIG_SCRATCH_MB": "64"}, 1e-5),       # many small factorisation batches
                                     # block elimination without the Newton-Schulz polish: the residual probe of
                                     # alfi_patches_factor (7e-6 here) would reject these inverses at its default 1e-6
                                     ({"ALFI_BIG_POLISH": "0", "ALFI_PATCH_CHECK_TOL": "1e-3"}, 5e-2),
                                     ({"ALFI_TRANSFER_REFINE": "0"}, 1e-3),       # explicit block inverses without refinement
                                     # condensed macro-star factors: the one-launch apply of rounds 1-2, and the settings of
                                     # the three-launch one that the defaults do not take on this hierarchy
                                     ({"ALFI_COND_SPLIT": "0"}, 1e-5), ({"ALFI_COND_SPLIT": "0", "ALFI_COND_BALANCE": "0"}, 1e-5),
                                     ({"ALFI_COND_GROUP_NT": "1"}, 1e-5),
                                     # three launches: a workgroup per PATCH in the group products (2), per chunk of groups
                                     # (3); the default picks by the number of patches of the launch
                                     ({"ALFI_COND_SPLIT": "2"}, 1e-5), ({"ALFI_COND_SPLIT": "3"}, 1e-5),
                                     ({"ALFI_COND_SPLIT": "3", "ALFI_COND_GROUP_NT": "1"}, 1e-5),
                                     ({"ALFI_COND_SPLIT": "2", "ALFI_COND_WAVES": "8", "ALFI_COND_GROUP_NT": "1"}, 1e-5),
                                     ({"ALFI_COND_SPLIT": "2", "ALFI_COND_WAVES": "8", "ALFI_COND_RU": "16"}, 1e-5),
                                     # EVERY condensed patch flagged by the residual probe: Schur complements formed again
                                     # and re-inverted in place by the pivoted LU (kernels_check.hip: cond_repair)
                                     ({"ALFI_PATCH_CHECK_TOL": "1e-14", "ALFI_PATCH_CHECK_FAIL": "1e-6"}, 1e-5)])
def test_switch(env, tol):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_env_variant_worker.py")],
                         env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=600)
    m = re.search(r"RELERR (\S+)", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-2000:] + out.stderr[-3000:]
    assert float(m.group(1)) < tol, (env, m.group(1))


@pytest.mark.parametrize("env", [{},                                                  # defaults: four-launch fused iteration on the small levels
                                 {"ALFI_PATCH_IL": "0"},                              # small patches applied from the row-piece storage
                                 {"ALFI_FUSED_SMOOTHER": "0"},                        # the general launch chain
                                 {"ALFI_FUSED_ALL": "1"},                             # normalisation folded behind the patch solves on every level
                                 {"ALFI_SMALL_LEVEL_WG": "0"},                        # a wave per patch on every level
                                 {"ALFI_SPMV_DEDUP": "0"},                            # direct x gathers in the large SpMV
                                 {"ALFI_SPMV_ALIGNED": "0", "ALFI_FUSED_SMOOTHER": "0"},   # de-duplicated SpMV on small levels too
                                 {"ALFI_SPMV_ALIGNED": "0", "ALFI_XCD_MAP": "1", "ALFI_FUSED_SMOOTHER": "0"},
                                 {"ALFI_SPMV_ALIGNED": "0", "ALFI_XCD_MAP": "2", "ALFI_FUSED_SMOOTHER": "0"},   # strips of 2 workgroups per XCD
                                 {"ALFI_SPMV_ALIGNED": "0", "ALFI_XCD_MAP": "0", "ALFI_FUSED_SMOOTHER": "0"},
                                 {"ALFI_FUSED_REDUCE_MAX": "0"}, {"ALFI_INVERT_MFMA": "0"}, {"ALFI_INVERT_MFMA": "2"},
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
