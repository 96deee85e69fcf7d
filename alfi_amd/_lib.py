"""ctypes binding of libalfi_hip.so -- the product's only compute path.  Fails loudly when the library is missing or
cannot be loaded; there is no CPU fallback (the oracle under oracle/ is test infrastructure and never imported here)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALFI_HIP_LIB: an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("ALFI_HIP_LIB") or os.path.join(_HERE, "libalfi_hip.so")

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class CsrHost(ctypes.Structure):
    _fields_ = [("nrows", ctypes.c_int64), ("ncols", ctypes.c_int64), ("rowptr", vp), ("colidx", vp), ("vals", vp)]


class BsrHost(ctypes.Structure):
    _fields_ = [("nbrows", ctypes.c_int64), ("nbcols", ctypes.c_int64), ("rowptr", vp), ("colidx", vp), ("vals", vp)]


# name -> (restype, argtypes); every symbol include/alfi_hip.h declares
SIGNATURES = {
    "alfi_ctx_create": (ctypes.c_int, [ctypes.c_int, vp, ctypes.POINTER(vp)]),
    "alfi_ctx_destroy": (ctypes.c_int, [vp]),
    "alfi_ctx_sync": (ctypes.c_int, [vp]),
    "alfi_last_error": (ctypes.c_char_p, [vp]),
    "alfi_malloc": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.POINTER(vp)]),
    "alfi_free": (ctypes.c_int, [vp, vp]),
    "alfi_memcpy_h2d": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64]),
    "alfi_memcpy_d2h": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64]),
    "alfi_memset0": (ctypes.c_int, [vp, vp, ctypes.c_int64]),
    "alfi_prof_enable": (ctypes.c_int, [vp, ctypes.c_int]),
    "alfi_prof_reset": (ctypes.c_int, [vp]),
    "alfi_prof_get": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "alfi_prof_get_level": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_int64)]),
    "alfi_ctx_set_comm": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64]),
    "alfi_comm_unique_id": (ctypes.c_int, [vp, ctypes.c_int64]),
    "alfi_ctx_comm_init": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int]),
    "alfi_level_set_sum_exchange": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int64, vp, vp, vp]),
    "alfi_ctx_comm_stats": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    "alfi_ctx_comm_size": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "alfi_ctx_comm_destroy": (ctypes.c_int, [vp]),
    "alfi_level_set_neighbours": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp]),
    "alfi_level_set_partition": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp, vp, vp,
                                                ctypes.c_int64]),
    "alfi_level_set_overlap": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int64]),
    "alfi_level_id": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int)]),
    "alfi_level_create": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64,
                                         ctypes.POINTER(vp)]),
    "alfi_level_destroy": (ctypes.c_int, [vp]),
    "alfi_level_update_values": (ctypes.c_int, [vp, vp]),
    "alfi_ctx_comm_allow_self": (ctypes.c_int, [vp, ctypes.c_int]),
    "alfi_vec_axpy": (ctypes.c_int, [vp, vp, vp, ctypes.c_double, ctypes.c_int64]),
    "alfi_vec_copy": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64]),
    "alfi_vec_gather": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int]),
    "alfi_level_zero_bc": (ctypes.c_int, [vp, vp]),
    "alfi_saddle_dot": (ctypes.c_int, [vp, vp, vp, ctypes.POINTER(ctypes.c_double)]),
    "alfi_transfer_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    "alfi_level_halo_sum": (ctypes.c_int, [vp, vp]),
    "alfi_level_set_assembly": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp]),
    "alfi_ctx_set_assembly_scratch": (ctypes.c_int, [vp, ctypes.c_int64]),
    "alfi_level_assemble_supg": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_int]),
    "alfi_level_assemble": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int]),
    "alfi_level_set_assembly_bc": (ctypes.c_int, [vp, vp, ctypes.c_int64]),
    "alfi_level_assembly_state_size": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64)]),
    "alfi_level_assemble_mult": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, vp, vp]),
    "alfi_level_set_supg": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, vp, vp]),
    "alfi_level_supg": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int, vp]),
    "alfi_level_apply_bc": (ctypes.c_int, [vp]),
    "alfi_level_get_values": (ctypes.c_int, [vp, vp]),
    "alfi_level_size": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64)]),
    "alfi_spmv": (ctypes.c_int, [vp, vp, vp]),
    "alfi_residual": (ctypes.c_int, [vp, vp, vp, vp]),
    "alfi_patches_set": (ctypes.c_int, [vp, ctypes.c_int64, vp, vp]),
    "alfi_patches_factor": (ctypes.c_int, [vp]),
    "alfi_patches_set_groups": (ctypes.c_int, [vp, vp]),
    "alfi_patches_factor_bytes": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64)]),
    "alfi_patches_check": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]),
    "alfi_patches_set_partition_of_unity": (ctypes.c_int, [vp, ctypes.c_int]),
    "alfi_patches_set_multiplicative": (ctypes.c_int, [vp, ctypes.c_int64, vp, ctypes.c_int]),
    "alfi_patches_multiplicative_levels": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64)]),
    "alfi_patch_apply": (ctypes.c_int, [vp, vp, vp]),
    "alfi_patches_stats": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int64)]),
    "alfi_patch_get_inverse": (ctypes.c_int, [vp, ctypes.c_int64, vp]),
    "alfi_smooth_fgmres": (ctypes.c_int, [vp, ctypes.c_int, vp, vp, ctypes.c_int]),
    "alfi_coarse_factor": (ctypes.c_int, [vp]),
    "alfi_coarse_factor_sparse": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int]),
    "alfi_coarse_factor_bytes": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int64)]),
    "alfi_coarse_residual": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double)]),
    "alfi_coarse_set_inverse": (ctypes.c_int, [vp, vp, ctypes.c_int]),
    "alfi_coarse_solve": (ctypes.c_int, [vp, vp, vp]),
    "alfi_transfer_create": (ctypes.c_int, [vp, vp, vp, ctypes.POINTER(BsrHost), ctypes.POINTER(BsrHost),
                                            ctypes.POINTER(BsrHost), ctypes.POINTER(BsrHost), ctypes.POINTER(BsrHost),
                                            ctypes.c_int64, ctypes.c_int, vp, vp, vp, ctypes.POINTER(vp)]),
    "alfi_transfer_destroy": (ctypes.c_int, [vp]),
    "alfi_transfer_update": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double]),
    "alfi_transfer_get_block_inverse": (ctypes.c_int, [vp, ctypes.c_int64, vp]),
    "alfi_prolong": (ctypes.c_int, [vp, vp, vp]),
    "alfi_transfer_set_injection": (ctypes.c_int, [vp, vp]),
    "alfi_transfer_set_injection_matrix": (ctypes.c_int, [vp, ctypes.POINTER(CsrHost)]),
    "alfi_inject": (ctypes.c_int, [vp, vp, vp]),
    "alfi_restrict": (ctypes.c_int, [vp, vp, vp, ctypes.c_int]),
    "alfi_mg_create": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_int,
                                      ctypes.c_int, ctypes.POINTER(vp)]),
    "alfi_mg_destroy": (ctypes.c_int, [vp]),
    "alfi_mg_vcycle": (ctypes.c_int, [vp, vp, vp]),
    "alfi_mg_fcycle": (ctypes.c_int, [vp, vp, vp]),
    "alfi_ctx_set_graph": (ctypes.c_int, [vp, ctypes.c_int]),
    "alfi_saddle_create": (ctypes.c_int, [vp, ctypes.POINTER(CsrHost), ctypes.POINTER(CsrHost), vp, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_int, ctypes.POINTER(vp)]),
    "alfi_saddle_destroy": (ctypes.c_int, [vp]),
    "alfi_saddle_set_mass_inverse": (ctypes.c_int, [vp, ctypes.POINTER(CsrHost)]),
    "alfi_saddle_update": (ctypes.c_int, [vp, ctypes.c_double, ctypes.c_double]),
    "alfi_saddle_solve": (ctypes.c_int, [vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "alfi_level_halo_forward": (ctypes.c_int, [vp, vp]),
    "alfi_level_halo_reverse_add": (ctypes.c_int, [vp, vp]),
    "alfi_csr_create": (ctypes.c_int, [vp, ctypes.POINTER(CsrHost), ctypes.POINTER(vp)]),
    "alfi_csr_destroy": (ctypes.c_int, [vp]),
    "alfi_csr_mult": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_double, ctypes.c_int]),
    "alfi_saddle_mult": (ctypes.c_int, [vp, vp, vp]),
    "alfi_saddle_precond": (ctypes.c_int, [vp, vp, vp]),
}

COMM_ID_BYTES = 128      # ALFI_COMM_ID_BYTES

EVENTS = ["PATCH_APPLY", "PATCH_SCATTER", "PATCH_FACTOR", "MATMULT", "BLAS1", "PROLONG", "RESTRICT", "COARSE", "COMM"]
# the PETSc log events the reference's report prints for the same work (alfi/driver.py:80; COMM: the PetscSF scatters around
# every patch apply and MatMult, alfi/solver.py:604-605)
PETSC_EVENT_NAMES = {"PATCH_APPLY": "PCPATCHApply", "PATCH_SCATTER": "PCPATCHScatter", "PATCH_FACTOR": "PCPatchComputeOp",
                     "MATMULT": "MatMult", "BLAS1": "KSPGMRESOrthog", "PROLONG": "SchoeberlProlong", "RESTRICT": "SchoeberlRestrict",
                     "COARSE": "MatSolve", "COMM": "SFBcastOpBegin"}

_lib = None


def comm_unique_id():
    """128-byte id for alfi_ctx_comm_init; call on ONE rank and distribute the bytes."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    lib = load()
    rc = lib.alfi_comm_unique_id(ctypes.cast(buf, vp), COMM_ID_BYTES)
    if rc != 0:
        raise RuntimeError("alfi_comm_unique_id failed (%d): %s" % (rc, lib.alfi_last_error(None).decode()))
    return buf.raw


def load():
    """Load libalfi_hip.so (needs libamdhip64; works without a GPU, but every compute call then returns an error)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7, and whichever copy is loaded first serves
    # both (same SONAME).  Loaded in the other order -- ours from /opt/rocm first, torch later -- torch's device init
    # reports "No HIP GPUs are available" (seen on the MI355X box).  torch supplies streams / torch.distributed to
    # alfi_amd.dist anyway, so import it first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libalfi_hip.so is missing (%s): build it with `python -m alfi_amd.build`; "
                           "alfi_amd has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError = ABI mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
