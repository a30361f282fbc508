"""ctypes binding of the C-ABI in ``include/vittles_hip.h`` (libvittles_hip.so).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``make -C vittles_amd/csrc``.  There is NO CPU fallback: if the library is
missing, or no HIP device is visible, every compute entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITTLES_HIP_LIB", os.path.join(_HERE, "libvittles_hip.so"))  # env: dev A/B builds

c_double_p = ctypes.c_void_p  # device pointers travel as integers
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_dbl = ctypes.c_double
c_ptr = ctypes.c_void_p

VT_OK = 0
FAMILY = {"logistic": 0, "poisson": 1, "gaussian": 2}
SMALL_D_MAX = 128

# name -> (restype, argtypes); every symbol include/vittles_hip.h declares.
SIGNATURES = {
    "vt_version": (c_int, []),
    "vt_error_string": (ctypes.c_char_p, [c_int]),
    "vt_device_count": (c_int, []),
    "vt_glm_hessian_workspace_bytes": (c_size_t, [c_i64, c_int]),
    "vt_glm_hessian": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int,
                               c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_potrf_workspace_bytes": (c_size_t, [c_int]),
    "vt_potrf": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_potrs": (c_int, [c_ptr, c_int, c_int, c_ptr, c_i64, c_i64, c_i64, c_ptr]),
    "vt_trsm_pack_bytes": (c_size_t, [c_int]),
    "vt_trsm_pack": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "vt_potrf_pack": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "vt_potrf_pack_from": (c_int, [c_ptr, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "vt_trsm_rows": (c_int, [c_ptr, c_int, c_ptr, c_i64, c_ptr, c_dbl, c_i64, c_ptr, c_i64, c_ptr]),
    "vt_rows_workspace_bytes": (c_size_t, [c_int]),
    "vt_glm_link": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_i64, c_int,
                            c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_grad": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int,
                            c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_hvp": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_hvp_noalloc": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "vt_gemv_t": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_gemv": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_gemv_t2": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_taylor2_rhs": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_ptr,
                                   c_ptr, c_size_t, c_ptr]),
    "vt_glm_taylor3_rhs": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_ptr,
                                   c_ptr, c_size_t, c_ptr]),
    "vt_gemm_tn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vt_gemm_tn": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_dbl, c_ptr, c_ptr, c_i64,
                           c_ptr, c_size_t, c_ptr]),
    "vt_gemm_tn_f32b": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_dbl, c_ptr, c_ptr, c_i64,
                                c_ptr, c_size_t, c_ptr]),
    "vt_cg_init": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "vt_cg_update": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "vt_cg_init_multi": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    "vt_cg_update_multi": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
    "vt_glm_hvp_multi_workspace_bytes": (c_size_t, [c_i64, c_int, c_int]),
    "vt_glm_hvp_multi": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_i64, c_int, c_ptr, c_i64,
                                 c_ptr, c_size_t, c_ptr]),
    "vt_comm_unique_id": (c_int, [c_ptr]),
    "vt_comm_init": (c_int, [ctypes.POINTER(c_ptr), c_int, c_int, c_ptr]),
    "vt_comm_destroy": (c_int, [c_ptr]),
    "vt_allreduce_sum": (c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "vt_spmv_csr": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "vt_dot3": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "vt_vi_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vt_vi_hvp": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_dbl, c_dbl, c_int, c_int, c_int, c_int, c_ptr,
                          c_ptr, c_size_t, c_ptr]),
    "vt_vi_grad": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_dbl, c_dbl, c_int, c_int, c_int, c_int, c_ptr,
                           c_ptr, c_size_t, c_ptr]),
    "vt_vi_value": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_dbl, c_dbl, c_int, c_int, c_int, c_int, c_ptr,
                            c_ptr, c_size_t, c_ptr]),
    "vt_block_colour_vec": (c_int, [c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    "vt_block_gather": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "vt_global_gather": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                 c_ptr, c_ptr]),
    "vt_arrow_factor": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "vt_arrow_local_solve": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int,
                                     c_int, c_int, c_ptr]),
    "vt_arrow_back": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "vt_glm_value": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_i64, c_int,
                             c_ptr, c_ptr, c_size_t, c_ptr]),
    # fp32 storage mode: same argument lists as the fp64 namesakes (pointers travel as integers)
    "vt_glm_hessian_f32_workspace_bytes": (c_size_t, [c_i64, c_int]),
    "vt_glm_hessian_f32": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int,
                                   c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_trsm_pack_f32_bytes": (c_size_t, [c_int]),
    "vt_trsm_pack_f32": (c_int, [c_ptr, c_int, c_ptr, c_ptr]),
    "vt_trsm_rows_f32": (c_int, [c_ptr, c_int, c_ptr, c_i64, c_ptr, c_dbl, c_i64, c_ptr, c_i64, c_ptr]),
    "vt_trsm_rows_f32_large_workspace_bytes": (c_size_t, [c_i64, c_int]),
    "vt_trsm_rows_f32_large": (c_int, [c_ptr, c_int, c_ptr, c_i64, c_ptr, c_dbl, c_i64, c_ptr, c_i64, c_ptr, c_size_t,
                                       c_ptr]),
    "vt_glm_link_f32": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_i64, c_int,
                                c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_grad_f32": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int,
                                c_ptr, c_ptr, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_hvp_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_dbl, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t,
                               c_ptr]),
    "vt_gemv_t_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_taylor_coeff": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_dbl, c_int, c_int, c_i64, c_ptr, c_ptr]),
    "vt_gemv_t2_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_gemv_f32": (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_glm_value_f32": (c_int, [c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_dbl, c_i64, c_int,
                                 c_ptr, c_ptr, c_size_t, c_ptr]),
    "vt_synth_normal": (c_int, [ctypes.c_uint64, ctypes.c_uint, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "vt_synth_normal_f32": (c_int, [ctypes.c_uint64, ctypes.c_uint, c_i64, c_i64, c_int, c_ptr, c_i64, c_ptr]),
    "vt_synth_response": (c_int, [ctypes.c_uint64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr]),
}

_lib = None


class VittlesHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and set the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VittlesHipError(
            "libvittles_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vittles_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def fn(lib, name, t):
    """The entry point ``name`` for the storage type of the CUDA tensor ``t`` (fp32 mode: ``name_f32``)."""
    import torch
    if t.dtype == torch.float32:
        return getattr(lib, name + "_f32")
    if t.dtype != torch.float64:
        raise VittlesHipError("%s: tensors must be float64 (or float32 in the fp32 storage mode), got %s"
                              % (name, t.dtype))
    return getattr(lib, name)


def check(code, what=""):
    if code != VT_OK:
        msg = load().vt_error_string(int(code)).decode()
        raise VittlesHipError("%s failed: %s (code %d)" % (what or "vittles_hip call", msg, code))
