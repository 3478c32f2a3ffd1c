"""ctypes binding of libspmf_hip.so (the C-ABI in include/spmf_hip.h).

There is no CPU fallback: if the library is missing or a call fails this
module raises.  PyTorch tensors are used purely as device storage
(``tensor.data_ptr()`` handed to ctypes).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: SPMF_LIB_PATH: another build of the same library (tools/build_variant.sh: kernel experiments)
LIB_PATH = os.environ.get("SPMF_LIB_PATH") or os.path.join(_HERE, "libspmf_hip.so")

NVARS = 12
ABI_VERSION = 6          # include/spmf_hip.h SPMF_ABI_VERSION
VI_STATE_LEN = 16
NPARTS = 14
#: variable order of the C-ABI = the reference's var_list (poisson.py:403-539,572)
VAR_ORDER = ("v", "w", "u", "u_eta", "u_tau", "s_eta", "s_tau", "s",
             "u_eta_a", "u_tau_a", "s_eta_a", "s_tau_a")
PART_ORDER = VAR_ORDER + ("z", "x")

FLAG_SCALE_ROWS = 1
FLAG_LOG_TRANSFORM = 2
FLAG_BERNOULLI = 4
FLAG_MIXED = 8
FLAG_ABS_HORSESHOE = 16
#: horshoe_plus=False: the four variables of poisson.py:378-398 in surrogate order (:540-565)
VAR_ORDER_ABS = ("v", "w", "s", "u")


class SpmfError(RuntimeError):
    pass


class CountsStruct(C.Structure):
    """struct spmf_counts (include/spmf_hip.h)."""
    _fields_ = [
        ("n_rows", C.c_int64), ("nnz", C.c_int64),
        ("n_cols", C.c_int32), ("n_panels", C.c_int32),
        ("panel_rows", C.c_int32), ("row_base", C.c_int32),
        ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("val", C.c_void_p),
        ("row_scale", C.c_void_p),
        ("pc_ptr", C.c_void_p), ("pc_row", C.c_void_p), ("pc_val", C.c_void_p),
        ("lgamma_sum", C.c_double),
        ("gval", C.c_void_p), ("pc_gval", C.c_void_p),
        ("item_ptr", C.c_void_p), ("items", C.c_void_p),
        ("max_items_per_panel", C.c_int32), ("pc_pad", C.c_int32),
        ("item_mid", C.c_void_p), ("col_split", C.c_int32),
        ("max_items_half", C.c_int32 * 2), ("struct_size", C.c_int32),
        ("ent", C.c_void_p), ("pc_ent", C.c_void_p),
        ("list_first", C.c_void_p), ("item_pos", C.c_void_p), ("n_items", C.c_int64),
    ]


class LayoutInfo(C.Structure):
    """struct spmf_layout_info"""
    _fields_ = [("struct_size", C.c_int32), ("n_panels", C.c_int32), ("panel_rows", C.c_int32),
                ("segment", C.c_int32), ("n_items", C.c_int64), ("packed_ent", C.c_int32),
                ("packed_pc_ent", C.c_int32), ("items_per_panel", C.c_void_p),
                ("items_lower", C.c_void_p)]


PtrArray = C.c_void_p * NVARS


class SurVar(C.Structure):
    """struct spmf_sur_var"""
    _fields_ = [("t0", C.c_void_p), ("t1", C.c_void_p), ("noise", C.c_void_p),
                ("dgda", C.c_void_p), ("theta", C.c_void_p), ("gtheta", C.c_void_p),
                ("g0", C.c_void_p), ("g1", C.c_void_p), ("n", C.c_int32), ("kind", C.c_int32),
                ("ident", C.c_void_p), ("noise_ld", C.c_int64)]


class AdamVar(C.Structure):
    """struct spmf_adam_var"""
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p),
                ("n", C.c_int32), ("reserved_", C.c_int32)]

#: every symbol include/spmf_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "spmf_version": (C.c_int, []),
    "spmf_sizeof_counts": (C.c_size_t, []),
    "spmf_sizeof_sur_var": (C.c_size_t, []),
    "spmf_sizeof_adam_var": (C.c_size_t, []),
    "spmf_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint,
                                  C.POINTER(C.c_void_p)]),
    "spmf_ctx_destroy": (None, [C.c_void_p]),
    "spmf_last_error": (C.c_char_p, [C.c_void_p]),
    "spmf_ctx_set_prior": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "spmf_ctx_set_column_types": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spmf_ctx_set_bernoulli_columns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "spmf_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int]),
    "spmf_det_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int]),
    "spmf_ctx_set_deterministic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spmf_ctx_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spmf_counts_stats": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 7
                          + [C.c_void_p]),
    "spmf_counts_colstats": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "spmf_counts_gvals": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "spmf_sizeof_layout_info": (C.c_size_t, []),
    "spmf_layout_sizes": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "spmf_layout_build": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.POINTER(CountsStruct),
                                    C.POINTER(LayoutInfo), C.c_void_p]),
    "spmf_layout_sizes_k": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "spmf_layout_build_k": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t, C.POINTER(CountsStruct),
                                      C.POINTER(LayoutInfo), C.c_void_p]),
    "spmf_layout_last_error": (C.c_char_p, []),
    "spmf_dense_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "spmf_dense_row_ptr": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "spmf_dense_fill_csr": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmf_data_pass": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_int,
                                 PtrArray, C.c_void_p, C.c_void_p]),
    "spmf_acc_ptr": (C.c_void_p, [C.c_void_p]),
    "spmf_acc_len": (C.c_int64, [C.c_void_p, C.c_int]),
    "spmf_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double,
                              C.c_double, PtrArray, C.c_void_p, C.c_void_p, PtrArray,
                              C.c_void_p, C.c_void_p]),
    "spmf_step_begin": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_int, C.c_double, PtrArray,
                                  C.c_void_p, C.c_void_p, PtrArray, C.c_void_p, C.c_void_p]),
    "spmf_step_end": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_void_p]),
    "spmf_elbo_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_int,
                                    C.c_double, PtrArray, C.c_void_p, C.c_void_p, PtrArray,
                                    C.c_void_p, C.c_void_p]),
    "spmf_encode": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct)] + [C.c_void_p] * 5),
    "spmf_dense_ll": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct)] + [C.c_void_p] * 8),
    "spmf_nonfinite_reduce": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "spmf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "spmf_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "spmf_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "spmf_comm_destroy": (C.c_int, [C.c_void_p]),
    "spmf_p2p_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "spmf_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spmf_p2p_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "spmf_p2p_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "spmf_p2p_disconnect": (C.c_int, [C.c_void_p]),
    "spmf_p2p_destroy": (C.c_int, [C.c_void_p]),
    "spmf_nonfinite_argmin": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                        C.c_void_p, C.c_void_p]),
    "spmf_nonfinite_lgamma": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "spmf_nonfinite_patch": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_int, PtrArray,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmf_surrogate_fwd": (C.c_int, [C.c_void_p, C.POINTER(SurVar), C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "spmf_sample_noise": (C.c_int, [C.c_void_p, C.POINTER(SurVar), C.c_int, C.c_int, C.c_uint64,
                                    C.c_uint64, C.c_void_p, C.c_void_p]),
    "spmf_sample_transform": (C.c_int, [C.c_void_p, C.POINTER(SurVar), C.c_int, C.c_int, C.c_uint64,
                                        C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmf_surrogate_bwd": (C.c_int, [C.c_void_p, C.POINTER(SurVar), C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_void_p]),
    "spmf_ctx_set_column_split": (C.c_int, [C.c_void_p, C.c_int]),
    "spmf_ctx_set_rows_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spmf_acc_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "spmf_data_pass_split": (C.c_int, [C.c_void_p, C.POINTER(CountsStruct), C.c_int, PtrArray,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "spmf_prior_async": (C.c_int, [C.c_void_p, C.c_int, C.c_double, PtrArray, C.c_void_p, C.c_void_p,
                                   PtrArray, C.c_void_p]),
    "spmf_vi_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                               C.c_double, C.c_void_p, C.c_void_p]),
    "spmf_adam_step_dev": (C.c_int, [C.c_void_p, C.POINTER(AdamVar), C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "spmf_surrogate_bwd_adam_dev": (C.c_int, [C.c_void_p, C.POINTER(SurVar), C.c_int, C.c_int,
                                              C.c_double, C.c_double, C.POINTER(AdamVar),
                                              C.c_void_p, C.c_void_p]),
    "spmf_adam_step": (C.c_int, [C.c_void_p, C.POINTER(AdamVar), C.c_int, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                                 C.c_void_p]),
    "spmf_ctx_set_e_cap": (C.c_int, [C.c_void_p, C.c_size_t]),
    "spmf_padded_k": (C.c_int, [C.c_void_p]),
    "spmf_z_ptr": (C.c_void_p, [C.c_void_p]),
    "spmf_gz_ptr": (C.c_void_p, [C.c_void_p]),
    "spmf_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "spmf_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """Load libspmf_hip.so (built in-tree by ``__graft_entry__.build()`` /
    ``make -C spmf_amd/csrc``).  Raises SpmfError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpmfError(
            f"{LIB_PATH} not found: build it with `make -C spmf_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    if lib.spmf_version() != ABI_VERSION:
        raise SpmfError(f"{LIB_PATH} has ABI version {lib.spmf_version()}, this binding is written "
                        f"for {ABI_VERSION} (include/spmf_hip.h SPMF_ABI_VERSION): rebuild the library")
    # the ctypes mirrors must have the library's own struct sizes
    for fn, st in ((lib.spmf_sizeof_counts, CountsStruct), (lib.spmf_sizeof_sur_var, SurVar),
                   (lib.spmf_sizeof_adam_var, AdamVar), (lib.spmf_sizeof_layout_info, LayoutInfo)):
        if fn() != C.sizeof(st):
            raise SpmfError(f"{st.__name__}: ctypes mirror is {C.sizeof(st)} bytes, the library's "
                            f"struct {fn()} (include/spmf_hip.h and spmf_amd/_lib.py disagree)")
    _lib = lib
    return lib


def check(ctx_handle, rc, what):
    if rc != 0:
        msg = load().spmf_last_error(ctx_handle)
        raise SpmfError(f"{what} failed (rc={rc}): "
                        f"{msg.decode() if msg else '?'}")
