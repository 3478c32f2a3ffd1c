"""Run under LD_PRELOAD=<asan runtime>: loads the AddressSanitizer build of the
C-ABI layer and walks its host-side paths that need no GPU (argument checks,
workspace arithmetic, struct handling, error strings, RCCL id).  Exit code 0 and
no AddressSanitizer report on stderr == pass (tests/test_host.py checks both)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spmf_amd import _lib  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "spmf_amd", "libspmf_hip_asan.so"))
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args

assert lib.spmf_version() == 6 and lib.spmf_sizeof_counts() == C.sizeof(_lib.CountsStruct)
h = C.c_void_p()
assert lib.spmf_ctx_create(0, 300, 10, 0, C.byref(h)) == -1
assert lib.spmf_ctx_create(0, 0, 10, 0, C.byref(h)) == -1
assert lib.spmf_ctx_create(0, 8, 10, _lib.FLAG_LOG_TRANSFORM | _lib.FLAG_MIXED, C.byref(h)) == -4
for flags in (0, 1, 2 | 1, 4, 4 | 2, 8, 16 | 1):
    for K in (1, 3, 16, 33, 64):
        assert lib.spmf_ctx_create(0, K, 1000, flags, C.byref(h)) == 0
        kp = lib.spmf_padded_k(h)
        assert kp >= K and kp in (4, 8, 16, 32, 64)
        for S in (1, 2, 20):
            n1 = lib.spmf_workspace_bytes(h, 0, S)
            n2 = lib.spmf_workspace_bytes(h, 4096, S)
            assert n2 > n1 > 0 and lib.spmf_acc_len(h, S) == S * (2 * 1000 * kp + 1000 + 2 * (6 + kp))
        assert lib.spmf_ctx_set_prior(h, 0.0, 1.0, 0.99) == -1 and b"> 0" in lib.spmf_last_error(h)
        assert lib.spmf_ctx_set_prior(h, 0.01, 1.0, 0.99) == 0
        assert lib.spmf_ctx_set_e_cap(h, 1000) == -1 and lib.spmf_ctx_set_e_cap(h, 1 << 26) == 0
        assert lib.spmf_ctx_set_workspace(h, None, 10) == -1
        assert lib.spmf_ctx_set_workspace(h, 12345, 1 << 20) == -1          # misaligned
        assert lib.spmf_ctx_set_column_split(h, 33) == -1
        rc = lib.spmf_ctx_set_column_split(h, 512)
        assert rc == (0 if flags in (0, 1, 16 | 1) else -4), (flags, rc)
        off, ln = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        assert lib.spmf_acc_split(h, off, ln) == 0 and off[1] == ln[0]
        assert lib.spmf_ctx_set_column_split(h, 0) == 0
        # calls that must fail on their argument checks before touching the device
        cs = _lib.CountsStruct()
        cs.n_cols = 999
        P = _lib.PtrArray()
        # ABI guard: a struct from a caller built against another header is refused first
        assert lib.spmf_data_pass(h, C.byref(cs), 1, P, 4096, None) == -1
        assert b"struct_size" in lib.spmf_last_error(h)
        cs.struct_size = C.sizeof(_lib.CountsStruct)
        assert lib.spmf_data_pass(h, C.byref(cs), 1, P, None, None) == -1
        assert lib.spmf_data_pass(h, None, 1, P, 4096, None) != 0
        assert lib.spmf_finish(h, 1, 10, 0.0, 1.0, P, 4096, 4096, P, None, None) == -1
        assert lib.spmf_encode(h, C.byref(cs), None, None, None, None, None) == -1
        assert lib.spmf_dense_ll(h, C.byref(cs), *([None] * 8)) == -1
        assert lib.spmf_nonfinite_reduce(h, -1, 4096, 0, 4096, None) == -1
        assert lib.spmf_nonfinite_argmin(h, 8, None, 0.0, 4096, None) == -1
        assert lib.spmf_nonfinite_patch(h, C.byref(cs), 1, P, None, None, None, None) == -1
        sv = (_lib.SurVar * 12)()
        assert lib.spmf_surrogate_fwd(h, sv, 12, 1, 4096, None) == -1
        assert lib.spmf_surrogate_bwd(h, sv, 13, 1, 1.0, 1.0, None) == -1
        assert lib.spmf_sample_noise(h, sv, 12, 1, 1, 0, None, None) == -1
        assert lib.spmf_sample_transform(h, sv, 12, 1, 1, 0, None, 4096, None) == -1     # every variable skipped
        assert lib.spmf_sample_transform(h, sv, 12, 1, 1, 0, None, None, None) == -1
        av = (_lib.AdamVar * 24)()
        assert lib.spmf_adam_step(h, av, 24, 1e-3, 0.9, 0.999, 1e-7, 1, 0.0, None) == -1
        assert lib.spmf_adam_step_dev(h, av, 25, 4096, None) == -1
        assert lib.spmf_vi_gate(h, None, None, None, 1, 1.0, 1.0, None, None) == -1
        assert lib.spmf_allreduce(h, 4096, 8, None) == -1 and b"comm_init" in lib.spmf_last_error(h)
        assert lib.spmf_comm_init(h, None, 0, 1) == -1
        # ABI 6: the step with its outputs up front, the peer-pointer collective, the rows event
        assert lib.spmf_step_end(h, 10, 0.0, None) == -1 and b"step_begin" in lib.spmf_last_error(h)
        assert lib.spmf_step_begin(h, C.byref(cs), 1, 1.0, P, None, 4096, P, None, None) == -1
        assert lib.spmf_step_begin(h, C.byref(cs), 1, 1.0, P, 4096, None, P, None, None) == -1
        assert lib.spmf_p2p_connect(h, None) == -1
        hb = (C.c_char * 64)()
        assert lib.spmf_p2p_connect(h, hb) == -1 and b"p2p_init" in lib.spmf_last_error(h)
        assert lib.spmf_p2p_init(h, 0, 17, 1024, 0, hb) == -1          # more than 16 ranks
        assert lib.spmf_p2p_init(h, 3, 2, 1024, 0, hb) == -1           # rank outside the world
        assert lib.spmf_p2p_init(h, 0, 2, 0, 0, hb) == -1              # nothing to reduce
        assert lib.spmf_p2p_status(h, None) == -1
        assert lib.spmf_p2p_enable(h, 1) == -1 and lib.spmf_p2p_enable(h, 0) == 0
        assert lib.spmf_p2p_disconnect(h) == 0
        assert lib.spmf_p2p_destroy(h) == 0
        assert lib.spmf_ctx_set_rows_event(h, None) == 0
        assert lib.spmf_comm_destroy(h) == 0
        # deterministic mode: scratch arithmetic, the contexts it is refused for
        n0, n1 = lib.spmf_det_scratch_bytes(h, 0, 1), lib.spmf_det_scratch_bytes(h, 1000, 1)
        assert n1 - n0 == (1000 * (2 * kp + 4) * 4 + 255) // 256 * 256 and lib.spmf_det_scratch_bytes(h, 1000, 3) == 3 * n1
        if flags in (0, 1, 16 | 1):
            assert lib.spmf_ctx_set_deterministic(h, 4096 + 8, n1) == -1      # misaligned
            assert lib.spmf_ctx_set_deterministic(h, 4096, 100) == -3          # too small
            assert lib.spmf_ctx_set_deterministic(h, 4096, n1) == 0
            cs_d = _lib.CountsStruct()
            cs_d.struct_size = C.sizeof(_lib.CountsStruct)
            cs_d.n_cols = 1000
            cs_d.n_rows, cs_d.nnz, cs_d.row_ptr, cs_d.col_idx, cs_d.val = 8, 8, 4096, 4096, 4096
            cs_d.pc_row = cs_d.pc_val = cs_d.item_ptr = cs_d.items = cs_d.pc_ptr = 4096
            cs_d.n_panels, cs_d.panel_rows = 1, 8
            assert lib.spmf_data_pass(h, C.byref(cs_d), 1, _lib.PtrArray(*([4096] * 12)), 4096, None) == -1
            assert b"list_first" in lib.spmf_last_error(h)                      # counts without the item order
            assert lib.spmf_ctx_set_deterministic(h, None, 0) == 0             # off again
        else:
            assert lib.spmf_ctx_set_deterministic(h, 4096, n1) == -4 and b"linear decoder" in lib.spmf_last_error(h)
        assert lib.spmf_counts_colstats(h, None, None, None, None) == -1
        assert lib.spmf_counts_colstats(h, C.byref(cs), None, None, None) == -1          # n_cols != D
        assert lib.spmf_counts_gvals(h, C.byref(cs), None, None, None, None) == -1
        ms = (C.c_float * 6)()
        assert lib.spmf_last_timing(h, ms) == -1
        lib.spmf_ctx_destroy(h)
# the layout builder: geometry and buffer arithmetic, everything checked before the first device call
assert lib.spmf_sizeof_layout_info() == C.sizeof(_lib.LayoutInfo)
lb, sb = C.c_size_t(), C.c_size_t()
for args in ((10, 5, 0, 4), (10, -1, 7, 4), (-1, 5, 7, 4), (10, 5, 7, 0)):
    assert lib.spmf_layout_sizes(0, *args, C.byref(lb), C.byref(sb)) == -1 and lib.spmf_layout_last_error()
assert lib.spmf_layout_sizes(0, 10, 2 ** 31, 7, 4, C.byref(lb), C.byref(sb)) == -4
assert lib.spmf_layout_sizes(0, 2 ** 31, 5, 7, 4, C.byref(lb), C.byref(sb)) == -4
assert lib.spmf_layout_sizes(0, 10, 5, 7, 4, None, None) == -1
assert lib.spmf_layout_sizes_k(0, 10, 5, 7, 4, -1, C.byref(lb), C.byref(sb)) == -1            # negative latent_dim hint
assert lib.spmf_layout_build_k(0, 10, 5, 7, 4096, 4096, 4096, 4, 0, -3, 4096, 1 << 20, 4096, 1 << 20,
                               C.byref(_lib.CountsStruct()), C.byref(_lib.LayoutInfo()), None) == -1
cs0, info = _lib.CountsStruct(), _lib.LayoutInfo()
assert lib.spmf_layout_build(0, 10, 5, 7, 4096, 4096, 4096, 4, 0, 4096, 1 << 20, 4096, 1 << 20,
                             C.byref(cs0), C.byref(info), None) == -1 and b"struct_size" in lib.spmf_layout_last_error()
info.struct_size = C.sizeof(_lib.LayoutInfo)
assert lib.spmf_layout_build(0, 10, 5, 7, 4096, None, None, 4, 0, 4096, 1 << 20, 4096, 1 << 20,
                             C.byref(cs0), C.byref(info), None) == -1            # col_idx / val missing
assert lib.spmf_layout_build(0, 10, 5, 7, 4096, 4096, 4096, 4, 9, 4096, 1 << 20, 4096, 1 << 20,
                             C.byref(cs0), C.byref(info), None) == -1            # col_split > n_cols
assert lib.spmf_layout_build(0, 10, 5, 7, 4096, 4096, 4096, 4, 0, 4097, 1 << 20, 4096, 1 << 20,
                             C.byref(cs0), C.byref(info), None) == -1            # misaligned buffer
assert lib.spmf_layout_build(0, 10, 5, 7, 4096, 4096, 4096, 4, 0, 4096, 1 << 20, 4096, 1 << 20,
                             None, C.byref(info), None) == -1
assert lib.spmf_dense_scratch_bytes(0) == 24 and lib.spmf_dense_scratch_bytes(1025) == 32
assert lib.spmf_dense_row_ptr(0, 5, 0, 4096, 4, 4096, 4096, 64, None) == -1
assert lib.spmf_dense_row_ptr(0, 5, 4, 4096, 3, 4096, 4096, 64, None) == -1     # ld < n_cols
assert lib.spmf_dense_row_ptr(0, 5, 4, 4096, 4, 4096, 4096, 8, None) == -3      # scratch too small
assert lib.spmf_dense_fill_csr(0, 5, 4, 4096, 3, 4096, 4096, 4096, None) == -1
assert lib.spmf_dense_fill_csr(0, 0, 4, None, 4, 4096, None, None, None) == 0
buf = (C.c_char * 128)()
assert lib.spmf_comm_unique_id(buf) in (0, -4) and lib.spmf_comm_unique_id(None) == -1
lib.spmf_ctx_destroy(None)
assert lib.spmf_last_error(None) == b"null ctx"
print("asan walk ok")
