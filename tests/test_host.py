"""CPU-side tests: the C-ABI library loads and exports every symbol the header
declares, the ctypes struct matches, and the host-side layout builder
(SparseCounts) produces a correct CSR + panel-CSC.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from spmf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from spmf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "spmf_hip.h")).read()
    declared = set(re.findall(r"\b(spmf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.spmf_version() == 6 and 'define SPMF_ABI_VERSION 6' in hdr


def test_library_exports_only_the_c_abi(lib):
    """-fvisibility=hidden + the linker export map: the dynamic symbol table holds
    the header's functions and nothing else (no launch wrappers, no kernel handles)."""
    import subprocess
    from spmf_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)


def test_struct_sizes_agree_between_library_ctypes_and_integration_stub(lib):
    """spmf_sizeof_*: the library's own struct sizes == the ctypes mirrors in
    spmf_amd/_lib.py == the stub INTEGRATION.md tells a maintainer to copy."""
    from spmf_amd import _lib
    assert lib.spmf_sizeof_counts() == C.sizeof(_lib.CountsStruct) == 200
    assert lib.spmf_sizeof_sur_var() == C.sizeof(_lib.SurVar)
    assert lib.spmf_sizeof_adam_var() == C.sizeof(_lib.AdamVar)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class SpmfCounts\(C\.Structure\):.*?\n(    _fields_ = \[.*?\])\n", doc, re.S)
    assert m, "INTEGRATION.md: SpmfCounts stub not found"
    ns = {"C": C}
    exec("class SpmfCounts(C.Structure):\n" + m.group(1), ns)
    stub = ns["SpmfCounts"]
    assert C.sizeof(stub) == lib.spmf_sizeof_counts()
    assert [(n, getattr(stub, n).offset) for n, *_ in stub._fields_] == \
        [(n, getattr(_lib.CountsStruct, n).offset) for n, *_ in _lib.CountsStruct._fields_]


def test_layout_builder_argument_errors_without_gpu(lib):
    """spmf_layout_*: what is checked before the first device call."""
    from spmf_amd import _lib
    assert lib.spmf_sizeof_layout_info() == C.sizeof(_lib.LayoutInfo) == 48
    lb, sb = C.c_size_t(), C.c_size_t()
    assert lib.spmf_layout_sizes(0, 10, 5, 0, 4, C.byref(lb), C.byref(sb)) == -1       # n_cols < 1
    assert b"n_cols" in lib.spmf_layout_last_error()
    assert lib.spmf_layout_sizes(0, 10, -1, 7, 4, C.byref(lb), C.byref(sb)) == -1      # nnz < 0
    assert lib.spmf_layout_sizes(0, 10, 2 ** 31, 7, 4, C.byref(lb), C.byref(sb)) == -4  # nnz must fit int32
    assert lib.spmf_layout_sizes(0, 10, 5, 7, 0, C.byref(lb), C.byref(sb)) == -1       # panel_rows < 1
    assert lib.spmf_layout_sizes(0, 2 ** 20, 5, 8192, 1, C.byref(lb), C.byref(sb)) == -4  # n_panels * n_cols >= 2^32
    assert b"larger panels" in lib.spmf_layout_last_error()
    assert lib.spmf_layout_sizes(0, 10, 5, 7, 4, None, C.byref(sb)) == -1
    cs, info = _lib.CountsStruct(), _lib.LayoutInfo()
    info.struct_size = C.sizeof(_lib.LayoutInfo)
    assert lib.spmf_layout_build(0, 10, 5, 7, None, None, None, 4, 0, None, 0, None, 0,
                                 C.byref(cs), C.byref(info), None) == -1               # null buffers
    assert b"null" in lib.spmf_layout_last_error()


def test_ctx_lifecycle_and_argument_errors_without_gpu(lib):
    h = C.c_void_p()
    assert lib.spmf_ctx_create(0, 300, 10, 0, C.byref(h)) == -1      # K > 256
    assert lib.spmf_ctx_create(0, 100, 10, 2, C.byref(h)) == -4   # dense-term contexts: K <= 64
    assert lib.spmf_ctx_create(0, 100, 10, 0, C.byref(h)) == 0       # the whole-wave passes (csrc/widek.hip)
    assert lib.spmf_padded_k(h) == 128
    lib.spmf_ctx_destroy(h)
    assert lib.spmf_ctx_create(0, 16, 1000, 1, C.byref(h)) == 0
    assert lib.spmf_padded_k(h) == 16
    assert lib.spmf_ctx_set_prior(h, -1.0, 1.0, 0.99) == -1
    assert b"must be > 0" in lib.spmf_last_error(h)
    n1 = lib.spmf_workspace_bytes(h, 1000, 1)
    n2 = lib.spmf_workspace_bytes(h, 2000, 1)
    assert n2 - n1 == 2 * 1000 * 16 * 4
    # acc_len = 2*D*KP + D + 2*(4+KP)
    assert lib.spmf_acc_len(h, 3) == 3 * (2 * 1000 * 16 + 1000 + 2 * (6 + 16))
    lib.spmf_ctx_destroy(h)
    h3 = C.c_void_p()
    assert lib.spmf_ctx_create(0, 3, 10, 0, C.byref(h3)) == 0
    assert lib.spmf_padded_k(h3) == 4
    lib.spmf_ctx_destroy(h3)


def test_counts_struct_abi_guard_rejects_foreign_layouts(lib):
    """spmf_counts grew fields without a version signal.  Since ABI version 3
    every entry point that takes the struct refuses one whose struct_size is not the library's own
    sizeof (a caller compiled against the shorter layout, or one that never initialised the former
    reserved slot), a negative pc_pad, and a packed list stream with panels too tall for 16 bits
    -- all on the host, before anything touches the device."""
    from spmf_amd import _lib
    h = C.c_void_p()
    assert lib.spmf_ctx_create(0, 8, 100, 1, C.byref(h)) == 0
    P = _lib.PtrArray()
    cs = _lib.CountsStruct()
    cs.n_cols, cs.n_rows = 100, 4
    for bad in (0, 160, 184):                      # zero-initialised, the round-2 layout, a longer one
        cs.struct_size = bad
        assert lib.spmf_data_pass(h, C.byref(cs), 1, P, 4096, None) == -1
        assert b"struct_size" in lib.spmf_last_error(h) and b"another spmf_hip.h" in lib.spmf_last_error(h)
        assert lib.spmf_encode(h, C.byref(cs), 4096, 4096, 4096, 4096, None) == -1
        assert b"struct_size" in lib.spmf_last_error(h)
    cs.struct_size = C.sizeof(_lib.CountsStruct)
    cs.pc_pad = -1
    assert lib.spmf_data_pass(h, C.byref(cs), 1, P, 4096, None) == -1
    assert b"pc_pad" in lib.spmf_last_error(h)
    cs.pc_pad = 64
    cs.pc_ent, cs.panel_rows = 4096, 70000
    assert lib.spmf_data_pass(h, C.byref(cs), 1, P, 4096, None) == -1
    assert b"panel_rows must be <= 65536" in lib.spmf_last_error(h)
    lib.spmf_ctx_destroy(h)


def test_counts_struct_layout_matches_header():
    from spmf_amd._lib import CountsStruct
    # 2*int64 + 4*int32 + 7 pointers + double + 2 pointers + (2 pointers, 2 int32)
    # + column split: pointer + 4 int32 + the packed entry streams' pointers
    # + the deterministic mode's item order: 2 pointers + int64
    assert C.sizeof(CountsStruct) == 16 + 16 + 7 * 8 + 8 + 16 + 16 + 8 + 8 + 16 + 16 + 24
    assert CountsStruct.ent.offset == 160 and CountsStruct.pc_ent.offset == 168
    assert CountsStruct.list_first.offset == 176 and CountsStruct.item_pos.offset == 184
    assert CountsStruct.n_items.offset == 192
    assert CountsStruct.item_mid.offset == 136 and CountsStruct.col_split.offset == 144
    assert CountsStruct.max_items_half.offset == 148
    assert CountsStruct.gval.offset == 96
    assert CountsStruct.max_items_per_panel.offset == 128
    assert CountsStruct.row_ptr.offset == 32
    assert CountsStruct.lgamma_sum.offset == 88


@pytest.mark.parametrize("N,D,P,density", [(50, 13, 16, 0.3), (64, 7, 64, 0.9), (10, 5, 3, 0.0),
                                           (33, 20, 1000, 0.2)])
def test_sparse_counts_layout(N, D, P, density):
    from spmf_amd.sparse import SparseCounts
    rng = np.random.default_rng(N + D)
    x = ((rng.random((N, D)) < density) * (1 + rng.poisson(2.0, size=(N, D)))).astype(np.float32)
    if N > 3:
        x[2] = 0
    sc = SparseCounts.from_dense(x, "cpu", P)
    ref = sp.csr_matrix(x)
    np.testing.assert_array_equal(sc.row_ptr.numpy(), ref.indptr)
    np.testing.assert_array_equal(sc.col_idx.numpy(), ref.indices)
    np.testing.assert_array_equal(sc.val.numpy(), ref.data)
    np.testing.assert_array_equal(sc.to_dense().numpy(), x)
    if sc.nnz:   # integer counts below 65536 in fewer than 65536 columns: col << 16 | count
        w = sc.ent.numpy().astype(np.int64) & 0xffffffff
        np.testing.assert_array_equal(w >> 16, ref.indices)
        np.testing.assert_array_equal((w & 0xffff).astype(np.float32), ref.data)
    # scipy CSR input gives the same layout
    sc2 = SparseCounts.from_any(ref, "cpu", P)
    np.testing.assert_array_equal(sc2.pc_row.numpy(), sc.pc_row.numpy())
    # panel-CSC: every (panel, column) list holds that panel's rows, ascending
    nP = sc.n_panels
    assert nP == max(1, -(-N // sc.panel_rows))
    ptr = sc.pc_ptr.numpy().reshape(nP, D + 1)
    assert ptr[0, 0] == 0 and ptr[-1, -1] == sc.nnz
    seen = np.zeros_like(x)
    for p in range(nP):
        for d in range(D):
            rows = sc.pc_row.numpy()[ptr[p, d]:ptr[p, d + 1]]
            vals = sc.pc_val.numpy()[ptr[p, d]:ptr[p, d + 1]]
            assert np.all(np.diff(rows) > 0)
            assert np.all(rows // sc.panel_rows == p)
            np.testing.assert_array_equal(x[rows, d], vals)
            seen[rows, d] = vals
        if p + 1 < nP:
            assert ptr[p, D] == ptr[p + 1, 0]
    np.testing.assert_array_equal(seen, x)


def test_energy_fails_loudly_without_gpu():
    from spmf_amd import PoissonFactorization
    from spmf_amd._lib import SpmfError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = PoissonFactorization(latent_dim=2, feature_dim=4, initialize_distributions=False)
    with pytest.raises(SpmfError):
        m.energy_and_grads({"counts": np.ones((3, 4))}, {})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spmf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


@pytest.mark.parametrize("seg", [3, 256])
def test_column_pass_work_items_cover_every_entry_once(seg, monkeypatch):
    import spmf_amd.sparse as S
    monkeypatch.setattr(S, "SEGMENT_ENTRIES", seg)
    rng = np.random.default_rng(9)
    x = ((rng.random((70, 11)) < 0.5) * (1 + rng.poisson(2.0, size=(70, 11)))).astype(np.float32)
    x[:, 4] = 0
    x[:, 7] = 1.0                                     # a hot column: split into segments
    sc = S.SparseCounts.from_dense(x, "cpu", 32)
    items, ip = sc.items.numpy(), sc.item_ptr.numpy()
    assert ip[0] == 0 and ip[-1] == len(items) and len(ip) == sc.n_panels + 1
    seen = np.zeros(sc.nnz, dtype=int)
    ptr = sc.pc_ptr.numpy().reshape(sc.n_panels, -1)
    for p in range(sc.n_panels):
        lens = items[ip[p]:ip[p + 1], 1]
        assert np.all(np.diff(lens) <= 0)             # sorted by length inside the panel
        for start, ln, col, _ in items[ip[p]:ip[p + 1]]:
            assert 0 < ln <= seg
            assert ptr[p, col] <= start and start + ln <= ptr[p, col + 1]
            seen[start:start + ln] += 1
    assert np.all(seen == 1)
    assert not np.any(items[:, 2] == 4)               # empty column has no item


def test_work_items_sorted_by_column_half():
    """col_split: a panel's items come lower half first (item_mid marks the
    boundary), each half sorted by length; every entry still covered once."""
    import spmf_amd.sparse as S
    rng = np.random.default_rng(4)
    x = ((rng.random((90, 70)) < 0.3) * (1 + rng.poisson(2.0, size=(90, 70)))).astype(np.float32)
    sc = S.SparseCounts.from_dense(x, "cpu", 32, col_split=32)
    items, ip, mid = sc.items.numpy(), sc.item_ptr.numpy(), sc.item_mid.numpy()
    assert len(mid) == sc.n_panels
    total = 0
    for p in range(sc.n_panels):
        lo, hi = items[ip[p]:mid[p]], items[mid[p]:ip[p + 1]]
        assert np.all(lo[:, 2] < 32) and np.all(hi[:, 2] >= 32)
        assert np.all(np.diff(lo[:, 1]) <= 0) and np.all(np.diff(hi[:, 1]) <= 0)
        total += lo[:, 1].sum() + hi[:, 1].sum()
        assert sc.items_per_half[0, p] == len(lo) and sc.items_per_half[1, p] == len(hi)
    assert total == sc.nnz
    cs = sc.batch_struct(1, 3)
    assert cs.col_split == 32 and cs.item_mid == sc.item_mid.data_ptr() + 4
    assert cs.max_items_half[0] == int(sc.items_per_half[0, 1:3].max())


@pytest.mark.timeout(600)
def test_c_abi_host_layer_under_address_sanitizer():
    """an ASAN host build of the C-ABI layer (make -C spmf_amd/csrc asan:
    api.hip compiled with -fsanitize=address for the host, linked with the regular kernel
    objects) walks every argument-check / workspace / descriptor path that needs no GPU."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "spmf_amd", "csrc")
    r = subprocess.run(["make", "-s", "-C", csrc, "-j8", "all", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = r.stdout.strip().splitlines()[-1]
    assert rt.endswith(".so") and os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    w = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_asan_walk.py")],
                       capture_output=True, text=True, env=env, timeout=500)
    assert w.returncode == 0 and "asan walk ok" in w.stdout, (w.stdout[-1500:], w.stderr[-3000:])
    assert "AddressSanitizer" not in w.stderr, w.stderr[-3000:]


def test_finish_fast_log_series():
    """csrc/finish.hip fast_log: range reduction to m in [sqrt(1/2), sqrt(2)), t = (m-1)/(m+1),
    log m = 2t(1 + t^2/3 + ... + t^22/23) -- restated here in fp64 numpy and compared with
    np.log over 60 decades (the device code differs only in how 1/(m+1) is formed)."""
    rng = np.random.default_rng(5)
    x = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), size=2_000_000))
    x = np.concatenate([x, np.float32(rng.uniform(0.5, 2.0, size=200_000)).astype(np.float64),
                        [1.0, np.sqrt(2.0), np.nextafter(np.sqrt(2.0), 2.0), 2.0 ** -126, 3.4e38]])
    m, e = np.frexp(x)            # m in [0.5, 1)
    m, e = 2.0 * m, e - 1         # [1, 2)
    big = m > 1.4142135623730951
    m = np.where(big, 0.5 * m, m)
    e = e + big
    t = (m - 1.0) / (m + 1.0)
    t2 = t * t
    p = np.full_like(t, 1.0 / 23.0)
    for c in (21, 19, 17, 15, 13, 11, 9, 7, 5, 3, 1):
        p = p * t2 + 1.0 / c
    got = e * 0.69314718055994530942 + 2.0 * t * p
    ref = np.log(x)
    assert np.abs(got - ref).max() <= 4e-16 * (np.abs(ref).max() + 1.0)
    assert (np.abs(got - ref) / (np.abs(ref) + 1.0)).max() <= 3e-16


def test_balanced_panel_rows_fill_l2_and_count_in_eights():
    """Panel geometry of the column pass (spmf_amd/sparse.py balanced_panel_rows): the z / xi*gz
    rows of a panel fit 3 MB, big shards get a multiple of 8 panels (one XCD class each),
    shards of up to two panels stay whole."""
    from spmf_amd.sparse import PANEL_TABLE_BYTES, balanced_panel_rows
    for n_rows, K in [(1_000_000, 32), (500_000, 64), (125_000, 32), (122_880, 32), (200_000, 32),
                      (100_000, 16), (999_983, 50), (3_000_000, 8)]:
        pr = balanced_panel_rows(n_rows, K)
        kp = 4
        while kp < K:
            kp *= 2
        n_panels = -(-n_rows // pr)
        assert pr * 8 * kp <= PANEL_TABLE_BYTES + 64 * 8 * kp
        assert n_panels % 8 == 0, (n_rows, K, pr, n_panels)
        assert (n_panels - 1) * pr < n_rows           # no empty panel
        assert n_rows - (n_panels - 1) * pr > pr // 2   # and no sliver at the end
    assert balanced_panel_rows(1_000_000, 32) == 11392      # 88 panels (C3)
    assert balanced_panel_rows(500_000, 64) == 5696         # 88 panels (C4)
    assert balanced_panel_rows(125_000, 32) == 7872         # 16 panels (an 8-GPU shard of C3)
    assert balanced_panel_rows(20_000, 16) == 20_000        # a reference-sized minibatch: one panel
    assert balanced_panel_rows(5, 2) == 5 and balanced_panel_rows(0, 2) == 1
    # latent dimensions above 64 (csrc/widek.hip: one work item per wave, 2*KP float atomics per item): 10k-row
    # panels whatever KP is, still a multiple of 8 of them
    for K in (65, 128, 200, 256):
        pr = balanced_panel_rows(100_000, K)
        assert pr == 6272 and (-(-100_000 // pr)) % 8 == 0
        assert balanced_panel_rows(15_000, K) == 15_000       # up to two such panels: one panel


def test_packed_entries_only_for_integer_counts_in_16_bits():
    """spmf_counts.ent (col << 16 | count) exists only when it is exact: real-valued or
    large entries, or more than 65536 columns, keep the canonical col / val stream."""
    from spmf_amd.sparse import SparseCounts
    x = np.zeros((6, 9), dtype=np.float32)
    x[1, 3], x[2, 8], x[4, 0] = 2.0, 65535.0, 7.0
    sc = SparseCounts.from_dense(x, "cpu", 4)
    w = sc.ent.numpy().astype(np.int64) & 0xffffffff
    assert w.tolist() == [(3 << 16) | 2, (8 << 16) | 65535, 7]
    for bad in (2.5, 65536.0):
        y = x.copy()
        y[1, 3] = bad
        assert SparseCounts.from_dense(y, "cpu", 4).ent is None
    wide = sp.csr_matrix((np.ones(2, np.float32), (np.array([0, 1]), np.array([5, 70000]))), shape=(2, 70001))
    assert SparseCounts.from_any(wide, "cpu", 2).ent is None
    top = sp.csr_matrix((np.ones(1, np.float32), (np.array([0]), np.array([65535]))), shape=(1, 65536))
    t = SparseCounts.from_any(top, "cpu", 1)
    assert (int(t.ent[0]) & 0xffffffff) == (65535 << 16) | 1


def test_float64_request_is_answered_once(capsys):
    """poisson.py:64 / bin/factorize_csv.py:119: the reference computes in float64 and its callers say so.  This
    build accepts the keyword and computes in float32 with float64 accumulation: a caller who asks for float64 is
    told so, once per process; the default construction says nothing extra."""
    import numpy as np
    from spmf_amd import PoissonFactorization
    PoissonFactorization._dtype_notice_given = False
    m = PoissonFactorization(latent_dim=2, feature_dim=5, device="cpu", initialize_distributions=False)
    assert "float64" not in capsys.readouterr().out and m.dtype is not None
    PoissonFactorization(latent_dim=2, feature_dim=5, dtype=np.float64, device="cpu", initialize_distributions=False)
    out = capsys.readouterr().out
    assert "float32 storage" in out and "float64 accumulation" in out and "min(y, 70)" in out
    PoissonFactorization(latent_dim=2, feature_dim=5, dtype="float64", device="cpu", initialize_distributions=False)
    assert "float32 storage" not in capsys.readouterr().out
