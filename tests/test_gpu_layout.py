"""The library's device layout builder (csrc/layout.hip, spmf_layout_build) against the torch
construction of the same arrays (spmf_amd/sparse.py _build_panel_csc / _build_items, the host-side
statement of the layout): integer work, so every array must agree bit for bit.  The layout is
what replaces the dense [B,D] batch the reference hands its model (poisson.py:170,182)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _csr(rows, D, density, seed, values="counts", unsorted=False, empty_rows=False, dense_col=False,
         duplicates=False):
    rng = np.random.default_rng(seed)
    cols, ptr = [], [0]
    for b in range(rows):
        if empty_rows and b % 3 == 1:
            n = 0
        else:
            n = rng.binomial(D, density)
        c = rng.choice(D, size=n, replace=False)
        if dense_col and D > 2:
            c = np.union1d(c, [1])
        if not unsorted:
            c = np.sort(c)
        if duplicates and len(c) > 1:
            c = np.concatenate([c, c[:1]])
        cols.append(c)
        ptr.append(ptr[-1] + len(c))
    col = np.concatenate(cols).astype(np.int64) if cols else np.zeros(0, np.int64)
    nnz = len(col)
    if values == "counts":
        val = rng.poisson(2.0, nnz) + 1.0
    elif values == "big":
        val = rng.integers(1, 200_000, nnz).astype(np.float64)
    else:
        val = rng.gamma(2.0, 1.0, nnz) + 0.25
    return np.asarray(ptr, np.int64), col, val.astype(np.float32)


def _build(ptr, col, val, rows, D, P, split, native, latent_dim=None):
    from spmf_amd.sparse import SparseCounts
    dev = torch.device("cuda", 0)
    old = os.environ.get("SPMF_NATIVE_LAYOUT")
    os.environ["SPMF_NATIVE_LAYOUT"] = "1" if native else "0"
    try:
        return SparseCounts(torch.as_tensor(ptr).to(dev), torch.as_tensor(col).to(dev),
                            torch.as_tensor(val).to(dev), rows, D, P, col_split=split, latent_dim=latent_dim)
    finally:
        if old is None:
            del os.environ["SPMF_NATIVE_LAYOUT"]
        else:
            os.environ["SPMF_NATIVE_LAYOUT"] = old


ARRAYS = ("pc_ptr", "pc_row", "pc_val", "pc_ent", "ent", "items", "item_ptr", "item_mid",
          "items_per_panel", "items_per_half", "list_first", "item_pos")


def _same(a, b, what):
    assert a.native_layout and not b.native_layout
    for k in ("n_rows", "n_cols", "nnz", "panel_rows", "n_panels", "pc_pad", "col_split"):
        assert getattr(a, k) == getattr(b, k), (what, k)
    for k in ARRAYS:
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None), (what, k, x is None, y is None)
        if x is not None:
            assert x.dtype == y.dtype and x.shape == y.shape, (what, k, x.dtype, y.dtype, x.shape, y.shape)
            assert torch.equal(x, y), (what, k)
    # the descriptors the kernels get, for the whole shard and for a panel range
    for pr in ((0, None), (1, 3)):
        if pr[0] >= a.n_panels:
            continue
        ca, cb = a.batch_struct(*pr), b.batch_struct(*pr)
        for k in ("n_rows", "nnz", "n_panels", "max_items_per_panel", "pc_pad", "row_base", "n_items"):
            assert getattr(ca, k) == getattr(cb, k), (what, pr, k)
        assert list(ca.max_items_half) == list(cb.max_items_half)


CASES = [
    # rows, D, density, P, split, kwargs
    ("counts, 12 panels", 3000, 700, 0.02, 256, 0, {}),
    ("counts, column split", 3000, 700, 0.02, 256, 300, {}),
    ("one panel", 900, 300, 0.05, 4096, 0, {}),
    ("ragged last panel + empty rows", 1001, 257, 0.03, 100, 0, {"empty_rows": True}),
    ("unsorted columns inside rows", 800, 200, 0.05, 64, 0, {"unsorted": True}),
    ("a pair stored twice", 500, 100, 0.05, 64, 40, {"duplicates": True}),
    ("real values: nothing packs", 1500, 400, 0.03, 128, 0, {"values": "real"}),
    ("counts above 65535", 600, 300, 0.04, 128, 0, {"values": "big"}),
    ("long lists: several segments per list", 5000, 30, 0.5, 5000, 0, {"dense_col": True}),
    ("long lists, split", 6000, 24, 0.6, 3000, 10, {"dense_col": True}),
    ("D above 65536: no packed rows", 300, 70_000, 0.0005, 64, 0, {}),
    ("one row", 1, 50, 0.3, 8, 0, {}),
    ("one column", 200, 1, 0.6, 32, 0, {}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_native_layout_equals_the_torch_construction(case):
    what, rows, D, dens, P, split, kw = case
    ptr, col, val = _csr(rows, D, dens, seed=len(what) * 31 + rows, **kw)
    a = _build(ptr, col, val, rows, D, P, split, native=True)
    b = _build(ptr, col, val, rows, D, P, split, native=False)
    _same(a, b, what)


@pytest.mark.parametrize("K,finer", [(2, True), (7, True), (16, False)])
def test_latent_dim_hint_cuts_finer_items_for_small_k_and_both_builders_agree(K, finer):
    """spmf_layout_build_k: a work item is one lane group of the column pass (K padded / 4 lanes), so at K <= 8 the
    lists are cut for 16 384 / 32 768 items per panel instead of 4096 (the reference CLI's default K = 2 on a dense
    5000 x 200 batch had 2.5k one-lane items = 39 waves).  Native == torch with the hint; the energy does not
    depend on the cut (a C1-shaped batch against the unhinted layout)."""
    rows, D = 5000, 200
    rng = np.random.default_rng(5)
    X = rng.poisson(1.0, size=(rows, D)).astype(np.float32)
    mask = X != 0
    ptr = np.concatenate([[0], np.cumsum(mask.sum(1))]).astype(np.int64)
    col = np.nonzero(mask)[1].astype(np.int64)
    val = X[mask]
    a = _build(ptr, col, val, rows, D, rows, 0, native=True, latent_dim=K)
    b = _build(ptr, col, val, rows, D, rows, 0, native=False, latent_dim=K)
    _same(a, b, f"hint K={K}")
    plain = _build(ptr, col, val, rows, D, rows, 0, native=True)
    assert (a.segment < plain.segment) == finer and (a.items.shape[0] > plain.items.shape[0]) == finer
    if K == 2:
        from spmf_amd import PoissonFactorization
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-3, device="cuda")
        torch.manual_seed(1)
        params = m.surrogate_distribution.sample(1)
        pa, ga, _ = m.energy_and_grads({"counts": a}, params)
        pb, gb, _ = m.energy_and_grads({"counts": plain}, params)
        for k in pa:
            assert abs(float(pa[k][0]) - float(pb[k][0])) <= 1e-9 * max(1.0, abs(float(pb[k][0]))), k
        for k in ga:
            assert float((ga[k] - gb[k]).abs().max()) <= 1e-5 * float(gb[k].abs().max()), k


def test_panels_of_more_than_65536_rows_and_an_empty_shard():
    rows, D = 70_000, 40
    ptr, col, val = _csr(rows, D, 0.05, seed=5)
    a = _build(ptr, col, val, rows, D, rows, 0, native=True)
    b = _build(ptr, col, val, rows, D, rows, 0, native=False)
    assert a.pc_ent is None and a.ent is not None
    _same(a, b, "one 70 000-row panel")
    z = np.zeros(0, np.int64)
    for n in (0, 5):
        a = _build(np.zeros(n + 1, np.int64), z, z.astype(np.float32), n, 30, 8, 0, native=True)
        b = _build(np.zeros(n + 1, np.int64), z, z.astype(np.float32), n, 30, 8, 0, native=False)
        _same(a, b, f"no stored entry, {n} rows")


def test_builder_is_deterministic_and_feeds_the_kernels():
    """Two builds give the same bytes; energy and gradients through a native layout equal those
    through the torch-built one (same arrays, same kernels; float atomics aside)."""
    import contextlib
    import sys
    from spmf_amd import PoissonFactorization
    rows, D, K = 4000, 500, 8
    ptr, col, val = _csr(rows, D, 0.03, seed=77)
    a = _build(ptr, col, val, rows, D, 512, 0, native=True)
    a2 = _build(ptr, col, val, rows, D, 512, 0, native=True)
    for k in ARRAYS:
        if getattr(a, k) is not None:
            assert torch.equal(getattr(a, k), getattr(a2, k)), k
    b = _build(ptr, col, val, rows, D, 512, 0, native=False)
    dev = torch.device("cuda", 0)
    outs = []
    for sc in (a, b):
        with contextlib.redirect_stdout(sys.stderr):
            m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=0.01, device=dev)
        colsum = torch.zeros(D, dtype=torch.float64, device=dev)
        colnnz = torch.zeros_like(colsum)
        sc.compute_stats(m._handle(), colsum, colnnz)
        torch.manual_seed(5)
        p = m.surrogate_distribution.sample(1)
        parts, grads, _ = m.energy_and_grads({"counts": sc}, p)
        outs.append((float(parts["x"][0]), float(parts["z"][0]), grads))
    for i in (0, 1):        # (fp64 atomics: the order of the adds is not fixed)
        assert abs(outs[0][i] - outs[1][i]) <= 1e-11 * abs(outs[1][i])
    for k in ("u", "v", "w", "s"):      # float atomics: run-to-run noise of the column pass, array norm
        ga, gb = outs[0][2][k], outs[1][2][k]
        assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()), k


def test_builder_rejects_what_would_make_the_kernels_read_out_of_bounds():
    from spmf_amd import _lib
    rows, D = 200, 50
    ptr, col, val = _csr(rows, D, 0.1, seed=9)
    bad = col.copy()
    bad[7] = D                                   # a column index outside [0, D)
    with pytest.raises(_lib.SpmfError, match="column index"):
        _build(ptr, bad, val, rows, D, 64, 0, native=True)
    bad = col.copy()
    bad[3] = -1
    with pytest.raises(_lib.SpmfError, match="column index"):
        _build(ptr, bad, val, rows, D, 64, 0, native=True)
    p2 = ptr.copy()
    p2[10], p2[11] = p2[11], p2[10]              # decreasing offsets
    if p2[10] != p2[11]:
        with pytest.raises(_lib.SpmfError, match="row_ptr"):
            _build(p2, col, val, rows, D, 64, 0, native=True)
    # offsets that zig-zag AT the panel edges (panels of 64 rows: edges at rows 64 and 128): neighbouring
    # panels get overlapping / reversed entry ranges, list "lengths" below zero or of a whole panel -- the
    # item stage must not be driven by them (it wrote outside its scratch)
    for edit in ({64: ptr[128], 128: ptr[64]}, {64: ptr[192], 128: ptr[1]}, {64: 0, 128: ptr[-1], 192: 0}):
        p4 = ptr.copy()
        for i, v in edit.items():
            p4[i] = v
        with pytest.raises(_lib.SpmfError, match="row_ptr"):
            _build(p4, col, val, rows, D, 64, 0, native=True)
    _build(ptr, col, val, rows, D, 64, 0, native=True)      # and the builder still works afterwards
    p3 = ptr.copy()
    p3[-1] += 4                                  # row_ptr[n_rows] != nnz
    with pytest.raises(_lib.SpmfError, match="row_ptr"):
        _build(p3, col, val, rows, D, 64, 0, native=True)
    # buffers smaller than asked for, a foreign info struct
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    lb, sb = C.c_size_t(), C.c_size_t()
    assert lib.spmf_layout_sizes(0, rows, len(col), D, 64, C.byref(lb), C.byref(sb)) == 0
    lay = torch.empty(lb.value, dtype=torch.uint8, device=dev)
    scr = torch.empty(sb.value, dtype=torch.uint8, device=dev)
    rp = torch.as_tensor(ptr).to(dev).to(torch.int32)
    ci = torch.as_tensor(col).to(dev).to(torch.int32)
    va = torch.as_tensor(val).to(dev)
    cs, info = _lib.CountsStruct(), _lib.LayoutInfo()
    info.struct_size = C.sizeof(_lib.LayoutInfo)
    st = torch.cuda.current_stream(dev).cuda_stream
    args = (0, rows, len(col), D, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), 64, 0)
    assert lib.spmf_layout_build(*args, lay.data_ptr(), lb.value - 256, scr.data_ptr(), sb.value,
                                 C.byref(cs), C.byref(info), st) == -3
    assert b"smaller" in lib.spmf_layout_last_error()
    info.struct_size = 8
    assert lib.spmf_layout_build(*args, lay.data_ptr(), lb.value, scr.data_ptr(), sb.value,
                                 C.byref(cs), C.byref(info), st) == -1
    info.struct_size = C.sizeof(_lib.LayoutInfo)
    assert lib.spmf_layout_build(*args, lay.data_ptr(), lb.value, scr.data_ptr(), sb.value,
                                 C.byref(cs), C.byref(info), st) == 0
    assert cs.struct_size == C.sizeof(_lib.CountsStruct) and cs.nnz == len(col) and info.n_items > 0


def test_statistics_csr_form_list_form_and_host_numbers_agree():
    """compute_scales' column sums (poisson.py:118-135) and the per-row sums: the CSR form of
    spmf_counts_stats, the list form of spmf_counts_colstats and numpy on the same data."""
    import contextlib
    import sys
    from scipy.special import gammaln
    from spmf_amd import PoissonFactorization, _lib
    rows, D = 3000, 400
    ptr, col, val = _csr(rows, D, 0.04, seed=21)
    val[::7] = 300.0 + (np.arange(len(val[::7])) % 5)       # counts beyond the lgamma table
    val[5::11] = 0.0                                          # stored zeros: not counted by colnnz
    sc = _build(ptr, col, val, rows, D, 512, 0, native=True)
    dev = torch.device("cuda", 0)
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=4, feature_dim=D, u_tau_scale=0.01, device=dev)
    lib, h = _lib.load(), m._handle()
    cs_l = torch.zeros(D, dtype=torch.float64, device=dev)
    cn_l = torch.zeros_like(cs_l)
    sc.compute_stats(h, cs_l, cn_l)                           # rows: CSR kernel, columns: lists
    cs_c = torch.zeros_like(cs_l)
    cn_c = torch.zeros_like(cs_l)
    st = torch.cuda.current_stream(dev).cuda_stream
    assert lib.spmf_counts_stats(h, rows, sc.row_ptr.data_ptr(), sc.col_idx.data_ptr(), sc.val.data_ptr(),
                                 cs_c.data_ptr(), cn_c.data_ptr(), None, None, st) == 0
    torch.cuda.synchronize()
    want_sum = np.bincount(col, weights=val.astype(np.float64), minlength=D)
    want_nnz = np.bincount(col, weights=(val > 0).astype(np.float64), minlength=D)
    for got in (cs_l, cs_c):
        assert np.array_equal(got.cpu().numpy(), want_sum)    # integer counts: exact in fp64
    for got in (cn_l, cn_c):
        assert np.array_equal(got.cpu().numpy(), want_nnz)
    rs = np.add.reduceat(np.append(val.astype(np.float64), 0.0), ptr[:-1])
    rs[np.diff(ptr) == 0] = 0.0
    np.testing.assert_allclose(sc.row_sum.cpu().numpy(), rs, rtol=1e-6)
    lg = np.add.reduceat(np.append(gammaln(val.astype(np.float64) + 1.0), 0.0), ptr[:-1])
    lg[np.diff(ptr) == 0] = 0.0
    np.testing.assert_allclose(sc.row_lgamma.cpu().numpy(), lg, rtol=1e-13, atol=1e-12)


def test_log_transform_streams_native_equal_torch():
    """g(x) = log(x / eta_d + 1) (encoder_function, poisson.py:41-42) in CSR and list order:
    spmf_counts_gvals against the torch statement, and against numpy in fp64."""
    import contextlib
    import sys
    from spmf_amd import PoissonFactorization
    rows, D = 2500, 300
    ptr, col, val = _csr(rows, D, 0.05, seed=31, empty_rows=True)
    dev = torch.device("cuda", 0)
    eta = torch.as_tensor(np.random.default_rng(3).gamma(2.0, 1.0, D).astype(np.float32) + 0.1).to(dev)
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=4, feature_dim=D, u_tau_scale=0.01, device=dev, log_transform=True)
    a = _build(ptr, col, val, rows, D, 512, 0, native=True)
    b = _build(ptr, col, val, rows, D, 512, 0, native=True)
    a.set_log_transform(eta, m._handle())        # the library
    b.set_log_transform(eta)                     # torch operators
    assert a.gval.shape == b.gval.shape and a.pc_gval.shape == b.pc_gval.shape
    torch.testing.assert_close(a.gval, b.gval, rtol=3e-7, atol=0)
    torch.testing.assert_close(a.pc_gval, b.pc_gval, rtol=3e-7, atol=0)
    assert float(a.pc_gval[a.nnz:].abs().max()) == 0.0            # the padding behind the last list
    want = np.log1p(val.astype(np.float64) / eta.cpu().numpy().astype(np.float64)[col])
    np.testing.assert_allclose(a.gval.cpu().numpy(), want, rtol=3e-7)


@pytest.mark.parametrize("shape", [(700, 333), (1, 5), (2049, 64), (5000, 1), (3, 70_000), (0, 7)])
def test_dense_batches_to_csr_native_equals_torch_and_numpy(shape):
    """The reference's own batch format (dense [B,D], tests/spmf_test.py:17-22) through the library's
    compaction kernels: same CSR as numpy's nonzero(), same layout as the torch path."""
    from spmf_amd.sparse import SparseCounts
    rows, D = shape
    rng = np.random.default_rng(rows * 7 + D)
    x = rng.poisson(0.05 if D > 1000 else 0.4, size=(rows, D)).astype(np.float32)
    if shape == (1, 5):
        x[:] = 0.0                                  # a batch without a stored cell
    if rows > 2:
        x[1] = 0.0                                  # an empty row
        x[2, :] = 3.0                               # a full row
    dev = torch.device("cuda", 0)
    a = SparseCounts.from_dense(x, dev, 256)
    old = os.environ.get("SPMF_NATIVE_LAYOUT")
    os.environ["SPMF_NATIVE_LAYOUT"] = "0"
    try:
        b = SparseCounts.from_dense(x, dev, 256)
    finally:
        if old is None:
            del os.environ["SPMF_NATIVE_LAYOUT"]
        else:
            os.environ["SPMF_NATIVE_LAYOUT"] = old
    r, c = np.nonzero(x)
    assert a.nnz == len(r) == b.nnz
    assert np.array_equal(a.row_ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))]))
    assert np.array_equal(a.col_idx.cpu().numpy(), c) and np.array_equal(a.val.cpu().numpy(), x[r, c])
    for k in ("row_ptr", "col_idx", "val"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    if rows > 0:
        _same(a, b, f"dense {shape}")
    # a strided view (leading dimension > D) and NaN cells, through the C entry points
    if rows >= 3 and D >= 5:
        from spmf_amd import _lib
        lib = _lib.load()
        big = torch.as_tensor(x).to(dev)
        big[0, 1] = float("nan")
        wide = torch.zeros(rows, D + 3, dtype=torch.float32, device=dev)
        wide[:, :D] = big
        rp = torch.empty(rows + 1, dtype=torch.int32, device=dev)
        scr = torch.empty(int(lib.spmf_dense_scratch_bytes(rows)) // 8 + 1, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        assert lib.spmf_dense_row_ptr(0, rows, D, wide.data_ptr(), D + 3, rp.data_ptr(), scr.data_ptr(),
                                      scr.numel() * 8, st) == 0
        n = int(rp[rows])
        ci = torch.empty(n, dtype=torch.int32, device=dev)
        va = torch.empty(n, dtype=torch.float32, device=dev)
        assert lib.spmf_dense_fill_csr(0, rows, D, wide.data_ptr(), D + 3, rp.data_ptr(), ci.data_ptr(),
                                       va.data_ptr(), st) == 0
        xb = big.cpu().numpy()
        rr, cc = np.nonzero((xb != 0) | np.isnan(xb))
        assert n == len(rr) and np.array_equal(ci.cpu().numpy(), cc)
        assert np.array_equal(va.cpu().numpy(), xb[rr, cc], equal_nan=True)


def test_torch_sparse_tensors_are_taken_where_they_are():
    """torch.sparse_csr / sparse_coo inputs (device resident) give the layout of the same matrix from scipy."""
    import scipy.sparse as sp
    from spmf_amd.sparse import SparseCounts
    rng = np.random.default_rng(3)
    X = sp.random(700, 300, density=0.03, format="csr", random_state=rng, data_rvs=lambda n: rng.poisson(2.0, n) + 1.0)
    X.sort_indices()
    dev = torch.device("cuda", 0)
    ref = SparseCounts.from_any(X, dev, 128)
    t_csr = torch.sparse_csr_tensor(torch.as_tensor(X.indptr, dtype=torch.int64), torch.as_tensor(X.indices, dtype=torch.int64),
                                    torch.as_tensor(X.data, dtype=torch.float32), size=X.shape).to(dev)
    coo = X.tocoo()
    t_coo = torch.sparse_coo_tensor(torch.as_tensor(np.stack([coo.row, coo.col])), torch.as_tensor(coo.data, dtype=torch.float32),
                                    size=X.shape).to(dev)
    for t in (t_csr, t_coo):
        sc = SparseCounts.from_any(t, dev, 128)
        for k in ("row_ptr", "col_idx", "val") + ARRAYS:
            a, b = getattr(sc, k), getattr(ref, k)
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), k
