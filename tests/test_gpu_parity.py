"""GPU parity: the HIP path through the C-ABI vs the fp64 CPU oracle on the
same seeded inputs.  Tolerance (north_star): 1e-5 relative, fp32 kernels.
Gradients are compared ENTRY by entry against 1e-5 of the sum of the absolute
contributions to that entry (tests/_gradcheck.py, oracle.energy_grad_scales):
entries of a sparse gradient cancel to ~0, so the yardstick is what was added up."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from _gradcheck import assert_grads_entrywise

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def make_problem(B, D, K, S, seed, density, scale_rows=True, empty=True, xmax=2.0):
    rng = np.random.default_rng(seed)
    mask = rng.random((B, D)) < density
    x = (mask * (1 + rng.poisson(xmax, size=(B, D)))).astype(np.float64)
    if empty and B > 4 and D > 4:
        x[1, :] = 0.0
        x[:, 2] = 0.0
        x[B - 1, :] = 0.0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows,
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(2.0, 6.0))
    params = O.random_params(cfg, S, seed + 1, fp32_exact=True)
    return cfg, x, params


def build_model(cfg, panel_rows=64):
    from spmf_amd import PoissonFactorization
    m = PoissonFactorization(
        latent_dim=cfg.latent_dim, feature_dim=cfg.feature_dim,
        u_tau_scale=cfg.u_tau_scale, s_tau_scale=cfg.s_tau_scale,
        symmetry_breaking_decay=cfg.symmetry_breaking_decay,
        scale_rows=cfg.scale_rows, log_transform=cfg.log_transform, column_norms=cfg.eta_i,
        initialize_distributions=False, device="cuda", panel_rows=panel_rows)
    m.xi_u_global = cfg.xi_u_global
    return m


def assert_close_parts(got, ref, rtol=RTOL):
    for k in ref:
        g = got[k].detach().cpu().numpy()
        r = ref[k].numpy()
        np.testing.assert_allclose(g, r, rtol=rtol, atol=rtol, err_msg=f"part {k}")


def assert_close_grads(got, ref, scales, rtol=RTOL, tag=""):
    """Entry-wise: |hip - oracle| <= rtol * sum of |contributions| to that entry."""
    assert_grads_entrywise(got, ref, scales, rtol, tag)


CASES = [
    # B,   D,   K,  S, density, scale_rows, panel_rows
    (37, 23, 3, 2, 0.3, True, 16),       # ragged, K padded 3->4
    (64, 40, 2, 1, 0.63, True, 64),      # config-1-like density, K=2
    (200, 150, 16, 2, 0.05, True, 64),   # config-2-like K
    (300, 257, 32, 1, 0.04, True, 128),  # config-3-like K
    (300, 257, 32, 2, 0.04, False, 128),  # scale_rows off
    (150, 90, 50, 1, 0.1, True, 32),     # reference test's P=50 -> KP=64
    (90, 70, 64, 1, 0.2, True, 1000),    # K=64, single panel
    (130, 33, 8, 3, 0.9, True, 16),      # near dense, rows > 64 nnz? no: D=33
    (50, 300, 4, 1, 0.8, True, 16),      # long rows (240 nnz) -> several chunks
]


@pytest.mark.parametrize("B,D,K,S,density,scale_rows,panel_rows", CASES)
def test_energy_parts_and_grads_match_oracle(B, D, K, S, density, scale_rows, panel_rows):
    cfg, x, params = make_problem(B, D, K, S, 1000 + B + D + K, density, scale_rows)
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, panel_rows)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))


@pytest.mark.parametrize("K,log_transform", [(32, False), (16, False), (64, False), (64, True)])
def test_row_lengths_around_the_chunk_boundaries(K, log_transform):
    """Rows of exactly 0, 1, 63 ... 257, 300 stored entries in a batch big enough for the
    resident-set launches (B >= 4096): the short-row path keeps two 64-entry chunks in
    registers, longer rows stream their chunks two ahead (row_pass.hip)."""
    B, D, S = 4200, 300, 1
    lengths = [0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300]
    rng = np.random.default_rng(77 + K)
    x = np.zeros((B, D))
    for b in range(B):
        n = lengths[b % len(lengths)]
        cols = rng.choice(D, size=n, replace=False)
        x[b, cols] = 1 + rng.poisson(1.5, size=n)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=True, log_transform=log_transform,
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(2.0, 6.0))
    params = O.random_params(cfg, S, 78 + K, fp32_exact=True)
    if log_transform:   # keep exp(<z, eta v>) in range: the largest exponent is 8
        T = torch.as_tensor
        z = O.encode(cfg, T(x), T(params["u"]), T(params["s"]))
        params["v"] *= 8.0 / float((torch.matmul(z, T(params["v"])) * cfg.eta_i).max())
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, 1024)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))


@pytest.mark.parametrize("kind", ["fractional", "large"])
def test_counts_outside_the_packed_format_take_the_canonical_streams(kind, monkeypatch):
    """spmf_counts.ent / pc_ent (col << 16 | count) exist only for integer counts below 65536:
    real-valued entries (the reference casts whatever it is given, poisson.py:43,182) or one
    count of 70 000 keep the col / val arrays -- same kernels' canonical instances, B >= 4096 so
    that the resident-set row launch and the four-per-lane column fetch are the ones that run."""
    from spmf_amd.sparse import SparseCounts
    monkeypatch.delenv("SPMF_PACKED_ENTRIES", raising=False)   # (a run of the suite with packing switched off)
    B, D, K, S = 4200, 300, 32, 1
    cfg, x, params = make_problem(B, D, K, S, 321, 0.2)
    if kind == "fractional":
        x = x * 0.5
    else:
        x[7, 11] = 70000.0
    sc = SparseCounts.from_any(x, "cuda", 1024)
    assert sc.ent is None and sc.pc_ent is None
    ok = SparseCounts.from_any(np.round(np.minimum(x, 100.0)), "cuda", 1024)
    assert ok.ent is not None and ok.pc_ent is not None
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, 1024)
    parts, grads, nnf = m.energy_and_grads({"counts": sc}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))


@pytest.mark.parametrize("K,log_transform", [(32, False), (64, True)])
def test_packed_and_canonical_entry_streams_agree_bit_for_bit(K, log_transform, monkeypatch):
    """The packed words (spmf_counts.ent / pc_ent) carry exactly what col / val and pc_row /
    pc_val carry, and the kernels' two instances do the same arithmetic: the results agree up to
    the order of the atomic sums (fp64 block sums of the row pass, float column sums)."""
    from spmf_amd.sparse import SparseCounts
    B, D, S = 4300, 257, 1
    cfg, x, params = make_problem(B, D, K, S, 99 + K, 0.25)
    cfg.log_transform = log_transform
    if log_transform:
        T = torch.as_tensor
        z = O.encode(cfg, T(x), T(params["u"]), T(params["s"]))
        params["v"] *= 8.0 / float((torch.matmul(z, T(params["v"])) * cfg.eta_i).max())
    monkeypatch.delenv("SPMF_PACKED_ENTRIES", raising=False)
    packed = SparseCounts.from_any(x, "cuda", 512)
    monkeypatch.setenv("SPMF_PACKED_ENTRIES", "0")
    canon = SparseCounts.from_any(x, "cuda", 512)
    monkeypatch.delenv("SPMF_PACKED_ENTRIES")
    assert packed.ent is not None and packed.pc_ent is not None and canon.ent is None and canon.pc_ent is None
    m = build_model(cfg, 512)
    pa, ga, _ = m.energy_and_grads({"counts": packed}, params)
    pb, gb, _ = m.energy_and_grads({"counts": canon}, params)
    for k in ("x", "z"):   # fp64 block sums added with atomics: equal up to their order
        np.testing.assert_allclose(pa[k].cpu().numpy(), pb[k].cpu().numpy(), rtol=1e-12, err_msg=k)
    for k in ga:
        a, b = ga[k].double(), gb[k].double()
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), k


def test_prior_weight_scales_only_prior_gradient():
    cfg, x, params = make_problem(60, 40, 8, 1, 7, 0.2)
    _, _, groups = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg)
    _, grads, _ = m.energy_and_grads({"counts": x}, params, prior_weight=0.25)
    ref = {k: groups["data"][k] + 0.25 * groups["prior"][k] for k in groups["data"]}
    assert_close_grads(grads, ref, O.energy_grad_scales(cfg, x, params, prior_weight=0.25))


def test_unormalized_log_prob_parts_surface():
    cfg, x, params = make_problem(40, 30, 4, 2, 3, 0.3)
    ref = O.unormalized_log_prob_parts(cfg, x, params)
    m = build_model(cfg)
    got = m.unormalized_log_prob_parts({"counts": torch.as_tensor(x)}, **params)
    assert set(got) == set(ref)
    assert_close_parts(got, ref)
    tot = m.unormalized_log_prob(data={"counts": x}, prior_weight=0.3, **params)
    np.testing.assert_allclose(tot.cpu().numpy(), O.unormalized_log_prob(cfg, x, params).numpy(),
                               rtol=RTOL)
    # un-batched parameters (no sample axis) give scalars, like the reference
    one = {k: v[0] for k, v in params.items()}
    got1 = m.unormalized_log_prob_parts({"counts": x}, **one)
    assert got1["x"].dim() == 0
    np.testing.assert_allclose(got1["x"].item(), ref["x"][0].item(), rtol=RTOL)


def test_encode_and_matrices_match_oracle():
    cfg, x, params = make_problem(70, 45, 5, 2, 11, 0.25)
    m = build_model(cfg)
    T = lambda a: torch.as_tensor(a)
    z_ref = O.encode(cfg, T(x), T(params["u"]), T(params["s"])).numpy()
    z = m.encode(x, u=T(params["u"]), s=T(params["s"])).cpu().numpy()
    np.testing.assert_allclose(z, z_ref, rtol=RTOL, atol=RTOL * np.abs(z_ref).max())
    A = m.encoding_matrix(T(params["u"]), T(params["s"])).cpu().numpy()
    np.testing.assert_allclose(A, O.encoding_matrix(T(params["u"]), T(params["s"])).numpy(),
                               rtol=1e-12)
    phi = m.intercept_matrix(T(params["w"]).cuda(), T(params["s"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(phi, O.intercept_matrix(cfg, T(params["w"]), T(params["s"])).numpy(),
                               rtol=1e-6)


def test_compute_scales_matches_oracle():
    rng = np.random.default_rng(5)
    x = (rng.random((500, 60)) < 0.2) * (1 + rng.poisson(3.0, size=(500, 60)))
    x = x.astype(np.float64)
    x[:, 7] = 0
    cfg = O.OracleConfig(latent_dim=3, feature_dim=60)
    from spmf_amd import PoissonFactorization
    m = PoissonFactorization(latent_dim=3, feature_dim=60, initialize_distributions=False,
                             device="cuda")
    batches = [x[:200], x[200:350], x[350:]]
    m.compute_scales(lambda: [{"counts": b} for b in batches])
    xs = np.delete(x, 7, axis=1)   # oracle xi is NaN with an empty column; compare without it
    O.compute_scales(cfg, [xs])
    eta = m.eta_i.cpu().numpy().reshape(-1)
    np.testing.assert_allclose(np.delete(eta, 7), cfg.eta_i.numpy().reshape(-1), rtol=1e-12)
    assert eta[7] == 1.0
    assert abs(m.xi_u_global - float(cfg.xi_u_global)) < 1e-9 * float(cfg.xi_u_global)


def test_minibatch_panels_equal_separate_batches():
    cfg, x, params = make_problem(256, 50, 8, 1, 21, 0.15, empty=False)
    m = build_model(cfg, panel_rows=64)
    from spmf_amd import SparseCounts
    sc = SparseCounts.from_any(x, "cuda", 64)
    for p0, p1 in [(0, 1), (1, 3), (3, 4)]:
        xb = x[p0 * 64:p1 * 64]
        pref, gref, _ = O.energy_and_grads(cfg, xb, params)
        parts, grads, _ = m.energy_and_grads({"counts": sc, "panels": (p0, p1)}, params)
        assert_close_parts(parts, pref)
        assert_close_grads(grads, gref, O.energy_grad_scales(cfg, xb, params))


def test_split_data_pass_allreduce_hook_sums_shards():
    """Two row shards processed on one GPU; accumulators summed by hand in the
    all_reduce hook == the unsharded batch (the N>1 path without RCCL)."""
    cfg, x, params = make_problem(128, 40, 4, 2, 33, 0.2, empty=False)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m0, m1 = build_model(cfg, 32), build_model(cfg, 32)
    other = {}

    def hook0(acc, rows, lg):
        other["acc"], other["rows"], other["lg"] = acc.clone(), rows, lg
        return None
    m0.energy_and_grads({"counts": x[:64]}, params, all_reduce=hook0)

    def hook1(acc, rows, lg):
        acc += other["acc"]
        return rows + other["rows"], lg + other["lg"]
    parts, grads, _ = m1.energy_and_grads({"counts": x[64:]}, params, all_reduce=hook1)
    assert_close_parts(parts, pref)
    assert_close_grads(grads, gref, O.energy_grad_scales(cfg, x, params))


def test_errors_are_loud():
    from spmf_amd import PoissonFactorization
    from spmf_amd._lib import SpmfError
    with pytest.raises(SpmfError):
        m = PoissonFactorization(latent_dim=300, feature_dim=10, initialize_distributions=False,
                                 device="cuda")
        m._handle()
    cfg, x, params = make_problem(20, 10, 2, 1, 1, 0.5, empty=False)
    m = build_model(cfg)
    with pytest.raises(ValueError):
        m.energy_and_grads({"counts": x[:, :5]}, params)
    bad = dict(params)
    bad["u"] = bad["u"][:, :, :1]
    with pytest.raises(ValueError):
        m.energy_and_grads({"counts": x}, bad)


def test_randomised_shapes_sweep():
    """40 random (B, D, K, S, density, scale_rows, panel_rows) draws, incl. K
    not a power of two, single rows/columns, all-zero batches and rows longer
    than two 64-entry chunks."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        B = int(rng.integers(1, 180))
        D = int(rng.integers(1, 400))
        K = int(rng.integers(1, 65))
        S = int(rng.integers(1, 4))
        density = float(rng.choice([0.0, 0.02, 0.2, 0.7, 1.0]))
        scale_rows = bool(rng.integers(0, 2))
        panel_rows = int(rng.choice([1, 7, 64, 10_000]))
        cfg, x, params = make_problem(B, D, K, S, 5000 + case, density, scale_rows, empty=False)
        pref, gref, _ = O.energy_and_grads(cfg, x, params)
        m = build_model(cfg, panel_rows)
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        tag = f"case {case}: B={B} D={D} K={K} S={S} dens={density} sr={scale_rows} P={panel_rows}"
        assert float(nnf.sum()) == 0, tag
        for k in pref:
            np.testing.assert_allclose(parts[k].cpu().numpy(), pref[k].numpy(), rtol=RTOL,
                                       atol=RTOL, err_msg=f"{tag} part {k}")
        assert_close_grads(grads, gref, O.energy_grad_scales(cfg, x, params), tag=tag)


def test_c_abi_argument_errors():
    """The C entry points validate their arguments and report through
    spmf_last_error instead of launching on bad pointers."""
    import ctypes as C
    from spmf_amd import _lib
    cfg, x, params = make_problem(20, 10, 2, 1, 1, 0.5, empty=False)
    m = build_model(cfg)
    lib, h = _lib.load(), m._handle()
    # finish without a data pass for this S
    S, P = m._pack_params(params)
    pin = _lib.PtrArray(*[P[n].data_ptr() for n in _lib.VAR_ORDER])
    g = {n: torch.empty_like(P[n]) for n in _lib.VAR_ORDER}
    gout = _lib.PtrArray(*[g[n].data_ptr() for n in _lib.VAR_ORDER])
    parts = torch.empty(S, _lib.NPARTS, dtype=torch.float64, device="cuda")
    eta = m._eta_device()
    rc = lib.spmf_finish(h, S, 20, 0.0, 1.0, pin, eta.data_ptr(), parts.data_ptr(), gout, None, None)
    assert rc != 0 and b"data pass" in lib.spmf_last_error(h)
    # surrogate: noise stride shorter than the variable
    sur = m.surrogate_distribution if m.surrogate_distribution is not None else None
    if sur is None:
        m.create_distributions()
        sur = m.surrogate_distribution
    noise = sur.draw_noise(2)
    theta = {n: torch.empty(noise[n][0].shape, dtype=torch.float32, device="cuda")
             for n in _lib.VAR_ORDER}
    arr = sur._table(2, noise, theta=theta)
    arr[0].noise_ld = 1
    logq = torch.empty(2, dtype=torch.float64, device="cuda")
    rc = lib.spmf_surrogate_fwd(h, arr, len(_lib.VAR_ORDER), 2, logq.data_ptr(), None)
    assert rc != 0 and b"noise_ld" in lib.spmf_last_error(h)
    # null state for the device-side optimiser
    assert lib.spmf_vi_gate(h, parts.data_ptr(), logq.data_ptr(), None, 1, 1.0, 20.0, None, None) != 0
    torch.cuda.synchronize()


class _TwoShardSplitReducer:
    """Stand-in for dist.ShardReducer in the column-split flow: shard 0 records
    its two accumulator ranges, shard 1 adds them (a 2-rank all-reduce by hand)."""

    def __init__(self, store, first):
        self.store, self.first, self.k = store, first, 0

    def start(self, piece):
        if self.first:
            self.store.setdefault("pieces", []).append(piece.clone())
        else:
            piece += self.store["pieces"][self.k]
        self.k += 1
        return None

    def wait(self, work):
        pass

    def totals(self, rows, lg):
        if self.first:
            self.store["rows"], self.store["lg"] = rows, lg
            return None
        return rows + self.store["rows"], lg + self.store["lg"]

    def __call__(self, acc, rows, lg):          # unsplit fallback must not be taken here
        raise AssertionError("column-split flow expected")


@pytest.mark.parametrize("K", [4, 32, 100])
def test_column_split_step_equals_unsplit(K):
    """spmf_ctx_set_column_split + spmf_data_pass_split: two row shards, each
    all-reducing its accumulators in two column-half ranges == the oracle on the
    whole batch; and the split layout with the plain hook-less call is unchanged."""
    cfg, x, params = make_problem(128, 100, K, 1, 35, 0.2, empty=False)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    store = {}
    m0, m1 = build_model(cfg, 32), build_model(cfg, 32)
    for m in (m0, m1):
        assert m.enable_column_split(64) == 64
    m0.energy_and_grads({"counts": x[:64]}, params, all_reduce=_TwoShardSplitReducer(store, True))
    assert len(store["pieces"]) == 2
    parts, grads, _ = m1.energy_and_grads({"counts": x[64:]}, params,
                                          all_reduce=_TwoShardSplitReducer(store, False))
    scales = O.energy_grad_scales(cfg, x, params)
    assert_close_parts(parts, pref)
    assert_close_grads(grads, gref, scales)
    # same model, no reducer: the split layout serves the single-device call too
    parts1, grads1, _ = m1.energy_and_grads({"counts": x}, params)
    assert_close_parts(parts1, pref)
    assert_close_grads(grads1, gref, scales)
    with pytest.raises(ValueError):
        m1.enable_column_split(50)


@pytest.mark.parametrize("K", [3, 16, 32, 64])
def test_column_pass_without_list_padding_uses_the_narrow_fetch(K):
    """spmf_counts.pc_pad = 0 (a C-ABI caller whose panel-CSC arrays carry no padding): the library
    must fall back to the entry-at-a-time fetch, which never reads behind a list, and give the same
    gradients as the 16-byte fetches (both against the oracle, entry by entry)."""
    from spmf_amd import SparseCounts
    cfg, x, params = make_problem(300, 150, K, 1, 4400 + K, 0.25, empty=True)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    scales = O.energy_grad_scales(cfg, x, params)
    out = {}
    for pad in (None, 0):
        m = build_model(cfg, panel_rows=64)
        sc = SparseCounts.from_any(x, "cuda", 64)
        if pad is not None:
            sc.pc_pad = pad
        parts, grads, nnf = m.energy_and_grads({"counts": sc}, params)
        assert float(nnf.sum()) == 0
        assert_close_parts(parts, pref)
        assert_close_grads(grads, gref, scales, tag=f"pc_pad={pad}")
        out[pad] = grads
    for k in ("u", "v", "w", "s"):
        a, b = out[None][k].double(), out[0][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * float(scales[k].max())
