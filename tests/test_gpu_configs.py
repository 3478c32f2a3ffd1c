"""GPU parity at the BASELINE.json config shapes.

C1 (5k x 200 dense Poisson(1), K=2, B=5000): against the dense fp64 oracle.
C2 (one 20000 x 5000 batch, 1 % nnz, K=16): against the scipy sparse-exact
    port (itself pinned to the dense oracle in tests/test_oracle.py).
C3 (1M x 20k, nnz ~1e8, K=32, full size): size-independent properties --
    additivity over row shards, the z-prior identity from encode(), the
    closed-form sum over implicit zeros against an independent dense-GEMV
    evaluation, and finiteness/sign of every part.
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import spmf_oracle as O
from oracle import sparse_exact as SE

pytestmark = pytest.mark.gpu


def _close_grads(got, ref, tol=1e-5):
    for k, r in ref.items():
        g = got[k].detach().cpu().double().numpy().reshape(np.shape(r))
        r = np.asarray(r)
        assert np.abs(g - r).max() <= tol * np.abs(r).max(), k


def test_c1_dense_poisson_noise_K2():
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(20241218 + 1)
    N, D, K = 5000, 200, 2
    x = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / math.sqrt(N * D))
    O.compute_scales(cfg, [x])
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             device="cuda")
    m.compute_scales(lambda: [{"counts": x}])
    np.testing.assert_allclose(m.eta_i.cpu().numpy(), cfg.eta_i.numpy(), rtol=1e-12)
    assert abs(m.xi_u_global - float(cfg.xi_u_global)) < 1e-9 * float(cfg.xi_u_global)
    torch.manual_seed(3)
    params = {k: v.double().cpu().numpy() for k, v in m.surrogate_distribution.sample(2).items()}
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, err_msg=k)
    _close_grads(grads, {k: v.numpy() for k, v in gref.items()})


def test_c2_batch_20000x5000_K16_vs_sparse_exact():
    from spmf_amd import PoissonFactorization, synth
    B, D, K = 20000, 5000, 16
    sc = synth.bernoulli_poisson(B, D, 0.01, torch.device("cuda"), 20241218 + 2)
    assert 0.8e6 < sc.nnz < 1.2e6
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(1e5 * D),
                             device="cuda")
    m.compute_scales(lambda: [{"counts": sc}])
    torch.manual_seed(5)
    params = m.surrogate_distribution.sample(1)
    # move away from the near-degenerate initial point so every term matters
    params["u"] = params["u"] * 40
    params["v"] = params["v"] * 40
    parts, grads, nnf = m.energy_and_grads({"counts": sc}, params)
    assert float(nnf.sum()) == 0
    X = sp.csr_matrix((sc.val.cpu().numpy().astype(np.float64), sc.col_idx.cpu().numpy(),
                       sc.row_ptr.cpu().numpy()), shape=(B, D))
    one = {k: v[0].double().cpu().numpy() for k, v in params.items()}
    eta = m._eta_device().double().cpu().numpy()
    ref = SE.data_term(X, eta, float(m.xi_u_global), True, one["u"], one["v"], one["w"], one["s"])
    np.testing.assert_allclose(parts["x"][0].item(), ref["x"], rtol=1e-5)
    np.testing.assert_allclose(parts["z"][0].item(), ref["z"], rtol=1e-5)
    pparts, pg = SE.prior_term(one, m.u_tau_scale, m.s_tau_scale,
                               m.symmetry_breaking_decay ** np.arange(K))
    for k, r in pparts.items():
        np.testing.assert_allclose(parts[k][0].item(), r, rtol=1e-5, err_msg=k)
    tot = {k: pg[k] + ref["grads"].get(k, 0.0) for k in pg}
    _close_grads({k: v[0] for k, v in grads.items()}, tot)


@pytest.mark.timeout(600)
def test_c3_full_size_properties():
    from spmf_amd import PoissonFactorization, synth
    dev = torch.device("cuda")
    N, D, K = 1_000_000, 20_000, 32
    sc = synth.linear_structure(N, D, 0.005, dev)
    assert 0.9e8 < sc.nnz < 1.1e8
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device=dev)
    m.compute_scales(lambda: [{"counts": sc}])
    torch.manual_seed(7)
    params = m.surrogate_distribution.sample(1)
    params["u"] = params["u"] * 20
    params["v"] = params["v"] * 20
    parts, grads, nnf = m.energy_and_grads({"counts": sc}, params)
    assert float(nnf.sum()) == 0
    for k, v in parts.items():
        assert bool(torch.isfinite(v).all()), k
    assert float(parts["x"]) < 0 and float(parts["z"]) < 0
    for k, g in grads.items():
        assert bool(torch.isfinite(g).all()), k
    # (1) z-prior identity from an independent encode() call
    z = m.encode(sc, u=params["u"][0], s=params["s"][0]).double()
    zref = N * K * 0.5 * math.log(2 / math.pi) - 0.5 * float((z * z).sum())
    assert abs(float(parts["z"]) - zref) <= 1e-5 * abs(zref)
    # (2) the closed-form sum of the rate over ALL cells == a dense evaluation
    #     sum_b <z_b, sum_d eta_d v_d> + N sum_d phi_d done with torch in fp64
    eta = m._eta_device().double()
    v = params["v"][0].double()
    s = params["s"][0].double()
    w = params["w"][0].double().reshape(-1)
    veta = (v * eta[None, :]).sum(1)
    phi = eta * (s[1] / (s[0] + s[1])) * w
    sum_r = float(z.sum(0) @ veta) + N * float(phi.sum())
    lg = float(sc.row_lgamma.sum())
    ll_nnz = float(parts["x"]) + lg + sum_r          # = sum_nnz x log r, must be reproducible:
    rows = torch.repeat_interleave(torch.arange(N, device=dev),
                                   (sc.row_ptr[1:] - sc.row_ptr[:-1]).long())
    idx = torch.arange(0, sc.nnz, 997, device=dev)   # a strided sample of stored cells
    r = (z[rows[idx]] * (v * eta[None, :]).T[sc.col_idx[idx].long()]).sum(1) + phi[sc.col_idx[idx].long()]
    assert bool((r > 0).all())
    est = float((sc.val[idx].double() * torch.log(r)).sum()) * 997
    assert abs(est - ll_nnz) <= 0.02 * abs(ll_nnz)   # sampled estimate: 2 %
    # (3) additivity over row shards: two halves accumulated == whole
    half = (sc.n_panels // 2)
    other = {}

    def hook0(acc, rows_, lg_):
        other["acc"], other["rows"], other["lg"] = acc.clone(), rows_, lg_
        return None
    m.energy_and_grads({"counts": sc, "panels": (0, half)}, params, all_reduce=hook0)

    def hook1(acc, rows_, lg_):
        acc += other["acc"]
        return rows_ + other["rows"], lg_ + other["lg"]
    p2, g2, _ = m.energy_and_grads({"counts": sc, "panels": (half, sc.n_panels)}, params,
                                   all_reduce=hook1)
    for k in parts:
        assert abs(float(p2[k]) - float(parts[k])) <= 1e-5 * abs(float(parts[k])), k
    for k in grads:
        a, b = g2[k].double(), grads[k].double()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), k
