"""GPU parity of BernoulliFactorization (mederrata_spmf/bernoulli.py): sparse
x*logit terms + dense f32-MFMA softplus/sigmoid sums vs the fp64 oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from _gradcheck import assert_grads_entrywise

pytestmark = pytest.mark.gpu


def problem(B, D, K, S, seed, density):
    rng = np.random.default_rng(seed)
    x = (rng.random((B, D)) < density).astype(np.float64)
    if B > 4 and D > 4:
        x[1, :] = 0
        x[:, 2] = 0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=False, likelihood="bernoulli",
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 2.0, size=(1, D)))
    params = O.random_params(cfg, S, seed + 1, fp32_exact=True)
    params["v"] = params["v"] * rng.choice([-1.0, 1.0], size=params["v"].shape)   # Identity bijector
    params["w"] = -3.0 * params["w"]
    return cfg, x, params


@pytest.mark.parametrize("bf16x3", ["1", "0"])
@pytest.mark.parametrize("B,D,K,S,density", [(37, 23, 3, 2, 0.3), (150, 90, 8, 1, 0.1),
                                             (260, 200, 32, 1, 0.05), (300, 129, 64, 2, 0.05),
                                             (1500, 700, 20, 1, 0.02)])
def test_bernoulli_energy_and_grads(monkeypatch, B, D, K, S, density, bf16x3):
    """Both dense paths (read at spmf_ctx_create): the bf16x3 sigmoid kernels with the fused row pass
    (default at K <= 32) and the exact-f32 MFMA kernels; the last case spans several Q tiles and chunks."""
    from spmf_amd import BernoulliFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    cfg, x, params = problem(B, D, K, S, 700 + B + K, density)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                               column_norms=cfg.eta_i, device="cuda", panel_rows=64)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                   err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")


def problem_logt(B, D, K, S, seed, density):
    """log_transform=True (bernoulli.py:49-50,60-61): g(x) = log(x/eta + 1), logit = exp(<z, eta v>) - 1 + phi."""
    cfg, x, params = problem(B, D, K, S, seed, density)
    cfg.log_transform = True
    T = torch.as_tensor
    z = O.encode(cfg, T(x), T(params["u"]), T(params["s"]))
    ymax = float((torch.matmul(z, T(params["v"])) * cfg.eta_i).abs().max())
    params["v"] *= 3.0 / max(ymax, 1e-30)          # exponents within [-3, 3]
    return cfg, x, params


@pytest.mark.parametrize("bf16x3", ["1", "0"])
@pytest.mark.parametrize("B,D,K,S,density", [(37, 23, 3, 2, 0.3), (150, 90, 8, 1, 0.1),
                                             (260, 200, 32, 1, 0.05), (300, 129, 64, 2, 0.05),
                                             (1300, 700, 20, 1, 0.03)])
def test_bernoulli_log_transform_energy_and_grads(monkeypatch, B, D, K, S, density, bf16x3):
    """Both dense paths (read at spmf_ctx_create): K <= 32 on the bf16x3 kernels (dense3.hip sigdot3, ACT 2),
    and the exact-f32 kernels; K = 64 runs the f32 kernels either way."""
    from spmf_amd import BernoulliFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    cfg, x, params = problem_logt(B, D, K, S, 1700 + B + K, density)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                               column_norms=cfg.eta_i, log_transform=True, device="cuda", panel_rows=64)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                   err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")
    # per-cell outputs (bernoulli.py:126-155): 'rate' is the logit
    T = torch.as_tensor
    got = m.log_likelihood_components(s=T(params["s"]), u=T(params["u"]), v=T(params["v"]),
                                      w=T(params["w"]), data={"counts": x})
    ref = O.log_likelihood_components(cfg, T(x), T(params["s"]), T(params["u"]), T(params["v"]),
                                      T(params["w"]))
    for k in ("rate", "log_likelihood"):
        a, b = got[k].cpu().double().numpy(), ref[k].numpy()
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1.0), k


@pytest.mark.parametrize("bf16x3", ["1", "0"])
def test_bernoulli_log_transform_sweep_and_fit(monkeypatch, bf16x3):
    from spmf_amd import BernoulliFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    rng = np.random.default_rng(79)
    for case in range(8):
        B, D = int(rng.integers(2, 300)), int(rng.integers(2, 300))
        K, S = int(rng.integers(1, 65)), int(rng.integers(1, 3))
        density = float(rng.choice([0.03, 0.2, 0.8]))
        cfg, x, params = problem_logt(B, D, K, S, 9700 + case, density)
        pref, gref, _ = O.energy_and_grads(cfg, x, params)
        m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                   column_norms=cfg.eta_i, log_transform=True, device="cuda",
                                   panel_rows=int(rng.choice([5, 64, 4096])))
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        tag = f"case {case}: B={B} D={D} K={K} S={S} dens={density}"
        for k, r in pref.items():
            np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                       err_msg=f"{tag} {k}")
        assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, tag)
    X = (rng.random((400, 20)) < 0.2).astype(np.float64)
    m = BernoulliFactorization(latent_dim=2, feature_dim=20, u_tau_scale=1 / math.sqrt(8000),
                               log_transform=True, device="cuda", panel_rows=100)
    torch.manual_seed(1)
    losses = m.fit(lambda: [{"counts": X}], dataset_size=400, sample_size=4, num_steps=15,
                   learning_rate=0.05, rel_tol=1e-9, verbose=False)
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]


def test_bernoulli_surrogate_and_fit_smoke():
    from spmf_amd import BernoulliFactorization
    rng = np.random.default_rng(0)
    X = (rng.random((400, 20)) < 0.2).astype(np.float64)
    m = BernoulliFactorization(latent_dim=2, feature_dim=20, u_tau_scale=1 / math.sqrt(8000),
                               device="cuda", panel_rows=100)
    th = m.surrogate_distribution.sample(3)
    assert float(th["v"].max()) < 0 and float(th["u"].min()) > 0     # v ~ N(-6, 5e-4), identity
    torch.manual_seed(1)
    losses = m.fit(lambda: [{"counts": X}], dataset_size=400, sample_size=4, num_steps=15,
                   learning_rate=0.05, rel_tol=1e-9, verbose=False)
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]


def test_bernoulli_randomised_sweep():
    from spmf_amd import BernoulliFactorization
    rng = np.random.default_rng(78)
    for case in range(12):
        B, D = int(rng.integers(2, 300)), int(rng.integers(2, 300))
        K, S = int(rng.integers(1, 65)), int(rng.integers(1, 3))
        density = float(rng.choice([0.03, 0.2, 0.8]))
        cfg, x, params = problem(B, D, K, S, 9500 + case, density)
        pref, gref, _ = O.energy_and_grads(cfg, x, params)
        m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                   column_norms=cfg.eta_i, device="cuda",
                                   panel_rows=int(rng.choice([5, 64, 4096])))
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        tag = f"case {case}: B={B} D={D} K={K} S={S} dens={density}"
        assert float(nnf.sum()) == 0, tag
        for k, r in pref.items():
            np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                       err_msg=f"{tag} {k}")
        assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, tag)
