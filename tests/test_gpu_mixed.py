"""GPU parity of the (build-defined) mixed Poisson/Bernoulli likelihood vs the
fp64 oracle's definition of it (BASELINE.json config 5; mixed.py is empty in
the reference, so this pins the build against its own restatement only)."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from _gradcheck import assert_grads_entrywise

pytestmark = pytest.mark.gpu


def problem(B, D, K, S, seed, density, scale_rows=True):
    rng = np.random.default_rng(seed)
    mask = (np.arange(D) % 2 == 1)                       # 50/50 columns, as config 5
    x = ((rng.random((B, D)) < density) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    x[:, mask] = (x[:, mask] > 0)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows, likelihood="mixed",
                         u_tau_scale=1.0 / math.sqrt(B * D), extra={"bernoulli_columns": mask})
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 2.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(2.0, 6.0))
    params = O.random_params(cfg, S, seed + 1, fp32_exact=True)
    sign = np.where(mask, rng.choice([-1.0, 1.0], size=D), 1.0)
    params["v"] = params["v"] * sign[None, None, :]
    params["w"] = params["w"] * np.where(mask, -3.0, 1.0)[None, None, :]
    return cfg, x, params, mask


@pytest.mark.parametrize("bf16x3", ["1", "0"])
@pytest.mark.parametrize("B,D,K,S,density,sr", [(37, 24, 3, 2, 0.3, True), (150, 90, 8, 1, 0.1, False),
                                                (260, 200, 32, 1, 0.05, True), (300, 129, 64, 2, 0.05, True)])
def test_mixed_energy_and_grads(monkeypatch, B, D, K, S, density, sr, bf16x3):
    """Both dense paths of the sigmoid sums (read at spmf_ctx_create): the bf16x3 kernels with the fused
    row pass (csrc/dense3.hip sigdot3, default at K <= 32) and the exact-f32 MFMA kernels."""
    from spmf_amd import MixedFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    cfg, x, params, mask = problem(B, D, K, S, 800 + B + K, density, sr)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = MixedFactorization(mask, latent_dim=K, u_tau_scale=cfg.u_tau_scale, scale_rows=sr,
                           column_norms=cfg.eta_i, device="cuda", panel_rows=64)
    m.xi_u_global = cfg.xi_u_global
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                   err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")


@pytest.mark.parametrize("bf16x3", ["0", "1"])
@pytest.mark.parametrize("kind", ["mixed", "bernoulli"])
def test_sigmoid_e_buffer_in_several_row_chunks(monkeypatch, kind, bf16x3):
    """bf16x3 = 0: the sigmoid form of the kept-E path (exact-f32 kernels) with a small
    spmf_ctx_set_e_cap: several row chunks, a last chunk that ends inside a 32-row round (row sums of E
    masked there), and -- mixed -- the scatter through the compacted Bernoulli column list.
    bf16x3 = 1: the same problem on the sigdot3 kernels (no E buffer): row and column counts off the
    128 / 256 tile edges, several Q chunks with float-atomic epilogues, the compacted column list."""
    from spmf_amd import BernoulliFactorization, MixedFactorization, _lib
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    B, D, K, S = 3001, 90, 32, 1
    if kind == "mixed":
        cfg, x, params, mask = problem(B, D, K, S, 5150, 0.04, True)
        m = MixedFactorization(mask, latent_dim=K, u_tau_scale=cfg.u_tau_scale, scale_rows=True,
                               column_norms=cfg.eta_i, device="cuda", panel_rows=512)
        m.xi_u_global = cfg.xi_u_global
    else:
        import test_gpu_bernoulli as TB
        cfg, x, params = TB.problem(B, D, K, S, 5151, 0.04)
        m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                   column_norms=cfg.eta_i, device="cuda", panel_rows=512)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    lib, h = _lib.load(), m._handle()
    _lib.check(h, lib.spmf_ctx_set_e_cap(h, 1 << 20), "spmf_ctx_set_e_cap")   # 2 chunks of 1536 / 1465 rows
    assert (1 << 20) // (96 * 4) < B
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")


def test_mixed_fit_smoke():
    from spmf_amd import MixedFactorization
    rng = np.random.default_rng(0)
    mask = np.arange(20) % 2 == 1
    X = rng.poisson(1.0, size=(400, 20)).astype(np.float64)
    X[:, mask] = (rng.random((400, 10)) < 0.2)
    m = MixedFactorization(mask, latent_dim=2, u_tau_scale=1 / math.sqrt(8000), device="cuda",
                           panel_rows=100)
    th = m.surrogate_distribution.sample(2)
    assert float(th["v"][:, :, mask].max()) < 0 < float(th["v"][:, :, ~mask].min())
    torch.manual_seed(1)
    losses = m.fit(lambda: [{"counts": X}], dataset_size=400, sample_size=4, num_steps=15,
                   learning_rate=0.05, rel_tol=1e-9, verbose=False)
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
