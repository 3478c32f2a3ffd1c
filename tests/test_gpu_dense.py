"""GPU: dense per-cell outputs (log_likelihood_components,
poisson.py:156-184) and the non-finite replacement rule (:606-616) vs the
fp64 oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from test_gpu_parity import build_model, make_problem

pytestmark = pytest.mark.gpu
T = torch.as_tensor


@pytest.mark.parametrize("K,logt", [(3, False), (16, False), (8, True)])
def test_log_likelihood_components_dense(K, logt):
    cfg, x, params = make_problem(70, 45, K, 2, 91 + K, 0.25)
    cfg.log_transform = logt
    if logt:
        params["v"] *= 0.05
    from spmf_amd import PoissonFactorization
    m = PoissonFactorization(latent_dim=K, feature_dim=45, u_tau_scale=cfg.u_tau_scale,
                             log_transform=logt, column_norms=cfg.eta_i,
                             initialize_distributions=False, device="cuda", panel_rows=32)
    m.xi_u_global = cfg.xi_u_global
    ref = O.log_likelihood_components(cfg, T(x), T(params["s"]), T(params["u"]),
                                      T(params["v"]), T(params["w"]))
    got = m.log_likelihood_components(s=params["s"], u=params["u"], v=params["v"],
                                      w=params["w"], data={"counts": x})
    for k in ("rate", "log_likelihood"):
        g = got[k].cpu().double().numpy()
        r = ref[k].numpy()
        assert g.shape == r.shape == (2, 70, 45)
        np.testing.assert_allclose(g, r, rtol=1e-5, atol=1e-5 * np.abs(r).max(), err_msg=k)
    pred = m.predictive_distribution(s=params["s"], u=params["u"], v=params["v"],
                                     w=params["w"], data={"counts": x})
    np.testing.assert_allclose(pred["ll"].cpu().double().numpy(),
                               ref["log_likelihood"].sum(-1).numpy(), rtol=1e-4)
    one = {k: T(v[0]) for k, v in params.items()}
    g1 = m.log_likelihood_components(s=one["s"], u=one["u"], v=one["v"], w=one["w"],
                                     data={"counts": x})
    assert tuple(g1["rate"].shape) == (70, 45)


def test_non_finite_rule_matches_reference_semantics():
    """A stored cell with rate 0 has log-pmf -inf: the reference replaces it by
    (global min over [S,B,D] - 10) (poisson.py:606-616)."""
    cfg, x, params = make_problem(24, 15, 2, 2, 3, 0.3, empty=False)
    params["w"][0, 0, 0] = 0.0          # phi = 0 for draw 0, column 0
    params["u"][0, 0, :] = 0.0          # column 0 feeds nothing into z
    x[:, 0] = 0
    x[0, :] = 0
    x[0, 0] = 3.0                       # row 0: only column 0 -> z_0 = 0 -> rate 0, x = 3
    ref = O.unormalized_log_prob_parts(cfg, x, params)
    ll = O.log_likelihood_components(cfg, T(x), T(params["s"]), T(params["u"]),
                                     T(params["v"]), T(params["w"]))["log_likelihood"]
    assert torch.isinf(ll[0, 0, 0]) and torch.isfinite(ll[1]).all()
    m = build_model(cfg, 8)
    got = m.unormalized_log_prob_parts({"counts": x}, **params)
    np.testing.assert_allclose(got["x"].cpu().numpy(), ref["x"].numpy(), rtol=1e-5)
    np.testing.assert_allclose(got["z"].cpu().numpy(), ref["z"].numpy(), rtol=1e-5)
    # and the energy/grad entry point reports the cell instead of hiding it
    _, _, nnf = m.energy_and_grads({"counts": x}, params)
    assert nnf.cpu().tolist() == [1.0, 0.0]


def test_waic_matches_pointwise_definition():
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(2)
    x = rng.poisson(1.0, size=(200, 12)).astype(np.float64)
    m = PoissonFactorization(latent_dim=2, feature_dim=12, u_tau_scale=1 / math.sqrt(2400),
                             device="cuda", panel_rows=64)
    torch.manual_seed(4)
    out = m.waic({"counts": x}, nsamples=50)
    assert set(out) == {"waic", "se", "lppd", "pwaic"}
    torch.manual_seed(4)
    th = m.surrogate_distribution.sample(50)
    cfg = O.OracleConfig(latent_dim=2, feature_dim=12)
    ll = O.log_likelihood_components(cfg, T(x), th["s"].double().cpu(), th["u"].double().cpu(),
                                     th["v"].double().cpu(), th["w"].double().cpu())["log_likelihood"]
    lppd = (torch.logsumexp(ll, 0) - math.log(50)).sum().item()
    pw = ll.var(0, unbiased=True).sum().item()
    assert abs(out["lppd"] - lppd) <= 1e-5 * abs(lppd)
    assert abs(out["pwaic"] - pw) <= 1e-3 * max(abs(pw), 1e-9) + 1e-9
    assert abs(out["waic"] + 2 * (lppd - pw)) <= 1e-5 * abs(out["waic"])
