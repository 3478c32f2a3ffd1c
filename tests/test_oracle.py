"""Pins for the CPU oracle (PARITY UNPINNED by the reference -- these are the
substitutes SURVEY.md section 8c lists): scipy densities, the HalfCauchy
marginal identity, finite differences, hand-computed known answers and
sparse-exact == dense."""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.stats as st
import torch
from scipy import integrate

from oracle import spmf_oracle as O
from oracle import sparse_exact as SE

F64 = torch.float64


def T(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64))


def test_densities_match_scipy():
    rng = np.random.default_rng(0)
    y = rng.uniform(0.05, 4.0, size=50)
    sc = rng.uniform(0.2, 3.0, size=50)
    np.testing.assert_allclose(
        O.halfnormal_log_prob(T(y), T(sc)).numpy(),
        st.halfnorm.logpdf(y, scale=sc), rtol=1e-13, atol=1e-13)
    a = rng.uniform(0.3, 4.0, size=50)
    np.testing.assert_allclose(
        O.inverse_gamma_log_prob(T(y), T(a), T(sc)).numpy(),
        st.invgamma.logpdf(y, a, scale=sc), rtol=1e-12, atol=1e-12)
    x = rng.poisson(3.0, size=50).astype(np.float64)
    r = rng.uniform(0.1, 9.0, size=50)
    np.testing.assert_allclose(
        O.poisson_log_prob(T(x), T(r)).numpy(), st.poisson.logpmf(x, r),
        rtol=1e-12, atol=1e-12)
    # 0 * log 0 := 0 (multiply_no_nan), tfd.Poisson semantics
    assert O.poisson_log_prob(T([0.0]), T([0.0])).item() == 0.0
    assert O.poisson_log_prob(T([2.0]), T([0.0])).item() == -math.inf


def test_sqrt_inverse_gamma_is_change_of_variables():
    # P(Y<=y) for Y=sqrt(X), X~InvGamma(a,b) equals InvGamma cdf at y^2
    a, b = 0.5, 1.7
    for y in (0.3, 1.0, 2.5):
        val, _ = integrate.quad(
            lambda t: math.exp(O.sqrt_inverse_gamma_log_prob(
                T(t), T(a), T(b)).item()), 0, y, limit=200)
        assert abs(val - st.invgamma.cdf(y * y, a, scale=b)) < 1e-7


@pytest.mark.parametrize("tau", [0.01, 1.0, 3.0])
def test_horseshoe_plus_hierarchy_marginalises_to_halfcauchy(tau):
    # y|a ~ SqrtInvGamma(1/2, 1/a), a ~ InvGamma(1/2, 1/tau^2) => y ~ HalfCauchy(tau)
    # (poisson.py:303-341 replaces the HalfCauchy entries at :252-295)
    for y in (0.2 * tau, tau, 4.0 * tau):
        f = lambda la: math.exp(
            O.sqrt_inverse_gamma_log_prob(T(y), T(0.5), T(math.exp(-la))).item()
            + O.inverse_gamma_log_prob(T(math.exp(la)), T(0.5), T(1 / tau ** 2)).item()
            + la)
        val, _ = integrate.quad(f, -40, 40, limit=400)
        assert abs(val / st.halfcauchy.pdf(y, scale=tau) - 1) < 1e-6


def _tiny_cfg():
    return O.OracleConfig(latent_dim=1, feature_dim=3, scale_rows=True,
                          eta_i=T([[2.0, 1.0, 4.0]]), xi_u_global=5.0)


def test_known_answer_B2_D3_K1():
    """Hand-computed: B=2, D=3, K=1."""
    cfg = _tiny_cfg()
    x = np.array([[2.0, 0.0, 4.0], [0.0, 3.0, 0.0]])
    u = np.array([[[0.5], [0.2], [0.1]]])
    s = np.array([[[1.0, 3.0, 1.0], [1.0, 1.0, 3.0]]])
    v = np.array([[[0.3, 0.6, 0.9]]])
    w = np.array([[[0.2, 0.4, 0.1]]])
    # A = w1*u : w1 = [.5,.75,.25] -> A = [.25,.15,.025]
    # g(x) = x/eta: row0 [1,0,1], row1 [0,3,0]
    # z (unscaled) = [0.275, 0.45]; xi = rowsum/5 = [1.2, 0.6]
    z = np.array([0.275 * 1.2, 0.45 * 0.6])
    np.testing.assert_allclose(
        O.encode(cfg, T(x), T(u), T(s)).numpy().reshape(-1), z, rtol=1e-14)
    phi = np.array([2.0 * 0.5 * 0.2, 1.0 * 0.25 * 0.4, 4.0 * 0.75 * 0.1])
    np.testing.assert_allclose(
        O.intercept_matrix(cfg, T(w), T(s)).numpy().reshape(-1), phi, rtol=1e-14)
    rate = z[:, None] * v.reshape(1, 3) * np.array([2.0, 1.0, 4.0]) + phi[None]
    ll = st.poisson.logpmf(x, rate).sum()
    zp = 2 * (0.5 * math.log(2 / math.pi)) - 0.5 * (z ** 2).sum()
    params = dict(u=u, s=s, v=v, w=w)
    for n, shp in O.var_shapes(3, 1).items():
        if n not in params:
            params[n] = np.ones((1,) + shp)
    parts = O.unormalized_log_prob_parts(cfg, x, params)
    assert abs(parts["x"].item() - ll) < 1e-12
    assert abs(parts["z"].item() - zp) < 1e-13
    # prior parts, all-ones hyper parameters: v ~ HalfNormal(.1)
    assert abs(parts["v"].item() - st.halfnorm.logpdf(v, scale=0.1).sum()) < 1e-12
    assert abs(parts["w"].item() - st.halfnorm.logpdf(w, scale=1.0).sum()) < 1e-12
    assert abs(parts["u"].item() - st.halfnorm.logpdf(
        u.reshape(3, 1), scale=1.0).sum()) < 1e-12
    # SqrtInvGamma(.5, 1)(1) = InvGamma(.5,1).pdf(1) * 2
    sig = math.log(st.invgamma.pdf(1.0, 0.5, scale=1.0) * 2.0)
    assert abs(parts["u_eta"].item() - 3 * sig) < 1e-12
    assert abs(parts["s_eta"].item() - 6 * sig) < 1e-12
    assert abs(parts["u_tau_a"].item()
               - st.invgamma.logpdf(1.0, 0.5, scale=1 / 0.01 ** 2)) < 1e-10
    assert set(parts) == {"v", "w", "u", "s", "u_eta", "u_tau", "s_eta", "s_tau",
                          "u_eta_a", "u_tau_a", "s_eta_a", "s_tau_a", "z", "x"}


def _problem(B, D, K, S, seed, density=0.3, scale_rows=True, empty=True):
    rng = np.random.default_rng(seed)
    mask = rng.random((B, D)) < density
    x = (mask * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    if empty and B > 2 and D > 2:
        x[1, :] = 0.0          # empty row
        x[:, 2] = 0.0          # empty column
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows,
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = T(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(2.0, 6.0))
    params = O.random_params(cfg, S, seed + 1)
    return cfg, x, params


def test_non_finite_rule():
    """poisson.py:606-616: non-finite cells are replaced by (global min - 10)."""
    cfg, x, params = _problem(4, 5, 2, 2, 3, empty=False)
    params["w"][0, 0, 0] = 0.0     # phi=0 for sample 0, column 0
    params["u"][0, 0, :] = 0.0
    x[:, 0] = np.array([3.0, 0, 0, 0])
    x[0, 1:] = 0                    # z_0 comes only from column 0 -> z_0 = 0, rate 0
    ll = O.log_likelihood_components(
        cfg, T(x), T(params["s"]), T(params["u"]), T(params["v"]),
        T(params["w"]))["log_likelihood"]
    assert torch.isinf(ll[0, 0, 0])
    fin = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    m = fin.min() - 10
    expect = torch.where(torch.isfinite(ll), ll, m).sum((-1, -2))
    got = O.unormalized_log_prob_parts(cfg, x, params)["x"]
    np.testing.assert_allclose(got.numpy(), expect.numpy(), rtol=1e-13)


def test_autograd_matches_finite_differences():
    cfg, x, params = _problem(6, 5, 2, 1, 11)
    parts, grads, _ = O.energy_and_grads(cfg, x, params)
    total = lambda p: float(O.unormalized_log_prob(cfg, x, p).sum())
    rng = np.random.default_rng(5)
    for name in O.VAR_ORDER:
        for _ in range(3):
            idx = tuple(rng.integers(0, n) for n in params[name].shape)
            h = 1e-6 * max(1.0, abs(params[name][idx]))
            pp = {k: v.copy() for k, v in params.items()}
            pm = {k: v.copy() for k, v in params.items()}
            pp[name][idx] += h
            pm[name][idx] -= h
            fd = (total(pp) - total(pm)) / (2 * h)
            an = grads[name][idx].item()
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (name, idx, fd, an)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("scale_rows", [True, False])
def test_sparse_exact_equals_dense_oracle(seed, scale_rows):
    B, D, K = [(7, 11, 3), (16, 9, 1), (5, 20, 4), (12, 12, 2), (3, 4, 5), (30, 17, 8)][seed]
    cfg, x, params = _problem(B, D, K, 2, 100 + seed, scale_rows=scale_rows)
    parts, grads, groups = O.energy_and_grads(cfg, x, params)
    decay = cfg.symmetry_breaking_decay ** np.arange(K)
    for smp in range(2):
        one = {k: v[smp] for k, v in params.items()}
        out = SE.data_term(sp.csr_matrix(x), cfg.eta_i.numpy().reshape(-1),
                           cfg.xi_u_global, scale_rows,
                           one["u"], one["v"], one["w"], one["s"])
        assert out["n_nonfinite"] == 0
        np.testing.assert_allclose(out["x"], parts["x"][smp].item(), rtol=1e-12)
        np.testing.assert_allclose(out["z"], parts["z"][smp].item(), rtol=1e-12)
        for n in ("u", "v", "w", "s"):
            ref = groups["data"][n][smp].numpy()
            np.testing.assert_allclose(out["grads"][n], ref, rtol=1e-10,
                                       atol=1e-11 * np.abs(ref).max())
        pparts, pg = SE.prior_term(one, cfg.u_tau_scale, cfg.s_tau_scale, decay)
        for n in O.VAR_ORDER:
            np.testing.assert_allclose(pparts[n], parts[n][smp].item(), rtol=1e-12)
            ref = groups["prior"][n][smp].numpy()
            np.testing.assert_allclose(pg[n], ref, rtol=1e-10,
                                       atol=1e-12 * max(1.0, np.abs(ref).max()))


def test_compute_scales_reference_semantics():
    """poisson.py:113-154: eta = colmean over non-zero entries where > 1 else 1;
    xi = sum of those column means."""
    x = np.array([[4.0, 0, 1, 0], [2.0, 1, 0, 0], [0, 1, 1, 0], [6.0, 0, 0, 2.0]])
    cfg = O.OracleConfig(latent_dim=2, feature_dim=4)
    O.compute_scales(cfg, [x[:2], x[2:]])
    np.testing.assert_allclose(cfg.eta_i.numpy().reshape(-1), [4.0, 1.0, 1.0, 2.0])
    assert abs(float(cfg.xi_u_global) - (4.0 + 1.0 + 1.0 + 2.0)) < 1e-14


def test_surrogate_initial_state_shapes_and_order():
    cfg = O.OracleConfig(latent_dim=3, feature_dim=5, u_tau_scale=0.02)
    st_ = O.surrogate_initial_state(cfg)
    assert tuple(st_.keys()) == O.VAR_ORDER
    assert st_["s"]["loc"][0, 0] == -2.0 and st_["s"]["loc"][1, 0] == -1.0
    assert st_["u_tau_a"]["scale"][0, 0] == pytest.approx(1 / 0.02 ** 2)
    # softplus(N(-6, 5e-4)) is tiny and positive, as the reference initialises u,v,w
    th, lq = O.surrogate_transform(
        "normal", T(st_["u"]["loc"]), T(O.softplus_inverse(st_["u"]["scale"])),
        torch.zeros(5, 3, dtype=F64))
    assert torch.all(th > 0) and th.max() < 3e-3 and torch.isfinite(lq)


def test_bernoulli_restatement():
    """mederrata_spmf/bernoulli.py: Bernoulli(logits) log-pmf, Normal priors on
    v,w, encode without row scaling."""
    rng = np.random.default_rng(4)
    l = rng.normal(0, 3, size=40)
    x = (rng.random(40) < 0.4).astype(np.float64)
    from scipy.special import expit
    np.testing.assert_allclose(
        O.bernoulli_log_prob(T(x), T(l)).numpy(),
        st.bernoulli.logpmf(x.astype(int), expit(l)), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(O.normal_log_prob(T(l), T(0.1)).numpy(),
                               st.norm.logpdf(l, scale=0.1), rtol=1e-12)
    cfg, xx, params = _problem(9, 7, 2, 1, 5)
    cfg.likelihood = "bernoulli"
    xb = (xx > 0).astype(np.float64)
    params["v"] = -params["v"]            # Identity bijector: any sign
    z = O.encode(cfg, T(xb), T(params["u"]), T(params["s"]))
    A = O.encoding_matrix(T(params["u"]), T(params["s"]))
    np.testing.assert_allclose(z.numpy(), (T(xb) / cfg.eta_i @ A).numpy(), rtol=1e-13)
    parts = O.unormalized_log_prob_parts(cfg, xb, params)
    assert abs(parts["v"].item() - st.norm.logpdf(params["v"], scale=0.1).sum()) < 1e-9
    logits = (z @ T(params["v"])) * cfg.eta_i + O.intercept_matrix(cfg, T(params["w"]), T(params["s"]))
    ll = st.bernoulli.logpmf(xb.astype(int), expit(logits.numpy()[0])).sum()
    assert abs(parts["x"].item() - ll) < 1e-9


from hypothesis import given, settings, strategies as hst


@settings(max_examples=40, deadline=None)
@given(B=hst.integers(1, 9), D=hst.integers(1, 9), K=hst.integers(1, 5),
       seed=hst.integers(0, 10_000), density=hst.floats(0.0, 1.0),
       xmax=hst.sampled_from([1, 3, 10_000]), scale_rows=hst.booleans())
def test_sparse_exact_equals_dense_property(B, D, K, seed, density, xmax, scale_rows):
    """SURVEY build-plan step 2: sparse == dense for random patterns, incl.
    empty rows/columns, all-zero matrices and counts up to 1e4."""
    rng = np.random.default_rng(seed)
    x = ((rng.random((B, D)) < density) * rng.integers(1, xmax + 1, size=(B, D))).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows)
    cfg.eta_i = T(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(1.0, 5.0))
    params = O.random_params(cfg, 1, seed + 1)
    parts, _, groups = O.energy_and_grads(cfg, x, params)
    one = {k: v[0] for k, v in params.items()}
    out = SE.data_term(sp.csr_matrix(x), cfg.eta_i.numpy().reshape(-1), cfg.xi_u_global,
                       scale_rows, one["u"], one["v"], one["w"], one["s"])
    np.testing.assert_allclose(out["x"], parts["x"][0].item(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(out["z"], parts["z"][0].item(), rtol=1e-11, atol=1e-11)
    for n in ("u", "v", "w", "s"):
        ref = groups["data"][n][0].numpy()
        np.testing.assert_allclose(out["grads"][n], ref, rtol=1e-9,
                                   atol=1e-10 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("likelihood,logt", [("poisson", False), ("poisson", True), ("mixed", False)])
def test_chunked_oracle_driver_equals_one_piece(likelihood, logt):
    """tests/_chunked_oracle.py (used by the C4 / C5 workload parity tests) adds the
    same oracle up over row chunks: parts and gradients equal the one-piece call."""
    import scipy.sparse as sp
    from _chunked_oracle import data_term, prior_term
    rng = np.random.default_rng(5)
    B, D, K = 53, 17, 3
    x = ((rng.random((B, D)) < 0.3) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    extra = {}
    if likelihood == "mixed":
        mask = np.arange(D) % 2 == 1
        x[:, mask] = x[:, mask] > 0
        extra = {"bernoulli_columns": mask}
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, log_transform=logt, likelihood=likelihood,
                         extra=extra)
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 5.0
    params = O.random_params(cfg, 1, 3)
    parts, grads, split = O.energy_and_grads(cfg, x, params)
    got = data_term(cfg, sp.csr_matrix(x), params, chunk=16)
    assert abs(got["x"] - float(parts["x"])) <= 1e-12 * abs(float(parts["x"]))
    assert abs(got["z"] - float(parts["z"])) <= 1e-12 * abs(float(parts["z"]))
    for k in ("u", "v", "w", "s"):
        np.testing.assert_allclose(got["grads"][k], split["data"][k].numpy(), rtol=1e-10, atol=1e-12)
    pp, pg = prior_term(cfg, params)
    for k, v in pp.items():
        assert abs(v - float(parts[k])) <= 1e-12 * abs(float(parts[k]))
    for k in O.VAR_ORDER:
        np.testing.assert_allclose(pg[k], split["prior"][k].numpy(), rtol=1e-10, atol=1e-12)


def test_c_port_equals_numpy_port():
    """oracle/sparse_exact_omp.c (the multithreaded port bench.py times as
    cpu_baseline) == oracle/sparse_exact.py to fp64 rounding, incl. empty rows and
    columns and an un-scaled run."""
    import scipy.sparse as sp
    from oracle import sparse_exact as SE
    from oracle import sparse_exact_c as SC
    rng = np.random.default_rng(8)
    B, D, K = 301, 77, 7
    x = ((rng.random((B, D)) < 0.1) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    x[3] = 0
    x[:, 5] = 0
    X = sp.csr_matrix(x)
    eta = rng.uniform(0.5, 3.0, size=D)
    u = np.abs(rng.normal(0.3, 0.1, size=(D, K)))
    v = np.abs(rng.normal(0.3, 0.1, size=(K, D)))
    w = np.abs(rng.normal(0.2, 0.05, size=(1, D)))
    s = np.abs(rng.normal(0.4, 0.1, size=(2, D)))
    for scale_rows in (True, False):
        a = SE.data_term(X, eta, 4.5, scale_rows, u, v, w, s)
        b = SC.data_term(X, eta, 4.5, scale_rows, u, v, w, s)
        assert abs(a["x"] - b["x"]) <= 1e-12 * abs(a["x"]) and abs(a["z"] - b["z"]) <= 1e-12 * abs(a["z"])
        assert a["n_nonfinite"] == b["n_nonfinite"] == 0
        for k in ("u", "v", "w", "s"):
            np.testing.assert_allclose(b["grads"][k], a["grads"][k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(b["z_rows"], a["z_rows"], rtol=1e-12, atol=1e-14)
    assert SC.max_threads() >= 1


@pytest.mark.parametrize("scale_rows", [True, False])
def test_c_port_yardstick_equals_oracle_energy_grad_scales(scale_rows):
    """The OpenMP port's entry-wise yardstick (Prepared.step(scales=True): per entry the sum over
    the stored-cell, minus-rate and z-prior pieces of |d piece / d entry|) is the data part of
    oracle.energy_grad_scales -- the metric the full-shard C2 / C3 parity tests and bench.py's
    parity_vs_port hold the HIP gradients to.  Includes empty rows and an empty column."""
    from oracle import sparse_exact_c as SC
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    B, D, K = 57, 23, 4
    x = ((rng.random((B, D)) < 0.2) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    x[4] = 0
    x[:, 7] = 0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(B * D),
                         scale_rows=scale_rows)
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 3.3
    params = O.random_params(cfg, 1, 22)
    ref = O.energy_grad_scales(cfg, x, params, prior=False)
    one = {k: np.asarray(v)[0] for k, v in params.items()}
    out = SC.Prepared(sp.csr_matrix(x), cfg.eta_i.numpy().reshape(-1), cfg.xi_u_global,
                      scale_rows).step(one["u"], one["v"], one["w"], one["s"], scales=True)
    for k in ("u", "v", "w", "s"):
        np.testing.assert_allclose(out["scales"][k], ref[k][0].numpy(), rtol=1e-11, atol=1e-300, err_msg=k)
        # and it dominates the gradient it measures
        assert (np.abs(out["grads"][k]) <= out["scales"][k] * (1 + 1e-12) + 1e-300).all(), k


def test_horseshoe_log_prob_restates_tfp_approximation_close_to_the_exact_density():
    """tfd.Horseshoe.log_prob is a closed-form approximation; the restatement in the
    oracle must sit within its known accuracy (< 1e-3 nats) of the exact
    HalfCauchy-Normal marginal (numerical quadrature) over eight decades of x/scale
    and be asymptotically exact at both ends -- any mis-remembered constant fails."""
    from scipy import integrate
    for scale in (1.0, 0.01, 3.0):
        for r in (1e-4, 1e-3, 1e-2, 0.1, 0.5, 1.0, 2.0, 5.0, 20.0, 100.0, 1e3):
            x = r * scale
            f = lambda lam: (np.exp(-0.5 * (x / (lam * scale)) ** 2) / (np.sqrt(2 * np.pi) * lam * scale)
                             * 2 / (np.pi * (1 + lam * lam)))
            exact = np.log(integrate.quad(f, 0, np.inf, limit=500)[0])
            got = float(O.horseshoe_log_prob(torch.tensor(x, dtype=torch.float64),
                                             torch.tensor(scale, dtype=torch.float64)))
            tol = 1e-3 if 0.05 < r < 10 else 2e-5
            assert abs(got - exact) < tol, (scale, r, got, exact)
    # folded density integrates to one
    tot = integrate.quad(lambda y: float(torch.exp(O.abs_horseshoe_log_prob(
        torch.tensor(y, dtype=torch.float64), torch.tensor(0.7, dtype=torch.float64)))), 0, np.inf,
        limit=500)[0]
    assert abs(tot - 1.0) < 2e-3


def test_abs_horseshoe_branch_parts_and_gradients():
    """horshoe_plus=False (poisson.py:378-398): four variables, AbsHorseshoe priors on
    u (scale u_tau_scale * decay^k) and s (scale s_tau_scale); autograd == finite
    differences."""
    rng = np.random.default_rng(11)
    B, D, K = 9, 7, 3
    x = rng.poisson(1.0, size=(B, D)).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, horseshoe_plus=False, u_tau_scale=0.05)
    p = O.random_params(cfg, 1, 2)
    assert tuple(p) == O.VAR_ORDER_ABS == tuple(cfg.var_order)
    parts = O.unormalized_log_prob_parts(cfg, x, p)
    assert set(parts) == {"v", "w", "u", "s", "z", "x"}
    decay = 0.99 ** np.arange(K)
    want_u = sum(float(O.abs_horseshoe_log_prob(torch.tensor(p["u"][0, d, k]),
                                                torch.tensor(0.05 * decay[k])))
                 for d in range(D) for k in range(K))
    assert abs(float(parts["u"]) - want_u) < 1e-10 * abs(want_u)
    _, grads, _ = O.energy_and_grads(cfg, x, p)
    tot = lambda q: float(sum(v.sum() for v in O.unormalized_log_prob_parts(cfg, x, q).values()))
    for name, idx in (("u", (0, 2, 1)), ("s", (0, 1, 3)), ("v", (0, 1, 4)), ("w", (0, 0, 2))):
        h = 1e-6 * p[name][idx]
        hi = {k: v.copy() for k, v in p.items()}
        lo = {k: v.copy() for k, v in p.items()}
        hi[name][idx] += h
        lo[name][idx] -= h
        fd = (tot(hi) - tot(lo)) / (2 * h)
        assert abs(float(grads[name][idx]) - fd) <= 1e-6 * max(1.0, abs(fd)), (name, fd)
    st = O.surrogate_initial_state(cfg)
    assert tuple(st) == O.VAR_ORDER_ABS and float(st["u"]["loc"][0, 0]) == -9.0


# ---- the entry-wise gradient yardstick (oracle.energy_grad_scales, tests/_gradcheck.py) ----
@pytest.mark.parametrize("lik,hp,logt", [("poisson", True, False), ("poisson", False, False),
                                         ("poisson", True, True), ("bernoulli", True, False)])
def test_prior_terms_sum_to_the_parts(lik, hp, logt):
    rng = np.random.default_rng(11)
    B, D, K, S = 12, 9, 3, 2
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(B * D),
                         log_transform=logt)
    cfg.likelihood, cfg.horseshoe_plus = lik, hp
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 3.0
    p = {k: torch.as_tensor(v) for k, v in O.random_params(cfg, S, 5).items()}
    parts, terms = O.prior_log_prob_parts(cfg, p), O.prior_log_prob_terms(cfg, p)
    assert set(parts) == set(terms)
    for k in parts:
        tot = sum((t * torch.ones_like(p[k])).sum((-1, -2)) for t in terms[k])
        np.testing.assert_allclose(tot.numpy(), parts[k].numpy(), rtol=1e-12, atol=1e-10)


def test_gradient_yardstick_is_the_sum_of_absolute_cell_contributions():
    """For u, v, w every cell's contribution to one piece has the same sign, so the yardstick
    must equal the brute-force sum over cells of |d ll_cell / d entry| + |d z-prior| + the
    prior terms; it always dominates |gradient|; entries nothing contributes to get 0."""
    rng = np.random.default_rng(12)
    B, D, K = 6, 5, 2
    x = ((rng.random((B, D)) < 0.5) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    x[:, 3] = 0.0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 3.0
    params = O.random_params(cfg, 1, 13)
    _, grads, _ = O.energy_and_grads(cfg, x, params)
    sc = O.energy_grad_scales(cfg, x, params)
    for k in grads:
        assert bool((grads[k].abs() <= sc[k] * (1 + 1e-12) + 1e-300).all()), k
    # brute force over cells for the data term of v: stored part and rate part separately
    dsc = O.energy_grad_scales(cfg, x, params, prior=False)
    p = {k: torch.as_tensor(v).clone().requires_grad_(True) for k, v in params.items()}
    xt = torch.as_tensor(x)
    rate = O.log_likelihood_components(cfg, xt, p["s"], p["u"], p["v"], p["w"])["rate"]
    brute = torch.zeros_like(p["v"])
    for b in range(B):
        for d in range(D):
            for piece in (torch.xlogy(xt[b, d], rate[0, b, d]), -rate[0, b, d]):
                g, = torch.autograd.grad(piece, p["v"], retain_graph=True, allow_unused=True)
                if g is not None:
                    brute += g.abs()
    np.testing.assert_allclose(dsc["v"].numpy(), brute.detach().numpy(), rtol=1e-12, atol=1e-300)
    # column 3 holds no entry: d(stored part)/dv is 0 there, only the rate part remains
    assert float(dsc["v"][0, :, 3].min()) > 0
