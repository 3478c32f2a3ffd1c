"""GPU: the non-finite replacement rule WITH its gradient (poisson.py:606-616),
Bernoulli / mixed dense per-cell outputs (bernoulli.py:126-155), the var_list
wrapper (poisson.py:703-709), descriptor-cache lifetime when models share one
resident matrix, the sharded VI step's global batch weighting, and the HIP
surrogate against oracle.surrogate_transform + fp64 autograd."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from test_gpu_parity import build_model, make_problem

pytestmark = pytest.mark.gpu
T = torch.as_tensor


def _rule_energy(cfg, x, params):
    """The energy of poisson.py:582-621 with the replacement rule written so that
    autograd carries what the rule MEANS: finite cells pass through clip with
    gradient one, replaced cells are worth min_val = (min over finite cells) - 10,
    whose derivative is that of the minimum's cell.  (Differentiating the
    reference's own graph literally -- TF or torch alike -- multiplies the zero
    cotangent of the unselected where() branch by d log(r)/dr = inf at a rate-0
    cell and yields NaN; see test_literal_autograd_through_the_rule_is_nan.)"""
    p = {k: T(np.asarray(v, dtype=np.float64)).clone().requires_grad_(True) for k, v in params.items()}
    xt = T(np.asarray(x, dtype=np.float64))
    parts = O.prior_log_prob_parts(cfg, p)
    theta = O.encode(cfg, xt, p["u"], p["s"])
    rate = O.decoder_function(cfg, torch.matmul(theta, p["v"])) + O.intercept_matrix(cfg, p["w"], p["s"])
    bad = (xt > 0) & ~(rate > 0)
    safe = torch.where(bad, torch.ones_like(rate), rate)
    ll = O.poisson_log_prob(xt, safe)
    good = ~bad
    mval = torch.minimum(ll[good.expand_as(ll)].min(), torch.zeros((), dtype=ll.dtype)) - 10.0
    parts["x"] = torch.where(good, ll, torch.zeros_like(ll)).sum((-1, -2)) + bad.sum((-1, -2)) * mval
    parts["z"] = (O.HALF_LOG_2_OVER_PI - 0.5 * theta ** 2).sum((-1, -2))
    tot = sum(v.sum() for v in parts.values())
    g = torch.autograd.grad(tot, [p[k] for k in O.VAR_ORDER])
    return ({k: v.detach() for k, v in parts.items()},
            {k: gk.numpy() for k, gk in zip(O.VAR_ORDER, g)})


def _bad_cell_problem(logt=False):
    cfg, x, params = make_problem(24, 15, 2, 2, 3, 0.3, empty=False)
    cfg.log_transform = logt
    if logt:
        params["v"] *= 0.3
    params["w"][0, 0, 0] = 0.0          # phi = 0 for draw 0, column 0
    params["u"][0, 0, :] = 0.0          # column 0 feeds nothing into z
    x[:, 0] = 0
    x[0, :] = 0
    x[0, 0] = 3.0                       # row 0: only column 0 -> z_0 = 0 -> rate 0 under x = 3
    x[5, 0] = 2.0                       # a second stored cell in column 0 (finite: z_5 > 0)
    return cfg, x, params


@pytest.mark.parametrize("logt", [False, True])
def test_rule_value_and_gradient_match_oracle(logt):
    cfg, x, params = _bad_cell_problem(logt)
    ref_v = O.unormalized_log_prob_parts(cfg, x, params)         # the literal restatement: values
    rparts, rgrads = _rule_energy(cfg, x, params)
    np.testing.assert_allclose(rparts["x"].numpy(), ref_v["x"].numpy(), rtol=1e-12)
    m = build_model(cfg, 8)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params, nonfinite="rule")
    assert nnf.cpu().tolist() == [1.0, 0.0]
    for k, r in ref_v.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, err_msg=k)
    for k, r in rgrads.items():
        g = grads[k].cpu().double().numpy().reshape(r.shape)
        assert np.isfinite(g).all(), k
        assert np.abs(g - r).max() <= 1e-5 * np.abs(r).max(), (k, np.abs(g - r).max(), np.abs(r).max())
    # the class surface applies the rule too
    got = m.unormalized_log_prob_parts({"counts": x}, **params)
    np.testing.assert_allclose(got["x"].cpu().numpy(), ref_v["x"].numpy(), rtol=1e-5)


def test_literal_autograd_through_the_rule_is_nan():
    """Documents why _rule_energy exists: autograd through the line-by-line
    restatement of poisson.py:606-616 gives NaN at a rate-0 stored cell."""
    cfg, x, params = _bad_cell_problem()
    _, grads, _ = O.energy_and_grads(cfg, x, params)
    assert not np.isfinite(grads["w"].numpy()).all()


def test_fit_trains_through_non_finite_cells():
    """A batch with a rate-0 stored cell is not skipped (round 1 did): the eager
    loop applies the rule and the optimiser moves."""
    from spmf_amd import PoissonFactorization
    from spmf_amd.vi import AdamHIP, elbo_step
    rng = np.random.default_rng(1)
    X = rng.poisson(1.0, size=(120, 10)).astype(np.float64)
    X[:, 0] = 0
    X[0, :] = 0
    X[0, 0] = 3.0
    m = PoissonFactorization(latent_dim=2, feature_dim=10, u_tau_scale=1 / math.sqrt(1200),
                             device="cuda", panel_rows=64)
    sur = m.surrogate_distribution
    with torch.no_grad():       # column 0: u -> softplus(-200) = 0 exactly, w -> 0: rate(0,0) = 0
        sur.params_of("u")[0][0, :] = -200.0
        sur.params_of("w")[0][0, 0] = -200.0
    torch.manual_seed(0)
    loss, grads, nnf = elbo_step(m, {"counts": X}, 120, 2)
    assert float(nnf.sum()) > 0 and math.isfinite(float(loss))
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    before = [p.detach().clone() for p in sur.trainable_variables]
    losses = m.fit(lambda: [{"counts": X}], dataset_size=120, sample_size=2, num_steps=3,
                   learning_rate=0.01, verbose=False, rel_tol=1e-12)
    assert len(losses) >= 2 and all(math.isfinite(v) for v in losses)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, sur.trainable_variables))


@pytest.mark.parametrize("mixed", [False, True])
def test_bernoulli_and_mixed_dense_outputs(mixed):
    """log_likelihood_components of BernoulliFactorization (bernoulli.py:126-155:
    'rate' is the logit) and of the build-defined mixed likelihood."""
    from spmf_amd import BernoulliFactorization, MixedFactorization
    rng = np.random.default_rng(9)
    B, D, K, S = 60, 34, 5, 2
    x = ((rng.random((B, D)) < 0.2) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    mask = np.arange(D) % 2 == 1
    if mixed:
        x[:, mask] = x[:, mask] > 0
        cfg = O.OracleConfig(latent_dim=K, feature_dim=D, likelihood="mixed",
                             extra={"bernoulli_columns": mask})
    else:
        x = (x > 0).astype(np.float64)
        cfg = O.OracleConfig(latent_dim=K, feature_dim=D, likelihood="bernoulli", scale_rows=False)
    cfg.eta_i = T(rng.uniform(0.5, 2.0, size=(1, D)))
    cfg.xi_u_global = 3.0
    params = O.random_params(cfg, S, 4, fp32_exact=True)
    params["w"] = params["w"] - 1.0
    ref = O.log_likelihood_components(cfg, T(x), T(params["s"]), T(params["u"]), T(params["v"]),
                                      T(params["w"]))
    if mixed:
        m = MixedFactorization(mask, latent_dim=K, column_norms=cfg.eta_i, device="cuda", panel_rows=32)
    else:
        m = BernoulliFactorization(latent_dim=K, feature_dim=D, column_norms=cfg.eta_i, device="cuda",
                                   panel_rows=32)
    m.xi_u_global = cfg.xi_u_global
    got = m.log_likelihood_components(s=params["s"], u=params["u"], v=params["v"], w=params["w"],
                                      data={"counts": x})
    for k in ("rate", "log_likelihood"):
        g, r = got[k].cpu().double().numpy(), ref[k].numpy()
        assert g.shape == r.shape == (S, B, D)
        np.testing.assert_allclose(g, r, rtol=1e-5, atol=1e-5 * np.abs(r).max(), err_msg=k)


def test_unormalized_log_prob_list_is_the_var_list_wrapper():
    """poisson.py:703-709: positional arguments in var_list order."""
    cfg, x, params = make_problem(30, 12, 3, 2, 17, 0.3)
    m = build_model(cfg, 16)
    m.var_list = list(O.VAR_ORDER)
    ref = O.unormalized_log_prob(cfg, x, params)
    got = m.unormalized_log_prob_list(*[params[n] for n in m.var_list], data={"counts": x})
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5)
    # a permuted argument order must change the answer (it is positional)
    perm = [params[n] for n in m.var_list]
    perm[3], perm[8] = perm[8], perm[3]            # u_eta <-> u_eta_a (same shape)
    other = m.unormalized_log_prob_list(*perm, data={"counts": x})
    assert not np.allclose(other.cpu().numpy(), ref.numpy(), rtol=1e-7)


def test_shared_matrix_descriptor_cache_survives_key_changes():
    """Two models with different xi_u_global / eta share one SparseCounts: going
    A -> B -> A must not read freed row scales or g(x) (the cached spmf_counts
    descriptors hold raw pointers)."""
    from spmf_amd import PoissonFactorization, SparseCounts
    rng = np.random.default_rng(3)
    B, D, K = 90, 40, 4
    x = ((rng.random((B, D)) < 0.25) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    sc = SparseCounts.from_any(x, "cuda", 32)
    models, refs, params = [], [], []
    for j, (xi, lo) in enumerate([(3.0, 0.5), (7.0, 1.5)]):
        cfg = O.OracleConfig(latent_dim=K, feature_dim=D, log_transform=True,
                             u_tau_scale=1 / math.sqrt(B * D))
        cfg.eta_i = T(np.random.default_rng(j).uniform(lo, lo + 1.0, size=(1, D)))
        cfg.xi_u_global = xi
        p = O.random_params(cfg, 1, 20 + j, fp32_exact=True)
        p["v"] *= 0.2
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                 log_transform=True, column_norms=cfg.eta_i,
                                 initialize_distributions=False, device="cuda", panel_rows=32)
        m.xi_u_global = xi
        models.append(m)
        params.append(p)
        refs.append(O.energy_and_grads(cfg, x, p))
    for rnd in range(3):
        for j in (0, 1):
            if rnd == 1:
                # churn the allocator between uses so a stale pointer would hit other data
                junk = [torch.randn(sc.nnz + 7 * i, device="cuda") for i in range(4)]
                del junk
            parts, grads, _ = models[j].energy_and_grads({"counts": sc}, params[j])
            pref, gref, _ = refs[j]
            for k, r in pref.items():
                np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                           err_msg=f"round {rnd} model {j} {k}")
            for k, r in gref.items():
                g = grads[k].cpu().double().numpy().reshape(r.shape)
                assert np.abs(g - r.numpy()).max() <= 1e-5 * np.abs(r.numpy()).max(), (rnd, j, k)


class _TwoShardReducer:
    """Stands in for dist.ShardReducer on ONE device: shard 0's accumulators are
    kept, shard 1's call adds them -- what the sum all-reduce does on rank 1."""

    def __init__(self, rows_global, lgamma_global):
        self.rows_global, self.lgamma_global = rows_global, lgamma_global
        self.saved = None
        self.calls = 0

    def totals(self, rows, lgamma_sum):
        return self.rows_global, self.lgamma_global

    def __call__(self, acc, rows, lgamma_sum):
        self.calls += 1
        if self.saved is None:
            self.saved = acc.clone()
        else:
            acc += self.saved
            self.saved = None
        return self.rows_global, self.lgamma_global


def test_sharded_elbo_step_equals_unsharded():
    """elbo_step over two row shards (global batch weight c = B_global/N, loss
    divisor and 1/(S*B) from the reducer's totals) == the unsharded step on the
    same noise.  Round 1 used the LOCAL row count: prior and entropy were
    down-weighted by 1/world."""
    from spmf_amd import PoissonFactorization, SparseCounts
    from spmf_amd.vi import elbo_step
    rng = np.random.default_rng(12)
    N, D, K, S = 256, 20, 3, 2
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(4 * N * D),
                             device="cuda", panel_rows=64)
    sc = SparseCounts.from_any(X, "cuda", 64)
    m.compute_scales(lambda: [{"counts": sc}])
    dataset_rows = 4 * N                                    # the batch is a quarter of the data
    torch.manual_seed(21)
    loss_u, grads_u, _ = elbo_step(m, {"counts": sc}, dataset_rows, S)
    half = sc.n_panels // 2
    m._batch({"counts": sc})                                # stats are there already
    lg = float(sc.row_lgamma.sum())
    red = _TwoShardReducer(N, lg)
    torch.manual_seed(21)
    elbo_step(m, {"counts": sc, "panels": (0, half)}, dataset_rows, S, all_reduce=red)
    torch.manual_seed(21)
    loss_s, grads_s, _ = elbo_step(m, {"counts": sc, "panels": (half, sc.n_panels)}, dataset_rows, S,
                                   all_reduce=red)
    assert red.calls == 2
    assert abs(float(loss_s) - float(loss_u)) <= 1e-6 * abs(float(loss_u))
    for a, b in zip(grads_s, grads_u):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-30)


def test_calibrate_advi_counts_panel_rows():
    from spmf_amd import PoissonMatrixFactorization, SparseCounts
    from spmf_amd.vi import batch_rows
    rng = np.random.default_rng(2)
    X = rng.poisson(1.0, size=(700, 9)).astype(np.float64)
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    f = PoissonMatrixFactorization(batches, latent_dim=2)
    assert sum(batch_rows(f, b) for b in batches) == 700
    assert [batch_rows(f, b) for b in batches] == [f._batch(b)[1].n_rows for b in batches]


@pytest.mark.parametrize("bernoulli", [False, True])
def test_hip_surrogate_matches_oracle_transform_and_fp64_autograd(bernoulli):
    """spmf_surrogate_fwd / _bwd against oracle.surrogate_transform (poisson.py:403-539
    as the build defines the parameterisation) with fp64 autograd for the chain to
    the trainables, on the kernels' own base noise and implicit gamma derivative."""
    from spmf_amd import BernoulliFactorization, PoissonFactorization
    from spmf_amd._lib import VAR_ORDER
    rng = np.random.default_rng(4)
    N, D, K, S = 150, 14, 3, 3
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    if bernoulli:
        X = (X > 1).astype(np.float64)
    cls = BernoulliFactorization if bernoulli else PoissonFactorization
    m = cls(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda", panel_rows=64)
    sur = m.surrogate_distribution
    torch.manual_seed(8)
    with torch.no_grad():
        for p in sur.trainable_variables:
            p.add_(0.05 * torch.randn_like(p))
    noise = sur.draw_noise(S)
    theta, logq = sur.forward_hip(m, S, noise)
    # oracle side, fp64
    t64 = [p.detach().double().cpu().requires_grad_(True) for p in sur.trainable_variables]
    th_ref, lq_ref = {}, 0.0
    sp = torch.nn.functional.softplus
    for i, n in enumerate(VAR_ORDER):
        t0, t1 = t64[2 * i], t64[2 * i + 1]
        nz, dg = noise[n]
        nz = nz.double().cpu()
        if sur.kinds[n] == "invgamma":
            # g ~ Gamma(a,1) enters with the implicit-reparameterisation derivative the
            # kernels are given: g(a) = g0 + dg/da (a - a0), exact to first order
            a0 = sp(t0).detach()
            g = nz + dg.double().cpu() * (sp(t0) - a0)
            th, lq = O.surrogate_transform("invgamma", t0, t1, g)
        elif sur.kinds[n] == "normal_identity":
            sigma = sp(t1)                          # tfb.Identity(Normal): bernoulli.py:187-193
            th = t0 + sigma * nz
            lq = (-0.5 * nz ** 2 - torch.log(sigma) - 0.5 * math.log(2 * math.pi)).sum((-1, -2))
        else:
            th, lq = O.surrogate_transform("normal", t0, t1, nz)
        th_ref[n] = th
        lq_ref = lq_ref + lq
        np.testing.assert_allclose(theta[n].cpu().double().numpy(), th.detach().numpy(),
                                   rtol=1e-5, atol=1e-7 * float(th.detach().abs().max()), err_msg=n)
    np.testing.assert_allclose(logq.cpu().numpy(), lq_ref.detach().numpy(), rtol=1e-5)
    # backward: the same dE/dtheta fed to both
    c, B = 0.5, N
    parts, g, _ = m.energy_and_grads({"counts": X}, theta, prior_weight=c)
    grads = sur.backward_hip(m, S, noise, g, 1.0 / (S * B), c)
    lin = sum((g[n].double().cpu() * th_ref[n]).sum() for n in VAR_ORDER)
    ref_loss = -(lin - c * lq_ref.sum()) / (S * B)
    ref_grads = torch.autograd.grad(ref_loss, t64)
    for i, (a, r) in enumerate(zip(grads, ref_grads)):
        a = a.double().cpu()
        assert float((a - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1e-12), (i, VAR_ORDER[i // 2])


@pytest.mark.parametrize("B,D,K,S,logt", [(40, 21, 3, 2, False), (120, 64, 32, 1, False), (60, 30, 8, 1, True)])
def test_abs_horseshoe_branch_energy_and_grads(B, D, K, S, logt):
    """horshoe_plus=False (poisson.py:378-398): four variables v, w, s, u with
    AbsHorseshoe priors on u and s (TFP's Horseshoe log-density approximation,
    folded) -- parts and gradients against the fp64 oracle."""
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(60 + B)
    x = ((rng.random((B, D)) < 0.25) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, horseshoe_plus=False, log_transform=logt,
                         u_tau_scale=1.0 / math.sqrt(B * D), s_tau_scale=0.7)
    cfg.eta_i = T(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 4.0
    params = O.random_params(cfg, S, 5, fp32_exact=True)
    if logt:
        params["v"] *= 0.2
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, s_tau_scale=0.7,
                             horshoe_plus=False, log_transform=logt, column_norms=cfg.eta_i,
                             device="cuda", panel_rows=32)
    m.xi_u_global = cfg.xi_u_global
    assert list(m.var_list) == ["v", "w", "s", "u"] and len(m.surrogate_vars) == 8
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0 and set(parts) == {"v", "w", "s", "u", "z", "x"}
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    for k, r in gref.items():
        g = grads[k].cpu().double().numpy().reshape(r.shape)
        assert np.abs(g - r.numpy()).max() <= 1e-5 * np.abs(r.numpy()).max(), k
    # the class surface and the driver work on the four-variable model
    tot = m.unormalized_log_prob(data={"counts": x}, **params)
    np.testing.assert_allclose(tot.cpu().numpy(), O.unormalized_log_prob(cfg, x, params).numpy(), rtol=1e-5)
    th = m.surrogate_distribution.sample(3)
    assert set(th) == {"v", "w", "s", "u"}
    assert float(th["u"].mean()) < 0.1 * float(th["v"].mean())      # u starts at softplus(-9)
    if not logt:
        torch.manual_seed(2)
        losses = m.fit(lambda: [{"counts": x}], dataset_size=B, sample_size=2, num_steps=6,
                       learning_rate=0.02, rel_tol=1e-12, verbose=False)
        assert len(losses) >= 3 and all(math.isfinite(v) for v in losses)


def test_library_rccl_allreduce_single_rank():
    """spmf_comm_unique_id / spmf_comm_init / spmf_allreduce (RCCL bound inside the
    library): a one-rank communicator on this GPU; the step through it equals the
    plain step (a sum over one rank is the identity) and the collective is
    stream-ordered with the kernels around it."""
    from spmf_amd.dist import LibraryComm, ShardReducer
    cfg, x, params = make_problem(120, 40, 8, 2, 33, 0.2)
    m = build_model(cfg, 32)
    p0, g0, _ = m.energy_and_grads({"counts": x}, params)
    comm = LibraryComm(m, rank=0, world=1)
    red = ShardReducer(comm=comm)
    p1, g1, _ = m.energy_and_grads({"counts": x}, params, all_reduce=red)
    for k in p0:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p0[k].cpu().numpy(), rtol=1e-7)
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 1e-6 * float(g0[k].abs().max())
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    comm.all_reduce_(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))


def test_custom_encoder_decoder_callables():
    """poisson.py:94-97: a model built with user callables takes the dense
    torch-on-device route (spmf_amd/custom_codec.py).  (1) callables equal to the
    built-in pair reproduce the HIP path and the oracle; (2) a genuinely different
    pair (sqrt / square) matches an inline fp64 evaluation; (3) fit runs."""
    from spmf_amd import PoissonFactorization
    cfg, x, params = make_problem(50, 24, 4, 2, 21, 0.3)
    eta = cfg.eta_i.to("cuda")
    mk = lambda **kw: PoissonFactorization(
        latent_dim=4, feature_dim=24, u_tau_scale=cfg.u_tau_scale, column_norms=cfg.eta_i,
        initialize_distributions=False, device="cuda", panel_rows=32, **kw)
    m = mk(encoder_function=lambda t: t / eta.to(t.dtype), decoder_function=lambda y: y * eta.to(y.dtype))
    m.xi_u_global = cfg.xi_u_global
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-6, err_msg=k)
    for k, r in gref.items():
        g = grads[k].cpu().double().numpy().reshape(r.shape)
        assert np.abs(g - r.numpy()).max() <= 1e-5 * np.abs(r.numpy()).max(), k
    hip = build_model(cfg, 32)
    p2, _, _ = hip.energy_and_grads({"counts": x}, params)
    np.testing.assert_allclose(parts["x"].cpu().numpy(), p2["x"].cpu().numpy(), rtol=1e-5)
    # (2) sqrt / square
    m2 = mk(encoder_function=lambda t: torch.sqrt(t), decoder_function=lambda y: y * y)
    m2.xi_u_global = cfg.xi_u_global
    parts2, grads2, _ = m2.energy_and_grads({"counts": x}, params)
    xt, pp = T(x), {k: T(v) for k, v in params.items()}
    w = pp["s"] / pp["s"].sum(-2, keepdim=True)
    th = torch.matmul(torch.sqrt(xt), w[..., 0, :].unsqueeze(-1) * pp["u"]) * (xt.sum(-1, keepdim=True) / cfg.xi_u_global)
    rate = torch.matmul(th, pp["v"]) ** 2 + cfg.eta_i * w[..., 1, :].unsqueeze(-2) * pp["w"]
    xref = O.poisson_log_prob(xt, rate).sum((-1, -2))
    np.testing.assert_allclose(parts2["x"].cpu().numpy(), xref.numpy(), rtol=1e-9)
    z = m2.encode(x, u=pp["u"][0], s=pp["s"][0])
    np.testing.assert_allclose(z.cpu().double().numpy(), th[0].numpy(), rtol=1e-5)
    # (3) the driver trains it (eager loop)
    m3 = PoissonFactorization(latent_dim=2, feature_dim=24, u_tau_scale=cfg.u_tau_scale, device="cuda",
                              panel_rows=32, encoder_function=lambda t: torch.log1p(t))
    torch.manual_seed(0)
    losses = m3.fit(lambda: [{"counts": x}], dataset_size=50, sample_size=2, num_steps=5,
                    learning_rate=0.02, rel_tol=1e-12, verbose=False)
    assert len(losses) >= 3 and all(math.isfinite(v) for v in losses)


def test_finish_is_deterministic_given_the_accumulators():
    """SURVEY section 5 (deterministic-reduction self-check): the finish kernel's
    cross-block sums are two-stage in block order, so two runs over the SAME
    accumulators give bit-identical parts and gradients -- what keeps the replicated
    parameters of a row-sharded job from drifting (every rank holds identical
    accumulators after the all-reduce).  The data pass itself uses float atomics in the
    column pass: two full evaluations agree to ~1e-7, not bitwise, which is also checked."""
    import ctypes as C
    from spmf_amd import _lib
    cfg, x, params = make_problem(400, 300, 32, 2, 5, 0.05)
    m = build_model(cfg, 64)
    p1, g1, _ = m.energy_and_grads({"counts": x}, params)
    lib, h = _lib.load(), m._handle()
    sc, cs = m._batch({"counts": x})
    S, P = m._pack_params(params)
    eta = m._eta_device()
    stream = torch.cuda.current_stream().cuda_stream
    pin = _lib.PtrArray(*[P[n].data_ptr() for n in _lib.VAR_ORDER])
    outs = []
    for rep in range(3):
        grads = {n: torch.full_like(P[n], float("nan")) for n in _lib.VAR_ORDER}
        gout = _lib.PtrArray(*[grads[n].data_ptr() for n in _lib.VAR_ORDER])
        parts = torch.full((S, 14), float("nan"), dtype=torch.float64, device="cuda")
        nnf = torch.empty(2 * S, dtype=torch.float64, device="cuda")
        _lib.check(h, lib.spmf_finish(h, S, cs.n_rows, float(cs.lgamma_sum), 1.0, pin, eta.data_ptr(),
                                      parts.data_ptr(), gout, nnf.data_ptr(), stream), "spmf_finish")
        torch.cuda.synchronize()
        outs.append((parts.clone(), {k: v.clone() for k, v in grads.items()}))
    for parts, grads in outs[1:]:
        assert torch.equal(parts, outs[0][0])
        for k in grads:
            assert torch.equal(grads[k], outs[0][1][k]), k
    # same values as the first full evaluation (same accumulators are still in the workspace)
    for i, n in enumerate(_lib.PART_ORDER):
        assert torch.equal(outs[0][0][:, i], p1[n]), n
    # a second full evaluation: float atomics reorder the column pass's sums
    p2, g2, _ = m.energy_and_grads({"counts": x}, params)
    for k in g1:
        assert float((g1[k] - g2[k]).abs().max()) <= 1e-6 * float(g1[k].abs().max()), k


def test_custom_callables_with_abs_horseshoe_and_dense_surface():
    """Corners that used to raise: user callables together with horshoe_plus=False
    (poisson.py:94-97 + :378-398) and log_likelihood_components of such a model (:156-184).
    Callables equal to the built-in pair must reproduce the oracle's AbsHorseshoe energy and the
    HIP model's dense surface."""
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(31)
    B, D, K, S = 40, 18, 3, 2
    x = ((rng.random((B, D)) < 0.3) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(B * D))
    cfg.horseshoe_plus = False
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 4.2
    params = O.random_params(cfg, S, 32, fp32_exact=True)
    eta = cfg.eta_i.to("cuda")
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, column_norms=cfg.eta_i,
                             horshoe_plus=False, initialize_distributions=False, device="cuda", panel_rows=16,
                             encoder_function=lambda t: t / eta.to(t.dtype),
                             decoder_function=lambda y: y * eta.to(y.dtype))
    m.xi_u_global = cfg.xi_u_global
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0 and set(parts) == set(pref) == {"v", "w", "u", "s", "z", "x"}
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-9, err_msg=k)
    from _gradcheck import assert_grads_entrywise
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "custom+abs-horseshoe")
    # dense surface of a callable model == the HIP model's (same built-in pair)
    hip = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, column_norms=cfg.eta_i,
                               horshoe_plus=False, initialize_distributions=False, device="cuda", panel_rows=16)
    hip.xi_u_global = cfg.xi_u_global
    pp = {k: T(v) for k, v in params.items()}
    a = m.log_likelihood_components(pp["s"], pp["u"], pp["v"], pp["w"], {"counts": x})
    b = hip.log_likelihood_components(pp["s"], pp["u"], pp["v"], pp["w"], {"counts": x})
    ref = O.log_likelihood_components(cfg, T(x), pp["s"], pp["u"], pp["v"], pp["w"])
    for key in ("log_likelihood", "rate"):
        assert tuple(a[key].shape) == (S, B, D)
        np.testing.assert_allclose(a[key].cpu().numpy(), ref[key].numpy(), rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(a[key].cpu().numpy(), b[key].cpu().numpy(), rtol=1e-5, atol=1e-5)
    one = m.log_likelihood_components(pp["s"][0], pp["u"][0], pp["v"][0], pp["w"][0], {"counts": x})
    assert tuple(one["rate"].shape) == (B, D)
