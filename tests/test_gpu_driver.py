"""GPU: the VI driver around the hot path -- gradient chain to the surrogate
trainables vs an autograd-through-the-oracle reference, a short fit that must
reduce the loss, and the CLI end to end on a tiny CSV."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(N=600, D=24, seed=3):
    rng = np.random.default_rng(seed)
    Z = np.abs(rng.normal(0, 1, size=(N, 2)))
    V = np.abs(rng.normal(1.5, 0.5, size=(2, D // 3)))
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    X[:, ::3] = rng.poisson(Z @ V)
    return X


def test_elbo_step_gradients_match_oracle_autograd():
    from spmf_amd import PoissonFactorization
    from _vi_reference import elbo_step_reference, rsample
    X = _data(200, 12)
    N, D = X.shape
    K, S = 3, 2
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device="cuda", panel_rows=64)
    batch = {"counts": X}
    torch.manual_seed(5)
    loss, grads, nnf = elbo_step_reference(m, batch, dataset_rows=N, sample_size=S)
    # same draw again on the reference side: re-seed and rebuild theta with autograd
    torch.manual_seed(5)
    sur = m.surrogate_distribution
    theta, logq = rsample(sur, S)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=m.u_tau_scale)
    th64 = {k: v.double().cpu() for k, v in theta.items()}
    parts = O.unormalized_log_prob_parts(cfg, X, th64)
    prior = sum(parts[n] for n in O.VAR_ORDER)
    c = 1.0   # full batch: B/N = 1
    ref_loss = -(parts["x"] + parts["z"] + c * prior - c * logq.double().cpu()).mean() / N
    ref_grads = torch.autograd.grad(ref_loss, sur.trainable_variables)
    assert abs(float(loss) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach()))
    for g, r in zip(grads, ref_grads):
        r = r.to(g)
        assert (g - r).abs().max() <= 1e-5 * max(float(r.abs().max()), 1e-12)


@pytest.mark.parametrize("bernoulli", [False, True])
def test_hip_vi_step_matches_torch_autograd_reference(bernoulli):
    """surrogate_fwd / surrogate_bwd / adam HIP kernels vs the plain torch
    restatement of the same step, on identical base noise."""
    from spmf_amd import BernoulliFactorization, PoissonFactorization
    from spmf_amd import vi
    from _vi_reference import Adam as RefAdam, GammaReparam
    X = _data(300, 15)
    if bernoulli:
        X = (X > 1).astype(np.float64)
    N, D = X.shape
    K, S = 4, 3
    cls = BernoulliFactorization if bernoulli else PoissonFactorization
    m = cls(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda",
            panel_rows=64)
    sur = m.surrogate_distribution
    torch.manual_seed(11)
    with torch.no_grad():                       # move off the symmetric initial point
        for p in sur.trainable_variables:
            p.add_(0.05 * torch.randn_like(p))
    noise = sur.draw_noise(S)
    theta, logq = sur.forward_hip(m, S, noise)
    # torch restatement on the same noise
    th_ref, lq_ref = {}, 0.0
    sp = torch.nn.functional.softplus
    for n in vi.VAR_ORDER:
        t0, t1 = sur.params_of(n)
        nz, _ = noise[n]
        if sur.kinds[n] == "invgamma":
            a, b = sp(t0), sp(t1)
            g = GammaReparam.apply(nz, a.expand(nz.shape))
            y = b / g
            lq = a * torch.log(b) - torch.lgamma(a) - (a + 1) * torch.log(y) - b / y
        else:
            sg = sp(t1)
            y = t0 + sg * nz
            lq = -0.5 * nz ** 2 - torch.log(sg) - 0.5 * math.log(2 * math.pi)
        if sur.kinds[n] == "normal_identity":
            th = y
        else:
            th = sp(y)
            lq = lq - torch.nn.functional.logsigmoid(y)
        th_ref[n] = th
        lq_ref = lq_ref + lq.sum((-1, -2))
    for n in vi.VAR_ORDER:
        np.testing.assert_allclose(theta[n].cpu().numpy(), th_ref[n].detach().cpu().numpy(),
                                   rtol=1e-5, atol=1e-7, err_msg=n)
    np.testing.assert_allclose(logq.cpu().numpy(), lq_ref.detach().double().cpu().numpy(), rtol=1e-5)
    # backward: same energy gradient fed to both
    c, B = 0.5, N
    parts, g, _ = m.energy_and_grads({"counts": X}, theta, prior_weight=c)
    grads = sur.backward_hip(m, S, noise, g, 1.0 / (S * B), c)
    lin = sum((g[n] * th_ref[n]).sum() for n in vi.VAR_ORDER)
    ref_loss = -(lin - c * lq_ref.sum()) / (S * B)
    ref_grads = torch.autograd.grad(ref_loss, sur.trainable_variables)
    for i, (a, r) in enumerate(zip(grads, ref_grads)):
        assert (a - r).abs().max() <= 2e-4 * max(float(r.abs().max()), 1e-12), i   # fp32 torch reference (its own rounding)
    # ... and against the SAME chain in fp64: the trainables, the base noise and the
    # implicit gamma gradient the kernel was handed (noise[n][1]) as doubles, torch autograd in fp64 --
    # what is left is the kernel's own fp32 arithmetic, held to the contract's 1e-5 of the largest entry
    class _GammaGiven(torch.autograd.Function):
        @staticmethod
        def forward(ctx, g, a, dgda):
            ctx.save_for_backward(dgda)
            return g

        @staticmethod
        def backward(ctx, grad):
            return None, grad * ctx.saved_tensors[0], None
    tv64 = [p.detach().double().requires_grad_(True) for p in sur.trainable_variables]
    idx = {id(p): j for j, p in enumerate(sur.trainable_variables)}
    lin64, lq64 = 0.0, 0.0
    for n in vi.VAR_ORDER:
        t0, t1 = sur.params_of(n)
        t0, t1 = tv64[idx[id(t0)]], tv64[idx[id(t1)]]
        nz, dg = noise[n]
        nz = nz.double()
        if sur.kinds[n] == "invgamma":
            a, b = sp(t0), sp(t1)
            gg = _GammaGiven.apply(nz, a.expand(nz.shape), dg.double())
            y = b / gg
            lq = a * torch.log(b) - torch.lgamma(a) - (a + 1) * torch.log(y) - b / y
        else:
            sg = sp(t1)
            y = t0 + sg * nz
            lq = -0.5 * nz ** 2 - torch.log(sg) - 0.5 * math.log(2 * math.pi)
        if sur.kinds[n] == "normal_identity":
            th = y
        else:
            th = sp(y)
            lq = lq - torch.nn.functional.logsigmoid(y)
        lin64 = lin64 + (g[n].double() * th).sum()
        lq64 = lq64 + lq.sum()
    ref64 = torch.autograd.grad(-(lin64 - c * lq64) / (S * B), tv64)
    for i, (a, r) in enumerate(zip(grads, ref64)):
        assert (a.double() - r).abs().max() <= 1e-5 * max(float(r.abs().max()), 1e-12), i
    # Adam: one fused step == the tensor-op Adam
    p0 = [p.detach().clone() for p in sur.trainable_variables]
    ref_params = [p.detach().clone().requires_grad_(False) for p in p0]
    ref_opt = RefAdam(ref_params, 0.01)
    ref_opt.step([gg.clamp(-3.0, 3.0) for gg in ref_grads])
    opt = vi.AdamHIP(m, sur.trainable_variables, 0.01)
    opt.step(list(ref_grads), clip_value=3.0)
    for p, r in zip(sur.trainable_variables, ref_params):
        assert (p.detach() - r).abs().max() <= 1e-6 * max(1.0, float(r.abs().max()))


def test_fit_reduces_loss_and_sets_expectations():
    from spmf_amd import PoissonFactorization, SparseCounts
    X = _data()
    N, D = X.shape
    m = PoissonFactorization(latent_dim=2, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device="cuda", panel_rows=100)
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    m.compute_scales(lambda: [{"counts": X}])
    torch.manual_seed(0)
    losses = m.fit(lambda: batches, dataset_size=N, sample_size=4, num_steps=30,
                   learning_rate=0.05, rel_tol=1e-9, verbose=False)
    assert len(losses) >= 10 and all(math.isfinite(v) for v in losses)
    assert np.mean(losses[-3:]) < losses[0] - 0.5
    A = m.encoding_matrix()
    assert tuple(A.shape) == (D, 2) and bool((A >= 0).all())
    z = m.encode(X)
    assert tuple(z.shape) == (N, 2)


def test_cli_end_to_end(tmp_path):
    X = _data(300, 9)
    f = tmp_path / "counts.csv"
    np.savetxt(f, X, delimiter=",", fmt="%d")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "factorize_csv.py"),
                        "-f", str(f), "-e", "5", "-d", "2", "-b", "100", "-lr", "0.05", "-rn"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    base = f"{f}_2D"
    enc = np.loadtxt(base + "_encoding_lt_False_rn_True.csv", delimiter=",", ndmin=2)
    assert enc.shape == (2, 9)
    rep = np.loadtxt(base + "_representation_lt_False_rn_True.csv", delimiter=",", ndmin=2)
    assert rep.shape == (300, 3) and np.array_equal(rep[:, 0], np.arange(300))
    assert os.path.exists(base + "_model_lt_False_rn_True.pkl")
    assert "Feature dim: 9 -> Latent dim 2" in r.stdout


def test_cli_log_transform_flag(tmp_path):
    """-lt (log_transform) through the CLI: dense exp kernels in the training loop, outputs named lt_True."""
    rng = np.random.default_rng(2)
    X = rng.poisson(rng.gamma(1.0, 1.0, size=(1, 12)) * rng.gamma(2.0, 0.5, size=(240, 1)))
    f = tmp_path / "c.csv"
    np.savetxt(f, X, delimiter=",", fmt="%d")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "factorize_csv.py"),
                        "-f", str(f), "-e", "4", "-d", "2", "-b", "80", "-lr", "0.02", "-lt"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    enc = np.loadtxt(f"{f}_2D_encoding_lt_True_rn_False.csv", delimiter=",", ndmin=2)
    assert enc.shape == (2, 12) and np.isfinite(enc).all()
    rep = np.loadtxt(f"{f}_2D_representation_lt_True_rn_False.csv", delimiter=",", ndmin=2)
    assert rep.shape == (240, 3) and np.isfinite(rep).all()


def test_scrnaseq_cli_end_to_end(tmp_path):
    """bin/factorize_scrnaseq_counts.py (reference :29-130, the C4 model's caller): log_transform
    with gene-mean column norms through the legacy constructor + calibrate_advi, dense and CSR
    inputs, the seven .npy outputs and their score algebra."""
    import scipy.sparse as sp
    rng = np.random.default_rng(4)
    N, D, P = 300, 40, 3
    types = rng.integers(0, P, size=N)
    prog = rng.gamma(0.3, 1.0, size=(P, D)) * (rng.random((P, D)) < 0.3) * 6.0 + 0.2
    depth = rng.lognormal(0.0, 0.4, size=(N, 1))
    X = rng.poisson(prog[types] * depth).astype(np.int64)
    names = np.array([f"GENE{j}" for j in range(D)], dtype=object)
    np.save(tmp_path / "toy_counts.npy", X)
    np.save(tmp_path / "toy_genenames.npy", names, allow_pickle=True)
    cmd = [sys.executable, os.path.join(ROOT, "bin", "factorize_scrnaseq_counts.py"),
           "--genes", str(tmp_path / "toy_genenames.npy"), "-d", str(P), "-b", "64", "-e", "6",
           "--seed", "3"]
    r = subprocess.run(cmd + ["--counts", str(tmp_path / "toy_counts.npy")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"Total observations={N}, Batch size=64: dropping {N % 64} observations." in r.stdout
    assert "factor 0: GENE" in r.stdout and "intercept: GENE" in r.stdout
    out = {k: np.load(tmp_path / f"toy_{k}_{P}.npy") for k in
           ("U", "V", "W", "Z", "cellscore", "genescore", "interceptscore")}
    assert out["U"].shape == (D, P) and out["V"].shape == (P, D) and out["W"].shape == (1, D)
    assert out["Z"].shape == (N, P)
    assert all(np.isfinite(v).all() for v in out.values())
    rsf = X.sum(1) / np.median(X.sum(1))
    cn = np.maximum(X.mean(0), 1e-3)
    np.testing.assert_allclose(out["cellscore"], out["Z"] * rsf[:, None], rtol=1e-6)
    np.testing.assert_allclose(out["genescore"], out["V"] * cn[None, :], rtol=1e-6)
    np.testing.assert_allclose(out["interceptscore"], out["W"] * cn[None, :], rtol=1e-6)
    # the same run from a CSR .npz: same seed, same batches -> same factors
    sp.save_npz(tmp_path / "toyb_counts.npz", sp.csr_matrix(X))
    r = subprocess.run(cmd + ["--counts", str(tmp_path / "toyb_counts.npz")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    np.testing.assert_allclose(np.load(tmp_path / f"toyb_V_{P}.npy"), out["V"], rtol=2e-3, atol=1e-6)


def test_qualitative_linear_structure_outcome():
    """notebooks/factorize_linear_structure.ipynb:53-66: every third column is
    driven by the latent factors, the rest is Poisson(1) noise.  After fitting,
    the factor columns must carry the encoder loadings and the
    per-row loss must land where the notebook's legacy run did (~47 nats/row)."""
    from spmf_amd import PoissonMatrixFactorization, SparseCounts
    rng = np.random.default_rng(0)
    N, P, D = 8000, 3, 30
    V = np.abs(rng.normal(1.5, 0.5, size=(P, 10)))
    Z = np.abs(rng.normal(0, 1, size=(N, P)))
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    X[:, ::3] = rng.poisson(Z @ V)
    sc = SparseCounts.from_any(X, "cuda", 1000)
    batches = [{"counts": sc, "panels": (p, p + 1)} for p in range(sc.n_panels)]
    factor = PoissonMatrixFactorization(batches, latent_dim=P, u_tau_scale=1 / math.sqrt(D * N))
    torch.manual_seed(0)
    losses = factor.calibrate_advi(num_steps=250, learning_rate=0.05, rel_tol=1e-9, verbose=False,
                                   max_decay_steps=1000)   # the plateau rule may fire early: do not stop on it
    assert losses[-1] < 50.0 < losses[0]
    load = factor.encoding_matrix().abs().sum(1).cpu().numpy()
    noise_cols = np.delete(np.arange(D), np.arange(0, D, 3))
    # (float-atomic ordering makes runs differ in the last digits and a run can hit the
    #  plateau/restore loop early: typical ratio 20-50, worst seen 8.8)
    assert load[::3].mean() > 5 * load[noise_cols].mean()
    assert load[::3].min() > 2 * load[noise_cols].max()


def test_reference_smoke_script_shape():
    """The reference's own test (tests/spmf_test.py:13-44), scaled down: same
    constructor, compute_scales(data_factory) and fit(...) keyword arguments."""
    from mederrata_spmf import PoissonFactorization
    rng = np.random.default_rng(0)
    N, D, P = 4000, 35, 5
    counts = rng.poisson(1.0, size=(N, D))
    data = {"counts": counts, "indices": np.arange(N), "normalization": np.ones(N)}

    def data_factory(batch_size=1000):
        perm = rng.permutation(N)                      # ds.shuffle(...).batch(bs)
        return [{k: v[perm[i:i + batch_size]] for k, v in data.items()}
                for i in range(0, N, batch_size)]

    factor = PoissonFactorization(latent_dim=P, feature_dim=D, u_tau_scale=1.0 / np.sqrt(N * D),
                                  dtype=np.float64)
    factor.compute_scales(data_factory=data_factory)
    losses = factor.fit(batched_data_factory=data_factory, dataset_size=N, batch_size=1000,
                        sample_size=20, sample_batches=4, num_steps=6, rel_tol=1e-4,
                        learning_rate=.01, verbose=False)
    assert len(losses) >= 2 and all(math.isfinite(v) for v in losses)
    assert losses[-1] < losses[0]


def _fresh_model(X, K=2, seed=11):
    from spmf_amd import PoissonFactorization
    N, D = X.shape
    torch.manual_seed(seed)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device="cuda", panel_rows=100)
    m.compute_scales(lambda: [{"counts": X}])
    return m


def test_device_gated_step_equals_host_driven_step():
    """spmf_vi_gate + spmf_adam_step_dev (no host read-back) == elbo_step +
    spmf_adam_step with the host deciding, on the same noise."""
    from spmf_amd.vi import AdamHIP, elbo_step, vi_step_dev
    X = _data(300, 18)
    N = X.shape[0]
    batch = {"counts": X}
    a, b = _fresh_model(X), _fresh_model(X)
    oa = AdamHIP(a, a.surrogate_distribution.trainable_variables, 0.05)
    ob = AdamHIP(b, b.surrogate_distribution.trainable_variables, 0.05)
    ob.init_state(clip_value=10.0)
    ref_losses = []
    for step in range(3):
        torch.manual_seed(100 + step)
        loss, grads, nnf = elbo_step(a, batch, N, 3)
        oa.step(grads, 10.0)
        ref_losses.append(float(loss))
        torch.manual_seed(100 + step)
        vi_step_dev(b, ob, batch, N, 3)
        st = ob.read_state()
        assert st[9] == 1.0 and st[7] == step + 1
        assert abs(st[8] - ref_losses[-1]) <= 1e-6 * abs(ref_losses[-1])   # float-atomic order differs run to run
    assert abs(st[10] - sum(ref_losses)) <= 1e-6 * abs(sum(ref_losses)) and st[11] == 3 and st[12] == 0
    for p, q in zip(a.surrogate_distribution.trainable_variables,
                    b.surrogate_distribution.trainable_variables):
        assert (p.detach() - q.detach()).abs().max() <= 1e-6 * max(1.0, float(p.detach().abs().max()))


def test_fused_chain_rule_adam_equals_the_two_separate_kernels():
    """spmf_surrogate_bwd_adam_dev (the step path) against spmf_surrogate_bwd +
    spmf_adam_step_dev (the keep= path): trainables AND both Adam moments after three
    steps on the same Philox key."""
    from spmf_amd.vi import AdamHIP, vi_step_dev
    X = _data(300, 18)
    N = X.shape[0]
    batch = {"counts": X}
    a, b = _fresh_model(X), _fresh_model(X)
    oa = AdamHIP(a, a.surrogate_distribution.trainable_variables, 0.05)
    ob = AdamHIP(b, b.surrogate_distribution.trainable_variables, 0.05)
    oa.init_state(clip_value=3.0)
    ob.init_state(clip_value=3.0)
    for step in range(3):
        vi_step_dev(a, oa, batch, N, 2, seed=1234)              # fused
        vi_step_dev(b, ob, batch, N, 2, keep={}, seed=1234)     # separate kernels
    assert oa.read_state()[7] == ob.read_state()[7] == 3
    for name, xs, ys in (("p", oa.params, ob.params), ("m", oa.m, ob.m), ("v", oa.v, ob.v)):
        for x, y in zip(xs, ys):
            x, y = x.detach(), y.detach()
            assert (x - y).abs().max() <= 2e-6 * max(float(y.abs().max()), 1e-30), name


def test_device_gate_skips_non_finite_step():
    from spmf_amd.vi import AdamHIP, vi_step_dev
    X = _data(200, 12)
    N = X.shape[0]
    m = _fresh_model(X)
    opt = AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.05)
    opt.init_state(None)
    sur = m.surrogate_distribution
    before = [p.detach().clone() for p in sur.trainable_variables]
    t0, _ = sur.params_of("v")
    saved = t0.detach().clone()
    with torch.no_grad():
        t0.fill_(float("nan"))                     # poisons theta -> loss is NaN
    vi_step_dev(m, opt, {"counts": X}, N, 2)
    st = opt.read_state()
    assert st[9] == 0.0 and st[7] == 0 and st[12] == 1 and st[11] == 0
    with torch.no_grad():
        t0.copy_(saved)
    for p, q in zip(sur.trainable_variables, before):
        assert torch.equal(p, q)                   # skipped: nothing moved, moments untouched
    assert all(float(mm.abs().max()) == 0.0 for mm in opt.m)


def test_graph_replay_runs_the_same_step():
    """StepRunner: eager on first sight, hipGraph capture on the second, replay
    after; a replayed step's own tensors are self-consistent with an eager
    evaluation, and training under replay behaves like the eager loop."""
    from spmf_amd import SparseCounts
    from spmf_amd.vi import AdamHIP, StepRunner
    X = _data()
    N, D = X.shape
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    m = _fresh_model(X)
    opt = AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.05)
    opt.init_state(10.0)
    run = StepRunner(m, opt, N, 2, use_graph=True)
    run.keep_tensors = True
    for ep in range(4):
        for b in batches:
            run.step(b)
    assert len(run.graphs) == len(batches) and run.replays == 3 * len(batches)
    st = opt.read_state()
    assert st[7] == 4 * len(batches) and st[12] == 0
    # the last replay of the last batch: its theta -> parts / gradient must equal an eager call
    keep = next(reversed(run.kept.values()))
    theta = {k: v.clone() for k, v in keep["theta"].items()}
    B = m._batch(batches[-1])[1].n_rows
    parts_g = keep["parts"].clone()
    g_g = {k: v.clone() for k, v in keep["g"].items()}
    parts_e, g_e, _ = m.energy_and_grads(batches[-1], theta, prior_weight=B / N)
    from spmf_amd._lib import PART_ORDER
    for i, n in enumerate(PART_ORDER):
        assert abs(float(parts_g[0, i]) - float(parts_e[n][0])) <= 1e-9 * max(1.0, abs(float(parts_e[n][0])))
    for k in g_e:
        assert (g_g[k] - g_e[k]).abs().max() <= 1e-6 * max(1e-30, float(g_e[k].abs().max()))
    # noise differs between replays (the philox offset advances inside the graph)
    n1 = keep["noise"]["u"][0].clone()
    run.step(batches[-1])
    assert not torch.equal(n1, keep["noise"]["u"][0])


def test_fit_with_and_without_graph_agree():
    from spmf_amd import SparseCounts
    X = _data()
    N, D = X.shape
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    out = []
    for use_graph in (False, True):
        m = _fresh_model(X)
        torch.manual_seed(0)
        out.append(m.fit(lambda: batches, dataset_size=N, sample_size=4, num_steps=25,
                         learning_rate=0.05, rel_tol=1e-9, verbose=False, use_graph=use_graph))
    a, b = out
    assert len(a) == len(b) == 25
    # same model, same optimiser, independent noise streams: the curves track each other
    assert abs(np.mean(a[-5:]) - np.mean(b[-5:])) < 0.05 * abs(np.mean(a[-5:]))
    assert np.mean(b[-3:]) < b[0] - 0.5


def test_sharded_device_loop_one_rank_rccl_graph_replay_equals_unsharded():
    """The row-sharded VI step ON THE DEVICE (vi.vi_step_dev with a ShardReducer): data pass on
    the shard, spmf_allreduce through the library's RCCL communicator, finish / gate / chain rule /
    Adam on the all-reduced values -- captured in a hipGraph and replayed (the collective is a
    plain stream-ordered ncclAllReduce).  With one rank the sum is the identity, so the run must
    reproduce the unsharded device loop on the same Philox key, step for step, and must never go
    through the host-driven elbo_step."""
    from spmf_amd import SparseCounts, vi
    from spmf_amd.dist import LibraryComm, ShardReducer
    X = _data()
    N, D = X.shape
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]

    def run(reducer_of, use_graph, forked=None):
        m = _fresh_model(X)
        red = reducer_of(m)
        if red is not None and forked is not None:
            red.overlap_prior = forked
        opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.05)
        opt.init_state(10.0)
        r = vi.StepRunner(m, opt, N, 2, use_graph=use_graph, all_reduce=red, seed=4242)
        losses = []
        for ep in range(4):
            for b in batches:
                r.step(b)
                losses.append(opt.read_state()[8])
        st = opt.read_state()
        assert st[7] == 4 * len(batches) and st[12] == 0
        return losses, [p.detach().clone() for p in m.surrogate_distribution.trainable_variables], r

    base, pb, _ = run(lambda m: None, True)
    for forked in (False, True):       # prior half inline / forked to the side stream under the collective
        got, pg, r = run(lambda m: ShardReducer(comm=LibraryComm(m, rank=0, world=1)), True, forked)
        assert r.use_graph and len(r.graphs) == len(batches) and r.replays == 3 * len(batches)
        np.testing.assert_allclose(got, base, rtol=2e-6)
        for a, b in zip(pg, pb):
            assert float((a - b).abs().max()) <= 2e-5 * max(1e-30, float(b.abs().max()))
    # a reducer without the library communicator (torch.distributed / gloo transport): the same
    # device-gated step, eagerly
    got, pg, r = run(lambda m: ShardReducer(), False)
    np.testing.assert_allclose(got, base, rtol=2e-6)


def test_sharded_fit_stays_on_the_device_loop(monkeypatch):
    """fit(all_reduce=ShardReducer) takes the device-resident loop (no per-step host read-back):
    the host-driven elbo_step is never called; a foreign hook still takes the eager loop."""
    from spmf_amd import vi
    from spmf_amd.dist import LibraryComm, ShardReducer
    X = _data(300, 18)
    N = X.shape[0]
    m = _fresh_model(X)
    calls = []
    real = vi.elbo_step
    monkeypatch.setattr(vi, "elbo_step", lambda *a, **k: calls.append(1) or real(*a, **k))
    red = ShardReducer(comm=LibraryComm(m, rank=0, world=1))
    losses = m.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=2, num_steps=6,
                   learning_rate=0.05, rel_tol=1e-12, verbose=False, all_reduce=red)
    assert len(losses) == 6 and not calls and all(np.isfinite(losses))
    m2 = _fresh_model(X)

    class Foreign:                 # a caller's own hook: sums nothing (one shard), knows the totals
        def __call__(self, acc, rows, lg):
            return rows, lg

        def totals(self, rows, lg):
            return rows, lg
    m2.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=2, num_steps=2, learning_rate=0.05,
           rel_tol=1e-12, verbose=False, all_reduce=Foreign())
    assert len(calls) == 2


def test_checkpoint_roundtrip(tmp_path):
    """factor.save -> pickle -> reconstitute BY POSITION (poisson.py:711-717):
    the restored model encodes identically."""
    import pickle
    from spmf_amd import PoissonFactorization
    X = _data(300, 12)
    m = _fresh_model(X)
    torch.manual_seed(3)
    m.fit(lambda: [{"counts": X}], dataset_size=X.shape[0], sample_size=2, num_steps=5,
          learning_rate=0.05, verbose=False)
    f = tmp_path / "model.pkl"
    m.save(str(f))
    state = pickle.load(open(f, "rb"))
    assert len(state["surrogate_vars"]) == 24 and list(state["var_list"]) == list(m.var_list)
    m2 = PoissonFactorization(latent_dim=m.latent_dim, feature_dim=m.feature_dim,
                              u_tau_scale=m.u_tau_scale, device="cuda", panel_rows=100)
    m2.reconstitute(state)
    for p, q in zip(m.surrogate_distribution.trainable_variables,
                    m2.surrogate_distribution.trainable_variables):
        assert torch.equal(p.detach(), q.detach())
    m2.eta_i, m2.xi_u_global = m.eta_i, m.xi_u_global
    torch.manual_seed(9)
    m.set_calibration_expectations()
    torch.manual_seed(9)
    m2.set_calibration_expectations()
    z1, z2 = m.encode(X), m2.encode(X)
    assert float((z1 - z2).abs().max()) <= 1e-6 * float(z1.abs().max())
    bad = dict(state, surrogate_vars=state["surrogate_vars"][:-1])
    with pytest.raises(ValueError):
        m2.reconstitute(bad)


def test_hip_sampler_statistics_and_implicit_gradient():
    """spmf_sample_noise: eps ~ N(0,1) and g ~ Gamma(a,1) pass moment and KS checks,
    the implicit-reparameterisation derivative dg/da equals (i) the finite-difference
    derivative of the inverse CDF at fixed probability (scipy, fp64) and (ii) torch's
    _standard_gamma_grad; draws are reproducible under torch.manual_seed, differ from
    call to call, and advance with the device step counter."""
    import scipy.special as sps
    import scipy.stats as sst
    from spmf_amd import PoissonFactorization
    m = PoissonFactorization(latent_dim=8, feature_dim=600, u_tau_scale=0.01, device="cuda",
                             panel_rows=64)
    sur = m.surrogate_distribution
    with torch.no_grad():     # spread the concentrations: a = softplus(t0) in ~[0.3, 6]
        t0, _ = sur.params_of("u_eta")
        t0.copy_(torch.linspace(-1.0, 6.0, t0.numel(), device=t0.device).view_as(t0))
    S = 16
    torch.manual_seed(123)
    n1 = sur.draw_noise(S)
    torch.manual_seed(123)
    n2 = sur.draw_noise(S)
    n3 = sur.draw_noise(S)
    for k in n1:
        assert torch.equal(n1[k][0], n2[k][0]) and not torch.equal(n1[k][0], n3[k][0]), k
    eps = n1["u"][0].flatten().double().cpu().numpy()
    assert abs(eps.mean()) < 0.02 and abs(eps.std() - 1.0) < 0.02
    assert sst.kstest(eps[:20000], "norm").pvalue > 1e-3
    g = n1["u_eta"][0].double().cpu().numpy().reshape(S, -1)
    dg = n1["u_eta"][1].double().cpu().numpy().reshape(S, -1)
    a = torch.nn.functional.softplus(sur.params_of("u_eta")[0].detach()).double().cpu().numpy().reshape(-1)
    # moments per concentration bucket
    for lo, hi in ((0.3, 1.0), (1.0, 3.0), (3.0, 6.1)):
        sel = (a >= lo) & (a < hi)
        z = (g[:, sel] - a[sel]) / np.sqrt(a[sel])           # standardised: mean 0, var 1
        assert abs(z.mean()) < 0.03 and abs(z.var() - 1.0) < 0.06, (lo, hi, z.mean(), z.var())
    j = np.argmin(np.abs(a - 2.0))
    assert sst.kstest(np.concatenate([n["u_eta"][0].double().cpu().numpy().reshape(S, -1)[:, j - 40:j + 40].ravel()
                                      / 1.0 for n in (n1, n3)]), "gamma", args=(a[j],)).pvalue > 1e-4
    # implicit gradient: d/da of the quantile at fixed probability
    idx = np.linspace(0, a.size - 1, 60).astype(int)
    for i in idx:
        ai, gi = a[i], g[0, i]
        pr = sps.gammainc(ai, gi)
        if not (1e-6 < pr < 1 - 1e-9):
            continue
        h = 1e-5 * ai
        fd = (sps.gammaincinv(ai + h, pr) - sps.gammaincinv(ai - h, pr)) / (2 * h)
        assert abs(dg[0, i] - fd) <= 2e-4 * max(abs(fd), 1e-3), (ai, gi, dg[0, i], fd)
    ref = torch._standard_gamma_grad(torch.as_tensor(np.broadcast_to(a, g.shape).copy()),
                                     torch.as_tensor(g)).numpy()
    ok = np.abs(dg - ref) <= 2e-3 * np.maximum(np.abs(ref), 1e-2)
    assert ok.mean() > 0.999
    # device step counter: same key, different counter -> different noise; same counter -> same
    st = torch.zeros(16, dtype=torch.float64, device="cuda")
    a0 = sur.draw_noise(2, seed=77, state=st)["v"][0].clone()
    a1 = sur.draw_noise(2, seed=77, state=st)["v"][0].clone()
    st[13] = 5.0
    a2 = sur.draw_noise(2, seed=77, state=st)["v"][0].clone()
    assert torch.equal(a0, a1) and not torch.equal(a0, a2)


@pytest.mark.parametrize("use_graph", [False, True])
def test_vi_step_with_the_hierarchy_beside_the_column_pass_equals_the_single_stream_step(monkeypatch, use_graph):
    """vi.vi_step_dev draws + transforms v, w, u, s, issues the data pass, and draws + transforms the eight
    variables of the scale hierarchy and runs the prior half of the finish on a side stream that starts when
    the row pass is done (spmf_ctx_set_rows_event).  Same draws (the sampler's counter does not depend on which
    call covers a variable), same kernels: in the deterministic mode the trainables after a few steps are the
    same BITS as without the split (the default: one stream), eager and replayed from a hipGraph
    that holds the fork and the join; the loss agrees to the last digits (log q is added up in two pieces)."""
    from spmf_amd import PoissonFactorization, SparseCounts
    from spmf_amd.vi import AdamHIP, StepRunner
    X = _data()
    N, D = X.shape
    sc = SparseCounts.from_any(X, "cuda", 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    finals, losses = [], []
    for overlap in ("1", "0"):
        monkeypatch.setenv("SPMF_VI_OVERLAP", overlap)
        torch.manual_seed(3)
        m = PoissonFactorization(latent_dim=3, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda",
                                 panel_rows=100, deterministic=True)
        m.compute_scales(lambda: [{"counts": X}])
        opt = AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.05)
        opt.init_state(10.0)
        run = StepRunner(m, opt, N, 2, use_graph=use_graph, seed=99)
        for ep in range(3):
            for b in batches:
                run.step(b)
        torch.cuda.synchronize()
        st = opt.read_state()
        assert st[11] == 3 * len(batches) and st[12] == 0
        if use_graph:
            assert run.replays == 2 * len(batches)
        assert (getattr(m, "_vi_side", None) is not None) == (overlap == "1")
        finals.append([t.detach().clone() for t in m.surrogate_distribution.trainable_variables])
        losses.append(st[10])
    for a, b in zip(*finals):
        assert torch.equal(a, b)
    assert abs(losses[0] - losses[1]) <= 1e-12 * abs(losses[1])


@pytest.mark.parametrize("D,K,S", [(24, 2, 3), (700, 32, 2), (5000, 16, 1)])
def test_sample_transform_in_one_launch_equals_the_two_calls(D, K, S):
    """spmf_sample_transform (the VI step's path since round 5): base noise, theta and log q in ONE launch --
    the per-workgroup log-q sums folded by the last workgroup to arrive -- give the same draws and the same theta
    BITS as spmf_sample_noise + spmf_surrogate_fwd, and the same log q to fp64 rounding (its partial sums group 256
    elements, the two-call form 1024); repeated (the arrival ticket resets), with the device step counter, and
    for a subset of the variables."""
    from spmf_amd import PoissonFactorization
    from spmf_amd import vi
    torch.manual_seed(4)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-3, device="cuda")
    sur = m.surrogate_distribution
    with torch.no_grad():
        for p in sur.trainable_variables:
            p.add_(0.05 * torch.randn_like(p))
    opt = vi.AdamHIP(m, sur.trainable_variables, 1e-3)
    opt.init_state(None)
    opt.state[13] = 7.0                                  # seven steps gated so far
    for rep, (seed, state, only) in enumerate(((11, None, None), (11, None, None), (12, opt.state, None),
                                               (13, None, frozenset(("u", "u_eta", "s_tau_a"))))):
        n1 = sur.draw_noise(S, seed=seed, state=state, only=only)
        th1 = {n: torch.zeros(n1[n][0].shape, dtype=torch.float32, device="cuda") for n in sur.var_order}
        _, lq1 = sur.forward_hip(m, S, n1, only=only, theta=th1)
        n2, th2, lq2 = sur.draw_and_forward(m, S, seed=seed, state=state, only=only)
        torch.cuda.synchronize()
        assert float((lq1 - lq2).abs().max()) <= 1e-13 * float(lq1.abs().max()), (rep, lq1, lq2)
        for n in (only or sur.var_order):
            assert torch.equal(n1[n][0], n2[n][0]), (rep, n)
            if n1[n][1] is not None:
                assert torch.equal(n1[n][1], n2[n][1]), (rep, n, "dgda")
            assert torch.equal(th1[n], th2[n]), (rep, n, "theta")
