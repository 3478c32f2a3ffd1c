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
    from spmf_amd.vi import elbo_step
    X = _data(200, 12)
    N, D = X.shape
    K, S = 3, 2
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device="cuda", panel_rows=64)
    batch = {"counts": X}
    torch.manual_seed(5)
    loss, grads, nnf = elbo_step(m, batch, dataset_rows=N, sample_size=S)
    # same draw again on the reference side: re-seed and rebuild theta with autograd
    torch.manual_seed(5)
    sur = m.surrogate_distribution
    theta, logq = sur.rsample(S)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=m.u_tau_scale)
    th64 = {k: v.double().cpu() for k, v in theta.items()}
    parts = O.unormalized_log_prob_parts(cfg, X, th64)
    prior = sum(parts[n] for n in O.VAR_ORDER)
    c = 1.0   # full batch: B/N = 1
    ref_loss = -(parts["x"] + parts["z"] + c * prior - c * logq.double().cpu()).mean() / N
    ref_grads = torch.autograd.grad(ref_loss, sur.trainable_variables)
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    for g, r in zip(grads, ref_grads):
        r = r.to(g)
        assert (g - r).abs().max() <= 2e-4 * max(float(r.abs().max()), 1e-12)


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
