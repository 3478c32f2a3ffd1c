"""Deterministic mode (spmf_ctx_set_deterministic; PoissonFactorization(deterministic=True)): the float
atomics of the column pass and the fp64 atomics of the row pass replaced by single-writer partial sums
added in a fixed order.  Bar: bit-identical parts and gradients from run to run, and the same numbers as
the default path to fp32 rounding (the default path is the one checked against the oracle everywhere)."""
import contextlib
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def _model(K, D, rows, deterministic, **kw):
    from spmf_amd import PoissonFactorization
    with contextlib.redirect_stdout(sys.stderr):
        return PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                    device="cuda", deterministic=deterministic, **kw)


def _counts(rows, D, density, seed, dense_col=False):
    rng = np.random.default_rng(seed)
    X = sp.random(rows, D, density=density, format="csr", random_state=rng,
                  data_rvs=lambda n: rng.poisson(2.0, n) + 1.0)
    if dense_col:
        X = X.tolil()
        X[:, 3] = rng.poisson(3.0, rows) + 1.0          # a column present in every row: lists of many segments
        X = X.tocsr()
    return X


def _run(m, batch, params):
    parts, grads, nnf = m.energy_and_grads(batch, params)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in parts.items()}, {k: v.clone() for k, v in grads.items()}, nnf.clone()


@pytest.mark.parametrize("rows,D,K,S,P,density,dense_col", [
    (3000, 400, 8, 1, 256, 0.03, False),
    (5000, 700, 32, 2, 512, 0.02, True),
    (900, 130, 3, 1, 64, 0.1, False),
    (2500, 257, 50, 1, 1024, 0.05, True),
])
def test_deterministic_mode_is_bit_reproducible_and_equals_the_default_path(rows, D, K, S, P, density, dense_col):
    from spmf_amd.sparse import SparseCounts
    X = _counts(rows, D, density, seed=rows + K, dense_col=dense_col)
    dev = torch.device("cuda", 0)
    sc = SparseCounts.from_any(X, dev, P)
    outs = {}
    for det in (True, False):
        m = _model(K, D, rows, det, panel_rows=P)
        colsum = torch.zeros(D, dtype=torch.float64, device=dev)
        colnnz = torch.zeros_like(colsum)
        sc.compute_stats(m._handle(), colsum, colnnz)
        cm = colsum / colnnz.clamp_min(1.0)
        m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
        m.xi_u_global = float(cm.sum())
        torch.manual_seed(11)
        params = m.surrogate_distribution.sample(S)
        runs = [_run(m, {"counts": sc}, params) for _ in range(3)]
        # a panel range of the same resident shard (a minibatch) through the same mode
        mb = [_run(m, {"counts": sc, "panels": (1, 3)}, params) for _ in range(2)]
        outs[det] = (runs, mb)
    for runs in outs[True]:
        p0, g0, _ = runs[0]
        for p1, g1, _ in runs[1:]:
            for k in p0:
                assert torch.equal(p0[k], p1[k]), ("part", k)
            for k in g0:
                assert torch.equal(g0[k], g1[k]), ("grad", k)
    # same numbers as the default (atomic) path, to rounding
    for a, b in zip(outs[True], outs[False]):
        pa, ga, na = a[0]
        pb, gb, nb = b[0]
        assert torch.equal(na, nb)
        for k in pa:
            torch.testing.assert_close(pa[k], pb[k], rtol=1e-11, atol=1e-9, msg=lambda s_, k=k: f"part {k}: {s_}")
        for k in ga:
            d = float((ga[k] - gb[k]).abs().max())
            assert d <= 2e-6 * max(float(gb[k].abs().max()), 1e-30), (k, d)


def test_deterministic_fit_repeats_exactly_and_unsupported_contexts_say_so():
    from spmf_amd import BernoulliFactorization, _lib
    rows, D, K = 1200, 90, 4
    X = _counts(rows, D, 0.08, seed=5)
    losses = []
    for rep in range(2):
        m = _model(K, D, rows, True, panel_rows=256)
        torch.manual_seed(3)
        m.create_distributions()
        torch.manual_seed(4)
        losses.append(m.fit(lambda: [{"counts": X}], dataset_size=rows, sample_size=2, num_steps=12,
                            learning_rate=0.05, rel_tol=1e-12, verbose=False))
    assert losses[0] == losses[1] and len(losses[0]) >= 5
    with contextlib.redirect_stdout(sys.stderr):
        b = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=0.01, device="cuda", deterministic=True)
    torch.manual_seed(1)
    with pytest.raises(_lib.SpmfError, match="linear decoder only"):
        b.energy_and_grads({"counts": (X > 0).astype(np.float64)}, b.surrogate_distribution.sample(1))


def test_deterministic_minibatch_training_under_graph_replay_repeats_exactly():
    """The device-gated VI loop over panel-range minibatches of growing size (the scratch is re-sized between
    captured steps), replayed from hipGraphs: two runs with the same seeds end in the same trainables, bit for bit."""
    from spmf_amd import vi
    from spmf_amd.sparse import SparseCounts
    rows, D, K = 4000, 300, 8
    X = _counts(rows, D, 0.04, seed=17)
    dev = torch.device("cuda", 0)
    finals = []
    for rep in range(2):
        sc = SparseCounts.from_any(X, dev, 250)
        m = _model(K, D, rows, True, panel_rows=250)
        torch.manual_seed(21)
        m.create_distributions()
        colsum = torch.zeros(D, dtype=torch.float64, device=dev)
        colnnz = torch.zeros_like(colsum)
        sc.compute_stats(m._handle(), colsum, colnnz)
        batches = [{"counts": sc, "panels": (0, 2)}, {"counts": sc, "panels": (2, 7)}, {"counts": sc, "panels": (7, 16)}]
        opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-2)
        opt.init_state(5.0)
        run = vi.StepRunner(m, opt, rows, 2, use_graph=True, seed=99)
        for ep in range(4):
            for b in batches:
                run.step(b)
        torch.cuda.synchronize()
        assert run.replays > 0
        finals.append([p.detach().clone() for p in m.surrogate_distribution.trainable_variables])
    for a, b in zip(*finals):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,D,K,S,density,scale_rows,panel_rows", [
    (37, 23, 3, 2, 0.3, True, 16), (300, 257, 32, 2, 0.04, False, 128), (150, 90, 50, 1, 0.1, True, 32),
    (50, 300, 4, 1, 0.8, True, 16), (1800, 640, 32, 1, 0.03, True, 256),
])
def test_deterministic_mode_against_the_oracle(B, D, K, S, density, scale_rows, panel_rows):
    """The same entry-wise bar as the default path (tests/test_gpu_parity.py): 14 parts to 1e-5, every gradient
    entry to 1e-5 of the sum of the absolute contributions to it, against the fp64 oracle."""
    from oracle import spmf_oracle as O
    from test_gpu_parity import assert_close_grads, assert_close_parts, build_model, make_problem
    cfg, x, params = make_problem(B, D, K, S, 4000 + B + D + K, density, scale_rows)
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, panel_rows)
    m.deterministic = True
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params), tag="deterministic")
