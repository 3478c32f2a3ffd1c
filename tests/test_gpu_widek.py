"""Latent dimensions above 64 (the reference's `latent_dim` defaults to `feature_dim`,
poisson.py:103-104): the whole-wave sparse passes of csrc/widek.hip (K padded to 128 / 256) against
the fp64 oracle at the same bar as every other shape -- 14 energy parts to 1e-5, 12 gradients
entry by entry to 1e-5 of the summed absolute contributions."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from test_gpu_parity import make_problem, build_model, assert_close_parts, assert_close_grads

pytestmark = pytest.mark.gpu

CASES = [
    # B,   D,   K,   S, density, scale_rows, panel_rows
    (70, 90, 65, 1, 0.2, True, 32),       # first K past the lane-group kernels: KP = 128
    (150, 200, 100, 2, 0.1, True, 64),    # two draws (one launch per kernel: small batch)
    (60, 120, 128, 1, 0.3, False, 16),    # K = KP = 128, scale_rows off
    (90, 260, 200, 1, 0.08, True, 32),    # KP = 256
    (40, 300, 256, 1, 0.5, True, 1000),   # K = KP = 256, rows of ~150 entries (several 64-entry chunks), one panel
]


@pytest.mark.parametrize("B,D,K,S,density,scale_rows,panel_rows", CASES)
def test_energy_parts_and_grads_match_oracle_above_k64(B, D, K, S, density, scale_rows, panel_rows):
    cfg, x, params = make_problem(B, D, K, S, 4000 + B + D + K, density, scale_rows)
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, panel_rows)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))


def test_feature_dim_default_latent_dim_and_encode():
    """latent_dim=None is the reference's default (K = D): a 100-column model encodes and evaluates."""
    from spmf_amd import PoissonFactorization
    B, D = 80, 100
    cfg, x, params = make_problem(B, D, D, 1, 4242, 0.15)
    m = PoissonFactorization(latent_dim=None, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             column_norms=cfg.eta_i, initialize_distributions=False, device="cuda",
                             panel_rows=32)
    assert m.latent_dim == D
    m.xi_u_global = cfg.xi_u_global
    T = torch.as_tensor
    z_ref = O.encode(cfg, T(x), T(params["u"]), T(params["s"]))
    z = m.encode(x, u=T(params["u"]), s=T(params["s"]))
    zz = z.detach().cpu().double().numpy().reshape(z_ref.shape)
    np.testing.assert_allclose(zz, z_ref.numpy(), rtol=1e-5, atol=1e-6)
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))


def test_minibatch_of_panels_and_nonfinite_cells_above_k64():
    """A panel-range minibatch (row_base != 0) and cells with a non-positive rate (counted, left out of
    'x' and its gradient) on the K > 64 kernels."""
    B, D, K = 96, 80, 70
    cfg, x, params = make_problem(B, D, K, 1, 4343, 0.2, empty=False)
    m = build_model(cfg, 32)
    xs = x[32:96]
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, xs, params)
    parts, grads, nnf = m.energy_and_grads({"counts": x, "panels": (1, 3)}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, xs, params))
    # negative intercepts on a few columns: stored cells with r <= 0 are counted, not propagated
    p2 = {k: np.array(v, copy=True) for k, v in params.items()}
    p2["w"][..., :5] = -1e7
    parts2, grads2, nnf2 = m.energy_and_grads({"counts": x}, p2)
    stored = int((x[:, :5] > 0).sum())
    assert 0 < int(nnf2.sum()) <= stored
    for g in grads2.values():
        assert torch.isfinite(g).all()


def test_contexts_that_stay_at_k64_say_so():
    from spmf_amd import PoissonFactorization
    from spmf_amd._lib import SpmfError
    with pytest.raises(SpmfError):
        m = PoissonFactorization(latent_dim=65, feature_dim=90, log_transform=True,
                                 initialize_distributions=False, device="cuda")
        m._handle()
    with pytest.raises(SpmfError):
        m = PoissonFactorization(latent_dim=257, feature_dim=300, initialize_distributions=False, device="cuda")
        m._handle()
    cfg, x, params = make_problem(40, 90, 65, 1, 5, 0.2)
    m = PoissonFactorization(latent_dim=65, feature_dim=90, initialize_distributions=False, device="cuda",
                             column_norms=cfg.eta_i, deterministic=True)
    with pytest.raises(SpmfError):
        m.energy_and_grads({"counts": x}, params)


def test_fit_trains_a_model_with_latent_dim_above_64():
    """The whole VI loop (sampler, surrogate, energy + gradient, Adam; hipGraph replay of the small batches)
    at K = 80: the loss falls and the fitted model encodes."""
    from spmf_amd import PoissonFactorization, SparseCounts
    rng = np.random.default_rng(9)
    N, D, K = 400, 96, 80
    Z = np.abs(rng.standard_normal((N, 3)))
    V = np.abs(1.5 + 0.5 * rng.standard_normal((3, D)))
    X = rng.poisson(0.3 * Z @ V).astype(np.float64) * (rng.random((N, D)) < 0.4)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             device="cuda", panel_rows=100)
    sc = SparseCounts.from_any(X, "cuda", 100, latent_dim=K)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    m.compute_scales(lambda: [{"counts": X}])
    torch.manual_seed(0)
    losses = m.fit(lambda: batches, dataset_size=N, sample_size=2, num_steps=30,
                   learning_rate=0.05, rel_tol=1e-9, verbose=False)
    assert len(losses) >= 10 and all(math.isfinite(v) for v in losses)
    assert np.mean(losses[-3:]) < losses[0]
    z = m.encode(X)
    assert tuple(z.shape) == (N, K) and bool(torch.isfinite(z).all())


def test_draws_in_turn_on_a_batch_too_big_for_one_launch_per_kernel():
    """More than 2048 rows and S table pairs above 3 MB: the draws run back to back (api.hip batched_draws),
    each through the K > 64 kernels with its own tables, row outputs and accumulators."""
    B, D, K, S = 2100, 1700, 128, 2
    cfg, x, params = make_problem(B, D, K, S, 4545, 0.01, empty=False)
    parts_ref, grads_ref, _ = O.energy_and_grads(cfg, x, params)
    m = build_model(cfg, 1024)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    assert_close_parts(parts, parts_ref)
    assert_close_grads(grads, grads_ref, O.energy_grad_scales(cfg, x, params))
