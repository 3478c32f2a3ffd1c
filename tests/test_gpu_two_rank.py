"""GPU: the row-sharded path END TO END with two processes (gloo as the transport,
both ranks on the one visible card): each rank owns a row shard as a resident
SparseCounts, the HIP data pass runs on it, ShardReducer sums the packed
accumulators, the finish kernel runs redundantly -- and the energy, the gradients,
the sharded VI step and a short sharded fit must equal the single-process results
on the whole matrix.  (The 8-GPU run over RCCL is the driver's; this is the same
code with a host-staged transport.)"""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    rng = np.random.default_rng(31)
    N, D = 512, 40
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    X[:, ::4] = rng.poisson(np.abs(rng.normal(0, 1, size=(N, 2))) @ np.abs(rng.normal(1.5, .5, size=(2, D // 4))))
    return X


def _model(D, N):
    from spmf_amd import PoissonFactorization
    torch.manual_seed(5)
    return PoissonFactorization(latent_dim=3, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                                device="cuda", panel_rows=64)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from spmf_amd import SparseCounts
    from spmf_amd.dist import ShardReducer, shard_bounds, sync_seed
    from spmf_amd.vi import elbo_step
    X = _data()
    N, D = X.shape
    r0, r1 = shard_bounds(N, world, rank, granule=64)
    m = _model(D, N)
    sc = SparseCounts.from_any(X[r0:r1], "cuda", 64)
    # compute_scales over the shards: one reduction of the column statistics
    red = ShardReducer()
    m.compute_scales(lambda: [{"counts": sc}], all_reduce=red)
    rows_g = red.dataset_rows
    sync_seed(77)
    params = m.surrogate_distribution.sample(2)
    parts, grads, _ = m.energy_and_grads({"counts": sc}, params, all_reduce=red)
    sync_seed(78)
    loss, g, _ = elbo_step(m, {"counts": sc}, rows_g, 2, all_reduce=red)
    sync_seed(79)
    # the sharded fit is the device-resident loop (vi.vi_step_dev with the reducer; gloo: eager,
    # no hipGraph); the host-driven step must not be what runs
    from spmf_amd import vi as _vi

    def _no_host_step(*a, **k):
        raise AssertionError("fit(all_reduce=ShardReducer) fell back to the host-driven elbo_step")
    _vi.elbo_step = _no_host_step
    losses = m.fit(lambda: [{"counts": sc}], dataset_size=rows_g, sample_size=2, num_steps=4,
                   learning_rate=0.02, rel_tol=1e-12, verbose=False, all_reduce=red, sync_every=2)
    drift = red.replicas_max_abs_diff(m.surrogate_distribution.trainable_variables)
    if rank == 0:
        q.put({"rows": rows_g, "eta": m.eta_i.cpu().numpy(), "xi": m.xi_u_global,
               "parts": {k: v.cpu().numpy() for k, v in parts.items()},
               "grads": {k: v.cpu().numpy() for k, v in grads.items()},
               "loss": float(loss), "g": [t.cpu().numpy() for t in g], "losses": losses, "drift": drift})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_energy_step_and_fit_equal_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=500)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process on the whole matrix, same seeds
    from spmf_amd.vi import elbo_step
    X = _data()
    N, D = X.shape
    m = _model(D, N)
    m.compute_scales(lambda: [{"counts": X}])
    assert res["rows"] == N and abs(res["xi"] - m.xi_u_global) <= 1e-9 * m.xi_u_global
    np.testing.assert_allclose(res["eta"], m.eta_i.cpu().numpy(), rtol=1e-12)
    torch.manual_seed(77)
    params = m.surrogate_distribution.sample(2)
    parts, grads, _ = m.energy_and_grads({"counts": X}, params)
    for k, v in parts.items():
        np.testing.assert_allclose(res["parts"][k], v.cpu().numpy(), rtol=1e-6, err_msg=k)
    for k, v in grads.items():
        a, b = res["grads"][k], v.cpu().numpy()
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), k
    torch.manual_seed(78)
    loss, g, _ = elbo_step(m, {"counts": X}, N, 2)
    assert abs(res["loss"] - float(loss)) <= 1e-6 * abs(float(loss))
    for a, b in zip(res["g"], g):
        b = b.cpu().numpy()
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-30)
    torch.manual_seed(79)
    from spmf_amd.dist import ShardReducer
    # an inactive reducer (no process group): the same device-resident loop the shards ran, on
    # the whole matrix (here replayed from a hipGraph; the Philox key is drawn the same way)
    losses = m.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=2, num_steps=4,
                   learning_rate=0.02, rel_tol=1e-12, verbose=False, all_reduce=ShardReducer())
    assert len(res["losses"]) == len(losses) == 4
    # same noise stream (same seed), same data: the sharded fit follows the single-process fit
    np.testing.assert_allclose(res["losses"], losses, rtol=2e-4)
    assert res["drift"] == 0.0


def _rule_worker(rank, world, port, q, flip):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from spmf_amd.dist import ShardReducer, shard_bounds
    from test_gpu_parity import build_model
    from test_gpu_rule_and_surface import _bad_cell_problem
    cfg, x, params = _bad_cell_problem()
    if flip:
        x = x[::-1].copy()          # the rate-0 cell (and the minimum's cell) move to the other shard
    r0, r1 = shard_bounds(x.shape[0], world, rank, granule=8)
    m = build_model(cfg, 8)
    red = ShardReducer()
    parts, grads, nnf = m.energy_and_grads({"counts": x[r0:r1]}, params, all_reduce=red, nonfinite="rule")
    if rank == 1:                   # the replicas must agree: report the one not checked elsewhere
        q.put({"parts": {k: v.cpu().numpy() for k, v in parts.items()},
               "grads": {k: v.cpu().numpy() for k, v in grads.items()}, "nnf": nnf.cpu().numpy()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flip", [False, True])
def test_two_rank_nonfinite_rule_equals_single_process(flip):
    """poisson.py:606-616 across row shards: minimum over the shard minima, the
    gradient term from the shard that holds the minimum's cell, value terms per
    shard -- against the single-process rule (itself pinned to the oracle in
    test_gpu_rule_and_surface.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rule_worker, args=(r, 2, port, q, flip)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=500)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import build_model
    from test_gpu_rule_and_surface import _bad_cell_problem
    cfg, x, params = _bad_cell_problem()
    if flip:
        x = x[::-1].copy()
    m = build_model(cfg, 8)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params, nonfinite="rule")
    assert res["nnf"].tolist() == nnf.cpu().tolist() == [1.0, 0.0]
    for k, v in parts.items():
        np.testing.assert_allclose(res["parts"][k], v.cpu().numpy(), rtol=1e-6, err_msg=k)
    for k, v in grads.items():
        a, b = res["grads"][k], v.cpu().numpy()
        assert np.isfinite(a).all(), k
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), (k, np.abs(a - b).max(), np.abs(b).max())


def _custom_model(cfg):
    from spmf_amd import PoissonFactorization
    m = PoissonFactorization(latent_dim=cfg.latent_dim, feature_dim=cfg.feature_dim, u_tau_scale=cfg.u_tau_scale,
                             column_norms=cfg.eta_i, initialize_distributions=False, device="cuda", panel_rows=8,
                             encoder_function=torch.sqrt, decoder_function=lambda y: y * y)
    m.xi_u_global = cfg.xi_u_global
    return m


def _custom_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from spmf_amd.dist import ShardReducer, shard_bounds
    from test_gpu_parity import make_problem
    cfg, x, params = make_problem(48, 20, 3, 2, 41, 0.3)
    r0, r1 = shard_bounds(x.shape[0], world, rank, granule=8)
    m = _custom_model(cfg)
    parts, grads, nnf = m.energy_and_grads({"counts": x[r0:r1]}, params, all_reduce=ShardReducer())
    if rank == 1:
        q.put({"parts": {k: v.cpu().numpy() for k, v in parts.items()},
               "grads": {k: v.cpu().numpy() for k, v in grads.items()}, "nnf": nnf.cpu().numpy()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_custom_callables_equal_single_process():
    """poisson.py:94-97 callables over row shards (spmf_amd/custom_codec.py _sharded): the class
    surface with a ShardReducer against the same model on the whole matrix."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_custom_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=500)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import make_problem
    cfg, x, params = make_problem(48, 20, 3, 2, 41, 0.3)
    parts, grads, nnf = _custom_model(cfg).energy_and_grads({"counts": x}, params)
    assert res["nnf"].tolist() == nnf.cpu().tolist() == [0.0, 0.0]
    for k, v in parts.items():
        np.testing.assert_allclose(res["parts"][k], v.cpu().numpy(), rtol=2e-7, err_msg=k)
    for k, v in grads.items():
        a, b = res["grads"][k], v.cpu().numpy()
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), (k, np.abs(a - b).max(), np.abs(b).max())


def _det_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from spmf_amd import PoissonFactorization, SparseCounts
    from spmf_amd.dist import ShardReducer, shard_bounds, sync_seed
    X = _data()
    N, D = X.shape
    r0, r1 = shard_bounds(N, world, rank, granule=64)
    torch.manual_seed(5)
    m = PoissonFactorization(latent_dim=3, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda",
                             panel_rows=64, deterministic=True)
    sc = SparseCounts.from_any(X[r0:r1], "cuda", 64)
    red = ShardReducer()
    m.compute_scales(lambda: [{"counts": sc}], all_reduce=red)
    sync_seed(79)
    calls = []
    orig = red.sync_replicas
    red.sync_replicas = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    losses = m.fit(lambda: [{"counts": sc}], dataset_size=red.dataset_rows, sample_size=2, num_steps=14,
                   learning_rate=0.02, rel_tol=1e-12, verbose=False, all_reduce=red, sync_every=2)
    n_sync = len(calls)                         # (the diagnostic below broadcasts through sync_replicas itself)
    drift = red.replicas_max_abs_diff(m.surrogate_distribution.trainable_variables)
    if rank == 0:
        q.put({"losses": losses, "drift": drift, "syncs": n_sync})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_deterministic_replicas_stay_identical_without_the_guard_broadcast():
    """deterministic=True: every rank applies bit-identical updates from the same all-reduced buffer, so
    `fit` drops ShardReducer.sync_replicas (asked for every 2 steps here, never called) and the replicas
    still agree exactly after 14 steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_det_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=500)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res["syncs"] == 0 and res["drift"] == 0.0 and len(res["losses"]) == 14
    assert all(math.isfinite(v) for v in res["losses"])
