"""N>1 path on CPU: world_size 2 and 4, gloo.  Each rank owns a row shard,
produces the packed accumulators of S = 2 draws (here by the numpy restatement,
since the HIP data pass needs a GPU), the product's ShardReducer all-reduces
them in ONE call, and the finished energy/gradients of every draw must equal the
unsharded fp64 oracle.  Also rehearsed: the column-split start/wait protocol,
the global batch weighting of the sharded VI step (B_global/N, not the shard's
rows), and the replica sync / drift diagnostic."""
import math
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import spmf_oracle as O
from oracle import sparse_exact as SE


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(77)
    B, D, K = 90, 31, 5
    x = ((rng.random((B, D)) < 0.25) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 3.7
    params = O.random_params(cfg, 2, 78)          # S = 2 draws
    return cfg, x, params


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spmf_amd.dist import ShardReducer, shard_bounds
    cfg, x, params = _problem()
    r0, r1 = shard_bounds(x.shape[0], world, rank, granule=16 if world <= 4 else 8)
    S = params["u"].shape[0]
    draws = [{k: v[i] for k, v in params.items()} for i in range(S)]
    eta = cfg.eta_i.numpy().reshape(-1)
    xs = sp.csr_matrix(x[r0:r1])
    # the real accumulator buffer: [S][acc_len], all draws reduced by one collective
    acc = torch.from_numpy(np.concatenate([
        SE.shard_accumulators(xs, eta, cfg.xi_u_global, True, d["u"], d["v"], d["w"], d["s"])
        for d in draws]))
    from scipy.special import gammaln
    lg = float(gammaln(xs.data + 1.0).sum())
    from spmf_amd.dist import sync_seed
    seed = sync_seed(1234 + rank)          # rank 0's seed wins everywhere
    assert seed == 1234 and torch.initial_seed() == 1234
    red = ShardReducer()
    colsum = torch.from_numpy(np.asarray(xs.sum(0)).reshape(-1).copy())
    colnnz = torch.from_numpy(np.asarray((xs > 0).sum(0)).reshape(-1).astype(np.float64))
    rows_g, lg_g = red.reduce_stats(colsum, colnnz, r1 - r0, lg, full_batch=True)
    # the column-split protocol (start / wait / totals) sums the same buffer in two ranges
    acc2 = acc.clone()
    half = acc2.numel() // 3
    w0 = red.start(acc2[:half])
    w1 = red.start(acc2[half:])
    red.wait(w0)
    red.wait(w1)
    assert red.totals(r1 - r0, lg) == (rows_g, lg_g)
    rg2, lg2 = red(acc, r1 - r0, lg)
    assert (rg2, lg2) == (rows_g, lg_g)
    # (two ranges vs one call: the ring splits the buffer differently, so with more
    #  than two ranks the fp32 sums differ in the last bit)
    assert torch.allclose(acc, acc2, rtol=1e-6, atol=1e-30) and (world > 2 or torch.equal(acc, acc2))
    al = acc.numel() // S
    outs = [SE.finish_from_acc(acc.numpy()[i * al:(i + 1) * al], rows_g, lg_g, eta, d["u"], d["v"],
                               d["w"], d["s"]) for i, d in enumerate(draws)]
    # the sharded VI step weights the batch by its GLOBAL row count
    from spmf_amd.vi import batch_rows_global

    class _CS:
        n_rows, lgamma_sum = r1 - r0, lg
    assert batch_rows_global(_CS, red) == rows_g == x.shape[0]
    assert batch_rows_global(_CS, None) == r1 - r0
    # ---- what the device-resident sharded loop (vi.vi_step_dev with the reducer) leans on -------------
    # rank 0's Philox key on every rank (three float32-exact 21-bit pieces through the sum all-reduce)
    key = (1 << 61) + 123456789012345
    assert red.share_int(key + 17 * rank) == key
    # per-batch global totals: reduced ONCE per batch object, then answered from the cache with no
    # communication (the step must not read anything back)
    red_mb = ShardReducer()                       # minibatched shards: no fixed per-step totals

    class _Batch:
        n_rows, lgamma_sum = r1 - r0, lg
    b0 = _Batch()
    assert red_mb.batch_totals(b0) == (rows_g, lg_g)
    real_sum = red_mb._sum

    def _no_comm(t):
        raise AssertionError("batch_totals communicated again for a batch it has seen")
    red_mb._sum = _no_comm
    assert red_mb.batch_totals(b0) == (rows_g, lg_g)
    red_mb._sum = real_sum
    assert red.batch_totals(b0) == (rows_g, lg_g)          # full-batch reducer: the fixed totals
    # torch.distributed as the transport: the step's collective is not graph-capturable, and the prior
    # half of the finish is forked under it only when there is more than one rank
    assert red.graph_safe is False and red.overlap_prior is (world > 1)
    assert ShardReducer(overlap_prior=False).overlap_prior is False
    # replicas: equal after a sync, and the diagnostic sees a planted difference
    rep = [torch.full((5,), float(rank)), torch.arange(3.0) + rank]
    assert red.replicas_max_abs_diff(rep) == float(world - 1)
    red.sync_replicas(rep)
    assert red.replicas_max_abs_diff(rep) == 0.0 and float(rep[0][0]) == 0.0
    if rank == 0:
        q.put((r0, r1, rows_g, colsum.numpy(), [o["x"] for o in outs], [o["z"] for o in outs],
               [{k: v for k, v in o["grads"].items()} for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_rows_once():
    from spmf_amd.dist import shard_bounds
    for total, world, g in [(1000, 8, 125), (90, 2, 16), (7, 4, 1), (1_000_000, 8, 125_000),
                            (100, 3, 64)]:
        edges = [shard_bounds(total, world, r, g) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        for a, b in zip(edges[:-1], edges[1:]):
            assert a[1] == b[0]


def test_shard_bounds_balance_stored_entries():
    """SURVEY 8e "nnz-balanced split points": with the row pointers, shard edges are the
    granule boundaries nearest to equal stored counts (deep rows first, shallow rows after:
    equal row counts would give rank 0 three times the entries of the last rank)."""
    from spmf_amd.dist import shard_bounds
    rng = np.random.default_rng(3)
    lens = np.concatenate([rng.poisson(300, 5000), rng.poisson(50, 15000)])
    rp = np.concatenate([[0], np.cumsum(lens)])
    total, g = 20000, 1000
    for world in (1, 2, 4, 8):
        edges = [shard_bounds(total, world, r, g, row_ptr=rp) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        for a, b in zip(edges[:-1], edges[1:]):
            assert a[1] == b[0] and a[0] < a[1]
        assert all(e[0] % g == 0 for e in edges)
        nnz = np.array([rp[b] - rp[a] for a, b in edges], dtype=np.float64)
        even = np.array([rp[b] - rp[a] for a, b in
                         (shard_bounds(total, world, r, g) for r in range(world))], dtype=np.float64)
        # never worse than the equal-rows split, and within one granule of deep rows of ideal
        assert nnz.max() <= even.max() + 1e-9
        assert nnz.max() - nnz.mean() <= 300 * g
    # torch row pointers and a list work too; a wrong length is refused
    a = shard_bounds(total, 4, 1, g, row_ptr=torch.as_tensor(rp))
    assert a == shard_bounds(total, 4, 1, g, row_ptr=list(rp))
    with pytest.raises(ValueError):
        shard_bounds(total, 4, 1, g, row_ptr=rp[:-1])
    # fewer granules than ranks: still a cover without overlap (some ranks get no rows)
    edges = [shard_bounds(3000, 4, r, 1000, row_ptr=np.arange(3001)) for r in range(4)]
    assert edges[0][0] == 0 and edges[-1][1] == 3000
    assert all(x[1] == y[0] for x, y in zip(edges[:-1], edges[1:]))


def test_shard_reducer_with_a_library_comm_alone_on_one_rank():
    """with only a LibraryComm (no torch.distributed process group) the reducer
    is active; totals / sync_replicas / replicas_max_abs_diff / gather_scalar must not ask
    torch.distributed for a backend.  The comm here is a stand-in that records calls (the real
    one needs RCCL and a GPU: tests/test_gpu_two_rank.py)."""
    from spmf_amd.dist import ShardReducer
    assert not dist.is_initialized()

    class FakeComm:
        rank, world = 0, 1

        def __init__(self):
            self.calls = 0

        def all_reduce_(self, t):
            self.calls += 1

    comm = FakeComm()
    red = ShardReducer(comm=comm)
    assert red.active and red.world == 1 and red.rank == 0
    assert red.totals(120, 3.5) == (120, 3.5)          # rows_global unset: computed, not raised
    red.set_batch_totals(500, 9.0)
    assert red.totals(120, 3.5) == (500, 9.0)
    t = torch.arange(6, dtype=torch.float32)
    red.sync_replicas([t])                               # nothing to broadcast with one rank
    assert red.replicas_max_abs_diff([t, torch.zeros(0)]) == 0.0
    assert red.gather_scalar(2.5, device=torch.device("cpu")) == [2.5]
    acc = torch.ones(8, dtype=torch.float32)
    assert red(acc, 120, 3.5) == (500, 9.0)
    assert comm.calls == 0                               # host tensors never go to the library comm


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_allreduce_matches_unsharded_oracle(world):
    """(world 8 = the rank count of the driver's scale run: the reducer's protocol, the global
    batch weighting and the device-loop helpers with eight processes over gloo)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1, rows_g, colsum, px, pz, grads = res
    cfg, x, params = _problem()
    from spmf_amd.dist import shard_bounds
    assert rows_g == x.shape[0] and (r0, r1) == shard_bounds(x.shape[0], world, 0, 16 if world <= 4 else 8)
    assert (r0, r1) == {2: (0, 48), 4: (0, 16), 8: (0, 8)}[world]
    np.testing.assert_allclose(colsum, x.sum(0), rtol=1e-12)
    parts, _, groups = O.energy_and_grads(cfg, x, params)
    # accumulators travel as fp32 (that is the wire format): 1e-5 tolerance
    for i in range(len(px)):
        np.testing.assert_allclose(px[i], parts["x"][i].item(), rtol=1e-5)
        np.testing.assert_allclose(pz[i], parts["z"][i].item(), rtol=1e-5)
        for k, g in grads[i].items():
            ref = groups["data"][k][i].numpy()
            assert np.abs(g - ref).max() <= 1e-5 * np.abs(ref).max(), (i, k)


# ---- custom encoder/decoder callables over row shards (spmf_amd/custom_codec.py) -------
class _StubModel:
    """What custom_codec reads of a PoissonFactorization (the class itself needs a GPU)."""

    def __init__(self, cfg, enc, dec):
        self.device = torch.device("cpu")
        self.horseshoe_plus = True
        self.latent_dim = cfg.latent_dim
        self.symmetry_breaking_decay = cfg.symmetry_breaking_decay
        self.u_tau_scale, self.s_tau_scale = cfg.u_tau_scale, cfg.s_tau_scale
        self.scale_rows = cfg.scale_rows
        self.xi_u_global = cfg.xi_u_global
        self._eta = cfg.eta_i
        self._custom_codec = (enc, dec)

    def _eta_device(self):
        return self._eta


def _custom_problem(bad):
    rng = np.random.default_rng(9)
    B, D, K, S = 40, 12, 3, 2
    x = ((rng.random((B, D)) < 0.4) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=True, u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = 3.0
    params = O.random_params(cfg, S, 10, fp32_exact=True)
    if bad:   # a stored cell under rate 0 in draw 0 (row `bad` holds nothing but column 0)
        params["w"][0, 0, 0] = 0.0
        params["u"][0, 0, :] = 0.0
        x[:, 0] = 0
        x[bad, :] = 0
        x[bad, 0] = 3.0
    return cfg, x, params


def _custom_worker(rank, world, port, q, bad):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spmf_amd import custom_codec
    from spmf_amd.dist import ShardReducer, shard_bounds
    cfg, x, params = _custom_problem(bad)
    m = _StubModel(cfg, torch.sqrt, lambda y: y * y)
    r0, r1 = shard_bounds(x.shape[0], world, rank)
    parts, grads, nbad = custom_codec.energy_and_grads(m, torch.as_tensor(x[r0:r1]), params, 0.7,
                                                       shard=ShardReducer())
    if rank == world - 1:
        q.put({"parts": {k: v.numpy() for k, v in parts.items()},
               "grads": {k: v.numpy() for k, v in grads.items()}, "nbad": nbad.numpy()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad", [0, 3, 37])
def test_custom_callables_over_two_row_shards_equal_one_process(bad):
    """sqrt / square callables: data terms summed over the shards in one packed buffer, prior
    evaluated on every rank; with a rate-0 stored cell the rule's minimum and its gradient come
    from whichever shard holds them (row 3: shard 0, row 37: shard 1; 0: no such cell)."""
    from spmf_amd import custom_codec
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_custom_worker, args=(r, 2, port, q, bad)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, x, params = _custom_problem(bad)
    m = _StubModel(cfg, torch.sqrt, lambda y: y * y)
    parts, grads, nbad = custom_codec.energy_and_grads(m, torch.as_tensor(x), params, 0.7)
    assert res["nbad"].tolist() == nbad.tolist() == ([1.0, 0.0] if bad else [0.0, 0.0])
    for k, v in parts.items():
        np.testing.assert_allclose(res["parts"][k], v.numpy(), rtol=2e-7, err_msg=k)
    for k, v in grads.items():
        a, b = res["grads"][k], v.numpy()
        assert np.isfinite(a).all(), k
        assert np.abs(a - b).max() <= 2e-6 * max(np.abs(b).max(), 1e-30), (k, np.abs(a - b).max(), np.abs(b).max())
