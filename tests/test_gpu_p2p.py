"""GPU: the step's collective as the library's own kernel over peer pointers (csrc/p2p.hip,
spmf_p2p_*; SURVEY 5 last row / 8e "direct reduce-scatter + all-gather ... or a 2-shot P2P kernel"),
run for real at world 2 and 4: the ranks are separate processes on the ONE visible card, their
regions reach each other through hipIpcGetMemHandle / hipIpcOpenMemHandle (RCCL refuses two ranks on
one device; IPC mappings do not).  On a node the same kernel pushes over xGMI.

Checked bit for bit:
  * the all-reduced buffer == the sum of the ranks' inputs in rank order, on every rank (each slice is
    reduced once, by its owner, in that order) -- lengths that are no multiple of 4 * world, shorter than
    the world, and the C3 accumulator length; many calls back to back (the inbox parities alternate);
  * a hipGraph that holds [refill, all-reduce] replays correctly (the call counter lives on the device);
  * communicators created one after another on the same model (the region is allocated once per context);
  * the row-sharded energy + gradient step and the device-resident VI step (captured WITH the collective
    and replayed) over this transport == the same over gloo's host-staged sum (deterministic mode, so a
    rank's accumulators are the same bits run to run; world 2: a two-term sum has one association).
gloo only carries the 64-byte handles and the test's own comparisons.
"""
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

LENGTHS = [1, 3, 7, 64, 1001, 4099, 262_147, 1_340_140]     # the last: C3's packed accumulators (D 20000, KP 32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(n, world, call):
    """Every rank's input of one call, reproducible on every rank: values of mixed sign and magnitude,
    so that the order of the adds shows in the last bits."""
    out = []
    for r in range(world):
        g = torch.Generator().manual_seed(1000 * call + 17 * r + n % 977)
        x = torch.randn(n, generator=g) * torch.exp(3 * torch.randn(n, generator=g))
        out.append(x.float())
    return out


def _ordered_sum(xs):
    s = xs[0].clone()
    for x in xs[1:]:
        s += x
    return s


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)


def _collective_worker(rank, world, port, q):
    _init(rank, world, port)
    from spmf_amd import PoissonFactorization
    from spmf_amd.dist import PeerComm
    m = PoissonFactorization(latent_dim=3, feature_dim=40, device="cuda", panel_rows=64)
    comm = PeerComm(m, n_max=max(LENGTHS))
    bad = []
    call = 0
    for n in LENGTHS:
        for rep in range(3):
            call += 1
            xs = _inputs(n, world, call)
            buf = xs[rank].cuda()
            comm.all_reduce_(buf)
            want = _ordered_sum(xs)
            if not torch.equal(buf.cpu(), want):
                bad.append((n, rep, float((buf.cpu() - want).abs().max())))
    done, gave_up = comm.status()
    # a graph that refills the buffer and reduces it
    n = 70_001
    xs = _inputs(n, world, 999)
    src = xs[rank].cuda()
    buf = torch.empty_like(src)
    buf.copy_(src)
    comm.all_reduce_(buf)              # eager once with this length
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.copy_(src)
        comm.all_reduce_(buf)
    want = _ordered_sum(xs)
    graph_ok = True
    for _ in range(4):                 # (the capture itself executes nothing)
        g.replay()
        torch.cuda.synchronize()
        graph_ok = graph_ok and torch.equal(buf.cpu(), want)
    done2, gave_up2 = comm.status()
    dist.barrier()
    comm.close()
    # a second and third communicator on the same model, other workgroup counts: the context's region is kept
    # and re-used (spmf_p2p_init), the peers map it again -- the sequence that stalled at world 4 while every
    # communicator freed and re-allocated its region
    again = []
    for nchunk in (16, 64):
        comm = PeerComm(m, n_max=max(LENGTHS), nchunk=nchunk)
        for rep in range(3):
            call += 1
            xs = _inputs(4099, world, call)
            buf = xs[rank].cuda()
            comm.all_reduce_(buf)
            again.append(torch.equal(buf.cpu(), _ordered_sum(xs)))
        again.append(comm.status() == (3, 0))
        dist.barrier()
        comm.close()
    q.put({"rank": rank, "bad": bad, "calls": done, "gave_up": gave_up, "graph_ok": graph_ok,
           "calls2": done2, "gave_up2": gave_up2, "again": again})
    dist.barrier()
    dist.destroy_process_group()


def _run(world, target, timeout=500):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(res, key=lambda d: d["rank"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_peer_allreduce_is_the_rank_ordered_sum_on_every_rank(world):
    res = _run(world, _collective_worker)
    for d in res:
        assert d["gave_up"] == 0 and d["gave_up2"] == 0, d
        assert d["bad"] == [], d["bad"][:3]
        assert d["calls"] == 3 * len(LENGTHS) and d["calls2"] == d["calls"] + 5    # 1 eager + 4 replays
        assert d["graph_ok"]
        assert all(d["again"]) and len(d["again"]) == 8, d["again"]


def _data():
    rng = np.random.default_rng(31)
    N, D = 768, 40
    X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    X[:, ::4] = rng.poisson(np.abs(rng.normal(0, 1, size=(N, 2))) @ np.abs(rng.normal(1.5, .5, size=(2, D // 4))))
    return X


def _step_worker(rank, world, port, q):
    _init(rank, world, port)
    from spmf_amd import PoissonFactorization, SparseCounts
    from spmf_amd import vi
    from spmf_amd.dist import PeerComm, ShardReducer, make_reducer, shard_bounds, sync_seed
    X = _data()
    N, D = X.shape
    r0, r1 = shard_bounds(N, world, rank, granule=64)
    out = {"rank": rank}
    finals = {}
    for name in ("p2p", "gloo"):
        torch.manual_seed(5)
        m = PoissonFactorization(latent_dim=3, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda",
                                 panel_rows=64, deterministic=True)
        sc = SparseCounts.from_any(X[r0:r1], "cuda", 64)
        # (make_reducer: what a caller of fit() uses -- the peer kernel first, RCCL / torch.distributed behind it)
        red = make_reducer(m) if name == "p2p" else ShardReducer()
        comm = red.comm
        assert (name == "p2p") == isinstance(comm, PeerComm)
        m.compute_scales(lambda: [{"counts": sc}], all_reduce=red)
        rows_g = red.dataset_rows
        sync_seed(77)
        params = m.surrogate_distribution.sample(2) if False else m.surrogate_distribution.sample(1)
        parts, grads, _ = m.energy_and_grads({"counts": sc}, params, all_reduce=red)
        torch.cuda.synchronize()
        out[name + "_parts"] = {k: v.cpu().numpy() for k, v in parts.items()}
        out[name + "_grads"] = {k: v.cpu().numpy() for k, v in grads.items()}
        # the device-resident VI step; with the peer transport the collective is INSIDE the captured graph
        opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.02)
        opt.init_state(3.0)
        run = vi.StepRunner(m, opt, rows_g, 1, use_graph=True, all_reduce=red, seed=4242)
        for _ in range(6):
            run.step({"counts": sc})
        torch.cuda.synchronize()
        out[name + "_replays"] = run.replays
        finals[name] = [t.detach().cpu().numpy().copy() for t in m.surrogate_distribution.trainable_variables]
        out[name + "_state"] = opt.read_state()
        if comm is not None:
            out["p2p_status"] = comm.status()
            dist.barrier()
            comm.close()
    # the column-split step (version-5 calls: the all-reduce of the lower column half is issued while the column
    # pass still produces the upper one) over the same kernel: two collectives per step on slices of the buffer
    torch.manual_seed(5)
    m2 = PoissonFactorization(latent_dim=3, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D), device="cuda", panel_rows=64)
    Dh = m2.enable_column_split(32)
    sc2 = SparseCounts.from_any(X[r0:r1], "cuda", 64, col_split=Dh)
    red2 = make_reducer(m2)
    m2.compute_scales(lambda: [{"counts": sc2}], all_reduce=red2)
    sync_seed(77)
    params2 = m2.surrogate_distribution.sample(1)
    parts2, grads2, _ = m2.energy_and_grads({"counts": sc2}, params2, all_reduce=red2)
    torch.cuda.synchronize()
    out["split_parts"] = {k: v.cpu().numpy() for k, v in parts2.items()}
    out["split_grads"] = {k: v.cpu().numpy() for k, v in grads2.items()}
    out["split_status"] = red2.comm.status()
    dist.barrier()
    red2.comm.close()
    out["final_equal"] = all(np.array_equal(a, b) for a, b in zip(finals["p2p"], finals["gloo"]))
    out["final_maxdiff"] = max(float(np.abs(a - b).max()) for a, b in zip(finals["p2p"], finals["gloo"]))
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sharded_step_and_captured_vi_step_over_the_peer_kernel_equal_gloo():
    res = _run(2, _step_worker)
    for d in res:
        assert d["p2p_status"][1] == 0, d["p2p_status"]
        for k, v in d["gloo_parts"].items():
            assert np.array_equal(d["p2p_parts"][k], v), k
        for k, v in d["gloo_grads"].items():
            assert np.array_equal(d["p2p_grads"][k], v), k
        assert d["p2p_replays"] >= 4 and d["gloo_replays"] == 0       # captured with the collective / eager
        assert d["final_equal"], d["final_maxdiff"]
        assert d["p2p_state"][11] == 6 and d["gloo_state"][11] == 6   # six applied steps each
        # the column-split step over the peer kernel == the one-piece step (float atomics: to rounding)
        assert d["split_status"][1] == 0 and d["split_status"][0] >= 2
        for k, v in d["p2p_parts"].items():
            np.testing.assert_allclose(d["split_parts"][k], v, rtol=1e-6, err_msg=k)
        for k, v in d["p2p_grads"].items():
            assert np.abs(d["split_grads"][k] - v).max() <= 1e-5 * np.abs(v).max(), k
    # and the replicas agree across the ranks
    a, b = res
    for k in a["p2p_grads"]:
        assert np.array_equal(a["p2p_grads"][k], b["p2p_grads"][k]), k
