"""The peer-pointer all-reduce on ONE GPU (ranks = processes on the card, IPC): us per call of C3's accumulator length over
the number of workgroups, world 2 and 4.  Not an xGMI number -- the protocol's own cost (launch, three flag
hand-offs, 3 passes over the buffer through one GPU's memory).  usage: p2p_sweep.py"""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1_300_076


def worker(rank, world, port, q, chunks):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from spmf_amd import PoissonFactorization
    from spmf_amd.dist import PeerComm
    m = PoissonFactorization(latent_dim=3, feature_dim=40, device="cuda", panel_rows=64)
    out = {}
    x = torch.randn(N, device="cuda")
    for nchunk in chunks:
        comm = PeerComm(m, n_max=N, nchunk=nchunk)
        for _ in range(5):
            comm.all_reduce_(x)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            comm.all_reduce_(x)
        torch.cuda.synchronize()
        out[nchunk] = round(1e6 * (time.perf_counter() - t0) / 50, 2)
        assert comm.status()[1] == 0
        dist.barrier()
        comm.close()
        x.normal_()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def one_set(world, chunks, wait):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q, chunks)) for r in range(world)]
    for p in ps:
        p.start()
    try:
        res = q.get(timeout=wait)
    except Exception:
        res = {c: "no answer in %d s" % wait for c in chunks}
    for p in ps:
        p.join(30)
        if p.is_alive():
            p.kill()
    return res


if __name__ == "__main__":
    if "--same-processes" in sys.argv:
        # every workgroup count inside ONE set of processes: communicators created one after another on the same
        # context, which keeps and re-uses its region (spmf_p2p_init)
        for world in (2, 4):
            print(json.dumps({"world": world, "floats": N, "same_processes": True,
                              "us_per_call_by_workgroups": one_set(world, (8, 16, 32, 64, 128), 300)}), flush=True)
        sys.exit(0)
    # one set of processes per (world, workgroup count): a region is created once per process (re-creating the
    # regions inside one process at world 4 is what stalled the first form of this sweep)
    for world in (2, 4):
        res = {}
        for nchunk in (8, 16, 32, 64, 128):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            ps = [ctx.Process(target=worker, args=(r, world, port, q, (nchunk,))) for r in range(world)]
            for p in ps:
                p.start()
            try:
                res.update(q.get(timeout=120))
            except Exception:
                res[nchunk] = "no answer in 120 s"
            for p in ps:
                p.join(30)
                if p.is_alive():
                    p.kill()
            print(json.dumps({"world": world, "workgroups": nchunk, "us_per_call": res.get(nchunk)}), flush=True)
        print(json.dumps({"world": world, "floats": N, "us_per_call_by_workgroups": res}), flush=True)
