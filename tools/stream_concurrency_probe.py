"""Probe: do the small per-draw kernel chains of two draws overlap when issued on two streams?
Two model instances (own context + workspace) on a 24576-row C2-shaped batch."""
import sys, time, contextlib, io
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth
rows, D, K = 24576, 5000, 16
dev = torch.device("cuda", 0)
sc = synth.bernoulli_poisson(rows, D, 0.01, dev, 7)
ms = []
for i in range(4):
    with contextlib.redirect_stdout(io.StringIO()):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-4, device=dev)
    m.eta_i = torch.ones(1, D, device=dev); m.xi_u_global = 10.0
    ms.append(m)
torch.manual_seed(0)
ps = [m.surrogate_distribution.sample(1) for m in ms]
batch = {"counts": sc}
streams = [torch.cuda.Stream() for _ in ms]
for m, p in zip(ms, ps):
    m.energy_and_grads(batch, p)
torch.cuda.synchronize()
def seq(n):
    for _ in range(n):
        for m, p in zip(ms, ps):
            m.energy_and_grads(batch, p)
def par(n):
    for _ in range(n):
        for m, p, st in zip(ms, ps, streams):
            with torch.cuda.stream(st):
                m.energy_and_grads(batch, p)
for name, fn in (("sequential", seq), ("4 streams", par)):
    fn(5); torch.cuda.synchronize(); t = time.perf_counter(); fn(100); torch.cuda.synchronize()
    print(f"{name}: {1e6*(time.perf_counter()-t)/400:.1f} us per draw")

# the same under graph replay (no host time): one stream vs forked streams inside the capture
def capture(parallel):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if parallel:
            ev = torch.cuda.Event(); ev.record(cur)
            for m, p, st in zip(ms, ps, streams):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    m.energy_and_grads(batch, p)
                e2 = torch.cuda.Event(); e2.record(st); cur.wait_event(e2)
        else:
            for m, p in zip(ms, ps):
                m.energy_and_grads(batch, p)
    return g
for name, par_ in (("graph, one stream", False), ("graph, 4 forked streams", True)):
    g = capture(par_)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print(f"{name}: {1e6*(time.perf_counter()-t)/800:.1f} us per draw")
