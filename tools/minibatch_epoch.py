"""Minibatch training epochs through vi.StepRunner (eager vs hipGraph replay): 3-panel (24576-row)
minibatches of the resident matrix.  usage: minibatch_epoch.py [c3|c2] [S]"""
import sys, time, contextlib
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi

cfgs = {"c3": (1_000_000, 20_000, 32, 0.005), "c2": (100_000, 5_000, 16, 0.01)}
rows, D, K, dens = cfgs[sys.argv[1] if len(sys.argv) > 1 else "c3"]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, dens, dev) if D == 20_000 else \
    synth.bernoulli_poisson(rows, D, dens, dev, 20241218 + 2)
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
colsum = torch.zeros(D, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
sc.compute_stats(m._handle(), colsum, colnnz)
cm = colsum / colnnz
m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
m.xi_u_global = float(torch.nansum(cm))
batches = [{"counts": sc, "panels": (p, min(p + 3, sc.n_panels))} for p in range(0, sc.n_panels, 3)]
for use_graph in (False, True):
    torch.manual_seed(0)
    opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-3)
    opt.init_state(3.0)
    run = vi.StepRunner(m, opt, rows, S, use_graph=use_graph)
    for ep in range(5):
        if ep == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.reset_epoch_counters()
        for b in batches:
            run.step(b)
        st = opt.read_state()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"use_graph={use_graph}: {len(batches)} steps/epoch, {1e3*dt:.2f} ms/epoch, {1e3*dt/len(batches):.3f} ms/step, "
          f"epoch loss {st[10]/max(st[11],1):.4f}, skipped {int(st[12])}, graphs {len(run.graphs)}")
