"""One whole VI step (sampler, surrogate, data pass, finish, gate, chain rule, Adam) on a
122 880-row shard of C3 (the per-GPU shard of the 8-GPU split), hipGraph replay.
usage: vi_shard_step.py [S] [steps]   (run under rocprofv3 --kernel-trace --stats for the split)"""
import sys, time, contextlib
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi

rows, D, K, dens = 1_000_000, 20_000, 32, 0.005
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
n_s = 122_880
sc = synth.linear_structure(n_s, D, dens, dev)
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
m.compute_scales(lambda: [{"counts": sc}])
batch = {"counts": sc}
for use_graph in (False, True):
    torch.manual_seed(0)
    opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-3)
    opt.init_state(3.0)
    run = vi.StepRunner(m, opt, rows, S, use_graph=use_graph)
    for _ in range(10):
        run.step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        run.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = opt.read_state()
    print(f"use_graph={use_graph}: S={S} {1e3*dt:.3f} ms per VI step on a {n_s}-row shard "
          f"(applied {int(st[11])}, skipped {int(st[12])})")
