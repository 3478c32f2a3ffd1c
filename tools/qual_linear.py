"""Qualitative acceptance run (notebooks/factorize_linear_structure.ipynb:53-66 shape):
every third column is driven by 3 latent factors, the rest is Poisson(1) noise."""
import math, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from spmf_amd import PoissonMatrixFactorization, SparseCounts

rng = np.random.default_rng(0)
N, D_factor, D_noise, P = 20000, 10, 20, 3
D = D_factor + D_noise
V = np.abs(rng.normal(1.5, 0.5, size=(P, D_factor)))
Z = np.abs(rng.normal(0, 1, size=(N, P)))
X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
X[:, ::3] = rng.poisson(Z @ V)
sc = SparseCounts.from_any(X, "cuda", 1000)
batches = [{"counts": sc, "panels": (p, p + 1), "indices": np.arange(p * 1000, (p + 1) * 1000)}
           for p in range(sc.n_panels)]
factor = PoissonMatrixFactorization(batches, latent_dim=P, u_tau_scale=1.0 / math.sqrt(D * N))
torch.manual_seed(int(__import__("os").environ.get("SEED", "0")))
t0 = time.time()
losses = factor.calibrate_advi(num_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 100,
                               learning_rate=0.05, rel_tol=1e-9, check_every=int(__import__("os").environ.get("EVERY", "20")),
                               use_graph=(len(sys.argv) < 3 or sys.argv[2] != 'eager'))
print("epochs", len(losses), "time", round(time.time() - t0, 1), "loss", losses[0], "->", losses[-1])
A = factor.encoding_matrix().cpu().numpy()          # [D, P]
load = np.abs(A).sum(1)
print("mean loading factor cols", load[::3].mean(), "noise cols", np.delete(load, np.arange(0, D, 3)).mean())
print(np.round(load, 4))
