"""Qualitative acceptance run (notebooks/factorizing_random_noise.ipynb:51-62 shape):
pure Poisson(1) noise, N=50000, D=30, P=4, batch 1000 -> the encoder should stay ~0."""
import math, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from spmf_amd import PoissonMatrixFactorization, SparseCounts

rng = np.random.default_rng(1)
N, D, P = 50000, 30, 4
X = rng.poisson(1.0, size=(N, D)).astype(np.float64)
sc = SparseCounts.from_any(X, "cuda", 1000)
batches = [{"counts": sc, "panels": (p, p + 1)} for p in range(sc.n_panels)]
factor = PoissonMatrixFactorization(batches, latent_dim=P, u_tau_scale=1.0 / math.sqrt(D * N))
torch.manual_seed(0)
t0 = time.time()
losses = factor.calibrate_advi(num_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 60,
                               learning_rate=0.05, rel_tol=1e-9, check_every=20)
print("epochs", len(losses), "time", round(time.time() - t0, 1), "loss", losses[0], "->", losses[-1])
A = factor.encoding_matrix().cpu().numpy()
print("max |A|", np.abs(A).max(), "intercept phi/eta mean",
      float((factor.intercept_matrix() / factor._eta_device()).mean()))
