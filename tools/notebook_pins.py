"""The only numbers the reference holds for this path: the recorded outputs of its two
notebooks (unseeded data, so a statistical comparison only).

  notebooks/factorizing_random_noise.ipynb:51-62,122,447   N=50000, D=30, P=4, Poisson(1):
      calibrate_advi(num_steps=200, rel_tol=1e-4, learning_rate=.05): loss 44.13 -> ~40.39,
      waic() of one 1000-row batch: lppd -37090.95
  notebooks/factorize_linear_structure.ipynb:53-66,447,468   N=50000, D=30 (10 factor cols), P=3:
      calibrate_advi(num_steps=200, learning_rate=.05): loss 54.5 -> ~46.97, lppd -41236.9

Runs the same calls on the HIP path and prints what it gets (tests/test_gpu_notebook_pins.py
asserts the bands).   python tools/notebook_pins.py [noise|linear] [epochs]"""
import json
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")


def noise_data(seed):
    rng = np.random.default_rng(seed)
    return rng.poisson(1.0, size=(50000, 30)).astype(np.float64), 4


def linear_data(seed):
    rng = np.random.default_rng(seed)
    N, Df, Dn, P = 50000, 10, 20, 3
    V = np.abs(rng.normal(1.5, 0.5, size=(P, Df)))
    Z = np.abs(rng.normal(0, 1, size=(N, P)))
    X = rng.poisson(1.0, size=(N, Df + Dn)).astype(np.float64)
    X[:, ::3] = rng.poisson(Z @ V)
    return X, P


def run(which, epochs=200, seed=0, verbose=False):
    from spmf_amd import PoissonMatrixFactorization, SparseCounts
    X, P = (noise_data if which == "noise" else linear_data)(20241218 + seed)
    N, D = X.shape
    sc = SparseCounts.from_any(X, "cuda", 1000)
    batches = [{"counts": sc, "panels": (p, p + 1)} for p in range(sc.n_panels)]
    factor = PoissonMatrixFactorization(batches, latent_dim=P, u_tau_scale=1.0 / math.sqrt(D * N))
    torch.manual_seed(seed)
    kw = {"rel_tol": 1e-4} if which == "noise" else {}
    t0 = time.time()
    losses = factor.calibrate_advi(num_steps=epochs, learning_rate=0.05, verbose=verbose, **kw)
    w = factor.waic({"counts": X[:1000]})
    return {"which": which, "epochs": len(losses), "first_loss": losses[0], "final_loss": losses[-1],
            "best_loss": min(losses), "lppd": w["lppd"], "waic": w["waic"], "pwaic": w["pwaic"],
            "seconds": round(time.time() - t0, 1)}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "noise"
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    print(json.dumps(run(which, epochs)))
