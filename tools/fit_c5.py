"""MixedFactorization (config 5: 200k x 10k, even columns Poisson, odd columns Bernoulli, K = 32) through the
reference's call sequence: construct -> compute_scales -> fit, full batch: wall time and loss of N epochs of the
device-side VI loop.   usage: fit_c5.py [epochs]   -> one JSON line"""
import contextlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import MixedFactorization, synth
from spmf_amd.sparse import balanced_panel_rows

rows, D, K = 200_000, 10_000, 32
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
pr = balanced_panel_rows(rows, K)
sc, mask = synth.mixed_c5(rows, D, dev, 20241218 + 5, panel_rows=pr, first_chunk=0, chunk_rows=synth.MIXED_CHUNK_ROWS)
with contextlib.redirect_stdout(sys.stderr):
    m = MixedFactorization(mask, latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                           panel_rows=pr)
    m.compute_scales(lambda: [{"counts": sc}])
    torch.manual_seed(0)
    m.fit(lambda: [{"counts": sc}], dataset_size=rows, sample_size=1, num_epochs=3, learning_rate=0.01,
          rel_tol=0.0, abs_tol=0.0, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = m.fit(lambda: [{"counts": sc}], dataset_size=rows, sample_size=1, num_epochs=epochs,
                   learning_rate=0.01, rel_tol=0.0, abs_tol=0.0, verbose=False)
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "C5 full batch, MixedFactorization.fit()", "epochs_run": len(losses), "wall_s": dt,
                  "ms_per_epoch": 1e3 * dt / max(1, len(losses)), "loss_first": losses[0], "loss_last": losses[-1],
                  "loss_min": min(losses), "all_finite": all(l == l and abs(l) != float("inf") for l in losses)}))
