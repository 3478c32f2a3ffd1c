"""Config 4 through the scRNA script's call sequence (bin/factorize_scrnaseq_counts.py:93-105 of the reference:
log_transform=True, column_norms = plain gene means, fit with lr 0.01 / clip 10) on the full 500k x 30k
scRNA-shaped matrix, K = 64, full batch: ms per epoch, the loss, and the decoder's saturation events per epoch
(exp evaluated at min(y, 70): they must decay to 0 as the exponents come down).
usage: fit_c4.py [epochs]   -> one JSON line"""
import contextlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi
from spmf_amd.sparse import balanced_panel_rows

rows, D, K = 500_000, 30_000, 64
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
sc = synth.scrna_like(rows, D, dev, 20241218 + 4, panel_rows=balanced_panel_rows(rows, K), chunk_rows=25_000,
                      target_density=0.03)
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                             log_transform=True)
colsum = torch.zeros(D, dtype=torch.float64, device=dev)
colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
sc.compute_stats(m._handle(), colsum, colnnz)
m.eta_i = (colsum / rows).clamp_min(1e-3).reshape(1, D)        # :93-99: column_norms = gene means (floored)
m.xi_u_global = float((colsum / rows).sum())
batch = {"counts": sc}
opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 0.01)
opt.init_state(10.0)
run = vi.StepRunner(m, opt, rows, 1, seed=7)
per_epoch = []
run.step(batch)                                                  # warm-up (workspace, first sight)
torch.cuda.synchronize()
t0 = time.perf_counter()
for ep in range(epochs):
    opt.reset_epoch_counters()
    run.step(batch)
    st = opt.read_state()
    per_epoch.append({"loss": st[10] / max(st[11], 1.0), "applied": int(st[11]), "skipped": int(st[12]),
                      "saturation_events": st[14]})
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": "C4 full batch (500k x 30k, K=64, log_transform), device-gated VI steps",
                  "epochs_run": epochs, "wall_s": dt, "ms_per_epoch": 1e3 * dt / epochs,
                  "loss_first": per_epoch[0]["loss"], "loss_last": per_epoch[-1]["loss"],
                  "skipped_total": sum(e["skipped"] for e in per_epoch),
                  "saturation_events_per_epoch": [e["saturation_events"] for e in per_epoch]}))
