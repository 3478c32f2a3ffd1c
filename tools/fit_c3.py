"""The reference's call sequence (tests/spmf_test.py:13-44: construct -> compute_scales -> fit) on the
C3 matrix, full batch: wall time and loss of N epochs of the device-side VI loop (hipGraph replay).
usage: fit_c3.py [epochs] [sample_size] [det]     -> one JSON line
("det": PoissonFactorization(deterministic=True), the whole fit run TWICE from the same seeds: the two loss
lists must be equal, float for float)"""
import contextlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth
from spmf_amd.sparse import balanced_panel_rows

rows, D, K, dens = 1_000_000, 20_000, 32, 0.005
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
det = len(sys.argv) > 3 and sys.argv[3] == "det"
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, dens, dev, panel_rows=balanced_panel_rows(rows, K))


def one_fit(seed):
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(seed)
        mm = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                  deterministic=True)
        mm.compute_scales(lambda: [{"counts": sc}])
        torch.manual_seed(seed + 1)
        return mm.fit(lambda: [{"counts": sc}], dataset_size=rows, sample_size=S, num_epochs=epochs,
                      learning_rate=0.01, rel_tol=0.0, abs_tol=0.0, verbose=False)


repeat_equal = None
if det:
    la, lb = one_fit(100), one_fit(100)
    repeat_equal = la == lb
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                             deterministic=det)
    m.compute_scales(lambda: [{"counts": sc}])
    torch.manual_seed(0)
    m.fit(lambda: [{"counts": sc}], dataset_size=rows, sample_size=S, num_epochs=3, learning_rate=0.01,
          rel_tol=0.0, abs_tol=0.0, verbose=False)                     # warm-up (graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = m.fit(lambda: [{"counts": sc}], dataset_size=rows, sample_size=S, num_epochs=epochs,
                   learning_rate=0.01, rel_tol=0.0, abs_tol=0.0, verbose=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"workload": "C3 full batch, fit()", "deterministic": det,
                  "two_fits_from_the_same_seeds_give_equal_loss_lists": repeat_equal, "epochs_run": len(losses), "sample_size": S,
                  "wall_s": dt, "ms_per_epoch": 1e3 * dt / max(1, len(losses)),
                  "loss_first": losses[0], "loss_last": losses[-1],
                  "loss_min": min(losses), "all_finite": all(l == l and abs(l) < 1e300 for l in losses),
                  "saturated_events": getattr(m, "saturated_events", 0)}))
