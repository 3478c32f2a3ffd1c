"""Where does replaying the VI step's hipGraph stop paying?  The same device-gated step
(vi.StepRunner) on minibatches of the resident C3 matrix, eager launches against graph replay, over the
batch size: launch-bound batches win with the graph, GPU-bound ones lose to its per-node cost.
Sets StepRunner.graph_max_nnz.   usage: graph_threshold_sweep.py   -> one line per batch size"""
import contextlib
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi
from spmf_amd.sparse import balanced_panel_rows

rows, D, K, dens = 1_000_000, 20_000, 32, 0.005
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, dens, dev, panel_rows=balanced_panel_rows(rows, K))
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    m.compute_scales(lambda: [{"counts": sc}])
for npan in (1, 2, 4, 7, 11, 16, 22):
    batch = {"counts": sc, "panels": (0, npan)}
    nnz = int(m._batch(batch)[1].nnz)
    out = {}
    for tag, use_graph in (("eager", False), ("graph", True)):
        opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-4)
        opt.init_state(3.0)
        run = vi.StepRunner(m, opt, rows, 1, use_graph=use_graph, seed=11)
        run.graph_max_nnz = 1 << 62
        for _ in range(5):
            run.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            run.step(batch)
        torch.cuda.synchronize()
        out[tag] = 1e3 * (time.perf_counter() - t0) / n
    print(f"panels {npan:2d}  rows {npan * sc.panel_rows:7d}  nnz {nnz:9d}  eager {out['eager']:.4f} ms  "
          f"graph {out['graph']:.4f} ms  graph/eager {out['graph'] / out['eager']:.3f}")
