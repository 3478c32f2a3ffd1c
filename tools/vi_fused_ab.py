"""A/B of the VI step's first launch: spmf_sample_transform (one launch) against spmf_sample_noise +
spmf_surrogate_fwd + the log-q fold (three), on the 122 880-row shard of C3 (eager), on C2-sized and on the reference
harness' shape (replayed from hipGraphs).  usage: vi_fused_ab.py"""
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi  # noqa: E402
from spmf_amd.sparse import SparseCounts, balanced_panel_rows  # noqa: E402

dev = torch.device("cuda", 0)


def timed(run, batch, n, w=10):
    for _ in range(w):
        run.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run.step(batch)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def case(name, m, batch, rows, S, n):
    out = {"case": name}
    for rep in range(2):
        for fused in ("1", "0"):
            os.environ["SPMF_VI_FUSED_SAMPLER"] = fused
            opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-3)
            opt.init_state(3.0)
            run = vi.StepRunner(m, opt, rows, S, use_graph=True)
            out.setdefault("one_launch_ms" if fused == "1" else "three_launches_ms", []).append(round(timed(run, batch, n), 4))
            del run, opt
    print(json.dumps(out), flush=True)


D, K, rows = 20_000, 32, 122_880
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    m.compute_scales(lambda: [{"counts": sc}])
case("c3_shard_122880_S1", m, {"counts": sc}, rows, 1, 100)
del sc, m
rng = np.random.default_rng(6)
x = rng.poisson(1.0, size=(5000, 350)).astype(np.float64)
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=50, feature_dim=350, u_tau_scale=1.0 / (5000 * 350) ** 0.5, device=dev)
    m.compute_scales(lambda: [{"counts": x}])
for r in (10, 1000):
    case(f"ref_harness_b{r}_S20", m, {"counts": SparseCounts.from_any(x[:r], dev, balanced_panel_rows(r, 50))}, 5000, 20, 200)
x = rng.poisson(1.0, size=(5000, 200)).astype(np.float64)
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=2, feature_dim=200, u_tau_scale=1.0 / (5000 * 200) ** 0.5, device=dev)
    m.compute_scales(lambda: [{"counts": x}])
case("c1_5000x200_K2_S1", m, {"counts": SparseCounts.from_any(x, dev, balanced_panel_rows(5000, 2), latent_dim=2)}, 5000, 1, 300)
