"""What does not shrink with the batch: the energy + gradient step of small batches.

Three shapes: the 122 880-row shard of C3's 8-GPU run (D = 20 000, K = 32), a 22 784-row minibatch of
it (two 11 392-row panels), and C2 as bench.py's c2_extra runs it (100k x 5k, 1 %, K = 16, one batch).  Per shape:
ms per step without the hipEvent taps, then with them (prep / row / col / finish) and
fixed_us = step - row - col.  SPMF_LEGACY_STEP=1 runs the version-5 call sequence for comparison.
usage: fixed_cost_probe.py [reps]      (under rocprofv3 --kernel-trace --stats for per-kernel times)
"""
import contextlib
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, _lib, synth  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def model_for(sc, rows, D, K):
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros_like(colsum)
    sc.compute_stats(m._handle(), colsum, colnnz)
    cm = colsum / colnnz
    m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    m.xi_u_global = float(torch.nansum(cm))
    return m


def run(name, m, batch, n=100):
    torch.manual_seed(7)
    params = m.surrogate_distribution.sample(1)
    h = m._handle()
    out = {"shape": name, "legacy": os.environ.get("SPMF_LEGACY_STEP", "0") == "1", "runs": []}
    for _ in range(reps):
        for _ in range(10):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        no_taps = 1e3 * (time.perf_counter() - t0) / n
        lib.spmf_ctx_enable_timing(h, 1)
        for _ in range(5):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        taps = 1e3 * (time.perf_counter() - t0) / n
        t6 = (C.c_float * 6)()
        lib.spmf_last_timing(h, t6)
        lib.spmf_ctx_enable_timing(h, 0)
        out["runs"].append({"ms_no_taps": round(no_taps, 4), "ms_taps": round(taps, 4),
                            "prep": round(t6[0], 4), "row": round(t6[1], 4), "col": round(t6[2], 4),
                            "finish": round(t6[3], 4),
                            "fixed_us": round(1e3 * (taps - t6[1] - t6[2]), 1),
                            "fixed_us_no_taps": round(1e3 * (no_taps - t6[1] - t6[2]), 1)})
    print(json.dumps(out), flush=True)


D, K = 20_000, 32
rows = 122_880
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
m = model_for(sc, rows, D, K)
run("c3_shard_122880", m, {"counts": sc})
del sc, m
rows = 11_392 * 8
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=11_392)
m = model_for(sc, rows, D, K)
run("c3_minibatch_22784", m, {"counts": sc, "panels": (0, 2)})
del sc, m
D, K = 5_000, 16
rows = 100_000
pr = balanced_panel_rows(rows, K)
sc = synth.linear_structure(rows, D, 0.01, dev, first_chunk=0, panel_rows=pr)      # bench.py c2_extra
m = model_for(sc, rows, D, K)
run("c2_100k_x_5k", m, {"counts": sc})
