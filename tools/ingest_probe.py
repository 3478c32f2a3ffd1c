"""Cost of building the device layout (SparseCounts: panel-CSC lists, work items, packed
streams) from CSR arrays already on the device, against the energy + gradient step it feeds.
usage: ingest_probe.py [rows ...]      (C3's generator: D = 20 000, 0.5 % stored)"""
import contextlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth  # noqa: E402
from spmf_amd.sparse import SparseCounts, balanced_panel_rows  # noqa: E402

D, K, dens = 20_000, 32, 0.005
dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [22_784, 122_880, 1_000_000]


def csr(rows):
    cnts, cols, vals, done, cid = [], [], [], 0, 0
    while done < rows:
        n = min(synth.CHUNK_ROWS, rows - done)
        cnt, c, x = synth.linear_structure_chunk(cid, n, D, dens, dev)
        cnts.append(cnt); cols.append(c); vals.append(x)
        done += n; cid += 1
    rp = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.cat(cnts), 0)
    return rp, torch.cat(cols), torch.cat(vals)


for rows in sizes:
    rp, col, val = csr(rows)
    P = balanced_panel_rows(rows, K)
    build = {}
    for mode in ("0", "1"):            # 0: torch operators, 1: spmf_layout_build
        os.environ["SPMF_NATIVE_LAYOUT"] = mode
        ts = []
        torch.cuda.reset_peak_memory_stats()
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc = SparseCounts(rp, col, val, rows, D, P)
            cs = sc.batch_struct()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            if it < 3 or mode == "0":
                del sc, cs
        build["native" if mode == "1" else "torch"] = {
            "ms": [round(1e3 * t, 3) for t in ts],
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.compute_stats(m._handle(), colsum, colnnz)
    torch.cuda.synchronize(); t_stats = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc.compute_stats(m._handle())
    torch.cuda.synchronize(); t_rows = time.perf_counter() - t0
    cm = colsum / colnnz
    m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    m.xi_u_global = float(torch.nansum(cm))
    torch.manual_seed(3); params = m.surrogate_distribution.sample(1)
    for _ in range(3):
        m.energy_and_grads({"counts": sc}, params)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m.energy_and_grads({"counts": sc}, params)
    torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / 10
    print(json.dumps({"rows": rows, "nnz": int(val.numel()), "panel_rows": P,
                      "layout_build": build, "native": bool(getattr(sc, "native_layout", False)),
                      "stats_ms": round(1e3 * t_stats, 3), "stats_rows_only_ms": round(1e3 * t_rows, 3),
                      "step_ms": round(1e3 * t_step, 3)}), flush=True)
    del sc, m, rp, col, val
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
