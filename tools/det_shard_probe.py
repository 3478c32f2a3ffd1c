"""Default (float atomics) against the deterministic column pass + ordered reduce over shard sizes of C3's
generator: ms per energy + gradient step and the kernel taps, three repeats each, interleaved.
usage: det_shard_probe.py [rows ...]"""
import contextlib
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, _lib, synth  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

D, K = 20_000, 32
dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [61_440, 122_880, 250_000, 500_000, 1_000_000]
lib = _lib.load()
for rows in sizes:
    sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
    ms = {}
    models = {}
    for det in (False, True):
        with contextlib.redirect_stdout(sys.stderr):
            m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                     deterministic=det)
        colsum = torch.zeros(D, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
        sc.compute_stats(m._handle(), colsum, colnnz)
        cm = colsum / colnnz
        m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
        m.xi_u_global = float(torch.nansum(cm))
        models[det] = m
    torch.manual_seed(7)
    params = models[False].surrogate_distribution.sample(1)
    out = {"rows": rows, "n_panels": sc.n_panels, "panel_rows": sc.panel_rows, "nnz": sc.nnz, "n_items": int(sc.items.shape[0])}
    for rep in range(3):
        for det in (False, True):
            m = models[det]
            h = m._handle()
            for _ in range(5):
                m.energy_and_grads({"counts": sc}, params)
            torch.cuda.synchronize()
            lib.spmf_ctx_enable_timing(h, 1)
            n = 50 if rows <= 250_000 else 15
            t0 = time.perf_counter()
            for _ in range(n):
                m.energy_and_grads({"counts": sc}, params)
            torch.cuda.synchronize()
            dt = 1e3 * (time.perf_counter() - t0) / n
            t6 = (C.c_float * 6)()
            lib.spmf_last_timing(h, t6)
            lib.spmf_ctx_enable_timing(h, 0)
            out.setdefault("det" if det else "atomics", []).append(
                {"ms": round(dt, 4), "row": round(t6[1], 4), "col": round(t6[2], 4), "finish": round(t6[3], 4)})
    print(json.dumps(out), flush=True)
    del sc, models
    torch.cuda.empty_cache()
