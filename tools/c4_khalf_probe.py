"""VERDICT r4 #7, measured before building it: would the encode sweep of C4 (z_k = xi sum_d g(x_bd) A'_dk, separable
in k) run faster on K-HALVED tables -- two sweeps over the same entries, each gathering 128-byte rows out of a
3.84 MB [D, 32] table that fits an XCD's 4 MB L2, instead of one sweep gathering 256-byte rows out of 7.7 MB?

What the halved form would execute is exactly two encode sweeps of a K = 32 model over C4's entries, so that is
what is timed: spmf_encode (prep + sweep 1 + the copy of z) on the SAME counts for K = 64 and K = 32, log_transform
on (sweep 1 reads g(x)), medians of 10.  Verdict line: 2 * t(K=32) against t(K=64).  Also the whole step's taps
at K = 64 for scale.  usage: c4_khalf_probe.py [rows]"""
import contextlib
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, _lib, synth  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
D = 30_000
lib = _lib.load()
pr = balanced_panel_rows(rows, 64)
sc = synth.scrna_like(rows, D, dev, 20241218 + 4, first_chunk=0, panel_rows=pr, chunk_rows=25_000, target_density=0.03)
out = {"rows": rows, "D": D, "nnz": int(sc.nnz)}
for K in (64, 32):
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                 panel_rows=pr, log_transform=True)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros_like(colsum)
    sc.compute_stats(m._handle(), colsum, colnnz)
    cmean = colsum / float(rows)
    m.eta_i = torch.clamp(cmean, min=1e-3).reshape(1, D)
    m.xi_u_global = float(cmean.sum())
    torch.manual_seed(20241218)
    params = m.surrogate_distribution.sample(1)
    batch = {"counts": sc}
    u, s = params["u"][0], params["s"][0]
    for _ in range(3):
        m.encode(batch, u, s)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.encode(batch, u, s)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    out[f"encode_ms_K{K}"] = round(0.5 * (ts[4] + ts[5]), 4)
    out[f"table_MB_K{K}"] = round(D * K * 4 / 1e6, 2)
    if K == 64:
        h = m._handle()
        for _ in range(2):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        lib.spmf_ctx_enable_timing(h, 1)
        for _ in range(3):
            m.energy_and_grads(batch, params)
        torch.cuda.synchronize()
        t6 = (C.c_float * 6)()
        lib.spmf_last_timing(h, t6)
        lib.spmf_ctx_enable_timing(h, 0)
        out["step_taps_K64"] = {"row_launches": round(t6[1], 3), "col": round(t6[2], 3), "dense": round(t6[5], 3)}
    del m, params
    torch.cuda.empty_cache()
out["two_halved_sweeps_ms"] = round(2 * out["encode_ms_K32"], 4)
out["verdict"] = ("halved tables would be %.1f %% %s than the one 256-byte-row sweep" % (
    100 * abs(1 - out["two_halved_sweeps_ms"] / out["encode_ms_K64"]),
    "faster" if out["two_halved_sweeps_ms"] < out["encode_ms_K64"] else "slower"))
print(json.dumps(out))
