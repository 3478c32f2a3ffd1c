"""Does the ORDER of a row's stored entries matter to the row pass?  (the flat gather probe reads
sorted neighbour indices 12 % slower than i.i.d. ones; a CSR row is sorted by column.)

Same matrix, same kernels; the entries of every row in (a) ascending column order (what the generators and
scipy give), (b) a fixed pseudo-random order inside the row, (c) a stride order (entry j of an n-entry row takes
the (j * 37 mod n')-th slot).  The row pass's sums change only in their rounding order; the column pass's lists
are ordered by (panel, column, row) whatever the CSR order is.  Prints ms per step and the kernel taps.
usage: row_order_probe.py [rows]"""
import contextlib
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, _lib, synth  # noqa: E402
from spmf_amd.sparse import SparseCounts, balanced_panel_rows  # noqa: E402

D, K = 20_000, 32
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lib = _lib.load()
pr = balanced_panel_rows(rows, K)
sc0 = synth.linear_structure(rows, D, 0.005, dev, panel_rows=pr)
rp, ci, va = sc0.row_ptr, sc0.col_idx, sc0.val
nnz = int(ci.numel())
row_of = torch.repeat_interleave(torch.arange(rows, device=dev, dtype=torch.int64),
                                 (rp[1:] - rp[:-1]).to(torch.int64))


def reorder(kind):
    if kind == "sorted":
        return ci, va
    pos = torch.arange(nnz, device=dev, dtype=torch.int64)
    if kind == "random":
        h = (pos * 2654435761 + 12345) & 0xFFFFF            # fixed hash of the entry position
    else:                                                   # stride: 37 * (position inside the row) mod 128
        inside = pos - rp[:-1].to(torch.int64)[row_of]
        h = (inside * 37) & 127
    order = torch.argsort(row_of * (1 << 21) + h, stable=True)
    return ci[order].contiguous(), va[order].contiguous()


out = {"rows": rows, "nnz": nnz, "panel_rows": pr}
ref = None
for kind in ("sorted", "random", "stride", "sorted"):
    c2, v2 = reorder(kind)
    sc = SparseCounts(rp, c2, v2, rows, D, pr)
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros_like(colsum)
    sc.compute_stats(m._handle(), colsum, colnnz)
    cm = colsum / colnnz
    m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    m.xi_u_global = float(torch.nansum(cm))
    torch.manual_seed(7)
    params = m.surrogate_distribution.sample(1)
    h = m._handle()
    batch = {"counts": sc}
    for _ in range(3):
        parts, grads, _ = m.energy_and_grads(batch, params)
    torch.cuda.synchronize()
    lib.spmf_ctx_enable_timing(h, 1)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        parts, grads, _ = m.energy_and_grads(batch, params)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    t6 = (C.c_float * 6)()
    lib.spmf_last_timing(h, t6)
    lib.spmf_ctx_enable_timing(h, 0)
    x = float(parts["x"][0])
    if ref is None:
        ref = (x, {k: v.clone() for k, v in grads.items()})
    dmax = max(float((grads[k] - ref[1][k]).abs().max()) / max(float(ref[1][k].abs().max()), 1e-30) for k in grads)
    out.setdefault("runs", []).append({"order": kind, "ms": round(ms, 4), "row": round(t6[1], 4), "col": round(t6[2], 4),
                                       "x_rel_diff": abs(x - ref[0]) / abs(ref[0]), "grad_max_rel_diff": dmax,
                                       "packed": sc.ent is not None})
    del sc, m, c2, v2
    torch.cuda.empty_cache()
print(json.dumps(out))
