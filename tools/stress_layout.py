"""Randomised comparison of the library's layout builder (spmf_layout_build, the dense compaction,
the list-form statistics) with the torch construction of the same arrays: shapes, densities, panel
sizes, column splits, value kinds and row orders drawn at random; every array must agree bit for bit.
usage: stress_layout.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from spmf_amd.sparse import SparseCounts  # noqa: E402

ARRAYS = ("row_ptr", "col_idx", "val", "pc_ptr", "pc_row", "pc_val", "pc_ent", "ent", "items", "item_ptr",
          "item_mid", "items_per_panel", "items_per_half", "list_first", "item_pos")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
t0 = time.time()
worst = {"nnz": 0, "lists": 0, "items": 0}
fails = 0
for case in range(n_cases):
    rows = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1000, 4097, 20000, 70001]))
    D = int(rng.choice([1, 2, 3, 31, 64, 257, 1000, 5000, 66000]))
    if rows * D > 3e8:
        D = max(1, int(3e8 // rows))
    dens = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3, 0.9]))
    P = int(rng.choice([1, 3, 16, 100, 256, 1024, 8192, 70000, 100000]))
    if rows * D // max(P, 1) > 4e7:           # keep n_panels * D (pc_ptr) small enough
        P = max(P, rows * D // int(4e7) + 1)
    split = int(rng.integers(0, D + 1)) if rng.random() < 0.4 else 0
    kind = rng.choice(["counts", "counts", "counts", "real", "big", "zeros"])
    dense_in = rng.random() < 0.3 and rows * D <= 2e7
    mask = rng.random((rows, D)) < dens if rows * D <= 2e7 else None
    if mask is None:                            # large: draw the stored cells directly
        nnz = int(rows * D * min(dens, 0.01))
        flat = np.unique(rng.integers(0, rows * D, nnz))
        r, c = flat // D, flat % D
    else:
        r, c = np.nonzero(mask)
    nnz = len(r)
    if kind == "counts":
        v = (rng.poisson(1.5, nnz) + 1).astype(np.float32)
    elif kind == "big":
        v = rng.integers(1, 100000, nnz).astype(np.float32)
    elif kind == "zeros":
        v = rng.poisson(0.7, nnz).astype(np.float32)       # stored zeros among them
    else:
        v = (rng.gamma(2.0, 1.0, nnz) + 0.1).astype(np.float32)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))]).astype(np.int64)
    cc, vv = c, v
    if rng.random() < 0.3 and nnz > 1:              # columns in descending order inside the rows
        order = np.lexsort((-c, r))
        cc, vv = c[order], v[order]
    use_dense = dense_in and kind != "zeros"
    built = []
    for mode in ("1", "0"):
        os.environ["SPMF_NATIVE_LAYOUT"] = mode
        if use_dense:
            x = np.zeros((rows, D), np.float32)
            x[r, c] = v
            sc = SparseCounts.from_dense(torch.as_tensor(x).to(dev), dev, P, col_split=split)
        else:
            sc = SparseCounts(torch.as_tensor(ptr).to(dev), torch.as_tensor(cc.astype(np.int64)).to(dev),
                              torch.as_tensor(vv).to(dev), rows, D, P, col_split=split)
        built.append(sc)
    a, b = built
    bad = []
    if not (a.native_layout and not b.native_layout):
        bad.append("mode")
    for k in ARRAYS:
        x, y = getattr(a, k), getattr(b, k)
        if (x is None) != (y is None):
            bad.append(k + ":none")
        elif x is not None and not (x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y)):
            bad.append(k)
    worst["nnz"] = max(worst["nnz"], nnz)
    worst["lists"] = max(worst["lists"], a.n_panels * D)
    worst["items"] = max(worst["items"], int(a.items.shape[0]))
    if bad:
        fails += 1
        print(f"case {case}: rows={rows} D={D} dens={dens} P={P} split={split} kind={kind} dense_in={dense_in} "
              f"nnz={nnz}: MISMATCH in {bad}", flush=True)
print(f"stress_layout: {n_cases} cases (seed {seed}), {fails} mismatching; largest case nnz={worst['nnz']}, "
      f"(panel, column) lists={worst['lists']}, work items={worst['items']}; {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if fails else 0)
