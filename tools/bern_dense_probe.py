"""BernoulliFactorization (bernoulli.py:126-216) energy + gradient step on synthetic Bernoulli(0.05) data:
ms per step and the dense sigmoid kernels' share, for K = 32 and 64, on the bf16x3 kernels (default) or with
SPMF_DENSE_BF16X3=0 on the exact-f32 ones.   usage: bern_dense_probe.py [rows] [cols] [logt]   -> one JSON line per K
(third argument "logt": log_transform=True, bernoulli.py:60-61, K = 32 only)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
from spmf_amd import BernoulliFactorization, SparseCounts, _lib

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000
logt = len(sys.argv) > 3 and sys.argv[3] == "logt"
rng = np.random.default_rng(5)
X = sp.random(rows, cols, density=0.05, format="csr", random_state=rng, data_rvs=lambda n: np.ones(n))
dev = torch.device("cuda", 0)
for K in ((32,) if logt else (32, 64)):
    sc = SparseCounts.from_any(X, dev, None, latent_dim=K)
    m = BernoulliFactorization(latent_dim=K, feature_dim=cols, u_tau_scale=1.0 / (rows * cols) ** 0.5, device=dev,
                               log_transform=logt)
    torch.manual_seed(1)
    p = m.surrogate_distribution.sample(1)
    lib, h = _lib.load(), m._handle()
    for _ in range(3):
        m.energy_and_grads({"counts": sc}, p)
    torch.cuda.synchronize()
    lib.spmf_ctx_enable_timing(h, 1)
    t0 = time.perf_counter()
    for _ in range(10):
        parts, _, _ = m.energy_and_grads({"counts": sc}, p)
    torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    t6 = (C.c_float * 6)()
    _lib.check(h, lib.spmf_last_timing(h, t6), "spmf_last_timing")
    lib.spmf_ctx_enable_timing(h, 0)
    tf = 6.0 * rows * cols * max(32, K) / (t6[5] * 1e-3) / 1e12
    print(json.dumps({"K": K, "log_transform": logt, "rows": rows, "cols": cols, "bf16x3": os.environ.get("SPMF_DENSE_BF16X3", "1"),
                      "ms_per_step": round(ms, 4), "dense_ms": round(t6[5], 4), "row_ms": round(t6[1], 4),
                      "col_ms": round(t6[2], 4), "dense_tflops_algorithmic": round(tf, 1),
                      "frac_f32_mfma_peak": round(tf / 157.3, 3), "x": float(parts["x"][0])}))
    del m, sc
