"""Worst entry-wise gradient error (in units of the yardstick, tests/_gradcheck.py) of the dense
log_transform path at K = 64 over the size of the largest exponent, for whichever dense kernels
the environment selects (SPMF_DENSE_BF16X3, SPMF_LIB_PATH).  Test infrastructure: uses the oracle."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import spmf_oracle as O
from _gradcheck import worst_entry
from test_gpu_logtransform import problem
from spmf_amd import PoissonFactorization
out = {}
for (B, D, ymax) in [(300, 129, 8.0), (513, 64, 30.0), (90, 1000, 45.0), (90, 1000, 60.0), (700, 333, 60.0)]:
    K = 64
    cfg, x, params = problem(B, D, K, 1, 2900 + B + D, 0.05)
    params["v"] *= ymax / 8.0
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    sc = O.energy_grad_scales(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, log_transform=True,
                             column_norms=cfg.eta_i, initialize_distributions=False, device="cuda", panel_rows=64)
    m.xi_u_global = cfg.xi_u_global
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    w = {k: worst_entry(grads[k].cpu().double().numpy(), gref[k].numpy(), sc[k].numpy())[0] for k in ("u", "v", "w", "s")}
    xr = abs(float(parts["x"].sum()) - float(pref["x"].sum())) / abs(float(pref["x"].sum()))
    print(f"B={B} D={D} ymax={ymax}: x_rel={xr:.2e} worst " + " ".join(f"{k}={v:.2e}" for k, v in w.items()), flush=True)
