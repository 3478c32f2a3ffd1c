#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the fp64 CPU oracle
(oracle/spmf_oracle.py).  The reference itself cannot run here (TensorFlow /
TFP / bayesianquilts absent -- ModuleNotFoundError at poisson.py:11), so these
are ORACLE outputs, i.e. parity stays "unpinned" in the sense of the task
statement; they freeze the restatement against regressions and give the GPU
tests data files that travel to the GPU box.

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import spmf_oracle as O  # noqa: E402

CASES = {
    # name: (B, D, K, S, density, scale_rows, seed)
    "g1_B24_D17_K3_S2": (24, 17, 3, 2, 0.35, True, 101),
    "g2_B40_D30_K8_S1": (40, 30, 8, 1, 0.15, True, 202),
    "g3_B33_D21_K2_S2_noscale": (33, 21, 2, 2, 0.6, False, 303),
    "g4_B64_D48_K32_S1": (64, 48, 32, 1, 0.1, True, 404),
}


def make(name, B, D, K, S, density, scale_rows, seed):
    rng = np.random.default_rng(seed)
    x = ((rng.random((B, D)) < density) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    x[1, :] = 0.0
    x[:, 2] = 0.0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows,
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(2.0, 6.0))
    params = O.random_params(cfg, S, seed + 1)
    parts, grads, _ = O.energy_and_grads(cfg, x, params)
    out = {"x": x, "eta": cfg.eta_i.numpy(), "xi": np.float64(cfg.xi_u_global),
           "u_tau_scale": np.float64(cfg.u_tau_scale), "scale_rows": np.bool_(scale_rows),
           "K": np.int64(K)}
    for k, v in params.items():
        out["p_" + k] = v
    for k, v in parts.items():
        out["part_" + k] = v.numpy()
    for k, v in grads.items():
        out["grad_" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    for name, args in CASES.items():
        make(name, *args)
        print("wrote", name)
