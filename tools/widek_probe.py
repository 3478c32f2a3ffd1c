"""Energy + gradient step of the C2 matrix (100k x 5k linear-structure counts at 1 %) at latent dimensions
16 (BASELINE's), 64, 128 and 256: what the whole-wave sparse passes of csrc/widek.hip (K > 64) cost next to the
lane-group kernels.  One JSON line per K; run on the GPU box: python tools/widek_probe.py"""
import contextlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from spmf_amd import PoissonFactorization, synth
from spmf_amd.sparse import balanced_panel_rows


def main():
    dev = torch.device("cuda:0")
    rows, D, density, _K, _desc = bench.WORKLOADS["c2"]
    sweep = [(16, None), (64, None), (128, None), (256, None)]
    if len(sys.argv) > 1 and sys.argv[1] == "only128":
        sweep = [(128, None)]
    elif len(sys.argv) > 1:      # panel-rows sweep of the wide kernels: python tools/widek_probe.py 4096 8192 ...
        sweep = [(K, int(a)) for K in (128, 256) for a in sys.argv[1:]]
    for K, pr_arg in sweep:
        pr = pr_arg or balanced_panel_rows(rows, K)
        sc = synth.linear_structure(rows, D, density, dev, first_chunk=0, panel_rows=pr)
        with contextlib.redirect_stdout(sys.stderr):
            model = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                         device=dev, panel_rows=pr)
        colsum = torch.zeros(D, dtype=torch.float64, device=dev)
        colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
        sc.compute_stats(model._handle(), colsum, colnnz)
        cm = colsum / colnnz
        model.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
        model.xi_u_global = float(torch.nansum(cm))
        torch.manual_seed(20241218)
        params = model.surrogate_distribution.sample(1)
        ms, t6, parts, nnf = bench._timed_steps(model, {"counts": sc}, params, 50, 10)
        nnz = int(sc.nnz)
        gathered = 4.0 * nnz * model._kp() * 4 if hasattr(model, "_kp") else None
        print(json.dumps({"K": K, "rows": rows, "cols": D, "nnz": nnz, "panel_rows": pr, "ms_per_step": round(ms, 4),
                          "kernel_ms": {"begin": round(t6[0], 4), "row_pass": round(t6[1], 4),
                                        "col_pass": round(t6[2], 4), "end": round(t6[3], 4)},
                          "gather_TBps": round(4.0 * nnz * max(4, 1 << (K - 1).bit_length()) * 4 / (t6[1] + t6[2]) / 1e9, 2),
                          "n_nonfinite": float(nnf.sum()), "elbo_x": float(parts["x"][0])}), flush=True)
        del model, sc, params
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
