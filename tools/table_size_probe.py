"""The fused row pass against the size of its two gathered tables: C3's rows and stored entries per row (100) at
D = 10 000 ... 40 000 columns, K = 32 -- A' + V' = 2.56 ... 10.2 MB against an XCD's 4 MB L2.  Does the fused pass
(as opposed to the encode-only sweep, profiles/r05_pmc_sparse_passes.txt) get faster when its tables are L2 resident?
One JSON line per D; python tools/table_size_probe.py [rows]"""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    K = 32
    for D in (10_000, 15_000, 20_000, 30_000, 40_000):
        density = 0.005 * 20_000 / D            # ~100 stored entries per row at every D
        pr = balanced_panel_rows(rows, K)
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
        ms, t6, parts, nnf = bench._timed_steps(model, {"counts": sc}, params, 20, 5)
        nnz = int(sc.nnz)
        print(json.dumps({"D": D, "rows": rows, "nnz": nnz, "tables_MB": round(2 * D * K * 4 / 1e6, 2),
                          "ms_per_step": round(ms, 4), "row_pass_ms": round(t6[1], 4), "col_pass_ms": round(t6[2], 4),
                          "row_ps_per_entry": round(1e9 * t6[1] / nnz, 3), "col_ps_per_entry": round(1e9 * t6[2] / nnz, 3),
                          "n_nonfinite": float(nnf.sum())}), flush=True)
        del model, sc, params
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
