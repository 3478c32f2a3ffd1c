"""Sweep 1 alone (spmf_encode: prep + the encode-only row pass, ONE 2.56 MB table that fits an XCD's L2) on C3, in a
loop for rocprofv3 --pmc: does the row pass's request rate rise when its table is L2 resident?  (round 5,
VERDICT r4 #3: the fused row pass holds ~73 requests per CU in flight at 301 clk of mean latency, the column pass 63 at
205 clk; tools/pmc_r05_summary.py)   usage: encode_only_loop.py [rows] [reps]"""
import contextlib
import sys

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

D, K = 20_000, 32
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
colsum = torch.zeros(D, dtype=torch.float64, device=dev)
colnnz = torch.zeros_like(colsum)
sc.compute_stats(m._handle(), colsum, colnnz)
cm = colsum / colnnz
m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
m.xi_u_global = float(torch.nansum(cm))
torch.manual_seed(7)
p = m.surrogate_distribution.sample(1)
for _ in range(reps):
    z = m.encode({"counts": sc}, p["u"][0], p["s"][0])
torch.cuda.synchronize()
print("z", float(z.sum()))
