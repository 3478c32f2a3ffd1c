"""Would restricting each XCD to 1/8 of the columns (tables of ~1 MB instead of 7.7 MB at K = 64)
speed up C4's row sweeps?  Proxy: the same 250k-row half of C4 restricted to its first D/8
columns (rows of ~112 entries, tables L2-resident) against the full matrix: gathered TB/s of the two
row launches and of the column pass."""
import sys, contextlib, ctypes as C
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, SparseCounts, synth, _lib

rows, D, K = 250_000, 30_000, 64
dev = torch.device("cuda", 0)
sc = synth.scrna_like(rows, D, dev, 20241218 + 4)
lens = (sc.row_ptr[1:] - sc.row_ptr[:-1]).long()
rid = torch.repeat_interleave(torch.arange(rows, device=dev, dtype=torch.int64), lens)
# a random eighth of the columns (same skew as the whole), relabelled 0..D/8-1
g = torch.Generator(device=dev); g.manual_seed(7)
perm = torch.randperm(D, device=dev, generator=g)
new_id = torch.full((D,), -1, dtype=torch.int64, device=dev)
Ds = D // 8
new_id[perm[:Ds]] = torch.arange(Ds, device=dev)
keep = new_id[sc.col_idx.long()] >= 0
cnt = torch.bincount(rid[keep], minlength=rows)
rp = torch.zeros(rows + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(cnt, 0)
newc = new_id[sc.col_idx.long()][keep]
order = torch.argsort(rid[keep] * Ds + newc)
sub = SparseCounts(rp, newc[order].to(torch.int32), sc.val[keep][order], rows, Ds, sc.panel_rows)
for tag, s, d in (("full D=30000", sc, D), ("one eighth D=3750", sub, Ds)):
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=d, u_tau_scale=1e-4, log_transform=True, device=dev)
    colsum = torch.zeros(d, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
    s.compute_stats(m._handle(), colsum, colnnz)
    m.eta_i = (colsum / rows).clamp_min(1e-3).reshape(1, d)
    m.xi_u_global = float((colsum / rows).sum())
    torch.manual_seed(1)
    params = m.surrogate_distribution.sample(1)
    params["v"] = params["v"] * 0.05
    lib, h = _lib.load(), m._handle()
    for _ in range(2):
        m.energy_and_grads({"counts": s}, params)
    torch.cuda.synchronize()
    lib.spmf_ctx_enable_timing(h, 1)
    for _ in range(5):
        m.energy_and_grads({"counts": s}, params)
    torch.cuda.synchronize()
    ms = (C.c_float * 6)()
    lib.spmf_last_timing(h, ms)
    tb_row = 2 * s.nnz * 256 / (ms[1] * 1e-3) / 1e12
    tb_col = 2 * s.nnz * 256 / (ms[2] * 1e-3) / 1e12
    print(f"{tag:20s} nnz {s.nnz:>10d} ({s.nnz / rows:.0f}/row): row launches {ms[1]:.3f} ms = {tb_row:.1f} TB/s gathered, "
          f"column pass {ms[2]:.3f} ms = {tb_col:.1f} TB/s", flush=True)
    del m
