"""does relabelling the columns by descending stored count (hot genes' 256-B table
rows contiguous) speed up the sparse passes of C4?  Same matrix, three labelings: as generated,
columns sorted by descending count (rows re-sorted ascending under the new ids), and a random
permutation (re-sorted).  A 250k-row half of C4 (the skew is per gene, not per row), K = 64,
log_transform; the row / column launch times come from the library's taps (the two row
launches of the log_transform step: sweep 1, then sweep 2; the dense kernels are not timed here)."""
import sys, contextlib, ctypes as C
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, SparseCounts, synth, _lib

rows, D, K = 250_000, 30_000, 64
dev = torch.device("cuda", 0)
sc = synth.scrna_like(rows, D, dev, 20241218 + 4)
cnt = torch.bincount(sc.col_idx.long(), minlength=D)
top = torch.sort(cnt, descending=True).values
print(f"nnz {sc.nnz}, share of the 640 / 2048 / 4096 most frequent genes: "
      f"{float(top[:640].sum()) / sc.nnz:.3f} / {float(top[:2048].sum()) / sc.nnz:.3f} / {float(top[:4096].sum()) / sc.nnz:.3f}")
lens = (sc.row_ptr[1:] - sc.row_ptr[:-1]).long()
rid = torch.repeat_interleave(torch.arange(rows, device=dev, dtype=torch.int64), lens)


def relabel(perm):
    newc = perm[sc.col_idx.long()].long()
    order = torch.argsort(rid * D + newc)
    return SparseCounts(sc.row_ptr, newc[order].to(torch.int32), sc.val[order], rows, D, sc.panel_rows)


by_count = torch.empty(D, dtype=torch.int64, device=dev)
by_count[torch.argsort(cnt, descending=True)] = torch.arange(D, device=dev)
g = torch.Generator(device=dev); g.manual_seed(5)
cases = (("as generated", sc), ("by descending count", relabel(by_count)),
         ("random permutation", relabel(torch.randperm(D, device=dev, generator=g))))
for tag, s in cases:
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-4, log_transform=True, device=dev)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
    s.compute_stats(m._handle(), colsum, colnnz)
    m.eta_i = (colsum / rows).clamp_min(1e-3).reshape(1, D)
    m.xi_u_global = float((colsum / rows).sum())
    torch.manual_seed(1)
    params = m.surrogate_distribution.sample(1)
    params["v"] = params["v"] * 0.05          # keep the exponents small: only the sparse launches matter here
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
    print(f"{tag:22s} row launches {ms[1]:.3f} ms, column pass {ms[2]:.3f} ms (dense {ms[5]:.3f})", flush=True)
    del m
