"""Does the ascending order of the column ids inside a row cost the row pass?  The same C3 matrix
with its columns relabelled by a random permutation (rows no longer ascending): step and kernel times
of both.  (tools/gather_rows_probe.hip saw 12 % between i.i.d. and sorted indices on the flat kernel.)"""
import sys, contextlib, ctypes as C
import torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, SparseCounts, synth, _lib

rows, D, K, dens = 1_000_000, 20_000, 32, 0.005
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, dens, dev)
g = torch.Generator(device=dev); g.manual_seed(5)
perm = torch.randperm(D, device=dev, generator=g).to(torch.int32)
sc2 = SparseCounts(sc.row_ptr, perm[sc.col_idx.long()], sc.val, sc.n_rows, D, sc.panel_rows)
# a third: relabelled AND re-sorted inside each row (ascending again, other ids)
key = torch.repeat_interleave(torch.arange(rows, device=dev, dtype=torch.int64),
                              (sc.row_ptr[1:] - sc.row_ptr[:-1]).long()) * D + sc2.col_idx.long()
order = torch.argsort(key)
sc3 = SparseCounts(sc.row_ptr, sc2.col_idx[order], sc.val[order], sc.n_rows, D, sc.panel_rows)
for tag, s in (("ascending ids", sc), ("permuted ids, order kept", sc2), ("permuted ids, re-sorted", sc3)):
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-4, device=dev)
    m.compute_scales(lambda: [{"counts": s}])
    torch.manual_seed(1)
    params = m.surrogate_distribution.sample(1)
    lib, h = _lib.load(), m._handle()
    for _ in range(3):
        m.energy_and_grads({"counts": s}, params)
    torch.cuda.synchronize()
    lib.spmf_ctx_enable_timing(h, 1)
    for _ in range(10):
        m.energy_and_grads({"counts": s}, params)
    torch.cuda.synchronize()
    ms = (C.c_float * 6)()
    lib.spmf_last_timing(h, ms)
    print(f"{tag:28s} row pass {ms[1]:.3f} ms, column pass {ms[2]:.3f} ms")
    del m
