"""Synthetic count matrices of the shapes BASELINE.json names.

Generated on the device with torch (data plumbing, not the hot path) in
fixed 125k-row chunks seeded by chunk id, so the global matrix is the same
however many ranks share it.
"""
from __future__ import annotations

import torch

from .sparse import SparseCounts

CHUNK_ROWS = 125_000


def linear_structure_chunk(chunk_id, rows, D, density, device, seed=20241218 + 3,
                           n_factors=8):
    """CSR pieces of one row chunk of the C3 workload: the generator of
    notebooks/factorize_linear_structure.ipynb:53-66 (Poisson(1) noise, every
    third column Poisson(Z.V) with Z=|N(0,1)|, V=|N(1.5,.5)|) thinned by a
    Bernoulli(p) mask so that the stored density is ~`density`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000 + chunk_id)
    gv = torch.Generator(device=device)
    gv.manual_seed(seed)                       # V is shared by every chunk
    nf = (D + 2) // 3
    V = (1.5 + 0.5 * torch.randn(n_factors, nf, device=device, generator=gv)).abs()
    Z = torch.randn(rows, n_factors, device=device, generator=g).abs()
    # retention of a candidate cell: P(Poisson>0) ~ (2/3)*0.632 + (1/3)*~1
    p_cand = min(1.0, density / 0.755)
    lam = torch.full((rows,), D * p_cand, device=device)
    n_cand = torch.poisson(lam, generator=g).clamp_(max=D).to(torch.int64)
    r = torch.repeat_interleave(torch.arange(rows, device=device), n_cand)
    c = torch.randint(0, D, (int(r.numel()),), device=device, generator=g)
    key = torch.unique(r * D + c)              # sorted, duplicate free
    r, c = key // D, key % D
    is_f = (c % 3) == 0
    rate = torch.ones(key.numel(), device=device)
    fi = is_f.nonzero(as_tuple=True)[0]
    rate[fi] = (Z[r[fi]] * V[:, c[fi] // 3].T).sum(1)
    x = torch.poisson(rate, generator=g)
    keep = x > 0
    r, c, x = r[keep], c[keep], x[keep]
    cnt = torch.bincount(r, minlength=rows)
    return cnt, c.to(torch.int32), x.to(torch.float32)


def linear_structure(rows, D, density, device, first_chunk=0, panel_rows=8192,
                     chunk_rows=CHUNK_ROWS, col_split=0):
    """SparseCounts of `rows` rows starting at global chunk `first_chunk`."""
    cnts, cols, vals = [], [], []
    done, cid = 0, first_chunk
    while done < rows:
        n = min(chunk_rows, rows - done)
        cnt, c, x = linear_structure_chunk(cid, n, D, density, device)
        cnts.append(cnt); cols.append(c); vals.append(x)
        done += n
        cid += 1
    cnt = torch.cat(cnts)
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(cnt, 0)
    return SparseCounts(row_ptr, torch.cat(cols), torch.cat(vals), rows, D, panel_rows,
                        col_split=col_split)


def bernoulli_poisson(rows, D, density, device, seed, mean=2.0, panel_rows=8192):
    """C2-style: Bernoulli(density) mask x (1 + Poisson(mean)) values."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lam = torch.full((rows,), D * density, device=device)
    n = torch.poisson(lam, generator=g).clamp_(max=D).to(torch.int64)
    r = torch.repeat_interleave(torch.arange(rows, device=device), n)
    c = torch.randint(0, D, (int(r.numel()),), device=device, generator=g)
    key = torch.unique(r * D + c)
    r, c = key // D, key % D
    x = 1.0 + torch.poisson(torch.full((key.numel(),), mean, device=device), generator=g)
    cnt = torch.bincount(r, minlength=rows)
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(cnt, 0)
    return SparseCounts(row_ptr, c.to(torch.int32), x.to(torch.float32), rows, D, panel_rows)


def scrna_like(rows, D, device, seed, first_chunk=0, panel_rows=8192, chunk_rows=25_000,
               target_density=0.03, max_gene_mean=None, size_clip=None):
    """C4-style scRNA-seq-shaped counts: per-gene mean ~
    LogNormal(-3.5, 1.5), per-cell size factor ~ LogNormal(0, 0.5),
    X ~ Poisson(size * mean), gene means rescaled so the stored density is
    ~`target_density`.  Generated dense chunk by chunk on the device.
    No clamping by default (round 1 capped the gene means at 4 and clipped the
    size factors at 2 sigma so that exp(<z, eta v>) could not overflow fp32 at
    the surrogate's initial values; the kernels now saturate the exponent
    instead -- csrc/common.h kYSat); ``max_gene_mean`` / ``size_clip`` (in
    sigmas) bring the old generator back."""
    gg = torch.Generator(device=device)
    gg.manual_seed(seed)
    mean = torch.exp(-3.5 + 1.5 * torch.randn(D, device=device, generator=gg))
    # P(x>0) = 1 - exp(-size*mean); calibrate a global scale on a size-factor sample
    sf = torch.exp(0.5 * torch.randn(2048, device=device, generator=gg))
    lo, hi = 1e-3, 1e3
    for _ in range(40):
        mid = (lo * hi) ** 0.5
        m_ = mean[None, :] * mid
        if max_gene_mean is not None:
            m_ = m_.clamp_max(max_gene_mean)
        dens = (1 - torch.exp(-(sf[:, None] * m_))).mean().item()
        if dens < target_density:
            lo = mid
        else:
            hi = mid
    mean = mean * ((lo * hi) ** 0.5)
    if max_gene_mean is not None:
        mean = mean.clamp_max(max_gene_mean)
    cnts, cols, vals = [], [], []
    done, cid = 0, first_chunk
    while done < rows:
        n = min(chunk_rows, rows - done)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000 + 7 + cid)
        lsz = 0.5 * torch.randn(n, device=device, generator=g)
        if size_clip is not None:
            lsz = lsz.clamp_(-0.5 * size_clip, 0.5 * size_clip)
        size = torch.exp(lsz)
        x = torch.poisson(size[:, None] * mean[None, :], generator=g)
        mask = x > 0
        cnts.append(mask.sum(1))
        nz = mask.nonzero(as_tuple=False)
        cols.append(nz[:, 1].to(torch.int32))
        vals.append(x[mask].to(torch.float32))
        del x, mask, nz
        done += n
        cid += 1
    cnt = torch.cat(cnts)
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(cnt, 0)
    return SparseCounts(row_ptr, torch.cat(cols), torch.cat(vals), rows, D, panel_rows)


MIXED_CHUNK_ROWS = 25_000


def mixed_c5(rows, D, device, seed, panel_rows=8192, first_chunk=0, chunk_rows=MIXED_CHUNK_ROWS):
    """C5: even columns Poisson as C2 (Bernoulli(0.01) mask x
    (1 + Poisson(2))), odd columns Bernoulli(0.05) 0/1.  Generated in row chunks
    seeded by chunk id (the global matrix does not depend on how many ranks share
    it); ``first_chunk`` is the global index of this shard's first chunk.  Returns
    (SparseCounts, bernoulli_column_mask)."""
    dens = 0.5 * 0.01 + 0.5 * 0.05
    cnts, cols, vals = [], [], []
    done, cid = 0, first_chunk
    while done < rows:
        n_rows = min(chunk_rows, rows - done)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000 + 11 + cid)
        lam = torch.full((n_rows,), D * dens * 1.05, device=device)
        n = torch.poisson(lam, generator=g).clamp_(max=D).to(torch.int64)
        r = torch.repeat_interleave(torch.arange(n_rows, device=device), n)
        # draw a column: odd (Bernoulli) with prob 5/6, even (Poisson) with prob 1/6
        odd = torch.rand(r.numel(), device=device, generator=g) < (0.05 / 0.06)
        half = torch.randint(0, D // 2, (int(r.numel()),), device=device, generator=g)
        c = 2 * half + odd.to(torch.int64)
        key = torch.unique(r * D + c)
        r, c = key // D, key % D
        x = torch.where(c % 2 == 1, torch.ones(key.numel(), device=device),
                        1.0 + torch.poisson(torch.full((key.numel(),), 2.0, device=device), generator=g))
        cnts.append(torch.bincount(r, minlength=n_rows))
        cols.append(c.to(torch.int32))
        vals.append(x.to(torch.float32))
        done += n_rows
        cid += 1
    cnt = torch.cat(cnts)
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(cnt, 0)
    mask = (torch.arange(D) % 2 == 1).numpy()
    return SparseCounts(row_ptr, torch.cat(cols), torch.cat(vals), rows, D, panel_rows), mask
