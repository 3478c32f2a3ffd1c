#!/usr/bin/env python3
"""Factorize a single-cell RNA-seq count matrix with the log-transform decoder
-- the numeric half of the reference's scRNA script
(bin/factorize_scrnaseq_counts.py:29-140), which is where BASELINE config C4's
model comes from: ``log_transform=True``, ``column_norms`` = plain gene means
(:60,93-99), ``u_tau_scale = 1/sqrt(D*N)``, ``calibrate_advi(num_steps=500,
learning_rate=0.01, abs_tol=1e-3, rel_tol=1e-3, clip_value=10)`` (:101-105).

The reference script hard-codes a dataset directory (:30-35) and draws figures
with matplotlib/scanpy (:142-293); here the paths are flags and the figures are
replaced by a text table of the top genes per factor (what the first figure
shows, :154-159).

Inputs
  --counts   <name>_counts.npy ([cells, genes] dense) or a scipy .npz CSR
  --genes    <name>_genenames.npy (optional)
Outputs next to --counts (reference :124-130), <stem> = the counts file minus
"_counts.npy":
  <stem>_U_<P>.npy  encoding matrix          <stem>_V_<P>.npy  decoding matrix
  <stem>_W_<P>.npy  intercept                <stem>_Z_<P>.npy  encode(X)
  <stem>_cellscore_<P>.npy       Z * row size factors        (:113-114)
  <stem>_genescore_<P>.npy       V * gene means              (:118-119)
  <stem>_interceptscore_<P>.npy  W * gene means              (:108-109)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mederrata_spmf import PoissonMatrixFactorization  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Log-transform PMF of a scRNA-seq count matrix")
    p.add_argument("--counts", required=True, help="<name>_counts.npy or a scipy .npz CSR")
    p.add_argument("--genes", default=None, help="<name>_genenames.npy")
    p.add_argument("-d", "--dimension", type=int, default=3, help="latent factors P (reference: 3)")
    p.add_argument("-b", "--batch-size", type=int, default=256, help="reference: 256")
    p.add_argument("-e", "--epoch", type=int, default=500, help="num_steps (reference: 500)")
    p.add_argument("-lr", "--learning-rate", type=float, default=0.01)
    p.add_argument("-c", "--clip-value", type=float, default=10.0)
    p.add_argument("--abs-tol", type=float, default=1e-3)
    p.add_argument("--rel-tol", type=float, default=1e-3)
    p.add_argument("--top", type=int, default=10, help="genes listed per factor")
    p.add_argument("--seed", type=int, default=None)
    return p


def load_counts(path):
    if path.endswith(".npz"):
        import scipy.sparse as sp
        return sp.load_npz(path).tocsr()
    return np.load(path)


def main(argv=None):
    args = build_parser().parse_args(sys.argv[1:] if argv is None else argv)
    if not os.path.exists(args.counts):
        sys.exit("File doesn't exist")
    import torch
    if args.seed is not None:
        torch.manual_seed(args.seed)
    X = load_counts(args.counts)
    N, D = X.shape
    P, B = args.dimension, args.batch_size
    gene_names = (np.load(args.genes, allow_pickle=True) if args.genes
                  else np.array([f"g{j}" for j in range(D)], dtype=object))

    row_sums = np.asarray(X.sum(1)).reshape(-1).astype(np.float64)
    row_size_factors = row_sums / np.median(row_sums)                     # :48-50
    col_norm = np.asarray(X.mean(0)).reshape(-1).astype(np.float64)       # :56,60
    # (a gene with no counts would divide by zero in g(x) = log(x/eta + 1))
    col_norm = np.maximum(col_norm, 1e-3)

    print((N, D))
    print(f"Total observations={N}, Batch size={B}: dropping {N % B} observations.")   # :76-77
    nb = N // B
    if nb == 0:
        sys.exit("fewer cells than one batch")
    # shuffle once, batch with drop_remainder (:86-87); the batches stay resident on the device
    perm = np.random.default_rng(args.seed).permutation(N)
    batches = []
    for i in range(nb):
        idx = np.sort(perm[i * B:(i + 1) * B])
        xb = X[idx]
        batches.append({"data": xb, "indices": idx, "normalization": row_size_factors[idx]})

    factor = PoissonMatrixFactorization(
        batches, latent_dim=P, strategy=None, scale_rates=True, column_norms=col_norm,
        log_transform=True, u_tau_scale=1.0 / np.sqrt(D * N), count_key="data")       # :91-99
    losses = factor.calibrate_advi(
        num_steps=args.epoch, learning_rate=args.learning_rate,
        abs_tol=args.abs_tol, rel_tol=args.rel_tol, clip_value=args.clip_value)        # :101-105

    U = factor.encoding_matrix().cpu().numpy()                                         # :111
    W = factor.intercept_matrix().cpu().numpy()                                        # :114
    intercept_score = W * col_norm[np.newaxis, :]
    Z = factor.encode(X).cpu().numpy()                                                 # :118
    cell_score = Z * row_size_factors[:, np.newaxis]
    V = factor.decoding_matrix().cpu().numpy()                                         # :122
    gene_score = V * col_norm[np.newaxis, :]

    stem = args.counts
    for suffix in ("_counts.npy", "_counts.npz", ".npy", ".npz"):
        if stem.endswith(suffix):
            stem = stem[:-len(suffix)]
            break
    for name, arr in (("U", U), ("V", V), ("W", W), ("Z", Z), ("cellscore", cell_score),
                      ("genescore", gene_score), ("interceptscore", intercept_score)):
        np.save(f"{stem}_{name}_{P}.npy", arr)                                         # :124-130

    # the first figure's content (:150-159): top genes per factor by gene score
    for p in range(P):
        top = np.argsort(gene_score[p, :])[::-1][:args.top]
        print(f"factor {p}: " + ", ".join(f"{gene_names[j]}({gene_score[p, j]:.3g})" for j in top))
    top = np.argsort(W[0, :])[::-1][:P * args.top]                                     # :174-175
    print("intercept: " + ", ".join(str(gene_names[j]) for j in top))
    return losses


if __name__ == "__main__":
    main()
