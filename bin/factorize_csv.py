#!/usr/bin/env python3
"""Train PMF on a CSV-formatted count matrix -- same flags, defaults and
output files as the reference CLI (bin/factorize_csv.py:19-204), running on
the MI355X hot path.

Outputs next to the input file (reference :128-200):
  <f>_<K>D_encoding_lt_<b>_rn_<b>.csv        rows of A^T            (:128-134)
  <f>_<K>D_model_lt_<b>_rn_<b>.pkl           factor.save            (:137-139)
  <f>_<K>D_representation_lt_<b>_rn_<b>.csv  index, z (x normalization) (:187-200)
The PDF figure (:143-185) needs matplotlib + arviz and is skipped when they
are not installed.
"""
import argparse
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mederrata_spmf import PoissonMatrixFactorization  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(
        description='Train PMF on CSV-formatted count matrix')
    parser.add_argument(
        '-f', '--csv-file', nargs='?', type=str,
        help="Enter the CSV file")
    parser.add_argument(
        '-e', '--epoch', nargs='?', type=int, default=300,
        help='Enter Epoch value: Default: 300')
    parser.add_argument(
        '-d', '--dimension', nargs='?', type=int, default=2,
        help='Enter embedding dimension. Default: 2')
    parser.add_argument(
        '-b', '--batch-size', nargs='?', type=int, default=5000,
        help='Enter batch size. Default: 5000')
    parser.add_argument(
        '-lr', '--learning-rate', nargs='?', type=float, default=0.01,
        help='Enter float. Default: 0.01')
    parser.add_argument(
        '-c', '--clip-value', nargs='?', type=float, default=3.,
        help='Gradient clip value. Default: 3.0')
    parser.add_argument(
        '-lt', '--log-transform',
        help='Log-transform?', action='store_true')
    parser.add_argument(
        '-rn', '--row-normalize',
        help='Row normalize based on counts?', action='store_true')
    return parser


def save_encoding_figure(encoding, rate_draws, filename):
    """The reference CLI's PDF (bin/factorize_csv.py:141-185): left, the [D, K] encoding matrix
    as a heat map (item 0 at the bottom); right, per item the 95 % (thin) and 65 % (thick)
    intervals and the median of the background rate over the surrogate draws
    (``rate_draws``: [n_draws, D]) -- what arviz.plot_forest shows there."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    encoding = np.asarray(encoding)
    D, K = encoding.shape
    fig, ax = plt.subplots(1, 2, figsize=(14, 8))
    pcm = ax[0].imshow(encoding[::-1, :], vmin=0, cmap="Blues", aspect="auto")
    ax[0].set_yticks(np.arange(D))
    ax[0].set_yticklabels(np.arange(D)[::-1])
    ax[0].set_ylabel("item")
    ax[0].set_xlabel("factor dimension")
    ax[0].set_xticks(np.arange(K))
    ax[0].set_xticklabels(np.arange(K))
    fig.colorbar(pcm, ax=ax[0], orientation="vertical")
    q = np.quantile(np.asarray(rate_draws, dtype=np.float64), [0.025, 0.175, 0.5, 0.825, 0.975], axis=0)
    y = np.arange(D)
    ax[1].hlines(y, q[0], q[4], color="C0", linewidth=1)
    ax[1].hlines(y, q[1], q[3], color="C0", linewidth=3)
    ax[1].plot(q[2], y, "o", color="white", markeredgecolor="C0", markersize=4)
    ax[1].set_yticks(y)
    ax[1].set_xlabel("background rate")
    ax[1].set_title("65% and 95% CI")
    ax[1].axvline(1.0, linestyle="dashed", color="black")
    fig.savefig(filename, bbox_inches="tight")
    plt.close(fig)


def main(argv=None):
    args = build_parser().parse_args(sys.argv[1:] if argv is None else argv)
    if args.csv_file is None:
        sys.exit("You need to specify a csv file")
    elif not os.path.exists(args.csv_file):
        sys.exit("File doesn't exist")
    _FILENAME = args.csv_file
    _BATCH_SIZE = args.batch_size
    _LOG_TRANSFORM = args.log_transform
    _EPOCH_NUMBER = args.epoch
    _DIMENSION = args.dimension
    _LEARNING_RATE = args.learning_rate
    _ROW_NORMALIZE = args.row_normalize
    _CLIP_VALUE = args.clip_value

    X = np.loadtxt(_FILENAME, delimiter=",", dtype=np.float64, ndmin=2)
    N, columns = X.shape
    colmeans = X.sum(0, keepdims=True) / N            # reference :90-98
    rowmean = colmeans.sum()                          # :99

    def batches(drop_remainder):
        out = []
        for lo in range(0, N, _BATCH_SIZE):
            hi = min(N, lo + _BATCH_SIZE)
            if drop_remainder and hi - lo < _BATCH_SIZE:
                break                                  # :110 drop_remainder=True
            b = {'indices': np.arange(lo, hi), 'counts': X[lo:hi]}
            if _ROW_NORMALIZE:                         # :101-108
                b['normalization'] = np.maximum(X[lo:hi].sum(1), 1.) / rowmean
            out.append(b)
        return out

    csv_data_batched = batches(drop_remainder=True)
    if not csv_data_batched:
        sys.exit("Batch size larger than the dataset (drop_remainder=True leaves nothing)")

    factor = PoissonMatrixFactorization(
        csv_data_batched, latent_dim=_DIMENSION, strategy=None,
        scale_columns=True, log_transform=_LOG_TRANSFORM,
        column_norms=colmeans,
        u_tau_scale=1.0 / np.sqrt(columns * N),
        dtype=np.float64)

    factor.calibrate_advi(
        num_steps=_EPOCH_NUMBER,
        rel_tol=1e-4, clip_value=_CLIP_VALUE,
        learning_rate=_LEARNING_RATE)

    print("Saving the encoding matrix")
    filename = f"{_FILENAME}_{_DIMENSION}D_encoding"
    filename += f"_lt_{_LOG_TRANSFORM}_rn_{_ROW_NORMALIZE}.csv"
    with open(filename, "w") as f:
        writer = csv.writer(f)
        encoding = factor.encoding_matrix().cpu().numpy().T
        for row in range(encoding.shape[0]):
            writer.writerow(encoding[row, :])

    print("Saving the trained model object")
    filename = f"{_FILENAME}_{_DIMENSION}D_model"
    filename += f"_lt_{_LOG_TRANSFORM}_rn_{_ROW_NORMALIZE}.pkl"
    factor.save(filename)

    # the reference's figure (bin/factorize_csv.py:141-185): encoding heat map + background-rate
    # intervals.  arviz is not needed: its forest plot is the 65 % and 95 % quantile intervals
    # of the surrogate draws, computed here with numpy.
    filename = f"{_FILENAME}_{_DIMENSION}D_encoding_"
    filename += f"lt_{_LOG_TRANSFORM}_rn_{_ROW_NORMALIZE}.pdf"
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        print("Skipping the figure with the encodings (matplotlib is not installed)")
    else:
        print("Saving figure with the encodings")
        draws = factor.surrogate_distribution.sample(250)
        w = draws['w'].reshape(250, -1).double().cpu().numpy()
        eta = np.asarray(factor.eta_i.cpu() if hasattr(factor.eta_i, "cpu") else factor.eta_i,
                         dtype=np.float64).reshape(1, -1)
        if 's' in draws:
            sd = draws['s'].double().cpu().numpy()
            rate = w * (sd[:, -1, :] / sd.sum(-2)) * eta
        else:
            rate = w * eta
        save_encoding_figure(factor.encoding_matrix().cpu().numpy(), rate, filename)

    print("Generating representations")
    filename = f"{_FILENAME}_{_DIMENSION}D_representation"
    filename += f"_lt_{_LOG_TRANSFORM}_rn_{_ROW_NORMALIZE}.csv"
    with open(filename, 'w') as f:
        writer = csv.writer(f)
        for record in batches(drop_remainder=False):
            # the reference reads record['data'] (:195) although the key is
            # 'counts' (:86): a latent bug; the counts key is used here.
            z = factor.encode(record['counts']).cpu().numpy()
            if _ROW_NORMALIZE:
                z *= record['normalization'][:, np.newaxis]
            ind = record['indices']
            for row in range(z.shape[0]):
                writer.writerow(np.concatenate([[ind[row]], z[row, :]]))


if __name__ == "__main__":
    main()
