"""CLI surface (bin/factorize_csv.py:20-57 of the reference): flags, defaults
and nargs='?' behaviour must be kept byte-for-byte."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli():
    spec = importlib.util.spec_from_file_location(
        "factorize_csv", os.path.join(ROOT, "bin", "factorize_csv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flags_and_defaults_match_reference():
    p = _cli().build_parser()
    a = p.parse_args([])
    assert (a.csv_file, a.epoch, a.dimension, a.batch_size) == (None, 300, 2, 5000)
    assert (a.learning_rate, a.clip_value, a.log_transform, a.row_normalize) == \
        (0.01, 3.0, False, False)
    a = p.parse_args(["-f", "x.csv", "-e", "7", "-d", "3", "-b", "10", "-lr", "0.5",
                      "-c", "2", "-lt", "-rn"])
    assert (a.csv_file, a.epoch, a.dimension, a.batch_size, a.learning_rate, a.clip_value,
            a.log_transform, a.row_normalize) == ("x.csv", 7, 3, 10, 0.5, 2.0, True, True)
    a = p.parse_args(["--csv-file", "y.csv", "--epoch", "--dimension", "4"])   # nargs='?'
    assert a.epoch is None and a.dimension == 4


def test_package_exports_both_class_names():
    import mederrata_spmf
    assert hasattr(mederrata_spmf, "PoissonFactorization")
    assert hasattr(mederrata_spmf, "PoissonMatrixFactorization")
    assert issubclass(mederrata_spmf.PoissonMatrixFactorization,
                      mederrata_spmf.PoissonFactorization)


def test_scrnaseq_cli_defaults_match_reference_script():
    """bin/factorize_scrnaseq_counts.py hard-codes P=3 (:40), BATCH_SIZE=256 (:46) and
    calibrate_advi(num_steps=500, learning_rate=0.01, abs_tol=1e-3, rel_tol=1e-3,
    clip_value=10) (:101-105): the flags default to those."""
    spec = importlib.util.spec_from_file_location(
        "factorize_scrnaseq_counts", os.path.join(ROOT, "bin", "factorize_scrnaseq_counts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.build_parser().parse_args(["--counts", "x_counts.npy"])
    assert (a.dimension, a.batch_size, a.epoch) == (3, 256, 500)
    assert (a.learning_rate, a.abs_tol, a.rel_tol, a.clip_value) == (0.01, 1e-3, 1e-3, 10.0)


def test_encoding_figure_is_written_without_arviz(tmp_path):
    """bin/factorize_csv.py:141-185 of the reference saves a PDF of the encodings and the
    background-rate intervals; here it needs matplotlib only (the forest plot is the 65 % /
    95 % quantile intervals of the surrogate draws)."""
    import numpy as np
    import pytest
    pytest.importorskip("matplotlib")
    rng = np.random.default_rng(0)
    enc = np.abs(rng.normal(size=(12, 3)))
    rate = np.abs(rng.normal(1.0, 0.1, size=(250, 12)))
    out = tmp_path / "toy_3D_encoding_lt_False_rn_False.pdf"
    _cli().save_encoding_figure(enc, rate, str(out))
    data = out.read_bytes()
    assert data[:5] == b"%PDF-" and len(data) > 2000
