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
