"""Drop-in import shim: ``from mederrata_spmf import PoissonFactorization``
(tests/spmf_test.py:6) and ``PoissonMatrixFactorization``
(bin/factorize_csv.py:14) resolve to the MI355X-native implementation."""
from spmf_amd import (BernoulliFactorization, PoissonFactorization,
                      PoissonMatrixFactorization, SparseCounts)

__all__ = ["PoissonFactorization", "PoissonMatrixFactorization", "BernoulliFactorization",
           "SparseCounts"]
