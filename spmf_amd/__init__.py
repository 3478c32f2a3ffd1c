"""spmf_amd -- MI355X-native hot path of mederrata/spmf.

Exports the reference's class surface (mederrata_spmf/__init__.py:1-3 plus the
legacy name the CLI and notebooks import, bin/factorize_csv.py:14).
"""
from .bernoulli import BernoulliFactorization
from .mixed import MixedFactorization
from .poisson import PoissonFactorization, PoissonMatrixFactorization
from .sparse import SparseCounts

__all__ = ["PoissonFactorization", "PoissonMatrixFactorization", "BernoulliFactorization",
           "MixedFactorization", "SparseCounts"]
