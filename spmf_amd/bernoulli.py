"""BernoulliFactorization -- mirror of mederrata_spmf/bernoulli.py:32-649 over
the HIP path.

Differences from PoissonFactorization, as in the reference:
  * likelihood  tfd.Bernoulli(logits=rate), rate = f(z B) + phi        :127-155
  * v, w        Identity bijector, Normal(0, .1) / Normal(0, 1) priors :185-217
  * encode      no row scaling                                        :572-589
The stored-cell term x*logit is linear (sparse row/column passes without any
division or log); the sum over ALL cells of softplus(logit) and its gradients
run on the f32 matrix cores (dense.hip, sigmoid/softplus variant).
Both decoders are built: with log_transform=True the logit is
exp(<z, eta v>) - 1 + phi (:60-61) and g(x) = log(x/eta + 1) (:49-50).  The dense
per-cell outputs of log_likelihood_components (:126-155) come from
spmf_dense_ll (dense_ll.hip).
"""
from __future__ import annotations

import torch

from . import _lib
from .poisson import PoissonFactorization


class BernoulliFactorization(PoissonFactorization):
    """Sparse (horseshoe) Bernoulli matrix factorization (bernoulli.py:32-36).
    Constructor keywords are the reference's (bernoulli.py:64-79)."""

    _likelihood_flag = _lib.FLAG_BERNOULLI
    _identity_vars = ("v", "w")          # bijectors: tfb.Identity (bernoulli.py:187-193)

    def __init__(
            self,
            latent_dim=None, feature_dim=None,
            u_tau_scale=0.01, s_tau_scale=1.0, symmetry_breaking_decay=0.99,
            strategy=None, encoder_function=None, decoder_function=None,
            log_transform=False, horshoe_plus=True, column_norms=None,
            count_key="counts", dtype=torch.float64, device=None,
            panel_rows=None, **kwargs):
        super().__init__(
            latent_dim=latent_dim, feature_dim=feature_dim,
            u_tau_scale=u_tau_scale, s_tau_scale=s_tau_scale,
            symmetry_breaking_decay=symmetry_breaking_decay, strategy=strategy,
            encoder_function=encoder_function, decoder_function=decoder_function,
            scale_columns=True, scale_rows=False, log_transform=log_transform,
            horshoe_plus=horshoe_plus, column_norms=column_norms, count_key=count_key,
            initialize_distributions=True, dtype=dtype, device=device,
            panel_rows=panel_rows, **kwargs)

    def create_distributions(self):
        super().create_distributions()
        self.bijectors["v"] = "identity"
        self.bijectors["w"] = "identity"

    # log_likelihood_components (bernoulli.py:126-155) is the base class method: the
    # context's likelihood code makes spmf_dense_ll return the logits as 'rate' and
    # x*logit - softplus(logit) as 'log_likelihood'.
