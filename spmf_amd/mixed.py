"""MixedFactorization -- per-column Poisson / Bernoulli likelihood.

BUILD-DEFINED: ``mederrata_spmf/mixed.py`` is a 0-byte file in the reference
(BASELINE.json config 5 names it), so there are no reference
semantics to match and parity is unpinned.  The definition here is the
natural combination of the two classes that do exist:

  * columns flagged in ``bernoulli_columns`` follow bernoulli.py:
    Bernoulli(logits = rate) likelihood (:147-155), Identity bijector and
    Normal(0,.1) / Normal(0,1) priors on their entries of v / w (:187-216);
  * the other columns follow poisson.py (Poisson(rate), Softplus bijector,
    HalfNormal priors);
  * one shared encoder z = xi * g(x) A over ALL columns (poisson.py:623-650).

On the device the Poisson columns use the closed-form sum over implicit zeros,
the Bernoulli columns the dense softplus/sigmoid f32-MFMA kernel with a logit
bias of -1e30 on the Poisson columns (sigmoid = softplus = 0 there).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .poisson import PoissonFactorization


class MixedFactorization(PoissonFactorization):
    _likelihood_flag = _lib.FLAG_MIXED

    def __init__(self, bernoulli_columns, latent_dim=None, feature_dim=None, **kwargs):
        mask = np.asarray(bernoulli_columns, dtype=bool).reshape(-1)
        if feature_dim is None:
            feature_dim = mask.size
        if mask.size != feature_dim:
            raise ValueError("bernoulli_columns must have one flag per feature")
        if kwargs.get("log_transform"):
            raise NotImplementedError("MixedFactorization supports the linear decoder only")
        self.bernoulli_columns = mask
        self.compact_bernoulli = bool(kwargs.pop("compact_bernoulli", True))
        self._ctype_dev = None
        super().__init__(latent_dim=latent_dim, feature_dim=feature_dim, **kwargs)

    def _handle(self):
        first = self._ctx is None
        h = super()._handle()
        if first:
            self._ctype_dev = torch.as_tensor(self.bernoulli_columns.astype(np.uint8)).to(self.device)
            _lib.check(h, _lib.load().spmf_ctx_set_column_types(h, self._ctype_dev.data_ptr()),
                       "spmf_ctx_set_column_types")
            # the dense softplus sums only need the Bernoulli columns: hand the library their list
            self._bcols_dev = torch.as_tensor(np.flatnonzero(self.bernoulli_columns).astype(np.int32)
                                              ).to(self.device)
            if self.compact_bernoulli and self._bcols_dev.numel() > 0:
                _lib.check(h, _lib.load().spmf_ctx_set_bernoulli_columns(
                    h, self._bcols_dev.data_ptr(), int(self._bcols_dev.numel())),
                    "spmf_ctx_set_bernoulli_columns")
        return h

    def create_distributions(self):
        # per-element Identity flags for the surrogate of v [K,D] and w [1,D]
        m = torch.as_tensor(self.bernoulli_columns.astype(np.uint8)).to(self.device)
        K, D = self.latent_dim, self.feature_dim
        self._identity_mask = {"v": m.view(1, D).expand(K, D).contiguous(),
                               "w": m.view(1, D).contiguous()}
        super().create_distributions()
        self.bijectors["v"] = "softplus | identity (per column)"
        self.bijectors["w"] = "softplus | identity (per column)"

    def _after_aux_ctx(self, h):
        self._handle()                       # makes sure the column-type vector exists
        _lib.check(h, _lib.load().spmf_ctx_set_column_types(h, self._ctype_dev.data_ptr()),
                   "spmf_ctx_set_column_types")
