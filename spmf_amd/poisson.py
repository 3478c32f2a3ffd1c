"""PoissonFactorization -- host-side mirror of the reference class surface
(mederrata_spmf/poisson.py:25-718) over the HIP hot path.

Only the energy path is re-implemented: ``unormalized_log_prob_parts`` and
its gradient run as hand-written HIP kernels behind the C-ABI of
``libspmf_hip.so`` (include/spmf_hip.h).  This module is argument plumbing:
it keeps the reference's names, argument meaning and error behaviour, owns
the torch tensors used as device storage and hands raw pointers to ctypes.
There is no CPU fallback; without the library / a GPU the energy raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict

import numpy as np
import torch

from . import _lib
from ._lib import PART_ORDER, VAR_ORDER, SpmfError
from .sparse import SparseCounts


def var_shapes(D: int, K: int) -> Dict[str, tuple]:
    """Event shapes of the 12 latent variables (poisson.py:228-377)."""
    return {
        "v": (K, D), "w": (1, D), "u": (D, K),
        "u_eta": (D, K), "u_tau": (1, K),
        "s_eta": (2, D), "s_tau": (1, D), "s": (2, D),
        "u_eta_a": (D, K), "u_tau_a": (1, K),
        "s_eta_a": (2, D), "s_tau_a": (1, D),
    }


class PoissonFactorization:
    """Sparse (horseshoe) poisson matrix factorization  (poisson.py:25-29).

    Constructor keywords are the reference's (poisson.py:56-64), including the
    ``horshoe_plus`` spelling.  ``device`` and ``panel_rows`` are additions.
    """
    bijectors = None
    var_list = []
    s_tau_scale = 1
    _dtype_notice_given = False   # the float64 notice is printed once per process
    _likelihood_flag = 0          # extra ctx flag of a subclass (BernoulliFactorization)
    _identity_vars = ()           # variables with an Identity bijector in the surrogate

    def __init__(
            self,
            latent_dim=None, feature_dim=None,
            u_tau_scale=0.01, s_tau_scale=1., symmetry_breaking_decay=0.99,
            strategy=None, encoder_function=None, decoder_function=None,
            scale_columns=True, scale_rows=True, log_transform=False,
            horshoe_plus=True, column_norms=None, count_key='counts',
            initialize_distributions=True,
            dtype=None, device=None, panel_rows=None,
            **kwargs):
        # poisson.py:94-97 lets callers swap g / f.  The kernels know the two built-in
        # pairs only; with a callable the energy takes the dense torch-on-device route of
        # spmf_amd/custom_codec.py -- announced, never silent.
        self._custom_codec = None
        if encoder_function is not None or decoder_function is not None:
            self._custom_codec = (encoder_function or self._builtin_encoder,
                                  decoder_function or self._builtin_decoder)
            print("custom encoder_function/decoder_function: the energy is evaluated densely with "
                  "torch ops on the device (spmf_amd/custom_codec.py); the HIP kernels cover the "
                  "built-in x/eta and log(x/eta+1) pairs only")
        self.strategy = strategy
        self.scale_rows = scale_rows
        self.scale_columns = scale_columns
        self.horseshoe_plus = horshoe_plus
        # the model's latent variables in the reference's surrogate order: all twelve
        # (poisson.py:403-539), or v, w, s, u for horshoe_plus=False (:378-398, :540-565)
        self.var_order = VAR_ORDER if horshoe_plus else _lib.VAR_ORDER_ABS
        self.eta_i = 1.
        self.xi_u_global = 1.
        if column_norms is not None:
            self.eta_i = column_norms
        self.count_key = count_key
        # the reference's default is float64 (poisson.py:64) and its CLI passes it (bin/factorize_csv.py:119);
        # a caller who ASKS for it is told once what runs instead
        self.dtype = torch.float64 if dtype is None else dtype
        if dtype is not None and "64" in str(dtype) and not PoissonFactorization._dtype_notice_given:
            PoissonFactorization._dtype_notice_given = True
            print("dtype=float64 requested: the device arithmetic of this build is float32 storage and FMA "
                  "with float64 accumulation of every scalar sum (energy parts to 1e-5 relative of the "
                  "float64 reference); the log_transform decoder evaluates exp(min(y, 70)) - 1 where "
                  "float64 is finite up to y = 709 (reported as 'Decoder saturated' while it happens)")
        self.symmetry_breaking_decay = symmetry_breaking_decay
        self.log_transform = log_transform
        self.feature_dim = feature_dim
        self.latent_dim = self.feature_dim if latent_dim is None else latent_dim
        self.u_tau_scale = u_tau_scale
        self.s_tau_scale = s_tau_scale
        self.panel_rows = panel_rows
        self.device = torch.device(
            device if device is not None else
            ("cuda" if torch.cuda.is_available() else "cpu"))
        self._ctx = None
        self._ws = None
        # deterministic=True (build-defined keyword, not in the reference): bit-reproducible energy and
        # gradients -- the step's atomics replaced by fixed-order sums (spmf_ctx_set_deterministic);
        # Poisson likelihood with the linear decoder only
        self.deterministic = bool(kwargs.pop("deterministic", False))
        self._det_buf = None
        self._det_ctx = None
        self._eta_dev = None
        self._eta_key = None
        self._batch_cache = {}
        self.max_cached_batches = 256
        self.max_cached_nnz = 1 << 28
        self.calibrated_expectations = {}
        self.surrogate_distribution = None
        self.surrogate_vars = []
        if initialize_distributions:
            self.create_distributions()
        print(
            f"Feature dim: {self.feature_dim} -> Latent dim {self.latent_dim}")

    def _builtin_encoder(self, x):
        """g (poisson.py:34-43) as a tensor function, for a model that swaps only f."""
        eta = self._eta_device().to(x.dtype)
        return torch.log(x / eta + 1.) if self.log_transform else x / eta

    def _builtin_decoder(self, y):
        """f (poisson.py:45-54) as a tensor function, for a model that swaps only g."""
        eta = self._eta_device().to(y.dtype)
        return torch.exp(y * eta) - 1. if self.log_transform else y * eta

    def _custom_energy(self, data, params, all_reduce, prior_weight):
        from . import custom_codec
        if all_reduce is not None and not (hasattr(all_reduce, "_sum") and hasattr(all_reduce, "gather_scalar")):
            raise NotImplementedError("custom encoder/decoder callables: row shards need a dist.ShardReducer "
                                      "as the all_reduce hook (a bare callable only sees the kernels' accumulators)")
        sc, cs = self._batch(data)
        x = sc.to_dense()
        pr = data.get("panels") if isinstance(data, dict) else None
        if pr is not None:
            r0 = pr[0] * sc.panel_rows
            x = x[r0:r0 + cs.n_rows]
        parts, grads, nbad = custom_codec.energy_and_grads(self, x, params, prior_weight, shard=all_reduce)
        S = nbad.shape[0]
        zero = torch.zeros(S, dtype=torch.float64, device=self.device)
        # (horshoe_plus=False has four variables: the other parts are 0, like the kernels')
        block = torch.stack([parts.get(n, zero) for n in PART_ORDER], 1).contiguous()
        self._last_parts = block
        self.last_saturated = torch.zeros(S, dtype=torch.float64, device=self.device)
        shapes = var_shapes(self.feature_dim, self.latent_dim)
        grads = {n: g.reshape((S,) + shapes[n]).contiguous() for n, g in grads.items()}
        return {n: block[:, i] for i, n in enumerate(PART_ORDER)
                if n in ("z", "x") or n in self.var_order}, grads, nbad

    # ------------------------------------------------------------------
    # native context
    # ------------------------------------------------------------------
    def _new_ctx(self, aux=False):
        if self.device.type != "cuda":
            raise SpmfError(
                "the HIP hot path needs a GPU device (no CPU fallback)")
        lib = _lib.load()
        flags = (_lib.FLAG_SCALE_ROWS if self.scale_rows else 0) | (
            _lib.FLAG_LOG_TRANSFORM if self.log_transform else 0) | self._likelihood_flag | (
            0 if self.horseshoe_plus else _lib.FLAG_ABS_HORSESHOE)
        h = C.c_void_p()
        rc = lib.spmf_ctx_create(self.device.index or 0, int(self.latent_dim),
                                 int(self.feature_dim), flags, C.byref(h))
        if rc != 0:
            raise SpmfError(
                f"spmf_ctx_create failed (rc={rc}); latent_dim must be in 1..256 (1..64 with "
                f"log_transform and for the Bernoulli / mixed likelihoods), got K={self.latent_dim}, "
                f"D={self.feature_dim}")
        _lib.check(h, lib.spmf_ctx_set_prior(
            h, float(self.u_tau_scale), float(self.s_tau_scale),
            float(self.symmetry_breaking_decay)), "spmf_ctx_set_prior")
        if aux:
            # the replacement rule's scan context never runs a data pass: keep its E buffer
            # (part of every workspace of a dense-term context) at the library's minimum
            _lib.check(h, lib.spmf_ctx_set_e_cap(h, 1 << 20), "spmf_ctx_set_e_cap")
        elif flags & (_lib.FLAG_LOG_TRANSFORM | _lib.FLAG_BERNOULLI | _lib.FLAG_MIXED):
            # room for E between the two dense contractions (dense.hip): a third of the device
            # memory torch can still hand out (free + its own cached blocks), at most 64 GiB
            # (the library's default is 8 GiB); fewer, larger row chunks fill the chip better
            free, _total = torch.cuda.mem_get_info(self.device)
            cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            cap = max(1 << 30, min(64 << 30, (int(free) + max(0, int(cached))) // 3))
            _lib.check(h, lib.spmf_ctx_set_e_cap(h, cap), "spmf_ctx_set_e_cap")
        return h

    def _handle(self):
        if self._ctx is None:
            self._ctx = self._new_ctx()
            if getattr(self, "column_split", 0):
                _lib.check(self._ctx, _lib.load().spmf_ctx_set_column_split(
                    self._ctx, int(self.column_split)), "spmf_ctx_set_column_split")
        return self._ctx

    def _aux(self, rows):
        """A second context with its own workspace for the dense fallback of the
        replacement rule: its launches must not re-carve the main workspace, whose
        accumulators the patch + finish that follow still need."""
        lib = _lib.load()
        if getattr(self, "_aux_ctx", None) is None:
            self._aux_ctx = self._new_ctx(aux=True)
            self._aux_ws = None
            self._after_aux_ctx(self._aux_ctx)
        need = lib.spmf_workspace_bytes(self._aux_ctx, int(rows), 1)
        if self._aux_ws is None or self._aux_ws.numel() < need + 256:
            self._aux_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            off = (-self._aux_ws.data_ptr()) % 256
            _lib.check(self._aux_ctx, lib.spmf_ctx_set_workspace(
                self._aux_ctx, self._aux_ws.data_ptr() + off, self._aux_ws.numel() - off),
                "spmf_ctx_set_workspace")
        return self._aux_ctx

    def _after_aux_ctx(self, h):
        """Hook for subclasses that configure a context further (column types)."""

    def enable_column_split(self, Dh=None):
        """Multi-GPU overlap (include/spmf_hip.h, spmf_ctx_set_column_split): lay
        the gradient accumulators out as two column halves so the all-reduce of
        the lower half runs while the column pass still produces the upper one.
        The batches must be built with the same split
        (``SparseCounts(..., col_split=model.column_split)``).  Returns Dh."""
        D = int(self.feature_dim)
        if Dh is None:
            Dh = (D // 2) // 32 * 32
        if Dh <= 0 or Dh >= D or Dh % 32:
            raise ValueError(f"column split {Dh} must be a multiple of 32 inside (0, {D})")
        self.column_split = int(Dh)
        if self._ctx is not None:
            _lib.check(self._ctx, _lib.load().spmf_ctx_set_column_split(self._ctx, int(Dh)),
                       "spmf_ctx_set_column_split")
        return self.column_split

    def __del__(self):
        try:
            for name in ("_ctx", "_aux_ctx"):
                if getattr(self, name, None) is not None:
                    _lib.load().spmf_ctx_destroy(getattr(self, name))
                    setattr(self, name, None)
        except Exception:
            pass

    def _ensure_workspace(self, rows, S):
        lib, h = _lib.load(), self._handle()
        need = lib.spmf_workspace_bytes(h, int(rows), int(S))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            base = self._ws.data_ptr()
            off = (-base) % 256
            self._ws_ptr = base + off
            _lib.check(h, lib.spmf_ctx_set_workspace(h, self._ws_ptr, self._ws.numel() - off),
                       "spmf_ctx_set_workspace")

    def _ensure_det_scratch(self, n_items, S):
        """Scratch of the deterministic mode for a batch of ``n_items`` work items and S draws."""
        lib, h = _lib.load(), self._handle()
        need = int(lib.spmf_det_scratch_bytes(h, int(n_items), int(S)))
        if self._det_buf is None or self._det_buf.numel() < need + 256 or self._det_ctx != h:
            if self._det_buf is None or self._det_buf.numel() < need + 256:
                # a step captured into a hipGraph (vi.StepRunner) keeps the pointer it was captured with:
                # an outgrown buffer stays alive beside the new one (growth by halves bounds how many)
                if self._det_buf is not None:
                    self.__dict__.setdefault("_det_old", []).append(self._det_buf)
                    need = max(need, 3 * self._det_buf.numel() // 2)
                self._det_buf = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._det_ctx = h
            base = self._det_buf.data_ptr()
            off = (-base) % 256
            _lib.check(h, lib.spmf_ctx_set_deterministic(h, base + off, self._det_buf.numel() - off),
                       "spmf_ctx_set_deterministic")

    def _eta_device(self):
        """eta_i as a [D] fp32 device vector (ones when unscaled)."""
        e = self.eta_i
        scalar = isinstance(e, (int, float))
        # arrays are compared by identity against a held reference (an id() alone
        # can be recycled once the old array is freed)
        if scalar:
            same = isinstance(self._eta_key, float) and self._eta_key == float(e)
        else:
            same = self._eta_key is e
        same = same and self._eta_dev is not None
        key = float(e) if scalar else e
        if not same:
            D = self.feature_dim
            if isinstance(e, (int, float)):
                t = torch.full((D,), float(e), dtype=torch.float32, device=self.device)
            else:
                if hasattr(e, "numpy") and not isinstance(e, torch.Tensor):
                    e = e.numpy()
                t = torch.as_tensor(np.asarray(e.cpu() if isinstance(e, torch.Tensor) else e,
                                               dtype=np.float64)).reshape(-1)
                if t.numel() == 1:
                    t = t.expand(D)
                t = t.to(torch.float32).to(self.device).contiguous()
            self._eta_dev, self._eta_key = t, key
        return self._eta_dev

    # ------------------------------------------------------------------
    # data plumbing
    # ------------------------------------------------------------------
    def _counts(self, data) -> SparseCounts:
        x = data[self.count_key] if isinstance(data, dict) else data
        if isinstance(x, SparseCounts):
            sc = x
        else:
            ck = id(x)
            hit = self._batch_cache.get(ck)
            if hit is not None and hit[0] is x:
                sc = hit[1]
            else:
                sc = SparseCounts.from_any(x, self.device, self.panel_rows,
                                             getattr(self, "column_split", 0), latent_dim=self.latent_dim)
                # device layouts of the most recent batches (an epoch loop over a
                # fixed list of host batches re-uses them; bounded by stored entries)
                self._batch_cache[ck] = (x, sc)
                tot = sum(h[1].nnz for h in self._batch_cache.values())
                while len(self._batch_cache) > 1 and (
                        len(self._batch_cache) > self.max_cached_batches
                        or tot > self.max_cached_nnz):
                    old = next(iter(self._batch_cache))
                    tot -= self._batch_cache.pop(old)[1].nnz
        if sc.n_cols != self.feature_dim:
            raise ValueError(
                f"counts have {sc.n_cols} features, model has {self.feature_dim}")
        if sc.row_sum is None:
            sc.compute_stats(self._handle())
        sc.set_row_scale(float(self.xi_u_global), self.scale_rows)
        if self.log_transform:
            sc.set_log_transform(self._eta_device(), self._handle())
        return sc

    def _batch(self, data):
        """-> (SparseCounts, spmf_counts struct) for a batch dict.  A batch may
        carry ``'panels': (p0, p1)`` to select a panel range of a resident
        shard (minibatching without re-sorting)."""
        sc = self._counts(data)
        pr = data.get("panels") if isinstance(data, dict) else None
        key = (pr, sc._xi_key, sc._g_key)
        cache = sc.__dict__.setdefault("_struct_cache", {})
        if key not in cache:
            cache[key] = sc.batch_struct(*(pr or (0, None)))
        return sc, cache[key]

    def _pack_params(self, params, names=None):
        """dict name -> tensor  =>  (S, {name: contiguous fp32 [S,*shape]})."""
        D, K = self.feature_dim, self.latent_dim
        shapes = var_shapes(D, K)
        out, S = {}, None
        for n in (names if names is not None else self.var_order):
            if n not in params:
                raise KeyError(f"missing parameter '{n}'")
            t = params[n]
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(np.asarray(t))
            t = t.to(device=self.device, dtype=torch.float32)
            if t.dim() == len(shapes[n]):
                t = t.unsqueeze(0)
            if tuple(t.shape[1:]) != shapes[n]:
                raise ValueError(f"parameter '{n}' has shape {tuple(t.shape)}, "
                                 f"expected [S,{shapes[n]}]")
            if S is None:
                S = t.shape[0]
            elif t.shape[0] != S:
                raise ValueError("all parameters must share the sample axis")
            out[n] = t.contiguous()
        return S, out

    # ------------------------------------------------------------------
    # the hot path
    # ------------------------------------------------------------------
    def energy_and_grads(self, data, params, all_reduce=None, prior_weight=1.0,
                         nonfinite="count", beside_columns=None):
        """All 14 energy parts (poisson.py:582-621) and d(sum of parts)/d(param)
        for every one of the 12 variables, for S draws, on the GPU.

        Returns ``(parts, grads, n_nonfinite)``: dict name -> [S] float64,
        dict name -> [S,*shape] float32 (gradient of x + z + prior_weight *
        prior parts), [S] float64.  ``all_reduce`` (optional
        callable taking the packed fp32 accumulator tensor) is invoked between
        the data pass and the finish kernel -- the single collective of the
        row-sharded multi-GPU path; it must also return the global
        (rows, lgamma_sum) via its return value or None for single shard.

        ``beside_columns`` (the VI step, spmf_amd/vi.py): ``((side_stream, rows_event), fn)`` -- the data
        pass reads v, w, u, s only, so the caller may hand over the OTHER parameter tensors unfilled: ``fn``
        fills them on ``side_stream`` once the row pass is done (the library records ``rows_event`` there:
        spmf_ctx_set_rows_event), the prior half of the finish follows it on that stream, and the finish
        joins -- all of it beside the column pass and the collective.

        ``nonfinite``: what happens when stored cells have a non-finite
        log-pmf (rate 0 under a positive count).  "count" (default, no host
        read-back): they are left out of 'x' and its gradient and counted in
        n_nonfinite.  "rule": the reference's replacement rule
        (poisson.py:606-616), value AND gradient -- one host read of the count
        per call, the dense fallback only when it is non-zero.  With row shards
        the fallback costs a second data pass and all-reduce (the minimum is taken
        over the shard minima, its gradient term comes from the shard that holds it).
        """
        if nonfinite not in ("count", "rule"):
            raise ValueError("nonfinite must be 'count' or 'rule'")
        if self._custom_codec is not None:
            return self._custom_energy(data, params, all_reduce, prior_weight)
        lib, h = _lib.load(), self._handle()
        sc, cs = self._batch(data)
        S, P = self._pack_params(params)
        self._ensure_workspace(cs.n_rows, S)
        if self.deterministic:
            self._ensure_det_scratch(cs.n_items, S)
        eta = self._eta_device()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # the C-ABI takes twelve slots in VAR_ORDER; variables the model does not have
        # (horshoe_plus=False) stay NULL
        pin = _lib.PtrArray(*[P[n].data_ptr() if n in P else None for n in VAR_ORDER])
        grads = {n: torch.empty_like(P[n]) for n in self.var_order}
        gout = _lib.PtrArray(*[grads[n].data_ptr() if n in grads else None for n in VAR_ORDER])
        parts = torch.empty(S, _lib.NPARTS, dtype=torch.float64, device=self.device)
        # [0:S] non-finite stored cells, [S:2S] saturated cells (log_transform)
        nnf2 = torch.empty(2 * S, dtype=torch.float64, device=self.device)
        nnf = nnf2[:S]
        rows_g, lg_g = cs.n_rows, cs.lgamma_sum
        # SPMF_LEGACY_STEP=1: the version-5 call sequence (spmf_data_pass [+ spmf_prior_async] + spmf_finish),
        # kept for A/B timing and for callers built against it; the results are the same
        legacy = os.environ.get("SPMF_LEGACY_STEP", "0") == "1"
        split = (all_reduce is not None and S == 1 and getattr(self, "column_split", 0) > 0
                 and sc.col_split == self.column_split and hasattr(all_reduce, "start"))
        if split:
            # column-split step: the all-reduce of the lower column half runs while the
            # column pass produces the upper half
            off = (C.c_int64 * 2)()
            ln = (C.c_int64 * 2)()
            _lib.check(h, lib.spmf_acc_split(h, off, ln), "spmf_acc_split")
            _lib.check(h, lib.spmf_data_pass_split(h, C.byref(cs), S, pin, eta.data_ptr(), 0, stream),
                       "spmf_data_pass_split")
            acc = _wrap_f32(lib.spmf_acc_ptr(h), off[1] + ln[1], self.device, self._ws)
            w0 = all_reduce.start(acc[off[0]:off[0] + ln[0]])
            _lib.check(h, lib.spmf_data_pass_split(h, C.byref(cs), S, pin, eta.data_ptr(), 1, stream),
                       "spmf_data_pass_split")
            _lib.check(h, lib.spmf_prior_async(h, S, float(prior_weight), pin, eta.data_ptr(),
                                               parts.data_ptr(), gout, stream), "spmf_prior_async")
            w1 = all_reduce.start(acc[off[1]:off[1] + ln[1]])
            all_reduce.wait(w0)
            all_reduce.wait(w1)
            r = all_reduce.totals(cs.n_rows, cs.lgamma_sum)
            if r is not None:
                rows_g, lg_g = r
        elif beside_columns is not None:
            (side, ev_rows), fill = beside_columns
            legacy = True                    # data pass, prior half on the side stream, finish joins it
            _lib.check(h, lib.spmf_ctx_set_rows_event(h, ev_rows.cuda_event), "spmf_ctx_set_rows_event")
            try:
                _lib.check(h, lib.spmf_data_pass(h, C.byref(cs), S, pin, eta.data_ptr(), stream),
                           "spmf_data_pass")
            finally:
                lib.spmf_ctx_set_rows_event(h, None)
            with torch.cuda.stream(side):
                side.wait_event(ev_rows)
                fill()
                _lib.check(h, lib.spmf_prior_async(h, S, float(prior_weight), pin, eta.data_ptr(),
                                                   parts.data_ptr(), gout, side.cuda_stream),
                           "spmf_prior_async")
        elif legacy:
            _lib.check(h, lib.spmf_data_pass(h, C.byref(cs), S, pin, eta.data_ptr(), stream),
                       "spmf_data_pass")
        else:
            # ABI 6: the outputs go in with the step's first call, so the prior half of the finish
            # (parameters only) runs inside the data pass's first launch
            _lib.check(h, lib.spmf_step_begin(h, C.byref(cs), S, float(prior_weight), pin, eta.data_ptr(),
                                              parts.data_ptr(), gout, nnf.data_ptr(), stream),
                       "spmf_step_begin")
        if all_reduce is not None and not split:
            # (version-5 flow only: the prior half of the finish on the library's side stream
            # while the collective has the GPU mostly idle; ShardReducer.overlap_prior)
            if legacy and beside_columns is None and getattr(all_reduce, "overlap_prior", True):
                _lib.check(h, lib.spmf_prior_async(h, S, float(prior_weight), pin, eta.data_ptr(),
                                                   parts.data_ptr(), gout, stream), "spmf_prior_async")
            n = lib.spmf_acc_len(h, S)
            acc = _wrap_f32(lib.spmf_acc_ptr(h), n, self.device, self._ws)
            r = all_reduce(acc, cs.n_rows, cs.lgamma_sum)
            if r is not None:
                rows_g, lg_g = r
        if split or legacy:
            _lib.check(h, lib.spmf_finish(h, S, int(rows_g), float(lg_g), float(prior_weight), pin,
                                          eta.data_ptr(), parts.data_ptr(), gout, nnf.data_ptr(), stream),
                       "spmf_finish")
        else:
            _lib.check(h, lib.spmf_step_end(h, int(rows_g), float(lg_g), stream), "spmf_step_end")
        if nonfinite == "rule" and float(nnf.sum()) > 0.0:
            io, nlg = self._nonfinite_scan(sc, cs, data, S, P)
            if all_reduce is not None:
                # row shards: nnf is already the global count (it came through the
                # all-reduce), so every rank is here.  The minimum is the minimum of the
                # shard minima; the shard that holds it owns the gradient term.  The
                # accumulators were summed in place, so this shard's own are rebuilt,
                # patched (value terms are per shard and add up; the gradient term is
                # the owner's, weighted by the GLOBAL count in io[2]) and summed again.
                if split or not hasattr(all_reduce, "gather_scalar"):
                    raise NotImplementedError(
                        "the non-finite replacement rule across row shards needs a reducer "
                        "with gather_scalar (spmf_amd.dist.ShardReducer) and the one-piece "
                        "all-reduce (no column split)")
                mins = all_reduce.gather_scalar(float(io[0]))
                m_g = min(mins)
                io[0] = m_g
                io[2] = float(nnf.sum())
                if all_reduce.rank != mins.index(m_g):
                    io[3] = float("inf")
                _lib.check(h, lib.spmf_data_pass(h, C.byref(cs), S, pin, eta.data_ptr(), stream),
                           "spmf_data_pass")
            _lib.check(h, lib.spmf_nonfinite_patch(h, C.byref(cs), S, pin, eta.data_ptr(),
                                                   io.data_ptr(), nlg.data_ptr(), stream),
                       "spmf_nonfinite_patch")
            if all_reduce is not None:
                all_reduce(_wrap_f32(lib.spmf_acc_ptr(h), lib.spmf_acc_len(h, S), self.device, self._ws),
                           cs.n_rows, cs.lgamma_sum)
            _lib.check(h, lib.spmf_finish(h, S, int(rows_g), float(lg_g), float(prior_weight), pin,
                                          eta.data_ptr(), parts.data_ptr(), gout, nnf.data_ptr(),
                                          stream), "spmf_finish")
        pd = {n: parts[:, i] for i, n in enumerate(PART_ORDER)
              if n in ("z", "x") or n in self.var_order}
        self._last_parts = parts                      # [S,14] block (spmf_vi_gate input)
        self.last_saturated = nnf2[S:]                # cells with exp() saturated (common.h kYSat)
        return pd, grads, nnf

    def unormalized_log_prob_parts(self, data, prior_weight=1., **params):
        """Energy function (poisson.py:582-621): dict of [S] tensors keyed
        v,w,u,...,z,x.  When a stored cell has a non-finite log-pmf the
        clip/replace rule (:606-616) is applied (dense fallback); otherwise it
        is the identity and the sparse fast path is the whole evaluation."""
        squeeze = params["u"].dim() == 2 if isinstance(params.get("u"), torch.Tensor) \
            else np.ndim(params["u"]) == 2
        parts, _, nnf = self.energy_and_grads(data, params, nonfinite="rule")
        out = {}
        for k, v in parts.items():
            if k not in ("x", "z"):
                v = v * prior_weight                       # poisson.py:591
            out[k] = v[0] if squeeze else v
        return out

    # ------------------------------------------------------------------
    # dense per-cell outputs (class surface; not the hot path)
    # ------------------------------------------------------------------
    def log_likelihood_components(self, s, u, v, w, data, *args, **kwargs):
        """Returns the log likelihood without summing along axes
        (poisson.py:156-184): {'log_likelihood': [S,B,D], 'rate': [S,B,D]}
        (no sample axis when the parameters have none)."""
        if self._custom_codec is not None:
            from . import custom_codec
            sc, cs = self._batch(data)
            xd = sc.to_dense()
            pr = data.get("panels") if isinstance(data, dict) else None
            if pr is not None:
                r0 = pr[0] * sc.panel_rows
                xd = xd[r0:r0 + cs.n_rows]
            return custom_codec.log_likelihood_components(self, xd, s, u, v, w)
        lib, h = _lib.load(), self._handle()
        sc, cs = self._batch(data)
        S, P = self._pack_params({"s": s, "u": u, "v": v, "w": w}, names=("s", "u", "v", "w"))
        single = (u.dim() if isinstance(u, torch.Tensor) else np.ndim(u)) == 2
        self._ensure_workspace(cs.n_rows, 1)
        eta = self._eta_device()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        B, D = cs.n_rows, self.feature_dim
        rate = torch.empty(S, B, D, dtype=torch.float32, device=self.device)
        ll = torch.empty(S, B, D, dtype=torch.float32, device=self.device)
        for i in range(S):
            _lib.check(h, lib.spmf_dense_ll(
                h, C.byref(cs), P["u"][i].data_ptr(), P["v"][i].data_ptr(),
                P["w"][i].data_ptr(), P["s"][i].data_ptr(), eta.data_ptr(),
                rate[i].data_ptr(), ll[i].data_ptr(), stream), "spmf_dense_ll")
        if single:
            rate, ll = rate[0], ll[0]
        return {'log_likelihood': ll, 'rate': rate}

    def predictive_distribution(self, s, u, v, w, data, *args, **kwargs):
        """poisson.py:187-210.  The reference sums a non-existent key 'll'
        (:206-208, KeyError whenever a sample axis is present); here the
        summed per-row log likelihood is added under that key instead."""
        prediction = self.log_likelihood_components(s=s, u=u, v=v, w=w, data=data)
        if prediction['log_likelihood'].dim() > 2:
            prediction['ll'] = prediction['log_likelihood'].sum(-1)
        return prediction

    def waic(self, data=None, nsamples=100):
        """Widely applicable information criterion on ONE batch, as the notebooks
        call it (notebooks/factorizing_random_noise.ipynb:447 prints
        {'waic','se','lppd','pwaic'}; the recorded lppd of -37091 is that of one
        1000 x 30 batch).  bayesianquilts' implementation is out of tree
        [UNVERIFIED-3P]; this is the standard pointwise definition over the
        cells of the batch: lppd_i = log mean_s p(x_i|theta_s),
        pwaic_i = var_s log p(x_i|theta_s), waic = -2 sum_i (lppd_i - pwaic_i),
        se = 2 sqrt(n var_i(lppd_i - pwaic_i))."""
        if data is None:
            src = getattr(self, "data", None)
            if src is None:
                raise ValueError("waic needs a batch (or a model built with data)")
            data = next(iter(src() if callable(src) else src))
        th = self.surrogate_distribution.sample(int(nsamples))
        ll = self.log_likelihood_components(s=th["s"], u=th["u"], v=th["v"], w=th["w"],
                                            data=data)["log_likelihood"].double()
        S = ll.shape[0]
        lppd_i = torch.logsumexp(ll, 0) - math.log(S)
        pwaic_i = ll.var(0, unbiased=True)
        elpd_i = lppd_i - pwaic_i
        n = elpd_i.numel()
        return {"waic": float(-2.0 * elpd_i.sum()),
                "se": float(2.0 * torch.sqrt(n * elpd_i.var(unbiased=True))),
                "lppd": float(lppd_i.sum()), "pwaic": float(pwaic_i.sum())}

    def _nonfinite_scan(self, sc, cs, data, S, P, max_cells=1 << 27):
        """Dense part of the replacement rule (poisson.py:606-616): the minimum
        of the per-cell log-pmf over ALL S*B*D cells (finite ones; the
        reference's where(finite, ll, 0) also puts a 0 into it) and the linear
        index of the cell that attains it.  Evaluated by spmf_dense_ll over row
        chunks of whole panels (at most ``max_cells`` cells at a time), so the
        dense [B,D] block never has to exist.  Returns (io, nlg): io = double[4]
        on the device, [0] minimum, [3] its cell (+inf if the minimum is the 0);
        nlg = double[S], per draw the sum of lgamma(x+1) over the replaced cells."""
        lib = _lib.load()
        eta = self._eta_device()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        D = self.feature_dim
        pr = data.get("panels") if isinstance(data, dict) else None
        p0, p1 = (pr or (0, None))
        p1 = sc.n_panels if p1 is None else min(int(p1), sc.n_panels)
        step = max(1, max_cells // max(1, sc.panel_rows * D))
        h = self._aux(min(cs.n_rows, step * sc.panel_rows))
        io = torch.zeros(4, dtype=torch.float64, device=self.device)
        io[3] = float("inf")
        nlg = torch.zeros(S, dtype=torch.float64, device=self.device)
        buf_rows = min(cs.n_rows, step * sc.panel_rows)
        rate = torch.empty(buf_rows * D, dtype=torch.float32, device=self.device)
        ll = torch.empty(buf_rows * D, dtype=torch.float32, device=self.device)
        key = (sc._xi_key, sc._g_key)

        def sweep(fn):
            for i in range(S):
                for q0 in range(int(p0), p1, step):
                    q1 = min(q0 + step, p1)
                    sub = sc.__dict__.setdefault("_struct_cache", {}).setdefault(
                        ((q0, q1),) + key, sc.batch_struct(q0, q1))
                    _lib.check(h, lib.spmf_dense_ll(
                        h, C.byref(sub), P["u"][i].data_ptr(), P["v"][i].data_ptr(),
                        P["w"][i].data_ptr(), P["s"][i].data_ptr(), eta.data_ptr(),
                        rate.data_ptr(), ll.data_ptr(), stream), "spmf_dense_ll")
                    r0 = (q0 - int(p0)) * sc.panel_rows
                    fn(sub, i, sub.n_rows * D, float(i) * cs.n_rows * D + float(r0) * D)

        def first(sub, i, n, base):
            _lib.check(h, lib.spmf_nonfinite_reduce(h, n, ll.data_ptr(), 0, io.data_ptr(), stream),
                       "spmf_nonfinite_reduce")
            _lib.check(h, lib.spmf_nonfinite_lgamma(h, C.byref(sub), rate.data_ptr(),
                                                    nlg[i:].data_ptr(), stream), "spmf_nonfinite_lgamma")
        sweep(first)
        sweep(lambda sub, i, n, base: _lib.check(h, lib.spmf_nonfinite_argmin(
            h, n, ll.data_ptr(), base, io.data_ptr(), stream), "spmf_nonfinite_argmin"))
        return io, nlg

    def unormalized_log_prob(self, data=None, prior_weight=1., **params):
        """poisson.py:575-580 -- NB: like the reference this ignores
        ``prior_weight`` and sums the parts with weight 1 (:577)."""
        prob_parts = self.unormalized_log_prob_parts(data, prior_weight=1., **params)
        return sum(prob_parts.values())

    def unormalized_log_prob_list(self, *x, data=None):
        """poisson.py:703-709: positional wrapper in var_list order."""
        return self.unormalized_log_prob(
            data=data, **{v: t for v, t in zip(self.var_list, x)})

    # ------------------------------------------------------------------
    # small O(D*K) helpers (plain tensor algebra, not on the hot path)
    # ------------------------------------------------------------------
    def _expect(self, name, value):
        if value is not None:
            return value if isinstance(value, torch.Tensor) else torch.as_tensor(
                np.asarray(value), device=self.device)
        if name not in self.calibrated_expectations:
            raise KeyError(
                f"no calibrated expectation for '{name}': fit the model or pass it")
        return self.calibrated_expectations[name]

    def encoding_matrix(self, u=None, s=None):
        """Output A = (alpha_ik)  (poisson.py:652-666): batch_shape x I x K"""
        u = self._expect("u", u)
        s = self._expect("s", s)
        weights = s / s.sum(-2, keepdim=True)
        return weights[..., 0, :].unsqueeze(-1) * u

    def decoding_matrix(self, v=None):
        """Output B = (beta_ki)  (poisson.py:668-678)"""
        return self._expect("v", v)

    def intercept_matrix(self, w=None, s=None):
        """export phi  (poisson.py:680-701): batch_shape x 1 x I"""
        w = self._expect("w", w)
        s = self._expect("s", s)
        weights = s / s.sum(-2, keepdim=True)
        eta = self._eta_device().to(w.dtype)
        return eta * weights[..., 1, :].unsqueeze(-2) * w

    def encode(self, x, u=None, s=None):
        """Returns theta given x (poisson.py:623-650), [B,K] (or [S,B,K])."""
        u = self._expect("u", u).to(self.device, torch.float32)
        s = self._expect("s", s).to(self.device, torch.float32)
        if self._custom_codec is not None:
            sc, cs = self._batch(x if isinstance(x, dict) else {self.count_key: x})
            xd = sc.to_dense().double()
            wts = (s / s.sum(-2, keepdim=True)).double()
            z = torch.matmul(self._custom_codec[0](xd), wts[..., 0, :].unsqueeze(-1) * u.double())
            if self.scale_rows:
                z = z * (xd.sum(-1, keepdim=True) / float(self.xi_u_global))
            return z.to(torch.float32)
        lib, h = _lib.load(), self._handle()
        sc, cs = self._batch(x if isinstance(x, dict) else {self.count_key: x})
        single = u.dim() == 2
        if single:
            u, s = u.unsqueeze(0), s.unsqueeze(0)
        self._ensure_workspace(cs.n_rows, 1)
        eta = self._eta_device()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        out = torch.empty(u.shape[0], cs.n_rows, self.latent_dim,
                          dtype=torch.float32, device=self.device)
        for i in range(u.shape[0]):
            ui, si = u[i].contiguous(), s[i].contiguous()
            _lib.check(h, lib.spmf_encode(h, C.byref(cs), ui.data_ptr(), si.data_ptr(),
                                          eta.data_ptr(), out[i].data_ptr(), stream),
                       "spmf_encode")
        return out[0] if single else out

    # ------------------------------------------------------------------
    # compute_scales (poisson.py:113-154)
    # ------------------------------------------------------------------
    def compute_scales(self, data_factory, compute_normalization=True, n=None, all_reduce=None):
        """``all_reduce``: with row shards (``data_factory`` yields THIS rank's rows)
        a spmf_amd.dist.ShardReducer; its reduce_stats sums the column statistics
        once over the ranks, so every rank ends with the same eta_i / xi_u_global
        and the reducer knows the dataset's global row count."""
        if self.scale_columns and compute_normalization:
            print("Looping through the entire dataset once to get some stats")
            D = self.feature_dim
            colsum = torch.zeros(D, dtype=torch.float64, device=self.device)
            colnnz = torch.zeros(D, dtype=torch.float64, device=self.device)
            N, lg = 0, 0.0
            h = self._handle()
            for batch in iter(data_factory()):
                x = batch[self.count_key] if isinstance(batch, dict) else batch
                sc = SparseCounts.from_any(x, self.device, self.panel_rows, latent_dim=self.latent_dim)
                sc.compute_stats(h, colsum, colnnz)
                N += sc.n_rows
                if all_reduce is not None:
                    lg += float(sc.row_lgamma.sum())
            if all_reduce is not None:
                all_reduce.reduce_stats(colsum, colnnz, N, lg)
            colmeans_nonzero = colsum / colnnz          # NaN for empty columns
            # poisson.py:139-140 sums NaNs into xi for an empty column; the
            # build defines xi over the non-empty columns.
            rowmean_nonzero = torch.nansum(colmeans_nonzero)
            self.eta_i = torch.where(colmeans_nonzero > 1, colmeans_nonzero,
                                     torch.ones_like(colmeans_nonzero)).reshape(1, D)
            if self.scale_rows:
                self.xi_u_global = float(rowmean_nonzero)
            else:
                self.xi_u_global = 1.

    # ------------------------------------------------------------------
    # distributions / driver hooks live in vi.py
    # ------------------------------------------------------------------
    def create_distributions(self):
        """poisson.py:212-573: bijectors, var_list and the surrogate posterior
        (initial values :403-539).  The prior itself lives in the finish
        kernel."""
        from .vi import Surrogate
        self.bijectors = {n: "softplus" for n in self.var_order}
        self.surrogate_distribution = Surrogate(self)
        self.surrogate_vars = self.surrogate_distribution.variables
        self.var_list = list(self.var_order)
        self.set_calibration_expectations()

    def set_calibration_expectations(self, samples=32):
        if self.device.type != "cuda":
            # sampling the surrogate runs in the HIP kernels: without a device the
            # expectations stay unset (encode()/encoding_matrix() then ask for them)
            self.calibrated_expectations = {}
            return
        self.calibrated_expectations = \
            self.surrogate_distribution.expectations(samples)

    def reconstitute(self, state):
        """poisson.py:711-717: rebuild, then assign surrogate variables BY
        POSITION (the ordering is part of the pickle format)."""
        self.create_distributions()
        tv = self.surrogate_distribution.trainable_variables
        if len(state['surrogate_vars']) != len(tv):
            raise ValueError(f"checkpoint holds {len(state['surrogate_vars'])} surrogate "
                             f"variables, the model has {len(tv)}")
        with torch.no_grad():
            for j, value in enumerate(state['surrogate_vars']):
                src = torch.as_tensor(np.asarray(value))
                if tuple(src.shape) != tuple(tv[j].shape):
                    raise ValueError(f"surrogate variable {j}: checkpoint shape "
                                     f"{tuple(src.shape)}, model {tuple(tv[j].shape)}")
                tv[j].copy_(src.to(tv[j]))
        for k in ("eta_i", "xi_u_global"):
            if k in state and state[k] is not None:
                setattr(self, k, state[k])

    # fit / calibrate_advi / save / waic are attached in vi.py
    def fit(self, *args, **kwargs):
        from .vi import fit
        return fit(self, *args, **kwargs)

    def calibrate_advi(self, *args, **kwargs):
        from .vi import calibrate_advi
        return calibrate_advi(self, *args, **kwargs)

    def save(self, filename):
        from .vi import save_model
        return save_model(self, filename)


def _wrap_f32(ptr, n, device, owner):
    """View n floats at device address ``ptr`` (inside ``owner``'s storage) as
    a torch tensor without copying."""
    base = owner.data_ptr()
    off = ptr - base
    assert off >= 0 and off % 4 == 0
    return owner[off:off + 4 * n].view(torch.float32)


class PoissonMatrixFactorization(PoissonFactorization):
    """Legacy name/constructor used by bin/factorize_csv.py:114-119, the scRNA
    script and every notebook: first positional argument is the (batched)
    dataset; feature_dim is inferred from it; unknown legacy keywords
    (scale_rates, with_s, encoder, decoder, ...) are swallowed."""

    def __init__(self, data=None, latent_dim=None, **kwargs):
        for legacy in ("scale_rates", "with_s", "encoder", "decoder",
                       "fn", "fn_inverse", "auxiliary_horseshoe"):
            kwargs.pop(legacy, None)
        feature_dim = kwargs.pop("feature_dim", None)
        self.data = data
        if feature_dim is None and data is not None:
            first = next(iter(data() if callable(data) else data))
            x = first[kwargs.get("count_key", "counts")] if isinstance(first, dict) else first
            feature_dim = x.n_cols if isinstance(x, SparseCounts) else x.shape[-1]
        super().__init__(latent_dim=latent_dim, feature_dim=feature_dim, **kwargs)
