"""Minimal VI driver -- the build's counterpart of the out-of-tree caller of
the hot path (bayesianquilts' BayesianModel.fit / legacy calibrate_advi;
call sites tests/spmf_test.py:35-43 and bin/factorize_csv.py:121-124).

[UNVERIFIED-3P] bayesianquilts is an un-vendored, un-pinned dependency, so
its exact loss scaling / plateau logic cannot be mirrored; this module
DEFINES:

    loss = - mean_S [ LL_x + logp_z + (B/N) (logp_prior - log q) ] / B

with the raw parts available from ``PoissonFactorization.energy_and_grads``.
The surrogate is the reference's (poisson.py:403-539): Softplus(Normal) for
u, v, w, s and Softplus(InverseGamma) for the scale hierarchy; positive
distribution parameters are softplus(raw) trainables.

The VI step runs on the device end to end in HIP kernels behind the C-ABI:
the base noise (Philox normal / gamma sampler with the implicit-
reparameterisation derivative), the transform to theta, log q, the energy and
its gradient, the chain back to the trainables and the Adam update
(surrogate.hip + the hot path).  torch supplies storage and the 64-bit seed.
(The plain torch-autograd restatement of the same step that the tests compare
against lives in tests/_vi_reference.py, not in the product.)
"""
from __future__ import annotations

import math
import os
from typing import Dict, List

import numpy as np
import torch


from . import _lib
from ._lib import VAR_ORDER

_sp = torch.nn.functional.softplus


def softplus_inverse(y):
    y = np.asarray(y, dtype=np.float64)
    return y + np.log(-np.expm1(-y))


def _initial_state(D, K, u_tau_scale, s_tau_scale, horseshoe_plus=True):
    """(kind, constrained initial parameters) per variable, poisson.py:403-539."""
    from .poisson import var_shapes
    sh = var_shapes(D, K)
    one = lambda n: np.ones(sh[n])
    return {
        "v": ("normal", -6.0 * one("v"), 5e-4 * one("v")),
        "w": ("normal", -6.0 * one("w"), 5e-4 * one("w")),
        "u": ("normal", (-6.0 if horseshoe_plus else -9.0) * one("u"), 5e-4 * one("u")),   # :427-437 / :556
        "u_eta": ("invgamma", 3.0 * one("u_eta"), one("u_eta")),
        "u_tau": ("invgamma", 3.0 * one("u_tau"), one("u_tau")),
        "s_eta": ("invgamma", one("s_eta"), one("s_eta")),
        "s_tau": ("invgamma", one("s_tau"), one("s_tau")),
        "s": ("normal", one("s") * np.array([[-2.0], [-1.0]]), 1e-3 * one("s")),
        "u_eta_a": ("invgamma", 2.0 * one("u_eta_a"), one("u_eta_a")),
        "u_tau_a": ("invgamma", 2.0 * one("u_tau_a"), one("u_tau_a") / u_tau_scale ** 2),
        "s_eta_a": ("invgamma", 2.0 * one("s_eta_a"), one("s_eta_a")),
        "s_tau_a": ("invgamma", 2.0 * one("s_tau_a"), one("s_tau_a") / s_tau_scale ** 2),
    }


class Surrogate:
    """Mean-field surrogate posterior, poisson.py:403-569.  Trainables are kept
    in the reference's order (two per variable, VAR_ORDER) because
    ``reconstitute`` assigns them by position (poisson.py:711-717)."""

    def __init__(self, model):
        import weakref
        D, K = model.feature_dim, model.latent_dim
        dev = model.device
        self.device = dev
        self._model_ref = weakref.ref(model)      # no cycle: the model owns the surrogate
        self.kinds: Dict[str, str] = {}
        self.trainable_variables: List[torch.Tensor] = []
        self._index = {}
        self.var_order = tuple(getattr(model, "var_order", VAR_ORDER))
        init = _initial_state(D, K, model.u_tau_scale, model.s_tau_scale,
                              getattr(model, "horseshoe_plus", True))
        identity = set(getattr(model, "_identity_vars", ()))
        # per-element Identity flags (mixed likelihood): name -> uint8 tensor
        self.ident_mask = dict(getattr(model, "_identity_mask", {}) or {})
        staged = []
        for n in self.var_order:
            kind, a, b = init[n]
            if n in identity:          # tfb.Identity(Normal): bernoulli.py:187-193,362-381
                kind = "normal_identity"
            self.kinds[n] = kind
            if kind in ("normal", "normal_identity"):
                t0, t1 = a, softplus_inverse(b)                 # loc, raw scale
            else:
                t0, t1 = softplus_inverse(a), softplus_inverse(b)  # raw conc, raw scale
            staged.append((n, kind, t0, t1))
        # the raw concentrations of all Softplus(InverseGamma) variables live in one
        # flat buffer (each trainable is a view of it), so one softplus + one gamma
        # draw per step covers them all without a concatenation
        self._gam_names = [n for n, kind, _, _ in staged if kind == "invgamma"]
        total = sum(int(np.asarray(t0).size) for n, kind, t0, _ in staged if kind == "invgamma")
        self._gam_flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for n, kind, t0, t1 in staged:
            self._index[n] = len(self.trainable_variables)
            if kind == "invgamma":
                src = torch.as_tensor(np.asarray(t0), dtype=torch.float32)
                view = self._gam_flat[off:off + src.numel()].view(src.shape)
                view.copy_(src)
                off += src.numel()
                view.requires_grad_(True)
                self.trainable_variables.append(view)
            else:
                self.trainable_variables.append(
                    torch.tensor(t0, dtype=torch.float32, device=dev, requires_grad=True))
            self.trainable_variables.append(
                torch.tensor(t1, dtype=torch.float32, device=dev, requires_grad=True))

    @property
    def variables(self):
        return self.trainable_variables

    def params_of(self, n):
        i = self._index[n]
        return self.trainable_variables[i], self.trainable_variables[i + 1]

    # ---- HIP path -------------------------------------------------------
    _KIND = {"normal": 0, "normal_identity": 1, "invgamma": 2}

    def alloc_noise(self, S):
        """Buffers of the base noise of all variables (name -> (noise, dgda or None)): a variable's noise is
        a column slice of an [S, total] buffer (row stride = total, passed as noise_ld)."""
        out = {}
        nor = [n for n in self.var_order if self.kinds[n] != "invgamma"]
        if nor:
            sizes = [self.params_of(n)[0].numel() for n in nor]
            eps = torch.empty(S, sum(sizes), device=self.device, dtype=torch.float32)
            for n, piece in zip(nor, eps.split(sizes, dim=1)):
                out[n] = (piece.view((S,) + tuple(self.params_of(n)[0].shape)), None)
        if self._gam_names:
            sizes = [self.params_of(n)[0].numel() for n in self._gam_names]
            g = torch.empty(S, sum(sizes), device=self.device, dtype=torch.float32)
            dg = torch.empty(S, sum(sizes), device=self.device, dtype=torch.float32)
            for n, gp, dp in zip(self._gam_names, g.split(sizes, dim=1), dg.split(sizes, dim=1)):
                shape = (S,) + tuple(self.params_of(n)[0].shape)
                out[n] = (gp.view(shape), dp.view(shape))
        return out

    @torch.no_grad()
    def draw_noise(self, S, seed=None, state=None, only=None, out=None):
        """Base noise per variable, drawn by the HIP sampler (spmf_sample_noise:
        Philox4x32-10; eps ~ N(0,1), or for the InverseGamma kinds g ~
        Gamma(softplus(t0), 1) and d g/d concentration by implicit
        reparameterisation).  One launch for ALL variables (or for the names in ``only``:
        the others are skipped, the draws do not depend on which call covers a variable).
        ``seed``: 64-bit key; None draws one from torch's CPU generator, so
        ``torch.manual_seed`` still makes a run reproducible and ranks that share a
        seed (dist.sync_seed) draw identical noise.  ``state``: the optimiser's
        device state; its step counter (advanced by spmf_vi_gate) is added to the
        Philox counter, which is what gives a hipGraph replay fresh noise.
        ``out``: buffers from alloc_noise (allocated here when None)."""
        model = self._model()
        lib, h = _lib.load(), model._handle()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        if out is None:
            out = self.alloc_noise(S)
        arr = self._table(S, out, only=only)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(h, lib.spmf_sample_noise(h, arr, len(self.var_order), S, seed, 0,
                                            state.data_ptr() if state is not None else None,
                                            stream), "spmf_sample_noise")
        return out

    def _table(self, S, noise, theta=None, gtheta=None, grads=None, only=None):
        arr = (_lib.SurVar * len(self.var_order))()
        for i, n in enumerate(self.var_order):
            if only is not None and n not in only:
                arr[i].n = 0                    # skipped by spmf_sample_noise / spmf_surrogate_fwd
                continue
            t0, t1 = self.params_of(n)
            nz, dg = noise[n]
            v = arr[i]
            v.t0, v.t1 = t0.data_ptr(), t1.data_ptr()
            v.noise = nz.data_ptr()
            v.noise_ld = nz.stride(0) if nz.dim() > 1 and nz.shape[0] > 1 else 0
            v.dgda = dg.data_ptr() if dg is not None else None
            v.theta = theta[n].data_ptr() if theta is not None else None
            v.gtheta = gtheta[n].data_ptr() if gtheta is not None else None
            v.g0 = grads[2 * i].data_ptr() if grads is not None else None
            v.g1 = grads[2 * i + 1].data_ptr() if grads is not None else None
            v.n, v.kind = t0.numel(), self._KIND[self.kinds[n]]
            im = self.ident_mask.get(n)
            v.ident = im.data_ptr() if im is not None else None
        return arr

    @torch.no_grad()
    def forward_hip(self, model, S, noise, only=None, theta=None, logq=None):
        """theta (dict name -> [S,*shape]) and logq [S] (float64) by the HIP kernel; ``only``: the
        variables to transform (logq is then THEIR share of log q); ``theta`` / ``logq``: outputs
        to fill (allocated here when None)."""
        lib, h = _lib.load(), model._handle()
        if theta is None:
            theta = {n: torch.empty(noise[n][0].shape, dtype=torch.float32, device=self.device)
                     for n in self.var_order}
        if logq is None:
            logq = torch.empty(S, dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        arr = self._table(S, noise, theta=theta, only=only)
        _lib.check(h, lib.spmf_surrogate_fwd(h, arr, len(self.var_order), S, logq.data_ptr(), stream),
                   "spmf_surrogate_fwd")
        return theta, logq

    @torch.no_grad()
    def draw_and_forward(self, model, S, seed=None, state=None, only=None):
        """draw_noise + forward_hip in ONE launch (spmf_sample_transform): -> (noise, theta, logq), the
        same draws and theta bits as the two calls, logq to fp64 rounding."""
        lib, h = _lib.load(), model._handle()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        noise = self.alloc_noise(S)
        theta = {n: torch.empty(noise[n][0].shape, dtype=torch.float32, device=self.device)
                 for n in self.var_order}
        logq = torch.empty(S, dtype=torch.float64, device=self.device)
        arr = self._table(S, noise, theta=theta, only=only)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(h, lib.spmf_sample_transform(h, arr, len(self.var_order), S, seed, 0,
                                                state.data_ptr() if state is not None else None,
                                                logq.data_ptr(), stream), "spmf_sample_transform")
        return noise, theta, logq

    @torch.no_grad()
    def backward_hip(self, model, S, noise, gtheta, inv_sb, c):
        """d loss / d trainables (list in trainable order) given dE/dtheta."""
        lib, h = _lib.load(), model._handle()
        grads = [torch.empty_like(p) for p in self.trainable_variables]
        gt = {n: gtheta[n].contiguous() for n in self.var_order}
        stream = torch.cuda.current_stream(self.device).cuda_stream
        arr = self._table(S, noise, gtheta=gt, grads=grads)
        _lib.check(h, lib.spmf_surrogate_bwd(h, arr, len(self.var_order), S, float(inv_sb), float(c),
                                             stream), "spmf_surrogate_bwd")
        return grads

    @torch.no_grad()
    def backward_adam_hip(self, model, S, noise, gtheta, inv_sb, c, opt):
        """backward_hip + AdamHIP.step_dev fused (spmf_surrogate_bwd_adam_dev): the
        step path of the training loop, gated by the optimiser's device state."""
        lib, h = _lib.load(), model._handle()
        if opt.params is not self.trainable_variables and any(
                a is not b for a, b in zip(opt.params, self.trainable_variables)):
            raise ValueError("the optimiser must own the surrogate's trainables, in order")
        gt = {n: gtheta[n].contiguous() for n in self.var_order}
        arr = self._table(S, noise, gtheta=gt)
        av = (_lib.AdamVar * len(opt.params))()
        for i, (p, m, v) in enumerate(zip(opt.params, opt.m, opt.v)):
            a = av[i]
            a.p, a.m, a.v, a.g, a.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), None, p.numel()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(h, lib.spmf_surrogate_bwd_adam_dev(
            h, arr, len(self.var_order), S, float(inv_sb), float(c), av, opt.state.data_ptr(),
            stream), "spmf_surrogate_bwd_adam_dev")

    def _model(self):
        m = self._model_ref() if self._model_ref is not None else None
        if m is None:
            raise RuntimeError("the surrogate's model is gone")
        return m

    @torch.no_grad()
    def sample(self, n=1):
        """n draws theta ~ q (dict name -> [n,*shape]); surrogate_distribution.sample(n)
        of the reference's callers (bin/factorize_csv.py:155)."""
        n = int(n)
        theta, _ = self.forward_hip(self._model(), n, self.draw_noise(n))
        return theta

    @torch.no_grad()
    def expectations(self, samples=32):
        return {k: v.mean(0) for k, v in self.sample(samples).items()}


class AdamHIP:
    """The same update as ``Adam`` in one HIP launch over all trainables
    (spmf_adam_step), value clipping fused."""

    def __init__(self, model, params, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.model, self.params, self.lr = model, params, lr
        self.b1, self.b2, self.eps, self.t = beta1, beta2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.state = None                                 # device state of step_dev()

    # -- device-resident state (include/spmf_hip.h, SPMF_VI_STATE_LEN) ----------
    def init_state(self, clip_value=None):
        st = torch.zeros(_lib.VI_STATE_LEN, dtype=torch.float64)
        st[0], st[1], st[2], st[3] = self.lr, self.b1, self.b2, self.eps
        st[4] = float(clip_value) if clip_value else 0.0
        st[5], st[6] = self.b1 ** self.t, self.b2 ** self.t
        st[7] = self.t
        self.state = st.to(self.params[0].device)
        return self.state

    def set_lr(self, lr):
        self.lr = lr
        if self.state is not None:
            self.state[0:1].fill_(lr)

    def read_state(self):
        """Host copy of the device state (one sync)."""
        return self.state.cpu().tolist()

    def reset_epoch_counters(self):
        self.state[10:13].zero_()
        self.state[14:15].zero_()

    @torch.no_grad()
    def step_dev(self, grads):
        """Adam update gated by state[9]; every scalar comes from the device
        state, so the launch is identical from step to step (graph-capturable)."""
        lib, h = _lib.load(), self.model._handle()
        arr = (_lib.AdamVar * len(self.params))()
        for i, (p, g, m, v) in enumerate(zip(self.params, grads, self.m, self.v)):
            if not g.is_contiguous():
                raise ValueError("step_dev needs contiguous gradients")
            a = arr[i]
            a.p, a.m, a.v, a.g, a.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p.numel()
        stream = torch.cuda.current_stream(p.device).cuda_stream
        _lib.check(h, lib.spmf_adam_step_dev(h, arr, len(self.params), self.state.data_ptr(),
                                             stream), "spmf_adam_step_dev")

    @torch.no_grad()
    def step(self, grads, clip_value=None):
        lib, h = _lib.load(), self.model._handle()
        self.t += 1
        arr = (_lib.AdamVar * len(self.params))()
        keep = []
        for i, (p, g, m, v) in enumerate(zip(self.params, grads, self.m, self.v)):
            g = g.contiguous()
            keep.append(g)
            a = arr[i]
            a.p, a.m, a.v, a.g, a.n = p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p.numel()
        stream = torch.cuda.current_stream(p.device).cuda_stream
        _lib.check(h, lib.spmf_adam_step(h, arr, len(self.params), float(self.lr), self.b1,
                                         self.b2, self.eps, self.t,
                                         float(clip_value) if clip_value else 0.0, stream),
                   "spmf_adam_step")


def batch_rows_global(cs, all_reduce):
    """Rows of the batch over ALL row shards.  With a reducer the data parts that
    come out of spmf_finish are global sums, so the batch weight c = B/N, the
    loss divisor and 1/(S*B) must use the global row count, not the shard's."""
    if all_reduce is None:
        return int(cs.n_rows)
    totals = getattr(all_reduce, "totals", None)
    if totals is None:
        raise ValueError("a sharded VI step needs the batch's global row count: the "
                         "all_reduce hook must provide totals(rows, lgamma_sum) "
                         "(spmf_amd.dist.ShardReducer does)")
    return int(totals(cs.n_rows, cs.lgamma_sum)[0])


def elbo_step(model, batch, dataset_rows, sample_size, all_reduce=None, nonfinite=None):
    """One stochastic ELBO evaluation + gradient wrt the surrogate trainables,
    all arithmetic in HIP kernels.  Returns (loss, grads list, n_nonfinite).
    ``dataset_rows`` is the size of the whole dataset (all shards).
    ``nonfinite``: "rule" applies the reference's replacement rule
    (poisson.py:606-616) with its gradient when stored cells have a non-finite
    log-pmf (the default, also across row shards when the reducer can gather the
    shard minima: dist.ShardReducer), "count" only counts them (the caller then
    skips the batch; the default with any other all_reduce hook)."""
    if nonfinite is None:
        nonfinite = "rule" if (all_reduce is None or hasattr(all_reduce, "gather_scalar")) else "count"
    sur = model.surrogate_distribution
    S = int(sample_size)
    noise = sur.draw_noise(S)
    theta, logq = sur.forward_hip(model, S, noise)
    sc, cs = model._batch(batch)
    B = batch_rows_global(cs, all_reduce)
    c = float(B) / float(dataset_rows)
    parts, g, nnf = model.energy_and_grads(batch, theta, all_reduce=all_reduce, prior_weight=c,
                                           nonfinite=nonfinite)
    rows = B
    prior = sum(parts[n] for n in model.var_order)
    energy = parts["x"] + parts["z"] + c * prior           # [S] float64
    loss = -(energy - c * logq).mean() / rows
    grads = sur.backward_hip(model, S, noise, g, 1.0 / (S * rows), c)
    return loss, grads, nnf


class _StepReducer:
    """all_reduce hook of ONE device-resident step over row shards: sums the packed
    accumulators through the reducer's transport and answers the batch totals from values
    the host already holds, so the step reads nothing back (and, with the library's RCCL
    communicator, is a fixed stream-ordered launch sequence a hipGraph can hold)."""

    def __init__(self, red, totals):
        self.red, self._tot = red, totals
        self.overlap_prior = getattr(red, "overlap_prior", True)

    def __call__(self, acc, rows, lgamma_sum):
        self.red.sum_(acc)
        return self._tot

    def totals(self, rows, lgamma_sum):
        return self._tot


def _hierarchy_beside_the_column_pass(model, sur):
    """The side stream + the "rows done" event of the model when the VI step is asked to split its surrogate
    work (vi_step_dev), else None: models with the scale hierarchy (horshoe_plus=True) on the HIP energy path,
    and only with SPMF_VI_OVERLAP=1.  OFF by default: on ONE GPU the split measured 2 % SLOWER than the
    single-stream order (the 122 880-row shard of C3: 0.509 against 0.499 ms per VI step, round 5 -- the
    side stream's kernels did not run beside the column pass, they took turns with it); what it is for is the
    row-sharded step, where the collective leaves the chip idle, and that has never been timed."""
    import os
    if os.environ.get("SPMF_VI_OVERLAP", "0") != "1" or getattr(model, "_custom_codec", None) is not None:
        return None
    if not any(sur.kinds[n] == "invgamma" for n in sur.var_order) or getattr(model, "column_split", 0):
        return None
    st = getattr(model, "_vi_side", None)
    if st is None:
        stream = torch.cuda.Stream(device=sur.device)
        ev_rows = torch.cuda.Event(enable_timing=False)
        ev_rows.record()                     # (creates the underlying hipEvent_t: its handle goes to the library)
        st = model._vi_side = (stream, ev_rows)
    return st


@torch.no_grad()
def vi_step_dev(model, opt, batch, dataset_rows, sample_size, keep=None, seed=None,
                all_reduce=None):
    """One whole VI step with no host read-back: noise, surrogate, energy +
    gradient, loss/skip decision (spmf_vi_gate), chain rule, gated Adam
    (spmf_adam_step_dev).  The launch sequence depends only on the batch
    object, so it can be captured in a hipGraph (StepRunner).

    ``all_reduce``: a dist.ShardReducer when ``batch`` is this rank's row shard of the
    step's batch (poisson.py:60,72 `strategy` is the reference's only hook).
    The data pass runs on the shard, the packed accumulators go through the reducer's ONE
    sum all-reduce, and everything after it -- finish, gate, chain rule, Adam -- runs
    redundantly on the all-reduced values with the batch's GLOBAL row count, so every rank
    takes the same decision and applies the same update.  The global totals come from
    ``ShardReducer.batch_totals`` (reduced once per batch object, on the host); the step
    itself reads nothing back."""
    lib, h = _lib.load(), model._handle()
    sur = model.surrogate_distribution
    S = int(sample_size)
    sc, cs = model._batch(batch)
    hook = None
    B = cs.n_rows
    if all_reduce is not None:
        if not hasattr(all_reduce, "batch_totals"):
            raise ValueError("the device-resident sharded step needs a dist.ShardReducer "
                             "(batch_totals / sum_); other hooks run through elbo_step")
        tot = all_reduce.batch_totals(cs)
        B = int(tot[0])
        hook = _StepReducer(all_reduce, (B, float(tot[1])))
    c = float(B) / float(dataset_rows)
    # seed given (StepRunner): a fixed key + the device step counter, so the launch
    # sequence is replayable; else a fresh key from torch's generator per call
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        state = None
    else:
        state = opt.state
    side = _hierarchy_beside_the_column_pass(model, sur)
    # draw + transform in one launch where the step is launch-bound (a model of up to ~1e6 surrogate elements:
    # the reference harness' shape 0.20 -> 0.15 ms per VI step, S = 20; C1 0.099 -> 0.095); the big models keep
    # the separate sampler, whose waves retire without a workgroup barrier behind the gamma draws' series
    # (the 8-GPU shard of C3, 2.7e6 elements: 0.488 against 0.498 ms; tools/vi_fused_ab.py)
    fused_env = os.environ.get("SPMF_VI_FUSED_SAMPLER")
    n_elem = sum(p.numel() for p in sur.trainable_variables) // 2
    fused = (n_elem <= 1_000_000) if fused_env is None else (fused_env != "0")
    if side is None and fused:
        noise, theta, logq = sur.draw_and_forward(model, S, seed=seed, state=state)     # one launch
        parts, g, nnf = model.energy_and_grads(batch, theta, all_reduce=hook, prior_weight=c)
    elif side is None:
        noise = sur.draw_noise(S, seed=seed, state=state)                               # (A/B: the three launches)
        theta, logq = sur.forward_hip(model, S, noise)
        parts, g, nnf = model.energy_and_grads(batch, theta, all_reduce=hook, prior_weight=c)
    else:
        # The data pass reads v, w, u, s only; the eight variables of the scale hierarchy (the gamma draws
        # with their implicit gradient, half of the transform) enter the prior alone.  So: draw + transform
        # the four, issue the data pass, and draw + transform the hierarchy and run the prior half of the
        # finish on a side stream that starts when the row pass is done -- beside the column pass (and the
        # all-reduce), which is bound by gathers while this work is arithmetic.  Same draws, same numbers:
        # the sampler's counter and the log q slots do not depend on which call covers a variable.
        names_d = frozenset(n for n in sur.var_order if sur.kinds[n] != "invgamma")
        names_h = frozenset(sur.var_order) - names_d
        noise = sur.alloc_noise(S)
        theta = {n: torch.empty(noise[n][0].shape, dtype=torch.float32, device=sur.device) for n in sur.var_order}
        logq_d = torch.empty(S, dtype=torch.float64, device=sur.device)
        logq_h = torch.empty(S, dtype=torch.float64, device=sur.device)
        sur.draw_noise(S, seed=seed, state=state, only=names_d, out=noise)
        sur.forward_hip(model, S, noise, only=names_d, theta=theta, logq=logq_d)

        def hierarchy():
            sur.draw_noise(S, seed=seed, state=state, only=names_h, out=noise)
            sur.forward_hip(model, S, noise, only=names_h, theta=theta, logq=logq_h)
        parts, g, nnf = model.energy_and_grads(batch, theta, all_reduce=hook, prior_weight=c,
                                               beside_columns=(side, hierarchy))
        logq = logq_d + logq_h
    stream = torch.cuda.current_stream(sur.device).cuda_stream
    _lib.check(h, lib.spmf_vi_gate(h, model._last_parts.data_ptr(), logq.data_ptr(),
                                   nnf.data_ptr(), S, c, float(B), opt.state.data_ptr(), stream),
               "spmf_vi_gate")
    if keep is None:
        # chain rule + gated Adam in one pass (the gradient never goes to memory)
        sur.backward_adam_hip(model, S, noise, g, 1.0 / (S * B), c, opt)
        return
    grads = sur.backward_hip(model, S, noise, g, 1.0 / (S * B), c)
    opt.step_dev(grads)
    if keep is not None:
        keep.update(noise=noise, theta=theta, logq=logq, parts=model._last_parts, g=g,
                    grads=grads, nnf=nnf)


class StepRunner:
    """Runs vi_step_dev per batch; the second time a batch object is seen its
    step is captured into a hipGraph (torch.cuda.CUDAGraph: the library's
    launches go to torch's capturing stream) and replayed from then on.  Small
    batches are launch-bound (about 40 launches per step), which is what the
    graph removes; batches above ``graph_max_nnz`` stored entries are GPU-bound
    and run the same device-gated step as plain launches (a replay is no faster
    there).  With ``all_reduce`` (a dist.ShardReducer) the step is this rank's
    share of a row-sharded step; the collective is part of the captured graph
    when the reducer carries the library's RCCL communicator."""

    def __init__(self, model, opt, dataset_rows, sample_size, use_graph=True, max_graphs=64,
                 all_reduce=None, seed=None):
        self.model, self.opt = model, opt
        self.dataset_rows, self.S = dataset_rows, int(sample_size)
        self.all_reduce = all_reduce
        # row shards: the collective is captured only when it is the library's own
        # stream-ordered ncclAllReduce (dist.LibraryComm); torch.distributed's collectives
        # and the host-staged gloo rehearsal run the same device-gated step eagerly
        self.use_graph = bool(use_graph) and (all_reduce is None or
                                              bool(getattr(all_reduce, "graph_safe", False)))
        self.max_graphs = max_graphs
        # batches with more stored entries than this run the SAME device-gated step eagerly: the
        # step is then GPU-bound (launches queue ahead of it) and a replayed hipGraph is no faster than
        # the plain launches, mostly slower (tools/graph_threshold_sweep.py on C3 minibatches,
        # profiles/r04_graph_threshold_sweep.txt: 2.3e6 entries 0.265 replayed / 0.269 eager, 4.5e6
        # 0.324 / 0.316, 1.25e7 0.581 / 0.571 ms); the replay pays for launch-bound steps (C1 / C2 sizes)
        self.graph_max_nnz = 2_500_000
        self.graphs = {}           # key -> (graph, workspace ptr, pinned refs)
        self.seen = {}
        self.pool = None
        self.replays = 0
        self.keep_tensors = False  # tests: keep the captured step's tensors
        self.kept = {}
        # Philox key of this runner's noise; the per-step variation is the device counter.
        # Row shards replicate the surrogate, so every rank must draw the SAME noise: rank
        # 0's key goes to the others through the reducer
        self.seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)) if seed is None else int(seed)
        if all_reduce is not None and hasattr(all_reduce, "share_int"):
            self.seed = all_reduce.share_int(self.seed)

    def _key(self, batch):
        sc, cs = self.model._batch(batch)
        return (id(cs), self.S, float(self.dataset_rows)), cs

    def step(self, batch):
        if not self.use_graph:
            vi_step_dev(self.model, self.opt, batch, self.dataset_rows, self.S, seed=self.seed,
                        all_reduce=self.all_reduce)
            return
        key, cs = self._key(batch)
        if cs.nnz > self.graph_max_nnz:
            vi_step_dev(self.model, self.opt, batch, self.dataset_rows, self.S, seed=self.seed,
                        all_reduce=self.all_reduce)
            return
        hit = self.graphs.get(key)
        ws = self.model._ws.data_ptr() if self.model._ws is not None else 0
        if hit is not None and hit[1] == ws and hit[2] is cs:
            hit[0].replay()
            self.replays += 1
            return
        if hit is not None:
            del self.graphs[key]
        if self.seen.get(key) is not cs or ws == 0:
            # first sight: eager (also the warm-up that sizes the workspace)
            vi_step_dev(self.model, self.opt, batch, self.dataset_rows, self.S, seed=self.seed,
                        all_reduce=self.all_reduce)
            if len(self.seen) > 4 * self.max_graphs:
                self.seen.clear()
            self.seen[key] = cs
            return
        if len(self.graphs) >= self.max_graphs:
            self.graphs.pop(next(iter(self.graphs)))
        if self.all_reduce is not None and hasattr(self.all_reduce, "batch_totals"):
            # the batch's global totals must already be known: reducing them is a collective plus a host
            # read, neither of which may happen inside the capture (ShardReducer.batch_totals refuses it)
            self.all_reduce.batch_totals(cs)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        graph = torch.cuda.CUDAGraph()
        keep = {} if self.keep_tensors else None
        torch.cuda.synchronize(self.model.device)
        # no garbage collection while capturing: a collected cycle that owns device
        # memory or another graph would free it inside the capture (process abort)
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph, pool=self.pool):
                vi_step_dev(self.model, self.opt, batch, self.dataset_rows, self.S, keep=keep,
                            seed=self.seed, all_reduce=self.all_reduce)
        finally:
            if gc_was_on:
                gc.enable()
        if keep is not None:
            self.kept[key] = keep
        self.graphs[key] = (graph, self.model._ws.data_ptr(), cs)
        graph.replay()             # capture does not execute: run the step once
        self.replays += 1


def _clip(grads, clip_value):
    if clip_value is None:
        return grads
    return [g.clamp(-clip_value, clip_value) for g in grads]


class PlateauController:
    """The epoch-level control of the legacy driver, as its recorded stdout shows it
    (notebooks/factorizing_random_noise.ipynb:122-420: "Saved a checkpoint", "We are
    in a loss plateau learning rate: ...", "Restoring from a checkpoint", "We have
    reset 25 times so quitting"; bayesianquilts itself is out of tree
    [UNVERIFIED-3P]).  Host logic only, no device work:

      update(epoch_loss) ->
        "improved"   new best: the caller checkpoints the trainables
        "converged"  the improvement fell under abs_tol / rel_tol: stop
        "plateau"    no improvement: lr *= lr_decay_factor, the caller restores
                     the best checkpoint
        "quit"       like "plateau", and max_decay_steps resets have happened: stop
    """

    def __init__(self, learning_rate, rel_tol=1e-6, abs_tol=1e-10, max_decay_steps=25,
                 lr_decay_factor=0.99):
        self.lr = float(learning_rate)
        self.rel_tol, self.abs_tol = rel_tol, abs_tol
        self.max_decay_steps, self.lr_decay_factor = int(max_decay_steps), lr_decay_factor
        self.best = math.inf
        self.decays = 0

    def update(self, ep_loss):
        if ep_loss < self.best:
            gain = self.best - ep_loss
            stop = gain < self.abs_tol or (math.isfinite(self.best) and
                                           gain / abs(self.best) < self.rel_tol)
            self.best = ep_loss
            return "converged" if stop else "improved"
        self.decays += 1
        self.lr *= self.lr_decay_factor
        return "quit" if self.decays >= self.max_decay_steps else "plateau"


def _warn_saturated(model, events, verbose):
    """The log_transform decoder evaluates exp(min(y, 70)) (csrc/common.h kYSat): a step with
    exponents in (70, 709) trains on values and gradients that differ from the fp64 reference
    (poisson.py:52-53) instead of being skipped.  Say so once per epoch while it happens, and
    keep the running count on the model (``model.saturated_events``)."""
    model.saturated_events = getattr(model, "saturated_events", 0.0) + float(events)
    if events and verbose:
        print(f"Decoder saturated: exp(y) evaluated at min(y, 70) in {int(events)} workgroup "
              "events this epoch -- values and gradients of those steps differ from the fp64 "
              "reference (poisson.py:52-53) until the exponents come down")


def fit(model, batched_data_factory, dataset_size, batch_size=None, sample_size=8,
        sample_batches=1, num_steps=100, num_epochs=None, rel_tol=1e-6, abs_tol=1e-10,
        learning_rate=0.01, clip_value=10.0, max_decay_steps=25, lr_decay_factor=0.99,
        check_every=1, set_expectations=True, all_reduce=None, verbose=True, **kwargs):
    """Counterpart of BayesianModel.fit (tests/spmf_test.py:35-43 kwargs).
    One epoch = one pass over ``batched_data_factory()``; stops on rel_tol /
    abs_tol of the epoch-mean loss; on a plateau decays the learning rate and
    restores the best trainables (behaviour evidenced by
    notebooks/factorizing_random_noise.ipynb:122-420)."""
    sur = model.surrogate_distribution
    opt = AdamHIP(model, sur.trainable_variables, learning_rate)
    epochs = num_epochs if num_epochs is not None else num_steps
    losses, best_state = [], None
    ctl = PlateauController(learning_rate, rel_tol, abs_tol, max_decay_steps, lr_decay_factor)
    # (custom encoder/decoder callables run torch autograd inside the step: eager loop; so does
    #  a foreign all_reduce hook, which only knows how to sum the accumulators)
    device_loop = getattr(model, "_custom_codec", None) is None and (
        all_reduce is None or hasattr(all_reduce, "batch_totals"))
    # (a deterministic model's replicas apply bit-identical updates on the device loop: the guard
    #  re-broadcast is not needed THERE; the eager loop a skipped batch falls back to runs the
    #  non-finite rule, whose accumulator patch uses atomics and is outside that guarantee, so the
    #  guard comes back with it -- see the fallback below)
    sync_default = int(kwargs.get("sync_every", 200))
    sync_every = 0 if getattr(model, "deterministic", False) else sync_default
    since_sync = 0
    if device_loop:
        opt.init_state(clip_value)
        runner = StepRunner(model, opt, dataset_size, sample_size,
                            use_graph=kwargs.get("use_graph", True), all_reduce=all_reduce)
    for ep in range(epochs):
        tot, nb, ep_sat = 0.0, 0, 0.0
        if device_loop:
            # no host read-back inside the epoch: the loss sum, the applied and
            # the skipped step counts live in the optimiser's device state
            opt.reset_epoch_counters()
            for batch in iter(batched_data_factory()):
                runner.step(batch)
            st = opt.read_state()
            tot, nb, skipped = st[10], int(st[11]), int(st[12])
            ep_sat = st[14]
            # the peer-pointer collective bounds every wait for a peer (20 s) and then goes on with whatever it
            # has: an epoch in which that happened must not pass for a trained one
            peer = getattr(all_reduce, "comm", None)
            if peer is not None and hasattr(peer, "status"):
                _, gave_up = peer.status()
                if gave_up:
                    raise RuntimeError(f"the step's collective gave up waiting for a peer in its call {gave_up} "
                                       "(spmf_p2p_status): a rank is gone or far behind; the replicas are no "
                                       "longer in step")
            if verbose and skipped:
                print(f"Batch loss NaN, skipping ({skipped} batches)")
            # row shards: the replicas took the same decisions on the same all-reduced
            # values; the guard re-broadcast of the eager loop, at epoch boundaries
            since_sync += nb
            if sync_every and since_sync >= sync_every and hasattr(all_reduce, "sync_replicas"):
                all_reduce.sync_replicas(list(sur.trainable_variables) + opt.m + opt.v)
                since_sync = 0
            if skipped:
                # the device-gated loop cannot apply the replacement rule (it needs a
                # host decision); from here on run the eager loop, which trains THROUGH
                # non-finite cells like the reference (poisson.py:606-616)
                device_loop = False
                sync_every = sync_default         # (deterministic models: the eager / rule path is not bit-reproducible)
                opt.t = int(st[7])
                if nb == 0:
                    continue                      # nothing applied yet: next epoch, eagerly
        else:
            for batch in iter(batched_data_factory()):
                loss, grads, nnf = elbo_step(model, batch, dataset_size, sample_size, all_reduce)
                lv = float(loss)
                # non-finite cells were handled by the rule (single shard, or row shards
                # behind a ShardReducer); behind any other hook they are only counted and
                # the batch is skipped
                ruled = all_reduce is None or hasattr(all_reduce, "gather_scalar")
                if not math.isfinite(lv) or (not ruled and float(nnf.sum()) > 0):
                    if verbose:
                        print("Batch loss NaN, skipping")
                    continue
                opt.step(grads, clip_value)
                tot += lv
                nb += 1
                ep_sat += float(model.last_saturated.sum())
                # row shards: keep the replicated trainables and Adam moments bit-identical
                # (ShardReducer.sync_replicas explains why they can drift)
                if sync_every and hasattr(all_reduce, "sync_replicas") and opt.t % sync_every == 0:
                    all_reduce.sync_replicas(list(sur.trainable_variables) + opt.m + opt.v)
        _warn_saturated(model, ep_sat, verbose)
        if nb == 0:
            # every batch of the epoch was skipped (non-finite loss): returning
            # quietly would look like a converged fit
            raise FloatingPointError(
                f"epoch {ep}: all batches were skipped (non-finite loss); nothing was "
                "trained -- check the column scales / initial values for overflow")
        ep_loss = tot / nb
        losses.append(ep_loss)
        if verbose and ep % check_every == 0:
            print(f"Epoch: {ep} average-batch loss: {ep_loss}")
        action = ctl.update(ep_loss)
        if action == "converged":
            break
        if action == "improved":
            # (one multi-tensor copy into buffers kept for the whole fit: two dozen clones per
            #  improving epoch were a tenth of a full-batch epoch on C3)
            cur = [p.detach() for p in sur.trainable_variables]
            if best_state is None:
                best_state = [p.clone() for p in cur]
            else:
                torch._foreach_copy_(best_state, cur)
            continue
        opt.set_lr(ctl.lr)
        if verbose:
            print(f"We are in a loss plateau learning rate: {opt.lr}")
        if best_state is not None:
            with torch.no_grad():
                for p, b in zip(sur.trainable_variables, best_state):
                    p.copy_(b)
            if verbose:
                print("Restoring from a checkpoint")
        if action == "quit":
            if verbose:
                print(f"We have reset {ctl.decays} times so quitting")
            break
    if set_expectations:
        model.set_calibration_expectations()
    return losses


def batch_rows(model, b):
    """Rows of one batch object WITHOUT building its device layout: a dense
    [B,D] array, a sparse matrix, a SparseCounts, or a panel range
    ``{'counts': sc, 'panels': (p0, p1)}`` of a resident SparseCounts (whose
    own n_rows is the whole shard, not the batch)."""
    x = b[model.count_key] if isinstance(b, dict) else b
    pr = b.get("panels") if isinstance(b, dict) else None
    if pr is not None and hasattr(x, "panel_rows"):
        p0, p1 = pr
        p1 = x.n_panels if p1 is None else min(int(p1), x.n_panels)
        return max(0, min(p1 * x.panel_rows, x.n_rows) - int(p0) * x.panel_rows)
    if isinstance(x, (tuple, list)) and len(x) == 4:      # (indptr, indices, data, shape)
        return int(x[3][0])
    return int(x.n_rows) if hasattr(x, "n_rows") else int(x.shape[0])


def calibrate_advi(model, num_steps=100, num_epochs=None, learning_rate=0.1, abs_tol=1e-10,
                   rel_tol=1e-8, clip_value=5.0, max_decay_steps=25, lr_decay_factor=0.99,
                   check_every=1, set_expectations=True, sample_size=4, data=None, **kwargs):
    """Legacy driver entry (bin/factorize_csv.py:121-124): iterates
    ``model.data`` (the batched dataset given to PoissonMatrixFactorization)."""
    data = data if data is not None else getattr(model, "data", None)
    if data is None:
        raise ValueError("calibrate_advi needs the dataset the model was built with")
    factory = data if callable(data) else (lambda: data)
    n = sum(batch_rows(model, b) for b in iter(factory()))
    return fit(model, factory, dataset_size=n, sample_size=sample_size,
               num_steps=num_steps, num_epochs=num_epochs, rel_tol=rel_tol, abs_tol=abs_tol,
               learning_rate=learning_rate, clip_value=clip_value,
               max_decay_steps=max_decay_steps, lr_decay_factor=lr_decay_factor,
               check_every=check_every, set_expectations=set_expectations, **kwargs)


def save_model(model, filename):
    """``factor.save(filename)`` (bin/factorize_csv.py:139): pickle of the
    ordered surrogate variables + constructor state, restorable with
    ``reconstitute`` (poisson.py:711-717)."""
    import pickle
    eta = model.eta_i
    if isinstance(eta, torch.Tensor):
        eta = eta.detach().cpu().numpy()
    state = {
        "surrogate_vars": [p.detach().cpu().numpy()
                           for p in model.surrogate_distribution.trainable_variables],
        "var_list": list(model.var_list),
        "latent_dim": model.latent_dim, "feature_dim": model.feature_dim,
        "u_tau_scale": model.u_tau_scale, "s_tau_scale": model.s_tau_scale,
        "symmetry_breaking_decay": model.symmetry_breaking_decay,
        "scale_columns": model.scale_columns, "scale_rows": model.scale_rows,
        "log_transform": model.log_transform, "eta_i": eta,
        "xi_u_global": float(model.xi_u_global), "count_key": model.count_key,
    }
    with open(filename, "wb") as f:
        pickle.dump(state, f)
    return state
