"""Plain torch restatement of the VI step around the hot path -- TEST
INFRASTRUCTURE ONLY (moved out of spmf_amd/vi.py: the product's step runs in
HIP kernels).  Sampling, log q and the chain rule to the trainables are torch
autograd ops here; the energy still comes from the model under test, so the
tests that use this isolate the surrogate / optimiser kernels.
"""
import math

import torch

from spmf_amd._lib import VAR_ORDER

_sp = torch.nn.functional.softplus


class GammaReparam(torch.autograd.Function):
    """g ~ Gamma(a, 1) with d g/d a from torch._standard_gamma_grad (implicit
    reparameterisation)."""

    @staticmethod
    def forward(ctx, g, a):
        ctx.save_for_backward(g, a)
        return g

    @staticmethod
    def backward(ctx, grad):
        g, a = ctx.saved_tensors
        return None, grad * torch._standard_gamma_grad(a.contiguous(), g.contiguous())


def rsample(sur, S, generator=None):
    """-> (theta: name -> [S,*shape] with autograd graph, logq [S]) for a
    spmf_amd.vi.Surrogate (poisson.py:403-569 as the build defines it)."""
    theta, logq = {}, 0.0
    for n in sur.var_order:
        t0, t1 = sur.params_of(n)
        shape = (S,) + tuple(t0.shape)
        if sur.kinds[n] in ("normal", "normal_identity"):
            sigma = _sp(t1)
            eps = torch.randn(shape, device=sur.device, dtype=torch.float32, generator=generator)
            y = t0 + sigma * eps
            lq = -0.5 * eps ** 2 - torch.log(sigma) - 0.5 * math.log(2 * math.pi)
        else:
            a, b = _sp(t0), _sp(t1)
            g = torch._standard_gamma(a.expand(shape).contiguous())
            g = GammaReparam.apply(g.detach(), a.expand(shape))
            g = g.clamp_min(1e-30)
            y = b / g
            lq = (a * torch.log(b) - torch.lgamma(a) - (a + 1.0) * torch.log(y) - b / y)
        if sur.kinds[n] == "normal_identity":
            th = y                                          # no Jacobian
        elif n in sur.ident_mask:                           # per-element bijector
            im = sur.ident_mask[n].bool()
            th = torch.where(im, y, _sp(y))
            lq = lq - torch.where(im, torch.zeros_like(y), torch.nn.functional.logsigmoid(y))
        else:
            th = _sp(y)
            lq = lq - torch.nn.functional.logsigmoid(y)
        theta[n] = th
        logq = logq + lq.sum((-1, -2))
    return theta, logq


def elbo_step_reference(model, batch, dataset_rows, sample_size, all_reduce=None, generator=None):
    """torch-autograd restatement of spmf_amd.vi.elbo_step.  Returns (loss,
    grads list, n_nonfinite) -- no optimiser update."""
    sur = model.surrogate_distribution
    theta, logq = rsample(sur, sample_size, generator)
    det = {k: v.detach() for k, v in theta.items()}
    sc, cs = model._batch(batch)
    B = cs.n_rows
    c = float(B) / float(dataset_rows)
    parts, g, nnf = model.energy_and_grads(batch, det, all_reduce=all_reduce, prior_weight=c)
    prior = sum(parts[n] for n in sur.var_order)
    energy = parts["x"] + parts["z"] + c * prior           # [S] float64
    loss = -(energy - c * logq.detach().double()).mean() / B
    lin = sum((g[n] * theta[n]).sum() for n in sur.var_order)
    sur_loss = -(lin - c * logq.sum()) / (sample_size * B)
    grads = torch.autograd.grad(sur_loss, sur.trainable_variables)
    return loss, list(grads), nnf


class Adam:
    """tf.keras-style Adam on a list of tensors (beta1 .9, beta2 .999, eps 1e-7)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.params, self.lr, self.b1, self.b2, self.eps = params, lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        c1 = 1.0 - self.b1 ** self.t
        c2 = 1.0 - self.b2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.addcdiv_(m / c1, (v / c2).sqrt_().add_(self.eps), value=-self.lr)
