"""Energy of a model built with USER-SUPPLIED ``encoder_function`` /
``decoder_function`` callables (mederrata_spmf/poisson.py:94-97 lets a caller
replace g and f).

The HIP kernels know the two built-in pairs only (x/eta <-> y*eta and
log(x/eta+1) <-> exp(y*eta)-1, poisson.py:34-54); an arbitrary Python callable
cannot be compiled into them.  SURVEY 8b therefore asks for a non-kernel path for
this one constructor option: the reference's own dense formulation
(poisson.py:156-184,582-621) written with torch tensor ops ON THE MODEL'S DEVICE,
gradients by autograd through the user's callables.  It is not a fallback of the
hot path -- nothing routes here unless the caller passed a callable -- and the
model announces it when it is constructed.  Cost: O(S*B*D) dense, like the
reference.
"""
from __future__ import annotations

import math

import torch

HALF_LOG_2_OVER_PI = 0.5 * math.log(2.0 / math.pi)
LGAMMA_HALF = math.lgamma(0.5)


def _halfnormal(y, scale):
    return HALF_LOG_2_OVER_PI - torch.log(scale) - 0.5 * (y / scale) ** 2


def _inv_gamma(x, conc, scale):
    return conc * torch.log(scale) - math.lgamma(conc) - (conc + 1.0) * torch.log(x) - scale / x


def _sqrt_inv_gamma(y, conc, scale):
    return _inv_gamma(y * y, conc, scale) + math.log(2.0) + torch.log(y)


def prior_parts(model, p):
    """The prior of create_distributions (poisson.py:228-377, horseshoe-plus)."""
    if not model.horseshoe_plus:
        raise NotImplementedError("custom encoder/decoder callables with horshoe_plus=False")
    dt = p["u"].dtype
    K = model.latent_dim
    decay = (model.symmetry_breaking_decay ** torch.arange(K, dtype=dt, device=p["u"].device))[None, :]
    one = torch.ones((), dtype=dt, device=p["u"].device)
    sm = lambda t: t.sum((-1, -2))
    out = {"v": sm(_halfnormal(p["v"], 0.1 * one)), "w": sm(_halfnormal(p["w"], one)),
           "u": sm(_halfnormal(p["u"], p["u_eta"] * p["u_tau"] * decay)),
           "s": sm(_halfnormal(p["s"], p["s_eta"] * p["s_tau"]))}
    for n, beta in (("u_eta", 1.0), ("u_tau", 1.0 / model.u_tau_scale ** 2),
                    ("s_eta", 1.0), ("s_tau", 1.0 / model.s_tau_scale ** 2)):
        out[n] = sm(_sqrt_inv_gamma(p[n], 0.5, 1.0 / p[n + "_a"]))
        out[n + "_a"] = sm(_inv_gamma(p[n + "_a"], 0.5, beta * one))
    return out


def energy_and_grads(model, x_dense, params, prior_weight=1.0):
    """parts (name -> [S] float64) and d(x + z + prior_weight*prior)/d(param)
    (float32, the shapes of ``params``) for a dense batch [B,D] on the device."""
    dev = model.device
    dt = torch.float64
    enc, dec = model._custom_codec
    p = {k: torch.as_tensor(v, device=dev).to(dt).clone().requires_grad_(True)
         for k, v in params.items()}
    if p["u"].dim() == 2:
        p = {k: v.unsqueeze(0) for k, v in p.items()}
    x = x_dense.to(dev, dt)
    eta = model._eta_device().to(dt)
    parts = prior_parts(model, p)
    s = p["s"]
    weights = s / s.sum(-2, keepdim=True)
    A = weights[..., 0, :].unsqueeze(-1) * p["u"]                      # poisson.py:652-666
    phi = eta * weights[..., 1, :].unsqueeze(-2) * p["w"]              # :680-701
    theta = torch.matmul(enc(x), A)                                    # :640-643
    if model.scale_rows:
        theta = theta * (x.sum(-1, keepdim=True) / float(model.xi_u_global))
    rate = dec(torch.matmul(theta, p["v"])) + phi                      # :174-177
    bad = (x > 0) & ~((rate > 0) & torch.isfinite(rate))
    safe = torch.where(bad, torch.ones_like(rate), rate)
    ll = torch.xlogy(x, safe) - torch.lgamma(x + 1.0) - safe           # tfd.Poisson.log_prob
    good = ~bad & torch.isfinite(ll)
    n_bad = (~good).sum((-1, -2)).to(dt)
    # replacement rule (:606-616); the identity when every cell is finite
    fin = torch.where(good, ll, torch.zeros_like(ll))
    mval = fin.min() - 10.0
    parts["x"] = torch.where(good, torch.clamp(ll, max=0.0), torch.zeros_like(ll)).sum((-1, -2)) + n_bad * mval
    parts["z"] = (HALF_LOG_2_OVER_PI - 0.5 * theta ** 2).sum((-1, -2))
    tot = parts["x"].sum() + parts["z"].sum() + prior_weight * sum(
        v.sum() for k, v in parts.items() if k not in ("x", "z"))
    names = list(p)
    g = torch.autograd.grad(tot, [p[n] for n in names], allow_unused=True)
    grads = {n: (gi if gi is not None else torch.zeros_like(p[n])).to(torch.float32)
             for n, gi in zip(names, g)}
    return {k: v.detach() for k, v in parts.items()}, grads, n_bad.detach()
