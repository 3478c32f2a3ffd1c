"""Sparse-exact CPU restatement (numpy + scipy.sparse) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/spmf_oracle.py header): pinned only against the
dense fp64 oracle, scipy densities and finite differences.

For the linear decoder (log_transform=False) the dense [S,B,D] likelihood of
mederrata_spmf/poisson.py:156-184 collapses exactly to work over the stored
entries plus closed-form sums over the implicit zeros:

    sum_{b,d} ll = sum_nnz [x log r - lgamma(x+1)] - sum_all r
    sum_all r    = <sum_b z_b, sum_d eta_d V_d> + B * sum_d phi_d

and the analytic gradients (derived from
poisson.py:582-701) are

    dE/dz_b  = sum_{d in nnz(b)} (x/r) eta_d V_d - V eta - z_b
    dE/dV_d  = eta_d ( sum_{b in nnz(d)} (x/r) z_b - sum_b z_b )
    dE/dphi_d= sum_{b in nnz(d)} x/r - B
    dE/dA_d  = sum_{b in nnz(d)} g(x_bd) xi_b dE/dz_b

This is the algorithm the HIP kernels implement; here it is written with
scipy.sparse so it can be (a) property-tested against the dense oracle and
(b) timed as bench.py's ``cpu_baseline`` (kind "port").

The non-finite rule (poisson.py:606-616) is the identity whenever every cell
is finite (Poisson log-pmf <= 0 and min-10 < min), which this fast path
assumes; it reports the number of non-finite stored cells so callers can
detect when the assumption fails.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
from scipy.special import gammaln

HALF_LOG_2_OVER_PI = 0.5 * math.log(2.0 / math.pi)


def data_term(X: sp.csr_matrix, eta, xi_global, scale_rows, u, v, w, s,
              dtype=np.float64):
    """Energy parts 'x' and 'z' and their gradients wrt (u, v, w, s) for ONE
    sample.  u [D,K], v [K,D], w [1,D], s [2,D]; eta [D] (or scalar)."""
    X = X.tocsr().astype(dtype)
    B, D = X.shape
    K = u.shape[1]
    eta = np.broadcast_to(np.asarray(eta, dtype=dtype).reshape(-1), (D,)) \
        if np.ndim(eta) else np.full(D, eta, dtype=dtype)
    u = u.astype(dtype); v = v.astype(dtype)
    w = w.astype(dtype).reshape(D); s = s.astype(dtype)
    T = s[0] + s[1]
    w1, w2 = s[0] / T, s[1] / T
    A = w1[:, None] * u                       # poisson.py:652-666
    phi = eta * w2 * w                        # poisson.py:680-701
    Ap = A / eta[:, None]                     # eta folded: g(x) A = x (A/eta)
    Vp = (v * eta[None, :]).T                 # [D,K]; f(y)=y*eta folded
    rowsum = np.asarray(X.sum(1)).reshape(B)
    xi = rowsum / dtype(xi_global) if scale_rows else np.ones(B, dtype)
    z = (X @ Ap) * xi[:, None]                # poisson.py:640-649
    # SDDMM on the stored pattern
    indptr, indices, xv = X.indptr, X.indices, X.data
    rows = np.repeat(np.arange(B), np.diff(indptr))
    r = np.einsum("nk,nk->n", z[rows], Vp[indices]) + phi[indices]
    with np.errstate(divide="ignore", invalid="ignore"):
        ll_nnz = xv * np.log(r) - gammaln(xv + 1.0)
    n_nonfinite = int((~np.isfinite(ll_nnz)).sum())
    zsum = z.sum(0)
    veta = Vp.sum(0)
    sum_r = zsum @ veta + B * phi.sum()
    part_x = ll_nnz.sum() - sum_r
    part_z = B * K * HALF_LOG_2_OVER_PI - 0.5 * (z * z).sum()
    # gradients
    c = xv / r
    C = sp.csr_matrix((c, indices, indptr), shape=(B, D))
    gz = C @ Vp - veta[None, :] - z
    gVp = C.T @ z - zsum[None, :]             # [D,K]
    gphi = np.asarray(C.sum(0)).reshape(D) - B
    gAp = X.T @ (gz * xi[:, None])            # [D,K]
    gA = gAp / eta[:, None]
    gv = (gVp * eta[:, None]).T               # [K,D]
    gu = w1[:, None] * gA
    gw = (eta * w2 * gphi)[None, :]
    GA = (u * gA).sum(1)
    Gphi = eta * w * gphi
    gs = np.stack([(GA - Gphi) * s[1] / T ** 2, (Gphi - GA) * s[0] / T ** 2])
    return {"x": part_x, "z": part_z, "n_nonfinite": n_nonfinite,
            "grads": {"u": gu, "v": gv, "w": gw, "s": gs},
            "z_rows": z, "gz_rows": gz}


def prior_term(p, u_tau_scale, s_tau_scale, decay):
    """Horseshoe-plus prior parts (poisson.py:228-377) and analytic gradients
    wrt all 12 variables, ONE sample.  decay: [K]."""
    f = np.float64
    out, g = {}, {k: np.zeros_like(np.asarray(val, dtype=f)) for k, val in p.items()}
    P = {k: np.asarray(val, dtype=f) for k, val in p.items()}
    c0 = HALF_LOG_2_OVER_PI
    lgh = math.lgamma(0.5)

    def halfnormal(y, sig):
        lp = c0 - np.log(sig) - 0.5 * (y / sig) ** 2
        return lp, -y / sig ** 2, -1.0 / sig + y ** 2 / sig ** 3

    def sqrt_ig_half(y, a):     # SqrtInvGamma(1/2, scale=1/a) at y
        lp = -0.5 * np.log(a) - lgh - 2.0 * np.log(y) - 1.0 / (a * y * y) + math.log(2.0)
        return lp, -2.0 / y + 2.0 / (a * y ** 3), -0.5 / a + 1.0 / (a * a * y * y)

    def ig_half(a, beta):       # InvGamma(1/2, beta) at a
        lp = 0.5 * math.log(beta) - lgh - 1.5 * np.log(a) - beta / a
        return lp, -1.5 / a + beta / a ** 2

    lp, gy, _ = halfnormal(P["v"], 0.1); out["v"] = lp.sum(); g["v"] += gy
    lp, gy, _ = halfnormal(P["w"], 1.0); out["w"] = lp.sum(); g["w"] += gy
    sig = P["u_eta"] * P["u_tau"] * decay[None, :]
    lp, gy, gs_ = halfnormal(P["u"], sig); out["u"] = lp.sum(); g["u"] += gy
    g["u_eta"] += gs_ * P["u_tau"] * decay[None, :]
    g["u_tau"] += (gs_ * P["u_eta"] * decay[None, :]).sum(0, keepdims=True)
    sig = P["s_eta"] * P["s_tau"]
    lp, gy, gs_ = halfnormal(P["s"], sig); out["s"] = lp.sum(); g["s"] += gy
    g["s_eta"] += gs_ * P["s_tau"]
    g["s_tau"] += (gs_ * P["s_eta"]).sum(0, keepdims=True)
    for nm, beta in (("u_eta", 1.0), ("u_tau", 1.0 / u_tau_scale ** 2),
                     ("s_eta", 1.0), ("s_tau", 1.0 / s_tau_scale ** 2)):
        lp, gy, ga = sqrt_ig_half(P[nm], P[nm + "_a"])
        out[nm] = lp.sum(); g[nm] += gy; g[nm + "_a"] += ga
        lp, ga = ig_half(P[nm + "_a"], beta)
        out[nm + "_a"] = lp.sum(); g[nm + "_a"] += ga
    return out, g


# --------------------------------------------------------------------------
# Packed-accumulator view (mirrors the HIP data pass / finish split so the
# row-sharded all-reduce path can be tested on CPU with gloo).
# Layout (include/spmf_hip.h, spmf_data_pass):
#   acc = [ gA'(D*KP) | gV'(D*KP) | gphi(D) | tail ],  KP = K padded to 4,8,..
#   tail = (hi,lo) float pairs of [sum x log r, sum z^2, nonfinite, dense sum,
#                                  saturated cells, 0, zsum[KP]]
# --------------------------------------------------------------------------
TAIL_HEAD = 6     # scalars in front of zsum (spmf_amd/csrc/common.h kDaccHead)


def padded_k(K):
    kp = 4
    while kp < K:
        kp <<= 1
    return kp


def shard_accumulators(X, eta, xi_global, scale_rows, u, v, w, s):
    """fp32 packed accumulators of ONE row shard, one sample."""
    X = X.tocsr().astype(np.float64)
    B, D = X.shape
    K = u.shape[1]
    KP = padded_k(K)
    eta = np.broadcast_to(np.asarray(eta, dtype=np.float64).reshape(-1), (D,))
    T = s[0] + s[1]
    Ap = (s[0] / T)[:, None] * u / eta[:, None]
    Vp = (v * eta[None, :]).T
    phi = eta * (s[1] / T) * w.reshape(D)
    rowsum = np.asarray(X.sum(1)).reshape(B)
    xi = rowsum / xi_global if scale_rows else np.ones(B)
    z = (X @ Ap) * xi[:, None]
    rows = np.repeat(np.arange(B), np.diff(X.indptr))
    r = np.einsum("nk,nk->n", z[rows], Vp[X.indices]) + phi[X.indices]
    c = X.data / r
    C = sp.csr_matrix((c, X.indices, X.indptr), shape=(B, D))
    veta = Vp.sum(0)
    gz = C @ Vp - veta[None, :] - z
    gVp = np.zeros((D, KP)); gVp[:, :K] = C.T @ z
    gAp = np.zeros((D, KP)); gAp[:, :K] = X.T @ (gz * xi[:, None])
    gphi = np.asarray(C.sum(0)).reshape(D)
    zsum = np.zeros(KP); zsum[:K] = z.sum(0)
    scal = np.concatenate([[(X.data * np.log(r)).sum(), (z * z).sum(), 0.0, 0.0, 0.0, 0.0], zsum])
    hi = scal.astype(np.float32)
    lo = (scal - hi.astype(np.float64)).astype(np.float32)
    tail = np.stack([hi, lo], 1).reshape(-1)
    return np.concatenate([gAp.reshape(-1), gVp.reshape(-1), gphi, tail]).astype(np.float32)


def finish_from_acc(acc, B_global, lgamma_sum, eta, u, v, w, s):
    """Numpy restatement of the finish kernel's data-term chain: packed
    accumulators (after the all-reduce) -> parts x, z and d/d(u,v,w,s)."""
    D, K = u.shape
    KP = padded_k(K)
    eta = np.broadcast_to(np.asarray(eta, dtype=np.float64).reshape(-1), (D,))
    acc = acc.astype(np.float64)
    gAp = acc[:D * KP].reshape(D, KP)[:, :K]
    gVp = acc[D * KP:2 * D * KP].reshape(D, KP)[:, :K]
    gphi_acc = acc[2 * D * KP:2 * D * KP + D]
    tail = acc[2 * D * KP + D:].reshape(-1, 2).sum(1)
    llx, zsq, zsum = tail[0], tail[1], tail[TAIL_HEAD:TAIL_HEAD + K]
    T = s[0] + s[1]
    w1, w2 = s[0] / T, s[1] / T
    Vp = (v * eta[None, :]).T
    phi = eta * w2 * w.reshape(D)
    part_x = llx - lgamma_sum - (zsum @ Vp.sum(0) + B_global * phi.sum())
    part_z = B_global * K * HALF_LOG_2_OVER_PI - 0.5 * zsq
    gA = gAp / eta[:, None]
    gu = w1[:, None] * gA
    gv = ((gVp - zsum[None, :]) * eta[:, None]).T
    dphi = gphi_acc - B_global
    gw = (eta * w2 * dphi)[None, :]
    GA, Gphi = (u * gA).sum(1), eta * w.reshape(D) * dphi
    gs = np.stack([(GA - Gphi) * s[1] / T ** 2, (Gphi - GA) * s[0] / T ** 2])
    return {"x": part_x, "z": part_z, "grads": {"u": gu, "v": gv, "w": gw, "s": gs}}
