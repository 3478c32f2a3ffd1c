"""Row-chunked driver of the dense fp64 oracle -- TEST INFRASTRUCTURE ONLY.

oracle/spmf_oracle.py evaluates the reference's dense [S,B,D] likelihood
(mederrata_spmf/poisson.py:156-184) in one piece, which does not fit for a
20 000 x 30 000 slice of the C4 / C5 workloads.  The data terms are sums over
rows (poisson.py:604,617-618), so this helper feeds the SAME oracle functions
one row chunk at a time and adds the parts and the autograd gradients up.  The
non-finite replacement rule (:606-616) is the identity when every cell is finite,
which is asserted per chunk.
"""
import numpy as np
import torch

from oracle import spmf_oracle as O


def data_term(cfg, X_csr, params, chunk=1024, scales=False):
    """'x', 'z' and d(x+z)/d(u,v,w,s) for ONE draw.  ``params``: name ->
    [1,*shape] float64 arrays (fp32-exact values); X_csr: scipy CSR."""
    names = ("u", "v", "w", "s")
    p = {k: torch.as_tensor(np.asarray(params[k], dtype=np.float64)).clone().requires_grad_(True)
         for k in names}
    B = X_csr.shape[0]
    tot_x, tot_z = 0.0, 0.0
    grads = {k: torch.zeros_like(p[k]) for k in names}
    # per additive piece (stored-cell part, minus-rate part, z prior): its gradient summed over
    # the chunks; sum over pieces of |.| is oracle.energy_grad_scales' entry-wise yardstick
    piece_g = [{k: torch.zeros_like(p[k]) for k in names} for _ in range(3)] if scales else None
    for r0 in range(0, B, chunk):
        x = torch.as_tensor(X_csr[r0:r0 + chunk].toarray().astype(np.float64))
        ll = O.log_likelihood_components(cfg, x, p["s"], p["u"], p["v"], p["w"])["log_likelihood"]
        assert bool(torch.isfinite(ll).all()), "non-finite cell: the rule would not be the identity"
        theta = O.encode(cfg, x, p["u"], p["s"])
        px = ll.sum()
        pz = (O.HALF_LOG_2_OVER_PI - 0.5 * theta ** 2).sum()
        if scales:
            pieces = O._data_pieces(cfg, x, p)
            for i, piece in enumerate(pieces):
                gi = torch.autograd.grad(piece, [p[k] for k in names], retain_graph=True,
                                         allow_unused=True)     # the z prior does not see v, w
                for k, gk in zip(names, gi):
                    if gk is not None:
                        piece_g[i][k] += gk
            del pieces
        g = torch.autograd.grad(px + pz, [p[k] for k in names])
        for k, gk in zip(names, g):
            grads[k] += gk
        tot_x += float(px)
        tot_z += float(pz)
        del x, ll, theta, px, pz, g
    out = {"x": tot_x, "z": tot_z, "grads": {k: v.numpy() for k, v in grads.items()}}
    if scales:
        out["scales"] = {k: sum(pg[k].abs() for pg in piece_g).numpy() for k in names}
    return out


def prior_scales(cfg, params, prior_weight=1.0):
    """Entry-wise yardstick of the prior gradients (oracle.energy_grad_scales, prior half)."""
    sc = O.energy_grad_scales(cfg, None, params, prior_weight=prior_weight, data=False)
    return {k: v.numpy() for k, v in sc.items()}


def prior_term(cfg, params):
    """The twelve prior parts and their gradients (one draw) from the oracle."""
    p = {k: torch.as_tensor(np.asarray(v, dtype=np.float64)).clone().requires_grad_(True)
         for k, v in params.items()}
    parts = O.prior_log_prob_parts(cfg, p)
    tot = sum(v.sum() for v in parts.values())
    g = torch.autograd.grad(tot, [p[k] for k in O.VAR_ORDER], allow_unused=True)
    grads = {k: (gk if gk is not None else torch.zeros_like(p[k])).numpy()
             for k, gk in zip(O.VAR_ORDER, g)}
    return {k: float(v.sum()) for k, v in parts.items()}, grads
