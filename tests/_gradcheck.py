"""Entry-wise gradient comparison -- TEST INFRASTRUCTURE ONLY.

north_star: "gradient values within 1e-5 relative".  Entries of a sparse gradient cancel
(columns with 40 stored entries next to columns with 40 000), so "relative" needs a per-entry
yardstick; round 2 used the maximum of the whole array, which lets an entry 1e-3 of that
maximum be wrong by 1 %.  Here every entry is held to

    |hip - oracle| <= tol * (sum over the energy's additive pieces of |d piece / d entry|)

with the sum from oracle.spmf_oracle.energy_grad_scales (stored-cell part of the likelihood,
minus-rate part, z prior, every additive term of the prior log-densities): what fp32 can be
asked to resolve is the magnitude of what was added up, not of what was left after cancelling.
"""
import numpy as np


def worst_entry(g, r, sc):
    """(max over entries of |g - r| / scale, flat index of that entry)."""
    g, r, sc = (np.asarray(a, dtype=np.float64) for a in (g, r, sc))
    g = g.reshape(r.shape)
    sc = np.broadcast_to(sc, r.shape)
    err = np.abs(g - r)
    # an entry nothing contributes to (empty column, unused variable) must be exactly equal
    ratio = np.where(sc > 0, err / np.where(sc > 0, sc, 1.0), np.where(err > 0, np.inf, 0.0))
    i = int(np.argmax(ratio))
    return float(ratio.reshape(-1)[i]), i


def assert_grads_entrywise(got, ref, scales, tol=1e-5, tag="", norm_tol=None):
    """got: name -> tensor / array (any float dtype, device); ref, scales: name -> array-like
    of the reference's shape.

    BOTH bounds must hold: the entry-wise one above, and the array-norm bound of rounds
    1-2, max|hip - oracle| <= norm_tol * max|oracle| (norm_tol defaults to tol).  The yardstick of a
    strongly cancelling entry can be orders of magnitude above |oracle| there, so on the DOMINANT
    entries the entry-wise bound alone is the looser of the two; the norm bound keeps those tight."""
    norm_tol = tol if norm_tol is None else norm_tol
    for k in ref:
        g = got[k]
        if hasattr(g, "detach"):
            g = g.detach().cpu().double().numpy()
        r = ref[k].numpy() if hasattr(ref[k], "numpy") else np.asarray(ref[k])
        sc = scales[k].numpy() if hasattr(scales[k], "numpy") else np.asarray(scales[k])
        w, i = worst_entry(g, r, sc)
        assert w <= tol, (tag, k, f"entry {i}: |hip-oracle| = {w:.3e} x its yardstick "
                          f"(ref {r.reshape(-1)[i]:.6e}, yardstick {np.broadcast_to(sc, r.shape).reshape(-1)[i]:.6e})")
        g64 = np.asarray(g, dtype=np.float64).reshape(r.shape)
        top = float(np.abs(r).max()) if r.size else 0.0
        err = float(np.abs(g64 - r).max()) if r.size else 0.0
        assert err <= norm_tol * top + 1e-300, (tag, k, f"array norm: max|hip-oracle| = {err:.3e} = "
                                                f"{err / max(top, 1e-300):.3e} x max|oracle| ({top:.3e})")
