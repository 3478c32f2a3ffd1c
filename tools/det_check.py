"""Deterministic mode against the default path on a C3-shaped shard: repeats of each, bit equality."""
import contextlib
import sys

import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

rows, D, K = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000, 20_000, 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
models = {}
for det in (False, True):
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                 deterministic=det)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev); colnnz = torch.zeros_like(colsum)
    sc.compute_stats(m._handle(), colsum, colnnz)
    cm = colsum / colnnz
    m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    m.xi_u_global = float(torch.nansum(cm))
    models[det] = m
torch.manual_seed(7)
params = models[False].surrogate_distribution.sample(S)
runs = {}
for det in (False, True, False, True):
    for rep in range(3):
        p, g, _ = models[det].energy_and_grads({"counts": sc}, params)
        runs.setdefault(det, []).append(({k: v.clone() for k, v in p.items()}, {k: v.clone() for k, v in g.items()}))
for det in (False, True):
    r = runs[det]
    eq = [all(torch.equal(r[0][1][k], x[1][k]) for k in r[0][1]) and all(torch.equal(r[0][0][k], x[0][k]) for k in r[0][0])
          for x in r[1:]]
    print(f"deterministic={det}, S={S}: {len(r)} runs, identical to the first: {eq}")
a, b = runs[True][0][1], runs[False][0][1]
for k in ("u", "v", "w", "s"):
    d = (a[k] - b[k]).abs()
    print(f"  det vs default, grad {k}: max abs {float(d.max()):.3e} of max {float(b[k].abs().max()):.3e}; differing entries {int((d > 0).sum())} of {d.numel()}")
