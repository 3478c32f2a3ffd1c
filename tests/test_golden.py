"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
CPU: the oracle and the sparse-exact restatement reproduce them; GPU: the HIP
path through the C-ABI matches them to 1e-5."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import spmf_oracle as O
from oracle import sparse_exact as SE

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(f):
    g = np.load(f)
    K = int(g["K"])
    D = g["x"].shape[1]
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=bool(g["scale_rows"]),
                         u_tau_scale=float(g["u_tau_scale"]))
    cfg.eta_i = torch.as_tensor(g["eta"])
    cfg.xi_u_global = float(g["xi"])
    params = {k[2:]: g[k] for k in g.files if k.startswith("p_")}
    parts = {k[5:]: g[k] for k in g.files if k.startswith("part_")}
    grads = {k[5:]: g[k] for k in g.files if k.startswith("grad_")}
    return cfg, g["x"], params, parts, grads


def test_golden_files_present():
    assert len(FILES) >= 4


@pytest.mark.parametrize("f", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(f):
    cfg, x, params, parts, grads = load(f)
    p2, g2, _ = O.energy_and_grads(cfg, x, params)
    for k in parts:
        np.testing.assert_allclose(p2[k].numpy(), parts[k], rtol=1e-12)
    for k in grads:
        np.testing.assert_allclose(g2[k].numpy(), grads[k], rtol=1e-10,
                                   atol=1e-12 * np.abs(grads[k]).max())


@pytest.mark.parametrize("f", FILES, ids=[os.path.basename(f) for f in FILES])
def test_sparse_exact_reproduces_golden(f):
    cfg, x, params, parts, grads = load(f)
    S = params["u"].shape[0]
    for s in range(S):
        one = {k: v[s] for k, v in params.items()}
        out = SE.data_term(sp.csr_matrix(x), cfg.eta_i.numpy().reshape(-1), cfg.xi_u_global,
                           cfg.scale_rows, one["u"], one["v"], one["w"], one["s"])
        np.testing.assert_allclose(out["x"], parts["x"][s], rtol=1e-12)
        np.testing.assert_allclose(out["z"], parts["z"][s], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("f", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_matches_golden(f):
    from spmf_amd import PoissonFactorization
    cfg, x, params, parts, grads = load(f)
    m = PoissonFactorization(latent_dim=cfg.latent_dim, feature_dim=cfg.feature_dim,
                             u_tau_scale=cfg.u_tau_scale, scale_rows=cfg.scale_rows,
                             column_norms=cfg.eta_i, initialize_distributions=False,
                             device="cuda", panel_rows=16)
    m.xi_u_global = cfg.xi_u_global
    got, gg, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k in parts:
        np.testing.assert_allclose(got[k].cpu().numpy(), parts[k], rtol=1e-5, atol=1e-5)
    # entry by entry against 1e-5 of the sum of |contributions| (tests/_gradcheck.py)
    from _gradcheck import assert_grads_entrywise
    assert_grads_entrywise(gg, grads, O.energy_grad_scales(cfg, x, params), 1e-5,
                           os.path.basename(f))
