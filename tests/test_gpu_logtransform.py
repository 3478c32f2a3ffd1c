"""GPU parity of the log_transform decoder (poisson.py:41-42,52-53): sparse
stored-cell terms + dense f32-MFMA exp sums vs the dense fp64 oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import spmf_oracle as O
from _gradcheck import assert_grads_entrywise

pytestmark = pytest.mark.gpu


def problem(B, D, K, S, seed, density, scale_rows=True):
    rng = np.random.default_rng(seed)
    x = ((rng.random((B, D)) < density) * (1 + rng.poisson(2.0, size=(B, D)))).astype(np.float64)
    if B > 4 and D > 4:
        x[1, :] = 0
        x[:, 2] = 0
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, scale_rows=scale_rows, log_transform=True,
                         u_tau_scale=1.0 / math.sqrt(B * D))
    cfg.eta_i = torch.as_tensor(rng.uniform(0.5, 3.0, size=(1, D)))
    cfg.xi_u_global = float(rng.uniform(4.0, 8.0))
    params = O.random_params(cfg, S, seed + 1, fp32_exact=True)
    # keep exp(<z, eta v>) in a sane range: rescale v so the largest exponent is 8
    T = torch.as_tensor
    z = O.encode(cfg, T(x), T(params["u"]), T(params["s"]))
    ymax = float((torch.matmul(z, T(params["v"])) * cfg.eta_i).max())
    params["v"] *= 8.0 / ymax
    return cfg, x, params


CASES = [(37, 23, 3, 2, 0.3, True), (150, 90, 8, 1, 0.1, True), (260, 200, 32, 1, 0.05, True),
         (130, 70, 50, 1, 0.1, False), (300, 129, 64, 2, 0.05, True)]


@pytest.mark.parametrize("bf16x3", ["1", "0"])
@pytest.mark.parametrize("B,D,K,S,density,scale_rows", CASES + [(1500, 700, 20, 1, 0.02, True)])
def test_log_transform_energy_and_grads(monkeypatch, B, D, K, S, density, scale_rows, bf16x3):
    """Both dense paths (read at spmf_ctx_create): the bf16x3 kernels (csrc/dense3.hip: expdot3 at K padded
    to 64, the sigdot3 family's exp form at K padded to 32) and the exact-f32 MFMA kernels; the last case
    spans several Q tiles and chunks."""
    from spmf_amd import PoissonFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", bf16x3)
    cfg, x, params = problem(B, D, K, S, 500 + B + K, density, scale_rows)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             scale_rows=scale_rows, log_transform=True, column_norms=cfg.eta_i,
                             initialize_distributions=False, device="cuda", panel_rows=64)
    m.xi_u_global = cfg.xi_u_global
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                   err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")
    T = torch.as_tensor
    z = m.encode(x, u=T(params["u"]), s=T(params["s"])).cpu().numpy()
    zr = O.encode(cfg, T(x), T(params["u"]), T(params["s"])).numpy()
    np.testing.assert_allclose(z, zr, rtol=1e-5, atol=1e-5 * np.abs(zr).max())


@pytest.mark.parametrize("B,D,K,S", [(300, 129, 64, 2), (260, 200, 32, 1)])
def test_two_launch_form_still_matches_oracle(monkeypatch, B, D, K, S):
    """SPMF_DENSE_E_ONCE=0 (read at spmf_ctx_create): the form that recomputes E in a second
    exp launch instead of keeping it in HBM -- same oracle, same contract."""
    from spmf_amd import PoissonFactorization
    monkeypatch.setenv("SPMF_DENSE_E_ONCE", "0")
    monkeypatch.setenv("SPMF_DENSE_BF16X3", "0")        # (a switch of the exact-f32 kernels)
    cfg, x, params = problem(B, D, K, S, 900 + B + K, 0.05)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             log_transform=True, column_norms=cfg.eta_i,
                             initialize_distributions=False, device="cuda", panel_rows=64)
    m.xi_u_global = cfg.xi_u_global
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")


def test_e_buffer_in_several_row_chunks(monkeypatch):
    """spmf_ctx_set_e_cap small enough that the rows go through the exact-f32 dense kernels (the ones
    that keep E: SPMF_DENSE_BF16X3=0) in several chunks (at C4 scale the Python class sizes the buffer for
    ONE chunk, so the chunk loop is exercised here): 4000 rows with a 1 MiB cap = 3 chunks of
    1408 / 1408 / 1184 rows."""
    from spmf_amd import PoissonFactorization, _lib
    monkeypatch.setenv("SPMF_DENSE_BF16X3", "0")
    B, D, K, S = 4000, 129, 64, 1
    cfg, x, params = problem(B, D, K, S, 4321, 0.03)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             log_transform=True, column_norms=cfg.eta_i,
                             initialize_distributions=False, device="cuda", panel_rows=512)
    m.xi_u_global = cfg.xi_u_global
    lib, h = _lib.load(), m._handle()
    _lib.check(h, lib.spmf_ctx_set_e_cap(h, 1 << 20), "spmf_ctx_set_e_cap")
    assert (1 << 20) // (160 * 4) < B          # the cap really is below one chunk of all rows
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, "")


def test_log_transform_randomised_sweep():
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(77)
    for case in range(12):
        B, D = int(rng.integers(2, 300)), int(rng.integers(2, 300))
        K, S = int(rng.integers(1, 65)), int(rng.integers(1, 3))
        density = float(rng.choice([0.03, 0.2, 0.8]))
        sr = bool(rng.integers(0, 2))
        cfg, x, params = problem(B, D, K, S, 9000 + case, density, sr)
        pref, gref, _ = O.energy_and_grads(cfg, x, params)
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                 scale_rows=sr, log_transform=True, column_norms=cfg.eta_i,
                                 initialize_distributions=False, device="cuda",
                                 panel_rows=int(rng.choice([5, 64, 4096])))
        m.xi_u_global = cfg.xi_u_global
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        tag = f"case {case}: B={B} D={D} K={K} S={S} dens={density} sr={sr}"
        assert float(nnf.sum()) == 0, tag
        for k, r in pref.items():
            np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5,
                                       err_msg=f"{tag} {k}")
        assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params), 1e-5, tag)


def test_log_transform_fit_smoke():
    """The training loop (device-gated steps, hipGraph replay from the second
    epoch on) over the log_transform decoder: dense exp kernels inside the graph."""
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(5)
    N, D, K = 400, 40, 3
    X = rng.poisson(rng.gamma(1.0, 1.0, size=(1, D)) * rng.gamma(2.0, 0.5, size=(N, 1))).astype(np.float64)
    colmean = np.maximum(X.mean(0, keepdims=True), 1e-3)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             log_transform=True, column_norms=colmean, device="cuda", panel_rows=128)
    m.xi_u_global = float(colmean.sum())
    torch.manual_seed(1)
    losses = m.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=4, num_steps=20,
                   learning_rate=0.02, rel_tol=1e-9, verbose=False)
    assert len(losses) == 20 and all(math.isfinite(v) for v in losses)
    assert np.mean(losses[-3:]) < losses[0]
    z = m.encode(X)
    assert tuple(z.shape) == (N, K) and bool(torch.isfinite(z).all())


def test_fit_reports_decoder_saturation(capsys):
    """a step whose exponents exceed kYSat = 70 (csrc/common.h) trains on
    exp(min(y, 70)); the device-gated loop must not do that silently.  A tiny xi_u_global makes
    the row scales -- and with them <z, eta v> -- huge at the initial values: the saturation
    count reaches the VI state ([14]), fit prints the warning and keeps the running count."""
    from spmf_amd import PoissonFactorization
    rng = np.random.default_rng(6)
    N, D, K = 300, 40, 3
    X = rng.poisson(3.0, size=(N, D)).astype(np.float64)
    colmean = np.maximum(X.mean(0, keepdims=True), 1e-3)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                             log_transform=True, column_norms=colmean, device="cuda", panel_rows=128)
    m.xi_u_global = 1e-7 * float(colmean.sum())
    torch.manual_seed(2)
    params = m.surrogate_distribution.sample(1)
    m.energy_and_grads({"counts": X}, params)
    assert float(m.last_saturated.sum()) > 0            # the premise: this point saturates
    losses = m.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=2, num_steps=3,
                   learning_rate=1e-3, rel_tol=1e-12, verbose=True)
    out = capsys.readouterr().out
    assert "Decoder saturated" in out and "min(y, 70)" in out
    assert m.saturated_events > 0
    assert len(losses) == 3 and all(math.isfinite(v) for v in losses)
    # a model that never saturates says nothing and counts nothing
    m2 = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1 / math.sqrt(N * D),
                              log_transform=True, column_norms=colmean, device="cuda", panel_rows=128)
    m2.xi_u_global = float(colmean.sum())
    m2.fit(lambda: [{"counts": X}], dataset_size=N, sample_size=2, num_steps=2,
           learning_rate=1e-3, rel_tol=1e-12, verbose=True)
    assert "Decoder saturated" not in capsys.readouterr().out
    assert getattr(m2, "saturated_events", 0.0) == 0.0


@pytest.mark.parametrize("B,D,S,ymax,K", [(300, 129, 2, 8.0, 64), (700, 333, 1, 8.0, 64), (513, 64, 1, 30.0, 64),
                                          (90, 1000, 1, 45.0, 64), (90, 1000, 1, 60.0, 64),
                                          (700, 333, 1, 8.0, 24), (513, 64, 1, 30.0, 32), (90, 1000, 1, 45.0, 17)])
def test_bf16x3_dense_path_matches_oracle(monkeypatch, B, D, S, ymax, K):
    """The default dense path (SPMF_DENSE_BF16X3, read at spmf_ctx_create): the exp sums on the bf16
    matrix cores with three-way split operands (csrc/dense3.hip: expdot3 at K padded to 64, the exp form of
    the sigdot3 family at K padded to 32).  Same oracle,
    same 1e-5 contract, gradients entry by entry; exponents up to 60 -- exp amplifies the absolute
    error of <z, eta v>, and there the exact-f32 MFMA kernels themselves are at 1.6 - 2.3e-5 of
    the yardstick (tools/b3_err.py) while this path, which keeps a1 b1 and
    the small partial products in separate accumulators, measured 6 - 8.5e-6; B, D off the
    32 / 64 / 256 tile edges; several Q chunks in the W-stationary launch (D = 1000)."""
    from spmf_amd import PoissonFactorization
    monkeypatch.setenv("SPMF_DENSE_BF16X3", "1")
    cfg, x, params = problem(B, D, K, S, 2900 + B + D, 0.05)
    params["v"] *= ymax / 8.0                  # problem() scaled the largest exponent to 8
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                             log_transform=True, column_norms=cfg.eta_i,
                             initialize_distributions=False, device="cuda", panel_rows=64)
    m.xi_u_global = cfg.xi_u_global
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    assert float(nnf.sum()) == 0 and float(m.last_saturated.sum()) == 0
    for k, r in pref.items():
        np.testing.assert_allclose(parts[k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    # The contract's 1e-5 up to exponents of 45; 1.5e-5 at 60, and the entry that needs it (round 5, asked for
    # by VERDICT r4 #6: the bound was tried at 1e-5): dE/dv[50462] of the 90 x 1000 case = -2.06e20, its
    # yardstick the same 2.06e20 -- ONE cell with an exponent y near 47 carries the whole entry, so the
    # relative error of the entry is d exp(y) / exp(y) = the ABSOLUTE error of y, a 64-term fp32 sum of that
    # size: ulp(47) = 3.8e-6 and a handful of roundings on the way measured 1.28e-5 (8.5e-6 on round 3's worst
    # entry).  No fp32 evaluation of <z, eta v> -- this kernel's or the exact-f32 MFMA one's, which is at
    # 1.6 - 2.3e-5 there -- has more to give; below exponents of 45 both hold the contract.
    assert_grads_entrywise(grads, gref, O.energy_grad_scales(cfg, x, params),
                           1e-5 if ymax <= 45.0 else 1.5e-5, f"bf16x3 {B}x{D}")
    # and it agrees with the exact-f32 MFMA form of the same library far inside the contract
    monkeypatch.setenv("SPMF_DENSE_BF16X3", "0")
    m2 = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                              log_transform=True, column_norms=cfg.eta_i,
                              initialize_distributions=False, device="cuda", panel_rows=64)
    m2.xi_u_global = cfg.xi_u_global
    parts2, grads2, _ = m2.energy_and_grads({"counts": x}, params)
    # (the two kernels are each within the contract of the oracle; at exponents of 45 - 60 both
    #  are a few 1e-6 from it, on either side)
    assert abs(float(parts["x"].sum()) - float(parts2["x"].sum())) <= 1e-5 * abs(float(parts2["x"].sum()))
    # ... and so do the gradients, entry by entry (pins the split -- P and Q in three bf16
    # planes, E = exp(X) in TWO in the second product, |dE| <= 2^-18 E per term -- against a regression):
    # the two kernels together may use the two bounds above, no more
    tol2 = 2e-5 if ymax <= 45.0 else 3e-5
    assert_grads_entrywise(grads, {k: v.double().cpu().numpy() for k, v in grads2.items()},
                           O.energy_grad_scales(cfg, x, params), tol2, f"bf16x3 vs f32 MFMA {B}x{D} ymax={ymax}")


@pytest.mark.parametrize("B,D,S", [(4500, 300, 1), (4700, 1500, 2)])
def test_fused_row_pass_of_the_exp_decoder_matches_the_two_launch_flow(monkeypatch, B, D, S):
    """SPMF_FUSE_ROWS=1 (off by default: slower on C4, profiles/r04_fused_rows_c4.txt): ONE row pass for the
    Poisson log_transform context at K = 64 -- sweep 1 reads g(x), sweep 2 takes the count out of the packed
    word (row_pass.hip FMT 2), mode 3 leaves xi (gz - z) in gzs -- and the (Z, W) expdot3 launch subtracts
    xi_b sum_d E_bd V'_d in its epilogue.  Needs >= 4096 rows (the LDS-phi launch shapes carry the two-stream
    form).  Against the oracle at the contract, and against the default flow of the same library."""
    from spmf_amd import PoissonFactorization
    K = 64
    cfg, x, params = problem(B, D, K, S, 4100 + D, 0.03)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SPMF_FUSE_ROWS", fuse)
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, log_transform=True,
                                 column_norms=cfg.eta_i, initialize_distributions=False, device="cuda")
        m.xi_u_global = cfg.xi_u_global
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        assert float(nnf.sum()) == 0
        out[fuse] = (parts, grads)
    scales = O.energy_grad_scales(cfg, x, params)
    for k, r in pref.items():
        np.testing.assert_allclose(out["1"][0][k].cpu().numpy(), r.numpy(), rtol=1e-5, atol=1e-5, err_msg=k)
    assert_grads_entrywise(out["1"][1], gref, scales, 1e-5, "fused exp row pass")
    for k in out["1"][0]:
        np.testing.assert_allclose(out["1"][0][k].cpu().numpy(), out["0"][0][k].cpu().numpy(), rtol=2e-6, err_msg=k)
    assert_grads_entrywise(out["1"][1], {k: v.double().cpu().numpy() for k, v in out["0"][1].items()}, scales,
                           4e-6, "fused vs two launches")
