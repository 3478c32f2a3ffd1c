"""Extended randomized parity runs for the other likelihood/decoder modes (log_transform, Bernoulli,
mixed), with the problem builders of the test modules.  Not part of the test suite."""
import sys, contextlib, io, math
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import spmf_oracle as O
import test_gpu_logtransform as TL, test_gpu_bernoulli as TB, test_gpu_mixed as TM
from spmf_amd import PoissonFactorization, BernoulliFactorization, MixedFactorization

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 99
rng = np.random.default_rng(seed0)

def check(tag, parts, grads, pref, gref, acc):
    for k in pref:
        a, r = parts[k].cpu().numpy(), pref[k].numpy()
        e = float(np.max(np.abs(a - r) / np.maximum(np.abs(r), 1.0)))
        acc[0] = max(acc[0], e)
        if e > 5e-6: print("part", tag, k, f"{e:.2e}")
    for k in gref:
        g = grads[k].cpu().double().numpy().reshape(gref[k].shape); r = gref[k].numpy()
        e = float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-30))
        acc[1] = max(acc[1], e)
        if e > 5e-6: print("grad", tag, k, f"{e:.2e}")

for mode in ("logt", "bern", "bernlogt", "mixed"):
    acc = [0.0, 0.0]
    for case in range(N):
        B = int(rng.integers(2, 500)); D = int(rng.integers(2, 600)); K = int(rng.integers(1, 65))
        S = int(rng.integers(1, 3)); density = float(rng.choice([0.02, 0.1, 0.5]))
        P = int(rng.choice([7, 128, 4096])); sr = bool(rng.integers(0, 2))
        seed = seed0 * 1000 + case
        tag = f"{mode} case {case}: B={B} D={D} K={K} S={S} dens={density} sr={sr} P={P}"
        with contextlib.redirect_stdout(io.StringIO()):
            if mode == "logt":
                cfg, x, params = TL.problem(B, D, K, S, seed, density, sr)
                m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale, scale_rows=sr,
                                         log_transform=True, column_norms=cfg.eta_i,
                                         initialize_distributions=False, device="cuda", panel_rows=P)
                m.xi_u_global = cfg.xi_u_global
            elif mode in ("bern", "bernlogt"):
                lt = mode == "bernlogt"
                cfg, x, params = (TB.problem_logt if lt else TB.problem)(B, D, K, S, seed, density)
                m = BernoulliFactorization(latent_dim=K, feature_dim=D, u_tau_scale=cfg.u_tau_scale,
                                           column_norms=cfg.eta_i, log_transform=lt, device="cuda", panel_rows=P)
            else:
                cfg, x, params, mask = TM.problem(B, D, K, S, seed, density, sr)
                m = MixedFactorization(mask, latent_dim=K, u_tau_scale=cfg.u_tau_scale, scale_rows=sr,
                                       column_norms=cfg.eta_i, device="cuda", panel_rows=P)
                m.xi_u_global = cfg.xi_u_global
        pref, gref, _ = O.energy_and_grads(cfg, x, params)
        parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
        if float(nnf.sum()) != 0:
            print("NONFINITE", tag); continue
        check(tag, parts, grads, pref, gref, acc)
        del m
    print(f"{mode}: {N} cases, worst part error {acc[0]:.3e}, worst gradient error {acc[1]:.3e}")
