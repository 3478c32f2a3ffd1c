"""Extended randomized parity run (not part of the test suite): N random linear-Poisson cases with
fresh seeds, bigger shapes than the tests use, against the fp64 oracle.  Prints the worst errors."""
import sys, contextlib, io
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from oracle import spmf_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed0)
worst_p, worst_g, fails = 0.0, 0.0, 0
for case in range(N):
    B = int(rng.integers(1, 700)); D = int(rng.integers(1, 900)); K = int(rng.integers(1, 65))
    if len(sys.argv) > 3 and sys.argv[3] == "widek":     # latent dimensions 65 .. 256 (csrc/widek.hip)
        K = int(rng.integers(65, 257)); B = min(B, 300); D = min(D, 500)
    S = int(rng.integers(1, 3)); density = float(rng.choice([0.0, 0.01, 0.05, 0.3, 1.0]))
    sr = bool(rng.integers(0, 2)); P = int(rng.choice([1, 33, 256, 8192]))
    cfg, x, params = T.make_problem(B, D, K, S, seed0 * 1000 + case, density, sr, empty=False)
    pref, gref, _ = O.energy_and_grads(cfg, x, params)
    with contextlib.redirect_stdout(io.StringIO()):
        m = T.build_model(cfg, P)
    parts, grads, nnf = m.energy_and_grads({"counts": x}, params)
    tag = f"case {case}: B={B} D={D} K={K} S={S} dens={density} sr={sr} P={P}"
    if float(nnf.sum()) != 0:
        print("NONFINITE", tag); fails += 1; continue
    for k in pref:
        a, r = parts[k].cpu().numpy(), pref[k].numpy()
        e = float(np.max(np.abs(a - r) / np.maximum(np.abs(r), 1.0)))
        worst_p = max(worst_p, e)
        if e > 2e-6: print("part>2e-6", tag, k, f"{e:.2e} got {float(a.reshape(-1)[0]):.9g} ref {float(r.reshape(-1)[0]):.9g}")
        if e > 1e-5: print("PART", tag, k, e); fails += 1
    for k in gref:
        g = grads[k].cpu().double().numpy().reshape(gref[k].shape); r = gref[k].numpy()
        e = float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-30))
        worst_g = max(worst_g, e)
        if e > 1e-5: print("GRAD", tag, k, e); fails += 1
    del m
print(f"{N} cases: worst part error {worst_p:.3e}, worst gradient error {worst_g:.3e}, failures {fails}")
