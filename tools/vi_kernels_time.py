"""Times the O(D*K) elementwise kernels of a VI step at C3 parameter sizes (no data needed)."""
import sys, time, contextlib, math
import numpy as np, torch
sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, vi
D, K = 20000, 32
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1e-4, device="cuda")
sur = m.surrogate_distribution
opt = vi.AdamHIP(m, sur.trainable_variables, 1e-3); opt.init_state(3.0)
opt.state[9] = 1.0
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t) / n
noise = sur.draw_noise(1)
theta, logq = sur.forward_hip(m, 1, noise)
g = {n: torch.randn_like(theta[n]) for n in theta}
grads = sur.backward_hip(m, 1, noise, g, 1e-6, 0.5)
print("draw_noise   %.1f us" % timeit(lambda: sur.draw_noise(1)))
print("surrogate_fwd %.1f us" % timeit(lambda: sur.forward_hip(m, 1, noise)))
print("surrogate_bwd %.1f us" % timeit(lambda: sur.backward_hip(m, 1, noise, g, 1e-6, 0.5)))
print("adam_dev     %.1f us" % timeit(lambda: opt.step_dev(grads)))
