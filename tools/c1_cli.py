"""C1 of BASELINE.json: bin/factorize_csv.py on a 5k x 200 dense random Poisson CSV, K=2, batch 5000
(notebooks/factorizing_random_noise.ipynb shape), default flags.  Prints wall-clock of the whole CLI
(CSV parse, 300 epochs, outputs) and of the training loop alone."""
import os, subprocess, sys, tempfile, time
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(0)
X = rng.poisson(1.0, size=(5000, 200))
d = tempfile.mkdtemp(prefix="spmf_c1_")
f = os.path.join(d, "noise.csv")
np.savetxt(f, X, fmt="%d", delimiter=",")
epochs = sys.argv[1] if len(sys.argv) > 1 else "300"
t0 = time.time()
p = subprocess.run([sys.executable, os.path.join(root, "bin", "factorize_csv.py"), "-f", f, "-e", epochs],
                   capture_output=True, text=True)
dt = time.time() - t0
lines = [l for l in p.stdout.splitlines() if l.startswith("Epoch")]
print("rc", p.returncode, "wall_s", round(dt, 2), "epochs_run", len(lines))
print(lines[0] if lines else "", "|", lines[-1] if lines else "")
print(sorted(os.listdir(d)))
if p.returncode:
    print(p.stderr[-2000:])
