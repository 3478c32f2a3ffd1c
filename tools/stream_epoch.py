"""Training epochs over minibatches that ARRIVE from the host every step (the reference's flow: a
tf.data iterator of batches, tests/spmf_test.py:17-43), as scipy CSR: upload +
device layout + statistics + the device-gated VI step, per batch.  Library layout builder against
the torch construction (SPMF_NATIVE_LAYOUT=0).  usage: stream_epoch.py [rows_total] [batch_rows]"""
import contextlib
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 22_784
D, K, dens = 20_000, 32, 0.005
dev = torch.device("cuda", 0)

cnts, cols, vals, done, cid = [], [], [], 0, 0
while done < rows:
    n = min(synth.CHUNK_ROWS, rows - done)
    cnt, c, x = synth.linear_structure_chunk(cid, n, D, dens, dev)
    cnts.append(cnt.cpu()); cols.append(c.cpu()); vals.append(x.cpu())
    done += n; cid += 1
indptr = np.concatenate([[0], np.cumsum(torch.cat(cnts).numpy())])
X = sp.csr_matrix((torch.cat(vals).numpy(), torch.cat(cols).numpy(), indptr), shape=(rows, D))
cuts = list(range(0, rows, bs))


slices = [X[r0:min(r0 + bs, rows)] for r0 in cuts]


def host_batches():
    for b in slices:      # a fresh object every step (same host arrays): no device layout is re-used
        yield {"counts": sp.csr_matrix((b.data, b.indices, b.indptr), shape=b.shape)}


for mode in ("0", "1"):
    os.environ["SPMF_NATIVE_LAYOUT"] = mode
    with contextlib.redirect_stdout(sys.stderr):
        m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
    m.max_cached_batches = 1
    cm = np.asarray(X.sum(0)).ravel() / np.maximum(np.asarray((X > 0).sum(0)).ravel(), 1)
    m.eta_i = torch.as_tensor(np.where(cm > 1, cm, 1.0)).reshape(1, D)
    m.xi_u_global = float(np.nansum(cm))
    torch.manual_seed(0)
    opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-3)
    opt.init_state(3.0)
    run = vi.StepRunner(m, opt, rows, 1, use_graph=False)
    ts = []
    for ep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.reset_epoch_counters()
        for b in host_batches():
            run.step(b)
        st = opt.read_state()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps({"batches": "scipy CSR on the host, new object per step", "layout": "library" if mode == "1" else "torch",
                      "rows": rows, "batch_rows": bs, "steps_per_epoch": len(cuts),
                      "batch_mb_over_the_host_link": round(slices[0].nnz * 8 / 1e6, 1),
                      "ms_per_step": [round(1e3 * t / len(cuts), 3) for t in ts],
                      "epoch_loss": round(st[10] / max(st[11], 1), 4), "skipped": int(st[12])}), flush=True)
    del m, opt, run
