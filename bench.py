#!/usr/bin/env python3
"""bench.py -- ELBO steps/s + achieved HBM GB/s of the spmf hot path on MI355X.

One *step* = one evaluation of all 14 energy parts plus the gradient wrt all
12 latent variables for one batch and S Monte-Carlo draws: the
prep, row-pass, column-pass and finish kernels behind the C-ABI, plus -- with
more than one rank -- the single RCCL sum-all-reduce of the packed gradient
accumulators over the row shards.  Inputs are resident in HBM before the
timed region starts.

Default workload = BASELINE.json configs[2] ("C3"): 1M x 20k linear-structure
counts, nnz ~ 1e8, K = 32, S = 1, full batch, row-sharded over the ranks
(strong scaling: the matrix is fixed, so steps/s should rise with N).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 measured copy)

WORKLOADS = {
    # name: (rows, cols, density, K, description)
    "c3": (1_000_000, 20_000, 0.005, 32,
           "C3: 1M x 20k linear-structure Poisson counts, nnz~1e8, K=32"),
    "c2": (100_000, 5_000, 0.01, 16,
           "C2: 100k x 5k, 1% nnz, K=16"),
    "small": (50_000, 2_000, 0.01, 32, "smoke-sized: 50k x 2k, 1% nnz, K=32"),
    # scRNA-shaped, row scaling + log_transform on (bin/factorize_scrnaseq_counts.py:93-99):
    # the dense exp sums run on the f32 matrix cores, roofline bound = mfma
    "c4": (500_000, 30_000, 0.03, 64,
           "C4: 500k x 30k scRNA-shaped counts, ~3% nnz, K=64, log_transform"),
    "c4small": (50_000, 6_000, 0.03, 64, "C4-shaped smoke: 50k x 6k, ~3% nnz, K=64, log_transform"),
    # build-defined mixed likelihood (mixed.py is empty in the reference)
    "c5": (200_000, 10_000, 0.03, 32,
           "C5: 200k x 10k, even columns Poisson (1% nnz), odd columns Bernoulli(0.05), K=32, mixed"),
}
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (datasheet; ~1.25 PF measured on random data)


def algorithmic_bytes(nnz, B, D, K, S):
    """SURVEY 8d two-pass model, canonical fp32 values + int32 indices."""
    row = 8 * nnz + 4 * (B + 1) + S * (8 * B * K + 8 * D * K + 8 * D)
    col = 8 * nnz + 4 * (D + 1) + S * (8 * B * K + 8 * D * K + 8 * D)
    return row, col, row + col


def cpu_baseline(sc, model, params, K, max_rows=125_000):
    """The CPU restatement (oracle/sparse_exact_omp.c through oracle/sparse_exact_c.py,
    kind "port": fp64, OpenMP over ALL host cores) timed on one whole row shard of
    the same workload -- 125k rows of C3, the per-GPU shard of the 8-GPU
    configuration; whole panels, so the HIP path runs on exactly the same rows and is
    compared with it.  3 warm-ups, median of 10 steps."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    from oracle import sparse_exact as SE
    from oracle import sparse_exact_c as SC
    npan = max(1, min(sc.n_panels, max_rows // sc.panel_rows))
    n = min(sc.n_rows, npan * sc.panel_rows)
    hi = int(sc.row_ptr[n])
    X = sp.csr_matrix((sc.val[:hi].cpu().numpy().astype(np.float64),
                       sc.col_idx[:hi].cpu().numpy(),
                       sc.row_ptr[:n + 1].cpu().numpy()), shape=(n, sc.n_cols))
    one = {k: v[0].double().cpu().numpy() for k, v in params.items()}
    eta = model._eta_device().double().cpu().numpy()
    decay = model.symmetry_breaking_decay ** np.arange(K)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    prep = SC.Prepared(X, eta, float(model.xi_u_global), model.scale_rows)
    # thread count: the fastest of a short sweep up to every host core (a 256-thread
    # box is not fastest at 256 threads on a gather-bound loop); `cores` reports the
    # count actually used for the timed steps
    best_t, best_n = None, cores
    for nthr in sorted({c for c in (16, 32, 64, 128, cores) if c <= cores}):
        SC.set_threads(nthr)
        prep.step(one["u"], one["v"], one["w"], one["s"])
        t0 = time.perf_counter()
        prep.step(one["u"], one["v"], one["w"], one["s"])
        t1 = time.perf_counter() - t0
        if best_t is None or t1 < best_t:
            best_t, best_n = t1, nthr
    cores = best_n
    SC.set_threads(cores)

    def step():
        out = prep.step(one["u"], one["v"], one["w"], one["s"])
        SE.prior_term(one, model.u_tau_scale, model.s_tau_scale, decay)
        return out
    # the checker's other job: the HIP path on the same rows, same draw (data term only),
    # entry by entry against the port's own yardstick (per entry the sum of |contributions| of the
    # stored-cell, minus-rate and z-prior pieces: tests/_gradcheck.py's metric, pinned to
    # oracle.energy_grad_scales in tests/test_oracle.py) -- and the array-norm figure beside it
    ref = prep.step(one["u"], one["v"], one["w"], one["s"], scales=True)
    SE.prior_term(one, model.u_tau_scale, model.s_tau_scale, decay)
    first = {k: v[:1] for k, v in params.items()}
    parts, grads, _ = model.energy_and_grads({"counts": sc, "panels": (0, npan)}, first,
                                             prior_weight=0.0)
    torch.cuda.synchronize()
    gerr, worst, worst_at = 0.0, 0.0, None
    for name in ("u", "v", "w", "s"):
        r = np.asarray(ref["grads"][name], dtype=np.float64)
        g = grads[name][0].double().cpu().numpy().reshape(r.shape)
        err = np.abs(g - r)
        gerr = max(gerr, float(err.max() / max(np.abs(r).max(), 1e-300)))
        ys = ref["scales"][name].reshape(r.shape)
        ratio = np.where(ys > 0, err / np.where(ys > 0, ys, 1.0), np.where(err > 0, np.inf, 0.0))
        i = int(np.argmax(ratio))
        if float(ratio.reshape(-1)[i]) >= worst:
            worst, worst_at = float(ratio.reshape(-1)[i]), f"{name}[{i}]"
    parity = {"rows": int(n),
              "x_rel": abs(float(parts["x"][0]) - ref["x"]) / abs(ref["x"]),
              "z_rel": abs(float(parts["z"][0]) - ref["z"]) / abs(ref["z"]),
              "grad_worst_entry": worst, "grad_worst_entry_at": worst_at,
              "grad_worst_entry_unit": "|hip - port| / sum of |contributions| to that entry; contract 1e-5",
              "grad_max_rel": gerr}
    for _ in range(2):
        step()                                   # 3 warm-ups with the one above
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = 0.5 * (ts[4] + ts[5])
    return n, hi, dt, parity, cores


def c1_dense_cpu_baseline():
    """The reference's own shape of computation on its CPU-runnable config (C1: 5k x 200
    dense Poisson(1) counts, K = 2, one batch of 5000): the dense fp64 oracle
    (oracle/spmf_oracle.py) + torch autograd on all host cores; 3 warm-ups, median of 10."""
    import math
    import numpy as np
    import torch
    from oracle import spmf_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # 5000 x 200: more threads than that only add overhead
    torch.set_num_threads(cores)
    rng = np.random.default_rng(20241218 + 1)
    N, D, K = 5000, 200, 2
    x = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    cfg = O.OracleConfig(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / math.sqrt(N * D))
    O.compute_scales(cfg, [x])
    p = O.random_params(cfg, 1, 3)
    ts = []
    for it in range(13):
        t0 = time.perf_counter()
        O.energy_and_grads(cfg, x, p)
        if it >= 3:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1.0 / (0.5 * (ts[4] + ts[5])), cores


def _timed_steps(model, batch, params, steps, warmup):
    """ms per energy + gradient step and the kernel taps (spmf_last_timing) of `steps` calls after `warmup`."""
    import ctypes as C
    import torch
    from spmf_amd import _lib
    lib, h = _lib.load(), model._handle()
    for _ in range(warmup):
        model.energy_and_grads(batch, params)
    torch.cuda.synchronize()
    lib.spmf_ctx_enable_timing(h, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        parts, _, nnf = model.energy_and_grads(batch, params)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    t6 = (C.c_float * 6)()
    _lib.check(h, lib.spmf_last_timing(h, t6), "spmf_last_timing")
    lib.spmf_ctx_enable_timing(h, 0)
    return ms, t6, parts, nnf


def c5_extra(dev, steps=10, warmup=3):
    """BASELINE config 5 (mixed likelihood: 200k x 10k, even columns Poisson, odd columns
    Bernoulli, K = 32) on this one GPU, for the `also` block of the default line: ms per energy +
    gradient step, the kernel taps, and the dense sigmoid kernels against the f32-MFMA peak on
    the algorithmic 6*B*D_bern*K.  Generated after the headline's timed region."""
    import contextlib
    import torch
    from spmf_amd import MixedFactorization, synth
    from spmf_amd.sparse import balanced_panel_rows
    rows, D, _density, K, _desc = WORKLOADS["c5"]
    pr = balanced_panel_rows(rows, K)
    sc, mask = synth.mixed_c5(rows, D, dev, 20241218 + 5, panel_rows=pr, first_chunk=0,
                              chunk_rows=synth.MIXED_CHUNK_ROWS)
    with contextlib.redirect_stdout(sys.stderr):
        model = MixedFactorization(mask, latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                   device=dev, panel_rows=pr)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
    sc.compute_stats(model._handle(), colsum, colnnz)
    cm = colsum / colnnz
    model.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    model.xi_u_global = float(torch.nansum(cm))
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(1)
    batch = {"counts": sc}
    ms, t6, parts, nnf = _timed_steps(model, batch, params, steps, warmup)
    d_bern = int(mask.sum())
    tf = 6.0 * rows * d_bern * max(32, K) / (t6[5] * 1e-3) / 1e12
    out = {"c5_ms_per_step": ms, "c5_steps_per_sec": 1e3 / ms, "c5_nnz": int(sc.nnz),
           "c5_kernel_ms": {"prep": round(t6[0], 4), "row_pass": round(t6[1], 4), "col_pass": round(t6[2], 4),
                            "finish": round(t6[3], 4), "dense": round(t6[5], 4)},
           "c5_dense_tflops_algorithmic": tf, "c5_dense_frac": tf / MFMA_F32_PEAK_TFLOPS,
           "c5_dense_frac_of": "algorithmic 6*B*D_bern*K against the f32-MFMA peak (157.3 TF/s)",
           "c5_n_nonfinite": float(nnf.sum()), "c5_elbo_x": float(parts["x"][0])}
    del model, sc, params
    torch.cuda.empty_cache()
    return out


def c4_extra(dev, steps=4, warmup=2):
    """BASELINE config 4 (the scRNA script's model: 500k x 30k counts at 3 %, log_transform decoder, K = 64)
    on this one GPU, for the `also` block of the default line, set up exactly as `--workload c4` does:
    ms per energy + gradient step, the kernel taps, the bf16x3 dense kernels on the algorithmic 6*B*D*K
    against the f32-MFMA peak (the same convention as C5).  After the headline's timed region."""
    import contextlib
    import torch
    from spmf_amd import PoissonFactorization, synth
    from spmf_amd.sparse import balanced_panel_rows
    rows, D, density, K, _desc = WORKLOADS["c4"]
    pr = balanced_panel_rows(rows, K)
    sc = synth.scrna_like(rows, D, dev, 20241218 + 4, first_chunk=0, panel_rows=pr, chunk_rows=25_000,
                          target_density=density)
    with contextlib.redirect_stdout(sys.stderr):
        model = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                     device=dev, panel_rows=pr, log_transform=True)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
    sc.compute_stats(model._handle(), colsum, colnnz)
    cmean = colsum / float(rows)              # bin/factorize_scrnaseq_counts.py:93-99, as in main()
    model.eta_i = torch.clamp(cmean, min=1e-3).reshape(1, D)
    model.xi_u_global = float(cmean.sum())
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(1)
    batch = {"counts": sc}
    ms, t6, parts, nnf = _timed_steps(model, batch, params, steps, warmup)
    tf = 6.0 * rows * D * K / (t6[5] * 1e-3) / 1e12
    # what the bf16 pipe itself issues: two launches x 88 MFMAs of 32x32x16 (6 + 5 partial products of
    # the three-way split operands) per 64 x 32 cells, against the dense bf16 datasheet peak
    tf_exec = 2.0 * 88 * 32768 / (64 * 32) * rows * D / (t6[5] * 1e-3) / 1e12
    out = {"c4_ms_per_step": ms, "c4_steps_per_sec": 1e3 / ms, "c4_nnz": int(sc.nnz),
           "c4_dense_executed_bf16_tflops": tf_exec, "c4_dense_executed_bf16_frac": tf_exec / MFMA_BF16_PEAK_TFLOPS,
           "c4_kernel_ms": {"prep": round(t6[0], 4), "row_pass": round(t6[1], 4), "col_pass": round(t6[2], 4),
                            "finish": round(t6[3], 4), "dense": round(t6[5], 4)},
           "c4_dense_tflops_algorithmic": tf, "c4_dense_frac": tf / MFMA_F32_PEAK_TFLOPS,
           "c4_dense_frac_of": "algorithmic 6*B*D*K against the f32-MFMA peak (157.3 TF/s); the kernels run on the "
                               "bf16 pipe with three-way split operands",
           "c4_saturated": float(model.last_saturated.sum()), "c4_n_nonfinite": float(nnf.sum()),
           "c4_elbo_x": float(parts["x"][0])}
    del model, sc, params
    torch.cuda.empty_cache()
    return out


def c2_extra(dev, steps=50, warmup=5):
    """BASELINE config 2 (100k x 5k at 1 %, K = 16, linear decoder) as `--workload c2` sets it up, for the
    `also` block of the default line: launch-latency-sized, ms per energy + gradient step."""
    import contextlib
    import torch
    from spmf_amd import PoissonFactorization, synth
    from spmf_amd.sparse import balanced_panel_rows
    rows, D, density, K, _desc = WORKLOADS["c2"]
    pr = balanced_panel_rows(rows, K)
    sc = synth.linear_structure(rows, D, density, dev, first_chunk=0, panel_rows=pr)
    with contextlib.redirect_stdout(sys.stderr):
        model = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                     device=dev, panel_rows=pr)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
    sc.compute_stats(model._handle(), colsum, colnnz)
    cm = colsum / colnnz
    model.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    model.xi_u_global = float(torch.nansum(cm))
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(1)
    batch = {"counts": sc}
    ms, t6, parts, nnf = _timed_steps(model, batch, params, steps, warmup)
    out = {"c2_ms_per_step": ms, "c2_steps_per_sec": 1e3 / ms, "c2_nnz": int(sc.nnz),
           "c2_kernel_ms": {"prep": round(t6[0], 4), "row_pass": round(t6[1], 4), "col_pass": round(t6[2], 4),
                            "finish": round(t6[3], 4)},
           "c2_n_nonfinite": float(nnf.sum()), "c2_elbo_x": float(parts["x"][0])}
    del model, sc, params
    torch.cuda.empty_cache()
    return out


def widek_extra(dev, K=128, steps=30, warmup=10):
    """Latent dimensions above 64 (the reference's latent_dim defaults to feature_dim, poisson.py:103-104): the C2
    matrix (100k x 5k at 1 %) at K = 128 on the whole-wave passes of csrc/widek.hip, ms per energy + gradient step."""
    import contextlib
    import torch
    from spmf_amd import PoissonFactorization, synth
    from spmf_amd.sparse import balanced_panel_rows
    rows, D, density, _K, _desc = WORKLOADS["c2"]
    pr = balanced_panel_rows(rows, K)
    sc = synth.linear_structure(rows, D, density, dev, first_chunk=0, panel_rows=pr)
    with contextlib.redirect_stdout(sys.stderr):
        model = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                     device=dev, panel_rows=pr)
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
    sc.compute_stats(model._handle(), colsum, colnnz)
    cm = colsum / colnnz
    model.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
    model.xi_u_global = float(torch.nansum(cm))
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(1)
    ms, t6, parts, nnf = _timed_steps(model, {"counts": sc}, params, steps, warmup)
    out = {f"k{K}_c2_ms_per_step": ms, f"k{K}_c2_kernel_ms": {"prep": round(t6[0], 4), "row_pass": round(t6[1], 4),
                                                              "col_pass": round(t6[2], 4), "finish": round(t6[3], 4)},
           f"k{K}_c2_n_nonfinite": float(nnf.sum()), f"k{K}_c2_elbo_x": float(parts["x"][0])}
    del model, sc, params
    torch.cuda.empty_cache()
    return out


def c1_gpu_extra(dev, steps=200, warmup=10):
    """BASELINE config 1 on the GPU, on the SAME seeded data as c1_dense_cpu_baseline (5000 x 200 dense
    Poisson(1) counts, K = 2, one batch of 5000): ms per energy + gradient step, and per whole VI step
    (what bin/factorize_csv.py runs per epoch at its defaults) replayed from its hipGraph."""
    import contextlib
    import numpy as np
    import torch
    from spmf_amd import PoissonFactorization, vi
    rng = np.random.default_rng(20241218 + 1)
    N, D, K = 5000, 200, 2
    x = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    with contextlib.redirect_stdout(sys.stderr):
        model = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (N * D) ** 0.5, device=dev)
        model.compute_scales(lambda: [{"counts": x}])
    from spmf_amd.sparse import SparseCounts, balanced_panel_rows
    batch = {"counts": SparseCounts.from_any(x, dev, balanced_panel_rows(N, K), latent_dim=K)}   # resident, as every workload here
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(1)
    ms, t6, parts, nnf = _timed_steps(model, batch, params, steps, warmup)
    opt = vi.AdamHIP(model, model.surrogate_distribution.trainable_variables, 0.1)
    opt.init_state(None)
    run = vi.StepRunner(model, opt, N, 1, use_graph=True)
    for _ in range(warmup):
        run.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step(batch)
    torch.cuda.synchronize()
    vi_ms = 1e3 * (time.perf_counter() - t0) / steps
    out = {"c1_gpu_ms_per_step": ms, "c1_gpu_steps_per_sec": 1e3 / ms, "c1_gpu_vi_step_ms": vi_ms,
           "c1_gpu_vi_graph_replays": run.replays, "c1_nnz": int(model._batch(batch)[0].nnz),
           "c1_gpu_elbo_x": float(parts["x"][0])}
    del model, run, opt
    torch.cuda.empty_cache()
    return out


def ref_harness_extra(dev, steps=100, warmup=10):
    """The shape of the reference's own harness (tests/spmf_test.py:12-43): D = 350 features, P = 50
    factors, dense Poisson(1) counts, batches of batch_size = 10 rows, sample_size = 20 draws per step.
    ms per energy + gradient evaluation of one such batch with S = 20 (all draws in ONE launch sequence:
    csrc/api.hip batched_draws), per whole VI step of it replayed from its hipGraph, and the same at a
    1000-row batch for scale."""
    import contextlib
    import numpy as np
    import torch
    from spmf_amd import PoissonFactorization, vi
    rng = np.random.default_rng(20241218 + 6)
    N, D, P, S = 5000, 350, 50, 20
    x = rng.poisson(1.0, size=(N, D)).astype(np.float64)
    with contextlib.redirect_stdout(sys.stderr):
        model = PoissonFactorization(latent_dim=P, feature_dim=D, u_tau_scale=1.0 / (N * D) ** 0.5, device=dev)
        model.compute_scales(lambda: [{"counts": x}])
    out = {"ref_harness_shape": f"D={D} P={P} S={S} dense Poisson(1) (tests/spmf_test.py:12-43)"}
    torch.manual_seed(20241218)
    params = model.surrogate_distribution.sample(S)
    from spmf_amd.sparse import SparseCounts, balanced_panel_rows
    for rows, tag in ((10, "b10"), (1000, "b1000")):
        batch = {"counts": SparseCounts.from_any(x[:rows], dev, balanced_panel_rows(rows, P))}
        ms, _, parts, _ = _timed_steps(model, batch, params, steps, warmup)
        out[f"ref_harness_{tag}_S20_ms"] = ms
        opt = vi.AdamHIP(model, model.surrogate_distribution.trainable_variables, 0.01)
        opt.init_state(None)
        run = vi.StepRunner(model, opt, N, S, use_graph=True)
        for _ in range(warmup):
            run.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run.step(batch)
        torch.cuda.synchronize()
        out[f"ref_harness_{tag}_S20_vi_step_ms"] = 1e3 * (time.perf_counter() - t0) / steps
        del run, opt
    del model
    torch.cuda.empty_cache()
    return out


def _pick_collective(model, lib_comm, dev, world, rank, S, want, all_reduce_):
    """-> (comm, transport string, record).  See the call site.  all_reduce_(tensor, op): torch.distributed's
    all-reduce of a device tensor (host-staged under gloo: the one-GPU rehearsal)."""
    import torch
    import torch.distributed as dist
    from spmf_amd import _lib
    from spmf_amd.dist import PeerComm
    lib, h = _lib.load(), model._handle()
    n = int(lib.spmf_acc_len(h, S))
    rec = {}
    transport_lib = "library RCCL communicator (spmf_allreduce on the step's stream)"

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        all_reduce_(t, dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def check_and_time(comm, reps=20):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        x = torch.randn(n, device=dev, generator=g)
        ref = x.clone()
        all_reduce_(ref, dist.ReduceOp.SUM)
        y = x.clone()
        comm.all_reduce_(y)
        torch.cuda.synchronize()
        good = bool(torch.isfinite(y).all()) and float((y - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        if not all_ok(good):
            return False, None
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            comm.all_reduce_(y)
        torch.cuda.synchronize()
        us = torch.tensor([1e6 * (time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        all_reduce_(us, dist.ReduceOp.MAX)
        # the buffer after the timed calls against the same number of torch.distributed all-reduces: every call of
        # the loop was a correct sum, not only the first (inbox parities, flags and the call counter all cycled)
        for _ in range(reps):
            all_reduce_(ref, dist.ReduceOp.SUM)
        good = bool(torch.isfinite(y).all()) and float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        if not all_ok(good):
            return False, float(us.item())
        return True, float(us.item())

    peer = None
    try:
        peer = PeerComm(model, max_draws=S)              # (collective inside; a local failure is agreed on)
    except Exception as e:
        rec["allreduce_p2p_error"] = str(e)[:160]
    if not all_ok(peer is not None):
        if peer is not None:
            peer.close(sync=False)
        rec.setdefault("allreduce_p2p_error", "unavailable on another rank")
        return lib_comm, transport_lib if lib_comm is not None else "torch.distributed", rec
    ok_p, us_p = check_and_time(peer)
    done, gave_up = peer.status()
    ok_p = all_ok(ok_p and gave_up == 0)
    rec["allreduce_p2p_ok"], rec["allreduce_p2p_us"] = ok_p, us_p
    us_l = None
    if lib_comm is not None:
        peer.enable(False)
        ok_l, us_l = check_and_time(lib_comm)
        rec["allreduce_rccl_ok"], rec["allreduce_rccl_us"] = ok_l, us_l
        peer.enable(True)
    use_p2p = ok_p and (want == "p2p" or us_l is None or us_p <= us_l)
    rec["allreduce_floats"] = n
    if use_p2p:
        return peer, ("library peer-pointer kernel (csrc/p2p.hip: direct reduce-scatter + all-gather over xGMI, "
                      "spmf_allreduce on the step's stream)"), rec
    peer.enable(False)
    return lib_comm, (transport_lib if lib_comm is not None else "torch.distributed"), rec


def main():
    # The contract is ONE JSON line on stdout.  Libraries below this script print there too
    # (RCCL writes a version banner from C when a communicator is created), so file
    # descriptor 1 is pointed at stderr for the whole run and the line goes to the real one.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=1, help="Monte-Carlo draws S per step")
    ap.add_argument("--rows", type=int, default=None, help="override total rows")
    ap.add_argument("--panel-rows", type=int, default=0,
                    help="rows per panel of the panel-CSC; 0 = spmf_amd.sparse.balanced_panel_rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the S=20 / minibatch lines")
    ap.add_argument("--split", action="store_true",
                    help="multi-rank: all-reduce the lower column half while the column pass "
                         "produces the upper one (off by default: on ONE rank it costs 56 us "
                         "per step; needs an 8-GPU measurement to decide)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from spmf_amd import PoissonFactorization, _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "torch.distributed.run --nproc-per-node N")
    # rehearsal on a one-GPU box: SPMF_BENCH_BACKEND=gloo SPMF_BENCH_ONE_GPU=1 puts every
    # rank on cuda:0 with host-staged collectives (same sharding/reducer/timing logic)
    backend = os.environ.get("SPMF_BENCH_BACKEND", "nccl")
    if os.environ.get("SPMF_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # launched by torch.distributed.run (also with one rank: exercises RCCL)
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce_(t, op=None):
        op = op or dist.ReduceOp.SUM
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h_ = t.cpu()
            dist.all_reduce(h_, op=op)
            t.copy_(h_)

    def broadcast_(t, src):
        if backend == "nccl":
            dist.broadcast(t, src)
        else:
            h_ = t.cpu()
            dist.broadcast(h_, src)
            t.copy_(h_)

    rows, D, density, K, desc = WORKLOADS[args.workload]
    if args.rows:
        rows = args.rows
    S = args.samples
    # strong scaling: contiguous shards of whole generator chunks
    chunk = {"c5": synth.MIXED_CHUNK_ROWS}.get(args.workload, synth.CHUNK_ROWS)
    if args.workload.startswith("c4"):
        chunk = 25_000
    nchunks = -(-rows // chunk)
    c0 = nchunks * rank // world
    c1 = nchunks * (rank + 1) // world
    my_rows = min(rows, c1 * chunk) - c0 * chunk
    if my_rows <= 0:
        raise SystemExit(f"{rows} rows are {nchunks} generator chunks of {chunk}: too few for "
                         f"{world} ranks")
    logt = args.workload.startswith("c4")
    mixed_mask = None
    from spmf_amd.sparse import balanced_panel_rows
    if not args.panel_rows:   # L2-sized panels, a multiple of 8 of them per shard (XCD balance)
        args.panel_rows = balanced_panel_rows(my_rows, K)
    if args.workload == "c5":
        # BASELINE.json config 5 names 4 GPUs: row shards like C3
        sc, mixed_mask = synth.mixed_c5(my_rows, D, dev, 20241218 + 5, panel_rows=args.panel_rows,
                                        first_chunk=c0, chunk_rows=chunk)
    elif logt:
        sc = synth.scrna_like(my_rows, D, dev, 20241218 + 4, first_chunk=c0,
                              panel_rows=args.panel_rows, chunk_rows=chunk,
                              target_density=density)
    else:
        # --split: gradient accumulators and work items in two column halves, so the
        # all-reduce of the lower half overlaps the upper half's column pass
        col_split = (D // 2) // 32 * 32 if (distributed and args.split) else 0
        sc = synth.linear_structure(my_rows, D, density, dev, first_chunk=c0,
                                    panel_rows=args.panel_rows, col_split=col_split)

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the class prints like the reference
        if mixed_mask is not None:
            from spmf_amd import MixedFactorization
            model = MixedFactorization(mixed_mask, latent_dim=K, feature_dim=D,
                                       u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                       panel_rows=args.panel_rows)
        else:
            model = PoissonFactorization(latent_dim=K, feature_dim=D,
                                         u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev,
                                         panel_rows=args.panel_rows, log_transform=logt)
    if getattr(sc, "col_split", 0):
        model.enable_column_split(sc.col_split)
    # compute_scales (poisson.py:113-154) over all shards: one pre-pass + all-reduce
    colsum = torch.zeros(D, dtype=torch.float64, device=dev)
    colnnz = torch.zeros(D, dtype=torch.float64, device=dev)
    sc.compute_stats(model._handle(), colsum, colnnz)
    tot = torch.tensor([float(sc.n_rows), float(sc.row_lgamma.sum()), float(sc.nnz)],
                       dtype=torch.float64, device=dev)
    if distributed:
        for t in (colsum, colnnz, tot):
            all_reduce_(t)
    cm = colsum / colnnz
    if logt:
        # bin/factorize_scrnaseq_counts.py:93-99: column_norms = plain column means
        # (floored: a zero mean would divide by zero in g(x) = log(x/eta + 1))
        model.eta_i = (colsum / float(tot[0])).clamp_min(1e-3).reshape(1, D)
        model.xi_u_global = float(torch.nansum(colsum / float(tot[0])))
    else:
        model.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
        model.xi_u_global = float(torch.nansum(cm))
    rows_g, lgam_g, nnz_g = int(tot[0]), float(tot[1]), int(tot[2])

    # parameters: S draws from the surrogate at its initial values (poisson.py:403-539)
    gen_seed = 20241218
    torch.manual_seed(gen_seed)
    params = model.surrogate_distribution.sample(S)
    if distributed:
        for n in _lib.VAR_ORDER:            # replicate rank 0's draw
            broadcast_(params[n], 0)
    batch = {"counts": sc}

    hook = None
    transport = None
    comm_pick = {}
    if distributed:
        from spmf_amd.dist import LibraryComm, ShardReducer
        # transport of the step's one collective: the library's own RCCL communicator
        # (spmf_allreduce: ncclAllReduce on the step's stream, 3.6 us against 15 us per one-rank
        # call through torch.distributed, and capturable in the VI step's hipGraph).
        # SPMF_BENCH_COMM=torch selects torch.distributed's; the gloo rehearsal always uses it.
        comm = None
        transport = f"torch.distributed ({backend})"
        comm_pick = {}
        want = os.environ.get("SPMF_BENCH_COMM", "auto")       # auto | p2p | lib | torch
        if want != "torch" and backend == "nccl":
            try:
                comm = LibraryComm(model)
                transport = "library RCCL communicator (spmf_allreduce on the step's stream)"
            except Exception as e:        # librccl not loadable from the library: every rank lands here alike
                print(f"LibraryComm unavailable ({e}); torch.distributed moves the accumulators", file=sys.stderr)
                comm = None
        # The library's own kernel over peer pointers (csrc/p2p.hip: direct reduce-scatter + all-gather,
        # every pair of ranks on its own xGMI link) against RCCL, ON THIS NODE: both are checked against
        # torch.distributed's all-reduce of the same buffer and timed back to back on the real payload; the
        # faster one that passed carries the step.  Every decision is agreed over all ranks first.
        # (gloo rehearsal on one GPU: only with SPMF_BENCH_COMM=p2p -- the ranks are processes on one card)
        if world > 1 and ((want in ("auto", "p2p") and backend == "nccl") or want == "p2p"):
            comm, transport, comm_pick = _pick_collective(model, comm, dev, world, rank, S, want, all_reduce_)
        hook = ShardReducer(comm=comm)
        hook.set_batch_totals(rows_g, lgam_g)

    lib, h = _lib.load(), model._handle()

    def step():
        return model.energy_and_grads(batch, params, all_reduce=hook)

    for _ in range(args.warmup):
        parts, grads, nnf = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    lib.spmf_ctx_enable_timing(h, 1)       # hipEvents between kernels, no syncs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parts, grads, nnf = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    import ctypes as C
    sat0 = float(model.last_saturated.sum())     # saturation events of the last timed step
    ms5 = (C.c_float * 6)()
    _lib.check(h, lib.spmf_last_timing(h, ms5), "spmf_last_timing")
    lib.spmf_ctx_enable_timing(h, 0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        all_reduce_(tmax, dist.ReduceOp.MAX)
    dt = float(tmax[0])

    # extras SURVEY 8(d) asks to report beside the headline (1 GPU only, not `value`):
    # S=20 draws per step (tests/spmf_test.py:39), a B~20000-row minibatch, the 125k-row
    # shard of the 8-GPU configuration with its N-independent part, a one-rank RCCL
    # all-reduce of the packed accumulators, and the step again after 50 seeded Adam steps
    extras = {}

    def timed(fn, n, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t_) / n

    import ctypes as C
    if world == 1 and not args.no_extras and not logt and mixed_mask is None:
        torch.manual_seed(gen_seed)
        p20 = model.surrogate_distribution.sample(20)
        extras["S20_ms_per_step"] = timed(lambda: model.energy_and_grads(batch, p20), 2, 1)
        del p20
        npan = max(1, min(sc.n_panels, -(-20000 // sc.panel_rows)))
        mb = {"counts": sc, "panels": (0, npan)}
        extras["minibatch_rows"] = min(sc.n_rows, npan * sc.panel_rows)
        extras["minibatch_ms_per_step"] = timed(lambda: model.energy_and_grads(mb, params), 50, 5)
        # the per-GPU shard of the 8-GPU run (122 880 of its 125 000 rows, as in rounds 1-2) on this
        # one GPU: its step, the kernel taps, and what does not shrink with N (step - row - col)
        # (a rank of the 8-GPU run lays out its OWN rows: panels sized and counted for 125k rows)
        from spmf_amd.sparse import SparseCounts
        n_s = min(sc.n_rows, 122_880)
        e_s = int(sc.row_ptr[n_s])
        ss = SparseCounts(sc.row_ptr[:n_s + 1].clone(), sc.col_idx[:e_s].clone(), sc.val[:e_s].clone(),
                          n_s, D, balanced_panel_rows(n_s, K))
        ss.compute_stats(h)
        npan_s = ss.n_panels
        sb = {"counts": ss}
        # (first without the hipEvent taps: five event records per step are ~20 us of a 0.4 ms step)
        extras["shard125k_ms_per_step_no_taps"] = timed(lambda: model.energy_and_grads(sb, params), 50, 5)
        lib.spmf_ctx_enable_timing(h, 1)
        sh_ms = timed(lambda: model.energy_and_grads(sb, params), 50, 5)
        ms_s = (C.c_float * 6)()
        _lib.check(h, lib.spmf_last_timing(h, ms_s), "spmf_last_timing")
        lib.spmf_ctx_enable_timing(h, 0)
        extras["shard125k_rows"] = n_s
        extras["shard125k_panel_rows"] = ss.panel_rows
        extras["shard125k_ms_per_step"] = sh_ms
        extras["shard125k_kernel_ms"] = {"prep": round(ms_s[0], 4), "row_pass": round(ms_s[1], 4),
                                         "col_pass": round(ms_s[2], 4), "finish": round(ms_s[3], 4)}
        extras["shard125k_fixed_us"] = 1e3 * (sh_ms - ms_s[1] - ms_s[2])
        # one-rank RCCL all-reduce of the packed accumulators through the library's own
        # communicator (the transport of SPMF_BENCH_COMM=lib), back to back on the stream
        try:
            from spmf_amd.dist import LibraryComm
            comm1 = LibraryComm(model, rank=0, world=1)
            n_acc = int(lib.spmf_acc_len(h, 1))
            accv = torch.zeros(n_acc, dtype=torch.float32, device=dev)
            extras["allreduce_1rank_us"] = 1e3 * timed(lambda: comm1.all_reduce_(accv), 200, 20)
            extras["allreduce_floats"] = n_acc
            # the multi-GPU step's critical path on one rank: data pass, prior half of the
            # finish on the side stream, spmf_allreduce, data half of the finish
            from spmf_amd.dist import ShardReducer
            red1 = ShardReducer(comm=comm1)          # one rank: prior half NOT forked (overlap_prior)
            red1.set_batch_totals(extras["shard125k_rows"],
                                  float(sc.row_lgamma[:extras["shard125k_rows"]].sum()))
            extras["shard125k_1rank_rccl_ms_per_step"] = timed(
                lambda: model.energy_and_grads(sb, params, all_reduce=red1), 50, 5)
            # the same with the prior half of the finish forked to the side stream under the
            # collective, as ranks of a world > 1 run it (what the fork/join costs on one rank)
            red1f = ShardReducer(comm=comm1, overlap_prior=True)
            red1f.set_batch_totals(red1.rows_global, red1.lgamma_global)
            extras["shard125k_1rank_rccl_forked_prior_ms_per_step"] = timed(
                lambda: model.energy_and_grads(sb, params, all_reduce=red1f), 50, 5)
            # the whole row-sharded VI step of that shard on the device (vi.vi_step_dev with the
            # reducer: noise, surrogate, data pass, spmf_allreduce, finish, gate, chain rule +
            # Adam; no host read-back), eager and replayed from its hipGraph
            from spmf_amd import vi as _vi
            sur = model.surrogate_distribution
            saved = [p_.detach().clone() for p_ in sur.trainable_variables]
            for key, force_graph, red in (("shard125k_vi_1rank_rccl_ms", False, red1),
                                          ("shard125k_vi_1rank_rccl_graph_ms", True, red1),
                                          ("shard125k_vi_no_comm_ms", False, None),
                                          # SPMF_VI_OVERLAP=1 (off by default): the scale hierarchy's draws / transform /
                                          # prior half on a side stream behind the row pass (vi.vi_step_dev)
                                          ("shard125k_vi_hierarchy_on_side_stream_ms", False, None)):
                os.environ["SPMF_VI_OVERLAP"] = "1" if "side_stream" in key else "0"
                o_ = _vi.AdamHIP(model, sur.trainable_variables, 1e-3)
                o_.init_state(3.0)
                # default: what fit() runs (StepRunner replays a hipGraph for launch-bound batches and
                # runs GPU-bound ones like this shard eagerly); forced: the captured step incl. the collective
                run_ = _vi.StepRunner(model, o_, rows_g, S, use_graph=True, all_reduce=red)
                if force_graph:
                    run_.graph_max_nnz = 1 << 62
                extras[key] = timed(lambda: run_.step(sb), 50, 5)
                extras[key.replace("_ms", "_replays")] = run_.replays
                del run_, o_
                with torch.no_grad():
                    for p_, q_ in zip(sur.trainable_variables, saved):
                        p_.copy_(q_)
            del saved
            os.environ.pop("SPMF_VI_OVERLAP", None)
            _lib.check(h, lib.spmf_comm_destroy(h), "spmf_comm_destroy")
        except Exception as e:                      # no librccl on the box: report, do not fail
            extras["allreduce_1rank_us"] = None
            extras["allreduce_error"] = str(e)[:120]
    if world == 1 and not args.no_extras and not logt and mixed_mask is None:
        # the deterministic mode (spmf_ctx_set_deterministic) on the same resident workload: ms per step,
        # the kernel taps, and whether two steps give the same bits (parts and all gradients)
        try:
            with contextlib.redirect_stdout(sys.stderr):
                mdet = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5,
                                            device=dev, panel_rows=args.panel_rows, deterministic=True)
            mdet.eta_i, mdet.xi_u_global = model.eta_i, model.xi_u_global
            pa, ga, _ = mdet.energy_and_grads(batch, params)
            pa = {k: v.clone() for k, v in pa.items()}
            ga = {k: v.clone() for k, v in ga.items()}
            pb, gb, _ = mdet.energy_and_grads(batch, params)
            same = all(torch.equal(pa[k], pb[k]) for k in pa) and all(torch.equal(ga[k], gb[k]) for k in ga)
            pd_, gd_, _ = model.energy_and_grads(batch, params)
            dmax = max(float((ga[k] - gd_[k]).abs().max()) / max(float(gd_[k].abs().max()), 1e-30) for k in ga)
            extras["det_vs_default_differing_entries"] = int(sum(int((ga[k] != gd_[k]).sum()) for k in ga))
            det_ms, t6d, _, _ = _timed_steps(mdet, batch, params, max(3, min(args.steps, 10)), 2)
            if "sb" in locals():       # the 8-GPU shard in the deterministic mode too
                extras["shard125k_det_ms_per_step"] = timed(lambda: mdet.energy_and_grads(sb, params), 50, 5)
            extras["det_ms_per_step"] = det_ms
            extras["det_kernel_ms"] = {"prep": round(t6d[0], 4), "row_pass": round(t6d[1], 4),
                                       "col_pass_and_reduce": round(t6d[2], 4), "finish": round(t6d[3], 4)}
            extras["det_bit_identical_repeat"] = bool(same)
            extras["det_vs_default_grad_max_rel"] = dmax
            del mdet, pa, ga, pb, gb, pd_, gd_
        except Exception as e:
            extras["det_error"] = str(e)[:200]
    if world == 1 and not args.no_extras:
        # "measure a copy kernel for the attainable HBM ceiling and report both": one
        # 1 GiB device-to-device copy (read + write bytes), beside the 8 TB/s nominal peak
        try:
            src_ = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            dst_ = torch.empty_like(src_)
            cp_ms = timed(lambda: dst_.copy_(src_), 10, 3)
            extras["hbm_copy_gbps"] = 2 * src_.numel() * 4 / (cp_ms * 1e-3) / 1e9
            del src_, dst_
        except Exception as e:
            extras["hbm_copy_error"] = str(e)[:120]
    if world == 1 and not args.no_extras:
        # the data-format step in front of the path: the device layout of this whole
        # workload built again from its resident CSR arrays by the library (csrc/layout.hip), and the
        # statistics pre-pass (row sums + column sums of compute_scales)
        from spmf_amd.sparse import SparseCounts as _SC

        def _build():
            return _SC(sc.row_ptr, sc.col_idx, sc.val, sc.n_rows, sc.n_cols, sc.panel_rows)
        try:
            sc2 = _build()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sc2 = _build()
            torch.cuda.synchronize()
            extras["layout_build_ms"] = 1e3 * (time.perf_counter() - t0) / 3
            extras["layout_native"] = bool(sc2.native_layout)
            cs2 = torch.zeros(D, dtype=torch.float64, device=dev)
            sc2.compute_stats(model._handle(), cs2, torch.zeros_like(cs2))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sc2.compute_stats(model._handle(), cs2, torch.zeros_like(cs2))
            torch.cuda.synchronize()
            extras["stats_ms"] = 1e3 * (time.perf_counter() - t0) / 3
            del sc2, cs2
        except Exception as e:
            extras["layout_error"] = str(e)[:200]
    if world == 1 and not args.no_extras and args.workload == "c3":
        # a second BASELINE config in the driver-timed record
        try:
            extras.update(c5_extra(dev))
        except Exception as e:
            extras["c5_error"] = str(e)[:200]
        try:
            extras.update(c2_extra(dev))
        except Exception as e:
            extras["c2_error"] = str(e)[:200]
        try:   # K above 64 (VERDICT r4 missing #6): C2's matrix at K = 128
            extras.update(widek_extra(dev))
        except Exception as e:
            extras["k128_error"] = str(e)[:200]
        # the reference's own shapes: C1 on the GPU beside its CPU number, and the harness
        # of tests/spmf_test.py (D = 350, P = 50, batches of 10 rows, sample_size = 20)
        try:
            extras.update(c1_gpu_extra(dev))
        except Exception as e:
            extras["c1_gpu_error"] = str(e)[:200]
        try:
            extras.update(ref_harness_extra(dev))
        except Exception as e:
            extras["ref_harness_error"] = str(e)[:200]
        # ... and a third (C4: seconds to generate on the device, 15 GB resident)
        if os.environ.get("SPMF_BENCH_C4_EXTRA", "1") != "0":
            try:
                extras.update(c4_extra(dev))
            except Exception as e:
                extras["c4_error"] = str(e)[:200]
    if world == 1 and not args.no_extras:
        # "surrogate at its init values and after 50 seeded Adam steps" -- the same
        # energy + gradient step timed again at draws from the trained surrogate
        from spmf_amd import vi as _vi
        opt50 = _vi.AdamHIP(model, model.surrogate_distribution.trainable_variables, 1e-2)
        opt50.init_state(10.0)
        torch.manual_seed(gen_seed + 50)
        for _ in range(50):
            _vi.vi_step_dev(model, opt50, batch, rows_g, S)
        st50 = opt50.read_state()
        p50 = model.surrogate_distribution.sample(S)
        extras["after50_ms_per_step"] = timed(lambda: model.energy_and_grads(batch, p50),
                                              max(3, min(args.steps, 20)), 2)
        extras["after50_saturated"] = float(model.last_saturated.sum())
        extras["after50_steps_applied"] = int(st50[11])
        extras["after50_loss"] = st50[8]
        del p50

    # extra (not the contract's `value`): the whole VI step -- base noise, surrogate
    # transform + log q, energy + gradient, chain to the trainables, Adam -- per step
    from spmf_amd import vi
    opt = vi.AdamHIP(model, model.surrogate_distribution.trainable_variables, 1e-3)
    opt.init_state(3.0)
    n_vi = max(2, min(5, args.steps))
    vi_seed = 20241218 + 77          # row shards replicate the surrogate: every rank draws the same noise
    for it in range(n_vi + 1):
        if it == 1:
            torch.cuda.synchronize()
            tv = time.perf_counter()
        # what fit() runs, one rank or many: no host read-back (row shards: the reducer's
        # all-reduce sits inside the device-gated step)
        vi.vi_step_dev(model, opt, batch, rows_g, S, all_reduce=hook,
                       seed=vi_seed if hook is not None else None)
    torch.cuda.synchronize()
    vi_ms = 1e3 * (time.perf_counter() - tv) / n_vi

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        value = args.steps / dt
        # roofline of the dominant kernel of THIS rank's shard
        b_row, b_col, b_tot = algorithmic_bytes(sc.nnz, sc.n_rows, D, K, S)
        kern = {"prep": ms5[0], "row_pass": ms5[1], "col_pass": ms5[2], "finish": ms5[3],
                "dense_expdot": ms5[5]}
        KD = max(32, K)
        # algorithmic count: 6*B*D*K -- X = Z W^T once, then the two gradient
        # products.  The kernels execute exactly that: E (exp, or the sigmoid of the Bernoulli
        # logits) is kept in HBM between the two contractions; SPMF_DENSE_E_ONCE=0 selects the
        # form that computes it in both launches (8*B*D*K executed)
        # (mixed likelihood: the dense sums run over the Bernoulli columns only)
        D_dense = int(mixed_mask.sum()) if mixed_mask is not None else D
        dense_flops = 6.0 * sc.n_rows * D_dense * KD
        e_once = os.environ.get("SPMF_DENSE_E_ONCE", "1")[:1] != "0"
        dense_flops_executed = (6.0 if e_once else 8.0) * sc.n_rows * D_dense * KD
        # K = 64 Poisson log_transform: the bf16x3 kernel (csrc/dense3.hip) unless switched off:
        # two launches x 88 bf16 MFMAs of 32x32x16 per 64 x 32 cells (6 + 5 partial products)
        bf16x3 = (logt and mixed_mask is None and KD == 64
                  and os.environ.get("SPMF_DENSE_BF16X3", "1")[:1] != "0")
        if bf16x3:
            dense_flops_executed = 2.0 * 88 * 32768 / (64 * 32) * sc.n_rows * D_dense
        if (logt or mixed_mask is not None) and ms5[5] >= max(ms5[1], ms5[2]):
            dom = "dense_expdot"
            achieved = dense_flops / (ms5[5] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": achieved,
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                    "executed_tflops": dense_flops_executed / (ms5[5] * 1e-3) / 1e12}
            if bf16x3:
                # achieved / peak / frac stay the ALGORITHMIC 6*B*D*K against the f32-MFMA peak
                # (no credit for a bf16 peak; > 1 means the work left that pipe); what the bf16 pipe
                # itself issues, against its dense datasheet peak, is reported beside it
                roof["executed_pipe"] = "bf16 (three-way split operands, csrc/dense3.hip)"
                roof["executed_peak"] = MFMA_BF16_PEAK_TFLOPS
                roof["executed_frac"] = roof["executed_tflops"] / MFMA_BF16_PEAK_TFLOPS
        else:
            dom = "col_pass" if ms5[2] >= ms5[1] else "row_pass"
            dom_bytes = (b_col if dom == "col_pass" else b_row) / S   # taps cover one draw
            achieved = dom_bytes / (kern[dom] * 1e-3) / 1e9
            roof = None
        # roofline.traffic: counter-measured bytes beyond L2 per launch of the dominant
        # kernel (tools/pmc.sh + tools/pmc_traffic.py), quoted only while the kernel
        # sources are the ones the counters were collected on
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from pmc_traffic import kernels_sha
                doc = json.load(open(pmc))
                # (collected on the workload at its named size and one GPU: not quoted for --rows runs or shards)
                if doc.get("_kernels_sha") == kernels_sha() and args.rows is None and world == 1:
                    traffic = doc.get(args.workload, {}).get(dom, {}).get("traffic_bytes")
            except Exception:
                traffic = None
        if roof is not None:
            roof["traffic"] = traffic
        # second roofline object (not the headline): the sparse passes against the
        # MEASURED ceiling of their own access pattern -- one 4*KP-byte table row per
        # stored entry and table, gathered into registers out of L2-resident tables
        # (tools/gather_ceiling.hip -> profiles/gather_ceiling.json)
        roof_l2 = None
        gc = os.path.join(ROOT, "profiles", "gather_ceiling.json")
        if roof is None and os.path.exists(gc):
            try:
                ceil = json.load(open(gc))
                rowb = 4 * (4 if K <= 4 else 1 << (K - 1).bit_length())
                key = f"{rowb}B_2tables_{'5.12' if dom == 'row_pass' else '2.56'}MB"
                peak = ceil["best_tbps"][key]
                gathered = 2.0 * sc.nnz * rowb                     # two table rows per stored entry
                ach = gathered / (kern[dom] * 1e-3) / 1e12
                roof_l2 = {"bound": "l2-gather", "kernel": dom, "achieved": ach, "peak": peak,
                           "unit": "TB/s", "frac": ach / peak, "peak_source": key,
                           "guide_l2_peak": 34.5}
            except Exception:
                roof_l2 = None
        # ... and what the counters say about that kernel's request path (measured once, round 5, on this
        # workload; a record riding along, not re-measured by this run)
        rp = os.path.join(ROOT, "profiles", "pmc_request_path.json")
        if roof_l2 is not None and os.path.exists(rp) and args.workload == "c3" and args.rows is None:
            try:
                doc = json.load(open(rp))
                roof_l2["request_path"] = dict(doc.get(dom, {}), guide_l2_bytes_per_clk_cu=doc.get("guide_l2_bytes_per_clk_cu"),
                                               flat_probe_ceiling_requests_per_clk_cu=doc.get("flat_probe_ceiling_requests_per_clk_cu"),
                                               source=doc.get("_source"))
            except Exception:
                pass
        _, _, b_tot_g = algorithmic_bytes(nnz_g, rows_g, D, K, S)
        out = {
            "metric": "elbo_steps_per_sec", "value": value, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc + f", S={S}, full batch, row-sharded x{world}",
                       "rows": rows_g, "cols": D, "nnz": nnz_g, "latent_dim": K,
                       "samples": S, "parallelism": f"row-shard dp{world}", "allreduce_transport": transport,
                       # what one timed step is: the energy step; the whole VI step is vi_step_ms
                       "step": "energy + gradient: all 14 energy parts and d/d(12 latent variables) of one batch "
                               "(spmf_step_begin .. [all-reduce] .. spmf_step_end); the surrogate's sampler / "
                               "transform / chain rule / Adam around it are NOT in `value`: the whole VI step is "
                               "vi_step_ms",
                       "panel_rows": args.panel_rows,
                       # physical entry streams (the algorithmic bytes above stay canonical: 8 B per entry and pass)
                       "entry_format": ("packed u32: col<<16|count (row pass), row-in-panel<<16|count (column pass)"
                                        if getattr(sc, "ent", None) is not None and getattr(sc, "pc_ent", None) is not None
                                        else "canonical int32 index + f32 value")},
            "achieved_hbm_gbps_step": b_tot_g / world / (ms_step * 1e-3) / 1e9,
            "frac_hbm_roofline_step": b_tot_g / world / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_step": b_tot_g,
            # the same fraction against the copy rate measured on this box (also.hbm_copy_gbps), when it was
            "frac_hbm_measured_copy_step": (b_tot_g / world / (ms_step * 1e-3) / 1e9 / extras["hbm_copy_gbps"]
                                            if extras.get("hbm_copy_gbps") else None),
            "kernel_ms": {k: round(float(v), 4) for k, v in kern.items()},
            "roofline": roof or {"bound": "hbm", "kernel": dom, "achieved": achieved,
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic},
            "roofline_l2": roof_l2,
            "vi_step_ms": vi_ms,
            "collective": (comm_pick if distributed else None),
            "also": extras,
            "n_nonfinite": float(nnf.sum()),
            "saturated": sat0,
            "elbo_x": float(parts["x"][0]),
        }
        if not args.no_cpu_baseline and world == 1 and not logt and mixed_mask is None:
            n_s, nnz_s, t_s, parity, cores = cpu_baseline(sc, model, params, K)
            out["cpu_baseline"] = {
                "value": 1.0 / (t_s * nnz_g / nnz_s), "unit": "steps/s", "cores": cores,
                "kind": "port", "parity_vs_port": parity,
                "shard_steps_per_sec": 1.0 / t_s,
                "sample": f"one whole shard = first {n_s} rows ({nnz_s} nnz) of the same matrix: "
                          f"fp64 OpenMP C port (oracle/sparse_exact_omp.c) on {cores} threads "
                          f"(fastest of a sweep up to all {os.cpu_count()} host cores), "
                          f"median of 10 after 3 warm-ups = {t_s:.4f} s per shard step "
                          f"(shard_steps_per_sec); value = that rate scaled by nnz to the "
                          f"full workload ({nnz_g / nnz_s:.2f} shards)"}
            c1, c1_cores = c1_dense_cpu_baseline()
            out["also"]["c1_dense_fp64_cpu_steps_per_sec"] = c1
            out["also"]["c1_dense_fp64_cpu_threads"] = c1_cores
        else:
            out["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
