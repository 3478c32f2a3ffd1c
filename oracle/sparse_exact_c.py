"""ctypes front end of oracle/sparse_exact_omp.c -- TEST INFRASTRUCTURE ONLY
(PARITY UNPINNED, see oracle/spmf_oracle.py).

The multithreaded fp64 C restatement of the sparse-exact data term
(oracle/sparse_exact.py, itself pinned to the dense oracle of
mederrata_spmf/poisson.py:156-184,582-701).  Two uses: a second, independent
checker for the HIP path at sizes where the numpy port is slow, and the
``cpu_baseline`` leg of bench.py (kind "port") on all host cores.
``__graft_entry__.build()`` compiles it; outputs go to oracle/_build/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "sparse_exact_omp.c")
OUT_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(OUT_DIR, "libsparse_exact_omp.so")
HALF_LOG_2_OVER_PI = 0.5 * math.log(2.0 / math.pi)

_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", SRC,
                               "-o", LIB, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB)
        lib.spx_max_threads.restype = C.c_int
        lib.spx_set_threads.argtypes = [C.c_int]
        lib.spx_data_term.restype = C.c_int
        lib.spx_data_term.argtypes = [C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 17
        lib.spx_grad_pieces.restype = C.c_int
        lib.spx_grad_pieces.argtypes = [C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 10
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Prepared:
    """One row shard laid out for the C port: CSR + a CSC copy (built once, like
    the GPU path's resident panel-CSC), fp64 values, row scales."""

    def __init__(self, X, eta, xi_global, scale_rows):
        X = sp.csr_matrix(X).astype(np.float64)
        X.sort_indices()
        self.B, self.D = X.shape
        self.row_ptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
        self.col = np.ascontiguousarray(X.indices, dtype=np.int32)
        self.val = np.ascontiguousarray(X.data, dtype=np.float64)
        Xc = X.tocsc()
        Xc.sort_indices()
        self.csc_ptr = np.ascontiguousarray(Xc.indptr, dtype=np.int32)
        self.csc_row = np.ascontiguousarray(Xc.indices, dtype=np.int32)
        self.csc_val = np.ascontiguousarray(Xc.data, dtype=np.float64)
        self.eta = np.ascontiguousarray(
            np.broadcast_to(np.asarray(eta, dtype=np.float64).reshape(-1), (self.D,)))
        rowsum = np.asarray(X.sum(1)).reshape(self.B)
        self.xi = np.ascontiguousarray(rowsum / float(xi_global)) if scale_rows else None

    def step(self, u, v, w, s, scales=False):
        """'x', 'z' and d(x+z)/d(u,v,w,s) of one draw: u [D,K], v [K,D], w [1,D], s [2,D].
        ``scales``: also return the entry-wise yardstick of the gradient comparison under
        "scales" -- per entry the sum over the three additive pieces of the data term (stored-cell
        part, minus-rate part, z prior) of |d piece / d entry|, the definition of
        oracle.spmf_oracle.energy_grad_scales(prior=False) (pinned to it in tests/test_oracle.py)."""
        lib = load()
        B, D = self.B, self.D
        K = u.shape[1]
        eta = self.eta
        u = np.asarray(u, dtype=np.float64)
        v = np.asarray(v, dtype=np.float64)
        w = np.asarray(w, dtype=np.float64).reshape(D)
        s = np.asarray(s, dtype=np.float64)
        T = s[0] + s[1]
        w1, w2 = s[0] / T, s[1] / T
        Ap = np.ascontiguousarray((w1 / eta)[:, None] * u)        # poisson.py:652-666, g(x)=x/eta folded
        Vp = np.ascontiguousarray((v * eta[None, :]).T)           # f(y)=y*eta folded, [D,K]
        phi = np.ascontiguousarray(eta * w2 * w)                  # poisson.py:680-701
        z = np.empty((B, K))
        gz = np.empty((B, K))
        gAp = np.empty((D, K))
        gVp = np.empty((D, K))
        gphi = np.empty(D)
        zsum = np.empty(K)
        sc = np.zeros(4)
        rc = lib.spx_data_term(B, D, K, _p(self.row_ptr), _p(self.col), _p(self.val), _p(self.csc_ptr),
                               _p(self.csc_row), _p(self.csc_val), _p(self.xi), _p(Ap), _p(Vp), _p(phi),
                               _p(z), _p(gz), _p(gAp), _p(gVp), _p(gphi), _p(zsum), _p(sc))
        if rc != 0:
            raise RuntimeError(f"spx_data_term failed: {rc}")
        veta = Vp.sum(0)
        sum_r = zsum @ veta + B * phi.sum()
        part_x = sc[0] - sc[3] - sum_r
        part_z = B * K * HALF_LOG_2_OVER_PI - 0.5 * sc[1]
        gVp = gVp - zsum[None, :]
        dphi = gphi - B
        gA = gAp / eta[:, None]
        gv = (gVp * eta[:, None]).T
        gu = w1[:, None] * gA
        gw = (eta * w2 * dphi)[None, :]
        GA = (u * gA).sum(1)
        Gphi = eta * w * dphi
        gs = np.stack([(GA - Gphi) * s[1] / T ** 2, (Gphi - GA) * s[0] / T ** 2])
        out = {"x": part_x, "z": part_z, "n_nonfinite": int(sc[2]),
               "grads": {"u": gu, "v": gv, "w": gw, "s": gs}, "z_rows": z, "gz_rows": gz}
        if scales:
            gA_pos = np.empty((D, K))
            gA_zp = np.empty((D, K))
            sxx = np.empty(D)
            rc = lib.spx_grad_pieces(B, D, K, _p(self.csc_ptr), _p(self.csc_row), _p(self.csc_val),
                                     _p(self.xi), _p(z), _p(gz), _p(veta), _p(gA_pos), _p(gA_zp), _p(sxx))
            if rc != 0:
                raise RuntimeError(f"spx_grad_pieces failed: {rc}")
            zero_dk, zero_d = np.zeros((D, K)), np.zeros(D)
            pieces = [   # (d piece / dA', d piece / dV', d piece / dphi)
                (gA_pos, gVp + zsum[None, :], gphi),                              # stored cells: x log r
                (-sxx[:, None] * veta[None, :], np.broadcast_to(-zsum, (D, K)), np.full(D, -float(B))),  # -sum r
                (gA_zp, zero_dk, zero_d)]                                         # z prior
            acc = {"u": np.zeros((D, K)), "v": np.zeros((K, D)), "w": np.zeros((1, D)), "s": np.zeros((2, D))}
            for pA, pV, pphi in pieces:
                a = pA / eta[:, None]
                acc["u"] += np.abs(w1[:, None] * a)
                acc["v"] += np.abs((pV * eta[:, None]).T)
                acc["w"] += np.abs(eta * w2 * pphi)[None, :]
                ga_ = (u * a).sum(1)
                gp_ = eta * w * pphi
                acc["s"] += np.abs(np.stack([(ga_ - gp_) * s[1] / T ** 2, (gp_ - ga_) * s[0] / T ** 2]))
            out["scales"] = acc
        return out


def data_term(X, eta, xi_global, scale_rows, u, v, w, s):
    """Same contract as oracle.sparse_exact.data_term."""
    return Prepared(X, eta, xi_global, scale_rows).step(u, v, w, s)


def set_threads(n):
    load().spx_set_threads(int(n))


def max_threads():
    return int(load().spx_max_threads())
