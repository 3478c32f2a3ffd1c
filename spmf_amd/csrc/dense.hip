// dense.hip -- the dense part of the log_transform decoder on the f32 matrix
// cores (gfx950).
//
// With f(y) = exp(y*eta) - 1 (mederrata_spmf/poisson.py:52-53) the sum of the
// Poisson rate over ALL B*D cells has no closed form:
//   sum_all r   = sum_{b,d} E_bd - B*D + B*sum_d phi_d,   E_bd = exp(<z_b, W_d>), W_d = eta_d v_d
//   d/dz_b     -= sum_d E_bd W_d          ("GZ dense")
//   d/dW_d     -= sum_b E_bd z_b          ("GW dense")
// Both are the same operator with the operands swapped:
//   expdot(P, Q):  out_p[k] = sum_q exp(<P_p, Q_q>) Q_q[k],   esum = sum_{p,q} exp(<P_p, Q_q>)
// GZ = expdot(Z, W), GW = expdot(W, Z).  The stored cells' x*log r term and its
// gradient stay in the sparse row/column passes.
//
// Kernel: P-stationary.  A wave owns 32 rows of P as the B operand of
//   X = Q_tile * P_tile^T      (v_mfma_f32_32x32x2_f32, KD/2 steps)
// so X has the P row on the lane and the Q row in the 16 accumulator
// registers; E = exp(X) is then ALREADY the B operand of the second product
//   out^T[k, p] += sum_q Q^T[k, q] E[q, p]      (16 steps per 32 features)
// (an accumulator tile feeds a following MFMA that sums over its row index with
// no lane movement; the k order of a step is the accumulator's row order
// rho_h(t) = (t&3) + 8(t>>2) + 4h, which the A operand reads from LDS).
// Q tiles (128 rows) are staged through LDS with an odd pitch so both the
// column-strided A reads of product 1 and the row reads of product 2 are
// conflict free.  exact-f32 MFMA == fmaf chain, so parity is that of fp32.
//
// Roofline: MFMA f32 (157 TF dense peak): 4*NP*NQ*KD flop per launch.
#include "common.h"
#include "kernels.h"

namespace spmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 128;  // Q rows per LDS stage

template <int KD>
__global__ __launch_bounds__(256) void expdot_kernel(int NP, int NQ, const float* __restrict__ P,
                                                     const float* __restrict__ Q,
                                                     float* __restrict__ out, float sign,
                                                     double* __restrict__ esum, int atomic_out) {
  constexpr int PITCH = KD + 1;
  constexpr int MT = KD / 32;  // 32-feature tiles of the second product
  __shared__ float qs[2][QT * PITCH];
  __shared__ double red[16];
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wid) * 32;
  const int p = p0 + c;
  // Q range of this block (gridDim.y chunks, whole QT tiles)
  const int ntiles = (NQ + QT - 1) / QT;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc;
  const int tile1 = min(ntiles, tile0 + tpc);

  // P fragment: B operand of product 1, B[k=2s+h][j=c] = P[p0+c][2s+h]
  float pb[KD / 2];
#pragma unroll
  for (int s = 0; s < KD / 2; ++s) pb[s] = p < NP ? P[(size_t)p * KD + 2 * s + h] : 0.f;

  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  double es = 0.0;

  // stage loader: 256 threads move QT*KD floats
  constexpr int PER = QT * KD / 256;
  float stage[PER];
  auto gload = [&](int tile) {
    const int q0 = tile * QT;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = i * 256 + t;
      const int r = e / KD, k = e % KD;
      stage[i] = (q0 + r < NQ) ? Q[(size_t)(q0 + r) * KD + k] : 0.f;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = i * 256 + t;
      const int r = e / KD, k = e % KD;
      qs[buf][r * PITCH + k] = stage[i];
    }
  };

  if (tile0 < tile1) {
    gload(tile0);
    swrite(0);
  }
  __syncthreads();
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = (tile - tile0) & 1;
    if (tile + 1 < tile1) gload(tile + 1);
    const float* qb = qs[buf];
    const int q0 = tile * QT;
#pragma unroll 1
    for (int sub = 0; sub < QT / 32; ++sub) {
      if (q0 + sub * 32 >= NQ) break;  // block-uniform
      const float* qt = qb + sub * 32 * PITCH;
      // ---- product 1: X[q][p] = sum_k Q[q][k] P[p][k] --------------------
      f32x16 x;
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = 0.f;
#pragma unroll
      for (int s = 0; s < KD / 2; ++s) {
        const float a = qt[c * PITCH + 2 * s + h];
        x = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[s], x, 0, 0, 0);
      }
      // ---- E = exp(X), masked outside [NQ) x [NP) ------------------------
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = q0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const float e = (q < NQ && p < NP) ? expf(x[i]) : 0.f;
        x[i] = e;
        part += e;
      }
      es += (double)part;
      // ---- product 2: out^T[k][p] += sum_q Q[q][k] E[q][p] ---------------
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
        const int row = (tt & 3) + 8 * (tt >> 2) + 4 * h;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float a = qt[row * PITCH + m * 32 + c];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[tt], acc[m], 0, 0, 0);
        }
      }
    }
    if (tile + 1 < tile1) swrite(buf ^ 1);
    __syncthreads();
  }
  // ---- store: lane holds features (i&3)+8(i>>2)+4h (+32m) of row p --------
  if (p < NP) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float* dst = out + (size_t)p * KD + m * 32 + 8 * g + 4 * h;
        if (atomic_out) {
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sign * acc[m][4 * g + j]);
        } else {
          *reinterpret_cast<float4*>(dst) =
              make_float4(sign * acc[m][4 * g + 0], sign * acc[m][4 * g + 1],
                          sign * acc[m][4 * g + 2], sign * acc[m][4 * g + 3]);
        }
      }
  }
  if (esum) {
    const double tot = block_sum(es, red);
    if (t == 0) atomicAdd(esum, tot);
  }
}

void launch_expdot(int KD, const ExpdotArgs& a, hipStream_t st) {
  const int nbx = (a.NP + 127) / 128;
  int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  dim3 grid(nbx, chunks);
  if (KD == 32)
    hipLaunchKernelGGL(expdot_kernel<32>, grid, dim3(256), 0, st, a.NP, a.NQ, a.P, a.Q, a.out,
                       a.sign, a.esum, a.atomic_out);
  else if (KD == 64)
    hipLaunchKernelGGL(expdot_kernel<64>, grid, dim3(256), 0, st, a.NP, a.NQ, a.P, a.Q, a.out,
                       a.sign, a.esum, a.atomic_out);
}

}  // namespace spmf
