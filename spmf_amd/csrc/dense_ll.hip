// dense_ll.hip -- per-cell outputs of log_likelihood_components
// (mederrata_spmf/poisson.py:156-184): rate[b,d] and the Poisson log-pmf for
// EVERY cell of a batch, materialised dense.  This is the reference's own
// output format ([B,D] per draw) and is output-bound (8 B written per cell);
// it serves the class surface (log_likelihood_components,
// predictive_distribution) and the non-finite replacement rule (:606-616),
// which needs a minimum over all B*D cells.  The hot path never calls it.
//
//   dense_rate_kernel : rate = f(<z_b, V'_d>) + phi_d ; ll = -rate   (x = 0 cells)
//   dense_fix_kernel  : stored cells: ll = x log r - r - lgamma(x+1)
//   nonfinite_kernel  : min over finite cells, then sum of the clipped /
//                       replaced log-likelihood (two launches)
#include "common.h"
#include "kernels.h"

namespace spmf {

template <int KP>
__global__ __launch_bounds__(256) void dense_rate_kernel(int64_t B, int D, int logt,
                                                         const float* __restrict__ z,
                                                         const float* __restrict__ Vp,
                                                         const float* __restrict__ phi,
                                                         float* __restrict__ rate,
                                                         float* __restrict__ ll) {
  __shared__ float vs[64][KP + 1];
  __shared__ float zs[4][KP];
  const int t = threadIdx.x;
  const int d0 = blockIdx.x * 64;
  const int64_t b0 = (int64_t)blockIdx.y * 4;
  for (int e = t; e < 64 * KP; e += 256) {
    const int dl = e / KP, k = e % KP;
    vs[dl][k] = (d0 + dl < D) ? Vp[(size_t)(d0 + dl) * KP + k] : 0.f;
  }
  for (int e = t; e < 4 * KP; e += 256) {
    const int r = e / KP, k = e % KP;
    zs[r][k] = (b0 + r < B) ? z[(size_t)(b0 + r) * KP + k] : 0.f;
  }
  __syncthreads();
  const int dl = t & 63, r = t >> 6;
  const int d = d0 + dl;
  const int64_t b = b0 + r;
  if (d < D && b < B) {
    float y = 0.f;
#pragma unroll 8
    for (int k = 0; k < KP; ++k) y = fmaf(zs[r][k], vs[dl][k], y);
    const float rt = (logt ? expf(y) - 1.f : y) + phi[d];
    rate[(size_t)b * D + d] = rt;
    ll[(size_t)b * D + d] = -rt;           // x = 0: 0*log r := 0 (multiply_no_nan)
  }
}

__global__ __launch_bounds__(256) void dense_fix_kernel(int64_t B, int D,
                                                        const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ col,
                                                        const float* __restrict__ val,
                                                        const float* __restrict__ rate,
                                                        float* __restrict__ ll) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    for (int i = start + lane; i < end; i += 64) {
      const float x = val[i];
      const size_t o = (size_t)b * D + col[i];
      const float r = rate[o];
      // tfd.Poisson.log_prob: multiply_no_nan(log r, x) - lgamma(x+1) - r
      const float xl = x == 0.f ? 0.f : x * logf(r);
      ll[o] = xl - lgammaf(x + 1.f) - r;
    }
  }
}

// pass 0: dmin = min over finite cells of ll (and 0, the reference's
// where(finite, ll, 0)); pass 1: sum of where(finite, clip(ll, m, 0), m).
__global__ __launch_bounds__(256) void nonfinite_kernel(int64_t n, const float* __restrict__ ll,
                                                        int pass, double* __restrict__ io) {
  __shared__ double red[16];
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (pass == 0) {
    double m = 0.0;
    for (int64_t i = i0; i < n; i += stride) {
      const float v = ll[i];
      if (isfinite(v)) m = fmin(m, (double)v);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmin(m, __shfl_xor(m, s));
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wid] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmin(m, red[w]);
      // atomic min on a double via CAS
      unsigned long long* a = reinterpret_cast<unsigned long long*>(&io[0]);
      unsigned long long old = *a, assumed;
      do {
        assumed = old;
        if (__longlong_as_double((long long)assumed) <= m) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(m));
      } while (assumed != old);
    }
  } else {
    const double mv = io[0] - 10.0;          // poisson.py:609
    double s = 0.0, nf = 0.0;
    for (int64_t i = i0; i < n; i += stride) {
      const float v = ll[i];
      if (isfinite(v)) {
        s += fmin(fmax((double)v, mv), 0.0);  // clip_by_value(ll, min_val, 0)
      } else {
        s += mv;                              // replaced by min_val (:612-616)
        nf += 1.0;
      }
    }
    const double ts = block_sum(s, red);
    const double tn = block_sum(nf, red);
    if (threadIdx.x == 0) {
      atomicAdd(&io[1], ts);
      if (tn != 0.0) atomicAdd(&io[2], tn);
    }
  }
}

template <int KP>
static void launch_dense_t(const DenseLLArgs& a, hipStream_t st) {
  dim3 grid((a.D + 63) / 64, (unsigned)((a.B + 3) / 4));
  hipLaunchKernelGGL(dense_rate_kernel<KP>, grid, dim3(256), 0, st, a.B, a.D, a.logt, a.z, a.Vp,
                     a.phi, a.rate, a.ll);
}

void launch_dense_ll(int KP, const DenseLLArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_dense_t<4>(a, st); break;
    case 8: launch_dense_t<8>(a, st); break;
    case 16: launch_dense_t<16>(a, st); break;
    case 32: launch_dense_t<32>(a, st); break;
    case 64: launch_dense_t<64>(a, st); break;
    default: return;
  }
  int64_t want = (a.B + 3) / 4;
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(dense_fix_kernel, dim3(nb), dim3(256), 0, st, a.B, a.D, a.row_ptr, a.col,
                     a.val, a.rate, a.ll);
}

void launch_nonfinite(int64_t n, const float* ll, int pass, double* io, hipStream_t st) {
  int64_t want = (n + 1023) / 1024;
  int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(nonfinite_kernel, dim3(nb), dim3(256), 0, st, n, ll, pass, io);
}

}  // namespace spmf
