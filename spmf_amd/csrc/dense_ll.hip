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
                                                         const uint8_t* __restrict__ ctype,
                                                         float* __restrict__ rate,
                                                         float* __restrict__ ll) {
  // logt = likelihood code: 0 Poisson linear, 1 Poisson log_transform, 2 Bernoulli(logits)
  // (bernoulli.py:126-155: "rate" is the logit), 3 mixed (per column, ctype)
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
    float ey_;
    const float rt = (lik_exp(logt) ? expm1_dec(fminf(y, kYSat), ey_) : y) + phi[d];
    rate[(size_t)b * D + d] = rt;
    const bool bern = lik_bern(logt) || (logt == 3 && ctype[d]);
    // x = 0 cell.  Poisson: 0*log r := 0 (multiply_no_nan) -> -r.  Bernoulli: -softplus(logit)
    ll[(size_t)b * D + d] = bern ? -(fmaxf(rt, 0.f) + log1pf(expf(-fabsf(rt)))) : -rt;
  }
}

__global__ __launch_bounds__(256) void dense_fix_kernel(int64_t B, int D, int logt,
                                                        const uint8_t* __restrict__ ctype,
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
      const int d = col[i];
      const size_t o = (size_t)b * D + d;
      const float r = rate[o];
      if (lik_bern(logt) || (logt == 3 && ctype[d])) {
        // tfd.Bernoulli(logits).log_prob(x) = x*l - softplus(l)  (bernoulli.py:147-155)
        ll[o] = x * r - (fmaxf(r, 0.f) + log1pf(expf(-fabsf(r))));
      } else {
        // tfd.Poisson.log_prob: multiply_no_nan(log r, x) - lgamma(x+1) - r
        const float xl = x == 0.f ? 0.f : x * logf(r);
        ll[o] = xl - lgammaf(x + 1.f) - r;
      }
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

// pass 2 of the rule: linear index (index_base + i) of the FIRST cell whose log-pmf equals
// the global minimum io[0]; io[3] (initialise to +inf) = min over matching cells.  The
// minimum's cell is where d(min_val) flows (tf.reduce_min's gradient).
__global__ __launch_bounds__(256) void nonfinite_argmin_kernel(int64_t n, const float* __restrict__ ll,
                                                               double index_base,
                                                               double* __restrict__ io) {
  __shared__ double red[4];
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float mv = (float)io[0];          // cells are fp32: the minimum is one of them, exactly
  double best = INFINITY;
  for (int64_t i = i0; i < n; i += stride)
    if (ll[i] == mv) { best = index_base + (double)i; break; }   // ascending i: first match wins
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) best = fmin(best, __shfl_xor(best, s));
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wid] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = fmin(best, red[w]);
    if (best < INFINITY) {
      unsigned long long* a = reinterpret_cast<unsigned long long*>(&io[3]);
      unsigned long long old = *a, assumed;
      do {                                  // atomic min on a non-negative double via CAS
        assumed = old;
        if (__longlong_as_double((long long)assumed) <= best) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(best));
      } while (assumed != old);
    }
  }
}

// sum of lgamma(x+1) over the stored (Poisson) cells of a dense_ll chunk whose rate is
// not a positive finite number -- the cells the rule replaces: their lgamma sits in
// the batch's pre-summed constant and has to be taken back out.  out[0] += sum.
__global__ __launch_bounds__(256) void nonfinite_lgamma_kernel(int64_t B, int D, int logt,
                                                               const uint8_t* __restrict__ ctype,
                                                               const int32_t* __restrict__ row_ptr,
                                                               const int32_t* __restrict__ col,
                                                               const float* __restrict__ val,
                                                               const float* __restrict__ rate,
                                                               double* __restrict__ out) {
  __shared__ double red[16];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    for (int i = start + lane; i < end; i += 64) {
      const int d = col[i];
      if (lik_bern(logt) || (logt == 3 && ctype[d])) continue;      // Bernoulli cell: no lgamma term
      const float r = rate[(size_t)b * D + d];
      const float x = val[i];
      if (x > 0.f && !(r > 0.f && r < INFINITY)) acc += (double)lgammaf(x + 1.f);
    }
  }
  const double t = block_sum(acc, red);
  if (threadIdx.x == 0 && t != 0.0) atomicAdd(out, t);
}

// The replacement rule's effect on the sparse fast path's accumulators
// (poisson.py:606-616).  The fast path left every non-finite stored cell out of
// sum x log r.  Under the rule each of them is worth min_val = m (the global
// minimum - 10) INSTEAD of its whole log-pmf, so per draw s
//     'x'_s += nnf_s * m + sum_{non-finite cells} lgamma(x+1)     (nlg_s[s])
// (the lgamma of those cells sits in the pre-summed constant; their rate is 0),
// and since m moves with the minimum's cell (s*, b*, d*), the gradient gains
//     N * d ll(s*, b*, d*) / d theta_{s*},   N = sum_s nnf_s,
// added here to draw s*'s gA' / gV' / gphi accumulators:
//     gV'[d*] += N c z_b*        c = d ll / d<z,V'> at the cell
//     gphi[d*]+= N (x/r - 1)     (Bernoulli: x - sigmoid)
//     gA'[d]  += N c xi_b* g(x_b*d) V'_d*   for every stored d of row b*
// Block s handles draw s; one block per draw, 256 threads.
template <int KP>
__global__ __launch_bounds__(256) void nonfinite_patch_kernel(
    int D, int K, int logt, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ row_scale, const float* __restrict__ u,
    const float* __restrict__ v, const float* __restrict__ w, const float* __restrict__ s,
    const float* __restrict__ eta, const uint8_t* __restrict__ ctype, float* __restrict__ acc,
    int64_t acc_stride, int Dh, const double* __restrict__ io, const double* __restrict__ nlg_s,
    int64_t rows_batch, int S) {
  const int sd = blockIdx.x;
  const int t = threadIdx.x;
  const AccLayout L{D, KP, Dh};
  float* tail = acc + (size_t)sd * acc_stride + L.tail_off();
  // totals over the draws
  double N = 0.0;
  for (int q = 0; q < S; ++q) {
    const float* tq = acc + (size_t)q * acc_stride + L.tail_off();
    N += (double)tq[2 * 2] + (double)tq[2 * 2 + 1];
  }
  if (io[2] > 0.0) N = io[2];                            // row shards: the count over ALL shards
  const double m = io[0] - 10.0;
  if (t == 0) {
    const double nnf = (double)tail[4] + (double)tail[5];
    const double v0 = (double)tail[0] + (double)tail[1] + nnf * m + nlg_s[sd];
    const float hi = (float)v0;
    tail[0] = hi;
    tail[1] = (float)(v0 - (double)hi);
  }
  const double idx = io[3];
  if (!(idx < INFINITY) || N == 0.0) return;          // minimum is the placeholder 0: no cell
  const double per_draw = (double)rows_batch * (double)D;
  const int s_star = (int)floor(idx / per_draw);
  if (s_star != sd) return;                             // block-uniform
  const double rem = idx - (double)s_star * per_draw;
  const int64_t b = (int64_t)floor(rem / (double)D);
  const int dstar = (int)(rem - (double)b * (double)D);
  u += (size_t)sd * D * K;
  v += (size_t)sd * K * D;
  w += (size_t)sd * D;
  s += (size_t)sd * 2 * D;
  float* accd = acc + (size_t)sd * acc_stride;
  const int hfs = dstar >= Dh ? 1 : 0;
  __shared__ float zs[KP], vs[KP];
  __shared__ float xstar_s;
  const int start = row_ptr[b], end = row_ptr[b + 1];
  const float xi = row_scale ? row_scale[b] : 1.f;
  if (t == 0) xstar_s = 0.f;
  __syncthreads();
  // z_b* (encode, poisson.py:640-649) and V'_d*; thread k < K owns feature k
  if (t < KP) {
    float zk = 0.f;
    if (t < K)
      for (int e = start; e < end; ++e) {
        const int d = col[e];
        const float x = val[e];
        const float s0 = s[d], s1 = s[D + d];
        const float w1 = s0 / (s0 + s1);
        const float a = lik_exp(logt) ? w1 * u[(size_t)d * K + t] : w1 * u[(size_t)d * K + t] / eta[d];
        const float gx = lik_exp(logt) ? log1pf(x / eta[d]) : x;
        zk = fmaf(gx, a, zk);
      }
    zs[t] = xi * zk;
    vs[t] = t < K ? eta[dstar] * v[(size_t)t * D + dstar] : 0.f;
  }
  for (int e = start + t; e < end; e += 256)
    if (col[e] == dstar) xstar_s = val[e];
  __syncthreads();
  float y = 0.f;
  for (int k = 0; k < KP; ++k) y = fmaf(zs[k], vs[k], y);
  const float s0 = s[dstar], s1 = s[D + dstar];
  const float phi = eta[dstar] * (s1 / (s0 + s1)) * w[dstar];
  const float x = xstar_s;
  float cy, cphi;
  if (lik_bern(logt) || (logt == 3 && ctype[dstar])) {
    const float ey = logt == 4 ? expf(fminf(y, kYSat)) : 1.f;
    const float sg = 1.f / (1.f + expf(-((logt == 4 ? ey - 1.f : y) + phi)));
    cphi = x - sg;
    cy = cphi * ey;
  } else {
    float ey = 1.f;
    const float r = (logt == 1 ? expm1_dec(fminf(y, kYSat), ey) : y) + phi;
    cphi = x / r - 1.f;
    cy = cphi * ey;
  }
  const float Nf = (float)N;
  float* gV = accd + L.gV_off(hfs);
  float* gph = accd + L.gphi_off(hfs);
  if (t < K) gV[(size_t)dstar * KP + t] += Nf * cy * zs[t];
  if (t == 0) gph[dstar] += Nf * cphi;
  // chain through z_b* to the encoder rows of row b*
  for (int i = t; i < (end - start) * KP; i += 256) {
    const int e = start + i / KP, k = i % KP;
    if (k < K) {
      const int d = col[e];
      const float gx = lik_exp(logt) ? log1pf(val[e] / eta[d]) : val[e];
      float* gA = accd + L.gA_off(d >= Dh ? 1 : 0);
      gA[(size_t)d * KP + k] += Nf * cy * xi * gx * vs[k];
    }
  }
}

void launch_nonfinite_argmin(int64_t n, const float* ll, double index_base, double* io, hipStream_t st) {
  int64_t want = (n + 1023) / 1024;
  int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(nonfinite_argmin_kernel, dim3(nb), dim3(256), 0, st, n, ll, index_base, io);
}

void launch_nonfinite_lgamma(const DenseLLArgs& a, double* out, hipStream_t st) {
  int64_t want = (a.B + 3) / 4;
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(nonfinite_lgamma_kernel, dim3(nb), dim3(256), 0, st, a.B, a.D, a.logt, a.ctype,
                     a.row_ptr, a.col, a.val, a.rate, out);
}

void launch_nonfinite_patch(int KP, const NfPatchArgs& a, hipStream_t st) {
#define SPMF_NFP(KP_)                                                                            \
  hipLaunchKernelGGL(nonfinite_patch_kernel<KP_>, dim3(a.S), dim3(256), 0, st, a.D, a.K, a.logt,  \
                     a.row_ptr, a.col, a.val, a.row_scale, a.u, a.v, a.w, a.s, a.eta, a.ctype,     \
                     a.acc, a.acc_stride, a.Dh, a.io, a.nlg, a.rows_batch, a.S)
  switch (KP) {
    case 4: SPMF_NFP(4); break;
    case 8: SPMF_NFP(8); break;
    case 16: SPMF_NFP(16); break;
    case 32: SPMF_NFP(32); break;
    case 64: SPMF_NFP(64); break;
    case 128: SPMF_NFP(128); break;
    case 256: SPMF_NFP(256); break;
    default: break;
  }
#undef SPMF_NFP
}

template <int KP>
static void launch_dense_t(const DenseLLArgs& a, hipStream_t st) {
  dim3 grid((a.D + 63) / 64, (unsigned)((a.B + 3) / 4));
  hipLaunchKernelGGL(dense_rate_kernel<KP>, grid, dim3(256), 0, st, a.B, a.D, a.logt, a.z, a.Vp,
                     a.phi, a.ctype, a.rate, a.ll);
}

void launch_dense_ll(int KP, const DenseLLArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_dense_t<4>(a, st); break;
    case 8: launch_dense_t<8>(a, st); break;
    case 16: launch_dense_t<16>(a, st); break;
    case 32: launch_dense_t<32>(a, st); break;
    case 64: launch_dense_t<64>(a, st); break;
    case 128: launch_dense_t<128>(a, st); break;
    case 256: launch_dense_t<256>(a, st); break;
    default: return;
  }
  int64_t want = (a.B + 3) / 4;
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(dense_fix_kernel, dim3(nb), dim3(256), 0, st, a.B, a.D, a.logt, a.ctype,
                     a.row_ptr, a.col, a.val, a.rate, a.ll);
}

void launch_nonfinite(int64_t n, const float* ll, int pass, double* io, hipStream_t st) {
  int64_t want = (n + 1023) / 1024;
  int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(nonfinite_kernel, dim3(nb), dim3(256), 0, st, n, ll, pass, io);
}

}  // namespace spmf
