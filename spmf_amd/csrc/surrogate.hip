// surrogate.hip -- the mean-field surrogate posterior of the VI step and the
// optimiser update, fused elementwise kernels (gfx950).
//
// The reference builds the surrogate in create_distributions
// (mederrata_spmf/poisson.py:403-569): Softplus(Normal(loc, scale)) for
// v, w, u, s (Identity instead of Softplus for v, w in bernoulli.py:187-193)
// and Softplus(InverseGamma(conc, scale)) for the horseshoe scale hierarchy;
// sampling, log q and the reparameterisation gradient live in the un-vendored
// bayesianquilts / TFP ([UNVERIFIED-3P]; the build defines: positive
// distribution parameters are softplus(raw) trainables).
//
//   kind 0  y = t0 + softplus(t1) * eps              theta = softplus(y)
//   kind 1  same, theta = y                          (Identity bijector)
//   kind 2  a = softplus(t0), b = softplus(t1), y = b / g, g ~ Gamma(a,1)
//           theta = softplus(y); dg/da is supplied (implicit reparameterisation)
//   log q(theta) = log q_y(y) - log sigmoid(y)       (no Jacobian for kind 1)
//
// surrogate_fwd : theta[S,n], logq[S] (fp64)          one launch for all 12 variables
// surrogate_bwd : d loss / d(t0,t1) given g = dE/dtheta from the finish kernel,
//                 loss = -(1/(S*B)) sum_s [E_s - c*logq_s]     
// adam_kernel   : tf.keras-style Adam with optional value clipping, all 24
//                 trainables in one launch.
// All O(D*K) elementwise: HBM bound, microseconds.
#include "common.h"
#include "kernels.h"

namespace spmf {

__device__ __forceinline__ float softplusf(float x) {
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf_(float x) {
  const float e = expf(-fabsf(x));
  const float inv = 1.f / (1.f + e);
  return x >= 0.f ? inv : e * inv;
}
// log sigmoid(y) = -softplus(-y)
__device__ __forceinline__ float logsigmoidf_(float y) { return -softplusf(-y); }
__device__ __forceinline__ float digammaf_(float x) {
  // sum_{j<m} 1/(x+j) as one fraction (see digammad_): one division instead of up to six
  float num = 0.f, den = 1.f;
  while (x < 6.f) {
    num = fmaf(num, x, den);
    den *= x;
    x += 1.f;
  }
  const float r = -num / den;
  const float i = 1.f / x, i2 = i * i;
  return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}

// lgamma(a), a > 0: shift to x >= 8 by the recurrence (one log of the product) + Stirling.
// Absolute error ~2e-6 (fp32 rounding of (x - 1/2) log x ~ 16), the level of the library
// lgammaf, at a third of its instructions.
__device__ __forceinline__ float lgamma_pos_(float a) {
  float prod = 1.f, x = a;
  while (x < 8.f) {
    prod *= x;
    x += 1.f;
  }
  const float i = 1.f / x, i2 = i * i;
  const float st = (x - 0.5f) * logf(x) - x + 0.91893853320467274178f +
                   i * (1.f / 12.f - i2 * (1.f / 360.f - i2 * (1.f / 1260.f)));
  return st - logf(prod);
}

// Elements per thread of the elementwise kernels below: a block covers
// 256*kEPT consecutive elements (coalesced per step), so that the per-block
// fp64 atomic on logq[s] -- all blocks add to the SAME address, ~13 ns each,
// serialised at the memory side -- is paid once per 2048 elements (it was 134 us
// of atomics for the 2.7 M trainable elements of C3 at one per 256).
#ifndef SPMF_EPT
#define SPMF_EPT 8
#endif
constexpr int kEPT = SPMF_EPT;

// logq: with `lqpart` every block stores its sum per draw in its own slot
// (lqpart[s][block]) and logq_reduce_kernel adds the slots in a fixed order -- no
// same-address fp64 atomics (1300 of them at C3 sizes, ~13 ns each, serialised)
// and a result that is bit-identical from run to run; without it (scratch too
// small for S * blocks) the blocks add to logq[s] atomically.
#ifndef SPMF_EPTF
#define SPMF_EPTF 4
#endif
constexpr int kEPTF = SPMF_EPTF;   // elements per thread of surrogate_fwd_kernel

__global__ __launch_bounds__(256) void surrogate_fwd_kernel(SurTable T, int S,
                                                            double* __restrict__ logq,
                                                            double* __restrict__ lqpart) {
  __shared__ double red[16];
  const SurVar v = T.v[blockIdx.y];
  const int base = blockIdx.x * (256 * kEPTF);
  const size_t nblk = (size_t)gridDim.x * gridDim.y, blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (base >= v.n) {          // block-uniform
    if (lqpart && threadIdx.x == 0)
      for (int s = 0; s < S; ++s) lqpart[(size_t)s * nblk + blk] = 0.0;
    return;
  }
  // per-element quantities that do not depend on the draw (softplus, lgamma, logs are the
  // expensive part): once, not once per draw
  float c0[kEPTF], c1[kEPTF], c2[kEPTF];
  bool soft[kEPTF];
#pragma unroll
  for (int e = 0; e < kEPTF; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    c0[e] = c1[e] = c2[e] = 0.f;
    soft[e] = false;
    if (i < v.n) {
      const float t0 = v.t0[i], t1 = v.t1[i];
      if (v.kind == 2) {
        const float a = softplusf(t0), b = softplusf(t1);
        c0[e] = a;
        c1[e] = b;
        c2[e] = a * logf(b) - lgamma_pos_(a);   // draw-independent part of log q_y
      } else {
        const float sg = softplusf(t1);
        c0[e] = t0;
        c1[e] = sg;
        c2[e] = -logf(sg) - 0.91893853320467274178f;
      }
      soft[e] = v.kind != 1 && !(v.ident && v.ident[i]);
    }
  }
  for (int s = 0; s < S; ++s) {
    double lq = 0.0;
    const float* __restrict__ nzp = v.noise + (size_t)s * v.ld;
    float* __restrict__ thp = v.theta + (size_t)s * v.n;
#pragma unroll
    for (int e = 0; e < kEPTF; ++e) {
      const int i = base + e * 256 + threadIdx.x;
      if (i < v.n) {
        const float nz = nzp[i];
        float y, l;
        if (v.kind == 2) {
          y = c1[e] / nz;
          l = c2[e] - (c0[e] + 1.f) * logf(y) - c1[e] / y;
        } else {
          y = c0[e] + c1[e] * nz;
          l = -0.5f * nz * nz + c2[e];
        }
        float th = y;
        if (soft[e]) {
          th = softplusf(y);
          l -= logsigmoidf_(y);
        }
        thp[i] = th;
        lq += (double)l;
      }
    }
    const double tot = block_sum(lq, red);
    if (threadIdx.x == 0) {
      if (lqpart) lqpart[(size_t)s * nblk + blk] = tot;
      else atomicAdd(&logq[s], tot);
    }
  }
}

__global__ __launch_bounds__(256) void logq_reduce_kernel(int nblk, const double* __restrict__ lqpart,
                                                          double* __restrict__ logq) {
  __shared__ double red[16];
  const double* p = lqpart + (size_t)blockIdx.x * nblk;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += p[b];
  const double tot = block_sum(s, red);
  if (threadIdx.x == 0) logq[blockIdx.x] = tot;
}

// (one element per thread measured faster here: 30 us vs 46 us at 8 -- the body is
// transcendental-heavy, parallelism matters more than block count)
constexpr int kEPTB = 1;

// d loss / d(t0, t1) of element i, summed over the S draws
__device__ __forceinline__ void sur_bwd_elem(const SurVar& v, int i, int S, float inv_sb, float c,
                                             float& g0, float& g1) {
  g0 = 0.f;
  g1 = 0.f;
  const float t0 = v.t0[i], t1 = v.t1[i];
  if (v.kind == 2) {
    const float a = softplusf(t0), b = softplusf(t1);
    const float lb = logf(b), dga = digammaf_(a);
    for (int s = 0; s < S; ++s) {
      const size_t o = (size_t)s * v.n + i, on = (size_t)s * v.ld + i;
      const float g = v.noise[on], dgda = v.dgda[on], ge = v.gtheta[o];
      const float y = b / g, sig = sigmoidf_(y);
      const float dlq_dy = -(a + 1.f) / y + b / (y * y) - (1.f - sig);
      const float dL_dy = inv_sb * (-ge * sig + c * dlq_dy);
      const float dy_da = -b / (g * g) * dgda, dy_db = 1.f / g;
      g0 += dL_dy * dy_da + inv_sb * c * (lb - dga - logf(y));
      g1 += dL_dy * dy_db + inv_sb * c * (a / b - 1.f / y);
    }
    g0 *= sigmoidf_(t0);
    g1 *= sigmoidf_(t1);
  } else {
    const float sg = softplusf(t1);
    const bool ident = v.kind != 0 || (v.ident && v.ident[i]);
    for (int s = 0; s < S; ++s) {
      const size_t o = (size_t)s * v.n + i, on = (size_t)s * v.ld + i;
      const float eps = v.noise[on], ge = v.gtheta[o];
      const float y = t0 + sg * eps;
      float dth = 1.f, dlq_dy = 0.f;
      if (!ident) {
        dth = sigmoidf_(y);
        dlq_dy = -(1.f - dth);
      }
      const float dL_dy = inv_sb * (-ge * dth + c * dlq_dy);
      g0 += dL_dy;
      g1 += dL_dy * eps - inv_sb * c / sg;
    }
    g1 *= sigmoidf_(t1);
  }
}

__global__ __launch_bounds__(256) void surrogate_bwd_kernel(SurTable T, int S, float inv_sb,
                                                            float c) {
  const SurVar v = T.v[blockIdx.y];
  const int base = blockIdx.x * (256 * kEPTB);
  if (base >= v.n) return;   // block-uniform
  float o0[kEPTB], o1[kEPTB];
  // results are stored after the loop: the stores may alias the inputs as far as
  // the compiler knows, and would otherwise serialise the loads of the steps
#pragma unroll
  for (int e = 0; e < kEPTB; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    o0[e] = o1[e] = 0.f;
    if (i < v.n) sur_bwd_elem(v, i, S, inv_sb, c, o0[e], o1[e]);
  }
#pragma unroll
  for (int e = 0; e < kEPTB; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < v.n) {
      v.g0[i] = o0[e];
      v.g1[i] = o1[e];
    }
  }
}

// Adam over 256*kEPT consecutive elements per block.
__device__ __forceinline__ float adam_elem(float& m, float& v, float p, float g, float lr, float b1,
                                           float b2, float eps, float c1, float c2, float clip) {
  if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  return p - lr * (m / c1) / (sqrtf(v / c2) + eps);
}

__device__ __forceinline__ void adam_block(const AdamVar& a, float lr, float b1, float b2, float eps,
                                           float c1, float c2, float clip) {
  const int base = blockIdx.x * (256 * kEPT);
  if (base >= a.n) return;
  // 16-B lanes when the four arrays allow it (dword streams run at about half the
  // dwordx4 rate on this part): block-uniform choice
  const bool vec = (a.n & 3) == 0 &&
                   (((uintptr_t)a.p | (uintptr_t)a.m | (uintptr_t)a.v | (uintptr_t)a.g) & 15) == 0;
  if (vec) {
    constexpr int NV = kEPT / 4;
    float4 p[NV], m[NV], vv[NV], g[NV];
    const int n4 = a.n >> 2, base4 = base >> 2;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const int i = base4 + e * 256 + threadIdx.x;
      if (i < n4) {
        p[e] = reinterpret_cast<const float4*>(a.p)[i];
        m[e] = reinterpret_cast<const float4*>(a.m)[i];
        vv[e] = reinterpret_cast<const float4*>(a.v)[i];
        g[e] = reinterpret_cast<const float4*>(a.g)[i];
      }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const int i = base4 + e * 256 + threadIdx.x;
      if (i < n4) {
        float4 o;
        o.x = adam_elem(m[e].x, vv[e].x, p[e].x, g[e].x, lr, b1, b2, eps, c1, c2, clip);
        o.y = adam_elem(m[e].y, vv[e].y, p[e].y, g[e].y, lr, b1, b2, eps, c1, c2, clip);
        o.z = adam_elem(m[e].z, vv[e].z, p[e].z, g[e].z, lr, b1, b2, eps, c1, c2, clip);
        o.w = adam_elem(m[e].w, vv[e].w, p[e].w, g[e].w, lr, b1, b2, eps, c1, c2, clip);
        reinterpret_cast<float4*>(a.m)[i] = m[e];
        reinterpret_cast<float4*>(a.v)[i] = vv[e];
        reinterpret_cast<float4*>(a.p)[i] = o;
      }
    }
    return;
  }
  // operands of all steps are loaded before the first store (p, m, v are updated in place)
  float p[kEPT], m[kEPT], vv[kEPT], g[kEPT];
#pragma unroll
  for (int e = 0; e < kEPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    const bool in = i < a.n;
    p[e] = in ? a.p[i] : 0.f;
    m[e] = in ? a.m[i] : 0.f;
    vv[e] = in ? a.v[i] : 0.f;
    g[e] = in ? a.g[i] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < kEPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < a.n) {
      const float o = adam_elem(m[e], vv[e], p[e], g[e], lr, b1, b2, eps, c1, c2, clip);
      a.m[i] = m[e];
      a.v[i] = vv[e];
      a.p[i] = o;
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTable T, float lr, float b1, float b2,
                                                   float eps, float c1, float c2, float clip) {
  adam_block(T.v[blockIdx.y], lr, b1, b2, eps, c1, c2, clip);
}

// ---- device-resident optimiser state (graph-capturable VI step) -------------
// state[]: see include/spmf_hip.h (SPMF_VI_*).  One thread: loss of this step
// from the 14 parts and log q, the apply/skip decision, Adam's running powers.
// ---------------------------------------------------------------------------
// Base noise of the surrogate, drawn on the device (replaces torch's randn /
// _standard_gamma / _standard_gamma_grad on the step path).
//   counter-based RNG: Philox4x32-10 (Salmon et al., SC'11), key = the 64-bit seed,
//   counter = (element, draw | variable << 16 | attempt << 24, step counter lo, hi);
//   the step counter comes from the device-resident VI state (state[13], advanced by
//   the gate kernel) when the step is replayed from a hipGraph, else from the host.
//   kind 0/1: eps ~ N(0,1) by Box-Muller.
//   kind 2:   g ~ Gamma(a, 1), a = softplus(t0), by Marsaglia & Tsang (2000) (a < 1:
//             Gamma(a+1) * U^(1/a)), and the implicit-reparameterisation derivative
//             (Figurnov et al. 2018)   dg/da = -(dP/da)(a, g) / p(g; a)
//             with P the regularised lower incomplete gamma function, from its series
//             P(a,x) = sum_n T_n,  T_n = x^(a+n) e^-x / Gamma(a+n+1),
//             dT_n/da = T_n (log x - psi(a+n+1)),  in fp64.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
__device__ __forceinline__ float normal_bm(uint32_t a, uint32_t b) {
  return sqrtf(-2.f * logf(u01(a))) * cospif(2.f * u01(b));
}
// fp64 reciprocal: v_rcp_f32 seed + two Newton steps (|rel err| ~ 1e-15 for the ranges used)
__device__ __forceinline__ double rcpd_(double d) {
  double rc = (double)__builtin_amdgcn_rcpf((float)d);
  rc = rc * (2.0 - d * rc);
  return rc * (2.0 - d * rc);
}
__device__ __forceinline__ double digammad_(double x) {
  // recurrence psi(x) = psi(x+m) - sum_{j<m} 1/(x+j): the sum as ONE fraction num/den
  // (den = prod (x+j) <= 8^8), so one reciprocal instead of up to eight fp64 divisions
  double num = 0.0, den = 1.0;
  while (x < 8.0) {
    num = num * x + den;
    den *= x;
    x += 1.0;
  }
  const double r = -num * rcpd_(den);
  const double i = rcpd_(x), i2 = i * i;
  return r + log(x) - 0.5 * i - i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0))));
}
// d g / d a of g ~ Gamma(a, 1) at the drawn value (implicit reparameterisation)
__device__ __forceinline__ float gamma_dgda(double a, double x) {
  // -dP/da / p = -sum_n R_n (log x - psi(a+n+1)),  R_n = T_n / p(x; a) = x^(n+1) / (a (a+1) ... (a+n)):
  // the density cancels analytically, so no exp / lgamma is needed.  fp64 throughout
  // (in the right tail the terms reach ~e^x before they cancel); the reciprocal is a
  // v_rcp_f32 seed + two Newton steps instead of an fp64 division; ~x + 40 terms.
  const double lx = log(x);
  double R = x / a;                                     // R_0
  double psi = digammad_(a + 1.0);
  double acc = 0.0;
  for (int n = 0; n < 4000; ++n) {
    const double term = R * (lx - psi);
    acc += term;
    if ((double)n > x && fabs(term) < 1e-8 * fabs(acc) + 1e-300) break;    // the result is stored as fp32 (6e-8)
    const double dn = a + (double)(n + 1);
    double rc = (double)__builtin_amdgcn_rcpf((float)dn);
    rc = rc * (2.0 - dn * rc);                 // one Newton step: ~1e-14 relative
    psi += rc;
    R *= x * rc;
  }
  return (float)(-acc);
}

// One element's base noise for draw s of variable `var` (index in the caller's table): eps ~ N(0,1), or
// g ~ Gamma(a, 1) with dgda its implicit-reparameterisation derivative (a = softplus(t0), given).
__device__ __forceinline__ void draw_elem(int kind, float a, int var, int s, int i, uint2 key, uint32_t clo,
                                          uint32_t chi, float& nz_out, float& dgda_out) {
  const uint32_t c1 = (uint32_t)s | ((uint32_t)var << 16);
  dgda_out = 0.f;
  if (kind != 2) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)i, c1, clo, chi), key);
    nz_out = normal_bm(r.x, r.y);
    return;
  }
  const float ab = a < 1.f ? a + 1.f : a;              // boost: Gamma(a) = Gamma(a+1) U^(1/a)
  const float d = ab - (1.f / 3.f), cc = rsqrtf(9.f * d);
  float g = d;                                          // (fallback after 16 rejected blocks: the mode)
  float ub = 1.f;
  // One Philox block serves TWO Marsaglia-Tsang candidates: both Box-Muller normals (cos and sin
  // branch of the same radius), a uniform each, and the boost uniform from the low bytes that
  // u01() does not look at.  A lane rejects a candidate 1 time in ~20, so a wave of 64 nearly
  // always needed a second block with one candidate per block; with two it rarely does.
  bool done = false;
  for (uint32_t att = 0; att < 16 && !done; ++att) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)i, c1 | (att << 24), clo, chi), key);
    if (att == 0)
      ub = ((float)(((r.x & 0xffu) << 16) | ((r.y & 0xffu) << 8) | (r.z & 0xffu)) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.f * logf(u01(r.x)));
    float sn, cs;
    sincospif(2.f * u01(r.y), &sn, &cs);
#pragma unroll
    for (int cand = 0; cand < 2; ++cand) {
      const float x = rad * (cand == 0 ? cs : sn);
      const float t = 1.f + cc * x;
      const float vv = t * t * t;
      const float u = u01(cand == 0 ? r.z : r.w);
      if (!done && t > 0.f && logf(u) < 0.5f * x * x + d - d * vv + d * logf(vv)) {
        g = d * vv;
        done = true;
      }
    }
  }
  if (a < 1.f) g *= powf(ub, 1.f / a);
  g = fmaxf(g, 1e-30f);
  nz_out = g;
  dgda_out = gamma_dgda((double)a, (double)g);
}

__global__ __launch_bounds__(256) void sample_noise_kernel(SurTable T, int S, uint32_t seed_lo,
                                                           uint32_t seed_hi, uint64_t counter,
                                                           const double* __restrict__ state) {
  const int var = blockIdx.y, s = blockIdx.z;
  const SurVar v = T.v[var];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= v.n) return;
  if (state) counter += (uint64_t)state[13];           // steps taken so far (device-resident)
  const uint2 key = make_uint2(seed_lo, seed_hi);
  float nz, dg;
  draw_elem(v.kind, v.kind == 2 ? softplusf(v.t0[i]) : 0.f, var, s, i, key, (uint32_t)counter,
            (uint32_t)(counter >> 32), nz, dg);
  const_cast<float*>(v.noise)[(size_t)s * v.ld + i] = nz;
  if (v.kind == 2) const_cast<float*>(v.dgda)[(size_t)s * v.ld + i] = dg;
}

// spmf_sample_noise + spmf_surrogate_fwd in ONE launch (the VI step's path): the sampler's decomposition -- one
// element and one draw per thread, grid (elements / 256, variables, draws): the gamma draws' series is what the
// launch's time is, and it wants every lane it can get (a first form with the transform's four elements x all
// draws per thread was 86 us SLOWER on the 8-GPU shard) -- and each thread transforms what it drew while it is
// in registers; noise / dgda (the chain rule needs them) and theta are written, the block sums of log q go to
// per-block slots that logq_reduce_kernel adds up in block order -- two launches instead of three, a fixed
// association (log q equals the two-call form to rounding, not bit for bit: the partial sums group 256 elements
// here, 1024 there).  (The fold by the LAST workgroup to arrive, in the same launch, was built and measured: 30 000
// workgroups taking a ticket at ONE address serialise at ~13 ns each -- 0.85 against 0.49 ms per VI step on the
// 8-GPU shard -- and every one of them pays an agent-scope release first.)
__global__ __launch_bounds__(256) void sample_fwd_kernel(SurTable T, int S, uint32_t seed_lo, uint32_t seed_hi,
                                                         uint64_t counter, const double* __restrict__ state,
                                                         double* __restrict__ lqpart) {
  __shared__ double red[16];
  const int var = blockIdx.y, s = blockIdx.z;
  const SurVar v = T.v[var];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const size_t nb2 = (size_t)gridDim.x * gridDim.y, blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (blockIdx.x * 256 >= v.n) {                       // block-uniform: past this variable's end (or a skipped one)
    if (threadIdx.x == 0) lqpart[(size_t)s * nb2 + blk] = 0.0;
    return;
  }
  if (state) counter += (uint64_t)state[13];
  double lq = 0.0;
  if (i < v.n) {
    const float t0 = v.t0[i], t1 = v.t1[i];
    float c0, c1, c2;
    if (v.kind == 2) {
      const float a = softplusf(t0), b = softplusf(t1);
      c0 = a;
      c1 = b;
      c2 = a * logf(b) - lgamma_pos_(a);
    } else {
      const float sg = softplusf(t1);
      c0 = t0;
      c1 = sg;
      c2 = -logf(sg) - 0.91893853320467274178f;
    }
    const bool soft = v.kind != 1 && !(v.ident && v.ident[i]);
    float nz, dg;
    draw_elem(v.kind, c0, var, s, i, make_uint2(seed_lo, seed_hi), (uint32_t)counter, (uint32_t)(counter >> 32), nz, dg);
    const_cast<float*>(v.noise)[(size_t)s * v.ld + i] = nz;
    if (v.kind == 2) const_cast<float*>(v.dgda)[(size_t)s * v.ld + i] = dg;
    float y, l;
    if (v.kind == 2) {
      y = c1 / nz;
      l = c2 - (c0 + 1.f) * logf(y) - c1 / y;
    } else {
      y = c0 + c1 * nz;
      l = -0.5f * nz * nz + c2;
    }
    float th = y;
    if (soft) {
      th = softplusf(y);
      l -= logsigmoidf_(y);
    }
    v.theta[(size_t)s * v.n + i] = th;
    lq = (double)l;
  }
  const double tot = block_sum(lq, red);               // (every thread of the block: a block past the variable's end adds 0)
  if (threadIdx.x == 0) lqpart[(size_t)s * nb2 + blk] = tot;
}

bool launch_sample_fwd(const SurTable& T, int nvars, int max_n, int S, uint64_t seed, uint64_t counter,
                       const double* state, double* logq, double* scratch, size_t scratch_doubles, hipStream_t st) {
  dim3 grid((max_n + 255) / 256, nvars, S);
  const size_t nb2 = (size_t)grid.x * grid.y;
  if (!scratch || nb2 * (size_t)S > scratch_doubles) return false;   // (the caller runs the separate kernels)
  hipLaunchKernelGGL(sample_fwd_kernel, grid, dim3(256), 0, st, T, S, (uint32_t)seed, (uint32_t)(seed >> 32), counter,
                     state, scratch);
  hipLaunchKernelGGL(logq_reduce_kernel, dim3(S), dim3(256), 0, st, (int)nb2, scratch, logq);
  return true;
}

void launch_sample_noise(const SurTable& T, int nvars, int max_n, int S, uint64_t seed, uint64_t counter,
                         const double* state, hipStream_t st) {
  dim3 grid((max_n + 255) / 256, nvars, S);
  hipLaunchKernelGGL(sample_noise_kernel, grid, dim3(256), 0, st, T, S, (uint32_t)seed,
                     (uint32_t)(seed >> 32), counter, state);
}

__global__ void vi_gate_kernel(const double* __restrict__ parts, const double* __restrict__ logq,
                               const double* __restrict__ nnf, int S, double c, double rows,
                               double* __restrict__ state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0, bad = 0.0, sat = 0.0;
  for (int s = 0; s < S; ++s) {
    double prior = 0.0;
    for (int i = 0; i < 12; ++i) prior += parts[s * 14 + i];
    acc += parts[s * 14 + 13] + parts[s * 14 + 12] + c * (prior - logq[s]);
    bad += nnf ? nnf[s] : 0.0;
    sat += nnf ? nnf[S + s] : 0.0;     // spmf_finish: [S + s] = saturation events of draw s
  }
  state[14] += sat;
  const double loss = -(acc / S) / rows;
  const bool ok = (loss - loss == 0.0) && bad == 0.0;      // finite and no non-finite cell
  state[13] += 1.0;                                        // RNG step counter (spmf_sample_noise)
  state[8] = loss;
  state[9] = ok ? 1.0 : 0.0;
  if (ok) {
    state[5] *= state[1];
    state[6] *= state[2];
    state[7] += 1.0;
    state[10] += loss;
    state[11] += 1.0;
  } else {
    state[12] += 1.0;
  }
}

__global__ __launch_bounds__(256) void adam_dev_kernel(AdamTable T, const double* __restrict__ state) {
  if (state[9] == 0.0) return;                              // step skipped (poisson.py fit: NaN batch)
  adam_block(T.v[blockIdx.y], (float)state[0], (float)state[1], (float)state[2], (float)state[3],
             (float)(1.0 - state[5]), (float)(1.0 - state[6]), (float)state[4]);
}

// Chain rule and gated Adam in one pass: the gradient of a trainable pair never goes
// to memory (the separate kernels move 21 floats per element, this one 16).  A.v[2*var],
// A.v[2*var+1] are the Adam records of variable var's t0 / t1 (p aliases T's t0 / t1).
__global__ __launch_bounds__(256) void surrogate_bwd_adam_kernel(SurTable T, AdamTable A, int S,
                                                                 float inv_sb, float c,
                                                                 const double* __restrict__ state) {
  if (state[9] == 0.0) return;                              // step skipped
  const SurVar v = T.v[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= v.n) return;
  float g0, g1;
  sur_bwd_elem(v, i, S, inv_sb, c, g0, g1);
  const float lr = (float)state[0], b1 = (float)state[1], b2 = (float)state[2], eps = (float)state[3];
  const float c1 = (float)(1.0 - state[5]), c2 = (float)(1.0 - state[6]), clip = (float)state[4];
  const AdamVar& a0 = A.v[2 * blockIdx.y];
  const AdamVar& a1 = A.v[2 * blockIdx.y + 1];
  float m0 = a0.m[i], v0 = a0.v[i], m1 = a1.m[i], v1 = a1.v[i];
  const float p0 = adam_elem(m0, v0, a0.p[i], g0, lr, b1, b2, eps, c1, c2, clip);
  const float p1 = adam_elem(m1, v1, a1.p[i], g1, lr, b1, b2, eps, c1, c2, clip);
  a0.m[i] = m0; a0.v[i] = v0; a0.p[i] = p0;
  a1.m[i] = m1; a1.v[i] = v1; a1.p[i] = p1;
}

void launch_surrogate_bwd_adam(const SurTable& T, const AdamTable& A, int nvars, int max_n, int S,
                               float inv_sb, float c, const double* state, hipStream_t st) {
  dim3 grid((max_n + 255) / 256, nvars);
  hipLaunchKernelGGL(surrogate_bwd_adam_kernel, grid, dim3(256), 0, st, T, A, S, inv_sb, c, state);
}

void launch_vi_gate(const double* parts, const double* logq, const double* nnf, int S, double c,
                    double rows, double* state, hipStream_t st) {
  hipLaunchKernelGGL(vi_gate_kernel, dim3(1), dim3(64), 0, st, parts, logq, nnf, S, c, rows, state);
}
void launch_adam_dev(const AdamTable& T, int ntensors, int max_n, const double* state, hipStream_t st) {
  dim3 grid((max_n + 256 * kEPT - 1) / (256 * kEPT), ntensors);
  hipLaunchKernelGGL(adam_dev_kernel, grid, dim3(256), 0, st, T, state);
}

void launch_surrogate_fwd(const SurTable& T, int nvars, int max_n, int S, double* logq,
                          double* scratch, size_t scratch_doubles, hipStream_t st) {
  dim3 grid((max_n + 256 * kEPTF - 1) / (256 * kEPTF), nvars);
  const size_t nblk = (size_t)grid.x * grid.y;
  if (scratch && nblk * (size_t)S <= scratch_doubles) {
    hipLaunchKernelGGL(surrogate_fwd_kernel, grid, dim3(256), 0, st, T, S, logq, scratch);
    hipLaunchKernelGGL(logq_reduce_kernel, dim3(S), dim3(256), 0, st, (int)nblk, scratch, logq);
    return;
  }
  launch_zero(logq, (size_t)S * sizeof(double), st);
  hipLaunchKernelGGL(surrogate_fwd_kernel, grid, dim3(256), 0, st, T, S, logq, (double*)nullptr);
}
void launch_surrogate_bwd(const SurTable& T, int nvars, int max_n, int S, float inv_sb, float c,
                          hipStream_t st) {
  dim3 grid((max_n + 256 * kEPTB - 1) / (256 * kEPTB), nvars);
  hipLaunchKernelGGL(surrogate_bwd_kernel, grid, dim3(256), 0, st, T, S, inv_sb, c);
}
void launch_adam(const AdamTable& T, int ntensors, int max_n, float lr, float b1, float b2,
                 float eps, float c1, float c2, float clip, hipStream_t st) {
  dim3 grid((max_n + 256 * kEPT - 1) / (256 * kEPT), ntensors);
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, T, lr, b1, b2, eps, c1, c2, clip);
}

}  // namespace spmf
