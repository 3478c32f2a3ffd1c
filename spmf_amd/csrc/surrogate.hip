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
  float r = 0.f;
  while (x < 6.f) {
    r -= 1.f / x;
    x += 1.f;
  }
  const float i = 1.f / x, i2 = i * i;
  return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
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

__global__ __launch_bounds__(256) void surrogate_fwd_kernel(SurTable T, int S,
                                                            double* __restrict__ logq) {
  __shared__ double red[16];
  const SurVar v = T.v[blockIdx.y];
  const int base = blockIdx.x * (256 * kEPT);
  if (base >= v.n) return;   // block-uniform
  // per-element quantities that do not depend on the draw (softplus, lgamma, logs are the
  // expensive part): once, not once per draw
  float c0[kEPT], c1[kEPT], c2[kEPT];
  bool soft[kEPT];
#pragma unroll
  for (int e = 0; e < kEPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    c0[e] = c1[e] = c2[e] = 0.f;
    soft[e] = false;
    if (i < v.n) {
      const float t0 = v.t0[i], t1 = v.t1[i];
      if (v.kind == 2) {
        const float a = softplusf(t0), b = softplusf(t1);
        c0[e] = a;
        c1[e] = b;
        c2[e] = a * logf(b) - lgammaf(a);       // draw-independent part of log q_y
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
    for (int e = 0; e < kEPT; ++e) {
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
    if (threadIdx.x == 0) atomicAdd(&logq[s], tot);
  }
}

// (one element per thread measured faster here: 30 us vs 46 us at 8 -- the body is
// transcendental-heavy, parallelism matters more than block count)
constexpr int kEPTB = 1;

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
    float g0 = 0.f, g1 = 0.f;
    if (i < v.n) {
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
    o0[e] = g0;
    o1[e] = g1;
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
__global__ void vi_gate_kernel(const double* __restrict__ parts, const double* __restrict__ logq,
                               const double* __restrict__ nnf, int S, double c, double rows,
                               double* __restrict__ state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0, bad = 0.0;
  for (int s = 0; s < S; ++s) {
    double prior = 0.0;
    for (int i = 0; i < 12; ++i) prior += parts[s * 14 + i];
    acc += parts[s * 14 + 13] + parts[s * 14 + 12] + c * (prior - logq[s]);
    bad += nnf ? nnf[s] : 0.0;
  }
  const double loss = -(acc / S) / rows;
  const bool ok = (loss - loss == 0.0) && bad == 0.0;      // finite and no non-finite cell
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

void launch_vi_gate(const double* parts, const double* logq, const double* nnf, int S, double c,
                    double rows, double* state, hipStream_t st) {
  hipLaunchKernelGGL(vi_gate_kernel, dim3(1), dim3(64), 0, st, parts, logq, nnf, S, c, rows, state);
}
void launch_adam_dev(const AdamTable& T, int ntensors, int max_n, const double* state, hipStream_t st) {
  dim3 grid((max_n + 256 * kEPT - 1) / (256 * kEPT), ntensors);
  hipLaunchKernelGGL(adam_dev_kernel, grid, dim3(256), 0, st, T, state);
}

void launch_surrogate_fwd(const SurTable& T, int nvars, int max_n, int S, double* logq,
                          hipStream_t st) {
  dim3 grid((max_n + 256 * kEPT - 1) / (256 * kEPT), nvars);
  hipLaunchKernelGGL(surrogate_fwd_kernel, grid, dim3(256), 0, st, T, S, logq);
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
