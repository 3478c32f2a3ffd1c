// widek.hip -- the sparse passes for latent dimensions above 64 (KP = 128, 256; gfx950, wave64).
//
// The reference's `latent_dim` defaults to `feature_dim` (poisson.py:103-104) and its own harness
// runs P = 50 (tests/spmf_test.py:21); the named configurations stop at K = 64, which is what the
// lane-group kernels of row_pass.hip / col_pass.hip are shaped for (a factor row = KP/4 lanes x
// float4, 64/(KP/4) stored entries per wave instruction).  Above that a factor row is the WHOLE
// wave -- lane l owns k = l*VPL .. l*VPL + VPL-1, VPL = KP/64 -- so one wave instruction gathers
// one stored entry's row as one contiguous 4*KP-byte read, dot products fold over the 64 lanes,
// and the per-entry scalar work (rate, log, reciprocal) is wave-uniform.  Same algebra, same
// outputs, same accumulator layout as the lane-group kernels:
//
//   row_widek_kernel   z_b = xi_b sum_d x A'_d ; r = <z_b, V'_d> + phi_d ; sum x log r ;
//                      gz_b = sum_d (x/r) V'_d - veta - z_b ; fp64 scalars     (poisson.py:640-649,174-183)
//   col_widek_kernel   gV'_d += (x/r) z_b ; gA'_d += x xi_b gz_b ; gphi_d += x/r over the
//                      panel-CSC work items, float atomics; block 0 packs the row pass's fp64
//                      scalars into the accumulator tail
//
// Scope: Poisson likelihood with the linear decoder (likelihood code 0), modes 0 (full) and 1
// (encode only), canonical (col, val) / (pc_row, pc_val) entry arrays.  The log_transform /
// Bernoulli / mixed contexts and the deterministic mode stay at K <= 64 (spmf_ctx_create and
// spmf_ctx_set_deterministic say so).  Four gather instructions are in flight per wave; KP = 128 runs
// the half-wave kernels further down (two entries per instruction).  This is the general form, not a
// tuned one: 2*KP*4 bytes gathered per stored entry and pass at the rate a wave-per-row loop reaches
// (C2's matrix: 6 - 11 TB/s against 16 for the lane-group kernels at K = 64, profiles/r05_widek_probe.txt).
#include "common.h"
#include "kernels.h"

namespace spmf {

namespace {

template <int VPL>
struct Vec {
  float v[VPL];
};
template <int VPL>
__device__ __forceinline__ Vec<VPL> load_row(const float* __restrict__ base, int row, int KP, int lane) {
  Vec<VPL> o;
  const float* p = base + (size_t)row * KP + lane * VPL;
  if constexpr (VPL == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    o.v[0] = t.x; o.v[1] = t.y;
  } else {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
  }
  return o;
}
template <int VPL>
__device__ __forceinline__ void store_row(float* __restrict__ base, int64_t row, int KP, int lane, const Vec<VPL>& x) {
  float* p = base + (size_t)row * KP + lane * VPL;
  if constexpr (VPL == 2) *reinterpret_cast<float2*>(p) = make_float2(x.v[0], x.v[1]);
  else *reinterpret_cast<float4*>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}

constexpr int kInFlight = 4;   // gather instructions issued back to back
#ifndef WIDEK_HALF128
#define WIDEK_HALF128 1        // KP = 128: the half-wave kernels below (0: the whole-wave form, 8 bytes per lane)
#endif

}  // namespace

template <int KP>
__global__ __launch_bounds__(256) void row_widek_kernel(
    int64_t B, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ row_scale, const float* __restrict__ Ap,
    const float* __restrict__ Vp, const float* __restrict__ phi, const double* __restrict__ dprep,
    float* __restrict__ z, float* __restrict__ gzs, double* __restrict__ dacc, int mode, int Dcols,
    int64_t dacc_stride) {
  constexpr int VPL = KP / 64;
  if (gridDim.y > 1) {   // S draws per launch
    const size_t sd = blockIdx.y;
    Ap += sd * (size_t)Dcols * KP;
    Vp += sd * (size_t)Dcols * KP;
    phi += sd * (size_t)Dcols;
    dprep += sd * (size_t)kPrepSeg * (KP + 1);
    z += sd * (size_t)B * KP;
    gzs += sd * (size_t)B * KP;
    dacc += sd * (size_t)dacc_stride;
  }
  const bool encode_only = mode == 1;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float veta[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) veta[j] = encode_only ? 0.f : (float)prep_sum(dprep, KP, lane * VPL + j);
  double ll_acc = 0.0, zsq_acc = 0.0, nnf_acc = 0.0;   // ll / nnf: wave-uniform; zsq: this lane's k
  float zsum[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) zsum[j] = 0.f;

  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    const float xi = row_scale ? row_scale[b] : 1.f;
    // ---- sweep 1: z_b ---------------------------------------------------------------
    Vec<VPL> zacc;
#pragma unroll
    for (int j = 0; j < VPL; ++j) zacc.v[j] = 0.f;
    for (int base = start; base < end; base += 64) {
      const int i = base + lane;
      const int c = i < end ? col[i] : 0;         // (slots behind the row's end: row 0, weight 0)
      const float x = i < end ? val[i] : 0.f;
      const int cnt = min(64, end - base);
      for (int e0 = 0; e0 < cnt; e0 += kInFlight) {
        Vec<VPL> a[kInFlight];
        float xe[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const int src = min(e0 + j, 63);
          const int cj = __shfl(c, src);
          xe[j] = e0 + j < cnt ? __shfl(x, src) : 0.f;
          a[j] = load_row<VPL>(Ap, cj, KP, lane);
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j)
#pragma unroll
          for (int q = 0; q < VPL; ++q) zacc.v[q] = fmaf(xe[j], a[j].v[q], zacc.v[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < VPL; ++q) zacc.v[q] *= xi;
    store_row<VPL>(z, b, KP, lane, zacc);
    if (encode_only) continue;
    // ---- sweep 2: rates, log-likelihood, gz_b -------------------------------------
    Vec<VPL> gz;
#pragma unroll
    for (int j = 0; j < VPL; ++j) gz.v[j] = 0.f;
    float llrow = 0.f;
    for (int base = start; base < end; base += 64) {
      const int i = base + lane;
      const int c = i < end ? col[i] : 0;
      const float x = i < end ? val[i] : 0.f;
      const int cnt = min(64, end - base);
      for (int e0 = 0; e0 < cnt; e0 += kInFlight) {
        Vec<VPL> vv[kInFlight];
        float xe[kInFlight], ph[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const int src = min(e0 + j, 63);
          const int cj = __shfl(c, src);
          xe[j] = e0 + j < cnt ? __shfl(x, src) : 0.f;
          vv[j] = load_row<VPL>(Vp, cj, KP, lane);
          ph[j] = phi[cj];
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          float d = 0.f;
#pragma unroll
          for (int q = 0; q < VPL; ++q) d = fmaf(zacc.v[q], vv[j].v[q], d);
          const float r = wave_sum(d) + ph[j];
          float cc = 0.f;
          if (xe[j] > 0.f) {                     // wave-uniform
            if (r > 0.f && r < INFINITY) {
              llrow = fmaf(xe[j], logf(r), llrow);
              cc = xe[j] * __builtin_amdgcn_rcpf(r);
            } else {
              // the replacement rule's cell (row_pass.hip sweep2): counted, weight +1 against the
              // closed-form -1 every cell gets
              nnf_acc += 1.0;
              cc = 1.f;
            }
          }
#pragma unroll
          for (int q = 0; q < VPL; ++q) gz.v[q] = fmaf(cc, vv[j].v[q], gz.v[q]);
        }
      }
    }
    Vec<VPL> o;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
      o.v[q] = xi * (gz.v[q] - veta[q] - zacc.v[q]);
      zsq_acc += (double)(zacc.v[q] * zacc.v[q]);
      zsum[q] += zacc.v[q];
    }
    store_row<VPL>(gzs, b, KP, lane, o);
    ll_acc += (double)llrow;
  }
  if (encode_only) return;
  // ---- one set of fp64 atomics per wave, into one of the kDaccRep replicas of the scalar block ----
  dacc += (size_t)(blockIdx.x % kDaccRep) * (kDaccHead + KP);
  const double zq = wave_sum(zsq_acc);
  if (lane == 0) {
    atomicAdd(&dacc[0], ll_acc);
    atomicAdd(&dacc[1], zq);
    if (nnf_acc != 0.0) atomicAdd(&dacc[2], nnf_acc);
  }
#pragma unroll
  for (int q = 0; q < VPL; ++q) atomicAdd(&dacc[kDaccHead + lane * VPL + q], (double)zsum[q]);
}

template <int KP>
__global__ __launch_bounds__(256) void col_widek_kernel(
    int D, int n_panels, int row_base, int blocks_per_panel, const int32_t* __restrict__ item_ptr,
    const int4* __restrict__ items, const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ Vp, const float* __restrict__ phi, const float* __restrict__ z,
    const float* __restrict__ gzs, float* __restrict__ gAp, float* __restrict__ gVp, float* __restrict__ gphi,
    const int32_t* __restrict__ item_mid, int half_sel, int64_t Brows, int64_t acc_stride,
    const double* __restrict__ pack_dacc, float* __restrict__ pack_tail, int64_t dacc_stride) {
  constexpr int VPL = KP / 64;
  if (pack_dacc && blockIdx.x == 0) {
    // the row pass's fp64 scalars -> (hi, lo) float pairs in the accumulator tail (col_pass.hip pack_block)
    const double* dacc = pack_dacc + (size_t)blockIdx.y * dacc_stride;
    float* tail = pack_tail + (size_t)blockIdx.y * acc_stride;
    for (int i = threadIdx.x; i < kDaccHead + KP; i += blockDim.x) {
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < kDaccRep; ++r) v += dacc[(size_t)r * (kDaccHead + KP) + i];
      const float hi = (float)v;
      tail[2 * i] = hi;
      tail[2 * i + 1] = (float)(v - (double)hi);
    }
    return;
  }
  if (gridDim.y > 1) {   // S draws per launch
    const size_t sd = blockIdx.y;
    Vp += sd * (size_t)D * KP;
    phi += sd * (size_t)D;
    z += sd * (size_t)Brows * KP;
    gzs += sd * (size_t)Brows * KP;
    gAp += sd * (size_t)acc_stride;
    gVp += sd * (size_t)acc_stride;
    gphi += sd * (size_t)acc_stride;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t L = (int64_t)blockIdx.x - (pack_dacc ? 1 : 0);
  const int p = (int)(L / blocks_per_panel), ib = (int)(L % blocks_per_panel);
  if (p >= n_panels) return;
  const int ilo = half_sel == 2 ? item_mid[p] : item_ptr[p];
  const int ihi = half_sel == 1 ? item_mid[p] : item_ptr[p + 1];
  const int it = ilo + ib * 4 + wid;               // one work item per wave
  if (it >= ihi) return;                            // wave-uniform
  const int4 im = items[it];
  const int d = im.z;
  const Vec<VPL> vp = load_row<VPL>(Vp, d, KP, lane);
  const float ph = phi[d];
  Vec<VPL> gV, gA;
#pragma unroll
  for (int q = 0; q < VPL; ++q) gV.v[q] = gA.v[q] = 0.f;
  float gph = 0.f;
  const int end = im.x + im.y;
  for (int base = im.x; base < end; base += 64) {
    const int i = base + lane;
    const int rb = i < end ? pc_row[i] - row_base : 0;   // (slots behind the item's end: row 0, weight 0)
    const float x = i < end ? pc_val[i] : 0.f;
    const int cnt = min(64, end - base);
    for (int e0 = 0; e0 < cnt; e0 += kInFlight) {
      Vec<VPL> zz[kInFlight], gg[kInFlight];
      float xe[kInFlight];
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int src = min(e0 + j, 63);
        const int b = __shfl(rb, src);
        xe[j] = e0 + j < cnt ? __shfl(x, src) : 0.f;
        zz[j] = load_row<VPL>(z, b, KP, lane);
        gg[j] = load_row<VPL>(gzs, b, KP, lane);
      }
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        float dd = 0.f;
#pragma unroll
        for (int q = 0; q < VPL; ++q) dd = fmaf(zz[j].v[q], vp.v[q], dd);
        const float r = wave_sum(dd) + ph;
        // (col_pass.hip: a cell the row pass counted as non-finite gets weight +1; padded slots stay weightless)
        const float xr = (r > 0.f && r < INFINITY) ? xe[j] * __builtin_amdgcn_rcpf(r) : (xe[j] > 0.f ? 1.f : 0.f);
#pragma unroll
        for (int q = 0; q < VPL; ++q) {
          gV.v[q] = fmaf(xr, zz[j].v[q], gV.v[q]);
          gA.v[q] = fmaf(xe[j], gg[j].v[q], gA.v[q]);
        }
        gph += xr;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < VPL; ++q) {
    const size_t o = (size_t)d * KP + lane * VPL + q;
    if (gV.v[q] != 0.f) atomicAdd(&gVp[o], gV.v[q]);
    if (gA.v[q] != 0.f) atomicAdd(&gAp[o], gA.v[q]);
  }
  if (lane == 0 && gph != 0.f) atomicAdd(&gphi[d], gph);
}

// ---- KP = 128: two stored entries per wave instruction ------------------------------------------------------
// A 512-byte factor row is 32 lanes x float4, so the two halves of a wave take alternate entries (an 8-byte-per-lane
// whole-wave row, the general form above, moves half the bytes per gather instruction: 0.76 against 0.5x ms per row
// pass on C2's matrix).  Half h of the wave owns entries e0 + 2j + h; partial sums of the halves meet through one
// cross-half exchange per row / item.
namespace {
__device__ __forceinline__ float half_sum(float v) {     // sum over the aligned 32 lanes, in every lane of them
  v = group_sum<16>(v);
  return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float4 xhalf_add(float4 v) {  // + the other half's value
  return make_float4(v.x + __shfl_xor(v.x, 32), v.y + __shfl_xor(v.y, 32), v.z + __shfl_xor(v.z, 32),
                     v.w + __shfl_xor(v.w, 32));
}
__device__ __forceinline__ float4 ld4(const float* __restrict__ base, int row, int sub) {
  return *reinterpret_cast<const float4*>(base + (size_t)row * 128 + sub * 4);
}
}  // namespace

__global__ __launch_bounds__(256) void row_widek128_kernel(
    int64_t B, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ row_scale, const float* __restrict__ Ap,
    const float* __restrict__ Vp, const float* __restrict__ phi, const double* __restrict__ dprep,
    float* __restrict__ z, float* __restrict__ gzs, double* __restrict__ dacc, int mode, int Dcols,
    int64_t dacc_stride) {
  constexpr int KP = 128;
  if (gridDim.y > 1) {   // S draws per launch
    const size_t sd = blockIdx.y;
    Ap += sd * (size_t)Dcols * KP;
    Vp += sd * (size_t)Dcols * KP;
    phi += sd * (size_t)Dcols;
    dprep += sd * (size_t)kPrepSeg * (KP + 1);
    z += sd * (size_t)B * KP;
    gzs += sd * (size_t)B * KP;
    dacc += sd * (size_t)dacc_stride;
  }
  const bool encode_only = mode == 1;
  const int lane = threadIdx.x & 63, sub = lane & 31, h = lane >> 5;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float4 veta = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!encode_only)
    veta = make_float4((float)prep_sum(dprep, KP, sub * 4 + 0), (float)prep_sum(dprep, KP, sub * 4 + 1),
                       (float)prep_sum(dprep, KP, sub * 4 + 2), (float)prep_sum(dprep, KP, sub * 4 + 3));
  double ll_acc = 0.0, zsq_acc = 0.0, nnf_acc = 0.0;   // per lane: ll / nnf of this half's cells, zsq of half 0's k
  float4 zsum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    const float xi = row_scale ? row_scale[b] : 1.f;
    float4 zacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = start; base < end; base += 64) {
      const int i = base + lane;
      const int c = i < end ? col[i] : 0;
      const float x = i < end ? val[i] : 0.f;
      const int cnt = min(64, end - base);
      for (int e0 = 0; e0 < cnt; e0 += 2 * kInFlight) {
        float4 a[kInFlight];
        float xe[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const int idx = e0 + 2 * j + h, src = min(idx, 63);
          const int cj = __shfl(c, src);
          const float xs = __shfl(x, src);
          xe[j] = idx < cnt ? xs : 0.f;
          a[j] = ld4(Ap, cj, sub);
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) zacc = fma4(xe[j], a[j], zacc);
      }
    }
    zacc = xhalf_add(zacc);
    zacc.x *= xi; zacc.y *= xi; zacc.z *= xi; zacc.w *= xi;
    if (h == 0) reinterpret_cast<float4*>(z + (size_t)b * KP)[sub] = zacc;
    if (encode_only) continue;
    float4 gz = make_float4(0.f, 0.f, 0.f, 0.f);
    float llrow = 0.f;
    for (int base = start; base < end; base += 64) {
      const int i = base + lane;
      const int c = i < end ? col[i] : 0;
      const float x = i < end ? val[i] : 0.f;
      const int cnt = min(64, end - base);
      for (int e0 = 0; e0 < cnt; e0 += 2 * kInFlight) {
        float4 vv[kInFlight];
        float xe[kInFlight], ph[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const int idx = e0 + 2 * j + h, src = min(idx, 63);
          const int cj = __shfl(c, src);
          const float xs = __shfl(x, src);
          xe[j] = idx < cnt ? xs : 0.f;
          vv[j] = ld4(Vp, cj, sub);
          ph[j] = phi[cj];
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
          const float r = half_sum(dot4(zacc, vv[j])) + ph[j];
          const bool on = xe[j] > 0.f, good = r > 0.f && r < INFINITY;   // uniform inside a half
          // (a cell with a non-positive rate: counted, weight +1 against the closed-form -1 -- row_pass.hip sweep2)
          const float cc = on ? (good ? xe[j] * __builtin_amdgcn_rcpf(r) : 1.f) : 0.f;
          if (on && good) llrow = fmaf(xe[j], logf(r), llrow);
          if (on && !good) nnf_acc += 1.0;
          gz = fma4(cc, vv[j], gz);
        }
      }
    }
    gz = xhalf_add(gz);
    if (h == 0) {
      float4 o;
      o.x = xi * (gz.x - veta.x - zacc.x);
      o.y = xi * (gz.y - veta.y - zacc.y);
      o.z = xi * (gz.z - veta.z - zacc.z);
      o.w = xi * (gz.w - veta.w - zacc.w);
      reinterpret_cast<float4*>(gzs + (size_t)b * KP)[sub] = o;
      zsq_acc += (double)dot4(zacc, zacc);
      zsum = add4(zsum, zacc);
    }
    ll_acc += (double)llrow;
  }
  if (encode_only) return;
  dacc += (size_t)(blockIdx.x % kDaccRep) * (kDaccHead + KP);
  // ll / nnf: every lane of a half carries that half's sum -- lanes 0 and 32 hold the two halves' values
  const double zq = wave_sum(zsq_acc);
  if (sub == 0) {
    atomicAdd(&dacc[0], ll_acc);
    if (nnf_acc != 0.0) atomicAdd(&dacc[2], nnf_acc);
  }
  if (lane == 0) atomicAdd(&dacc[1], zq);
  if (h == 0) {
    atomicAdd(&dacc[kDaccHead + sub * 4 + 0], (double)zsum.x);
    atomicAdd(&dacc[kDaccHead + sub * 4 + 1], (double)zsum.y);
    atomicAdd(&dacc[kDaccHead + sub * 4 + 2], (double)zsum.z);
    atomicAdd(&dacc[kDaccHead + sub * 4 + 3], (double)zsum.w);
  }
}

__global__ __launch_bounds__(256) void col_widek128_kernel(
    int D, int n_panels, int row_base, int blocks_per_panel, const int32_t* __restrict__ item_ptr,
    const int4* __restrict__ items, const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ Vp, const float* __restrict__ phi, const float* __restrict__ z,
    const float* __restrict__ gzs, float* __restrict__ gAp, float* __restrict__ gVp, float* __restrict__ gphi,
    const int32_t* __restrict__ item_mid, int half_sel, int64_t Brows, int64_t acc_stride,
    const double* __restrict__ pack_dacc, float* __restrict__ pack_tail, int64_t dacc_stride) {
  constexpr int KP = 128;
  if (pack_dacc && blockIdx.x == 0) {
    const double* dacc = pack_dacc + (size_t)blockIdx.y * dacc_stride;
    float* tail = pack_tail + (size_t)blockIdx.y * acc_stride;
    for (int i = threadIdx.x; i < kDaccHead + KP; i += blockDim.x) {
      double v = 0.0;
#pragma unroll
      for (int r = 0; r < kDaccRep; ++r) v += dacc[(size_t)r * (kDaccHead + KP) + i];
      const float hi = (float)v;
      tail[2 * i] = hi;
      tail[2 * i + 1] = (float)(v - (double)hi);
    }
    return;
  }
  if (gridDim.y > 1) {
    const size_t sd = blockIdx.y;
    Vp += sd * (size_t)D * KP;
    phi += sd * (size_t)D;
    z += sd * (size_t)Brows * KP;
    gzs += sd * (size_t)Brows * KP;
    gAp += sd * (size_t)acc_stride;
    gVp += sd * (size_t)acc_stride;
    gphi += sd * (size_t)acc_stride;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane & 31, h = lane >> 5;
  const int64_t L = (int64_t)blockIdx.x - (pack_dacc ? 1 : 0);
  const int p = (int)(L / blocks_per_panel), ib = (int)(L % blocks_per_panel);
  if (p >= n_panels) return;
  const int ilo = half_sel == 2 ? item_mid[p] : item_ptr[p];
  const int ihi = half_sel == 1 ? item_mid[p] : item_ptr[p + 1];
  const int it = ilo + ib * 4 + wid;
  if (it >= ihi) return;
  const int4 im = items[it];
  const int d = im.z;
  const float4 vp = ld4(Vp, d, sub);
  const float ph = phi[d];
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;
  const int end = im.x + im.y;
  for (int base = im.x; base < end; base += 64) {
    const int i = base + lane;
    const int rb = i < end ? pc_row[i] - row_base : 0;
    const float x = i < end ? pc_val[i] : 0.f;
    const int cnt = min(64, end - base);
    for (int e0 = 0; e0 < cnt; e0 += 2 * kInFlight) {
      float4 zz[kInFlight], gg[kInFlight];
      float xe[kInFlight];
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int idx = e0 + 2 * j + h, src = min(idx, 63);
        const int b = __shfl(rb, src);
        const float xs = __shfl(x, src);
        xe[j] = idx < cnt ? xs : 0.f;
        zz[j] = ld4(z, b, sub);
        gg[j] = ld4(gzs, b, sub);
      }
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const float r = half_sum(dot4(zz[j], vp)) + ph;
        const float xr = (r > 0.f && r < INFINITY) ? xe[j] * __builtin_amdgcn_rcpf(r) : (xe[j] > 0.f ? 1.f : 0.f);
        gV = fma4(xr, zz[j], gV);
        gA = fma4(xe[j], gg[j], gA);
        gph += xr;
      }
    }
  }
  gV = xhalf_add(gV);
  gA = xhalf_add(gA);
  gph += __shfl_xor(gph, 32);
  if (h == 0) {
    float* dv = gVp + (size_t)d * KP + sub * 4;
    float* da = gAp + (size_t)d * KP + sub * 4;
    if (gV.x != 0.f) atomicAdd(dv + 0, gV.x);
    if (gV.y != 0.f) atomicAdd(dv + 1, gV.y);
    if (gV.z != 0.f) atomicAdd(dv + 2, gV.z);
    if (gV.w != 0.f) atomicAdd(dv + 3, gV.w);
    if (gA.x != 0.f) atomicAdd(da + 0, gA.x);
    if (gA.y != 0.f) atomicAdd(da + 1, gA.y);
    if (gA.z != 0.f) atomicAdd(da + 2, gA.z);
    if (gA.w != 0.f) atomicAdd(da + 3, gA.w);
    if (lane == 0 && gph != 0.f) atomicAdd(&gphi[d], gph);
  }
}

// false: not a shape this file covers (nothing launched)
bool launch_row_widek(int KP, const RowArgs& a, hipStream_t st) {
  if (a.logt != 0 || (a.mode != 0 && a.mode != 1) || a.det_slots || a.dual) return false;
  const int64_t want = (a.B + 3) / 4;
  const int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
#define SPMF_ROWW(KP_)                                                                                     \
  hipLaunchKernelGGL((row_widek_kernel<KP_>), dim3(nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.B, a.row_ptr, \
                     a.col, a.val, a.row_scale, a.Ap, a.Vp, a.phi, a.dprep, a.z, a.gzs, a.dacc, a.mode, a.D, \
                     a.dacc_stride)
  switch (KP) {
    case 128:
#if WIDEK_HALF128
      hipLaunchKernelGGL(row_widek128_kernel, dim3(nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.B, a.row_ptr, a.col,
                         a.val, a.row_scale, a.Ap, a.Vp, a.phi, a.dprep, a.z, a.gzs, a.dacc, a.mode, a.D,
                         a.dacc_stride);
#else
      SPMF_ROWW(128);
#endif
      return true;
    case 256: SPMF_ROWW(256); return true;
    default: return false;
  }
#undef SPMF_ROWW
}

bool launch_col_widek(int KP, const ColArgs& a, hipStream_t st) {
  if (a.logt != 0 || a.det_part) return false;
  const int bpp = (a.max_items_per_panel + 3) / 4;
  if (bpp < 1) return false;
  const int64_t nb = (int64_t)a.n_panels * bpp + (a.pack_dacc ? 1 : 0);
  const int4* items = reinterpret_cast<const int4*>(a.items);
#define SPMF_COLW(KP_)                                                                                      \
  hipLaunchKernelGGL((col_widek_kernel<KP_>), dim3((unsigned)nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.D,   \
                     a.n_panels, a.row_base, bpp, a.item_ptr, items, a.pc_row, a.pc_val, a.Vp, a.phi, a.z,    \
                     a.gzs, a.gAp, a.gVp, a.gphi, a.item_mid, a.half_sel, a.B, a.acc_stride, a.pack_dacc,     \
                     a.pack_tail, a.dacc_stride)
  switch (KP) {
    case 128:
#if WIDEK_HALF128
      hipLaunchKernelGGL(col_widek128_kernel, dim3((unsigned)nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.D,
                         a.n_panels, a.row_base, bpp, a.item_ptr, items, a.pc_row, a.pc_val, a.Vp, a.phi, a.z, a.gzs,
                         a.gAp, a.gVp, a.gphi, a.item_mid, a.half_sel, a.B, a.acc_stride, a.pack_dacc, a.pack_tail,
                         a.dacc_stride);
#else
      SPMF_COLW(128);
#endif
      return true;
    case 256: SPMF_COLW(256); return true;
    default: return false;
  }
#undef SPMF_COLW
}

}  // namespace spmf
