// common.h -- shared device helpers for libspmf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPMF_WAVE 64

namespace spmf {

constexpr double kHalfLog2OverPi = -0.22579135264472743236;  // 0.5*log(2/pi)
constexpr double kLgammaHalf = 0.57236494292470008707;      // lgamma(0.5)
constexpr double kLog2 = 0.69314718055994530942;

// Layout of the fp64 scalar block a data pass accumulates per draw.
//   [0] sum_nnz x*log r   [1] sum z^2   [2] non-finite stored cells
//   [3] dense sum (sum E / sum softplus)   [4] saturated cells (log_transform)
//   [5] reserved          [6 .. 6+KP)   sum_b z_b
constexpr int kDaccHead = 6;
// log_transform decoder (poisson.py:52-53): f(y) = exp(y) - 1 is evaluated as
// exp(min(y, kYSat)) - 1.  fp32 cannot hold exp(y) beyond y ~ 88.7 (the fp64
// reference overflows at 709); saturating keeps every sum and gradient of a
// step finite (70: D * e^70 * |V'| stays far below FLT_MAX for any supported D),
// so a batch with a few runaway cells still trains -- its gradient pushes their
// exponents down -- instead of being skipped.  Saturation events are counted
// (dacc[4], workgroup granularity); while that count is 0 the decoder is exact.
// Likelihood code of a context (api.hip likelihood_code): 0 Poisson / linear decoder, 1 Poisson /
// exp decoder (log_transform), 2 Bernoulli(logits) / linear, 3 mixed per column / linear,
// 4 Bernoulli(logits) / exp decoder (bernoulli.py:60-61: logit = exp(<z, eta v>) - 1 + phi).
__host__ __device__ constexpr bool lik_exp(int l) { return l == 1 || l == 4; }
__host__ __device__ constexpr bool lik_bern(int l) { return l == 2 || l == 4; }

constexpr float kYSat = 70.0f;
// exp(y) - 1 of the log_transform decoder (poisson.py:52-53) for an exponent already clamped
// at kYSat; `ey` returns exp(y) (the derivative).  The difference formed in fp32 loses its
// leading digits for small |y| -- a rarely expressed gene has eta = 1e-3 and y ~ 1e-5, where
// expf(y) - 1.f keeps two digits -- while the fp64 reference keeps them all: the rate
// r = exp(y) - 1 + phi of such a column, and with it x/r in three gradients, was off by up to
// 1e-3 relative (found by the entry-wise gradient check on the C4 slice).  Series below 1/4
// (next term y^8/9! < 5e-11 relative), exp - 1 above (relative error <= 4.6 ulp there).
__device__ __forceinline__ float expm1_dec(float y, float& ey) {
  float p = 1.f / 40320.f;
  p = fmaf(p, y, 1.f / 5040.f);
  p = fmaf(p, y, 1.f / 720.f);
  p = fmaf(p, y, 1.f / 120.f);
  p = fmaf(p, y, 1.f / 24.f);
  p = fmaf(p, y, 1.f / 6.f);
  p = fmaf(p, y, 0.5f);
  p = fmaf(p, y, 1.f);
  const float e = expf(y);
  const bool small = fabsf(y) < 0.25f;
  const float em1 = small ? y * p : e - 1.f;
  ey = small ? 1.f + em1 : e;
  return em1;
}

// The prep kernel's closed-form column sums (veta[KP], phisum) are written as kPrepSeg
// partial sums over column segments, dprep[seg][KP+1]: one writer per slot (no atomics,
// no zero fill), and every reader folds the segments in index order (prep_sum).
constexpr int kPrepSeg = 8;
// columns per workgroup of the finish kernel (the per-block slots ppart/putau are sized by it)
#ifndef FINISH_FTD
#define FINISH_FTD 32
#endif
constexpr int kFinishCols = FINISH_FTD;
// the block sums land in one of kDaccRep replicas (blockIdx % kDaccRep) so the
// fp64 atomics of thousands of blocks do not serialise on 4+KP addresses;
// the pack kernel folds the replicas.
constexpr int kDaccRep = 16;
// Deterministic mode (spmf_ctx_set_deterministic): per-workgroup scalar slots of the row pass instead of
// its fp64 atomics -- at most this many workgroups (ROW_MAX_BLOCKS), slot -1 = the launch's workgroup count;
// per-item partial sums of the column pass: 2*KP + 4 floats per work item (gV', gA', gphi + padding)
constexpr int kDetMaxBlocks = 4096;
constexpr int kDetMeta = 8;      // doubles in front of the slots ([0] = workgroups of the row-pass launch)
__host__ __device__ inline int det_part_len(int KP) { return 2 * KP + 4; }

// fp32 accumulator tail: the fp64 scalars as (hi,lo) float pairs.
__host__ __device__ inline int acc_tail_len(int KP) { return 2 * (kDaccHead + KP); }
// Packed accumulators of one draw: [H0: gA' | gV' | gphi][H1: gA' | gV' | gphi][tail],
// H0 = columns [0, Dh), H1 = [Dh, D).  Dh = D (no column split) is the plain
// [gA' | gV' | gphi | tail].  With a split the two halves are contiguous ranges, so
// the multi-GPU step can all-reduce H0 while the column pass still produces H1.
struct AccLayout {
  int D, KP, Dh;
  __host__ __device__ int64_t half_len(int h) const {
    return (int64_t)(h ? D - Dh : Dh) * (2 * KP + 1);
  }
  // base pointers such that base[(size_t)d * KP + k] (or base[d] for gphi) addresses
  // column d of half h directly
  __host__ __device__ int64_t gA_off(int h) const { return h ? half_len(0) - (int64_t)Dh * KP : 0; }
  __host__ __device__ int64_t gV_off(int h) const {
    return h ? half_len(0) + (int64_t)(D - Dh) * KP - (int64_t)Dh * KP : (int64_t)Dh * KP;
  }
  __host__ __device__ int64_t gphi_off(int h) const {
    return h ? half_len(0) + (int64_t)2 * (D - Dh) * KP - Dh : (int64_t)2 * Dh * KP;
  }
  __host__ __device__ int64_t tail_off() const { return (int64_t)2 * D * KP + D; }
};
__host__ __device__ inline int64_t acc_len(int D, int KP) {
  return (int64_t)2 * D * KP + D + acc_tail_len(KP);
}

__device__ __forceinline__ double prep_sum(const double* __restrict__ dprep, int KP, int i) {
  double t = 0.0;
#pragma unroll
  for (int g = 0; g < kPrepSeg; ++g) t += dprep[(size_t)g * (KP + 1) + i];
  return t;
}

__device__ __forceinline__ float4 shfl4(float4 v, int src) {
  return make_float4(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src), __shfl(v.w, src));
}
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m),
                     __shfl_xor(v.w, m));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
  return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z),
                     fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Gather the `sub`-th float4 of row `row` of a [rows, 4*LPN]-float table with
// a 32-bit byte offset (tables are < 4 GiB, checked on the host): lets the
// compiler use the saddr + 32-bit voffset form instead of 64-bit VALU
// address arithmetic per gather.
template <int LPN>
__device__ __forceinline__ float4 gather4(const float* __restrict__ base, int row, int sub) {
  const uint32_t off = ((uint32_t)row * (uint32_t)LPN + (uint32_t)sub) * 16u;
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}

// The same gather through a buffer descriptor whose record count is the table's size.  What it buys is the
// hardware range check: a lane whose offset lies behind the table is DROPPED by the address unit -- it returns
// zeros without a tag lookup in the vector cache and without a request to L2 -- so the padded slots of a row's
// or list's last group of gathers (they used to read row 0 with weight 0: an L1 hit, but still one of the
// vector cache's 64-B tag cycles per four lanes, and the sparse passes run that pipe at 98 % of its cycles:
// DESIGN.md section 4, round 5 (e)) cost an issue slot and nothing else, with no branch and no exec mask in
// the straight-line groups of loads.  A padded slot asks for row kPadRow: -1 wraps to the last KP*4 bytes
// below 4 GiB, behind every table the host admits (api.hip checks rows*KP*4 < 4 GiB - KP*4).
#ifndef SPMF_OOB_PAD
#define SPMF_OOB_PAD 1
#endif
#if SPMF_OOB_PAD
constexpr int kPadRow = -1;
constexpr uint32_t kPadWord = 0xffff0000u;     // packed entry (column or panel row 65 535, count 0)
struct GTable {
  __amdgpu_buffer_rsrc_t r;
};
// base and rows must be wave-uniform (kernel arguments, blockIdx)
__device__ __forceinline__ GTable gtable(const float* base, int64_t rows, int KP) {
  GTable t;
  t.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)((uint64_t)rows * KP * 4u),
                                          0x00020000);
  return t;
}
template <int LPN>
__device__ __forceinline__ float4 gather4(const GTable& t, int row, int sub) {
  const uint32_t off = ((uint32_t)row * (uint32_t)LPN + (uint32_t)sub) * 16u;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(t.r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#else
constexpr int kPadRow = 0;
constexpr uint32_t kPadWord = 0u;
struct GTable {
  const float* p;
};
__device__ __forceinline__ GTable gtable(const float* base, int64_t, int) { return GTable{base}; }
template <int LPN>
__device__ __forceinline__ float4 gather4(const GTable& t, int row, int sub) {
  return gather4<LPN>(t.p, row, sub);
}
#endif

// ---- DPP cross-lane adds (no LDS traffic, fold into v_add_f32_dpp) --------
// ctrl: quad_perm[1,0,3,2]=0xB1 (xor 1), quad_perm[2,3,0,1]=0x4E (xor 2),
// row_half_mirror=0x141 (i <-> 7-i), row_mirror=0x140 (i <-> 15-i),
// row_ror:8=0x128.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(t);
}
// keep + (send of the DPP partner)
template <int CTRL>
__device__ __forceinline__ float dpp_add_to(float keep, float send) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true);
  return keep + __int_as_float(t);
}
// <a, b> with the packed forms (v_pk_mul_f32, v_pk_fma_f32, one add: three instructions instead of four)
__device__ __forceinline__ float dot4p(const float4& a, const float4& b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p = f2{a.x, a.y} * f2{b.x, b.y};
  p = __builtin_elementwise_fma(f2{a.z, a.w}, f2{b.z, b.w}, p);
  return p.x + p.y;
}
// Sum over the aligned group of N lanes (N = 1,2,4,8,16); every lane of the
// group ends up with the total.
template <int N>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (N >= 2) v = dpp_add<0xB1>(v);
  if constexpr (N >= 4) v = dpp_add<0x4E>(v);
  if constexpr (N >= 8) v = dpp_add<0x141>(v);
  if constexpr (N >= 16) v = dpp_add<0x140>(v);
  return v;
}
// Sum over the 64/LPN groups of a wave for lanes holding the same `sub`
// (lane = grp*LPN + sub): xor butterflies over the lane bits >= log2(LPN).
template <int LPN>
__device__ __forceinline__ float across_groups_sum(float v) {
  // inside a 16-lane DPP row: rotate-and-add by LPN, 2*LPN, ... , 8
  if constexpr (LPN <= 1) v = dpp_add<0x121>(v);  // row_ror:1
  if constexpr (LPN <= 2) v = dpp_add<0x122>(v);  // row_ror:2
  if constexpr (LPN <= 4) v = dpp_add<0x124>(v);  // row_ror:4
  if constexpr (LPN <= 8) v = dpp_add<0x128>(v);  // row_ror:8
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
template <int LPN>
__device__ __forceinline__ float4 across_groups_sum4(float4 v) {
  return make_float4(across_groups_sum<LPN>(v.x), across_groups_sum<LPN>(v.y),
                     across_groups_sum<LPN>(v.z), across_groups_sum<LPN>(v.w));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Block-wide fp64 sum of `v`; result valid in thread 0. `red` >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

}  // namespace spmf
