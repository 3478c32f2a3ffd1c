// dense3f.hip -- the Bernoulli / mixed dense sums in ONE kernel: the sigmoid of a tile is computed once and
// feeds BOTH contractions (dense3.hip's sigdot3 form computes it in two launches, (Z, W) and (W, Z)).
//
//   l_pq   = <P_p, Q_q> + bias_q                 P = z rows (the batch), Q = V' rows of the Bernoulli columns
//   E_pq   = sigmoid(l_pq)
//   out_p += sign  * p_scale_p * sum_q E_pq Q_q  (the dense row term, subtracted from gzs: fused row pass)
//   gV_q  += sign2 * sum_p E_pq P_p              (d/dV' of the softplus sum)
//   gphi_q += sign2 * sum_p E_pq                 (d/dphi)
//   esum  += sum softplus(l_pq)                  (part 'x')
// (mederrata_spmf/bernoulli.py:127-155: ll = x l - softplus(l).)  K padded to 32 only.
//
// A wave owns PBW = 4 blocks of 32 P rows (their out accumulators stay in registers for the whole kernel);
// a 256-thread workgroup (4 waves, 512 P rows) stages 128-row Q tiles in LDS as three bf16 planes, like
// sigdot3.  Per (P block, 32-row Q sub-tile):
//   A   X = Q P^T                 12 MFMAs, bias in the accumulator's initial value, lane = p, registers = q
//   E   sigmoid / softplus, E split into three bf16 planes (V' has mixed signs: dense3.hip)
//   C   out^T += Q^T E            12 MFMAs, E straight from the registers (the accumulator layout IS the B operand)
//   T   E (two planes) to the wave's own LDS scratch as [p][q'] rows, read back column-wise with the
//       transposing load: lane = q', registers = p  -- the layout the second contraction needs
//   D   G[q', k] = sum_p E P      10 MFMAs against P^T planes prepared by the split kernel, + 4 against ones
//       for the row sums; G is added into the workgroup's LDS tile (ds_add_f32), flushed to gV / gphi
//       with float atomics once per Q tile: 512 P rows per flush = B/512 * D * (K+1) * 4 bytes (0.25 GB on C5).
// The P operands (both layouts, three planes) come pre-split from split kernel output in the workspace, so
// reloading them per (P block, Q tile) costs loads, not vector instructions.
#include "common.h"
#include "kernels.h"

namespace spmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }

constexpr int kQT = 128;       // Q rows per staged tile
constexpr int kQPitch = 80;    // bytes per row of a Q plane (32 bf16 + 16: the 16 rows of a b128 group cover all banks)
constexpr int kEPitch = 80;    // bytes per row of a wave's E scratch plane ([p][q'], 32 bf16 + 16)
constexpr int kNW = 4;         // waves per workgroup
constexpr int kPBW = 4;        // P blocks (32 rows) per wave
constexpr int kKD = 32;
}  // namespace

// P [NP][32] fp32 -> three bf16 planes in the two layouts the fused kernel reads:
//   pz[plane][np_pad][32]                          rows as they are (B operand of X = Q P^T)
//   zt[block][plane][s2][h][k][8]                  the block's 32 rows transposed, 8 rows per lane in the k order
//                                                   of an accumulator tile (B operand of G = E^T P)
__global__ __launch_bounds__(256) void sigf_split_kernel(int NP, int np_pad, const float* __restrict__ P,
                                                          uint16_t* __restrict__ pz, uint16_t* __restrict__ zt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)np_pad * kKD) return;
  const int p = (int)(i / kKD), k = (int)(i % kKD);
  float v = p < NP ? P[(size_t)p * kKD + k] : 0.f;
  uint16_t pl[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const uint32_t pk = pack_bf16(v, 0.f);
    pl[j] = (uint16_t)(pk & 0xffffu);
    v -= bf16_lo(pk);
  }
  const int b = p >> 5, pl_ = p & 31, s2 = pl_ >> 4, rr = pl_ & 15, blk = rr >> 2;
  const int h = blk & 1, j8 = (blk >> 1) * 4 + (rr & 3);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    pz[((size_t)j * np_pad + p) * kKD + k] = pl[j];
    zt[(((((size_t)b * 3 + j) * 2 + s2) * 2 + h) * kKD + k) * 8 + j8] = pl[j];
  }
}

template <bool EDGE>
__device__ __forceinline__ void sigf_unit(const unsigned char* __restrict__ qimg, uint32_t q_tr_base,
                                          unsigned char* __restrict__ epw, uint32_t ep_tr_base,
                                          const float* __restrict__ bq, const uint16_t* __restrict__ pzb, int np_pad,
                                          const uint16_t* __restrict__ ztb, f32x16& acc, f32x16& g, f32x16& gs,
                                          float& es_tile, int sub, int r, int h, int q0, int NQ, bool p_in) {
  constexpr int IMG = kQT * kQPitch, EIMG = 32 * kEPitch;
  // the block's P planes (rows as they are), L2 resident: B operand of X = Q P^T
  bf16x8 pbz[2][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      pbz[s][pl] = *reinterpret_cast<const bf16x8*>(pzb + ((size_t)pl * np_pad + r) * kKD + 16 * s + 8 * h);
  // ---- A: X = Q P^T (+ bias) --------------------------------------------------------------------
  f32x16 x;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 b4 = *reinterpret_cast<const float4*>(bq + 32 * sub + 8 * g + 4 * h);
    x[4 * g + 0] = b4.x; x[4 * g + 1] = b4.y; x[4 * g + 2] = b4.z; x[4 * g + 3] = b4.w;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 a[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      a[pl] = *reinterpret_cast<const bf16x8*>(qimg + pl * IMG + (32 * sub + r) * kQPitch + (16 * s + 8 * h) * 2);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pbz[s][0], x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pbz[s][1], x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pbz[s][2], x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pbz[s][0], x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pbz[s][1], x, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pbz[s][0], x, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- E: sigmoid, softplus sum, three-plane split ------------------------------------------------
  u32x4 eb[2][3];
  {
    constexpr float kLog2e = 1.4426950408889634f;
    float pm0 = 0.f, pm1 = 0.f, dp0 = 1.f, dp1 = 1.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      float l0 = x[i], l1 = x[i + 1];
      if (EDGE) {
        const int qa = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
        l0 = (p_in && qa < NQ) ? l0 : -INFINITY;
        l1 = (p_in && qa + 1 < NQ) ? l1 : -INFINITY;
      }
      const float en0 = __builtin_amdgcn_exp2f(-fabsf(l0) * kLog2e);
      const float en1 = __builtin_amdgcn_exp2f(-fabsf(l1) * kLog2e);
      const float d0 = 1.f + en0, d1 = 1.f + en1;
      const float i0 = __builtin_amdgcn_rcpf(d0), i1 = __builtin_amdgcn_rcpf(d1);
      const bool g0 = l0 >= 0.f, g1 = l1 >= 0.f;
      const float s0 = (g0 ? 1.f : en0) * i0, s1 = (g1 ? 1.f : en1) * i1;
      pm0 += g0 ? l0 : 0.f;
      pm1 += g1 ? l1 : 0.f;
      dp0 *= d0;
      dp1 *= d1;
      const uint32_t p1 = pack_bf16(s0, s1);
      const float r0 = s0 - bf16_lo(p1), r1 = s1 - bf16_hi(p1);
      const uint32_t p2 = pack_bf16(r0, r1);
      eb[i >> 3][0][(i & 7) >> 1] = p1;
      eb[i >> 3][1][(i & 7) >> 1] = p2;
      eb[i >> 3][2][(i & 7) >> 1] = pack_bf16(r0 - bf16_lo(p2), r1 - bf16_hi(p2));
    }
    es_tile += (pm0 + pm1) + 0.69314718056f * (__builtin_amdgcn_logf(dp0) + __builtin_amdgcn_logf(dp1));
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- T (store): planes 1 and 2 of E as row p of the wave's scratch image [p][q' = 16 h + i] ---------
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    unsigned char* row = epw + pl * EIMG + r * kEPitch + h * 32;
    *reinterpret_cast<u32x4*>(row) = eb[0][pl];
    *reinterpret_cast<u32x4*>(row + 16) = eb[1][pl];
  }
  // ---- C: out^T += Q^T E --------------------------------------------------------------------------
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    bf16x8 a[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const uint32_t ad = q_tr_base + pl * IMG + (32 * sub + 16 * s2) * kQPitch;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)ad);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(uintptr_t)(ad + 8 * kQPitch));
      a[pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    const bf16x8 e1 = __builtin_bit_cast(bf16x8, eb[s2][0]), e2 = __builtin_bit_cast(bf16x8, eb[s2][1]);
    const bf16x8 e3 = __builtin_bit_cast(bf16x8, eb[s2][2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], e1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e1, acc, 0, 0, 0);
  }
  // ---- D: G[q', k] = sum_p E[q', p] P[p, k], row sums of E against a plane of ones ---------------------
  __builtin_amdgcn_wave_barrier();          // the wave's own LDS writes above are read back below
  __builtin_amdgcn_sched_barrier(0);        // (phases stay apart: the live ranges of x / eb / g must not overlap)
  // the block's P^T planes for this product only (6 KB per block, L2 resident): [plane][s2][h][k][8]
  bf16x8 pbt[2][3];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      pbt[s2][pl] = *reinterpret_cast<const bf16x8*>(ztb + ((((size_t)pl * 2 + s2) * 2 + h) * kKD + r) * 8);
  bf16x8 e[2][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const uint32_t ad = ep_tr_base + pl * EIMG + (16 * s2) * kEPitch;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)ad);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(uintptr_t)(ad + 8 * kEPitch));
      e[s2][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  __builtin_amdgcn_wave_barrier();          // the scratch image may be overwritten by the next unit from here on
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][1], pbt[s2][1], g, 0, 0, 0);
    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][0], pbt[s2][2], g, 0, 0, 0);
    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][1], pbt[s2][0], g, 0, 0, 0);
    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][0], pbt[s2][1], g, 0, 0, 0);
    g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][0], pbt[s2][0], g, 0, 0, 0);
    gs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][1], ones, gs, 0, 0, 0);
    gs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e[s2][0], ones, gs, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(kNW * 64, 2) void sigfused3_kernel(
    int NP, int NQ, int np_pad, const uint16_t* __restrict__ pz, const uint16_t* __restrict__ zt,
    const float* __restrict__ Q, const float* __restrict__ bias_q, float* __restrict__ out, float sign,
    const float* __restrict__ p_scale, int atomic_out, double* __restrict__ esum, float* __restrict__ gV,
    float* __restrict__ gphi, const int32_t* __restrict__ out_rows, float sign2) {
  constexpr int IMG = kQT * kQPitch, EIMG = 32 * kEPitch, NT = kNW * 64;
  constexpr int NPC = kQT * (kKD / 8) / NT;       // (row, 8 k) pieces per loader thread: 2
  __shared__ __attribute__((aligned(16))) unsigned char qimg[3][IMG];
  __shared__ __attribute__((aligned(16))) unsigned char ep[kNW][2][EIMG];
  __shared__ __attribute__((aligned(16))) float bqs[kQT];
  __shared__ __attribute__((aligned(16))) float gslot[kNW][32 * 32 + 32];   // a wave's G tile + row sums of one sub-tile
  __shared__ double red[16];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int blk0 = (blockIdx.x * kNW + wid) * kPBW;          // first 32-row P block of this wave
  const int ntiles = (NQ + kQT - 1) / kQT;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc, tile1 = min(ntiles, tile0 + tpc);
  const uint32_t tr_off = (uint32_t)((((lane & 15) >> 2) + 4 * h) * kQPitch + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const uint32_t q_tr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&qimg[0][0] + tr_off;
  const uint32_t ep_tr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&ep[wid][0][0] + tr_off;
  static_assert(kQPitch == kEPitch, "one transposed-read lane offset serves both images");

  f32x16 acc[kPBW];
#pragma unroll
  for (int b = 0; b < kPBW; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  double es = 0.0;

  for (int tile = tile0; tile < tile1; ++tile) {
    const int q0 = tile * kQT;
    __syncthreads();                     // the previous tile's image / G tile are free (first trip: zero fill done)
    // ---- stage the Q tile: three bf16 planes + the logit biases ----------------------------------
    if (t < kQT) bqs[t] = (bias_q && q0 + t < NQ) ? bias_q[q0 + t] : 0.f;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int pc = t + NT * j, row = pc >> 2, k8 = (pc & 3) * 8;
      const int qr = min(q0 + row, NQ - 1);
      const float keep = q0 + row < NQ ? 1.f : 0.f;
      const float* src = Q + (size_t)qr * kKD + k8;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      const float v[8] = {a.x * keep, a.y * keep, a.z * keep, a.w * keep, b.x * keep, b.y * keep, b.z * keep, b.w * keep};
      u32x4 c3[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x0 = v[2 * e], x1 = v[2 * e + 1];
        const uint32_t p1 = pack_bf16(x0, x1);
        x0 -= bf16_lo(p1);
        x1 -= bf16_hi(p1);
        const uint32_t p2 = pack_bf16(x0, x1);
        x0 -= bf16_lo(p2);
        x1 -= bf16_hi(p2);
        c3[0][e] = p1; c3[1][e] = p2; c3[2][e] = pack_bf16(x0, x1);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(&qimg[pl][row * kQPitch + k8 * 2]) = c3[pl];
    }
    __syncthreads();
    const bool q_edge = q0 + kQT > NQ;
    float es_tile = 0.f;
#pragma unroll 1
    for (int sub = 0; sub < kQT / 32; ++sub) {
      if (q0 + 32 * sub >= NQ) break;                     // block-uniform
      // G (and the row sums of E) of this sub-tile over the wave's P blocks, in registers
      f32x16 g, gs;
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = gs[i] = 0.f;
#pragma unroll
      for (int b = 0; b < kPBW; ++b) {
        const int blk = blk0 + b;
        if (blk * 32 >= NP) continue;                     // wave-uniform
        const uint16_t* pzb = pz + (size_t)blk * 32 * kKD;
        const uint16_t* ztb = zt + (size_t)blk * 3 * 2 * 2 * kKD * 8;
        if (q_edge || blk * 32 + 32 > NP)                 // wave-uniform
          sigf_unit<true>(&qimg[0][0], q_tr, &ep[wid][0][0], ep_tr, bqs, pzb, np_pad, ztb, acc[b], g, gs, es_tile, sub, r,
                          h, q0, NQ, blk * 32 + r < NP);
        else
          sigf_unit<false>(&qimg[0][0], q_tr, &ep[wid][0][0], ep_tr, bqs, pzb, np_pad, ztb, acc[b], g, gs, es_tile, sub, r,
                           h, q0, NQ, true);
      }
      // register t of lane (k = r, h) is row q' = rho_h(t) of G: two whole 128-B rows per store instruction
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
        const int qp = (tt & 3) + 8 * (tt >> 2) + 4 * h;
        gslot[wid][qp * 32 + r] = g[tt];
        if (r == 0) gslot[wid][32 * 32 + qp] = gs[tt];
      }
      __syncthreads();
      // the four waves' tiles -> gV / gphi (rows in q' order: q' = 16 hh + i  <->  q = (i & 3) + 8 (i >> 2) + 4 hh)
#pragma unroll
      for (int j = 0; j < (32 * 32 + 32 + NT - 1) / NT; ++j) {
        const int e = t + NT * j;
        if (e < 32 * 32 + 32) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < kNW; ++w) v += gslot[w][e];
          const int qp = e < 32 * 32 ? (e >> 5) : (e - 32 * 32);
          const int i = qp & 15, hh = qp >> 4;
          const int q = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (q < NQ && v != 0.f) {
            const size_t row = out_rows ? out_rows[q] : q;
            if (e < 32 * 32) atomicAdd(gV + row * kKD + (e & 31), sign2 * v);
            else if (gphi) atomicAdd(gphi + row, sign2 * v);
          }
        }
      }
      __syncthreads();                   // the slots are free for the next sub-tile
    }
    es += (double)es_tile;
  }
  // ---- the P rows' sums -------------------------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < kPBW; ++b) {
    const int p = (blk0 + b) * 32 + r;
    if (p >= NP) continue;
    const float sc = sign * (p_scale ? p_scale[p] : 1.f);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float* dst = out + (size_t)p * kKD + 8 * g4 + 4 * h;
      const float4 v = make_float4(sc * acc[b][4 * g4 + 0], sc * acc[b][4 * g4 + 1], sc * acc[b][4 * g4 + 2],
                                   sc * acc[b][4 * g4 + 3]);
      if (atomic_out) {
        atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
      } else {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        *reinterpret_cast<float4*>(dst) = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
      }
    }
  }
  if (esum) {
    const double tsum = block_sum(es, red);
    if (t == 0) atomicAdd(esum, tsum);
  }
}

size_t sigfused3_scratch_bytes(int64_t rows) {
  const int64_t np_pad = (rows + 31) / 32 * 32;
  return (size_t)2 * 3 * np_pad * kKD * sizeof(uint16_t);
}
int sigfused3_rows_per_wg() { return kNW * kPBW * 32; }

// out[p] += sign * p_scale[p] * sum_q E Q_q (plain read-modify-write, float atomics with q_chunks > 1);
// gV[out_rows[q]] += sign2 * sum_p E P_p; gphi[out_rows[q]] += sign2 * sum_p E; esum += sum softplus.
bool launch_sigfused3(int KD, const SigFusedArgs& a, hipStream_t st) {
  if (KD != 32 || !a.scratch || a.NP < 1 || a.NQ < 1) return false;
  const int np_pad = (a.NP + 31) / 32 * 32;
  uint16_t* pz = reinterpret_cast<uint16_t*>(a.scratch);
  uint16_t* zt = pz + (size_t)3 * np_pad * kKD;
  const int64_t n = (int64_t)np_pad * kKD;
  hipLaunchKernelGGL(sigf_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.NP, np_pad, a.P, pz, zt);
  const int rpw = kNW * kPBW * 32;
  const int nbx = (a.NP + rpw - 1) / rpw;
  const int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  hipLaunchKernelGGL(sigfused3_kernel, dim3(nbx, chunks), dim3(kNW * 64), 0, st, a.NP, a.NQ, np_pad, pz, zt, a.Q,
                     a.bias_q, a.out, a.sign, a.p_scale, chunks > 1 ? 1 : 0, a.esum, a.gV, a.gphi, a.out_rows, a.sign2);
  return true;
}

}  // namespace spmf
