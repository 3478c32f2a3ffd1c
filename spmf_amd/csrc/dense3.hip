// dense3.hip -- the dense exp sums of the log_transform decoder (dense.hip: expdot) on the
// BF16 matrix cores with fp32 accuracy: every fp32 operand is split three ways,
//     a = a1 + a2 + a3,  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// (3 x 8 significant bits = fp32's 24; the two subtractions are exact in fp32), and a product
// is the six partial products with i + j <= 4,
//     a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Dropped: a2 b3, a3 b2, a3 b3 <= 3 * 2^-24
// |a b| (each factor pair carries 2^-8 * 2^-16 or smaller), the size of fp32's own product
// rounding (adding a2 b3 and a3 b2 back changed nothing measurable); error analysis and the
// measured parity: DESIGN.md section 4 (dense path), tools/b3_err.py.
// Six bf16 MFMAs of 32 cycles replace eight f32 MFMAs of 64 (32x32x2 covers k = 2, 32x32x16
// covers k = 16): 2.67 x less matrix-pipe time per product.
// NOT every operand is split three ways: in the SECOND product E = exp(X) is carried in TWO bf16
// planes (e1 = bf16(E), e2 = bf16(E - e1): 16 significant bits, |E - e1 - e2| <= 2^-18 E per term)
// against Q in three, five MFMAs: e1 (q1 + q2 + q3) + e2 (q1 + q2).  Every term of those sums is
// positive (E > 0 times one sign of z or eta v per entry), so the bound on a sum is 2^-18 = 3.8e-6 of
// the result in the worst case and the rounding is unbiased in practice; sum E for part 'x' comes from
// the unsplit fp32 E.  Pinned by tests/test_gpu_logtransform.py (oracle at 1e-5 / 1.5e-5 up to
// exponents of 60, and against the exact-f32 kernels entry by entry).
//
//   expdot3(P, Q):  out_p[k] = sum_q exp(min(<P_p, Q_q>, kYSat)) Q_q[k],  esum = sum_{p,q} exp(.)
// the same operator as dense.hip's expdot_kernel<KD, 0> (Poisson likelihood, exp decoder:
// mederrata_spmf/poisson.py:52-53,174-183), launched as (Z, W) and (W, Z): E is recomputed
// in the second launch instead of making a B*D*4-byte round trip through HBM -- at this
// matrix rate that round trip (120 GB per step at C4) would be the bound.
//
// Layout.  A workgroup is 8 waves; each wave owns 32 rows of P as the B operand of
//     X = Q_tile P_tile^T                 (A = rows of Q from LDS, 4 k-steps x 6 products;
//                                          a1 b1 and the five small products accumulate apart)
// so X has the P row on the lane and the Q row in the 16 accumulator registers; E = exp(X),
// split in registers into TWO planes, is the B operand of
//     out^T[k, p] += sum_q Q^T[k, q] E[q, p]   (A = columns of Q from LDS, 2 x 2 x 5 MFMAs)
// with the k order of a step fixed by the accumulator's row order (cdna guide, "an accumulator
// tile as the next MFMA's operand": element j of lane half h is row 16 s + 8 (j >> 2) + 4 h +
// (j & 3)).  A Q tile (128 rows) is split once while it is staged -- a thread owns 8 consecutive k
// of a row: two 16-byte loads, three 16-byte LDS stores -- and kept in LDS as ONE image per
// plane, [q][k] with 144-byte rows (36 dwords: the 16 lanes of a ds_read_b128 service group
// cover all 64 banks).  The first product reads its A fragments as rows (ds_read_b128), the
// second reads the same image column-wise with the transposing LDS load ds_read_b64_tr_b16
// (two 4-row blocks per fragment: rows 16 s2 + 4 h .. + 3 and 16 s2 + 8 + 4 h .. + 3).  Two
// buffers of 3 x 18 KB: one barrier per 176 MFMAs and wave.
// Measured on C4 (two launches): 43.1 ms serial body, six products everywhere, E in three planes,
// two LDS images of 64-row tiles -> 38.3 pipelined body + two-plane E -> 33.8 staging loads at
// immediate offsets + paired conversions + launches sized to whole rounds of workgroups -> 31.3
// one image + transposing loads + 128-row tiles.  88 MFMAs per 64 x 32 cells = 1.35 PFLOP/s of
// bf16 issue, at the rate the guide measures for bf16 GEMMs on random data (1.25 PFLOP/s: the
// chip lowers its clock under matrix load); without the exp/split vector work 31.4 -> 28 ms,
// without restaging ~ the same: what is left is the clock.
#include "common.h"
#include "kernels.h"

namespace spmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kPitch3 = 144;      // bytes per LDS row (128 of data + 16)

// two floats -> one register of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }

struct Split3 {
  __bf16 a, b, c;
};
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  s.a = (__bf16)x;
  const float r1 = x - (float)s.a;
  s.b = (__bf16)r1;
  s.c = (__bf16)(r1 - (float)s.b);
  return s;
}

}  // namespace

// One Q tile (four 32-row sub-tiles) for one wave, branch free (EDGE is a template parameter) so
// that the whole body is ONE scheduling region, written in the order the in-order issue should
// see it (A = first product of a sub-tile, E = exp + split, C = second product):
//   A0 | A1 + E0 | C0 | A2 + E1 | C1 | A3 + E2 | C2 + E3 | C3
// with sched_group_barrier asking for one MFMA per few vector instructions where both are present.
// SPMF_EXP3_WHATIF (timing experiments only, WRONG numbers): bit 0 = the first product reads one LDS plane
// instead of three, bit 1 = the second product does: what the matrix pipe does when two thirds of those reads go;
// bit 2 = no v_exp in the epilogue
#ifndef SPMF_EXP3_WHATIF
#define SPMF_EXP3_WHATIF 0
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int kQT3 = 128;

template <int KD, bool EDGE>
__device__ __forceinline__ void expdot3_tile(const unsigned char* __restrict__ img, uint32_t lds_tr_base,
                                              const bf16x8 (&pb)[KD / 16][3], f32x16 (&acc)[KD / 32],
                                              float& es_tile, float& xmax, int r, int h, int q0, int NQ,
                                              bool p_in) {
  constexpr int KS = KD / 16, MT = KD / 32, NSUB = kQT3 / 32;
  constexpr int IMG = kQT3 * kPitch3;
  f32x16 xh[NSUB], xl[NSUB];
  u32x4 eb[NSUB][2][2];          // [sub][s2][plane]
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto productA = [&](int sub) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[pl] = *reinterpret_cast<const bf16x8*>(img + (SPMF_EXP3_WHATIF & 1 ? 0 : pl) * IMG + (32 * sub + r) * kPitch3 + (16 * s + 8 * h) * 2);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pb[s][0], s == 0 ? zero16 : xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][1], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][2], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][0], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][1], xl[sub], 0, 0, 0);
      xh[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][0], s == 0 ? zero16 : xh[sub], 0, 0, 0);
    }
  };
  auto expsplit = [&](int sub) {
    f32x2 part2 = {0.f, 0.f};
    constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f32x2 xa = {xh[sub][i], xh[sub][i + 1]}, xb = {xl[sub][i], xl[sub][i + 1]};
      f32x2 x = xa + xb;
      if (EDGE) {
        const int qa = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
        x[0] = (p_in && qa < NQ) ? x[0] : -INFINITY;          // exp -> 0, and never the maximum
        x[1] = (p_in && qa + 1 < NQ) ? x[1] : -INFINITY;
      }
      xmax = __builtin_fmaxf(xmax, __builtin_fmaxf(x[0], x[1]));
      const f32x2 y = x * kLog2e;
      f32x2 e;
      if (SPMF_EXP3_WHATIF & 4) {                       // (timing experiment: no exponential)
        e = y;
      } else {
        e[0] = __builtin_amdgcn_exp2f(fminf(y[0], kYSat * kLog2e));
        e[1] = __builtin_amdgcn_exp2f(fminf(y[1], kYSat * kLog2e));
      }
      part2 += e;
      const uint32_t p1 = pack_bf16(e[0], e[1]);
      const f32x2 hi = {bf16_lo(p1), bf16_hi(p1)};
      const f32x2 rr = e - hi;
      const uint32_t p2 = pack_bf16(rr[0], rr[1]);
      eb[sub][i >> 3][0][(i & 7) >> 1] = p1;
      eb[sub][i >> 3][1][(i & 7) >> 1] = p2;
    }
    es_tile += part2[0] + part2[1];
  };
  auto productC = [&](int sub) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          // lds_tr_base: this lane's address inside a 4-row x 16-column block (rows = q, columns = k)
          const uint32_t ad = lds_tr_base + (SPMF_EXP3_WHATIF & 2 ? 0 : pl) * IMG + (32 * sub + 16 * s2) * kPitch3 + 32 * m * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(uintptr_t)ad);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(uintptr_t)(ad + 8 * kPitch3));
          const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          a[pl] = __builtin_bit_cast(bf16x8, both);
        }
        const bf16x8 e1v = __builtin_bit_cast(bf16x8, eb[sub][s2][0]);
        const bf16x8 e2v = __builtin_bit_cast(bf16x8, eb[sub][s2][1]);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e1v, acc[m], 0, 0, 0);
      }
  };
  // issue order: A0 | A1 + E0 | C0 | A2 + E1 | C1 | A3 + E2 | C2 + E3 | C3
  // SPMF_EXP3_DSLEAD (experiment, 0 = off): LDS reads as a scheduling group of their own -- a lead of that
  // many reads at the head of a phase, then SPMF_EXP3_DSPER reads behind every MFMA -- instead of wherever the
  // register-pressure heuristic sinks them (next to their use)
#ifndef SPMF_EXP3_DSLEAD
#define SPMF_EXP3_DSLEAD 0
#endif
#ifndef SPMF_EXP3_DSPER
#define SPMF_EXP3_DSPER 1
#endif
#define SPMF_EXP3_PHASE(N_, V_)                                                                   \
  do {                                                                                            \
    if (SPMF_EXP3_DSLEAD > 0) __builtin_amdgcn_sched_group_barrier(0x100, SPMF_EXP3_DSLEAD, 0);    \
    _Pragma("unroll") for (int i = 0; i < (N_); ++i) {                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      if (SPMF_EXP3_DSLEAD > 0) __builtin_amdgcn_sched_group_barrier(0x100, SPMF_EXP3_DSPER, 0);   \
      __builtin_amdgcn_sched_group_barrier(0x002, (V_), 0);                                       \
    }                                                                                             \
  } while (0)
  productA(0);
  productA(1);
  expsplit(0);
  SPMF_EXP3_PHASE(24, 5);
  productC(0);
  productA(2);
  expsplit(1);
  SPMF_EXP3_PHASE(44, 3);
  productC(1);
  productA(3);
  expsplit(2);
  SPMF_EXP3_PHASE(44, 3);
  productC(2);
  expsplit(3);
  SPMF_EXP3_PHASE(20, 6);
  productC(3);
#undef SPMF_EXP3_PHASE
}

#ifndef SPMF_EXP3_NW
#define SPMF_EXP3_NW 8       // waves per workgroup (32 P rows each)
#endif
#ifndef SPMF_EXP3_NBUF
#define SPMF_EXP3_NBUF 2     // LDS buffers of the staged Q tile (1: the next tile waits in registers for a barrier)
#endif
constexpr int kExpNW = SPMF_EXP3_NW, kExpNBUF = SPMF_EXP3_NBUF;
template <int KD>
__global__ __launch_bounds__(kExpNW * 64, 2) void expdot3_kernel(int NP, int NQ, const float* __restrict__ P,
                                                          const float* __restrict__ Q, float* __restrict__ out,
                                                          float sign, double* __restrict__ esum, int atomic_out,
                                                          int accumulate, const float* __restrict__ p_scale) {
  static_assert(KD == 64, "expdot3b: K padded to 64");
  constexpr int KS = KD / 16, MT = KD / 32, NW = kExpNW, NT = NW * 64;
  constexpr int IMG = kQT3 * kPitch3;
  constexpr int NPC = kQT3 * (KD / 8) / NT;    // (row, 8 k) pieces per loader thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[kExpNBUF][3][IMG];
  __shared__ double red[16];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int p = (blockIdx.x * NW + wid) * 32 + r;
  const int ntiles = (NQ + kQT3 - 1) / kQT3;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc, tile1 = min(ntiles, tile0 + tpc);
  // transposed-read lane address inside a block: lane 4 q' + pp of a 16-lane group supplies
  // row q' (+ 4 h: the lane half's rows), columns 16 g16 + 4 pp .. + 3
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0][0][0];
  const uint32_t tr_lane = (uint32_t)((((lane & 15) >> 2) + 4 * h) * kPitch3 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);

  bf16x8 pb[KS][3];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float v[8];
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      const float4 f = p < NP ? *reinterpret_cast<const float4*>(P + (size_t)p * KD + 16 * s + 8 * h + 4 * q4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      v[4 * q4 + 0] = f.x; v[4 * q4 + 1] = f.y; v[4 * q4 + 2] = f.z; v[4 * q4 + 3] = f.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const Split3 sp = split3(v[j]);
      pb[s][0][j] = sp.a; pb[s][1][j] = sp.b; pb[s][2][j] = sp.c;
    }
  }
  constexpr int FOLD = 4;                       // 4 tiles of 128 = runs of 512 terms, as before
  f32x16 acc[MT], tot[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = tot[m][i] = 0.f;
  double es = 0.0;
  float xmax = 0.f;

  // ---- stage loader: piece pc = t + NT j: row pc / 8 of the tile, k = 8 (pc % 8) .. + 7 -------
  // (single-buffer form: the pieces of a thread go through the staging registers two at a time)
  constexpr int NST = kExpNBUF == 2 ? NPC : (NPC < 2 ? NPC : 2);
  float4 st0[NST], st1[NST];
  auto gload = [&](int tile, int j0 = 0) {
    const int q0 = tile * kQT3;
#pragma unroll
    for (int jj = 0; jj < NST; ++jj) {
      const int j = j0 + jj;
      const int pc = t + NT * j, row = pc >> 3, k8 = (pc & 7) * 8;
      // rows past NQ read the last row (always in bounds) and are zeroed: no branch, no
      // exec masking around the loads
      const int qr = min(q0 + row, NQ - 1);
      const float* src = Q + (size_t)qr * KD + k8;
      const float keep = q0 + row < NQ ? 1.f : 0.f;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      st0[jj] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
      st1[jj] = make_float4(b.x * keep, b.y * keep, b.z * keep, b.w * keep);
    }
  };
  auto swrite = [&](int buf, int j0 = 0) {
#pragma unroll
    for (int jj = 0; jj < NST; ++jj) {
      const int j = j0 + jj;
      const int pc = t + NT * j, row = pc >> 3, k8 = (pc & 7) * 8;
      const float v[8] = {st0[jj].x, st0[jj].y, st0[jj].z, st0[jj].w, st1[jj].x, st1[jj].y, st1[jj].z, st1[jj].w};
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
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(&lds[buf][pl][row * kPitch3 + k8 * 2]) = c3[pl];
    }
  };

  if (tile0 < tile1) {
#pragma unroll
    for (int j0 = 0; j0 < NPC; j0 += NST) {
      gload(tile0, j0);
      swrite(0, j0);
    }
  }
  __syncthreads();
  const bool p_edge = (int)(blockIdx.x * NW * 32 + NW * 32) > NP;      // block-uniform
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = kExpNBUF == 2 ? ((tile - tile0) & 1) : 0;
    const bool more = tile + 1 < tile1;                    // block-uniform
    if (kExpNBUF == 2 && more) gload(tile + 1);            // lands under this tile's MFMAs
    const int q0 = tile * kQT3;
    float es_tile = 0.f;
    const uint32_t trb = lds0 + buf * 3 * IMG + tr_lane;
    if (p_edge || q0 + kQT3 > NQ)
      expdot3_tile<KD, true>(&lds[buf][0][0], trb, pb, acc, es_tile, xmax, r, h, q0, NQ, p < NP);
    else
      expdot3_tile<KD, false>(&lds[buf][0][0], trb, pb, acc, es_tile, xmax, r, h, q0, NQ, true);
    es += (double)es_tile;
    if (((tile - tile0) % FOLD) == FOLD - 1) {             // block-uniform: close the run
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          tot[m][i] += acc[m][i];
          acc[m][i] = 0.f;
        }
    }
    if (kExpNBUF == 2) {
      if (more) swrite(buf ^ 1);
      __syncthreads();
    } else {
      __syncthreads();                                     // every wave has read this tile
      if (more) {                                          // (the CU's other workgroup computes meanwhile)
#pragma unroll
        for (int j0 = 0; j0 < NPC; j0 += NST) {
          gload(tile + 1, j0);
          swrite(0, j0);
        }
      }
      __syncthreads();
    }
  }
  if (p < NP) {
    const float sc = sign * (p_scale ? p_scale[p] : 1.f);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float* dst = out + (size_t)p * KD + m * 32 + 8 * g4 + 4 * h;
        const float4 v = make_float4(sc * (acc[m][4 * g4 + 0] + tot[m][4 * g4 + 0]),
                                     sc * (acc[m][4 * g4 + 1] + tot[m][4 * g4 + 1]),
                                     sc * (acc[m][4 * g4 + 2] + tot[m][4 * g4 + 2]),
                                     sc * (acc[m][4 * g4 + 3] + tot[m][4 * g4 + 3]));
        if (atomic_out) {
          atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        } else if (accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(dst);      // this lane is the only writer of dst
          *reinterpret_cast<float4*>(dst) = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        } else {
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
  }
  if (esum) {
    const double tsum = block_sum(es, red);
    if (t == 0) atomicAdd(esum, tsum);
    const double ts = block_sum(xmax > kYSat ? 1.0 : 0.0, red);
    if (t == 0 && ts != 0.0) atomicAdd(esum + 1, 1.0);
  }
}

// ---------------------------------------------------------------------------------------------
// sigdot3: the Bernoulli / mixed dense sums (dense.hip ACT 1; mederrata_spmf/bernoulli.py:127-155:
// ll = x l - softplus(l), l = <z_b, V'_d> + phi_d) on the same bf16x3 operands:
//     out_p[k] = sum_q sigmoid(l_pq) Q_q[k],   esum = sum softplus(l_pq),   out2[p] = sum_q sigmoid(l_pq)
// launched as (Z, W) with the column bias on the Q rows and as (W, Z) with it on the P rows; the
// sigmoid is recomputed in the second launch (keeping it costs a B*D*4-byte write and read: 8 GB per
// step on C5, more than the second launch's whole time).  KD = 32 or 64.
// Differences from expdot3: (i) the bias rides in the accumulator's initial value, and ONE fp32
// accumulator takes all six partial products of a step (the logits are O(10): no large-exponent
// regime whose absolute error a second accumulator would have to protect -- d sigmoid = s(1-s) dX
// <= dX/4, and the 2*KD/16*6 MFMA adds of the chain are the fmaf chain of the exact-f32 kernel, just
// shorter); (ii) the epilogue is the one of dense.hip ACT 1 -- sigmoid from one v_exp and one v_rcp,
// the softplus sum as max(l, 0) plus ONE log2 of the product of a sub-tile's sixteen (1 + e^-|l|)
// factors -- followed by the two-plane split of E = sigmoid in [0, 1] (|E - e1 - e2| <= 2^-18 E term by
// term, every term of the sums positive: <= 3.8e-6 of the result in the worst case, unbiased rounding
// in practice); (iii) the kernel is VALU-bound (about 15 vector issue slots per cell against 22 MFMAs
// per 1024 cells at KD = 32), so the body is left to the compiler's scheduler: other waves' MFMAs fill
// the matrix pipe under a wave's epilogue.
#ifndef SPMF_SIG3_VALU
#define SPMF_SIG3_VALU 230
#endif
constexpr int kSigValu = SPMF_SIG3_VALU;
// ESUM: this launch owns the softplus sum (part 'x'); CSUM: it owns the sigmoid sums per P row (out2);
// EPL: bf16 planes of E = sigmoid in the second product.  Two planes leave |dE| <= 2^-18 E per term: fine
// where every term of the sum has one sign (Q = z rows, z >= 0); where the Q rows have mixed signs (Q =
// V' under the Normal priors of bernoulli.py:187-216) a sum can cancel to a small fraction of its terms
// and the entry-wise 1e-5 needs the third plane (2^-26).
// ACT 1: the sigmoid / softplus operator described above.  ACT 0: the exp operator of expdot3 (Poisson,
// log_transform decoder: E = exp(min(X, kYSat)), esum = sum E, no bias) at K padded to 32, where expdot3
// itself (K padded to 64) does not apply: a1 b1 and the five small products in separate accumulators as there
// (exp amplifies the absolute error of X), E in two planes (every term of both sums is positive).
template <int KD, int ACT, bool EDGE, bool BQ, bool ESUM, bool CSUM, int EPL>
__device__ __forceinline__ void sigdot3_tile(const unsigned char* __restrict__ img, uint32_t lds_tr_base,
                                              const float* __restrict__ bq, float bp,
                                              const bf16x8 (&pb)[KD / 16][3], f32x16 (&acc)[KD / 32],
                                              float& es_tile, float& colsum, float& xmax, int r, int h, int q0,
                                              int NQ, bool p_in) {
  constexpr int KS = KD / 16, MT = KD / 32, NSUB = kQT3 / 32;
  constexpr int PITCH = KD * 2 + 16;
  constexpr int IMG = kQT3 * PITCH;
  constexpr bool XSPLIT = ACT != 1;      // the exp forms: a1 b1 and the five small products apart
  f32x16 x[NSUB], xlo[XSPLIT ? NSUB : 1];
  u32x4 eb[NSUB][2][EPL];          // [sub][s2][plane]
  auto productA = [&](int sub) {
    f32x16 xi;
    if (XSPLIT) {
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 xl = zero16;
      xi = zero16;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[pl] = *reinterpret_cast<const bf16x8*>(img + pl * IMG + (32 * sub + r) * PITCH + (16 * s + 8 * h) * 2);
        xl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pb[s][0], xl, 0, 0, 0);
        xl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][1], xl, 0, 0, 0);
        xl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][2], xl, 0, 0, 0);
        xl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][0], xl, 0, 0, 0);
        xl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][1], xl, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][0], xi, 0, 0, 0);
      }
      x[sub] = xi;
      xlo[XSPLIT ? sub : 0] = xl;
      return;
    }
    if (BQ) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bq + 32 * sub + 8 * g + 4 * h);
        xi[4 * g + 0] = b4.x; xi[4 * g + 1] = b4.y; xi[4 * g + 2] = b4.z; xi[4 * g + 3] = b4.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) xi[i] = bp;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[pl] = *reinterpret_cast<const bf16x8*>(img + pl * IMG + (32 * sub + r) * PITCH + (16 * s + 8 * h) * 2);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pb[s][0], xi, 0, 0, 0);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][1], xi, 0, 0, 0);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][2], xi, 0, 0, 0);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][0], xi, 0, 0, 0);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][1], xi, 0, 0, 0);
      xi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][0], xi, 0, 0, 0);
    }
    x[sub] = xi;
  };
  auto sigsplit = [&](int sub) {
    constexpr float kLog2e = 1.4426950408889634f;
    if (ACT == 0) {
      f32x2 part2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float x0 = x[sub][i] + xlo[XSPLIT ? sub : 0][i], x1 = x[sub][i + 1] + xlo[XSPLIT ? sub : 0][i + 1];
        if (EDGE) {
          const int qa = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
          x0 = (p_in && qa < NQ) ? x0 : -INFINITY;          // exp -> 0, and never the maximum
          x1 = (p_in && qa + 1 < NQ) ? x1 : -INFINITY;
        }
        xmax = __builtin_fmaxf(xmax, __builtin_fmaxf(x0, x1));
        const float e0 = __builtin_amdgcn_exp2f(fminf(x0 * kLog2e, kYSat * kLog2e));
        const float e1 = __builtin_amdgcn_exp2f(fminf(x1 * kLog2e, kYSat * kLog2e));
        part2[0] += e0;
        part2[1] += e1;
        const uint32_t p1 = pack_bf16(e0, e1);
        eb[sub][i >> 3][0][(i & 7) >> 1] = p1;
        eb[sub][i >> 3][1][(i & 7) >> 1] = pack_bf16(e0 - bf16_lo(p1), e1 - bf16_hi(p1));
      }
      es_tile += part2[0] + part2[1];
      return;
    }
    float pm0 = 0.f, pm1 = 0.f, dp0 = 1.f, dp1 = 1.f, cs0 = 0.f, cs1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      float l0 = x[sub][i], l1 = x[sub][i + 1];
      float ex0 = 1.f, ex1 = 1.f;
      if (ACT == 2) {
        // Bernoulli + log_transform (bernoulli.py:60-61): l = exp(X) - 1 + bias, saturating like the Poisson
        // form; the second product takes E = sigmoid(l) exp(X) = d softplus(l) / dX
        ex0 = __builtin_amdgcn_exp2f(fminf((l0 + xlo[XSPLIT ? sub : 0][i]) * kLog2e, kYSat * kLog2e));
        ex1 = __builtin_amdgcn_exp2f(fminf((l1 + xlo[XSPLIT ? sub : 0][i + 1]) * kLog2e, kYSat * kLog2e));
        const float b0 = BQ ? bq[32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h] : bp;
        const float b1 = BQ ? bq[32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h + 1] : bp;
        l0 = (ex0 - 1.f) + b0;
        l1 = (ex1 - 1.f) + b1;
      }
      if (EDGE) {
        const int qa = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
        l0 = (p_in && qa < NQ) ? l0 : -INFINITY;          // sigmoid -> 0, softplus -> 0
        l1 = (p_in && qa + 1 < NQ) ? l1 : -INFINITY;
      }
      float s0, s1;
      if (ESUM) {
        // softplus(l) = max(l, 0) + log(1 + e^-|l|); sigmoid from the same e^-|l|
        const float en0 = __builtin_amdgcn_exp2f(-fabsf(l0) * kLog2e);      // in (0, 1]
        const float en1 = __builtin_amdgcn_exp2f(-fabsf(l1) * kLog2e);
        const float d0 = 1.f + en0, d1 = 1.f + en1;
        const float i0 = __builtin_amdgcn_rcpf(d0), i1 = __builtin_amdgcn_rcpf(d1);
        const bool g0 = l0 >= 0.f, g1 = l1 >= 0.f;
        s0 = (g0 ? 1.f : en0) * i0;
        s1 = (g1 ? 1.f : en1) * i1;
        pm0 += g0 ? l0 : 0.f;            // (a select, not fmaxf: no canonicalising v_max in front of it)
        pm1 += g1 ? l1 : 0.f;
        dp0 *= d0;
        dp1 *= d1;
      } else {
        // sigmoid(l) = 1 / (1 + e^-l): e^-l overflows to +inf for l < -88 and the reciprocal gives the 0
        // that sigmoid rounds to there; relative accuracy 2 ulp everywhere else
        s0 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-l0 * kLog2e));
        s1 = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-l1 * kLog2e));
      }
      if (CSUM) {
        cs0 += s0;
        cs1 += s1;
      }
      if (ACT == 2) {
        s0 *= ex0;
        s1 *= ex1;
      }
      const uint32_t p1 = pack_bf16(s0, s1);
      const float r0 = s0 - bf16_lo(p1), r1 = s1 - bf16_hi(p1);
      const uint32_t p2 = pack_bf16(r0, r1);
      eb[sub][i >> 3][0][(i & 7) >> 1] = p1;
      eb[sub][i >> 3][1][(i & 7) >> 1] = p2;
      if (EPL == 3) eb[sub][i >> 3][EPL - 1][(i & 7) >> 1] = pack_bf16(r0 - bf16_lo(p2), r1 - bf16_hi(p2));
    }
    // sum_i log(1 + e^-|l_i|) = ln2 * log2 of the product (eight factors in (1, 2] per partial product)
    if (ESUM)
      es_tile += (pm0 + pm1) + 0.69314718056f * (__builtin_amdgcn_logf(dp0) + __builtin_amdgcn_logf(dp1));
    if (CSUM) colsum += cs0 + cs1;
  };
  auto productC = [&](int sub) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint32_t ad = lds_tr_base + pl * IMG + (32 * sub + 16 * s2) * PITCH + 32 * m * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(uintptr_t)ad);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(uintptr_t)(ad + 8 * PITCH));
          const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          a[pl] = __builtin_bit_cast(bf16x8, both);
        }
        const bf16x8 e1v = __builtin_bit_cast(bf16x8, eb[sub][s2][0]);
        const bf16x8 e2v = __builtin_bit_cast(bf16x8, eb[sub][s2][1]);
        if (EPL == 3)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], __builtin_bit_cast(bf16x8, eb[sub][s2][EPL - 1]),
                                                            acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e1v, acc[m], 0, 0, 0);
      }
  };
  // issue order (one scheduling region, as expdot3_tile): A0 | A1 + E0 | C0 + A2 + E1 | C1 + A3 + E2 |
  // C2 + E3 | C3 -- the vector work of a sub-tile's epilogue is spread under the MFMAs of the next
  // sub-tile's first product and the previous one's second (kSigValu ~ vector instructions of one
  // epilogue; a group that asks for more than there are is simply shorter)
  constexpr int NA = 6 * KS, NC = (2 * EPL + 1) * 2 * MT;
  constexpr int NV = ESUM ? kSigValu : (kSigValu * 2) / 3;
  productA(0);
  productA(1);
  sigsplit(0);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV + NA - 1) / NA, 0);
  }
  productC(0);
  productA(2);
  sigsplit(1);
#pragma unroll
  for (int i = 0; i < NA + NC; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV + NA + NC - 1) / (NA + NC), 0);
  }
  productC(1);
  productA(3);
  sigsplit(2);
#pragma unroll
  for (int i = 0; i < NA + NC; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV + NA + NC - 1) / (NA + NC), 0);
  }
  productC(2);
  sigsplit(3);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (NV + NC - 1) / NC, 0);
  }
  productC(3);
}

#ifndef SPMF_SIG3_WPS
#define SPMF_SIG3_WPS 2      // waves per SIMD the kernel is compiled for (register budget 512 / WPS)
#endif
#ifndef SPMF_SIG3_NW
#define SPMF_SIG3_NW 4       // waves per workgroup (32 P rows each) at KD = 32: two 256-thread workgroups per CU, so
                             // a workgroup at its per-tile barrier leaves the CU to the other (C5 dense: 1.85 -> 1.52 ms)
#endif
// KD = 64: the staged Q tile is 110 KB (two buffers x three planes), one workgroup per CU: eight waves, like expdot3
__host__ __device__ constexpr int sig_nw(int KD) { return KD == 64 ? 8 : SPMF_SIG3_NW; }
template <int KD, bool BQ, bool ESUM, bool CSUM, int EPL, int ACT = 1>
__global__ __launch_bounds__(sig_nw(KD) * 64, SPMF_SIG3_WPS) void sigdot3_kernel(
    int NP, int NQ, const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ out, float sign,
    double* __restrict__ esum, int atomic_out, const float* __restrict__ bias, float* __restrict__ out2,
    const int32_t* __restrict__ out_rows, int accumulate, const float* __restrict__ p_scale) {
  static_assert(KD == 32 || KD == 64, "sigdot3: K padded to 32 or 64");
  static_assert(ACT != 0 || (EPL == 2 && !BQ && !CSUM), "the exp form: two planes of E, no bias, no row sums");
  constexpr int KS = KD / 16, MT = KD / 32, NW = sig_nw(KD), NT = NW * 64;
  constexpr int PITCH = KD * 2 + 16;            // bytes per LDS row; 144 (KD 64) / 80 (KD 32): the 16 rows of a
                                                // ds_read_b128 service group cover all 64 banks
  constexpr int IMG = kQT3 * PITCH;
  constexpr int PCS = kQT3 * (KD / 8);          // (row, 8 k) pieces of a tile
  constexpr int NPC = (PCS + NT - 1) / NT;      // pieces per loader thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][3][IMG];
  __shared__ __attribute__((aligned(16))) float bqs[2][kQT3];
  __shared__ double red[16];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int p = (blockIdx.x * NW + wid) * 32 + r;
  const int ntiles = (NQ + kQT3 - 1) / kQT3;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc, tile1 = min(ntiles, tile0 + tpc);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0][0][0];
  const uint32_t tr_lane = (uint32_t)((((lane & 15) >> 2) + 4 * h) * PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);

  bf16x8 pb[KS][3];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float v[8];
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      const float4 f = p < NP ? *reinterpret_cast<const float4*>(P + (size_t)p * KD + 16 * s + 8 * h + 4 * q4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      v[4 * q4 + 0] = f.x; v[4 * q4 + 1] = f.y; v[4 * q4 + 2] = f.z; v[4 * q4 + 3] = f.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const Split3 sp = split3(v[j]);
      pb[s][0][j] = sp.a; pb[s][1][j] = sp.b; pb[s][2][j] = sp.c;
    }
  }
  const float bp = (!BQ && bias && p < NP) ? bias[p] : 0.f;
  // two-level accumulation (runs of 4 tiles x 128 = 512 terms per fp32 chain) at KD = 64 only: at KD = 32
  // the second accumulator set does not fit the 256-register budget of two waves per SIMD, the terms are
  // sigmoids in [0, 1] and a chain is one Q chunk long (a few thousand terms), like dense.hip's KD = 32 form
  constexpr int FOLD = 4;
  constexpr bool TWO = KD == 64;
  f32x16 acc[MT], tot[TWO ? MT : 1];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[m][i] = 0.f;
      if (TWO) tot[m][i] = 0.f;
    }
  double es = 0.0;
  float colsum = 0.f, coltot = 0.f;
  float xmax = 0.f;                 // ACT 0: largest exponent this lane saw (saturation at kYSat)

  float4 st0[NPC], st1[NPC];
  float bst = 0.f;
  auto gload = [&](int tile) {
    const int q0 = tile * kQT3;
    if (BQ && t < kQT3) bst = (bias && q0 + t < NQ) ? bias[q0 + t] : 0.f;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int pc = t + NT * j, row = (pc / (KD / 8)) % kQT3, k8 = (pc % (KD / 8)) * 8;
      const int qr = min(q0 + row, NQ - 1);           // rows past NQ: read the last row, zeroed below
      const float* src = Q + (size_t)qr * KD + k8;
      const float keep = q0 + row < NQ ? 1.f : 0.f;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      st0[j] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
      st1[j] = make_float4(b.x * keep, b.y * keep, b.z * keep, b.w * keep);
    }
  };
  auto swrite = [&](int buf) {
    if (BQ && t < kQT3) bqs[buf][t] = bst;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int pc = t + NT * j, row = (pc / (KD / 8)) % kQT3, k8 = (pc % (KD / 8)) * 8;
      if (PCS % NT != 0 && pc >= PCS) continue;
      const float v[8] = {st0[j].x, st0[j].y, st0[j].z, st0[j].w, st1[j].x, st1[j].y, st1[j].z, st1[j].w};
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
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(&lds[buf][pl][row * PITCH + k8 * 2]) = c3[pl];
    }
  };

  if (tile0 < tile1) {
    gload(tile0);
    swrite(0);
  }
  __syncthreads();
  const bool p_edge = (int)(blockIdx.x * NW * 32 + NW * 32) > NP;      // block-uniform
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = (tile - tile0) & 1;
    const bool more = tile + 1 < tile1;                    // block-uniform
    if (more) gload(tile + 1);                             // lands under this tile's MFMAs
    const int q0 = tile * kQT3;
    float es_tile = 0.f;
    const uint32_t trb = lds0 + buf * 3 * IMG + tr_lane;
    if (p_edge || q0 + kQT3 > NQ)
      sigdot3_tile<KD, ACT, true, BQ, ESUM, CSUM, EPL>(&lds[buf][0][0], trb, bqs[buf], bp, pb, acc, es_tile, colsum,
                                                       xmax, r, h, q0, NQ, p < NP);
    else
      sigdot3_tile<KD, ACT, false, BQ, ESUM, CSUM, EPL>(&lds[buf][0][0], trb, bqs[buf], bp, pb, acc, es_tile, colsum,
                                                        xmax, r, h, q0, NQ, true);
    if (ESUM || ACT == 0) es += (double)es_tile;
    if (((tile - tile0) % FOLD) == FOLD - 1) {             // block-uniform: close the run
      coltot += colsum;
      colsum = 0.f;
      if (TWO) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            tot[m][i] += acc[m][i];
            acc[m][i] = 0.f;
          }
      }
    }
    if (more) swrite(buf ^ 1);
    __syncthreads();
  }
  const int prow = (out_rows && p < NP) ? out_rows[p] : p;
  if (p < NP) {
    const float sc = sign * (p_scale ? p_scale[p] : 1.f);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float* dst = out + (size_t)prow * KD + m * 32 + 8 * g4 + 4 * h;
        float4 v = make_float4(acc[m][4 * g4 + 0], acc[m][4 * g4 + 1], acc[m][4 * g4 + 2], acc[m][4 * g4 + 3]);
        if (TWO) {
          v.x += tot[m][4 * g4 + 0]; v.y += tot[m][4 * g4 + 1]; v.z += tot[m][4 * g4 + 2]; v.w += tot[m][4 * g4 + 3];
        }
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        if (atomic_out) {
          atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        } else if (accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(dst);      // this lane is the only writer of dst
          *reinterpret_cast<float4*>(dst) = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        } else {
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
  }
  if (CSUM && out2) {
    float cs = colsum + coltot;
    cs += __shfl_xor(cs, 32);                         // the two lane halves hold disjoint q rows
    if (h == 0 && p < NP && cs != 0.f) atomicAdd(&out2[prow], sign * cs);
  }
  if ((ESUM || ACT == 0) && esum) {
    const double tsum = block_sum(es, red);
    if (t == 0) atomicAdd(esum, tsum);
    if (ACT == 0) {
      // esum[1] = dacc[4]: workgroups in which an exponent exceeded kYSat and was saturated there
      const double ts = block_sum(xmax > kYSat ? 1.0 : 0.0, red);
      if (t == 0 && ts != 0.0) atomicAdd(esum + 1, 1.0);
    }
  }
}

// launch geometry for the caller's chunk choice: P rows per workgroup, resident workgroups per CU
// (two waves per SIMD by registers: 8 waves per CU)
int expdot3_rows_per_wg() { return kExpNW * 32; }
int expdot3_wgs_per_cu() { return (8 / kExpNW) < (kExpNBUF == 2 ? 1 : 2) ? (8 / kExpNW) : (kExpNBUF == 2 ? 1 : 2); }
int sigdot3_rows_per_wg(int KD) { return sig_nw(KD) * 32; }
int sigdot3_wgs_per_cu(int KD) { return 8 / sig_nw(KD); }

// (Z, W)-type launch: bias on the Q rows (bias_q); (W, Z)-type: bias on the P rows (bias_p).
bool launch_sigdot3(int KD, const ExpdotArgs& a, hipStream_t st) {
  if (a.act == 0) {
    // the exp operator at K padded to 32 (launch_expdot3 covers 64): no biases, no row sums, no E store
    if (KD != 32 || a.est || a.bias_p || a.bias_q || a.out2 || a.out_rows) return false;
    const int chunks0 = a.q_chunks < 1 ? 1 : a.q_chunks;
    if (chunks0 > 1 && !a.atomic_out) return false;
    const int nbx0 = (a.NP + sig_nw(32) * 32 - 1) / (sig_nw(32) * 32);
    hipLaunchKernelGGL((sigdot3_kernel<32, false, false, false, 2, 0>), dim3(nbx0, chunks0), dim3(sig_nw(32) * 64), 0,
                       st, a.NP, a.NQ, a.P, a.Q, a.out, a.sign, a.esum, a.atomic_out, nullptr, nullptr, nullptr,
                       a.accumulate, a.p_scale);
    return true;
  }
  if ((KD != 32 && KD != 64) || (a.act != 1 && a.act != 2) || a.est || (a.bias_p && a.bias_q)) return false;
  const int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  if (chunks > 1 && !a.atomic_out) return false;
  if (a.p_scale && a.out_rows) return false;      // (p_scale is indexed by the P row, not the output row)
  const int nw = sig_nw(KD);
  const int nbx = (a.NP + nw * 32 - 1) / (nw * 32);
  const bool bq = a.bias_q != nullptr;
  const float* bias = bq ? a.bias_q : a.bias_p;
  const bool es = a.esum != nullptr, cs = a.out2 != nullptr;
  const int epl = a.e_planes == 3 ? 3 : 2;
#define SPMF_SIG3(KD_, BQ_, ES_, CS_, EPL_)                                                                  \
  hipLaunchKernelGGL((sigdot3_kernel<KD_, BQ_, ES_, CS_, EPL_>), dim3(nbx, chunks), dim3(sig_nw(KD_) * 64), 0, \
                     st, a.NP, a.NQ, a.P, a.Q, a.out, a.sign, a.esum, a.atomic_out, bias, a.out2, a.out_rows,  \
                     a.accumulate, a.p_scale)
  // the two shapes the step uses; KD = 32 also has one general form for any other caller (at KD = 64 the
  // general form spills at two waves per SIMD: not built, the caller falls back to dense.hip)
  const bool zw = bq && es && !cs && epl == 3, wz = !bq && !es && cs && epl == 2;
  if (a.act == 2) {
    // Bernoulli + log_transform: the two shapes of the step at K padded to 32
    if (KD != 32 || !(zw || wz)) return false;
    if (zw)
      hipLaunchKernelGGL((sigdot3_kernel<32, true, true, false, 3, 2>), dim3(nbx, chunks), dim3(sig_nw(32) * 64), 0, st,
                         a.NP, a.NQ, a.P, a.Q, a.out, a.sign, a.esum, a.atomic_out, bias, a.out2, a.out_rows,
                         a.accumulate, a.p_scale);
    else
      hipLaunchKernelGGL((sigdot3_kernel<32, false, false, true, 2, 2>), dim3(nbx, chunks), dim3(sig_nw(32) * 64), 0, st,
                         a.NP, a.NQ, a.P, a.Q, a.out, a.sign, a.esum, a.atomic_out, bias, a.out2, a.out_rows,
                         a.accumulate, a.p_scale);
    return true;
  }
  if (KD == 64) {
    if (zw) SPMF_SIG3(64, true, true, false, 3);
    else if (wz) SPMF_SIG3(64, false, false, true, 2);
    else return false;
    return true;
  }
  if (zw) SPMF_SIG3(32, true, true, false, 3);
  else if (wz) SPMF_SIG3(32, false, false, true, 2);
  else if (bq) SPMF_SIG3(32, true, true, true, 3);
  else SPMF_SIG3(32, false, true, true, 3);
#undef SPMF_SIG3
  return true;
}

bool launch_expdot3(int KD, const ExpdotArgs& a, hipStream_t st) {
  if (KD != 64 || a.act != 0 || a.bias_p || a.bias_q || a.out2 || a.out_rows || a.est) return false;
  const int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  const int nbx = (a.NP + kExpNW * 32 - 1) / (kExpNW * 32);
  hipLaunchKernelGGL((expdot3_kernel<64>), dim3(nbx, chunks), dim3(kExpNW * 64), 0, st, a.NP, a.NQ, a.P, a.Q, a.out,
                     a.sign, a.esum, a.atomic_out, a.accumulate, a.p_scale);
  return true;
}

}  // namespace spmf
