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
        a[pl] = *reinterpret_cast<const bf16x8*>(img + pl * IMG + (32 * sub + r) * kPitch3 + (16 * s + 8 * h) * 2);
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
      e[0] = __builtin_amdgcn_exp2f(fminf(y[0], kYSat * kLog2e));
      e[1] = __builtin_amdgcn_exp2f(fminf(y[1], kYSat * kLog2e));
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
          const uint32_t ad = lds_tr_base + pl * IMG + (32 * sub + 16 * s2) * kPitch3 + 32 * m * 2;
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
  productA(0);
  productA(1);
  expsplit(0);
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
  }
  productC(0);
  productA(2);
  expsplit(1);
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  }
  productC(1);
  productA(3);
  expsplit(2);
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  }
  productC(2);
  expsplit(3);
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
  }
  productC(3);
}

template <int KD>
__global__ __launch_bounds__(512, 2) void expdot3_kernel(int NP, int NQ, const float* __restrict__ P,
                                                          const float* __restrict__ Q, float* __restrict__ out,
                                                          float sign, double* __restrict__ esum, int atomic_out) {
  static_assert(KD == 64, "expdot3b: K padded to 64");
  constexpr int KS = KD / 16, MT = KD / 32, NW = 8, NT = NW * 64;
  constexpr int IMG = kQT3 * kPitch3;
  constexpr int NPC = kQT3 * (KD / 8) / NT;    // (row, 8 k) pieces per loader thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][3][IMG];
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
  float4 st0[NPC], st1[NPC];
  auto gload = [&](int tile) {
    const int q0 = tile * kQT3;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int pc = t + NT * j, row = pc >> 3, k8 = (pc & 7) * 8;
      // rows past NQ read the last row (always in bounds) and are zeroed: no branch, no
      // exec masking around the loads
      const int qr = min(q0 + row, NQ - 1);
      const float* src = Q + (size_t)qr * KD + k8;
      const float keep = q0 + row < NQ ? 1.f : 0.f;
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      st0[j] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
      st1[j] = make_float4(b.x * keep, b.y * keep, b.z * keep, b.w * keep);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int pc = t + NT * j, row = pc >> 3, k8 = (pc & 7) * 8;
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
        *reinterpret_cast<u32x4*>(&lds[buf][pl][row * kPitch3 + k8 * 2]) = c3[pl];
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
    if (more) swrite(buf ^ 1);
    __syncthreads();
  }
  if (p < NP) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float* dst = out + (size_t)p * KD + m * 32 + 8 * g4 + 4 * h;
        const float4 v = make_float4(sign * (acc[m][4 * g4 + 0] + tot[m][4 * g4 + 0]),
                                     sign * (acc[m][4 * g4 + 1] + tot[m][4 * g4 + 1]),
                                     sign * (acc[m][4 * g4 + 2] + tot[m][4 * g4 + 2]),
                                     sign * (acc[m][4 * g4 + 3] + tot[m][4 * g4 + 3]));
        if (atomic_out) {
          atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
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

bool launch_expdot3(int KD, const ExpdotArgs& a, hipStream_t st) {
  if (KD != 64 || a.act != 0 || a.bias_p || a.bias_q || a.out2 || a.out_rows || a.est) return false;
  const int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  const int nbx = (a.NP + 255) / 256;
  hipLaunchKernelGGL((expdot3_kernel<64>), dim3(nbx, chunks), dim3(512), 0, st, a.NP, a.NQ, a.P, a.Q, a.out,
                     a.sign, a.esum, a.atomic_out);
  return true;
}

}  // namespace spmf
