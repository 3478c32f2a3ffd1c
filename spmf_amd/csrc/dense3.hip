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
//     X = Q_tile P_tile^T                 (A = rows of Q from LDS, 4 k-steps x 6 products)
// so X has the P row on the lane and the Q row in the 16 accumulator registers; E = exp(X),
// split in registers, is the B operand of
//     out^T[k, p] += sum_q Q^T[k, q] E[q, p]   (A = columns of Q from LDS, 2 x 2 x 6 MFMAs)
// with the k order of a step fixed by the accumulator's row order (cdna guide, "an
// accumulator tile as the next MFMA's operand": element j of lane half h is row
// 16 s + 8 (j >> 2) + 4 h + (j & 3)).  A Q tile (64 rows) is split once while it is staged and
// kept in LDS as two images per plane: [q][k] for the first product and [k][q in that row
// order] for the second, both with 144-byte rows (36 dwords: the 16 lanes of a ds_read_b128
// service group then cover all 64 banks).
#include "common.h"
#include "kernels.h"

namespace spmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kQT3 = 64;          // Q rows per stage
constexpr int kPitch3 = 144;      // bytes per LDS row (128 of data + 16)
constexpr int kNW3 = 8;           // waves per workgroup

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

// One Q tile (two 32-row sub-tiles) for one wave, branch free (EDGE is a template
// parameter) so that the whole body is ONE scheduling region, written in the order the
// in-order issue should see it:
//   A(s0)                       24 MFMAs, nothing to overlap yet
//   A(s1)  beside  exp/split(s0)    the VALU work of a sub-tile fits the issue slots 24 MFMAs leave
//   C(s0)  beside  exp/split(s1)
//   C(s1)  beside  the split + LDS writes of the NEXT tile's staging registers
// A(.) = X = Q_sub P^T (6 partial products per k-step; a1 b1 and the five small ones accumulate
// apart and are added once: measured, the entry-wise gradient error at exponents of 45 - 60
// falls from 1.1 - 1.6e-5 to 6 - 8e-6, below the exact-f32 kernel's 1.6 - 2.3e-5);
// C(.) = out^T += Q^T E with E in TWO planes: |E - (e1 + e2)| <= 2^-18 E term by term, all terms of
// the sum positive, so 3.8e-6 of the result at worst; e1 (q1 + q2 + q3) + e2 (q1 + q2) = 5 MFMAs.
template <int KD, bool EDGE>
__device__ __forceinline__ void expdot3_tile(const unsigned char* __restrict__ img_row,
                                             const unsigned char* __restrict__ img_col,
                                             const bf16x8 (&pb)[KD / 16][3], f32x16 (&acc)[KD / 32],
                                             float& es_tile, float& xmax, int r, int h, int q0, int NQ, bool p_in) {
  constexpr int KS = KD / 16, MT = KD / 32;
  constexpr int IMG = kQT3 * kPitch3;
  f32x16 xh[2], xl[2];
  u32x4 eb[2][2][2];             // [sub][s2][plane]: 8 bf16 = the B fragment of one k-step
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto productA = [&](int sub) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[pl] = *reinterpret_cast<const bf16x8*>(img_row + pl * IMG + (32 * sub + r) * kPitch3 + (16 * s + 8 * h) * 2);
      // (the first MFMA of each chain takes the constant 0 as its accumulator: no register zeroing)
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pb[s][0], s == 0 ? zero16 : xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][1], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][2], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[s][0], xl[sub], 0, 0, 0);
      xl[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][1], xl[sub], 0, 0, 0);
      xh[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[s][0], s == 0 ? zero16 : xh[sub], 0, 0, 0);
    }
  };
  auto expsplit = [&](int sub) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      float x0 = xh[sub][i] + xl[sub][i], x1 = xh[sub][i + 1] + xl[sub][i + 1];
      float e0, e1;
      if (EDGE) {
        const int qa = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool in0 = p_in && qa < NQ, in1 = p_in && qa + 1 < NQ;
        x0 = in0 ? x0 : 0.f;
        x1 = in1 ? x1 : 0.f;
        e0 = in0 ? __expf(fminf(x0, kYSat)) : 0.f;
        e1 = in1 ? __expf(fminf(x1, kYSat)) : 0.f;
      } else {
        e0 = __expf(fminf(x0, kYSat));
        e1 = __expf(fminf(x1, kYSat));
      }
      xmax = __builtin_fmaxf(xmax, __builtin_fmaxf(x0, x1));
      part += e0 + e1;
      // two planes, two elements per register: cvt_pk, unpack (shift / mask), subtract, cvt_pk
      const uint32_t p1 = pack_bf16(e0, e1);
      const uint32_t p2 = pack_bf16(e0 - bf16_lo(p1), e1 - bf16_hi(p1));
      eb[sub][i >> 3][0][(i & 7) >> 1] = p1;
      eb[sub][i >> 3][1][(i & 7) >> 1] = p2;
    }
    es_tile += part;
  };
  auto productC = [&](int sub) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[pl] = *reinterpret_cast<const bf16x8*>(img_col + pl * IMG + (32 * m + r) * kPitch3 +
                                                   (32 * sub + 16 * s2 + 8 * h) * 2);
        const bf16x8 e1v = __builtin_bit_cast(bf16x8, eb[sub][s2][0]);
        const bf16x8 e2v = __builtin_bit_cast(bf16x8, eb[sub][s2][1]);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e2v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], e1v, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], e1v, acc[m], 0, 0, 0);
      }
  };
  productA(0);
  productA(1);
  expsplit(0);
  // issue order wanted: one MFMA of A(s1), then ~6 VALU of exp/split(s0), 24 times
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
  }
  productC(0);
  expsplit(1);
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
  }
  productC(1);
}

template <int KD>
__global__ __launch_bounds__(kNW3 * 64, 2) void expdot3_kernel(int NP, int NQ, const float* __restrict__ P,
                                                               const float* __restrict__ Q,
                                                               float* __restrict__ out, float sign,
                                                               double* __restrict__ esum, int atomic_out) {
  static_assert(KD == 64, "expdot3: K padded to 64 (BASELINE config 4); other K use the f32-MFMA kernels");
  constexpr int KS = KD / 16;     // k-steps of the first product
  constexpr int MT = KD / 32;     // 32-feature tiles of the second
  constexpr int IMG = kQT3 * kPitch3;          // bytes of one plane of one image (64 rows either way)
  // [buffer][image: 0 rows of Q, 1 columns of Q][plane]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][3][IMG];
  __shared__ double red[16];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int p0 = (blockIdx.x * kNW3 + wid) * 32;
  const int p = p0 + r;
  const int ntiles = (NQ + kQT3 - 1) / kQT3;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc, tile1 = min(ntiles, tile0 + tpc);

  // ---- P fragments: B operand of the first product, B[k = 16 s + 8 h + j][col r] -------
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
  // two-level accumulation as in dense.hip: runs of FOLD tiles (512 terms) in the MFMA
  // accumulator, finished runs summed in `tot`
  constexpr int FOLD = 8;
  f32x16 acc[MT], tot[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = tot[m][i] = 0.f;
  double es = 0.0;
  float xmax = 0.f;

  // ---- stage loader: thread (kcol, g) moves Q[8 rows of group g][kcol] ------------------
  // group g = (a, b): rows 16 a + 4 b + {0..3} and 16 a + 8 + 4 b + {0..3} -- the eight rows
  // whose slots in the column image are contiguous (16 a + 8 b .. + 7)
  const int kcol = t % KD, g = t / KD, ga = g >> 1, gb = g & 1;
  float stage[8];
  auto gload = [&](int tile) {
    const int q0 = tile * kQT3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = q0 + 16 * ga + 8 * (i >> 2) + 4 * gb + (i & 3);
      stage[i] = q < NQ ? Q[(size_t)q * KD + kcol] : 0.f;
    }
  };
  auto swrite = [&](int buf) {
    bf16x8 c3[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const Split3 sp = split3(stage[i]);
      c3[0][i] = sp.a; c3[1][i] = sp.b; c3[2][i] = sp.c;
      const int ql = 16 * ga + 8 * (i >> 2) + 4 * gb + (i & 3);
      // row image: [q][k]
      *reinterpret_cast<__bf16*>(&lds[buf][0][0][ql * kPitch3 + kcol * 2]) = sp.a;
      *reinterpret_cast<__bf16*>(&lds[buf][0][1][ql * kPitch3 + kcol * 2]) = sp.b;
      *reinterpret_cast<__bf16*>(&lds[buf][0][2][ql * kPitch3 + kcol * 2]) = sp.c;
    }
    // column image: [k][slot], the eight slots of this group are one 16-byte piece
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<bf16x8*>(&lds[buf][1][pl][kcol * kPitch3 + (16 * ga + 8 * gb) * 2]) = c3[pl];
  };

  if (tile0 < tile1) {
    gload(tile0);
    swrite(0);
  }
  __syncthreads();
  const bool p_edge = (int)(blockIdx.x * kNW3 * 32 + kNW3 * 32) > NP;      // block-uniform
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = (tile - tile0) & 1;
    const bool more = tile + 1 < tile1;                    // block-uniform
    if (more) gload(tile + 1);                             // lands under this tile's MFMAs
    const int q0 = tile * kQT3;
    float es_tile = 0.f;
    if (p_edge || q0 + kQT3 > NQ)
      expdot3_tile<KD, true>(&lds[buf][0][0][0], &lds[buf][1][0][0], pb, acc, es_tile, xmax, r, h, q0, NQ, p < NP);
    else
      expdot3_tile<KD, false>(&lds[buf][0][0][0], &lds[buf][1][0][0], pb, acc, es_tile, xmax, r, h, q0, NQ, true);
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
  // ---- store: lane holds features (i&3) + 8(i>>2) + 4h (+ 32 m) of row p ------------------
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
    // esum[1] = dacc[4]: workgroups in which an exponent exceeded kYSat (dense.hip)
    const double ts = block_sum(xmax > kYSat ? 1.0 : 0.0, red);
    if (t == 0 && ts != 0.0) atomicAdd(esum + 1, 1.0);
  }
}

bool launch_expdot3(int KD, const ExpdotArgs& a, hipStream_t st) {
  if (KD != 64 || a.act != 0 || a.bias_p || a.bias_q || a.out2 || a.out_rows || a.est) return false;
  const int nbx = (a.NP + kNW3 * 32 - 1) / (kNW3 * 32);
  const int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  hipLaunchKernelGGL((expdot3_kernel<64>), dim3(nbx, chunks), dim3(kNW3 * 64), 0, st, a.NP, a.NQ, a.P, a.Q,
                     a.out, a.sign, a.esum, a.atomic_out);
  return true;
}

}  // namespace spmf
