// dense.hip -- the dense part of the log_transform decoder on the f32 matrix
// cores (gfx950).
//
// With f(y) = exp(y*eta) - 1 (mederrata_spmf/poisson.py:52-53) the sum of the
// Poisson rate over ALL B*D cells has no closed form:
//   sum_all r   = sum_{b,d} E_bd - B*D + B*sum_d phi_d,   E_bd = exp(<z_b, W_d>), W_d = eta_d v_d
//   d/dz_b     -= sum_d E_bd W_d          ("GZ dense")
//   d/dW_d     -= sum_b E_bd z_b          ("GW dense")
// Both are the same operator with the operands swapped:
//   expdot(P, Q):  out_p[k] = sum_q exp(<P_p, Q_q>) Q_q[k],   esum = sum_{p,q} exp(<P_p, Q_q>)
// GZ = expdot(Z, W), GW = expdot(W, Z).  The stored cells' x*log r term and its
// gradient stay in the sparse row/column passes.
//
// Kernel: P-stationary.  A wave owns 32 rows of P as the B operand of
//   X = Q_tile * P_tile^T      (v_mfma_f32_32x32x2_f32, KD/2 steps)
// so X has the P row on the lane and the Q row in the 16 accumulator
// registers; E = exp(X) is then ALREADY the B operand of the second product
//   out^T[k, p] += sum_q Q^T[k, q] E[q, p]      (16 steps per 32 features)
// (an accumulator tile feeds a following MFMA that sums over its row index with
// no lane movement; the k order of a step is the accumulator's row order
// rho_h(t) = (t&3) + 8(t>>2) + 4h, which the A operand reads from LDS).
// Q tiles (128 rows) are staged through LDS with an odd pitch so both the
// column-strided A reads of product 1 and the row reads of product 2 are
// conflict free.  exact-f32 MFMA == fmaf chain, so parity is that of fp32.
//
// Roofline: MFMA f32 (157 TF dense peak): 4*NP*NQ*KD flop per launch.
#include "common.h"
#include "kernels.h"

namespace spmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 128;  // Q rows per LDS stage

template <int PB>
struct XB {
  f32x16 v[PB];
};

// ACT 0: E = exp(X)                         esum += sum E          (Poisson, log_transform)
// ACT 2: l = exp(X) - 1 + bias, E = sigmoid(l) * exp(X) (= d softplus(l)/dX), esum += sum softplus(l),
//        out2 gets the sums of sigmoid(l)    (Bernoulli logits with the exp decoder, bernoulli.py:60-61)
// ACT 1: E = sigmoid(X + bias)              esum += sum softplus(X + bias)   (Bernoulli logits,
//        bernoulli.py:147-155; bias = phi of the column: bias_q when Q rows are columns,
//        bias_p when P rows are columns); out2[p] += sign * sum_q E (the d/dphi column sums)
template <int KD, int ACT>
__global__ __launch_bounds__(256, (KD == 32 && ACT == 1) ? 2 : 1) void expdot_kernel(int NP, int NQ, const float* __restrict__ P,
                                                     const float* __restrict__ Q,
                                                     float* __restrict__ out, float sign,
                                                     double* __restrict__ esum, int atomic_out,
                                                     const float* __restrict__ bias_p,
                                                     const float* __restrict__ bias_q,
                                                     float* __restrict__ out2,
                                                     const int32_t* __restrict__ out_rows,
                                                     float* __restrict__ est, int64_t ldE) {
  constexpr int PITCH = KD + 4;   // 16-B aligned rows; (KD+4) % 64 = 4 keeps b128 column reads conflict free
  constexpr int KH = KD / 2;      // lane half h covers k in [h*KH, (h+1)*KH): any k order is valid
                                  // as long as A and B agree, and this one makes A a contiguous read
  constexpr int MT = KD / 32;     // 32-feature tiles of the second product
  // P blocks (of 32 rows) per wave: at KD = 32 a sub-tile is only 32 MFMAs, so a
  // wave carries two P blocks that share every LDS operand read and the loop
  // overhead (the MFMA : overhead ratio of the KD = 64 form)
  constexpr int PB = KD == 32 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float qs[2][QT * PITCH];
  __shared__ double red[16];
  __shared__ float bqs[2][QT];   // ACT 1: logit bias of the staged Q rows
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int p0 = (blockIdx.x * 4 + wid) * 32 * PB;
  // Q range of this block (gridDim.y chunks, whole QT tiles)
  const int ntiles = (NQ + QT - 1) / QT;
  const int tpc = (ntiles + gridDim.y - 1) / gridDim.y;
  const int tile0 = blockIdx.y * tpc;
  const int tile1 = min(ntiles, tile0 + tpc);

  // P fragments: B operand of product 1, step s: B[k][j=c] = P[p][h*KH + s]
  float pb[PB][KH];
  float bp[PB];
  // Two-level accumulation: `acc` is the MFMA accumulator of the current run of FOLD
  // Q tiles (FOLD*128 terms per output, all of one sign), `tot` the sum of the
  // finished runs.  One fp32 chain over all NQ terms (30 000 at C4) has a rounding
  // error of ~sqrt(NQ)*2^-24 of the SUM, which the difference with the stored-cell
  // term then amplifies (6.9e-5 of max|du| on a C4 slice); runs of 512 bring the
  // chain error down to ~sqrt(512)*2^-24.
  // (not in the KD = 32 sigmoid form: it runs two waves per SIMD on a 256-register
  // budget that the second accumulator set would spill)
  constexpr int FOLD = 4;
  constexpr bool TWO_LEVEL = !(KD == 32 && ACT == 1);
  f32x16 acc[PB][MT], tot[TWO_LEVEL ? PB : 1][TWO_LEVEL ? MT : 1];
  float colsum[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    const int p = p0 + b * 32 + c;
#pragma unroll
    for (int s4 = 0; s4 < KH / 4; ++s4) {
      const float4 v = p < NP ? *reinterpret_cast<const float4*>(P + (size_t)p * KD + h * KH + 4 * s4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[b][4 * s4 + 0] = v.x; pb[b][4 * s4 + 1] = v.y;
      pb[b][4 * s4 + 2] = v.z; pb[b][4 * s4 + 3] = v.w;
    }
    bp[b] = (ACT >= 1 && bias_p && p < NP) ? bias_p[p] : 0.f;
    colsum[b] = 0.f;                                     // ACT 1: sum_q E for this lane's p
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[b][m][i] = 0.f;
        if (TWO_LEVEL) tot[b][m][i] = 0.f;
      }
  }
  double es = 0.0;
  float xmax = 0.f;       // ACT 0: largest exponent this lane saw (saturation at kYSat)

  // stage loader: 256 threads move the QT*KD floats of a tile in QT/32 parts
  // (one 32-row part per sub-tile of the compute loop: only PER4 registers live)
  constexpr int PER4 = 32 * KD / 256 / 4;   // float4 per thread per 32-row part
  float4 stage[PER4];
  float bstage = 0.f;
  auto gload = [&](int tile, int part) {
    const int q0 = tile * QT + part * 32;
    if (ACT >= 1 && t < 32) bstage = (bias_q && q0 + t < NQ) ? bias_q[q0 + t] : 0.f;
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = (i * 256 + t) * 4;
      const int r = e / KD, k = e % KD;
      stage[i] = (q0 + r < NQ) ? *reinterpret_cast<const float4*>(Q + (size_t)(q0 + r) * KD + k)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto swrite = [&](int buf, int part) {
    if (ACT >= 1 && t < 32) bqs[buf][part * 32 + t] = bstage;
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = (i * 256 + t) * 4;
      const int r = e / KD, k = e % KD;
      *reinterpret_cast<float4*>(&qs[buf][(part * 32 + r) * PITCH + k]) = stage[i];
    }
  };

  if (tile0 < tile1) {
#pragma unroll
    for (int part = 0; part < QT / 32; ++part) {
      gload(tile0, part);
      swrite(0, part);
    }
  }
  __syncthreads();
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = (tile - tile0) & 1;
    const bool more = tile + 1 < tile1;     // block-uniform
    // interior tiles need no masking (block-uniform)
    const bool edge = (tile * QT + QT > NQ) || ((int)(blockIdx.x * 128 * PB + 128 * PB) > NP);
    const float* qb = qs[buf];
    const int q0 = tile * QT;
    // Software pipeline over the 4 sub-tiles of the stage: product 1 of
    // sub-tile s+1 (MFMA chains) is issued in the same basic block as the
    // exp() VALU work of sub-tile s, so the two pipes overlap inside one wave.
    // Rows past NQ hold zeros in LDS and are masked in E, so no early exit.
    auto product1 = [&](const float* qt, int sub) {
      XB<PB> xb;
      f32x16* x = xb.v;
      // ACT 1: the logit biases ride in the accumulator instead of the epilogue
#pragma unroll
      for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          x[b][i] = ACT == 1 ? bp[b] + bqs[buf][sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] : 0.f;
#pragma unroll
      for (int s4 = 0; s4 < KH / 4; ++s4) {
        const float4 a = *reinterpret_cast<const float4*>(qt + c * PITCH + h * KH + 4 * s4);
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          x[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, pb[b][4 * s4 + 0], x[b], 0, 0, 0);
          x[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, pb[b][4 * s4 + 1], x[b], 0, 0, 0);
          x[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, pb[b][4 * s4 + 2], x[b], 0, 0, 0);
          x[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, pb[b][4 * s4 + 3], x[b], 0, 0, 0);
        }
      }
      return xb;
    };
    float es_tile = 0.f;
    // (with two P blocks per wave the two MFMA chains already interleave; carrying
    // a second pair of tiles across sub-tiles only costs registers)
    constexpr bool PIPE = PB == 1;
    XB<PB> xc, xn;
    if (PIPE) xc = product1(qb, 0);
    f32x16* xcur = xc.v;
    constexpr int UNR = PIPE ? QT / 32 : 1;
#pragma unroll UNR
    for (int sub = 0; sub < QT / 32; ++sub) {
      const float* qt = qb + sub * 32 * PITCH;
      if (more) gload(tile + 1, sub);       // lands under this sub-tile's MFMAs
      if (!PIPE) xc = product1(qt, sub);
      if (PIPE && sub + 1 < QT / 32) xn = product1(qt + 32 * PITCH, sub + 1);
      // A operands of product 2 for the whole sub-tile, issued before the exp
      // block so their LDS latency is not paid per MFMA pair
      float aq[16][MT];
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
        const int row = (tt & 3) + 8 * (tt >> 2) + 4 * h;
#pragma unroll
        for (int m = 0; m < MT; ++m) aq[tt][m] = qt[row * PITCH + m * 32 + c];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- E = act(X), masked outside [NQ) x [NP) on edge tiles ------------
      // exp via v_exp_f32 (2^x): |rel err| ~ 1e-7 * (1 + |x|), inside the 1e-5 budget
      float part = 0.f;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const int p = p0 + b * 32 + c;
        if (ACT == 2) {
          // l = exp(X) - 1 + bias; E = sigmoid(l) * exp(X); sum softplus(l); colsum = sum sigmoid(l)
          float pmax = 0.f, plog = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int ql = sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool in = !edge || (q0 + ql < NQ && p < NP);
            const float ex = __expf(fminf(xcur[b][i], kYSat));
            const float l = ex - 1.f + bp[b] + bqs[buf][ql];
            const float en = __expf(-fabsf(l));
            const float d = 1.f + en;
            const float inv = __builtin_amdgcn_rcpf(d);
            const float sg = in ? (l >= 0.f ? inv : en * inv) : 0.f;
            xcur[b][i] = sg * ex;
            pmax += in ? fmaxf(l, 0.f) : 0.f;
            plog += in ? __builtin_amdgcn_logf(d) : 0.f;
            colsum[b] += sg;
          }
          part += pmax + 0.69314718056f * plog;
        } else if (ACT == 1) {
          // softplus(l) = max(l,0) + ln2*log2(1+e^-|l|): the two sums are kept apart
          // so the ln2 factor is applied once per sub-tile
          float pmax = 0.f, plog = 0.f, dprod = 1.f;
          if (edge) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int ql = sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
              const bool in = q0 + ql < NQ && p < NP;
              const float l = xcur[b][i];
              const float en = __expf(-fabsf(l));             // exp(-|l|) in (0,1]
              const float d = 1.f + en;
              const float inv = __builtin_amdgcn_rcpf(d);
              const float sg = in ? (l >= 0.f ? inv : en * inv) : 0.f;   // sigmoid(l)
              xcur[b][i] = sg;
              pmax += in ? fmaxf(l, 0.f) : 0.f;
              dprod *= in ? d : 1.f;
              colsum[b] += sg;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float l = xcur[b][i];
              const float en = __expf(-fabsf(l));
              const float d = 1.f + en;
              const float inv = __builtin_amdgcn_rcpf(d);
              const float sg = l >= 0.f ? inv : en * inv;
              xcur[b][i] = sg;
              pmax += fmaxf(l, 0.f);
              dprod *= d;
              colsum[b] += sg;
            }
          }
          // sum_i log2(1 + e^-|l_i|) = log2 of the product: sixteen factors in (1, 2] stay below
          // 2^16, so ONE v_log per sub-tile and lane replaces sixteen (quarter-rate instructions)
          plog = __builtin_amdgcn_logf(dprod);
          part += pmax + 0.69314718056f * plog;
        } else if (edge) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int q = q0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const bool in = q < NQ && p < NP;
            const float xv = in ? xcur[b][i] : 0.f;
            const float e = in ? __expf(fminf(xv, kYSat)) : 0.f;   // saturating: common.h kYSat
            xmax = fmaxf(xmax, xv);
            xcur[b][i] = e;
            part += e;
          }
        } else {
#pragma unroll
          // (the largest exponent is tracked two elements per v_max3: the epilogue's VALU
          //  work is not free beside the MFMAs -- a compare + count per element cost 12 %)
          for (int i = 0; i < 16; i += 2) {
            const float x0 = xcur[b][i], x1 = xcur[b][i + 1];
            xmax = __builtin_fmaxf(xmax, __builtin_fmaxf(x0, x1));
            const float e0 = __expf(fminf(x0, kYSat)), e1 = __expf(fminf(x1, kYSat));
            xcur[b][i] = e0;
            xcur[b][i + 1] = e1;
            part += e0 + e1;
          }
        }
      }
      es_tile += part;
      // ---- E kept for the second contraction (estdot_kernel): 32-row Q tiles, P-major inside a
      // tile: est[((q/32)*ldE + p)*32 + q%32]; a lane's four register runs are 16-B pieces of
      // row p's 128-B line, which the lane pair (c, h=0/1) writes whole
      if (est && q0 + sub * 32 < NQ) {     // (sub-tiles wholly past NQ have no E tile)
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          const int p = p0 + b * 32 + c;
          if (p < NP) {
            float* dst = est + ((size_t)((q0 + sub * 32) >> 5) * (size_t)ldE + (size_t)p) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(dst + 8 * g) =
                  make_float4(xcur[b][4 * g + 0], xcur[b][4 * g + 1], xcur[b][4 * g + 2], xcur[b][4 * g + 3]);
          }
        }
      }
      // ---- product 2: out^T[k][p] += sum_q Q[q][k] E[q][p] ---------------
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int b = 0; b < PB; ++b)
            acc[b][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[tt][m], xcur[b][tt], acc[b][m], 0, 0, 0);
      }
      if (PIPE && sub + 1 < QT / 32) xc = xn;
      if (more) swrite(buf ^ 1, sub);
    }
    es += (double)es_tile;
    if (TWO_LEVEL && ((tile - tile0) % FOLD) == FOLD - 1) {      // block-uniform: close the run
#pragma unroll
      for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            tot[b][m][i] += acc[b][m][i];
            acc[b][m][i] = 0.f;
          }
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < PB; ++b)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (TWO_LEVEL) acc[b][m][i] += tot[b][m][i];
  // ---- store: lane holds features (i&3)+8(i>>2)+4h (+32m) of row p --------
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    const int p = p0 + b * 32 + c;
    // out_rows: P holds a compacted subset of rows (mixed likelihood: the Bernoulli columns);
    // results go to the rows of the full-size output they came from
    const int prow = (out_rows && p < NP) ? out_rows[p] : p;
    if (p < NP) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* dst = out + (size_t)prow * KD + m * 32 + 8 * g + 4 * h;
          if (atomic_out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sign * acc[b][m][4 * g + j]);
          } else {
            *reinterpret_cast<float4*>(dst) =
                make_float4(sign * acc[b][m][4 * g + 0], sign * acc[b][m][4 * g + 1],
                            sign * acc[b][m][4 * g + 2], sign * acc[b][m][4 * g + 3]);
          }
        }
    }
    if (ACT >= 1 && out2) {
      float cs = colsum[b];
      cs += __shfl_xor(cs, 32);                         // the two lane halves hold disjoint q rows
      if (h == 0 && p < NP && cs != 0.f) atomicAdd(&out2[prow], sign * cs);
    }
  }
  if (esum) {
    const double tot = block_sum(es, red);
    if (t == 0) atomicAdd(esum, tot);
    if (ACT == 0) {
      // esum[1] = dacc[4]: number of workgroups (128 or 256 P rows x their Q range) in
      // which an exponent exceeded kYSat and was saturated: 0 <=> the decoder was exact
      const double ts = block_sum(xmax > kYSat ? 1.0 : 0.0, red);
      if (t == 0 && ts != 0.0) atomicAdd(esum + 1, 1.0);
    }
  }
}

// Second contraction of the exp terms from the STORED E (written by expdot_kernel, layout above):
//   out[q][k] += sign * sum_p E[q][p] P[p][k]
// E is computed once per step (6*B*D*K flop instead of 8) at the price of one write and one
// read of B*D floats in HBM, which the 288 GB part has room for (chunked by the caller).
// A wave owns IT = 2 tiles of 32 Q rows as the A operand rows (lane = q, read straight from
// global memory: one step = the two adjacent 128-B lines of p = 2s, 2s+1), the P rows of a
// round (32 of them) are staged in LDS as the B operand for all 8 waves; the accumulator has
// q in its registers and k on the lane, so the epilogue writes 128-B rows.
// Roofline: MFMA f32, 2*NQ*NP*KD flop; HBM NQ*NP*4 bytes (2.5 TB/s at the MFMA rate).
template <int KD, int NW, bool RS>   // RS: also the row sums of E (out2)
__global__ __launch_bounds__(NW * 64, 8 / NW) void estdot_kernel(int NQ, int NP, int64_t ldE,
                                                                 const float* __restrict__ est,
                                                                 const float* __restrict__ P,
                                                                 float* __restrict__ out, float sign,
                                                                 float* __restrict__ out2,
                                                                 const int32_t* __restrict__ out_rows) {
  // out2 (Bernoulli: d/dphi): out2[q] += sign * sum_p E[q][p]; out_rows: Q is a compacted
  // subset of the output rows (mixed likelihood)
  constexpr int MT = KD / 32;
  constexpr int IT = 2;
  constexpr int RP = 32;                      // P rows per round
  constexpr int ZP = KD == 64 ? 96 : 32;      // pitch % 64 == 32: the two lane halves (rows 2s, 2s+1) hit disjoint banks
  constexpr int FOLD = 16;                    // rounds per fp32 run (two-level accumulation, as in expdot_kernel)
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) float zs[2][RP * ZP];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int qtiles = (NQ + 31) / 32;
  const int qt0 = (blockIdx.x * NW + wid) * IT;
  // P range of this block: gridDim.y chunks of whole rounds
  const int rounds = (NP + RP - 1) / RP;
  const int rpc = (rounds + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rpc, r1 = min(rounds, r0 + rpc);
  f32x16 acc[IT][MT], tot[IT][MT];
  float rs[IT], rst[IT];            // row sums of E (two-level like the accumulators), out2 only
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    rs[it] = rst[it] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[it][m][i] = tot[it][m][i] = 0.f;
  }
  // stage loader: NT threads move RP*KD floats
  constexpr int PER4 = (RP * KD / 4 + NT - 1) / NT;
  float4 stage[PER4];
  auto gload = [&](int r) {
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = (i * NT + t) * 4;
      const int row = e / KD, k = e % KD;
      const int p = r * RP + row;
      stage[i] = (e < RP * KD && p < NP) ? *reinterpret_cast<const float4*>(P + (size_t)p * KD + k)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int e = (i * NT + t) * 4;
      const int row = e / KD, k = e % KD;
      if (e < RP * KD) *reinterpret_cast<float4*>(&zs[buf][row * ZP + k]) = stage[i];
    }
  };
  float a0[IT][16], a1[IT][16];
  // A fragments of round r.  No predicates (a predicated load is a branch and a full wait each):
  // a wave past the last Q tile reads the last tile (its results are dropped in the epilogue) and
  // rows past NP are clamped to NP-1 -- their partners in LDS are zeros.  Whole rounds (all but
  // possibly the last) use immediate offsets from one base address.
  auto aload = [&](float (&dst)[IT][16], int r) {
    const bool whole = (r + 1) * RP <= NP;                  // block-uniform
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int qt = min(qt0 + it, qtiles - 1);
      const float* tile = est + (size_t)qt * (size_t)ldE * 32 + c;
      if (whole) {
        const float* base = tile + ((size_t)r * RP + h) * 32;
#pragma unroll
        for (int s = 0; s < 16; ++s) dst[it][s] = base[s * 64];
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) dst[it][s] = tile[(size_t)min(r * RP + 2 * s + h, NP - 1) * 32];
      }
    }
  };
  auto compute = [&](const float (&a)[IT][16], const float* zb, int r) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float bz[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) bz[m] = zb[(2 * s + h) * ZP + m * 32 + c];
#pragma unroll
      for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[it][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it][s], bz[m], acc[it][m], 0, 0, 0);
    }
    if (RS) {
      // (rows past NP were read clamped to NP-1: zero partners in the product, but not here)
      const int nv = NP - r * RP;
#pragma unroll
      for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int s = 0; s < 16; ++s) rs[it] += (2 * s + h < nv) ? a[it][s] : 0.f;
    }
  };
  auto fold = [&]() {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (RS) rst[it] += rs[it];
      rs[it] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          tot[it][m][i] += acc[it][m][i];
          acc[it][m][i] = 0.f;
        }
  };
  if (r0 < r1) {
    gload(r0);
    swrite(0);
    aload(a0, r0);
  }
  __syncthreads();
  // two rounds per trip: the A fragments ping-pong between two register sets (no copies)
  for (int r = r0; r < r1; r += 2) {
    const bool more1 = r + 1 < r1, more2 = r + 2 < r1;     // block-uniform
    if (more1) {
      gload(r + 1);
      aload(a1, r + 1);
    }
    compute(a0, zs[0], r);
    if (more1) swrite(1);
    __syncthreads();
    if (more1) {
      if (more2) {
        gload(r + 2);
        aload(a0, r + 2);
      }
      compute(a1, zs[1], r + 1);
      if (more2) swrite(0);
      __syncthreads();
    }
    if ((((r - r0) >> 1) % (FOLD / 2)) == FOLD / 2 - 1) fold();
  }
  // epilogue: register i of tile (it, m) is row q = 32*(qt0+it) + rho_h(i), lane c is feature m*32 + c
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (qt0 + it >= qtiles) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = (qt0 + it) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (q < NQ) {
          const int row = out_rows ? out_rows[q] : q;
          atomicAdd(out + (size_t)row * KD + m * 32 + c, sign * (acc[it][m][i] + tot[it][m][i]));
        }
      }
    if (RS && out2) {
      // lane (c, h) summed E[q = 32*(qt0+it) + c][p of half h]
      float r = rs[it] + rst[it];
      r += __shfl_xor(r, 32);
      const int q = (qt0 + it) * 32 + c;
      if (h == 0 && q < NQ && r != 0.f) atomicAdd(&out2[out_rows ? out_rows[q] : q], sign * r);
    }
  }
}

#ifndef SPMF_ESTDOT_NW
#define SPMF_ESTDOT_NW 8
#endif
void launch_estdot(int KD, int NQ, int NP, int64_t ldE, const float* est, const float* P, float* out, float sign,
                   float* out2, const int32_t* out_rows, hipStream_t st) {
  const int qtiles = (NQ + 31) / 32;
  constexpr int NW = SPMF_ESTDOT_NW;
  const int nbx = (qtiles + 2 * NW - 1) / (2 * NW);     // NW waves x 2 tiles per workgroup
  const int rounds = (NP + 31) / 32;
  int chunks = (768 * 8 / NW) / nbx;                    // 3 rounds of resident workgroups, no partial fourth
  if (chunks > rounds) chunks = rounds;
  if (chunks < 1) chunks = 1;
  dim3 grid(nbx, chunks);
#define SPMF_ESTDOT(KD_, RS_)                                                                        \
  hipLaunchKernelGGL((estdot_kernel<KD_, NW, RS_>), grid, dim3(NW * 64), 0, st, NQ, NP, ldE, est, P, out, \
                     sign, out2, out_rows)
  if (KD == 64 && out2) SPMF_ESTDOT(64, true);
  else if (KD == 64) SPMF_ESTDOT(64, false);
  else if (KD == 32 && out2) SPMF_ESTDOT(32, true);
  else if (KD == 32) SPMF_ESTDOT(32, false);
#undef SPMF_ESTDOT
}

// Mixed likelihood: the dense softplus/sigmoid sums run over the Bernoulli columns only.
// Vb[j] = V'[cols[j]], bb[j] = phi[cols[j]] (one wave per 64 floats of a row).
__global__ __launch_bounds__(256) void compact_rows_kernel(int n, int KD, const int32_t* __restrict__ cols,
                                                           const float* __restrict__ Vp,
                                                           const float* __restrict__ phi,
                                                           float* __restrict__ Vb, float* __restrict__ bb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)n * KD;
  if (i < tot) {
    const int j = (int)(i / KD), k = (int)(i % KD);
    const int d = cols[j];
    Vb[i] = Vp[(size_t)d * KD + k];
    if (k == 0) bb[j] = phi[d];
  }
}
void launch_compact_rows(int n, int KD, const int32_t* cols, const float* Vp, const float* phi, float* Vb,
                         float* bb, hipStream_t st) {
  if (n <= 0) return;
  const int64_t tot = (int64_t)n * KD;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, KD, cols, Vp,
                     phi, Vb, bb);
}

void launch_expdot(int KD, const ExpdotArgs& a, hipStream_t st) {
  const int pb = KD == 32 ? 2 : 1;
  const int nbx = (a.NP + 128 * pb - 1) / (128 * pb);
  int chunks = a.q_chunks < 1 ? 1 : a.q_chunks;
  dim3 grid(nbx, chunks);
#define SPMF_ED_LAUNCH(KD_, ACT_)                                                              \
  hipLaunchKernelGGL((expdot_kernel<KD_, ACT_>), grid, dim3(256), 0, st, a.NP, a.NQ, a.P, a.Q, \
                     a.out, a.sign, a.esum, a.atomic_out, a.bias_p, a.bias_q, a.out2, a.out_rows,  \
                     a.est, a.ldE)
  if (KD == 32 && a.act == 0) SPMF_ED_LAUNCH(32, 0);
  else if (KD == 32 && a.act == 1) SPMF_ED_LAUNCH(32, 1);
  else if (KD == 32) SPMF_ED_LAUNCH(32, 2);
  else if (KD == 64 && a.act == 0) SPMF_ED_LAUNCH(64, 0);
  else if (KD == 64 && a.act == 1) SPMF_ED_LAUNCH(64, 1);
  else if (KD == 64) SPMF_ED_LAUNCH(64, 2);
#undef SPMF_ED_LAUNCH
}

}  // namespace spmf
