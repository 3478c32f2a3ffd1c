// row_pass.hip -- row-resident fused forward of the sparse Poisson energy
// (gfx950, wave64).
//
// Per stored row b of the batch (one wavefront per row, rows grid-strided):
//   sweep 1  z_b   = xi_b * sum_{d in nnz(b)} x_bd A'_d          encode,
//                                               poisson.py:640-649
//   sweep 2  r_bd  = <z_b, V'_d> + phi_d                          :174-177
//            ll   += x_bd log r_bd   (lgamma(x+1) is parameter free and is
//                                     pre-summed per batch)       :178-183
//            gz_b  = sum_d (x_bd/r_bd) V'_d - veta - z_b          d(x+z)/dz_b
//   out      z_b, xi_b*gz_b (the column pass needs exactly that product),
//            fp64: sum x log r, sum z^2, sum_b z_b, #non-finite cells.
//
// Lane layout: a gathered factor row is KP floats = LPN=KP/4 lanes x float4,
// so one wave instruction serves NPI=64/LPN stored entries; the LPN lanes of
// an entry fold their partial dot products with DPP adds.  The transcendental
// part (log, divide) runs once per 64 entries with one entry per lane: lane
// (grp,sub) keeps the dot product of iteration `sub`.  col/val are read as
// one coalesced 256-B wave load per 64 entries and distributed by
// ds_bpermute.  Gathers are issued in straight-line groups of up to four
// (padded entries ask for a row behind the table: dropped by the buffer range
// check, common.h GTable) so several L2 round trips are in
// flight per wave; rows of <= 128 entries keep col/val in registers for both
// sweeps.
//
// Roofline: HBM traffic is 8 B per stored entry (+ an L2-served re-read for
// long rows) + 8*KP B per row written; the factor-row gathers (2 x 4*KP B per
// entry) are served by L2/Infinity Cache because A' and V' (D*KP*4 B each)
// stay resident.  Algorithmic bytes: DESIGN.md section 4.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace spmf {

#ifndef ROW_MAX_BLOCKS
#define ROW_MAX_BLOCKS 4096
#endif
// share of a wave's rows that the counters hand out (the dynamic tail of the row loop, below)
#ifndef ROW_DYN_NUM
#define ROW_DYN_NUM 1
#define ROW_DYN_DEN 8
#endif
static_assert(ROW_MAX_BLOCKS <= spmf::kDetMaxBlocks,
              "the deterministic mode's per-workgroup slots (common.h kDetMaxBlocks) are sized for at most that many row-pass workgroups");
#ifndef ROW_GRP
#define ROW_GRP 4
#endif
// short rows: the V' gathers of a row's first chunk are issued before sweep 1 (RowCtx::s2_load)
#ifndef ROW_V_AHEAD
#define ROW_V_AHEAD 2
#endif
#ifndef ROW_WAVES_PER_SIMD
#define ROW_WAVES_PER_SIMD 1
#endif

namespace {

// The intercept phi_d is needed once per stored entry ("one entry per lane": 64 different
// cache lines per wave load).  tools/gather_rows_probe.hip: that third gather stream costs
// the row pass 0.35 ms of 1.9 on C3 (mode 6 vs mode 2); staged in LDS once per workgroup
// (4*D bytes, 80 KB at D = 20 000, two 512-thread workgroups per CU) it costs 0.1.
__device__ __forceinline__ float* lds_dyn() {
  extern __shared__ __attribute__((aligned(16))) float spmf_row_lds[];
  return spmf_row_lds;
}

// One stored entry: from the packed stream (col << 16 | count: half the bytes) when the batch
// carries one, from the canonical col / val arrays otherwise.  PACKED is a template parameter of the
// kernel (the resident-set launches have both forms).  load_entry only LOADS (into the pipeline registers of the row prefetch);
// unpack_entry turns them into (column, count) where they are used, one row later -- arithmetic
// on the loaded word right away would put the wait for the prefetch in front of it.
// FMT 0: canonical (col, val).  FMT 1: packed word.  FMT 2: packed word + a second value stream
// (`val` = g(x) of the log_transform encoder, poisson.py:41-42): the fused pass of the exp decoders
// reads g(x) in sweep 1 and the count out of the word in sweep 2 -- the register cost of the canonical pair.
template <int FMT, bool NT>
__device__ __forceinline__ void load_entry(const int32_t* __restrict__ col, const float* __restrict__ val,
                                           const uint32_t* __restrict__ ent, int i, bool ok, int& ra,
                                           float& rb) {
  if (FMT == 1) {
    ra = ok ? (int)(NT ? __builtin_nontemporal_load(&ent[i]) : ent[i]) : (int)kPadWord;
    rb = 0.f;
  } else if (FMT == 2) {
    ra = ok ? (int)(NT ? __builtin_nontemporal_load(&ent[i]) : ent[i]) : (int)kPadWord;
    rb = ok ? (NT ? __builtin_nontemporal_load(&val[i]) : val[i]) : 0.f;
  } else {
    ra = ok ? (NT ? __builtin_nontemporal_load(&col[i]) : col[i]) : kPadRow;
    rb = ok ? (NT ? __builtin_nontemporal_load(&val[i]) : val[i]) : 0.f;
  }
}
// SWEEP 1: the encoder's value (g(x) under FMT 2); SWEEP 2: the count
template <int FMT, int SWEEP = 2>
__device__ __forceinline__ void unpack_entry(int ra, float rb, int& c, float& x) {
  if (FMT == 1 || (FMT == 2 && SWEEP == 2)) {
    c = (int)((uint32_t)ra >> 16);
    x = (float)((uint32_t)ra & 0xffffu);
  } else if (FMT == 2) {
    c = (int)((uint32_t)ra >> 16);
    x = rb;
  } else {
    c = ra;
    x = rb;
  }
}

// Lane `base/4 + IMM` 's value: ds_bpermute with the compile-time part of the source lane in the instruction's
// offset field (the row pass broadcasts from lanes (g0 + j) * NPI + grp and grp * LPN + g0 + j: one base
// register per form for the whole kernel instead of a shift / add per broadcast -- ROW_BPERM_IMM)
// sweep 2 at K = 32: the chunk's eight dot products through one transpose-reduce (RowCtx::sweep2_loaded)
// (measured: 1.443 against 1.411 ms on C3 -- the fourteen selects and two spilled registers cost more than the
//  seventeen folds they replace; built, parity-tested, off: profiles/r05_row_valu_ab.txt)
#ifndef ROW_DOT_BUTTERFLY
#define ROW_DOT_BUTTERFLY 0
#endif
// sweep 2's dot products with the packed multiply / fma (three instructions instead of four): measured, no
// difference (1.409 against 1.402 ms): off
#ifndef ROW_DOT_PK
#define ROW_DOT_PK 0
#endif
#ifndef ROW_BPERM_IMM
#define ROW_BPERM_IMM 1
#endif
// (imm_lanes is a constant after the unrolled callers are inlined.  BASE false = the __shfl form: at K = 64 the
//  sixteen base + constant sums of a chunk get hoisted into registers the kernel does not have -- 132 instead of
//  76 bytes of scratch, C4's row launches 12.7 -> 13.2 ms -- so that instantiation keeps the shift per call)
template <bool BASE>
__device__ __forceinline__ int bperm_i(int base_bytes, int imm_lanes, int v) {
  if constexpr (BASE) return __builtin_amdgcn_ds_bpermute(base_bytes + imm_lanes * 4, v);
  else return __shfl(v, (base_bytes >> 2) + imm_lanes);
}
template <bool BASE>
__device__ __forceinline__ float bperm_f(int base_bytes, int imm_lanes, float v) {
  return __int_as_float(bperm_i<BASE>(base_bytes, imm_lanes, __float_as_int(v)));
}

// LIK: 0 Poisson / linear decoder, 1 Poisson / log_transform, 2 Bernoulli(logits) / linear
// LDSPHI: phi is read from the workgroup's LDS copy instead of global memory
template <int KP, int LIK, bool LDSPHI = false>
struct RowCtx {
  static constexpr int LPN = KP / 4;
  static constexpr int NPI = 64 / LPN;
  static constexpr int GRP = LPN < ROW_GRP ? LPN : ROW_GRP;  // gathers issued back to back
  // broadcasts through a base register: K <= 32; not the exp decoder's instantiations (they have no register to
  // spare either: 12 - 20 bytes of scratch with it)
  static constexpr bool BPI = ROW_BPERM_IMM && LPN <= 8 && LIK != 1;
  GTable Ap, Vp;          // the gathered tables (common.h: slots behind a row's end are dropped by the range check)
  const float* phi;
  const uint8_t* ctype;   // LIK 3 (mixed): 1 = Bernoulli column
  int lane, sub, grp;
  int bp_grp, bp_row;     // byte addresses of lanes grp and grp * LPN (bperm_i / bperm_f)

  // z partial: zacc += sum over the chunk of x * A'_d
  template <int CNT>
  __device__ __forceinline__ void s1_group(int c, float x, int g0, float4& zacc) const {
    float4 a[CNT];
    float xv[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const int d = bperm_i<BPI>(bp_grp, (g0 + j) * NPI, c);
      xv[j] = bperm_f<BPI>(bp_grp, (g0 + j) * NPI, x);
      a[j] = gather4<LPN>(Ap, d, sub);
    }
#pragma unroll
    for (int j = 0; j < CNT; ++j) zacc = fma4(xv[j], a[j], zacc);
  }
  __device__ __forceinline__ void sweep1(int c, float x, int nchunk, float4& zacc) const {
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (g0 * NPI < nchunk) s1_group<GRP>(c, x, g0, zacc);  // wave-uniform
    }
  }

  // sweep-2 pieces of one group of GRP gather instructions of which CNT carry entries
  template <int CNT>
  __device__ __forceinline__ void s2_gather(int c, int g0, const float4& z, float4 (&vv)[LPN],
                                            float& rmine) const {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const int d = bperm_i<BPI>(bp_grp, (g0 + j) * NPI, c);
      vv[g0 + j] = gather4<LPN>(Vp, d, sub);
    }
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const float dot = group_sum<LPN>(dot4(z, vv[g0 + j]));
      if (sub == g0 + j) rmine = dot;
    }
#pragma unroll
    for (int j = CNT; j < GRP; ++j) vv[g0 + j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  template <int CNT, bool PERM = false>
  __device__ __forceinline__ void s2_back(float cc, int g0, const float4 (&vv)[LPN],
                                          float4& gz) const {
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const int q = g0 + j;                                  // the lane of the group that owns gather q's weight
      const float cb = bperm_f<BPI>(bp_row, PERM ? (q < 4 ? q : 11 - q) : q, cc);
      gz = fma4(cb, vv[g0 + j], gz);
    }
  }

  // The V' gathers of one chunk, ISSUED and not waited for (short rows, ROW_V_AHEAD: a row's first chunk asks for
  // its V' rows before sweep 1 starts -- they depend on the column indices only, not on z -- so they are in
  // flight beside the A' gathers and sweep 2 of that chunk starts on loaded registers; a later chunk asks for all
  // of its groups at once instead of group by group)
  __device__ __forceinline__ void s2_load(int c, int nchunk, float4 (&vv)[LPN]) const {
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (g0 * NPI < nchunk) {
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const int d = bperm_i<BPI>(bp_grp, (g0 + j) * NPI, c);
          vv[g0 + j] = gather4<LPN>(Vp, d, sub);
        }
      } else {
#pragma unroll
        for (int j = 0; j < GRP; ++j) vv[g0 + j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // sweep 2 of a chunk whose V' rows were asked for by s2_load
  __device__ __forceinline__ void sweep2_loaded(int c, float x, int nchunk, const float4& z, float4& gz, float& ll,
                                                double& nnf, const float4 (&vv)[LPN]) const {
    if constexpr (LPN == 8 && ROW_DOT_BUTTERFLY) {
      // Eight dot products over eight lanes as ONE transpose-reduce instead of eight three-step folds + eight
      // selects: at every step a lane keeps half of its values and hands the other half to its partner, so it
      // ends with the ONE total it owns (7 exchanges + 14 selects against 24 + 8; the row pass is as close
      // to its VALU issue rate as to its request rate: DESIGN section 4 "Round 5 (g)").  Partners: lane ^ 1, lane ^ 2
      // (quad_perm), then 7 - lane (row_half_mirror: the upper quad therefore keeps by the complement of its lane
      // bits, eff), so lane `sub` ends with the total of gather own(sub) = sub < 4 ? sub : 11 - sub.
      float d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = dot4p(z, vv[j]);
      const int eff = sub < 4 ? sub : 7 - sub;
      const bool m0 = (eff & 1) != 0, m1 = (eff & 2) != 0, hi = sub >= 4;
      float r4[4], q2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = m0 ? d[2 * i + 1] : d[2 * i], send = m0 ? d[2 * i] : d[2 * i + 1];
        r4[i] = dpp_add_to<0xB1>(keep, send);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = m1 ? r4[2 * i + 1] : r4[2 * i], send = m1 ? r4[2 * i] : r4[2 * i + 1];
        q2[i] = dpp_add_to<0x4E>(keep, send);
      }
      const float keep = hi ? q2[1] : q2[0], send = hi ? q2[0] : q2[1];
      const float rmine = dpp_add_to<0x141>(keep, send);
      s2_cells<true>(c, x, nchunk, rmine, vv, gz, ll, nnf);
      return;
    }
    float rmine = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (g0 * NPI < nchunk) {
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const float dot = group_sum<LPN>(ROW_DOT_PK ? dot4p(z, vv[g0 + j]) : dot4(z, vv[g0 + j]));
          if (sub == g0 + j) rmine = dot;
        }
      }
    }
    s2_cells(c, x, nchunk, rmine, vv, gz, ll, nnf);
  }

  // rates, log-likelihood and gz partial over one chunk
  __device__ __forceinline__ void sweep2(int c, float x, int nchunk, const float4& z,
                                         float4& gz, float& ll, double& nnf) const {
    float4 vv[LPN];
    float rmine = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (g0 * NPI < nchunk) s2_gather<GRP>(c, g0, z, vv, rmine);
      else s2_gather<0>(c, g0, z, vv, rmine);
    }
    s2_cells(c, x, nchunk, rmine, vv, gz, ll, nnf);
  }
  // the per-cell part of sweep 2 (one entry per lane) and the gz partial
  // PERM: the lane owns gather own(sub) = sub < 4 ? sub : 11 - sub (what the transpose-reduce of sweep2_loaded leaves)
  template <bool PERM = false>
  __device__ __forceinline__ void s2_cells(int c, float x, int nchunk, float rmine, const float4 (&vv)[LPN],
                                           float4& gz, float& ll, double& nnf) const {
    // one entry per lane: lane (grp,sub) owns slot own*NPI+grp
    const int own = PERM ? (sub < 4 ? sub : 11 - sub) : sub;
    const int slot = own * NPI + grp;
    const float xs = __shfl(x, slot);
    const int cs = __shfl(c, slot);
    float cc = 0.f;
    if (slot < nchunk && xs > 0.f) {
      if (LIK == 2 || LIK == 4 || (LIK == 3 && ctype[cs])) {
        // Bernoulli(logits = f(<z,V'>) + phi) (bernoulli.py:147-155): stored-cell part x*logit;
        // LIK 4: f = exp - 1 (saturating like the Poisson form)
        float ey = 1.f;
        const float fy = LIK == 4 ? expm1_dec(fminf(rmine, kYSat), ey) : rmine;
        const float lg = fy + (LDSPHI ? lds_dyn()[cs] : phi[cs]);
        if (lg > -INFINITY && lg < INFINITY) {
          ll = fmaf(xs, lg, ll);
          cc = LIK == 4 ? xs * ey : xs;              // d(x*logit)/d<z,V'>
        } else {
          nnf += 1.0;
        }
      } else {
        // linear decoder: r = <z,V'> + phi; log_transform: r = exp(<z,V'>) - 1 + phi
        float ey = 1.f;
        const float fy = LIK == 1 ? expm1_dec(fminf(rmine, kYSat), ey) : rmine;
        const float r = fy + (LDSPHI ? lds_dyn()[cs] : phi[cs]);
        if (r > 0.f && r < INFINITY) {
          ll = fmaf(xs, logf(r), ll);
          cc = xs * ey * __builtin_amdgcn_rcpf(r);   // d(x log r)/d<z,V'>
        } else {
          // the replacement rule (poisson.py:606-616) swaps the WHOLE log-pmf of this
          // cell for min-10 (spmf_nonfinite_patch; its lgamma(x+1) is taken back out of
          // the pre-summed constant there).  Its share of the closed-form / dense "-sum
          // over all cells of r" is cancelled with weight +1 here against the -1 every
          // cell gets there.  (No lgammaf in this kernel: inlined into the hot loop it
          // cost 20 % of the pass, 1.88 -> 2.25 ms on C3.)
          nnf += 1.0;
          cc = ey;
        }
      }
    }
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (g0 * NPI < nchunk) s2_back<GRP, PERM>(cc, g0, vv, gz);
    }
  }
};

}  // namespace

// BT: threads per workgroup.  256: phi from global memory.  512 / 1024: phi staged in LDS
// (4*D bytes of dynamic LDS: two workgroups per CU up to D = 20 480, one up to 40 960),
// four waves per SIMD either way.
template <int KP, int LIK, int BT = 256, int PACKED = 0>
__global__ __launch_bounds__(BT, BT == 256 ? ROW_WAVES_PER_SIMD : 4) void row_pass_kernel(
    int64_t B, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ row_scale,
    const float* __restrict__ Ap, const float* __restrict__ Vp, const float* __restrict__ phi,
    const double* __restrict__ dprep, float* __restrict__ z, float* __restrict__ gzs,
    double* __restrict__ dacc, int mode, const float* __restrict__ gzd,
    const uint8_t* __restrict__ ctype, int Dcols, int64_t dacc_stride,
    const uint32_t* __restrict__ ent, double* __restrict__ det_slots, int64_t det_stride, int dyn_tail) {
  if (gridDim.y > 1) {   // S draws per launch: tables, outputs and accumulators of draw blockIdx.y
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
  constexpr int LPN = KP / 4;
  constexpr bool LDSPHI = BT != 256;
  if (LDSPHI && !encode_only) {
    float* pl = lds_dyn();
    for (int i = threadIdx.x; i < Dcols; i += BT) pl[i] = phi[i];
    __syncthreads();
  }
  RowCtx<KP, LIK, LDSPHI> cx;
  cx.Ap = gtable(Ap, Dcols, KP);
  cx.Vp = gtable(Vp, Dcols, KP);
  cx.phi = phi;
  cx.ctype = ctype;
  cx.lane = threadIdx.x & 63;
  cx.sub = cx.lane % LPN;
  cx.grp = cx.lane / LPN;
  cx.bp_grp = cx.grp * 4;
  cx.bp_row = cx.grp * LPN * 4;
  const int lane = cx.lane, sub = cx.sub, grp = cx.grp;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

  float4 veta4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!encode_only && (mode != 2 || LIK == 3 || !gzd))
    veta4 = make_float4((float)prep_sum(dprep, KP, sub * 4 + 0), (float)prep_sum(dprep, KP, sub * 4 + 1),
                        (float)prep_sum(dprep, KP, sub * 4 + 2), (float)prep_sum(dprep, KP, sub * 4 + 3));
  double ll_acc = 0.0, zsq_acc = 0.0, nnf_acc = 0.0;
  float4 zsum = make_float4(0.f, 0.f, 0.f, 0.f);

  // Which rows a wave takes.  Rows [0, B_static) are dealt out with a fixed stride (row w, w + nwaves, ...:
  // every wave the same count); the LAST eighth of a wave's share is not fixed: rows [B_static, B) are handed
  // out one at a time from sixteen counters (wave w draws from counter w % 16, which owns a sixteenth of that
  // range), so a wave that fell behind -- longer rows, a slower corner of the chip -- takes fewer of them and
  // the launch no longer ends with most waves waiting for the unluckiest one (the stored entries of a wave's
  // fixed share scatter by sqrt(rows per wave): +6.7 % at the largest of 4096 waves on the 8-GPU shard of
  // C3, +2.4 % on the whole matrix).  The counters are the spare fp64 slot [5] of the sixteen replicas of the
  // scalar block (common.h: zeroed by the prep launch with everything else in it, never read as a sum).
  // Off (B_static = B) in the deterministic mode -- which workgroup sums which rows must not depend on timing
  // there -- and for launches that share a step's scalar block with another row launch (dyn_tail = 0).
  int64_t B_static = B;
  int64_t dyn_lo = B, dyn_hi = B;
  unsigned int* dyn_ctr = nullptr;
  if (dyn_tail && !det_slots && !encode_only) {
    const int64_t per_wave = B / nwaves;
    if (per_wave >= 8) {
      int64_t fixed = per_wave - (per_wave * ROW_DYN_NUM) / ROW_DYN_DEN;
      if (fixed < 2) fixed = 2;
      B_static = fixed * nwaves;
      const int r = (int)(wave & (kDaccRep - 1));
      const int64_t each = (B - B_static + kDaccRep - 1) / kDaccRep;
      dyn_lo = B_static + r * each;
      dyn_hi = dyn_lo + each < B ? dyn_lo + each : B;
      if (dyn_lo > B) dyn_lo = B;
      dyn_ctr = reinterpret_cast<unsigned int*>(dacc + (size_t)r * (kDaccHead + KP) + 5);
    }
  }
  // the row after `cur` in this wave's sequence when it is a fixed one, else -1 (ask the counter)
  auto fixed_next = [&](int64_t cur) -> int64_t {
    if (cur >= B) return B;
    if (cur + nwaves < B_static) return cur + nwaves;
    return dyn_ctr ? (int64_t)-1 : B;
  };
  auto take_dynamic = [&]() -> int64_t {          // (wave-uniform; the wait for the atomic is here)
    unsigned int k = 0;
    if (lane == 0) k = atomicAdd(dyn_ctr, 1u);
    k = (unsigned int)__builtin_amdgcn_readfirstlane((int)k);
    const int64_t row = dyn_lo + (int64_t)k;
    return row < dyn_hi ? row : B;
  };

  // Software pipeline across rows: the pointers of the row after next and the first
  // two 64-entry chunks of the next row are in flight while a row is processed,
  // so the row_ptr -> col/val dependent latency is off the per-row critical path.
  int start = 0, end = 0, pc0 = 0, pc1 = 0, nstart = 0, nend = 0;
  float xi = 1.f, px0 = 0.f, px1 = 0.f, nxi = 1.f;
  int64_t b = wave < B_static ? wave : B;
  int64_t bn = fixed_next(b);
  if (bn < 0) bn = take_dynamic();
  int64_t bnn = fixed_next(bn);
  if (bnn < 0) bnn = take_dynamic();
  if (b < B) {
    start = row_ptr[b];
    end = row_ptr[b + 1];
    xi = row_scale ? row_scale[b] : 1.f;
    const int f0 = min(end - start, 64);
    const int i0 = start + lane, i1 = start + f0 + lane;
    load_entry<PACKED, true>(col, val, ent, i0, lane < f0, pc0, px0);
    load_entry<PACKED, true>(col, val, ent, i1, i1 < end, pc1, px1);
  }
  if (bn < B) {
    nstart = row_ptr[bn];
    nend = row_ptr[bn + 1];
    nxi = row_scale ? row_scale[bn] : 1.f;
  }
  while (b < B) {
    const int n = end - start;
    // prefetch: chunks of the next row, pointers of the one after
    int qc0 = 0, qc1 = 0, nnstart = 0, nnend = 0;
    float qx0 = 0.f, qx1 = 0.f, nnxi = 1.f;
    if (bn < B) {
      const int f0 = min(nend - nstart, 64);
      const int j0 = nstart + lane, j1 = nstart + f0 + lane;
      load_entry<PACKED, true>(col, val, ent, j0, lane < f0, qc0, qx0);
      load_entry<PACKED, true>(col, val, ent, j1, j1 < nend, qc1, qx1);
    }
    if (bnn < B) {
      nnstart = row_ptr[bnn];
      nnend = row_ptr[bnn + 1];
      nnxi = row_scale ? row_scale[bnn] : 1.f;
    }
    // the row three ahead: a fixed one, or a counter's next (asked for now, looked at when this row is done)
    int64_t bnnn = fixed_next(bnn);
    unsigned int dyn_k = 0;
    const bool dyn_ask = bnnn < 0;                  // wave-uniform
    if (dyn_ask && lane == 0) dyn_k = atomicAdd(dyn_ctr, 1u);
    float4 zacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gz = make_float4(0.f, 0.f, 0.f, 0.f);
    float llrow = 0.f;
    if (n <= 128) {
      // ---- short row: col/val stay in registers for both sweeps ----------
      int c0, c1;
      float x0, x1;
      unpack_entry<PACKED>(pc0, px0, c0, x0);
      unpack_entry<PACKED>(pc1, px1, c1, x1);
      const int n0 = min(n, 64), n1 = n - 64;
      // ROW_V_AHEAD (K <= 32; at K = 64 a chunk's V' rows are 64 registers): 1 = the first chunk's V' gathers
      // are issued before sweep 1, 3 = between sweep 1 and its cross-group sum, 2 = not ahead, but a chunk's
      // sweep 2 issues all of its gathers before its first dot product; 0 = group by group (the round-3 form)
      constexpr int VAHEAD = KP <= 32 ? ROW_V_AHEAD : 0;
      float4 vv0[VAHEAD ? LPN : 1];
      if constexpr (VAHEAD == 1) {
        if (!encode_only) cx.s2_load(c0, n0, vv0);
      }
      if (mode != 2) {
        if (PACKED == 2) {              // sweep 1 reads the encoder's values, not the counts
          cx.sweep1(c0, px0, n0, zacc);
          if (n1 > 0) cx.sweep1(c1, px1, n1, zacc);
        } else {
          cx.sweep1(c0, x0, n0, zacc);
          if (n1 > 0) cx.sweep1(c1, x1, n1, zacc);
        }
        if constexpr (VAHEAD == 3) {
          if (!encode_only) cx.s2_load(c0, n0, vv0);
        }
        zacc = across_groups_sum4<LPN>(zacc);
        zacc.x *= xi; zacc.y *= xi; zacc.z *= xi; zacc.w *= xi;
        if (grp == 0) reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = zacc;
      } else {
        zacc = gather4<LPN>(z, (int)b, sub);
      }
      if (!encode_only) {
      if constexpr (VAHEAD != 0) {
        if constexpr (VAHEAD == 2) cx.s2_load(c0, n0, vv0);
        if constexpr (VAHEAD == 3) {
          if (mode == 2) cx.s2_load(c0, n0, vv0);
        }
        cx.sweep2_loaded(c0, x0, n0, zacc, gz, llrow, nnf_acc, vv0);
        if (n1 > 0) {
          float4 vv1[LPN];
          cx.s2_load(c1, n1, vv1);
          cx.sweep2_loaded(c1, x1, n1, zacc, gz, llrow, nnf_acc, vv1);
        }
      } else {
        cx.sweep2(c0, x0, n0, zacc, gz, llrow, nnf_acc);
        if (n1 > 0) cx.sweep2(c1, x1, n1, zacc, gz, llrow, nnf_acc);
      }
      }
    } else {
      // ---- long row: stream the row twice (second read is L2 served) -----
      // Chunks 0 and 1 came with the row's prefetch; chunk i+2 is fetched while chunk i is
      // processed, so no chunk waits for its own col/val (14 chunks a row on C4).
      if (mode != 2) {
        int ra = pc0, ra1 = pc1;
        float rb = px0, rb1 = px1;
        for (int base = start; base < end; base += 64) {
          const int i2 = base + 128 + lane;
          int ra2;
          float rb2;
          load_entry<PACKED, false>(col, val, ent, i2, i2 < end, ra2, rb2);
          int c;
          float x;
          unpack_entry<PACKED, 1>(ra, rb, c, x);
          cx.sweep1(c, x, min(64, end - base), zacc);
          ra = ra1; rb = rb1; ra1 = ra2; rb1 = rb2;
        }
        zacc = across_groups_sum4<LPN>(zacc);
        zacc.x *= xi; zacc.y *= xi; zacc.z *= xi; zacc.w *= xi;
        if (grp == 0) reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = zacc;
      } else {
        zacc = gather4<LPN>(z, (int)b, sub);
      }
      if (!encode_only) {
        int ra = pc0, ra1 = pc1;
        float rb = px0, rb1 = px1;
        for (int base = start; base < end; base += 64) {
          const int i2 = base + 128 + lane;
          int ra2;
          float rb2;
          load_entry<PACKED, false>(col, val, ent, i2, i2 < end, ra2, rb2);
          int c;
          float x;
          unpack_entry<PACKED>(ra, rb, c, x);
          cx.sweep2(c, x, min(64, end - base), zacc, gz, llrow, nnf_acc);
          ra = ra1; rb = rb1; ra1 = ra2; rb1 = rb2;
        }
      }
    }
    // rotate the pipeline registers
    const float xi_cur = xi;
    const int64_t b_cur = b;
    start = nstart; end = nend; xi = nxi;
    pc0 = qc0; px0 = qx0; pc1 = qc1; px1 = qx1;
    nstart = nnstart; nend = nnend; nxi = nnxi;
    if (dyn_ask) {
      const int64_t row = dyn_lo + (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)dyn_k);
      bnnn = row < dyn_hi ? row : B;
    }
    b = bn; bn = bnn; bnn = bnnn;
    if (encode_only) continue;
    gz = across_groups_sum4<LPN>(gz);
    if (grp == 0) {
      zsq_acc += (double)dot4(zacc, zacc);
      zsum = add4(zsum, zacc);
      // minus the derivative of sum_d r_bd over ALL columns: closed form veta
      // (linear decoder) or the dense exp term of this row (log_transform)
      // (mixed: closed-form Poisson-column part veta PLUS the dense Bernoulli-column term)
      float4 dn = (mode == 2 && gzd) ? gather4<LPN>(gzd, (int)b_cur, sub) : veta4;
      if (LIK == 3 && mode == 2 && gzd) dn = add4(dn, veta4);
      // mode 3 (both sweeps, dense row term subtracted LATER by the dense kernel's epilogue,
      // gzs_b -= xi_b * sum_d E_bd V'_d): only the closed-form Poisson-column part is known here
      if (mode == 3 && LIK != 3) dn = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = xi_cur * (gz.x - dn.x - zacc.x);
      o.y = xi_cur * (gz.y - dn.y - zacc.y);
      o.z = xi_cur * (gz.z - dn.z - zacc.z);
      o.w = xi_cur * (gz.w - dn.w - zacc.w);
      reinterpret_cast<float4*>(gzs)[(size_t)b_cur * LPN + sub] = o;
    }
    ll_acc += (double)llrow;
  }
  if (encode_only) return;

  // ---- block reduction, one set of fp64 atomics per block ---------------
  __shared__ double red[16];
  __shared__ double zred_static[BT == 256 ? 4 : 1][KP];
  // with phi in LDS the per-wave z sums reuse that region (phi is no longer needed once
  // every wave has left the row loop): keeps two 80 KB workgroups inside 160 KB
  if (LDSPHI) __syncthreads();
  double (*zred)[KP] = LDSPHI ? reinterpret_cast<double (*)[KP]>(lds_dyn()) : zred_static;
  const int wid = threadIdx.x >> 6;
  if (grp == 0) {
    zred[wid][sub * 4 + 0] = (double)zsum.x;
    zred[wid][sub * 4 + 1] = (double)zsum.y;
    zred[wid][sub * 4 + 2] = (double)zsum.z;
    zred[wid][sub * 4 + 3] = (double)zsum.w;
  }
  const double ll_b = block_sum(ll_acc, red);
  const double zq_b = block_sum(zsq_acc, red);
  const double nf_b = block_sum(nnf_acc, red);
  if (det_slots) {
    // deterministic mode: this workgroup's sums in its own slot; the pack block of the column pass adds
    // the slots up in workgroup order
    double* sl = det_slots + (size_t)blockIdx.y * det_stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) sl[0] = (double)gridDim.x;
    sl += kDetMeta + (size_t)blockIdx.x * (kDaccHead + KP);
    if (threadIdx.x == 0) {
      sl[0] = ll_b;
      sl[1] = zq_b;
      sl[2] = nf_b;
      sl[3] = sl[4] = sl[5] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x < KP) {
      double t = 0.0;
      const int nw = blockDim.x >> 6;
      for (int i = 0; i < nw; ++i) t += zred[i][threadIdx.x];
      sl[kDaccHead + threadIdx.x] = t;
    }
    return;
  }
  dacc += (size_t)(blockIdx.x % kDaccRep) * (kDaccHead + KP);
  if (threadIdx.x == 0) {
    atomicAdd(&dacc[0], ll_b);
    atomicAdd(&dacc[1], zq_b);
    if (nf_b != 0.0) atomicAdd(&dacc[2], nf_b);
  }
  __syncthreads();
  if (threadIdx.x < KP) {
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += zred[i][threadIdx.x];
    atomicAdd(&dacc[kDaccHead + threadIdx.x], t);
  }
}

// false: the device does not grant the dynamic LDS this form needs (the caller then launches
// the 256-thread form, which reads phi from global memory)
template <int KP, int LIK, int BT, int PACKED>
static bool launch_row_lds_t(const RowArgs& a, hipStream_t st) {
  // (the encode-only sweep reads no phi: it takes this launch shape, not the LDS)
  const size_t lds = a.mode == 1 ? 0 : std::max((size_t)a.D * 4, (size_t)(BT / 64) * KP * sizeof(double));
  // opt in to more than 64 KB of dynamic LDS.  The attribute is per DEVICE (and this is one
  // instantiation per process), so the grant is remembered per device ordinal.
  constexpr int kMaxDev = 64;
  static size_t granted[kMaxDev] = {};
  if (lds > 64 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
      (void)hipGetLastError();
      return false;
    }
    if (lds > __atomic_load_n(&granted[dev], __ATOMIC_RELAXED)) {
      if (hipFuncSetAttribute((const void*)row_pass_kernel<KP, LIK, BT, PACKED>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
      __atomic_store_n(&granted[dev], lds, __ATOMIC_RELAXED);
    }
  }
  const int64_t want = (a.B + (BT / 64) - 1) / (BT / 64);
  // exactly the resident set (two 512-thread or one 1024-thread workgroup per CU), rows
  // grid-strided: every workgroup stages phi once.  Measured on C3 / a 125k-row shard:
  // 1.587 / 0.211 ms, against 1.596 / 0.227 with twice and 1.621 / 0.247 with eight times
  // as many workgroups (each re-stages 4*D bytes before its first gather).
#ifndef ROW_LDS_CAP
#define ROW_LDS_CAP 1024
#endif
  const int64_t cap = (int64_t)ROW_LDS_CAP * 256 / BT;
  const int nb = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  hipLaunchKernelGGL((row_pass_kernel<KP, LIK, BT, PACKED>), dim3(nb, a.S > 1 ? a.S : 1), dim3(BT), lds, st,
                     a.B, a.row_ptr, a.col, a.val, a.row_scale, a.Ap, a.Vp, a.phi, a.dprep, a.z,
                     a.gzs, a.dacc, a.mode, a.gzd, a.ctype, a.D, a.dacc_stride, a.ent, a.det_slots, a.det_stride, a.dyn_tail);
  return true;
}

// packed entry stream (spmf_counts.ent) when the batch carries one
template <int KP, int LIK, int BT>
static bool launch_row_lds(const RowArgs& a, hipStream_t st) {
  if constexpr (LIK == 1) {
    // fused pass of the exp decoder: packed word + the g(x) stream (RowArgs.dual)
    if (a.dual) return a.ent ? launch_row_lds_t<KP, LIK, BT, 2>(a, st) : false;
  }
  return a.ent ? launch_row_lds_t<KP, LIK, BT, 1>(a, st) : launch_row_lds_t<KP, LIK, BT, 0>(a, st);
}

#ifndef ROW_LDS_PHI
#define ROW_LDS_PHI 1
#endif

template <int KP>
static bool launch_row_t(const RowArgs& a, hipStream_t st) {
  int64_t want = (a.B + 3) / 4;  // 4 waves (rows in flight) per 256-thread block
  int nb = (int)(want < 1 ? 1 : (want > ROW_MAX_BLOCKS ? ROW_MAX_BLOCKS : want));
  // phi from LDS: the sweep-2 forms of the Poisson likelihoods at the K of the named
  // configs, when 4*D bytes fit (and the batch is big enough to fill the wider blocks)
  if constexpr (KP >= 16) {
    if (ROW_LDS_PHI && a.mode == 1 && a.B >= 4096) {
      // sweep 1 alone (z from the encoder side) does not depend on the likelihood: the
      // resident-set launch of the 512-thread form, without the phi copy.  C4: 8.6 -> 6.4 ms
      // against the 256-thread grid (profiles/r03_sparse_pass_attempts.txt e25)
      if (launch_row_lds<KP, 0, 512>(a, st)) return true;
    }
    if (ROW_LDS_PHI && a.mode != 1 && a.logt >= 0 && a.logt <= 3 && a.B >= 4096) {
      // (likelihood codes 2 and 3, Bernoulli and mixed, since the end of round 3: their
      //  stored-cell sweep read phi one entry per lane from global memory, C5 0.61 ms)
      const size_t need = (size_t)a.D * 4;
      bool done = false;
      if (need <= 80 * 1024) {
        done = a.logt == 0   ? launch_row_lds<KP, 0, 512>(a, st)
               : a.logt == 1 ? launch_row_lds<KP, 1, 512>(a, st)
               : a.logt == 2 ? launch_row_lds<KP, 2, 512>(a, st)
                             : launch_row_lds<KP, 3, 512>(a, st);
      } else if (need <= 160 * 1024 - 1024) {
        done = a.logt == 0   ? launch_row_lds<KP, 0, 1024>(a, st)
               : a.logt == 1 ? launch_row_lds<KP, 1, 1024>(a, st)
               : a.logt == 2 ? launch_row_lds<KP, 2, 1024>(a, st)
                             : launch_row_lds<KP, 3, 1024>(a, st);
      }
      if (done) return true;
    }
  }
  if (a.dual) return false;      // only the LDS-phi shapes have the two-stream form
#define SPMF_ROW_LAUNCH(L_)                                                                    \
  hipLaunchKernelGGL((row_pass_kernel<KP, L_>), dim3(nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, \
                     a.B, a.row_ptr, a.col, a.val, a.row_scale, a.Ap, a.Vp, a.phi, a.dprep, a.z, \
                     a.gzs, a.dacc, a.mode, a.gzd, a.ctype, a.D, a.dacc_stride, a.ent, a.det_slots, a.det_stride, a.dyn_tail)
  if (a.logt == 4) SPMF_ROW_LAUNCH(4);
  else if (a.logt == 3) SPMF_ROW_LAUNCH(3);
  else if (a.logt == 2) SPMF_ROW_LAUNCH(2);
  else if (a.logt == 1) SPMF_ROW_LAUNCH(1);
  else SPMF_ROW_LAUNCH(0);
#undef SPMF_ROW_LAUNCH
  return true;
}

bool launch_row_pass(int KP, const RowArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: return launch_row_t<4>(a, st);
    case 8: return launch_row_t<8>(a, st);
    case 16: return launch_row_t<16>(a, st);
    case 32: return launch_row_t<32>(a, st);
    case 64: return launch_row_t<64>(a, st);
    case 128: case 256: return launch_row_widek(KP, a, st);   // widek.hip: a factor row is the whole wave
    default: return false;
  }
}

}  // namespace spmf
