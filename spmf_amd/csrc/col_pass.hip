// col_pass.hip -- column-side (transposed) accumulation of the factor
// gradients over the panel-CSC copy of the batch (gfx950, wave64).
//
// For every feature column d (SURVEY 8a gradient formulas, derived from
// poisson.py:156-184,582-701):
//   r_bd     = <z_b, V'_d> + phi_d                 recomputed, V'_d in registers
//   gV'_d   += (x_bd/r_bd) z_b                     -> d/dv   after the chain
//   gphi_d  += x_bd/r_bd                           -> d/dw, d/ds
//   gA'_d   += g(x_bd) * (xi_b gz_b)               -> d/du, d/ds
// (log_transform: r = exp(<z,V'>) - 1 + phi and the gV' weight is x*E/r.)
// The minus-one part of (x/r - 1) over ALL rows (stored or not) is closed
// form (sum_b z_b and B) and is applied by the finish kernel.
//
// Work items.  The host cuts every non-empty (row panel, column) list of the
// panel-CSC into segments of at most SEG entries and sorts the items of a
// panel by length (spmf_amd/sparse.py): item = {start, len, column}.  That
// (a) skips empty lists, (b) bounds the work of one item, so a hot column of
// a skewed matrix (scRNA-seq genes, vocabulary heads) is spread over many
// waves instead of serialising one, and (c) gives the lane groups of a wave
// items of similar length.
//
// Mapping: a gathered row (z_b or xi_b*gz_b) is KP floats = LPN=KP/4 lanes x
// float4, so a wave holds NG=64/LPN lane groups; EACH GROUP OWNS ONE ITEM and
// keeps that item's gV'/gA'/gphi slices in registers, so there is no
// cross-lane reduction besides the DPP fold of the dot product.  A group
// streams its item LPN entries per fetch (one per lane, a 4*LPN-B contiguous
// read), two fetches ahead of use, and broadcasts an entry inside the group
// with ds_bpermute; gathers are issued four entries (eight 16-B loads per
// lane) at a time.
//
// L2 residency is what makes the 2 x 4*KP-B-per-entry gathers affordable
// (measured: served from Infinity Cache instead, the pass runs at ~8.6 TB/s
// of gather traffic = 3 ms on the C3 shape).  Panels are `panel_rows`
// consecutive rows (2*panel_rows*KP*4 B of z / xi*gz, 2 MB at the default)
// and workgroup ids are ordered panel-major with p = 8t + blockIdx%8, so with
// the observed round-robin workgroup->XCD placement each XCD's resident
// workgroups share one panel.  Speed heuristic only: results do not depend
// on placement.
//
// A wave leaves through an LDS transpose and 8 float-atomic wave instructions,
// each covering two whole 128-B gradient rows (the shape MI355X runs atomics
// at full rate).  Atomic traffic is n_items * (2KP+1)*4 B (0.63 GB on C3).
// Float atomics make the low-order bits of the gradient run-to-run dependent;
// parity tolerance is 1e-5 relative (north_star).
#include "common.h"
#include "kernels.h"

namespace spmf {

#ifndef COL_WIDE
#define COL_WIDE 1
#endif
#ifndef COL_WIDE_WAVES
#define COL_WIDE_WAVES 4
#endif
#ifndef COL_GRP
#define COL_GRP 4
#endif

// The extra first block of a launch (pack_dacc != null): fold the kDaccRep replicas of the row
// pass's fp64 scalars into the accumulator tail as (hi, lo) float pairs, so that ONE fp32
// all-reduce finishes the step.  The row pass is a previous launch: its sums are complete.
template <int KP>
__device__ __forceinline__ void pack_block(const double* __restrict__ dacc, float* __restrict__ tail) {
  const int i = threadIdx.x;
  if (i < kDaccHead + KP) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < kDaccRep; ++r) v += dacc[(size_t)r * (kDaccHead + KP) + i];
    const float hi = (float)v;
    tail[2 * i] = hi;
    tail[2 * i + 1] = (float)(v - (double)hi);
  }
}

// LIK: 0 Poisson / linear, 1 Poisson / log_transform, 2 Bernoulli(logits) / linear
template <int KP, int LIK>
__global__ __launch_bounds__(256) void col_pass_kernel(
    int D, int n_panels, int row_base, int blocks_per_panel,
    const int32_t* __restrict__ item_ptr, const int4* __restrict__ items,
    const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ pc_gval, const float* __restrict__ Vp,
    const float* __restrict__ phi, const float* __restrict__ z, const float* __restrict__ gzs,
    float* __restrict__ gAp, float* __restrict__ gVp, float* __restrict__ gphi,
    const uint8_t* __restrict__ ctype, const int32_t* __restrict__ item_mid, int half_sel,
    int64_t Brows, int64_t acc_stride, const double* __restrict__ pack_dacc,
    float* __restrict__ pack_tail, int64_t dacc_stride) {
  if (pack_dacc && blockIdx.x == 0) {
    pack_block<KP>(pack_dacc + (size_t)blockIdx.y * dacc_stride, pack_tail + (size_t)blockIdx.y * acc_stride);
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
  constexpr int LPN = KP / 4;
  constexpr int NG = 64 / LPN;                        // items per wave
  constexpr int GRP = LPN < COL_GRP ? LPN : COL_GRP;  // entries gathered back to back
  __shared__ __attribute__((aligned(16))) float stage[4][NG][2 * KP];
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int wid = threadIdx.x >> 6;
  // block id -> (panel, block of 4*NG items); blockIdx % 8 = panel residue
  // (batches of fewer than 8 panels use a flat mapping: the residue mapping
  // would leave the XCDs of the missing residues with empty workgroups only)
  const int64_t L = (int64_t)blockIdx.x - (pack_dacc ? 1 : 0);   // block 0 is the pack block when asked
  int p, ib;
  if (n_panels < 8) {
    p = (int)(L / blocks_per_panel);
    ib = (int)(L % blocks_per_panel);
  } else {
    const int x = (int)(L & 7);
    const int64_t q = L >> 3;
    p = 8 * (int)(q / blocks_per_panel) + x;
    ib = (int)(q % blocks_per_panel);
  }
  if (p >= n_panels) return;                          // block-uniform
  // item range of this launch: the whole panel, or one column half of it (the host
  // sorts a panel's items by half first: multi-GPU overlap of the all-reduce)
  const int ilo = half_sel == 2 ? item_mid[p] : item_ptr[p];
  const int ihi = half_sel == 1 ? item_mid[p] : item_ptr[p + 1];
  const int i0 = ilo + ib * 4 * NG;
  if (i0 >= ihi) return;                              // block-uniform
  const int it = i0 + wid * NG + grp;
  const bool ok = it < ihi;
  int cur = 0, end = 0, d = 0;
  if (ok) {
    const int4 im = items[it];
    cur = im.x;
    end = im.x + im.y;
    d = im.z;
  }
  const float4 vp = ok ? gather4<LPN>(Vp, d, sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float ph = ok ? phi[d] : 1.f;
  const bool bern = LIK == 2 || LIK == 4 || (LIK == 3 && ok && ctype[d]);   // item's column is Bernoulli
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;

  auto fetch = [&](int& rr_, float& xx_, float& gx_, int& cnt_) {
    cnt_ = min(LPN, end - cur);                       // 0 once the item is exhausted
    const int e = cur + sub;
    rr_ = sub < cnt_ ? pc_row[e] - row_base : 0;
    xx_ = sub < cnt_ ? pc_val[e] : 0.f;
    gx_ = ((LIK == 1 || LIK == 4) && sub < cnt_) ? pc_gval[e] : 0.f;  // g(x) = log(x/eta+1), data side
    cur += cnt_;
  };

  int rr0, cnt0, rr1, cnt1;
  float xx0, xx1, gx0, gx1;
  fetch(rr0, xx0, gx0, cnt0);
  fetch(rr1, xx1, gx1, cnt1);
  while (__any(cnt0 > 0)) {
    int rr2, cnt2;
    float xx2, gx2;
    fetch(rr2, xx2, gx2, cnt2);                       // two fetches ahead of use
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (__any(cnt0 > g0)) {                         // wave-uniform
        float4 zz[GRP], gg[GRP];
        float xv[GRP], gv[GRP];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const int src = grp * LPN + g0 + j;
          const int b = __shfl(rr0, src);
          xv[j] = __shfl(xx0, src);
          gv[j] = (LIK == 1 || LIK == 4) ? __shfl(gx0, src) : xv[j];
          zz[j] = gather4<LPN>(z, b, sub);
          gg[j] = gather4<LPN>(gzs, b, sub);
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          if (bern) {
            // Bernoulli: d(x*logit)/dV' = x z_b, d/dphi = x: no rate needed
            // (LIK 4, logit = exp(y) - 1 + phi: the V' weight is x * exp(y))
            float wv = xv[j];
            if (LIK == 4) wv *= expf(fminf(group_sum<LPN>(dot4(zz[j], vp)), kYSat));
            gV = fma4(wv, zz[j], gV);
            gA = fma4(gv[j], gg[j], gA);
            gph += xv[j];
          } else {
            const float y = group_sum<LPN>(dot4(zz[j], vp));
            float ey = 1.f;                                            // saturating: common.h kYSat
            const float r = (LIK == 1 ? expm1_dec(fminf(y, kYSat), ey) : y) + ph;
            // r <= 0 / NaN cells were counted by the row pass; the replacement rule
            // (poisson.py:606-616) drops such a cell whole, so it gets weight +1 to cancel
            // the -1 that the closed-form sum over ALL cells gives it (finish kernel)
            // (padded slots carry x = 0 and must stay weightless)
            const float xr = (r > 0.f && r < INFINITY) ? xv[j] * __builtin_amdgcn_rcpf(r)
                                                       : (xv[j] > 0.f ? 1.f : 0.f);
            gV = fma4(LIK == 1 ? xr * ey : xr, zz[j], gV);
            gA = fma4(gv[j], gg[j], gA);
            gph += xr;
          }
        }
      }
    }
    rr0 = rr1; xx0 = xx1; gx0 = gx1; cnt0 = cnt1;
    rr1 = rr2; xx1 = xx2; gx1 = gx2; cnt1 = cnt2;
  }
  // ---- transpose through LDS so each atomic instruction covers whole rows --
  float4* st4 = reinterpret_cast<float4*>(&stage[wid][grp][0]);
  st4[sub] = gV;
  st4[LPN + sub] = gA;
  __builtin_amdgcn_wave_barrier();
  const float* flat = &stage[wid][0][0];
#pragma unroll
  for (int i = 0; i < (NG * 2 * KP) / 64; ++i) {
    const int e = i * 64 + lane;
    const int c = e / (2 * KP), rem = e % (2 * KP);
    const int dd = __shfl(d, c * LPN);                // column of group c's item
    const int okc = __shfl((int)ok, c * LPN);
    const float v = flat[e];
    if (okc && v != 0.f) {
      float* dst = (rem >= KP ? gAp + (size_t)dd * KP + (rem - KP) : gVp + (size_t)dd * KP + rem);
      atomicAdd(dst, v);
    }
  }
  if (ok && sub == 0 && gph != 0.f) atomicAdd(&gphi[d], gph);
}

// ---- the same pass with 16-B entry fetches (needs padded panel-CSC arrays) ----
// LIK: 0 Poisson / linear, 1 Poisson / log_transform, 2 Bernoulli(logits) / linear
template <int KP, int LIK, bool PACKED>
__global__ __launch_bounds__(256, COL_WIDE_WAVES) void col_pass_wide_kernel(
    int D, int n_panels, int row_base, int blocks_per_panel,
    const int32_t* __restrict__ item_ptr, const int4* __restrict__ items,
    const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ pc_gval, const float* __restrict__ Vp,
    const float* __restrict__ phi, const float* __restrict__ z, const float* __restrict__ gzs,
    float* __restrict__ gAp, float* __restrict__ gVp, float* __restrict__ gphi,
    const uint8_t* __restrict__ ctype, const int32_t* __restrict__ item_mid, int half_sel,
    int64_t Brows, int64_t acc_stride, const double* __restrict__ pack_dacc,
    float* __restrict__ pack_tail, int64_t dacc_stride, const uint32_t* __restrict__ pc_ent,
    int panel_rows) {
  if (pack_dacc && blockIdx.x == 0) {
    pack_block<KP>(pack_dacc + (size_t)blockIdx.y * dacc_stride, pack_tail + (size_t)blockIdx.y * acc_stride);
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
  constexpr int LPN = KP / 4;
  constexpr int NG = 64 / LPN;                        // items per wave
  constexpr int GRP = LPN < COL_GRP ? LPN : COL_GRP;  // entries gathered back to back
  __shared__ __attribute__((aligned(16))) float stage[4][NG][2 * KP];
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int wid = threadIdx.x >> 6;
  // block id -> (panel, block of 4*NG items); blockIdx % 8 = panel residue
  // (batches of fewer than 8 panels use a flat mapping: the residue mapping
  // would leave the XCDs of the missing residues with empty workgroups only)
  const int64_t L = (int64_t)blockIdx.x - (pack_dacc ? 1 : 0);   // block 0 is the pack block when asked
  int p, ib;
  if (n_panels < 8) {
    p = (int)(L / blocks_per_panel);
    ib = (int)(L % blocks_per_panel);
  } else {
    const int x = (int)(L & 7);
    const int64_t q = L >> 3;
    p = 8 * (int)(q / blocks_per_panel) + x;
    ib = (int)(q % blocks_per_panel);
  }
  if (p >= n_panels) return;                          // block-uniform
  // item range of this launch: the whole panel, or one column half of it (the host
  // sorts a panel's items by half first: multi-GPU overlap of the all-reduce)
  const int ilo = half_sel == 2 ? item_mid[p] : item_ptr[p];
  const int ihi = half_sel == 1 ? item_mid[p] : item_ptr[p + 1];
  const int i0 = ilo + ib * 4 * NG;
  if (i0 >= ihi) return;                              // block-uniform
  const int it = i0 + wid * NG + grp;
  const bool ok = it < ihi;
  int cur = 0, end = 0, d = 0;
  if (ok) {
    const int4 im = items[it];
    cur = im.x;
    end = im.x + im.y;
    d = im.z;
  }
  const float4 vp = ok ? gather4<LPN>(Vp, d, sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float ph = ok ? phi[d] : 1.f;
  const bool bern = LIK == 2 || LIK == 4 || (LIK == 3 && ok && ctype[d]);   // item's column is Bernoulli
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;

  // Wide fetch: every lane reads FOUR consecutive entries of its group's list (one 16-B
  // load per array; list starts are only 4-B aligned), so a fetch covers 4*LPN entries =
  // one 128-B line per group and array at K = 32, against one 32-B piece of a line per
  // 8 entries in col_pass_kernel (four times the vector-cache line slots for the same
  // bytes).  Reads up to 4*LPN - 1 entries past the end of a list: the caller guarantees
  // that much readable padding behind the panel-CSC arrays (spmf_counts.pc_pad).
  struct __attribute__((packed, aligned(4))) I4 { int x, y, z, w; };
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  constexpr int FE = 4 * LPN;                         // entries per fetch and group
  // fetch_raw only LOADS (the fetch for the next iteration stays untouched until then: decoding
  // it right away would put the wait for it in front of this iteration's gathers); decode turns
  // the words of the CURRENT fetch into batch rows / counts where they are used.
  // PACKED: one word per entry, row inside the panel << 16 | count (spmf_counts.pc_ent).
  auto fetch_raw = [&](I4& r, F4& x, F4& g, int& cnt_) {
    cnt_ = min(FE, end - cur);                        // 0 once the item is exhausted
    const int e = cur + 4 * sub;
    r = {0, 0, 0, 0};
    x = {0.f, 0.f, 0.f, 0.f};
    g = {0.f, 0.f, 0.f, 0.f};
    if (4 * sub < cnt_) {
      if (PACKED) {
        r = *reinterpret_cast<const I4*>(reinterpret_cast<const int32_t*>(pc_ent) + e);
      } else {
        r = *reinterpret_cast<const I4*>(pc_row + e);
        x = *reinterpret_cast<const F4*>(pc_val + e);
      }
      if (LIK == 1 || LIK == 4) g = *reinterpret_cast<const F4*>(pc_gval + e);
    }
    cur += cnt_;
  };
  const int pbase = PACKED ? p * panel_rows : -row_base;   // batch row of a list word
  auto decode = [&](const I4& r, const F4& x, const F4& g, int cnt_, int (&rr_)[4], float (&xx_)[4],
                    float (&gx_)[4]) {
    const int left = cnt_ - 4 * sub;                  // valid components of this lane
    const int rw[4] = {r.x, r.y, r.z, r.w};
    const float xw[4] = {x.x, x.y, x.z, x.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool on = left > t;
      if (PACKED) {
        const uint32_t w = (uint32_t)rw[t];
        rr_[t] = on ? (int)(w >> 16) + pbase : 0;
        xx_[t] = on ? (float)(w & 0xffffu) : 0.f;
      } else {
        rr_[t] = on ? rw[t] + pbase : 0;
        xx_[t] = on ? xw[t] : 0.f;
      }
      gx_[t] = on ? gw[t] : 0.f;
    }
  };

  I4 rA, rB;
  F4 xA, xB, gA_, gB_;
  int cnt0, cnt1;
  fetch_raw(rA, xA, gA_, cnt0);
  while (__any(cnt0 > 0)) {
    fetch_raw(rB, xB, gB_, cnt1);                     // one fetch (4*LPN entries) ahead of use
    int rr0[4];
    float xx0[4], gx0[4];
    decode(rA, xA, gA_, cnt0, rr0, xx0, gx0);
#pragma unroll
    for (int g0 = 0; g0 < FE; g0 += GRP) {
      if (__any(cnt0 > g0)) {                         // wave-uniform
        float4 zz[GRP], gg[GRP];
        float xv[GRP], gv[GRP];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const int q = g0 + j;                       // entry q of the fetch: lane q/4, component q%4
          const int src = grp * LPN + q / 4;
          const int b = __shfl(rr0[q % 4], src);
          xv[j] = __shfl(xx0[q % 4], src);
          gv[j] = (LIK == 1 || LIK == 4) ? __shfl(gx0[q % 4], src) : xv[j];
          zz[j] = gather4<LPN>(z, b, sub);
          gg[j] = gather4<LPN>(gzs, b, sub);
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          if (bern) {
            float wv = xv[j];
            if (LIK == 4) wv *= expf(fminf(group_sum<LPN>(dot4(zz[j], vp)), kYSat));
            gV = fma4(wv, zz[j], gV);
            gA = fma4(gv[j], gg[j], gA);
            gph += xv[j];
          } else {
            const float y = group_sum<LPN>(dot4(zz[j], vp));
            float ey = 1.f;
            const float r = (LIK == 1 ? expm1_dec(fminf(y, kYSat), ey) : y) + ph;
            const float xr = (r > 0.f && r < INFINITY) ? xv[j] * __builtin_amdgcn_rcpf(r)
                                                       : (xv[j] > 0.f ? 1.f : 0.f);
            gV = fma4(LIK == 1 ? xr * ey : xr, zz[j], gV);
            gA = fma4(gv[j], gg[j], gA);
            gph += xr;
          }
        }
      }
    }
    rA = rB; xA = xB; gA_ = gB_;
    cnt0 = cnt1;
  }
  // ---- transpose through LDS so each atomic instruction covers whole rows --
  float4* st4 = reinterpret_cast<float4*>(&stage[wid][grp][0]);
  st4[sub] = gV;
  st4[LPN + sub] = gA;
  __builtin_amdgcn_wave_barrier();
  const float* flat = &stage[wid][0][0];
#pragma unroll
  for (int i = 0; i < (NG * 2 * KP) / 64; ++i) {
    const int e = i * 64 + lane;
    const int c = e / (2 * KP), rem = e % (2 * KP);
    const int dd = __shfl(d, c * LPN);                // column of group c's item
    const int okc = __shfl((int)ok, c * LPN);
    const float v = flat[e];
    if (okc && v != 0.f) {
      float* dst = (rem >= KP ? gAp + (size_t)dd * KP + (rem - KP) : gVp + (size_t)dd * KP + rem);
      atomicAdd(dst, v);
    }
  }
  if (ok && sub == 0 && gph != 0.f) atomicAdd(&gphi[d], gph);
}

template <int KP>
static bool launch_col_t(const ColArgs& a, hipStream_t st) {
  constexpr int NG = 64 / (KP / 4);
  const int per_block = 4 * NG;
  const int bpp = (a.max_items_per_panel + per_block - 1) / per_block;
  if (bpp < 1) return false;
  const int64_t nt = (a.n_panels + 7) / 8;
  // + 1: the pack block (kernels: blockIdx.x == 0, the item blocks shift by one)
  const int64_t nb = (a.n_panels < 8 ? (int64_t)a.n_panels * bpp : nt * bpp * 8) + (a.pack_dacc ? 1 : 0);
  const int4* items = reinterpret_cast<const int4*>(a.items);
#define SPMF_COL_ARGS                                                                            \
  dim3((unsigned)nb, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.D, a.n_panels, a.row_base, bpp,   \
      a.item_ptr, items, a.pc_row, a.pc_val, a.pc_gval, a.Vp, a.phi, a.z, a.gzs, a.gAp, a.gVp, \
      a.gphi, a.ctype, a.item_mid, a.half_sel, a.B, a.acc_stride, a.pack_dacc, a.pack_tail,     \
      a.dacc_stride
  const bool wide = COL_WIDE && a.pc_pad >= KP - 1;   // 4*LPN - 1 entries of readable padding
  // packed lists (spmf_counts.pc_ent: row in panel << 16 | count) when the batch carries them
  const bool packed = wide && a.pc_ent && a.panel_rows > 0 && a.panel_rows <= 65536;
#define SPMF_COL_LAUNCH(L_)                                                                        \
  do {                                                                                             \
    if (packed)                                                                                    \
      hipLaunchKernelGGL((col_pass_wide_kernel<KP, L_, true>), SPMF_COL_ARGS, a.pc_ent,            \
                         a.panel_rows);                                                            \
    else if (wide)                                                                                 \
      hipLaunchKernelGGL((col_pass_wide_kernel<KP, L_, false>), SPMF_COL_ARGS, a.pc_ent,           \
                         a.panel_rows);                                                            \
    else                                                                                           \
      hipLaunchKernelGGL((col_pass_kernel<KP, L_>), SPMF_COL_ARGS);                                \
  } while (0)
  if (a.logt == 4) SPMF_COL_LAUNCH(4);
  else if (a.logt == 3) SPMF_COL_LAUNCH(3);
  else if (a.logt == 2) SPMF_COL_LAUNCH(2);
  else if (a.logt == 1) SPMF_COL_LAUNCH(1);
  else SPMF_COL_LAUNCH(0);
#undef SPMF_COL_LAUNCH
#undef SPMF_COL_ARGS
  return true;
}

bool launch_col_pass(int KP, const ColArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: return launch_col_t<4>(a, st);
    case 8: return launch_col_t<8>(a, st);
    case 16: return launch_col_t<16>(a, st);
    case 32: return launch_col_t<32>(a, st);
    case 64: return launch_col_t<64>(a, st);
    default: return false;
  }
}

}  // namespace spmf
