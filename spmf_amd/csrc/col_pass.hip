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
// at full rate).  Atomic traffic is n_items * (2KP+1)*4 B (0.49 GB on C3 with 88 panels).
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
__device__ __forceinline__ void pack_block(const double* __restrict__ dacc, float* __restrict__ tail,
                                           const double* __restrict__ det_slots) {
  if (det_slots) {
    // deterministic mode: the row pass's workgroups left their sums in their own slots (slots 3..5 -- dense
    // sum, saturation, spare -- are not produced by this mode's linear decoder and stay 0).  A fixed
    // association: 256 / 128 thread groups each add a contiguous eighth (quarter ... ) of the slots in
    // workgroup order, then the partial sums are added in group order -- the same tree for the same
    // workgroup count, 8 x shorter chains than one pass (4096 slots on the 256-thread launch shape)
    constexpr int LEN = kDaccHead + KP;                 // 10 ... 70 values
    constexpr int SEG = 256 / (LEN <= 32 ? 32 : (LEN <= 64 ? 64 : 128));
    constexpr int W = 256 / SEG;
    __shared__ double seg_sum[SEG][W];
    const int i = threadIdx.x % W, g = threadIdx.x / W;
    const int nb = min(max((int)det_slots[0], 0), kDetMaxBlocks);   // (never past the slot area, whatever the word says)
    const int per = (nb + SEG - 1) / SEG;
    double v = 0.0;
    if (i < LEN) {
      const double* sl = det_slots + kDetMeta + i;
      const int b1 = min(nb, (g + 1) * per);
      for (int b = g * per; b < b1; ++b) v += sl[(size_t)b * LEN];
    }
    seg_sum[g][i] = v;
    __syncthreads();
    if (threadIdx.x < LEN) {
      double t = 0.0;
#pragma unroll
      for (int s = 0; s < SEG; ++s) t += seg_sum[s][threadIdx.x];
      const float hi = (float)t;
      tail[2 * threadIdx.x] = hi;
      tail[2 * threadIdx.x + 1] = (float)(t - (double)hi);
    }
    return;
  }
  const int i = threadIdx.x;
  if (i < kDaccHead + KP) {
    double v = 0.0;
    {
#pragma unroll
      for (int r = 0; r < kDaccRep; ++r) v += dacc[(size_t)r * (kDaccHead + KP) + i];
    }
    const float hi = (float)v;
    tail[2 * i] = hi;
    tail[2 * i + 1] = (float)(v - (double)hi);
  }
}

// ONE kernel body for both fetch shapes (the round-3 file carried it twice):
//   EPL = 4 ("wide", the default): every lane reads FOUR consecutive entries of its group's list (one
//     16-B load per array; list starts are only 4-B aligned), so a fetch covers 4*LPN entries = one 128-B
//     line per group and array at K = 32, against one 32-B piece of a line per 8 entries with EPL = 1
//     (four times the vector-cache line slots for the same bytes).  Reads up to 4*LPN - 1 entries past the
//     end of a list: the caller guarantees that much readable padding behind the panel-CSC arrays
//     (spmf_counts.pc_pad).  PACKED: one word per entry, row inside the panel << 16 | count
//     (spmf_counts.pc_ent).
//   EPL = 1 ("narrow"): one entry per lane and fetch, never reads behind a list (pc_pad = 0: a C-ABI
//     caller whose arrays carry no padding).
// LIK: 0 Poisson / linear, 1 Poisson / log_transform, 2 Bernoulli(logits) / linear, 3 mixed, 4 Bernoulli / exp
template <int KP, int LIK, int EPL, bool PACKED>
__global__ __launch_bounds__(256, EPL == 4 ? COL_WIDE_WAVES : 1) void col_pass_kernel(
    int D, int n_panels, int row_base, int blocks_per_panel,
    const int32_t* __restrict__ item_ptr, const int4* __restrict__ items,
    const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ pc_gval, const float* __restrict__ Vp,
    const float* __restrict__ phi, const float* __restrict__ z, const float* __restrict__ gzs,
    float* __restrict__ gAp, float* __restrict__ gVp, float* __restrict__ gphi,
    const uint8_t* __restrict__ ctype, const int32_t* __restrict__ item_mid, int half_sel,
    int64_t Brows, int64_t acc_stride, const double* __restrict__ pack_dacc,
    float* __restrict__ pack_tail, int64_t dacc_stride, const uint32_t* __restrict__ pc_ent,
    int panel_rows, float* __restrict__ det_part, int64_t det_part_stride,
    const double* __restrict__ det_slots, int64_t det_stride) {
  static_assert(EPL == 1 || EPL == 4, "entries per lane and fetch: 1 or 4");
  static_assert(!PACKED || EPL == 4, "the packed lists are read by the wide fetch only");
  if (pack_dacc && blockIdx.x == 0) {
    pack_block<KP>(pack_dacc + (size_t)blockIdx.y * dacc_stride, pack_tail + (size_t)blockIdx.y * acc_stride,
                   det_slots ? det_slots + (size_t)blockIdx.y * det_stride : nullptr);
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
  // z and xi*gz as range-checked tables: the slots behind a list's end (and the lane groups without an item)
  // ask for row kPadRow, which the address unit drops (common.h GTable)
  const GTable zt = gtable(z, Brows, KP), gt = gtable(gzs, Brows, KP);
  // entries gathered back to back: COL_GRP of the 4 * LPN a wide fetch holds (also at K <= 8, where a lane group
  // is one or two lanes and used to keep ONE or two entries in flight); the narrow fetch holds LPN
  constexpr int GRP = EPL == 4 ? COL_GRP : (LPN < COL_GRP ? LPN : COL_GRP);
  static_assert((EPL * LPN) % GRP == 0, "a fetch is a whole number of gather groups");
  __shared__ __attribute__((aligned(16))) float stage[4][NG][2 * KP];
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int bp_row = grp * LPN * 4;                   // byte address of the group's first lane (ds_bpermute)
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

  struct __attribute__((packed, aligned(4))) I4 { int x, y, z, w; };
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  constexpr int FE = EPL * LPN;                       // entries per fetch and group
  // One fetch of a lane: EPL list words per array.  fetch_raw only LOADS (the fetch for a later
  // iteration stays untouched until then: decoding it right away would put the wait for it in front of
  // this iteration's gathers); decode turns the words of the CURRENT fetch into batch rows / counts /
  // g(x) where they are used.
  struct Raw {
    int r[EPL];
    float x[EPL], g[EPL];
    int cnt;
  };
  auto fetch_raw = [&](Raw& f) {
    f.cnt = min(FE, end - cur);                       // 0 once the item is exhausted
    const int e = cur + EPL * sub;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
      f.r[t] = 0;
      f.x[t] = 0.f;
      f.g[t] = 0.f;
    }
    if (EPL * sub < f.cnt) {
      if constexpr (EPL == 4) {
        if (PACKED) {
          const I4 w = *reinterpret_cast<const I4*>(reinterpret_cast<const int32_t*>(pc_ent) + e);
          f.r[0] = w.x; f.r[1] = w.y; f.r[2] = w.z; f.r[3] = w.w;
        } else {
          const I4 w = *reinterpret_cast<const I4*>(pc_row + e);
          const F4 v = *reinterpret_cast<const F4*>(pc_val + e);
          f.r[0] = w.x; f.r[1] = w.y; f.r[2] = w.z; f.r[3] = w.w;
          f.x[0] = v.x; f.x[1] = v.y; f.x[2] = v.z; f.x[3] = v.w;
        }
        if (LIK == 1 || LIK == 4) {                   // g(x) = log(x/eta+1), data side
          const F4 v = *reinterpret_cast<const F4*>(pc_gval + e);
          f.g[0] = v.x; f.g[1] = v.y; f.g[2] = v.z; f.g[3] = v.w;
        }
      } else {
        f.r[0] = pc_row[e];
        f.x[0] = pc_val[e];
        if (LIK == 1 || LIK == 4) f.g[0] = pc_gval[e];
      }
    }
    cur += f.cnt;
  };
  const int pbase = PACKED ? p * panel_rows : -row_base;   // batch row of a list word
  auto decode = [&](const Raw& f, int (&rr_)[EPL], float (&xx_)[EPL], float (&gx_)[EPL]) {
    const int left = f.cnt - EPL * sub;               // valid components of this lane
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
      const bool on = left > t;
      if (PACKED) {
        const uint32_t w = (uint32_t)f.r[t];
        rr_[t] = on ? (int)(w >> 16) + pbase : kPadRow;
        xx_[t] = on ? (float)(w & 0xffffu) : 0.f;
      } else {
        rr_[t] = on ? f.r[t] + pbase : kPadRow;
        xx_[t] = on ? f.x[t] : 0.f;
      }
      gx_[t] = on ? f.g[t] : 0.f;
    }
  };

  // the wide fetch runs one fetch (4*LPN entries) ahead of use, the narrow one two (2*LPN entries)
  Raw fa, fb, fc;
  fetch_raw(fa);
  if (EPL == 1) fetch_raw(fb);
  while (__any(fa.cnt > 0)) {
    if (EPL == 1) fetch_raw(fc);
    else fetch_raw(fb);
    int rr0[EPL];
    float xx0[EPL], gx0[EPL];
    decode(fa, rr0, xx0, gx0);
    const int cnt0 = fa.cnt;
#pragma unroll
    for (int g0 = 0; g0 < FE; g0 += GRP) {
      if (__any(cnt0 > g0)) {                         // wave-uniform
        float4 zz[GRP], gg[GRP];
        float xv[GRP], gv[GRP];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const int q = g0 + j;                       // entry q of the fetch: lane q / EPL, component q % EPL
          // (source lane grp * LPN + q / EPL: the lane-dependent part is one byte-address register of the kernel,
          //  the compile-time part an add -- __shfl shifts the index per call; row_pass.hip ROW_BPERM_IMM)
          const int sa = bp_row + (q / EPL) * 4;
          const int b = __builtin_amdgcn_ds_bpermute(sa, rr0[q % EPL]);
          xv[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(xx0[q % EPL])));
          gv[j] = (LIK == 1 || LIK == 4) ? __int_as_float(__builtin_amdgcn_ds_bpermute(sa, __float_as_int(gx0[q % EPL])))
                                         : xv[j];
          zz[j] = gather4<LPN>(zt, b, sub);
          gg[j] = gather4<LPN>(gt, b, sub);
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
    if (EPL == 1) {
      fa = fb;
      fb = fc;
    } else {
      fa = fb;
    }
  }
  // ---- transpose through LDS so each atomic instruction covers whole rows --
  // (a wave leaves through 8 float-atomic wave instructions, each covering two whole 128-B gradient
  //  rows: the shape MI355X runs atomics at full rate; n_items * (2KP+1)*4 B, 0.49 GB on C3.  The
  //  owner form that needs no atomics was measured and rejected: profiles/r04_col_owner_probe.txt)
  if (det_part) {
    // deterministic mode: the item's sums go to its own slot (single writer, plain 16-B stores);
    // det_reduce_kernel adds a column's items up in (panel, segment) order
    if (ok) {
      float4* dst = reinterpret_cast<float4*>(det_part + (size_t)blockIdx.y * det_part_stride +
                                              (size_t)(it - item_ptr[0]) * det_part_len(KP));
      dst[sub] = gV;
      dst[LPN + sub] = gA;
      if (sub == 0) dst[2 * LPN] = make_float4(gph, 0.f, 0.f, 0.f);
    }
    return;
  }
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
      a.dacc_stride, a.pc_ent, a.panel_rows, a.det_part, a.det_part_stride, a.det_slots, a.det_stride
  const bool wide = COL_WIDE && a.pc_pad >= KP - 1;   // 4*LPN - 1 entries of readable padding
  // packed lists (spmf_counts.pc_ent: row in panel << 16 | count) when the batch carries them
  const bool packed = wide && a.pc_ent && a.panel_rows > 0 && a.panel_rows <= 65536;
#define SPMF_COL_LAUNCH(L_)                                                                        \
  do {                                                                                             \
    if (packed)                                                                                    \
      hipLaunchKernelGGL((col_pass_kernel<KP, L_, 4, true>), SPMF_COL_ARGS);                       \
    else if (wide)                                                                                 \
      hipLaunchKernelGGL((col_pass_kernel<KP, L_, 4, false>), SPMF_COL_ARGS);                      \
    else                                                                                           \
      hipLaunchKernelGGL((col_pass_kernel<KP, L_, 1, false>), SPMF_COL_ARGS);                      \
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

// Deterministic mode: one lane group per column adds the partial sums of the column's work items in
// generation order -- panel by panel, segment by segment -- into gV', gA', gphi (zeroed by the prep
// launch; single writer per column).  Reads n_items * det_part_len floats once.
template <int KP>
__global__ __launch_bounds__(256) void det_reduce_kernel(int D, int n_panels, const int32_t* __restrict__ list_first,
                                                         const int32_t* __restrict__ item_pos,
                                                         const int32_t* __restrict__ item_ptr,
                                                         const float* __restrict__ part, int64_t part_stride,
                                                         float* __restrict__ gAp, float* __restrict__ gVp,
                                                         float* __restrict__ gphi, int64_t acc_stride) {
  constexpr int LPN = KP / 4;
  const int sub = threadIdx.x % LPN;
  const int d = blockIdx.x * (256 / LPN) + threadIdx.x / LPN;
  if (d >= D) return;
  part += (size_t)blockIdx.y * part_stride;
  gAp += (size_t)blockIdx.y * acc_stride;
  gVp += (size_t)blockIdx.y * acc_stride;
  gphi += (size_t)blockIdx.y * acc_stride;
  const int it0 = item_ptr[0];
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;
  auto add_item = [&](int r) {
    const float4* src = reinterpret_cast<const float4*>(part + (size_t)(item_pos[r] - it0) * det_part_len(KP));
    const float4 a = src[sub], b = src[LPN + sub];
    gV.x += a.x; gV.y += a.y; gV.z += a.z; gV.w += a.w;
    gA.x += b.x; gA.y += b.y; gA.z += b.z; gA.w += b.w;
    if (sub == 0) gph += src[2 * LPN].x;
  };
  // panels in chunks of DP: the list bounds, then the first items' positions, then their slots are
  // fetched for the whole chunk before anything is added (the ADDS stay in panel order; a panel's
  // further segments -- lists longer than one item -- follow in its turn)
  constexpr int DP = 8;
  for (int p0 = 0; p0 < n_panels; p0 += DP) {
    int r0[DP], r1[DP], jp[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      const bool on = p0 + i < n_panels;
      const int32_t* lf = list_first + (size_t)(on ? p0 + i : p0) * D + d;
      r0[i] = on ? lf[0] : 0;
      r1[i] = on ? lf[1] : 0;
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) jp[i] = r1[i] > r0[i] ? item_pos[r0[i]] - it0 : 0;
    float4 a[DP], b[DP];
    float g[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      const float4* src = reinterpret_cast<const float4*>(part + (size_t)jp[i] * det_part_len(KP));
      const bool on = r1[i] > r0[i];
      a[i] = on ? src[sub] : make_float4(0.f, 0.f, 0.f, 0.f);
      b[i] = on ? src[LPN + sub] : make_float4(0.f, 0.f, 0.f, 0.f);
      g[i] = (on && sub == 0) ? src[2 * LPN].x : 0.f;
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      if (r1[i] > r0[i]) {            // (an empty list adds nothing: not even +0, which could turn -0 into +0)
        gV.x += a[i].x; gV.y += a[i].y; gV.z += a[i].z; gV.w += a[i].w;
        gA.x += b[i].x; gA.y += b[i].y; gA.z += b[i].z; gA.w += b[i].w;
        if (sub == 0) gph += g[i];
        for (int r = r0[i] + 1; r < r1[i]; ++r) add_item(r);
      }
    }
  }
  reinterpret_cast<float4*>(gVp + (size_t)d * KP)[sub] = gV;
  reinterpret_cast<float4*>(gAp + (size_t)d * KP)[sub] = gA;
  if (sub == 0) gphi[d] = gph;
}

void launch_det_reduce(const DetReduceArgs& a, hipStream_t st) {
  if (a.D <= 0 || a.n_panels <= 0) return;
#define SPMF_DET(KP_)                                                                                        \
  hipLaunchKernelGGL((det_reduce_kernel<KP_>), dim3((a.D + 256 / (KP_ / 4) - 1) / (256 / (KP_ / 4)),          \
                                                    a.S > 1 ? a.S : 1),                                       \
                     dim3(256), 0, st, a.D, a.n_panels, a.list_first, a.item_pos, a.item_ptr, a.part,        \
                     a.part_stride, a.gAp, a.gVp, a.gphi, a.acc_stride)
  switch (a.KP) {
    case 4: SPMF_DET(4); break;
    case 8: SPMF_DET(8); break;
    case 16: SPMF_DET(16); break;
    case 32: SPMF_DET(32); break;
    case 64: SPMF_DET(64); break;
    default: break;
  }
#undef SPMF_DET
}

bool launch_col_pass(int KP, const ColArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: return launch_col_t<4>(a, st);
    case 8: return launch_col_t<8>(a, st);
    case 16: return launch_col_t<16>(a, st);
    case 32: return launch_col_t<32>(a, st);
    case 64: return launch_col_t<64>(a, st);
    case 128: case 256: return launch_col_widek(KP, a, st);   // widek.hip
    default: return false;
  }
}

}  // namespace spmf
