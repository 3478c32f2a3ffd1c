// col_pass.hip -- column-side (transposed) accumulation of the factor
// gradients over the panel-CSC copy of the batch (gfx950, wave64).
//
// For every feature column d (SURVEY 8a gradient formulas, derived from
// poisson.py:156-184,582-701):
//   r_bd     = <z_b, V'_d> + phi_d                 recomputed, V'_d in registers
//   gV'_d   += (x_bd/r_bd) z_b                     -> d/dv   after the chain
//   gphi_d  += x_bd/r_bd                           -> d/dw, d/ds
//   gA'_d   += x_bd * (xi_b gz_b)                  -> d/du, d/ds
// The minus-one part of (x/r - 1) over ALL rows (stored or not) is closed
// form (sum_b z_b and B) and is applied by the finish kernel.
//
// Mapping: a gathered row (z_b or xi_b*gz_b) is KP floats = LPN=KP/4 lanes x
// float4, so a wave holds NG=64/LPN lane groups; EACH GROUP OWNS ONE COLUMN of
// one row panel and keeps that column's gV'/gA'/gphi slices in registers, so
// there is no cross-lane reduction besides the DPP fold of the dot product.
// A group streams its (panel, column) list LPN entries per fetch (one per
// lane, a 4*LPN-B contiguous read), two fetches ahead of use, and broadcasts
// an entry inside the group with ds_bpermute; gathers are issued four entries
// (eight 16-B loads per lane) at a time.
//
// L2 residency is what makes the 2 x 4*KP-B-per-entry gathers affordable
// (measured: served from Infinity Cache instead, the pass runs at ~8.6 TB/s
// of gather traffic = 3 ms on the C3 shape).  Panels are `panel_rows`
// consecutive rows (2*panel_rows*KP*4 B of z / xi*gz, 2 MB at the default),
// a wave handles ONE panel, and workgroup ids are ordered panel-major with
// p = 8t + blockIdx%8, so with the observed round-robin workgroup->XCD
// placement each XCD's resident workgroups share one panel.  Speed heuristic
// only: results do not depend on placement.
//
// A wave leaves through an LDS transpose and 8 float-atomic wave instructions,
// each covering two whole 128-B gradient rows (the shape MI355X runs atomics
// at full rate).  Atomic traffic is nnz/len * (2KP+1)*4 B with len = mean
// entries per (panel, column) list (~41 at the default) -- 0.63 GB on C3.
// Float atomics make the low-order bits of the gradient run-to-run dependent;
// parity tolerance is 1e-5 relative (north_star).
#include "common.h"
#include "kernels.h"

namespace spmf {

#ifndef COL_GRP
#define COL_GRP 4
#endif

template <int KP, int TP, bool LOGT>
__global__ __launch_bounds__(256) void col_pass_kernel(
    int D, int n_panels, int row_base, const int32_t* __restrict__ pc_ptr,
    const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ Vp, const float* __restrict__ phi, const float* __restrict__ z,
    const float* __restrict__ gzs, float* __restrict__ gAp, float* __restrict__ gVp,
    float* __restrict__ gphi, const float* __restrict__ pc_gval) {
  constexpr int LPN = KP / 4;
  constexpr int NG = 64 / LPN;                  // columns per wave
  constexpr int GRP = LPN < COL_GRP ? LPN : COL_GRP;  // entries gathered back to back
  __shared__ __attribute__((aligned(16))) float stage[4][NG][2 * KP];
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int wid = threadIdx.x >> 6;
  // block id -> (panel, block of 4*NG columns); blockIdx % 8 = panel residue
  const int64_t L = blockIdx.x;
  const int x = (int)(L & 7);
  const int64_t q = L >> 3;
  const int ncbb = (D + 4 * NG - 1) / (4 * NG);
  const int t = (int)(q / ncbb), cbb = (int)(q % ncbb);
  // super-panel = TP consecutive panels walked in order by the same wave
  const int p0 = (8 * t + x) * TP;
  if (p0 >= n_panels) return;                   // block-uniform
  const int d0 = (cbb * 4 + wid) * NG;
  const int d = d0 + grp;
  const bool colok = d < D;

  int seg_s[TP], seg_e[TP];
#pragma unroll
  for (int i = 0; i < TP; ++i) {
    seg_s[i] = seg_e[i] = 0;
    if (colok && p0 + i < n_panels) {
      const int64_t pb = (int64_t)(p0 + i) * (D + 1) + d;
      seg_s[i] = pc_ptr[pb];
      seg_e[i] = pc_ptr[pb + 1];
    }
  }
  int cur = seg_s[0], end = seg_e[0], seg = 1;
  const float4 vp = colok ? reinterpret_cast<const float4*>(Vp)[(size_t)d * LPN + sub]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
  const float ph = colok ? phi[d] : 1.f;
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;

  auto fetch = [&](int& rr_, float& xx_, float& gx_, int& cnt_) {
#pragma unroll
    for (int i = 1; i < TP; ++i)                // next non-empty list of the super-panel
      if (cur == end && seg == i) {
        cur = seg_s[i];
        end = seg_e[i];
        seg = i + 1;
      }
    cnt_ = min(LPN, end - cur);                 // 0 once the lists are exhausted
    const int e = cur + sub;
    rr_ = sub < cnt_ ? pc_row[e] - row_base : 0;
    xx_ = sub < cnt_ ? pc_val[e] : 0.f;
    gx_ = (LOGT && sub < cnt_) ? pc_gval[e] : 0.f;   // g(x) = log(x/eta+1), data side
    cur += cnt_;
  };

  int rr0, cnt0, rr1, cnt1;
  float xx0, xx1, gx0, gx1;
  fetch(rr0, xx0, gx0, cnt0);
  fetch(rr1, xx1, gx1, cnt1);
  while (__any(cnt0 > 0)) {
    int rr2, cnt2;
    float xx2, gx2;
    fetch(rr2, xx2, gx2, cnt2);                 // two fetches ahead of use
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      if (__any(cnt0 > g0)) {                   // wave-uniform
        float4 zz[GRP], gg[GRP];
        float xv[GRP], gv[GRP];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const int src = grp * LPN + g0 + j;
          const int b = __shfl(rr0, src);
          xv[j] = __shfl(xx0, src);
          gv[j] = LOGT ? __shfl(gx0, src) : xv[j];
          zz[j] = gather4<LPN>(z, b, sub);
          gg[j] = gather4<LPN>(gzs, b, sub);
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
          const float y = group_sum<LPN>(dot4(zz[j], vp));
          const float ey = LOGT ? expf(y) : 1.f;
          const float r = (LOGT ? ey - 1.f : y) + ph;
          // r <= 0 / NaN cells were counted by the row pass; +inf gives 0
          const float xr = r > 0.f ? xv[j] * __builtin_amdgcn_rcpf(r) : 0.f;
          gV = fma4(LOGT ? xr * ey : xr, zz[j], gV);
          gA = fma4(gv[j], gg[j], gA);
          gph += xr;
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
    const int dd = d0 + c;
    const float v = flat[e];
    if (dd < D && v != 0.f) {
      float* dst = (rem >= KP ? gAp + (size_t)dd * KP + (rem - KP) : gVp + (size_t)dd * KP + rem);
      atomicAdd(dst, v);
    }
  }
  if (colok && sub == 0 && gph != 0.f) atomicAdd(&gphi[d], gph);
}

template <int KP>
static void launch_col_t(const ColArgs& a, hipStream_t st) {
  constexpr int NG = 64 / (KP / 4);
  const int64_t ncbb = (a.D + 4 * NG - 1) / (4 * NG);
  const int tp = a.logt ? 1 : (a.panels_per_wave >= 4 ? 4 : (a.panels_per_wave >= 2 ? 2 : 1));
  const int64_t nsp = (a.n_panels + tp - 1) / tp;   // super-panels
  const int64_t nt = (nsp + 7) / 8;
  const int64_t nb = nt * ncbb * 8;
#define SPMF_COL_LAUNCH(TP_, LT_)                                                              \
  hipLaunchKernelGGL((col_pass_kernel<KP, TP_, LT_>), dim3((unsigned)nb), dim3(256), 0, st,    \
                     a.D, a.n_panels, a.row_base, a.pc_ptr, a.pc_row, a.pc_val, a.Vp, a.phi,   \
                     a.z, a.gzs, a.gAp, a.gVp, a.gphi, a.pc_gval)
  if (a.logt) SPMF_COL_LAUNCH(1, true);
  else if (tp == 4) SPMF_COL_LAUNCH(4, false);
  else if (tp == 2) SPMF_COL_LAUNCH(2, false);
  else SPMF_COL_LAUNCH(1, false);
#undef SPMF_COL_LAUNCH
}

void launch_col_pass(int KP, const ColArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_col_t<4>(a, st); break;
    case 8: launch_col_t<8>(a, st); break;
    case 16: launch_col_t<16>(a, st); break;
    case 32: launch_col_t<32>(a, st); break;
    case 64: launch_col_t<64>(a, st); break;
    default: break;
  }
}

}  // namespace spmf
