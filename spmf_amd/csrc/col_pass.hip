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
// Work item = (row panel p, column d), one wavefront each.  Panels are
// `panel_rows` consecutive rows so that the z / xi*gz rows a panel gathers
// (2 * panel_rows * KP * 4 B) stay L2 resident; consecutive workgroup ids go
// to panels p = 8t + (blockIdx % 8), so with the observed round-robin
// workgroup->XCD placement each XCD's L2 holds one panel at a time.  That is
// a speed heuristic only: results do not depend on placement.
//
// Per-(p,d) partial sums are folded across the wave with xor shuffles and
// leave as ONE float-atomic wave instruction covering two whole 128-B rows
// (the shape MI355X runs atomics at full rate).  Float atomics make the
// low-order bits of the gradient run-to-run dependent; parity tolerance is
// 1e-5 relative (north_star).
#include "common.h"
#include "kernels.h"

namespace spmf {

template <int KP>
__global__ __launch_bounds__(256) void col_pass_kernel(
    int D, int n_panels, int row_base, const int32_t* __restrict__ pc_ptr,
    const int32_t* __restrict__ pc_row, const float* __restrict__ pc_val,
    const float* __restrict__ Vp, const float* __restrict__ phi, const float* __restrict__ z,
    const float* __restrict__ gzs, float* __restrict__ gAp, float* __restrict__ gVp,
    float* __restrict__ gphi) {
  constexpr int LPN = KP / 4;
  constexpr int NPI = 64 / LPN;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int wid = threadIdx.x >> 6;
  // block id -> (panel, column chunk); blockIdx % 8 selects the panel residue
  const int64_t L = blockIdx.x;
  const int x = (int)(L & 7);
  const int64_t q = L >> 3;
  const int nch = (D + 3) >> 2;
  const int t = (int)(q / nch), cch = (int)(q % nch);
  const int p = 8 * t + x;
  const int d = cch * 4 + wid;
  if (p >= n_panels || d >= D) return;
  const int64_t pb = (int64_t)p * (D + 1) + d;
  const int start = pc_ptr[pb], end = pc_ptr[pb + 1];
  if (start >= end) return;

  const float4* z4 = reinterpret_cast<const float4*>(z);
  const float4* g4 = reinterpret_cast<const float4*>(gzs);
  const float4 vp = reinterpret_cast<const float4*>(Vp)[(size_t)d * LPN + sub];
  const float ph = phi[d];
  float4 gV = make_float4(0.f, 0.f, 0.f, 0.f), gA = gV;
  float gph = 0.f;
  for (int base = start; base < end; base += 64) {
    const int idx = base + lane;
    const bool valid = idx < end;
    const int rr = valid ? pc_row[idx] - row_base : 0;
    const float xx = valid ? pc_val[idx] : 0.f;
    const int nchunk = min(64, end - base);
    const int nit = (nchunk + NPI - 1) / NPI;
    for (int it = 0; it < nit; ++it) {
      const int src = it * NPI + grp;
      const int b = __shfl(rr, src);
      const float xv = __shfl(xx, src);
      const float4 zz = z4[(size_t)b * LPN + sub];
      const float4 gg = g4[(size_t)b * LPN + sub];
      float dot = dot4(zz, vp);
#pragma unroll
      for (int m = 1; m < LPN; m <<= 1) dot += __shfl_xor(dot, m);
      const float r = dot + ph;
      const float cb = (xv > 0.f && r > 0.f && r < INFINITY) ? xv / r : 0.f;
      gV = fma4(cb, zz, gV);
      gA = fma4(xv, gg, gA);
      gph += cb;
    }
  }
#pragma unroll
  for (int m = LPN; m < 64; m <<= 1) {
    gV = add4(gV, shfl_xor4(gV, m));
    gA = add4(gA, shfl_xor4(gA, m));
    gph += __shfl_xor(gph, m);
  }
  // every lane now holds the full sums of its k-slice; spread the 2*KP adds
  // over the wave: group g adds component (q&3) of (q>>2 ? gA : gV), q = g.
  constexpr int NG = NPI < 8 ? NPI : 8;
  if (grp < NG) {
#pragma unroll
    for (int q0 = 0; q0 < 8; q0 += NG) {
      const int qq = q0 + grp;
      const float4 src = (qq & 4) ? gA : gV;
      const int j = qq & 3;
      const float v = j == 0 ? src.x : (j == 1 ? src.y : (j == 2 ? src.z : src.w));
      float* dst = ((qq & 4) ? gAp : gVp) + (size_t)d * KP + sub * 4 + j;
      atomicAdd(dst, v);
    }
  }
  if (lane == 0) atomicAdd(&gphi[d], gph);
}

template <int KP>
static void launch_col_t(const ColArgs& a, hipStream_t st) {
  const int64_t nch = (a.D + 3) / 4;
  const int64_t nt = (a.n_panels + 7) / 8;
  const int64_t nb = nt * nch * 8;
  hipLaunchKernelGGL(col_pass_kernel<KP>, dim3((unsigned)nb), dim3(256), 0, st, a.D, a.n_panels,
                     a.row_base, a.pc_ptr, a.pc_row, a.pc_val, a.Vp, a.phi, a.z, a.gzs, a.gAp,
                     a.gVp, a.gphi);
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
