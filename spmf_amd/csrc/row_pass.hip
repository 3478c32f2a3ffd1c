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
// an entry reduce their partial dot products with xor shuffles.  The
// transcendental part (log, divide) runs once per 64 entries with one entry
// per lane: lane (grp,sub) keeps the dot product of iteration `sub`.
// col/val are read as one coalesced 256-B wave load per 64 entries and
// distributed by ds_bpermute.
//
// Roofline: HBM traffic is 8 B per stored entry per sweep (second sweep hits
// L2) + 8*KP B per row written; the factor-row gathers (2 x 4*KP B per
// entry) are served by L2/Infinity Cache because A' and V' (D*KP*4 B each)
// stay resident.  Algorithmic bytes: DESIGN.md section 4.
#include "common.h"
#include "kernels.h"

namespace spmf {

template <int KP>
__global__ __launch_bounds__(256) void row_pass_kernel(
    int64_t B, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ row_scale,
    const float* __restrict__ Ap, const float* __restrict__ Vp, const float* __restrict__ phi,
    const double* __restrict__ dprep, float* __restrict__ z, float* __restrict__ gzs,
    double* __restrict__ dacc, int encode_only) {
  constexpr int LPN = KP / 4;    // lanes per stored entry
  constexpr int NPI = 64 / LPN;  // entries per wave iteration
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float4* Ap4 = reinterpret_cast<const float4*>(Ap);
  const float4* Vp4 = reinterpret_cast<const float4*>(Vp);

  float4 veta4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!encode_only)
    veta4 = make_float4((float)dprep[sub * 4 + 0], (float)dprep[sub * 4 + 1],
                        (float)dprep[sub * 4 + 2], (float)dprep[sub * 4 + 3]);
  double ll_acc = 0.0, zsq_acc = 0.0, nnf_acc = 0.0;
  float4 zsum = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    const float xi = row_scale ? row_scale[b] : 1.f;
    // ---- sweep 1: z_b ---------------------------------------------------
    float4 zacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = start; base < end; base += 64) {
      const int idx = base + lane;
      const bool valid = idx < end;
      const int c = valid ? col[idx] : 0;
      const float x = valid ? val[idx] : 0.f;
      const int nchunk = min(64, end - base);
      const int nit = (nchunk + NPI - 1) / NPI;
      for (int it = 0; it < nit; ++it) {
        const int src = it * NPI + grp;
        const int d = __shfl(c, src);
        const float xv = __shfl(x, src);
        zacc = fma4(xv, Ap4[(size_t)d * LPN + sub], zacc);
      }
    }
#pragma unroll
    for (int m = LPN; m < 64; m <<= 1) zacc = add4(zacc, shfl_xor4(zacc, m));
    zacc.x *= xi; zacc.y *= xi; zacc.z *= xi; zacc.w *= xi;
    if (grp == 0) reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = zacc;
    if (encode_only) continue;
    if (grp == 0) {
      zsq_acc += (double)dot4(zacc, zacc);
      zsum = add4(zsum, zacc);
    }
    // ---- sweep 2: rates, log-likelihood, gz_b ---------------------------
    float4 gz = make_float4(0.f, 0.f, 0.f, 0.f);
    float llrow = 0.f;
    for (int base = start; base < end; base += 64) {
      const int idx = base + lane;
      const bool valid = idx < end;
      const int c = valid ? col[idx] : 0;
      const float x = valid ? val[idx] : 0.f;
      const int nchunk = min(64, end - base);
      const int nit = (nchunk + NPI - 1) / NPI;
      float4 vv[LPN];
      float rmine = 0.f;
#pragma unroll
      for (int it = 0; it < LPN; ++it) {
        vv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < nit) {  // wave-uniform
          const int d = __shfl(c, it * NPI + grp);
          vv[it] = Vp4[(size_t)d * LPN + sub];
          float dot = dot4(zacc, vv[it]);
#pragma unroll
          for (int m = 1; m < LPN; m <<= 1) dot += __shfl_xor(dot, m);
          if (sub == it) rmine = dot;
        }
      }
      // one entry per lane: lane (grp,sub) owns slot sub*NPI+grp
      const int slot = sub * NPI + grp;
      const float xs = __shfl(x, slot);
      const int cs = __shfl(c, slot);
      const bool sv = slot < nchunk;
      float cc = 0.f;
      if (sv && xs > 0.f) {
        const float r = rmine + phi[cs];
        if (r > 0.f && r < INFINITY) {
          llrow = fmaf(xs, logf(r), llrow);
          cc = xs / r;
        } else {
          nnf_acc += 1.0;
        }
      }
#pragma unroll
      for (int it = 0; it < LPN; ++it) {
        if (it < nit) {
          const float cb = __shfl(cc, grp * LPN + it);
          gz = fma4(cb, vv[it], gz);
        }
      }
    }
#pragma unroll
    for (int m = LPN; m < 64; m <<= 1) gz = add4(gz, shfl_xor4(gz, m));
    if (grp == 0) {
      float4 o;
      o.x = xi * (gz.x - veta4.x - zacc.x);
      o.y = xi * (gz.y - veta4.y - zacc.y);
      o.z = xi * (gz.z - veta4.z - zacc.z);
      o.w = xi * (gz.w - veta4.w - zacc.w);
      reinterpret_cast<float4*>(gzs)[(size_t)b * LPN + sub] = o;
    }
    ll_acc += (double)llrow;
  }
  if (encode_only) return;

  // ---- block reduction, one set of fp64 atomics per block ---------------
  __shared__ double red[16];
  __shared__ double zred[4][KP];
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

template <int KP>
static void launch_row_t(const RowArgs& a, hipStream_t st) {
  int64_t want = (a.B + 3) / 4;  // 4 waves (rows in flight) per 256-thread block
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(row_pass_kernel<KP>, dim3(nb), dim3(256), 0, st, a.B, a.row_ptr, a.col,
                     a.val, a.row_scale, a.Ap, a.Vp, a.phi, a.dprep, a.z, a.gzs, a.dacc,
                     a.encode_only);
}

void launch_row_pass(int KP, const RowArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_row_t<4>(a, st); break;
    case 8: launch_row_t<8>(a, st); break;
    case 16: launch_row_t<16>(a, st); break;
    case 32: launch_row_t<32>(a, st); break;
    case 64: launch_row_t<64>(a, st); break;
    default: break;
  }
}

}  // namespace spmf
