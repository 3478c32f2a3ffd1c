// prep.hip -- per-draw prologue of the sparse data term (gfx950).
//
// From one draw of (u, v, w, s) and the column scales eta it forms, in the
// layouts the row/column passes gather from:
//   A'[d,k] = w1_d * u[d,k] / eta_d      encoding_matrix (poisson.py:652-666)
//                                        with g(x)=x/eta (:43) folded in
//   V'[d,k] = eta_d * v[k,d]             decoding_matrix (:668-678) with
//                                        f(y)=y*eta (:54) folded in, transposed
//   phi[d]  = eta_d * w2_d * w[d]        intercept_matrix (:680-701)
//   veta[k] = sum_d V'[d,k], phisum = sum_d phi[d]   (fp64; closed-form sum
//                                        of the rate over all B*D cells; WRITTEN,
//                                        not accumulated: dprep needs no zero fill)
// with w1 = s0/(s0+s1), w2 = s1/(s0+s1).  Rows are padded to KP floats so a
// gathered row is a whole number of 16-byte lanes.
//
// HBM-bound elementwise + a [K,D]->[D,K] transpose staged through LDS; the
// arrays are O(D*K) (2.5 MB at D=20k,K=32), i.e. noise next to the nnz
// passes.
#include "common.h"
#include "kernels.h"

namespace spmf {

constexpr int TD = 64;  // columns (features) per block

// Grid: tile blocks [0, nt) form A', V', phi for TD columns each and zero a slice of the
// step's accumulators (the zero fill used to be a launch of its own); the (KP+1)*kPrepSeg
// blocks behind them are the SUM blocks: block (j, seg) owns the partial sum of
// veta[j] = sum_d fl(eta_d v[j,d]) (row j of v is contiguous; j = KP: phisum) over column
// segment seg.  One writer per slot, fp64, fixed order: no atomics (313 blocks adding into
// the same 33 doubles made this kernel 16 of its 22 us on C3), nothing to zero first,
// bit-identical from run to run; the readers fold the segments (common.h prep_sum).
template <int KP>
__global__ __launch_bounds__(256) void prep_kernel(int D, int K, const float* __restrict__ u,
                                                   const float* __restrict__ v,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ s,
                                                   const float* __restrict__ eta,
                                                   float* __restrict__ Ap, float* __restrict__ Vp,
                                                   float* __restrict__ phi,
                                                   double* __restrict__ dprep, int logt,
                                                   const uint8_t* __restrict__ ctype,
                                                   float* __restrict__ dbias, int nt,
                                                   uint4* __restrict__ zero_p, size_t zero_n16) {
  __shared__ float tile[KP][TD + 1];
  __shared__ float w1ie[TD], etas[TD];
  __shared__ double red[16];
  const int t = threadIdx.x;
  {   // draw of this block (gridDim.y draws per launch)
    const size_t sd = blockIdx.y;
    u += sd * (size_t)D * K;
    if (v) v += sd * (size_t)K * D;
    if (w) w += sd * (size_t)D;
    s += sd * (size_t)2 * D;
    Ap += sd * (size_t)D * KP;
    Vp += sd * (size_t)D * KP;
    phi += sd * (size_t)D;
    dprep += sd * (size_t)kPrepSeg * (KP + 1);
    if (dbias) dbias += sd * (size_t)D;
  }
  if ((int)blockIdx.x >= nt) {
    // ---- sum block: one segment of one closed-form column sum, the same fp32 products the
    // tiles store ----
    const int j = (blockIdx.x - nt) / kPrepSeg, seg = (blockIdx.x - nt) % kPrepSeg;
    const int per = (D + kPrepSeg - 1) / kPrepSeg;
    const int d_lo = seg * per, d_hi = min(D, d_lo + per);
    double acc = 0.0;
    if (j < K && v) {
      const float* vr = v + (size_t)j * D;
      for (int d = d_lo + t; d < d_hi; d += 256)
        if (!(ctype && ctype[d])) {
          const float p = vr[d] * eta[d];
          acc += (double)p;
        }
    } else if (j == KP && w) {
      for (int d = d_lo + t; d < d_hi; d += 256)
        if (!(ctype && ctype[d])) {
          const float s0 = s[d], s1 = s[D + d];
          const float p = eta[d] * (s1 / (s0 + s1)) * w[d];
          acc += (double)p;
        }
    }
    const double tot = block_sum(acc, red);
    if (t == 0) dprep[(size_t)seg * (KP + 1) + j] = tot;
    return;
  }
  // ---- zero slice (acc | dacc of every draw of the step; 16-B stores) -----------------
  if (zero_p) {
    const size_t nblk = (size_t)nt * gridDim.y, bid = (size_t)blockIdx.y * nt + blockIdx.x;
    for (size_t i = bid * 256 + t; i < zero_n16; i += nblk * 256) zero_p[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int d0 = blockIdx.x * TD;
  if (t < TD) {
    const int d = d0 + t;
    float e = 1.f, a = 0.f;
    if (d < D) {
      e = eta[d];
      const float s0 = s[d], s1 = s[D + d];
      const float T = s0 + s1;
      a = logt ? (s0 / T) : (s0 / T) / e;
      const float p = w ? e * (s1 / T) * w[d] : 0.f;
      phi[d] = p;
      // mixed likelihood: the closed-form sums cover the Poisson columns only;
      // Bernoulli columns go through the dense softplus kernel, whose logit bias
      // is phi there and -1e30 (sigmoid = softplus = 0) on Poisson columns
      const bool bern = ctype && ctype[d];
      if (dbias) dbias[d] = bern ? p : -1e30f;
    }
    w1ie[t] = a;
    etas[t] = e;
  }
  // stage v[k][d0..d0+63] (coalesced along d) into LDS
  for (int e = t; e < KP * TD; e += 256) {
    const int k = e / TD, dl = e % TD;
    const int d = d0 + dl;
    tile[k][dl] = (v && k < K && d < D) ? v[(size_t)k * D + d] : 0.f;
  }
  __syncthreads();
  for (int e = t; e < KP * TD; e += 256) {
    const int dl = e / KP, k = e % KP;
    const int d = d0 + dl;
    if (d < D) {
      const float uv = (k < K) ? u[(size_t)d * K + k] : 0.f;
      Ap[(size_t)d * KP + k] = uv * w1ie[dl];
      Vp[(size_t)d * KP + k] = tile[k][dl] * etas[dl];
    }
  }
}

template <int KP>
static void launch_prep_t(const PrepArgs& a, hipStream_t st) {
  const int nt = (a.D + TD - 1) / TD;
  hipLaunchKernelGGL(prep_kernel<KP>, dim3(nt + (KP + 1) * kPrepSeg, a.S > 1 ? a.S : 1), dim3(256), 0, st, a.D, a.K, a.u, a.v,
                     a.w, a.s, a.eta, a.Ap, a.Vp, a.phi, a.dprep, a.logt, a.ctype, a.dbias, nt,
                     (uint4*)a.zero_p, a.zero_bytes / 16);
}

void launch_prep(int KP, const PrepArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_prep_t<4>(a, st); break;
    case 8: launch_prep_t<8>(a, st); break;
    case 16: launch_prep_t<16>(a, st); break;
    case 32: launch_prep_t<32>(a, st); break;
    case 64: launch_prep_t<64>(a, st); break;
    default: break;
  }
}

}  // namespace spmf
