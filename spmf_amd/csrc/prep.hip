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
#include "finish_body.h"
#include "kernels.h"

namespace spmf {

// columns (features) per block: 64; 32 at K above 128 (the transpose tile is KP x (TD + 1) floats of static LDS)
template <int KP>
constexpr int prep_td() { return KP > 128 ? 32 : 64; }

// Grid: tile blocks [0, nt) form A', V', phi for TD columns each and zero a slice of the
// step's accumulators (the zero fill used to be a launch of its own); the (KP+1)*kPrepSeg
// blocks behind them are the SUM blocks: block (j, seg) owns the partial sum of
// veta[j] = sum_d fl(eta_d v[j,d]) (row j of v is contiguous; j = KP: phisum) over column
// segment seg.  One writer per slot, fp64, fixed order: no atomics (313 blocks adding into
// the same 33 doubles made this kernel 16 of its 22 us on C3), nothing to zero first,
// bit-identical from run to run; the readers fold the segments (common.h prep_sum).
// kernel arguments of the prep body (it runs inside two kernels)
struct PrepK {
  int D, K;
  const float *u, *v, *w, *s, *eta;
  float *Ap, *Vp, *phi;
  double* dprep;
  int logt;
  const uint8_t* ctype;
  float* dbias;
  int nt;
  uint4* zero_p;
  size_t zero_n16;
};

// block bx of the prep grid (nt tile blocks + (KP+1)*kPrepSeg sum blocks), draw by of nby
template <int KP>
__device__ __forceinline__ void prep_body(const PrepK& a_, const int bx, const int by, const int nby) {
  const int D = a_.D, K = a_.K;
  constexpr int TD = prep_td<KP>();
  const float* __restrict__ u = a_.u;
  const float* __restrict__ v = a_.v;
  const float* __restrict__ w = a_.w;
  const float* __restrict__ s = a_.s;
  const float* __restrict__ eta = a_.eta;
  float* __restrict__ Ap = a_.Ap;
  float* __restrict__ Vp = a_.Vp;
  float* __restrict__ phi = a_.phi;
  double* __restrict__ dprep = a_.dprep;
  const int logt = a_.logt;
  const uint8_t* __restrict__ ctype = a_.ctype;
  float* __restrict__ dbias = a_.dbias;
  const int nt = a_.nt;
  uint4* __restrict__ zero_p = a_.zero_p;
  const size_t zero_n16 = a_.zero_n16;
  __shared__ float tile[KP][TD + 1];
  __shared__ float w1ie[TD], etas[TD];
  __shared__ double red[16];
  const int t = threadIdx.x;
  {   // draw of this block (nby draws per launch)
    const size_t sd = by;
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
  if (bx >= nt) {
    // ---- sum block: one segment of one closed-form column sum, the same fp32 products the
    // tiles store ----
    const int j = (bx - nt) / kPrepSeg, seg = (bx - nt) % kPrepSeg;
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
    const size_t nblk = (size_t)nt * nby, bid = (size_t)by * nt + bx;
    for (size_t i = bid * 256 + t; i < zero_n16; i += nblk * 256) zero_p[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int d0 = bx * TD;
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
__global__ __launch_bounds__(256) void prep_kernel(const PrepK a) {
  prep_body<KP>(a, blockIdx.x, blockIdx.y, gridDim.y);
}

// The step's FIRST launch (spmf_step_begin): the prior half of the finish (finish_body.h PHASE 1: all twelve
// prior log-densities and prior_weight * d prior / d theta; it reads parameters only) in the same grid as
// the prep tiles and sums -- both are O(D*K) work on the parameters, neither reads an accumulator, so
// the prior half costs the step no launch of its own and nothing on the critical path beyond what it
// adds to this grid.  Blocks [0, nfin) are finish blocks (dispatched first: theirs is the longer chain),
// the prep grid follows.
template <int KP, bool HS>
__global__ __launch_bounds__(256) void begin_kernel(const PrepK pa, const FinishK fa, int nfin) {
  if ((int)blockIdx.x < nfin) {
    finish_body<KP, 1, HS>(fa, blockIdx.x, blockIdx.y, nfin, gridDim.y);
    return;
  }
  prep_body<KP>(pa, (int)blockIdx.x - nfin, blockIdx.y, gridDim.y);
}

template <int KP>
static PrepK make_prep_k(const PrepArgs& a) {
  constexpr int TD = prep_td<KP>();
  return PrepK{a.D, a.K, a.u, a.v, a.w, a.s, a.eta, a.Ap, a.Vp, a.phi, a.dprep, a.logt, a.ctype, a.dbias,
               (a.D + TD - 1) / TD, (uint4*)a.zero_p, a.zero_bytes / 16};
}

template <int KP>
static void launch_prep_t(const PrepArgs& a, hipStream_t st) {
  const PrepK k = make_prep_k<KP>(a);
  hipLaunchKernelGGL(prep_kernel<KP>, dim3(k.nt + (KP + 1) * kPrepSeg, a.S > 1 ? a.S : 1), dim3(256), 0, st, k);
}
template <int KP>
static void launch_begin_t(const PrepArgs& a, const FinishArgs& f, hipStream_t st) {
  const PrepK k = make_prep_k<KP>(a);
  const FinishK fk = make_finish_k(f);
  const int nfin = finish_blocks(f.D);
  const dim3 grid(nfin + k.nt + (KP + 1) * kPrepSeg, a.S > 1 ? a.S : 1);
  if (f.abs_horseshoe) hipLaunchKernelGGL((begin_kernel<KP, true>), grid, dim3(256), 0, st, k, fk, nfin);
  else hipLaunchKernelGGL((begin_kernel<KP, false>), grid, dim3(256), 0, st, k, fk, nfin);
}

void launch_prep(int KP, const PrepArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_prep_t<4>(a, st); break;
    case 8: launch_prep_t<8>(a, st); break;
    case 16: launch_prep_t<16>(a, st); break;
    case 32: launch_prep_t<32>(a, st); break;
    case 64: launch_prep_t<64>(a, st); break;
    case 128: launch_prep_t<128>(a, st); break;
    case 256: launch_prep_t<256>(a, st); break;
    default: break;
  }
}

// prep + the prior half of the finish in one launch (begin_kernel); f.S must equal a.S (or both 1)
void launch_step_begin(int KP, const PrepArgs& a, const FinishArgs& f, hipStream_t st) {
  switch (KP) {
    case 4: launch_begin_t<4>(a, f, st); break;
    case 8: launch_begin_t<8>(a, f, st); break;
    case 16: launch_begin_t<16>(a, f, st); break;
    case 32: launch_begin_t<32>(a, f, st); break;
    case 64: launch_begin_t<64>(a, f, st); break;
    case 128: launch_begin_t<128>(a, f, st); break;
    case 256: launch_begin_t<256>(a, f, st); break;
    default: break;
  }
}

}  // namespace spmf
