// finish.hip -- the finish kernels: chain rule from the sparse accumulators to d/d(u,v,w,s), the
// prior parts and gradients for all 12 variables, and the 14 energy parts (gfx950).  The body is
// finish_body.h (shared with the step's first launch, prep.hip begin_kernel, which runs the PRIOR
// half beside the prep tiles); this file holds
//   finish_kernel<KP, PHASE>   the body as a launch of its own (PHASE 0 whole, 1 prior half, 2 data half)
//   finish_reduce_kernel       second stage of the prior half's cross-block sums
//   end_kernel<KP>             the step's LAST launch (spmf_step_end): the data half + that second
//                              stage in the same grid (the reduce reads what the step's first launch
//                              wrote, the data half what the passes / the all-reduce left)
//   pack_kernel                fp64 scalars -> (hi, lo) float pairs (batches without stored entries)
#include "finish_body.h"
#include "kernels.h"

namespace spmf {

template <int KP, int PHASE, bool HS>
__global__ __launch_bounds__(256) void finish_kernel(const FinishK a) {
  finish_body<KP, PHASE, HS>(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// Second stage of the prior half's cross-block sums: one wave per output (12 prior parts + K u_tau
// gradients) and draw.  Lane l adds blocks l, l+64, ... in order, then a butterfly folds the lanes: a
// fixed association, so the result does not depend on timing; the lanes only exist to have the
// loads in flight.
__device__ __forceinline__ void finish_reduce_slot(int slot, int sd, int lane, int nb, int K, int KP, int hs,
                                                   const double* __restrict__ ppart,
                                                   const float* __restrict__ putau,
                                                   double* __restrict__ parts, float* gutau,
                                                   int64_t utau_stride) {
  if (slot < 12) {
    ppart += (size_t)sd * nb * 12;
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += ppart[(size_t)b * 12 + slot];
    s = wave_sum(s);
    if (lane == 0) parts[(size_t)sd * 14 + slot] = s;
  } else if (slot < 12 + K && !hs && gutau) {
    const int k = slot - 12;
    putau += (size_t)sd * nb * KP;
    float s = 0.f;
    for (int b = lane; b < nb; b += 64) s += putau[(size_t)b * KP + k];
    s = wave_sum(s);
    if (lane == 0) gutau[(size_t)sd * utau_stride + k] = s;
  }
}
__global__ __launch_bounds__(64) void finish_reduce_kernel(int nb, int K, int KP, int hs,
                                                          const double* __restrict__ ppart,
                                                          const float* __restrict__ putau,
                                                          double* __restrict__ parts, float* gutau,
                                                          int64_t utau_stride) {
  finish_reduce_slot(blockIdx.x, blockIdx.y, threadIdx.x, nb, K, KP, hs, ppart, putau, parts, gutau, utau_stride);
}

// The step's last launch (spmf_step_end): blocks [0, nb) are the data half of the finish (PHASE 2: it adds
// the chain rule from the accumulators to what the prior half left in G, and stores parts 'z', 'x'); the
// blocks behind them fold the prior half's per-block sums, one wave per output -- those were written by
// the step's FIRST launch (prep.hip begin_kernel), so they are complete, and no data-half block reads
// what a reduce block writes (the twelve prior parts, the u_tau gradient).
template <int KP>
__global__ __launch_bounds__(256) void end_kernel(const FinishK a, int nb, int64_t utau_stride) {
  if ((int)blockIdx.x < nb) {
    finish_body<KP, 2, false>(a, blockIdx.x, blockIdx.y, nb, gridDim.y);   // (the data half has no prior code)
    return;
  }
  const int slot = ((int)blockIdx.x - nb) * 4 + (threadIdx.x >> 6);
  finish_reduce_slot(slot, blockIdx.y, threadIdx.x & 63, nb, a.K, KP, a.hs, a.ppart, a.putau, a.parts,
                     a.G.p[UTAU_], utau_stride);
}

__global__ void pack_kernel(int KP, const double* __restrict__ dacc, float* __restrict__ tail,
                            int64_t dacc_stride, int64_t acc_stride) {
  dacc += (size_t)blockIdx.y * dacc_stride;     // draw of this block
  tail += (size_t)blockIdx.y * acc_stride;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kDaccHead + KP) {
    double v = 0.0;
    for (int r = 0; r < kDaccRep; ++r) v += dacc[(size_t)r * (kDaccHead + KP) + i];
    const float hi = (float)v;
    tail[2 * i] = hi;
    tail[2 * i + 1] = (float)(v - (double)hi);
  }
}

void launch_pack(const PackArgs& a, hipStream_t st) {
  const int n = kDaccHead + a.KP;
  hipLaunchKernelGGL(pack_kernel, dim3((n + 63) / 64, a.S > 1 ? a.S : 1), dim3(64), 0, st, a.KP,
                     a.dacc, a.tail, a.dacc_stride, a.acc_stride);
}

template <int KP>
static void launch_step_end_t(const FinishArgs& a, hipStream_t st);

// phase 1: the prior half alone + the fold of its per-block sums; 2: the data half alone; 0: both, as the
// prior-half launch followed by the step's end launch (data half + fold) -- the whole finish of a caller
// that hands the outputs over late (spmf_finish without spmf_prior_async)
template <int KP>
static void launch_finish_t(const FinishArgs& a, int phase, hipStream_t st) {
  const FinishK k = make_finish_k(a);
  const int nb = finish_blocks(a.D);
  const dim3 grid(nb, a.S > 1 ? a.S : 1);
  const bool hs = a.abs_horseshoe != 0;
  if (phase == 2) {
    hipLaunchKernelGGL((finish_kernel<KP, 2, false>), grid, dim3(256), 0, st, k);
    return;
  }
  if (hs) hipLaunchKernelGGL((finish_kernel<KP, 1, true>), grid, dim3(256), 0, st, k);
  else hipLaunchKernelGGL((finish_kernel<KP, 1, false>), grid, dim3(256), 0, st, k);
  if (phase == 0) launch_step_end_t<KP>(a, st);
  else   // the prior half produced per-block partials: add them up in block order
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(12 + a.K, a.S > 1 ? a.S : 1), dim3(64), 0, st, nb, a.K, KP,
                       a.abs_horseshoe, a.ppart, a.putau, a.parts, a.grads[4], a.vstride[4]);
}

// phase 0: whole finish; 1: prior half (no accumulators read); 2: data half (adds to G)
void launch_finish(int KP, const FinishArgs& a, int phase, hipStream_t st) {
  switch (KP) {
    case 4: launch_finish_t<4>(a, phase, st); break;
    case 8: launch_finish_t<8>(a, phase, st); break;
    case 16: launch_finish_t<16>(a, phase, st); break;
    case 32: launch_finish_t<32>(a, phase, st); break;
    case 64: launch_finish_t<64>(a, phase, st); break;
    case 128: launch_finish_t<128>(a, phase, st); break;
    case 256: launch_finish_t<256>(a, phase, st); break;
    default: break;
  }
}

// the step's last launch: data half + the fold of the prior half's per-block sums (end_kernel)
template <int KP>
static void launch_step_end_t(const FinishArgs& a, hipStream_t st) {
  const FinishK k = make_finish_k(a);
  const int nb = finish_blocks(a.D);
  const int nred = (12 + a.K + 3) / 4;
  hipLaunchKernelGGL((end_kernel<KP>), dim3(nb + nred, a.S > 1 ? a.S : 1), dim3(256), 0, st, k, nb, a.vstride[4]);
}
void launch_step_end(int KP, const FinishArgs& a, hipStream_t st) {
  switch (KP) {
    case 4: launch_step_end_t<4>(a, st); break;
    case 8: launch_step_end_t<8>(a, st); break;
    case 16: launch_step_end_t<16>(a, st); break;
    case 32: launch_step_end_t<32>(a, st); break;
    case 64: launch_step_end_t<64>(a, st); break;
    case 128: launch_step_end_t<128>(a, st); break;
    case 256: launch_step_end_t<256>(a, st); break;
    default: break;
  }
}

}  // namespace spmf
