// stats.hip -- one streaming pre-pass over a CSR block (gfx950):
// compute_scales' column sums and per-column counts of x>0
// (mederrata_spmf/poisson.py:118-135), the row sums encode() scales by
// (:645-648) and the parameter-free per-row sums of lgamma(x+1) of the Poisson
// log-pmf (:178-183).  HBM-bound: 8 B per stored entry read once; column
// accumulators are fp64 atomics (D is small, contention is spread by the
// random column pattern).
#include "common.h"
#include "kernels.h"

namespace spmf {

__global__ __launch_bounds__(256) void stats_kernel(int64_t B, const int32_t* __restrict__ row_ptr,
                                                    const int32_t* __restrict__ col,
                                                    const float* __restrict__ val,
                                                    double* __restrict__ colsum,
                                                    double* __restrict__ colnnz,
                                                    float* __restrict__ row_sum,
                                                    double* __restrict__ row_lgamma) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    double rs = 0.0, lg = 0.0;
    for (int i = start + lane; i < end; i += 64) {
      const float x = val[i];
      const int c = col[i];
      rs += (double)x;
      if (colsum) atomicAdd(&colsum[c], (double)x);
      if (colnnz && x > 0.f) atomicAdd(&colnnz[c], 1.0);
      lg += lgamma((double)x + 1.0);
    }
    rs = wave_sum(rs);
    lg = wave_sum(lg);
    if (lane == 0 && row_sum) row_sum[b] = (float)rs;
    if (lane == 0 && row_lgamma) row_lgamma[b] = lg;
  }
}

void launch_stats(const StatsArgs& a, hipStream_t st) {
  int64_t want = (a.B + 3) / 4;
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(stats_kernel, dim3(nb), dim3(256), 0, st, a.B, a.row_ptr, a.col, a.val,
                     a.colsum, a.colnnz, a.row_sum, a.row_lgamma);
}

// Zero fill as a KERNEL.  hipMemsetAsync is avoided on the step path: captured into a
// hipGraph its memset node was observed to run out of order with the kernels of the
// PREVIOUS graph launch on the same stream (rare, timing dependent: accumulators zeroed
// while the preceding step's finish kernel still read them; gone with per-step syncs).
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = 0u;
}

void launch_zero(void* p, size_t bytes, hipStream_t st) {
  const size_t n4 = bytes / 4;          // every buffer zeroed here is float / double sized
  if (n4 == 0) return;
  size_t nb = (n4 + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)nb), dim3(256), 0, st, (uint32_t*)p, n4);
}

}  // namespace spmf
