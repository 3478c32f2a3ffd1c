// stats.hip -- one streaming pre-pass over a CSR block (gfx950):
// compute_scales' column sums and per-column counts of x>0
// (mederrata_spmf/poisson.py:118-135), the row sums encode() scales by
// (:645-648) and the parameter-free per-row sums of lgamma(x+1) of the Poisson
// log-pmf (:178-183).  HBM-bound: 8 B per stored entry read once.  The column
// sums have two forms: fp64 atomics per stored entry from the CSR block, or
// (spmf_counts_colstats) one atomic per (panel, column) list of a built layout.
#include "common.h"
#include "kernels.h"

namespace spmf {

// lgamma(n + 1) of the small integer counts, once per workgroup in LDS: the values are the same
// lgamma() calls the per-entry form made (bit-identical sums), 256 per workgroup instead of
// one per stored entry (C3: the row statistics 1.84 -> see profiles/r04_layout_build.txt)
constexpr int kLgTable = 256;

__global__ __launch_bounds__(256) void stats_kernel(int64_t B, const int32_t* __restrict__ row_ptr,
                                                    const int32_t* __restrict__ col,
                                                    const float* __restrict__ val,
                                                    double* __restrict__ colsum,
                                                    double* __restrict__ colnnz,
                                                    float* __restrict__ row_sum,
                                                    double* __restrict__ row_lgamma) {
  __shared__ double lg_tab[kLgTable];
  if (row_lgamma) {
    lg_tab[threadIdx.x] = lgamma((double)threadIdx.x + 1.0);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int start = row_ptr[b], end = row_ptr[b + 1];
    double rs = 0.0, lg = 0.0;
    for (int i = start + lane; i < end; i += 64) {
      const float x = val[i];
      rs += (double)x;
      if (colsum || colnnz) {
        const int c = col[i];
        if (colsum) atomicAdd(&colsum[c], (double)x);
        if (colnnz && x > 0.f) atomicAdd(&colnnz[c], 1.0);
      }
      if (row_lgamma) {
        const int xi = (int)x;
        lg += (x >= 0.f && x < (float)kLgTable && (float)xi == x) ? lg_tab[xi] : lgamma((double)x + 1.0);
      }
    }
    rs = wave_sum(rs);
    if (lane == 0 && row_sum) row_sum[b] = (float)rs;
    if (row_lgamma) {
      lg = wave_sum(lg);
      if (lane == 0) row_lgamma[b] = lg;
    }
  }
}

void launch_stats(const StatsArgs& a, hipStream_t st) {
  int64_t want = (a.B + 3) / 4;
  int nb = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(stats_kernel, dim3(nb), dim3(256), 0, st, a.B, a.row_ptr, a.col, a.val,
                     a.colsum, a.colnnz, a.row_sum, a.row_lgamma);
}

// compute_scales' column sums from the panel-CSC lists: sixteen lanes walk one (panel, column)
// list (the lists of a wave are neighbours in memory) and add its fp64 sum to the column's slot
// once -- n_panels atomics per column instead of one per stored entry (the CSR form above:
// 1e8 fp64 atomics on C3, 6.7 of its 8.5 ms).
constexpr int kColLanes = 16;      // lanes per list

__global__ __launch_bounds__(256) void colstats_kernel(int64_t nlists, int D, const int32_t* __restrict__ pc_ptr,
                                                       const float* __restrict__ pc_val,
                                                       double* __restrict__ colsum,
                                                       double* __restrict__ colnnz) {
  const int g = threadIdx.x % kColLanes;
  const int64_t stride = (int64_t)gridDim.x * (256 / kColLanes);
  const int64_t k0 = (int64_t)blockIdx.x * (256 / kColLanes) + threadIdx.x / kColLanes;
  const int64_t rounds = (nlists + stride - 1) / stride;       // the same for every lane: the shuffles need all
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t k = k0 + r * stride;
    double s = 0.0, n = 0.0;
    int d = 0;
    bool any = false;
    if (k < nlists) {
      const int64_t p = k / D;
      d = (int)(k - p * D);
      const int32_t* pp = pc_ptr + p * (D + 1) + d;
      const int a = pp[0], e = pp[1];
      any = e > a;
      for (int i = a + g; i < e; i += kColLanes) {
        const float x = pc_val[i];
        s += (double)x;
        n += x > 0.f ? 1.0 : 0.0;
      }
    }
#pragma unroll
    for (int o = kColLanes / 2; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, kColLanes);
      n += __shfl_xor(n, o, kColLanes);
    }
    if (any && g == 0) {
      if (colsum) atomicAdd(&colsum[d], s);
      if (colnnz) atomicAdd(&colnnz[d], n);
    }
  }
}

void launch_colstats(int n_panels, int D, const int32_t* pc_ptr, const float* pc_val, double* colsum,
                     double* colnnz, hipStream_t st) {
  const int64_t nlists = (int64_t)n_panels * D;
  if (nlists <= 0) return;
  int64_t nb = (nlists + (256 / kColLanes) - 1) / (256 / kColLanes);
  if (nb > (1 << 20)) nb = 1 << 20;
  hipLaunchKernelGGL(colstats_kernel, dim3((unsigned)nb), dim3(256), 0, st, nlists, D,
                     pc_ptr, pc_val, colsum, colnnz);
}

// encoder_function of the log_transform models (poisson.py:41-42): g(x) = log(x / eta_d + 1) per
// stored entry, in CSR order and in list order -- data side, once per (batch, eta).
__global__ __launch_bounds__(256) void gval_rows_kernel(int64_t nnz, const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ col,
                                                        const float* __restrict__ val,
                                                        const float* __restrict__ eta, float* __restrict__ gval) {
  const int64_t first = row_ptr[0];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride)
    gval[first + i] = log1pf(val[first + i] / eta[col[first + i]]);
}

__global__ __launch_bounds__(256) void gval_lists_kernel(int64_t nlists, int D, const int32_t* __restrict__ pc_ptr,
                                                         const float* __restrict__ pc_val,
                                                         const float* __restrict__ eta,
                                                         float* __restrict__ pc_gval) {
  const int g = threadIdx.x % kColLanes;
  const int64_t stride = (int64_t)gridDim.x * (256 / kColLanes);
  for (int64_t k = (int64_t)blockIdx.x * (256 / kColLanes) + threadIdx.x / kColLanes; k < nlists; k += stride) {
    const int64_t p = k / D;
    const int d = (int)(k - p * D);
    const int32_t* pp = pc_ptr + p * (D + 1) + d;
    const int a = pp[0], e = pp[1];
    if (e <= a) continue;
    const float et = eta[d];
    for (int i = a + g; i < e; i += kColLanes) pc_gval[i] = log1pf(pc_val[i] / et);
  }
}

void launch_gvals(int64_t nnz, int n_panels, int D, const int32_t* row_ptr, const int32_t* col, const float* val,
                  const int32_t* pc_ptr, const float* pc_val, const float* eta, float* gval, float* pc_gval,
                  hipStream_t st) {
  if (nnz <= 0) return;
  if (gval) {
    int64_t nb = (nnz + 1023) / 1024;
    if (nb > (1 << 20)) nb = 1 << 20;
    hipLaunchKernelGGL(gval_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, nnz, row_ptr, col, val, eta, gval);
  }
  if (pc_gval) {
    const int64_t nlists = (int64_t)n_panels * D;
    int64_t nb = (nlists + (256 / kColLanes) - 1) / (256 / kColLanes);
    if (nb > (1 << 20)) nb = 1 << 20;
    hipLaunchKernelGGL(gval_lists_kernel, dim3((unsigned)nb), dim3(256), 0, st, nlists, D, pc_ptr, pc_val, eta,
                       pc_gval);
  }
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
