// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access shapes of the sparse passes?
// MI355X_MICROARCH.md (HBM): "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern".  profiles/README.md (round 1) and tools/pmc_traffic.py (round 3) disagreed on
// whether the x2 applies to the 16-B-per-lane GATHERS of factor rows (8 lanes x 16 B = one 128-B row per
// stored entry).  Three kernels with a KNOWN byte count each, far beyond the 256 MB Infinity Cache:
//   stream       every lane reads 16 B, consecutive lanes consecutive addresses       bytes = table
//   stream4      every lane reads 4 B, consecutive lanes consecutive addresses        bytes = table
//   gather128    rows of 128 B read ONCE each in a random order, 8 lanes x 16 B a row  bytes = table (+ 4 B index per row)
//   gather256    rows of 256 B (K = 64), 16 lanes x 16 B a row                         bytes = table (+ index)
// Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv  (tools/fetch_calib.sh) and divide.
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ t, int64_t n4, float* __restrict__ sink) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = t[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;      // never true: keeps the loads
}

// 4 B per lane, consecutive lanes consecutive addresses (the row pass's entry stream: one 256-B wave load)
__global__ __launch_bounds__(256) void calib_stream4(const float* __restrict__ t, int64_t n, float* __restrict__ sink) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += t[i];
  if (acc == 12345.678f) sink[0] = acc;
}

// LPN lanes x 16 B per row; a wave instruction gathers 64 / LPN rows; four gathers back to back
template <int LPN>
__global__ __launch_bounds__(256) void calib_gather(const float* __restrict__ t, const int32_t* __restrict__ idx,
                                                    int64_t nrows, float* __restrict__ sink) {
  constexpr int NPI = 64 / LPN;
  const int lane = threadIdx.x & 63, sub = lane % LPN, grp = lane / LPN;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = wave * 64; base < nrows; base += nwaves * 64) {
    const int64_t i = base + lane;
    const int r = i < nrows ? idx[i] : 0;
    const int n = (int)(nrows - base < 64 ? nrows - base : 64);
#pragma unroll
    for (int g0 = 0; g0 < LPN; ++g0) {
      const int src = g0 * NPI + grp;
      const int row = __shfl(r, src);
      if (src < n) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(t) +
                                                          ((size_t)row * LPN + sub) * 16u);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30);     // table size
  float *t, *sink;
  CHECK(hipMalloc(&t, bytes));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(t, 0, bytes));
  std::mt19937_64 rng(7);
  for (int rowb : {128, 256}) {
    const int64_t nrows = (int64_t)(bytes / rowb);
    std::vector<int32_t> perm(nrows);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    int32_t* idx;
    CHECK(hipMalloc(&idx, nrows * 4));
    CHECK(hipMemcpy(idx, perm.data(), nrows * 4, hipMemcpyHostToDevice));
    if (rowb == 128) hipLaunchKernelGGL(calib_gather<8>, dim3(4096), dim3(256), 0, 0, t, idx, nrows, sink);
    else hipLaunchKernelGGL(calib_gather<16>, dim3(4096), dim3(256), 0, 0, t, idx, nrows, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_gather<%d>: rows of %d B read once each in random order: table %.0f bytes + index %.0f bytes\n",
           rowb / 16, rowb, (double)bytes, (double)nrows * 4);
    CHECK(hipFree(idx));
  }
  hipLaunchKernelGGL(calib_stream, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(t),
                     (int64_t)(bytes / 16), sink);
  CHECK(hipDeviceSynchronize());
  printf("calib_stream: 16 B per lane, coalesced: %.0f bytes\n", (double)bytes);
  hipLaunchKernelGGL(calib_stream4, dim3(8192), dim3(256), 0, 0, t, (int64_t)(bytes / 4), sink);
  CHECK(hipDeviceSynchronize());
  printf("calib_stream4: 4 B per lane, coalesced: %.0f bytes\n", (double)bytes);
  return 0;
}
