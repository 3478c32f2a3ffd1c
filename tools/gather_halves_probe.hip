// gather_halves_probe.hip -- does it matter HOW a wave asks for a 128-byte table row?  (round 5)
//
// Counters of the sparse passes (profiles/r05_pmc_sparse_passes.txt): ~2.3 vector-cache tag accesses per L2
// request -- a 128-B row fetched by 8 lanes x 16 B is two 64-B accesses of ONE line, issued back to back -- and
// the vector cache "pending"-stalled for 34 - 38 % of its busy cycles, at a request rate (0.24 - 0.31 per clock
// and CU) that does not move with where the line is served from (L2 at 177 clk or Infinity Cache: same rate).
// Hypothesis: the second half of a row stalls on the first half's miss, which is still pending.  Test: the flat
// gather of tools/gather_ceiling.hip with the two halves of a row asked for by SEPARATE instructions:
//   FULL   8 lanes x float4 per row, 8 rows per wave instruction                       (what the passes do)
//   HALF   4 lanes per row, 16 rows per wave instruction: all first halves (k 0..15), then all second halves
//   HALFG  as HALF, but G first-half instructions before their G second-half instructions (distance in time)
//   QUART  2 lanes per row, 32 rows per instruction, four instructions per row
// Two tables of 2.56 MB (the row pass's A', V'), i.i.d. indices, 1e8 entries; ms per launch and lines / s.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_halves_probe.hip -o tools/bin/gather_halves_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ float4 ld16(const float* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void acc4(float4& a, const float4& v) {
  a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}

// LPR lanes per row (8: FULL, 4: HALF, 2: QUART); PIECES = 8 / LPR instructions per row; G rows' pieces grouped
template <int LPR, int G, int TABLES>
__global__ __launch_bounds__(256) void gather_kernel(const int32_t* __restrict__ idx, int64_t n,
                                                      const float* __restrict__ t0, const float* __restrict__ t1,
                                                      float* __restrict__ out) {
  constexpr int RPI = 64 / LPR;        // rows per wave instruction
  constexpr int PIECES = 8 / LPR;      // 16-B pieces of a row per lane
  constexpr int STEPS = 64 / RPI;      // instructions (per piece) that cover a chunk of 64 indices
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t c = wave * 64;
  int cur = c + lane < n ? idx[c + lane] : 0;
  for (; c < n; c += nwaves * 64) {
    const int64_t cn = c + nwaves * 64;
    const int nxt = cn + lane < n ? idx[cn + lane] : 0;
#pragma unroll
    for (int s0 = 0; s0 < STEPS; s0 += G) {
      float4 a[G][PIECES], b[G][PIECES];
      uint32_t off[G];
#pragma unroll
      for (int j = 0; j < G; ++j) off[j] = (uint32_t)__shfl(cur, (s0 + j) * RPI + grp) * 128u + (uint32_t)sub * 16u;
      // piece p of every row of the group, then the next piece: the pieces of ONE row are G instructions apart
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          a[j][p] = ld16(t0, off[j] + (uint32_t)p * (uint32_t)LPR * 16u);
          if (TABLES == 2) b[j][p] = ld16(t1, off[j] + (uint32_t)p * (uint32_t)LPR * 16u);
        }
      }
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
          acc4(acc, a[j][p]);
          if (TABLES == 2) acc4(acc, b[j][p]);
        }
    }
    cur = nxt;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int LPR, int G, int TABLES>
static void run(const char* name, const int32_t* idx, int64_t n, const float* t0, const float* t1, float* out,
                int blocks, int rows) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 5;
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((gather_kernel<LPR, G, TABLES>), dim3(blocks), dim3(256), 0, 0, idx, n, t0, t1, out);
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((gather_kernel<LPR, G, TABLES>), dim3(blocks), dim3(256), 0, 0, idx, n, t0, t1, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  ms /= reps;
  printf("%-8s lanes/row %d  group %d  tables %d  table rows %7d  blocks %5d   %8.4f ms  %7.2f G lines/s  %6.2f TB/s\n", name,
         LPR, G, TABLES, rows, blocks, ms, (double)n * TABLES / (ms * 1e-3) / 1e9, (double)n * TABLES * 128 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000LL;
  CHECK(hipSetDevice(0));
  int32_t* idx;
  float *tab, *out;
  const int rows = 20000;                                  // 2.56 MB per table
  CHECK(hipMalloc(&idx, n * 4));
  CHECK(hipMalloc(&tab, (size_t)2 * rows * 128));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(tab, 0, (size_t)2 * rows * 128));
  std::vector<int32_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (int32_t)(s % (uint64_t)rows);
  }
  CHECK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  const float* t0 = tab;
  const float* t1 = tab + (size_t)rows * 32;
  for (int blocks : {1024, 2048}) {                        // 4 and 8 waves per SIMD
    run<8, 4, 2>("FULL", idx, n, t0, t1, out, blocks, rows);
    run<8, 8, 2>("FULL", idx, n, t0, t1, out, blocks, rows);
    run<4, 1, 2>("HALF", idx, n, t0, t1, out, blocks, rows);
    run<4, 2, 2>("HALFG", idx, n, t0, t1, out, blocks, rows);
    run<4, 4, 2>("HALFG", idx, n, t0, t1, out, blocks, rows);
    run<2, 1, 2>("QUART", idx, n, t0, t1, out, blocks, rows);
    run<2, 2, 2>("QUART", idx, n, t0, t1, out, blocks, rows);
    run<8, 4, 1>("FULL", idx, n, t0, t1, out, blocks, rows);
    run<4, 2, 1>("HALFG", idx, n, t0, t1, out, blocks, rows);
    run<4, 4, 1>("HALFG", idx, n, t0, t1, out, blocks, rows);
  }
  return 0;
}
