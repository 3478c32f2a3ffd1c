// gather_ceiling.hip -- what the chip delivers for THE access pattern of the
// sparse passes: rows of 128 B / 256 B gathered by a random index INTO
// REGISTERS out of a table of 1 - 8 MB (A' / V' are 2.56 MB each on C3, the
// pair 5.1 MB; a z / xi*gz panel pair is 2 MB), with the index stream read
// coalesced like col_idx.  The kernel is the row pass's sweep 1 with the
// arithmetic reduced to one add per gathered float4 and nothing else: LPN lanes
// x float4 per row, 64/LPN rows per wave instruction, GRP gathers issued back to
// back, 32-bit saddr offsets.  Whatever this measures is the ceiling of any
// "one table row per stored entry out of L2" formulation;
// the guide's figures (MI355X_MICROARCH.md: L2 ~34.5 TB/s, 1152-B rows into LDS
// 16.8-18.8 TB/s "lower bound") are for other shapes.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_ceiling.hip -o /tmp/gather_ceiling
//   /tmp/gather_ceiling            (prints one line per configuration)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <int LPN>
__device__ __forceinline__ float4 gather4(const float* __restrict__ base, int row, int sub) {
  const uint32_t off = ((uint32_t)row * (uint32_t)LPN + (uint32_t)sub) * 16u;
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}

// TABLES = 1: one table (sweep-1 like).  TABLES = 2: two tables gathered per entry
// (row pass: A' and V'; column pass: z and xi*gz).
template <int LPN, int GRP, int TABLES>
__global__ __launch_bounds__(256) void gather_kernel(const int32_t* __restrict__ idx, int64_t n,
                                                          const float* __restrict__ t0,
                                                          const float* __restrict__ t1,
                                                          float* __restrict__ out) {
  constexpr int NPI = 64 / LPN;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPN, grp = lane / LPN;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // each wave walks chunks of 64 indices, grid-strided; next chunk prefetched
  int64_t c = wave * 64;
  int cur = c + lane < n ? idx[c + lane] : 0;
  for (; c < n; c += nwaves * 64) {
    const int64_t cn = c + nwaves * 64;
    const int nxt = cn + lane < n ? idx[cn + lane] : 0;
#pragma unroll
    for (int g0 = 0; g0 < LPN; g0 += GRP) {
      float4 a[GRP], b[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int d = __shfl(cur, (g0 + j) * NPI + grp);
        a[j] = gather4<LPN>(t0, d, sub);
        if (TABLES == 2) b[j] = gather4<LPN>(t1, d, sub);
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        acc.x += a[j].x; acc.y += a[j].y; acc.z += a[j].z; acc.w += a[j].w;
        if (TABLES == 2) { acc.x += b[j].x; acc.y += b[j].y; acc.z += b[j].z; acc.w += b[j].w; }
      }
    }
    cur = nxt;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;   // keep the loads alive
}

struct Cfg {
  const char* name;
  int lpn, grp, tables, wps;
};

template <int LPN, int GRP, int TABLES>
static float run(const int32_t* idx, int64_t n, const float* t0, const float* t1, float* out,
                 int blocks, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((gather_kernel<LPN, GRP, TABLES>), dim3(blocks), dim3(256), 0, 0, idx, n, t0,
                       t1, out);
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((gather_kernel<LPN, GRP, TABLES>), dim3(blocks), dim3(256), 0, 0, idx, n, t0,
                       t1, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000LL;   // entries (C3: 1e8)
  CHECK(hipSetDevice(0));
  int32_t* idx;
  float *tab, *out;
  const size_t max_tab = 2 * (16u << 20);
  CHECK(hipMalloc(&idx, n * 4));
  CHECK(hipMalloc(&tab, max_tab));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(tab, 0, max_tab));
  std::vector<int32_t> h(n);
  printf("# entries %lld; time per launch, gathered TB/s = entries * tables * row bytes / time\n",
         (long long)n);
  printf("%-10s %-6s %-7s %-4s %-7s %9s %9s\n", "table_MB", "rowB", "tables", "grp", "blocks", "ms",
         "TB/s");
  // argv[2]: comma-separated list of TOTAL resident MB (default: the round-2 sweep);
  // 0.016 / 0.064 = L1-resident / just past L1: what the per-CU vector cache alone delivers
  std::vector<double> table_mb = {1.0, 2.56, 4.0, 5.12, 8.0};
  if (argc > 2) {
    table_mb.clear();
    for (char* t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) table_mb.push_back(atof(t));
  }
  for (double mb : table_mb) {
    for (int rowb : {128, 256}) {
      for (int tables : {1, 2}) {
        // `mb` is the TOTAL resident bytes: with two tables each holds mb/2
        const int64_t rows = (int64_t)(mb * 1e6 / tables / rowb);
        uint64_t s = 88172645463325252ull;
        for (int64_t i = 0; i < n; ++i) {
          s ^= s << 13; s ^= s >> 7; s ^= s << 17;
          h[i] = (int32_t)(s % (uint64_t)rows);
        }
        CHECK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        const float* t0 = tab;
        const float* t1 = tab + (size_t)rows * rowb / 4;
        {
          // 256-thread blocks: 256 blocks = 1 wave per SIMD, 2048 = 8 (the register-light
          // kernel is resident in full; the row pass itself holds ~2-4 waves per SIMD)
          for (int blocks : {256, 512, 1024, 2048}) {
            for (int grp : {4, 8}) {
              float ms = -1.f;
#define CASE(L_, G_, T_)                                              \
  if (rowb == L_ * 16 && grp == G_ && tables == T_)                       \
    ms = run<L_, G_, T_>(idx, n, t0, t1, out, blocks, 5);
              CASE(8, 4, 1) CASE(8, 8, 1) CASE(8, 4, 2) CASE(8, 8, 2)
              CASE(16, 4, 1) CASE(16, 8, 1) CASE(16, 4, 2) CASE(16, 8, 2)
#undef CASE
              if (ms < 0) continue;
              const double tbs = (double)n * tables * rowb / (ms * 1e-3) / 1e12;
              printf("%-10.2f %-6d %-7d %-4d %-7d %9.3f %9.2f\n", mb, rowb, tables, grp, blocks, ms, tbs);
              fflush(stdout);
            }
          }
        }
      }
    }
  }
  return 0;
}
