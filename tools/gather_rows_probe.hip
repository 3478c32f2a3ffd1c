// gather_rows_probe.hip -- where does the row pass lose against the flat gather
// ceiling?  tools/gather_ceiling.hip gathers 1e8 x 2 rows of 128 B out of two
// 2.56 MB tables in 1.25 ms (20.6 TB/s) as a FLAT stream; the row pass moves the
// same bytes in 1.9 ms.  This probe adds the row pass's structure to the flat
// kernel one piece at a time (rows of ~100 entries, CSR pointers, per-row
// reduction + 128-B store, the dependent second phase) and tries the candidate
// re-organisations, each a few lines here instead of a rewrite of row_pass.hip.
//
//   mode 0  flat stream, two tables, no rows                      (the ceiling)
//   mode 1  wave per row: all A gathers of the row, cross-group sum, store z
//   mode 2  mode 1, then the V gathers of the row (dependent on z), dots, store gz
//   mode 3  mode 2 with the V gathers issued TOGETHER with the A gathers
//   mode 4  lane group per row (8 rows per wave, entries walked 1 per group and step,
//           no cross-group reduction), two phases like mode 2
//   mode 5  mode 4 with both tables gathered in one sweep (z of the row is not
//           needed for the gathers themselves)
//
//   mode 6  mode 2 + the per-entry intercept phi[col] read from global memory, one entry
//           per lane (64 different cache lines per wave instruction), like the row pass
//   mode 7  mode 6 with phi staged in LDS once per workgroup (80 KB at D = 20 000)
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_rows_probe.hip -o /tmp/grp && /tmp/grp
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

constexpr int LPN = 8, NPI = 8;

__device__ __forceinline__ float4 gather4(const float* __restrict__ base, int row, int sub) {
  const uint32_t off = ((uint32_t)row * (uint32_t)LPN + (uint32_t)sub) * 16u;
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 xgroups(float4 v) {   // sum over the 8 lane groups
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {
    v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m);
    v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
  }
  return v;
}
__device__ __forceinline__ float ingroup(float v) {     // sum over the 8 lanes of a group
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return v;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                        const float* __restrict__ val, int64_t B, int64_t n,
                                        const float* __restrict__ tA, const float* __restrict__ tV,
                                        float* __restrict__ z, float* __restrict__ gz,
                                        const float* __restrict__ phi, int D) {
  extern __shared__ float phis[];
  if (MODE == 7) {   // (run with 256-thread blocks: 2 per CU; or 1024-thread blocks: 1 per CU, 4 waves/SIMD)
    for (int i = threadIdx.x; i < D; i += blockDim.x) phis[i] = phi[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, sub = lane % LPN, grp = lane / LPN;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (MODE == 0) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t c = wave * 64; c < n; c += nwaves * 64) {
      const int cur = c + lane < n ? col[c + lane] : 0;
#pragma unroll
      for (int g0 = 0; g0 < LPN; g0 += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = __shfl(cur, (g0 + j) * NPI + grp);
          a[j] = gather4(tA, d, sub);
          b[j] = gather4(tV, d, sub);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = add4(acc, add4(a[j], b[j]));
      }
    }
    if (acc.x == 123.456f) z[0] = acc.x;
    return;
  }
  if ((MODE >= 1 && MODE <= 3) || MODE >= 6) {
    for (int64_t b = wave; b < B; b += nwaves) {
      const int s = row_ptr[b], e = row_ptr[b + 1];
      float4 zacc = make_float4(0, 0, 0, 0), gacc = zacc;
      // rows of <= 128 entries (the generator guarantees it): two chunks in registers
      int c[2];
      float x[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = s + 64 * h + lane;
        c[h] = i < e ? col[i] : 0;
        x[h] = i < e ? val[i] : 0.f;
      }
      float4 vbuf[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nchunk = e - s - 64 * h;
#pragma unroll
        for (int g0 = 0; g0 < LPN; g0 += 4) {
          if (g0 * NPI < nchunk) {
            float4 a[4];
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int d = __shfl(c[h], (g0 + j) * NPI + grp);
              xv[j] = __shfl(x[h], (g0 + j) * NPI + grp);
              a[j] = gather4(tA, d, sub);
              if (MODE == 3) vbuf[h * 8 + g0 + j] = gather4(tV, d, sub);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) zacc = fma4(xv[j], a[j], zacc);
          }
        }
      }
      zacc = xgroups(zacc);
      if (grp == 0) reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = zacc;
      if (MODE == 1) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nchunk = e - s - 64 * h;
        float ph = 1.f;
        if (MODE == 6 && lane < nchunk) ph = phi[c[h]];        // one entry per lane
        if (MODE == 7 && lane < nchunk) ph = phis[c[h]];
#pragma unroll
        for (int g0 = 0; g0 < LPN; g0 += 4) {
          if (g0 * NPI < nchunk) {
            float4 v[4];
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              xv[j] = __shfl(x[h], (g0 + j) * NPI + grp);
              if (MODE == 3) v[j] = vbuf[h * 8 + g0 + j];
              else v[j] = gather4(tV, __shfl(c[h], (g0 + j) * NPI + grp), sub);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float pj = MODE >= 6 ? __shfl(ph, (g0 + j) * NPI + grp) : 1.f;
              const float r = ingroup(dot4(zacc, v[j])) + pj;
              gacc = fma4(xv[j] * __builtin_amdgcn_rcpf(r), v[j], gacc);
            }
          }
        }
      }
      gacc = xgroups(gacc);
      if (grp == 0) reinterpret_cast<float4*>(gz)[(size_t)b * LPN + sub] = gacc;
    }
    return;
  }
  // MODE 4 / 5: one row per lane group, 8 rows per wave
  for (int64_t b0 = wave * 8; b0 < B; b0 += nwaves * 8) {
    const int64_t b = b0 + grp;
    const bool ok = b < B;
    const int s = ok ? row_ptr[b] : 0, e = ok ? row_ptr[b + 1] : 0;
    float4 zacc = make_float4(0, 0, 0, 0), gacc = zacc;
    if (MODE == 5) {
      // single sweep: z needs the whole row before r, so accumulate z and keep the V rows?  No
      // registers for that (100 x 16 B per lane); instead gather both tables per step and
      // accumulate sum x*A and sum x*V (upper bound for a "both tables, one sweep" form)
      for (int i = s; __any(i < e); i += 8) {
        const int idx = i + sub;
        const int cc = idx < e ? col[idx] : 0;
        const float xx = idx < e ? val[idx] : 0.f;
        float4 a[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = __shfl(cc, grp * 8 + j);
          a[j] = gather4(tA, d, sub);
          v[j] = gather4(tV, d, sub);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xv = __shfl(xx, grp * 8 + j);
          zacc = fma4(xv, a[j], zacc);
          gacc = fma4(xv, v[j], gacc);
        }
      }
    } else {
      for (int i = s; __any(i < e); i += 8) {
        const int idx = i + sub;
        const int cc = idx < e ? col[idx] : 0;
        const float xx = idx < e ? val[idx] : 0.f;
        float4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = gather4(tA, __shfl(cc, grp * 8 + j), sub);
#pragma unroll
        for (int j = 0; j < 8; ++j) zacc = fma4(__shfl(xx, grp * 8 + j), a[j], zacc);
      }
      for (int i = s; __any(i < e); i += 8) {
        const int idx = i + sub;
        const int cc = idx < e ? col[idx] : 0;
        const float xx = idx < e ? val[idx] : 0.f;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gather4(tV, __shfl(cc, grp * 8 + j), sub);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float r = ingroup(dot4(zacc, v[j])) + 1.f;
          gacc = fma4(__shfl(xx, grp * 8 + j) * __builtin_amdgcn_rcpf(r), v[j], gacc);
        }
      }
    }
    if (ok) {
      reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = zacc;
      reinterpret_cast<float4*>(gz)[(size_t)b * LPN + sub] = gacc;
    }
  }
}

template <int MODE>
static float run(int blocks, const int32_t* rp, const int32_t* col, const float* val, int64_t B, int64_t n,
                 const float* tA, const float* tV, float* z, float* gz, const float* phi, int D,
                 int threads = 256) {
  const size_t lds = MODE == 7 ? (size_t)D * 4 : 0;
  if (lds) CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, rp, col, val, B, n, tA, tV, z, gz, phi, D);
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, rp, col, val, B, n, tA, tV, z, gz, phi, D);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  return ms / 5;
}

int main() {
  const int64_t B = 1000000;
  const int D = 20000;
  std::vector<int32_t> rp(B + 1), col;
  std::vector<float> val;
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  rp[0] = 0;
  col.reserve(B * 101);
  for (int64_t b = 0; b < B; ++b) {
    const int len = 90 + (int)(rnd() % 21);           // 90..110 entries, sorted columns
    int cprev = -1;
    for (int i = 0; i < len; ++i) {
      int c = (int)((rnd() % (uint64_t)(D / len)) + (int64_t)i * (D / len));
      if (c <= cprev) c = cprev + 1;
      cprev = c;
      col.push_back(c);
    }
    rp[b + 1] = (int32_t)col.size();
  }
  const int64_t n = col.size();
  val.assign(n, 1.0f);
  int32_t *d_rp, *d_col;
  float *d_val, *tab, *z, *gz;
  CHECK(hipMalloc(&d_rp, (B + 1) * 4));
  CHECK(hipMalloc(&d_col, n * 4));
  CHECK(hipMalloc(&d_val, n * 4));
  CHECK(hipMalloc(&tab, 2 * (size_t)D * 128 + (size_t)D * 4));
  CHECK(hipMalloc(&z, B * 128));
  CHECK(hipMalloc(&gz, B * 128));
  CHECK(hipMemcpy(d_rp, rp.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_col, col.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_val, val.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<float> ones((size_t)2 * D * 32 + D, 0.001f);
  CHECK(hipMemcpy(tab, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
  const float* tA = tab;
  const float* tV = tab + (size_t)D * 32;
  const float* phi = tab + (size_t)2 * D * 32;
  printf("# rows %lld, entries %lld, tables 2 x %.2f MB; ms per launch (row pass on the same shape: 1.88)\n",
         (long long)B, (long long)n, D * 128 / 1e6);
  printf("%-6s %-8s %9s %9s\n", "mode", "blocks", "ms", "TB/s");
  for (int blocks : {512, 1024, 2048, 4096}) {
    float ms[8];
    ms[0] = run<0>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[1] = run<1>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[2] = run<2>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[3] = run<3>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[4] = run<4>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[5] = run<5>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[6] = run<6>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    ms[7] = run<7>(blocks, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D);
    if (blocks == 512) {
      for (int nb : {256, 512}) {
        const float t8 = run<7>(nb, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D, 1024);
        printf("%-6s %-8d %9.3f %9.2f   (mode 7 with 1024-thread blocks)\n", "7w", nb, t8,
               (double)n * 2 * 128 / (t8 * 1e-3) / 1e12);
        const float t6 = run<6>(nb, d_rp, d_col, d_val, B, n, tA, tV, z, gz, phi, D, 1024);
        printf("%-6s %-8d %9.3f %9.2f   (mode 6 with 1024-thread blocks)\n", "6w", nb, t6,
               (double)n * 2 * 128 / (t6 * 1e-3) / 1e12);
      }
    }
    for (int m = 0; m < 8; ++m) {
      const double tables = m == 1 ? 1.0 : 2.0;
      printf("%-6d %-8d %9.3f %9.2f\n", m, blocks, ms[m], (double)n * tables * 128 / (ms[m] * 1e-3) / 1e12);
    }
  }
  return 0;
}
