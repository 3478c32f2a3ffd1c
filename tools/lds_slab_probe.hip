// lds_slab_probe.hip -- ONE product of the sparse step (sweep 1 of the row pass, z = X A') in the
// formulation north_star names ("LDS-staged tiles of the dense factor"), measured instead of sized
// on paper:
//
//   row block x column slab: a 1024-thread workgroup owns RB = 2048 rows; their z accumulators stay in
//   REGISTERS for the whole kernel (lane group g of wave w owns 16 rows: 16 float4 per lane); A' is
//   streamed through LDS in slabs of SC = 1024 columns (128 KB, the whole table once per workgroup);
//   every gather is a ds_read_b128 out of the slab instead of a 128-B line through the vector cache.
//
// What that costs, and the probe pays all of it: at 0.5 % density a (row, slab) cell holds 5 entries,
// the 8 lane groups of a wave walk their rows' cells in lock step (register indices are static, so all
// groups are on the same row slot) and every slot runs to the longest of the 8 lists.  The entry stream
// is materialised on the host in exactly that lock-step order -- (row block, slab, wave, slot, step,
// group), padded with weightless entries -- so the kernel has no divergence and reads it as one
// coalesced 256-B load per 8 steps; trip counts per (row block, slab, wave, slot) are wave-uniform bytes.
//
// Reference in the same program: the present formulation of the same product (wave per row, 8 entries
// per wave instruction gathered from the L2-resident table, cross-group sum, 128-B store) on the same
// matrix -- the encode-only launch of row_pass.hip in miniature (tools/gather_rows_probe.hip mode 1).
//
//   hipcc -O3 --offload-arch=gfx950 tools/lds_slab_probe.hip -o /tmp/lsp && /tmp/lsp [rows]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

constexpr int KP = 32, LPN = 8;
constexpr int RB = 2048;        // rows per workgroup
constexpr int SC = 1024;        // columns per slab (128 KB of LDS)
constexpr int NW = 16;          // waves per workgroup
constexpr int NSLOT = 16;       // rows per lane group

__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// ---- reference: wave per row, gathers out of the L2-resident table ----------------------------------
__global__ __launch_bounds__(256) void ref_sweep1(int64_t B, const int32_t* __restrict__ row_ptr,
                                                  const uint32_t* __restrict__ ent, const float* __restrict__ Ap,
                                                  float* __restrict__ z) {
  const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int s0 = row_ptr[b], s1 = row_ptr[b + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = s0; base < s1; base += 64) {
      const int i = base + lane;
      const uint32_t w = i < s1 ? ent[i] : 0u;
      const int n = min(64, s1 - base);
#pragma unroll
      for (int g0 = 0; g0 < 8; g0 += 4) {
        if (g0 * 8 < n) {
          float4 a[4];
          float xv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t wj = __shfl(w, (g0 + j) * 8 + grp);
            xv[j] = (float)(wj & 0xffffu);
            a[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Ap) + ((wj >> 16) * 8u + sub) * 16u);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = fma4(xv[j], a[j], acc);
        }
      }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
      acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m);
      acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
    }
    if (grp == 0) reinterpret_cast<float4*>(z)[(size_t)b * LPN + sub] = acc;
  }
}

// ---- the slab formulation ------------------------------------------------------------------------------
// stream: per (row block, slab, wave) one run of lock-step steps, 8 words per step (word g of a step = the
//   entry of lane group g: column inside the slab << 16 | count; padding words are 0: column 0, weight 0)
// run_off[(rb * nslab + s) * NW + w]: first word of the run (a multiple of 64); trips[... * NSLOT + j]: steps
//   of row slot j in that run (the longest of the 8 groups' lists)
__global__ __launch_bounds__(NW * 64) void slab_sweep1(int nslab, int D, const uint32_t* __restrict__ stream,
                                                       const int64_t* __restrict__ run_off,
                                                       const uint8_t* __restrict__ trips,
                                                       const float* __restrict__ Ap, float* __restrict__ z) {
  extern __shared__ __attribute__((aligned(16))) float slab[];        // [SC][KP]
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int sub = lane & 7, grp = lane >> 3;
  const int rb = blockIdx.x;
  float4 acc[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t sub16 = (uint32_t)sub * 16u;
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();                                     // every wave is done with the previous slab
    {
      const int c0 = s * SC;
      const float4* src = reinterpret_cast<const float4*>(Ap) + (size_t)c0 * LPN;
      float4* dst = reinterpret_cast<float4*>(slab);
      const int nf4 = min(SC, D - c0) * LPN;
#pragma unroll
      for (int i = 0; i < SC * LPN / (NW * 64); ++i) {
        const int idx = i * NW * 64 + t;
        if (idx < nf4) dst[idx] = src[idx];
      }
    }
    __syncthreads();
    const size_t rid = ((size_t)rb * nslab + s) * NW + wid;
    const uint32_t* run = stream + run_off[rid];
    // wave-uniform trip counts of the 16 row slots (one 16-byte scalar-ish load)
    const uint4 tr = *reinterpret_cast<const uint4*>(trips + rid * NSLOT);
    const uint32_t trw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)tr.x),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.z),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.w)};
    uint32_t ebuf = run[lane];                           // steps 0..7
    uint32_t enext = run[64 + lane];                     // (every run carries one chunk of slack)
    int step = 0;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int n = (int)((trw[j >> 2] >> (8 * (j & 3))) & 0xffu);
      for (int i = 0; i < n; ++i) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((((step & 7) << 3) + grp) << 2, (int)ebuf);
        const float x = (float)(w & 0xffffu);
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slab) + ((w >> 16) << 7) + sub16);
        acc[j] = fma4(x, a, acc[j]);
        ++step;
        if ((step & 7) == 0) {                           // wave-uniform: next chunk of 8 steps
          ebuf = enext;
          enext = run[(size_t)(step + 8) * 8 + lane];
        }
      }
    }
  }
  const size_t row0 = (size_t)rb * RB + (size_t)wid * (RB / NW) + (size_t)grp * NSLOT;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) reinterpret_cast<float4*>(z)[(row0 + j) * LPN + sub] = acc[j];
}

// The same kernel with the per-step dependent chain (entry word -> slab row -> fma) software-pipelined two
// deep: the word of step t + 2 and the slab row of step t + 1 are in flight while step t is accumulated
// (steps are consecutive inside a run whatever slot they belong to, and a word past the end of the run is
// another run's first word or slack: a valid column of the slab either way).
__global__ __launch_bounds__(NW * 64) void slab_sweep1_pipe(int nslab, int D, const uint32_t* __restrict__ stream,
                                                            const int64_t* __restrict__ run_off,
                                                            const uint8_t* __restrict__ trips,
                                                            const float* __restrict__ Ap, float* __restrict__ z) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int sub = lane & 7, grp = lane >> 3;
  const int rb = blockIdx.x;
  float4 acc[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t sub16 = (uint32_t)sub * 16u;
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();
    {
      const int c0 = s * SC;
      const float4* src = reinterpret_cast<const float4*>(Ap) + (size_t)c0 * LPN;
      float4* dst = reinterpret_cast<float4*>(slab);
      const int nf4 = min(SC, D - c0) * LPN;
#pragma unroll
      for (int i = 0; i < SC * LPN / (NW * 64); ++i) {
        const int idx = i * NW * 64 + t;
        if (idx < nf4) dst[idx] = src[idx];
      }
    }
    __syncthreads();
    const size_t rid = ((size_t)rb * nslab + s) * NW + wid;
    const uint32_t* run = stream + run_off[rid];
    const uint4 tr = *reinterpret_cast<const uint4*>(trips + rid * NSLOT);
    const uint32_t trw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)tr.x),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.z),
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)tr.w)};
    uint32_t ebuf = run[lane], enext = run[64 + lane];
    // word(q): the entry of this lane's group at step q, q in [8 c, 8 c + 16) for the current chunk c
    auto word = [&](int q, int chunk0) {
      const uint32_t src = (q >> 3) == chunk0 ? ebuf : enext;
      return (uint32_t)__builtin_amdgcn_ds_bpermute((((q & 7) << 3) + grp) << 2, (int)src);
    };
    auto rowof = [&](uint32_t w) {
      return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slab) + ((w >> 16) << 7) + sub16);
    };
    int step = 0, chunk0 = 0;
    uint32_t w0 = word(0, 0), w1 = word(1, 0);
    float4 a0 = rowof(w0);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int n = (int)((trw[j >> 2] >> (8 * (j & 3))) & 0xffu);
      for (int i = 0; i < n; ++i) {
        const uint32_t w2 = word(step + 2, chunk0);      // two steps ahead (this chunk or the prefetched one)
        const float4 a1 = rowof(w1);                     // one step ahead
        acc[j] = fma4((float)(w0 & 0xffffu), a0, acc[j]);
        w0 = w1; w1 = w2; a0 = a1;
        ++step;
        if ((step & 7) == 0) {                           // wave-uniform: the next chunk becomes the current one
          ebuf = enext;
          enext = run[(size_t)(step + 8) * 8 + lane];
          chunk0 = step >> 3;
        }
      }
    }
  }
  const size_t row0 = (size_t)rb * RB + (size_t)wid * (RB / NW) + (size_t)grp * NSLOT;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) reinterpret_cast<float4*>(z)[(row0 + j) * LPN + sub] = acc[j];
}

int main(int argc, char** argv) {
  int64_t B = argc > 1 ? atoll(argv[1]) : 1000000;
  B = (B + RB - 1) / RB * RB;
  const int D = 20000, mean_len = 100;
  const int nslab = (D + SC - 1) / SC;
  const int64_t nrb = B / RB;
  printf("rows %lld (row blocks %lld), D %d, slabs %d x %d columns, ~%d entries per row\n", (long long)B,
         (long long)nrb, D, nslab, SC, mean_len);
  // ---- the matrix: Poisson(100) distinct uniform columns per row, counts 1..4 --------------------------
  std::mt19937_64 rng(20241218);
  std::poisson_distribution<int> plen(mean_len);
  std::vector<int32_t> row_ptr(B + 1, 0);
  std::vector<uint32_t> ent;
  ent.reserve((size_t)B * (mean_len + 2));
  std::vector<uint32_t> cols;
  for (int64_t b = 0; b < B; ++b) {
    const int n = plen(rng);
    cols.clear();
    for (int i = 0; i < n; ++i) cols.push_back((uint32_t)(rng() % D));
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (uint32_t c : cols) ent.push_back(c << 16 | (uint32_t)(1 + (rng() & 3)));
    row_ptr[b + 1] = (int32_t)ent.size();
  }
  const int64_t nnz = (int64_t)ent.size();
  // ---- the lock-step stream ----------------------------------------------------------------------------
  std::vector<uint32_t> stream;
  stream.reserve((size_t)nnz * 2);
  std::vector<int64_t> run_off((size_t)nrb * nslab * NW);
  std::vector<uint8_t> trips((size_t)nrb * nslab * NW * NSLOT);
  int64_t steps_total = 0;
  {
    // per row: cursor into its (sorted) entries, advanced slab by slab
    std::vector<int32_t> cur(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t rb = 0; rb < nrb; ++rb)
      for (int s = 0; s < nslab; ++s)
        for (int w = 0; w < NW; ++w) {
          const size_t rid = ((size_t)rb * nslab + s) * NW + w;
          run_off[rid] = (int64_t)stream.size();
          const uint32_t chi = (uint32_t)std::min(D, (s + 1) * SC);
          for (int j = 0; j < NSLOT; ++j) {
            int cnt[8], mx = 0;
            int32_t st[8];
            for (int g = 0; g < 8; ++g) {
              const int64_t row = rb * RB + (int64_t)w * (RB / NW) + g * NSLOT + j;
              int32_t c = cur[row];
              st[g] = c;
              while (c < row_ptr[row + 1] && (ent[c] >> 16) < chi) ++c;
              cnt[g] = c - st[g];
              cur[row] = c;
              mx = std::max(mx, cnt[g]);
            }
            if (mx > 255) { fprintf(stderr, "trip count overflow\n"); return 1; }
            trips[rid * NSLOT + j] = (uint8_t)mx;
            steps_total += mx;
            for (int i = 0; i < mx; ++i)
              for (int g = 0; g < 8; ++g) {
                uint32_t wv = 0;
                if (i < cnt[g]) {
                  const uint32_t e = ent[st[g] + i];
                  wv = ((e >> 16) - (uint32_t)(s * SC)) << 16 | (e & 0xffffu);
                }
                stream.push_back(wv);
              }
          }
          // pad the run to whole 64-word chunks plus one chunk of slack (the kernel prefetches one ahead)
          while (stream.size() % 64) stream.push_back(0);
          for (int i = 0; i < 64; ++i) stream.push_back(0);
        }
  }
  for (int i = 0; i < 128; ++i) stream.push_back(0);
  printf("nnz %lld; lock-step slots %lld (8 per step): %.1f %% carry an entry; stream %.2f GB against %.2f GB packed\n",
         (long long)nnz, (long long)steps_total * 8, 100.0 * nnz / (steps_total * 8.0), stream.size() * 4e-9, nnz * 4e-9);
  // ---- device --------------------------------------------------------------------------------------------
  std::vector<float> hA((size_t)D * KP);
  for (auto& v : hA) v = (float)((rng() >> 40) * (1.0 / (1 << 24)));
  int32_t* d_rp; uint32_t *d_ent, *d_stream; int64_t* d_off; uint8_t* d_trips; float *d_A, *d_z0, *d_z1;
  CHECK(hipMalloc(&d_rp, (B + 1) * 4));
  CHECK(hipMalloc(&d_ent, nnz * 4));
  CHECK(hipMalloc(&d_stream, stream.size() * 4));
  CHECK(hipMalloc(&d_off, run_off.size() * 8));
  CHECK(hipMalloc(&d_trips, trips.size()));
  CHECK(hipMalloc(&d_A, hA.size() * 4));
  CHECK(hipMalloc(&d_z0, (size_t)B * KP * 4));
  CHECK(hipMalloc(&d_z1, (size_t)B * KP * 4));
  CHECK(hipMemcpy(d_rp, row_ptr.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_ent, ent.data(), nnz * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_stream, stream.data(), stream.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_off, run_off.data(), run_off.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_trips, trips.data(), trips.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)SC * KP * 4;
  CHECK(hipFuncSetAttribute((const void*)slab_sweep1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-46s %8.4f ms   %7.2f G entries/s\n", name, ms, nnz / (ms * 1e6));
    return ms;
  };
  const float t_ref = timeit("reference: wave per row, gathers from L2", [&] {
    hipLaunchKernelGGL(ref_sweep1, dim3(4096), dim3(256), 0, 0, B, d_rp, d_ent, d_A, d_z0);
  });
  const float t_slab = timeit("slab: row block x LDS column slab, lock step", [&] {
    hipLaunchKernelGGL(slab_sweep1, dim3((unsigned)nrb), dim3(NW * 64), lds, 0, nslab, D, d_stream, d_off, d_trips,
                       d_A, d_z1);
  });
  CHECK(hipFuncSetAttribute((const void*)slab_sweep1_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float t_pipe = timeit("slab, entry -> row -> fma pipelined two deep", [&] {
    hipLaunchKernelGGL(slab_sweep1_pipe, dim3((unsigned)nrb), dim3(NW * 64), lds, 0, nslab, D, d_stream, d_off,
                       d_trips, d_A, d_z1);
  });
  printf("slab (pipelined) / reference = %.3f  (speed-up %.2f x)\n", t_pipe / t_ref, t_ref / t_pipe);
  // ---- same numbers? ---------------------------------------------------------------------------------------
  std::vector<float> z0((size_t)B * KP), z1((size_t)B * KP);
  CHECK(hipMemcpy(z0.data(), d_z0, z0.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(z1.data(), d_z1, z1.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, big = 0.0;
  for (size_t i = 0; i < z0.size(); ++i) {
    worst = std::max(worst, (double)fabsf(z0[i] - z1[i]));
    big = std::max(big, (double)fabsf(z0[i]));
  }
  printf("max |z_slab - z_ref| = %.3e (max |z| %.3e)\n", worst, big);
  printf("slab / reference = %.3f  (speed-up %.2f x)\n", t_slab / t_ref, t_ref / t_slab);
  return worst <= 1e-4 * big ? 0 : 2;
}
