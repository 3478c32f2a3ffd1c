// col_owner_probe.hip -- the column pass WITHOUT global float atomics, measured:
// "persistent workgroups that own their columns for every panel of their XCD's residue class and keep
// gV' / gA' / gphi in registers until the end, one plain store".
//
//   owner form: the grid is exactly the resident set; workgroup b serves the panels of residue class
//   b % 8 (one class per XCD, as the product's column pass maps them); inside a class every lane group
//   owns NC columns (sorted by stored count and dealt round-robin, so the 8 groups of a wave walk lists of
//   similar length) and walks  for panel in class: for slot in 0..NC-1: list(panel, column[slot])  with
//   the slot's accumulators (gV' 4 + gA' 4 + gphi 1 registers per lane) selected statically.  A class
//   writes its columns' sums once (plain stores, one owner per (class, column): deterministic); a small
//   kernel adds the 8 class partials.
//
// Reference in the same program: THE PRODUCT'S kernel (spmf_amd/csrc/col_pass.hip is compiled into this
// file: col_pass_kernel<32, 0, 4, packed>, items sorted by length per panel, float-atomic flush) on the
// same matrix, same z / xi*gz / V' / phi.
//
//   hipcc -O3 --offload-arch=gfx950 -Ispmf_amd/csrc -Iinclude tools/col_owner_probe.hip -o tools/bin/col_owner_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#include "../spmf_amd/csrc/col_pass.hip"

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

using namespace spmf;

constexpr int KP = 32, LPN = 8, NG = 8;

// NC: column slots per lane group; WPS: waves per SIMD the kernel is compiled for
template <int NC, int WPS>
__global__ __launch_bounds__(256, WPS) void col_owner_kernel(
    int D, int n_panels, int panel_rows, int groups_per_class, const int32_t* __restrict__ col_sorted,
    const int32_t* __restrict__ pc_ptr, const uint32_t* __restrict__ pc_ent, const float* __restrict__ Vp,
    const float* __restrict__ phi, const float* __restrict__ z, const float* __restrict__ gzs,
    float* __restrict__ partial /* [8][D][2*KP+1] */) {
  const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3, wid = threadIdx.x >> 6;
  const int cls = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int gi = (q * 4 + wid) * NG + grp;              // lane group inside the class
  int dcol[NC];
  float4 vp[NC], gV[NC], gA[NC];
  float ph[NC], gph[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int sidx = c * groups_per_class + gi;
    dcol[c] = sidx < D ? col_sorted[sidx] : -1;
    vp[c] = dcol[c] >= 0 ? gather4<LPN>(Vp, dcol[c], sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    ph[c] = dcol[c] >= 0 ? phi[dcol[c]] : 1.f;
    gV[c] = gA[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    gph[c] = 0.f;
  }
  struct __attribute__((packed, aligned(4))) I4 { int x, y, z, w; };
  constexpr int FE = 4 * LPN;
  for (int p = cls; p < n_panels; p += 8) {
    const int pbase = p * panel_rows;
    const int32_t* pp = pc_ptr + (size_t)p * (D + 1);
    // list bounds of all NC slots of this panel up front (independent loads)
    int ls[NC], le[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ls[c] = dcol[c] >= 0 ? pp[dcol[c]] : 0;
      le[c] = dcol[c] >= 0 ? pp[dcol[c] + 1] : 0;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int cur = ls[c];
      const int end = le[c];
      I4 rA = {0, 0, 0, 0}, rB;
      int cnt0 = min(FE, end - cur), cnt1;
      if (4 * sub < cnt0) rA = *reinterpret_cast<const I4*>(reinterpret_cast<const int32_t*>(pc_ent) + cur + 4 * sub);
      cur += cnt0;
      while (__any(cnt0 > 0)) {
        cnt1 = min(FE, end - cur);
        rB = {0, 0, 0, 0};
        if (4 * sub < cnt1) rB = *reinterpret_cast<const I4*>(reinterpret_cast<const int32_t*>(pc_ent) + cur + 4 * sub);
        cur += cnt1;
        const int left = cnt0 - 4 * sub;
        const int rw[4] = {rA.x, rA.y, rA.z, rA.w};
        int rr[4];
        float xx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t w = (uint32_t)rw[t];
          rr[t] = left > t ? (int)(w >> 16) + pbase : 0;
          xx[t] = left > t ? (float)(w & 0xffffu) : 0.f;
        }
#pragma unroll
        for (int g0 = 0; g0 < FE; g0 += 4) {
          if (__any(cnt0 > g0)) {
            float4 zz[4], gg[4];
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int e = g0 + j, src = grp * LPN + e / 4;
              const int b = __shfl(rr[e % 4], src);
              xv[j] = __shfl(xx[e % 4], src);
              zz[j] = gather4<LPN>(z, b, sub);
              gg[j] = gather4<LPN>(gzs, b, sub);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float r = group_sum<LPN>(dot4(zz[j], vp[c])) + ph[c];
              const float xr = (r > 0.f && r < INFINITY) ? xv[j] * __builtin_amdgcn_rcpf(r) : (xv[j] > 0.f ? 1.f : 0.f);
              gV[c] = fma4(xr, zz[j], gV[c]);
              gA[c] = fma4(xv[j], gg[j], gA[c]);
              gph[c] += xr;
            }
          }
        }
        rA = rB;
        cnt0 = cnt1;
      }
    }
  }
  // one plain store per (class, column): the lane group is its only owner
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (dcol[c] < 0) continue;
    float* dst = partial + ((size_t)cls * D + dcol[c]) * (2 * KP + 1);
    // (2 KP + 1 floats: rows are 260 B, 4-byte aligned only)
    dst[4 * sub + 0] = gV[c].x; dst[4 * sub + 1] = gV[c].y; dst[4 * sub + 2] = gV[c].z; dst[4 * sub + 3] = gV[c].w;
    dst[KP + 4 * sub + 0] = gA[c].x; dst[KP + 4 * sub + 1] = gA[c].y; dst[KP + 4 * sub + 2] = gA[c].z; dst[KP + 4 * sub + 3] = gA[c].w;
    if (sub == 0) dst[2 * KP] = gph[c];
  }
}

__global__ __launch_bounds__(256) void class_reduce_kernel(int D, const float* __restrict__ partial,
                                                           float* __restrict__ gAp, float* __restrict__ gVp,
                                                           float* __restrict__ gphi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = 2 * KP + 1;
  if (i >= (int64_t)D * W) return;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += partial[(size_t)c * D * W + i];       // fixed order
  const int d = (int)(i / W), k = (int)(i % W);
  if (k < KP) gVp[(size_t)d * KP + k] = s;
  else if (k < 2 * KP) gAp[(size_t)d * KP + (k - KP)] = s;
  else gphi[d] = s;
}

int main(int argc, char** argv) {
  const int D = 20000, panel_rows = 11392, n_panels = 88, mean_len = 100;
  const int64_t B = (int64_t)panel_rows * n_panels;
  printf("rows %lld = %d panels x %d, D %d, ~%d entries per row\n", (long long)B, n_panels, panel_rows, D, mean_len);
  std::mt19937_64 rng(20241218);
  std::poisson_distribution<int> plen(mean_len);
  // per panel: lists by column.  First pass: rows -> (col, count); then a counting sort per panel.
  std::vector<int32_t> pc_ptr((size_t)n_panels * (D + 1), 0);
  std::vector<uint32_t> pc_ent;
  pc_ent.reserve((size_t)B * (mean_len + 2) + 128);
  std::vector<int64_t> col_total(D, 0);
  {
    std::vector<uint32_t> cols, rws, cts;      // one panel's entries
    std::vector<uint32_t> c1;
    for (int p = 0; p < n_panels; ++p) {
      cols.clear(); rws.clear(); cts.clear();
      for (int r = 0; r < panel_rows; ++r) {
        const int n = plen(rng);
        c1.clear();
        for (int i = 0; i < n; ++i) c1.push_back((uint32_t)(rng() % D));
        std::sort(c1.begin(), c1.end());
        c1.erase(std::unique(c1.begin(), c1.end()), c1.end());
        for (uint32_t c : c1) {
          cols.push_back(c);
          rws.push_back((uint32_t)r);
          cts.push_back((uint32_t)(1 + (rng() & 3)));
        }
      }
      std::vector<int32_t> cnt(D + 1, 0);
      for (uint32_t c : cols) cnt[c + 1]++;
      for (int d = 0; d < D; ++d) cnt[d + 1] += cnt[d];
      const int32_t base = (int32_t)pc_ent.size();
      int32_t* pp = &pc_ptr[(size_t)p * (D + 1)];
      for (int d = 0; d <= D; ++d) pp[d] = base + cnt[d];
      pc_ent.resize(pc_ent.size() + cols.size());
      std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
      for (size_t i = 0; i < cols.size(); ++i) pc_ent[(size_t)base + fill[cols[i]]++] = rws[i] << 16 | cts[i];   // rows ascend
      for (int d = 0; d < D; ++d) col_total[d] += cnt[d + 1] - cnt[d];
    }
  }
  const int64_t nnz = (int64_t)pc_ent.size();
  for (int i = 0; i < 128; ++i) pc_ent.push_back(0);          // pc_pad
  // items of the product's kernel: every non-empty (panel, column) list, sorted by length inside a panel
  std::vector<int32_t> items, item_ptr(n_panels + 1, 0);
  int max_items = 0;
  for (int p = 0; p < n_panels; ++p) {
    const int32_t* pp = &pc_ptr[(size_t)p * (D + 1)];
    std::vector<std::pair<int, int>> v;      // (len, column)
    for (int d = 0; d < D; ++d)
      if (pp[d + 1] > pp[d]) v.push_back({pp[d + 1] - pp[d], d});
    std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
    for (auto& it : v) {
      items.push_back(pp[it.second]);
      items.push_back(it.first);
      items.push_back(it.second);
      items.push_back(0);
    }
    item_ptr[p + 1] = (int32_t)(items.size() / 4);
    max_items = std::max(max_items, (int)v.size());
  }
  std::vector<int32_t> col_sorted(D);
  std::iota(col_sorted.begin(), col_sorted.end(), 0);
  std::sort(col_sorted.begin(), col_sorted.end(), [&](int a, int b) { return col_total[a] > col_total[b]; });
  printf("nnz %lld, %zu (panel, column) lists, %.1f entries a list\n", (long long)nnz, items.size() / 4,
         (double)nnz / (items.size() / 4));

  std::vector<float> hz((size_t)B * KP), hg((size_t)B * KP), hV((size_t)D * KP), hphi(D);
  auto unif = [&]() { return (float)((rng() >> 40) * (1.0 / (1 << 24))); };
  for (auto& v : hz) v = 0.05f + unif();
  for (auto& v : hg) v = unif() - 0.5f;
  for (auto& v : hV) v = 0.05f + unif();
  for (auto& v : hphi) v = 0.1f + unif();
  int32_t *d_ptr, *d_items, *d_iptr, *d_cs; uint32_t* d_ent; float *d_z, *d_g, *d_V, *d_phi, *d_acc0, *d_acc1, *d_part;
  const size_t accn = (size_t)2 * D * KP + D;
  CHECK(hipMalloc(&d_ptr, pc_ptr.size() * 4));
  CHECK(hipMalloc(&d_items, items.size() * 4));
  CHECK(hipMalloc(&d_iptr, item_ptr.size() * 4));
  CHECK(hipMalloc(&d_cs, D * 4));
  CHECK(hipMalloc(&d_ent, pc_ent.size() * 4));
  CHECK(hipMalloc(&d_z, hz.size() * 4));
  CHECK(hipMalloc(&d_g, hg.size() * 4));
  CHECK(hipMalloc(&d_V, hV.size() * 4));
  CHECK(hipMalloc(&d_phi, D * 4));
  CHECK(hipMalloc(&d_acc0, accn * 4));
  CHECK(hipMalloc(&d_acc1, accn * 4));
  CHECK(hipMalloc(&d_part, (size_t)8 * D * (2 * KP + 1) * 4));
  CHECK(hipMemcpy(d_ptr, pc_ptr.data(), pc_ptr.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_items, items.data(), items.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_iptr, item_ptr.data(), item_ptr.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_cs, col_sorted.data(), D * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_ent, pc_ent.data(), pc_ent.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_V, hV.data(), hV.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_phi, hphi.data(), D * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(d_part, 0, (size_t)8 * D * (2 * KP + 1) * 4));

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
    printf("%-64s %8.4f ms\n", name, ms);
    return ms;
  };
  // ---- the product's kernel (zero fill + launch, as a step does it) ------------------------------------
  ColArgs ca{D, n_panels, 0, max_items, d_iptr, d_items, nullptr, nullptr, d_V, d_phi, d_z, d_g, d_acc0,
             d_acc0 + (size_t)D * KP, d_acc0 + (size_t)2 * D * KP, 0, nullptr, nullptr, nullptr, 0, 1, B, (int64_t)accn, 64};
  ca.pc_row = reinterpret_cast<const int32_t*>(d_ent);        // (unused by the packed form; must be non-null)
  ca.pc_val = reinterpret_cast<const float*>(d_ent);
  ca.pc_ent = d_ent;
  ca.panel_rows = panel_rows;
  const float t_ref = timeit("product: col_pass_kernel<32,0,4,packed>, float-atomic flush (+ zero fill)", [&] {
    (void)hipMemsetAsync(d_acc0, 0, accn * 4, 0);
    launch_col_pass(KP, ca, 0);
  });
  // ---- owner forms ------------------------------------------------------------------------------------------
  auto owner = [&](const char* name, auto kern, int nc, int wps) {
    int occ = 0;                                              // what the registers of this instance allow
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
    if (occ < wps) { printf("%s: occupancy %d workgroups per CU < %d, skipped\n", name, occ, wps); return 0.f; }
    const int wgs_per_cu = wps * 4 / 4;                       // 256-thread workgroups: 4 waves each
    const int wgs_per_class = 32 * wgs_per_cu;                // 32 CUs per XCD
    const int gpc = wgs_per_class * 4 * NG;
    if ((int64_t)gpc * nc < D) { printf("%s: %d slots x %d groups < D, skipped\n", name, nc, gpc); return 0.f; }
    return timeit(name, [&] {
      hipLaunchKernelGGL(kern, dim3(8 * wgs_per_class), dim3(256), 0, 0, D, n_panels, panel_rows, gpc, d_cs, d_ptr,
                         d_ent, d_V, d_phi, d_z, d_g, d_part);
      hipLaunchKernelGGL(class_reduce_kernel, dim3((unsigned)((accn + 255) / 256)), dim3(256), 0, 0, D, d_part,
                         d_acc1, d_acc1 + (size_t)D * KP, d_acc1 + (size_t)2 * D * KP);
    });
  };
  float best = 1e9f;
  float t;
  // (second template argument: the register budget the instance is COMPILED for; the launch asks for the
  //  occupancy in the name and is skipped if the instance's registers do not allow it)
  t = owner("owner: 10 column slots, 2 waves/SIMD (+ class reduce)", col_owner_kernel<10, 2>, 10, 2); if (t > 0) best = std::min(best, t);
  t = owner("owner:  7 column slots, 3 waves/SIMD (+ class reduce)", col_owner_kernel<7, 2>, 7, 3); if (t > 0) best = std::min(best, t);
  t = owner("owner:  7 column slots, 3 waves/SIMD, 168-register build", col_owner_kernel<7, 3>, 7, 3); if (t > 0) best = std::min(best, t);
  t = owner("owner:  5 column slots, 4 waves/SIMD (+ class reduce)", col_owner_kernel<5, 3>, 5, 4); if (t > 0) best = std::min(best, t);
  t = owner("owner:  5 column slots, 4 waves/SIMD, 128-register build", col_owner_kernel<5, 4>, 5, 4); if (t > 0) best = std::min(best, t);
  // ---- same numbers? (the last owner launch against the product's) -------------------------------------------
  std::vector<float> a0(accn), a1(accn);
  CHECK(hipMemcpy(a0.data(), d_acc0, accn * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(a1.data(), d_acc1, accn * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, big = 0.0;
  for (size_t i = 0; i < accn; ++i) {
    worst = std::max(worst, (double)fabsf(a0[i] - a1[i]));
    big = std::max(big, (double)fabsf(a0[i]));
  }
  printf("max |owner - product| = %.3e (max |acc| %.3e)\n", worst, big);
  printf("best owner form / product = %.3f\n", best / t_ref);
  return worst <= 1e-4 * big ? 0 : 2;
}
