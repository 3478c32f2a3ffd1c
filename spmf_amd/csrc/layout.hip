// layout.hip -- builds the device layout of one row shard from its CSR arrays (gfx950):
// the row-panel CSC lists, the column-pass work items and the packed entry streams of a
// spmf_counts (include/spmf_hip.h).  The reference feeds its model dense [B,D] batches
// (mederrata_spmf/poisson.py:170,182; tests/spmf_test.py:17-22); this is the data-format step
// between a caller's sparse batch and the kernels of row_pass.hip / col_pass.hip.
//
// One stable sort of the stored entries by (panel, column) does the transposition: CSR order is
// ascending in the row, so a stable sort leaves every list ascending in the row, and the value
// sorted along is the packed word of the column pass itself (row inside the panel << 16 | count)
// whenever the counts fit it -- the lists are written by the sort, nothing is gathered
// afterwards.  The sort passes and the two prefix sums are rocPRIM's device primitives (radix
// sort over exactly the key bits in use: 21 for C3, three 8-bit passes); everything else is
// below.  Per stored entry: 8 B read + 8 B written by the key pass, 3 x 16 B by the sort,
// 8 B read + 8 B written by the unpack: ~80 B, against ~1 kB through the torch operators this
// replaces (int64 keys, eight sort passes, gathers through the permutation).
//
//   keys      wave per row: key = panel * D + column, word, ent = column << 16 | count; input checks
//   sort      (key, word) -> (sorted key, pc_ent)
//   bounds    thread per (panel, column): lower bound of the key in its panel's range -> pc_ptr
//   unpack    pc_row, pc_val from the packed words (or gathered through the sorted entry indices
//             when the values do not pack), zero padding behind the last list
//   items     segments per list -> prefix sum -> {start, len, column, 0} + sort key
//             ((panel * 2 + half) * (seg + 1) + seg - len), stable sort, gather, per-panel bounds
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <string>

#include "spmf_hip.h"
#include "common.h"
#include "kernels.h"

namespace spmf {
namespace {

constexpr int kPcPad = 64;        // zero entries behind the last list (spmf_counts.pc_pad)
constexpr int kSegMax = 256;      // longest run of one column a single lane group streams
constexpr uint32_t kPadKey = 0xffffffffu;   // item slots beyond n_items: all key bits set, and behind every real
                                            // item in the input, so the stable sort leaves them at the end

thread_local std::string g_layout_err;
int lfail(int code, const std::string& m) {
  g_layout_err = m;
  return code;
}
#define LCHK(call)                                                                     \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return lfail(SPMF_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
  } while (0)

inline size_t up256(size_t x) { return (x + 255) & ~size_t(255); }

// The entry points below take a device ordinal: they make it current for their own launches and put
// the caller's current device back on every return path (a process that drives several GPUs keeps its own).
struct DeviceScope {
  int prev = -1;
  hipError_t err;
  explicit DeviceScope(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    else if (err != hipSuccess) prev = -1;
    else prev = -1;                     // already current: nothing to restore
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct Geo {
  int64_t B, nnz, nkeys, max_items;
  int D, P, nP, seg, key_bits, item_bits;
};

// info block on the device
enum { I_FLAGS = 0, I_NITEMS, I_MAXPANEL, I_MAXLO, I_MAXHI, I_LEN = 8 };
enum { F_VALUE = 1, F_COLUMN = 2, F_ROWPTR = 4 };

int bits_for(uint64_t n) {         // bits needed for keys in [0, n)
  int b = 1;
  while (b < 64 && (uint64_t(1) << b) < n) ++b;
  return b;
}

// khint: the latent dimension of the model the layout is for (0 = unknown).  A work item is one lane GROUP of the
// column pass, KP / 4 lanes (KP = K padded to 4, 8, ...): at K <= 8 a wave carries 64 or 32 items, so "a few
// thousand items per panel" is a few dozen waves -- C1 (5000 x 200 dense, K = 2) ran its column pass on 39
// waves with one gather in flight per lane.  Small K asks for proportionally more, shorter items.
int make_geo(int64_t B, int64_t nnz, int D, int P, int khint, Geo* g) {
  if (B < 0 || nnz < 0 || D < 1 || P < 1) return lfail(SPMF_E_ARG, "layout: n_rows, nnz >= 0, n_cols, panel_rows >= 1");
  if (nnz >= (int64_t(1) << 31)) return lfail(SPMF_E_UNSUPPORTED, "layout: nnz per shard must fit int32");
  if (B >= (int64_t(1) << 31)) return lfail(SPMF_E_UNSUPPORTED, "layout: n_rows per shard must fit int32");
  g->B = B;
  g->nnz = nnz;
  g->D = D;
  g->P = (int)(P < (B > 1 ? B : 1) ? P : (B > 1 ? B : 1));
  g->nP = (int)(B > 0 ? (B + g->P - 1) / g->P : 1);
  g->nkeys = (int64_t)g->nP * D;
  // the sort key panel * D + column is 32 bits wide (and pc_ptr would be 16 GB beyond that): larger panels
  if (g->nkeys >= (int64_t(1) << 32)) return lfail(SPMF_E_UNSUPPORTED,
      "layout: n_panels * n_cols must stay below 2^32 (choose larger panels)");
  // segment length: a panel should offer a few thousand items (spmf_amd/sparse.py _build_items)
  const double per_panel = (double)nnz / (double)(g->nP > 0 ? g->nP : 1);
  const double want_items = khint >= 1 && khint <= 4 ? 32768.0 : (khint >= 5 && khint <= 8 ? 16384.0 : 4096.0);
  int seg = 16;
  while (seg < kSegMax && per_panel / seg > want_items) seg *= 2;
  g->seg = seg;
  const int64_t lists = nnz < g->nkeys ? nnz : g->nkeys;
  g->max_items = lists + nnz / seg + 1;
  g->key_bits = bits_for((uint64_t)(g->nkeys > 1 ? g->nkeys : 2));
  const uint64_t item_keys = (uint64_t)g->nP * 2 * (seg + 1);
  if (item_keys >= (uint64_t(1) << 32)) return lfail(SPMF_E_UNSUPPORTED, "layout: too many panels");
  g->item_bits = bits_for(item_keys > 1 ? item_keys : 2);
  return SPMF_OK;
}

struct LayoutCarve {
  size_t pc_ptr, pc_row, pc_val, pc_ent, ent, item_ptr, item_mid, per_panel, lower, items, list_first, item_pos, total;
};
LayoutCarve carve_layout(const Geo& g) {
  LayoutCarve c{};
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += up256(bytes); return at; };
  c.pc_ptr = take(((size_t)g.nkeys + g.nP) * 4);
  c.pc_row = take(((size_t)g.nnz + kPcPad) * 4);
  c.pc_val = take(((size_t)g.nnz + kPcPad) * 4);
  c.pc_ent = take(((size_t)g.nnz + kPcPad) * 4);
  c.ent = take(((size_t)g.nnz + 1) * 4);
  c.item_ptr = take(((size_t)g.nP + 1) * 4);
  c.item_mid = take((size_t)g.nP * 4);
  c.per_panel = take((size_t)g.nP * 4);
  c.lower = take((size_t)g.nP * 4);
  c.items = take((size_t)g.max_items * 16);
  c.list_first = take(((size_t)g.nkeys + 1) * 4);     // raw index of each list's first work item
  c.item_pos = take((size_t)g.max_items * 4);          // raw item -> position in `items`
  c.total = o;
  return c;
}

struct ScratchCarve {
  size_t key_in, key_out, word_in, excl, nseg, raw, ikey_in, ikey_out, ival_in, ival_out, info, temp,
      temp_bytes, total;
};

template <typename KeyT>
hipError_t sort_temp_bytes(size_t n, int bits, hipStream_t st, size_t* bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (KeyT*)nullptr, (KeyT*)nullptr, (uint32_t*)nullptr,
                                   (uint32_t*)nullptr, n, 0u, (unsigned)bits, st);
}

int carve_scratch(const Geo& g, hipStream_t st, ScratchCarve* out) {
  ScratchCarve c{};
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += up256(bytes); return at; };
  const size_t ks = 4;
  c.key_in = take(((size_t)g.nnz + 1) * ks);
  c.key_out = take(((size_t)g.nnz + 1) * ks);
  c.word_in = take(((size_t)g.nnz + 1) * 4);
  c.excl = take(((size_t)g.nkeys + 1) * 4);
  c.nseg = take(((size_t)g.nkeys + 1) * 4);
  c.raw = take((size_t)g.max_items * 16);
  c.ikey_in = take((size_t)g.max_items * 4);
  c.ikey_out = take((size_t)g.max_items * 4);
  c.ival_in = take((size_t)g.max_items * 4);
  c.ival_out = take((size_t)g.max_items * 4);
  c.info = take(I_LEN * 4);
  size_t t1 = 0, t2 = 0, t3 = 0;
  hipError_t e = sort_temp_bytes<uint32_t>((size_t)g.nnz, g.key_bits, st, &t1);
  if (e == hipSuccess) e = sort_temp_bytes<uint32_t>((size_t)g.max_items, g.item_bits, st, &t2);
  if (e == hipSuccess)
    e = rocprim::exclusive_scan(nullptr, t3, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)g.nkeys + 1,
                                rocprim::plus<int32_t>(), st);
  if (e != hipSuccess) return lfail(SPMF_E_HIP, std::string("layout: rocprim size query: ") + hipGetErrorString(e));
  c.temp_bytes = t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
  c.temp = take(c.temp_bytes + 256);
  c.total = o;
  *out = c;
  return SPMF_OK;
}

// ---- kernels ----------------------------------------------------------------------------

// wave per row.  BY_INDEX: the value sorted along is the entry's CSR position (the values do
// not pack, or a panel has more than 65536 rows); otherwise the column pass's packed word.
template <typename KeyT, bool BY_INDEX>
__global__ __launch_bounds__(256) void layout_keys_kernel(int64_t B, int64_t nnz, int D, int P,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ val,
                                                          KeyT* __restrict__ key, uint32_t* __restrict__ word,
                                                          uint32_t* __restrict__ ent, int* __restrict__ info) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int flags = 0;
  if (wave == 0 && lane == 0 && (row_ptr[0] != 0 || (int64_t)row_ptr[B] != nnz)) flags |= F_ROWPTR;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int64_t start = row_ptr[b], end = row_ptr[b + 1];
    if (start < 0 || end < start || end > nnz) {      // never write outside [0, nnz)
      flags |= F_ROWPTR;
      continue;
    }
    const int64_t panel = b / P;
    const uint32_t rin = (uint32_t)(b - panel * P);
    const KeyT base = (KeyT)panel * (KeyT)D;
    for (int64_t i = start + lane; i < end; i += 64) {
      int c = col[i];
      const float x = val[i];
      if (c < 0 || c >= D) {
        flags |= F_COLUMN;
        c = 0;
      }
      const bool ok = x >= 0.f && x <= 65535.f && x == floorf(x);
      if (!ok) flags |= F_VALUE;
      const uint32_t xi = ok ? (uint32_t)x : 0u;
      key[i] = base + (KeyT)c;
      word[i] = BY_INDEX ? (uint32_t)i : ((rin << 16) | xi);
      if (ent) ent[i] = ((uint32_t)c << 16) | xi;
    }
  }
  if (flags) atomicOr(&info[I_FLAGS], flags);
}

// thread per (panel, column) key: first sorted position >= key, searched inside the panel's
// own entry range (rows of a panel are contiguous in CSR, so are its sorted entries)
template <typename KeyT>
__global__ __launch_bounds__(256) void layout_bounds_kernel(int64_t nkeys, int D, int P, int64_t B, int64_t nnz,
                                                            const int32_t* __restrict__ row_ptr,
                                                            const KeyT* __restrict__ skey,
                                                            int32_t* __restrict__ excl,
                                                            int32_t* __restrict__ pc_ptr) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nkeys) return;
  if (k == nkeys) {
    excl[k] = (int32_t)nnz;
    pc_ptr[k + k / D - 1] = (int32_t)nnz;            // slot [n_panels - 1][D]
    return;
  }
  const int64_t p = k / D;
  const int64_t r0 = p * P < B ? p * P : B, r1 = (p + 1) * P < B ? (p + 1) * P : B;
  int64_t lo = row_ptr[r0], hi = row_ptr[r1];
  lo = lo < 0 ? 0 : (lo > nnz ? nnz : lo);            // (a bad row_ptr is reported by the key pass)
  hi = hi < lo ? lo : (hi > nnz ? nnz : hi);
  const KeyT kk = (KeyT)k;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skey[mid] < kk) lo = mid + 1;
    else hi = mid;
  }
  excl[k] = (int32_t)lo;
  pc_ptr[k + p] = (int32_t)lo;
  if (k - p * D == 0 && p > 0) pc_ptr[k + p - 1] = (int32_t)lo;   // slot [p - 1][D]
}

template <typename KeyT, bool BY_INDEX>
__global__ __launch_bounds__(256) void layout_unpack_kernel(int64_t nnz, int D, int P, int64_t B,
                                                            const int32_t* __restrict__ row_ptr,
                                                            const float* __restrict__ val,
                                                            const KeyT* __restrict__ skey,
                                                            uint32_t* __restrict__ sword,
                                                            int32_t* __restrict__ pc_row,
                                                            float* __restrict__ pc_val) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nnz + kPcPad) return;
  if (j >= nnz) {
    pc_row[j] = 0;
    pc_val[j] = 0.f;
    sword[j] = 0u;
    return;
  }
  int64_t p = (int64_t)(skey[j] / (KeyT)D);
  const int64_t last = B > 0 ? (B - 1) / P : 0;
  p = p > last ? last : p;                                 // (keys of a rejected input may be anything)
  const uint32_t w = sword[j];
  if (!BY_INDEX) {
    pc_row[j] = (int32_t)(p * P + (w >> 16));
    pc_val[j] = (float)(w & 0xffffu);
  } else {
    // row of CSR position w: last row of the panel whose row_ptr <= w
    int64_t lo = p * P, hi = (p + 1) * P < B ? (p + 1) * P : B;   // row_ptr[lo] <= w < row_ptr[hi]
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)row_ptr[mid] <= (int64_t)w) lo = mid;
      else hi = mid;
    }
    pc_row[j] = (int32_t)lo;
    pc_val[j] = (int64_t)w < nnz ? val[w] : 0.f;
  }
}

// (an input the key pass rejected -- a row_ptr that is no CSR, a column out of range -- can give
//  neighbouring panels overlapping or reversed entry ranges, hence list "lengths" below zero or of up to nnz per
//  panel: such an input gets NO work items at all, so nothing below writes outside the scratch carve; the
//  key pass ran earlier on this stream, its flags are complete)
__global__ __launch_bounds__(256) void layout_nseg_kernel(int64_t nkeys, int seg, const int32_t* __restrict__ excl,
                                                          int32_t* __restrict__ nseg, const int* __restrict__ info) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nkeys) return;
  const bool rejected = (info[I_FLAGS] & (F_ROWPTR | F_COLUMN)) != 0;
  const int len = k < nkeys ? excl[k + 1] - excl[k] : 0;
  nseg[k] = (rejected || len <= 0) ? 0 : (len + seg - 1) / seg;
}

__global__ __launch_bounds__(256) void layout_fill_kernel(int64_t n, uint32_t v, uint32_t* __restrict__ p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// thread per list: its segments as raw items + their sort keys, in (panel, column, segment) order
__global__ __launch_bounds__(256) void layout_items_kernel(int64_t nkeys, int D, int seg, int col_split,
                                                           const int32_t* __restrict__ excl,
                                                           const int32_t* __restrict__ first,
                                                           int4* __restrict__ raw, uint32_t* __restrict__ ikey,
                                                           uint32_t* __restrict__ ival, int* __restrict__ info,
                                                           int64_t max_items) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool rejected = (info[I_FLAGS] & (F_ROWPTR | F_COLUMN)) != 0;
  if (k == 0) {
    const int64_t n = rejected ? 0 : first[nkeys];
    info[I_NITEMS] = (int)(n < 0 ? 0 : (n > max_items ? max_items : n));
  }
  if (k >= nkeys || rejected) return;
  const int start = excl[k], len = excl[k + 1] - start;
  if (len <= 0) return;
  const int64_t p = k / D;
  const int d = (int)(k - p * D);
  const uint32_t half = col_split > 0 && d >= col_split ? 1u : 0u;
  const uint32_t kb = ((uint32_t)p * 2u + half) * (uint32_t)(seg + 1);
  const int f = first[k];
  for (int s = 0, off = 0; off < len; ++s, off += seg) {
    if (f < 0 || (int64_t)f + s >= max_items) return;      // (cannot happen on an accepted input: max_items bounds it)
    const int l = len - off < seg ? len - off : seg;
    raw[f + s] = make_int4(start + off, l, d, 0);
    ikey[f + s] = kb + (uint32_t)(seg - l);
    ival[f + s] = (uint32_t)(f + s);
  }
}

__global__ __launch_bounds__(256) void layout_gather_items_kernel(const int* __restrict__ info,
                                                                  const int4* __restrict__ raw,
                                                                  const uint32_t* __restrict__ sval,
                                                                  int4* __restrict__ items,
                                                                  int32_t* __restrict__ item_pos) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < info[I_NITEMS]) {
    const uint32_t r = sval[j];
    items[j] = raw[r];
    item_pos[r] = (int32_t)j;        // the permutation inverted: raw (generation-order) item -> sorted position
  }
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// thread per panel: bounds of its items in the sorted item keys
__global__ __launch_bounds__(256) void layout_panel_items_kernel(int nP, int seg, const uint32_t* __restrict__ skey,
                                                                 int32_t* __restrict__ item_ptr,
                                                                 int32_t* __restrict__ item_mid,
                                                                 int32_t* __restrict__ per_panel,
                                                                 int32_t* __restrict__ lower, int* __restrict__ info) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nP) return;
  const int n = info[I_NITEMS];
  const uint32_t w = (uint32_t)(seg + 1);
  const int a = lower_bound_u32(skey, n, (uint32_t)p * 2u * w);
  const int m = lower_bound_u32(skey, n, ((uint32_t)p * 2u + 1u) * w);
  const int e = lower_bound_u32(skey, n, ((uint32_t)p * 2u + 2u) * w);
  item_ptr[p] = a;
  item_mid[p] = m;
  if (p == nP - 1) item_ptr[nP] = e;
  per_panel[p] = e - a;
  lower[p] = m - a;
  atomicMax(&info[I_MAXPANEL], e - a);
  atomicMax(&info[I_MAXLO], m - a);
  atomicMax(&info[I_MAXHI], e - m);
}

// ---- dense [B,D] batches (the reference's own input: data[count_key], poisson.py:170) -> CSR ----
constexpr int kRowsPerBlock = 1024;     // rows whose counts one workgroup of the offset pass scans

__device__ __forceinline__ int wave_count_row(const float* __restrict__ row, int D, int lane) {
  int n = 0;
  for (int c = lane; c < D; c += 64) n += row[c] != 0.f ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  return n;
}

// wave per row: stored cells of the row -> cnt[b]; a block's rows are consecutive
__global__ __launch_bounds__(256) void dense_count_kernel(int64_t B, int D, const float* __restrict__ dense,
                                                          int64_t ld, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int n = wave_count_row(dense + b * ld, D, lane);
    if (lane == 0) cnt[b] = n;
  }
}

// sums of kRowsPerBlock consecutive counts
__global__ __launch_bounds__(256) void dense_blocksum_kernel(int64_t B, const int32_t* __restrict__ cnt,
                                                             int64_t* __restrict__ bsum) {
  __shared__ int64_t red[4];
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  int64_t s = 0;
  for (int i = threadIdx.x; i < kRowsPerBlock; i += 256)
    if (r0 + i < B) s += cnt[r0 + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup: exclusive scan of the block sums in place, the total behind them
__global__ __launch_bounds__(256) void dense_scan_blocks_kernel(int64_t nblk, int64_t* __restrict__ bsum) {
  __shared__ int64_t part[256];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblk; base += 256) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = i < nblk ? bsum[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {           // Hillis-Steele, inclusive
      const int64_t t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) bsum[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += part[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nblk] = carry;
}

// workgroup per kRowsPerBlock rows: counts -> offsets (in place: cnt[b] becomes row_ptr[b]; row_ptr[B] = total)
__global__ __launch_bounds__(256) void dense_offsets_kernel(int64_t B, const int64_t* __restrict__ bsum,
                                                            int32_t* __restrict__ row_ptr) {
  __shared__ int64_t part[256];
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x * 4;
  const int64_t total = bsum[gridDim.x];
  int c[4];
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = r0 + j < B ? row_ptr[r0 + j] : 0;
    s += c[j];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int64_t t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t off = bsum[blockIdx.x] + part[threadIdx.x] - s;
  const int64_t cap = 2147483647;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (r0 + j < B) row_ptr[r0 + j] = (int32_t)(off < cap ? off : cap);
    off += c[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) row_ptr[B] = (int32_t)(total < cap ? total : cap);
}

// wave per row: the stored cells of the row in column order
__global__ __launch_bounds__(256) void dense_fill_kernel(int64_t B, int D, const float* __restrict__ dense, int64_t ld,
                                                         const int32_t* __restrict__ row_ptr,
                                                         int32_t* __restrict__ col, float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const float* __restrict__ row = dense + b * ld;
    int64_t at = row_ptr[b];
    for (int c0 = 0; c0 < D; c0 += 64) {
      const int c = c0 + lane;
      const float x = c < D ? row[c] : 0.f;
      const bool st = c < D && x != 0.f;
      const unsigned long long m = __ballot(st);
      if (st) {
        const int64_t i = at + __popcll(m & ((1ull << lane) - 1ull));
        col[i] = c;
        val[i] = x;
      }
      at += __popcll(m);
    }
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

template <typename KeyT>
int run_build(const Geo& g, const int32_t* row_ptr, const int32_t* col, const float* val, int col_split,
              char* layout, const LayoutCarve& L, char* scratch, const ScratchCarve& S, bool* packed_words,
              int host_info[I_LEN], hipStream_t st) {
  KeyT* key_in = (KeyT*)(scratch + S.key_in);
  KeyT* key_out = (KeyT*)(scratch + S.key_out);
  uint32_t* word_in = (uint32_t*)(scratch + S.word_in);
  uint32_t* pc_ent = (uint32_t*)(layout + L.pc_ent);
  uint32_t* ent = g.D <= 65536 ? (uint32_t*)(layout + L.ent) : nullptr;
  int* info = (int*)(scratch + S.info);
  int32_t* excl = (int32_t*)(scratch + S.excl);
  int32_t* first = (int32_t*)(layout + L.list_first);
  void* temp = scratch + S.temp;
  size_t tb = S.temp_bytes;
  const int64_t want = (g.B + 3) / 4;
  const unsigned nb = (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
  bool by_index = g.P > 65536;
  for (int attempt = 0; attempt < 2; ++attempt) {
    LCHK(hipMemsetAsync(info, 0, I_LEN * 4, st));
    if (g.nnz > 0) {
      if (by_index)
        hipLaunchKernelGGL((layout_keys_kernel<KeyT, true>), dim3(nb), dim3(256), 0, st, g.B, g.nnz, g.D, g.P,
                           row_ptr, col, val, key_in, word_in, ent, info);
      else
        hipLaunchKernelGGL((layout_keys_kernel<KeyT, false>), dim3(nb), dim3(256), 0, st, g.B, g.nnz, g.D, g.P,
                           row_ptr, col, val, key_in, word_in, ent, info);
      LCHK(rocprim::radix_sort_pairs(temp, tb, key_in, key_out, word_in, pc_ent, (size_t)g.nnz, 0u,
                                     (unsigned)g.key_bits, st));
    }
    hipLaunchKernelGGL((layout_bounds_kernel<KeyT>), dim3(blocks_for(g.nkeys + 1)), dim3(256), 0, st, g.nkeys, g.D,
                       g.P, g.B, g.nnz, row_ptr, key_out, excl, (int32_t*)(layout + L.pc_ptr));
    if (by_index)
      hipLaunchKernelGGL((layout_unpack_kernel<KeyT, true>), dim3(blocks_for(g.nnz + kPcPad)), dim3(256), 0, st,
                         g.nnz, g.D, g.P, g.B, row_ptr, val, key_out, pc_ent, (int32_t*)(layout + L.pc_row),
                         (float*)(layout + L.pc_val));
    else
      hipLaunchKernelGGL((layout_unpack_kernel<KeyT, false>), dim3(blocks_for(g.nnz + kPcPad)), dim3(256), 0, st,
                         g.nnz, g.D, g.P, g.B, row_ptr, val, key_out, pc_ent, (int32_t*)(layout + L.pc_row),
                         (float*)(layout + L.pc_val));
    if (attempt == 0) {
      // work items: they depend on the list lengths only
      int32_t* nseg = (int32_t*)(scratch + S.nseg);
      hipLaunchKernelGGL(layout_nseg_kernel, dim3(blocks_for(g.nkeys + 1)), dim3(256), 0, st, g.nkeys, g.seg, excl,
                         nseg, info);
      LCHK(rocprim::exclusive_scan(temp, tb, nseg, first, 0, (size_t)g.nkeys + 1, rocprim::plus<int32_t>(), st));
      uint32_t* ikey_in = (uint32_t*)(scratch + S.ikey_in);
      uint32_t* ikey_out = (uint32_t*)(scratch + S.ikey_out);
      uint32_t* ival_in = (uint32_t*)(scratch + S.ival_in);
      uint32_t* ival_out = (uint32_t*)(scratch + S.ival_out);
      hipLaunchKernelGGL(layout_fill_kernel, dim3(blocks_for(g.max_items)), dim3(256), 0, st, g.max_items, kPadKey,
                         ikey_in);
      hipLaunchKernelGGL(layout_fill_kernel, dim3(blocks_for(g.max_items)), dim3(256), 0, st, g.max_items, 0u,
                         ival_in);
      hipLaunchKernelGGL(layout_items_kernel, dim3(blocks_for(g.nkeys + 1)), dim3(256), 0, st, g.nkeys, g.D, g.seg,
                         col_split, excl, first, (int4*)(scratch + S.raw), ikey_in, ival_in, info,
                         (int64_t)g.max_items);
      LCHK(rocprim::radix_sort_pairs(temp, tb, ikey_in, ikey_out, ival_in, ival_out, (size_t)g.max_items, 0u,
                                     (unsigned)g.item_bits, st));
      hipLaunchKernelGGL(layout_gather_items_kernel, dim3(blocks_for(g.max_items)), dim3(256), 0, st, info,
                         (const int4*)(scratch + S.raw), ival_out, (int4*)(layout + L.items),
                         (int32_t*)(layout + L.item_pos));
      hipLaunchKernelGGL(layout_panel_items_kernel, dim3(blocks_for(g.nP)), dim3(256), 0, st, g.nP, g.seg, ikey_out,
                         (int32_t*)(layout + L.item_ptr), (int32_t*)(layout + L.item_mid),
                         (int32_t*)(layout + L.per_panel), (int32_t*)(layout + L.lower), info);
    }
    LCHK(hipGetLastError());
    int hi[I_LEN];
    LCHK(hipMemcpyAsync(hi, info, I_LEN * 4, hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st));
    if (attempt == 0)
      for (int i = 0; i < I_LEN; ++i) host_info[i] = hi[i];
    else
      host_info[I_FLAGS] |= hi[I_FLAGS];
    if (hi[I_FLAGS] & (F_COLUMN | F_ROWPTR)) break;
    if (by_index || !(hi[I_FLAGS] & F_VALUE)) break;
    by_index = true;      // the values do not pack: once more, sorting the entry positions along
  }
  *packed_words = !by_index;
  return SPMF_OK;
}

}  // namespace

const char* layout_last_error() { return g_layout_err.c_str(); }

}  // namespace spmf

using namespace spmf;

extern "C" {

size_t spmf_dense_scratch_bytes(int64_t n_rows) {
  const int64_t nblk = n_rows > 0 ? (n_rows + kRowsPerBlock - 1) / kRowsPerBlock : 1;
  return (size_t)(nblk + 2) * 8;
}

int spmf_dense_row_ptr(int device, int64_t n_rows, int32_t n_cols, const float* dense, int64_t ld,
                       int32_t* row_ptr, void* scratch, size_t scratch_bytes, void* stream) {
  if (n_rows < 0 || n_cols < 1 || !row_ptr || !scratch || (n_rows > 0 && !dense) || ld < n_cols)
    return lfail(SPMF_E_ARG, "dense_row_ptr: bad arguments (n_rows >= 0, n_cols >= 1, ld >= n_cols, non-null buffers)");
  if (n_rows >= (int64_t(1) << 31)) return lfail(SPMF_E_UNSUPPORTED, "dense_row_ptr: n_rows must fit int32");
  if (scratch_bytes < spmf_dense_scratch_bytes(n_rows) || ((uintptr_t)scratch & 7))
    return lfail(SPMF_E_WORKSPACE, "dense_row_ptr: scratch smaller than spmf_dense_scratch_bytes or not 8-byte aligned");
  DeviceScope dev_scope(device);
  LCHK(dev_scope.err);
  hipStream_t st = (hipStream_t)stream;
  int64_t* bsum = (int64_t*)scratch;
  const int64_t nblk = n_rows > 0 ? (n_rows + kRowsPerBlock - 1) / kRowsPerBlock : 1;
  if (n_rows > 0) {
    const int64_t want = (n_rows + 3) / 4;
    hipLaunchKernelGGL(dense_count_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, st, n_rows,
                       (int)n_cols, dense, ld, row_ptr);
  }
  hipLaunchKernelGGL(dense_blocksum_kernel, dim3((unsigned)nblk), dim3(256), 0, st, n_rows, row_ptr, bsum);
  hipLaunchKernelGGL(dense_scan_blocks_kernel, dim3(1), dim3(256), 0, st, nblk, bsum);
  hipLaunchKernelGGL(dense_offsets_kernel, dim3((unsigned)nblk), dim3(256), 0, st, n_rows, bsum, row_ptr);
  LCHK(hipGetLastError());
  return SPMF_OK;
}

int spmf_dense_fill_csr(int device, int64_t n_rows, int32_t n_cols, const float* dense, int64_t ld,
                        const int32_t* row_ptr, int32_t* col_idx, float* val, void* stream) {
  if (n_rows < 0 || n_cols < 1 || !row_ptr || ld < n_cols)
    return lfail(SPMF_E_ARG, "dense_fill_csr: bad arguments");
  if (n_rows == 0) return SPMF_OK;
  if (!dense || !col_idx || !val) return lfail(SPMF_E_ARG, "dense_fill_csr: null buffers");
  DeviceScope dev_scope(device);
  LCHK(dev_scope.err);
  const int64_t want = (n_rows + 3) / 4;
  hipLaunchKernelGGL(dense_fill_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0,
                     (hipStream_t)stream, n_rows, (int)n_cols, dense, ld, row_ptr, col_idx, val);
  LCHK(hipGetLastError());
  return SPMF_OK;
}

size_t spmf_sizeof_layout_info(void) { return sizeof(spmf_layout_info); }

const char* spmf_layout_last_error(void) { return layout_last_error(); }

int spmf_layout_sizes(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, int32_t panel_rows,
                      size_t* layout_bytes, size_t* scratch_bytes) {
  return spmf_layout_sizes_k(device, n_rows, nnz, n_cols, panel_rows, 0, layout_bytes, scratch_bytes);
}

int spmf_layout_sizes_k(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, int32_t panel_rows,
                        int32_t latent_dim, size_t* layout_bytes, size_t* scratch_bytes) {
  if (!layout_bytes || !scratch_bytes) return lfail(SPMF_E_ARG, "layout_sizes: null output");
  if (latent_dim < 0) return lfail(SPMF_E_ARG, "layout_sizes: latent_dim >= 0 (0 = unknown)");
  Geo g;
  int rc = make_geo(n_rows, nnz, n_cols, panel_rows, latent_dim, &g);
  if (rc) return rc;
  *layout_bytes = carve_layout(g).total;
  DeviceScope dev_scope(device);
  LCHK(dev_scope.err);
  ScratchCarve s;
  rc = carve_scratch(g, nullptr, &s);
  if (rc) return rc;
  *scratch_bytes = s.total;
  return SPMF_OK;
}

int spmf_layout_build(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, const int32_t* row_ptr,
                      const int32_t* col_idx, const float* val, int32_t panel_rows, int32_t col_split,
                      void* layout, size_t layout_bytes, void* scratch, size_t scratch_bytes, spmf_counts* out,
                      spmf_layout_info* info, void* stream) {
  return spmf_layout_build_k(device, n_rows, nnz, n_cols, row_ptr, col_idx, val, panel_rows, col_split, 0, layout,
                             layout_bytes, scratch, scratch_bytes, out, info, stream);
}

int spmf_layout_build_k(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, const int32_t* row_ptr,
                        const int32_t* col_idx, const float* val, int32_t panel_rows, int32_t col_split,
                        int32_t latent_dim, void* layout, size_t layout_bytes, void* scratch, size_t scratch_bytes,
                        spmf_counts* out, spmf_layout_info* info, void* stream) {
  if (!out || !info || !layout || !scratch || !row_ptr) return lfail(SPMF_E_ARG, "layout_build: null argument");
  if (latent_dim < 0) return lfail(SPMF_E_ARG, "layout_build: latent_dim >= 0 (0 = unknown)");
  if (info->struct_size != (int32_t)sizeof(spmf_layout_info))
    return lfail(SPMF_E_ARG, "layout_build: spmf_layout_info.struct_size differs: built against another spmf_hip.h");
  if (nnz > 0 && (!col_idx || !val)) return lfail(SPMF_E_ARG, "layout_build: col_idx / val missing");
  if (col_split < 0 || col_split > n_cols) return lfail(SPMF_E_ARG, "layout_build: col_split outside [0, n_cols]");
  if (((uintptr_t)layout | (uintptr_t)scratch) & 255)
    return lfail(SPMF_E_ARG, "layout_build: buffers must be 256-byte aligned");
  Geo g;
  int rc = make_geo(n_rows, nnz, n_cols, panel_rows, latent_dim, &g);
  if (rc) return rc;
  DeviceScope dev_scope(device);
  LCHK(dev_scope.err);
  hipStream_t st = (hipStream_t)stream;
  const LayoutCarve L = carve_layout(g);
  ScratchCarve S;
  rc = carve_scratch(g, st, &S);
  if (rc) return rc;
  if (layout_bytes < L.total || scratch_bytes < S.total)
    return lfail(SPMF_E_WORKSPACE, "layout_build: buffers smaller than spmf_layout_sizes reports");
  bool packed_words = false;
  int hi[I_LEN] = {0};
  rc = run_build<uint32_t>(g, row_ptr, col_idx, val, col_split, (char*)layout, L, (char*)scratch, S,
                           &packed_words, hi, st);
  if (rc) return rc;
  if (hi[I_FLAGS] & F_ROWPTR)
    return lfail(SPMF_E_ARG, "layout_build: row_ptr is not a CSR offset array of this shard (row_ptr[0] == 0, "
                             "non-decreasing, row_ptr[n_rows] == nnz)");
  if (hi[I_FLAGS] & F_COLUMN) return lfail(SPMF_E_ARG, "layout_build: a column index lies outside [0, n_cols)");
  const bool values_pack = !(hi[I_FLAGS] & F_VALUE) && g.nnz > 0;
  char* lb = (char*)layout;
  spmf_counts c{};
  c.struct_size = (int32_t)sizeof(spmf_counts);
  c.n_rows = g.B;
  c.nnz = g.nnz;
  c.n_cols = g.D;
  c.n_panels = g.nP;
  c.panel_rows = g.P;
  c.row_base = 0;
  c.row_ptr = row_ptr;
  c.col_idx = col_idx;
  c.val = val;
  c.pc_ptr = (const int32_t*)(lb + L.pc_ptr);
  c.pc_row = (const int32_t*)(lb + L.pc_row);
  c.pc_val = (const float*)(lb + L.pc_val);
  c.item_ptr = (const int32_t*)(lb + L.item_ptr);
  c.items = (const int32_t*)(lb + L.items);
  c.item_mid = (const int32_t*)(lb + L.item_mid);
  c.max_items_per_panel = hi[I_MAXPANEL];
  c.max_items_half[0] = hi[I_MAXLO];
  c.max_items_half[1] = hi[I_MAXHI];
  c.col_split = col_split;
  c.pc_pad = kPcPad;
  c.list_first = (const int32_t*)(lb + L.list_first);
  c.item_pos = (const int32_t*)(lb + L.item_pos);
  c.n_items = hi[I_NITEMS];
  c.ent = values_pack && g.D <= 65536 ? (const uint32_t*)(lb + L.ent) : nullptr;
  c.pc_ent = values_pack && packed_words ? (const uint32_t*)(lb + L.pc_ent) : nullptr;
  *out = c;
  info->n_panels = g.nP;
  info->panel_rows = g.P;
  info->segment = g.seg;
  info->n_items = hi[I_NITEMS];
  info->packed_ent = c.ent != nullptr;
  info->packed_pc_ent = c.pc_ent != nullptr;
  info->items_per_panel = (const int32_t*)(lb + L.per_panel);
  info->items_lower = (const int32_t*)(lb + L.lower);
  return SPMF_OK;
}

}  // extern "C"
