// api.hip -- the C-ABI of libspmf_hip.so (see include/spmf_hip.h).
// Host-side orchestration only: argument checks, workspace carving, stream
// ordered launches.  No device allocation, no host<->device copies, no
// synchronisation on the hot path (timing taps excepted, off by default).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/spmf_hip.h"
#include "common.h"
#include "kernels.h"

using namespace spmf;

struct spmf_ctx {
  int device = 0, K = 0, D = 0, KP = 0;
  unsigned flags = 0;
  double u_tau_scale = 0.01, s_tau_scale = 1.0, decay = 0.99;  // poisson.py:59
  // workspace carve
  char* ws = nullptr;
  size_t ws_bytes = 0;
  int64_t ws_rows = 0;
  int ws_S = 0;
  float* acc = nullptr;
  double* dacc = nullptr;
  double* dprep = nullptr;
  // deterministic mode (spmf_ctx_set_deterministic): caller-owned scratch for the row pass's per-workgroup
  // scalar slots and the column pass's per-item partial sums; null = off
  char* det_buf = nullptr;
  size_t det_bytes = 0;
  double* fpart = nullptr;   // finish kernel: per-block prior-part sums
  float* futau = nullptr;    //                per-block u_tau gradient sums
  float *Ap = nullptr, *Vp = nullptr, *phi = nullptr, *z = nullptr, *gzs = nullptr, *gzd = nullptr, *dbias = nullptr;
  const uint8_t* ctype = nullptr;   // mixed likelihood: 1 = Bernoulli column (device, caller-owned)
  const int32_t* bcols = nullptr;   // mixed likelihood: ascending indices of the Bernoulli columns (optional)
  int n_bcols = 0;
  float *Vb = nullptr, *bb = nullptr;   // compacted V' rows / logit biases of those columns
  // timing taps
  int timing = 0;
  static constexpr int kSets = 64;  // ring of event sets: no sync inside a timed loop
  hipEvent_t evs[kSets][8] = {};   // 0..3 data pass, 4..5 finish, 6..7 dense exp kernels
  hipEvent_t* ev = evs[0];
  int ev_set = -1;      // set used by the call in flight
  int ev_count = 0;     // complete sets recorded since enable
  int ev_valid = 0;
  // prior half of the finish on a side stream (spmf_prior_async)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int Dh = 0;                     // column split of the accumulator layout (0 = none)
  int batched = 0;                // the bound workspace holds per-draw tables (S draws per launch)
  int prior_pending = 0;          // S of the launched prior half, 0 = none
  const double* prior_parts = nullptr;
  // the library's only device allocation: a small scratch for per-block partial sums of the
  // O(D*K) surrogate kernels (fixed-order reductions instead of same-address atomics)
  // log_transform: E = exp(<z_b, W_d>) is computed once and kept for the second contraction
  // (dense.hip, estdot_kernel), in row chunks of at most kEstCapBytes
  float* est = nullptr;
  int64_t est_rows = 0;           // rows per chunk of the bound workspace
  int e_once = 1;                 // SPMF_DENSE_E_ONCE=0: recompute E in a second launch instead
  int fuse_rows = -1;             // dense-term contexts: one fused row pass + dense-kernel epilogue instead of
                                  // encode sweep -> dense -> stored-cell sweep.  -1 = where it measured faster:
                                  // the sigmoid forms at KP = 32 (C5 row launches 0.97 -> 0.84 ms); NOT the exp
                                  // decoder at KP = 64, where the two-stream fused kernel (20 spilled registers
                                  // at four waves per SIMD) ran 13.9 ms against 12.8 for the two launches on C4
                                  // (profiles/r04_fused_rows_c4.txt).  SPMF_FUSE_ROWS=1 / 0 forces it on / off.
  int dyn_rows = 1;               // the row pass hands the last eighth of its rows out dynamically (row_pass.hip);
                                  // SPMF_ROW_DYNAMIC=0: the fixed stride everywhere
  int dense3 = 1;                 // exp sums on the bf16 matrix cores with three-way split operands
                                  // (dense3.hip; Poisson log_transform at KP = 64 only);
                                  // SPMF_DENSE_BF16X3=0 selects the exact-f32 MFMA kernels (dense.hip)
  size_t est_cap_bytes = (size_t)8 << 30;   // spmf_ctx_set_e_cap
  double* scratch = nullptr;
  static constexpr size_t kScratchDoubles = 1u << 20;   // 8 MiB
  void* comm = nullptr;           // ncclComm_t of the row-shard collective (spmf_comm_init)
  hipEvent_t rows_event = nullptr; // caller's event recorded behind the row stage of a data pass (spmf_ctx_set_rows_event)
  // spmf_step_begin .. spmf_step_end (ABI 6): the step's outputs are known from its first call on, so the
  // prior half of the finish rides in the prep launch (fused = 1) and spmf_step_end launches the data half only
  struct StepOut {
    int active = 0, fused = 0, S = 0;
    double prior_weight = 1.0;
    const float* params[SPMF_NVARS] = {};
    float* grads[SPMF_NVARS] = {};
    const float* eta = nullptr;
    double* parts = nullptr;
    double* nnf = nullptr;
  } step;
  int comm_rank = 0, comm_world = 1;
  // the hand-written collective over peer pointers (spmf_p2p_*; p2p.hip): this rank's fine-grained region
  // and the peers' regions as mapped into this process
  struct P2P {
    char* region = nullptr;          // own allocation: [rs | ag | flags | seq]
    size_t bytes = 0, alloc_bytes = 0, off_ag = 0, off_flags = 0, off_seq = 0;
    int64_t n_max = 0, slice_cap = 0;
    int rank = 0, world = 0, nchunk = 0, connected = 0;
    char* peer[kP2PMaxWorld] = {};   // peer regions (own rank: own region); opened with hipIpcOpenMemHandle
  } p2p;
  std::string err;
};

// ---- RCCL, bound at run time ------------------------------------------------
// The library does not link librccl: single-GPU users never need it.  spmf_comm_*
// dlopen it on first use; the few entry points used are declared here with the
// ABI of <rccl/rccl.h> (ncclUniqueId = 128 opaque bytes, passed by value).
namespace {
struct RcclId { char internal[128]; };
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(void**, int, RcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (r.h) {
      r.get_id = (fn_get_id)dlsym(r.h, "ncclGetUniqueId");
      r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
      r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
      r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
      r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
      if (!r.get_id || !r.init_rank || !r.allreduce || !r.destroy) r.h = nullptr;
    }
  }
  return r.h ? &r : nullptr;
}
constexpr int kNcclFloat = 7, kNcclSum = 0;   // ncclFloat32, ncclSum (rccl.h enums)
}  // namespace

static int fail(spmf_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}
#define HIPCHK(c, call)                                                            \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail((c), SPMF_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

static int padded_k(int K) {
  int kp = 4;
  while (kp < K) kp <<= 1;
  return kp;
}
static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t var_size(const spmf_ctx* c, int i) {
  const size_t D = c->D, K = c->K;
  switch (i) {
    case 0: case 2: case 3: case 8: return D * K;
    case 4: case 9: return K;
    case 1: case 6: case 11: return D;
    default: return 2 * D;  // 5 s_eta, 7 s, 10 s_eta_a
  }
}

struct Carve {
  size_t acc, dacc, dprep, ppart, putau, Ap, Vp, phi, dbias, Vb, bb, z, gzs, gzd, est, total;
};
// Small batches run all S draws in ONE launch per kernel (gridDim.y = S): the per-draw tables and
// row outputs then exist S times.  Only for the linear Poisson decoder.  Two cases:
//   * the S table pairs stay L2 sized (<= 3 MB): any batch up to 256 MB of row outputs -- beyond that size a
//     draw is gather-bound, its tables want the L2 to themselves, and draws run in turn;
//   * the batch is launch-bound whatever the tables weigh (<= kSmallBatchRows rows: the reference's own
//     harness trains on batches of 10 rows with sample_size = 20, tests/spmf_test.py:35-43, where D = 350,
//     K = 50 gives 3.58 MB of tables -- just over the cut above -- and 20 launch sequences per step): a
//     draw's tables are only touched by that draw's few rows, residency is moot, and S x 4 launches
//     become 4.  Bounded by 512 MB of tables.
constexpr int64_t kSmallBatchRows = 2048;
static bool batched_draws(const spmf_ctx* c, int64_t rows, int S) {
  if (S < 2 || c->Dh > 0) return false;
  if (c->flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED)) return false;
  const size_t tables = (size_t)S * 2 * c->D * c->KP * sizeof(float);
  const size_t rowbuf = (size_t)S * 2 * (size_t)rows * c->KP * sizeof(float);
  if (rows <= kSmallBatchRows && tables <= ((size_t)512 << 20)) return true;
  return tables <= (3u << 20) && rowbuf <= (256u << 20);
}

static int fail(spmf_ctx* c, int code, const std::string& msg);
// a bf16x3 dense launcher answered "shape not covered": the dense term of the step would be missing from
// gzs / gV' / the softplus sum -- fail instead (the use_* predicates below are meant to make this unreachable)
static int dense3_uncovered(spmf_ctx* c) {
  return fail(c, SPMF_E_UNSUPPORTED, "data_pass: the bf16x3 dense kernels do not cover this launch shape "
      "(set SPMF_DENSE_BF16X3=0 for the exact-f32 kernels)");
}

static int likelihood_code(const spmf_ctx* c) {   // common.h: lik_exp / lik_bern
  const bool lt = (c->flags & SPMF_FLAG_LOG_TRANSFORM) != 0;
  if (c->flags & SPMF_FLAG_BERNOULLI) return lt ? 4 : 2;
  if (c->flags & SPMF_FLAG_MIXED) return 3;
  return lt ? 1 : 0;
}

// rows of one E chunk: whole 128-row workgroups of the exp kernel, at most kEstCapBytes
static int64_t est_chunk_rows(const spmf_ctx* c, int64_t rows) {
  const size_t per_row = (size_t)((c->D + 31) / 32) * 32 * sizeof(float);
  int64_t cap = (int64_t)(c->est_cap_bytes / per_row) / 128 * 128;
  if (cap < 128) cap = 128;
  return rows < cap ? rows : cap;
}

// The bf16x3 form of the exp sums (dense3.hip) covers the Poisson exp decoder at KP = 64 and
// recomputes E in its second launch.  The exact-f32 exp and sigmoid forms keep E (exp / sigmoid
// of the logits) in HBM between their two contractions; Bernoulli + log_transform (code 4: E
// would have to carry exp(X) too) recomputes.
static bool uses_dense3(const spmf_ctx* c) {
  return c->dense3 && likelihood_code(c) == 1 && c->KP == 64;
}
// The bf16x3 form of the sigmoid / softplus sums (dense3.hip sigdot3: Bernoulli and mixed contexts with the
// linear decoder) at KP = 32 and 64; it recomputes the sigmoid in its second launch too.
static bool uses_sig3(const spmf_ctx* c) {
  const int lik = likelihood_code(c);
  return c->dense3 && (lik == 2 || lik == 3) && (c->KP == 32 || c->KP == 64);
}
// The Poisson exp decoder at KP = 32 (K <= 32: the reference's scRNA script runs P = 3,
// bin/factorize_scrnaseq_counts.py) on the bf16x3 form as well: dense3.hip's sigdot3 family with ACT 0.
static bool uses_exp3_32(const spmf_ctx* c) {
  return c->dense3 && likelihood_code(c) == 1 && c->KP == 32;
}
// Bernoulli + log_transform (code 4: logit = exp(<z, V'>) - 1 + phi, bernoulli.py:60-61) at KP = 32 on the same
// family (ACT 2: exp accumulators as ACT 0, sigmoid / softplus epilogue as ACT 1, E = sigmoid * exp).
static bool uses_sigexp3(const spmf_ctx* c) {
  return c->dense3 && likelihood_code(c) == 4 && c->KP == 32;
}
static bool uses_e_buffer(const spmf_ctx* c) {
  return (c->flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED)) && c->e_once &&
         likelihood_code(c) != 4 && !uses_dense3(c) && !uses_sig3(c) && !uses_exp3_32(c);
}

// Q chunks (gridDim.y) of a P-stationary dense launch of nbx workgroup columns over ntiles Q tiles on
// `slots` resident workgroups: the count that fills whole rounds of the chip best, charging every
// workgroup a fixed prologue (P fragments, first tile) of about 0.7 tile-times.
static int pick_chunks(int nbx, int ntiles, int slots, int max_chunks) {
  int best = 1;
  double best_eff = -1.0;
  for (int c = 1; c <= ntiles && c <= max_chunks; ++c) {
    const int tpc = (ntiles + c - 1) / c;
    if ((ntiles + tpc - 1) / tpc != c) continue;          // some chunk would get no tile
    const long n = (long)nbx * c;
    const long rounds = (n + slots - 1) / slots;
    const double eff = ((double)n / (double)(rounds * slots)) * ((double)ntiles / ((double)c * tpc)) *
                       ((double)tpc / (tpc + 0.7));
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = c;
    }
  }
  return best;
}

static Carve carve(const spmf_ctx* c, int64_t rows, int S) {
  Carve k;
  size_t o = 0;
  const size_t KP = c->KP, D = c->D;
  const size_t nd = batched_draws(c, rows, S) ? (size_t)S : 1;   // per-draw copies
  k.acc = o;   o += al((size_t)S * acc_len(c->D, c->KP) * sizeof(float));
  k.dacc = o;  o += al((size_t)S * kDaccRep * (kDaccHead + KP) * sizeof(double));
  k.dprep = o; o += al((size_t)S * kPrepSeg * (KP + 1) * sizeof(double));
  const size_t fnb = (D + kFinishCols - 1) / kFinishCols;                     // workgroups of the finish kernel
  k.ppart = o; o += al((size_t)S * fnb * 12 * sizeof(double));
  k.putau = o; o += al((size_t)S * fnb * KP * sizeof(float));
  k.Ap = o;    o += al(nd * D * KP * sizeof(float));
  k.Vp = o;    o += al(nd * D * KP * sizeof(float));
  k.phi = o;   o += al(nd * D * sizeof(float));
  k.dbias = o; o += al(D * sizeof(float));
  k.Vb = o;    if (c->flags & SPMF_FLAG_MIXED) o += al(D * KP * sizeof(float));
  k.bb = o;    if (c->flags & SPMF_FLAG_MIXED) o += al(D * sizeof(float));
  k.z = o;     o += al(nd * (size_t)rows * KP * sizeof(float));
  k.gzs = o;   o += al(nd * (size_t)rows * KP * sizeof(float));
  k.gzd = o;   if (c->flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED)) o += al((size_t)rows * KP * sizeof(float));
  k.est = o;
  if (uses_e_buffer(c))
    o += al((size_t)((D + 31) / 32) * 32 * (size_t)est_chunk_rows(c, rows) * sizeof(float));
  k.total = o;
  return k;
}

extern "C" {

int spmf_version(void) { return SPMF_ABI_VERSION; }
size_t spmf_sizeof_counts(void) { return sizeof(spmf_counts); }
size_t spmf_sizeof_sur_var(void) { return sizeof(spmf_sur_var); }
size_t spmf_sizeof_adam_var(void) { return sizeof(spmf_adam_var); }

static void p2p_release(spmf_ctx* c);

int spmf_ctx_create(int device, int K, int D, unsigned flags, spmf_ctx** out) {
  if (!out) return SPMF_E_ARG;
  *out = nullptr;
  if (K < 1 || K > 256 || D < 1) return SPMF_E_ARG;
  // K above 64: the whole-wave passes of widek.hip -- Poisson likelihood with the linear decoder only (the
  // dense exp / sigmoid operators of the other contexts are built for K padded to 32 or 64)
  if (K > 64 && (flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED))) return SPMF_E_UNSUPPORTED;
  if ((int64_t)D * 64 * 4 >= (1LL << 32) - 256) return SPMF_E_ARG;   // 32-bit gather offsets into [D,KP]; the last KP*4 bytes below 4 GiB are the padded slots' (common.h kPadRow)
  spmf_ctx* c = new spmf_ctx();
  c->device = device;
  c->K = K;
  c->D = D;
  c->KP = padded_k(K);
  // the dense exp kernels of the log_transform decoder work on 32-feature MFMA tiles
  if ((flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED)) && c->KP < 32) c->KP = 32;
  if ((flags & SPMF_FLAG_LOG_TRANSFORM) && (flags & SPMF_FLAG_MIXED)) {
    delete c;
    return SPMF_E_UNSUPPORTED;   // the per-column mix with the exp decoder is not built
  }
  c->flags = flags;
  if (const char* e = getenv("SPMF_DENSE_E_ONCE")) c->e_once = e[0] != '0';
  if (const char* e = getenv("SPMF_DENSE_BF16X3")) c->dense3 = e[0] != '0';
  if (const char* e = getenv("SPMF_FUSE_ROWS")) c->fuse_rows = e[0] != '0' ? 1 : 0;
  if (const char* e = getenv("SPMF_ROW_DYNAMIC")) c->dyn_rows = e[0] != '0' ? 1 : 0;
  *out = c;
  return SPMF_OK;
}

void spmf_ctx_destroy(spmf_ctx* c) {
  if (!c) return;
  for (auto& set : c->evs)
    for (auto& e : set)
      if (e) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->comm) {
    Rccl* r = rccl();
    if (r) (void)r->destroy(c->comm);
  }
  p2p_release(c);
  delete c;
}

const char* spmf_last_error(const spmf_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int spmf_ctx_set_prior(spmf_ctx* c, double u_tau_scale, double s_tau_scale, double decay) {
  if (!c || !(u_tau_scale > 0) || !(s_tau_scale > 0) || !(decay > 0)) return fail(c, SPMF_E_ARG,
      "set_prior: scales must be > 0");
  c->u_tau_scale = u_tau_scale;
  c->s_tau_scale = s_tau_scale;
  c->decay = decay;
  return SPMF_OK;
}

int spmf_padded_k(const spmf_ctx* c) { return c ? c->KP : 0; }

int spmf_ctx_set_column_types(spmf_ctx* c, const uint8_t* column_is_bernoulli) {
  if (!c || !(c->flags & SPMF_FLAG_MIXED) || !column_is_bernoulli) return fail(c, SPMF_E_ARG,
      "set_column_types: needs a ctx created with SPMF_FLAG_MIXED and a [D] device array");
  c->ctype = column_is_bernoulli;
  return SPMF_OK;
}

int spmf_ctx_set_bernoulli_columns(spmf_ctx* c, const int32_t* cols, int n) {
  if (!c || !(c->flags & SPMF_FLAG_MIXED) || n < 0 || n > c->D || (n > 0 && !cols)) return fail(c, SPMF_E_ARG,
      "set_bernoulli_columns: needs a SPMF_FLAG_MIXED ctx and n indices in [0, D)");
  c->bcols = n > 0 ? cols : nullptr;
  c->n_bcols = n;
  return SPMF_OK;
}

// Forget the carve of the previous workspace: until the next data pass binds the new one the
// entry points that only READ a bound workspace (spmf_prior_async, spmf_finish,
// spmf_nonfinite_patch, spmf_acc_ptr) fail with SPMF_E_WORKSPACE / return NULL instead of
// touching memory the caller may have freed.
static void unbind_ws(spmf_ctx* c) {
  // a prior half still running on the side stream writes per-workgroup partials into the OLD
  // workspace and the caller's gradients: let it finish before the caller may free / reuse that
  // memory (torch's caching allocator does not synchronise on free)
  if (c->prior_pending && c->ev_join) (void)hipEventSynchronize(c->ev_join);
  c->ws_rows = -1;
  c->ws_S = 0;
  c->acc = nullptr;
  c->dacc = nullptr;
  c->dprep = nullptr;
  c->fpart = nullptr;
  c->futau = nullptr;
  c->prior_pending = 0;
  c->step.active = 0;
}

int spmf_ctx_set_e_cap(spmf_ctx* c, size_t bytes) {
  if (!c || bytes < ((size_t)1 << 20)) return fail(c, SPMF_E_ARG, "set_e_cap: at least 1 MiB");
  c->est_cap_bytes = bytes;
  unbind_ws(c);               // the carve changes: the next data pass re-binds (and re-checks) the workspace
  return SPMF_OK;
}

size_t spmf_workspace_bytes(const spmf_ctx* c, int64_t max_rows, int S) {
  if (!c || max_rows < 0 || S < 1) return 0;
  size_t need = carve(c, max_rows, S).total;
  // (a batch of at most kSmallBatchRows rows keeps S table pairs: cover it too, so that every batch of up
  //  to max_rows rows fits the workspace this sizes)
  if (max_rows > kSmallBatchRows) {
    const size_t small = carve(c, kSmallBatchRows, S).total;
    if (small > need) need = small;
  }
  return need;
}

int spmf_ctx_set_workspace(spmf_ctx* c, void* workspace, size_t bytes) {
  if (!c) return SPMF_E_ARG;
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(c, SPMF_E_ARG,
      "workspace must be 256-byte aligned and non-null");
  c->ws = (char*)workspace;
  c->ws_bytes = bytes;
  unbind_ws(c);
  return SPMF_OK;
}

static int bind_ws(spmf_ctx* c, int64_t rows, int S) {
  if (!c->ws) return fail(c, SPMF_E_WORKSPACE, "no workspace set (spmf_ctx_set_workspace)");
  Carve k = carve(c, rows, S);
  if (k.total > c->ws_bytes) {
    char b[160];
    snprintf(b, sizeof b, "workspace too small: need %zu bytes for rows=%lld S=%d, have %zu", k.total,
        (long long)rows, S, c->ws_bytes);
    return fail(c, SPMF_E_WORKSPACE, b);
  }
  c->acc = (float*)(c->ws + k.acc);
  c->dacc = (double*)(c->ws + k.dacc);
  c->dprep = (double*)(c->ws + k.dprep);
  c->fpart = (double*)(c->ws + k.ppart);
  c->futau = (float*)(c->ws + k.putau);
  c->Ap = (float*)(c->ws + k.Ap);
  c->Vp = (float*)(c->ws + k.Vp);
  c->phi = (float*)(c->ws + k.phi);
  c->dbias = (float*)(c->ws + k.dbias);
  c->Vb = (float*)(c->ws + k.Vb);
  c->bb = (float*)(c->ws + k.bb);
  c->z = (float*)(c->ws + k.z);
  c->gzs = (float*)(c->ws + k.gzs);
  c->gzd = (float*)(c->ws + k.gzd);
  c->est = uses_e_buffer(c) ? (float*)(c->ws + k.est) : nullptr;
  c->est_rows = est_chunk_rows(c, rows);
  c->ws_rows = rows;
  c->ws_S = S;
  c->batched = batched_draws(c, rows, S) ? 1 : 0;
  return SPMF_OK;
}

float* spmf_acc_ptr(const spmf_ctx* c) { return c ? c->acc : nullptr; }
int64_t spmf_acc_len(const spmf_ctx* c, int S) { return c ? (int64_t)S * acc_len(c->D, c->KP) : 0; }
const float* spmf_z_ptr(const spmf_ctx* c) {
  return c ? c->z + (c->batched ? (size_t)(c->ws_S - 1) * c->ws_rows * c->KP : 0) : nullptr;
}
const float* spmf_gz_ptr(const spmf_ctx* c) {
  return c ? c->gzs + (c->batched ? (size_t)(c->ws_S - 1) * c->ws_rows * c->KP : 0) : nullptr;
}

int spmf_ctx_enable_timing(spmf_ctx* c, int on) {
  if (!c) return SPMF_E_ARG;
  if (on && !c->evs[0][0])
    for (auto& set : c->evs)
      for (auto& e : set) HIPCHK(c, hipEventCreate(&e));
  c->timing = on;
  c->ev_valid = 0;
  c->ev_set = -1;
  c->ev_count = 0;
  return SPMF_OK;
}

int spmf_last_timing(spmf_ctx* c, float* ms5) {
  if (!c || !ms5) return SPMF_E_ARG;
  const bool logt = (c->flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED)) != 0;
  if (!c->timing || c->ev_count < 1) return fail(c, SPMF_E_ARG,
      "no timing recorded (enable timing, run data_pass + finish)");
  // average over the (up to kSets) most recent complete steps
  const int n = c->ev_count < spmf_ctx::kSets ? c->ev_count : spmf_ctx::kSets;
  double acc[4] = {0, 0, 0, 0};
  for (int k = 0; k < n; ++k) {
    hipEvent_t* e = c->evs[((c->ev_set - k) % spmf_ctx::kSets + spmf_ctx::kSets) % spmf_ctx::kSets];
    HIPCHK(c, hipEventSynchronize(e[5]));
    float t;
    for (int i = 0; i < 3; ++i) {
      HIPCHK(c, hipEventElapsedTime(&t, e[i], e[i + 1]));
      acc[i] += t;
    }
    HIPCHK(c, hipEventElapsedTime(&t, e[4], e[5]));
    acc[3] += t;
  }
  for (int i = 0; i < 4; ++i) ms5[i] = (float)(acc[i] / n);
  ms5[4] = ms5[0] + ms5[1] + ms5[2] + ms5[3];
  ms5[5] = 0.f;
  if (logt) {   // the two dense launches sit inside the row interval: report them apart
    double d = 0;
    for (int k = 0; k < n; ++k) {
      hipEvent_t* e = c->evs[((c->ev_set - k) % spmf_ctx::kSets + spmf_ctx::kSets) % spmf_ctx::kSets];
      float t;
      HIPCHK(c, hipEventElapsedTime(&t, e[6], e[7]));
      d += t;
    }
    ms5[5] = (float)(d / n);
    ms5[1] -= ms5[5];
  }
  return SPMF_OK;
}

static int check_counts(spmf_ctx* c, const spmf_counts* ct) {
  if (!ct) return fail(c, SPMF_E_ARG, "counts is null");
  if (ct->struct_size != (int32_t)sizeof(spmf_counts)) {
    char b[200];
    snprintf(b, sizeof b, "counts.struct_size is %d, this library's spmf_counts has %zu bytes: the caller was "
        "built against another spmf_hip.h (ABI version %d)", (int)ct->struct_size, sizeof(spmf_counts),
        SPMF_ABI_VERSION);
    return fail(c, SPMF_E_ARG, b);
  }
  if (ct->pc_pad < 0) return fail(c, SPMF_E_ARG, "counts.pc_pad must be >= 0");
  if (ct->pc_ent && ct->panel_rows > 65536) return fail(c, SPMF_E_ARG,
      "counts.pc_ent packs the row inside its panel into 16 bits: panel_rows must be <= 65536");
  if (ct->n_cols != c->D) return fail(c, SPMF_E_ARG, "counts.n_cols != ctx D");
  if (ct->n_rows < 0 || ct->nnz < 0 || ct->nnz > 2147483647LL) return fail(c, SPMF_E_ARG,
      "counts: bad n_rows/nnz (nnz must fit int32)");
  // the kernels gather factor rows with 32-bit byte offsets: B*KP*4 must stay below 4 GiB
  if (ct->n_rows * (int64_t)c->KP * 4 >= (1LL << 32) - c->KP * 4) return fail(c, SPMF_E_ARG,
      "counts: too many rows in one batch for this K (B*KP*4 must be < 4 GiB)");
  if (!ct->row_ptr || (ct->nnz > 0 && (!ct->col_idx || !ct->val))) return fail(c, SPMF_E_ARG,
      "counts: null CSR arrays");
  if (ct->ent && c->D > 65536) return fail(c, SPMF_E_ARG,
      "counts.ent packs the column into 16 bits: D must be <= 65536");
  return SPMF_OK;
}

int spmf_counts_stats(spmf_ctx* c, int64_t n_rows, const int32_t* row_ptr, const int32_t* col_idx, const float* val,
    double* colsum, double* colnnz, float* row_sum, double* row_lgamma, void* stream) {
  if (!c || !row_ptr || n_rows < 0) return fail(c, SPMF_E_ARG, "counts_stats: bad arguments");
  if (n_rows == 0) return SPMF_OK;
  StatsArgs a{n_rows, row_ptr, col_idx, val, colsum, colnnz, row_sum, row_lgamma};
  launch_stats(a, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_counts_colstats(spmf_ctx* c, const spmf_counts* ct, double* colsum, double* colnnz, void* stream) {
  if (!c || !ct) return fail(c, SPMF_E_ARG, "counts_colstats: bad arguments");
  int rc = check_counts(c, ct);
  if (rc) return rc;
  if (ct->nnz == 0 || ct->n_rows == 0 || (!colsum && !colnnz)) return SPMF_OK;
  if (!ct->pc_ptr || !ct->pc_val || ct->n_panels < 1)
    return fail(c, SPMF_E_ARG, "counts_colstats: panel-CSC arrays missing");
  launch_colstats(ct->n_panels, ct->n_cols, ct->pc_ptr, ct->pc_val, colsum, colnnz, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_counts_gvals(spmf_ctx* c, const spmf_counts* ct, const float* eta, float* gval, float* pc_gval,
    void* stream) {
  if (!c || !ct || !eta) return fail(c, SPMF_E_ARG, "counts_gvals: bad arguments");
  int rc = check_counts(c, ct);
  if (rc) return rc;
  if (ct->nnz == 0 || ct->n_rows == 0) return SPMF_OK;
  if (pc_gval && (!ct->pc_ptr || !ct->pc_val || ct->n_panels < 1))
    return fail(c, SPMF_E_ARG, "counts_gvals: panel-CSC arrays missing");
  launch_gvals(ct->nnz, ct->n_panels, ct->n_cols, ct->row_ptr, ct->col_idx, ct->val, ct->pc_ptr, ct->pc_val, eta,
               gval, pc_gval, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

// deterministic-mode scratch, per draw: [kDetMeta + kDetMaxBlocks * (kDaccHead + KP) doubles | n_items * det_part_len floats]
static size_t det_slot_doubles(int KP) { return (size_t)kDetMeta + (size_t)kDetMaxBlocks * (kDaccHead + KP); }
static size_t det_slots_bytes(int KP) { return (det_slot_doubles(KP) * sizeof(double) + 255) & ~size_t(255); }
static size_t det_part_bytes(int KP, int64_t n_items) {
  return ((size_t)(n_items > 0 ? n_items : 0) * det_part_len(KP) * sizeof(float) + 255) & ~size_t(255);
}
size_t spmf_det_scratch_bytes(const spmf_ctx* c, int64_t n_items, int S) {
  if (!c || S < 1) return 0;
  return (size_t)S * (det_slots_bytes(c->KP) + det_part_bytes(c->KP, n_items));
}
int spmf_ctx_set_deterministic(spmf_ctx* c, void* scratch, size_t bytes) {
  if (!c) return SPMF_E_ARG;
  if (!scratch) {
    c->det_buf = nullptr;
    c->det_bytes = 0;
    return SPMF_OK;
  }
  if (likelihood_code(c) != 0) return fail(c, SPMF_E_UNSUPPORTED,
      "set_deterministic: Poisson likelihood with the linear decoder only (the dense sums of the other contexts "
      "accumulate with float atomics)");
  if (c->Dh > 0) return fail(c, SPMF_E_UNSUPPORTED, "set_deterministic: not together with the column split");
  if (c->KP > 64) return fail(c, SPMF_E_UNSUPPORTED, "set_deterministic: latent dimensions up to 64 only");
  if ((uintptr_t)scratch & 255) return fail(c, SPMF_E_ARG, "set_deterministic: scratch must be 256-byte aligned");
  if (bytes < spmf_det_scratch_bytes(c, 0, 1)) return fail(c, SPMF_E_WORKSPACE, "set_deterministic: scratch too small");
  c->det_buf = (char*)scratch;
  c->det_bytes = bytes;
  return SPMF_OK;
}

// parts: bit 0 = zero, prep, row pass and the column pass of the lower column half (all columns
// without a split); bit 1 = column pass of the upper half and the fp64 pack
static int data_pass_impl(spmf_ctx* c, const spmf_counts* ct, int S, const float* const params[SPMF_NVARS],
    const float* eta, int parts_mask, void* stream, spmf_ctx::StepOut* so = nullptr) {
  if (!c || !params || !eta || S < 1) return fail(c, SPMF_E_ARG, "data_pass: bad arguments");
  if (!so) c->step.active = 0;   // a plain data pass ends a step begun earlier
  // likelihood / decoder code of the kernels: 0 Poisson linear, 1 Poisson log_transform, 2 Bernoulli
  const int logt = likelihood_code(c);
  int rc = check_counts(c, ct);
  if (!rc && logt == 3 && !c->ctype) rc = fail(c, SPMF_E_ARG,
      "mixed likelihood: spmf_ctx_set_column_types was not called");
  if (!rc && lik_exp(logt) && ct->nnz > 0 && (!ct->gval || !ct->pc_gval)) rc = fail(c, SPMF_E_ARG,
      "counts: log_transform needs gval / pc_gval");
  if (rc) return rc;
  if (ct->n_rows > 0 && ct->nnz > 0 && (!ct->pc_row || !ct->pc_val || !ct->item_ptr || !ct->items || ct->n_panels < 1)) return fail(c, SPMF_E_ARG, "counts: panel-CSC arrays / work items missing");
  for (int i : {0, 1, 2, 7})
    if (!params[i]) return fail(c, SPMF_E_ARG, "data_pass: params v,w,u,s must be non-null");
  const bool det = c->det_buf != nullptr;
  if (det) {
    if (logt != 0 || c->Dh > 0 || parts_mask != 3) return fail(c, SPMF_E_UNSUPPORTED,
        "deterministic mode: Poisson / linear decoder without the column split only");
    if (ct->nnz > 0 && ct->n_rows > 0 && (!ct->list_first || !ct->item_pos || ct->n_items < 0)) return fail(c,
        SPMF_E_ARG, "deterministic mode: counts.list_first / item_pos / n_items missing (spmf_layout_build fills them)");
    if (spmf_det_scratch_bytes(c, ct->n_items, S) > c->det_bytes) return fail(c, SPMF_E_WORKSPACE,
        "deterministic mode: scratch smaller than spmf_det_scratch_bytes for this batch");
  }
  rc = bind_ws(c, ct->n_rows, S);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int KP = c->KP, D = c->D;
  const size_t al_ = acc_len(D, KP);
  const bool split = c->Dh > 0;
  const AccLayout L{D, KP, split ? c->Dh : D};
  if (split && ct->n_rows > 0 && ct->nnz > 0 && (!ct->item_mid || ct->col_split != c->Dh))
    return fail(c, SPMF_E_ARG, "counts: work items are not sorted for this context's column split");
  if (!split && parts_mask != 3) return fail(c, SPMF_E_ARG, "data_pass_split needs spmf_ctx_set_column_split");
  if (parts_mask != 3 && S != 1) return fail(c, SPMF_E_UNSUPPORTED, "data_pass_split: one draw per step only");
  const bool first = parts_mask & 1, second = parts_mask & 2;
  const size_t det_draw = det ? det_slots_bytes(KP) + det_part_bytes(KP, ct->n_items) : 0;
  // acc | dacc (contiguous in the carve) are zeroed by the first prep launch of the step,
  // slice by slice in its tile blocks; dprep is written, not accumulated (prep.hip)
  if (c->timing && first) {
    c->ev_set = (c->ev_set + 1) % spmf_ctx::kSets;
    c->ev = c->evs[c->ev_set];
    c->ev_valid = 0;
  }
  // batched: one pass of the loop launches every kernel once for all S draws (gridDim.y)
  const int nbat = c->batched ? S : 1;
  const int64_t dacc_stride = (int64_t)kDaccRep * (kDaccHead + KP);
  for (int s = 0; s < S; s += nbat) {
    const bool tm = c->timing && s + nbat == S;
    float* acc = c->acc + (size_t)s * al_;
    double* dacc = c->dacc + (size_t)s * dacc_stride;
    double* dprep = c->dprep + (size_t)s * kPrepSeg * (KP + 1);
    float* gVp = acc + L.gV_off(0);
    bool packed = false;
    if (first) {
    if (tm) HIPCHK(c, hipEventRecord(c->ev[0], st));
    PrepArgs pa{D, c->K, params[2] + s * var_size(c, 2), params[0] + s * var_size(c, 0), params[1] + s * var_size(c,
        1), params[7] + s * var_size(c, 7), eta, c->Ap, c->Vp, c->phi, dprep, lik_exp(logt) ? 1 : 0,
        logt == 3 ? c->ctype : nullptr, logt == 3 ? c->dbias : nullptr, nbat};
    if (s == 0) {
      pa.zero_p = c->acc;
      pa.zero_bytes = (size_t)((char*)c->dprep - (char*)c->acc);
    }
    if (so && nbat == S) {
      // spmf_step_begin with every draw in this launch: the prior half of the finish (all twelve prior
      // log-densities and prior_weight * d prior / d theta: parameters only) runs in the prep launch
      const bool hsf = (c->flags & SPMF_FLAG_ABS_HORSESHOE) != 0;
      FinishArgs fa{D, c->K, 0, 0.0, c->u_tau_scale, c->s_tau_scale, c->decay, so->prior_weight, nullptr, nullptr,
          so->params, so->eta, so->grads, so->parts, nullptr, logt, c->ctype, c->Dh, S, 0, {}, hsf ? 1 : 0,
          c->fpart, c->futau};
      for (int i = 0; i < SPMF_NVARS; ++i) fa.vstride[i] = (int64_t)var_size(c, i);
      launch_step_begin(KP, pa, fa, st);
      so->fused = 1;
    } else {
      launch_prep(KP, pa, st);
    }
    if (tm) HIPCHK(c, hipEventRecord(c->ev[1], st));
    const float* rscale = (c->flags & SPMF_FLAG_SCALE_ROWS) ? ct->row_scale : nullptr;
    if (ct->n_rows > 0 && !logt) {
      RowArgs ra{ct->n_rows, ct->row_ptr, ct->col_idx, ct->val, rscale, c->Ap, c->Vp, c->phi, dprep, c->z, c->gzs,
          dacc, 0, 0, nullptr, nullptr, nbat, D, dacc_stride};
      ra.ent = ct->ent;
      ra.dyn_tail = c->dyn_rows;
      if (det) {
        ra.det_slots = (double*)(c->det_buf + (size_t)s * det_draw);
        ra.det_stride = (int64_t)(det_draw / sizeof(double));
      }
      launch_row_pass(KP, ra, st);
    } else if (ct->n_rows > 0 && uses_sig3(c) && c->fuse_rows != 0) {
      // (SPMF_FUSE_ROWS=0: the three-launch flow below, sweep 1 -> sigdot3 -> sweep 2)
      // Bernoulli / mixed columns with the linear decoder on the bf16x3 sigmoid kernels: ONE fused row
      // pass (both sweeps read the same counts; mode 3 leaves xi_b (gz_b - [veta] - z_b) in gzs), then
      // the (Z, W) launch subtracts the dense row term in its epilogue, gzs_b -= xi_b sum_d sigmoid(l_bd) V'_d,
      // instead of encode-only sweep -> dense -> stored-cell sweep (two row launches re-stream the
      // entries and pass z through HBM: DESIGN section 4)
      RowArgs rf{ct->n_rows, ct->row_ptr, ct->col_idx, ct->val, rscale, c->Ap, c->Vp, c->phi, dprep, c->z, c->gzs,
          dacc, 3, logt, nullptr, c->ctype, 1, D, dacc_stride};
      rf.ent = ct->ent;
      rf.dyn_tail = c->dyn_rows;
      launch_row_pass(KP, rf, st);
      if (tm) HIPCHK(c, hipEventRecord(c->ev[6], st));
      const float* lbias = logt == 3 ? c->dbias : c->phi;   // mixed: -1e30 masks the Poisson columns
      float* gphi_acc = acc + L.gphi_off(0);
      const bool compact = logt == 3 && c->bcols && c->n_bcols > 0;
      const float* Wd = c->Vp;
      int Dd = D;
      const int32_t* orows = nullptr;
      if (compact) {
        launch_compact_rows(c->n_bcols, KP, c->bcols, c->Vp, c->phi, c->Vb, c->bb, st);
        Wd = c->Vb;
        Dd = c->n_bcols;
        lbias = c->bb;
        orows = c->bcols;
      }
      // Two waves per SIMD by registers: chunk counts that fill whole rounds of the resident workgroups.
      const int zt = (Dd + 127) / 128, wt = (int)((ct->n_rows + 127) / 128);
      const int rpw = sigdot3_rows_per_wg(KP), slots = 256 * sigdot3_wgs_per_cu(KP);
      const int znb = (int)((ct->n_rows + rpw - 1) / rpw), wnb = (Dd + rpw - 1) / rpw;
      const int zc = pick_chunks(znb, zt, slots, 16), wc = pick_chunks(wnb, wt, slots, 256);
      ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzs, -1.f, dacc + 3, zc, zc > 1 ? 1 : 0, 1, nullptr, lbias,
          nullptr, nullptr};
      ez.e_planes = 3;              // V' rows have mixed signs under the Normal priors: third plane of E
      ez.accumulate = 1;
      ez.p_scale = rscale;
      if (!launch_sigdot3(KP, ez, st)) return dense3_uncovered(c);   // gzs_b -= xi_b sum_d sigmoid(l_bd) V'_d ; dacc[3] = sum softplus
      ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, wc, 1, 1, lbias, nullptr, gphi_acc, orows};
      if (!launch_sigdot3(KP, ew, st)) return dense3_uncovered(c);   // gV'_d -= sum_b sigmoid z_b ; gphi_d -= sum_b sigmoid
      if (tm) HIPCHK(c, hipEventRecord(c->ev[7], st));
    } else if (ct->n_rows > 0 && [&]() {
      // Poisson log_transform on the bf16x3 exp kernels with a packed entry stream: ONE fused row pass
      // too -- sweep 1 reads g(x) (counts.gval), sweep 2 takes the count out of the packed word
      // (RowArgs.dual: the register cost of the canonical (col, val) pair) -- and the (Z, W) launch
      // subtracts xi_b sum_d E_bd V'_d from gzs in its epilogue.  Only the LDS-phi launch shapes
      // have the two-stream form: false = nothing was launched, take the three-launch flow below.
      if (!uses_dense3(c) || !ct->ent || !ct->gval || c->fuse_rows != 1) return false;
      RowArgs rf{ct->n_rows, ct->row_ptr, ct->col_idx, ct->gval, rscale, c->Ap, c->Vp, c->phi, dprep, c->z, c->gzs,
          dacc, 3, logt, nullptr, c->ctype, 1, D, dacc_stride};
      rf.ent = ct->ent;
      rf.dual = 1;
      return launch_row_pass(KP, rf, st);
    }()) {
      if (tm) HIPCHK(c, hipEventRecord(c->ev[6], st));
      ExpdotArgs ez{(int)ct->n_rows, D, c->z, c->Vp, c->gzs, -1.f, dacc + 3, 1, 0, 0, nullptr, nullptr, nullptr, nullptr};
      ez.accumulate = 1;
      ez.p_scale = rscale;
      if (!launch_expdot3(KP, ez, st)) return dense3_uncovered(c);   // gzs_b -= xi_b sum_d E_bd V'_d ; dacc[3] = sum E
      const int ch3 = pick_chunks((D + expdot3_rows_per_wg() - 1) / expdot3_rows_per_wg(),
                                  (int)((ct->n_rows + 127) / 128), 256 * expdot3_wgs_per_cu(), 64);
      ExpdotArgs ew{D, (int)ct->n_rows, c->Vp, c->z, gVp, -1.f, nullptr, ch3, 1, 0, nullptr, nullptr, nullptr, nullptr};
      if (!launch_expdot3(KP, ew, st)) return dense3_uncovered(c);   // gV'_d -= sum_b E_bd z_b
      if (tm) HIPCHK(c, hipEventRecord(c->ev[7], st));
    } else if (ct->n_rows > 0) {
      // log_transform: z from g(x) (sweep 1), dense exp terms on the matrix
      // cores, then the stored-cell terms (sweep 2) with the dense row term.
      RowArgs r1{ct->n_rows, ct->row_ptr, ct->col_idx, lik_exp(logt) ? ct->gval : ct->val, rscale, c->Ap, c->Vp, c->phi,
          dprep, c->z, c->gzs, dacc, 1, logt, nullptr, c->ctype, 1, D, dacc_stride};
      if (!lik_exp(logt)) r1.ent = ct->ent;   // (the exp encoders read g(x), not the counts)
      launch_row_pass(KP, r1, st);
      if (tm) HIPCHK(c, hipEventRecord(c->ev[6], st));
      const int act = logt == 4 ? 2 : (logt >= 2 ? 1 : 0);   // dense.hip ACT
      const float* lbias = logt == 3 ? c->dbias : c->phi;   // mixed: -1e30 masks the Poisson columns
      float* gphi_acc = acc + L.gphi_off(0);
      // mixed likelihood with the Bernoulli column list set: the dense sums run over those
      // columns only (compacted V' rows), instead of over all D with the Poisson half masked
      const bool compact = logt == 3 && c->bcols && c->n_bcols > 0;
      const float* Wd = c->Vp;
      int Dd = D;
      const int32_t* orows = nullptr;
      if (compact) {
        launch_compact_rows(c->n_bcols, KP, c->bcols, c->Vp, c->phi, c->Vb, c->bb, st);
        Wd = c->Vb;
        Dd = c->n_bcols;
        lbias = c->bb;
        orows = c->bcols;
      }
      if (uses_dense3(c) && !compact) {
        // bf16x3: E is recomputed by the second launch (at this matrix rate a B*D*4-byte round
        // trip through HBM would be the bound)
        ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzd, 1.f, dacc + 3, 1, 0, 0, nullptr, nullptr, nullptr,
            nullptr};
        if (!launch_expdot3(KP, ez, st)) return dense3_uncovered(c);   // gzd_b = sum_d E_bd V'_d ; dacc[3] = sum E
        // Q chunks of the W-stationary launch: whole rounds of the resident workgroups (one 110 KB
        // workgroup per CU: 118 column blocks x 13 chunks = 6 rounds of 256 on C4; 5 chunks = 590
        // workgroups ran 2.3 rounds, the last one a third full)
        const int ch3 = pick_chunks((Dd + expdot3_rows_per_wg() - 1) / expdot3_rows_per_wg(),
                                    (int)((ct->n_rows + 127) / 128), 256 * expdot3_wgs_per_cu(), 64);
        ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, ch3, 1, 0, nullptr, nullptr, nullptr,
            nullptr};
        if (!launch_expdot3(KP, ew, st)) return dense3_uncovered(c);   // gV'_d -= sum_b E_bd z_b
      } else if (uses_exp3_32(c) && !compact) {
        // the same two launches at K padded to 32 (sigdot3 family, ACT 0); chunk counts that fill whole
        // rounds of the resident workgroups
        const int zt = (Dd + 127) / 128, wt = (int)((ct->n_rows + 127) / 128);
        const int rpw = sigdot3_rows_per_wg(KP), slots = 256 * sigdot3_wgs_per_cu(KP);
        const int zc = pick_chunks((int)((ct->n_rows + rpw - 1) / rpw), zt, slots, 16);
        const int wc = pick_chunks((Dd + rpw - 1) / rpw, wt, slots, 256);
        if (zc > 1) launch_zero(c->gzd, (size_t)ct->n_rows * KP * sizeof(float), st);
        ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzd, 1.f, dacc + 3, zc, zc > 1 ? 1 : 0, 0, nullptr, nullptr,
            nullptr, nullptr};
        if (!launch_sigdot3(KP, ez, st)) return dense3_uncovered(c);   // gzd_b = sum_d E_bd V'_d ; dacc[3] = sum E ; dacc[4]: saturation
        ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, wc, 1, 0, nullptr, nullptr, nullptr, nullptr};
        if (!launch_sigdot3(KP, ew, st)) return dense3_uncovered(c);   // gV'_d -= sum_b E_bd z_b
      } else if (uses_sig3(c)) {
        // the sigmoid / softplus sums on the bf16x3 kernels without the fused row pass (SPMF_FUSE_ROWS=0):
        // (Z, W) writes gzd_b = sum_d sigmoid(l_bd) V'_d for the stored-cell sweep, (W, Z) as in the fused flow
        const int zt = (Dd + 127) / 128, wt = (int)((ct->n_rows + 127) / 128);
        const int rpw = sigdot3_rows_per_wg(KP), slots = 256 * sigdot3_wgs_per_cu(KP);
        const int zc = pick_chunks((int)((ct->n_rows + rpw - 1) / rpw), zt, slots, 16);
        const int wc = pick_chunks((Dd + rpw - 1) / rpw, wt, slots, 256);
        if (zc > 1) launch_zero(c->gzd, (size_t)ct->n_rows * KP * sizeof(float), st);
        ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzd, 1.f, dacc + 3, zc, zc > 1 ? 1 : 0, 1, nullptr, lbias,
            nullptr, nullptr};
        ez.e_planes = 3;
        if (!launch_sigdot3(KP, ez, st)) return dense3_uncovered(c);
        ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, wc, 1, 1, lbias, nullptr, gphi_acc, orows};
        if (!launch_sigdot3(KP, ew, st)) return dense3_uncovered(c);
      } else if (uses_sigexp3(c) && !compact) {
        // Bernoulli + log_transform at K padded to 32: (Z, W) with the bias on the Q rows, the softplus sum and
        // three planes of E = sigmoid exp (V' has mixed signs), (W, Z) with the bias on the P rows and the
        // sigmoid row sums for d/dphi
        const int zt = (Dd + 127) / 128, wt = (int)((ct->n_rows + 127) / 128);
        const int rpw = sigdot3_rows_per_wg(KP), slots = 256 * sigdot3_wgs_per_cu(KP);
        const int zc = pick_chunks((int)((ct->n_rows + rpw - 1) / rpw), zt, slots, 16);
        const int wc = pick_chunks((Dd + rpw - 1) / rpw, wt, slots, 256);
        if (zc > 1) launch_zero(c->gzd, (size_t)ct->n_rows * KP * sizeof(float), st);
        ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzd, 1.f, dacc + 3, zc, zc > 1 ? 1 : 0, 2, nullptr, lbias,
            nullptr, nullptr};
        ez.e_planes = 3;
        if (!launch_sigdot3(KP, ez, st)) return dense3_uncovered(c);   // gzd_b = sum_d sigmoid(l) exp(X) V'_d ; dacc[3] = sum softplus(l)
        ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, wc, 1, 2, lbias, nullptr, gphi_acc, orows};
        if (!launch_sigdot3(KP, ew, st)) return dense3_uncovered(c);   // gV'_d -= sum_b E_bd z_b ; gphi_d -= sum_b sigmoid(l)
      } else if (c->est && act != 2) {   // (act 2: E carries exp(X) too, its row sums are not the d/dphi sums)
        // E once: per row chunk, the Z-stationary kernel keeps E (exp, or the sigmoid of the
        // Bernoulli logits) and the second contraction (gV'_d -= sum_b E_bd z_b; Bernoulli:
        // gphi_d -= sum_b E_bd too) reads it back instead of recomputing it
        // (chunks of equal size, whole 128-row workgroups: a short last chunk would run the
        //  chip half empty)
        const int64_t nch = (ct->n_rows + c->est_rows - 1) / c->est_rows;
        int64_t step = ((ct->n_rows + nch - 1) / nch + 127) / 128 * 128;
        if (step > c->est_rows) step = c->est_rows;
        for (int64_t r0 = 0; r0 < ct->n_rows; r0 += step) {
          const int nr = (int)((ct->n_rows - r0) < step ? (ct->n_rows - r0) : step);
          ExpdotArgs ez{nr, Dd, c->z + (size_t)r0 * KP, Wd, c->gzd + (size_t)r0 * KP, 1.f, dacc + 3, 1, 0, act,
              nullptr, act ? lbias : nullptr, nullptr, nullptr, c->est, (int64_t)nr};
          launch_expdot(KP, ez, st);
          launch_estdot(KP, Dd, nr, (int64_t)nr, c->est, c->z + (size_t)r0 * KP, gVp, -1.f,
                        act ? gphi_acc : nullptr, orows, st);
        }
      } else {
      // Z-stationary: Q rows are columns d -> bias_q = phi (Bernoulli logits)
      ExpdotArgs ez{(int)ct->n_rows, Dd, c->z, Wd, c->gzd, 1.f, dacc + 3, 1, 0, act, nullptr,
          act ? lbias : nullptr, nullptr, nullptr};
      launch_expdot(KP, ez, st);   // gzd_b = sum_d E_bd V'_d ; dacc[3] = sum E (or sum softplus)
      // W-stationary launch has only D/128 workgroups: split the row (Q) range
      // into chunks until ~4 workgroups per CU are in flight
      const int nbx = (Dd + 127) / 128;
      const int qtiles = (int)((ct->n_rows + 127) / 128);
      int chunks = (1024 + nbx - 1) / nbx;
      if (chunks > qtiles) chunks = qtiles;
      if (chunks < 1) chunks = 1;
      // W-stationary: P rows are columns d -> bias_p = phi; Bernoulli also needs the
      // column sums of sigmoid for d/dphi (subtracted from the gphi accumulators)
      ExpdotArgs ew{Dd, (int)ct->n_rows, Wd, c->z, gVp, -1.f, nullptr, chunks, 1, act, act ? lbias : nullptr,
          nullptr, act ? gphi_acc : nullptr, orows};
      launch_expdot(KP, ew, st);   // gV'_d -= sum_b E_bd z_b
      }
      if (tm) HIPCHK(c, hipEventRecord(c->ev[7], st));
      RowArgs r2{ct->n_rows, ct->row_ptr, ct->col_idx, ct->val, rscale, c->Ap, c->Vp, c->phi, dprep, c->z, c->gzs,
          dacc, 2, logt, c->gzd, c->ctype, 1, D, dacc_stride};
      r2.ent = ct->ent;
      r2.dyn_tail = c->dyn_rows;   // (the encode-only launch above never touches the counters: this one has them alone)
      launch_row_pass(KP, r2, st);
    }
    if (tm) HIPCHK(c, hipEventRecord(c->ev[2], st));
    // the caller's marker "the row stage of the last draw has been issued": what it makes wait for this event
    // runs beside the column pass instead of beside the resident-set row launch (spmf_ctx_set_rows_event)
    if (c->rows_event && s + nbat == S) HIPCHK(c, hipEventRecord(c->rows_event, st));
    }   // first
    if (ct->n_rows > 0 && ct->nnz > 0) {
      for (int hf = 0; hf < 2; ++hf) {
        if (!(hf == 0 ? first : second)) continue;
        if (!split && hf == 1) continue;
        ColArgs ca{D, ct->n_panels, ct->row_base, split ? ct->max_items_half[hf] : ct->max_items_per_panel,
            ct->item_ptr, ct->items, ct->pc_row, ct->pc_val, c->Vp, c->phi, c->z, c->gzs, acc + L.gA_off(hf),
            acc + L.gV_off(hf), acc + L.gphi_off(hf), logt, ct->pc_gval, c->ctype, split ? ct->item_mid : nullptr,
            split ? hf + 1 : 0, nbat, ct->n_rows, (int64_t)al_, ct->pc_pad};
        ca.pc_ent = ct->pc_ent;
        ca.panel_rows = ct->panel_rows;
        if (det) {
          ca.det_slots = (const double*)(c->det_buf + (size_t)s * det_draw);
          ca.det_stride = (int64_t)(det_draw / sizeof(double));
          ca.det_part = (float*)(c->det_buf + (size_t)s * det_draw + det_slots_bytes(KP));
          ca.det_part_stride = (int64_t)(det_draw / sizeof(float));
        }
        if (hf == (split ? 1 : 0)) {
          // the fp64 scalars of the row pass are complete before this launch starts: its
          // extra first block folds them into the accumulator tail (the former pack launch)
          ca.pack_dacc = dacc;
          ca.pack_tail = acc + L.tail_off();
          ca.dacc_stride = dacc_stride;
          packed = launch_col_pass(KP, ca, st);
        } else {
          launch_col_pass(KP, ca, st);
        }
      }
    }
    if (det && ct->n_rows > 0 && ct->nnz > 0) {
      // the per-item partial sums, column by column in (panel, segment) order, into the zeroed accumulators
      DetReduceArgs dr{D, KP, ct->n_panels, nbat, ct->list_first, ct->item_pos, ct->item_ptr,
          (const float*)(c->det_buf + (size_t)s * det_draw + det_slots_bytes(KP)), (int64_t)(det_draw / sizeof(float)),
          acc + L.gA_off(0), acc + L.gV_off(0), acc + L.gphi_off(0), (int64_t)al_};
      launch_det_reduce(dr, st);
    }
    if (second) {
      if (!packed) {
        PackArgs pk{KP, dacc, acc + L.tail_off(), nbat, dacc_stride, (int64_t)al_};
        launch_pack(pk, st);
      }
      if (tm) {
        HIPCHK(c, hipEventRecord(c->ev[3], st));
        c->ev_valid = 1;
      }
    }
  }
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_data_pass(spmf_ctx* c, const spmf_counts* ct, int S, const float* const params[SPMF_NVARS],
    const float* eta, void* stream) {
  return data_pass_impl(c, ct, S, params, eta, 3, stream);
}

int spmf_data_pass_split(spmf_ctx* c, const spmf_counts* ct, int S, const float* const params[SPMF_NVARS],
    const float* eta, int part, void* stream) {
  if (part != 0 && part != 1) return fail(c, SPMF_E_ARG, "data_pass_split: part must be 0 or 1");
  return data_pass_impl(c, ct, S, params, eta, part == 0 ? 1 : 2, stream);
}

int spmf_ctx_set_rows_event(spmf_ctx* c, void* event) {
  if (!c) return SPMF_E_ARG;
  c->rows_event = (hipEvent_t)event;
  return SPMF_OK;
}

int spmf_ctx_set_column_split(spmf_ctx* c, int Dh) {
  if (!c) return SPMF_E_ARG;
  if (Dh == 0 || Dh == c->D) {
    c->Dh = 0;
    return SPMF_OK;
  }
  if (Dh < 0 || Dh > c->D || (Dh % 32) != 0) return fail(c, SPMF_E_ARG,
      "column split must be a multiple of 32 inside (0, D)");
  if (c->flags & (SPMF_FLAG_LOG_TRANSFORM | SPMF_FLAG_BERNOULLI | SPMF_FLAG_MIXED))
    return fail(c, SPMF_E_UNSUPPORTED, "column split: linear Poisson decoder only");
  c->Dh = Dh;
  return SPMF_OK;
}

int spmf_acc_split(const spmf_ctx* c, int64_t off[2], int64_t len[2]) {
  if (!c || !off || !len) return SPMF_E_ARG;
  const AccLayout L{c->D, c->KP, c->Dh > 0 ? c->Dh : c->D};
  off[0] = 0;
  len[0] = L.half_len(0);
  off[1] = len[0];
  len[1] = acc_len(c->D, c->KP) - len[0];
  return SPMF_OK;
}


int spmf_prior_async(spmf_ctx* c, int S, double prior_weight, const float* const params[SPMF_NVARS],
    const float* eta, double* parts, float* const grads[SPMF_NVARS], void* stream) {
  if (!c || !params || !grads || !eta || !parts || S < 1) return fail(c, SPMF_E_ARG, "prior_async: bad arguments");
  const bool hsf = (c->flags & SPMF_FLAG_ABS_HORSESHOE) != 0;
  for (int i = 0; i < SPMF_NVARS; ++i)
    if ((!params[i] || !grads[i]) && !(hsf && i != 0 && i != 1 && i != 2 && i != 7))
      return fail(c, SPMF_E_ARG, "prior_async: params/grads must be non-null (all 12; v,w,u,s with ABS_HORSESHOE)");
  if (!c->fpart) return fail(c, SPMF_E_WORKSPACE, "prior_async: no data pass has bound the workspace yet");
  hipStream_t st = (hipStream_t)stream;
  if (!c->side) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  // the side stream forks off `stream` (outputs need no zero fill: single writers)
  HIPCHK(c, hipEventRecord(c->ev_fork, st));
  HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
  {
    // one launch for all S draws (gridDim.y)
    FinishArgs fa{c->D, c->K, 0, 0.0, c->u_tau_scale, c->s_tau_scale, c->decay, prior_weight, nullptr, nullptr,
        params, eta, grads, parts, nullptr, likelihood_code(c), c->ctype, c->Dh, S, 0, {}, hsf ? 1 : 0,
        c->fpart, c->futau};
    for (int i = 0; i < SPMF_NVARS; ++i) fa.vstride[i] = (int64_t)var_size(c, i);
    launch_finish(c->KP, fa, 1, c->side);
  }
  HIPCHK(c, hipEventRecord(c->ev_join, c->side));
  HIPCHK(c, hipGetLastError());
  c->prior_pending = S;
  c->prior_parts = parts;
  return SPMF_OK;
}

int spmf_finish(spmf_ctx* c, int S, int64_t n_rows_global, double lgamma_sum_global, double prior_weight,
    const float* const params[SPMF_NVARS], const float* eta, double* parts, float* const grads[SPMF_NVARS],
    double* n_nonfinite, void* stream) {
  if (!c || !params || !grads || !eta || !parts || S < 1) return fail(c, SPMF_E_ARG, "finish: bad arguments");
  if (!c->acc || c->ws_S < S) return fail(c, SPMF_E_ARG, "finish: no data pass precedes it for this S");
  const bool hsf = (c->flags & SPMF_FLAG_ABS_HORSESHOE) != 0;
  for (int i = 0; i < SPMF_NVARS; ++i)
    if ((!params[i] || !grads[i]) && !(hsf && i != 0 && i != 1 && i != 2 && i != 7))
      return fail(c, SPMF_E_ARG, "finish: params/grads must be non-null (all 12; v,w,u,s with ABS_HORSESHOE)");
  hipStream_t st = (hipStream_t)stream;
  const int KP = c->KP, D = c->D;
  const size_t al_ = acc_len(D, KP);
  // the prior half may already be running on the side stream (spmf_prior_async
  // with these outputs): join it and add the data half only
  const bool joined = c->prior_pending == S && c->prior_parts == parts;
  c->prior_pending = 0;
  c->prior_parts = nullptr;
  if (joined) {
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
  }
  // (no zero fill: finish_reduce_kernel writes the twelve prior parts and the u_tau gradient
  //  whole, the data half stores parts 12 and 13 -- single writers)
  {
    // one launch for all S draws (gridDim.y)
    const bool tm = c->timing;
    FinishArgs fa{D, c->K, n_rows_global, lgamma_sum_global, c->u_tau_scale, c->s_tau_scale, c->decay, prior_weight,
        c->acc, c->dprep, params, eta, grads, parts, n_nonfinite, likelihood_code(c), c->ctype, c->Dh, S,
        (int64_t)al_, {}, hsf ? 1 : 0, c->fpart, c->futau};
    for (int i = 0; i < SPMF_NVARS; ++i) fa.vstride[i] = (int64_t)var_size(c, i);
    if (tm) HIPCHK(c, hipEventRecord(c->ev[4], st));
    launch_finish(KP, fa, joined ? 2 : 0, st);
    if (tm) {
      HIPCHK(c, hipEventRecord(c->ev[5], st));
      if (c->ev_valid == 1) {
        c->ev_valid = 3;
        c->ev_count++;
      }
    }
  }
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

// ---- the step with its outputs known up front (ABI 6) -------------------------
int spmf_step_begin(spmf_ctx* c, const spmf_counts* ct, int S, double prior_weight,
    const float* const params[SPMF_NVARS], const float* eta, double* parts, float* const grads[SPMF_NVARS],
    double* n_nonfinite, void* stream) {
  if (!c || !params || !grads || !eta || !parts || S < 1) return fail(c, SPMF_E_ARG, "step_begin: bad arguments");
  const bool hsf = (c->flags & SPMF_FLAG_ABS_HORSESHOE) != 0;
  for (int i = 0; i < SPMF_NVARS; ++i)
    if ((!params[i] || !grads[i]) && !(hsf && i != 0 && i != 1 && i != 2 && i != 7))
      return fail(c, SPMF_E_ARG, "step_begin: params/grads must be non-null (all 12; v,w,u,s with ABS_HORSESHOE)");
  spmf_ctx::StepOut so;
  so.S = S;
  so.prior_weight = prior_weight;
  for (int i = 0; i < SPMF_NVARS; ++i) {
    so.params[i] = params[i];
    so.grads[i] = grads[i];
  }
  so.eta = eta;
  so.parts = parts;
  so.nnf = n_nonfinite;
  c->step.active = 0;
  const int rc = data_pass_impl(c, ct, S, params, eta, 3, stream, &so);
  if (rc) return rc;
  so.active = 1;
  c->step = so;
  return SPMF_OK;
}

int spmf_step_end(spmf_ctx* c, int64_t n_rows_global, double lgamma_sum_global, void* stream) {
  if (!c) return SPMF_E_ARG;
  if (!c->step.active || !c->acc) return fail(c, SPMF_E_ARG, "step_end: no spmf_step_begin precedes it");
  spmf_ctx::StepOut& so = c->step;
  so.active = 0;
  if (!so.fused)   // the draws ran in turn (S > 1 on a large batch): the whole finish, as spmf_finish runs it
    return spmf_finish(c, so.S, n_rows_global, lgamma_sum_global, so.prior_weight, so.params, so.eta, so.parts,
                       so.grads, so.nnf, stream);
  hipStream_t st = (hipStream_t)stream;
  const bool hsf = (c->flags & SPMF_FLAG_ABS_HORSESHOE) != 0;
  const bool tm = c->timing;
  FinishArgs fa{c->D, c->K, n_rows_global, lgamma_sum_global, c->u_tau_scale, c->s_tau_scale, c->decay,
      so.prior_weight, c->acc, c->dprep, so.params, so.eta, so.grads, so.parts, so.nnf, likelihood_code(c),
      c->ctype, c->Dh, so.S, (int64_t)acc_len(c->D, c->KP), {}, hsf ? 1 : 0, c->fpart, c->futau};
  for (int i = 0; i < SPMF_NVARS; ++i) fa.vstride[i] = (int64_t)var_size(c, i);
  if (tm) HIPCHK(c, hipEventRecord(c->ev[4], st));
  launch_step_end(c->KP, fa, st);   // data half + the fold of the prior half's per-block sums
  if (tm) {
    HIPCHK(c, hipEventRecord(c->ev[5], st));
    if (c->ev_valid == 1) {
      c->ev_valid = 3;
      c->ev_count++;
    }
  }
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_elbo_fwd_bwd(spmf_ctx* c, const spmf_counts* ct, int S, double prior_weight,
    const float* const params[SPMF_NVARS], const float* eta, double* parts, float* const grads[SPMF_NVARS],
    double* n_nonfinite, void* stream) {
  int rc = spmf_step_begin(c, ct, S, prior_weight, params, eta, parts, grads, n_nonfinite, stream);
  if (rc) return rc;
  return spmf_step_end(c, ct->n_rows, ct->lgamma_sum, stream);
}

int spmf_encode(spmf_ctx* c, const spmf_counts* ct, const float* u, const float* s, const float* eta, float* z_out,
    void* stream) {
  if (!c || !u || !s || !eta || !z_out) return fail(c, SPMF_E_ARG, "encode: bad arguments");
  const int logt = (c->flags & SPMF_FLAG_LOG_TRANSFORM) ? 1 : 0;
  int rc = check_counts(c, ct);
  if (!rc && logt && ct->nnz > 0 && !ct->gval) rc = fail(c, SPMF_E_ARG, "encode: log_transform needs counts.gval");
  if (rc) return rc;
  if (ct->n_rows == 0) return SPMF_OK;
  rc = bind_ws(c, ct->n_rows, 1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  PrepArgs pa{c->D, c->K, u, nullptr, nullptr, s, eta, c->Ap, c->Vp, c->phi, c->dprep, logt, nullptr, nullptr};
  launch_prep(c->KP, pa, st);
  RowArgs ra{ct->n_rows, ct->row_ptr, ct->col_idx, logt ? ct->gval : ct->val,
      (c->flags & SPMF_FLAG_SCALE_ROWS) ? ct->row_scale : nullptr, c->Ap, c->Vp, c->phi, c->dprep, c->z, c->gzs,
      c->dacc, 1, logt, nullptr, nullptr, 1, c->D, 0};
  if (!logt) ra.ent = ct->ent;
  launch_row_pass(c->KP, ra, st);
  HIPCHK(c, hipMemcpy2DAsync(z_out, (size_t)c->K * sizeof(float), c->z, (size_t)c->KP * sizeof(float),
      (size_t)c->K * sizeof(float), (size_t)ct->n_rows, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_dense_ll(spmf_ctx* c, const spmf_counts* ct, const float* u, const float* v, const float* w,
    const float* s, const float* eta, float* rate_out, float* ll_out, void* stream) {
  if (!c || !u || !v || !w || !s || !eta || !rate_out || !ll_out) return fail(c, SPMF_E_ARG,
      "dense_ll: bad arguments");
  const int lik = likelihood_code(c);
  if (lik == 3 && !c->ctype) return fail(c, SPMF_E_ARG, "dense_ll: spmf_ctx_set_column_types was not called");
  const int logt = lik_exp(lik) ? 1 : 0;
  int rc = check_counts(c, ct);
  if (!rc && logt && ct->nnz > 0 && !ct->gval) rc = fail(c, SPMF_E_ARG, "dense_ll: log_transform needs counts.gval");
  if (rc) return rc;
  if (ct->n_rows == 0) return SPMF_OK;
  rc = bind_ws(c, ct->n_rows, 1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  PrepArgs pa{c->D, c->K, u, v, w, s, eta, c->Ap, c->Vp, c->phi, c->dprep, logt, nullptr, nullptr};
  launch_prep(c->KP, pa, st);
  RowArgs ra{ct->n_rows, ct->row_ptr, ct->col_idx, logt ? ct->gval : ct->val,
      (c->flags & SPMF_FLAG_SCALE_ROWS) ? ct->row_scale : nullptr, c->Ap, c->Vp, c->phi, c->dprep, c->z, c->gzs,
      c->dacc, 1, logt, nullptr, nullptr, 1, c->D, 0};
  if (!logt) ra.ent = ct->ent;
  launch_row_pass(c->KP, ra, st);
  DenseLLArgs da{ct->n_rows, c->D, lik, c->z, c->Vp, c->phi, c->ctype, ct->row_ptr, ct->col_idx, ct->val, rate_out,
      ll_out};
  launch_dense_ll(c->KP, da, st);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_nonfinite_reduce(spmf_ctx* c, int64_t n, const float* ll, int pass, double* io, void* stream) {
  if (!c || !ll || !io || n < 0 || (pass != 0 && pass != 1)) return fail(c, SPMF_E_ARG,
      "nonfinite_reduce: bad arguments");
  if (n == 0) return SPMF_OK;
  launch_nonfinite(n, ll, pass, io, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

// ---- row-shard collective inside the library (spmf_allreduce) ---------
// (1) hand-written, over peer pointers: p2p.hip.  Region layout of a rank:
//     rs[2][kP2PMaxWorld][slice_cap] | ag[2][kP2PMaxWorld][slice_cap] | flags[2][2][kP2PMaxWorld][kP2PMaxChunks] | seq[8]
static void p2p_unmap(spmf_ctx* c) {
  spmf_ctx::P2P& p = c->p2p;
  for (int i = 0; i < p.world; ++i) {
    if (p.peer[i] && p.peer[i] != p.region) (void)hipIpcCloseMemHandle(p.peer[i]);
    p.peer[i] = nullptr;
  }
  p.connected = 0;
}
static void p2p_release(spmf_ctx* c) {
  spmf_ctx::P2P& p = c->p2p;
  p2p_unmap(c);
  if (p.region) (void)hipFree(p.region);
  p = spmf_ctx::P2P();
}

static int p2p_allreduce(spmf_ctx* c, float* buf, int64_t n, hipStream_t st) {
  spmf_ctx::P2P& p = c->p2p;
  if (n > p.n_max) {
    char b[160];
    snprintf(b, sizeof b, "allreduce: %lld floats, the peer regions were sized for %lld (spmf_p2p_init n_max)",
        (long long)n, (long long)p.n_max);
    return fail(c, SPMF_E_WORKSPACE, b);
  }
  if (((uintptr_t)buf & 15) != 0) return fail(c, SPMF_E_ARG, "allreduce: buffer must be 16-byte aligned");
  if (n == 0 || p.world == 1) return SPMF_OK;
  P2PLaunch L{};
  L.buf = buf;
  L.n = n;
  L.rank = p.rank;
  L.world = p.world;
  L.nchunk = p.nchunk;
  L.slice_cap = p.slice_cap;
  L.rs = (float*)p.region;
  L.ag = (float*)(p.region + p.off_ag);
  L.flags = (uint64_t*)(p.region + p.off_flags);
  L.seq = (uint64_t*)(p.region + p.off_seq);
  for (int i = 0; i < p.world; ++i) {
    L.peer_rs[i] = (float*)p.peer[i];
    L.peer_ag[i] = (float*)(p.peer[i] + p.off_ag);
    L.peer_flags[i] = (uint64_t*)(p.peer[i] + p.off_flags);
  }
  launch_p2p_allreduce(L, st);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_p2p_init(spmf_ctx* c, int rank, int world, int64_t n_max, int nchunk, void* handle_out64) {
  if (!c || !handle_out64 || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world || n_max < 1)
    return fail(c, SPMF_E_ARG, "p2p_init: bad arguments (world <= 16)");
  if (nchunk <= 0) nchunk = 32;
  if (nchunk > kP2PMaxChunks) nchunk = kP2PMaxChunks;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  spmf_ctx::P2P& p = c->p2p;
  const int64_t per = ((n_max + world - 1) / world + 3) & ~(int64_t)3;
  const int64_t slice_cap = (per + 63) & ~(int64_t)63;           // 256-byte slots
  const size_t box = (size_t)2 * kP2PMaxWorld * slice_cap * sizeof(float);
  const size_t off_flags = 2 * box;
  const size_t off_seq = off_flags + (size_t)2 * 2 * kP2PMaxWorld * kP2PMaxChunks * sizeof(uint64_t);
  const size_t bytes = off_seq + 64;
  // A context's region is allocated ONCE and re-used while it is large enough: freeing a region and exporting a
  // new allocation inside one process is what the runtime's IPC bookkeeping did not survive (round 5, world 4:
  // "hipIpcGetMemHandle: invalid argument", or peers that mapped something stale and never saw a flag).
  p2p_unmap(c);
  char* keep = (p.region && p.alloc_bytes >= bytes) ? p.region : nullptr;
  const size_t keep_bytes = keep ? p.alloc_bytes : 0;
  if (p.region && !keep) (void)hipFree(p.region);
  p = spmf_ctx::P2P();
  p.rank = rank;
  p.world = world;
  p.nchunk = nchunk;
  p.n_max = n_max;
  p.slice_cap = slice_cap;
  p.off_ag = box;
  p.off_flags = off_flags;
  p.off_seq = off_seq;
  p.bytes = bytes;
  hipError_t e = hipSuccess;
  if (keep) {
    p.region = keep;
    p.alloc_bytes = keep_bytes;
  } else {
    // fine-grained: coherent across agents INSIDE a kernel (a coarse-grained allocation is only
    // coherent at kernel boundaries); what RCCL allocates for its own buffers
    void* reg = nullptr;
    e = hipExtMallocWithFlags(&reg, p.bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(c, SPMF_E_HIP, std::string("p2p_init: hipExtMallocWithFlags(fine-grained): ") + hipGetErrorString(e));
    }
    p.region = (char*)reg;
    p.alloc_bytes = p.bytes;
  }
  HIPCHK(c, hipMemset(p.region + p.off_flags, 0, p.bytes - p.off_flags));
  HIPCHK(c, hipDeviceSynchronize());
  hipIpcMemHandle_t hnd;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "spmf_p2p_init hands out 64-byte handles");
  e = hipIpcGetMemHandle(&hnd, p.region);
  if (e != hipSuccess) {
    // seen once (round 5, world 4, a region re-created in a process whose earlier region a peer had still mapped
    // when it was freed): the runtime refused to export the new allocation.  One more try at another address.
    (void)hipGetLastError();
    void* again = nullptr;
    if (hipExtMallocWithFlags(&again, p.bytes, hipDeviceMallocFinegrained) == hipSuccess) {
      (void)hipFree(p.region);
      p.region = (char*)again;
      p.alloc_bytes = p.bytes;
      (void)hipMemset(p.region + p.off_flags, 0, p.bytes - p.off_flags);
      (void)hipDeviceSynchronize();
      e = hipIpcGetMemHandle(&hnd, p.region);
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    p2p_release(c);
    return fail(c, SPMF_E_HIP, std::string("p2p_init: hipIpcGetMemHandle: ") + hipGetErrorString(e));
  }
  memcpy(handle_out64, &hnd, 64);
  return SPMF_OK;
}

int spmf_p2p_connect(spmf_ctx* c, const void* handles) {
  if (!c || !handles) return fail(c, SPMF_E_ARG, "p2p_connect: bad arguments");
  spmf_ctx::P2P& p = c->p2p;
  if (!p.region) return fail(c, SPMF_E_ARG, "p2p_connect: spmf_p2p_init was not called");
  HIPCHK(c, hipSetDevice(c->device));
  for (int i = 0; i < p.world; ++i) {
    if (i == p.rank) {
      p.peer[i] = p.region;
      continue;
    }
    hipIpcMemHandle_t hnd;
    memcpy(&hnd, (const char*)handles + (size_t)i * 64, 64);
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, hnd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      char b[200];
      snprintf(b, sizeof b, "p2p_connect: hipIpcOpenMemHandle(rank %d): %s", i, hipGetErrorString(e));
      return fail(c, SPMF_E_HIP, b);
    }
    p.peer[i] = (char*)ptr;
    // a region that lives on ANOTHER device of this process's view (ranks = GPUs of one node): kernels here will
    // store into it, so the link must be there -- an inaccessible peer is refused now (the caller falls back to
    // RCCL: dist.PeerComm agrees on the failure over all ranks) instead of faulting in the first launch.
    // Same device (ranks = processes on one GPU: the one-GPU tests) or attributes unavailable: nothing to check.
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) == hipSuccess) {
      if (at.device >= 0 && at.device != c->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, at.device) == hipSuccess && !can) {
          char b[200];
          snprintf(b, sizeof b, "p2p_connect: device %d has no peer access to device %d (rank %d)", c->device,
                   at.device, i);
          return fail(c, SPMF_E_UNSUPPORTED, b);
        }
      }
    } else {
      (void)hipGetLastError();
    }
  }
  p.connected = 1;
  return SPMF_OK;
}

int spmf_p2p_enable(spmf_ctx* c, int on) {
  if (!c) return SPMF_E_ARG;
  if (on && (!c->p2p.region || !c->p2p.peer[c->p2p.world > 0 ? c->p2p.world - 1 : 0]))
    return fail(c, SPMF_E_ARG, "p2p_enable: spmf_p2p_connect has not connected the peers");
  c->p2p.connected = on ? 1 : 0;
  return SPMF_OK;
}

int spmf_p2p_status(spmf_ctx* c, uint64_t out3[3]) {
  if (!c || !out3) return SPMF_E_ARG;
  if (!c->p2p.region) return fail(c, SPMF_E_ARG, "p2p_status: spmf_p2p_init was not called");
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(out3, c->p2p.region + c->p2p.off_seq, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SPMF_OK;
}

int spmf_p2p_disconnect(spmf_ctx* c) {
  if (!c) return SPMF_E_ARG;
  (void)hipDeviceSynchronize();
  p2p_unmap(c);
  return SPMF_OK;
}

int spmf_p2p_destroy(spmf_ctx* c) {
  if (!c) return SPMF_E_ARG;
  (void)hipDeviceSynchronize();
  p2p_release(c);
  return SPMF_OK;
}

// (2) RCCL, bound at run time
int spmf_comm_unique_id(void* out128) {
  if (!out128) return SPMF_E_ARG;
  Rccl* r = rccl();
  if (!r) return SPMF_E_UNSUPPORTED;
  RcclId id;
  const int rc = r->get_id(&id);
  if (rc != 0) return SPMF_E_HIP;
  memcpy(out128, &id, sizeof id);
  return SPMF_OK;
}

int spmf_comm_init(spmf_ctx* c, const void* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, SPMF_E_ARG, "comm_init: bad arguments");
  Rccl* r = rccl();
  if (!r) return fail(c, SPMF_E_UNSUPPORTED, "comm_init: librccl.so.1 could not be loaded");
  if (c->comm) {
    (void)r->destroy(c->comm);
    c->comm = nullptr;
  }
  HIPCHK(c, hipSetDevice(c->device));
  RcclId id;
  memcpy(&id, id128, sizeof id);
  void* comm = nullptr;
  const int rc = r->init_rank(&comm, world, id, rank);
  if (rc != 0) return fail(c, SPMF_E_HIP, std::string("ncclCommInitRank: ") + (r->errstr ? r->errstr(rc) : "?"));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return SPMF_OK;
}

int spmf_allreduce(spmf_ctx* c, float* buf, int64_t n, void* stream) {
  if (!c || !buf || n < 0) return fail(c, SPMF_E_ARG, "allreduce: bad arguments");
  if (c->p2p.connected) return p2p_allreduce(c, buf, n, (hipStream_t)stream);
  if (!c->comm) return fail(c, SPMF_E_ARG, "allreduce: neither spmf_comm_init nor spmf_p2p_connect was called");
  if (n == 0) return SPMF_OK;
  Rccl* r = rccl();
  const int rc = r->allreduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, c->comm, (hipStream_t)stream);
  if (rc != 0) return fail(c, SPMF_E_HIP, std::string("ncclAllReduce: ") + (r->errstr ? r->errstr(rc) : "?"));
  return SPMF_OK;
}

int spmf_comm_destroy(spmf_ctx* c) {
  if (!c) return SPMF_E_ARG;
  if (c->comm) {
    Rccl* r = rccl();
    if (r) (void)r->destroy(c->comm);
    c->comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
  }
  return SPMF_OK;
}

int spmf_nonfinite_argmin(spmf_ctx* c, int64_t n, const float* ll, double index_base, double* io, void* stream) {
  if (!c || !ll || !io || n < 0 || !(index_base >= 0.0)) return fail(c, SPMF_E_ARG, "nonfinite_argmin: bad arguments");
  if (n == 0) return SPMF_OK;
  launch_nonfinite_argmin(n, ll, index_base, io, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_nonfinite_lgamma(spmf_ctx* c, const spmf_counts* ct, const float* rate, double* out, void* stream) {
  if (!c || !rate || !out) return fail(c, SPMF_E_ARG, "nonfinite_lgamma: bad arguments");
  int rc = check_counts(c, ct);
  if (rc) return rc;
  if (ct->n_rows == 0 || ct->nnz == 0) return SPMF_OK;
  DenseLLArgs da{ct->n_rows, c->D, likelihood_code(c), nullptr, nullptr, nullptr, c->ctype, ct->row_ptr, ct->col_idx,
      ct->val, const_cast<float*>(rate), nullptr};
  launch_nonfinite_lgamma(da, out, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_nonfinite_patch(spmf_ctx* c, const spmf_counts* ct, int S, const float* const params[SPMF_NVARS],
    const float* eta, const double* io, const double* nlg, void* stream) {
  if (!c || !params || !eta || !io || !nlg || S < 1) return fail(c, SPMF_E_ARG, "nonfinite_patch: bad arguments");
  int rc = check_counts(c, ct);
  if (rc) return rc;
  if (!c->acc || c->ws_S < S || c->ws_rows != ct->n_rows) return fail(c, SPMF_E_ARG,
      "nonfinite_patch: no data pass over this batch and S precedes it");
  for (int i : {0, 1, 2, 7})
    if (!params[i]) return fail(c, SPMF_E_ARG, "nonfinite_patch: params v,w,u,s must be non-null");
  const int logt = likelihood_code(c);
  if (lik_exp(logt) && ct->nnz > 0 && !ct->gval) return fail(c, SPMF_E_ARG, "nonfinite_patch: log_transform needs counts.gval");
  NfPatchArgs a{c->D, c->K, logt, ct->row_ptr, ct->col_idx, ct->val,
      (c->flags & SPMF_FLAG_SCALE_ROWS) ? ct->row_scale : nullptr, params[2], params[0], params[1], params[7], eta,
      c->ctype, c->acc, (int64_t)acc_len(c->D, c->KP), c->Dh > 0 ? c->Dh : c->D, io, nlg, ct->n_rows, S};
  launch_nonfinite_patch(c->KP, a, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

// the library's one device allocation, at its first use (never inside a stream capture: a step is run eagerly
// before it is captured); zero-filled: its last word is the arrival ticket of sample_fwd_kernel.  Without it the
// surrogate kernels fall back (atomics / two launches).
static void ensure_scratch(spmf_ctx* c, hipStream_t st) {
  if (c->scratch) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  if (cap != hipStreamCaptureStatusNone) return;
  if (hipMalloc((void**)&c->scratch, spmf_ctx::kScratchDoubles * sizeof(double)) != hipSuccess ||
      hipMemset(c->scratch, 0, spmf_ctx::kScratchDoubles * sizeof(double)) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr;
    (void)hipGetLastError();
  }
}

int spmf_sample_transform(spmf_ctx* c, const spmf_sur_var* vars, int nvars, int S, uint64_t seed, uint64_t counter,
    const double* state, double* logq, void* stream) {
  if (!c || !vars || nvars < 1 || nvars > 12 || S < 1 || S > 65535 || !logq) return fail(c, SPMF_E_ARG,
      "sample_transform: bad arguments");
  SurTable T;
  int max_n = 0;
  for (int i = 0; i < nvars; ++i) {
    const spmf_sur_var& v = vars[i];
    if (v.n == 0) {
      T.v[i] = SurVar{};
      continue;
    }
    if (!v.t0 || !v.t1 || !v.noise || !v.theta || v.n < 1 || v.kind < 0 || v.kind > 2 || (v.kind == 2 && !v.dgda))
      return fail(c, SPMF_E_ARG, "sample_transform: bad variable (t0 / t1 / noise / dgda / theta buffers)");
    if (v.noise_ld != 0 && v.noise_ld < v.n) return fail(c, SPMF_E_ARG, "surrogate: noise_ld < n");
    T.v[i] = SurVar{v.t0, v.t1, v.noise, v.dgda, v.theta, v.gtheta, v.g0, v.g1, v.n, v.kind, v.ident,
        v.noise_ld ? v.noise_ld : (int64_t)v.n};
    if (v.n > max_n) max_n = v.n;
  }
  if (max_n < 1) return fail(c, SPMF_E_ARG, "sample_transform: every variable is skipped (n = 0)");
  hipStream_t st = (hipStream_t)stream;
  ensure_scratch(c, st);
  if (!launch_sample_fwd(T, nvars, max_n, S, seed, counter, state, logq, c->scratch, spmf_ctx::kScratchDoubles, st)) {
    // no scratch for the per-block sums: the two separate launches (same numbers)
    launch_sample_noise(T, nvars, max_n, S, seed, counter, state, st);
    launch_surrogate_fwd(T, nvars, max_n, S, logq, c->scratch, spmf_ctx::kScratchDoubles, st);
  }
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_surrogate_fwd(spmf_ctx* c, const spmf_sur_var* vars, int nvars, int S, double* logq, void* stream) {
  if (!c || !vars || nvars < 1 || nvars > 12 || S < 1 || !logq) return fail(c, SPMF_E_ARG,
      "surrogate_fwd: bad arguments");
  SurTable T;
  int max_n = 0;
  for (int i = 0; i < nvars; ++i) {
    const spmf_sur_var& v = vars[i];
    if (v.n == 0) {                 // skipped variable (its slot keeps its index: the RNG counter, the logq slots)
      T.v[i] = SurVar{};
      continue;
    }
    if (!v.t0 || !v.t1 || !v.noise || !v.theta || v.n < 1 || v.kind < 0 || v.kind > 2) return fail(c, SPMF_E_ARG,
        "surrogate_fwd: bad variable");
    if (v.noise_ld != 0 && v.noise_ld < v.n) return fail(c, SPMF_E_ARG, "surrogate: noise_ld < n");
    T.v[i] = SurVar{v.t0, v.t1, v.noise, v.dgda, v.theta, v.gtheta, v.g0, v.g1, v.n, v.kind, v.ident,
        v.noise_ld ? v.noise_ld : (int64_t)v.n};
    if (v.n > max_n) max_n = v.n;
  }
  if (max_n < 1) return fail(c, SPMF_E_ARG, "surrogate_fwd: every variable is skipped (n = 0)");
  hipStream_t st = (hipStream_t)stream;
  ensure_scratch(c, st);
  launch_surrogate_fwd(T, nvars, max_n, S, logq, c->scratch, spmf_ctx::kScratchDoubles, st);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_sample_noise(spmf_ctx* c, const spmf_sur_var* vars, int nvars, int S, uint64_t seed, uint64_t counter,
    const double* state, void* stream) {
  if (!c || !vars || nvars < 1 || nvars > 12 || S < 1 || S > 65535) return fail(c, SPMF_E_ARG,
      "sample_noise: bad arguments");
  SurTable T;
  int max_n = 0;
  for (int i = 0; i < nvars; ++i) {
    const spmf_sur_var& v = vars[i];
    if (v.n == 0) {                 // skipped variable: nothing is drawn for it, the others keep their indices
      T.v[i] = SurVar{};
      continue;
    }
    if (!v.t0 || !v.noise || v.n < 1 || v.kind < 0 || v.kind > 2 || (v.kind == 2 && !v.dgda)) return fail(c,
        SPMF_E_ARG, "sample_noise: bad variable (t0 / noise / dgda buffers)");
    if (v.noise_ld != 0 && v.noise_ld < v.n) return fail(c, SPMF_E_ARG, "surrogate: noise_ld < n");
    T.v[i] = SurVar{v.t0, v.t1, v.noise, v.dgda, v.theta, v.gtheta, v.g0, v.g1, v.n, v.kind, v.ident,
        v.noise_ld ? v.noise_ld : (int64_t)v.n};
    if (v.n > max_n) max_n = v.n;
  }
  if (max_n < 1) return fail(c, SPMF_E_ARG, "sample_noise: every variable is skipped (n = 0)");
  launch_sample_noise(T, nvars, max_n, S, seed, counter, state, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_surrogate_bwd(spmf_ctx* c, const spmf_sur_var* vars, int nvars, int S, double inv_sb, double cw,
    void* stream) {
  if (!c || !vars || nvars < 1 || nvars > 12 || S < 1) return fail(c, SPMF_E_ARG, "surrogate_bwd: bad arguments");
  SurTable T;
  int max_n = 0;
  for (int i = 0; i < nvars; ++i) {
    const spmf_sur_var& v = vars[i];
    if (!v.t0 || !v.t1 || !v.noise || !v.gtheta || !v.g0 || !v.g1 || v.n < 1 || v.kind < 0 || v.kind > 2 || (v.kind == 2 && !v.dgda)) return fail(c, SPMF_E_ARG, "surrogate_bwd: bad variable");
    if (v.noise_ld != 0 && v.noise_ld < v.n) return fail(c, SPMF_E_ARG, "surrogate: noise_ld < n");
    T.v[i] = SurVar{v.t0, v.t1, v.noise, v.dgda, v.theta, v.gtheta, v.g0, v.g1, v.n, v.kind, v.ident,
        v.noise_ld ? v.noise_ld : (int64_t)v.n};
    if (v.n > max_n) max_n = v.n;
  }
  launch_surrogate_bwd(T, nvars, max_n, S, (float)inv_sb, (float)cw, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_adam_step(spmf_ctx* c, const spmf_adam_var* tensors, int ntensors, double lr, double beta1, double beta2,
    double eps, int step, double clip, void* stream) {
  if (!c || !tensors || ntensors < 1 || ntensors > 24 || step < 1) return fail(c, SPMF_E_ARG,
      "adam_step: bad arguments");
  AdamTable T;
  int max_n = 0;
  for (int i = 0; i < ntensors; ++i) {
    const spmf_adam_var& a = tensors[i];
    if (!a.p || !a.m || !a.v || !a.g || a.n < 1) return fail(c, SPMF_E_ARG, "adam_step: bad tensor");
    T.v[i] = AdamVar{a.p, a.m, a.v, a.g, a.n};
    if (a.n > max_n) max_n = a.n;
  }
  const double c1 = 1.0 - pow(beta1, step), c2 = 1.0 - pow(beta2, step);
  launch_adam(T, ntensors, max_n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)c1, (float)c2,
      (float)clip, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_vi_gate(spmf_ctx* c, const double* parts, const double* logq, const double* n_nonfinite, int S, double cw,
    double rows, double* state, void* stream) {
  if (!c || !parts || !logq || !state || S < 1 || !(rows > 0.0)) return fail(c, SPMF_E_ARG, "vi_gate: bad arguments");
  launch_vi_gate(parts, logq, n_nonfinite, S, cw, rows, state, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_adam_step_dev(spmf_ctx* c, const spmf_adam_var* tensors, int ntensors, const double* state, void* stream) {
  if (!c || !tensors || ntensors < 1 || ntensors > 24 || !state) return fail(c, SPMF_E_ARG,
      "adam_step_dev: bad arguments");
  AdamTable T;
  int max_n = 0;
  for (int i = 0; i < ntensors; ++i) {
    const spmf_adam_var& a = tensors[i];
    if (!a.p || !a.m || !a.v || !a.g || a.n < 1) return fail(c, SPMF_E_ARG, "adam_step_dev: bad tensor");
    T.v[i] = AdamVar{a.p, a.m, a.v, a.g, a.n};
    if (a.n > max_n) max_n = a.n;
  }
  launch_adam_dev(T, ntensors, max_n, state, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

int spmf_surrogate_bwd_adam_dev(spmf_ctx* c, const spmf_sur_var* vars, int nvars, int S, double inv_sb, double cw,
    const spmf_adam_var* tensors, const double* state, void* stream) {
  if (!c || !vars || !tensors || !state || nvars < 1 || nvars > 12 || S < 1) return fail(c, SPMF_E_ARG,
      "surrogate_bwd_adam_dev: bad arguments");
  SurTable T;
  AdamTable A;
  int max_n = 0;
  for (int i = 0; i < nvars; ++i) {
    const spmf_sur_var& v = vars[i];
    if (!v.t0 || !v.t1 || !v.noise || !v.gtheta || v.n < 1 || v.kind < 0 || v.kind > 2 || (v.kind == 2 && !v.dgda))
      return fail(c, SPMF_E_ARG, "surrogate_bwd_adam_dev: bad variable");
    if (v.noise_ld != 0 && v.noise_ld < v.n) return fail(c, SPMF_E_ARG, "surrogate: noise_ld < n");
    T.v[i] = SurVar{v.t0, v.t1, v.noise, v.dgda, v.theta, v.gtheta, nullptr, nullptr, v.n, v.kind, v.ident,
        v.noise_ld ? v.noise_ld : (int64_t)v.n};
    if (v.n > max_n) max_n = v.n;
    // tensors[2i], tensors[2i+1]: the Adam records of this variable's t0 / t1
    for (int j = 0; j < 2; ++j) {
      const spmf_adam_var& a = tensors[2 * i + j];
      if (!a.p || !a.m || !a.v || a.n != v.n || a.p != (j ? v.t1 : v.t0)) return fail(c, SPMF_E_ARG,
          "surrogate_bwd_adam_dev: tensors[2i+j] must be the Adam record (p, m, v, n) of vars[i].t{j}");
      A.v[2 * i + j] = AdamVar{a.p, a.m, a.v, nullptr, a.n};
    }
  }
  launch_surrogate_bwd_adam(T, A, nvars, max_n, S, (float)inv_sb, (float)cw, state, (hipStream_t)stream);
  HIPCHK(c, hipGetLastError());
  return SPMF_OK;
}

}  // extern "C"
