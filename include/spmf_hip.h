/* spmf_hip.h -- C-ABI of libspmf_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of mederrata/spmf: the per-batch energy
 * (ELBO integrand) of PoissonFactorization and its gradient through the
 * factor matrices.  The reference has no FFI; the seam this library sits
 * behind is the Python method
 *     PoissonFactorization.unormalized_log_prob_parts(data, **params)
 *         mederrata_spmf/poisson.py:582-621
 * (called once per optimiser step by bayesianquilts' fit/calibrate_advi,
 * tests/spmf_test.py:35-43, bin/factorize_csv.py:121-124) and the helpers it
 * calls: encode :623-650, encoding_matrix :652-666, intercept_matrix
 * :680-701, log_likelihood_components :156-184, compute_scales :113-154 and
 * the prior of create_distributions :212-401.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensors on
 *    the Python side) unless marked "host"; the caller supplies the workspace
 *    (spmf_workspace_bytes); the library's only own device allocation is an
 *    8 MiB scratch for the fixed-order reductions of the surrogate kernels.
 *  - all calls are asynchronous on the hipStream_t passed as `stream`
 *    (void* so the header needs no HIP include); one ctx per (process,
 *    device); calls on one ctx are stream-ordered and not re-entrant.
 *  - return value: 0 = ok, <0 = error (SPMF_E_*); spmf_last_error() gives
 *    the message.  Nothing throws across this boundary.
 *  - arithmetic: fp32 storage and fp32 FMA, fp64 accumulation of every
 *    scalar reduction ("dtype": "f32" in bench.py).
 *  - S = number of Monte-Carlo draws = leading axis of every parameter.
 *
 * Variable order everywhere (= the reference's surrogate var_list,
 * poisson.py:403-539,572):
 *   0 v[K,D] 1 w[1,D] 2 u[D,K] 3 u_eta[D,K] 4 u_tau[1,K] 5 s_eta[2,D]
 *   6 s_tau[1,D] 7 s[2,D] 8 u_eta_a[D,K] 9 u_tau_a[1,K] 10 s_eta_a[2,D]
 *   11 s_tau_a[1,D]
 * Energy parts order: the 12 prior parts in that order, then 12 = 'z',
 * 13 = 'x' (poisson.py:604,619).
 */
#ifndef SPMF_HIP_H
#define SPMF_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every function below is exported; nothing else is (the library is built with
 * -fvisibility=hidden). */
#if defined(SPMF_BUILD)
#pragma GCC visibility push(default)
#endif

#define SPMF_NVARS 12
#define SPMF_NPARTS 14
#define SPMF_PART_Z 12
#define SPMF_PART_X 13

#define SPMF_OK 0
#define SPMF_E_ARG (-1)       /* bad argument (null pointer, bad size, K unsupported) */
#define SPMF_E_HIP (-2)       /* a HIP runtime call failed */
#define SPMF_E_WORKSPACE (-3) /* caller's workspace too small */
#define SPMF_E_UNSUPPORTED (-4)

/* flags for spmf_ctx_create */
#define SPMF_FLAG_SCALE_ROWS 1u     /* poisson.py:61,644-649 */
#define SPMF_FLAG_BERNOULLI 4u      /* BernoulliFactorization (mederrata_spmf/bernoulli.py): Bernoulli(logits=rate)
                                     * likelihood :147-155, Normal priors on v,w :187-216; with SPMF_FLAG_LOG_TRANSFORM
                                     * the logit is exp(<z, eta v>) - 1 + phi (:60-61) */
#define SPMF_FLAG_MIXED 8u          /* build-defined (mederrata_spmf/mixed.py is an empty file): per-column likelihood,
                                     * Bernoulli(logits) on the columns flagged by spmf_ctx_set_column_types, Poisson
                                     * on the others; linear decoder only */
#define SPMF_FLAG_LOG_TRANSFORM 2u  /* poisson.py:41-42,52-53: sparse stored-cell terms + dense f32-MFMA exp sums */
#define SPMF_FLAG_ABS_HORSESHOE 16u /* horshoe_plus=False (poisson.py:378-398): AbsHorseshoe priors on u (scale
                                     * u_tau_scale*decay^k) and s (scale s_tau_scale); only the variables v, w, u, s
                                     * (indices 0, 1, 2, 7) exist: the other params / grads pointers are ignored
                                     * (may be NULL) and their energy parts are 0 */

typedef struct spmf_ctx spmf_ctx;

/* One batch (or one row shard) of the count matrix, device resident.
 * Row-major CSR for the row pass and a row-panel CSC ("panel-CSC": for each
 * panel of `panel_rows` consecutive rows, a CSC of that panel) for the
 * column pass; the column pass walks the panels of residue class blockIdx % 8
 * on one XCD, so a panel's z and xi*gz rows (panel_rows * 2 * KP floats) should
 * fit that XCD's L2 and n_panels should be a multiple of 8 for big batches:
 * spmf_amd/sparse.py balanced_panel_rows).  Replaces the dense [B,D] tensor data[count_key] the
 * reference feeds to encode/log_likelihood_components (poisson.py:170,182).
 */
typedef struct spmf_counts {
  int64_t n_rows;           /* B: rows of this batch */
  int64_t nnz;              /* stored entries of this batch */
  int32_t n_cols;           /* D */
  int32_t n_panels;         /* row panels covering the batch */
  int32_t panel_rows;       /* rows per panel (last one may be short) */
  int32_t row_base;         /* row id stored in pc_row for batch row 0 */
  const int32_t* row_ptr;   /* [B+1] absolute offsets into col_idx/val */
  const int32_t* col_idx;   /* base pointer (not offset) */
  const float* val;         /* base pointer: raw counts x_bd */
  const float* row_scale;   /* [B] xi_b = rowsum_b / xi_u_global, or NULL => 1 */
  const int32_t* pc_ptr;    /* [n_panels*(D+1)] absolute offsets into pc_row/pc_val */
  const int32_t* pc_row;    /* base pointer: row id (row_base + batch row) */
  const float* pc_val;      /* base pointer */
  double lgamma_sum;        /* sum over the batch of lgamma(x+1) (parameter free) */
  /* log_transform only (poisson.py:41-42): g(x) = log(x/eta + 1) per stored
   * entry, in CSR and in panel-CSC order (base pointers); NULL otherwise. */
  const float* gval;
  const float* pc_gval;
  /* Column-pass work items: every non-empty (panel, column) list cut into
   * segments of bounded length, sorted by length inside a panel.
   *   items[i] = {start, len, column, 0} (int32 x 4, start absolute into
   *   pc_row/pc_val); item_ptr[p], item_ptr[p+1] bound panel p's items
   *   (absolute item indices; pointer already offset to the batch's first
   *   panel); max_items_per_panel sizes the launch. */
  const int32_t* item_ptr;
  const int32_t* items;
  int32_t max_items_per_panel;
  /* Readable int32/float entries behind the END of the last list in pc_row, pc_val, pc_ent
   * AND pc_gval (every list array the context's decoder reads must carry the same padding:
   * with log_transform that includes pc_gval).  With pc_pad >= 4*ceil(K/4) - 1 (K padded to 4,
   * 8, 16, 32, 64) the column pass fetches list entries four at a time (16-B loads that may run
   * past a list's end); 0 selects the entry-at-a-time fetch, which never reads behind a list.
   * Negative values are rejected.  (Until version 2 this slot was a reserved field: a caller
   * that left it uninitialised is caught by struct_size below.) */
  int32_t pc_pad;
  /* Optional column split of the work items (multi-GPU overlap): inside a panel
   * the items are sorted by column half first (columns < col_split, then the
   * rest), then by length; item_mid[p] is the first item of panel p's upper
   * half (absolute item index, pointer offset like item_ptr) and
   * max_items_half[h] sizes a launch over one half.  NULL / 0 when unused. */
  const int32_t* item_mid;
  int32_t col_split;
  int32_t max_items_half[2];
  /* ABI guard (spmf_version() >= 3): sizeof(spmf_counts) as the CALLER was compiled.  Every entry
   * point that takes a spmf_counts rejects a struct whose struct_size differs from the library's
   * own sizeof (SPMF_E_ARG, "built against another spmf_hip.h"): a caller compiled against an older,
   * shorter layout can no longer have the fields appended since (ent, pc_ent) read from whatever
   * follows its struct.  Zero-initialise the struct, then set this first. */
  int32_t struct_size;
  /* Optional packed copy of the CSR entries for the row pass: ent[i] = col_idx[i] << 16 | count,
   * valid when D <= 65536 and every stored value is an integer in [0, 65535] (counts are:
   * tests/spmf_test.py:19); NULL otherwise.  Halves the entry stream of the sweeps that read the
   * raw counts (4 instead of 8 bytes per stored entry); col_idx / val stay the canonical arrays. */
  const uint32_t* ent;
  /* The same for the column pass's lists: pc_ent[i] = (row - first row of its panel) << 16 | count,
   * in the order (and with the pc_pad) of pc_row / pc_val; needs panel_rows <= 65536 as well
   * (rejected otherwise).  Used by the four-per-lane fetch only; NULL otherwise. */
  const uint32_t* pc_ent;
  /* Optional, for the deterministic mode (spmf_ctx_set_deterministic): the work items in their
   * GENERATION order -- (panel, column, segment) -- so that the per-item partial sums of the column
   * pass can be added up column by column in a fixed order:
   *   list_first[p * D + d] .. list_first[p * D + d + 1]   raw indices of the items cut from list
   *                       (p, d) (pointer already offset to the batch's first panel, like pc_ptr with
   *                       stride D; n_panels * D + 1 entries are read)
   *   item_pos[r]         position of raw item r in `items` (absolute item index, base pointer)
   *   n_items             items of THIS batch: item_ptr[n_panels] - item_ptr[0]
   * spmf_layout_build fills them; NULL / 0 when unused. */
  const int32_t* list_first;
  const int32_t* item_pos;
  int64_t n_items;
} spmf_counts;

/* ABI version of this header: 6.  (2 -> 3: spmf_counts.struct_size replaces a reserved slot
 * and is verified; pc_pad / ent / pc_ent are validated.  3 -> 4: the device layout builder
 * spmf_layout_* added.  4 -> 5: spmf_counts grows by list_first / item_pos / n_items for the
 * deterministic mode -- a caller built against 4 is refused by the struct_size check, not
 * misread.  5 -> 6: spmf_step_begin / spmf_step_end added -- the step with its outputs handed over
 * up front -- and spmf_p2p_*, the step's collective as a hand-written kernel over peer pointers; no
 * struct changed, every version-5 entry point keeps its meaning.)  A binding checks it
 * at load time. */
#define SPMF_ABI_VERSION 6
int spmf_version(void);

/* sizeof(spmf_counts) / sizeof(spmf_sur_var) / sizeof(spmf_adam_var) as this
 * library was built: a binding (ctypes, cgo, JNI ...) asserts its own mirror of
 * the struct against it before the first call. */
size_t spmf_sizeof_counts(void);
size_t spmf_sizeof_sur_var(void);
size_t spmf_sizeof_adam_var(void);

/* K = latent_dim, D = feature_dim (poisson.py:58,102-104).  1 <= K <= 256 for the Poisson likelihood with
 * the linear decoder (K <= 64: the lane-group sparse passes; above: one wave per factor row,
 * csrc/widek.hip); 1 <= K <= 64 with SPMF_FLAG_LOG_TRANSFORM / BERNOULLI / MIXED (SPMF_E_UNSUPPORTED above)
 * and for the deterministic mode. */
int spmf_ctx_create(int device, int K, int D, unsigned flags, spmf_ctx** out);
void spmf_ctx_destroy(spmf_ctx* ctx);
const char* spmf_last_error(const spmf_ctx* ctx); /* host string */

/* Hyper-parameters of the prior (poisson.py:59,106-107,225-226). */
int spmf_ctx_set_prior(spmf_ctx* ctx, double u_tau_scale, double s_tau_scale,
                       double symmetry_breaking_decay);

/* SPMF_FLAG_MIXED only: column_is_bernoulli[D] (uint8, device, caller-owned,
 * must outlive the ctx calls). */
int spmf_ctx_set_column_types(spmf_ctx* ctx, const uint8_t* column_is_bernoulli);
/* Optional with SPMF_FLAG_MIXED: the ascending indices of the Bernoulli columns (int32,
 * device, caller-owned, must outlive the ctx calls).  With it the dense softplus /
 * sigmoid sums run over those n columns only (compacted rows of V'); without it over
 * all D columns with the Poisson ones masked by a -1e30 logit bias. */
int spmf_ctx_set_bernoulli_columns(spmf_ctx* ctx, const int32_t* cols, int n);

/* Upper bound in bytes of the buffer that keeps E between the two dense contractions
 * (log_transform / Bernoulli / mixed contexts; default 8 GiB).  The rows of a batch are
 * processed in equal chunks that fit it; fewer, larger chunks fill the chip better (at
 * C4, 500k x 30k: 8 chunks at 8 GiB, one at 64 GiB).  Call before
 * spmf_workspace_bytes / spmf_ctx_set_workspace: it changes the workspace size. */
int spmf_ctx_set_e_cap(spmf_ctx* ctx, size_t bytes);

/* Deterministic mode (Poisson likelihood, linear decoder, no column split): the step's float and
 * fp64 atomics are replaced by single-writer partial sums added up in a fixed order -- per-item
 * partials of the column pass summed column by column in (panel, segment) order, per-workgroup
 * scalar sums of the row pass summed in workgroup order -- so the 14 parts and all 12 gradients of
 * a step are bit-identical from run to run (and replicas of a row-sharded job cannot drift apart
 * through rounding order).  Costs the column pass its atomics' bandwidth twice (write + read of
 * n_items * (2*KP + 4) floats): C3 2.81 -> 2.87 ms per step.  `scratch`: caller-owned device buffer of
 * spmf_det_scratch_bytes(ctx, n_items, S) bytes for the largest batch (n_items = spmf_counts.n_items),
 * 256-byte aligned; NULL switches the mode off.  The counts must carry list_first / item_pos.
 * (Covers the step as spmf_data_pass + spmf_finish run it; the replacement rule for non-finite cells,
 * spmf_nonfinite_patch, adds its correction with atomics and is outside the guarantee.) */
size_t spmf_det_scratch_bytes(const spmf_ctx* ctx, int64_t n_items, int S);
int spmf_ctx_set_deterministic(spmf_ctx* ctx, void* scratch, size_t bytes);

/* Bytes of caller-owned device workspace needed for batches of up to
 * max_rows rows and S draws.  With SPMF_FLAG_LOG_TRANSFORM / BERNOULLI / MIXED this
 * includes the buffer that keeps E = exp(<z_b, eta_d v_d>) (or the sigmoid of the
 * Bernoulli logits) between the two dense contractions
 * (min(max_rows, chunk) * D floats, chunk chosen so that it stays within
 * spmf_ctx_set_e_cap; the
 * environment variable SPMF_DENSE_E_ONCE=0, read at spmf_ctx_create, selects the
 * form that recomputes E instead and needs no such buffer). */
size_t spmf_workspace_bytes(const spmf_ctx* ctx, int64_t max_rows, int S);
int spmf_ctx_set_workspace(spmf_ctx* ctx, void* workspace, size_t bytes);

/* ---- dataset pre-pass ------------------------------------------------- */
/* compute_scales (poisson.py:113-154): column sums and per-column counts of
 * x>0 of a CSR block, ACCUMULATED into colsum[D] (fp64) / colnnz[D] (fp64),
 * plus per-row sums row_sum[B] (fp32) and per-row sums of lgamma(x+1)
 * row_lgamma[B] (fp64).  Any output pointer may be NULL. */
int spmf_counts_stats(spmf_ctx* ctx, int64_t n_rows, const int32_t* row_ptr,
                      const int32_t* col_idx, const float* val, double* colsum,
                      double* colnnz, float* row_sum, double* row_lgamma,
                      void* stream);

/* The column half of the same statistics from a built layout: colsum[D] / colnnz[D] (fp64,
 * ACCUMULATED, either may be NULL) from the panel-CSC lists of `counts` -- one atomic per
 * (panel, column) list instead of one per stored entry (C3: 6.7 -> 0.3 ms).  Sums of integer
 * counts are exact in fp64, so both forms give the same numbers. */
int spmf_counts_colstats(spmf_ctx* ctx, const spmf_counts* counts, double* colsum,
                         double* colnnz, void* stream);

/* log_transform contexts: the encoder side g(x) = log(x / eta_d + 1) (encoder_function,
 * poisson.py:41-42) of every stored entry of `counts`, in CSR order into gval (base pointer,
 * indexed like col_idx / val) and in list order into pc_gval (base pointer, indexed like pc_val);
 * either may be NULL.  eta[D] fp32.  The pc_pad entries behind the last list of the shard are the
 * caller's (zero-filled): only list entries are written.  Depends on the counts and the column
 * scales only: once per (shard, eta), then set spmf_counts.gval / pc_gval. */
int spmf_counts_gvals(spmf_ctx* ctx, const spmf_counts* counts, const float* eta, float* gval,
                      float* pc_gval, void* stream);

/* ---- device layout builder ------------------------------------------- */
/* Builds everything of a spmf_counts that is derived from the CSR arrays of one row shard
 * (the reference hands its model a dense [B,D] batch, poisson.py:170,182; a caller of this
 * library hands it CSR and gets the row-panel CSC, the column-pass work items and the packed
 * entry streams back): on the device, stream-ordered, deterministic (the same input gives the
 * same bytes), into ONE caller-owned buffer.
 *
 *   spmf_layout_sizes   bytes of the layout buffer (kept as long as the spmf_counts is used)
 *                       and of the scratch buffer (free after the call returns)
 *   spmf_layout_build   fills `layout`, then *out (zero-initialised by the callee; row_ptr /
 *                       col_idx / val are the caller's arrays, lgamma_sum / row_scale / gval /
 *                       pc_gval stay 0 / NULL: spmf_counts_stats and the model supply them) and
 *                       *info.  Synchronises `stream` once, at the end, to read back the item
 *                       count and the input checks.
 *
 * Input: canonical CSR of the shard -- row_ptr[0] == 0, row_ptr[n_rows] == nnz, non-decreasing,
 * 0 <= col_idx < n_cols, no (row, column) pair stored twice (columns need not be sorted inside
 * a row).  The first four are verified on the device (SPMF_E_ARG, nothing usable in *out).
 * panel_rows >= 1 (spmf_amd/sparse.py balanced_panel_rows chooses it from K); col_split = 0 or
 * the column split of spmf_ctx_set_column_split.  nnz < 2^31, n_rows < 2^31 and n_panels * n_cols < 2^32
 * (SPMF_E_UNSUPPORTED otherwise: choose larger panels).
 *
 * Layout produced (what spmf_amd/sparse.py built with torch sorts until version 3 of
 * this header; the two are compared array by array in tests/test_gpu_layout.py):
 *   lists     entries ordered by (panel, column, row): pc_ptr, pc_row, pc_val with pc_pad = 64
 *   items     every non-empty list cut into segments of <= info->segment entries (16 ... 256,
 *             so that a panel offers a few thousand items), inside a panel ordered by column
 *             half (col_split), then by descending length, ties in (column, segment) order
 *   ent / pc_ent  when every stored value is an integer count in [0, 65535] and n_cols
 *             (resp. panel_rows) <= 65536; NULL in *out otherwise
 */
typedef struct spmf_layout_info {
  int32_t struct_size;            /* in: sizeof(spmf_layout_info) of the caller */
  int32_t n_panels;
  int32_t panel_rows;
  int32_t segment;                /* longest work item */
  int64_t n_items;
  int32_t packed_ent;             /* 1: out->ent is set */
  int32_t packed_pc_ent;          /* 1: out->pc_ent is set */
  const int32_t* items_per_panel; /* [n_panels] device, inside `layout`: for max_items_per_panel of a panel range */
  const int32_t* items_lower;     /* [n_panels] items of the lower column half (all of them without a split) */
} spmf_layout_info;
size_t spmf_sizeof_layout_info(void);
int spmf_layout_sizes(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, int32_t panel_rows,
                      size_t* layout_bytes, size_t* scratch_bytes);
int spmf_layout_build(int device, int64_t n_rows, int64_t nnz, int32_t n_cols,
                      const int32_t* row_ptr, const int32_t* col_idx, const float* val,
                      int32_t panel_rows, int32_t col_split, void* layout, size_t layout_bytes,
                      void* scratch, size_t scratch_bytes, spmf_counts* out,
                      spmf_layout_info* info, void* stream);
/* The same with the model's latent dimension as a hint (ABI 6; 0 = unknown = the two calls above): a work item
 * is one lane group of the column pass, K padded / 4 lanes, so at K <= 8 a wave carries 32 or 64 items and the
 * lists are cut into proportionally more, shorter items (16 384 / 32 768 per panel instead of 4096) -- the
 * reference CLI's default K = 2 ran its column pass on a few dozen waves otherwise.  Sizes and build must get
 * the same hint. */
int spmf_layout_sizes_k(int device, int64_t n_rows, int64_t nnz, int32_t n_cols, int32_t panel_rows,
                        int32_t latent_dim, size_t* layout_bytes, size_t* scratch_bytes);
int spmf_layout_build_k(int device, int64_t n_rows, int64_t nnz, int32_t n_cols,
                        const int32_t* row_ptr, const int32_t* col_idx, const float* val,
                        int32_t panel_rows, int32_t col_split, int32_t latent_dim, void* layout,
                        size_t layout_bytes, void* scratch, size_t scratch_bytes, spmf_counts* out,
                        spmf_layout_info* info, void* stream);
/* Message of the last failed spmf_layout_* call of the calling thread (host string). */
const char* spmf_layout_last_error(void);

/* Dense batches -- the reference's own input, data[count_key] as a [B,D] array
 * (poisson.py:170,182; tests/spmf_test.py:17-22) -- to the CSR arrays above, on the device:
 *   spmf_dense_row_ptr   row_ptr[n_rows+1] = offsets of the rows' stored cells (x != 0, so NaN cells
 *                        are stored); scratch: spmf_dense_scratch_bytes(n_rows), 8-byte aligned;
 *                        the total is row_ptr[n_rows] (saturating at 2^31-1: a shard must stay below)
 *   spmf_dense_fill_csr  col_idx / val (sized by that total) in row-major, ascending-column order
 * dense: fp32, row-major with leading dimension ld >= n_cols.  Both are stream-ordered and do not
 * synchronise; the caller reads row_ptr[n_rows] between the two to size col_idx / val. */
size_t spmf_dense_scratch_bytes(int64_t n_rows);
int spmf_dense_row_ptr(int device, int64_t n_rows, int32_t n_cols, const float* dense, int64_t ld,
                       int32_t* row_ptr, void* scratch, size_t scratch_bytes, void* stream);
int spmf_dense_fill_csr(int device, int64_t n_rows, int32_t n_cols, const float* dense, int64_t ld,
                        const int32_t* row_ptr, int32_t* col_idx, float* val, void* stream);

/* ---- the hot path ------------------------------------------------------ */
/* Phase 1: sparse data term for S draws.  Reads u,v,w,s (params[2,0,1,7])
 * and eta[D] (eta_i, poisson.py:88-91,142-149; ones when unscaled).  Leaves
 * per-draw accumulators in the workspace:
 *   acc[S][acc_len] fp32 = [ gA'(D*KP) | gV'(D*KP) | gphi(D) | tail ]
 * where the tail carries the fp64 scalars (sum x log r, sum z^2, non-finite
 * count, dense sum, saturated count, one spare, sum_b z_b[KP]) as (hi,lo)
 * float pairs so ONE fp32 sum-all-reduce of
 * acc over row shards finishes the reduction. */
int spmf_data_pass(spmf_ctx* ctx, const spmf_counts* counts, int S,
                   const float* const params[SPMF_NVARS], const float* eta,
                   void* stream);
float* spmf_acc_ptr(const spmf_ctx* ctx);      /* device pointer into the workspace */
int64_t spmf_acc_len(const spmf_ctx* ctx, int S); /* floats, all S draws */

/* Optional overlap for the multi-GPU step: launch the prior half of
 * spmf_finish (all twelve prior log-densities and prior_weight * d prior /
 * d theta; it reads no accumulator) on the context's side stream, forked from
 * `stream`.  Meant to be called between spmf_data_pass and the all-reduce, so
 * it runs while the collective leaves the GPU mostly idle (beside the sparse
 * passes it costs them more than it hides: 3.60 -> 3.80 ms on C3).
 * spmf_finish with the same S, parts and grads then joins the side stream and
 * adds the data half only; without this call it does both halves itself.
 * Order: it writes per-workgroup partial sums into the workspace a data pass has
 * bound, so it must FOLLOW a spmf_data_pass on the current workspace;
 * spmf_ctx_set_workspace / spmf_ctx_set_e_cap forget that binding, and until the next
 * data pass spmf_prior_async, spmf_finish and spmf_nonfinite_patch return
 * SPMF_E_WORKSPACE / SPMF_E_ARG (spmf_acc_ptr: NULL) instead of touching the old buffer. */
int spmf_prior_async(spmf_ctx* ctx, int S, double prior_weight,
                     const float* const params[SPMF_NVARS], const float* eta, double* parts,
                     float* const grads[SPMF_NVARS], void* stream);

/* Optional marker inside the data pass (ABI 6): `event` (a hipEvent_t of the caller, NULL = none) is recorded
 * on the pass's stream right behind its row stage -- after the row pass of the last draw has been issued,
 * before the column pass.  Work of the caller that does not feed the data pass (the VI step's draws and
 * transform of the scale hierarchy, the prior half of the finish: spmf_amd/vi.py) waits for it on another
 * stream and so runs beside the column pass -- many short workgroups that share the chip gracefully -- and
 * not beside the row pass, whose launch is exactly the resident set and is delayed as a whole by any
 * workgroup that holds a slot when it starts. */
int spmf_ctx_set_rows_event(spmf_ctx* ctx, void* event);

/* Column split of the packed accumulators (multi-GPU: all-reduce one half while
 * the column pass still produces the other).  With Dh set (a multiple of 32 in
 * (0, D); 0 or D = none; linear Poisson decoder only) the accumulators of a
 * draw are laid out [cols < Dh: gA'|gV'|gphi][cols >= Dh: gA'|gV'|gphi][tail];
 * spmf_acc_split returns the two contiguous element ranges (the second one
 * includes the fp64 tail) relative to spmf_acc_ptr.  spmf_data_pass_split runs
 * part 0 (everything up to and including the lower half's column pass) or
 * part 1 (upper half + pack) of one draw's data pass; the counts must carry
 * item_mid for the same split.  Order per step:
 *   data_pass_split(0) -> all-reduce range 0 (async) -> data_pass_split(1)
 *   -> all-reduce range 1 -> spmf_finish. */
int spmf_ctx_set_column_split(spmf_ctx* ctx, int Dh);
int spmf_acc_split(const spmf_ctx* ctx, int64_t off[2], int64_t len[2]);
int spmf_data_pass_split(spmf_ctx* ctx, const spmf_counts* counts, int S,
                         const float* const params[SPMF_NVARS], const float* eta, int part,
                         void* stream);

/* ---- the step's one collective, inside the library (RCCL over xGMI) ---------
 * Row shards of the count matrix live on different GPUs, one process per GPU
 * (the reference has no counterpart: only the `strategy` pass-through,
 * poisson.py:60,72).  Between spmf_data_pass and spmf_finish every rank sums the
 * packed accumulators:  spmf_allreduce(ctx, spmf_acc_ptr(ctx), spmf_acc_len(ctx,S),
 * stream) -- ncclAllReduce(float, sum) on the caller's stream, so the collective is
 * stream-ordered with the kernels around it and needs no host synchronisation.
 * librccl is bound at run time (dlopen); without it these return SPMF_E_UNSUPPORTED
 * and single-GPU use is unaffected.  Set-up: rank 0 calls spmf_comm_unique_id, the
 * 128 bytes travel to the other ranks by any channel the host has (the Python
 * mirror uses a torch.distributed broadcast), every rank calls spmf_comm_init. */
int spmf_comm_unique_id(void* out128);
int spmf_comm_init(spmf_ctx* ctx, const void* id128, int rank, int world);
int spmf_allreduce(spmf_ctx* ctx, float* buf, int64_t n, void* stream);
int spmf_comm_destroy(spmf_ctx* ctx);

/* ---- the same collective as a hand-written kernel over peer pointers (ABI 6) -----
 * SURVEY 5 (last row) / 8e ask for a direct reduce-scatter + all-gather over the xGMI mesh instead of a
 * ring: csrc/p2p.hip.  Every rank pushes slice q of its buffer straight into rank q's inbox (N-1 links at
 * once), rank q adds the N contributions IN RANK ORDER and pushes the reduced slice to every peer, every
 * rank copies the N-1 reduced slices home: ONE kernel launch per rank on the caller's stream (capturable
 * in a hipGraph: the call counter lives on the device), flags with system-scope release / acquire, and all
 * ranks end with the SAME bits (each slice is reduced once, by its owner, in a fixed order).  The memory
 * the peers write is a fine-grained region this library allocates and exports with hipIpcGetMemHandle; the
 * peers may be other GPUs of the node (xGMI) or other processes on the same GPU (how the one-GPU tests run
 * it at world 2 and 4) -- the kernel is the same.
 *   spmf_p2p_init     allocates this rank's region for buffers of up to n_max floats and returns its 64-byte
 *                     IPC handle; nchunk = workgroups of the kernel (0: default 32; <= 256)
 *   [ the host exchanges the handles: the Python mirror uses a torch.distributed all_gather ]
 *   spmf_p2p_connect  handles = world x 64 bytes in rank order; opens the peers' regions
 *   spmf_allreduce    then runs this kernel instead of ncclAllReduce (n <= n_max, buf 16-byte aligned)
 *   spmf_p2p_enable   a context may hold both transports (bench.py times them against each other): which one
 *                     spmf_allreduce uses; spmf_p2p_connect leaves the kernel selected
 *   spmf_p2p_status   SYNCHRONISES; out3 = {calls completed, 0, first call in which a workgroup gave up
 *                     waiting for a peer (0 = none)}: every spin is bounded, a lost peer cannot hang the GPU
 *   spmf_p2p_disconnect / spmf_p2p_destroy   an orderly shutdown frees no region a peer still has mapped: every rank
 *                     disconnects (unmaps the peers), the host synchronises the ranks, every rank destroys (unmaps
 *                     what is left and frees; also done by spmf_ctx_destroy)
 * All ranks must call spmf_allreduce with the same n, the same number of times. */
int spmf_p2p_init(spmf_ctx* ctx, int rank, int world, int64_t n_max, int nchunk, void* handle_out64);
int spmf_p2p_connect(spmf_ctx* ctx, const void* handles);
int spmf_p2p_enable(spmf_ctx* ctx, int on);   /* 0: spmf_allreduce goes back to RCCL (if spmf_comm_init was called); 1: the kernel again */
int spmf_p2p_status(spmf_ctx* ctx, uint64_t out3[3]);
int spmf_p2p_disconnect(spmf_ctx* ctx);   /* unmap the peers' regions, keep the own one (first half of an orderly shutdown) */
int spmf_p2p_destroy(spmf_ctx* ctx);

/* Phase 2: chain the accumulators to d/d(u,v,w,s), add the prior's parts and
 * gradients -- the horseshoe-plus hierarchy over all 12 variables (poisson.py:228-377)
 * or, for a context created with SPMF_FLAG_ABS_HORSESHOE, the AbsHorseshoe priors on u
 * and s with v, w as before (poisson.py:378-398; variables 0, 1, 2, 7 only) -- and
 * finish the 14 energy parts.  n_rows_global / lgamma_sum_global are the
 * batch totals over all shards (= this shard's when single GPU).
 *   parts[S][14] fp64 (unweighted), grads[i] has the shape of params[i]
 *   (fp32) and holds d(x + z + prior_weight * sum of prior parts)/d(param):
 *   prior_weight = 1 is the reference's energy (poisson.py:577 passes the
 *   literal 1.); a minibatch driver passes B/N.
 *   n_nonfinite[2*S] (fp64, may be NULL): [s] = stored cells of draw s whose
 *   log-pmf was not finite; the sparse fast path assumes 0 (poisson.py:606-616
 *   is then the identity).  [S+s] = saturation events of draw s: the number of
 *   workgroups of the dense exp kernel (128 rows x all columns each) in which a
 *   log_transform exponent exceeded 70 and was saturated there: fp32 cannot
 *   hold exp(y) beyond y ~ 88 where the fp64 reference still can, so the decoder
 *   is evaluated as exp(min(y,70)) - 1; 0 means the decoder was exact. */
int spmf_finish(spmf_ctx* ctx, int S, int64_t n_rows_global,
                double lgamma_sum_global, double prior_weight,
                const float* const params[SPMF_NVARS], const float* eta,
                double* parts, float* const grads[SPMF_NVARS],
                double* n_nonfinite, void* stream);

/* ---- the step with its outputs known up front (ABI 6) ------------------------
 * poisson.py:582-621 is ONE call; spmf_data_pass + spmf_finish split it in two so that the row-shard
 * all-reduce fits between them, but handed the gradient outputs over only at the second call, which
 * forced the prior's twelve log-densities and their gradients (poisson.py:590-591: parameters only, no
 * accumulator) into the step's LAST launch, behind the collective.  This pair is the same step with
 * parts / grads / n_nonfinite known from the first call on:
 *   spmf_step_begin  = spmf_data_pass, and the prior half of the finish runs INSIDE the data pass's
 *                      first launch, beside the A' / V' / phi tiles (O(D*K) work on the parameters, both);
 *   [ the caller's all-reduce of spmf_acc_ptr / spmf_acc_len, row shards only ]
 *   spmf_step_end    = the data half of spmf_finish (chain rule from the accumulators ADDED to what
 *                      the prior half left in grads, parts 'z' and 'x', the fold of the prior half's
 *                      per-workgroup sums) in ONE launch.
 * Four launches per step instead of five, none of them on a side stream, and what is independent of
 * the batch size shrinks to the prep launch + a 7 us data half (C3 sizes).  Arguments as
 * spmf_data_pass / spmf_finish; params, eta, parts, grads, n_nonfinite must stay valid until
 * spmf_step_end returns (the pointer ARRAYS are copied, the caller's may go).  Results are those of
 * spmf_data_pass + spmf_finish.  When the S draws of a large batch run in turn (S > 1 beyond the
 * batched-draw size) spmf_step_end falls back to the whole finish.  A spmf_data_pass, or anything that
 * re-binds the workspace, between the two calls cancels the step (spmf_step_end: SPMF_E_ARG). */
int spmf_step_begin(spmf_ctx* ctx, const spmf_counts* counts, int S, double prior_weight,
                    const float* const params[SPMF_NVARS], const float* eta, double* parts,
                    float* const grads[SPMF_NVARS], double* n_nonfinite, void* stream);
int spmf_step_end(spmf_ctx* ctx, int64_t n_rows_global, double lgamma_sum_global, void* stream);

/* spmf_step_begin + spmf_step_end back to back (single shard). */
int spmf_elbo_fwd_bwd(spmf_ctx* ctx, const spmf_counts* counts, int S,
                      double prior_weight,
                      const float* const params[SPMF_NVARS], const float* eta,
                      double* parts, float* const grads[SPMF_NVARS],
                      double* n_nonfinite, void* stream);

/* encode (poisson.py:623-650) for one (u,s) draw: z[B,K] row-major fp32. */
int spmf_encode(spmf_ctx* ctx, const spmf_counts* counts, const float* u,
                const float* s, const float* eta, float* z_out, void* stream);

/* log_likelihood_components (poisson.py:156-184, bernoulli.py:126-155) for ONE
 * draw, dense like the reference: rate[B,D] and the log-pmf ll[B,D] of every
 * cell (fp32, row-major): Poisson(rate), or for a Bernoulli context / the
 * Bernoulli columns of a mixed one rate = the logit and ll = x*logit -
 * softplus(logit).  Output bound (8 B per cell); not on the hot path.  Serves
 * the class surface and the non-finite replacement rule. */
int spmf_dense_ll(spmf_ctx* ctx, const spmf_counts* counts, const float* u,
                  const float* v, const float* w, const float* s,
                  const float* eta, float* rate_out, float* ll_out, void* stream);

/* Reductions of the non-finite rule (poisson.py:606-616) over a dense ll
 * buffer of n cells; io = double[3] on the device.
 *   pass 0: io[0] = min(io[0], min over finite cells)  (initialise io[0]=0:
 *           the reference's where(finite, ll, 0) puts 0 into the minimum)
 *   pass 1: io[1] += sum where(finite, clip(ll, io[0]-10, 0), io[0]-10),
 *           io[2] += number of non-finite cells. */
int spmf_nonfinite_reduce(spmf_ctx* ctx, int64_t n, const float* ll, int pass,
                          double* io, void* stream);

/* Third reduction of the rule: io[3] = min(io[3], index_base + i) over the
 * cells i of ll[0..n) whose value equals the global minimum io[0] (initialise
 * io[3] = +inf): the linear index (draw * B*D + row * D + column, built by the
 * caller through index_base) of the minimum's cell, where d(min_val) flows. */
int spmf_nonfinite_argmin(spmf_ctx* ctx, int64_t n, const float* ll, double index_base,
                          double* io, void* stream);

/* out[0] += sum of lgamma(x+1) over the stored Poisson cells of `counts` whose entry
 * in rate[n_rows, D] (spmf_dense_ll's output for the same counts) is not a positive
 * finite number: the cells the rule replaces.  Their lgamma is part of the batch's
 * pre-summed constant (counts.lgamma_sum) and spmf_nonfinite_patch takes it back out. */
int spmf_nonfinite_lgamma(spmf_ctx* ctx, const spmf_counts* counts, const float* rate,
                          double* out, void* stream);

/* Apply the rule (poisson.py:606-616) to the packed accumulators that
 * spmf_data_pass left for S draws of `counts`, VALUE AND GRADIENT, so that a
 * following spmf_finish returns the energy the reference computes when stored
 * cells have a non-finite log-pmf (rate 0 under a positive count):
 *   'x'_s  += nnf_s * (io[0] - 10) + sum over those cells of lgamma(x+1)
 *   draw s* (the draw of the minimum's cell io[3]): gA' / gV' / gphi gain
 *   (sum_s nnf_s) * d ll(cell) / d theta   -- clip is the identity on finite
 *   cells, replaced cells are worth min_val, whose derivative is that of the
 *   minimum's cell (tf.reduce_min).
 * io = double[4]: [0] global minimum over all S*B*D cells (spmf_nonfinite_reduce
 * pass 0 over spmf_dense_ll output), [3] its cell (spmf_nonfinite_argmin);
 * nlg = double[S]: spmf_nonfinite_lgamma per draw.
 * Row shards: call it on the shard's OWN (not yet summed) accumulators with
 * io[0] = the minimum over the shard minima, io[2] = sum_s nnf_s over ALL shards
 * (0: taken from the accumulators, the single-shard case) and io[3] = +inf on
 * every shard but the one that holds the minimum's cell; the value terms are
 * per shard and add up in the all-reduce that follows. */
int spmf_nonfinite_patch(spmf_ctx* ctx, const spmf_counts* counts, int S,
                         const float* const params[SPMF_NVARS], const float* eta,
                         const double* io, const double* nlg, void* stream);

/* ---- VI step around the energy: surrogate posterior and optimiser ------- */
/* One latent variable of the mean-field surrogate (poisson.py:403-569).
 * kind 0: theta = softplus(t0 + softplus(t1)*eps)      (tfb.Softplus(Normal))
 * kind 1: theta = t0 + softplus(t1)*eps                (tfb.Identity(Normal), bernoulli.py:187-193)
 * kind 2: theta = softplus(softplus(t1) / g), g ~ Gamma(softplus(t0), 1)
 *                                                      (tfb.Softplus(InverseGamma))
 * noise = eps or g, [S,n]; dgda = d g / d concentration [S,n] (kind 2). */
typedef struct spmf_sur_var {
  const float* t0;
  const float* t1;
  const float* noise;
  const float* dgda;
  float* theta;        /* [S,n] written by spmf_surrogate_fwd */
  const float* gtheta; /* [S,n] dE/dtheta, read by spmf_surrogate_bwd */
  float* g0;           /* [n] d loss / d t0, written by spmf_surrogate_bwd */
  float* g1;           /* [n] d loss / d t1 */
  int32_t n;
  int32_t kind;
  /* optional per-element override for kind 0 (mixed likelihood): ident[i] != 0
   * means element i has the Identity bijector (kind 1). NULL otherwise. */
  const uint8_t* ident;
  /* elements between consecutive draws in noise / dgda (they may be slices of
   * one [S, total] buffer drawn for all variables at once); 0 means n. */
  int64_t noise_ld;
} spmf_sur_var;

/* (spmf_sample_noise and spmf_surrogate_fwd skip a variable whose n is 0: nothing is read or written for it,
 * the other variables keep their indices -- the Philox counter of a draw and the order of the log q sum do
 * not depend on which variables a call covers, so a caller may draw / transform the variables in two calls.) */
/* Base noise for every variable, drawn on the device into the caller's noise
 * (and, kind 2, dgda) buffers: eps ~ N(0,1), or g ~ Gamma(softplus(t0), 1) with
 * its implicit-reparameterisation derivative d g / d concentration.  Counter-based
 * (Philox4x32-10): the draw is a pure function of (seed, counter [+ state[13]],
 * variable index, draw, element), so replicas on different ranks that pass the
 * same seed draw the same noise, and a step replayed from a hipGraph gets fresh
 * noise through the device-resident step counter state[13] (spmf_vi_gate advances
 * it; pass state = NULL to use `counter` alone). */
int spmf_sample_noise(spmf_ctx* ctx, const spmf_sur_var* vars, int nvars, int S, uint64_t seed,
                      uint64_t counter, const double* state, void* stream);

/* spmf_sample_noise + spmf_surrogate_fwd with the draw and the transform in ONE launch (ABI 6; the VI step's path): a
 * thread draws its element's base noise, transforms it while it is in registers, writes noise / dgda
 * (spmf_surrogate_bwd needs them), theta and its share of log q; a second small launch adds the per-workgroup sums up
 * in a fixed order.  Same draws and same theta bits as the two calls, logq equal to fp64 rounding (other partial-sum
 * groups); three launches become two. */
int spmf_sample_transform(spmf_ctx* ctx, const spmf_sur_var* vars, int nvars, int S, uint64_t seed,
                          uint64_t counter, const double* state, double* logq, void* stream);

/* theta for every variable and logq[S] (fp64) = sum over variables and
 * elements of log q(theta). */
int spmf_surrogate_fwd(spmf_ctx* ctx, const spmf_sur_var* vars, int nvars, int S,
                       double* logq, void* stream);
/* Gradient of  loss = -(1/(S*B)) sum_s [E_s - c*logq_s]  wrt the trainables,
 * given gtheta = dE/dtheta; inv_sb = 1/(S*B).  (E = x + z +
 * c*prior, c = B/N.) */
int spmf_surrogate_bwd(spmf_ctx* ctx, const spmf_sur_var* vars, int nvars, int S,
                       double inv_sb, double c, void* stream);

typedef struct spmf_adam_var {
  float* p;
  float* m;
  float* v;
  const float* g;
  int32_t n;
  int32_t reserved_;
} spmf_adam_var;
/* tf.keras-style Adam (bias-corrected, eps outside the sqrt) over up to 24
 * tensors in one launch; clip > 0 clips each gradient element to [-clip, clip]
 * first (clip_value, bin/factorize_csv.py:44-47); step is 1-based. */
int spmf_adam_step(spmf_ctx* ctx, const spmf_adam_var* tensors, int ntensors, double lr,
                   double beta1, double beta2, double eps, int step, double clip,
                   void* stream);

/* spmf_surrogate_bwd and spmf_adam_step_dev in ONE pass over the trainables (the
 * gradient of a trainable never goes to memory): tensors[2i], tensors[2i+1] are the
 * Adam records (p, m, v, n; g unused) of vars[i].t0 / vars[i].t1, p pointing at the
 * same memory; vars[i].g0 / g1 are not written.  Gated by state[9] like
 * spmf_adam_step_dev.  The training loop's step path; the two separate calls remain
 * for callers that want the gradient. */
int spmf_surrogate_bwd_adam_dev(spmf_ctx* ctx, const spmf_sur_var* vars, int nvars, int S,
                                double inv_sb, double c, const spmf_adam_var* tensors,
                                const double* state, void* stream);

/* Device-resident optimiser state, so that a whole VI step (noise, surrogate,
 * energy + gradient, chain rule, Adam) is a fixed launch sequence with no host
 * read-back and can be captured in a hipGraph and replayed.  state is a device
 * array of SPMF_VI_STATE_LEN doubles:
 *   [0] lr  [1] beta1  [2] beta2  [3] eps  [4] clip (0 = off)     (host-written)
 *   [5] beta1^t  [6] beta2^t  [7] t                               (init 1, 1, 0)
 *   [8] loss of the last step  [9] 1 if it was applied, 0 if skipped
 *   [10] sum of applied losses  [11] applied steps  [12] skipped steps
 *   [13] steps gated so far, applied or not (the RNG step counter of spmf_sample_noise)
 *   [14] saturation events (log_transform decoder: exp evaluated at min(y, 70); counted per
 *        workgroup) summed over the steps gated since the caller last zeroed it
 * spmf_vi_gate computes loss = -mean_s[x + z + c*(prior - log q)]/rows from the
 * [S,14] parts of spmf_finish and log q of spmf_surrogate_fwd (SURVEY 8a row
 * 14), marks the step skipped when the loss is not finite or a stored cell's
 * log-pmf was not ("Batch loss NaN, skipping",
 * notebooks/factorizing_random_noise.ipynb:122-420), and advances [5..7],
 * [10..12], [14].  n_nonfinite is the [2*S] array spmf_finish wrote ([0:S] non-finite stored
 * cells, [S:2S] saturation events) or NULL.  spmf_adam_step_dev is spmf_adam_step reading
 * every scalar from state; it does nothing when [9] == 0. */
#define SPMF_VI_STATE_LEN 16
int spmf_vi_gate(spmf_ctx* ctx, const double* parts, const double* logq,
                 const double* n_nonfinite, int S, double c, double rows, double* state,
                 void* stream);
int spmf_adam_step_dev(spmf_ctx* ctx, const spmf_adam_var* tensors, int ntensors,
                       const double* state, void* stream);

/* Test/diagnostic taps: per-row z and d/dz of the LAST draw processed by
 * spmf_data_pass, [B,KP] fp32 with KP = spmf_padded_k(). */
int spmf_padded_k(const spmf_ctx* ctx);
const float* spmf_z_ptr(const spmf_ctx* ctx);
const float* spmf_gz_ptr(const spmf_ctx* ctx);

/* Per-kernel device time, averaged over the (up to 64) most recent
 * spmf_data_pass + spmf_finish pairs issued since timing was enabled
 * (hipEvents recorded on `stream` between the kernels of the last draw; the
 * query synchronises, the hot path does not): ms[6] = prep, row pass (sparse
 * launches), column pass, finish, sum of all, dense exp kernels
 * (log_transform only, else 0).  For bench.py's roofline. */
int spmf_ctx_enable_timing(spmf_ctx* ctx, int on);
int spmf_last_timing(spmf_ctx* ctx, float* ms6);

#if defined(SPMF_BUILD)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* SPMF_HIP_H */
