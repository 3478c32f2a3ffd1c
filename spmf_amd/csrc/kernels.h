// kernels.h -- host-side launch interface of the libspmf_hip kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spmf {

struct PrepArgs {
  int D, K;
  const float *u, *v, *w, *s, *eta;
  float *Ap, *Vp, *phi;
  double* dprep;  // [kPrepSeg][KP+1]: partial sums of veta[KP], phisum (written, not accumulated;
                  // readers fold the segments: common.h prep_sum)
  int logt;       // log_transform: A' = w1*u (g(x) is data side), V' = eta*v^T
  const uint8_t* ctype;  // mixed likelihood: 1 = Bernoulli column (may be null)
  float* dbias;          // mixed likelihood: dense-kernel logit bias per column (may be null)
  // S > 1: the launch covers S draws (gridDim.y); pointers are those of draw 0, draw s adds
  // s * D*K to u / v, s * D to w, s * 2D to s, s * D*KP to Ap / Vp, s * D to phi, s * kPrepSeg*(KP+1) to dprep
  int S;
  // optional zero fill folded into the launch (the step's acc | dacc): 16-byte multiple, or null
  void* zero_p = nullptr;
  size_t zero_bytes = 0;
};
void launch_prep(int KP, const PrepArgs& a, hipStream_t st);

struct RowArgs {
  int64_t B;
  const int32_t* row_ptr;
  const int32_t* col;
  const float* val;
  const float* row_scale;  // may be null
  const float *Ap, *Vp, *phi;
  const double* dprep;
  float *z, *gzs;
  double* dacc;  // [kDaccHead+KP] (zeroed by the caller)
  int mode;            // 0 full (linear decoder), 1 sweep 1 only (encode), 2 sweep 2 only, 3 full with the dense
                       // row term left to the dense kernel's epilogue (launch_sigdot3 p_scale / accumulate)
  int logt;            // log_transform rate r = exp(<z,V'>) - 1 + phi
  const float* gzd;    // mode 2: per-row dense term sum_d E_bd V'_d  [B,KP]
  const uint8_t* ctype;  // likelihood code 3 (mixed): column types
  // S > 1 (mode 0 only): S draws per launch (gridDim.y); draw s adds s * D*KP to Ap / Vp, s * D to
  // phi, s * kPrepSeg*(KP+1) to dprep, s * B*KP to z / gzs, s * dacc_stride to dacc
  int S, D;
  int64_t dacc_stride;
  const uint32_t* ent = nullptr;   // packed col << 16 | count copy of (col, val), or null (spmf_counts.ent)
  // mode 3 with the exp decoder (LIK 1): `val` is g(x) (sweep 1) and the counts of sweep 2 come out of `ent`
  // (must be non-null); only the LDS-phi launch shapes have this form: launch_row_pass returns false otherwise
  int dual = 0;
  // deterministic mode: the workgroups' scalar sums go to their own slots (kDetMeta + kDetMaxBlocks *
  // (kDaccHead + KP) doubles per draw, stride det_stride) instead of the fp64 atomics on dacc
  double* det_slots = nullptr;
  int64_t det_stride = 0;
  // 1: the last eighth of every wave's rows is handed out by counters (the spare slot [5] of the kDaccRep replicas
  // of `dacc`, zeroed by the caller with the rest of it) instead of by the fixed stride; only for the ONE
  // full row launch of a step (modes 0 and 3), ignored in the deterministic mode
  int dyn_tail = 0;
};
bool launch_row_pass(int KP, const RowArgs& a, hipStream_t st);   // false: a.dual asked for a form this shape lacks (nothing launched)
bool launch_row_widek(int KP, const RowArgs& a, hipStream_t st);  // KP = 128, 256 (widek.hip): Poisson / linear decoder, modes 0 and 1

struct ColArgs {
  int D, n_panels, row_base;
  int max_items_per_panel;
  const int32_t* item_ptr;  // [n_panels+1]
  const int32_t* items;     // [n_items][4] = {start, len, column, 0}
  const int32_t* pc_row;
  const float* pc_val;
  const float *Vp, *phi, *z, *gzs;
  float *gAp, *gVp, *gphi;  // accumulated with float atomics (zeroed by the caller)
  int logt;
  const float* pc_gval;     // log_transform: g(x) per panel-CSC entry
  const uint8_t* ctype;     // likelihood code 3 (mixed): column types
  const int32_t* item_mid;  // [n_panels] first item of the upper column half, or null
  int half_sel;             // 0 all items; 1 / 2: lower / upper column half only (needs item_mid)
  // S > 1: S draws per launch (gridDim.y); draw s adds s * D*KP to Vp, s * D to phi, s * B*KP to
  // z / gzs and s * acc_stride to gAp / gVp / gphi
  int S;
  int64_t B, acc_stride;
  int pc_pad = 0;           // readable entries behind the last list of pc_row / pc_val / pc_gval
  const uint32_t* pc_ent = nullptr;   // packed (row in panel << 16 | count) copy of pc_row / pc_val, or null
  int panel_rows = 0;
  // optional: one extra block folds the row pass's fp64 scalar block (kDaccRep replicas) into
  // the accumulator tail as (hi, lo) float pairs -- what pack_kernel does as its own launch
  const double* pack_dacc = nullptr;
  float* pack_tail = nullptr;
  int64_t dacc_stride = 0;
  // deterministic mode: per-item partial sums (det_part_len(KP) floats per item of the batch, item index
  // relative to item_ptr[0]; stride det_part_stride floats per draw) instead of the float atomics, and the
  // pack block reads the row pass's per-workgroup slots
  float* det_part = nullptr;
  int64_t det_part_stride = 0;
  const double* det_slots = nullptr;
  int64_t det_stride = 0;
};
// deterministic mode: adds the per-item partials up column by column in (panel, segment) order
struct DetReduceArgs {
  int D, KP, n_panels, S;
  const int32_t* list_first;   // [n_panels * D + 1], offset to the batch's first panel
  const int32_t* item_pos;
  const int32_t* item_ptr;     // [0] = first item of the batch
  const float* part;
  int64_t part_stride;
  float *gAp, *gVp, *gphi;
  int64_t acc_stride;
};
void launch_det_reduce(const DetReduceArgs& a, hipStream_t st);

struct ExpdotArgs {
  int NP, NQ;
  const float *P, *Q;  // [NP,KD], [NQ,KD]
  float* out;          // [NP,KD]
  float sign;
  double* esum;        // may be null
  int q_chunks;        // gridDim.y; >1 needs atomic_out
  int atomic_out;
  int act;             // 0 exp (Poisson log_transform), 1 sigmoid/softplus (Bernoulli)
  const float *bias_p, *bias_q;  // act 1: logit bias per P row / per Q row (one of them)
  float* out2;         // act 1: out2[p] += sign * sum_q sigmoid (may be null)
  const int32_t* out_rows;   // P is a compacted row subset: out / out2 rows to write (may be null)
  float* est = nullptr;      // keep E (exp or sigmoid) for launch_estdot (layout: dense.hip), ldE = its P extent
  int64_t ldE = 0;
  int e_planes = 2;          // launch_sigdot3: bf16 planes of E in the second product (3 where the Q rows have mixed signs)
  // launch_sigdot3: out[p] += sign * p_scale[p] * result instead of out[p] = sign * result (p_scale may be
  // null = 1; plain read-modify-write unless atomic_out): the fused row pass leaves xi_b (gz_b - z_b) in
  // gzs and this launch subtracts xi_b * sum_d E_bd V'_d from it
  int accumulate = 0;
  const float* p_scale = nullptr;
};
void launch_expdot(int KD, const ExpdotArgs& a, hipStream_t st);
// dense3.hip: the same operator (act 0, KD = 64, no biases / E store) on the bf16 matrix cores with
// three-way split operands; false = this shape is not covered (use launch_expdot)
bool launch_expdot3(int KD, const ExpdotArgs& a, hipStream_t st);
// dense3.hip: the sigmoid / softplus operator (act 1: Bernoulli and mixed columns, KD = 32 or 64 (64: the step's two launch shapes only), one of
// bias_p / bias_q, out2, out_rows; no E store) on the same bf16x3 operands; false = not covered
bool launch_sigdot3(int KD, const ExpdotArgs& a, hipStream_t st);
int expdot3_rows_per_wg();
int expdot3_wgs_per_cu();
int sigdot3_rows_per_wg(int KD);
int sigdot3_wgs_per_cu(int KD);
void launch_estdot(int KD, int NQ, int NP, int64_t ldE, const float* est, const float* P, float* out, float sign,
                   float* out2, const int32_t* out_rows, hipStream_t st);
void launch_compact_rows(int n, int KD, const int32_t* cols, const float* Vp, const float* phi, float* Vb,
                         float* bb, hipStream_t st);

struct DenseLLArgs {
  int64_t B;
  int D, logt;       // logt = likelihood code 0..3
  const float *z, *Vp, *phi;
  const uint8_t* ctype;
  const int32_t* row_ptr;
  const int32_t* col;
  const float* val;
  float *rate, *ll;  // [B,D] each
};
void launch_dense_ll(int KP, const DenseLLArgs& a, hipStream_t st);
// pass 0: io[0] = min(io[0], finite ll); pass 1: io[1] += clipped/replaced sum, io[2] += #non-finite
void launch_nonfinite(int64_t n, const float* ll, int pass, double* io, hipStream_t st);
void launch_nonfinite_argmin(int64_t n, const float* ll, double index_base, double* io, hipStream_t st);
struct NfPatchArgs {
  int D, K, logt;
  const int32_t* row_ptr;
  const int32_t* col;
  const float* val;
  const float* row_scale;          // may be null
  const float *u, *v, *w, *s;      // draw 0; draw q adds q * D*K / K*D / D / 2D
  const float* eta;
  const uint8_t* ctype;
  float* acc;                      // packed accumulators of draw 0
  int64_t acc_stride;
  int Dh;                          // column split of the accumulator layout (D = none)
  const double* io;                // [0] global minimum, [3] linear index of its cell
  const double* nlg;               // [S] sum of lgamma(x+1) over each draw's replaced cells
  int64_t rows_batch;
  int S;
};
void launch_nonfinite_patch(int KP, const NfPatchArgs& a, hipStream_t st);
void launch_nonfinite_lgamma(const DenseLLArgs& a, double* out, hipStream_t st);   // uses B, D, logt, ctype, CSR, rate
bool launch_col_pass(int KP, const ColArgs& a, hipStream_t st);   // true: launched, with the pack block if asked
bool launch_col_widek(int KP, const ColArgs& a, hipStream_t st);  // KP = 128, 256 (widek.hip)

struct PackArgs {
  int KP;
  const double* dacc;
  float* tail;  // acc tail: 2*(kDaccHead+KP) floats
  int S;        // draws (gridDim.y): draw s adds s * dacc_stride / s * acc_stride
  int64_t dacc_stride, acc_stride;
};
void launch_pack(const PackArgs& a, hipStream_t st);
void launch_zero(void* p, size_t bytes, hipStream_t st);   // zero fill as a kernel (see stats.hip)

struct FinishArgs {
  int D, K;
  int64_t B_global;
  double lgamma_sum;
  double u_tau_scale, s_tau_scale, decay, prior_weight;
  const float* acc;     // this draw's accumulators (after any all-reduce)
  const double* dprep;  // this draw's veta/phisum
  const float* const* params;  // 12 device pointers (this draw)
  const float* eta;
  float* const* grads;  // 12 device pointers (this draw)
  double* parts;        // [14] (zeroed by the caller)
  double* n_nonfinite;  // [1] or null
  int logt;
  const uint8_t* ctype;  // likelihood code 3 (mixed): column types
  int Dh;                // column split of the accumulator layout (0 / D = none; multiple of 32)
  // S > 1: S draws per launch (gridDim.y); draw s adds s * vstride[i] to params[i] / grads[i],
  // s * acc_stride to acc, s * kPrepSeg*(KP+1) to dprep, s * 14 to parts, s to n_nonfinite
  int S;
  int64_t acc_stride;
  int64_t vstride[12];
  int abs_horseshoe;     // horshoe_plus=False: only params/grads 0, 1, 2, 7 are used
  double* ppart;         // [S][ceil(D/32)][12] per-block prior-part sums (workspace)
  float* putau;          // [S][ceil(D/32)][KP] per-block u_tau gradient sums (workspace)
};
void launch_finish(int KP, const FinishArgs& a, int phase, hipStream_t st);
// The step as spmf_step_begin / spmf_step_end run it (ABI 6): the prior half of the finish inside the prep
// launch (prep.hip begin_kernel; f.acc / f.dprep / f.n_nonfinite unused, f.S == a.S), and the data half with
// the fold of the prior half's per-block sums as the step's last launch (finish.hip end_kernel).
void launch_step_begin(int KP, const PrepArgs& a, const FinishArgs& f, hipStream_t st);
void launch_step_end(int KP, const FinishArgs& a, hipStream_t st);

// ---- surrogate posterior / optimiser (surrogate.hip) ----------------------
struct SurVar {
  const float* t0;      // loc            | raw concentration
  const float* t1;      // raw scale      | raw scale
  const float* noise;   // [S,n] eps ~ N(0,1) | g ~ Gamma(softplus(t0), 1)
  const float* dgda;    // [S,n] d g / d a (kind 2 only)
  float* theta;         // [S,n] out (fwd)
  const float* gtheta;  // [S,n] dE/dtheta (bwd)
  float *g0, *g1;       // [n] out (bwd)
  int n, kind;          // kind 0 softplus-normal, 1 identity-normal, 2 softplus-invgamma
  const uint8_t* ident; // optional per-element kind-1 override of kind 0
  int64_t ld;           // stride between draws in noise / dgda
};
struct SurTable {
  SurVar v[12];
};
struct AdamVar {
  float *p, *m, *v;
  const float* g;
  int n;
};
struct AdamTable {
  AdamVar v[24];
};
void launch_surrogate_fwd(const SurTable& T, int nvars, int max_n, int S, double* logq, double* scratch,
                          size_t scratch_doubles, hipStream_t st);
void launch_sample_noise(const SurTable& T, int nvars, int max_n, int S, uint64_t seed, uint64_t counter,
                         const double* state, hipStream_t st);
// sample + transform in one launch + the fold of the log q block sums; false: scratch too small for S * blocks
// partial sums (nothing launched)
bool launch_sample_fwd(const SurTable& T, int nvars, int max_n, int S, uint64_t seed, uint64_t counter,
                       const double* state, double* logq, double* scratch, size_t scratch_doubles, hipStream_t st);
void launch_surrogate_bwd(const SurTable& T, int nvars, int max_n, int S, float inv_sb, float c, hipStream_t st);
void launch_surrogate_bwd_adam(const SurTable& T, const AdamTable& A, int nvars, int max_n, int S, float inv_sb,
                               float c, const double* state, hipStream_t st);
void launch_vi_gate(const double* parts, const double* logq, const double* nnf, int S, double c, double rows, double* state, hipStream_t st);
void launch_adam_dev(const AdamTable& T, int ntensors, int max_n, const double* state, hipStream_t st);
void launch_adam(const AdamTable& T, int ntensors, int max_n, float lr, float b1, float b2, float eps, float c1, float c2, float clip, hipStream_t st);

// ---- the step's collective over peer pointers (p2p.hip) ---------------------------------
constexpr int kP2PMaxWorld = 16;    // ranks of one node
constexpr int kP2PMaxChunks = 256;  // workgroups of the collective's kernel
struct P2PLaunch {
  float* buf;
  int64_t n;
  int rank, world, nchunk;
  int64_t slice_cap;
  float *rs, *ag;
  uint64_t *flags, *seq;
  float* peer_rs[kP2PMaxWorld];
  float* peer_ag[kP2PMaxWorld];
  uint64_t* peer_flags[kP2PMaxWorld];
};
void launch_p2p_allreduce(const P2PLaunch& L, hipStream_t st);

struct StatsArgs {
  int64_t B;
  const int32_t* row_ptr;
  const int32_t* col;
  const float* val;
  double *colsum, *colnnz;
  float* row_sum;
  double* row_lgamma;
};
void launch_stats(const StatsArgs& a, hipStream_t st);
void launch_gvals(int64_t nnz, int n_panels, int D, const int32_t* row_ptr, const int32_t* col, const float* val,
                  const int32_t* pc_ptr, const float* pc_val, const float* eta, float* gval, float* pc_gval,
                  hipStream_t st);
void launch_colstats(int n_panels, int D, const int32_t* pc_ptr, const float* pc_val, double* colsum,
                     double* colnnz, hipStream_t st);

}  // namespace spmf
