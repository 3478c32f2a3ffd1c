// kernels.h -- host-side launch interface of the libspmf_hip kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spmf {

struct PrepArgs {
  int D, K;
  const float *u, *v, *w, *s, *eta;
  float *Ap, *Vp, *phi;
  double* dprep;  // [KP+1]: veta[KP], phisum   (zeroed by the caller)
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
  int encode_only;
};
void launch_row_pass(int KP, const RowArgs& a, hipStream_t st);

struct ColArgs {
  int D, n_panels, row_base;
  int panels_per_wave;  // 1, 2 or 4 consecutive panels accumulated before the atomics
  const int32_t* pc_ptr;
  const int32_t* pc_row;
  const float* pc_val;
  const float *Vp, *phi, *z, *gzs;
  float *gAp, *gVp, *gphi;  // accumulated with float atomics (zeroed by the caller)
};
void launch_col_pass(int KP, const ColArgs& a, hipStream_t st);

struct PackArgs {
  int KP;
  const double* dacc;
  float* tail;  // acc tail: 2*(kDaccHead+KP) floats
};
void launch_pack(const PackArgs& a, hipStream_t st);

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
};
void launch_finish(int KP, const FinishArgs& a, hipStream_t st);

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

}  // namespace spmf
