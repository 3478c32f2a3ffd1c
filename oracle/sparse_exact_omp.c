/* sparse_exact_omp.c -- multithreaded C restatement of oracle/sparse_exact.py's
 * data_term (linear decoder) -- TEST INFRASTRUCTURE ONLY: the checker, and the
 * "port" that bench.py times on the host cores as cpu_baseline.  PARITY UNPINNED
 * like the rest of oracle/ (see oracle/spmf_oracle.py): pinned against the dense
 * fp64 oracle by tests/test_oracle.py.
 *
 * What it restates (fp64, one draw), from mederrata_spmf/poisson.py:
 *   z_b    = xi_b * sum_{d in nnz(b)} x_bd A'_d                 encode      :623-650
 *   r_bd   = <z_b, V'_d> + phi_d   on stored cells             rate        :174-177
 *   'x'    = sum_nnz [x log r - lgamma(x+1)] - sum_all r       log-pmf sum :178-183,617-619
 *            (sum over the implicit zeros in closed form: <sum_b z_b, sum_d V'_d> + B sum_d phi_d)
 *   'z'    = B K log(2/pi)/2 - sum z^2 / 2                     z-prior     :599-604
 *   gz_b   = sum_{d in nnz(b)} (x/r) V'_d - sum_d V'_d - z_b
 *   gV'_d  = sum_{b in nnz(d)} (x/r) z_b - sum_b z_b ; gphi_d = sum_b x/r - B
 *   gA'_d  = sum_{b in nnz(d)} x_bd xi_b gz_b
 * with A' = w1 u / eta, V' = eta v^T, phi = eta w2 w prepared by the caller
 * (oracle/sparse_exact.py does the O(D K) chain to u, v, w, s).  The row sweep is
 * parallel over rows, the transposed sweep over the columns of a CSC copy that the
 * caller builds once (the GPU path keeps its panel-CSC resident the same way).
 *
 *   gcc -O3 -fopenmp -shared -fPIC oracle/sparse_exact_omp.c -o oracle/_build/libsparse_exact_omp.so -lm
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int spx_max_threads(void) { return omp_get_max_threads(); }
void spx_set_threads(int n) { omp_set_num_threads(n); }

/* out_scalars: [0] sum_nnz x log r, [1] sum z^2, [2] non-finite stored cells,
 *              [3] sum_nnz lgamma(x+1)
 * zsum[K]; z, gz: [B,K] row-major; gAp, gVp: [D,K]; gphi: [D].  gVp / gphi hold the
 * stored-cell sums only (the closed-form -sum_b z_b and -B are the caller's). */
int spx_data_term(int64_t B, int32_t D, int32_t K, const int32_t* row_ptr, const int32_t* col,
                  const double* val, const int32_t* csc_ptr, const int32_t* csc_row,
                  const double* csc_val, const double* xi /* [B] or NULL */, const double* Ap,
                  const double* Vp, const double* phi, double* z, double* gz, double* gAp,
                  double* gVp, double* gphi, double* zsum, double* out_scalars) {
  if (B < 0 || D < 1 || K < 1 || K > 256) return -1;
  double veta[256];
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int32_t d = 0; d < D; ++d) s += Vp[(size_t)d * K + k];
    veta[k] = s;
  }
  double llx = 0.0, zsq = 0.0, nnf = 0.0, lgs = 0.0;
  const int nt = omp_get_max_threads();
  double* zs_t = (double*)calloc((size_t)nt * K, sizeof(double));
  if (!zs_t) return -2;
#pragma omp parallel reduction(+ : llx, zsq, nnf, lgs)
  {
    double* zs = zs_t + (size_t)omp_get_thread_num() * K;
    double zb[256], gb[256];
#pragma omp for schedule(dynamic, 256)
    for (int64_t b = 0; b < B; ++b) {
      const int32_t s0 = row_ptr[b], s1 = row_ptr[b + 1];
      const double x_i = xi ? xi[b] : 1.0;
      for (int k = 0; k < K; ++k) zb[k] = 0.0;
      for (int32_t e = s0; e < s1; ++e) {
        const double x = val[e];
        const double* a = Ap + (size_t)col[e] * K;
        for (int k = 0; k < K; ++k) zb[k] += x * a[k];
      }
      for (int k = 0; k < K; ++k) {
        zb[k] *= x_i;
        gb[k] = 0.0;
      }
      for (int32_t e = s0; e < s1; ++e) {
        const double x = val[e];
        const double* v = Vp + (size_t)col[e] * K;
        double r = phi[col[e]];
        for (int k = 0; k < K; ++k) r += zb[k] * v[k];
        lgs += lgamma(x + 1.0);
        if (r > 0.0 && r < INFINITY) {
          llx += x * log(r);
          const double c = x / r;
          for (int k = 0; k < K; ++k) gb[k] += c * v[k];
        } else {
          nnf += 1.0;
        }
      }
      double* zo = z + (size_t)b * K;
      double* go = gz + (size_t)b * K;
      for (int k = 0; k < K; ++k) {
        zo[k] = zb[k];
        go[k] = gb[k] - veta[k] - zb[k];
        zsq += zb[k] * zb[k];
        zs[k] += zb[k];
      }
    }
  }
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int t = 0; t < nt; ++t) s += zs_t[(size_t)t * K + k];
    zsum[k] = s;
  }
  free(zs_t);
  /* transposed sweep over the CSC copy */
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t d = 0; d < D; ++d) {
    double gv[256], ga[256];
    for (int k = 0; k < K; ++k) gv[k] = ga[k] = 0.0;
    double gp = 0.0;
    const double* v = Vp + (size_t)d * K;
    for (int32_t e = csc_ptr[d]; e < csc_ptr[d + 1]; ++e) {
      const int64_t b = csc_row[e];
      const double x = csc_val[e];
      const double* zb = z + (size_t)b * K;
      const double* gb = gz + (size_t)b * K;
      double r = phi[d];
      for (int k = 0; k < K; ++k) r += zb[k] * v[k];
      const double c = (r > 0.0 && r < INFINITY) ? x / r : 0.0;
      const double xx = x * (xi ? xi[b] : 1.0);
      for (int k = 0; k < K; ++k) {
        gv[k] += c * zb[k];
        ga[k] += xx * gb[k];
      }
      gp += c;
    }
    for (int k = 0; k < K; ++k) {
      gVp[(size_t)d * K + k] = gv[k];
      gAp[(size_t)d * K + k] = ga[k];
    }
    gphi[d] = gp;
  }
  out_scalars[0] = llx;
  out_scalars[1] = zsq;
  out_scalars[2] = nnf;
  out_scalars[3] = lgs;
  return 0;
}

/* The yardstick of an entry-wise gradient comparison (oracle/spmf_oracle.py
 * energy_grad_scales, data pieces): d/dA' of the three additive pieces of the data term
 * SEPARATELY, from the z / gz rows spx_data_term left:
 *   stored-cell piece  sum_nnz x log r :  gA'_pos_d = sum_b x xi_b (gz_b + sum_d V'_d + z_b)
 *   z prior           -sum z^2 / 2     :  gA'_zp_d  = -sum_b x xi_b z_b
 *   minus-rate piece  -sum_all r       :  gA'_neg_d = -(sum_b x xi_b) sum_d V'_d  -> sxx_d returned
 * (the pieces of gV' and gphi need no sweep: stored part = spx_data_term's raw gVp / gphi,
 * rate part = -sum_b z_b and -B.)  veta[K] = sum_d V'_d. */
int spx_grad_pieces(int64_t B, int32_t D, int32_t K, const int32_t* csc_ptr, const int32_t* csc_row,
                    const double* csc_val, const double* xi, const double* z, const double* gz,
                    const double* veta, double* gA_pos, double* gA_zp, double* sxx) {
  if (B < 0 || D < 1 || K < 1 || K > 256) return -1;
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t d = 0; d < D; ++d) {
    double gp[256], gq[256];
    for (int k = 0; k < K; ++k) gp[k] = gq[k] = 0.0;
    double sx = 0.0;
    for (int32_t e = csc_ptr[d]; e < csc_ptr[d + 1]; ++e) {
      const int64_t b = csc_row[e];
      const double xx = csc_val[e] * (xi ? xi[b] : 1.0);
      const double* zb = z + (size_t)b * K;
      const double* gb = gz + (size_t)b * K;
      for (int k = 0; k < K; ++k) {
        gp[k] += xx * (gb[k] + veta[k] + zb[k]);
        gq[k] -= xx * zb[k];
      }
      sx += xx;
    }
    for (int k = 0; k < K; ++k) {
      gA_pos[(size_t)d * K + k] = gp[k];
      gA_zp[(size_t)d * K + k] = gq[k];
    }
    sxx[d] = sx;
  }
  return 0;
}
