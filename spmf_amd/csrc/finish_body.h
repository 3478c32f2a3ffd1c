// finish_body.h -- body of the finish kernel (chain rule from the sparse accumulators, the prior's
// parts and gradients, the 14 energy parts) as a device function, so that its PRIOR half can run
// inside the step's first launch beside the prep tiles (prep.hip begin_kernel: neither reads an
// accumulator) and its DATA half in the step's last launch (finish.hip end_kernel).
//
//   data term:   dA = gA'/eta        du = w1*dA
//       dv[k,d] = eta_d (gV'[d,k] - sum_b z_b[k])
//       dphi_d  = gphi_d - B ; dw = eta w2 dphi
//       ds0 = (GA - Gphi) s1/T^2, ds1 = (Gphi - GA) s0/T^2,
//       GA_d = sum_k u dA,  Gphi_d = eta w dphi,  T = s0+s1
//   prior (poisson.py:228-377, tfd.HalfNormal / tfd.InverseGamma and
//   bayesianquilts SqrtInverseGamma restated):
//       HalfNormal(sig)(y)      = c0 - log sig - y^2/(2 sig^2)
//       SqrtInvGamma(1/2,1/a)(y)= -log(a)/2 - lgamma(1/2) - 2 log y - 1/(a y^2) + log 2
//       InvGamma(1/2,beta)(a)   = log(beta)/2 - lgamma(1/2) - 3/2 log a - beta/a
//   parts 'x','z' (poisson.py:599-619) from the fp64 scalars in the
//   accumulator tail and the closed-form sum of the rate over all cells.
//
// One block per FTD features; [D,K]-shaped arrays are walked flat (coalesced),
// the [K,D]-shaped v / dv through an LDS transpose tile.  O(D*K) elementwise
// work with fp64 block reductions.
#pragma once
#include "common.h"
#include "kernels.h"

namespace spmf {

constexpr int FTD = kFinishCols;
constexpr int SPMF_NPARTS_LOCAL = 12;

struct Ptrs12 {
  const float* p[12];
};
struct MPtrs12 {
  float* p[12];
};
struct VStride {
  int64_t v[12];
};

enum { V_ = 0, W_, U_, UETA_, UTAU_, SETA_, STAU_, S_, UETAA_, UTAUA_, SETAA_, STAUA_ };

__device__ __forceinline__ double unpack(const float* tail, int i) {
  return (double)tail[2 * i] + (double)tail[2 * i + 1];
}

// fp64 log for the log-densities.  The library routine is ~100 fp64 instructions; a block of
// this kernel is one wave per SIMD working through a few thousand dependent instructions, so
// the kernel's time IS that chain.  Arguments here are positive, finite fp32 values (or
// products of two): x = m 2^e with m in [sqrt(1/2), sqrt(2)), t = (m-1)/(m+1), |t| <= 0.1716,
// log m = 2t (1 + t^2/3 + ... + t^22/23): truncation 0.0295^12/25 ~ 2e-20 relative.  The
// series and the range reduction are restated in numpy and compared with np.log in
// tests/test_host.py::test_finish_fast_log_series; the device path (v_rcp_f64 + two Newton
// steps) is covered by every GPU parity test of the energy parts.  Zero, negative, infinite and
// NaN arguments take the library log, so a parameter that underflowed still yields -inf / NaN.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);      // v_rcp_f64: ~1e-8 relative
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// (out of line and cold: inlined at every call site the library log was most of this body's code, and
//  the body ran out of the instruction cache)
__device__ __attribute__((noinline, cold)) static double slow_log(double x) { return log(x); }
// fp32 reciprocal of the prior's gradient terms: v_rcp_f32 (1 ulp) instead of the IEEE division sequence
// (ten instructions each, three per element in the [D,K] loop)
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
#ifndef FIN_ABL
#define FIN_ABL 0   // timing-only ablations of the prior half (tools/build_variant.sh): 1 main, 2 vectors, 4 v, 8 sums
#endif
__device__ __forceinline__ double fast_log(double x) {
  if (__builtin_expect(!(x > 0.0) || x > 1.7e308, 0)) return slow_log(x);
  const long long bits = __double_as_longlong(x);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);   // [1, 2)
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  const double t = (m - 1.0) * fast_rcp(m + 1.0);
  const double t2 = t * t;
  double p = 1.0 / 23.0;
  p = fma(p, t2, 1.0 / 21.0);
  p = fma(p, t2, 1.0 / 19.0);
  p = fma(p, t2, 1.0 / 17.0);
  p = fma(p, t2, 1.0 / 15.0);
  p = fma(p, t2, 1.0 / 13.0);
  p = fma(p, t2, 1.0 / 11.0);
  p = fma(p, t2, 1.0 / 9.0);
  p = fma(p, t2, 1.0 / 7.0);
  p = fma(p, t2, 1.0 / 5.0);
  p = fma(p, t2, 1.0 / 3.0);
  p = fma(p, t2, 1.0);
  return fma((double)e, kLog2, 2.0 * t * p);
}
// decay^t for integer t >= 0 by squaring (pow() is several hundred fp64 instructions and sat
// on every block's prologue)
__device__ __forceinline__ double ipow(double b, int t) {
  double r = 1.0;
  while (t) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

// The log-densities (the energy PARTS) are evaluated in fp64: a part is a sum
// of O(D*K) terms of either sign, so fp32 term error would be amplified by the
// cancellation (seen: 2.6e-5 relative on a 4307-term part).  Gradients stay fp32.
// HalfNormal(sig) at y: log-prob, d/dy, d/dsig
__device__ __forceinline__ void halfnormal(float y, float sig, double& lp, float& gy, float& gs) {
  const float is = frcp(sig);
  const float q = y * is;
  const double qd = (double)y * fast_rcp((double)sig);
  lp = kHalfLog2OverPi - fast_log((double)sig) - 0.5 * qd * qd;
  gy = -q * is;
  gs = (q * q - 1.f) * is;
}
// SqrtInvGamma(1/2, scale=1/a) at y: log-prob, d/dy, d/da
__device__ __forceinline__ void sqrt_ig(float y, float a, double& lp, float& gy, float& ga) {
  const float iy = frcp(y), ia = frcp(a);
  const float t = ia * iy * iy;  // 1/(a y^2)
  const double yd = (double)y, ad = (double)a;
  lp = -0.5 * fast_log(ad) - kLgammaHalf - 2.0 * fast_log(yd) - fast_rcp(ad * yd * yd) + kLog2;
  gy = -2.f * iy + 2.f * t * iy;
  ga = -0.5f * ia + t * ia;
}
// InvGamma(1/2, beta) at a: log-prob, d/da
__device__ __forceinline__ void ig_half(float a, float beta, float half_log_beta, double& lp,
                                        float& ga) {
  const float ia = frcp(a);
  const double ad = (double)a;
  lp = 0.5 * fast_log((double)beta) - kLgammaHalf - 1.5 * fast_log(ad) - (double)beta * fast_rcp(ad);
  ga = -1.5f * ia + beta * ia * ia;
}

// tfd.Horseshoe(scale).log_prob (TFP's closed-form approximation of the HalfCauchy-Normal
// marginal, tensorflow_probability/python/distributions/horseshoe.py; restated and pinned
// against quadrature in oracle/spmf_oracle.py horseshoe_log_prob) folded onto x >= 0
// (bayesianquilts AbsHorseshoe, poisson.py:382,391), and its derivative in x.
__device__ __forceinline__ void abs_horseshoe(double x, double sigma, double& lp, double& dlp) {
  constexpr double g = 0.5614594835668851, b = 1.0420764938351215, h_inf = 1.0801359952503342;
  constexpr double p = 1.0919284281983377;
  const double xs = x / sigma;
  const double t = 0.5 * xs * xs;
  const double st = sqrt(t), t15 = t * st;
  const double q = (20.0 / 47.0) * pow(t, p);
  const double h = 1.0 / (1.0 + t15) + h_inf * q / (1.0 + q);
  const double a = (log1p(-g) - log(g)) - t / (1.0 - g);
  const double sp = a > 0.0 ? a + log1p(exp(-a)) : log1p(exp(a));
  const double hb = h + b * t;
  const double A = g / t - (1.0 - g) / (hb * hb);
  const double L = log1p(A);
  lp = -sp + log(L) - 0.5 * log(2.0 * 3.14159265358979323846 * 3.14159265358979323846 * 3.14159265358979323846)
       - log(g * sigma) + kLog2;
  const double sg = 1.0 / (1.0 + exp(-a));
  const double dq = p * q / t;
  const double dh = -1.5 * st / ((1.0 + t15) * (1.0 + t15)) + h_inf * dq / ((1.0 + q) * (1.0 + q));
  const double dA = -g / (t * t) + 2.0 * (1.0 - g) * (dh + b) / (hb * hb * hb);
  dlp = (sg / (1.0 - g) + dA / ((1.0 + A) * L)) * x / (sigma * sigma);
}


// The twelve per-lane fp64 part sums -> their twelve wave sums in 14 exchanges instead of 72: the same
// butterfly (xor 32, 16, 8, 4, 2, 1) as wave_sum, but at the first four steps a lane KEEPS half of the
// values it still holds and hands the other half to its partner, which keeps exactly those -- every sum is
// formed from the same pairs in the same order as wave_sum forms it (bit-identical), just not in every
// lane.  The lane with (lane & 3) == 0 of each 4-lane group ends up holding part
// 6*b5 + 3*b4 + (b3 ? 2 : b2) (b_i = bit i of the lane id) and stores it to pred[part][wid].
__device__ __forceinline__ void parts_wave_sums(const double (&part)[12], int lane, int wid, double (*pred)[4]) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
  double a[6], c[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double keep = b5 ? part[i + 6] : part[i], send = b5 ? part[i] : part[i + 6];
    a[i] = keep + __shfl_xor(send, 32);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double keep = b4 ? a[i + 3] : a[i], send = b4 ? a[i] : a[i + 3];
    c[i] = keep + __shfl_xor(send, 16);
  }
  // three values: the b3 = 0 side keeps c0, c1, the b3 = 1 side keeps c2
  const double r0 = __shfl_xor(b3 ? c[0] : c[2], 8), r1 = __shfl_xor(b3 ? c[1] : 0.0, 8);
  const double e0 = (b3 ? c[2] : c[0]) + r0, e1 = b3 ? 0.0 : c[1] + r1;
  // b3 = 0: two values, b2 = 0 keeps e0, b2 = 1 keeps e1; b3 = 1: the plain butterfly on e0
  const double keep = b3 ? e0 : (b2 ? e1 : e0), send = b3 ? e0 : (b2 ? e0 : e1);
  double f = keep + __shfl_xor(send, 4);
  f += __shfl_xor(f, 2);
  f += __shfl_xor(f, 1);
  const int idx = (b5 ? 6 : 0) + (b4 ? 3 : 0) + (b3 ? 2 : (b2 ? 1 : 0));
  if ((lane & 3) == 0 && !(b3 && b2)) pred[idx][wid] = f;
}

// kernel arguments of the finish body (one struct: the body runs inside three kernels)
struct FinishK {
  int D, K;
  double Bglob, lgamma_sum;
  float u_tau_scale, s_tau_scale;
  double decay;
  float pw;
  const float* acc;
  const double* dprep;
  Ptrs12 P;
  const float* eta;
  MPtrs12 G;
  double* parts;
  double* nnf_out;
  int logt;
  const uint8_t* ctype;
  int Dh;
  int64_t acc_stride;
  VStride VS;
  int hs;
  double* ppart;
  float* putau;
};

// PHASE 0: everything (one launch).  PHASE 1: the prior half only -- all twelve
// log-densities and pw * d prior/d theta written to G; needs nothing from the
// data pass.  PHASE 2: the data half only -- the chain rule from the accumulators ADDED to
// G, and parts 'z', 'x'.  PHASE 1 then 2 == PHASE 0.
// (bx, by) of (nbx, nby): the block's place in the finish grid (ceil(D / FTD) blocks x S draws).
// HS: horshoe_plus=False (the AbsHorseshoe branch, a few thousand instructions of fp64 library code) is a
// separate instantiation, so the default model's kernels do not carry it.
template <int KP, int PHASE, bool HS>
__device__ __forceinline__ void finish_body(const FinishK& a_, const int bx, const int by, const int nbx,
                                            const int nby) {
  const int D = a_.D, K = a_.K;
  const double Bglob = a_.Bglob, lgamma_sum = a_.lgamma_sum;
  const float u_tau_scale = a_.u_tau_scale, s_tau_scale = a_.s_tau_scale;
  const double decay = a_.decay;
  const float pw = a_.pw;
  const float* __restrict__ acc = a_.acc;
  const double* __restrict__ dprep = a_.dprep;
  Ptrs12 P = a_.P;
  const float* __restrict__ eta = a_.eta;
  MPtrs12 G = a_.G;
  double* __restrict__ parts = a_.parts;
  double* __restrict__ nnf_out = a_.nnf_out;
  const int logt = a_.logt;
  const uint8_t* __restrict__ ctype = a_.ctype;
  const int Dh = a_.Dh;
  const int64_t acc_stride = a_.acc_stride;
  const VStride VS = a_.VS;
  constexpr bool hs = HS;
  double* __restrict__ ppart = a_.ppart;
  float* __restrict__ putau = a_.putau;
  // hs: horshoe_plus=False (poisson.py:378-398): AbsHorseshoe priors on u and s, no
  // scale hierarchy -- only P/G[V_, W_, U_, S_] are touched
  constexpr bool PRIOR = PHASE != 2, DATA = PHASE != 1;
  if (nby > 1) {   // S draws per launch: everything per draw moves by its stride
    const size_t sd = by;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (P.p[i]) P.p[i] += sd * (size_t)VS.v[i];
      if (G.p[i]) G.p[i] += sd * (size_t)VS.v[i];
    }
    if (acc) acc += sd * (size_t)acc_stride;
    if (dprep) dprep += sd * (size_t)kPrepSeg * (KP + 1);
    parts += sd * 14;
    if (nnf_out) nnf_out += sd;
  }
  // Cross-block sums (the twelve prior parts, the u_tau gradient) go to per-block slots
  // that finish_reduce_kernel adds up in block order: given the same accumulators the
  // results are bit-identical from run to run and from rank to rank (the replicated
  // parameters of a row-sharded job cannot drift apart through this kernel).
  if (ppart) ppart += ((size_t)by * nbx + bx) * 12;
  if (putau) putau += ((size_t)by * nbx + bx) * KP;
  __shared__ float tile[KP][FTD + 1];
  __shared__ float w1s[FTD], ietas[FTD], etas_[FTD], GAs[FTD];
  __shared__ float zsum_s[KP], utau_s[KP], dec_s[KP], gutau_s[KP];
  __shared__ double lsc_s[KP];     // log(u_tau_k * decay^k)
  __shared__ double scd_s[KP];     // u_tau_k * decay^k in fp64 (an fp32 product is a per-k systematic
                                   // error of the quadratic term: 8e-6 of part 'u' at K ~ 60)
  __shared__ float gred[256];
  __shared__ int bern_s[FTD];
  __shared__ float GAw[KP > 64 ? KP / 64 : 1][FTD];   // K above 64: per-wave parts of GA_d
  __shared__ double rterm_s[KP > 64 ? KP : 1];
  const int t = threadIdx.x;
  const int d0 = bx * FTD;
  // column-split layout (common.h AccLayout): Dh is a multiple of FTD, so a block's
  // columns lie in one half and the choice of base pointers is block-uniform
  const AccLayout L{D, KP, Dh};
  const int hf = d0 >= Dh ? 1 : 0;
  const float* gAp = acc + L.gA_off(hf);
  const float* gVp = acc + L.gV_off(hf);
  const float* gph = acc + L.gphi_off(hf);
  const float* tail = acc + L.tail_off();
  double part[SPMF_NPARTS_LOCAL];
#pragma unroll
  for (int i = 0; i < SPMF_NPARTS_LOCAL; ++i) part[i] = 0.0;

  // Every global operand of the kernel is fetched here, before the first barrier and the
  // first store: the grid is only a few waves per SIMD, so each later batch of loads
  // (the output pointers may alias the inputs as far as the compiler knows, and a load
  // cannot move above a barrier or an earlier store) would cost its own memory round
  // trip in series -- five of them were most of this kernel's time.
  constexpr int NIT = (KP * FTD + 255) / 256;
  float in_u[NIT], in_ue[NIT], in_ua[NIT], in_ga[NIT], in_gv[NIT], in_v[NIT];
  float old_u[NIT], old_v[NIT];   // PHASE 2 adds to what the prior half left in G
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = t + it * 256;
    {
      const int dl = e / KP, k = e % KP;
      const int d = d0 + dl;
      const bool on = e < KP * FTD && d < D && k < K;
      const size_t i = on ? (size_t)d * K + k : 0;
      in_u[it] = on ? P.p[U_][i] : 0.f;
      in_ue[it] = (PRIOR && on && !hs) ? P.p[UETA_][i] : 1.f;
      in_ua[it] = (PRIOR && on && !hs) ? P.p[UETAA_][i] : 1.f;
      in_ga[it] = (DATA && on) ? gAp[(size_t)d * KP + k] : 0.f;
      in_gv[it] = (DATA && e < KP * FTD && d < D) ? gVp[(size_t)d * KP + k] : 0.f;
      old_u[it] = (PHASE == 2 && on) ? G.p[U_][i] : 0.f;
    }
    {
      const int k = e / FTD, dl = e % FTD;
      const int d = d0 + dl;
      const bool on = e < KP * FTD && k < K && d < D;
      in_v[it] = (PRIOR && on) ? P.p[V_][(size_t)k * D + d] : 0.f;
      old_v[it] = (PHASE == 2 && on) ? G.p[V_][(size_t)k * D + d] : 0.f;
    }
  }
  // The prior-only launch shape of the default model spreads the [.,D] vectors' log-densities over three
  // lane groups of FTD lanes -- group 0: the s0 chain (s0 | s_eta0 | s_eta_a0), group 1: the s1 chain, group 2:
  // w, s_tau, s_tau_a -- three density evaluations in a row per lane instead of nine (they were a quarter of
  // the launch's fp64 chain, all on half a wave); G[s_tau] needs a term from each group: through LDS.
  constexpr bool VSPREAD = PHASE == 1 && !HS;
  static_assert(!VSPREAD || 3 * FTD <= 256, "three lane groups of FTD lanes");
  const int vg = t / FTD, vdl = t % FTD;
  const bool von = VSPREAD && vg < 3 && d0 + vdl < D;
  const int vd = von ? d0 + vdl : 0;
  const size_t vrow = vg == 1 ? (size_t)D : 0;
  float v_a = 1.f, v_b = 1.f, v_c = 1.f, v_st = 1.f, v_keep = 0.f;
  if (VSPREAD && von) {
    v_a = vg == 2 ? P.p[W_][vd] : P.p[S_][vrow + vd];
    v_b = vg == 2 ? P.p[STAU_][vd] : P.p[SETA_][vrow + vd];
    v_c = vg == 2 ? P.p[STAUA_][vd] : P.p[SETAA_][vrow + vd];
    v_st = vg < 2 ? P.p[STAU_][vd] : 1.f;
  }
  // [.,D] vectors of this thread's column (threads t < FTD)
  const bool dcol = t < FTD && d0 + t < D;
  const bool hier = PRIOR && !hs && !VSPREAD;
  const int dme = dcol ? d0 + t : 0;
  const float c_eta = dcol ? eta[dme] : 1.f;
  const float c_s0 = dcol ? P.p[S_][dme] : 1.f, c_s1 = dcol ? P.p[S_][D + dme] : 1.f;
  const float c_w = dcol ? P.p[W_][dme] : 0.f;
  const float c_se0 = (hier && dcol) ? P.p[SETA_][dme] : 1.f, c_se1 = (hier && dcol) ? P.p[SETA_][D + dme] : 1.f;
  const float c_stau = (hier && dcol) ? P.p[STAU_][dme] : 1.f, c_sta = (hier && dcol) ? P.p[STAUA_][dme] : 1.f;
  const float c_sa0 = (hier && dcol) ? P.p[SETAA_][dme] : 1.f, c_sa1 = (hier && dcol) ? P.p[SETAA_][D + dme] : 1.f;
  const float c_gph = (DATA && dcol) ? gph[dme] : 0.f;
  const float old_w = (PHASE == 2 && dcol) ? G.p[W_][dme] : 0.f;
  const float old_s0 = (PHASE == 2 && dcol) ? G.p[S_][dme] : 0.f, old_s1 = (PHASE == 2 && dcol) ? G.p[S_][D + dme] : 0.f;
  // block 0 also closes the data term: lane k of its first wave holds
  // (sum_b z_bk) * (sum_d A'_dk), the closed-form rate sum of the linear decoder
  double rterm = 0.0;
  if (DATA && bx == 0 && t < KP && !lik_bern(logt) && logt != 1)
    rterm = unpack(tail, kDaccHead + t) * prep_sum(dprep, KP, t);
  const int c_bern = (lik_bern(logt) || (logt == 3 && dcol && ctype[dme])) ? 1 : 0;
  // [1,K] vectors (their own prior is block 0's)
  const bool kown = PRIOR && !hs && t < K;
  const float c_ut = (PRIOR && !hs && t < K) ? P.p[UTAU_][t] : 1.f;   // (the data half never touches u_tau: NULL with ABS_HORSESHOE)
  const float c_uta = (kown && bx == 0) ? P.p[UTAUA_][t] : 1.f;

  if (t < KP) {
    // log_transform: the dense kernel already subtracted sum_b E_bd z_b from gV'
    // (mixed, code 3: still needed for the Poisson columns)
    zsum_s[t] = (!DATA || lik_exp(logt) || lik_bern(logt)) ? 0.f : (float)unpack(tail, kDaccHead + t);
    utau_s[t] = hs ? u_tau_scale : c_ut;   // hs: scale = u_tau_scale * decay^k (c_ut is 1 for k >= K)
    dec_s[t] = (float)ipow(decay, t);   // powf is ~1e-6 off at t~60: a systematic part error
    gutau_s[t] = 0.f;
    if (PRIOR) {
      lsc_s[t] = fast_log((double)utau_s[t]) + (double)t * fast_log(decay);
      scd_s[t] = (double)utau_s[t] * ipow(decay, t);
    }
  }
  if (t < FTD) {
    const int d = d0 + t;
    w1s[t] = d < D ? c_s0 / (c_s0 + c_s1) : 0.f;
    etas_[t] = c_eta;
    ietas[t] = lik_exp(logt) ? 1.f : 1.f / c_eta;   // A' = w1*u/eta (linear) or w1*u (log_transform)
    GAs[t] = 0.f;
    // column follows the Bernoulli likelihood: all of them (code 2) or by type (mixed, code 3)
    bern_s[t] = c_bern;
  }
  __syncthreads();

  // ---- [D,K] arrays: u, u_eta, u_eta_a and gA' ---------------------------
  constexpr int LW = KP < 64 ? KP : 64;  // lanes sharing one d inside a wave
  // 256 % KP == 0, so a thread keeps the same k in every iteration: gut_acc
  // sums its d's in a register (LDS float atomics cost ~200 cycles each here)
  float gut_acc = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = t + it * 256;
    if (e >= KP * FTD) break;
    const int dl = e / KP, k = e % KP;
    const int d = d0 + dl;
    float ga_u = 0.f, gut = 0.f;
    if (d < D && k < K) {
      const size_t i = (size_t)d * K + k;
      const float u = in_u[it];
      float du = 0.f;
      if (DATA) {
        const float dA = in_ga[it] * ietas[dl];
        ga_u = u * dA;
        du = w1s[dl] * dA;
      }
      if (PRIOR && hs) {
        double lp, dlp;
        abs_horseshoe((double)u, scd_s[k], lp, dlp);
        part[U_] += lp;
        G.p[U_][i] = du + pw * (float)dlp;
      } else if (PRIOR && !(FIN_ABL & 1)) {
        const float ue = in_ue[it], ua = in_ua[it];
        const float sc = utau_s[k] * dec_s[k];
        // the three log-densities share their fp64 logs (software fp64 log is what
        // this kernel's time goes to): log sig = log ue + log(utau_k dec_k)
        const double Lue = fast_log((double)ue), Lua = fast_log((double)ua);
        const float sig = ue * sc, is = frcp(sig), q = u * is;
        const double qd = (double)u * fast_rcp((double)ue * scd_s[k]);
        part[U_] += kHalfLog2OverPi - (Lue + lsc_s[k]) - 0.5 * qd * qd;
        const float gy = -q * is, gs = (q * q - 1.f) * is;
        G.p[U_][i] = du + pw * gy;
        gut = pw * gs * ue * dec_s[k];
        const float iy = frcp(ue), ia = frcp(ua);
        const float tt = ia * iy * iy;                       // 1/(ua ue^2)
        part[UETA_] += -0.5 * Lua - kLgammaHalf - 2.0 * Lue
                       - fast_rcp((double)ua * (double)ue * (double)ue) + kLog2;
        const float gy2 = -2.f * iy + 2.f * tt * iy, ga2 = -0.5f * ia + tt * ia;
        G.p[UETA_][i] = pw * (gs * sc + gy2);
        part[UETAA_] += -kLgammaHalf - 1.5 * Lua - fast_rcp((double)ua);   // InvGamma(1/2, 1)
        const float ga3 = -1.5f * ia + ia * ia;
        G.p[UETAA_][i] = pw * (ga2 + ga3);
      } else {
        G.p[U_][i] = old_u[it] + du;
      }
    }
    if (DATA) {
      // GA_d = sum_k u*dA : fold over the LW lanes that share d
#pragma unroll
      for (int m = 1; m < LW; m <<= 1) ga_u += __shfl_xor(ga_u, m);
      if constexpr (KP <= 64) {
        if ((k % LW) == 0 && d < D) GAs[dl] = ga_u;   // one writer per dl
      } else {
        // K above 64: a column's KP values span KP / 64 waves (of one iteration: 256 % KP == 0); each leaves its
        // part in its own slot, added in k order below
        if ((k % LW) == 0 && d < D) GAw[k / 64][dl] = ga_u;
      }
    }
    gut_acc += gut;
  }
  // fold gut over the threads that share k: t, t+KP, t+2KP, ... (256/KP of them)
  gred[t] = gut_acc;
  __syncthreads();
  if constexpr (KP > 64) {
    if (DATA && t < FTD) {
      float g = 0.f;
#pragma unroll
      for (int j = 0; j < KP / 64; ++j) g += GAw[j][t];
      GAs[t] = g;                       // read behind the next barriers ([.,D] vectors)
    }
  }
  if (t < KP) {
    float g = 0.f;
    for (int j = t; j < 256; j += KP) g += gred[j];
    gutau_s[t] = g;
  }
  // ---- v / dv through the transpose tile ---------------------------------
  if (DATA) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = t + it * 256;
      if (e >= KP * FTD) break;
      const int dl = e / KP, k = e % KP;
      const int d = d0 + dl;
      tile[k][dl] = (d < D) ? (in_gv[it] - (bern_s[dl] ? 0.f : zsum_s[k])) * etas_[dl] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = t + it * 256;
    if (e >= KP * FTD) break;
    const int k = e / FTD, dl = e % FTD;
    const int d = d0 + dl;
    if (k < K && d < D) {
      const size_t i = (size_t)k * D + d;
      const float dv = DATA ? tile[k][dl] : 0.f;
      if (PRIOR && !(FIN_ABL & 4)) {
        const float v = in_v[it];
        double lp;
        float gy, gs;
        halfnormal(v, 0.1f, lp, gy, gs);
        if (bern_s[dl]) lp -= kLog2;   // Bernoulli column: v ~ Normal(0,.1) (bernoulli.py:187-200)
        part[V_] += (double)lp;
        G.p[V_][i] = dv + pw * gy;
      } else {
        G.p[V_][i] = old_v[it] + dv;
      }
    }
  }
  // ---- [.,D] vectors: w, s, s_eta, s_tau, s_eta_a, s_tau_a ---------------
  __shared__ float xs_s[2][FTD];
  if (VSPREAD) {
    if (von && !(FIN_ABL & 2)) {
      double lp, a_lp, c_lp;
      float gy, gs, a_gy, a_ga, c_ga;
      if (vg < 2) {                    // row vg of s, s_eta, s_eta_a
        halfnormal(v_a, v_b * v_st, lp, gy, gs);
        sqrt_ig(v_b, v_c, a_lp, a_gy, a_ga);
        ig_half(v_c, 1.f, 0.f, c_lp, c_ga);
        part[S_] += lp;
        part[SETA_] += a_lp;
        part[SETAA_] += c_lp;
        G.p[S_][vrow + vd] = pw * gy;
        G.p[SETA_][vrow + vd] = pw * (gs * v_st + a_gy);
        G.p[SETAA_][vrow + vd] = pw * (a_ga + c_ga);
        xs_s[vg][vdl] = gs * v_b;      // d/d s_tau through this row's scale
      } else {                         // w, s_tau, s_tau_a
        halfnormal(v_a, 1.f, lp, gy, gs);
        if (bern_s[vdl]) lp -= kLog2;   // Bernoulli column: w ~ Normal(0,1) (bernoulli.py:201-216)
        sqrt_ig(v_b, v_c, a_lp, a_gy, a_ga);
        const float beta = 1.f / (s_tau_scale * s_tau_scale);
        ig_half(v_c, beta, 0.5f * logf(beta), c_lp, c_ga);
        part[W_] += lp;
        part[STAU_] += a_lp;
        part[STAUA_] += c_lp;
        G.p[W_][vd] = pw * gy;
        G.p[STAUA_][vd] = pw * (a_ga + c_ga);
        v_keep = a_gy;
      }
    }
  } else if (t < FTD && d0 + t < D) {
    const int d = d0 + t;
    const float e = c_eta, s0 = c_s0, s1 = c_s1, w = c_w;
    const float se0 = c_se0, se1 = c_se1, stau = c_stau, sta = c_sta, sa0 = c_sa0, sa1 = c_sa1;
    const float gph_d = c_gph;
    float dw = 0.f, ds0 = 0.f, ds1 = 0.f;
    if (DATA) {
      const float T = s0 + s1, iT2 = 1.f / (T * T);
      const float w2 = s1 / T;
      // Poisson: sum_b x/r - B; Bernoulli: sum_nnz x - sum_b sigmoid (dense kernel already applied)
      const float dphi = bern_s[t] ? gph_d : gph_d - (float)Bglob;
      const float GA = GAs[t], Gphi = e * w * dphi;
      dw = e * w2 * dphi;
      ds0 = (GA - Gphi) * s1 * iT2;
      ds1 = (Gphi - GA) * s0 * iT2;
    }
    if (PRIOR && !(FIN_ABL & 2)) {
      double lp;
      float gy, gs;
      halfnormal(w, 1.f, lp, gy, gs);
      if (bern_s[t]) lp -= kLog2;     // Bernoulli column: w ~ Normal(0,1) (bernoulli.py:201-216)
      part[W_] += (double)lp;
      G.p[W_][d] = dw + pw * gy;
      if (hs) {
        double l0, g0, l1, g1;
        abs_horseshoe((double)s0, (double)s_tau_scale, l0, g0);
        abs_horseshoe((double)s1, (double)s_tau_scale, l1, g1);
        part[S_] += l0 + l1;
        G.p[S_][d] = ds0 + pw * (float)g0;
        G.p[S_][D + d] = ds1 + pw * (float)g1;
      } else {
      double lp0, lp1;
      float gy0, gs0, gy1, gs1;
      halfnormal(s0, se0 * stau, lp0, gy0, gs0);
      halfnormal(s1, se1 * stau, lp1, gy1, gs1);
      part[S_] += (double)lp0 + (double)lp1;
      G.p[S_][d] = ds0 + pw * gy0;
      G.p[S_][D + d] = ds1 + pw * gy1;
      double a_lp, b_lp, c_lp;
      float a_gy, a_ga, b_gy, b_ga, c_ga;
      sqrt_ig(se0, sa0, a_lp, a_gy, a_ga);
      sqrt_ig(se1, sa1, b_lp, b_gy, b_ga);
      part[SETA_] += (double)a_lp + (double)b_lp;
      G.p[SETA_][d] = pw * (gs0 * stau + a_gy);
      G.p[SETA_][D + d] = pw * (gs1 * stau + b_gy);
      ig_half(sa0, 1.f, 0.f, c_lp, c_ga);
      part[SETAA_] += (double)c_lp;
      G.p[SETAA_][d] = pw * (a_ga + c_ga);
      ig_half(sa1, 1.f, 0.f, c_lp, c_ga);
      part[SETAA_] += (double)c_lp;
      G.p[SETAA_][D + d] = pw * (b_ga + c_ga);
      sqrt_ig(stau, sta, a_lp, a_gy, a_ga);
      part[STAU_] += (double)a_lp;
      G.p[STAU_][d] = pw * (gs0 * se0 + gs1 * se1 + a_gy);
      const float beta = 1.f / (s_tau_scale * s_tau_scale);
      ig_half(sta, beta, 0.5f * logf(beta), c_lp, c_ga);
      part[STAUA_] += (double)c_lp;
      G.p[STAUA_][d] = pw * (a_ga + c_ga);
      }
    } else {
      G.p[W_][d] = old_w + dw;
      G.p[S_][d] = old_s0 + ds0;
      G.p[S_][D + d] = old_s1 + ds1;
    }
  }
  __syncthreads();
  if (VSPREAD && von && vg == 2 && !(FIN_ABL & 2)) G.p[STAU_][vd] = pw * (xs_s[0][vdl] + xs_s[1][vdl] + v_keep);
  if (PRIOR) {
    // ---- [1,K] vectors: u_tau, u_tau_a (block 0 adds their own prior) ------
    if (t < K && !hs) {
      float g = gutau_s[t];
      if (bx == 0) {
        const float ut = c_ut, uta = c_uta;
        double lp, lp2;
        float gy, ga, ga2;
        sqrt_ig(ut, uta, lp, gy, ga);
        part[UTAU_] += (double)lp;
        g += pw * gy;
        const float beta = 1.f / (u_tau_scale * u_tau_scale);
        ig_half(uta, beta, 0.5f * logf(beta), lp2, ga2);
        part[UTAUA_] += (double)lp2;
        G.p[UTAUA_][t] = pw * (ga + ga2);
      }
      putau[t] = g;
    }
    // ---- energy parts ------------------------------------------------------
    __shared__ double pred[12][4];
    const int wid = t >> 6, lane = t & 63;
    if (FIN_ABL & 8) {
#pragma unroll
      for (int i = 0; i < 12; ++i)
        if (lane == 0) pred[i][wid] = part[i];
    } else {
      parts_wave_sums(part, lane, wid, pred);
    }
    __syncthreads();
    if (t < 12) ppart[t] = pred[t][0] + pred[t][1] + pred[t][2] + pred[t][3];
  }
  if constexpr (KP > 64) {               // rterm lives in lanes t < KP of KP / 64 waves: through LDS to the first
    if (DATA && bx == 0) {               // (block-uniform)
      if (t < KP) rterm_s[t] = rterm;
      __syncthreads();
      if (t < 64) {
        rterm = 0.0;
#pragma unroll
        for (int j = 0; j < KP / 64; ++j) rterm += rterm_s[j * 64 + t];
      }
    }
  }
  if (DATA && bx == 0 && t < 64) {
    const double rsum = wave_sum(rterm);
    if (t == 0) {
    const double llx = unpack(tail, 0), zsq = unpack(tail, 1);
    // sum of the rate over ALL cells: closed form (linear) or the dense exp sum
    double sum_r = Bglob * prep_sum(dprep, KP, KP);
    if (lik_bern(logt))
      sum_r = unpack(tail, 3);                 // sum over all cells of softplus(logit)
    else if (logt == 1)
      sum_r += unpack(tail, 3) - Bglob * (double)D;
    else {
      sum_r += rsum;
      if (logt == 3) sum_r += unpack(tail, 3);   // mixed: + softplus over the Bernoulli columns
    }
    parts[13] = llx - (lik_bern(logt) ? 0.0 : lgamma_sum) - sum_r;      // single writer
    parts[12] = Bglob * (double)K * kHalfLog2OverPi - 0.5 * zsq;
    if (nnf_out) {
      nnf_out[0] = unpack(tail, 2);
      nnf_out[nby] = unpack(tail, 4);     // [S + s]: saturated cells (log_transform)
    }
    }
  }
}

// host side: the kernel-argument struct from the launch interface of kernels.h
inline FinishK make_finish_k(const FinishArgs& a) {
  FinishK k{};
  k.D = a.D;
  k.K = a.K;
  k.Bglob = (double)a.B_global;
  k.lgamma_sum = a.lgamma_sum;
  k.u_tau_scale = (float)a.u_tau_scale;
  k.s_tau_scale = (float)a.s_tau_scale;
  k.decay = a.decay;
  k.pw = (float)a.prior_weight;
  k.acc = a.acc;
  k.dprep = a.dprep;
  for (int i = 0; i < 12; ++i) {
    k.P.p[i] = a.params[i];
    k.G.p[i] = a.grads[i];
    k.VS.v[i] = a.vstride[i];
  }
  k.eta = a.eta;
  k.parts = a.parts;
  k.nnf_out = a.n_nonfinite;
  k.logt = a.logt;
  k.ctype = a.ctype;
  k.Dh = a.Dh > 0 ? a.Dh : a.D;
  k.acc_stride = a.acc_stride;
  k.hs = a.abs_horseshoe;
  k.ppart = a.ppart;
  k.putau = a.putau;
  return k;
}
inline int finish_blocks(int D) { return (D + FTD - 1) / FTD; }

}  // namespace spmf
