// gx_math.h -- scalar device math of the Genrich hot path (gfx950).
//
// Pileup values are exact integers in units of 1/120 ("V120"); the float the reference
// would hold is re-materialised by gx_getval exactly as getVal does (Genrich.c:1902-1907).
// p-values are double math rounded once to float (calcPval, Genrich.c:1628-1653).  The device
// uses OCML's double log/exp/log1p, the reference the host's libm; the two agree to a few ulp of
// the DOUBLE, so the float differs only when the double lies next to a float rounding boundary.
// round_checked() detects exactly that (distance to the boundary below RISK_B of the value, far
// above any libm difference): such a value is "risky", goes on a list, and is re-evaluated by the
// very same routines compiled for the host (every function here is __host__ __device__), i.e. with
// the libm the reference itself would call on this machine.  A handful per run; all others are
// bit-identical by the error bound.  Compile with -ffp-contract=off: the reference is built
// without FMA contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include <math.h>

#define GX_UNIT 120
#define GX_SKIPF (-1.0f)
#define GX_HD __host__ __device__

namespace gx {

// residue mod 120 -> packed (eighths | sixths<<4 | tenths<<8) with 15e+20s+12t == r (mod 120),
// e<8, s<3, t<5 (a bijection; frac bit layout of the reference: Genrich.c:2299-2305).
// mod 3: 15e+20s+12t = 2s ; mod 5: = 2t ; mod 8: = 7e + 4(s+t).
GX_HD __forceinline__ uint32_t est_of_residue(int r) {
  uint32_t s = (uint32_t)(r * 2) % 3u;
  uint32_t t = (uint32_t)(r * 3) % 5u;
  uint32_t e = (7u * ((uint32_t)r + 4u * (s + t))) & 7u;
  return e | (s << 4) | (t << 8);
}

// getVal (1902-1907) of the canonical (cov, e, s, t) state of the exact sum v/120.
// *neg is set when the canonical integer part is negative (updateVal's ERRPILE, 1921/1969).
GX_HD __forceinline__ float getval(int32_t v, bool* neg) {
  int32_t q = v / GX_UNIT, r = v - q * GX_UNIT;
  if (r < 0) { r += GX_UNIT; q -= 1; }
  if (r == 0) { *neg = q < 0; return (float)q; }
  uint32_t est = est_of_residue(r);
  int e = est & 15, s = (est >> 4) & 15, t = est >> 8;
  int32_t cov = q - (15 * e + 20 * s + 12 * t - r) / GX_UNIT;
  *neg = cov < 0;
  return (float)cov + ((float)e / 8.0f) + ((float)s / 6.0f) + ((float)t / 10.0f);
}

// updateVal's ERRPILE test alone (1921 / 1969): is the integer part of the canonical (cov, e, s, t) state of
// the exact pileup v / 120 negative?  (It can be for a small positive fractional pileup: 59/120 = -1 + 7/8 + 2/6 + 2/10 ... .)
GX_HD __forceinline__ bool getval_neg(int32_t v) {
  int32_t q = v / GX_UNIT, r = v - q * GX_UNIT;
  if (r < 0) { r += GX_UNIT; q -= 1; }
  if (r == 0) return q < 0;
  const uint32_t est = est_of_residue(r);
  const int e = est & 15, s = (est >> 4) & 15, t = est >> 8;
  return q - (15 * e + 20 * s + 12 * t - r) / GX_UNIT < 0;
}

// ---- the one rounding of a p-value: double -> float, with the "risky" test ----------------------
// Device and host evaluate the same IEEE operations around different libm calls (<= ~1.5 ulp of the
// double apart per call); through calcPval / pchisq that propagates to a relative difference in the double
// result that grows towards the lower tail, where a relative difference dz in z = (log expt - meanlog) / sdlog
// becomes z^2 dz in the result (z^2 / 2 <= 103 for any p that is not zero as a float).  A result whose distance
// to the nearest float rounding boundary (the midpoint of two neighbouring floats) is below RISK_B of its
// value is not rounded here but flagged, and the host evaluates it.  That is sound as long as the two doubles
// never differ by RISK_B or more -- SWEPT over the reachable domain in round 5 (tools/sweep_risk_margin.py on
// MI355X against glibc 2.35, profiles/r05_risk_margin.txt): every exact pileup V in [0, 2^18) x 307 values of
// lambda in [1e-4, 2000] (80 M pairs) and the 256 x 256 table of whole pileups x 26 factors x 7 lambda (12 M):
// max |device - host| / |host| = 7.4e-13 = 0.20 x RISK_B, reached for lambda > 1500 (pileups far below a control in the
// thousands: p ~ 1e-31); 0.03 x RISK_B for every lambda <= 240, 0.055 x RISK_B over the pair tables; 0 floats differ
// after the re-evaluation.  (Round 2's sample of 400,000 random pairs had seen 7e-14 at the 99.99th percentile and
// claimed a factor of 30: true below lambda ~ 240, a factor of 5 over everything.)
// RISK_B = 2^-38 = 3.6e-12.  About 2 * RISK_B / 2^-24 = 1.2e-4 of all values are flagged.
#define GX_RISK_B 0x1p-38

GX_HD __forceinline__ float bits_float(uint32_t u) { union { uint32_t u; float f; } x; x.u = u; return x.f; }
GX_HD __forceinline__ uint32_t float_bits(float f) { union { uint32_t u; float f; } x; x.f = f; return x.u; }

GX_HD inline float round_checked(double pv, bool* risky) {  // |pv| <= FLT_MAX
  const float f = (float)pv;
  const double a = fabs(pv), fd = fabs((double)f);  // rounding is symmetric: work on magnitudes
  const double err = fabs(a - fd);                  // exact
  const uint32_t b = float_bits(f) & 0x7FFFFFFFu;
  // the neighbouring float on pv's side (beyond zero lies the smallest subnormal of the other sign)
  const double nb = a >= fd ? (double)bits_float(b + 1u) : (b ? (double)bits_float(b - 1u) : -1.401298464324817e-45);
  const double half = 0.5 * fabs(nb - fd);  // (an infinite neighbour gives an infinite distance)
  if (half - err < a * GX_RISK_B) *risky = true;
  return f;
}

// ---- log-normal -log10 p : calcPval (1628-1653), plnorm (1617), pnorm (1509-1607) ----

GX_HD __forceinline__ double do_del(double y, double temp, bool lower) {  // 1497-1503
  double xsq = trunc(y * 16) / 16;
  double del = (y - xsq) * (y + xsq);
  if (lower) return log1p(-exp((-xsq * xsq - del) / 2.0) * temp);
  return (-xsq * xsq - del) / 2.0 + log(temp);
}

GX_HD inline double pnorm_upper_log(double x) {
  const double a0 = 2.2352520354606839287, a1 = 161.02823106855587881,
               a2 = 1067.6894854603709582, a3 = 18154.981253343561249,
               a4 = 0.065682337918207449113;
  const double b0 = 47.20258190468824187, b1 = 976.09855173777669322,
               b2 = 10260.932208618978205, b3 = 45507.789335026729956;
  const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                       597.27027639480026226,  2494.5375852903726711, 6848.1904505362823326,
                       11602.651437647350124,  9842.7148383839780218, 1.0765576773720192317e-8};
  const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                       6485.558298266760755,  18615.571640885098091, 34900.952721145977266,
                       38912.003286093271411, 19685.429676859990727};
  const double p[6] = {0.21589853405795699,   0.1274011611602473639,   0.022235277870649807,
                       0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
  const double q[5] = {1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
                       0.00378239633202758244, 7.29751555083966205e-5};
  double y = fabs(x), num, den, t;
  if (y <= 0.67448975) {
    if (y > DBL_EPSILON * 0.5) {
      double xsq = x * x;
      num = a4 * xsq;
      den = xsq;
      num = (num + a0) * xsq; den = (den + b0) * xsq;
      num = (num + a1) * xsq; den = (den + b1) * xsq;
      num = (num + a2) * xsq; den = (den + b2) * xsq;
      t = x * (num + a3) / (den + b3);
    } else
      t = x * a3 / b3;
    return log(0.5 - t);
  }
  if (y <= sqrt(32.0)) {
    num = c[8] * y;
    den = y;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      num = (num + c[i]) * y;
      den = (den + d[i]) * y;
    }
    t = (num + c[7]) / (den + d[7]);
    return do_del(y, t, x <= 0.0);
  }
  if (y < 1e170) {
    double xsq = 1.0 / (x * x);
    num = p[5] * xsq;
    den = xsq;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      num = (num + p[i]) * xsq;
      den = (den + q[i]) * xsq;
    }
    t = xsq * (num + p[4]) / (den + q[4]);
    t = (1 / sqrt(2 * 3.14159265358979323846) - t) / y;
    return do_del(x, t, x <= 0.0);
  }
  return -0.0;
}

// meanlog / sdlog of the null for control value mu (1637-1648)
GX_HD __forceinline__ void lnorm_params(float ctrl, double* meanlog, double* sdlog) {
  double mu = ctrl;
  if (mu > 7.0) {
    double sd = 10.0 * log10(mu);
    mu *= mu;
    sd *= sd;
    *meanlog = log(mu / sqrt(sd + mu));
    *sdlog = sqrt(log1p(sd / mu));
  } else {
    *meanlog = log(mu) - 0.445999019652555; /* LOGSQRT, Genrich.h:52 */
    *sdlog = 0.944456478248262;             /* SQRTLOG, Genrich.h:53 */
  }
}

// the double the reference rounds (1651-1652), from log(expt) and the control's parameters
GX_HD __forceinline__ double pval_double(float expt, double logExpt, double meanlog, double sdlog) {
  if (sdlog == 0.0) return (double)expt < meanlog ? 0.0 : (double)FLT_MAX;
  return -pnorm_upper_log((logExpt - meanlog) / sdlog) / 2.30258509299404568402;
}

GX_HD __forceinline__ float pval_round(double pv, bool* risky) {
  // a zero is +0: the reference's -pnorm(..) / M_LN10 never yields -0 (an underflowing tail is
  // log1p(-0.0) = -0.0 with glibc, negated), while the device's log1p(-0.0) is +0.0
  if (pv == 0.0) return 0.0f;
  return pv > (double)FLT_MAX ? FLT_MAX : round_checked(pv, risky);
}

GX_HD __forceinline__ float pval_given(float expt, double meanlog, double sdlog, bool* risky) {
  return pval_round(pval_double(expt, log((double)expt), meanlog, sdlog), risky);
}

GX_HD inline float calc_pval(float expt, float ctrl, bool* risky) {
  if (ctrl == GX_SKIPF) return GX_SKIPF;
  if (ctrl == 0.0f) return expt == 0.0f ? 0.0f : FLT_MAX;
  if (expt == 0.0f) return 0.0f;
  double ml, sl;
  lnorm_params(ctrl, &ml, &sl);
  return pval_given(expt, ml, sl, risky);
}

// the pileup floats of the two samples of a p-interval from their exact values (1/120 units; V_MARK
// inside an excluded -E region)
#define GX_V_MARK ((int32_t)0x80000000)
GX_HD __forceinline__ float expt_val(int v, bool* neg) {
  if (v == GX_V_MARK) { *neg = false; return 0.0f; }  // excluded region: 2248 / 2273
  return getval(v, neg);
}

GX_HD __forceinline__ float ctrl_net(int v, float factor, float lambda, bool* neg) {
  if (v == GX_V_MARK) { *neg = false; return GX_SKIPF; }  // excluded region: 2124 / 2141
  float val = factor * getval(v, neg);  // 2107 / 2118: float product
  return val > lambda ? val : lambda;   // MAX(val, lambda)
}

// ---- chi-squared upper tail, log scale: pchisq (555-559) and helpers (407-545) ----

GX_HD __forceinline__ double log1_exp(double x) {  // R_Log1_Exp, 407
  return x > -0.693147180559945309417 ? log(-expm1(x)) : log1p(-exp(x));
}

GX_HD inline double bd0(double x, double np) {  // 412-430
  if (fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np);
    double s = (x - np) * v;
    if (fabs(s) < DBL_MIN) return s;
    double ej = 2 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      double s1 = s + ej / ((j << 1) + 1);
      if (s1 == s) return s1;
      s = s1;
    }
  }
  return x * log(x / np) + np - x;
}

GX_HD inline double stirlerr(double n) {  // 436-469
  const double sferr[16] = {0.0,
                            0.0810614667953272582196702,
                            0.0413406959554092940938221,
                            0.02767792568499833914878929,
                            0.02079067210376509311152277,
                            0.01664469118982119216319487,
                            0.01387612882307074799874573,
                            0.01189670994589177009505572,
                            0.010411265261972096497478567,
                            0.009255462182712732917728637,
                            0.008330563433362871256469318,
                            0.007573675487951840794972024,
                            0.006942840107209529865664152,
                            0.006408994188004207068439631,
                            0.005951370112758847735624416,
                            0.005554733551962801371038690};
  const double S0 = 1.0 / 12, S1 = 1.0 / 360, S2 = 1.0 / 1260, S3 = 1.0 / 1680, S4 = 1.0 / 1188;
  double nn = n * n;
  if (n > 80.0) return (S0 - (S1 - S2 / nn) / nn) / n;
  if (n > 35.0) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
  if (n > 15.0) return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
  return sferr[(int)n];
}

GX_HD inline double dpois_log(double x, double lambda) {  // 474-477
  return -0.5 * log(2.0 * 3.14159265358979323846 * x) - stirlerr(x) - bd0(x, lambda);
}

GX_HD inline double pgamma_upper_log(double x, double alph) {  // 528-545
  if (x < 1) {  // pgamma_smallx 509-522
    double sum = 0.0, c = alph, n = 0.0, term;
    do {
      n++;
      c *= -x / n;
      term = c / (alph + n);
      sum += term;
    } while (fabs(term) > DBL_EPSILON * fabs(sum));
    double lf2 = alph * log(x) - lgamma(alph + 1);
    return log1_exp(log1p(sum) + lf2);
  }
  if (x <= alph - 1) {  // pd_upper_series 482-491
    double a = alph, term = x / a, sum = term;
    do {
      a++;
      term *= x / a;
      sum += term;
    } while (term > sum * DBL_EPSILON);
    return log1_exp(log(sum) + dpois_log(alph - 1, x));
  }
  double y = alph - 1, term = 1, sum = 0;  // pd_lower_series 496-504
  while (y >= 1 && term > sum * DBL_EPSILON) {
    term *= y / x;
    sum += term;
    y--;
  }
  return log1p(sum) + dpois_log(alph - 1, x);
}

// multPval's tail (577-582): sum of -log10 p over df/2 replicates -> combined -log10 p
GX_HD inline double fisher_double(double sum, int df) {
  return -pgamma_upper_log((2.0 * sum / 0.434294481903251827651) / 2.0, df / 2.0) / 2.30258509299404568402;
}

// The same tail in closed form (round 6).  df = 2 k is even -- two per replicate -- and the chi-squared upper tail of an even df is a
// finite sum: with y = x / 2 = sum * ln 10,
//     Q(x; 2k) = e^-y  S_k(y),  S_k(y) = sum_{j<k} y^j / j!        so   -log10 Q = sum - log10 S_k(y).
// Up to eight replicates (k <= 8: k_mergeN_w) S_k - 1 = y (1 + y/2 (1 + y/3 (...))) by Horner and ONE log1p; beyond, from the
// largest term down, y^(k-1) / (k-1)! (1 + (k-1)/y + ...), the reference's own pd_lower_series (496-504) with the Poisson density
// written out (no overflow).  Where sum and log10 S_k nearly cancel -- y far below k, Q next to 1 -- the lower tail
// P = e^-y y^k / k! (1 + y/(k+1) + ...) is small and -log10 (1 - P) = -log1p(-P) / ln 10 keeps its digits: below FISHER_Y0[k]
// (the y from which the cancellation costs less than a factor of 200; k - 1 beyond eight replicates).
// No data-dependent series of pgamma, no branch per regime of bd0 / stirlerr -- which is what lets the merge kernels evaluate every
// merged interval on the spot instead of probing a device-wide table of results (4 GB of 128-byte lines for 16-byte probes at
// hg38 x 3 replicates).  It is NOT the reference's sequence of operations, so it is only ever rounded through round_checked:
// against the reference's algorithm with the host's libm (fisher_double) the two doubles differ by at most 6.3e-14 = 0.017 x
// RISK_B over 11 M sums in [1e-6, 1e38] x df 4 .. 64 (tests/test_abi.py repeats a sweep on the CPU, the GPU suite on the device),
// and a value next to a float rounding boundary goes to the host, which evaluates fisher_double.
// log1p for the closed form below: the algorithm of fdlibm's s_log1p.c (argument reduction to [sqrt(2)/2, sqrt(2)), the correction
// term of 1 + x, a degree-14 polynomial in s = f / (2 + f); error below 1 ulp -- checked against glibc's over [-0.5, 1e300]) spelled
// out here because the device library's log1p costs the merge kernel more than everything else in it (hg38 x 3 replicates: 2.0 of
// k_mergeN_w's 3.9 ms were the combination's arithmetic).  The same operations on the host and on the device (IEEE divisions, no
// contraction), so the host-side sweep of the closed form against the reference's algorithm holds for both.
GX_HD inline double fast_log1p(double x) {  // x > -1, finite
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double f, c = 0.0;
  int k = 0;
  if (x > -0.2928932188134524 && x < 0.41421356237309503) {  // sqrt(2)/2 <= 1 + x < sqrt(2): no rescaling, f = x exactly
    const double ax = fabs(x);
    if (ax < 0x1p-29) return ax < 0x1p-54 ? x : x - x * x * 0.5;
    f = x;
  } else {
    union { double d; uint64_t u; } v;
    if (x < 0x1p53) {
      v.d = 1.0 + x;
      k = (int)((v.u >> 52) & 0x7FFu) - 1023;
      c = (k > 0 ? 1.0 - (v.d - x) : x - (v.d - 1.0)) / v.d;  // what the rounding of 1 + x lost
    } else {
      v.d = x;
      k = (int)((v.u >> 52) & 0x7FFu) - 1023;
    }
    const uint64_t hu = (v.u >> 32) & 0x000FFFFFu;
    if (hu < 0x6A09Eu) {
      v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;   // u in [1, sqrt 2)
    } else {
      k += 1;
      v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FE0000000000000ull;   // u / 2 in [sqrt(2)/2, 1)
    }
    f = v.d - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f), z = s * s;
  const double R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  const double dk = (double)k;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + (dk * ln2_lo + c))) - f);
}

// 1 / m for m < 64, as the division gives them (literals folded by the compiler: the same doubles on the host and on the device):
// the divisions of the closed form's Horner scheme and of its lower-tail series were a dozen instructions each on the device
GX_HD inline double fisher_rcp(int m) {
  static const double RK[64] = {
      0.0,        1.0 / 1.0,  1.0 / 2.0,  1.0 / 3.0,  1.0 / 4.0,  1.0 / 5.0,  1.0 / 6.0,  1.0 / 7.0,  1.0 / 8.0,  1.0 / 9.0,  1.0 / 10.0,
      1.0 / 11.0, 1.0 / 12.0, 1.0 / 13.0, 1.0 / 14.0, 1.0 / 15.0, 1.0 / 16.0, 1.0 / 17.0, 1.0 / 18.0, 1.0 / 19.0, 1.0 / 20.0, 1.0 / 21.0,
      1.0 / 22.0, 1.0 / 23.0, 1.0 / 24.0, 1.0 / 25.0, 1.0 / 26.0, 1.0 / 27.0, 1.0 / 28.0, 1.0 / 29.0, 1.0 / 30.0, 1.0 / 31.0, 1.0 / 32.0,
      1.0 / 33.0, 1.0 / 34.0, 1.0 / 35.0, 1.0 / 36.0, 1.0 / 37.0, 1.0 / 38.0, 1.0 / 39.0, 1.0 / 40.0, 1.0 / 41.0, 1.0 / 42.0, 1.0 / 43.0,
      1.0 / 44.0, 1.0 / 45.0, 1.0 / 46.0, 1.0 / 47.0, 1.0 / 48.0, 1.0 / 49.0, 1.0 / 50.0, 1.0 / 51.0, 1.0 / 52.0, 1.0 / 53.0, 1.0 / 54.0,
      1.0 / 55.0, 1.0 / 56.0, 1.0 / 57.0, 1.0 / 58.0, 1.0 / 59.0, 1.0 / 60.0, 1.0 / 61.0, 1.0 / 62.0, 1.0 / 63.0};
  return m < 64 ? RK[m] : 1.0 / (double)m;
}

GX_HD inline double fisher_fast_double(double sum, int df) {
  const double FISHER_Y0[9] = {0, 0, 0.0102, 0.1857, 0.5772, 1.1132, 1.7419, 2.4190, 3.1646};
  const double LN10 = 2.30258509299404568402, INV_LN10 = 0.43429448190325182765;
  const int k = df >> 1;
  const double y = sum * LN10;
  if (y >= (k <= 8 ? FISHER_Y0[k] : (double)(k - 1))) {
    if (k <= 8) {
      double S = 1.0;
      for (int j = k - 1; j >= 2; j--) S = 1.0 + S * (y * fisher_rcp(j));
      return sum - fast_log1p(y * S) * INV_LN10;
    }
    const double iy = 1.0 / y;
    double term = 1.0, T = 1.0, fact = 1.0;
    for (int j = k - 1; j >= 1; j--) {
      term *= (double)j * iy;
      T += term;
    }
    for (int j = 2; j <= k - 1; j++) fact *= (double)j;
    return sum - (double)(k - 1) * log10(y) + log10(fact) - log10(T);
  }
  double term = 1.0, s = 1.0, fact = 1.0, yk = 1.0;
  for (int j = 1; j < 400; j += 4) {  // (four terms a turn: their reciprocals are asked for together; a term too many changes nothing)
    const double r0 = fisher_rcp(k + j), r1 = fisher_rcp(k + j + 1), r2 = fisher_rcp(k + j + 2), r3 = fisher_rcp(k + j + 3);
    term *= y * r0;
    s += term;
    term *= y * r1;
    s += term;
    term *= y * r2;
    s += term;
    term *= y * r3;
    s += term;
    if (term < s * 1e-18) break;
  }
  for (int j = 2; j <= k; j++) fact *= (double)j;
  for (int j = 0; j < k; j++) yk *= y;
  const double P = exp(-y) * yk / fact * s;
  return -fast_log1p(-P) * INV_LN10;
}

// multPval (567-583) through the closed form; df <= 64 (32 replicates)
GX_HD inline float fisher_fast(double sum, int df, bool* risky) {
  if (df == 0) return GX_SKIPF;
  if (df == 2 || sum == 0.0) return (float)sum;
  return pval_round(fisher_fast_double(sum, df), risky);
}

GX_HD inline float fisher_combine(double sum, int df, bool* risky) {
  if (df == 0) return GX_SKIPF;
  if (df == 2 || sum == 0.0) return (float)sum;
  return pval_round(fisher_double(sum, df), risky);
}

}  // namespace gx
