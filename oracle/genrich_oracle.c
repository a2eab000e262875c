/*
 * genrich_oracle.c -- CPU restatement of the Genrich v0.6.2 hot path
 *                     (events -> pileup -> p -> q -> peaks).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under genrich_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Every function names the reference code it restates (file:line into
 * /root/reference/Genrich.c unless noted).  It is written from the semantic
 * description in SURVEY.md Appendix A, not transliterated: the reference's
 * 8-bit "tenths/sixths/eighths" fraction codec is restated as exact integer
 * arithmetic in units of 1/120 (every weight 1/count, count in
 * {1,2,3,4,5,6,8,10}, is a multiple of 1/120), with the float value
 * re-materialised exactly as getVal() does.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle.py,
 * run in the build container) against the unmodified reference compiled into
 * oracle/_ref/ and against the committed golden fixtures under tests/golden/
 * that the reference produced (tests/golden/make_golden.py), plus the seven
 * calcPval known answers of README.md:243-249.
 *
 * Build: gcc -O2 -std=gnu99 -ffp-contract=off (the reference is built for
 * baseline x86-64, i.e. without FMA contraction).
 */
#include <float.h>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/genrich_amd.h"

#define LOGSQRT 0.445999019652555 /* Genrich.h:52  log(sqrt(2.44)) */
#define SQRTLOG 0.944456478248262 /* Genrich.h:53  sqrt(log(2.44)) */
#define UNIT 120                  /* pileup unit: 1/120 */

typedef struct {
  uint32_t* end;
  float* cov;
  uint32_t len, mem;
} Rle; /* Pileup, Genrich.h:173-176 */

typedef struct {
  uint32_t len;
  bool skip, save;
  uint32_t* bed;
  int bedLen;
  int32_t* diff; /* Diff (Genrich.h:178-181) restated: per-base delta in 1/120 units */
  Rle expt, ctrl;
  Rle** pval; /* one per replicate (+ combined), NULL when the chromosome was not saved */
  int sample;
  float* qval; /* Chrom.qval->cov */
} OChrom;       /* Chrom, Genrich.h:183-201 */

typedef struct gxo_ctx {
  gx_params par;
  OChrom* chrom;
  int chromLen;
  int sample;      /* replicates finished (runProgram's `sample`) */
  int phase;       /* 0 idle, 1 treatment open, 2 treatment done, 3 control open, 4 ctrl done */
  double fragLen;  /* of the current replicate */
  /* The same sum WITHOUT the reference's rounding, for the tests (not part of the restatement): every product of
   * 2246 is a float, hence a multiple of 2^-27 here (val >= 1/10), so integer part and fraction * 2^27 add up exactly
   * in two int64; fragInexact counts the additions of 2246 / 2271 that rounded (each by at most half a unit in the
   * last place of the sum at that time).  A sum that never rounded IS the exact one. */
  long long fragHi, fragLo;
  uint64_t fragInexact;
  gx_peak* peaks;
  size_t nPeaks, memPeaks;
  uint64_t genomeLenUsed, peakBP;
  int final;       /* index of the p-value array peaks were called on */
  char err[256];
  uint64_t skippedOverflow;
} gxo_ctx;

/* ---- fraction codec --------------------------------------------------- */

static uint8_t g_est[UNIT][3]; /* residue mod 120 -> (eighths, sixths, tenths) */
static bool g_estInit = false;

/* The reference state is (cov, e<8, s<3, t<5) with value cov + e/8 + s/6 + t/10
 * (frac bit layout, Genrich.c:2299-2305).  15e+20s+12t mod 120 is a bijection of
 * (e,s,t) by CRT, so the state is a pure function of the exact sum. */
static void initEst(void) {
  if (g_estInit) return;
  for (int e = 0; e < 8; e++)
    for (int s = 0; s < 3; s++)
      for (int t = 0; t < 5; t++) {
        int r = (15 * e + 20 * s + 12 * t) % UNIT;
        g_estInit = true;
        g_est[r][0] = (uint8_t)e;
        g_est[r][1] = (uint8_t)s;
        g_est[r][2] = (uint8_t)t;
      }
}

/* canonical integer part of an exact sum v (units of 1/120) */
static int32_t canonCov(int64_t v, int* e, int* s, int* t) {
  int r = (int)(((v % UNIT) + UNIT) % UNIT);
  *e = g_est[r][0];
  *s = g_est[r][1];
  *t = g_est[r][2];
  return (int32_t)((v - (15 * *e + 20 * *s + 12 * *t)) / UNIT);
}

/* getVal (1902-1907) of the canonical state of v; *neg set when cov < 0
 * (updateVal's ERRPILE check, 1921 / 1969). */
float gxo_getval(int64_t v, int* neg) {
  initEst();
  int e, s, t;
  int32_t cov = canonCov(v, &e, &s, &t);
  if (neg) *neg = cov < 0;
  return (float)cov + (e / 8.0f) + (s / 6.0f) + (t / 10.0f);
}

/* ---- log-normal p-value (1490-1653) ----------------------------------- */

/* do_del (1497-1503) */
static double doDel(double y, double temp, bool lower) {
  double xsq = trunc(y * 16) / 16;
  double del = (y - xsq) * (y + xsq);
  if (lower) return log1p(-exp((-xsq * xsq - del) / 2.0) * temp);
  return (-xsq * xsq - del) / 2.0 + log(temp);
}

/* pnorm (1509-1607): log of the upper tail; Cody's rational approximations with
 * the coefficients of R-3.5.0 pnorm.c */
static double pnormUpperLog(double x) {
  static const double a[5] = {2.2352520354606839287, 161.02823106855587881,
                              1067.6894854603709582, 18154.981253343561249,
                              0.065682337918207449113};
  static const double b[4] = {47.20258190468824187, 976.09855173777669322,
                              10260.932208618978205, 45507.789335026729956};
  static const double c[9] = {0.39894151208813466764, 8.8831497943883759412,
                              93.506656132177855979,  597.27027639480026226,
                              2494.5375852903726711,  6848.1904505362823326,
                              11602.651437647350124,  9842.7148383839780218,
                              1.0765576773720192317e-8};
  static const double d[8] = {22.266688044328115691, 235.38790178262499861,
                              1519.377599407554805,  6485.558298266760755,
                              18615.571640885098091, 34900.952721145977266,
                              38912.003286093271411, 19685.429676859990727};
  static const double p[6] = {0.21589853405795699,    0.1274011611602473639,
                              0.022235277870649807,   0.001421619193227893466,
                              2.9112874951168792e-5,  0.02307344176494017303};
  static const double q[5] = {1.28426009614491121, 0.468238212480865118,
                              0.0659881378689285515, 0.00378239633202758244,
                              7.29751555083966205e-5};
  double y = fabs(x), num, den, t;
  if (y <= 0.67448975) {
    if (y > DBL_EPSILON * 0.5) {
      double xsq = x * x;
      num = a[4] * xsq;
      den = xsq;
      for (int i = 0; i < 3; i++) {
        num = (num + a[i]) * xsq;
        den = (den + b[i]) * xsq;
      }
      t = x * (num + a[3]) / (den + b[3]);
    } else
      t = x * a[3] / b[3];
    return log(0.5 - t);
  }
  if (y <= sqrt(32.0)) {
    num = c[8] * y;
    den = y;
    for (int i = 0; i < 7; i++) {
      num = (num + c[i]) * y;
      den = (den + d[i]) * y;
    }
    t = (num + c[7]) / (den + d[7]);
    return doDel(y, t, x <= 0.0);
  }
  if (y < 1e170) {
    double xsq = 1.0 / (x * x);
    num = p[5] * xsq;
    den = xsq;
    for (int i = 0; i < 4; i++) {
      num = (num + p[i]) * xsq;
      den = (den + q[i]) * xsq;
    }
    t = xsq * (num + p[4]) / (den + q[4]);
    t = (1 / sqrt(2 * M_PI) - t) / y;
    return doDel(x, t, x <= 0.0);
  }
  return -0.0;
}

/* calcPval (1628-1653) with plnorm (1617-1621) folded in */
float gxo_calc_pval(float expt, float ctrl) {
  if (ctrl == GX_SKIP) return GX_SKIP;
  if (ctrl == 0.0f) return expt == 0.0f ? 0.0f : FLT_MAX;
  if (expt == 0.0f) return 0.0f;
  double meanlog, sdlog, mu = ctrl;
  if (mu > 7.0) {
    double sd = 10.0 * log10(mu);
    mu *= mu;
    sd *= sd;
    meanlog = log(mu / sqrt(sd + mu));
    sdlog = sqrt(log1p(sd / mu));
  } else {
    meanlog = log(mu) - LOGSQRT;
    sdlog = SQRTLOG;
  }
  double pv;
  if (sdlog == 0.0)
    pv = (double)expt < meanlog ? 0.0 : FLT_MAX;
  else
    pv = -pnormUpperLog((log((double)expt) - meanlog) / sdlog) / M_LN10;
  return pv > FLT_MAX ? FLT_MAX : (float)pv;
}

/* ---- chi-squared upper tail for Fisher's method (403-559) -------------- */

static double log1Exp(double x) { /* R_Log1_Exp, 407 */
  return x > -M_LN2 ? log(-expm1(x)) : log1p(-exp(x));
}

static double bd0(double x, double np) { /* 412-430 */
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

static double stirlerr(double n) { /* 436-469 */
  static const double sferr[16] = {0.0,
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
  const double S0 = 1.0 / 12, S1 = 1.0 / 360, S2 = 1.0 / 1260, S3 = 1.0 / 1680,
               S4 = 1.0 / 1188;
  double nn = n * n;
  if (n > 80.0) return (S0 - (S1 - S2 / nn) / nn) / n;
  if (n > 35.0) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
  if (n > 15.0) return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
  return sferr[(int)n];
}

static double dpoisLog(double x, double lambda) { /* 474-477 */
  return -0.5 * log(2.0 * M_PI * x) - stirlerr(x) - bd0(x, lambda);
}

static double pgammaUpperLog(double x, double alph) { /* pgamma 528-545 + helpers 482-522 */
  if (x < 1) { /* pgamma_smallx 509 */
    double sum = 0.0, c = alph, n = 0.0, term;
    do {
      n++;
      c *= -x / n;
      term = c / (alph + n);
      sum += term;
    } while (fabs(term) > DBL_EPSILON * fabs(sum));
    double lf2 = alph * log(x) - lgamma(alph + 1);
    return log1Exp(log1p(sum) + lf2);
  }
  if (x <= alph - 1) { /* pd_upper_series 482 */
    double a = alph, term = x / a, sum = term;
    do {
      a++;
      term *= x / a;
      sum += term;
    } while (term > sum * DBL_EPSILON);
    return log1Exp(log(sum) + dpoisLog(alph - 1, x));
  }
  /* pd_lower_series 496 */
  double y = alph - 1, term = 1, sum = 0;
  while (y >= 1 && term > sum * DBL_EPSILON) {
    term *= y / x;
    sum += term;
    y--;
  }
  return log1p(sum) + dpoisLog(alph - 1, x);
}

/* pchisq (555-559): -log10 of the upper tail; df even in [4,400]; returns NaN on bad df */
double gxo_pchisq(double x, int df) {
  if (df < 4 || df > 400 || (df & 1)) return NAN;
  return -pgammaUpperLog(x / 2.0, df / 2.0) / M_LN10;
}

/* ---- small utilities --------------------------------------------------- */

static void rlePush(Rle* r, uint32_t end, float cov) {
  if (r->len == r->mem) {
    r->mem = r->mem ? r->mem * 2 : 16;
    r->end = (uint32_t*)realloc(r->end, (size_t)r->mem * sizeof(uint32_t));
    r->cov = (float*)realloc(r->cov, (size_t)r->mem * sizeof(float));
  }
  r->end[r->len] = end;
  r->cov[r->len] = cov;
  r->len++;
}

static void rleFree(Rle* r) {
  free(r->end);
  free(r->cov);
  memset(r, 0, sizeof(*r));
}

/* -E cursor shared by the four per-base walks (2185-2195, 2091-2101, 1991-2001) */
typedef struct {
  int idx;
  uint32_t pos;
  bool save;
} BedCur;

static uint32_t bedAt(const OChrom* c, int idx) {
  return idx < c->bedLen ? c->bed[idx] : c->len + 1;
}

static BedCur bedInit(const OChrom* c) {
  BedCur b = {0, bedAt(c, 0), true};
  if (b.pos == 0) {
    b.save = false;
    b.idx = 1;
    b.pos = bedAt(c, 1);
  }
  return b;
}

static void bedStep(const OChrom* c, BedCur* b) {
  b->save = !b->save;
  b->idx++;
  b->pos = bedAt(c, b->idx);
}

/* ---- life cycle -------------------------------------------------------- */

int gxo_create(gxo_ctx** out, const gx_params* par) {
  initEst();
  gxo_ctx* x = (gxo_ctx*)calloc(1, sizeof(gxo_ctx));
  if (!x) return GX_ERR_MEM;
  x->par = *par;
  *out = x;
  return GX_OK;
}

static void chromFree(OChrom* c) {
  free(c->bed);
  free(c->diff);
  rleFree(&c->expt);
  rleFree(&c->ctrl);
  for (int j = 0; j < c->sample; j++)
    if (c->pval[j]) {
      rleFree(c->pval[j]);
      free(c->pval[j]);
    }
  free(c->pval);
  free(c->qval);
}

void gxo_destroy(gxo_ctx* x) {
  if (!x) return;
  for (int i = 0; i < x->chromLen; i++) chromFree(x->chrom + i);
  free(x->chrom);
  free(x->peaks);
  free(x);
}

const char* gxo_last_error(const gxo_ctx* x) { return x->err; }

/* saveChrom (4220-4270) for a whole table at once */
int gxo_set_chroms(gxo_ctx* x, int n, const uint32_t* len, const uint8_t* skip,
                   const uint32_t* const* bed, const int32_t* bedLen) {
  x->chrom = (OChrom*)calloc((size_t)n, sizeof(OChrom));
  x->chromLen = n;
  for (int i = 0; i < n; i++) {
    OChrom* c = x->chrom + i;
    c->len = len[i];
    c->skip = skip && skip[i];
    if (bed && bedLen && bedLen[i] > 0 && !c->skip) {
      c->bedLen = bedLen[i];
      c->bed = (uint32_t*)malloc((size_t)c->bedLen * sizeof(uint32_t));
      memcpy(c->bed, bed[i], (size_t)c->bedLen * sizeof(uint32_t));
    }
  }
  return GX_OK;
}

/* runProgram 5463-5464 (reset save) and 5503-5510 (re-zero diff arrays) */
int gxo_sample_begin(gxo_ctx* x, int isCtrl, const uint8_t* save) {
  if (!isCtrl) {
    if (x->phase != 0) return GX_ERR_ORDER;
    for (int i = 0; i < x->chromLen; i++) x->chrom[i].save = save ? save[i] != 0 : true;
    x->fragLen = 0.0;
    x->phase = 1;
  } else {
    if (x->phase != 2) return GX_ERR_ORDER;
    x->phase = 3;
  }
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->diff) memset(c->diff, 0, ((size_t)c->len + 1) * sizeof(int32_t));
  }
  return GX_OK;
}

/* saveInterval's accumulate step (2546-2583) incl. the int16 saturation skips */
int gxo_push_events(gxo_ctx* x, const gx_event* ev, size_t n) {
  if (x->phase != 1 && x->phase != 3) return GX_ERR_ORDER;
  for (size_t i = 0; i < n; i++) {
    const gx_event* e = ev + i;
    if ((int)e->chrom >= x->chromLen) return GX_ERR_ORDER;
    OChrom* c = x->chrom + e->chrom;
    if (e->start >= c->len) return GX_ERR_POS; /* 2531 */
    uint32_t end = e->end > c->len ? c->len : e->end;
    int w;
    switch (e->count) {
      case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10:
        w = UNIT / (int)e->count;
        break;
      default:
        return GX_ERR_ALNS; /* 2400, 2483 */
    }
    if (!c->diff) { /* 2547-2555 */
      c->diff = (int32_t*)calloc((size_t)c->len + 1, sizeof(int32_t));
      if (!c->diff) return GX_ERR_MEM;
    }
    int a, b, d;
    if (canonCov(c->diff[e->start], &a, &b, &d) == INT16_MAX ||
        canonCov(c->diff[end], &a, &b, &d) == INT16_MIN) { /* 2558-2573 */
      x->skippedOverflow++;
      continue;
    }
    c->diff[e->start] += w;
    c->diff[end] -= w;
  }
  return GX_OK;
}

/* ---- pileups ------------------------------------------------------------ */

static int fail(gxo_ctx* x, int code, const char* msg) {
  snprintf(x->err, sizeof x->err, "%s", msg);
  return code;
}

/* fragLen += term (2246, 2271), and beside it the exact parts / the count of roundings (see gxo_ctx) */
static void addTerm(gxo_ctx* x, double* sum, float term) {
  const double a = *sum, b = (double)term, s = a + b;
  const double bb = s - a, err = (a - (s - bb)) + (b - bb); /* TwoSum: err != 0 <=> the addition rounded */
  if (err != 0.0) x->fragInexact++;
  *sum = s;
  const float fl = floorf(term);
  x->fragHi += (long long)fl;
  x->fragLo += (long long)((term - fl) * 134217728.0f);
}

/* savePileupExpt (2168-2295) */
static int pileupExpt(gxo_ctx* x, double* fragOut) {
  double fragLen = 0.0;
  x->fragHi = x->fragLo = 0;
  x->fragInexact = 0;
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->save) continue;
    c->expt.len = 0;
    if (!c->diff) { /* 2178-2182 */
      rlePush(&c->expt, c->len, 0.0f);
      continue;
    }
    BedCur b = bedInit(c);
    const int32_t* d = c->diff;
    int64_t v = d[0];
    int neg;
    float val = gxo_getval(v, &neg);
    if (neg) return fail(x, GX_ERR_PILE, "Invalid pileup value (< 0)");
    uint32_t start = 0, j;
    for (j = 1; j < c->len; j++) {
      if (j == b.pos || (b.save && d[j])) { /* 2241 */
        if (b.save) {
          rlePush(&c->expt, j, val);
          addTerm(x, &fragLen, (float)(j - start) * val); /* 2246: uint32*float in float, summed in double */
        } else
          rlePush(&c->expt, j, 0.0f);
        start = j;
      }
      if (d[j]) { /* 2254 */
        v += d[j];
        val = gxo_getval(v, &neg);
        if (neg) return fail(x, GX_ERR_PILE, "Invalid pileup value (< 0)");
      }
      if (j == b.pos) bedStep(c, &b);
    }
    if (b.save) { /* 2268-2273 */
      rlePush(&c->expt, j, val);
      addTerm(x, &fragLen, (float)(j - start) * val);
    } else
      rlePush(&c->expt, j, 0.0f);
    if (v + d[j] != 0) /* 2283-2289 */
      return fail(x, GX_ERR_ARR, "Experimental pileup does not finish at 0.0");
  }
  if (fragLen == 0.0) return fail(x, GX_ERR_EXPT, "Experimental sample has no analyzable fragments");
  *fragOut = fragLen;
  return GX_OK;
}

/* calcLambda (1817-1832) */
static int calcLambda(gxo_ctx* x, float* lambda) {
  uint64_t g = x->par.genome_len;
  if (!g) {
    for (int i = 0; i < x->chromLen; i++) {
      OChrom* c = x->chrom + i;
      if (!c->skip && c->save) {
        g += c->len;
        for (int j = 0; j < c->bedLen; j += 2) g -= c->bed[j + 1] - c->bed[j];
      }
    }
    if (!g) return fail(x, GX_ERR_GEN, "No analyzable genome (length=0)");
  }
  *lambda = (float)(x->fragLen / (double)g);
  return GX_OK;
}

/* saveLambda (1838-1877) */
static void saveLambda(OChrom* c, float lambda) {
  c->ctrl.len = 0;
  if (c->bedLen == 0) {
    rlePush(&c->ctrl, c->len, lambda);
    return;
  }
  int num = c->bedLen + 1, idx = 0;
  bool save = true;
  if (c->bed[0] == 0) {
    num--;
    idx++;
    save = false;
  }
  if (c->bed[c->bedLen - 1] == c->len) num--;
  for (int j = 0; j < num - 1; j++) {
    rlePush(&c->ctrl, c->bed[idx], save ? lambda : GX_SKIP);
    save = !save;
    idx++;
  }
  rlePush(&c->ctrl, c->len, save ? lambda : GX_SKIP);
}

/* calcFactor (1980-2046) */
static float calcFactor(gxo_ctx* x, int* err) {
  double ctrlFrag = 0.0;
  *err = GX_OK;
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->save || !c->diff) continue;
    BedCur b = bedInit(c);
    const int32_t* d = c->diff;
    int64_t v = d[0];
    int neg;
    float val = gxo_getval(v, &neg);
    if (neg) { *err = GX_ERR_PILE; return 0; }
    uint32_t start = 0, j;
    for (j = 1; j < c->len; j++) {
      if (j == b.pos || (b.save && d[j])) {
        if (b.save) ctrlFrag += (float)(j - start) * val; /* 2018 */
        start = j;
      }
      if (d[j]) {
        v += d[j];
        val = gxo_getval(v, &neg);
        if (neg) { *err = GX_ERR_PILE; return 0; }
      }
      if (j == b.pos) bedStep(c, &b);
    }
    if (b.save) ctrlFrag += (float)(j - start) * val;
  }
  if (!ctrlFrag) return 1.0f;
  return (float)(x->fragLen / ctrlFrag);
}

/* savePileupCtrl (2052-2161) */
static int pileupCtrl(gxo_ctx* x, float* lambdaOut, float* factorOut) {
  float lambda;
  int rc = calcLambda(x, &lambda);
  if (rc) return rc;
  float factor = calcFactor(x, &rc);
  if (rc) return fail(x, rc, "Invalid pileup value (< 0)");
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->save) continue;
    if (!c->diff) {
      saveLambda(c, lambda);
      continue;
    }
    c->ctrl.len = 0;
    BedCur b = bedInit(c);
    const int32_t* d = c->diff;
    int64_t v = d[0];
    int neg;
    float val = factor * gxo_getval(v, &neg); /* 2107 */
    if (neg) return fail(x, GX_ERR_PILE, "Invalid pileup value (< 0)");
    float net = val > lambda ? val : lambda;    /* MAX(val, lambda) 2109 */
    uint32_t j;
    for (j = 1; j < c->len; j++) {
      if (d[j]) { /* 2117 */
        v += d[j];
        val = factor * gxo_getval(v, &neg);
        if (neg) return fail(x, GX_ERR_PILE, "Invalid pileup value (< 0)");
      }
      float cur = val > lambda ? val : lambda;
      if (j == b.pos || (b.save && net != cur)) /* 2122 */
        rlePush(&c->ctrl, j, b.save ? net : GX_SKIP);
      net = cur;
      if (j == b.pos) bedStep(c, &b);
    }
    rlePush(&c->ctrl, j, b.save ? net : GX_SKIP); /* 2140 */
    if (v + d[j] != 0) return fail(x, GX_ERR_ARR, "Control pileup does not finish at 0.0");
  }
  if (lambdaOut) *lambdaOut = lambda;
  if (factorOut) *factorOut = factor;
  return GX_OK;
}

int gxo_sample_end(gxo_ctx* x, double* fragLen, float* lambda, float* factor) {
  if (x->phase == 1) {
    int rc = pileupExpt(x, &x->fragLen);
    if (rc) return rc;
    if (fragLen) *fragLen = x->fragLen;
    x->phase = 2;
    return GX_OK;
  }
  if (x->phase == 3) {
    int rc = pileupCtrl(x, lambda, factor);
    if (rc) return rc;
    if (fragLen) *fragLen = x->fragLen;
    x->phase = 4;
    return GX_OK;
  }
  return GX_ERR_ORDER;
}

/* savePileupNoCtrl (1883-1896) */
int gxo_sample_no_control(gxo_ctx* x, float* lambdaOut) {
  if (x->phase != 2) return GX_ERR_ORDER;
  float lambda;
  int rc = calcLambda(x, &lambda);
  if (rc) return rc;
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->save) continue;
    saveLambda(c, lambda);
  }
  if (lambdaOut) *lambdaOut = lambda;
  x->phase = 4;
  return GX_OK;
}

/* ---- p-values per replicate: savePval (1720-1794) ----------------------- */

static void padPval(OChrom* c, int n) { /* 1735-1741 */
  if (c->sample < n) {
    c->pval = (Rle**)realloc(c->pval, (size_t)n * sizeof(Rle*));
    for (int j = c->sample; j < n; j++) c->pval[j] = NULL;
    c->sample = n;
  }
}

static void appendPval(OChrom* c, Rle* r) {
  c->pval = (Rle**)realloc(c->pval, (size_t)(c->sample + 1) * sizeof(Rle*));
  c->pval[c->sample++] = r;
}

/* pile != NULL: also write the -k lines (printPileHeader 1680, printPile 1697);
 * names[] = chromosome names */
int gxo_pvalues_k(gxo_ctx* x, FILE* pile, const char* const* names, const char* exptName,
                  const char* ctrlName) {
  if (x->phase != 4) return GX_ERR_ORDER;
  int n = x->sample;
  if (pile) {
    fprintf(pile, "# experimental file: %s; control file: %s\n", exptName,
            ctrlName && strcmp(ctrlName, "null") ? ctrlName : "NA");
    fprintf(pile, "chr\tstart\tend\texperimental\tcontrol\t-log(p)\n");
  }
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip) continue;
    padPval(c, n);
    if (!c->save) { /* 1744-1750 */
      appendPval(c, NULL);
      continue;
    }
    Rle* p = (Rle*)calloc(1, sizeof(Rle));
    uint32_t start = 0, j = 0, k = 0;
    for (;;) { /* two-pointer merge, 1768-1791 */
      uint32_t end;
      float ev = c->expt.cov[j], cv = c->ctrl.cov[k];
      bool last = false;
      if (c->ctrl.end[k] < c->expt.end[j]) {
        end = c->ctrl.end[k];
        k++;
      } else {
        end = c->expt.end[j];
        if (c->ctrl.end[k] == c->expt.end[j]) k++;
        j++;
        last = j == c->expt.len;
      }
      float pv = gxo_calc_pval(ev, cv);
      rlePush(p, end, pv);
      if (pile) {
        if (cv == GX_SKIP)
          fprintf(pile, "%s\t%d\t%d\t%f\t%f\t%s\n", names[i], start, end, ev, 0.0f, "NA");
        else
          fprintf(pile, "%s\t%d\t%d\t%f\t%f\t%f\n", names[i], start, end, ev, cv, pv);
      }
      start = end;
      if (last) break;
    }
    appendPval(c, p);
  }
  x->sample++;
  x->phase = 0;
  return GX_OK;
}

int gxo_pvalues(gxo_ctx* x) { return gxo_pvalues_k(x, NULL, NULL, NULL, NULL); }

/* ---- Fisher combine: multPval (567-583), combinePval (612-667) ---------- */

static float multPval(OChrom* c, int n, const uint32_t* idx, int* err) {
  double sum = 0.0;
  int df = 0;
  for (int j = 0; j < n; j++)
    if (c->pval[j] && c->pval[j]->cov[idx[j]] != GX_SKIP) {
      sum += c->pval[j]->cov[idx[j]];
      df += 2;
    }
  if (df == 0) return GX_SKIP;
  if (df == 2 || !sum) return (float)sum;
  double p = gxo_pchisq(2.0 * sum / M_LOG10E, df);
  if (isnan(p)) {
    *err = GX_ERR_DF;
    return 0;
  }
  return p > FLT_MAX ? FLT_MAX : (float)p;
}

static int combinePval(gxo_ctx* x, int n) {
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip) continue;
    padPval(c, n);
    int j;
    for (j = 0; j < n; j++)
      if (c->pval[j]) break;
    if (j == n) { /* 625-631 */
      appendPval(c, NULL);
      continue;
    }
    /* union of breakpoints: walk by next-smallest end instead of per base (649-664) */
    Rle* r = (Rle*)calloc(1, sizeof(Rle));
    uint32_t* idx = (uint32_t*)calloc((size_t)n, sizeof(uint32_t));
    int err = GX_OK;
    for (;;) {
      uint32_t k = UINT32_MAX;
      for (j = 0; j < n; j++)
        if (c->pval[j] && idx[j] < c->pval[j]->len && c->pval[j]->end[idx[j]] < k)
          k = c->pval[j]->end[idx[j]];
      if (k == UINT32_MAX) break;
      rlePush(r, k, multPval(c, n, idx, &err));
      for (j = 0; j < n; j++)
        if (c->pval[j] && idx[j] < c->pval[j]->len && c->pval[j]->end[idx[j]] == k) idx[j]++;
    }
    free(idx);
    appendPval(c, r);
    if (err) return fail(x, err, "Invalid df in pchisq()");
  }
  return GX_OK;
}

/* ---- q-values: computeQval (352-401), saveQval (212-250) --------------- */

typedef struct {
  float p;
  uint64_t len;
} PEnt;

static int cmpPEnt(const void* a, const void* b) {
  float x = ((const PEnt*)a)->p, y = ((const PEnt*)b)->p;
  return (x > y) - (x < y);
}

/* distinct-value table {p -> total bp} (hashPval 300-327 + collectPval 333-347);
 * built with an open-addressing table keyed by the float's bits (+0/-0 folded as C's ==
 * does, 282), then sorted ascending (quickSort 154-188: keys are distinct so any sort
 * gives the same order). */
static PEnt* collectPvals(gxo_ctx* x, int n, size_t* count, uint64_t* checkLen) {
  size_t cap = 1 << 16, used = 0;
  PEnt* tab = (PEnt*)calloc(cap, sizeof(PEnt));
  uint8_t* occ = (uint8_t*)calloc(cap, 1);
  *checkLen = 0;
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->pval[n]) continue;
    Rle* p = c->pval[n];
    uint32_t start = 0;
    for (uint32_t m = 0; m < p->len; m++) {
      float v = p->cov[m];
      if (v != GX_SKIP) {
        if (v == 0.0f) v = 0.0f; /* fold -0 */
        uint32_t bits;
        memcpy(&bits, &v, 4);
        size_t h = (bits * 2654435761u) & (cap - 1);
        while (occ[h] && tab[h].p != v) h = (h + 1) & (cap - 1);
        if (!occ[h]) {
          occ[h] = 1;
          tab[h].p = v;
          tab[h].len = 0;
          used++;
        }
        tab[h].len += p->end[m] - start;
        *checkLen += p->end[m] - start;
        if (used * 2 > cap) { /* grow */
          size_t ncap = cap * 2;
          PEnt* nt = (PEnt*)calloc(ncap, sizeof(PEnt));
          uint8_t* no = (uint8_t*)calloc(ncap, 1);
          for (size_t q = 0; q < cap; q++)
            if (occ[q]) {
              uint32_t b2;
              memcpy(&b2, &tab[q].p, 4);
              size_t h2 = (b2 * 2654435761u) & (ncap - 1);
              while (no[h2]) h2 = (h2 + 1) & (ncap - 1);
              no[h2] = 1;
              nt[h2] = tab[q];
            }
          free(tab);
          free(occ);
          tab = nt;
          occ = no;
          cap = ncap;
        }
      }
      start = p->end[m];
    }
  }
  PEnt* out = (PEnt*)malloc((used ? used : 1) * sizeof(PEnt));
  size_t k = 0;
  for (size_t q = 0; q < cap; q++)
    if (occ[q]) out[k++] = tab[q];
  free(tab);
  free(occ);
  qsort(out, used, sizeof(PEnt), cmpPEnt);
  *count = used;
  return out;
}

static int computeQval(gxo_ctx* x, uint64_t genomeLen, bool genomeOpt, int n) {
  size_t pLen;
  uint64_t checkLen;
  PEnt* tab = collectPvals(x, n, &pLen, &checkLen);
  if (genomeOpt && checkLen != genomeLen) { /* 377-382 */
    free(tab);
    return fail(x, GX_ERR_PVAL, "Genome length does not match p-value length");
  }
  /* saveQval 220-229: from the most significant value down */
  float* qv = (float*)malloc((pLen + 1) * sizeof(float));
  uint64_t k = 1;
  float logN = -log10f((float)genomeLen);
  qv[pLen] = FLT_MAX;
  for (int64_t i = (int64_t)pLen - 1; i > -1; i--) {
    float raw = tab[i].p + logN + log10f((float)k);
    float m = raw < qv[i + 1] ? raw : qv[i + 1];
    qv[i] = m > 0.0f ? m : 0.0f;
    k += tab[i].len;
  }
  for (int i = 0; i < x->chromLen; i++) { /* 232-242 */
    OChrom* c = x->chrom + i;
    if (c->skip || !c->pval[n]) continue;
    Rle* p = c->pval[n];
    free(c->qval);
    c->qval = (float*)malloc((size_t)p->len * sizeof(float));
    for (uint32_t j = 0; j < p->len; j++) {
      float v = p->cov[j];
      if (v == GX_SKIP) {
        c->qval[j] = GX_SKIP;
        continue;
      }
      size_t lo = 0, hi = pLen; /* exact-match binary search (lookup 196-206) */
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (tab[mid].p < v) lo = mid + 1; else hi = mid;
      }
      c->qval[j] = qv[lo];
    }
  }
  free(qv);
  free(tab);
  return GX_OK;
}

/* ---- peak sweep: callPeaks (977-1069) with the -f log (808-831) --------- */

static void peakPush(gxo_ctx* x, gx_peak pk) {
  if (x->nPeaks == x->memPeaks) {
    x->memPeaks = x->memPeaks ? x->memPeaks * 2 : 1024;
    x->peaks = (gx_peak*)realloc(x->peaks, x->memPeaks * sizeof(gx_peak));
  }
  x->peaks[x->nPeaks++] = pk;
}

static void logHeader(FILE* f, int n, bool q, bool sig) { /* printLogHeader 674-717 */
  if (n) {
    fprintf(f, "chr\tstart\tend");
    for (int i = 0; i < n; i++) fprintf(f, "\t-log(p)_%d", i);
    fprintf(f, "\t-log(p)_comb");
  } else
    fprintf(f, "chr\tstart\tend\texperimental\tcontrol\t-log(p)");
  if (q) fprintf(f, "\t-log(q)");
  if (sig) fprintf(f, "\tsignif");
  fprintf(f, "\n");
}

/* one -f row: printInterval (770-803) / printIntervalN (724-763) */
static void logRow(FILE* f, const char* name, OChrom* c, uint32_t start, int n, uint32_t m,
                   uint32_t j, uint32_t k, uint32_t* idx, bool qOpt, bool sig) {
  Rle* p = c->pval[n];
  float pv = p->cov[m], qv = qOpt ? c->qval[m] : GX_SKIP;
  if (!n) {
    float cv = c->ctrl.cov[k];
    if (cv == GX_SKIP) {
      fprintf(f, "%s\t%d\t%d\t%f\t%f\t%s", name, start, p->end[m], c->expt.cov[j], 0.0f, "NA");
      if (qOpt) fprintf(f, "\t%s", "NA");
      fprintf(f, "\n");
    } else {
      fprintf(f, "%s\t%d\t%d\t%f\t%f\t%f", name, start, p->end[m], c->expt.cov[j], cv, pv);
      if (qOpt) fprintf(f, "\t%f", qv);
      fprintf(f, "%s\n", sig ? "\t*" : "");
    }
    return;
  }
  fprintf(f, "%s\t%d\t%d", name, start, p->end[m]);
  for (int r = 0; r < n; r++)
    if (!c->pval[r] || c->pval[r]->cov[idx[r]] == GX_SKIP)
      fprintf(f, "\t%s", "NA");
    else
      fprintf(f, "\t%f", c->pval[r]->cov[idx[r]]);
  if (pv == GX_SKIP) {
    fprintf(f, "\t%s", "NA");
    if (qOpt) fprintf(f, "\t%s", "NA");
  } else {
    fprintf(f, "\t%f", pv);
    if (qOpt) fprintf(f, "\t%f", qv);
  }
  fprintf(f, "%s\n", sig ? "\t*" : "");
  for (int r = 0; r < n; r++) /* 826-829 */
    if (c->pval[r] && c->pval[r]->end[idx[r]] == p->end[m]) idx[r]++;
}

static void callPeaks(gxo_ctx* x, int n, FILE* log, const char* const* names, bool peaksOpt) {
  const bool qOpt = x->par.qval_opt != 0;
  const float thr = x->par.thr;
  if (log) logHeader(log, n, qOpt, peaksOpt);
  x->nPeaks = 0;
  x->peakBP = 0;
  for (int i = 0; i < x->chromLen; i++) {
    OChrom* c = x->chrom + i;
    if (c->skip || !c->pval[n]) continue;
    Rle* p = c->pval[n];
    uint32_t j = 0, k = 0;
    uint32_t* idx = (uint32_t*)calloc((size_t)(n ? n : 1), sizeof(uint32_t));
    float auc = 0.0f, summitVal = -1.0f, summitP = -1.0f, summitQ = -1.0f;
    int64_t peakStart = -1, peakEnd = -1;
    uint32_t summitPos = 0, summitLen = 0, start = 0;
    for (uint32_t m = 0; m <= p->len; m++) {
      bool closeIt = m == p->len, sig = false;
      if (!closeIt && peaksOpt) {
        float pq = qOpt ? c->qval[m] : p->cov[m];
        if (pq > thr) { /* updatePeak 943-970 */
          sig = true;
          uint32_t len = p->end[m] - start;
          auc += (float)len * (pq - thr);
          if (peakStart == -1) peakStart = start;
          peakEnd = p->end[m];
          if (pq > summitVal) {
            summitVal = pq;
            summitP = p->cov[m];
            summitQ = qOpt ? c->qval[m] : GX_SKIP;
            summitPos = (uint32_t)((p->end[m] + start) / 2 - peakStart);
            summitLen = len;
          } else if (pq == summitVal && len > summitLen) {
            summitPos = (uint32_t)((p->end[m] + start) / 2 - peakStart);
            summitLen = len;
          }
        } else if (pq == GX_SKIP || (int64_t)p->end[m] - peakEnd > x->par.max_gap) /* 1031 */
          closeIt = true;
      }
      if (closeIt && peaksOpt) { /* checkPeak 916-927, resetVars 932-938 */
        if (peakStart != -1 && auc >= x->par.min_auc && peakEnd - peakStart >= x->par.min_len) {
          gx_peak pk = {(uint32_t)i, (uint32_t)peakStart, (uint32_t)peakEnd, summitPos,
                        auc,         summitP,             summitQ};
          peakPush(x, pk);
          x->peakBP += (uint64_t)(peakEnd - peakStart);
        }
        peakStart = -1;
        summitVal = -1.0f;
        summitLen = 0;
        auc = 0.0f;
      }
      if (m == p->len) break;
      if (log) logRow(log, names[i], c, start, n, m, j, k, idx, qOpt, sig);
      if (!n) { /* 1049-1057 */
        if (c->ctrl.end[k] < c->expt.end[j])
          k++;
        else {
          if (c->ctrl.end[k] == c->expt.end[j]) k++;
          j++;
        }
      }
      start = p->end[m];
    }
    free(idx);
  }
}

/* findPeaks (1076-1137).  log/names may be NULL.  peaksOpt = 0 is -X (logIntervals 837). */
int gxo_find_peaks_f(gxo_ctx* x, size_t* nPeaks, uint64_t* genomeLenOut, uint64_t* peakBP,
                     FILE* log, const char* const* names, int peaksOpt) {
  if (x->phase != 0 || x->sample < 1) return GX_ERR_ORDER;
  int sample = x->sample;
  if (sample > 1) {
    int rc = combinePval(x, sample);
    if (rc) return rc;
    sample++;
  } else
    for (int i = 0; i < x->chromLen; i++)
      if (!x->chrom[i].skip) padPval(x->chrom + i, sample);
  int n = sample - 1;
  uint64_t g = x->par.genome_len;
  bool genomeOpt = false;
  if (!g) { /* 1091-1101 */
    genomeOpt = true;
    for (int i = 0; i < x->chromLen; i++) {
      OChrom* c = x->chrom + i;
      if (!c->skip && c->pval[n]) {
        g += c->len;
        for (int j = 0; j < c->bedLen; j += 2) g -= c->bed[j + 1] - c->bed[j];
      }
    }
  }
  x->genomeLenUsed = g;
  x->final = n;
  if (x->par.qval_opt) {
    int rc = computeQval(x, g, genomeOpt, n);
    if (rc) return rc;
  }
  callPeaks(x, n, log, names, peaksOpt != 0);
  if (nPeaks) *nPeaks = x->nPeaks;
  if (genomeLenOut) *genomeLenOut = g;
  if (peakBP) *peakBP = x->peakBP;
  return GX_OK;
}

int gxo_find_peaks(gxo_ctx* x, size_t* nPeaks, uint64_t* genomeLen, uint64_t* peakBP) {
  return gxo_find_peaks_f(x, nPeaks, genomeLen, peakBP, NULL, NULL, 1);
}

int gxo_get_peaks(gxo_ctx* x, gx_peak* out, size_t cap) {
  size_t n = x->nPeaks < cap ? x->nPeaks : cap;
  memcpy(out, x->peaks, n * sizeof(gx_peak));
  return GX_OK;
}

/* printPeak (885-909); the (unsigned int) cast of an out-of-range float follows x86-64
 * gcc (cvttss2si to 64 bits, low 32 kept) */
void gxo_write_narrowpeak(gxo_ctx* x, FILE* out, const char* const* names) {
  for (size_t i = 0; i < x->nPeaks; i++) {
    const gx_peak* k = x->peaks + i;
    int64_t len = (int64_t)k->end - (int64_t)k->start;
    float sc = 1000.0f * k->auc / len + 0.5f;
    unsigned int u = (unsigned int)(int64_t)sc;
    fprintf(out, "%s\t%ld\t%ld\tpeak_%d\t%d\t.\t%f\t%f", names[k->chrom], (long)k->start,
            (long)k->end, (int)i, u < 1000u ? u : 1000u, k->auc, k->p);
    if (k->q == GX_SKIP)
      fprintf(out, "\t-1\t%d\n", k->summit);
    else
      fprintf(out, "\t%f\t%d\n", k->q, k->summit);
  }
}

/* ---- interval access (mirrors gx_interval_count / gx_get_intervals) ---- */

static Rle* whichRle(gxo_ctx* x, int which, int chrom, int* n) {
  if (chrom < 0 || chrom >= x->chromLen) return NULL;
  OChrom* c = x->chrom + chrom;
  int w = which == GX_IV_FINAL ? x->final : which;
  *n = w;
  if (c->skip || w < 0 || w >= c->sample) return NULL;
  return c->pval[w];
}

int gxo_interval_count(gxo_ctx* x, int which, int chrom, size_t* n) {
  int w;
  Rle* p = whichRle(x, which, chrom, &w);
  *n = p ? p->len : 0;
  return GX_OK;
}

int gxo_get_intervals(gxo_ctx* x, int which, int chrom, size_t cap, uint32_t* end, float* expt,
                      float* ctrl, float* pv, float* qv) {
  int w;
  Rle* p = whichRle(x, which, chrom, &w);
  if (!p) return GX_OK;
  OChrom* c = x->chrom + chrom;
  uint32_t j = 0, k = 0;
  bool single = x->sample == 1;
  for (uint32_t m = 0; m < p->len && m < cap; m++) {
    if (end) end[m] = p->end[m];
    if (pv) pv[m] = p->cov[m];
    if (qv) qv[m] = (c->qval && w == x->final) ? c->qval[m] : GX_SKIP;
    if (single) {
      if (expt) expt[m] = c->expt.cov[j];
      if (ctrl) ctrl[m] = c->ctrl.cov[k];
      if (c->ctrl.end[k] < c->expt.end[j])
        k++;
      else {
        if (c->ctrl.end[k] == c->expt.end[j]) k++;
        j++;
      }
    }
  }
  return GX_OK;
}

/* file-path conveniences for ctypes callers */
int gxo_pvalues_path(gxo_ctx* x, const char* pilePath, int append, const char* const* names,
                     const char* exptName, const char* ctrlName) {
  FILE* f = pilePath ? fopen(pilePath, append ? "a" : "w") : NULL;
  int rc = gxo_pvalues_k(x, f, names, exptName, ctrlName);
  if (f) fclose(f);
  return rc;
}

int gxo_find_peaks_path(gxo_ctx* x, const char* outPath, const char* logPath,
                        const char* const* names, int peaksOpt, size_t* nPeaks,
                        uint64_t* genomeLen, uint64_t* peakBP) {
  FILE* lf = logPath ? fopen(logPath, "w") : NULL;
  int rc = gxo_find_peaks_f(x, nPeaks, genomeLen, peakBP, lf, names, peaksOpt);
  if (lf) fclose(lf);
  if (rc == GX_OK && outPath && peaksOpt) {
    FILE* of = fopen(outPath, "w");
    gxo_write_narrowpeak(x, of, names);
    fclose(of);
  }
  return rc;
}

double gxo_frag_len(const gxo_ctx* x) { return x->fragLen; }
/* the treatment's fragLen as the exact sum of the same products, rounded once; how many of the reference's additions rounded */
double gxo_frag_len_exact(const gxo_ctx* x) { return (double)x->fragHi + (double)x->fragLo * (1.0 / 134217728.0); }
uint64_t gxo_frag_inexact(const gxo_ctx* x) { return x->fragInexact; }
uint64_t gxo_skipped_overflow(const gxo_ctx* x) { return x->skippedOverflow; }
