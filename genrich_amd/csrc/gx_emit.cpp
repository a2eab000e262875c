// gx_emit.cpp -- host-side text emitters of the drop-in surface: ENCODE narrowPeak (-o), the
// bedgraph-ish log (-f) and the pileup log (-k).  Pure formatting of arrays fetched through the
// C ABI (gx_get_peaks / gx_get_intervals); byte format follows the reference's printf calls:
//   printPeak       Genrich.c:885-909      printLogHeader 674-717
//   printInterval   770-803                printIntervalN 724-763
//   printPileHeader 1680-1691              printPile      1697-1715
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/genrich_amd.h"

namespace {

struct Iv {
  std::vector<uint32_t> end;
  std::vector<float> expt, ctrl, p, q;
  size_t n = 0;
};

int fetch(gx_ctx* ctx, int which, int chrom, Iv& iv, bool piles, bool qv) {
  size_t n = 0;
  int rc = gx_interval_count(ctx, which, chrom, &n);
  if (rc) return rc;
  iv.n = n;
  iv.end.assign(n, 0);
  iv.p.assign(n, 0);
  iv.expt.assign(piles ? n : 0, 0);
  iv.ctrl.assign(piles ? n : 0, 0);
  iv.q.assign(qv ? n : 0, 0);
  if (!n) return GX_OK;
  return gx_get_intervals(ctx, which, chrom, n, iv.end.data(), piles ? iv.expt.data() : nullptr,
                          piles ? iv.ctrl.data() : nullptr, iv.p.data(), qv ? iv.q.data() : nullptr);
}

}  // namespace

extern "C" {

// -o: one line per peak; peak_N numbers the peaks in output order (callPeaks' `count`).  Several contexts
// (chromosomes sharded over GPUs): their peak lists are merged into chromosome-table order, then position --
// the order in which the reference meets them (986, 925).
int gx_write_narrowpeak_group(gx_ctx* const* ctxs, int n_ctx, const char* const* names, FILE* out) {
  std::vector<gx_peak> pk;
  for (int g = 0; g < n_ctx; g++) {
    size_t k = 0;
    int rc = gx_peak_count(ctxs[g], &k);
    if (rc) return rc;
    const size_t at = pk.size();
    pk.resize(at + k);
    if (k && (rc = gx_get_peaks(ctxs[g], pk.data() + at, k))) return rc;
  }
  if (n_ctx > 1)
    std::stable_sort(pk.begin(), pk.end(), [](const gx_peak& a, const gx_peak& b) {
      return a.chrom != b.chrom ? a.chrom < b.chrom : a.start < b.start;
    });
  const size_t n = pk.size();
  for (size_t i = 0; i < n; i++) {
    const gx_peak& k = pk[i];
    long start = (long)k.start, end = (long)k.end;
    // MIN((unsigned int)(1000.0f * signal / (end - start) + 0.5f), 1000): the float -> unsigned
    // conversion of an out-of-range value follows x86-64 gcc (64-bit cvttss2si, low 32 bits)
    float sc = 1000.0f * k.auc / (float)(end - start) + 0.5f;
    unsigned int u = (unsigned int)(long long)sc;
    fprintf(out, "%s\t%ld\t%ld\tpeak_%d\t%d\t.\t%f\t%f", names[k.chrom], start, end, (int)i, u < 1000u ? u : 1000u, k.auc,
            k.p);
    if (k.q == GX_SKIP)
      fprintf(out, "\t-1\t%d\n", k.summit);
    else
      fprintf(out, "\t%f\t%d\n", k.q, k.summit);
  }
  return GX_OK;
}

int gx_write_narrowpeak(gx_ctx* ctx, const char* const* names, FILE* out) { return gx_write_narrowpeak_group(&ctx, 1, names, out); }

// -k for replicate `rep` (owner[c] = index into ctxs of the context that computed chromosome c; NULL: ctxs[0])
int gx_write_pile_group(gx_ctx* const* ctxs, const int* owner, int rep, const char* const* names, int n_chrom,
                        const char* expt_name, const char* ctrl_name, FILE* out) {
  fprintf(out, "# experimental file: %s; control file: %s\n", expt_name,
          ctrl_name && strcmp(ctrl_name, "null") ? ctrl_name : "NA");
  fprintf(out, "chr\tstart\tend\texperimental\tcontrol\t-log(p)\n");
  Iv iv;
  for (int c = 0; c < n_chrom; c++) {
    int rc = fetch(ctxs[owner ? owner[c] : 0], rep, c, iv, true, false);
    if (rc) return rc;
    uint32_t start = 0;
    for (size_t m = 0; m < iv.n; m++) {
      if (iv.ctrl[m] == GX_SKIP)
        fprintf(out, "%s\t%d\t%d\t%f\t%f\t%s\n", names[c], start, iv.end[m], iv.expt[m], 0.0f, "NA");
      else
        fprintf(out, "%s\t%d\t%d\t%f\t%f\t%f\n", names[c], start, iv.end[m], iv.expt[m], iv.ctrl[m], iv.p[m]);
      start = iv.end[m];
    }
  }
  return GX_OK;
}

int gx_write_pile(gx_ctx* ctx, int rep, const char* const* names, int n_chrom, const char* expt_name,
                  const char* ctrl_name, FILE* out) {
  return gx_write_pile_group(&ctx, nullptr, rep, names, n_chrom, expt_name, ctrl_name, out);
}

// -f after gx_find_peaks.  n_rep = number of replicates; peaks_opt = 0 for -X (logIntervals 837).
// thr / qval_opt as given to gx_create.
int gx_write_log_group(gx_ctx* const* ctxs, const int* owner, int n_rep, const char* const* names, int n_chrom, int qval_opt,
                       int peaks_opt, float thr, FILE* out) {
  const bool multi = n_rep > 1;
  if (multi) {
    fprintf(out, "chr\tstart\tend");
    for (int i = 0; i < n_rep; i++) fprintf(out, "\t-log(p)_%d", i);
    fprintf(out, "\t-log(p)_comb");
  } else
    fprintf(out, "chr\tstart\tend\texperimental\tcontrol\t-log(p)");
  if (qval_opt) fprintf(out, "\t-log(q)");
  if (peaks_opt) fprintf(out, "\tsignif");
  fprintf(out, "\n");
  Iv fin;
  std::vector<Iv> reps(multi ? n_rep : 0);
  for (int c = 0; c < n_chrom; c++) {
    gx_ctx* ctx = ctxs[owner ? owner[c] : 0];
    int rc = fetch(ctx, GX_IV_FINAL, c, fin, !multi, qval_opt != 0);
    if (rc) return rc;
    if (!fin.n) continue;
    std::vector<size_t> idx(multi ? n_rep : 0, 0);
    for (int r = 0; multi && r < n_rep; r++)
      if ((rc = fetch(ctx, r, c, reps[r], false, false))) return rc;
    uint32_t start = 0;
    for (size_t m = 0; m < fin.n; m++) {
      const float pv = fin.p[m], qv = qval_opt ? fin.q[m] : GX_SKIP;
      const float pq = qval_opt ? qv : pv;
      const bool sig = peaks_opt && pq > thr;  // callPeaks 1015
      if (!multi) {
        if (fin.ctrl[m] == GX_SKIP) {
          fprintf(out, "%s\t%d\t%d\t%f\t%f\t%s", names[c], start, fin.end[m], fin.expt[m], 0.0f, "NA");
          if (qval_opt) fprintf(out, "\t%s", "NA");
          fprintf(out, "\n");
        } else {
          fprintf(out, "%s\t%d\t%d\t%f\t%f\t%f", names[c], start, fin.end[m], fin.expt[m], fin.ctrl[m], pv);
          if (qval_opt) fprintf(out, "\t%f", qv);
          fprintf(out, "%s\n", sig ? "\t*" : "");
        }
      } else {
        fprintf(out, "%s\t%d\t%d", names[c], start, fin.end[m]);
        for (int r = 0; r < n_rep; r++)
          if (!reps[r].n || reps[r].p[idx[r]] == GX_SKIP)
            fprintf(out, "\t%s", "NA");
          else
            fprintf(out, "\t%f", reps[r].p[idx[r]]);
        if (pv == GX_SKIP) {
          fprintf(out, "\t%s", "NA");
          if (qval_opt) fprintf(out, "\t%s", "NA");
        } else {
          fprintf(out, "\t%f", pv);
          if (qval_opt) fprintf(out, "\t%f", qv);
        }
        fprintf(out, "%s\n", sig ? "\t*" : "");
        for (int r = 0; r < n_rep; r++)  // printLog 826-829
          if (reps[r].n && reps[r].end[idx[r]] == fin.end[m]) idx[r]++;
      }
      start = fin.end[m];
    }
  }
  return GX_OK;
}

// path-taking conveniences for FFI callers without a FILE*
int gx_write_log(gx_ctx* ctx, int n_rep, const char* const* names, int n_chrom, int qval_opt, int peaks_opt, float thr,
                 FILE* out) {
  return gx_write_log_group(&ctx, nullptr, n_rep, names, n_chrom, qval_opt, peaks_opt, thr, out);
}

int gx_write_narrowpeak_path(gx_ctx* ctx, const char* const* names, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return GX_ERR_ORDER;
  int rc = gx_write_narrowpeak(ctx, names, f);
  fclose(f);
  return rc;
}
int gx_write_pile_path(gx_ctx* ctx, int rep, const char* const* names, int n_chrom, const char* expt_name,
                       const char* ctrl_name, const char* path, int append) {
  FILE* f = fopen(path, append ? "a" : "w");
  if (!f) return GX_ERR_ORDER;
  int rc = gx_write_pile(ctx, rep, names, n_chrom, expt_name, ctrl_name, f);
  fclose(f);
  return rc;
}
int gx_write_log_path(gx_ctx* ctx, int n_rep, const char* const* names, int n_chrom, int qval_opt, int peaks_opt,
                      float thr, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return GX_ERR_ORDER;
  int rc = gx_write_log(ctx, n_rep, names, n_chrom, qval_opt, peaks_opt, thr, f);
  fclose(f);
  return rc;
}

}  // extern "C"
