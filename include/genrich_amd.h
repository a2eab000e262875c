/*
 * genrich_amd.h -- C ABI of the MI355X-native Genrich hot path.
 *
 * The reference (jsh58/Genrich v0.6.2) is a monolith with no plugin/FFI
 * interface; this header is the drop-in boundary cut where its data narrows
 * (SURVEY.md section 8b).  Every entry point names the reference code it
 * replaces (file:line into Genrich.c / Genrich.h).  Plain C types only: no
 * torch, no HIP types.  Handle based, not thread-safe (the reference is
 * single-threaded, README.md:535).
 *
 * Call order for one run (mirrors runProgram, Genrich.c:5386-5607):
 *
 *   gx_create -> gx_set_chroms
 *   for each replicate:
 *     gx_sample_begin(ctx, 0, save) ; gx_push_events* ; gx_sample_end   (treatment)
 *     either  gx_sample_begin(ctx, 1, NULL) ; gx_push_events* ; gx_sample_end   (control)
 *     or      gx_sample_no_control
 *     gx_pvalues
 *   gx_find_peaks -> gx_get_peaks / gx_get_intervals
 *   gx_destroy
 */
#ifndef GENRICH_AMD_H
#define GENRICH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GX_SKIP (-1.0f) /* Genrich.h:27  SKIP: statistic of an excluded (-E) region */

/* Status codes.  0 = success; negative values map 1:1 onto the reference's
 * errCode enum (Genrich.h:97-106) so a host can print the identical
 * "Error! <msg>" text (Genrich.c:78-81) and exit(1). */
enum gx_status {
  GX_OK = 0,
  GX_ERR_MEM = -1,      /* ERRMEM    "Cannot allocate memory" */
  GX_ERR_GEN = -2,      /* ERRGEN    "No analyzable genome (length=0)"           Genrich.c:1828 */
  GX_ERR_EXPT = -3,     /* ERREXPT   "Experimental sample has no analyzable fragments" :2292 */
  GX_ERR_PILE = -4,     /* ERRPILE   "Invalid pileup value (< 0)"                :1921,1969 */
  GX_ERR_POS = -5,      /* ERRPOS    ": read aligned beyond reference end"       :2531 */
  GX_ERR_ALNS = -6,     /* ERRALNS   "Disallowed number of alignments"           :2402,2485 */
  GX_ERR_ARR = -7,      /* ERRARR / ERRARRC / "finishes at %f (not 0.0)"         :2144,2276,2153,2284 */
  GX_ERR_PVAL = -8,     /* ERRPVAL / genome-length mismatch                      :344,377 */
  GX_ERR_DF = -9,       /* ERRDF     "Invalid df in pchisq()"                    :556 */
  GX_ERR_ORDER = -10,   /* API called out of order / bad argument */
  GX_ERR_DEVICE = -11   /* HIP runtime failure (message via gx_last_error) */
  /* (-12 was GX_ERR_OVERFLOW until round 5; the reference's int16 saturation skips, Genrich.c:2558-2573, are
   * reproduced -- gx_filter_saturation -- so nothing reports it: retired, the number is not reused) */
};

/* One alignment-derived interval AFTER saveInterval's clamping
 * (Genrich.c:2522-2544): 0 <= start < len(chrom), start <= end <= len.
 * weight = 1/count, count in {1,2,3,4,5,6,8,10} (Genrich.c:2576-2583,
 * addFrac 2311, subFrac 2412). */
typedef struct gx_event {
  uint32_t chrom; /* index into the table given to gx_set_chroms */
  uint32_t start;
  uint32_t end;
  uint32_t count;
} gx_event;

/* Peak-calling parameters: the tail of runProgram's argument list
 * (Genrich.c:5390-5395) after getArgs' conversions (5796-5817). */
typedef struct gx_params {
  float thr;           /* minPQval = -log10f(-p or -q value)           Genrich.c:5817 */
  int32_t qval_opt;    /* 1: significance on q-values (-q)             :5790 */
  float min_auc;       /* -a, DEFAUC 200.0f                            Genrich.h:31 */
  int32_t min_len;     /* -l, DEFMINLEN 0 */
  int32_t max_gap;     /* -g, DEFMAXGAP 100 */
  int32_t device;      /* HIP device ordinal (ignored by the CPU oracle) */
  uint64_t genome_len; /* -L, 0 = compute from the chromosome table    :1819,1091 */
} gx_params;

/* One called peak = the arguments of printPeak (Genrich.c:885-887). */
typedef struct gx_peak {
  uint32_t chrom;
  uint32_t start;
  uint32_t end;
  uint32_t summit; /* offset of the summit from start (narrowPeak column 10) */
  float auc;       /* signal (column 7) */
  float p;         /* summit -log10 p (column 8) */
  float q;         /* summit -log10 q (column 9) or GX_SKIP without -q */
} gx_peak;

typedef struct gx_ctx gx_ctx;

/* ---- life cycle ------------------------------------------------------- */

int gx_create(gx_ctx** ctx, const gx_params* params);
void gx_destroy(gx_ctx* ctx);
/* Forget all replicates and results but keep the chromosome table and device buffers
 * (a fresh run without re-allocating; what the reference does by exiting the process). */
int gx_reset(gx_ctx* ctx);
const char* gx_last_error(const gx_ctx* ctx);
const char* gx_strerror(int status); /* the reference's errMsg text, Genrich.h:107-154 */

/* Chromosome table in header order = output order (saveChrom, Genrich.c:4220-4270).
 * skip[i]   : chromosome listed in -e (Chrom.skip, :4243).
 * bed[i]    : merged, sorted -E coordinates as start/end pairs (Chrom.bed, saveXBed :1144);
 *             bed_len[i] = number of coordinates (even). bed/bed_len may be NULL. */
int gx_set_chroms(gx_ctx* ctx, int n, const uint32_t* len, const uint8_t* skip,
                  const uint32_t* const* bed, const int32_t* bed_len);

/* ---- per-sample event stream (replaces saveInterval's accumulate step,
 *      Genrich.c:2546-2583, and the re-zeroing of diff arrays, :5503-5510) ---- */

/* is_ctrl = 0: start the treatment file of the next replicate; save[i] = chromosome
 * appears in this treatment file's header (Chrom.save, :4230-4244, reset :5463).
 * is_ctrl = 1: start the matching control file (save must be NULL). */
int gx_sample_begin(gx_ctx* ctx, int is_ctrl, const uint8_t* save);

/* Host-resident events; consumed before return (the caller may reuse its buffer at once).  They go through
 * two pinned staging buffers and asynchronous copies on a side stream: the call does not wait for the
 * upload, so the host parses the next batch while this one travels, and gx_sample_end starts its first
 * kernel on the pieces that have arrived while the rest is still on its way. */
int gx_push_events(gx_ctx* ctx, const gx_event* events, size_t n);

/* The same for a caller whose buffer is page-locked (hipHostMalloc / hipHostRegister) and stays untouched
 * until gx_sample_end: the upload reads it in place, no staging copy. */
int gx_push_events_pinned(gx_ctx* ctx, const gx_event* events, size_t n);

/* Events already resident in device memory (HIP path only; the pointer must stay
 * valid until gx_sample_end). */
int gx_push_events_device(gx_ctx* ctx, const gx_event* d_events, size_t n);

/* The same event in 8 bytes (round 6): what saveInterval receives (Genrich.c:2516-2519) is a start, a length that a read pair
 * bounds, one of eight weights and a chromosome -- 16 bytes of gx_event are the step's largest stream (0.8 GB for 50 M
 * fragments) and twice what PCIe has to carry.
 *   start : as gx_event's
 *   lcc   : [15:0] end - start (< 65535);  [18:16] count class 0..7 = count 1, 2, 3, 4, 5, 6, 8, 10;  [31:19] chromosome (< 8192)
 * An event that does not fit (65,535 bases or more, an end before its start, a chromosome index of 8192 or more) travels as a
 * gx_event through gx_push_events; the two calls may be mixed within a sample in any order (the pileup is a sum; only the
 * replay of the reference's int16 decisions, gx_filter_saturation, looks at the order, and takes the pushes as they came).
 * gx_event8_pack: 1 and *out when `in` fits, 0 otherwise (host helper, no context).
 * where: GX_EVENTS_HOST (pageable memory, consumed before return), GX_EVENTS_PINNED (page-locked, untouched until
 * gx_sample_end), GX_EVENTS_DEVICE (device memory, valid until gx_sample_end; a 16-byte aligned buffer is read in place). */
typedef struct gx_event8 {
  uint32_t start;
  uint32_t lcc;
} gx_event8;
#define GX_EVENTS_HOST 0
#define GX_EVENTS_PINNED 1
#define GX_EVENTS_DEVICE 2
int gx_event8_pack(const gx_event* in, gx_event8* out);
int gx_push_events_packed(gx_ctx* ctx, const gx_event8* events, size_t n, int where);

/* The reference's int16 saturation rule (Genrich.c:2558-2573): saveInterval drops an alignment
 * when the int16 part of diff[start] already holds INT16_MAX or that of diff[end] INT16_MIN, which
 * depends on the order of the alignments.  keep[i] = 0 for the events (in input order, as for
 * gx_push_events; lengths as for gx_set_chroms) that it would drop, 1 otherwise.  Host-only: no
 * context, no device.  gx_sample_end applies the same rule itself when the device finds a base
 * that can saturate at all.  Returns the number of events dropped, or a negative gx_status. */
long long gx_filter_saturation(const gx_event* events, size_t n, int n_chrom, const uint32_t* len, uint8_t* keep);

/* The OPEN sample's difference array on [pos0, pos0 + n) of one chromosome, from the events pushed so far (between
 * gx_sample_begin and gx_sample_end): net[i] = weight, in 1/120 units, of the events that start at pos0 + i minus the
 * weight of those that end there -- the exact value of saveInterval's diff[pos0 + i] (Genrich.c:2576-2583; an end beyond
 * the chromosome counts at its length, 2536-2544, so pos0 + i may be the length itself; events that gx_sample_end would
 * reject are left out).  For a caller that needs saveInterval's int16 decisions READ BY READ -- the -v warnings
 * "skipped due to overflow / underflow", the missing -b line, the length 0 towards the -x average (2558-2573) --
 * and not only their effect on the pileup (which gx_sample_end reproduces by itself): it counts starts and ends per
 * window as it pushes, asks for a window's exact state when one comes near 32,767, and keeps that window itself from
 * then on (genrich_amd/host/genrich_amd.cpp: HotWindows).  One pass over the pushed events; waits for their uploads.
 * n <= 65536. */
int gx_window_net(gx_ctx* ctx, uint32_t chrom, uint32_t pos0, uint32_t n, long long* net);

/* PCR duplicates (-r): the membership half of findDupsPr / findDupsDc / findDupsSn (Genrich.c:3616-3690, 3761-3880, 3886-3944;
 * a discordant combination is looked up in both orders of its ends and stored in one -- the table is keyed on the unordered
 * pair, so the host hands the two ends over in a canonical order; the tables'
 * keys are the fields jenkins_hash_aln hashes, 3408-3450, packed by the host into four words -- an alignment-type tag
 * with the chromosome(s), the 5' end(s), the strand(s)).  keys[0..n) are the alignments of a file's sets in the order
 * in which findDups visits them (highest quality sum first; anything added to a table unconditionally -- the ends of
 * kept pairs in the singleton table, checkAndAdd 3514 -- comes as a record like any other); multi[i] != 0 when record
 * i belongs to a set with several alignments.  owner[i] = index of the FIRST record with record i's key (i itself: the
 * key was free), with bit 31 set when some record with that key belongs to a multi-alignment set: a set of one
 * alignment whose owner word has no bit 31 is a duplicate exactly when owner[i] != i, and of the set owner[i] belongs
 * to; everything that carries bit 31 is left to the caller, who walks those few sets in order as the reference does.
 * n < 2^31.  Uses the context's device and stream; independent of the sample state. */
typedef struct { uint32_t w[4]; } gx_dup_key;
int gx_dups_first(gx_ctx* ctx, const gx_dup_key* keys, const uint8_t* multi, size_t n, uint32_t* owner);

/* Treatment: == savePileupExpt (Genrich.c:2168-2295), returns fragLen.
 * Control : == savePileupCtrl (:2052-2161), returns lambda and factor.
 * Any out pointer may be NULL. */
int gx_sample_end(gx_ctx* ctx, double* frag_len, float* lambda, float* factor);

/* How many alignments of the sample just closed (gx_sample_end) the reference's int16 saturation rule dropped
 * (Genrich.c:2558-2573; normally 0).  The library has dropped them as the reference does; a host program that
 * printed their -b lines or counted their lengths beforehand can at least say so. */
int gx_saturation_dropped(gx_ctx* ctx, long long* n);

/* == savePileupNoCtrl (Genrich.c:1883-1896): missing or "null" control. */
int gx_sample_no_control(gx_ctx* ctx, float* lambda);

/* == savePval (Genrich.c:1720-1794) for the current replicate; closes the replicate. */
int gx_pvalues(gx_ctx* ctx);

/* ---- genome-wide statistics and peaks (replaces findPeaks, Genrich.c:1076-1137:
 *      combinePval 612, computeQval 352, callPeaks 977) ---- */

int gx_find_peaks(gx_ctx* ctx, size_t* n_peaks, uint64_t* genome_len, uint64_t* peak_bp);

/* Copies min(cap, n_peaks) peaks, chromosome-table order then position. */
int gx_peak_count(gx_ctx* ctx, size_t* n_peaks);
int gx_get_peaks(gx_ctx* ctx, gx_peak* out, size_t cap);

/* Interval arrays for host-side -f / -k formatting (printLog 808, printPile 1697).
 * which: replicate index r (0..n-1) = that replicate's p-value intervals,
 *        GX_IV_FINAL = the intervals peaks were called on (combined when n > 1).
 * Arrays (any may be NULL) receive n_iv entries: end[], expt[] and ctrl[] pileups
 * (only meaningful for a single replicate, as in the reference), p[], q[].
 * q (with -q, GX_IV_FINAL only; saveQval 212-250): gx_find_peaks looks q up where updatePeak reads it -- inside the candidate
 * peaks -- and the whole array is made when it is first asked for here, from the run's {p -> q} table, which stays on the device
 * until the next gx_find_peaks of the context takes it (then: GX_ERR_ORDER for an array that was never asked for). */
#define GX_IV_FINAL (-1)
int gx_interval_count(gx_ctx* ctx, int which, int chrom, size_t* n_iv);
int gx_total_intervals(gx_ctx* ctx, int which, size_t* n_iv); /* over all chromosomes */
int gx_get_intervals(gx_ctx* ctx, int which, int chrom, size_t cap, uint32_t* end,
                     float* expt, float* ctrl, float* p, float* q);

/* ---- host-side text emitters of the drop-in surface (gx_emit.cpp); byte format of the
 *      reference's printf calls.  names[i] = chromosome names in table order. ---- */
#include <stdio.h>
/* -o  ENCODE narrowPeak: printPeak, Genrich.c:885-909 */
int gx_write_narrowpeak(gx_ctx* ctx, const char* const* names, FILE* out);
/* -k  pileup log of replicate rep: printPileHeader 1680-1691, printPile 1697-1715 */
int gx_write_pile(gx_ctx* ctx, int rep, const char* const* names, int n_chrom, const char* expt_name,
                  const char* ctrl_name, FILE* out);
/* -f  bedgraph-ish log: printLogHeader 674-717, printInterval 770-803, printIntervalN 724-763;
 *     peaks_opt = 0 is -X (logIntervals 837-878) */
int gx_write_log(gx_ctx* ctx, int n_rep, const char* const* names, int n_chrom, int qval_opt, int peaks_opt,
                 float thr, FILE* out);
/* The same for a run whose chromosomes are sharded over several contexts (one per GPU): peaks of all
 * contexts in chromosome-table order (peak_N numbering, Genrich.c:986, 925); owner[c] = index into ctxs of
 * the context that computed chromosome c. */
int gx_write_narrowpeak_group(gx_ctx* const* ctxs, int n_ctx, const char* const* names, FILE* out);
int gx_write_pile_group(gx_ctx* const* ctxs, const int* owner, int rep, const char* const* names, int n_chrom,
                        const char* expt_name, const char* ctrl_name, FILE* out);
int gx_write_log_group(gx_ctx* const* ctxs, const int* owner, int n_rep, const char* const* names, int n_chrom,
                       int qval_opt, int peaks_opt, float thr, FILE* out);
int gx_write_narrowpeak_path(gx_ctx* ctx, const char* const* names, const char* path);
int gx_write_pile_path(gx_ctx* ctx, int rep, const char* const* names, int n_chrom, const char* expt_name,
                       const char* ctrl_name, const char* path, int append);
int gx_write_log_path(gx_ctx* ctx, int n_rep, const char* const* names, int n_chrom, int qval_opt, int peaks_opt,
                      float thr, const char* path);

/* ---- multi-GPU hooks: chromosomes are sharded across ranks; the
 *      three genome-wide quantities are exchanged through host-supplied callbacks
 *      (RCCL via torch.distributed in bench.py; a no-op on one GPU). ---- */

/* Sum `n` int64 values over all ranks in place.  EVERY exchange of the library goes through this one callback in the
 * callback mode: the fragLen / ctrlFrag fixed-point parts and the ranks' flags (n = 3, twice per sample when lambda is
 * exchanged ahead of the tile stage), the dense p-value histogram of a run without a control (n = 2^18 + 8,194 per
 * rank), and -- as sums of disjoint regions of a zeroed buffer, i.e. concatenations -- the samples, counts, totals,
 * minima and the all-to-all segments of the range-partitioned BH exchange (gx_bhx.h).  buf is host memory. */
typedef int (*gx_allreduce_i64_fn)(int64_t* buf, size_t n, void* user);
/* (Rounds 1-5 also took a table all-gather callback here; the BH table travels by all-reduces since round 4, and round 6
 * dropped the parameter.) */
int gx_set_collectives(gx_ctx* ctx, int rank, int world, gx_allreduce_i64_fn allreduce, void* user);
/* The library's own collectives: RCCL over xGMI on device buffers, on the library's stream (no host
 * hop in the data path).  One rank calls gx_rccl_unique_id and hands the 128 bytes to every rank
 * by whatever channel the host program has; then every rank calls gx_set_rccl (collective: it
 * returns when all `world` ranks have called it).  Replaces gx_set_collectives' callbacks, which stay
 * for host programs without RCCL (the tests' gloo mode).  librccl is opened at run time. */
int gx_rccl_unique_id(void* out, size_t cap /* >= 128 */);
int gx_set_rccl(gx_ctx* ctx, int rank, int world, const void* unique_id);
/* Ranks of the library's communicator as RCCL itself reports them (ncclCommCount); 0 without one. */
int gx_rccl_nranks(gx_ctx* ctx, int* n);
/* owned[i] = 1: this rank computes chromosome i (default: all).  The full table still goes to
 * gx_set_chroms on every rank, so genome lengths and output order are global; device work and
 * memory are laid out for the owned chromosomes only.  Call after gx_set_chroms and before the
 * first gx_sample_begin (or right after gx_reset): GX_ERR_ORDER otherwise. */
int gx_set_owned(gx_ctx* ctx, const uint8_t* owned);

/* keep = 0: the treatment / control pileup floats of the p-value intervals are not written to
 * device memory (they are only ever read by gx_get_intervals, i.e. by the -f / -k emitters, which
 * then fail with GX_ERR_ORDER); interval ends, p and q are unaffected.  Default 1.  The reference
 * always keeps them (Pileup arrays, Genrich.h:208-214) and prints them only with -f / -k. */
int gx_set_keep_pileups(gx_ctx* ctx, int keep);

/* ---- introspection used by bench.py / tests ---- */

/* Per-phase device times (ms, HIP events on the library's stream) of the last
 * gx_* call sequence; names are NUL-separated in *names. Returns count.
 * gx_set_phase_timing chooses what is timed: 0 nothing (default: an event record costs a
 * ~5 us bubble on the stream), 1 the tile stage only ("t.tile" / "c.tile": what
 * bench.py's roofline needs inside its timed region), 2 every phase.
 * Independently of the level, a context made with GX_ROCTX=1 in the environment (or gx_set_knob) brackets every phase
 * with a roctx range of the phase's name ("gx:t.tile", ...) on the calling thread, so that a rocprofv3 --marker-trace
 * --kernel-trace run attributes every kernel to its phase (tools/make_counters_json.py; host-side markers, no stream
 * bubble; the roctx library is opened at run time and its absence is not an error). */
int gx_set_phase_timing(gx_ctx* ctx, int level);
/* Like level 1, for another phase: only the phases called `name` ("sort1", "tile", "bucket" -- per sample, reported
 * as "t.<name>" / "c.<name>" --, "pval", "merge", "fisher", "bh", "sweep") are bracketed by events.  bench.py times
 * the phase of its roofline kernel this way inside the timed region. */
int gx_set_phase_filter(gx_ctx* ctx, const char* name);
int gx_phase_times(gx_ctx* ctx, const char** names, const float** ms);
/* A hint, before the first sample: the run may hold fractional weights (count > 1: Genrich's -s).  The device path then
 * writes its pair records with a weight class from the start; without the hint the first sample that shows a fractional
 * weight is built a second time, on the general chain (same results either way). */
int gx_expect_fractional(gx_ctx* ctx, int on);
/* Test and measurement switches (the GX_* names of DESIGN.md's knob table: GX_NO_FUSED, GX_NO_LOOSE, GX_SBSHIFT, ...).  The
 * library reads them from the environment once, in gx_create; this sets one on a live context (bench.py times the
 * "materialised" step that way).  value: a number as text; NULL or "" = 1.  An unknown name is GX_ERR_ORDER.  Not part of
 * what a host program needs: every switch only forces a path that the default run chooses by itself. */
int gx_set_knob(gx_ctx* ctx, const char* name, const char* value);
/* Which device path the last calls took (tests assert that the fast paths really run):
 * bit 0: the last sample's tile stage was k_sbtile (level 2 of the sort fused with the tile passes, gx_sbtile.h);
 * bit 1: the last gx_find_peaks swept the tile stage's loose slots (no k_pack_pval, gx_kernels.h LooseCtl);
 * bit 2: a sample of this context was sent back to the general chain (a super-bucket beyond k_sbtile's LDS, or
 *        fractional weights);
 * bit 3: a sample of this context was built again with larger page tables (reads piled up in one super-bucket
 *        beyond what a row of the level-1 page table held). */
#define GX_PATH_FUSED 1u
#define GX_PATH_LOOSE_SWEEP 2u
#define GX_PATH_FELL_BACK 4u
#define GX_PATH_PT_GREW 8u
#define GX_PATH_PAIRS 16u    /* bit 4: level 1 of the sort wrote one record per fragment (k_sort_a / k_sort_b) for the fused kernel */
#define GX_PATH_RANGE_BH 64u /* bit 6: several ranks with a control / replicates, -q: the range-partitioned BH exchange */
#define GX_PATH_DENSE_BH 32u /* bit 5: several ranks, no control, -q: the p-value histogram travelled as ONE dense all-reduce */
#define GX_PATH_PILES_MADE 256u /* bit 8: pileup floats (Pileup.cov, printed by -f / -k only) were written since the last gx_reset */
#define GX_PATH_PACKED 512u  /* bit 9: the last sample's level 1 read 8-byte events in place (k_sort_a<.., PACKED>: gx_push_events_packed) */
#define GX_PATH_PACK_HIST 2048u /* bit 11: -q on one replicate without control: BH's table was made of the "bp at pileup V" sums that the tight table's kernel left (k_pack_pval<.., HIST>), not by k_bh_hist */
#define GX_PATH_LAZY_Q 8192u /* bit 13: -q: the sweep's significance bits came from one compare per interval against the smallest significant p, q was looked up inside the candidates only (k_sig_from_p + k_q_fill_cands; the whole q array on request) */
#define GX_PATH_LATE_LOOSE 16384u /* bit 14: ... GX_PATH_LOOSE_SWEEP on a sample whose lambda came with its end (fractional weights): the bits and the fillers were written after the tile stage (k_loose_late) */
#define GX_PATH_Q_LOOSE 32768u /* bit 15: -q on one replicate without control: no tight table -- BH's histogram from the loose slots, q by pileup, the sweep on the loose slots (k_bh_small, k_loose_late, k_peak_both<.., PVQ>) */
#define GX_PATH_MERGE_P 1024u /* bit 10: the last control merge scored its intervals itself and left (end, p) in its loose slots (k_merge2<.., true> + k_pack_ep2) */
#define GX_PATH_FRAC_PAIRS 128u /* bit 7: ... and the pair records carried a weight class (k_sort_a<FRAC> / k_sbtile<.., FRAC>: -s multimapping) */
int gx_path_info(gx_ctx* ctx, unsigned* flags);

/* Evaluate one scalar device function on n inputs (numerics tests):
 * what 0: log10f as the host libm computes it (saveQval 221/226)   out = f(a)
 *      1: calcPval(expt = a, ctrl = b)          (Genrich.c:1628)
 *      2: getVal of the exact pileup whose int32 bits are in a (1/120 units, :1902)
 *      3: multPval's combination of sum = a over df = b (567-583), by the reference's own algorithm (pgamma's series)
 *      4: the same by the closed form of the even-df tail that the merge kernels evaluate (gx_math.h fisher_fast); a value next
 *         to a float rounding boundary is re-evaluated by 3's algorithm on the host, like everywhere */
int gx_selftest(gx_ctx* ctx, int what, const float* a, const float* b, float* out, size_t n);
/* The same, plus (what 1 and 3) the double each result was rounded from in out_double (may be NULL)
 * and the number of results that lay next to a float rounding boundary and were therefore
 * re-evaluated with the host's libm ("risky", gx_math.h) in *n_risky (may be NULL). */
int gx_selftest2(gx_ctx* ctx, int what, const float* a, const float* b, float* out, double* out_double,
                 size_t n, size_t* n_risky);
/* what 1, 3 and 4 evaluated by the host build of the same routines (this machine's libm, as the
 * reference would call it); no context and no device needed. */
int gx_selftest_host(int what, const float* a, const float* b, float* out, double* out_double, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* GENRICH_AMD_H */
