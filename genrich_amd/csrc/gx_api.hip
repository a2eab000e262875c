// gx_api.hip -- C ABI (include/genrich_amd.h) over the HIP kernels.  Host code here only
// sizes buffers, launches kernels on one stream and moves scalars; every per-base /
// per-interval computation of the hot path runs on the device.  There is no CPU fallback:
// a missing device or a failed launch is reported as GX_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>
#include <unistd.h>

#include "gx_merge.h"
#include "gx_rccl.h"
#include "gx_sort.h"
#include "gx_tile_fast.h"
#include "gx_sbtile.h"
#include "gx_dups.h"
#include "gx_bhx.h"
#include "gx_saturate.h"

using namespace gx;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), own(o.own) { o.p = nullptr; o.cap = 0; o.own = true; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; own = o.own; o.p = nullptr; o.cap = 0; o.own = true; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p && own) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    own = true;
  }
  void view(void* ptr, size_t bytes) {  // non-owning window into another allocation
    release();
    p = ptr;
    cap = bytes;
    own = false;
  }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
  bool own = true;
};

// Small device -> host read-backs land in one pinned block: a copy into pageable memory is staged and
// blocks the caller, a copy into pinned memory is an ordinary asynchronous packet.
struct HostMail {
  Scalars scal;
  long long acc[2];
  uint64_t peakBP, genome;
  u32 nF, nIv, status, R, nPeaks, nMerged, D, n, hot, bhOvf;
  long long coll[4];   // this rank's / all ranks' {fragLen parts, saturation flag}
  u32 counts[64];      // BH records per rank (all-gather)
  u32 closeState;      // k_close: 1 the sample is closed, 2 the separate kernels have to run
  u32 statusKeep;      // (host -> device: the status bits a repeated tile stage must keep)
  u32 seq;             // k_mail's last write (mail_sync polls it)
};

struct PinnedBuf {
  void* p = nullptr;
  void* dp = nullptr;  // the same memory as the device sees it (kernels write results straight into it)
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocCoherent);  // (fine-grained: the kernels write mail and peaks into it while the host polls)
    if (e == hipSuccess) {
      cap = want;
      e = hipHostGetDevicePointer(&dp, p, 0);
    }
    return e;
  }
};

struct Pileup {  // run-length pileup of one sample (treatment or control)
  DevBuf ivEnd, ivV, tileIvOff, chromIvOff;
  u32 nIv = 0;
  bool packed = false;  // ivEnd / ivV filled (otherwise the intervals still sit in the loose slots)
  // a sample that waits for its control merge where the tile stage left it (stash_loose): its loose slots and its tile
  // descriptors (slot, carry) taken out of the context, which builds the next sample into other buffers
  DevBuf looseEnd, looseV, meta;
  bool inLoose = false;
};

struct PArray {  // p-value intervals of one replicate (or the Fisher combination)
  DevBuf end, p, expt, ctrl, chromOff, tileOff, q, dPresent;
  u32 n = 0;
  bool loose = false;     // no control: the intervals still sit in the tile kernel's loose slots (ctx->looseEnd / looseV);
                          // the tight table is made when somebody asks for it (materialize_rep)
  bool looseSweep = false;    // ... and the tile stage left the sweep's significance bits for them (LooseCtl): gx_find_peaks on
                              // this replicate alone, with -p, walks the loose slots as they are
  DevBuf chromLooseOff;       // [nChrom + 1] first loose slot of each chromosome
  size_t looseStride = 0;     // words between the sig / brk masks in swMask
  bool pilesPending = false;  // no control: the pileup floats are wanted but not made yet (ensure_piles)
  // ... and what they will be made from once the context has built another sample into its loose slots: this replicate's
  // exact pileups and tile descriptors, taken out of the context (no copy; the context takes other buffers: pooled)
  DevBuf keptV, keptMeta;
  bool keptLoose = false;
  bool hasPiles = false;  // expt/ctrl filled (single-replicate logging)
  bool pilesDropped = false;  // ... deliberately not (gx_set_keep_pileups(0))
  float ctrlConst = 0.0f; // control value when ctrl is not materialised (no -E, no control file)
  bool ctrlIsConst = false;
  std::vector<uint8_t> present;  // per chromosome: p-values exist (pval[n] != NULL)
};

struct Phase {
  std::string name;
  hipEvent_t a, b;
};

// Test and measurement switches.  They are read from the environment ONCE, when the context is made (gx_create), or
// set on a live context by gx_set_knob; nothing in a build or a sweep calls getenv.  Every one of them is exercised
// by tests/ (a forced path must give the oracle's bits like the default one) or by tools/.
struct Knobs {
  int debug = 0;          // GX_DEBUG: synchronise after every launch and say which kernel it was
  int debugRetry = 0;     // GX_DEBUG_RETRY: say why a sample was built again
  int noSpin = 0;         // GX_NO_SPIN: block in the runtime instead of polling for the mail
  int forceRec64 = 0;     // GX_FORCE_REC64: 8-byte records although the genome fits 4-byte keys (only > 4.29 Gbp takes them naturally)
  int forceSlowFrag = 0;  // GX_FORCE_SLOWFRAG: the general fragLen path
  int noFused = 0;        // GX_NO_FUSED: the general chain instead of k_sbtile
  int noLoose = 0;        // GX_NO_LOOSE: lambda after the tile stage, tight table, the sweep on it
  int noPairs = 0;        // GX_NO_PAIRS: k_sort1's start / end keys for k_sbtile
  int noFracPairs = 0;    // GX_NO_FRAC_PAIRS: fractional weights take the general chain
  int forceHalfBins = 0;  // GX_FORCE_HALF_BINS: the 128-key level 1 on a small input
  int noHalfBins = 0;     // GX_NO_HALF_BINS
  int noEarlyColl = 0;    // GX_NO_EARLY_COLL: no all-reduce of the closed form of fragLen ahead of the tile stage
  int noDenseBh = 0;      // GX_NO_DENSE_BH: the range-partitioned exchange also without a control
  int qtMulti = 0;        // GX_QT_MULTI: the chunked BH table kernels for a small table
  int forceColl = 0;      // GX_FORCE_COLL: run the collectives with a single rank too
  int sbShift = -1;       // GX_SBSHIFT: tiles per super-bucket (log2)
  long long runCapMin = 0;  // GX_RUN_CAP_MIN: the sweep's first guess of the run count (a tiny one forces the second pass)
  int bhCapLog = 0;       // GX_BH_CAPLOG: log2 of the BH table's first size
  int ptJmax = 0;         // GX_PT_JMAX: pages per level-1 list at first
  int fault = 0;          // GX_FAULT: fault injection for the tests of the device-side invariants.  1: the weight of the ends at
                          // chromosome 0's length is damaged behind level 1 of the sort (as if an end record had been lost)
};
struct KnobDef { const char* name; int Knobs::*i; long long Knobs::*ll; };
const KnobDef KNOBS[] = {
    {"GX_DEBUG", &Knobs::debug, nullptr}, {"GX_DEBUG_RETRY", &Knobs::debugRetry, nullptr}, {"GX_NO_SPIN", &Knobs::noSpin, nullptr},
    {"GX_FORCE_REC64", &Knobs::forceRec64, nullptr}, {"GX_FORCE_SLOWFRAG", &Knobs::forceSlowFrag, nullptr},
    {"GX_NO_FUSED", &Knobs::noFused, nullptr}, {"GX_NO_LOOSE", &Knobs::noLoose, nullptr}, {"GX_NO_PAIRS", &Knobs::noPairs, nullptr},
    {"GX_NO_FRAC_PAIRS", &Knobs::noFracPairs, nullptr}, {"GX_FORCE_HALF_BINS", &Knobs::forceHalfBins, nullptr},
    {"GX_NO_HALF_BINS", &Knobs::noHalfBins, nullptr}, {"GX_NO_EARLY_COLL", &Knobs::noEarlyColl, nullptr},
    {"GX_NO_DENSE_BH", &Knobs::noDenseBh, nullptr}, {"GX_QT_MULTI", &Knobs::qtMulti, nullptr}, {"GX_FORCE_COLL", &Knobs::forceColl, nullptr},
    {"GX_SBSHIFT", &Knobs::sbShift, nullptr}, {"GX_RUN_CAP_MIN", nullptr, &Knobs::runCapMin}, {"GX_BH_CAPLOG", &Knobs::bhCapLog, nullptr},
    {"GX_PT_JMAX", &Knobs::ptJmax, nullptr}, {"GX_FAULT", &Knobs::fault, nullptr},
};
// a switch that is merely present counts as 1 (GX_NO_LOOSE= is "on", as it was with getenv() != nullptr)
bool set_knob(Knobs& k, const char* name, const char* value) {
  for (const KnobDef& d : KNOBS)
    if (!strcmp(d.name, name)) {
      const long long v = value && *value ? atoll(value) : 1;
      if (d.i) k.*(d.i) = (int)v; else k.*(d.ll) = v;
      return true;
    }
  return false;
}

}  // namespace

struct gx_ctx {
  gx_params par{};
  Knobs knob;
  int device = 0;
  hipStream_t stream = nullptr;
  bool keepPiles = true;        // materialise the pileup floats of the p-value intervals
  int maskIdx = -1;             // reps[] entry whose sig / skip masks sit in swMask (k_pack_pval)
  u32 maskN = 0;
  size_t maskStride = 0;        // words between the sig / skip / brk masks in swMask
  hipStream_t side = nullptr;   // small read-backs that must not stall the main stream
  hipEvent_t sideEv = nullptr;
  std::string err;
  // chromosome table
  std::vector<uint32_t> len;
  std::vector<uint8_t> skip, save, owned;
  std::vector<std::vector<uint32_t>> bed;
  std::vector<DChrom> hChrom;
  u32 nChrom = 0, nTiles = 0;
  int sbShift = 0;
  u32 nSB = 0;
  DevBuf dChrom, dTileChrom, dBedTileOff, dBedEdge, dTileSave0;
  bool hasBed = false;
  bool bedGiven = false;        // some chromosome (of any rank) has -E regions: what every rank knows alike
  size_t nBedEdges = 0;
  // per-sample state
  int phase = 0;  // 0 idle, 1 treatment open, 2 treatment done, 3 control open, 4 control done
  int sample = 0;
  // The sample's events, in push order: device-resident segments of the caller (gx_push_events_device) and
  // pieces of the library's own device chunks, filled from host memory by asynchronous copies on `side`
  // (`ready` = the copy has arrived: the main stream waits for it before the kernel that reads the piece).
  struct Seg { const gx_event* p; size_t n; hipEvent_t ready; };
  std::vector<Seg> segs;
  std::vector<DevBuf> evChunks;   // device chunks of EV_CHUNK events, reused sample after sample
  size_t evChunkIdx = 0, evChunkFill = 0;
  PinnedBuf stage[2];             // pinned staging of gx_push_events (the caller's buffer is free on return)
  hipEvent_t stageFree[2] = {nullptr, nullptr};
  int stageNext = 0;
  std::vector<hipEvent_t> evPool; // `ready` events, reused
  size_t evPoolUsed = 0;
  DevBuf satBuf;                  // what gx_filter_saturation left of the sample
  struct Stream {  // one record stream of the bucket sort
    DevBuf a, pool, pt, cursor, sbOff;  // level-2 output; level-1 pages, page table, list cursors; super-bucket offsets
  };
  size_t b2LdsSet = 0;          // dynamic LDS the level-2 kernel was last configured for
  bool sbtLdsSet = false;       // ... and k_sbtile
  bool sawFrac = false;         // a sample of this context held fractional weights: k_sbtile is not tried again
  bool fusedOff = false;        // this sample: a super-bucket did not fit k_sbtile (the general chain runs instead)
  bool fusedUsed = false;       // the last build went through k_sbtile
  bool pairsUsed = false;       // ... on level 1's pair records (k_sort_a / k_sort_b)
  bool fracPairsUsed = false;   // ... with a weight class per record (fractional weights)
  bool earlyColl = false;       // this build: the ranks exchange the closed form of fragLen ahead of the tile stage
  bool earlyOwed = false;       // ... and this rank has not taken part in that all-reduce yet (poison_allreduce)
  bool earlyPending = false;
  int fusedBackoff[2] = {0, 0}; // treatment / control samples for which k_sbtile is not tried (after one that did not fit)
  bool looseSwept = false;      // the last gx_find_peaks swept the loose slots
  bool pilesMade = false;       // pileup floats were written since the last gx_reset (ensure_piles)
  DevBuf lbSweep, lbSweep2;     // look-back granules of the sweep's one-pass compactions (generation-tagged)
  u32 sweepGen = 0;
  FragSelect closeSel{};        // the sample's k_frag_select arguments (k_close took them; finish_scalars may need them again)
  u32 closeSeq = 0;             // sequence number of the mail k_close sends (0: the separate kernels were launched)
  bool beginPending = false;    // gx_sample_begin's clearing of the scalars is still to be done (k_build_init / flush_begin)
  u64 beginGenome = 0;
  bool fellBack = false;        // some sample was sent back from k_sbtile to the general chain
  bool ptGrew = false;          // some sample was built again with larger page tables (RETRY_PT)
  bool fragFused = false;       // the last build's tile kernel adds the general fragLen path's terms itself (TileIn::fragAcc)
  bool looseOk = false;         // the treatment sample's tile stage left valid sweep bits on the loose slots
  bool riskNearThr = false;     // a re-evaluated table entry lies next to the significance threshold
  size_t looseStride = 0;       // words between the sig / brk masks the tile stage wrote into swMask
  DevBuf tileSlot, chromW0, chromLooseOff, looseCtl;
  u32 ptJmax = 16;              // pages per (XCD class, super-bucket) list; grown after ST_PT_FULL
  DevBuf lbIv;
  Stream str[3];  // S (start keys), E (end keys), F (fractional records)
  DevBuf tileCnt[3], tileOff[3];
  DevBuf looseC, pairLogE, pairCtab, pairP2d, fragSum, tileDeep, fragList, zeroArena, endAtLen, binNet, curC, ptC, poolC, auxC, nWide, wideList, heavyList;
  DevBuf tileMeta, tileWsum, tileCarry, lb, misc, dScal, dStatus, looseEnd, looseV, tileIvCount, tileLastEnd, tilePrevEnd;
  Pileup expt, ctrl;
  Scalars hScal{};  // host copy of the device scalars (refreshed from the mail block)
  std::vector<PArray> reps;
  int finalIdx = -1;
  // BH
  DevBuf pvLut, dRisk, dDeep;
  PinnedBuf riskHost;           // count + records of the risky p-values, as read back / as sent with the host's values
  bool pairTabsReady = false;   // the control's p-value tables were built when its sample was closed
  DevBuf fisherCache;  // k_mergeN's device-wide table of (sum, df) -> p
  DevBuf bhKeys, bhLens, bhOutKeys, bhOutSlot, bhSortKeys, bhSortSlot, bhQ, bhRaw, bhDl, bhTmp, bhRecs;
  PinnedBuf hostRecs;           // this rank's BH records for the all-gather
  bool satDone = false;         // this sample's events already went through the saturation filter
  long long satDropped = 0;     // ... which dropped this many of them (gx_saturation_dropped)
  bool bhDirty = false;         // the BH table was left with entries (an error path): wipe it before use
  u32 bhCapLog = 22;            // log2 of its slots (grows by 3 after ST_HASH_FULL)
  // sweep
  DevBuf swStart, swEnd, swMask, cand, valid, peaks, headPos, candHdr, longList;
  PinnedBuf hPeaks;             // the peak list on the host (pinned: the read-back is asynchronous)
  size_t nHostPeaks = 0;
  u32* nIvTarget = nullptr;
  PinnedBuf mailBuf;
  HostMail* mail = nullptr;
  uint64_t genomeLenUsed = 0, peakBP = 0;
  // collectives
  int rank = 0, world = 1;
  gx_allreduce_i64_fn allreduce = nullptr;
  gx_allgather_tab_fn allgather = nullptr;
  void* user = nullptr;
  ncclComm_t comm = nullptr;    // the library's own collectives (gx_set_rccl): RCCL on device buffers, on `stream`
  bool forceColl = false;       // GX_FORCE_COLL=1: run the collectives with a single rank too (tests)
  DevBuf bigBins;               // pair mode: the super-buckets k_sbtile's first launch leaves to its second
  DevBuf dColl, dCounts, dGather, bhDense, bhxSmall, bhxRecv, bhxKeys, bhxLens, bhxQ, bhxOut, bhxAns;
  bool rangeBhUsed = false;     // the last gx_find_peaks took the range-partitioned BH exchange
  bool denseBhUsed = false;     // the last gx_find_peaks exchanged the p-value histogram as one dense all-reduce
  int phaseLevel = 0;       // gx_set_phase_timing
  std::string phaseFilter = "tile";  // level 1: the one phase that is timed (gx_set_phase_filter)
  u32 mailSeq = 0;          // mail_sync: the sequence number the next k_mail writes
  u32 statusSeen = 1;       // status bits read back since the device word was last cleared (1: not cleared yet)
  u64 runCap = 0, runSeen = 0;  // run_sweep: runs its arrays are sized for; runs of the last sweep
  bool phaseOpen = false;
  int numCU = 0, resTile = 0, resTileHalf = 0, resTileFast = 0, resSweep = 0;  // co-resident workgroups per kernel class
  // recycled device buffers (gx_reset keeps allocations alive across runs)
  std::vector<DevBuf> pool;
  // timing
  std::vector<Phase> phases;
  size_t nPhases = 0;
  std::vector<float> phaseMs;
  std::string phaseNames;
};

#define HIPCHECK(x)                                                                      \
  do {                                                                                   \
    hipError_t e__ = (x);                                                                \
    if (e__ != hipSuccess) {                                                             \
      ctx->err = std::string(#x) + ": " + hipGetErrorString(e__);                        \
      return GX_ERR_DEVICE;                                                              \
    }                                                                                    \
  } while (0)

namespace {

// buffer of at least `bytes`, recycled from the context's pool when possible
hipError_t pooled(gx_ctx* ctx, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes) return hipSuccess;
  if (b.p) ctx->pool.push_back(std::move(b));
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); i++)
    if (ctx->pool[i].cap >= bytes && (best < 0 || ctx->pool[i].cap < ctx->pool[best].cap)) best = i;
  if (best >= 0) {
    b = std::move(ctx->pool[best]);
    ctx->pool.erase(ctx->pool.begin() + best);
    return hipSuccess;
  }
  return b.ensure(bytes);
}
void recycle(gx_ctx* ctx, DevBuf& b) {
  if (b.p) ctx->pool.push_back(std::move(b));
}

// misc device words (u32 indices into ctx->misc)
enum { M_TICKET = 0, M_NIV = 1, M_BHCOUNT = 5, M_ALLONE = 6, M_BHOVF = 7, M_GENOME = 10 /* u64 */, M_NMERGED = 15,
       // the sweep's counters are contiguous: one memset clears them
       M_TICKET2 = 16, M_SWCOUNT = 17, M_NPEAKS = 18, M_TICKET3 = 19, M_TICKET4 = 20, M_NHEADS = 21, M_PEAKBP = 22 /* u64 */,
       M_SWEEP_FIRST = 16, M_SWEEP_WORDS = 8, M_WORDS = 32 };

// GX_DEBUG=1: synchronise after every launch and say which kernel it was (hang / fault triage)
int dbg_sync(gx_ctx* ctx, const char* what) {
  if (!ctx->knob.debug) return GX_OK;
  fprintf(stderr, "[gx] %s ...", what);
  fflush(stderr);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  fflush(stderr);
  if (e != hipSuccess) {
    ctx->err = std::string(what) + ": " + hipGetErrorString(e);
    return GX_ERR_DEVICE;
  }
  return GX_OK;
}

// phase timers: the event pairs are created once and reused run after run
// (an event record costs a ~5 us bubble on the stream: gx_set_phase_timing chooses none / the tile stage / all)
static bool phase_wanted(const gx_ctx* ctx, const char* name) {
  if (ctx->phaseLevel >= 2) return true;
  if (ctx->phaseLevel != 1) return false;
  const char* base = name[0] && name[1] == '.' ? name + 2 : name;  // "t.tile" / "c.tile" -> "tile"
  return ctx->phaseFilter == base;
}
void phase_begin(gx_ctx* ctx, const char* name) {
  ctx->phaseOpen = phase_wanted(ctx, name);
  if (!ctx->phaseOpen) return;
  if (ctx->nPhases == ctx->phases.size()) {
    Phase ph;
    (void)hipEventCreate(&ph.a);
    (void)hipEventCreate(&ph.b);
    ctx->phases.push_back(ph);
  }
  Phase& ph = ctx->phases[ctx->nPhases++];
  ph.name = name;
  (void)hipEventRecord(ph.a, ctx->stream);
}
void phase_end(gx_ctx* ctx) {
  if (ctx->phaseOpen) (void)hipEventRecord(ctx->phases[ctx->nPhases - 1].b, ctx->stream);
  ctx->phaseOpen = false;
}

int status_to_rc(gx_ctx* ctx, u32 st) {
  ctx->statusSeen |= st;  // (gx_reset clears the device word only when something was ever raised)
  if (!st) return GX_OK;
  struct { u32 bit; int rc; const char* msg; } tab[] = {
      {ST_LOOKBACK, GX_ERR_DEVICE, "look-back / page-table spin limit reached"},
      {ST_BAD_CHROM, GX_ERR_ORDER, "event on an unknown chromosome"},
      {ST_BAD_POS, GX_ERR_POS, ": read aligned beyond reference end"},
      {ST_BAD_COUNT, GX_ERR_ALNS, "Disallowed number of alignments"},
      {ST_NEG_PILE, GX_ERR_PILE, "Invalid pileup value (< 0)"},
      {ST_NO_FRAGS, GX_ERR_EXPT, "Experimental sample has no analyzable fragments"},
      {ST_SAT16, GX_ERR_OVERFLOW, "per-base difference beyond the reference's int16 range"},
      {ST_HASH_FULL, GX_ERR_DEVICE, "p-value table full"},
      {ST_BAD_DF, GX_ERR_DF, "Invalid df in pchisq()"},
      {ST_PT_FULL, GX_ERR_MEM, "level-1 page table full"},
      {ST_END_PILE, GX_ERR_ARR, "pileup of a chromosome does not return to 0 behind its last base"},
      {ST_BH_LEN, GX_ERR_PVAL, "Genome length does not match p-value length"},
  };
  for (auto& t : tab)
    if (st & t.bit) {
      ctx->err = t.msg;
      return t.rc;
    }
  ctx->err = "unknown device status";
  return GX_ERR_DEVICE;
}

int read_status(gx_ctx* ctx) {
  HIPCHECK(hipMemcpyAsync(&ctx->mail->status, ctx->dStatus.p, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return status_to_rc(ctx, ctx->mail->status);
}

// ---- risky p-values (gx_math.h round_checked; gx_kernels.h RiskBuf) --------------------------------
// mail_sync: at a synchronisation the host needs anyway, the list's count and first records come along.
// risk_apply: after it, evaluate the listed values with the host's libm and send them back (k_risk_apply).
// One small kernel writes everything the host wants to know into pinned memory (scalars, status, flags, the
// risky list's count and first records), then the stream is synchronised.  Null pointers: not wanted.
// The host does not wait in hipStreamSynchronize (an interrupt and a wake-up: 20-30 us after the kernel): k_mail
// writes a sequence number behind everything else and the host polls that word in pinned memory (a few us).  After
// 20 ms of polling -- or with GX_NO_SPIN -- it blocks in the runtime after all, which also reports a device fault.
MailOut mail_out(gx_ctx* ctx) {
  HostMail* dm = static_cast<HostMail*>(ctx->mailBuf.dp);
  return MailOut{&dm->scal, &dm->status, &dm->hot, &dm->nIv, &dm->coll[2], &dm->nMerged, reinterpret_cast<u64*>(&dm->peakBP),
                 static_cast<RiskBuf*>(ctx->riskHost.dp), &dm->seq};
}

int mail_wait(gx_ctx* ctx, u32 seq);

int mail_sync(gx_ctx* ctx, const Scalars* ds, const u32* hot, const u32* nIv, const long long* coll, const u32* extra,
              const u64* extra64 = nullptr) {
  const u32 seq = ++ctx->mailSeq;
  hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, ctx->stream, ds, ctx->dStatus.as<u32>(), hot, nIv, coll, extra,
                     ctx->dRisk.as<RiskBuf>(), mail_out(ctx), seq, extra64);
  HIPCHECK(hipGetLastError());  // (a launch that failed is reported now, not after the polling gives up)
  return mail_wait(ctx, seq);
}

// (the mail kernel -- k_mail, or k_close -- has been launched with this sequence number)
int mail_wait(gx_ctx* ctx, u32 seq) {
  const bool spin = !ctx->knob.noSpin;
  volatile u32* word = &ctx->mail->seq;
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    for (u32 it = 0; *word != seq; it++) {
      __builtin_ia32_pause();
      if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
  }
  if (*word != seq) HIPCHECK(hipStreamSynchronize(ctx->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  return GX_OK;
}

struct RiskHostIn { const float* a; const float* b; };  // RK_SELF: the caller's inputs

float risk_host_value(const gx_ctx* ctx, const RiskRec& r, const RiskHostIn& in) {
  const float lambda = ctx->hScal.lambda, factor = ctx->hScal.factor;
  bool ng = false, rk = false;
  switch (r.kind) {
    case RK_LUT:
    case RK_DEEP: return calc_pval(getval((int)r.a, &ng), lambda, &rk);  // no control: the control value is lambda
    case RK_TAB2D:
      return calc_pval((float)(r.a / PT_N), ctrl_net((int)((r.a % PT_N) * GX_UNIT), factor, lambda, &ng), &rk);
    case RK_PAIR: return calc_pval(expt_val((int)r.b, &ng), ctrl_net((int)r.c, factor, lambda, &ng), &rk);
    case RK_FISHER: return fisher_combine(r.x, (int)r.c, &rk);
    case RK_SELF:
      if (r.b == 1) return calc_pval(in.a[r.a], in.b[r.a], &rk);
      if (r.b == 3) return fisher_combine((double)in.a[r.a], (int)in.b[r.a], &rk);
      return 0.0f;
    default: return 0.0f;
  }
}

int risk_apply(gx_ctx* ctx, RiskTargets T, RiskHostIn in = RiskHostIn{nullptr, nullptr}) {
  RiskBuf* hb = static_cast<RiskBuf*>(ctx->riskHost.p);
  const u32 n = hb->count;
  if (!n) return GX_OK;
  hipStream_t s = ctx->stream;
  if (n > RISK_CAP) {
    HIPCHECK(hipMemsetAsync(ctx->dRisk.p, 0, 4, s));
    ctx->err = "more p-values next to a float rounding boundary than the list holds";
    return GX_ERR_DEVICE;
  }
  if (n > RISK_PREFIX) {  // (rare: the count and the first records came with the synchronisation already paid for)
    HIPCHECK(hipMemcpyAsync(hb->rec + RISK_PREFIX, ctx->dRisk.as<RiskBuf>()->rec + RISK_PREFIX,
                            (size_t)(n - RISK_PREFIX) * sizeof(RiskRec), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
  }
  for (u32 i = 0; i < n; i++) {
    const float pn = risk_host_value(ctx, hb->rec[i], in);
    hb->rec[i].pnew = pn;
    // (the tile stage compared the device's table entry, one float away at most, with the threshold: LooseCtl)
    const float thr = ctx->par.thr;
    if (hb->rec[i].kind == RK_LUT && ((pn > thr) != (nextafterf(pn, -INFINITY) > thr) || (pn > thr) != (nextafterf(pn, INFINITY) > thr)))
      ctx->riskNearThr = true;
  }
  // (the pinned records stay untouched until the next mail_sync)
  // A short list is read by the kernel where it lies (mapped pinned memory): no copy launch.
  const RiskRec* src = static_cast<const RiskBuf*>(ctx->riskHost.dp)->rec;
  if (n > RISK_PREFIX) {
    HIPCHECK(hipMemcpyAsync(ctx->dRisk.as<RiskBuf>()->rec, hb->rec, (size_t)n * sizeof(RiskRec), hipMemcpyHostToDevice, s));
    src = ctx->dRisk.as<RiskBuf>()->rec;
  }
  T.lutP = ctx->pvLut.as<float>();
  T.p2d = ctx->pairP2d.as<float>();
  T.deep = ctx->dDeep.as<DeepTab>();
  hipLaunchKernelGGL(k_risk_apply, dim3(1), dim3(256), 0, s, ctx->dRisk.as<RiskBuf>(), src, n, T);
  hb->count = 0;
  return dbg_sync(ctx, "k_risk_apply");
}

uint64_t genome_len_for(const gx_ctx* ctx, const std::vector<uint8_t>& present) {
  // calcLambda 1819-1827 / findPeaks 1091-1101
  uint64_t g = 0;
  for (u32 i = 0; i < ctx->nChrom; i++)
    if (!ctx->skip[i] && present[i]) {
      g += ctx->len[i];
      for (size_t j = 0; j + 1 < ctx->bed[i].size(); j += 2) g -= ctx->bed[i][j + 1] - ctx->bed[i][j];
    }
  return g;
}

int upload_chroms(gx_ctx* ctx, bool force = true) {
  bool changed = force;
  for (u32 i = 0; i < ctx->nChrom; i++) {
    const u32 f = (ctx->skip[i] ? CH_SKIP : 0) | (ctx->save[i] ? CH_SAVE : 0) | (ctx->owned[i] ? CH_OWNED : 0);
    changed |= f != ctx->hChrom[i].flags;
    ctx->hChrom[i].flags = f;
  }
  if (!changed) return GX_OK;  // (the table on the device is this one already: no copy launch per sample)
  // (hChrom may be rewritten by the next call while this copy is in flight: pageable memory is staged by the runtime)
  HIPCHECK(hipMemcpyAsync(ctx->dChrom.p, ctx->hChrom.data(), ctx->nChrom * sizeof(DChrom), hipMemcpyHostToDevice,
                          ctx->stream));
  return GX_OK;
}

// loose slots -> tight (end, V) arrays of a pileup (only needed ahead of a control merge)
int pack_pileup(gx_ctx* ctx, Pileup& P) {
  if (P.packed) return GX_OK;
  hipStream_t s = ctx->stream;
  const u32 nTiles = ctx->nTiles;
  HIPCHECK(pooled(ctx, P.ivV, P.ivEnd.cap));
  PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), P.tileIvOff.as<u32>()};
  hipLaunchKernelGGL(k_pack, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU)))), dim3(256), 0, s, pin, nTiles,
                     P.ivEnd.as<u32>(), P.ivV.as<int>());
  if (int rc__ = dbg_sync(ctx, "k_pack")) return rc__;
  P.packed = true;
  return GX_OK;
}

// gx_sample_begin's clearing of the replicate's scalars, when no k_build_init is going to do it
int flush_begin(gx_ctx* ctx) {
  if (!ctx->beginPending) return GX_OK;
  hipLaunchKernelGGL(k_begin_sample, dim3(1), dim3(64), 0, ctx->stream, ctx->dScal.as<Scalars>(), ctx->beginGenome);
  ctx->beginPending = false;
  return GX_OK;
}

// A sample about to be merged with its control stays in its loose slots (k_merge2<true> reads them there): the
// buffers leave the context -- no copy -- and the context takes others for the next build (pooled).  With -E regions
// the merge needs tight arrays after all (gx_merge.h): pack_pileup.
int stash_or_pack(gx_ctx* ctx, Pileup& P) {
  if (ctx->hasBed) return pack_pileup(ctx, P);
  if (P.inLoose) return GX_OK;
  recycle(ctx, P.looseEnd);
  recycle(ctx, P.looseV);
  recycle(ctx, P.meta);
  P.looseEnd = std::move(ctx->looseEnd);
  P.looseV = std::move(ctx->looseV);
  P.meta = std::move(ctx->tileMeta);
  P.inLoose = true;
  return GX_OK;
}

// events -> tile-bucketed endpoint records -> run-length pileup (loose slots + offsets) and fragLen
int allreduce_words(gx_ctx* ctx, long long* d, size_t n);

// the most level-1 chunks (workgroups of k_sort_a) any XCD class gets: class = blockIdx % NXCD of each piece's launch
template <typename Segs> static u32 class_chunks(const Segs& segs) {
  u32 best = 0;
  for (u32 x = 0; x < (u32)NXCD; x++) {
    u32 c = 0;
    for (auto& sg : segs) {
      const u32 b = (u32)((sg.n + S2_CHUNK - 1) / S2_CHUNK);
      c += b / NXCD + (b % NXCD > x ? 1u : 0u);
    }
    best = std::max(best, c);
  }
  return best;
}

// reuseSort: the sample was built a moment ago and only its tile stage has to be done again on the general chain
// (k_sbtile sent it back): level 1 of the sort -- the pages, the cursors, the closed form of fragLen -- is still
// there, so k_sort1 does not run again and only what the first tile stage and the scans wrote is cleared.
int build_pileup(gx_ctx* ctx, Pileup& out, int isCtrl, bool reuseSort = false) {
  const Knobs& K = ctx->knob;
  // Several ranks: lambda needs every rank's fragLen.  Its closed form (the sum of the fragment lengths, k_sort1) is
  // known BEFORE the tile stage, so the ranks exchange that (`earlyColl`: one all-reduce of three words behind
  // k_sort1; decided by what every rank knows alike) and each of them has the table p(V) and the sweep's bits from the
  // tile stage, as a single rank has.  The all-reduce behind the tile stage (finish_scalars) still carries the exact
  // parts and the ranks' flags; a rank whose lambda came out different there falls back to k_pack_pval as before.
  // (decided ahead of everything that can fail -- the size check, the allocations: a rank that leaves this function
  // early owes the others BOTH all-reduces, and poison_allreduce reads earlyOwed to know)
  const bool multiRank = ctx->world > 1 || ctx->forceColl;
  const bool forceSlowFrag = K.forceSlowFrag != 0, noFused = K.noFused != 0, noLoose = K.noLoose != 0;
  const bool earlyColl = multiRank && !isCtrl && !ctx->par.qval_opt && !ctx->bedGiven && !noLoose && !forceSlowFrag && !K.noEarlyColl;
  ctx->earlyColl = earlyColl;
  ctx->earlyOwed = earlyColl;
  // (host-pushed events sit in the library's device chunks, device-resident segments are used in place)
  const std::vector<gx_ctx::Seg>& segs = ctx->segs;
  size_t n = 0;
  for (auto& sg : segs) n += sg.n;
  if (2 * n >= 0xFFFFFFFFull) {
    ctx->err = "too many events in one sample for 32-bit record offsets";
    return GX_ERR_MEM;
  }
  const u32 nEv = (u32)n;
  const u32 nTiles = ctx->nTiles, nSB = ctx->nSB, nChrom = ctx->nChrom;
  // tile id + offset fit a 4-byte key (GX_FORCE_REC64=1 forces the wide-record path: used by the tests,
  // since only a genome beyond 4.29 Gbp takes it naturally)
  const bool unit32 = nTiles < MAX_TILES32 && !K.forceRec64;
  hipStream_t s = ctx->stream;
  gx_ctx::Stream& SS = ctx->str[0];
  gx_ctx::Stream& SE = ctx->str[1];
  gx_ctx::Stream& SF = ctx->str[2];
  const u32 nL1base = nSB - 1;  // level-1 bins = super-buckets (records without a tile are not scattered at all)
  // ---- what the tile stage will be -------------------------------------------------------------------------
  // k_sbtile (gx_sbtile.h): level 2 of the sort fused with the tile passes -- unit weights, no -E regions, at most
  // 2^8 tiles per super-bucket, and bins that fit its LDS (a bin that does not raises ST_SB_FULL, a fractional
  // record ST_SB_FRAC: finish_scalars then has the sample built again on the general chain).
  // (a sample whose predecessor of the same kind did not fit is not even tried for a while: the same experiment's
  // next replicate, or the next run on the same data, has the same pile-ups)
  const bool backoff = !ctx->fusedOff && ctx->fusedBackoff[isCtrl ? 1 : 0] > 0;
  if (backoff) ctx->fusedBackoff[isCtrl ? 1 : 0]--;
  const bool pairsAllowed = !K.noPairs;
  // (fractional weights ride the pair records -- k_sort_a<true>, k_sbtile<.., true> -- once a sample of this context has
  // shown one; the start / end keys of the other fused variant cannot carry a weight)
  const bool fracOk = pairsAllowed && !K.noFracPairs;
  const bool fused = !backoff && unit32 && !ctx->hasBed && ctx->sbShift <= SBT_MAXSHIFT && !noFused && (!ctx->sawFrac || fracOk) &&
                     !ctx->fusedOff && !forceSlowFrag && (size_t)nEv <= (size_t)std::max(1u, nL1base) * 64000 &&
                     (size_t)2 * nEv + nTiles + 64 < ((size_t)1 << 30);  // (k_sbtile's stores use 32-bit byte offsets)
  ctx->fusedUsed = fused;
  // ... and with it level 1: one record per fragment (k_sort_a / k_sort_b) when k_sbtile will read it
  const bool pairs = fused && !reuseSort && pairsAllowed;
  const bool fracPairs = pairs && ctx->sawFrac;
  ctx->pairsUsed = pairs;
  ctx->fracPairsUsed = fracPairs;

  // A sample so dense that the average bin holds more keys than k_sbtile's key array (ATAC cut sites of a deep library)
  // takes bins of half the size -- level 1 of the pair mode reaches 64 x 128 of them -- so that a bin is one round of
  // the tile kernel again; the general chain (a later fall-back) keeps the context's own bin size.
  int sbS = ctx->sbShift;
  u32 nL1 = nL1base;
  // (not with fractional weights: measured at config 4, the tile passes with weights and the fragLen terms cost more per
  // key than the rounds of full-size bins -- 2.98 against 2.67 ms)
  const bool forceHalf = K.forceHalfBins != 0;  // (tests: the 128-key level 1 on a small input)
  if (pairs && (!fracPairs || forceHalf) && sbS > 0 &&
      (forceHalf || (size_t)2 * nEv > (size_t)std::max(1u, nL1base) * (SBT_KEYCAP - SBT_KEYCAP / 4)) &&
      ((nTiles + (1u << (sbS - 1)) - 1) >> (sbS - 1)) <= (u32)MAX_BINS_P && !K.noHalfBins) {
    sbS--;
    nL1 = (nTiles + (1u << sbS) - 1) >> sbS;
  }

  // level-2 output: 16-bit tile offsets (S, E) / whole records (F), tile-contiguous
  if (unit32) {
    HIPCHECK(SS.a.ensure((size_t)nEv * 2 + 16));
    HIPCHECK(SE.a.ensure((size_t)nEv * 2 + 16));
  }
  HIPCHECK(SF.a.ensure((size_t)nEv * 16 + 16));  // worst case: every event fractional
  // level-1 page pools (gx_sort.h): every record lands in one page of its (XCD class, bin) list
  const u32 jmax = ctx->ptJmax;
  u32 poolPages[3];
  // (page 0: sink; NXCD * nL1 fixed first pages; at most records / page-size further ones)
  poolPages[0] = poolPages[1] = (u32)(nEv >> PgCfg<u32>::SHIFT) + NXCD * nL1 + 4;
  poolPages[2] = (u32)(((size_t)2 * nEv) >> PgCfg<u64>::SHIFT) + NXCD * nL1 + 4;
  for (int q = 0; q < 3; q++) HIPCHECK(ctx->str[q].pool.ensure((size_t)poolPages[q] * PG_BYTES));
  // lambda ahead of the tile stage (closed form of fragLen; LooseCtl): one rank, a treatment sample, -p
  const bool wantEarly = !isCtrl && (!multiRank || earlyColl) && !ctx->par.qval_opt && !ctx->hasBed && unit32 && !noLoose && !forceSlowFrag &&
                         !ctx->sawFrac;  // (fractional weights: the closed form of fragLen is off, lambda only comes with the sample's end)
  const size_t looseCap = (size_t)2 * nEv + nTiles + ctx->nBedEdges + 16;  // slot t: records before + t (+ edges before)
  u64* sigMask = nullptr;
  if (wantEarly) {
    // the sweep's masks in loose-slot index space: [significant | first of its chromosome]
    ctx->looseStride = (looseCap + 63) / 64 + 2;
    HIPCHECK(ctx->swMask.ensure(ctx->looseStride * 8 * 3));
    sigMask = ctx->swMask.as<u64>();
    ctx->maskIdx = -1;
  }
  // everything that must start at zero lives in one arena: one launch per sample clears it (k_build_init: with the
  // sweep's masks and the replicate's scalars)
  const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tileBytes = up((size_t)(nTiles + 1) * 4);
    const size_t ffBytes = up(sizeof(FragFix));
    const size_t endBytes = up((size_t)(nChrom + 1) * 4);
    const size_t curBytes = up((size_t)NXCD * nL1 * 4 + 64);          // cursors + (last word) pages handed out
    const size_t ptBytes = up((size_t)NXCD * nL1 * jmax * 4);
    const size_t lbTBytes = up((size_t)3 * (tChunks + 2) * 8), lbIBytes = up((size_t)(2 * tChunks + 4) * 8);
    const size_t ctlBytes = up(sizeof(LooseCtl));
    const size_t netBytes = up((size_t)(MAX_BINS_P + 2) * 4);  // pair mode: the singles' weight per level-1 bin
    // pair mode in two passes (k_sort_a / k_sort_b): the coarse lists' cursors and page tables
    const u32 nCoarse = (std::max(1u, nL1) + (1u << s2_fine_shift(nL1)) - 1) >> s2_fine_shift(nL1);
    const u32 jmaxC = class_chunks(segs) + 3;   // (a class's workgroups cannot fill more pages than that in one list)
    const size_t curCBytes = up((size_t)NXCD * nCoarse * 4 + 64), ptCBytes = up((size_t)NXCD * nCoarse * jmaxC * 4);
    const size_t total = ffBytes + 256 + ctlBytes + endBytes + netBytes + curCBytes + ptCBytes + 3 * (curBytes + ptBytes) + 5 * tileBytes + lbTBytes + lbIBytes;
    HIPCHECK(ctx->zeroArena.ensure(total));
    char* base = ctx->zeroArena.as<char>();
    ctx->fragSum.view(base, ffBytes);
    base += ffBytes;
    ctx->nWide.view(base, 256);
    base += 256;
    ctx->looseCtl.view(base, ctlBytes);
    base += ctlBytes;
    ctx->endAtLen.view(base, endBytes);
    base += endBytes;
    ctx->binNet.view(base, netBytes);
    base += netBytes;
    ctx->curC.view(base, curCBytes);
    base += curCBytes;
    ctx->ptC.view(base, ptCBytes);
    base += ptCBytes;
    for (int q = 0; q < 3; q++) {
      ctx->str[q].cursor.view(base, curBytes);
      base += curBytes;
      ctx->str[q].pt.view(base, ptBytes);
      base += ptBytes;
    }
    for (int q = 0; q < 3; q++, base += tileBytes) ctx->tileCnt[q].view(base, tileBytes);
    ctx->tileWsum.view(base, tileBytes);
    base += tileBytes;
    ctx->tileDeep.view(base, tileBytes);
    base += tileBytes;
    ctx->lb.view(base, lbTBytes);      // k_scan_tiles' three look-back arrays
    base += lbTBytes;
    ctx->lbIv.view(base, lbIBytes);    // k_scan_iv's two
    if (!reuseSort) {
      const size_t nA = total / 16, nB = wantEarly ? ctx->looseStride * 8 * 2 / 16 : 0;
      static_assert(sizeof(Scalars) / 8 <= 256, "one workgroup clears the scalars");
      hipLaunchKernelGGL(k_build_init, dim3((u32)std::min<size_t>((nA + nB + 1023) / 1024, 4096)), dim3(256), 0, s,
                         ctx->dScal.as<Scalars>(), ctx->beginPending ? 1 : 0, ctx->beginGenome, ctx->zeroArena.as<uint4>(), nA,
                         wantEarly ? ctx->swMask.as<uint4>() : (uint4*)nullptr, nB);
      ctx->beginPending = false;
    } else {
      if (int rc__ = flush_begin(ctx)) return rc__;
      if (wantEarly) HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, ctx->looseStride * 8 * 2, s));
      // what the tile stage and the scans of the first attempt left: the per-tile tables and look-back arrays (the
      // arena's tail), the loose-sweep block, the correction words of fragLen and the wide-tile count
      char* tail = ctx->tileCnt[0].as<char>();
      HIPCHECK(hipMemsetAsync(tail, 0, (size_t)(ctx->zeroArena.as<char>() + total - tail), s));
      HIPCHECK(hipMemsetAsync(ctx->looseCtl.p, 0, ctlBytes, s));
      FragFix* f0 = ctx->fragSum.as<FragFix>();
      HIPCHECK(hipMemsetAsync(&f0->nList, 0, 12, s));   // nList, corr (the partial sums and the slow flag stay)
      HIPCHECK(hipMemsetAsync(ctx->nWide.p, 0, 4, s));  // (word 1, the int16 flag of k_sort1, stays)
      HIPCHECK(hipMemsetAsync(ctx->nWide.as<u32>() + 2, 0, 4, s));
    }
  }
  for (int q = 0; q < 3; q++) {
    HIPCHECK(ctx->str[q].sbOff.ensure((MAX_BINS_P + 2) * 4));
    HIPCHECK(ctx->tileOff[q].ensure((size_t)(nTiles + 2) * 4));
  }
  HIPCHECK(ctx->tileCarry.ensure((size_t)(nTiles + 1) * 4));
  // an interval closes at every base with a non-zero difference (<= one per record), at every -E edge,
  // plus one per chromosome
  const size_t ivCap = (size_t)2 * nEv + nChrom + ctx->nBedEdges + 16;
  HIPCHECK(pooled(ctx, out.ivEnd, ivCap * 4));
  HIPCHECK(pooled(ctx, out.tileIvOff, (size_t)(nTiles + 2) * 4));
  HIPCHECK(pooled(ctx, out.chromIvOff, (size_t)(nChrom + 2) * 4));

  phase_begin(ctx, isCtrl ? "c.sort1" : "t.sort1");
  // fragLen: closed form (sum of fragment lengths) unless something sets the slow flag
  HIPCHECK(ctx->fragList.ensure((size_t)(nTiles + 1) * 4));
  FragFix* ff = ctx->fragSum.as<FragFix>();
  u32* slowFrag = &ff->slow;
  if (ctx->hasBed || !unit32 || forceSlowFrag) HIPCHECK(hipMemsetAsync(slowFrag, 1, 4, s));
  PagedStream PG3[3];
  for (int q = 0; q < 3; q++) {
    gx_ctx::Stream& st = ctx->str[q];
    PG3[q] = PagedStream{st.pool.p, st.pt.as<u32>(), st.cursor.as<u32>(), st.cursor.as<u32>() + NXCD * nL1, jmax, poolPages[q],
                         NXCD * nL1};
  }
  Sort1Out so1{ff->fragSum, slowFrag, ctx->endAtLen.as<u32>(), ctx->nWide.as<u32>() + 1};
  PagedStream pcLast{};
  u32 ncLast = 0, gridB = 0;
  for (auto& seg : segs) {
    if (!seg.n || reuseSort) continue;
    // (a piece that is still on its way from the host: the main stream waits for that copy only, so the
    // scatter of the pieces that have arrived overlaps the upload of the rest)
    if (seg.ready) HIPCHECK(hipStreamWaitEvent(s, seg.ready, 0));
    const u32 blocks = (u32)((seg.n + S1_CHUNK - 1) / S1_CHUNK);
    if (pairs) {
      // two passes: coarse bins, then the fine ones (gx_sort.h)
      u32 nWG1 = 0;
      for (auto& sg : segs) nWG1 += (u32)((sg.n + S2_CHUNK - 1) / S2_CHUNK);
      const u32 nCoarse = (std::max(1u, nL1) + (1u << s2_fine_shift(nL1)) - 1) >> s2_fine_shift(nL1);
      const u32 perClass = class_chunks(segs), jmaxC = perClass + 3, nListsC = NXCD * nCoarse;
      const u32 pagesC = nWG1 + 2 * nListsC + 8;
      HIPCHECK(ctx->poolC.ensure((size_t)pagesC * PG_BYTES));
      HIPCHECK(ctx->auxC.ensure((size_t)pagesC << PgCfg<u32>::SHIFT));
      PagedStream PC{ctx->poolC.p, ctx->ptC.as<u32>(), ctx->curC.as<u32>(), ctx->curC.as<u32>() + nListsC, jmaxC, pagesC, nListsC};
      if (fracPairs)
        hipLaunchKernelGGL(k_sort_a<true>, dim3(blocks), dim3(S2_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom, sbS,
                           nL1, nCoarse, PC, ctx->auxC.as<uint8_t>(), PG3[2], ctx->binNet.as<int>(), so1, ctx->dStatus.as<u32>());
      else
        hipLaunchKernelGGL(k_sort_a<false>, dim3(blocks), dim3(S2_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom, sbS,
                           nL1, nCoarse, PC, ctx->auxC.as<uint8_t>(), PG3[2], ctx->binNet.as<int>(), so1, ctx->dStatus.as<u32>());
      pcLast = PC;
      ncLast = nCoarse;
      gridB = NXCD * (perClass + nCoarse);   // (a class's lists hold at most its chunks' + one partly filled page each)
    } else if (unit32)
      hipLaunchKernelGGL(k_sort1<true>, dim3(blocks), dim3(S1_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom,
                         sbS, nL1, PG3[0], PG3[1], PG3[2], so1, ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL(k_sort1<false>, dim3(blocks), dim3(S1_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom,
                         sbS, nL1, PG3[0], PG3[1], PG3[2], so1, ctx->dStatus.as<u32>());
  }
  if (gridB)  // the coarse lists (all pieces' events) -> the fine bins' lists
    hipLaunchKernelGGL(k_sort_b, dim3(gridB), dim3(S2_NT), 0, s, pcLast, (const uint8_t*)ctx->auxC.as<uint8_t>(), ncLast, nL1, PG3[0],
                       ctx->dStatus.as<u32>());
  if (int rc__ = dbg_sync(ctx, "k_sort1")) return rc__;
  phase_end(ctx);
  if (K.fault == 1 && !reuseSort) HIPCHECK(hipMemsetAsync(ctx->endAtLen.p, 0x01, 4, s));  // (tests: ST_END_PILE must catch it)
  long long* earlyWords = nullptr;
  if (earlyColl) {
    // this rank's closed form, whether it is valid here (unit weights so far, no -E regions, 4-byte keys), [2] unused
    earlyWords = ctx->dColl.as<long long>() + 4;
    hipLaunchKernelGGL(k_early_words, dim3(1), dim3(64), 0, s, (const FragFix*)ff, wantEarly ? 0 : 1, earlyWords);
    if (int rc__ = allreduce_words(ctx, earlyWords, 3)) return rc__;
    ctx->earlyOwed = false;
  }

  LooseCtl* ctl = ctx->looseCtl.as<LooseCtl>();
  HIPCHECK(ctx->tileSlot.ensure((size_t)(nTiles + 2) * 4));
  HIPCHECK(ctx->chromW0.ensure((size_t)(nChrom + 1) * 4));
  HIPCHECK(pooled(ctx, ctx->chromLooseOff, (size_t)(nChrom + 2) * 4));  // (moves into the replicate's record: gx_pvalues)
  phase_begin(ctx, isCtrl ? "c.bucket" : "t.bucket");
  {
    auto capOf = [&](int shift) -> u32 { return jmax >= (1u << (31 - shift)) ? 0x7FFFFFFFu : jmax << shift; };  // (list_cap)
    BinScan bs{{SS.cursor.as<u32>(), SE.cursor.as<u32>(), SF.cursor.as<u32>()},
               {capOf(PgCfg<u32>::SHIFT), capOf(PgCfg<u32>::SHIFT), capOf(PgCfg<u64>::SHIFT)},
               {SS.sbOff.as<u32>(), SE.sbOff.as<u32>(), SF.sbOff.as<u32>()},
               ctx->endAtLen.as<u32>(), ctx->chromW0.as<int>(), nChrom, ff, ctx->dScal.as<Scalars>(), ctl, wantEarly ? 1 : 0,
               pairs ? 1 : 0, ctx->binNet.as<int>(), ctx->nWide.as<u32>() + 12, earlyWords};
    static_assert(PV_LUT % 1024 == 0, "k_bins_lut: four of k_pval_lut's workgroups per block");
    if (wantEarly)  // with the table p(V) for that lambda, and from which pileup on an interval is significant
      hipLaunchKernelGGL(k_bins_lut, dim3(4 + PV_LUT / 1024), dim3(1024), 0, s, bs, nL1, ctx->pvLut.as<float>(),
                         ctx->dRisk.as<RiskBuf>(), ctx->dDeep.as<DeepTab>(), ctx->par.thr, ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL(k_scan_bins, dim3(4), dim3(1024), 0, s, bs, nL1);
  }
  if (!fused) {
    // level 2: one workgroup per super-bucket
    const size_t lds2 = std::max(b2_lds_bytes<u32>(1u << sbS), b2_lds_bytes<u64>(1u << sbS));
    if (ctx->b2LdsSet != lds2) {
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket2p), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      ctx->b2LdsSet = lds2;
    }
    // (the F stream: multimapped reads, or everything beyond 4.29 Gbp; a run without them finds every bin empty)
    Bucket2Jobs BJ{{{PG3[0], SS.a.p, SS.sbOff.as<u32>(), ctx->tileCnt[0].as<u32>()},
                    {PG3[1], SE.a.p, SE.sbOff.as<u32>(), ctx->tileCnt[1].as<u32>()},
                    {PG3[2], SF.a.p, SF.sbOff.as<u32>(), ctx->tileCnt[2].as<u32>()}}};
    hipLaunchKernelGGL(k_bucket2p, dim3(std::max(1u, nL1), 3), dim3(B2_NT), lds2, s, BJ, nL1, sbS, nTiles,
                       ctx->tileWsum.as<int>());
    if (int rc__ = dbg_sync(ctx, "k_bucket2p")) return rc__;
  }
  TileTabs tt{};
  for (int q = 0; q < 3; q++) {
    tt.cnt[q] = ctx->tileCnt[q].as<u32>();
    tt.off[q] = ctx->tileOff[q].as<u32>();
  }
  tt.wsumF = ctx->tileWsum.as<int>();
  tt.prefW = ctx->tileCarry.as<int>();
  if (!fused) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s, tt, nTiles,
                       ctx->lb.as<u64>(), ctx->lb.as<u64>() + tChunks + 2, ctx->lb.as<u64>() + 2 * (tChunks + 2),
                       ctx->dStatus.as<u32>());
    if (int rc__ = dbg_sync(ctx, "k_scan_tiles")) return rc__;
  }
  HIPCHECK(pooled(ctx, ctx->looseEnd, looseCap * 4));  // (pooled: a sample stashed for its control merge took the last ones along)
  HIPCHECK(pooled(ctx, ctx->looseV, looseCap * 4));
  HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->tileLastEnd.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->tilePrevEnd.ensure((size_t)(nTiles + 1) * 4));
  Scalars* ds = ctx->dScal.as<Scalars>();
  long long* acc = isCtrl ? ds->ctrlAcc : ds->fragAcc;  // zero since gx_sample_begin(treatment)
  TileOut to{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(),
             ctx->tileDeep.as<u32>(), sigMask, wantEarly ? ctl : (LooseCtl*)nullptr};
  BedIn bin{ctx->dBedTileOff.as<u32>(), ctx->dBedEdge.as<u32>(), ctx->dTileSave0.as<uint8_t>()};
  HIPCHECK(pooled(ctx, ctx->tileMeta, (size_t)(nTiles + 1) * sizeof(TileMeta)));
  HIPCHECK(ctx->wideList.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->heavyList.ensure((size_t)(nTiles + 1) * 4));
  if (!fused)
    hipLaunchKernelGGL(k_tile_meta, dim3((nTiles + 255) / 256), dim3(256), 0, s, ctx->tileOff[0].as<u32>(),
                       ctx->tileOff[1].as<u32>(), ctx->tileOff[2].as<u32>(), ctx->tileCarry.as<int>(), ctx->dTileChrom.as<u32>(),
                       ctx->dChrom.as<DChrom>(), ctx->hasBed ? ctx->dBedTileOff.as<u32>() : (const u32*)nullptr, nTiles,
                       ctx->tileMeta.as<TileMeta>(), ctx->wideList.as<u32>(), ctx->nWide.as<u32>(), ctx->tileSlot.as<u32>(),
                       ctx->hasBed ? (u32*)nullptr : ctx->heavyList.as<u32>());
  phase_end(ctx);

  phase_begin(ctx, isCtrl ? "c.tile" : "t.tile");  // k_tile alone: the dominant kernel (bench.py's roofline)
  TileIn tin{SS.a.as<uint16_t>(), SE.a.as<uint16_t>(), SF.a.as<u64>(), ctx->tileMeta.as<TileMeta>()};
  // the tile stage is k_tile_fast (+ k_tile_heavy): the general fragLen path's terms ride in it (TileIn::fragAcc)
  ctx->fragFused = (!fused && !ctx->hasBed) || ctx->fracPairsUsed;
  if (ctx->fragFused) {
    tin.ff = ff;
    tin.fragAcc = acc;
  }
  // narrow tiles with 16-bit LDS counters (twice the tiles in flight), then the wide ones from their list
  // (whose length stays on the device: an empty list costs one idle launch)
  const u32* wl = ctx->wideList.as<u32>();
  const u32* nw = ctx->nWide.as<u32>();
  const dim3 gHalf(std::min<u32>(nTiles, (u32)ctx->resTileHalf)), gWide(std::min<u32>(nTiles, (u32)ctx->resTile));
  if (fused) {
    // level 2 of the sort and the tile passes in one kernel, one workgroup per super-bucket (gx_sbtile.h)
    if (!ctx->sbtLdsSet) {
      for (const void* f : {reinterpret_cast<const void*>(k_sbtile<false, false, false>), reinterpret_cast<const void*>(k_sbtile<true, false, false>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, false>), reinterpret_cast<const void*>(k_sbtile<true, false, true>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, true>)})
        HIPCHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SbtLds)));
      ctx->sbtLdsSet = true;
    }
    HIPCHECK(ctx->bigBins.ensure((size_t)(MAX_BINS_P + 4) * 4));
    SbtIn si{PG3[0], PG3[1], PG3[2], SS.sbOff.as<u32>(), SE.sbOff.as<u32>(), SF.sbOff.as<u32>(), ctx->dTileChrom.as<u32>(),
             ctx->dChrom.as<DChrom>(), ctx->chromW0.as<int>(), nL1, nTiles, sbS,
             ctx->fracPairsUsed ? (const FragFix*)ff : (const FragFix*)nullptr, ctx->fracPairsUsed ? acc : (long long*)nullptr};
    SbtOut so2{to, ctx->tileMeta.as<TileMeta>(), ctx->tileSlot.as<u32>(), ctx->nWide.as<u32>() + 1, ctx->nWide.as<u32>() + 13,
               ctx->bigBins.as<u32>(), ctx->heavyList.as<u32>(), ctx->nWide.as<u32>() + 2};
    const dim3 gAll(std::max(1u, nL1)), gBig(std::max(1u, std::min(nL1, (u32)ctx->numCU)));
    // a sample so dense that the average bin already holds more keys than the key array (ATAC cut sites of a deep
    // library): every bin takes the rounds of the second launch, the first one would only find that out bin by bin
    const bool dense = ctx->pairsUsed && (size_t)2 * nEv > (size_t)std::max(1u, nL1) * (SBT_KEYCAP - SBT_KEYCAP / 16);
    if (dense) {
      so2.bigList = nullptr;
      if (ctx->fracPairsUsed)
        hipLaunchKernelGGL((k_sbtile<true, true, true>), gAll, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
      else
        hipLaunchKernelGGL((k_sbtile<true, true, false>), gAll, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
    } else if (ctx->fracPairsUsed) {
      hipLaunchKernelGGL((k_sbtile<true, false, true>), gAll, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
      hipLaunchKernelGGL((k_sbtile<true, true, true>), gBig, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
    } else if (ctx->pairsUsed) {
      hipLaunchKernelGGL((k_sbtile<true, false, false>), gAll, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
      // the bins it left on its list (reads piled up: more keys than the key array holds, a tile with thousands of keys):
      // usually none -- an idle launch
      hipLaunchKernelGGL((k_sbtile<true, true, false>), gBig, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
    } else
      hipLaunchKernelGGL((k_sbtile<false, false, false>), gAll, dim3(SBT_NT), sizeof(SbtLds), s, si, so2, ctx->dStatus.as<u32>());
  } else if (ctx->hasBed) {
    hipLaunchKernelGGL((k_tile<true, true>), gHalf, dim3(TL_NT), TL_LDS_HALF * 4, s, tin, nTiles, wl, nw, bin, to,
                       ctx->dStatus.as<u32>());
    hipLaunchKernelGGL((k_tile<true, false>), gWide, dim3(TL_NT), TL_LDS * 4, s, tin, nTiles, wl, nw, bin, to,
                       ctx->dStatus.as<u32>());
  } else {
    // the common case: one wavefront per tile, work laid out by touched base, unit-weight and fractional records
    // alike (gx_tile_fast.h)
    hipLaunchKernelGGL(k_tile_fast, dim3(std::min<u32>(nTiles, (u32)ctx->resTileFast)), dim3(64), 0, s, tin, nTiles, nw, to,
                       ctx->dStatus.as<u32>());
    // the tiles with thousands of records (pile-ups): a workgroup each, a counter per base (usually none: an idle launch)
    hipLaunchKernelGGL(k_tile_heavy, dim3(64), dim3(TH_NT), 0, s, tin, ctx->heavyList.as<u32>(), nw + 2, to, ctx->dStatus.as<u32>());
  }
  if (int rc__ = dbg_sync(ctx, "k_tile")) return rc__;
  phase_end(ctx);
  // (word 1 of the nWide block: the "a base can reach the int16 limits" flag, also set by k_convert)
  // (a bin that fits k_sbtile holds fewer than 32,767 records of a stream: no base of it can reach the limits)
  // (a tile that can hold such a base has >= 32,766 records: it is on the list of the heavy tiles -- walking the list of
  // the WIDE tiles instead cost config 4, where every tile holds fractional records and is "wide", 2.1 ms of header reads)
  if (!fused) {
    const bool haveHeavy = !ctx->hasBed;
    hipLaunchKernelGGL(k_hot_check, dim3(std::min<u32>(nTiles, 256u)), dim3(256), 0, s, tin, haveHeavy ? ctx->heavyList.as<u32>() : wl,
                       haveHeavy ? nw + 2 : nw, ctx->nWide.as<u32>() + 1);
  }
  if (int rc__ = dbg_sync(ctx, "k_hot_check")) return rc__;

  phase_begin(ctx, isCtrl ? "c.pack" : "t.pack");
  const u32 ivChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  IvScanOut so{out.tileIvOff.as<u32>(), ctx->tilePrevEnd.as<u32>(), out.chromIvOff.as<u32>(), ctx->misc.as<u32>() + M_NIV,
               ctx->tileSlot.as<u32>(), ctx->chromLooseOff.as<u32>(), ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctl,
               ctx->tileDeep.as<u32>(), ff, ctx->fragList.as<u32>(), ctx->fragFused ? acc : (long long*)nullptr,
               ctx->endAtLen.as<u32>()};
  const bool closeInScan = wantEarly && !multiRank;  // (k_scan_iv_close, below)
  if (!closeInScan)
    hipLaunchKernelGGL(k_scan_iv, dim3(std::min<u32>(ivChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                       ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->dTileChrom.as<u32>(),
                       ctx->dChrom.as<DChrom>(), nTiles, ctx->lbIv.as<u64>(), ctx->lbIv.as<u64>() + ivChunks + 1, so,
                       ctx->dStatus.as<u32>());
  if (int rc__ = dbg_sync(ctx, "k_scan_iv")) return rc__;
  {
    const u32* lE = ctx->looseEnd.as<u32>();
    const int* lV = ctx->looseV.as<int>();
    const TileMeta* tm = ctx->tileMeta.as<TileMeta>();
    const u32* tOff = out.tileIvOff.as<u32>();
    const u32* tPrev = ctx->tilePrevEnd.as<u32>();
    // (k_frag_fix1's pass over the tiles -- deep-tile list, long first intervals -- rides in k_scan_iv)
    FragSelect fsel{ff, acc, ctx->world > 1 || ctx->forceColl ? ctx->dColl.as<long long>() : (long long*)nullptr,
                    ctx->nWide.as<u32>() + 1, ctx->dStatus.as<u32>(), ctx->dChrom.as<DChrom>(), nChrom, out.chromIvOff.as<u32>(),
                    ctx->misc.as<u32>() + M_NIV, ds, isCtrl, ctx->chromLooseOff.as<u32>(), ctx->tileSlot.as<u32>(), nTiles, ctl,
                    wantEarly ? ctx->swMask.as<u64>() + ctx->looseStride : (u64*)nullptr};
    ctx->closeSel = fsel;
    ctx->closeSeq = 0;
    if (closeInScan) {
      // lambda was known before the tile stage: k_frag_select's work and the mail ride in the scan's launch; if a deep tile,
      // the general fragLen path or a changed lambda stands in the way, finish_scalars runs the separate kernels after all
      ctx->closeSeq = ++ctx->mailSeq;
      ctx->mail->nMerged = 0;
      ctx->mail->closeState = 0;
      HostMail* dm = static_cast<HostMail*>(ctx->mailBuf.dp);
      {
        // (the scan's last workgroup closes the sample: one launch)
        CloseArgs ca{fsel, ctx->misc.as<u32>() + M_NIV, (const u32*)&ctl->ok, ctx->dRisk.as<RiskBuf>(), mail_out(ctx),
                     &dm->closeState, ctx->closeSeq, ctx->nWide.as<u32>() + 8};
        hipLaunchKernelGGL(k_scan_iv_close, dim3(std::min<u32>(ivChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                           ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->dTileChrom.as<u32>(),
                           ctx->dChrom.as<DChrom>(), nTiles, ctx->lbIv.as<u64>(), ctx->lbIv.as<u64>() + ivChunks + 1, so,
                           ctx->dStatus.as<u32>(), ca);
      }
      if (int rc__ = dbg_sync(ctx, "k_close")) return rc__;
    } else {
    hipLaunchKernelGGL(k_frag_walk, dim3(std::max(1u, std::min((nTiles + 3) / 4, 4096u))), dim3(256), 0, s, lE, lV, tm, tOff,
                       tPrev, nTiles, ff, ctx->fragList.as<u32>(), acc,
                       ctx->fragFused ? ctx->heavyList.as<u32>() : (const u32*)nullptr, ctx->nWide.as<u32>() + 2);
    // (single thread: chromosome offsets of the chromosomes without tiles, closed form -> accumulator pair,
    // this rank's words of the all-reduce)
    hipLaunchKernelGGL(k_frag_select, dim3(1), dim3(1), 0, s, fsel);
    }
  }
  if (int rc__ = dbg_sync(ctx, "k_frag")) return rc__;
  out.packed = false;
  out.inLoose = false;
  if (isCtrl) {  // a control is always merged against the treatment
    int rc = stash_or_pack(ctx, out);
    if (rc) return rc;
  }
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  ctx->nIvTarget = &out.nIv;  // filled from the mail block once finish_scalars has synchronised
  return GX_OK;
}

constexpr int RETRY_GENERAL = 3;    // (internal) k_sbtile could not take the sample: build it again on the general chain
constexpr int RETRY_SATURATED = 1;  // (internal) finish_scalars: filter the events and build the sample again
constexpr int RETRY_PT = 2;         // (internal) a level-1 page list overflowed: build again with a longer page table
// (the page tables -- NXCD x bins x jmax x 4 bytes, three streams -- at the cap and hg38's 2,946 bins: 18.5 GB, which a
// 288 GB device holds; 2^20, round 2's cap, would have asked for 50 GB per stream.  A list beyond 2^16 pages holds
// more than 5 x 10^8 keys of ONE super-bucket: such a sample fails with "could not be rebuilt")
constexpr u32 PT_JMAX_CAP = 1u << 16;

// n (<= 4) 64-bit words on the device, summed over all ranks in place: RCCL in stream order (no host hop), or the host
// program's callback (a copy down, a synchronisation, a copy up)
int allreduce_words(gx_ctx* ctx, long long* d, size_t n) {
  hipStream_t s = ctx->stream;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(&ctx->err);
    if (!api) return GX_ERR_DEVICE;
    ncclResult_t r = api->allReduce(d, d, n, ncclInt64, ncclSum, ctx->comm, s);
    if (r != ncclSuccess) {
      ctx->err = std::string("ncclAllReduce: ") + api->getErrorString(r);
      return GX_ERR_DEVICE;
    }
  } else if (ctx->allreduce) {
    long long* acc = ctx->mail->coll;
    // (a big payload -- the dense p-value histogram, the all-to-all buffer of the range exchange -- is timed as a phase of
    // its own, "xfer": the trips to the host are this mode's stand-in for RCCL, not part of the phase they interrupt)
    const bool big = n > 4, wasOpen = ctx->phaseOpen;
    const std::string resume = wasOpen && ctx->nPhases ? ctx->phases[ctx->nPhases - 1].name : std::string();
    if (big) {
      HIPCHECK(ctx->hostRecs.ensure(n * 8));
      acc = static_cast<long long*>(ctx->hostRecs.p);
      phase_end(ctx);
      phase_begin(ctx, "xfer");
    }
    struct Resume {
      gx_ctx* c; std::string nm; bool on;
      ~Resume() { if (on) { phase_end(c); if (!nm.empty()) phase_begin(c, nm.c_str()); } }
    } resumeGuard{ctx, resume, big};
    HIPCHECK(hipMemcpyAsync(acc, d, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (ctx->allreduce(reinterpret_cast<int64_t*>(acc), n, ctx->user)) {
      ctx->err = "allreduce callback failed";
      return GX_ERR_DEVICE;
    }
    HIPCHECK(hipMemcpyAsync(d, acc, n * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));  // (the pinned words are reused by the next exchange)
  } else {
    ctx->err = "several ranks but no collectives (gx_set_rccl / gx_set_collectives)";
    return GX_ERR_ORDER;
  }
  return GX_OK;
}

// fragLen / ctrlFrag partial sums -> (all ranks) -> lambda, factor
int finish_scalars(gx_ctx* ctx, int isCtrl) {
  hipStream_t s = ctx->stream;
  Scalars* ds = ctx->dScal.as<Scalars>();
  const bool multi = ctx->world > 1 || ctx->forceColl;
  long long* dcoll = multi ? ctx->dColl.as<long long>() : nullptr;
  if (multi) {
    // The third word sums the ranks' "build this sample again" flags, so that every rank learns from the one
    // synchronisation below whether the sums are final.
    if (int rc__ = allreduce_words(ctx, dcoll, 3)) return rc__;
    ctx->earlyPending = false;
    // (one rank: k_frag_select has done it).  With lambda known to every rank before the tile stage (the early
    // all-reduce of build_pileup), this is also where a rank learns whether its sweep bits were written with the
    // lambda that turned out final.
    hipLaunchKernelGGL(k_finish_frag, dim3(1), dim3(1), 0, s, ds, isCtrl, ctx->dStatus.as<u32>(), (const long long*)dcoll,
                       !isCtrl && ctx->earlyColl ? ctx->looseCtl.as<LooseCtl>() : (LooseCtl*)nullptr);
    if (int rc__ = dbg_sync(ctx, "k_finish_frag")) return rc__;
  }
  bool closed = false;
  if (ctx->closeSeq) {
    // k_close has sent the mail (build_pileup); only if something stood in its way do the separate kernels run
    if (int rc__ = mail_wait(ctx, ctx->closeSeq)) return rc__;
    ctx->closeSeq = 0;
    closed = ctx->mail->closeState == 1;
    if (!closed) {
      const u32 nTiles = ctx->nTiles;
      hipLaunchKernelGGL(k_frag_walk, dim3(std::max(1u, std::min((nTiles + 3) / 4, 4096u))), dim3(256), 0, s, ctx->looseEnd.as<u32>(),
                         ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), ctx->expt.tileIvOff.as<u32>(),
                         ctx->tilePrevEnd.as<u32>(), nTiles, ctx->fragSum.as<FragFix>(), ctx->fragList.as<u32>(), ctx->closeSel.acc,
                         ctx->fragFused ? ctx->heavyList.as<u32>() : (const u32*)nullptr, ctx->nWide.as<u32>() + 2);
      hipLaunchKernelGGL(k_frag_select, dim3(1), dim3(1), 0, s, ctx->closeSel);
      if (int rc__ = dbg_sync(ctx, "k_frag (after k_close)")) return rc__;
    }
  }
  if (!closed) {
  // lambda (and with a control the factor) is final: build the p-value tables now, so that the values the
  // host has to re-evaluate (risky ones) travel with the synchronisation that returns the scalars
  if (!isCtrl) {
    PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), ctx->expt.tileIvOff.as<u32>()};
    // (when the tile stage had lambda already -- LooseCtl -- and it has not changed, only the deep tiles' part runs)
    hipLaunchKernelGGL(k_pval_lut, dim3(PV_LUT / 256 + DEEP_BLOCKS), dim3(256), 0, s, ds, ctx->pvLut.as<float>(),
                       ctx->dRisk.as<RiskBuf>(), ctx->dDeep.as<DeepTab>(), pin, ctx->fragSum.as<FragFix>(),
                       ctx->fragList.as<u32>(), ctx->looseCtl.as<LooseCtl>(), 0, ctx->par.thr);
  } else {
    hipLaunchKernelGGL(k_pair_tabs, dim3(PAIR_LUT / 256), dim3(256), 0, s, ds, ctx->pairLogE.as<double>(),
                       ctx->pairCtab.as<CtrlEntry>());
    hipLaunchKernelGGL(k_pair_tab2d, dim3(PT_N * PT_N / 256), dim3(256), 0, s, ds, ctx->pairLogE.as<double>(),
                       ctx->pairCtab.as<CtrlEntry>(), ctx->pairP2d.as<float>(), ctx->dRisk.as<RiskBuf>());
    ctx->pairTabsReady = true;
  }
  if (int rc__ = dbg_sync(ctx, "p-value tables")) return rc__;
  ctx->mail->nMerged = 0;
  if (int rc__ = mail_sync(ctx, ds, ctx->nWide.as<u32>() + 1, ctx->misc.as<u32>() + M_NIV, dcoll,
                           isCtrl ? (const u32*)nullptr : &ctx->looseCtl.as<LooseCtl>()->ok))
    return rc__;
  }
  ctx->hScal = ctx->mail->scal;
  ctx->riskNearThr = false;
  const int rcRisk = risk_apply(ctx, RiskTargets{});
  if (!isCtrl) ctx->looseOk = ctx->mail->nMerged != 0 && !ctx->riskNearThr;
  // (with several ranks: if any of them has to rebuild its sample, all go round again with it)
  const long long again = multi ? ctx->mail->coll[2]
                                : (long long)(ctx->mail->hot ? 1 : 0) + ((ctx->mail->status & ST_PT_FULL) ? 65536 : 0) +
                                      ((ctx->mail->status & (ST_SB_FULL | ST_SB_FRAC)) ? (1ll << 32) : 0);
  if (again >> 48) {
    ctx->err = "another rank could not build its sample";
    return GX_ERR_DEVICE;
  }
  if (again >> 32) {
    // k_sbtile could not take some rank's sample (a bin beyond its LDS, or fractional weights): once more, on the
    // general chain
    // (fractional weights in a unit-weight build: its singles may also have overfilled a bin -- that says nothing about
    // the next sample, which writes pair records with a weight class)
    if (ctx->knob.debugRetry) fprintf(stderr, "[gx] sample sent back to the general chain: status %u (fused %d pairs %d frac %d)\n",
                                          ctx->mail->status, (int)ctx->fusedUsed, (int)ctx->pairsUsed, (int)ctx->fracPairsUsed);
    if (ctx->mail->status & ST_SB_FRAC) ctx->sawFrac = true;
    else if (ctx->mail->status & ST_SB_FULL) ctx->fusedBackoff[isCtrl ? 1 : 0] = 8;
    ctx->fusedOff = true;
    ctx->fellBack = true;
    static_cast<RiskBuf*>(ctx->riskHost.p)->count = 0;
    HIPCHECK(hipMemsetAsync(ctx->dRisk.p, 0, 4, s));
    return RETRY_GENERAL;
  }
  if ((again & 0xFFFFFFFFll) >= 65536 && ctx->ptJmax < PT_JMAX_CAP) return RETRY_PT;
  int rc = status_to_rc(ctx, ctx->mail->status);
  if ((again & 0xFFFF) && !ctx->satDone) return RETRY_SATURATED;
  if (ctx->nIvTarget) *ctx->nIvTarget = ctx->mail->nIv;
  ctx->nIvTarget = nullptr;
  return rc ? rc : rcRisk;
}

// The sample holds a base that can reach the reference's int16 limits: bring the events to the host,
// drop the ones saveInterval would drop (gx_saturate.h) and stage what is left for a second build.
int drop_saturated(gx_ctx* ctx, int isCtrl) {
  hipStream_t s = ctx->stream;
  size_t total = 0;
  for (auto& sg : ctx->segs) total += sg.n;
  std::vector<gx_event> all(total);
  size_t at = 0;
  HIPCHECK(hipStreamSynchronize(ctx->side));  // (the uploads have long arrived: the sample was built once)
  for (auto& sg : ctx->segs) {  // in push order: the replay depends on it
    if (sg.n) HIPCHECK(hipMemcpyAsync(all.data() + at, sg.p, sg.n * sizeof(gx_event), hipMemcpyDeviceToHost, s));
    at += sg.n;
  }
  HIPCHECK(hipStreamSynchronize(s));
  // only the chromosomes this context works on (the others' events are ignored by k_convert too)
  std::vector<uint32_t> len(ctx->nChrom);
  for (u32 i = 0; i < ctx->nChrom; i++) len[i] = ctx->hChrom[i].tileBase == NULL_TILE ? 0u : ctx->len[i];
  std::vector<uint8_t> keep(total);
  const long long dropped = gxsat::filter(all.data(), total, (int)ctx->nChrom, len.data(), keep.data());
  ctx->satDropped = dropped > 0 ? dropped : 0;
  size_t kept = 0;
  if (dropped > 0) {
    for (size_t i = 0; i < total; i++)
      if (keep[i]) all[kept++] = all[i];
  } else
    kept = total;
  // (also when nothing was dropped: the second build must not see the caller's segments twice)
  HIPCHECK(ctx->satBuf.ensure(std::max<size_t>(kept, 1) * sizeof(gx_event)));
  if (kept) HIPCHECK(hipMemcpyAsync(ctx->satBuf.p, all.data(), kept * sizeof(gx_event), hipMemcpyHostToDevice, s));
  HIPCHECK(hipStreamSynchronize(s));  // `all` goes out of scope
  ctx->segs.clear();
  if (kept) ctx->segs.push_back({ctx->satBuf.as<gx_event>(), kept, nullptr});
  ctx->satDone = true;
  // what the first build left behind: its status bits and its contribution to fragLen / ctrlFrag
  Scalars* ds = ctx->dScal.as<Scalars>();
  HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, s));
  HIPCHECK(hipMemsetAsync(isCtrl ? ds->ctrlAcc : ds->fragAcc, 0, 16, s));
  return GX_OK;
}

constexpr long long COLL_FAILED = 1ll << 48;  // third all-reduce word: some rank could not build its sample

void poison_allreduce(gx_ctx* ctx) {
  if (!(ctx->world > 1 || ctx->forceColl)) return;
  hipStream_t s = ctx->stream;
  long long w[3] = {0, 0, COLL_FAILED};
  // (a build that fails ahead of its early all-reduce -- the closed form of fragLen, build_pileup -- owes the other
  // ranks that one too: they are in it, or about to be)
  const int rounds = ctx->earlyOwed ? 2 : 1;
  ctx->earlyOwed = false;
  for (int r = 0; r < rounds; r++) {
    if (ctx->comm) {
      const gxrccl::Api* api = gxrccl::load(nullptr);
      if (!api || !ctx->dColl.p) return;
      if (hipMemcpyAsync(ctx->dColl.p, w, sizeof w, hipMemcpyHostToDevice, s) != hipSuccess) return;
      (void)api->allReduce(ctx->dColl.p, ctx->dColl.p, 3, ncclInt64, ncclSum, ctx->comm, s);
      (void)hipStreamSynchronize(s);
    } else if (ctx->allreduce) {
      int64_t buf[3] = {w[0], w[1], w[2]};
      (void)ctx->allreduce(buf, 3, ctx->user);
    }
  }
}

int close_sample(gx_ctx* ctx, Pileup& P, int isCtrl) {
  ctx->fusedOff = false;
  if (!isCtrl) ctx->looseOk = false;
  auto wipe = [&]() -> int {  // what a build that is repeated left behind: status bits, its part of fragLen / ctrlFrag
    Scalars* ds = ctx->dScal.as<Scalars>();
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
    HIPCHECK(hipMemsetAsync(isCtrl ? ds->ctrlAcc : ds->fragAcc, 0, 16, ctx->stream));
    return GX_OK;
  };
  bool reuseSort = false;
  for (int attempt = 0; attempt < 12; attempt++) {
    int rc = build_pileup(ctx, P, isCtrl, reuseSort);
    reuseSort = false;
    if (rc) {
      // With several ranks the others are about to wait for this one in the fragLen all-reduce: take part in it with a
      // "this rank has failed" word, so that every rank returns an error instead of one returning and the rest hanging.
      const std::string why = ctx->err;
      poison_allreduce(ctx);
      ctx->err = why;
      return rc;
    }
    rc = finish_scalars(ctx, isCtrl);
    if (rc == RETRY_GENERAL) {
      // k_sbtile could not take the sample (finish_scalars has switched it off for this one): the general chain,
      // on the pages level 1 of the sort has already filled
      if (int w = wipe()) return w;
      // (pair records are of no use to the general chain: level 1 runs again as start / end keys)
      reuseSort = !ctx->pairsUsed;
      if (reuseSort) {
        // (k_sort1 does not run again: the status bits IT raised -- bad counts, positions, chromosomes -- must survive)
        // (and only those: what the abandoned tile stage raised -- e.g. "negative pileup" from carries that count the
        // dropped ends of fractional records it never saw -- means nothing)
        ctx->mail->statusKeep = ctx->mail->status & (ST_BAD_CHROM | ST_BAD_POS | ST_BAD_COUNT | ST_PT_FULL | ST_LOOKBACK);
        HIPCHECK(hipMemcpyAsync(ctx->dStatus.p, &ctx->mail->statusKeep, 4, hipMemcpyHostToDevice, ctx->stream));
      }
    } else if (rc == RETRY_PT) {
      // a (XCD class, super-bucket) list needed more pages than its table row holds -- reads piled up in one
      // spot: what the first build left behind goes, the table grows, the sample is built again
      // ... to what the longest list asked for (k_scan_bins: the cursors count every reservation), with a quarter to
      // spare -- not by a blind factor: the table is NXCD x bins x jmax words per stream, cleared for every sample
      u32 need = 0;
      HIPCHECK(hipMemcpy(&need, ctx->nWide.as<u32>() + 12, 4, hipMemcpyDeviceToHost));
      u32 want = std::max(ctx->ptJmax * 2, need + need / 4 + 2);
      ctx->ptJmax = std::min(want, PT_JMAX_CAP);
      ctx->ptGrew = true;
      if (int w = wipe()) return w;
    } else if (rc == RETRY_SATURATED) {
      if ((rc = drop_saturated(ctx, isCtrl))) return rc;  // (sets satDone: finish_scalars asks for this once)
    } else
      return rc;
  }
  ctx->err = "sample could not be rebuilt";
  return GX_ERR_DEVICE;
}

// tile space, super-buckets, -E edge lists and the chromosome table for the chromosomes this
// context works on: not skipped (-e), not empty, and owned by this rank (gx_set_owned)
int layout_tiles(gx_ctx* ctx) {
  const int n = (int)ctx->nChrom;
  const std::vector<uint32_t>& len = ctx->len;
  ctx->hChrom.assign(n, DChrom{});
  std::vector<u32> tileChrom;
  u32 t = 0;
  for (int i = 0; i < n; i++) {
    DChrom& c = ctx->hChrom[i];
    c.len = len[i];
    if (ctx->skip[i] || !ctx->owned[i] || len[i] == 0) {
      c.tileBase = NULL_TILE;
      c.nTiles = 0;
      continue;
    }
    c.tileBase = t;
    c.nTiles = (u32)(((uint64_t)len[i] + TILE - 1) >> TB);
    for (u32 k = 0; k < c.nTiles; k++) tileChrom.push_back((u32)i);
    t += c.nTiles;
  }
  if (t == 0) {
    // a rank that owns nothing still needs a (dormant) tile space: the first analyzable chromosome's
    for (int i = 0; i < n && t == 0; i++)
      if (!ctx->skip[i] && len[i] != 0) {
        DChrom& c = ctx->hChrom[i];
        c.tileBase = 0;
        c.nTiles = (u32)(((uint64_t)len[i] + TILE - 1) >> TB);
        tileChrom.assign(c.nTiles, (u32)i);
        t = c.nTiles;
      }
  }
  ctx->nTiles = t;
  if (t == 0) {
    ctx->err = "No analyzable genome (length=0)";
    return GX_ERR_GEN;
  }
  int lg = 0;
  while ((1u << lg) < t) lg++;
  // tiles per super-bucket: the level-1 scatter wants few bins (long runs per bin and chunk); level 2 wants
  // a super-bucket's keys to fit its one-pass LDS sort (45 K keys: ~2^9 tiles at hg38 / 50 M fragments) and
  // enough super-buckets for every CU; GX_SBSHIFT overrides for experiments
  // (k_sbtile, the fused level 2 + tile kernel, takes super-buckets of up to 2^8 tiles: hg38 = 2,946 bins)
  ctx->sbShift = std::min(SBT_MAXSHIFT, std::max(0, (lg - 1) / 2));
  if (ctx->knob.sbShift >= 0) ctx->sbShift = std::max(0, std::min(11, ctx->knob.sbShift));
  while (((t + (1u << ctx->sbShift) - 1) >> ctx->sbShift) + 1 > (u32)MAX_BINS) ctx->sbShift++;
  if ((1u << ctx->sbShift) > (u32)MAX_BINS) {
    ctx->err = "genome too large for the two-level tile sort";
    return GX_ERR_MEM;
  }
  ctx->nSB = ((t + (1u << ctx->sbShift) - 1) >> ctx->sbShift) + 1;  // + the null bucket
  // -E edges per tile (Genrich.c:2185-2195: a region starting at 0 only flips the initial state)
  {
    std::vector<u32> bedOff(t + 1, 0), edges;
    std::vector<uint8_t> save0(t, 1);
    ctx->hasBed = false;
    for (int i = 0; i < n; i++) {
      const DChrom& c = ctx->hChrom[i];
      if (c.tileBase == NULL_TILE) continue;
      const std::vector<uint32_t>& b = ctx->bed[i];
      if (!b.empty()) ctx->hasBed = true;
      bool state = b.empty() || b[0] != 0;
      size_t k = (!b.empty() && b[0] == 0) ? 1 : 0;
      for (u32 tl = 0; tl < c.nTiles; tl++) {
        const uint64_t lo = (uint64_t)tl << TB, hi = lo + TILE;
        save0[c.tileBase + tl] = state;
        bedOff[c.tileBase + tl] = (u32)edges.size();
        while (k < b.size() && b[k] < hi && b[k] < c.len) {
          edges.push_back((u32)(b[k] - lo));
          state = !state;
          k++;
        }
      }
    }
    bedOff[t] = (u32)edges.size();
    ctx->nBedEdges = edges.size();
    HIPCHECK(ctx->dBedTileOff.ensure((size_t)(t + 1) * 4));
    HIPCHECK(ctx->dBedEdge.ensure(edges.size() * 4 + 16));
    HIPCHECK(ctx->dTileSave0.ensure((size_t)t + 16));
    HIPCHECK(hipMemcpy(ctx->dBedTileOff.p, bedOff.data(), (size_t)(t + 1) * 4, hipMemcpyHostToDevice));
    if (!edges.empty()) HIPCHECK(hipMemcpy(ctx->dBedEdge.p, edges.data(), edges.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dTileSave0.p, save0.data(), (size_t)t, hipMemcpyHostToDevice));
  }
  HIPCHECK(ctx->dChrom.ensure((size_t)n * sizeof(DChrom)));
  HIPCHECK(ctx->dTileChrom.ensure((size_t)t * 4));
  HIPCHECK(hipMemcpyAsync(ctx->dTileChrom.p, tileChrom.data(), (size_t)t * 4, hipMemcpyHostToDevice, ctx->stream));
  return upload_chroms(ctx);
}

// What the sweep walks: the interval arrays (end, p[, q]) of the final p-array.
struct SweepSrc {
  const u32* end = nullptr;
  const float* p = nullptr;
  const float* q = nullptr;
  // the sweep on the loose slots (LooseCtl): `end` = the loose ends, p = the table p(V) looked up with the slots' exact
  // pileups `V`; the masks are [significant | first of its chromosome], `mStride` apart, and there are no SKIP intervals
  const int* V = nullptr;
  bool haveMasks = false, hasSkip = true;
  const u32* chromOff = nullptr;
  u32 nChrom = 0, nWords = 0;
  size_t mStride = 0;   // words between the sig / skip / brk masks in swMask
};

// callPeaks (Genrich.c:977-1069) on bit masks: runs of adjacent significant intervals -> candidates -> in-order AUC.
// ONE synchronisation, at the end: the run / candidate arrays are sized by a guess (the largest run count seen so
// far, with headroom), every kernel reads the counts on the device, the true run count comes back with the mail, and
// only when it exceeds the guess is the sweep repeated with arrays that fit.  Counts travel through pinned memory
// written by the kernels themselves, and the peak list is written straight into pinned host memory.
int run_sweep(gx_ctx* ctx, const SweepSrc& S, u32* nPeaksOut) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  HostMail* dm = static_cast<HostMail*>(ctx->mailBuf.dp);
  const u32 nWords = S.nWords, nChrom = S.nChrom;
  const u32 wChunks = (nWords + SW_CHUNK - 1) / SW_CHUNK;
  SweepMasks SM{ctx->swMask.as<u64>(), ctx->swMask.as<u64>() + S.mStride, ctx->swMask.as<u64>() + 2 * S.mStride, nWords};
  if (S.V) SM = SweepMasks{ctx->swMask.as<u64>(), nullptr, ctx->swMask.as<u64>() + S.mStride, nWords};
  u32 R = 0, nPeaks = 0;
  ctx->peakBP = 0;
  ctx->nHostPeaks = 0;
  if (nWords) {
    // (in loose-slot index space the chromosome starts were marked when the sample was closed: k_close / k_frag_select)
    if (!S.V) hipLaunchKernelGGL(k_brk_mask, dim3((nChrom + 255) / 256), dim3(256), 0, s, S.chromOff, nChrom, SM.brk);
    if (!S.haveMasks)
      hipLaunchKernelGGL(k_sig_mask, dim3(std::max(1u, std::min((nWords + 15) / 16, 4096u))), dim3(256), 0, s, S.p, S.q,
                         misc + M_NIV, ctx->par.thr, SM);
    // look-back granules of the three one-pass compactions (generation-tagged: never cleared between calls)
    {
      const size_t need = (size_t)2 * (wChunks + 8) * 8;
      if (ctx->lbSweep.cap < need) {
        HIPCHECK(ctx->lbSweep.ensure(need));
        HIPCHECK(hipMemsetAsync(ctx->lbSweep.p, 0, ctx->lbSweep.cap, s));
      }
    }
    for (int attempt = 0;; attempt++) {
      // arrays for `cap` runs (never more runs than intervals)
      const u64 capMin = ctx->knob.runCapMin > 0 ? (u64)ctx->knob.runCapMin : (u64)1 << 16;  // (tests: a tiny first guess)
      // (first guess: one run per 256 intervals -- several times what a default threshold leaves on a genome --
      // so that a single call does not pay for a second pass)
      const u64 guess = ctx->knob.runCapMin > 0 ? capMin : std::max<u64>(capMin, (u64)nWords / 4);
      const u32 cap = std::min<u64>(std::max<u64>(ctx->runCap, std::max<u64>(guess, 1)), (u64)nWords * 64);
      HIPCHECK(ctx->swStart.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->swEnd.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->headPos.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->cand.ensure((size_t)cap * sizeof(gx_peak)));
      HIPCHECK(ctx->valid.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->hPeaks.ensure((size_t)cap * sizeof(gx_peak) + 16));  // (at most one peak per run)
      HIPCHECK(ctx->candHdr.ensure((size_t)cap * sizeof(uint4)));
      HIPCHECK(ctx->longList.ensure((size_t)cap * 4 + 16));
      const u32 rChunks = (cap + RC_CHUNK - 1) / RC_CHUNK;
      {
        const size_t need = (size_t)2 * (rChunks + 8) * 8;
        if (ctx->lbSweep2.cap < need) {
          HIPCHECK(ctx->lbSweep2.ensure(need));
          HIPCHECK(hipMemsetAsync(ctx->lbSweep2.p, 0, ctx->lbSweep2.cap, s));
        }
      }
      if (++ctx->sweepGen >= (1u << 24)) {  // (the generation field wraps: start over with clean arrays)
        ctx->sweepGen = 1;
        HIPCHECK(hipMemsetAsync(ctx->lbSweep.p, 0, ctx->lbSweep.cap, s));
        HIPCHECK(hipMemsetAsync(ctx->lbSweep2.p, 0, ctx->lbSweep2.cap, s));
      }
      const u32 gen = ctx->sweepGen;
      u64* lbS = ctx->lbSweep.as<u64>();
      u64* lbE = lbS + wChunks + 8;
      u64* lbC = ctx->lbSweep2.as<u64>();
      u64* lbP = lbC + rChunks + 8;
      u32* runStart = ctx->swStart.as<u32>();
      u32* runEnd = ctx->swEnd.as<u32>();
      const u64* skipM = S.hasSkip ? SM.skip : (const u64*)nullptr;
      const u32 gridP = (u32)std::max(1, ctx->resSweep);
      // runs: count, place and write in one pass; the true count goes to the host, at most `cap` to the kernels
      hipLaunchKernelGGL(k_runs, dim3(std::min<u32>(wChunks, gridP)), dim3(SW_NT), 0, s, SM, lbS, lbE, gen, runStart, runEnd, cap,
                         misc + M_SWCOUNT, &dm->R, misc + M_TICKET3, reinterpret_cast<u64*>(misc + M_PEAKBP), ctx->dStatus.as<u32>());
      // candidates (chunks beyond the device-side run count leave at once)
      hipLaunchKernelGGL(k_cands, dim3(std::min<u32>(rChunks, gridP)), dim3(SW_NT), 0, s, SM, skipM, S.end, runStart, runEnd,
                         misc + M_SWCOUNT, ctx->par.max_gap, S.chromOff, nChrom, lbC, gen, ctx->headPos.as<u32>(), misc + M_NHEADS,
                         ctx->dStatus.as<u32>());
      hipLaunchKernelGGL(k_cand_hdr, dim3(std::max(1u, std::min((cap + 255) / 256, 4096u))), dim3(256), 0, s, SM, S.end, runStart,
                         runEnd, misc + M_SWCOUNT, ctx->headPos.as<u32>(), misc + M_NHEADS, ctx->candHdr.as<uint4>(),
                         ctx->longList.as<u32>(), misc + M_TICKET3);
      {
        const dim3 grid(std::max(1u, std::min((cap + 15) / 16, 16384u)));  // 16 candidates per workgroup and round
        const dim3 gridW(std::max(1u, std::min((cap + 3) / 4, (u32)(8 * ctx->numCU))));
// (k_peak_both: the short candidates' workgroups first, the long candidates' behind them, one launch)
#define GX_LAUNCH_PEAKS(Q, V, NSHORT, QPTR)                                                                               \
  hipLaunchKernelGGL((k_peak_both<Q, V>), dim3((NSHORT) + gridW.x), dim3(256), 0, s, (u32)(NSHORT), ctx->candHdr.as<uint4>(), \
                     S.end, S.p, QPTR, S.chromOff, nChrom, misc + M_NHEADS, ctx->longList.as<u32>(), misc + M_TICKET3,       \
                     ctx->par.thr, ctx->par.min_auc, ctx->par.min_len, ctx->cand.as<gx_peak>(), ctx->valid.as<u32>())
        if (S.V) {  // p from the table p(V) (`q` carries the exact pileups); every workgroup copies the table's compact form to LDS
          const u32 nShortV = std::min<u32>(grid.x, (u32)(8 * ctx->numCU));
          GX_LAUNCH_PEAKS(false, true, nShortV, reinterpret_cast<const float*>(S.V));
        } else if (S.q)
          GX_LAUNCH_PEAKS(true, false, grid.x, S.q);
        else
          GX_LAUNCH_PEAKS(false, false, grid.x, S.q);
#undef GX_LAUNCH_PEAKS
      }
      // the peaks, in order, into pinned host memory; their number with them
      hipLaunchKernelGGL(k_peaks, dim3(std::min<u32>(rChunks, gridP)), dim3(SW_NT), 0, s, ctx->cand.as<gx_peak>(), ctx->valid.as<u32>(),
                         misc + M_NHEADS, lbP, gen, static_cast<gx_peak*>(ctx->hPeaks.dp), misc + M_NPEAKS, &dm->nPeaks,
                         ctx->dStatus.as<u32>(), reinterpret_cast<u64*>(misc + M_PEAKBP));
      if (int rc__ = dbg_sync(ctx, "sweep kernels")) return rc__;
      // the end: status, counts, the peaks' total length (and whatever else is pending) through the mail kernel, one
      // synchronisation
      if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const u64*>(misc + M_PEAKBP)))
        return rc__;
      R = ctx->mail->R;
      ctx->runSeen = R;
      if (R <= cap) break;
      if (attempt >= 2) {
        ctx->err = "peak sweep: run count changed between attempts";
        return GX_ERR_DEVICE;
      }
      ctx->runCap = (u64)R + R / 4 + 1024;  // the guess was too small: once more, with arrays that fit
    }
    ctx->runCap = std::max<u64>(ctx->runCap, (u64)R + R / 4 + 1024);
  } else {
    if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc__;
  }
  if (R) nPeaks = ctx->mail->nPeaks;
  ctx->nHostPeaks = nPeaks;
  ctx->peakBP = R ? ctx->mail->peakBP : 0;  // (callPeaks 925: summed by k_peaks)
  *nPeaksOut = nPeaks;
  return status_to_rc(ctx, ctx->mail->status);
}

// Loose slots -> the tight interval table (end, p[, pileups]) of a replicate without control: savePval
// (Genrich.c:1720-1794) against the constant control lambda.  Needs the sample's loose slots, tile tables and
// p(V) table, i.e. must run before the next sample is built (gx_sample_begin sees to that).
int materialize_rep(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  if (!pa.loose) return GX_OK;
  hipStream_t s = ctx->stream;
  const u32 n = pa.n;
  HIPCHECK(pooled(ctx, pa.p, (size_t)n * 4 + 16));
  phase_begin(ctx, "pval");
  // (the table p(V) was built when the treatment sample was closed: finish_scalars)
  PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), pa.tileOff.as<u32>()};
  // p-mode: the sweep's significance / skip masks are filled on the way (gx_find_peaks reuses them
  // when this replicate turns out to be the only one)
  u64 *sigM = nullptr, *skipM = nullptr;
  ctx->maskIdx = -1;
  if (!ctx->par.qval_opt) {
    const u32 nWords = (n + 63) / 64;
    HIPCHECK(ctx->swMask.ensure((size_t)(nWords + 2) * 8 * 3));
    HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, (size_t)(nWords + 2) * 8 * 3, s));
    sigM = ctx->swMask.as<u64>();
    skipM = sigM + (nWords + 2);
    ctx->maskIdx = idx;
    ctx->maskN = n;
    ctx->maskStride = nWords + 2;
  }
  {
    const dim3 grid(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
    if (sigM)
      hipLaunchKernelGGL((k_pack_pval<true>), grid, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                         ctx->pvLut.as<float>(), ctx->expt.ivEnd.as<u32>(), pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                         ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL((k_pack_pval<false>), grid, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                         ctx->pvLut.as<float>(), ctx->expt.ivEnd.as<u32>(), pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                         ctx->dStatus.as<u32>());
  }
  hipLaunchKernelGGL(k_pval_deep, dim3(256), dim3(256), 0, s, pin, ctx->fragSum.as<FragFix>(), ctx->fragList.as<u32>(),
                     ctx->dScal.as<Scalars>(), ctx->dDeep.as<DeepTab>(), pa.p.as<float>(), ctx->par.thr, sigM);
  if (int rc__ = dbg_sync(ctx, "k_pack_pval")) return rc__;
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  pa.end = std::move(ctx->expt.ivEnd);
  pa.hasPiles = false;
  pa.pilesPending = ctx->keepPiles;  // made when somebody asks (ensure_piles), from the exact pileups in the loose slots
  pa.pilesDropped = !ctx->keepPiles;
  pa.loose = false;
  return GX_OK;
}

// The pileup floats of a no-control replicate (Pileup.cov of the reference: only -f / -k print them): made on
// request from the exact pileups, while the sample's loose slots are still there.
int ensure_piles(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  if (pa.loose)
    if (int rc = materialize_rep(ctx, idx)) return rc;
  if (!pa.pilesPending) return GX_OK;
  hipStream_t s = ctx->stream;
  HIPCHECK(pooled(ctx, pa.expt, (size_t)pa.n * 4 + 16));
  if (ctx->hasBed) HIPCHECK(pooled(ctx, pa.ctrl, (size_t)pa.n * 4 + 16));
  PackIn pin{ctx->looseEnd.as<u32>(), pa.keptLoose ? pa.keptV.as<int>() : ctx->looseV.as<int>(),
             pa.keptLoose ? pa.keptMeta.as<TileMeta>() : ctx->tileMeta.as<TileMeta>(), pa.tileOff.as<u32>()};
  const dim3 grid(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
  // (the control value of a replicate without control is its lambda: saveLambda 1847-1876)
  if (ctx->hasBed)
    hipLaunchKernelGGL(k_piles_from_loose<true>, grid, dim3(256), 0, s, pin, ctx->nTiles, pa.ctrlConst, pa.expt.as<float>(),
                       pa.ctrl.as<float>());
  else
    hipLaunchKernelGGL(k_piles_from_loose<false>, grid, dim3(256), 0, s, pin, ctx->nTiles, pa.ctrlConst, pa.expt.as<float>(),
                       (float*)nullptr);
  if (int rc__ = dbg_sync(ctx, "k_piles_from_loose")) return rc__;
  pa.hasPiles = true;
  pa.pilesPending = false;
  ctx->pilesMade = true;
  if (pa.keptLoose) {
    recycle(ctx, pa.keptV);
    recycle(ctx, pa.keptMeta);
    pa.keptLoose = false;
  }
  return GX_OK;
}

// The context's loose slots are about to be reused (a further replicate is built, or the Fisher combination writes its
// merged intervals there): a replicate whose pileup floats are still pending keeps what they are made of -- the exact
// pileups (looseV) and the tile descriptors -- instead of having the floats written now for nobody (k_piles_from_loose:
// 0.36 ms and 0.8 GB per replicate at hg38 / 50 M fragments; 0.4 GB of a 288 GB device kept instead).
int keep_loose_for_piles(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  if (pa.loose)
    if (int rc = materialize_rep(ctx, idx)) return rc;
  if (!pa.pilesPending || pa.keptLoose) return GX_OK;
  pa.keptV = std::move(ctx->looseV);
  pa.keptMeta = std::move(ctx->tileMeta);
  pa.keptLoose = true;
  return GX_OK;
}

}  // namespace

// ================================ C ABI ==================================================

extern "C" {

const char* gx_strerror(int status) {
  switch (status) {
    case GX_OK: return "";
    case GX_ERR_MEM: return "Cannot allocate memory";
    case GX_ERR_GEN: return "No analyzable genome (length=0)";
    case GX_ERR_EXPT: return "Experimental sample has no analyzable fragments";
    case GX_ERR_PILE: return "Invalid pileup value (< 0)";
    case GX_ERR_POS: return ": read aligned beyond reference end";
    case GX_ERR_ALNS: return "Disallowed number of alignments";
    case GX_ERR_ARR: return "Failure creating experimental pileup";
    case GX_ERR_PVAL: return "Failure collecting p-values";
    case GX_ERR_DF: return "Invalid df in pchisq()";
    case GX_ERR_ORDER: return "API called out of order";
    case GX_ERR_DEVICE: return "HIP device failure";
    case GX_ERR_OVERFLOW: return "per-base difference beyond the reference's int16 range";
    default: return "Unknown error";
  }
}

int gx_create(gx_ctx** out, const gx_params* par) {
  if (!out || !par) return GX_ERR_ORDER;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= par->device) {
    fprintf(stderr, "genrich_amd: no HIP device %d (found %d) -- there is no CPU fallback\n", par->device, nd);
    return GX_ERR_DEVICE;
  }
  gx_ctx* ctx = new gx_ctx();
  ctx->par = *par;
  ctx->device = par->device;
  for (const KnobDef& d : KNOBS)
    if (const char* e = getenv(d.name)) set_knob(ctx->knob, d.name, e);
  if (ctx->knob.bhCapLog) ctx->bhCapLog = (u32)std::max(4, std::min(28, ctx->knob.bhCapLog));  // (tests: a tiny first table)
  if (ctx->knob.ptJmax) ctx->ptJmax = (u32)std::max(1, std::min(1 << 16, ctx->knob.ptJmax));   // (tests: short page-table rows)
  *out = ctx;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  HIPCHECK(ctx->mailBuf.ensure(sizeof(HostMail)));
  ctx->mail = static_cast<HostMail*>(ctx->mailBuf.p);
  memset(ctx->mail, 0, sizeof(HostMail));
  HIPCHECK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  HIPCHECK(hipEventCreateWithFlags(&ctx->sideEv, hipEventDisableTiming));
  HIPCHECK(ctx->misc.ensure(M_WORDS * 4));
  HIPCHECK(ctx->dScal.ensure(sizeof(Scalars)));
  HIPCHECK(ctx->dStatus.ensure(64));
  // p-value tables and the list of risky values: fixed sizes, built while a sample is closed
  HIPCHECK(ctx->pvLut.ensure((size_t)(PV_LUT + PV_WHOLE) * 4));  // p(V), and p of the whole pileups once more (compact)
  HIPCHECK(ctx->pairLogE.ensure((size_t)PAIR_LUT * 8));
  HIPCHECK(ctx->pairCtab.ensure((size_t)PAIR_LUT * sizeof(CtrlEntry)));
  HIPCHECK(ctx->pairP2d.ensure((size_t)PT_N * PT_N * 4));
  HIPCHECK(ctx->dColl.ensure(64));
  HIPCHECK(ctx->dRisk.ensure(sizeof(RiskBuf)));
  HIPCHECK(ctx->dDeep.ensure(sizeof(DeepTab)));
  HIPCHECK(ctx->riskHost.ensure(sizeof(RiskBuf)));
  memset(ctx->riskHost.p, 0, 32);
  HIPCHECK(hipMemsetAsync(ctx->dRisk.p, 0, 32, ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dDeep.p, 0, sizeof(DeepTab), ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->misc.p, 0, M_WORDS * 4, ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dScal.p, 0, sizeof(Scalars), ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
  HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS * 4));
  {
    // persistent kernels: the grid must not exceed what is co-resident (look-back forward progress)
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, ctx->device));
    ctx->numCU = prop.multiProcessorCount;
    int nb = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile<true, false>, TL_NT, TL_LDS * 4));
    ctx->resTile = std::max(1, nb) * ctx->numCU;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile<true, true>, TL_NT, TL_LDS_HALF * 4));
    ctx->resTileHalf = std::max(1, nb) * ctx->numCU;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile_fast, 64, 0));
    ctx->resTileFast = std::max(1, std::min(nb, TF_WG_PER_CU)) * ctx->numCU;
    if (ctx->knob.debug)
      fprintf(stderr, "k_tile workgroups: half %d, wide %d, fast %d\n", ctx->resTileHalf, ctx->resTile, ctx->resTileFast);
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scan_iv, STL_NT, 0));
    ctx->resSweep = std::max(1, std::min(nb, 4)) * ctx->numCU;
  }
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

void gx_destroy(gx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) {
    if (const gxrccl::Api* api = gxrccl::load(nullptr)) (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  for (auto& ph : ctx->phases) {
    (void)hipEventDestroy(ph.a);
    (void)hipEventDestroy(ph.b);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->sideEv) (void)hipEventDestroy(ctx->sideEv);
  for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->stageFree) if (e) (void)hipEventDestroy(e);
  delete ctx;
}

const char* gx_last_error(const gx_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int gx_set_chroms(gx_ctx* ctx, int n, const uint32_t* len, const uint8_t* skip, const uint32_t* const* bed,
                  const int32_t* bed_len) {
  if (!ctx || n <= 0 || !len) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  ctx->nChrom = (u32)n;
  ctx->len.assign(len, len + n);
  ctx->skip.assign(n, 0);
  ctx->save.assign(n, 1);
  ctx->owned.assign(n, 1);
  ctx->bed.assign(n, {});
  ctx->bedGiven = false;
  for (int i = 0; i < n; i++) {
    ctx->skip[i] = skip && skip[i];
    if (bed && bed_len && bed_len[i] > 0 && !ctx->skip[i]) {
      ctx->bed[i].assign(bed[i], bed[i] + bed_len[i]);
      ctx->bedGiven = true;
    }
  }
  int rc = layout_tiles(ctx);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

int gx_set_collectives(gx_ctx* ctx, int rank, int world, gx_allreduce_i64_fn allreduce, gx_allgather_tab_fn allgather,
                       void* user) {
  if (!ctx || world < 1 || rank < 0 || rank >= world) return GX_ERR_ORDER;
  if (ctx->comm && (allreduce || allgather)) {  // callbacks replace a communicator of gx_set_rccl
    if (const gxrccl::Api* api = gxrccl::load(nullptr)) (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->rank = rank;
  ctx->world = world;
  ctx->allreduce = allreduce;
  ctx->allgather = allgather;
  ctx->user = user;
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_rccl_unique_id(void* out, size_t cap) {
  if (!out || cap < sizeof(ncclUniqueId)) return GX_ERR_ORDER;
  std::string err;
  const gxrccl::Api* api = gxrccl::load(&err);
  if (!api) {
    fprintf(stderr, "genrich_amd: %s\n", err.c_str());
    return GX_ERR_DEVICE;
  }
  ncclUniqueId id;
  if (api->getUniqueId(&id) != ncclSuccess) return GX_ERR_DEVICE;
  memcpy(out, &id, sizeof id);
  return GX_OK;
}

int gx_set_rccl(gx_ctx* ctx, int rank, int world, const void* unique_id) {
  if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world || world > 64) return GX_ERR_ORDER;
  const gxrccl::Api* api = gxrccl::load(&ctx->err);
  if (!api) return GX_ERR_DEVICE;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->comm) {
    (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  ncclResult_t r = api->commInitRank(&ctx->comm, world, id, rank);
  if (r != ncclSuccess) {
    ctx->err = std::string("ncclCommInitRank: ") + api->getErrorString(r);
    ctx->comm = nullptr;
    return GX_ERR_DEVICE;
  }
  ctx->rank = rank;
  ctx->world = world;
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_set_keep_pileups(gx_ctx* ctx, int keep) {
  if (!ctx) return GX_ERR_ORDER;
  ctx->keepPiles = keep != 0;
  return GX_OK;
}

int gx_set_owned(gx_ctx* ctx, const uint8_t* owned) {
  if (!ctx || !owned || ctx->nChrom == 0) return GX_ERR_ORDER;
  if (ctx->phase != 0 || ctx->sample != 0) return GX_ERR_ORDER;  // the tile space changes: between runs only
  ctx->owned.assign(owned, owned + ctx->nChrom);
  int rc = layout_tiles(ctx);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

int gx_reset(gx_ctx* ctx) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  for (Pileup* P : {&ctx->expt, &ctx->ctrl}) {  // (samples stashed for a control merge give their buffers back)
    recycle(ctx, P->looseEnd); recycle(ctx, P->looseV); recycle(ctx, P->meta);
    P->inLoose = false;
  }
  for (auto& pa : ctx->reps) {
    recycle(ctx, pa.end); recycle(ctx, pa.p); recycle(ctx, pa.expt); recycle(ctx, pa.ctrl);
    recycle(ctx, pa.chromOff); recycle(ctx, pa.q); recycle(ctx, pa.tileOff); recycle(ctx, pa.dPresent);
    recycle(ctx, pa.chromLooseOff); recycle(ctx, pa.keptV); recycle(ctx, pa.keptMeta);
  }
  ctx->reps.clear();
  ctx->pilesMade = false;
  ctx->sample = 0;
  ctx->phase = 0;
  ctx->finalIdx = -1;
  ctx->segs.clear();
  ctx->evChunkIdx = ctx->evChunkFill = ctx->evPoolUsed = 0;
  ctx->nHostPeaks = 0;
  if (ctx->statusSeen) {  // (a clean run leaves the status words at zero: no fill launch)
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
    ctx->statusSeen = 0;
  }
  return GX_OK;
}

int gx_sample_begin(gx_ctx* ctx, int is_ctrl, const uint8_t* save) {
  if (!ctx || ctx->nChrom == 0) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  if (!is_ctrl) {
    if (ctx->phase != 0) return GX_ERR_ORDER;
    // (a further replicate: the previous one's loose slots, tile tables and p(V) table are about to be reused)
    for (size_t r = 0; r < ctx->reps.size(); r++)
      if (ctx->reps[r].loose || ctx->reps[r].pilesPending)
        if (int rc = keep_loose_for_piles(ctx, (int)r)) return rc;
    for (u32 i = 0; i < ctx->nChrom; i++) ctx->save[i] = save ? (save[i] != 0) : 1;
    int rc = upload_chroms(ctx, false);
    if (rc) return rc;
    uint64_t g = ctx->par.genome_len ? ctx->par.genome_len : genome_len_for(ctx, ctx->save);
    if (!g) {
      ctx->err = "No analyzable genome (length=0)";
      return GX_ERR_GEN;  // calcLambda 1828
    }
    Scalars z{};
    z.genomeLen = g;
    ctx->hScal = z;
    // (the device's copy is cleared by the first launch of the build: k_build_init)
    ctx->beginPending = true;
    ctx->beginGenome = (u64)g;
    ctx->nPhases = 0;
    ctx->phase = 1;
  } else {
    if (ctx->phase != 2) return GX_ERR_ORDER;
    int rc = stash_or_pack(ctx, ctx->expt);  // the control's tiles are about to be built: the treatment steps aside
    if (rc) return rc;
    ctx->phase = 3;
  }
  ctx->segs.clear();
  ctx->evChunkIdx = ctx->evChunkFill = ctx->evPoolUsed = 0;  // (the previous sample's uploads were consumed: gx_sample_end synchronised)
  ctx->satDone = false;
  ctx->satDropped = 0;
  return GX_OK;
}

namespace {

constexpr size_t EV_CHUNK = (size_t)1 << 22;   // events per device chunk (64 MiB)
constexpr size_t EV_STAGE = (size_t)1 << 20;   // events per pinned staging buffer (16 MiB)

hipEvent_t ready_event(gx_ctx* ctx) {
  if (ctx->evPoolUsed == ctx->evPool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    ctx->evPool.push_back(e);
  }
  return ctx->evPool[ctx->evPoolUsed++];
}

// Host events -> the library's device chunks, by asynchronous copies on the side stream.  `pinned`: the
// caller's memory is page-locked and stays untouched until gx_sample_end, so it is the copy's source itself;
// otherwise the events go through two pinned staging buffers (the caller's buffer is free on return, and the
// host keeps parsing while a buffer is in flight).  Nothing here waits for a copy to arrive: every piece
// carries an event that the main stream waits for before the kernel that reads it (build_pileup).
int push_host(gx_ctx* ctx, const gx_event* events, size_t n, bool pinned) {
  while (n) {
    if (ctx->evChunkIdx == ctx->evChunks.size()) ctx->evChunks.emplace_back();
    DevBuf& chunk = ctx->evChunks[ctx->evChunkIdx];
    HIPCHECK(chunk.ensure(EV_CHUNK * sizeof(gx_event)));
    size_t take = std::min(n, EV_CHUNK - ctx->evChunkFill);
    if (!pinned) take = std::min(take, EV_STAGE);
    gx_event* dst = chunk.as<gx_event>() + ctx->evChunkFill;
    const gx_event* src = events;
    if (!pinned) {
      const int k = ctx->stageNext;
      ctx->stageNext ^= 1;
      HIPCHECK(ctx->stage[k].ensure(EV_STAGE * sizeof(gx_event)));
      if (!ctx->stageFree[k]) HIPCHECK(hipEventCreateWithFlags(&ctx->stageFree[k], hipEventDisableTiming));
      else HIPCHECK(hipEventSynchronize(ctx->stageFree[k]));  // its previous upload has left the buffer
      memcpy(ctx->stage[k].p, events, take * sizeof(gx_event));
      src = static_cast<const gx_event*>(ctx->stage[k].p);
      HIPCHECK(hipMemcpyAsync(dst, src, take * sizeof(gx_event), hipMemcpyHostToDevice, ctx->side));
      HIPCHECK(hipEventRecord(ctx->stageFree[k], ctx->side));
    } else
      HIPCHECK(hipMemcpyAsync(dst, src, take * sizeof(gx_event), hipMemcpyHostToDevice, ctx->side));
    hipEvent_t ev = ready_event(ctx);
    if (!ev) { ctx->err = "hipEventCreate failed"; return GX_ERR_DEVICE; }
    HIPCHECK(hipEventRecord(ev, ctx->side));
    ctx->segs.push_back({dst, take, ev});
    ctx->evChunkFill += take;
    if (ctx->evChunkFill == EV_CHUNK) { ctx->evChunkIdx++; ctx->evChunkFill = 0; }
    events += take;
    n -= take;
  }
  return GX_OK;
}

}  // namespace

int gx_push_events(gx_ctx* ctx, const gx_event* events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  return push_host(ctx, events, n, false);
}

int gx_push_events_pinned(gx_ctx* ctx, const gx_event* events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  return push_host(ctx, events, n, true);
}

long long gx_filter_saturation(const gx_event* events, size_t n, int n_chrom, const uint32_t* len, uint8_t* keep) {
  if ((!events && n) || (!keep && n) || n_chrom < 0 || (!len && n_chrom)) return GX_ERR_ORDER;
  return gxsat::filter(events, n, n_chrom, len, keep);
}

int gx_push_events_device(gx_ctx* ctx, const gx_event* d_events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (n) ctx->segs.push_back({d_events, n, nullptr});
  return GX_OK;
}

int gx_sample_end(gx_ctx* ctx, double* frag_len, float* lambda, float* factor) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->phase == 1) {
    int rc = close_sample(ctx, ctx->expt, 0);
    if (rc) return rc;
    ctx->phase = 2;
  } else if (ctx->phase == 3) {
    int rc = close_sample(ctx, ctx->ctrl, 1);
    if (rc) return rc;
    ctx->phase = 4;
  } else
    return GX_ERR_ORDER;
  if (frag_len) *frag_len = ctx->hScal.fragLen;
  if (lambda) *lambda = ctx->hScal.lambda;
  if (factor) *factor = ctx->hScal.factor;
  return GX_OK;
}

int gx_expect_fractional(gx_ctx* ctx, int on) {
  if (!ctx) return GX_ERR_ORDER;
  // (a hint can only add knowledge: what the library has learned from a sample by itself stays)
  if (on) ctx->sawFrac = true;
  return GX_OK;
}

int gx_set_knob(gx_ctx* ctx, const char* name, const char* value) {
  if (!ctx || !name) return GX_ERR_ORDER;
  if (!set_knob(ctx->knob, name, value)) {
    ctx->err = std::string("unknown switch ") + name;
    return GX_ERR_ORDER;
  }
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_dups_first(gx_ctx* ctx, const gx_dup_key* keys, const uint8_t* multi, size_t n, uint32_t* owner) {
  // (the table has 2^k >= 2 n slots addressed by 32-bit indices: n <= 2^30)
  if (!ctx || (n && (!keys || !multi || !owner)) || n > ((size_t)1 << 30)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  u32 cap = 1024;
  while ((size_t)cap < 2 * n) cap <<= 1;   // (<= 2^31: no wrap)
  DevBuf dKeys, dMulti, dOwner, dTab;
  HIPCHECK(dKeys.ensure(n * 16));
  HIPCHECK(dMulti.ensure(n + 16));
  HIPCHECK(dOwner.ensure(n * 4));
  HIPCHECK(dTab.ensure((size_t)cap * 12));
  HIPCHECK(hipMemcpyAsync(dKeys.p, keys, n * 16, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(dMulti.p, multi, n, hipMemcpyHostToDevice, s));
  DupTab T{dTab.as<u32>(), dTab.as<u32>() + cap, dTab.as<u32>() + 2 * (size_t)cap, cap - 1};
  HIPCHECK(hipMemsetAsync(T.rep, 0xFF, (size_t)cap * 8, s));   // rep = free, first = "no index yet"
  HIPCHECK(hipMemsetAsync(T.multi, 0, (size_t)cap * 4, s));
  const u32 blocks = (u32)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192));
  hipLaunchKernelGGL(k_dups_insert, dim3(blocks), dim3(256), 0, s, dKeys.as<uint4>(), (const uint8_t*)dMulti.as<uint8_t>(), (u32)n, T);
  hipLaunchKernelGGL(k_dups_lookup, dim3(blocks), dim3(256), 0, s, dKeys.as<uint4>(), (u32)n, T, dOwner.as<u32>());
  HIPCHECK(hipMemcpyAsync(owner, dOwner.p, n * 4, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

int gx_saturation_dropped(gx_ctx* ctx, long long* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = ctx->satDropped;
  return GX_OK;
}

int gx_window_net(gx_ctx* ctx, uint32_t chrom, uint32_t pos0, uint32_t n, long long* net) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3) || !net || !n || n > (1u << 16) || chrom >= ctx->nChrom) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf dNet;
  HIPCHECK(dNet.ensure((size_t)n * 8));
  HIPCHECK(hipMemsetAsync(dNet.p, 0, (size_t)n * 8, s));
  for (auto& sg : ctx->segs) {
    if (!sg.n) continue;
    if (sg.ready) HIPCHECK(hipStreamWaitEvent(s, sg.ready, 0));  // (its upload, on the side stream)
    const u32 blocks = (u32)std::min<size_t>((sg.n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_window_net, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(sg.p), sg.n, chrom, ctx->len[chrom],
                       pos0, n, dNet.as<unsigned long long>());
  }
  HIPCHECK(hipMemcpyAsync(net, dNet.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

int gx_sample_no_control(gx_ctx* ctx, float* lambda) {
  if (!ctx || ctx->phase != 2) return GX_ERR_ORDER;
  if (lambda) *lambda = ctx->hScal.lambda;  // computed with fragLen (calcLambda 1831)
  ctx->phase = 5;
  return GX_OK;
}

int gx_pvalues(gx_ctx* ctx) {
  if (!ctx || (ctx->phase != 4 && ctx->phase != 5)) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  PArray pa;
  pa.present.assign(ctx->nChrom, 0);
  for (u32 i = 0; i < ctx->nChrom; i++) pa.present[i] = !ctx->skip[i] && ctx->save[i];
  if (ctx->phase == 5) {
    // no control: the p-intervals are the treatment intervals, and they stay where the tile kernel left them
    // (loose slots) until somebody needs the tight table: gx_find_peaks on a single sample with -p does not
    pa.n = ctx->expt.nIv;
    pa.chromOff = std::move(ctx->expt.chromIvOff);
    pa.tileOff = std::move(ctx->expt.tileIvOff);
    pa.ctrlIsConst = !ctx->hasBed;
    pa.ctrlConst = ctx->hScal.lambda;
    pa.loose = true;
    // (the tile stage wrote the sweep's significance bits, in loose-slot index space: LooseCtl)
    pa.looseSweep = ctx->looseOk && !ctx->par.qval_opt && !ctx->knob.noLoose;
    pa.looseStride = ctx->looseStride;
    if (pa.looseSweep) pa.chromLooseOff = std::move(ctx->chromLooseOff);
    ctx->looseOk = false;
    ctx->reps.push_back(std::move(pa));
    ctx->sample++;
    ctx->phase = 0;
    return GX_OK;
  } else {
    // treatment + control: tile-local union of breakpoints (savePval 1768-1791)
    const u32 nTiles = ctx->nTiles, nChrom = ctx->nChrom;
    const size_t cap = (size_t)ctx->expt.nIv + ctx->ctrl.nIv + 16;
    HIPCHECK(pooled(ctx, pa.end, cap * 4));
    const bool keep = ctx->keepPiles;
    if (keep) {
      HIPCHECK(pooled(ctx, pa.expt, cap * 4));
      HIPCHECK(pooled(ctx, pa.ctrl, cap * 4));
    }
    HIPCHECK(pooled(ctx, pa.p, cap * 4));
    HIPCHECK(pooled(ctx, pa.tileOff, (size_t)(nTiles + 2) * 4));
    HIPCHECK(pooled(ctx, pa.chromOff, (size_t)(nChrom + 2) * 4));
    // loose slots: the tile kernel's loose buffers, or others like them (+ one more int array)
    HIPCHECK(pooled(ctx, ctx->looseEnd, cap * 4));
    HIPCHECK(pooled(ctx, ctx->looseV, cap * 4));
    HIPCHECK(ctx->looseC.ensure(cap * 4));
    HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
    u32* misc = ctx->misc.as<u32>();
    phase_begin(ctx, "merge");
    const bool fromLoose = ctx->expt.inLoose && ctx->ctrl.inLoose;
    if (!fromLoose && (ctx->expt.inLoose || ctx->ctrl.inLoose || !ctx->expt.packed || !ctx->ctrl.packed)) {
      ctx->err = "control merge: the two samples are not in the same form";
      return GX_ERR_ORDER;
    }
    Merge2Out mo{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->tileIvCount.as<u32>()};
    // (pos0 / len / flags of a tile do not depend on the sample: the control build's descriptors serve)
    const dim3 gridM(std::min(nTiles, (u32)(8 * ctx->numCU)));
    if (fromLoose) {
      RleIn A{ctx->expt.looseEnd.as<u32>(), ctx->expt.looseV.as<int>(), ctx->expt.tileIvOff.as<u32>(), ctx->expt.meta.as<TileMeta>()};
      RleIn Bc{ctx->ctrl.looseEnd.as<u32>(), ctx->ctrl.looseV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), ctx->ctrl.meta.as<TileMeta>()};
      hipLaunchKernelGGL(k_merge2<true>, gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->ctrl.meta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>());
    } else {
      RleIn A{ctx->expt.ivEnd.as<u32>(), ctx->expt.ivV.as<int>(), ctx->expt.tileIvOff.as<u32>(), nullptr};
      RleIn Bc{ctx->ctrl.ivEnd.as<u32>(), ctx->ctrl.ivV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), nullptr};
      hipLaunchKernelGGL(k_merge2<false>, gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->tileMeta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>());
    }
    if (int rc__ = dbg_sync(ctx, "k_merge2")) return rc__;
    const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
    HIPCHECK(hipMemsetAsync(ctx->lb.p, 0, (size_t)(tChunks + 2) * 8, s));
    hipLaunchKernelGGL(k_scan_counts, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                       ctx->tileIvCount.as<u32>(), ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles,
                       ctx->lb.as<u64>(), pa.tileOff.as<u32>(), pa.chromOff.as<u32>(), misc + M_NMERGED,
                       ctx->dStatus.as<u32>());
    hipLaunchKernelGGL(k_fix_chrom_off, dim3(1), dim3(1), 0, s, ctx->dChrom.as<DChrom>(), nChrom, pa.chromOff.as<u32>(),
                       misc + M_NMERGED);
    if (int rc__ = dbg_sync(ctx, "k_scan_counts")) return rc__;
    phase_end(ctx);
    phase_begin(ctx, "pval");
    // (the control's tables -- log(treatment), control parameters, p of whole pileup pairs -- were built when
    // its sample was closed: finish_scalars)
    PackPairsIn ppi{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->expt.tileIvOff.as<u32>(),
                    ctx->ctrl.tileIvOff.as<u32>(), pa.tileOff.as<u32>()};
    HIPCHECK(ctx->fragList.ensure((size_t)(nTiles + 1) * 4));
    // p-mode: the sweep's masks are filled on the way (the interval count is only bounded here, so the
    // masks are laid out for the bound and gx_find_peaks is told the stride)
    u64 *sigM = nullptr, *skipM = nullptr;
    ctx->maskIdx = -1;
    if (!ctx->par.qval_opt) {
      const size_t stride = (cap + 63) / 64 + 2;
      HIPCHECK(ctx->swMask.ensure(stride * 8 * 3));
      HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, stride * 8 * 3, s));
      sigM = ctx->swMask.as<u64>();
      skipM = sigM + stride;
      ctx->maskIdx = (int)ctx->reps.size();
      ctx->maskStride = stride;
    }
    HIPCHECK(hipMemsetAsync(misc + M_TICKET, 0, 4, s));
    {
      const dim3 grid(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
      const bool msk = sigM != nullptr;
#define GX_LAUNCH_PACK_PAIRS(K, M)                                                                                     \
  hipLaunchKernelGGL((k_pack_pairs<K, M>), grid, dim3(256), 0, s, ppi, nTiles, ctx->pairCtab.as<CtrlEntry>(),           \
                     ctx->pairP2d.as<float>(), pa.end.as<u32>(), pa.expt.as<float>(), pa.ctrl.as<float>(),              \
                     pa.p.as<float>(), ctx->par.thr, sigM, skipM, ctx->fragList.as<u32>(), misc + M_TICKET)
      if (keep) { if (msk) GX_LAUNCH_PACK_PAIRS(true, true); else GX_LAUNCH_PACK_PAIRS(true, false); }
      else { if (msk) GX_LAUNCH_PACK_PAIRS(false, true); else GX_LAUNCH_PACK_PAIRS(false, false); }
#undef GX_LAUNCH_PACK_PAIRS
    }
    hipLaunchKernelGGL(k_pack_pairs_full, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(4 * ctx->numCU)))), dim3(256), 0, s,
                       ppi, ctx->fragList.as<u32>(), misc + M_TICKET, ctx->dScal.as<Scalars>(), ctx->pairLogE.as<double>(),
                       ctx->pairCtab.as<CtrlEntry>(), keep ? pa.expt.as<float>() : (float*)nullptr,
                       keep ? pa.ctrl.as<float>() : (float*)nullptr, pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                       ctx->dStatus.as<u32>(), ctx->dRisk.as<RiskBuf>());
    if (int rc__ = dbg_sync(ctx, "k_pack_pairs")) return rc__;
    phase_end(ctx);
    HIPCHECK(hipGetLastError());
    if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_NMERGED)) return rc__;
    int rc = status_to_rc(ctx, ctx->mail->status);
    {
      RiskTargets T{};
      T.pairP = pa.p.as<float>();
      T.sigMask = sigM;
      T.thr = ctx->par.thr;
      const int rcRisk = risk_apply(ctx, T);
      if (!rc) rc = rcRisk;
    }
    if (rc) return rc;
    pa.n = ctx->mail->nMerged;
    ctx->maskN = pa.n;
    pa.hasPiles = keep;
    pa.pilesDropped = !keep;
    pa.ctrlIsConst = false;
  }
  ctx->reps.push_back(std::move(pa));
  ctx->sample++;
  ctx->phase = 0;
  return GX_OK;
}

// every rank's `per` 64-bit words -- written at [rank * per, ...) of a buffer that is zero elsewhere -- to every rank
// (a sum of disjoint regions is their concatenation: the fixed-size exchanges need no counts and no host)
static int coll_concat(gx_ctx* ctx, long long* d, size_t per) { return allreduce_words(ctx, d, per * (size_t)std::max(1, ctx->world)); }

// all-to-all with the counts of M (M[src * W + dst] elements of elemBytes from src to dst; sOff / rOff: this rank's
// send / receive offsets in elements).  RCCL: grouped send / recv on the library's stream.  Host callbacks (the
// validation mode of the tests): one all-reduce of a buffer in which every rank fills its outgoing segments.
static int coll_alltoallv(gx_ctx* ctx, const void* dSend, const std::vector<size_t>& sOff, void* dRecv, const std::vector<size_t>& rOff,
                          const std::vector<u32>& M, size_t elemBytes) {
  hipStream_t s = ctx->stream;
  const u32 W = (u32)std::max(1, ctx->world), me = (u32)ctx->rank;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(&ctx->err);
    if (!api) return GX_ERR_DEVICE;
    ncclResult_t r = api->groupStart();
    for (u32 p = 0; p < W && r == ncclSuccess; p++) {
      const size_t ns = (sOff[p + 1] - sOff[p]) * elemBytes, nr = (rOff[p + 1] - rOff[p]) * elemBytes;
      if (ns) r = api->send(static_cast<const char*>(dSend) + sOff[p] * elemBytes, ns, ncclChar, (int)p, ctx->comm, s);
      if (nr && r == ncclSuccess) r = api->recv(static_cast<char*>(dRecv) + rOff[p] * elemBytes, nr, ncclChar, (int)p, ctx->comm, s);
    }
    const ncclResult_t r2 = api->groupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) {
      ctx->err = std::string("ncclSend / ncclRecv: ") + api->getErrorString(r);
      return GX_ERR_DEVICE;
    }
    return GX_OK;
  }
  // (src-major layout of all segments; this rank's outgoing ones are contiguous in it, as in its send buffer)
  std::vector<size_t> segOff((size_t)W * W + 1, 0);
  for (size_t i = 0; i < (size_t)W * W; i++) segOff[i + 1] = segOff[i] + M[i];
  const size_t words = (segOff[(size_t)W * W] * elemBytes + 7) / 8 + 1;
  HIPCHECK(ctx->dGather.ensure(words * 8));
  HIPCHECK(hipMemsetAsync(ctx->dGather.p, 0, words * 8, s));
  const size_t mine = segOff[(size_t)(me + 1) * W] - segOff[(size_t)me * W];
  if (mine)
    HIPCHECK(hipMemcpyAsync(ctx->dGather.as<char>() + segOff[(size_t)me * W] * elemBytes, dSend, mine * elemBytes, hipMemcpyDeviceToDevice, s));
  if (int rc = allreduce_words(ctx, ctx->dGather.as<long long>(), words)) return rc;
  for (u32 src = 0; src < W; src++) {
    const size_t n = M[(size_t)src * W + me];
    if (n)
      HIPCHECK(hipMemcpyAsync(static_cast<char*>(dRecv) + rOff[src] * elemBytes, ctx->dGather.as<char>() + segOff[(size_t)src * W + me] * elemBytes,
                              n * elemBytes, hipMemcpyDeviceToDevice, s));
  }
  return GX_OK;
}

// gx_bhx.h: this rank's table T (Dlocal distinct values, their slots in bhOutKeys / bhOutSlot) -> the q of every one of
// them in ctx->bhQ, by slot
static int bh_range_exchange(gx_ctx* ctx, const BhTable& T, u32 Dlocal, u32 capLocal) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  const u32 W = (u32)std::max(1, ctx->world), me = (u32)ctx->rank;
  if (W > 64) { ctx->err = "more than 64 ranks"; return GX_ERR_ORDER; }
  (void)capLocal;
  // 1: this rank's distinct values in order
  HIPCHECK(ctx->bhSortKeys.ensure((size_t)std::max(Dlocal, 1u) * 4));
  HIPCHECK(ctx->bhSortSlot.ensure((size_t)std::max(Dlocal, 1u) * 4));
  if (Dlocal) {
    size_t tmpBytes = 0;
    HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                                       ctx->bhSortSlot.as<u32>(), Dlocal, 0, 32, s));
    HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
    HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                                       ctx->bhSortSlot.as<u32>(), Dlocal, 0, 32, s));
  }
  // the small fixed-size exchanges share one buffer: samples | counts matrix | range totals | range minima
  const size_t oSamp = 0, oCnt = oSamp + (size_t)W * BHX_SAMPLES, oTot = oCnt + (size_t)W * W, oMin = oTot + W, nSmall = oMin + W;
  HIPCHECK(ctx->bhxSmall.ensure(nSmall * 8 + (size_t)(2 * W + 4) * 4));
  u64* small = ctx->bhxSmall.as<u64>();
  u32* dSpl = reinterpret_cast<u32*>(small + nSmall);
  u32* dSendOff = dSpl + W + 1;
  HIPCHECK(hipMemsetAsync(small, 0, nSmall * 8, s));
  hipLaunchKernelGGL(k_bhx_samples, dim3(1), dim3(64), 0, s, (const u32*)ctx->bhSortKeys.as<u32>(), Dlocal, small + oSamp + (size_t)me * BHX_SAMPLES);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oSamp), BHX_SAMPLES)) return rc;
  hipLaunchKernelGGL(k_bhx_splitters, dim3(1), dim3(1024), 0, s, (const u64*)(small + oSamp), W * BHX_SAMPLES, W, dSpl);
  // 2: the counts, and the one synchronisation
  hipLaunchKernelGGL(k_bhx_offsets, dim3(1), dim3(128), 0, s, (const u32*)ctx->bhSortKeys.as<u32>(), Dlocal, (const u32*)dSpl, W, dSendOff,
                     small + oCnt + (size_t)me * W);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oCnt), W)) return rc;
  std::vector<u64> M64((size_t)W * W);
  HIPCHECK(hipMemcpyAsync(M64.data(), small + oCnt, M64.size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  std::vector<u32> M((size_t)W * W);
  for (size_t i = 0; i < M.size(); i++) M[i] = (u32)M64[i];
  std::vector<size_t> sOff(W + 1, 0), rOff(W + 1, 0);
  for (u32 p = 0; p < W; p++) {
    sOff[p + 1] = sOff[p] + M[(size_t)me * W + p];
    rOff[p + 1] = rOff[p] + M[(size_t)p * W + me];
  }
  if (sOff[W] != Dlocal) { ctx->err = "BH exchange: the counts do not add up"; return GX_ERR_DEVICE; }
  const size_t R = rOff[W];
  if (R >= ((size_t)1 << 31)) { ctx->err = "p-value table full"; return GX_ERR_MEM; }
  // 3: records out, records in
  HIPCHECK(ctx->bhRecs.ensure((size_t)std::max(Dlocal, 1u) * sizeof(BhRec)));
  HIPCHECK(ctx->bhxRecv.ensure(std::max<size_t>(R, 1) * sizeof(BhRec)));
  if (Dlocal)
    hipLaunchKernelGGL(k_bh_pack, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s, ctx->bhSortKeys.as<u32>(),
                       ctx->bhSortSlot.as<u32>(), ctx->bhLens.as<u64>(), Dlocal, ctx->bhRecs.as<BhRec>());
  if (int rc = coll_alltoallv(ctx, ctx->bhRecs.p, sOff, ctx->bhxRecv.p, rOff, M, sizeof(BhRec))) return rc;
  // 4: the owner's table of its range: merged, sorted, scored
  u32 cap2 = 1024;
  while ((size_t)cap2 < 4 * R) cap2 <<= 1;
  const u32 Rb = (u32)std::max<size_t>(R, 1);
  HIPCHECK(ctx->bhxKeys.ensure((size_t)cap2 * 4));
  HIPCHECK(ctx->bhxLens.ensure((size_t)cap2 * 8));
  HIPCHECK(ctx->bhxQ.ensure((size_t)cap2 * 4));
  HIPCHECK(ctx->bhxOut.ensure((size_t)Rb * 16));   // claimed keys | slots | sorted keys | sorted slots
  HIPCHECK(hipMemsetAsync(ctx->bhxKeys.p, 0xFF, (size_t)cap2 * 4, s));
  HIPCHECK(hipMemsetAsync(ctx->bhxLens.p, 0, (size_t)cap2 * 8, s));
  HIPCHECK(hipMemsetAsync(ctx->bhxOut.p, 0xFF, (size_t)Rb * 8, s));  // (unclaimed entries sort behind every value)
  u32* oKeys = ctx->bhxOut.as<u32>();
  u32 *oSlot = oKeys + Rb, *sKeys = oSlot + Rb, *sSlot = sKeys + Rb;
  u32* cnt2 = misc + M_BHOVF;  // (free again: the dense exchange is not this run's)
  HIPCHECK(hipMemsetAsync(cnt2, 0, 4, s));
  BhTable T2{ctx->bhxKeys.as<u32>(), ctx->bhxLens.as<u64>(), cap2 - 1, oKeys, oSlot, cnt2};
  if (R)
    hipLaunchKernelGGL(k_bh_insert, dim3((u32)std::max<size_t>(1, std::min<size_t>((R + 255) / 256, 1024))), dim3(256), 0, s,
                       (const BhRec*)ctx->bhxRecv.as<BhRec>(), (u32)R, T2, ctx->dStatus.as<u32>());
  {
    size_t tmpBytes = 0;
    HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, oKeys, sKeys, oSlot, sSlot, Rb, 0, 32, s));
    HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
    HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, oKeys, sKeys, oSlot, sSlot, Rb, 0, 32, s));
  }
  const u32 nCh = (Rb + QT_CHUNK - 1) / QT_CHUNK;
  HIPCHECK(ctx->bhDl.ensure((size_t)Rb * 8 + (size_t)nCh * 12 + 64));
  HIPCHECK(ctx->bhRaw.ensure((size_t)Rb * 4));
  u64* dl = ctx->bhDl.as<u64>();
  u64* chunkSum = dl + Rb;
  float* chunkMin = reinterpret_cast<float*>(chunkSum + nCh);
  hipLaunchKernelGGL(k_qt_sums, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sSlot, (const u64*)ctx->bhxLens.as<u64>(), 0u, dl, chunkSum, (const u32*)cnt2);
  hipLaunchKernelGGL(k_bhx_reduce, dim3(1), dim3(256), 0, s, (const u64*)chunkSum, (const float*)nullptr, nCh, small + oTot + me, (u64*)nullptr);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oTot), 1)) return rc;
  hipLaunchKernelGGL(k_qt_raw, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sKeys, (const u64*)dl, 0u, reinterpret_cast<const u64*>(misc + M_GENOME),
                     (const u64*)chunkSum, ctx->bhRaw.as<float>(), chunkMin, (const u32*)cnt2, (const u64*)(small + oTot), W, me,
                     ctx->par.genome_len == 0 ? ctx->dStatus.as<u32>() : (u32*)nullptr);
  hipLaunchKernelGGL(k_bhx_reduce, dim3(1), dim3(256), 0, s, (const u64*)nullptr, (const float*)chunkMin, nCh, (u64*)nullptr, small + oMin + me);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oMin), 1)) return rc;
  hipLaunchKernelGGL(k_qt_apply, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sSlot, (const float*)ctx->bhRaw.as<float>(), 0u, (const float*)chunkMin,
                     ctx->bhxQ.as<float>(), (u32*)nullptr, (const u32*)cnt2, (const u64*)(small + oMin), W, me);
  // 5: the answers, back along the same counts
  HIPCHECK(ctx->bhxAns.ensure(std::max<size_t>(R, 1) * 4 + (size_t)std::max(Dlocal, 1u) * 4));
  float* ansOut = ctx->bhxAns.as<float>();
  float* ansIn = ansOut + std::max<size_t>(R, 1);
  if (R)
    hipLaunchKernelGGL(k_bhx_answer, dim3((u32)std::max<size_t>(1, std::min<size_t>((R + 255) / 256, 1024))), dim3(256), 0, s,
                       (const BhRec*)ctx->bhxRecv.as<BhRec>(), (u32)R, (const u32*)ctx->bhxKeys.as<u32>(), cap2 - 1, (const float*)ctx->bhxQ.as<float>(), ansOut);
  std::vector<u32> Mt((size_t)W * W);
  for (u32 a = 0; a < W; a++)
    for (u32 b = 0; b < W; b++) Mt[(size_t)a * W + b] = M[(size_t)b * W + a];
  if (int rc = coll_alltoallv(ctx, ansOut, rOff, ansIn, sOff, Mt, 4)) return rc;
  if (Dlocal)
    hipLaunchKernelGGL(k_bhx_scatter, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s, (const u32*)ctx->bhSortSlot.as<u32>(),
                       (const float*)ansIn, Dlocal, ctx->bhQ.as<float>());
  (void)T;
  ctx->rangeBhUsed = true;
  return GX_OK;
}

int gx_find_peaks(gx_ctx* ctx, size_t* n_peaks, uint64_t* genome_len, uint64_t* peak_bp) {
  if (!ctx || ctx->phase != 0 || ctx->sample < 1) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  // One replicate without a control, -p, and the tile stage left the sweep's bits on the loose slots (LooseCtl): the
  // sweep walks them where they are -- no tight interval table is made unless somebody asks for it later
  const bool looseFast = ctx->sample == 1 && ctx->reps.size() == 1 && ctx->reps[0].loose && ctx->reps[0].looseSweep &&
                         !ctx->par.qval_opt;
  for (size_t r = 0; r < ctx->reps.size(); r++) {
    if (ctx->reps[r].loose && !looseFast)
      if (int rc = materialize_rep(ctx, (int)r)) return rc;
    // (the Fisher combination of several replicates reuses the loose slots: the last replicate keeps what its pileup
    // floats, if somebody asks for them, are made of)
    if (ctx->sample > 1 && ctx->reps[r].pilesPending)
      if (int rc = keep_loose_for_piles(ctx, (int)r)) return rc;
  }
  if (ctx->sample > 1 && (int)ctx->reps.size() == ctx->sample) {
    // combinePval (612-667): union of all replicates' breakpoints, Fisher's method per interval
    const int nr = ctx->sample;
    if (nr > MAX_REPS) {
      ctx->err = "more than 32 replicates are not supported";
      return GX_ERR_DF;
    }
    const u32 nTiles = ctx->nTiles, nChrom = ctx->nChrom;
    PArray comb;
    comb.present.assign(nChrom, 0);
    size_t cap = nChrom + 16;
    RepSet S{};
    S.n = nr;
    for (int r = 0; r < nr; r++) {
      PArray& pa = ctx->reps[r];
      HIPCHECK(pooled(ctx, pa.dPresent, nChrom + 16));
      HIPCHECK(hipMemcpyAsync(pa.dPresent.p, pa.present.data(), nChrom, hipMemcpyHostToDevice, s));
      for (u32 i = 0; i < nChrom; i++) comb.present[i] |= pa.present[i];
      cap += pa.n;
      S.r[r] = RepIn{pa.end.as<u32>(), pa.p.as<float>(), pa.tileOff.as<u32>(), pa.dPresent.as<uint8_t>()};
    }
    HIPCHECK(pooled(ctx, comb.end, cap * 4));
    HIPCHECK(pooled(ctx, comb.p, cap * 4));
    HIPCHECK(pooled(ctx, comb.tileOff, (size_t)(nTiles + 2) * 4));
    HIPCHECK(pooled(ctx, comb.chromOff, (size_t)(nChrom + 2) * 4));
    phase_begin(ctx, "fisher");
    HIPCHECK(pooled(ctx, ctx->looseEnd, cap * 4));
    HIPCHECK(pooled(ctx, ctx->looseV, cap * 4));   // (the last replicate may have kept the previous one: keep_loose_for_piles)
    HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
    MergeNOut mo{ctx->looseEnd.as<u32>(), ctx->looseV.as<float>(), ctx->tileIvCount.as<u32>()};
    const size_t lds = mergeN_lds_bytes((int)nr);
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mergeN), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    // the device-wide table of Fisher results: empty at the start of every run (within a run the same pairs recur
    // across tiles; across runs it would be a cache of outputs)
    HIPCHECK(ctx->fisherCache.ensure(((size_t)16 << MN_GLOBAL_LOG)));
    HIPCHECK(hipMemsetAsync(ctx->fisherCache.p, 0, (size_t)16 << MN_GLOBAL_LOG, s));
    int mnBlocks = 0;
    if (nr <= MNW_MAXREP) {
      // one wavefront per tile (no workgroup barrier in the tile loop, twenty tiles in flight per CU)
      const size_t ldsw = mergeNw_lds_bytes((int)nr);
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mergeN_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
      HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&mnBlocks, k_mergeN_w, MNW_NW * 64, ldsw));
      const u32 want = (nTiles + MNW_NW - 1) / MNW_NW;
      hipLaunchKernelGGL(k_mergeN_w, dim3(std::max(1u, std::min(want, (u32)(std::max(1, mnBlocks) * ctx->numCU)))), dim3(MNW_NW * 64), ldsw, s, S,
                         ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles, mo, ctx->dStatus.as<u32>(),
                         ctx->dRisk.as<RiskBuf>(), ctx->fisherCache.as<uint4>());
    } else {
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&mnBlocks, k_mergeN, MG_NT, lds));
    hipLaunchKernelGGL(k_mergeN, dim3(std::min(nTiles, (u32)(std::max(1, mnBlocks) * ctx->numCU))), dim3(MG_NT), lds, s, S,
                       ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles, mo, ctx->dStatus.as<u32>(),
                       ctx->dRisk.as<RiskBuf>(), ctx->fisherCache.as<uint4>());
    }
    if (int rc__ = dbg_sync(ctx, "k_mergeN")) return rc__;
    const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
    HIPCHECK(hipMemsetAsync(ctx->lb.p, 0, (size_t)(tChunks + 2) * 8, s));
    hipLaunchKernelGGL(k_scan_counts, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                       ctx->tileIvCount.as<u32>(), ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles,
                       ctx->lb.as<u64>(), comb.tileOff.as<u32>(), comb.chromOff.as<u32>(), misc + M_NMERGED,
                       ctx->dStatus.as<u32>());
    hipLaunchKernelGGL(k_fix_chrom_off, dim3(1), dim3(1), 0, s, ctx->dChrom.as<DChrom>(), nChrom, comb.chromOff.as<u32>(),
                       misc + M_NMERGED);
    hipLaunchKernelGGL(k_pack_ep, dim3(std::max(1u, std::min((nTiles + 3) / 4, 8192u))), dim3(256), 0, s, S,
                       ctx->looseEnd.as<u32>(), ctx->looseV.as<float>(), comb.tileOff.as<u32>(), nTiles, comb.end.as<u32>(),
                       comb.p.as<float>());
    if (int rc__ = dbg_sync(ctx, "k_pack_ep")) return rc__;
    phase_end(ctx);
    HIPCHECK(hipGetLastError());
    if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_NMERGED)) return rc__;
    int rc = status_to_rc(ctx, ctx->mail->status);
    {
      RiskTargets T{};
      T.fisherP = comb.p.as<float>();
      T.fisherTileOff = comb.tileOff.as<u32>();
      const int rcRisk = risk_apply(ctx, T);
      if (!rc) rc = rcRisk;
    }
    if (rc) return rc;
    comb.n = ctx->mail->nMerged;
    ctx->reps.push_back(std::move(comb));
  }
  ctx->finalIdx = (int)ctx->reps.size() - 1;
  PArray& fa = ctx->reps[ctx->finalIdx];
  const u32 n = fa.n, nChrom = ctx->nChrom;
  // genome length (findPeaks 1091-1101)
  uint64_t g = ctx->par.genome_len;
  const bool genomeOpt = g == 0;
  if (genomeOpt) g = genome_len_for(ctx, fa.present);
  ctx->genomeLenUsed = g;
  ctx->mail->genome = g;
  ctx->mail->n = n;
  // genome length and interval count for the kernels that read them through pointers (BH, k_sig_mask): one tiny
  // kernel instead of two copy launches, and none at all when the masks came with the p-values
  const bool masksReady = looseFast || (ctx->maskIdx == ctx->finalIdx && ctx->maskN == n);
  if (ctx->par.qval_opt || !masksReady)
    hipLaunchKernelGGL(k_set_misc, dim3(1), dim3(1), 0, s, misc, (u32)M_NIV, (u32)M_GENOME, (u64)g, n);

  if (ctx->par.qval_opt) {
    phase_begin(ctx, "bh");
    // The table of distinct p-values: open addressing, 2^bhCapLog slots.  It starts at 2^22 (16 MiB of keys: L2 /
    // Infinity-Cache resident for the per-interval look-ups) and grows by 8x, for good, whenever an insertion
    // gives up (ST_HASH_FULL: bh_global_add stops after BH_MAX_PROBE steps instead of crawling through a full
    // table) -- the reference's chained hash (recordPval 277-295) has no limit either.
    u32 cap = 1u << ctx->bhCapLog;
    auto bh_table = [&](u32 c) -> int {
      const bool fresh = ctx->bhKeys.cap < (size_t)c * 4;
      HIPCHECK(ctx->bhKeys.ensure((size_t)c * 4));
      HIPCHECK(ctx->bhLens.ensure((size_t)c * 8));
      HIPCHECK(ctx->bhQ.ensure((size_t)c * 4));
      HIPCHECK(ctx->bhOutKeys.ensure((size_t)c * 4));
      HIPCHECK(ctx->bhOutSlot.ensure((size_t)c * 4));
      if (fresh || ctx->bhDirty) {  // normally the table comes back clean from the previous call (k_bh_clear)
        HIPCHECK(hipMemsetAsync(ctx->bhKeys.p, 0xFF, (size_t)c * 4, s));
        HIPCHECK(hipMemsetAsync(ctx->bhLens.p, 0, (size_t)c * 8, s));
      }
      ctx->bhDirty = true;
      return GX_OK;
    };
    auto bh_grow = [&]() -> int {
      if (ctx->bhCapLog >= 28) {
        ctx->err = "p-value table full";
        return GX_ERR_MEM;
      }
      ctx->bhCapLog += 3;
      cap = 1u << ctx->bhCapLog;
      ctx->bhDirty = true;  // (whatever the failed attempt left behind is wiped)
      return GX_OK;
    };
    BhTable T{};
    u32 Dlocal = 0;
    for (;;) {
      if (int rc = bh_table(cap)) return rc;
      HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 8, s));
      T = BhTable{ctx->bhKeys.as<u32>(), ctx->bhLens.as<u64>(), cap - 1, ctx->bhOutKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                  misc + M_BHCOUNT};
      hipLaunchKernelGGL(k_bh_hist, dim3(std::max(1u, std::min((n + 4095) / 4096, 2048u))), dim3(256), 0, s,
                         fa.end.as<u32>(), fa.p.as<float>(), fa.chromOff.as<u32>(), nChrom, misc + M_NIV, T,
                         ctx->dStatus.as<u32>());
      if (int rc__ = dbg_sync(ctx, "k_bh_hist")) return rc__;
      if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_BHCOUNT)) return rc__;
      Dlocal = ctx->mail->nMerged;
      if (!(ctx->mail->status & ST_HASH_FULL)) break;
      if (ctx->mail->status != ST_HASH_FULL) return status_to_rc(ctx, ctx->mail->status & ~ST_HASH_FULL);
      HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 4, s));
      if (int rc = bh_grow()) return rc;
    }
    const bool multi = ctx->world > 1 || ctx->forceColl;
    // (computeQval 377-382: checked when the genome length was computed -- not with -L)
    u32* lenCheck = genomeOpt ? ctx->dStatus.as<u32>() : (u32*)nullptr;
    u32 D = 0;
    bool denseDone = false;
    ctx->denseBhUsed = false;
    ctx->rangeBhUsed = false;
    // One sample without a control: p is a function of the pileup, every rank holds the same table p(V), and the
    // genome-wide histogram is ONE all-reduce of a dense "bp at V" array (gx_stats.h: k_bh_dense_fill) -- decided by what
    // every rank knows alike
    if (multi && ctx->sample == 1 && ctx->reps.size() == 1 && fa.ctrlIsConst && !ctx->bedGiven && (u32)std::max(1, ctx->world) <= 64 &&
        !ctx->knob.noDenseBh) {
      const u32 W = (u32)std::max(1, ctx->world);
      const size_t words = bhd_words(W);
      HIPCHECK(ctx->bhDense.ensure(words * 8));
      HIPCHECK(hipMemsetAsync(ctx->bhDense.p, 0, words * 8, s));
      HIPCHECK(hipMemsetAsync(misc + M_BHOVF, 0, 4, s));  // (the "a rank's region overflowed" word)
      hipLaunchKernelGGL(k_bh_dense_fill, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s,
                         ctx->bhOutKeys.as<u32>(), ctx->bhOutSlot.as<u32>(), ctx->bhLens.as<u64>(), misc + M_BHCOUNT,
                         ctx->pvLut.as<float>(), ctx->bhDense.as<u64>(), (u32)ctx->rank);
      if (int rc__ = allreduce_words(ctx, ctx->bhDense.as<long long>(), words)) return rc__;
      hipLaunchKernelGGL(k_bh_clear, dim3(64), dim3(256), 0, s, T);
      HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 4, s));
      hipLaunchKernelGGL(k_bh_from_dense, dim3(256), dim3(256), 0, s, (const u64*)ctx->bhDense.as<u64>(), ctx->pvLut.as<float>(), W, T,
                         misc + M_BHOVF, ctx->dStatus.as<u32>());
      HIPCHECK(hipMemcpyAsync(&ctx->mail->D, misc + M_BHCOUNT, 4, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipMemcpyAsync(&ctx->mail->bhOvf, misc + M_BHOVF, 4, hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      if (ctx->mail->bhOvf == 0) {
        D = ctx->mail->D;
        denseDone = true;
        ctx->denseBhUsed = true;
      } else {
        // some rank holds more values outside the table than its region takes: every rank saw it, all go back to their own
        // tables and take the general exchange
        hipLaunchKernelGGL(k_bh_clear, dim3(64), dim3(256), 0, s, T);
        HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 8, s));
        hipLaunchKernelGGL(k_bh_hist, dim3(std::max(1u, std::min((n + 4095) / 4096, 2048u))), dim3(256), 0, s,
                           fa.end.as<u32>(), fa.p.as<float>(), fa.chromOff.as<u32>(), nChrom, misc + M_NIV, T,
                           ctx->dStatus.as<u32>());
        if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_BHCOUNT)) return rc__;
        Dlocal = ctx->mail->nMerged;
      }
    }
    bool rangeDone = false;
    if (denseDone) {
    } else if (multi) {
      // a control / replicates: the range-partitioned exchange (gx_bhx.h) leaves every value's q in this rank's table
      if (int rc = bh_range_exchange(ctx, T, Dlocal, cap)) return rc;
      rangeDone = true;
      D = 0;
    } else
      D = Dlocal;
    if (D && !rangeDone) {
      HIPCHECK(ctx->bhSortKeys.ensure((size_t)D * 4));
      HIPCHECK(ctx->bhSortSlot.ensure((size_t)D * 4));
      HIPCHECK(ctx->bhRaw.ensure((size_t)D * 4));
      size_t tmpBytes = 0;
      HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(),
                                         ctx->bhOutSlot.as<u32>(), ctx->bhSortSlot.as<u32>(), D, 0, 32, s));
      HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
      HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(),
                                         ctx->bhOutSlot.as<u32>(), ctx->bhSortSlot.as<u32>(), D, 0, 32, s));
      if (D <= 16384 && !ctx->knob.qtMulti) {
        hipLaunchKernelGGL(k_qtable, dim3(1), dim3(1024), 0, s, ctx->bhSortKeys.as<u32>(), ctx->bhSortSlot.as<u32>(),
                           ctx->bhLens.as<u64>(), D, reinterpret_cast<const u64*>(misc + M_GENOME), ctx->bhQ.as<float>(),
                           ctx->bhRaw.as<float>(), misc + M_ALLONE, lenCheck);
      } else {  // many distinct values (Fisher-combined replicates): the chunked kernels
        const u32 nCh = (D + QT_CHUNK - 1) / QT_CHUNK;
        HIPCHECK(ctx->bhDl.ensure((size_t)D * 8 + (size_t)nCh * 12 + 64));
        u64* dl = ctx->bhDl.as<u64>();
        u64* chunkSum = dl + D;
        float* chunkMin = reinterpret_cast<float*>(chunkSum + nCh);
        hipLaunchKernelGGL(k_qt_sums, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortSlot.as<u32>(), ctx->bhLens.as<u64>(), D, dl,
                           chunkSum, (const u32*)nullptr);
        hipLaunchKernelGGL(k_qt_raw, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortKeys.as<u32>(), dl, D,
                           reinterpret_cast<const u64*>(misc + M_GENOME), chunkSum, ctx->bhRaw.as<float>(), chunkMin,
                           (const u32*)nullptr, (const u64*)nullptr, 1u, 0u, lenCheck);
        hipLaunchKernelGGL(k_qt_apply, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortSlot.as<u32>(), ctx->bhRaw.as<float>(), D,
                           chunkMin, ctx->bhQ.as<float>(), misc + M_ALLONE, (const u32*)nullptr, (const u64*)nullptr, 1u, 0u);
      }
  if (int rc__ = dbg_sync(ctx, "k_qtable")) return rc__;
    }
    HIPCHECK(pooled(ctx, fa.q, (size_t)n * 4 + 16));
    {  // q-values and, on the way, the sweep's significance / SKIP masks
      const size_t stride = (size_t)((n + 63) / 64) + 2;
      HIPCHECK(ctx->swMask.ensure(stride * 8 * 3));
      HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, stride * 8 * 3, s));
      ctx->maskIdx = ctx->finalIdx;
      ctx->maskN = n;
      ctx->maskStride = stride;
      hipLaunchKernelGGL(k_qlookup, dim3(std::max(1u, std::min((n + 4095) / 4096, 4096u))), dim3(256), 0, s, fa.p.as<float>(),
                         misc + M_NIV, ctx->bhKeys.as<u32>(), ctx->bhQ.as<float>(), cap - 1, fa.q.as<float>(), ctx->par.thr,
                         ctx->swMask.as<u64>(), ctx->swMask.as<u64>() + stride);
    }
    hipLaunchKernelGGL(k_bh_clear, dim3(64), dim3(256), 0, s, T);
    ctx->bhDirty = false;
  if (int rc__ = dbg_sync(ctx, "k_qlookup")) return rc__;
    phase_end(ctx);
    HIPCHECK(hipGetLastError());
  }

  // peak sweep
  phase_begin(ctx, "sweep");
  SweepSrc src{};
  src.nChrom = nChrom;
  if (looseFast) {
    src.end = ctx->looseEnd.as<u32>();
    src.V = ctx->looseV.as<int>();
    src.p = ctx->pvLut.as<float>();
    src.chromOff = fa.chromLooseOff.as<u32>();
    src.mStride = fa.looseStride;
    src.nWords = (u32)(fa.looseStride - 2);
    src.haveMasks = true;
    src.hasSkip = false;
  } else {
    src.end = fa.end.as<u32>();
    src.p = fa.p.as<float>();
    src.q = ctx->par.qval_opt ? fa.q.as<float>() : (const float*)nullptr;
    src.chromOff = fa.chromOff.as<u32>();
    src.nWords = (n + 63) / 64;
    // the masks were filled by the pack kernels (p mode, one replicate) or by k_qlookup (q mode)
    src.haveMasks = ctx->maskIdx == ctx->finalIdx && ctx->maskN == n;
    src.mStride = src.haveMasks ? ctx->maskStride : (size_t)src.nWords + 2;
    src.hasSkip = true;
    HIPCHECK(ctx->swMask.ensure(src.mStride * 8 * 3));
    // (whoever filled the sig / skip masks zeroed all three: the chromosome-start mask is still clear)
    if (!src.haveMasks) HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, src.mStride * 8 * 3, s));
  }
  ctx->maskIdx = -1;
  ctx->looseSwept = looseFast;
  u32 nPeaks = 0;
  if (int rc = run_sweep(ctx, src, &nPeaks)) return rc;
  phase_end(ctx);
  if (n_peaks) *n_peaks = nPeaks;
  if (genome_len) *genome_len = g;
  if (peak_bp) *peak_bp = ctx->peakBP;
  return GX_OK;
}

int gx_peak_count(gx_ctx* ctx, size_t* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = ctx->nHostPeaks;
  return GX_OK;
}

int gx_get_peaks(gx_ctx* ctx, gx_peak* out, size_t cap) {
  if (!ctx || !out) return GX_ERR_ORDER;
  size_t n = std::min(cap, ctx->nHostPeaks);
  if (n) memcpy(out, ctx->hPeaks.p, n * sizeof(gx_peak));
  return GX_OK;
}

static const PArray* which_array(gx_ctx* ctx, int which, int chrom, u32* lo, u32* hi) {
  if (chrom < 0 || (u32)chrom >= ctx->nChrom) return nullptr;
  int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
  if (w < 0 || w >= (int)ctx->reps.size()) return nullptr;
  if (ctx->reps[w].loose && materialize_rep(ctx, w) != GX_OK) return nullptr;
  const PArray& pa = ctx->reps[w];
  if (!pa.present[chrom]) return nullptr;
  u32 off[2];
  if (hipMemcpy(off, pa.chromOff.as<u32>() + chrom, 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  *lo = off[0];
  *hi = off[1];
  return &pa;
}

int gx_interval_count(gx_ctx* ctx, int which, int chrom, size_t* n_iv) {
  if (!ctx || !n_iv) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  u32 lo = 0, hi = 0;
  const PArray* pa = which_array(ctx, which, chrom, &lo, &hi);
  *n_iv = pa ? hi - lo : 0;
  return GX_OK;
}

int gx_get_intervals(gx_ctx* ctx, int which, int chrom, size_t cap, uint32_t* end, float* expt, float* ctrl, float* p,
                     float* q) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  u32 lo = 0, hi = 0;
  const PArray* pa = which_array(ctx, which, chrom, &lo, &hi);
  if (!pa) return GX_OK;
  size_t n = std::min<size_t>(cap, hi - lo);
  if ((expt || ctrl) && pa->pilesPending) {
    const int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
    if (int rc = ensure_piles(ctx, w)) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
  }
  if ((expt || ctrl) && pa->pilesDropped) {
    ctx->err = "the pileup values were not kept (gx_set_keep_pileups)";
    return GX_ERR_ORDER;
  }
  if (!n) return GX_OK;
  if (end) HIPCHECK(hipMemcpy(end, pa->end.as<u32>() + lo, n * 4, hipMemcpyDeviceToHost));
  if (p) HIPCHECK(hipMemcpy(p, pa->p.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
  if (expt) {
    if (pa->hasPiles) HIPCHECK(hipMemcpy(expt, pa->expt.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    else std::fill(expt, expt + n, 0.0f);
  }
  if (ctrl) {
    if (pa->hasPiles && !pa->ctrlIsConst) HIPCHECK(hipMemcpy(ctrl, pa->ctrl.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    else std::fill(ctrl, ctrl + n, pa->ctrlIsConst ? pa->ctrlConst : 0.0f);
  }
  if (q) {
    if (pa->q.p && ctx->par.qval_opt) HIPCHECK(hipMemcpy(q, pa->q.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    else std::fill(q, q + n, GX_SKIP);
  }
  return GX_OK;
}

int gx_selftest2(gx_ctx* ctx, int what, const float* a, const float* b, float* out, double* out_double, size_t n,
                 size_t* n_risky) {
  if (!ctx || !a || !out || !n || n > 0xFFFFFFFFull) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  DevBuf da, db, dout, ddbl;
  HIPCHECK(da.ensure(n * 4));
  HIPCHECK(db.ensure(n * 4));
  HIPCHECK(dout.ensure(n * 4));
  if (out_double) HIPCHECK(ddbl.ensure(n * 8));
  HIPCHECK(hipMemcpy(da.p, a, n * 4, hipMemcpyHostToDevice));
  if (b) HIPCHECK(hipMemcpy(db.p, b, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest, dim3(1024), dim3(256), 0, ctx->stream, what, da.as<float>(), db.as<float>(),
                     dout.as<float>(), ddbl.as<double>(), (u32)n, ctx->dRisk.as<RiskBuf>());
  if (int rc__ = dbg_sync(ctx, "k_selftest")) return rc__;
  if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc__;
  if (n_risky) *n_risky = static_cast<RiskBuf*>(ctx->riskHost.p)->count;
  RiskTargets T{};
  T.selfOut = dout.as<float>();
  if (int rc__ = risk_apply(ctx, T, RiskHostIn{a, b})) return rc__;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  HIPCHECK(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
  if (out_double) HIPCHECK(hipMemcpy(out_double, ddbl.p, n * 8, hipMemcpyDeviceToHost));
  return GX_OK;
}

int gx_selftest(gx_ctx* ctx, int what, const float* a, const float* b, float* out, size_t n) {
  return gx_selftest2(ctx, what, a, b, out, nullptr, n, nullptr);
}

// the same scalar routines compiled for the host (what 1: calcPval, 3: multPval's tail): what the library
// evaluates risky values with, i.e. with this machine's libm; no context, no device
int gx_selftest_host(int what, const float* a, const float* b, float* out, double* out_double, size_t n) {
  if (!a || !b || !out || (what != 1 && what != 3)) return GX_ERR_ORDER;
  for (size_t i = 0; i < n; i++) {
    bool rk = false;
    double d = 0.0;
    if (what == 1) {
      out[i] = calc_pval(a[i], b[i], &rk);
      if (a[i] > 0.0f && b[i] > 0.0f) {
        double ml, sl;
        lnorm_params(b[i], &ml, &sl);
        d = pval_double(a[i], log((double)a[i]), ml, sl);
      }
    } else {
      out[i] = fisher_combine((double)a[i], (int)b[i], &rk);
      if ((int)b[i] > 2 && a[i] != 0.0f) d = fisher_double((double)a[i], (int)b[i]);
    }
    if (out_double) out_double[i] = d;
  }
  return GX_OK;
}

int gx_total_intervals(gx_ctx* ctx, int which, size_t* n_iv) {
  if (!ctx || !n_iv) return GX_ERR_ORDER;
  int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
  if (w < 0 || w >= (int)ctx->reps.size()) return GX_ERR_ORDER;
  *n_iv = ctx->reps[w].n;
  return GX_OK;
}

int gx_set_phase_filter(gx_ctx* ctx, const char* name) {
  if (!ctx || !name) return GX_ERR_ORDER;
  ctx->phaseFilter = name;
  ctx->phaseLevel = 1;
  return GX_OK;
}

int gx_rccl_nranks(gx_ctx* ctx, int* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = 0;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(nullptr);
    if (api && api->commCount && api->commCount(ctx->comm, n) != ncclSuccess) *n = 0;
  }
  return GX_OK;
}

int gx_path_info(gx_ctx* ctx, unsigned* flags) {
  if (!ctx || !flags) return GX_ERR_ORDER;
  *flags = (ctx->fusedUsed ? GX_PATH_FUSED : 0u) | (ctx->fusedUsed && ctx->pairsUsed ? GX_PATH_PAIRS : 0u) | (ctx->denseBhUsed ? GX_PATH_DENSE_BH : 0u) | (ctx->rangeBhUsed ? GX_PATH_RANGE_BH : 0u) | (ctx->looseSwept ? GX_PATH_LOOSE_SWEEP : 0u) |
           (ctx->fellBack ? GX_PATH_FELL_BACK : 0u) | (ctx->ptGrew ? GX_PATH_PT_GREW : 0u) | (ctx->fusedUsed && ctx->fracPairsUsed ? GX_PATH_FRAC_PAIRS : 0u) |
           (ctx->pilesMade ? GX_PATH_PILES_MADE : 0u);
  return GX_OK;
}

int gx_set_phase_timing(gx_ctx* ctx, int level) {
  if (!ctx || level < 0 || level > 2) return GX_ERR_ORDER;
  ctx->phaseLevel = level;
  return GX_OK;
}

int gx_phase_times(gx_ctx* ctx, const char** names, const float** ms) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->phaseMs.clear();
  ctx->phaseNames.clear();
  for (size_t i = 0; i < ctx->nPhases; i++) {
    Phase& ph = ctx->phases[i];
    float t = 0;
    (void)hipEventElapsedTime(&t, ph.a, ph.b);
    ctx->phaseMs.push_back(t);
    ctx->phaseNames += ph.name;
    ctx->phaseNames.push_back('\0');
  }
  if (names) *names = ctx->phaseNames.c_str();
  if (ms) *ms = ctx->phaseMs.data();
  return (int)ctx->nPhases;
}

}  // extern "C"
