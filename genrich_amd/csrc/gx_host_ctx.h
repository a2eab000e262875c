// gx_host_ctx.h -- the context behind the C ABI: device / pinned buffers, the per-run records, the test switches, the mail
// block and the re-evaluation of risky p-values (gx_math.h round_checked).
// (a part of gx_api.hip's translation unit: the kernels are templates and inline functions of the headers it includes;
// split by phase -- context / build / stats / sweep / collectives -- in round 5)
#pragma once
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
  bool pilesPending = false;  // the pileup floats are wanted but not made yet (ensure_piles)
  bool pairPending = false;   // ... with a control: to be made from the merge's loose arrays, the two samples' tile offsets and the
                              // control's tables, which all last until the next sample begins (make_pair_piles)
  bool mergedP = false;       // ... by a merge that wrote p-values, not pileups, into its loose slots (k_merge2<.., true>): the
                              // pileups' merge runs when the floats are asked for
  // ... and what they will be made from once the context has built another sample into its loose slots: this replicate's
  // exact pileups and tile descriptors, taken out of the context (no copy; the context takes other buffers: pooled)
  DevBuf keptV, keptMeta;
  bool keptLoose = false;
  bool latePending = false, lateLoose = false;   // the loose sweep's bits are still to be written (k_loose_late) / were written late
  bool qLazy = false;     // -q: q[] holds the candidates' intervals only; the rest on request (ensure_q)
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
  int noBedFused = 0;     // GX_NO_BED_FUSED: a run with -E regions takes the general chain (k_tile<BED>), as until round 5
  int bhVariant = -1;     // GX_BH_VARIANT: the instance of k_bh_hist / k_qlookup (0: 256 threads, 2048-entry LDS tables -- rounds 2-5; 1: 1024 threads,
                          // one workgroup per CU, 8192 / 16384 entries; 2: 512 threads, 4096 / 8192 entries); default: chosen by the run
  int mergeWg = 0;        // GX_MERGE_WG: the control merge by k_merge2 (a workgroup per tile, rounds 2-5) instead of k_merge2w (a wavefront per tile)
  int noQLoose = 0;       // GX_NO_Q_LOOSE: -q on one replicate without control makes the tight table and sweeps it, as until round 6
  int noLateLoose = 0;    // GX_NO_LATE_LOOSE: a sample whose lambda comes with its end takes the tight table (k_pack_pval), as until round 6
  int noLazyQ = 0;        // GX_NO_LAZY_Q: q of every interval by k_qlookup, as until round 6
  int noPackHist = 0;     // GX_NO_PACK_HIST: BH's histogram by k_bh_hist from the tight table also for a single replicate without control
  int noMergeP = 0;       // GX_NO_MERGE_P: the control merge leaves both pileups in its loose slots and k_pack_pairs scores them, as until round 5
  int forceHalfBins = 0;  // GX_FORCE_HALF_BINS: the 128-key level 1 on a small input
  int noHalfBins = 0;     // GX_NO_HALF_BINS
  int fracHalfBins = 0;   // GX_FRAC_HALF_BINS: half-size bins also for a dense sample with fractional weights (measurements)
  int noEarlyColl = 0;    // GX_NO_EARLY_COLL: no all-reduce of the closed form of fragLen ahead of the tile stage
  int noDenseBh = 0;      // GX_NO_DENSE_BH: the range-partitioned exchange also without a control
  int qtMulti = 0;        // GX_QT_MULTI: the chunked BH table kernels for a small table
  int forceColl = 0;      // GX_FORCE_COLL: run the collectives with a single rank too
  int sbShift = -1;       // GX_SBSHIFT: tiles per super-bucket (log2)
  long long runCapMin = 0;  // GX_RUN_CAP_MIN: the sweep's first guess of the run count (a tiny one forces the second pass)
  int bhCapLog = 0;       // GX_BH_CAPLOG: log2 of the BH table's first size
  int ptJmax = 0;         // GX_PT_JMAX: pages per level-1 list at first
  int sbtTr = 0;          // GX_SBT_TR: 384 / 448: which instance of k_sbtile's dense launch runs (measurements; default: by the sample's density)
  int roctx = 0;          // GX_ROCTX: a roctx range around every phase (rocprofv3 --marker-trace: kernel -> phase attribution)
  int fault = 0;          // GX_FAULT: fault injection for the tests of the device-side invariants.  1: the weight of the ends at
                          // chromosome 0's length is damaged behind level 1 of the sort (as if an end record had been lost)
};
struct KnobDef { const char* name; int Knobs::*i; long long Knobs::*ll; };
const KnobDef KNOBS[] = {
    {"GX_DEBUG", &Knobs::debug, nullptr}, {"GX_DEBUG_RETRY", &Knobs::debugRetry, nullptr}, {"GX_NO_SPIN", &Knobs::noSpin, nullptr},
    {"GX_FORCE_REC64", &Knobs::forceRec64, nullptr}, {"GX_FORCE_SLOWFRAG", &Knobs::forceSlowFrag, nullptr},
    {"GX_NO_FUSED", &Knobs::noFused, nullptr}, {"GX_NO_LOOSE", &Knobs::noLoose, nullptr}, {"GX_NO_PAIRS", &Knobs::noPairs, nullptr},
    {"GX_NO_FRAC_PAIRS", &Knobs::noFracPairs, nullptr}, {"GX_NO_BED_FUSED", &Knobs::noBedFused, nullptr}, {"GX_NO_MERGE_P", &Knobs::noMergeP, nullptr}, {"GX_NO_PACK_HIST", &Knobs::noPackHist, nullptr}, {"GX_NO_LAZY_Q", &Knobs::noLazyQ, nullptr}, {"GX_NO_LATE_LOOSE", &Knobs::noLateLoose, nullptr}, {"GX_NO_Q_LOOSE", &Knobs::noQLoose, nullptr}, {"GX_MERGE_WG", &Knobs::mergeWg, nullptr}, {"GX_BH_VARIANT", &Knobs::bhVariant, nullptr}, {"GX_FORCE_HALF_BINS", &Knobs::forceHalfBins, nullptr},
    {"GX_NO_HALF_BINS", &Knobs::noHalfBins, nullptr}, {"GX_FRAC_HALF_BINS", &Knobs::fracHalfBins, nullptr}, {"GX_NO_EARLY_COLL", &Knobs::noEarlyColl, nullptr},
    {"GX_NO_DENSE_BH", &Knobs::noDenseBh, nullptr}, {"GX_QT_MULTI", &Knobs::qtMulti, nullptr}, {"GX_FORCE_COLL", &Knobs::forceColl, nullptr},
    {"GX_SBSHIFT", &Knobs::sbShift, nullptr}, {"GX_RUN_CAP_MIN", nullptr, &Knobs::runCapMin}, {"GX_BH_CAPLOG", &Knobs::bhCapLog, nullptr},
    {"GX_PT_JMAX", &Knobs::ptJmax, nullptr}, {"GX_FAULT", &Knobs::fault, nullptr}, {"GX_SBT_TR", &Knobs::sbtTr, nullptr},
    {"GX_ROCTX", &Knobs::roctx, nullptr},
};
// a switch that is merely present counts as 1 (GX_NO_LOOSE= is "on", as it was with getenv() != nullptr), and so does a
// value that is not a number (GX_NO_LOOSE=yes)
bool set_knob(Knobs& k, const char* name, const char* value) {
  for (const KnobDef& d : KNOBS)
    if (!strcmp(d.name, name)) {
      long long v = 1;
      if (value && *value) {
        char* end = nullptr;
        v = strtoll(value, &end, 10);
        if (end == value) v = 1;
      }
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
  struct Seg { const gx_event* p; size_t n; hipEvent_t ready; bool packed = false; };   // (packed: 8-byte gx_event8 records behind p)
  std::vector<DevBuf> unpackBufs; // packed pieces as 16-byte events, for the paths that read those (unpack_segs); reused sample after sample
  size_t unpackUsed = 0;
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
  bool sawFrac = false;         // a sample of this context HELD fractional weights (learned: ST_SB_FRAC, Scalars::fracSeen): no closed
                                // form of fragLen, no early lambda, no loose-slot sweep from then on
  bool fracHint = false;        // gx_expect_fractional: only selects the kernels that can carry a weight class (k_sort_a<true>,
                                // k_sbtile<.., true>); on unit-weight data they give what the unit-weight instances give
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
  int denseHistIdx = -1;        // the replicate whose "bp at V" histogram k_pack_pval<.., HIST> left in bhDense (and its deep values in the table)
  bool denseHistUsed = false;   // ... and the last BH table was made from it
  bool mergePUsed = false;      // the last control merge wrote p-values into its loose slots (k_merge2<.., true>)
  DevBuf lbSweep, lbSweep2;     // look-back granules of the sweep's one-pass compactions (generation-tagged)
  u32 sweepGen = 0;
  FragSelect closeSel{};        // the sample's k_frag_select arguments (k_close took them; finish_scalars may need them again)
  u32 closeSeq = 0;             // sequence number of the mail k_close sends (0: the separate kernels were launched)
  bool beginPending = false;    // gx_sample_begin's clearing of the scalars is still to be done (k_build_init / flush_begin)
  u64 beginGenome = 0;
  bool fellBack = false;        // some sample was sent back from k_sbtile to the general chain
  bool ptGrew = false;          // some sample was built again with larger page tables (RETRY_PT)
  bool packedUsed = false;      // the last build read a piece of 8-byte events in place (k_sort_a<.., PACKED>: gx_path_info)
  bool fragFused = false;       // the last build's tile kernel adds the general fragLen path's terms itself (TileIn::fragAcc)
  bool looseOk = false;         // the treatment sample's tile stage left valid sweep bits on the loose slots
  bool riskNearThr = false;     // a re-evaluated table entry lies next to the significance threshold
  size_t looseStride = 0;       // words between the sig / brk masks the tile stage wrote into swMask
  DevBuf tileSlot, chromW0, chromLooseOff, looseCtl;
  u32 ptJmax = 16;              // pages per (XCD class, super-bucket) list; grown after ST_PT_FULL
  DevBuf lbIv;
  Stream str[3];  // S (start keys), E (end keys), F (fractional records)
  DevBuf tileCnt[3], tileOff[3];
  DevBuf looseC, looseC2, pairLogE, pairCtab, pairP2d, fragSum, tileDeep, fragList, zeroArena, endAtLen, binNet, curC, ptC, poolC, auxC, nWide, wideList, heavyList;
  DevBuf tileMeta, tileWsum, tileCarry, lb, misc, dScal, dStatus, looseEnd, looseV, tileIvCount, tileLastEnd, tilePrevEnd;
  Pileup expt, ctrl;
  Scalars hScal{};  // host copy of the device scalars (refreshed from the mail block)
  std::vector<PArray> reps;
  int finalIdx = -1;
  // BH
  DevBuf pvLut, dRisk, dDeep;
  PinnedBuf riskHost;           // count + records of the risky p-values, as read back / as sent with the host's values
  bool pairTabsReady = false;   // the control's p-value tables were built when its sample was closed
  DevBuf bhKeys, bhLens, bhOutKeys, bhOutSlot, bhSortKeys, bhSortSlot, bhQ, bhKQ, bhRaw, bhDl, bhTmp, bhRecs;
  PinnedBuf hostRecs;           // this rank's BH records for the all-gather
  bool satDone = false;         // this sample's events already went through the saturation filter
  long long satDropped = 0;     // ... which dropped this many of them (gx_saturation_dropped)
  bool qLooseBad = false;       // -q on the loose slots was called off once (ST_Q_LOOSE): this context takes the tight table from now on
  bool qLooseUsed = false;      // the last gx_find_peaks: -q, the sweep on the loose slots (GX_PATH_Q_LOOSE)
  DevBuf qLut;                  // ... its q by whole pileup (k_bh_small)
  bool lateLooseUsed = false;   // ... and the replicate the sweep walked was such a sample (GX_PATH_LATE_LOOSE)
  bool lateLoose = false;       // this sample: the sweep's bits on the loose slots come after the table p(V) (k_loose_late)
  bool lazyQUsed = false;       // the last -q run took k_sig_from_p / k_q_fill_cands (GX_PATH_LAZY_Q)
  bool bhLive = false;          // the table of the last -q run is still there, with {key, q} of every value (lazy q: ensure_q)
  u32 bhLiveCap = 0;
  int bhLiveIdx = -1;           // ... and it belongs to reps[bhLiveIdx]
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
  bool roctxOpen = false;       // (GX_ROCTX: the range of the open phase)
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
enum { M_TICKET = 0, M_NIV = 1, M_BHCOUNT = 5, M_ALLONE = 6, M_BHOVF = 7, M_GENOME = 10 /* u64 */, M_NMERGED = 15, M_PSTAR = 14, M_VQ = 12 /* two words: k_bh_small */,
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
// roctx ranges ("roctx ranges per phase"): host-side markers of the profiler, opened at run time like RCCL --
// librocprofiler-sdk-roctx is what rocprofv3 --marker-trace listens to, libroctx64 its predecessor.  Absent: no ranges.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
static const RoctxApi* roctx_api() {
  static RoctxApi api;
  static std::once_flag once;
  std::call_once(once, [&]() {
    for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      api.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      api.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (api.push && api.pop) break;
      api.push = nullptr;
      api.pop = nullptr;
    }
  });
  return api.push ? &api : nullptr;
}
void phase_begin(gx_ctx* ctx, const char* name) {
  if (ctx->knob.roctx)
    if (const RoctxApi* r = roctx_api()) {
      char buf[48];
      snprintf(buf, sizeof buf, "gx:%s", name);
      (void)r->push(buf);
      ctx->roctxOpen = true;
    }
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
  if (ctx->roctxOpen) {
    (void)roctx_api()->pop();
    ctx->roctxOpen = false;
  }
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
      if (r.b == 3 || r.b == 4) return fisher_combine((double)in.a[r.a], (int)in.b[r.a], &rk);
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

}  // namespace
