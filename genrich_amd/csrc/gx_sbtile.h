// gx_sbtile.h -- level 2 of the bucket sort FUSED with the tile stage, for samples without -E regions (unit weights;
// since round 4 also fractional weights, on pair records with a weight class).
//
// Replaces k_scan_bins' consumer chain  k_bucket2p -> k_scan_tiles -> k_tile_meta -> k_tile_fast  (gx_sort.h,
// gx_kernels.h, gx_tile_fast.h) and with it savePileupExpt's two per-base passes (Genrich.c:2197-2273):
// round 2 wrote a super-bucket's keys back to HBM as 16-bit tile offsets (0.34 GB), scanned the per-tile counts
// in two more launches, and then let one wavefront per tile fetch ~130 two-byte keys and a 48-byte descriptor --
// a kernel bounded by the latency of those small loads (28 % of the HBM peak, 58 % of the wave cycles waiting).
// Here ONE workgroup owns a super-bucket (2^sbShift <= 256 tiles) from its level-1 pages to its run-length
// intervals:
//   1  slot descriptors: the bin's sixteen page lists (8 XCD classes x start / end keys) cut into slots of 256
//      keys = one 16-byte load per lane of one wavefront; a slot never crosses a page
//   2  all slots of the bin in flight at once (up to 8 + 8 loads of 16 bytes per lane: the whole bin, ~136 KB,
//      is requested before anything waits), the tiles' chromosome records next to them
//   3  per-tile histogram (LDS atomics: starts in the low half of a word, ends in the high half), one block
//      scan -> where a tile's keys lie, its loose output slot and its carry-in pileup:
//        slot(t)  = start keys before t + end keys before t + t           (as k_tile_meta's)
//        carry(t) = 120 (starts before t - ends before t) - weight of the ends that earlier chromosomes dropped
//      (an end at the chromosome's length has no record: k_sort1 counts them per chromosome, endAtLen)
//   4  the keys are scattered to their tiles' lists in LDS (13-bit values: offset + "is an end"; 92 KB hold 47 K
//      keys, 1.4x what a bin of config 2 holds -- round 5 moved 24 KB from the keys to the wavefronts' scratch, 448
//      touched bases per round instead of 192) and the sixteen wavefronts take tiles from an LDS counter:
//      k_tile_fast's passes (occupancy bitmap -> rank of every touched base -> counters by rank -> 64 touched
//      bases per step) with the keys coming from LDS -- no global load in the tile loop at all
// Pair mode (round 4; the default): level 1 leaves ONE 4-byte record per fragment (gx_sort.h: k_sort_a / k_sort_b), step 1
// reads eight lists instead of sixteen, a pair whose ends share a tile is one atomic in steps 3 and 4.
// A tile also leaves (a) its descriptor for the kernels downstream (k_pack_pval, k_frag_*), (b) when the
// significance threshold is known already (gx_loose.h: lambda from the closed form of fragLen), the sweep's
// significance bits for its intervals, in loose-slot index space, and (c) its unused slots filled with
// zero-length intervals (end = the tile's last end), so that the peak sweep can walk the loose slots as they
// are and k_pack_pval's 1.6 GB round trip disappears from the common run.
// A bin that does not fit (more than SBT_SLOTS slots in a stream, more than SBT_KEYCAP keys: reads
// piled up in one place) raises ST_SB_FULL and leaves empty tiles behind; the host then runs the general chain
// on the same pages.
#pragma once
#include "gx_sort.h"
#include "gx_tile_fast.h"

namespace gx {

// measurement hook (tools/build_variant.sh -DGX_EXP_SBT=n): k_sbtile stops after its loads (1), its histogram and
// scan (2), its scatter (3) and leaves empty tiles behind -- where the kernel's time goes; 4: at once, 5: after the
// lists' lengths (both leave garbage behind: timing only).  0: the product.
#ifndef GX_EXP_SBT
#define GX_EXP_SBT 0
#endif

constexpr int SBT_NT = 1024;
constexpr int SBT_NW = SBT_NT / 64;
constexpr int SBT_TILES = 1 << SBT_MAXSHIFT;          // tiles per super-bucket the LDS tables are made for
constexpr int SBT_K = 8;                               // 16-byte loads per lane and stream (start / end keys)
#ifndef GX_SBT_KP
#define GX_SBT_KP 8
#endif
constexpr int SBT_KP = GX_SBT_KP;                      // ... of pair records kept in registers (pair mode: one stream)
constexpr int SBT_KX = 16;                             // ... and slots per wavefront in all: those beyond SBT_KP (a bin of more than 32 K pairs:
                                                       // reads piled up) are read from global memory by every pass that wants them
static_assert(SBT_KX <= 2 * SBT_K && SBT_KP <= SBT_KX, "the slot tables hold 2 x SBT_K x SBT_NW descriptors");
constexpr u32 SBT_SLOT = 256;                          // keys per slot
constexpr u32 SBT_SLOTS = SBT_K * SBT_NW;              // slots per stream (32 K keys)
#ifndef GX_SBT_TR
#define GX_SBT_TR 448
#endif
constexpr u32 SBT_PEAK_KEYS = 200;                    // keys from which a tile is drawn ahead of the ordinary ones (a peak: ~575; ordinary: ~90)
constexpr u32 SBT_HEAVY = 4096;                        // keys from which a tile is the whole workgroup's (a counter per base), not one wavefront's
constexpr int SBT_MAXR = 8;                            // pair mode: rounds of a super-bucket whose keys do not fit the LDS at once
constexpr u32 SBT_FCAP = 16384;                        // pair mode: singles of a super-bucket (read where they lie, twice)
// Touched bases per round of a tile's passes (TR).  A round walks ALL keys of the tile (rank, range check, counter) and then
// the round's touched bases in steps of 64; a tile with more touched bases than TR takes several rounds.  192 (round 3,
// k_tile_fast's TR_CAP) fits the ordinary tile -- but a tile with a peak on it (hg38 / 50 M fragments: one tile in twelve,
// ~575 keys on ~400 bases, at most 459; an ATAC sample: every tile) then pays three rounds -- 4.06 passes over a tile's keys
// on average instead of 2.65 (counted on the bench's stream).  448 covers them in one.  The price is LDS (6 bytes per base
// and wavefront), which comes out of the key array: the two share what the tables leave of the 160 KiB (sbt_keycap).
//   The ordinary launch is compiled for SBT_TR = 448 (47 K keys: a bin of that sample holds 34 K, the fullest 37 K).  The
// launch of a DENSE sample -- every bin beyond the key array, worked off in rounds of tiles -- exists for 448 and for
// SBT_TR_DENSE = 384 (50 K keys): what costs there is the number of rounds, every one of which reads the bin's records
// again, so the host picks the instance that gives fewer rounds (an ATAC sample with 95 K keys per bin: two rounds with
// 384, three with 448: measured 2.21 against 2.51 ms).  (TR as a run-time value of one instance: 2.48 ms -- the rounds
// kernel sits at 126 VGPRs and spills scalars; constants matter there.)
constexpr int SBT_TR = GX_SBT_TR;
constexpr int SBT_TR_DENSE = 384;
// a wavefront's scratch, in words: occupancy bitmap (+ a dummy word that absorbs the lanes without a key), the words'
// prefix counts (+ a dummy entry that ranks those lanes out of every round), counters and offsets by rank
constexpr int SBT_OCCW = TILE / 32 + 4, SBT_PREW = TILE / 64 + 4;
__host__ __device__ constexpr u32 sbt_tw(u32 tr) { return (u32)(SBT_OCCW + SBT_PREW) + tr + tr / 2; }   // words; 3,488 bytes at 448
constexpr u32 SBT_NOKEY = 0x1000u;                     // "no key": offset 4096 = bit 0 of the dummy bitmap word
static_assert(SBT_TR % 64 == 0, "the last step of a round reads whole wavefronts of counters");
static_assert((1u << PgCfg<u32>::SHIFT) % SBT_SLOT == 0, "a slot never crosses a page");
static_assert(TILE == 4096, "13-bit LDS keys: 12 bits of offset and the stream");
static_assert((SBT_NW * sbt_tw(64)) % 4 == 0 && (SBT_NW * 96) % 4 == 0, "the scratch is cleared by 16-byte stores (TR: a multiple of 64)");
constexpr u32 SBT_LDS_BYTES = 160u * 1024u - 512u;    // what a launch asks for (dynamic; the kernel's few static words come on top): one workgroup per CU

struct SbtLds {
  u32 hist[SBT_TILES];                     // [15:0] start keys, [31:16] end keys of the tile
  u32 startC[SBT_TILES + 1];               // keys (both streams) of the bin before the tile
  int netPref[SBT_TILES];                  // start keys - end keys of the bin before the tile
  u32 cur[2 * SBT_TILES];                  // scatter cursors: starts, ends
  uint4 tinfo[SBT_TILES];                  // what a wavefront needs to start a tile, one 16-byte read: pos0, chromosome length,
                                           // TM_ flags | keys of the tile << 8, carry-in pileup (1/120)
  u32 slotOff[2 * SBT_SLOTS];              // first key of a slot, as an index into its stream's page pool
  u32 slotCnt[2 * SBT_SLOTS];
  u32 pre[2][NXCD + 1];
  __attribute__((aligned(8))) u32 scratch[40];
  u32 work;
  u32 overflow;
  u32 rnd[SBT_MAXR + 1];                   // pair mode: the tiles of round r are rnd[r] .. rnd[r + 1] - 1 (a bin beyond the key array)
  u32 nRounds;
  u32 nHeavy;                              // tiles of the round that the whole workgroup takes (sbt_heavy)
  uint16_t heavy[SBT_TILES];
  int netW[SBT_TILES];                     // fractional pairs: weight (1/120) a tile hands on (ends it receives count negative)
  u32 vsRed[2];                            // loose_vsig's reduction words (its own: tid 0 initialises the others right after)
  // what is left of the 160 KiB, split by TR: k_tile_fast's scratch, one per wavefront -- int [SBT_NW][sbt_tw(TR)] --, then
  // the bin's keys, tile after tile -- uint16_t [sbt_keycap(TR) + 192]: [11:0] offset, [15] end key (+ slack: a tile's
  // first 192 keys are read without a bounds check)
  __attribute__((aligned(16))) int dyn[4];
};
constexpr u32 SBT_DYN_OFF = (u32)offsetof(SbtLds, dyn);
// keys of a super-bucket (both streams) that fit the LDS next to the scratch for TR touched bases per round
__host__ __device__ constexpr u32 sbt_keycap(u32 tr) { return ((SBT_LDS_BYTES - SBT_DYN_OFF - SBT_NW * sbt_tw(tr) * 4u) / 2u - 192u) / 64u * 64u; }
constexpr u32 SBT_KEYCAP = sbt_keycap(SBT_TR);        // ... of the ordinary launch
static_assert(sbt_keycap(192) > sbt_keycap(448) && sbt_keycap(448) >= 40960, "the split of the LDS");

// -E regions on the fused path (round 6; Genrich.c:2185-2263).  Three kinds of tile: (A) `save` on at its first base and no edge
// inside -- the ordinary tile, untouched; (B) inside a region (`save` off, no edge, not its chromosome's last): nothing to emit,
// the tile runs as an inactive one; (C) a tile with an edge, or a chromosome's last tile that ends inside a region (its closing
// interval carries V_MARK): TM_BEDX -- its bin goes to the second launch and the tile to the whole workgroup with a counter per
// base (sbt_heavy), where a base is a breakpoint when it is an edge or (save and its difference != 0), as in k_tile<BED>.
// fragLen keeps its closed form: the sum of the fragments' lengths counts every covered base, the reference only the saved ones
// (2246 adds nothing while `save` is off), so the tile stage sums the pileup over the excluded bases -- a tile of kind (B) from its
// keys alone (a key at offset o with weight w lifts the TILE - o bases from o on), a tile of kind (C) base by base --, exact integers
// in 1/120 units, and takes a wavefront's total off one of the closed form's partial sums (FragFix::fragSum: one atomic per
// wavefront and bin -- 22,000 tiles adding to one word cost the tile stage 0.2 ms).
constexpr u32 TM_BEDX = 16u;   // (in SbtLds::tinfo only; TileMeta::flags keeps the chromosome's state)
constexpr u32 TM_BEDIN = 32u;  // (likewise) kind (B) on a saved chromosome
struct SbtIn {
  PagedStream PS, PE;         // (pair mode: PS = the pair records' lists, PE unused)
  PagedStream PF;             // pair mode: the singles' lists (8-byte signed-weight records)
  const u32* sbOffS;          // [nSeg + 1] start keys in the bins before (k_scan_bins); pair mode: endpoint keys before
  const u32* sbOffE;          //            end keys in the bins before; pair mode: weight (1/120) handed on by the bins before
  const u32* sbOffF;          // (only its total is looked at: any fractional record sends the sample to the general chain)
  const u32* tileChrom;
  const DChrom* chroms;
  const int* chromW0;         // [nChrom] weight (1/120) of the ends dropped at the ends of the chromosomes before
  u32 nSeg, nTiles;
  int sbShift;
  const FragFix* ff;          // fractional pairs: the general fragLen path's switch and accumulator pair (k_tile_fast's TileIn::ff / fragAcc)
  long long* fragAcc;
  BedIn bed;                  // -E regions (bedTileOff == nullptr: none): a tile's slot also counts the edges before it
  u64* fragSum;               // ... and FragFix::fragSum, the closed form of fragLen: the pileup over the excluded bases comes off it
};

struct SbtOut {
  TileOut to;
  TileMeta* meta;
  u32* tileSlot;              // [nTiles + 1] first loose slot of a tile (a compact copy of meta[].slot)
  u32* hot;                   // set when a tile holds enough starts (or ends) for a base to reach the reference's int16 limits
  u32* nBig;                  // pair mode: the bins left to the second launch (k_sbtile<true, true>), and how many
  u32* bigList;
  u32* heavyList;             // fractional pairs: the tiles sbt_heavy took (k_frag_walk adds their fragLen terms), and how many
  u32* nHeavyG;
};

// LDS operations of ONE wavefront execute in order; what is needed between a wavefront's phases is only that
// the compiler keeps them in order (no workgroup barrier: the wavefronts of k_sbtile walk different tiles)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 32-bit byte offsets against a uniform base pointer: one shift instead of 64-bit address arithmetic per store
__device__ __forceinline__ void st_u32(void* base, u32 index, u32 v) {
  *reinterpret_cast<u32*>(static_cast<char*>(base) + (size_t)(index << 2)) = v;
}

// One tile, one wavefront: the passes of k_tile_fast with the keys in LDS (kl[0 .. n): [11:0] offset, [15] end).
// Lanes without a key carry SBT_NOKEY instead of sitting out: their bit goes to a dummy bitmap word and their rank
// (the dummy prefix entry: 0xFFFF) lies outside every round, so the three passes over the register-held keys run
// without per-lane predicates.
// FRAC (fractional pair records, k_sort_a<true>): a key carries its weight class in [14:12] -- weight 120 / count for
// count 1, 2, 3, 4, 5, 6, 8, 10 -- so "no key" moves to bit 16 of the register; with the general fragLen path on
// (FragFix::slow) the tile also adds the exact term of every interval but its first to fhi / flo, as k_tile_fast does.
constexpr u32 SBT_NOKEY_F = 0x10000u;
__device__ __forceinline__ int sbt_weight(u32 key) {  // 1/120 units, signed: [14:12] class, [15] end
  const int w = (int)((0x0C0F14181E283C78ull >> (8 * ((key >> 12) & 7u))) & 0xFFu);
  return (key & 0x8000u) ? -w : w;
}
__device__ __forceinline__ u32 sbt_class_of(int w) {  // weight (> 0) -> class; 8: not one of the eight
  u32 c = 8;
#pragma unroll
  for (u32 k = 0; k < 8; k++) c = (int)((0x0C0F14181E283C78ull >> (8 * k)) & 0xFFu) == w ? k : c;
  return c;
}

template <bool FRAC, bool BED>
__device__ __forceinline__ void sbt_tile(int* lds, const u32 TR_CAP, const uint16_t* __restrict__ kl, u32 n, u32 t, u32 pos0, u32 len, u32 flags,
                                         int carry, u32 slot, int vsig, const SbtOut& out, u32& bad, bool fragTerms, long long& fhi,
                                         long long& flo, long long& bedExcl) {
  constexpr u32 NOKEY = FRAC ? SBT_NOKEY_F : SBT_NOKEY;
  if (BED && (flags & TM_BEDIN)) {  // wave-uniform, -E runs only: a tile inside an excluded region -- no interval, its pileup off the closed form
    long long s = 0;
    for (u32 k = lane_id(); k < n; k += 64) {
      const u32 key = kl[k];
      const int w = FRAC ? sbt_weight(key) : ((key & 0x8000u) ? -GX_UNIT : GX_UNIT);
      s += (long long)w * (long long)(TILE - (key & (TILE - 1)));
    }
    bedExcl += wave_sum(s) + (long long)carry * TILE;
    if (lane_id() == 0) out.to.tileCount[t] = 0;
    return;
  }
  // a key's offset with "no key" at 4096 (the dummy bitmap word)
  auto offOf = [](u32 key) -> u32 { return FRAC ? (key & (TILE - 1)) | ((key >> 4) & (u32)TILE) : key & (2 * TILE - 1); };
  u32* occ = reinterpret_cast<u32*>(lds);
  u32* pre = reinterpret_cast<u32*>(lds + SBT_OCCW);
  const uint16_t* pre16 = reinterpret_cast<const uint16_t*>(pre);
  int* cnt = lds + SBT_OCCW + SBT_PREW;
  uint16_t* list = reinterpret_cast<uint16_t*>(lds + SBT_OCCW + SBT_PREW + TR_CAP);   // (TR_CAP: a constant in the ordinary launch)
  const int lane = lane_id();
  if (!(flags & TM_ACTIVE)) {  // wave-uniform: a tile of a chromosome that is not saved -- nothing to emit, nothing touched
    if (lane == 0) out.to.tileCount[t] = 0;
    return;
  }
  constexpr bool active = true;
  const u32 negPos0 = 0u - pos0;   // (pos0 + p != 0  <=>  p != -pos0)
  const bool lastTile = (flags & TM_LAST) != 0;
#ifndef GX_SBT_KR
#define GX_SBT_KR 2
#endif
  // keys per lane kept in registers: 128 per tile (an ordinary tile of hg38 / 50 M fragments holds ~90, nine in ten at most
  // 128; the third register of round 4 was all "no key" there and cost its three passes: 0.577 -> 0.567 ms.  Skipping it by
  // a scalar branch where the tile is small -- knob 8 -- cost more than it saved: 0.572 -> 0.585)
  constexpr int KR = GX_SBT_KR;
  u32 kr[KR];
#pragma unroll
  for (int q = 0; q < KR; q++) {
    const u32 v = kl[lane + q * 64];  // (in bounds: the key array has 192 entries of slack)
    kr[q] = (u32)lane + q * 64 < n ? v : NOKEY;
  }
  // ---- A1: keys -> occupancy bitmap
  auto mark = [&](u32 key) { const u32 off = offOf(key); atomicOr(&occ[off >> 5], 1u << (off & 31)); };
  // (predicated after all: the lanes without a key would all hit the one dummy word, and same-address LDS atomics
  // serialise -- measured 0.71 -> 0.81 ms for the kernel)
#pragma unroll
  for (int q = 0; q < KR; q++)
    if (kr[q] != NOKEY) mark(kr[q]);
  for (u32 k = KR * 64 + lane; k < n; k += 64) mark(kl[k]);
  wave_lds_sync();
  // ---- B: touched bases before each bitmap word
  const uint2 ww = *reinterpret_cast<const uint2*>(occ + 2 * lane);
  const u32 w0 = ww.x, w1 = ww.y;
  const int c0 = __popc(w0), c = c0 + __popc(w1);
  const int incC = dpp_scan_add(c);
  const u32 exc = (u32)(incC - c);
  const u32 T = (u32)__builtin_amdgcn_readlane(incC, 63);
  pre[lane] = exc | ((exc + (u32)c0) << 16);
  wave_lds_sync();
  int runBase = carry;
  u32 outCount = 0, lastEnd = 0;
  u64 negM = 0, bigM = carry >= FRAG_FAST_MAXV ? ~0ull : 0ull;
  auto rankOf = [&](u32 key) -> u32 {
    const u32 off = offOf(key), wi = off >> 5;
    return (u32)pre16[wi] + (u32)__popc(occ[wi] & ((1u << (off & 31)) - 1u));
  };
  u32 rr[KR];
#pragma unroll
  for (int q = 0; q < KR; q++) rr[q] = rankOf(kr[q]);
  for (u32 r0 = 0; r0 < T; r0 += TR_CAP) {
    if (r0) wave_lds_sync();
    // ---- A2: keys -> cnt[rank], list[rank]
    auto put = [&](u32 r, u32 key) {
      r -= r0;
      if (r < (u32)TR_CAP) {
        list[r] = (uint16_t)(key & (TILE - 1));
        atomicAdd(&cnt[r], FRAC ? sbt_weight(key) : ((key & 0x8000u) ? -GX_UNIT : GX_UNIT));
      }
    };
#pragma unroll
    for (int q = 0; q < KR; q++) put(rr[q], kr[q]);
    for (u32 k = KR * 64 + lane; k < n; k += 64) {
      const u32 key = kl[k];
      put(rankOf(key), key);
    }
    wave_lds_sync();
    // ---- C: 64 touched bases per step (whole wavefronts: the counters behind the last touched base are zero)
    const u32 nL = min((u32)TR_CAP, T - r0);
    auto loadEl = [&](u32 j, u32& p, int& d120) {
      p = list[j];
      d120 = __hip_atomic_exchange(&cnt[j], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // read and clear
    };
    // one step's 64 touched bases: p their offsets, d120 their differences, incS the inclusive scan of the differences
    auto emit = [&](const u32 p, const int d120, const int incS) {
      const int after = runBase + incS;
      const int before = after - d120;                          // the pileup of the interval that ends here (2244)
      // nz = d120 != 0 && pos0 + p != 0 (2241: base 0 closes nothing), as a lane mask made of the compares'
      // own scalar results: __ballot() of a composite condition sends it through a VGPR and back (two more VALU
      // instructions per ballot, four ballots per step)
      const u64 mask = __builtin_amdgcn_uicmp((u32)d120, 0u, 33 /* ne */) & __builtin_amdgcn_uicmp(p, negPos0, 33);
      const bool nz = __builtin_amdgcn_inverse_ballot_w64(mask);
      const u32 orank = __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
      if (nz) {
        const u32 o = slot + outCount + orank;
        st_u32(out.to.looseEnd, o, pos0 + p);
        st_u32(out.to.looseV, o, (u32)before);
      }
      if (vsig != 0x7FFFFFFF)  // wave-uniform
        sig_flush_m(out.to.sigMask, slot + outCount, mask & __builtin_amdgcn_sicmp(before, vsig, 39 /* sge */), orank, mask);
      // (a pileup below zero or at the table's end: one unsigned compare finds either, the rare step that has one says which)
      if (__builtin_amdgcn_uicmp((u32)after, (u32)FRAG_FAST_MAXV, 35 /* uge */)) {  // wave-uniform
        negM |= __ballot(after < 0);
        bigM |= __ballot(after >= FRAG_FAST_MAXV);
      }
      runBase += __builtin_amdgcn_readlane(incS, 63);
      if (FRAC && fragTerms && mask) {  // wave-uniform
        // the interval that ends here starts at the end before it: the lane's, the steps', or -- the tile's first
        // interval -- somewhere before the tile (k_scan_iv adds that one: it knows where)
        const u64 below = mask & ((1ull << lane) - 1ull);
        const int prevLane = below ? 63 - __builtin_clzll(below) : 0;
        const u32 pp = (u32)__shfl((int)p, prevLane, 64);
        if (nz && (below || outCount)) frag_term(pos0 + p - (below ? pos0 + pp : lastEnd), before, fhi, flo);
      }
      if (mask) {  // wave-uniform
        outCount += (u32)__popcll(mask);
        lastEnd = pos0 + (u32)__builtin_amdgcn_readlane((int)p, 63 - __builtin_clzll(mask));
      }
    };
    for (u32 j0 = 0; j0 < nL; j0 += 64) {
      u32 p;
      int d120;
      loadEl(j0 + lane, p, d120);
      emit(p, d120, dpp_scan_add(d120));
    }
  }
  *reinterpret_cast<uint2*>(occ + 2 * lane) = make_uint2(0u, 0u);
  u32 total = 0;
  if (active) {  // wave-uniform
    total = outCount + (lastTile ? 1u : 0u);
    if (lane == 0) {
      if (lastTile) {  // closing interval [.., len): 2268-2273
        const u32 o = slot + outCount;
        out.to.looseEnd[o] = len;
        out.to.looseV[o] = runBase;
        if (runBase >= vsig) atomicOr((unsigned long long*)&out.to.sigMask[o >> 6], 1ull << (o & 63));
        if (FRAC && fragTerms && outCount) frag_term(len - lastEnd, runBase, fhi, flo);  // (the tile's first interval: k_scan_iv)
        lastEnd = len;
      }
      if (total) out.to.tileLastEnd[t] = lastEnd;
    }
    lastEnd = (u32)__builtin_amdgcn_readfirstlane((int)lastEnd);
    if (negM) bad |= ST_NEG_PILE;
    if (bigM && lane == 0) {  // rare
      atomicOr(&out.to.tileDeep[t], 1u);
      if (out.to.ctl) atomicOr(&out.to.ctl->bad, 1u);  // a pileup beyond (or close to the end of) the table p(V)
    }
  }
  if (lane == 0) out.to.tileCount[t] = total;
  // the unused slots of the tile: zero-length intervals behind its last one (a tile without intervals does not
  // know where the previous one ended: k_scan_iv fills its slots)
  // (round 6: also when the sweep's bits come later -- a sample whose lambda is not known yet, LooseCtl handed over all the same)
  if (total && (vsig != 0x7FFFFFFF || out.to.ctl)) {  // wave-uniform
    const u32 size = n + 1;
    for (u32 j = total + lane; j < size; j += 64) {
      st_u32(out.to.looseEnd, slot + j, lastEnd);
      st_u32(out.to.looseV, slot + j, 0u);
    }
  }
  wave_lds_sync();
}

// A tile with thousands of keys (reads piled up on a few bases: a tower, chrM) by the WHOLE workgroup with a counter per
// base, as k_tile_heavy does on the general chain -- one wavefront walking 30,000 keys in rounds of 192 touched bases held
// its workgroup, and the kernel, for a millisecond.  The counters take the place of the wavefronts' scratch (all of
// them are through with their tiles); every thread owns four consecutive bases.  Same outputs as sbt_tile.
// (FRAC with the general fragLen path on: the tile goes on the list of the heavy tiles, whose terms k_frag_walk adds)
template <bool FRAC, bool BED>
__device__ __forceinline__ void sbt_heavy(SbtLds& L, const u32 scrWords, const uint16_t* __restrict__ kl, u32 n, u32 t, u32 pos0, u32 len, u32 flags,
                                          int carry, u32 slot, int vsig, const SbtOut& out, u32& bad, bool fragTerms, const BedIn& bed,
                                          long long& bedExcl) {
  static_assert(SBT_NW * sbt_tw(192) >= TILE, "the counters fit the wavefronts' scratch");
  static_assert(TILE == SBT_NT * 4, "four bases per thread");
  int* cnt = L.dyn;
  const int tid = threadIdx.x;
  const bool active = flags & TM_ACTIVE, lastTile = (flags & TM_LAST) != 0;
  for (u32 i = (u32)tid * 4; i < scrWords; i += SBT_NT * 4) *reinterpret_cast<int4*>(cnt + i) = make_int4(0, 0, 0, 0);
  __syncthreads();
  for (u32 k = tid; k < n; k += SBT_NT) {
    const u32 key = kl[k];
    atomicAdd(&cnt[key & (TILE - 1)], FRAC ? sbt_weight(key) : ((key & 0x8000u) ? -GX_UNIT : GX_UNIT));
  }
  __syncthreads();
  const int4 d4 = *reinterpret_cast<const int4*>(cnt + tid * 4);
  const int d[4] = {d4.x, d4.y, d4.z, d4.w};
  int tot;
  const int ex = block_excl_scan<int, SBT_NT>(d[0] + d[1] + d[2] + d[3], reinterpret_cast<int*>(L.scratch), &tot);
  int run = carry + ex;  // the pileup before this thread's first base
  bool nz[4];
  int before[4];
  u32 mine = 0, neg = 0, big = (u32)(tid == 0 && carry >= FRAG_FAST_MAXV);
  // -E (TM_BEDX): the tile's edges -- a handful, tile-local offsets in ascending order -- are breakpoints whatever the difference
  // holds, a difference inside a region is none, and an interval that ends inside one carries V_MARK (2241-2263, as k_tile<BED>).
  // `save` at this thread's first base = the tile's state ^ the parity of the edges before it.
  const bool bedx = BED && (flags & TM_BEDX) != 0;
  bool save = true, saveEnd = true;
  u32 edgeM = 0;
  if (bedx) {  // block-uniform
    const u32 e0 = bed.bedTileOff[t], e1 = bed.bedTileOff[t + 1];
    u32 pre = 0;
    for (u32 i = e0; i < e1; i++) {
      const u32 off = bed.bedEdge[i];
      pre += (u32)(off < (u32)tid * 4);
      if (off - (u32)tid * 4 < 4u) edgeM |= 1u << (off - (u32)tid * 4);
    }
    const bool s0 = bed.tileSave0[t] != 0;
    save = s0 ^ ((pre & 1u) != 0);
    saveEnd = s0 ^ (((e1 - e0) & 1u) != 0);
  }
  long long excl = 0;   // the pileup over this thread's excluded bases (a base belongs to the region from its start edge on, 2258-2263)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool edge = (edgeM >> j) & 1u;
    before[j] = save ? run : V_MARK;
    run += d[j];
    nz[j] = (edge || (save && d[j] != 0)) && active && (pos0 + (u32)tid * 4 + j != 0);  // 2241: base 0 closes nothing
    if (edge) save = !save;
    if (bedx && active && !save && pos0 + (u32)tid * 4 + j < len) excl += run;
    mine += nz[j];
    neg |= (u32)(run < 0);
    big |= (u32)(run >= FRAG_FAST_MAXV);
  }
  if (bedx) bedExcl += wave_sum(excl);  // block-uniform (every wavefront keeps its own total)
  u32 outCount;
  u32 o = slot + block_excl_scan<u32, SBT_NT>(mine, L.scratch, &outCount);
  u32 lastPos = 0;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (nz[j]) {
      const u32 p = pos0 + (u32)tid * 4 + j;
      out.to.looseEnd[o] = p;
      out.to.looseV[o] = before[j];
      if (before[j] >= vsig) atomicOr((unsigned long long*)&out.to.sigMask[o >> 6], 1ull << (o & 63));  // (vsig: INT_MAX when no bits are wanted)
      lastPos = p;
      o++;
    }
  // the last interval's end: the largest position written (positions grow with the thread index)
  u32 lastEnd;
  {
    u32 m = lastPos;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) m = max(m, (u32)__shfl_xor((int)m, dd, 64));
    __syncthreads();
    if (lane_id() == 0) L.scratch[tid >> 6] = m;
    __syncthreads();
    m = 0;
    for (int w = 0; w < SBT_NW; w++) m = max(m, L.scratch[w]);
    lastEnd = m;
    __syncthreads();
  }
  const int endRun = carry + tot;  // the pileup behind the tile's last base
  u32 total = 0;
  if (active) {
    total = outCount + (lastTile ? 1u : 0u);
    if (tid == 0) {
      if (lastTile) {  // closing interval [.., len): 2268-2273
        const u32 oc = slot + outCount;
        out.to.looseEnd[oc] = len;
        out.to.looseV[oc] = saveEnd ? endRun : V_MARK;   // (a chromosome that ends inside a -E region: 2268-2273 with save off)
        if (saveEnd && endRun >= vsig) atomicOr((unsigned long long*)&out.to.sigMask[oc >> 6], 1ull << (oc & 63));
      }
      if (total) out.to.tileLastEnd[t] = lastTile ? len : lastEnd;
    }
    if (lastTile) lastEnd = len;
    if (neg) bad |= ST_NEG_PILE;
    if (__syncthreads_or((int)big) && tid == 0) {
      atomicOr(&out.to.tileDeep[t], 1u);
      if (out.to.ctl) atomicOr(&out.to.ctl->bad, 1u);  // a pileup beyond (or close to the end of) the table p(V)
    }
  }
  if (tid == 0) out.to.tileCount[t] = total;
  if (FRAC && fragTerms && tid == 0) out.heavyList[atomicAdd(out.nHeavyG, 1u)] = t;
  if (total && (vsig != 0x7FFFFFFF || out.to.ctl))  // the unused slots: zero-length intervals behind the last one (the sweep walks the loose slots)
    for (u32 j = total + tid; j < n + 1; j += SBT_NT) {
      out.to.looseEnd[slot + j] = lastEnd;
      out.to.looseV[slot + j] = 0;
    }
  __syncthreads();
}

// PAIRS: level 1 was k_sort1p (gx_sort.h): one 4-byte record per fragment (start within the bin, length) in PS, the few
// other records ("singles") as 8-byte signed-weight records in PF -- half the bytes to load, one LDS atomic per fragment
// in the histogram and in the scatter where both ends share a tile (19 of 20).
// BIG (pair mode only): the second launch, for the few bins the first one put on its list -- more keys than the key
// array holds (worked off in rounds of tiles), a tile with thousands of keys (sbt_heavy: the whole workgroup), more than
// 32 K pair records.  It keeps no record in registers (every pass reads the bin's slots from global memory: they are
// in L2), so that none of this costs the first launch -- the one every bin of an ordinary sample takes -- a register.
template <bool PAIRS, bool BIG, bool FRAC, int TRC, bool BED>
__device__ __forceinline__ void sbt_bin(const SbtIn& in, const SbtOut& out, u32* __restrict__ st, const u32 seg, SbtLds& L) {
  static_assert(PAIRS || !BIG, "the second launch exists in pair mode only");
  static_assert(PAIRS || !FRAC, "fractional weights ride pair records only");
  static_assert(!BED || (PAIRS && !FRAC), "-E regions: unit-weight pair records (the tiles with an edge are the second launch's)");
  constexpr u32 LENB = FRAC ? 9u : PAIR_LEN_BITS;      // a pair record's length bits (fractional: [11:9] the weight class)
  const bool fragTerms = FRAC && in.fragAcc != nullptr && (u32)__builtin_amdgcn_readfirstlane((int)in.ff->slow) != 0u;
  long long fhi = 0, flo = 0, bedExcl = 0;
  constexpr int K = BIG ? 0 : (PAIRS ? SBT_KP : SBT_K);   // slots per wavefront of the (first) stream held in registers
  constexpr int KX = BIG ? SBT_KX : K;                    // ... and in all
  constexpr int KR = K ? K : 1;                           // (array sizes)
  constexpr u32 NSLOTS = (u32)KX * SBT_NW;
  const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
  // how the LDS behind the tables is split: constants of the instance (runtime values cost the rounds launch 12 %: measured)
  static_assert(TRC % 64 == 0 && TRC >= 192 && TRC <= 448, "touched bases per round");
  constexpr u32 trCap = (u32)TRC;
  constexpr u32 tw = sbt_tw(trCap), scrWords = SBT_NW * tw, keyCap = sbt_keycap(trCap);
  int* const scr = L.dyn;
  uint16_t* const keysL = reinterpret_cast<uint16_t*>(L.dyn + scrWords);
  const u32 nSeg = in.nSeg;
  const u32 nT = 1u << in.sbShift;           // tiles per super-bucket (<= SBT_TILES)
  const u32 segTileBase = seg << in.sbShift;
  const int vsig = (int)__builtin_amdgcn_readfirstlane(loose_vsig(out.to.ctl, !BIG && seg == 0 && tid == 0, L.vsRed));
  // (one workgroup per CU: a global round trip in this prologue is a round trip of the whole CU.  The lists' lengths and the
  // tiles' chromosome records -- two dependent loads -- are asked for first and arrive while the scratch is cleared)
  u32 myLen = 0;
  auto loadTi = [&]() {   // the tiles' chromosome records (thread b: tile b of the bin)
    uint4 ti = make_uint4(0u, 0u, 0u, 0u);
    if (tid < (int)nT) {
      const u32 t = segTileBase + tid;
      if (t < in.nTiles) {
        const u32 ci = in.tileChrom[t];
        const DChrom c = in.chroms[ci];
        const u32 tl = t - c.tileBase;
        ti.x = tl << TB;
        ti.y = c.len;
        ti.z = (chrom_active(c) ? TM_ACTIVE : 0u) | (tl + 1 == c.nTiles ? TM_LAST : 0u);
        ti.w = (u32)in.chromW0[ci];
      }
    }
    return ti;
  };
  // -E regions: the edges before this thread's tile (they take loose slots of their own) and the tile's kind (TM_BEDX: an edge
  // inside, or a chromosome's last tile that ends inside a region; `inside`: nothing of the tile is saved)
  constexpr bool hasBed = BED;   // (instances of their own: the plumbing cost the run without regions 0.012 ms as a run-time switch)
  u32 bedBefore = 0, bedMine = 0;
  bool bedInside = false;
  auto loadBed = [&](const uint4& ti) {
    if (hasBed && tid < (int)nT && segTileBase + tid < in.nTiles) {
      const u32 t = segTileBase + tid;
      bedBefore = in.bed.bedTileOff[t];
      bedMine = in.bed.bedTileOff[t + 1] - bedBefore;
      const bool s0 = in.bed.tileSave0[t] != 0;
      bedInside = !bedMine && !s0 && !(ti.z & TM_LAST);
      if (bedInside && (ti.z & TM_ACTIVE)) return TM_BEDIN;
      if (bedMine || (!s0 && (ti.z & TM_LAST))) return TM_BEDX;
    }
    return 0u;
  };
  auto expLeave = [&]() {   // (measurement exits: empty tiles with valid slots behind them)
    if (tid < (int)nT && segTileBase + tid < in.nTiles) {
      const u32 t = segTileBase + tid;
      TileMeta m{};
      m.slot = t;
      out.meta[t] = m;
      out.tileSlot[t] = t;
      if (t + 1 == in.nTiles) out.tileSlot[in.nTiles] = in.nTiles;
      out.to.tileCount[t] = 0;
    }
  };
  if (GX_EXP_SBT == 4) { expLeave(); return; }   // (measurement: what launching the workgroups costs)
  if (tid < 2 * NXCD) {
    const u32 li = (u32)(tid & (NXCD - 1)) * nSeg + seg;
    if (PAIRS)
      myLen = tid < NXCD ? list_len<u32>(in.PS, li) : list_len<u64>(in.PF, li);
    else
      myLen = list_len<u32>(tid < NXCD ? in.PS : in.PE, li);
  }
  uint4 ti = loadTi();
  const u32 bedx = loadBed(ti);
  // scratch and tables start at zero
  for (u32 i = (u32)tid * 4; i < scrWords; i += SBT_NT * 4) *reinterpret_cast<int4*>(scr + i) = make_int4(0, 0, 0, 0);
  if (tid < SBT_TILES) {
    L.hist[tid] = 0;
    if (FRAC) L.netW[tid] = 0;
  }
  if (tid == 0) { L.overflow = 0; L.nHeavy = 0; L.nRounds = 0; }
  if (tid < 2 * NXCD) L.scratch[tid] = myLen;
  __syncthreads();
  if (tid < SBT_NW) scr[(u32)tid * tw + SBT_OCCW + TILE / 64] = -1;  // the prefix entry of the dummy bitmap word: no rank at all
  if (tid < 2) {
    u32 a = 0;
    for (int x = 0; x < NXCD; x++) {
      L.pre[tid][x] = a;
      a += L.scratch[tid * NXCD + x];
    }
    L.pre[tid][NXCD] = a;
  }
  __syncthreads();
  if (GX_EXP_SBT == 5) { expLeave(); return; }   // (measurement: ... and clearing the scratch, the lists' lengths)
  // ---- 1: slot descriptors (thread k of the first 2 SBT_SLOTS: slot k & 127 of stream k >> 7)
  if (tid < (PAIRS ? 1 : 2) * (int)NSLOTS) {
    const int q = tid / (int)NSLOTS;
    const u32 k = (u32)tid % NSLOTS;
    const PagedStream& P = q ? in.PE : in.PS;
    const u32* pre = L.pre[q];
    u32 x = NXCD, k0 = 0, acc = 0;
    for (u32 y = 0; y < NXCD; y++) {
      const u32 ns = (pre[y + 1] - pre[y] + SBT_SLOT - 1) / SBT_SLOT;
      if (x == NXCD && k < acc + ns) {
        x = y;
        k0 = acc;
      }
      acc += ns;
    }
    if (k == 0 && acc > NSLOTS) L.overflow = 1;  // more slots than descriptors: the bin does not fit
    u32 ptr = 0;
    u32 cnt = 0;
    if (x < NXCD) {
      const u32 off = (k - k0) * SBT_SLOT;
      const u32 jp = off >> PgCfg<u32>::SHIFT, li = x * nSeg + seg;
      const u32 page = jp ? P.pt[(size_t)li * P.jmax + jp] - 1u : first_page(li);
      ptr = (page << PgCfg<u32>::SHIFT) + (off & ((1u << PgCfg<u32>::SHIFT) - 1u));
      cnt = min(SBT_SLOT, pre[x + 1] - pre[x] - off);
    }
    L.slotOff[tid] = ptr;
    L.slotCnt[tid] = cnt;
  }
  __syncthreads();
  const bool ovfSlots = L.overflow != 0;
  // ---- 2: the bin's keys, all loads in flight together
  const uint4* __restrict__ poolS = reinterpret_cast<const uint4*>(in.PS.pool);
  const uint4* __restrict__ poolE = reinterpret_cast<const uint4*>(in.PE.pool);
  uint4 kS[KR], kE[PAIRS ? 1 : KR];
  u32 cS[KR], cE[PAIRS ? 1 : KR];
#pragma unroll
  for (int i = 0; i < K; i++)  // (wave-uniform: scalar registers)
    cS[i] = ovfSlots ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)L.slotCnt[i * SBT_NW + wv]);
#pragma unroll
  for (int i = 0; i < (PAIRS ? 1 : K); i++)
    cE[i] = ovfSlots || PAIRS ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)L.slotCnt[NSLOTS + i * SBT_NW + wv]);
#pragma unroll
  for (int i = 0; i < K; i++) {
    kS[i] = make_uint4(0u, 0u, 0u, 0u);
    if ((u32)lane * 4 < cS[i]) kS[i] = poolS[(L.slotOff[i * SBT_NW + wv] >> 2) + lane];
  }
#pragma unroll
  for (int i = 0; i < (PAIRS ? 1 : K); i++) {
    kE[i] = make_uint4(0u, 0u, 0u, 0u);
    if (!PAIRS && (u32)lane * 4 < cE[i]) kE[i] = poolE[(L.slotOff[NSLOTS + i * SBT_NW + wv] >> 2) + lane];
  }
  // pair mode: the bin's singles (a handful; 8-byte records of weight +-120 = a start / an end key; anything else is a
  // fractional weight: the sample goes to the general chain) are read where they lie, once per pass
  const BinSrc<u64> srcF{reinterpret_cast<const u64*>(in.PF.pool), in.PF.pt, nSeg, in.PF.jmax, seg, L.pre[1]};
  const u32 nF = PAIRS && !ovfSlots ? L.pre[1][NXCD] : 0u;
  // (round 6: a thread's first single is asked for HERE, with the records, and kept for both passes -- read in the histogram's
  // and in the scatter's loop each was a global round trip of its own, with the whole CU waiting: one workgroup per CU)
#ifndef GX_SBT_FREG
#define GX_SBT_FREG 1
#endif
  u64 fReg = 0;
  if (GX_SBT_FREG && PAIRS && (u32)tid < nF) fReg = srcF.at((u32)tid);
  auto singleAt = [&](u32 i) -> u64 { return GX_SBT_FREG && i == (u32)tid ? fReg : srcF.at(i); };
  // (16-bit counts per tile: the bin's pairs and singles together stay below 2^16)
  if (PAIRS && tid == 0 && (nF > SBT_FCAP || L.pre[0][NXCD] + nF > 65535u)) L.overflow = 1;  // (read behind the histogram's barrier)
  if (GX_EXP_SBT == 1) {
    u32 x = 0;
#pragma unroll
    for (int i = 0; i < K; i++) x ^= kS[i].x ^ kS[i].y ^ kS[i].z ^ kS[i].w ^ kE[PAIRS ? 0 : i].x;
    if (x == 0xDEADBEEFu) atomicOr(st, 1u << 30);
  }
  // ---- 3: per-tile histogram
  auto keyAt = [](const uint4& v, int j) -> u32 { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
  // every record of this wavefront's slots i0 .. KX - 1, read from global memory four slots at a time
  auto slotsFrom = [&](int i0, auto&& f) {
    for (int i = i0; i < KX; i += 4) {
      u32 c[4];
      uint4 v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        c[q] = i + q < KX ? (u32)__builtin_amdgcn_readfirstlane((int)L.slotCnt[(i + q) * SBT_NW + wv]) : 0u;
        v[q] = make_uint4(0u, 0u, 0u, 0u);
        if ((u32)lane * 4 < c[q]) v[q] = poolS[(L.slotOff[(i + q) * SBT_NW + wv] >> 2) + lane];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!c[q]) continue;  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; j++)
          if ((u32)lane * 4 + j < c[q]) f(keyAt(v[q], j));
      }
    }
  };
  // (a slot is full -- 256 keys -- unless it is the last of its list: the full ones without per-key predicates;
  // the counts are wave-uniform)
  // (pair record: [31:12] start within the bin, [11:0] length; both ends lie in this bin)
  auto pairTs = [](u32 r) -> u32 { return r >> (PAIR_LEN_BITS + TB); };
  auto pairEnd = [](u32 r) -> u32 { return (r >> PAIR_LEN_BITS) + (r & ((1u << LENB) - 1u)); };
  auto pairCls = [](u32 r) -> u32 { return FRAC ? ((r >> 9) & 7u) << 12 : 0u; };  // (where a key carries it)
  auto fCls = [](u64 r) -> u32 {  // a single's weight class, where a key carries it
    const int w = (int)(int8_t)(r & 0xFF);
    return FRAC ? (sbt_class_of(w < 0 ? -w : w) & 7u) << 12 : 0u;
  };
  auto histPair = [&](u32 r) {
    const u32 ts = pairTs(r), te = pairEnd(r) >> TB;
    atomicAdd(&L.hist[ts], ts == te ? 65537u : 1u);
    if (ts != te) {
      atomicAdd(&L.hist[te], 65536u);
      if (FRAC) {  // (the pileup a tile hands on: only what starts in one tile and ends in another changes it)
        const int w = sbt_weight(pairCls(r));
        atomicAdd(&L.netW[ts], w);
        atomicAdd(&L.netW[te], -w);
      }
    }
  };
  if (PAIRS) {
#pragma unroll
    for (int i = 0; i < K; i++) {
      if (cS[i] == SBT_SLOT) {
#pragma unroll
        for (int j = 0; j < 4; j++) histPair(keyAt(kS[i], j));
      } else if (cS[i]) {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if ((u32)lane * 4 + j < cS[i]) histPair(keyAt(kS[i], j));
      }
    }
    // (the second launch: no record in registers -- the slots are read four at a time, their loads in flight together)
    if (!ovfSlots) slotsFrom(K, [&](u32 r) { histPair(r); });
    if (nF <= SBT_FCAP)
      for (u32 i = tid; i < nF; i += SBT_NT) {
        const u64 r = singleAt(i);
        const int w = (int)(int8_t)(r & 0xFF);
        const bool ok = FRAC ? sbt_class_of(w < 0 ? -w : w) < 8u : (w == GX_UNIT || w == -GX_UNIT);
        if (ok) {
          atomicAdd(&L.hist[(u32)(r >> 32) - segTileBase], w > 0 ? 1u : 65536u);
          if (FRAC) atomicAdd(&L.netW[(u32)(r >> 32) - segTileBase], w);
        } else
          L.overflow = 2;
      }
  } else if constexpr (!PAIRS) {
  if (GX_EXP_SBT != 1)
#pragma unroll
  for (int i = 0; i < SBT_K; i++) {
    if (cS[i] == SBT_SLOT) {
#pragma unroll
      for (int j = 0; j < 4; j++) atomicAdd(&L.hist[(keyAt(kS[i], j) >> TB) - segTileBase], 1u);
    } else if (cS[i]) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((u32)lane * 4 + j < cS[i]) atomicAdd(&L.hist[(keyAt(kS[i], j) >> TB) - segTileBase], 1u);
    }
    if (cE[i] == SBT_SLOT) {
#pragma unroll
      for (int j = 0; j < 4; j++) atomicAdd(&L.hist[(keyAt(kE[i], j) >> TB) - segTileBase], 65536u);
    } else if (cE[i]) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((u32)lane * 4 + j < cE[i]) atomicAdd(&L.hist[(keyAt(kE[i], j) >> TB) - segTileBase], 65536u);
    }
  }
  }
  __syncthreads();
  {
    u32 h = tid < (int)nT ? L.hist[tid] : 0u;
    const u32 nS = h & 0xFFFFu, nE = h >> 16;
    // (what the tiles before hand on: starts - ends in records; fractional pairs: the net weight itself)
    const u64 v = (u64)(nS + nE) | ((u64)(u32)(FRAC ? (tid < (int)nT ? L.netW[tid] : 0) : (int)nS - (int)nE) << 32);
    u64 tot;
    const u64 ex = block_excl_scan<u64, SBT_NT>(v, reinterpret_cast<u64*>(L.scratch), &tot);
    if (tid < (int)nT) {
      L.startC[tid] = (u32)ex;
      L.netPref[tid] = (int)(u32)(ex >> 32);
    }
    if (tid == 0) L.startC[nT] = (u32)tot;
  }
  __syncthreads();
  if constexpr (PAIRS && !BIG) {
    // what this launch does not take: it goes on the list of the second one, untouched
    const u32 hh = tid < (int)nT ? L.hist[tid] : 0u;
    const bool heavyTile = (hh & 0xFFFFu) + (hh >> 16) > SBT_HEAVY || (bedx & TM_BEDX);   // (a tile with a -E edge: the whole workgroup's too)
    const bool big = __syncthreads_or((int)heavyTile) || ovfSlots || L.startC[nT] > keyCap || L.overflow == 1;
    if (big) {  // block-uniform
      if (tid == 0) out.bigList[atomicAdd(out.nBig, 1u)] = seg;
      return;
    }
  }
  // Pair mode: a bin with more keys than the key array holds (reads piled up: a tower, chrM) is worked off in ROUNDS of
  // consecutive tiles whose keys fit -- the records stay in their registers, every round scatters the keys of its tiles
  // only.  What still does not fit: more records than the slots take, a single tile beyond the array, too many rounds.
  if (BIG && tid == 0) {
    u32 nr = 0;
    bool fits = true;
    if (L.startC[nT] <= keyCap) {
      L.rnd[0] = 0;
      L.rnd[1] = nT;
      nr = 1;
    } else if (!ovfSlots) {
      u32 tb = 0;
      L.rnd[0] = 0;
      while (tb < nT && fits) {
        // the last tile whose keys still fit behind tb's (startC grows: a bisection)
        u32 lo = tb, hi = nT;
        const u32 lim = L.startC[tb] + keyCap;
        while (lo < hi) {
          const u32 mid = (lo + hi + 1) >> 1;
          if (L.startC[mid] <= lim) lo = mid; else hi = mid - 1;
        }
        const u32 te = lo;
        if (te == tb || nr == (u32)SBT_MAXR) fits = false;
        else {
          L.rnd[++nr] = te;
          tb = te;
        }
      }
    }
    L.nRounds = fits ? nr : 0u;
  }
  if (BIG) __syncthreads();
  const u32 nRounds = BIG ? L.nRounds : 1u;
  const u32 ovfWord = PAIRS ? L.overflow : 0u;  // (pair mode: 1 too many singles, 2 a fractional weight among them)
  const bool ovfReal = ovfSlots || (BIG ? nRounds == 0 : L.startC[nT] > keyCap) || ovfWord != 0;
  const bool ovf = ovfReal || GX_EXP_SBT == 1 || GX_EXP_SBT == 2;
  // (the slot capacity bounds a stream at 32 K keys -- pair mode: 32 K pairs and SBT_FCAP singles --, so a tile's 16-bit
  // counts cannot have wrapped)
  if (ovfReal && tid == 0) atomicOr(st, ovfWord == 2 ? ST_SB_FRAC : ST_SB_FULL);
  if (!PAIRS && seg == 0 && tid == 0 && in.sbOffF[nSeg] != 0) atomicOr(st, ST_SB_FRAC);
  // descriptors for the kernels downstream, scatter cursors
  const u32 segS = in.sbOffS[seg], segE = in.sbOffE[seg];
  const u32 segSlot = PAIRS ? segS + segTileBase : segS + segE + segTileBase;
  // weight (1/120) that the bins before hand on
  const int segNet = PAIRS ? (int)segE : GX_UNIT * ((int)segS - (int)segE);
  if (tid < (int)nT) {
    const u32 t = segTileBase + tid;
    const u32 h = ovf ? 0u : L.hist[tid];
    const u32 nS = h & 0xFFFFu, nE = h >> 16;
    const u32 sc = ovf ? 0u : L.startC[tid];
    L.cur[tid] = sc;
    L.cur[SBT_TILES + tid] = sc + nS;
    // (a base can only reach the reference's int16 limits where 32,766 starts, or ends, share a tile: the host then
    // replays the events -- gx_saturate.h; a bin worked off in rounds can hold that many)
    if (nS >= HOT16 / GX_UNIT || nE >= HOT16 / GX_UNIT) atomicOr(out.hot, 1u);
    if (t < in.nTiles) {
      TileMeta m;
      m.sb = 0; m.eb = 0; m.fb = 0;
      m.nS = nS; m.nE = nE; m.nF = 0;
      m.carry = ovf ? 0 : segNet + (FRAC ? 1 : GX_UNIT) * L.netPref[tid] - (int)ti.w;
      m.ci = 0;
      m.pos0 = ti.x; m.len = ti.y; m.flags = ti.z;
      m.slot = segSlot + sc + (u32)tid + bedBefore;   // (a tile holds <= records + edges + 1 intervals)
      out.meta[t] = m;
      // (what the tile passes see: a tile inside a -E region runs as an inactive one, a tile with an edge is the workgroup's)
      L.tinfo[tid] = make_uint4(ti.x, ti.y, ((bedInside ? ti.z & ~TM_ACTIVE : ti.z) | bedx) | ((nS + nE) << 8), (u32)m.carry);
      static_assert((TM_BEDX | TM_BEDIN) < 256u, "the tile's flags take the low byte, its key count the rest");
      if (hasBed) L.netPref[tid] = (int)bedBefore;   // (its prefix has gone into the carry: the word now holds the slot's extra)
      out.tileSlot[t] = m.slot;
      if (t + 1 == in.nTiles) out.tileSlot[in.nTiles] = m.slot + nS + nE + 1 + bedMine;
    }
  }
  if constexpr (PAIRS && !BIG) {
    // the order in which the wavefronts draw the tiles (round 6): the ones with a peak -- one in twelve, several times an ordinary
    // tile's keys -- first, so that none of them is the last thing a wavefront starts while the others have run out of work
    // (tile stage 0.572 -> 0.556 ms at config 2).  L.heavy / L.nHeavy / L.nRounds are the second launch's: free here.
    if (tid < (int)nT && !ovf) {
      const u32 hA = L.hist[tid];
      const u32 nk = (hA & 0xFFFFu) + (hA >> 16);
      const u32 pos = nk > SBT_PEAK_KEYS ? atomicAdd(&L.nHeavy, 1u) : nT - 1u - atomicAdd(&L.nRounds, 1u);
      L.heavy[pos] = (uint16_t)tid;
    }
  }
  if (ovf) {
    // leave empty tiles behind (the host repeats the sample on the general chain)
    if (tid < (int)nT && segTileBase + tid < in.nTiles) out.to.tileCount[segTileBase + tid] = 0;
    return;
  }
  u32 bad = 0;
  // ---- 4: the keys to their tiles' lists in LDS; then the wavefronts take tiles from a counter
  if (tid == 0) L.work = 0;
  __syncthreads();  // the cursors are there
  auto place = [&](u32 key, u32 curBase, u32 flag) {
    keysL[atomicAdd(&L.cur[curBase + (key >> TB) - segTileBase], 1u)] = (uint16_t)((key & (TILE - 1)) | flag);
  };
  // the wavefronts take the tiles [L.work .. tileEnd) from a counter; keyBase: where the first key in LDS lies in the bin's order
  auto tiles = [&](u32 tileEnd, u32 keyBase) {
    // (from a counter, not dealt in turn: one tile in twelve carries a peak and costs several ordinary ones -- a static deal
    // measured 0.628 against 0.590 ms, the workgroup waits for the wavefront that drew three of them)
    for (;;) {
      u32 b = 0;
      if (lane == 0) b = atomicAdd(&L.work, 1u);
      b = (u32)__builtin_amdgcn_readfirstlane((int)b);
      if (b >= tileEnd) break;
      if constexpr (PAIRS && !BIG) b = (u32)__builtin_amdgcn_readfirstlane((int)L.heavy[b]);   // (peaks first)
      const u32 t = segTileBase + b;
      if (t >= in.nTiles) continue;
      const uint4 tf = L.tinfo[b];
      const u32 sc = L.startC[b];
      const u32 fz = (u32)__builtin_amdgcn_readfirstlane((int)tf.z), n = fz >> 8;
      // (a tile inside a -E region stays with its wavefront however many keys it holds: sbt_tile sums them, nothing is emitted --
      // sbt_heavy would see an inactive tile and leave its pileup in the closed form of fragLen)
      if (BIG && !(BED && (fz & TM_BEDIN)) && (n > SBT_HEAVY || (BED && (fz & TM_BEDX)))) {  // wave-uniform: left to the whole workgroup
        if (lane == 0) L.heavy[atomicAdd(&L.nHeavy, 1u)] = (uint16_t)b;
        continue;
      }
      const u32 bedSlots = hasBed ? (u32)__builtin_amdgcn_readfirstlane(L.netPref[b]) : 0u;
      sbt_tile<FRAC, BED>(scr + (u32)__builtin_amdgcn_readfirstlane(wv) * tw, trCap, keysL + (sc - keyBase), n, t, tf.x, tf.y, fz & 0xFFu, (int)tf.w,
                     segSlot + sc + b + bedSlots, vsig, out, bad, fragTerms, fhi, flo, bedExcl);
    }
    if constexpr (BIG) {
    __syncthreads();  // (every wavefront is through with its tiles; the list of the heavy ones is complete)
    const u32 nH = L.nHeavy;
    if (nH) {  // block-uniform
      for (u32 i = 0; i < nH; i++) {
        const u32 b = L.heavy[i], t = segTileBase + b;
        const uint4 tf = L.tinfo[b];
        const u32 sc = L.startC[b], n = tf.z >> 8;
        sbt_heavy<FRAC, BED>(L, scrWords, keysL + (sc - keyBase), n, t, tf.x, tf.y, tf.z & 0xFFu, (int)tf.w,
                        segSlot + sc + b + (hasBed ? (u32)L.netPref[b] : 0u), vsig, out, bad, fragTerms, in.bed, bedExcl);
      }
      // the wavefronts' scratch as the next round's tiles expect it
      for (u32 i = (u32)tid * 4; i < scrWords; i += SBT_NT * 4) *reinterpret_cast<int4*>(scr + i) = make_int4(0, 0, 0, 0);
      __syncthreads();
      if (tid < SBT_NW) scr[(u32)tid * tw + SBT_OCCW + TILE / 64] = -1;
      if (tid == 0) L.nHeavy = 0;
      __syncthreads();
    }
    }
  };
  if constexpr (PAIRS && !BIG) {
    {
      // one cursor per tile (starts and ends of a tile share its list: a key says which it is); a pair whose ends share a
      // tile takes its two places with one atomic
      auto placePair = [&](u32 r) {
        const u32 e = pairEnd(r), ts = pairTs(r), te = e >> TB;
        const u32 so = ((r >> PAIR_LEN_BITS) & (TILE - 1)) | pairCls(r), eo = (e & (TILE - 1)) | 0x8000u | pairCls(r);
        const u32 ps = atomicAdd(&L.cur[ts], ts == te ? 2u : 1u);
        const u32 pe = ts == te ? ps + 1u : atomicAdd(&L.cur[te], 1u);
        keysL[ps] = (uint16_t)so;
        keysL[pe] = (uint16_t)eo;
      };
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (cS[i] == SBT_SLOT) {
#pragma unroll
          for (int j = 0; j < 4; j++) placePair(keyAt(kS[i], j));
        } else if (cS[i]) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if ((u32)lane * 4 + j < cS[i]) placePair(keyAt(kS[i], j));
        }
      }
      for (u32 i = tid; i < nF; i += SBT_NT) {
        const u64 r = singleAt(i);
        const u32 off = (u32)(r >> 8) & (TILE - 1);
        keysL[atomicAdd(&L.cur[(u32)(r >> 32) - segTileBase], 1u)] = (uint16_t)(off | ((r & 0x80) ? 0x8000u : 0u) | fCls(r));
      }
      __syncthreads();
      if (GX_EXP_SBT == 3) {
        if (tid < (int)nT && segTileBase + tid < in.nTiles) out.to.tileCount[segTileBase + tid] = keysL[L.startC[tid]] == 0xFFFFu;
        return;
      }
      tiles(nT, 0u);
    }
  } else if constexpr (BIG) {
    {
      // Rounds (one, when only a heavy tile brought the bin here).  Every round loads the bin's records -- they are in L2 -- and
      // scatters the ends that lie in its tiles; the cursors count in the bin's order, the key array from the round's first
      // key.  (Keeping the records in their registers across the tile loop cost the common path its registers.)
      for (u32 round = 0; round < nRounds; round++) {
        const u32 tileBeg = L.rnd[round], tileEnd = L.rnd[round + 1], keyBase = L.startC[tileBeg];
        if (round) {
          __syncthreads();  // (the previous round's tiles are through with the keys)
          if (tid == 0) L.work = tileBeg;
          __syncthreads();
        }
        slotsFrom(0, [&](u32 r) {
          const u32 e = pairEnd(r), ts = pairTs(r), te = e >> TB;
          if (ts - tileBeg < tileEnd - tileBeg) keysL[atomicAdd(&L.cur[ts], 1u) - keyBase] = (uint16_t)(((r >> PAIR_LEN_BITS) & (TILE - 1)) | pairCls(r));
          if (te - tileBeg < tileEnd - tileBeg) keysL[atomicAdd(&L.cur[te], 1u) - keyBase] = (uint16_t)((e & (TILE - 1)) | 0x8000u | pairCls(r));
        });
        for (u32 i = tid; i < nF; i += SBT_NT) {
          const u64 r = singleAt(i);
          const u32 tl = (u32)(r >> 32) - segTileBase, off = (u32)(r >> 8) & (TILE - 1);
          if (tl - tileBeg < tileEnd - tileBeg) keysL[atomicAdd(&L.cur[tl], 1u) - keyBase] = (uint16_t)(off | ((r & 0x80) ? 0x8000u : 0u) | fCls(r));
        }
        __syncthreads();
        tiles(tileEnd, keyBase);
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < SBT_K; i++) {
    if (cS[i] == SBT_SLOT) {
      // (the four cursor atomics in flight together, then the four stores)
      u32 ps[4];
#pragma unroll
      for (int j = 0; j < 4; j++) ps[j] = atomicAdd(&L.cur[(keyAt(kS[i], j) >> TB) - segTileBase], 1u);
#pragma unroll
      for (int j = 0; j < 4; j++) keysL[ps[j]] = (uint16_t)(keyAt(kS[i], j) & (TILE - 1));
    } else if (cS[i]) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((u32)lane * 4 + j < cS[i]) place(keyAt(kS[i], j), 0u, 0u);
    }
    if (cE[i] == SBT_SLOT) {
      u32 ps[4];
#pragma unroll
      for (int j = 0; j < 4; j++) ps[j] = atomicAdd(&L.cur[SBT_TILES + (keyAt(kE[i], j) >> TB) - segTileBase], 1u);
#pragma unroll
      for (int j = 0; j < 4; j++) keysL[ps[j]] = (uint16_t)((keyAt(kE[i], j) & (TILE - 1)) | 0x8000u);
    } else if (cE[i]) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((u32)lane * 4 + j < cE[i]) place(keyAt(kE[i], j), (u32)SBT_TILES, 0x8000u);
    }
  }
  __syncthreads();
  if (GX_EXP_SBT == 3) {
    if (tid < (int)nT && segTileBase + tid < in.nTiles) out.to.tileCount[segTileBase + tid] = keysL[L.startC[tid]] == 0xFFFFu;
    return;
  }
  tiles(nT, 0u);
  }
  if (FRAC && fragTerms) {  // wave-uniform
    fhi = wave_sum(fhi);
    flo = wave_sum(flo);
    if (lane == 0) {
      if (fhi) atomicAdd((u64*)&in.fragAcc[0], (u64)fhi);
      if (flo) atomicAdd((u64*)&in.fragAcc[1], (u64)flo);
    }
  }
  // (-E: this wavefront's share of the pileup over excluded bases, off the closed form of fragLen -- whole bases with unit weights;
  // with fractional ones the closed form is not used, FRAG_SLOW_FRAC)
  if (bedExcl && lane == 0) atomicAdd(&in.fragSum[(seg * SBT_NW + (u32)wv) % FRAG_SLOTS], (u64)(-(bedExcl / GX_UNIT)));
  if (bad && lane == 0) atomicOr(st, bad);
}

template <bool PAIRS, bool BIG, bool FRAC, int TRC = SBT_TR, bool BED = false>
__global__ __launch_bounds__(SBT_NT) void k_sbtile(SbtIn in, SbtOut out, u32* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sbt_raw[];
  SbtLds& L = *reinterpret_cast<SbtLds*>(sbt_raw);
  if constexpr (!BIG)
    sbt_bin<PAIRS, false, FRAC, TRC, BED>(in, out, st, blockIdx.x, L);
  else {
    // (no list: a sample so dense that most bins need rounds -- the host sends every bin here and skips the first launch)
    const u32 nBig = out.bigList ? *out.nBig : in.nSeg;
    for (u32 item = blockIdx.x; item < nBig; item += gridDim.x) {  // (usually none)
      sbt_bin<PAIRS, true, FRAC, TRC, BED>(in, out, st, out.bigList ? out.bigList[item] : item, L);
      __syncthreads();
    }
  }
}

}  // namespace gx
