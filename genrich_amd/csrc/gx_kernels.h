// gx_kernels.h -- hand-written HIP kernels of the Genrich hot path for gfx950 (CDNA4).
//
// Data layout:
//   * every analysed chromosome is cut into tiles of TILE = 2^TB bases; a tile belongs to
//     one chromosome.  The reference's per-base difference array (Diff, Genrich.h:178-181;
//     written by saveInterval, Genrich.c:2575-2583) never exists in HBM: each tile's slice
//     lives in LDS for the lifetime of one workgroup.
//   * an alignment interval of unit weight becomes a 4-byte start key and a 4-byte end key
//       [31:TB] tile   [TB-1:0] offset in tile
//     (sign and weight implied by the stream; 16-bit offsets after the sort), one of fractional
//     weight two 8-byte records  [63:32] tile  [31:8] offset  [7:0] signed weight in 1/120 units;
//     a two-level bucket sort (super-bucket, then tile) groups them by tile.
//   * pileups leave the tile kernel as run-length intervals (end, V120) exactly where the
//     reference breaks them (savePileupExpt, Genrich.c:2239-2273): V120 is the exact pileup
//     in 1/120 units, from which getVal's float is re-materialised when needed.
// No MFMA anywhere: there is no dense contraction in this workload.  All cross-workgroup
// accumulation is integer (atomics / fixed point), so results are run-to-run identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "gx_math.h"
#include "../../include/genrich_amd.h"

namespace gx {

typedef unsigned long long u64;
typedef uint32_t u32;

#ifndef GX_TB
#define GX_TB 12  // 4,096-base tiles measured best of 2^11 / 2^12 / 2^13 (compile-time knob: -DGX_TB=13)
#endif
constexpr int TB = GX_TB;              // tile bits
constexpr int TILE = 1 << TB;          // bases per tile
constexpr u32 NULL_TILE = 0xFFFFFFFFu; // dropped endpoint
constexpr int MAX_BINS = 4096;         // bins of one bucket-sort level

// device status bits (checked by the host at every sync point)
enum : u32 {
  ST_BAD_CHROM = 1u,     // event on an unknown chromosome
  ST_BAD_POS = 2u,       // start >= chromosome length      (ERRPOS, Genrich.c:2531)
  ST_BAD_COUNT = 4u,     // count not in {1,2,3,4,5,6,8,10}  (ERRALNS, :2402)
  ST_NEG_PILE = 8u,      // negative pileup                  (ERRPILE, :1921)
  ST_SAT16 = 16u,        // (unused since the reference's int16 saturation skips are reproduced: k_hot_check)
  ST_LOOKBACK = 32u,     // decoupled look-back spin limit hit (internal)
  ST_HASH_FULL = 64u,    // p-value table overflow (internal)
  ST_NO_FRAGS = 128u,    // fragLen == 0                     (ERREXPT, :2292)
  ST_BAD_DF = 256u,      // more than 200 replicates         (ERRDF, :556)
  // (512 / 1024 / 2048: ST_PT_FULL / ST_SB_FULL / ST_SB_FRAC, gx_sort.h -- internal, the host retries)
  ST_END_PILE = 4096u,   // a chromosome's pileup does not return to 0 behind its last base ("finishes at %f (not 0.0)", :2283-2289)
  ST_BH_LEN = 8192u,     // the lengths behind the p-value table do not add up to the genome length (ERRISSUE, :377-382)
  ST_Q_LOOSE = 16384u,   // (internal) -q on the loose slots: q is no threshold on the pileup -- the host takes the tight table
};

// ---- "risky" p-values (gx_math.h: round_checked) ------------------------------------------------
// Every kernel that rounds a p-value to float appends the few results that lie next to a float
// rounding boundary to this list; at the next point where the host synchronises anyway it
// re-evaluates them with the host's libm (the same __host__ __device__ routines) and k_risk_apply
// writes the values back.  What a record refers to depends on its kind.
enum : u32 {
  RK_LUT = 1,    // a = V: entry of the no-control table p(V)
  RK_DEEP = 2,   // a = V: a pileup beyond that table (k_pval_deep)
  RK_TAB2D = 3,  // a = index into the PT_N x PT_N table of whole (treatment, control) pileups
  RK_PAIR = 4,   // a = interval, b / c = exact treatment / control pileups (k_pack_pairs_full)
  RK_FISHER = 5, // a = tile, b = interval within the tile, c = df, x = sum of -log10 p (k_mergeN)
  RK_SELF = 6,   // a = index, b = function (gx_selftest)
};
struct RiskRec { u32 kind, a, b, c; double x; float pnew; u32 pad; };
// (about 1.2e-4 of the directly evaluated p-values are risky: 2^20 records carry a run of ~10^10 evaluations -- 32 MB on
// the device and as many pinned on the host; round 2's 2^14 made a control / Fisher run beyond ~10^8 evaluations fail)
constexpr u32 RISK_CAP = 1u << 20, RISK_PREFIX = 64;  // records kept / records the host reads with the count
struct RiskBuf { u32 count; u32 pad[7]; RiskRec rec[RISK_CAP]; };
static_assert(sizeof(RiskRec) == 32 && sizeof(RiskBuf) == 32 + 32 * RISK_CAP, "layout shared with the host");

__device__ __forceinline__ void risk_add(RiskBuf* rb, u32 kind, u32 a, u32 b, u32 c, double x) {
  const u32 j = atomicAdd(&rb->count, 1u);  // (beyond RISK_CAP only the count grows: the host reports it)
  if (j < RISK_CAP) rb->rec[j] = RiskRec{kind, a, b, c, x, 0.0f, 0u};
}

// values of pileups beyond the no-control table that the host re-evaluated (RK_DEEP)
constexpr u32 DEEP_TAB = 64;
struct DeepTab { u32 n; u32 pad; int v[DEEP_TAB]; float p[DEEP_TAB]; };

struct DChrom {
  u32 len;
  u32 tileBase;  // first tile of this chromosome (NULL_TILE when it has none)
  u32 nTiles;
  u32 flags;     // bit0 skip (-e), bit1 save (this replicate), bit2 owned by this rank
};
constexpr u32 CH_SKIP = 1, CH_SAVE = 2, CH_OWNED = 4;

__device__ __forceinline__ bool chrom_active(const DChrom& c) {
  return (c.flags & (CH_SKIP | CH_SAVE | CH_OWNED)) == (CH_SAVE | CH_OWNED) && c.tileBase != NULL_TILE;
}

// ---- small block-level helpers (wave = 64 lanes) ------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 64-lane inclusive add-scan in 6 DPP adds (no LDS crossbar traffic): rows of 16 lanes are scanned
// Kogge-Stone fashion with row_shr 1/2/4/8 (lanes without a source read 0), then row_bcast:15 /
// row_bcast:31 carry the row totals across the wave (gfx9 DPP controls).  Each step is x += dpp(x), the
// form the compiler folds into one v_add_u32_dpp.
__device__ __forceinline__ int dpp_scan_add(int v) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return x;
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
  if constexpr (sizeof(T) == 4) {
    return (T)dpp_scan_add((int)v);
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      T o = __shfl_up(v, d, 64);
      if (lane_id() >= d) v += o;
    }
    return v;
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// exclusive block scan; `scratch` holds one T per wave (+1); returns exclusive prefix and the
// block total.  All threads must call it.
template <typename T, int NT>
__device__ __forceinline__ T block_excl_scan(T v, T* scratch, T* total) {
  constexpr int NW = NT / 64;
  T inc = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    T x = threadIdx.x < NW ? scratch[threadIdx.x] : T(0);
    T xi = wave_incl_scan(x);
    if (threadIdx.x < NW) scratch[threadIdx.x] = xi - x;
    if (threadIdx.x == NW - 1) scratch[NW] = xi;
  }
  __syncthreads();
  T res = inc - v + scratch[w];
  *total = scratch[NW];
  __syncthreads();
  return res;
}

// ---- decoupled look-back (single-pass chained scan across workgroups) ------------------------
// lb[i] is one 8-byte granule {flag:2, value:62}: 0 = nothing yet, AGG = this block's own
// aggregate, INC = inclusive prefix through this block.  The value and its flag travel in ONE
// relaxed agent-scope 8-byte store / load, so no fence is needed (the data is the flag).
// Forward progress: every launch that uses this is a PERSISTENT kernel whose workgroups are all
// co-resident (grid <= occupancy x CUs, computed by the host) and take work items round-robin
// (item = blockIdx.x + round * gridDim.x), so an item only ever waits on items that a resident
// workgroup is processing or has finished -- without a per-item ticket atomic (one contended
// word hands out only ~90 tickets/us on MI355X, i.e. > 2 ms for hg38's 188,507 tiles).  Every
// spin is bounded (ST_LOOKBACK).  Call from the first wave (64 lanes) only; returns the
// exclusive prefix of `aggregate` over items [0, id) in every lane.  lb zeroed before launch.
constexpr u64 LB_AGG = 1ull << 62, LB_INC = 2ull << 62, LB_MASK = (1ull << 62) - 1;
constexpr u32 LB_SPIN_LIMIT = 2000000;  // ~ seconds; a timed-out launch is reported, never silent

__device__ __forceinline__ u64 lookback_excl(u64* lb, u32 id, u64 aggregate, u32* st) {
  u64 excl = 0;
  if (id > 0) {
    if (lane_id() == 0)
      __hip_atomic_store(&lb[id], LB_AGG | (aggregate & LB_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = (int)id - 1;
    u32 spins = 0;
    bool done = false;
    while (!done) {
      int idx = look - lane_id();  // lane 0 = nearest predecessor
      u64 v = idx >= 0 ? __hip_atomic_load(&lb[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : LB_INC;   // virtual block -1: inclusive prefix 0
      u64 flag = v >> 62;
      u64 invalidMask = __ballot(flag == 0);
      u64 incMask = __ballot(flag == 2);
      int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
      int firstInc = incMask ? __builtin_ctzll(incMask) : 64;
      if (firstInc < firstInvalid) {
        excl += wave_sum(lane_id() <= firstInc ? (v & LB_MASK) : 0ull);
        done = true;
      } else if (firstInvalid > 0) {
        excl += wave_sum(lane_id() < firstInvalid ? (v & LB_MASK) : 0ull);
        look -= firstInvalid;
      } else {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        // give up when this wave has waited too long, or as soon as any other wave has
        // (so one failure cannot cascade into a launch that spins for minutes)
        bool abort_ = spins > LB_SPIN_LIMIT;
        if (!abort_ && (spins & 1023u) == 0)
          abort_ = (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_LOOKBACK) != 0;
        if (abort_) {
          if (lane_id() == 0) atomicOr(st, ST_LOOKBACK);
          done = true;
        }
      }
    }
  }
  excl &= LB_MASK;  // packed multi-field values rely on modular arithmetic inside the 62 bits
  if (lane_id() == 0)
    __hip_atomic_store(&lb[id], LB_INC | ((excl + aggregate) & LB_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// lazily re-seeking cursor: which chromosome owns interval i of an array laid out by
// chromIvOff[nChrom+1]?  A binary search runs only when i leaves the current range.
struct ChromCursor {
  u32 c = 0, lo = 0, hi = 0;
  __device__ __forceinline__ void seek(const u32* __restrict__ off, u32 nChrom, u32 i) {
    if (i >= lo && i < hi) return;
    u32 a = 0, b = nChrom;  // last c with off[c] <= i  (empty ranges share an offset: take the last)
    while (b - a > 1) {
      u32 mid = (a + b) >> 1;
      if (off[mid] <= i) a = mid; else b = mid;
    }
    c = a;
    lo = off[a];
    hi = off[a + 1];
  }
};

// ---- 1. events -> endpoint records (the kernels are in gx_sort.h) ---------------------------
// Replaces the accumulate step of saveInterval (Genrich.c:2546-2583): instead of a
// read-modify-write on diff[start] / diff[end], emit (+w at start) and (-w at end).
// An end at the chromosome length never influences a base < len and is dropped.
//
// Three record streams feed the bucket sort:
//   S, E : 4-byte keys  [31:TB] tile  [TB-1:0] offset  for the starts / ends of unit-weight
//          (count == 1) intervals -- sign and weight are implied by the stream, which halves
//          the sort traffic of the common case.
//   F    : 8-byte records  [63:32] tile  [31:8] offset  [7:0] signed weight (1/120 units) for
//          multimapped (fractional) intervals; also carries everything when the genome has too
//          many tiles for a 4-byte key.
// weight (1/120 units) of starts, or of ends, below which a base cannot reach the reference's int16 limits
constexpr u32 HOT16 = 32766u * GX_UNIT;
constexpr u32 NULL32 = 0xFFFFFFFFu;
constexpr u32 MAX_TILES32 = (1u << (32 - TB)) - 1;  // tile ids representable in a 4-byte key

template <typename R> struct RecT;
template <> struct RecT<u64> {
  static __device__ __forceinline__ u32 tile(u64 r) { return (u32)(r >> 32); }
};
template <> struct RecT<u32> {
  static __device__ __forceinline__ u32 tile(u32 r) { return r == NULL32 ? NULL_TILE : r >> TB; }
};

__device__ __forceinline__ u64 make_rec64(u32 tile, u32 off, int w) {
  return ((u64)tile << 32) | ((u64)off << 8) | (u64)(uint8_t)(int8_t)w;
}

constexpr int FRAG_SLOTS = 64;  // partial sums of the closed form of fragLen (see k_frag)

// Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2: a 128-byte line completed
// by short runs from several XCDs costs ~1.5x the time of one completed within a single L2
// (tools/bw_probe3.hip: 232 vs 154 us for the level-1 scatter's shape).  Hence one output list per
// (blockIdx % 8, super-bucket) in level 1 of the bucket sort (if the dispatch order differs, only speed
// is affected), and the XCD-local tile order of the tile kernels.
constexpr int NXCD = 8;
// logical index of a workgroup such that consecutive logical indices sit on one XCD: the grid's
// first NXCD * (G / NXCD) workgroups are regrouped, a remainder keeps its index
__device__ __forceinline__ u32 xcd_local_block(u32 b, u32 G) {
  const u32 per = G / NXCD, full = per * NXCD;
  return b < full ? (b % NXCD) * per + b / NXCD : b;
}

// per-tile record counts of the three streams -> offsets; per-tile weight -> genome-wide
// prefix tilePrefW (the carry-in of a tile is tilePrefW[t] - tilePrefW[first tile of its
// chromosome]); weight of a tile = 120 (#S - #E) + sum of its F weights.
// Persistent multi-workgroup chained scan (one look-back per stream), 2048 tiles per item.
constexpr int STL_NT = 256;
constexpr int STL_ITEMS = 8;
constexpr int STL_CHUNK = STL_NT * STL_ITEMS;

struct TileTabs {
  const u32* cnt[3];   // S, E, F record counts per tile (F may be nullptr)
  const int* wsumF;    // sum of F weights per tile (nullptr without F)
  u32* off[3];         // [nTiles+1]
  int* prefW;          // [nTiles]
};

__global__ __launch_bounds__(STL_NT) void k_scan_tiles(TileTabs T, u32 nTiles, u64* __restrict__ lb0,
                                                       u64* __restrict__ lb1, u64* __restrict__ lb2,
                                                       u32* __restrict__ st) {
  __shared__ u32 scratch[8];
  __shared__ u64 s_base[3];
  const u32 nChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  u64* lbs[3] = {lb0, lb1, lb2};
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 tb = id * STL_CHUNK + threadIdx.x * STL_ITEMS;
    u32 c[3][STL_ITEMS];
    int w[STL_ITEMS];
    u32 cs[3] = {0, 0, 0};
    int ws = 0;
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      u32 t = tb + k;
      bool ok = t < nTiles;
      c[0][k] = ok ? T.cnt[0][t] : 0;
      c[1][k] = ok ? T.cnt[1][t] : 0;
      c[2][k] = ok && T.cnt[2] ? T.cnt[2][t] : 0;
      w[k] = 120 * ((int)c[0][k] - (int)c[1][k]) + (ok && T.wsumF ? T.wsumF[t] : 0);
      cs[0] += c[0][k];
      cs[1] += c[1][k];
      cs[2] += c[2][k];
      ws += w[k];
    }
    u32 tot[3];
    u32 ex[3];
    int wtot;
    for (int q = 0; q < 3; q++) ex[q] = block_excl_scan<u32, STL_NT>(cs[q], scratch, &tot[q]);
    int wex = block_excl_scan<int, STL_NT>(ws, reinterpret_cast<int*>(scratch), &wtot);
    if (threadIdx.x < 64) {
      // stream 0's granule also carries the weight prefix (mod 2^30) in bits 32..61.  A chunk's
      // weight sum may be negative, but every PREFIX is a pileup value at a tile boundary
      // (>= 0, < 2^30 in 1/120 units), so modular sums reproduce it exactly.
      u64 agg0 = (u64)tot[0] | ((u64)((u32)wtot & 0x3FFFFFFFu) << 32);
      u64 e0 = lookback_excl(lbs[0], id, agg0, st);
      u64 e1 = lookback_excl(lbs[1], id, (u64)tot[1], st);
      u64 e2 = lookback_excl(lbs[2], id, (u64)tot[2], st);
      if (threadIdx.x == 0) {
        s_base[0] = e0;
        s_base[1] = e1;
        s_base[2] = e2;
        if (id == nChunks - 1) {
          T.off[0][nTiles] = (u32)e0 + tot[0];
          T.off[1][nTiles] = (u32)e1 + tot[1];
          T.off[2][nTiles] = (u32)e2 + tot[2];
        }
      }
    }
    __syncthreads();
    ex[0] += (u32)s_base[0];
    ex[1] += (u32)s_base[1];
    ex[2] += (u32)s_base[2];
    wex = (int)(((u32)wex + (u32)(s_base[0] >> 32)) & 0x3FFFFFFFu);
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      u32 t = tb + k;
      if (t < nTiles) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          T.off[q][t] = ex[q];
        }
        T.prefW[t] = wex;
      }
      ex[0] += c[0][k];
      ex[1] += c[1][k];
      ex[2] += c[2][k];
      wex = (int)(((u32)wex + (u32)w[k]) & 0x3FFFFFFFu);
    }
    __syncthreads();
  }
}

// ---- 3. bucket scatter (both levels, either record width) -----------------------------------
// One workgroup takes CHUNK records of one segment, ranks them per bin with LDS atomics,
// reserves a run per (workgroup, bin) with one global atomic, sorts the chunk in LDS and
// writes bin-contiguous runs.  Order inside a bin is arbitrary: every consumer is a
// commutative integer sum.
constexpr int SC_NT = 1024;  // 16 waves per workgroup: the kernel is latency-, not bandwidth-bound
template <typename R> struct ScCfg { static constexpr int ITEMS = 4; };    // 4096 x 8 B = 32 KiB staged
template <> struct ScCfg<u32> { static constexpr int ITEMS = 8; };        // 8192 x 4 B = 32 KiB staged

// level 2 in one kernel: a workgroup owns one whole super-bucket, so the per-tile histogram, its
// scan and the scatter cursors all live in LDS -- no global atomics, no separate histogram pass,
// and the tile counts are plain stores.  Pass 1 streams the super-bucket and counts (records
// per tile, and for F the signed weight); pass 2 streams it again (it was just read: L2 /
// Infinity Cache) in chunks that are ranked and sorted in LDS and written as tile-contiguous runs.
constexpr int B2_NT = 1024;
// What leaves level 2: for the 4-byte key streams only the offset inside the tile (2 bytes; the
// tile is implied by the record's position and the per-tile offsets), for F the record as it is.
template <typename R> struct B2Out { typedef R type; };
template <> struct B2Out<u32> { typedef uint16_t type; };
static_assert(TB <= 16, "tile offsets are stored in 16 bits");
// one-pass path: records of a super-bucket held in registers (FITEMS per thread) and sorted in LDS
template <typename R> struct B2Cfg { static constexpr int FITEMS = 16; };     // 16 K x 8 B staged
template <> struct B2Cfg<u32> { static constexpr int FITEMS = 48; };          // 48 slots of 1024 keys: up to 48 K x 2 B staged
constexpr int B2_LDS_MAX = 160 * 1024;
__host__ __device__ constexpr u32 b2_table_bytes(u32 nBins) { return (4 * nBins + 32) * 4; }
template <typename R>
__host__ __device__ constexpr u32 b2_stage_bytes() {
  // the larger of: one-pass staging (output elements), one chunk of the two-pass path (records)
  return (u32)B2_NT * (u32)(B2Cfg<R>::FITEMS * sizeof(typename B2Out<R>::type) > ScCfg<R>::ITEMS * sizeof(R)
                                ? B2Cfg<R>::FITEMS * sizeof(typename B2Out<R>::type)
                                : ScCfg<R>::ITEMS * sizeof(R));
}
template <typename R>
__host__ __device__ constexpr size_t b2_lds_bytes(u32 nBins) { return (size_t)b2_stage_bytes<R>() + b2_table_bytes(nBins); }

// ---- 4. the tile kernel: LDS difference array -> prefix sum -> run-length pileup -----------
// Replaces savePileupExpt's two per-base passes (Genrich.c:2197-2273; and the per-base walk
// of calcFactor/savePileupCtrl for a control).  One workgroup owns one tile at a time:
//   zero the LDS slice; add every endpoint record of the tile (ds_add, integer);
//   blocked prefix sum (32 bases per thread, DPP wave scans + one cross-wave step);
//   a base j >= 1 with a non-zero difference closes the interval [.., j) whose value is the
//   prefix BEFORE j (:2241-2251); the chromosome's last tile also closes [.., len) (:2268).
// No inter-workgroup dependency: a tile writes its intervals into its own slot of a LOOSE
// array (slot t starts at tileOff[t] + t: a tile cannot close more intervals than it has
// endpoint records, plus the chromosome-closing one) and reports how many it wrote.
// k_scan_iv then turns the counts into tight offsets and the pack kernels (k_pack_pval, or k_pack
// ahead of a control merge) move the slots to their tight place.  (A fused
// decoupled look-back was measured first: with ~512 resident tiles the look-back distance made
// it latency-bound at ~10 us per tile.)
#ifndef GX_TL_EPT
#define GX_TL_EPT 32
#endif
constexpr int TL_EPT = GX_TL_EPT;                 // bases per thread (= one occupancy word): 32, or 64 for one wavefront per 2^12-base tile
static_assert(TL_EPT == 32 || TL_EPT == 64, "the occupancy word is 32 or 64 bits");
constexpr int TL_NT = TILE / TL_EPT;
constexpr int TL_NW = TL_NT / 64;
static_assert(TL_NT % 64 == 0, "whole wavefronts");
#ifndef GX_TL_REG
#define GX_TL_REG (GX_TL_EPT == 64 ? 6 : 4)
#endif
constexpr int TL_REG = GX_TL_REG;                 // touched bases per thread held in registers
constexpr int TL_KPT = (TL_NT >= 128 ? 1 : 128 / TL_NT);  // prefetched keys per thread and stream (128 per tile)
using occ_t = std::conditional<TL_EPT == 64, unsigned long long, u32>::type;
__device__ __forceinline__ int occ_ctz(u32 v) { return __builtin_ctz(v); }
__device__ __forceinline__ int occ_ctz(unsigned long long v) { return __builtin_ctzll(v); }
__device__ __forceinline__ int occ_popc(u32 v) { return __popc(v); }
__device__ __forceinline__ int occ_popc(unsigned long long v) { return __popcll(v); }
constexpr int TL_LDS = TILE + 64 + 2 * (TILE / 32); // ints: slice, scan scratch, occupancy + -E edge bitmaps
// k_tile<.., HALF = true> packs the differences of two neighbouring bases into one LDS word as 16-bit
// counts of unit-weight records (word = 65536 * odd base + even base as one integer sum, so borrows
// between the halves undo themselves on decoding): half the LDS per tile = twice the tiles in
// flight on a CU, which is what bounds this kernel.  Tiles that could overflow a half (>= 32767
// records in a stream) or that hold fractional records are "wide" and go through the 32-bit variant.
constexpr int TL_LDS_HALF = TILE / 2 + 64 + 2 * (TILE / 32);
constexpr u32 TL_HALF_MAX = 32767;                // records per stream at which a tile turns wide
constexpr u32 TM_ACTIVE = 1u, TM_LAST = 2u, TM_WIDE = 4u, TM_HEAVY = 8u;  // TileMeta::flags
// A tile with more records than this is not walked by one wavefront (k_tile_fast spends ~3 passes of n / 64 steps
// on a tile: a pile-up of 16,000 fragments in one place held the whole kernel for a millisecond) but by a workgroup
// with a counter per base (k_tile_heavy).
constexpr u32 HEAVY_MIN = 4096;
constexpr int FRAG_FAST_MAXV = ((1 << 24) / (2 * TILE)) * GX_UNIT;  // len < 2 TILE and V below this: len * val < 2^24
constexpr int V_MARK = (int)0x80000000;           // "pileup" of an interval inside an excluded (-E) region

// -E support (Genrich.c:2185-2263): bed edges are breakpoints whatever the difference array
// holds, breakpoints inside an excluded region are suppressed, and intervals there carry
// V_MARK (treatment value 0.0f, control SKIP).  Edges of a tile come as a CSR list built on
// the host; tileSave0 is the `save` state in effect at the tile's first base.
struct BedIn {
  const u32* bedTileOff;  // [nTiles+1]
  const u32* bedEdge;     // tile-local offsets
  const uint8_t* tileSave0;
};

// ---- the peak sweep on the loose slots (gx_find_peaks, one replicate, no control, -p) ------------------------
// Without a control p is a function of the exact pileup V (the table p(V), k_pval_lut), and with the closed
// form of fragLen lambda -- hence the table -- is known BEFORE the tile stage.  The tile kernels then write the
// sweep's significance bits themselves, in loose-slot index space, and fill a tile's unused slots with
// zero-length intervals, so that the sweep walks the loose slots as they are (no k_pack_pval round trip).
// One block of words per sample (zero arena) says whether that is valid.
constexpr u32 PV_LUT = 1u << 18;  // entries of the table p(V)
constexpr u32 PV_WHOLE = (PV_LUT + GX_UNIT - 1) / GX_UNIT;  // whole pileups in the table (2,185): p(120 c), compact copy
struct LooseCtl {
  u32 bad;        // something forbids the loose sweep (a pileup beyond the table, a table that is not monotone)
  u32 earlyBits;  // lambda as the tile stage knew it (float bits)
  u32 enabled;    // lambda was known before the tile stage
  u32 ok;         // the verdict for the host (k_frag_select): enabled, nothing bad, lambda unchanged
  u32 ready;      // k_bins_lut: the three words above are written (its table blocks wait for this one)
  u32 pad[3];
  // one slot per workgroup of k_pval_lut (plain stores: two contended words took 8,192 same-address atomics, 45 us)
  u32 sigInv[PV_LUT / 256]; // max of (PV_LUT - V) over the workgroup's significant table entries V (0: none)
  u32 nonP1[PV_LUT / 256];  // max of (V + 1) over the others
};
// pileups (1/120 units) from which an interval is significant; INT_MAX: the tile kernels write no bits.
// Call with the whole workgroup (`red`: two words of LDS; contains a barrier).
// (late: lambda came with the sample's end -- k_loose_late reads the slots the table's second launch left, `enabled` or not)
__device__ __forceinline__ int loose_vsig(LooseCtl* __restrict__ c, bool reporter, u32* __restrict__ red, bool late = false) {
  if (!c || (!late && !c->enabled)) return 0x7FFFFFFF;  // (block-uniform)
  u32 a = 0, b = 0;
  for (u32 i = threadIdx.x; i < PV_LUT / 256; i += blockDim.x) {
    a = max(a, c->sigInv[i]);
    b = max(b, c->nonP1[i]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    a = max(a, (u32)__shfl_xor((int)a, d, 64));
    b = max(b, (u32)__shfl_xor((int)b, d, 64));
  }
  if (blockDim.x > 64) {
    if (threadIdx.x < 2) red[threadIdx.x] = 0;
    __syncthreads();
    if (lane_id() == 0) {
      atomicMax(&red[0], a);
      atomicMax(&red[1], b);
    }
    __syncthreads();
    a = red[0];
    b = red[1];
  }
  const u32 minSig = PV_LUT - a;
  if (minSig < b) {  // p(V) > thr is not a threshold on V: leave it to the general path
    if (reporter) atomicOr(&c->bad, 2u);
    return 0x7FFFFFFF;
  }
  return (int)minSig;
}

// The significance bits of one step of a tile kernel (up to 64 intervals written to consecutive loose slots from
// `pos` on; lane's interval is the rank-th of them and significant iff sg): gathered into one 64-bit word by a
// wave-wide sum of disjoint bits and ORed into the mask by ONE lane.  (One atomic per significant interval was the
// first version: peaks are dense in breakpoints, a third of all intervals of config 2 are significant, and 64 lanes
// hammering one 8-byte word serialise in the L2.)  Call with all lanes active.
// `written`: the lanes that wrote an interval (rank = a lane's number among them).  When they are the lanes 0 .. k-1 --
// nearly always: a touched base whose starts and ends cancel is rare -- a lane's rank is its number and the word is
// the ballot itself (two scalar instructions instead of two DPP scans).
__device__ __forceinline__ void sig_flush_m(u64* __restrict__ sigMask, u32 pos, u64 sgm, u32 rank, u64 written);
__device__ __forceinline__ void sig_flush(u64* __restrict__ sigMask, u32 pos, bool sg, u32 rank, u64 written) {
  sig_flush_m(sigMask, pos, __ballot(sg), rank, written);
}
// (sgm: the lanes whose interval is significant, as a mask)
__device__ __forceinline__ void sig_flush_m(u64* __restrict__ sigMask, u32 pos, u64 sgm, u32 rank, u64 written) {
  if (!sgm) return;  // wave-uniform
  u64 m;
  if (((written + 1ull) & written) == 0ull) {  // wave-uniform
    m = sgm;
  } else {
    const bool sg = __builtin_amdgcn_inverse_ballot_w64(sgm);
    const int lo = sg && rank < 32u ? (int)(1u << rank) : 0, hi = sg && rank >= 32u ? (int)(1u << (rank - 32u)) : 0;
    const u32 mlo = (u32)__builtin_amdgcn_readlane(dpp_scan_add(lo), 63), mhi = (u32)__builtin_amdgcn_readlane(dpp_scan_add(hi), 63);
    m = (u64)mlo | ((u64)mhi << 32);
  }
  if (lane_id() == 0) {
    const u32 w = pos >> 6, sh = pos & 63;
    atomicOr((unsigned long long*)&sigMask[w], (unsigned long long)(m << sh));
    if (sh && (m >> (64 - sh))) atomicOr((unsigned long long*)&sigMask[w + 1], (unsigned long long)(m >> (64 - sh)));
  }
}

struct TileOut {
  u32* looseEnd;    // interval end (chromosome coordinate), loose slots
  int* looseV;      // pileup in 1/120 units
  u32* tileCount;   // [nTiles] intervals written by each tile
  u32* tileLastEnd; // [nTiles] end of the tile's last interval (valid when tileCount > 0)
  u32* tileDeep;    // [nTiles] pre-zeroed; set when a pileup of the tile reaches FRAG_FAST_MAXV (see k_frag)
  u64* sigMask;     // bit i: loose slot i holds a significant interval (pre-zeroed; see LooseCtl)
  LooseCtl* ctl;
};

// savePileupExpt 2246/2271, calcFactor 2018/2038: `fragLen += (j - start) * val` is a float
// product added into a double.  Every product is a multiple of 2^-27 (val >= 1/10 when
// non-zero), so the sum is accumulated exactly in two int64 (integer part, fraction * 2^27):
// deterministic for any launch geometry or rank count, and equal to the reference's double
// sum whenever that sum is exact (always, for unit weights).
__device__ __forceinline__ void frag_term(u32 len, int v, long long& hi, long long& lo) {
  if (v != 0 && v != V_MARK) {
    bool ng;
    const float term = (float)len * getval(v, &ng);
    const float fl = floorf(term);
    hi += (long long)fl;
    lo += (long long)((term - fl) * 134217728.0f);
  }
}

// everything k_tile needs to know about a tile in one 48-byte record, so the persistent loop can
// prefetch the next tile's descriptor without a dependent-load chain (tileChrom -> chroms -> prefW)
struct __attribute__((aligned(16))) TileMeta {
  u32 sb, eb, fb, nS;     // record ranges of the three streams
  u32 nE, nF;
  int carry;              // pileup (1/120 units) at the tile's first base
  u32 ci;                 // chromosome
  u32 pos0, len;          // first base of the tile, chromosome length
  u32 flags;              // bit0 active, bit1 last tile of the chromosome
  u32 slot;               // first loose output slot
};

struct FragFix;
struct TileIn {
  const uint16_t* S;  const uint16_t* E;  // start / end offsets inside the tile, bucketed by tile
  const u64* F;                           // fractional-weight records
  const TileMeta* meta;
  // k_tile_fast only: with the general fragLen path on (FragFix::slow: fractional weights), the tile kernel adds the
  // exact term of every interval but a tile's first -- it knows where each one starts -- to `fragAcc` itself
  // (frag_term; k_scan_iv adds the tiles' first intervals, k_frag_walk what k_tile_heavy emitted): round 2 walked
  // all loose slots once more for that (k_frag_walk over every tile: 0.87 ms at config 4)
  const FragFix* ff = nullptr;
  long long* fragAcc = nullptr;
};

__global__ __launch_bounds__(256) void k_tile_meta(const u32* __restrict__ offS, const u32* __restrict__ offE,
                                                   const u32* __restrict__ offF, const int* __restrict__ prefW,
                                                   const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                                   const u32* __restrict__ bedTileOff, u32 nTiles,
                                                   TileMeta* __restrict__ meta, u32* __restrict__ wideList,
                                                   u32* __restrict__ nWide, u32* __restrict__ tileSlot,
                                                   u32* __restrict__ heavyList /* or null; its length: nWide[2] */) {
  // (the grid covers the tiles exactly once: every wavefront reaches the ballot below together)
  for (u32 t0 = blockIdx.x * 256; t0 < nTiles; t0 += gridDim.x * 256) {
    const u32 t = t0 + threadIdx.x;
    bool wide = false;
    if (t < nTiles) {
    const u32 ci = tileChrom[t];
    const DChrom c = chroms[ci];
    TileMeta m;
    m.sb = offS[t]; m.nS = offS[t + 1] - m.sb;
    m.eb = offE[t]; m.nE = offE[t + 1] - m.eb;
    m.fb = offF[t]; m.nF = offF[t + 1] - m.fb;
    m.carry = prefW[t] - prefW[c.tileBase];
    m.ci = ci;
    const u32 tl = t - c.tileBase;
    m.pos0 = tl << TB;
    m.len = c.len;
    wide = m.nF != 0 || m.nS >= TL_HALF_MAX || m.nE >= TL_HALF_MAX;
    const bool heavy = heavyList && (u64)m.nS + m.nE + m.nF > HEAVY_MIN;
    m.flags = (chrom_active(c) ? TM_ACTIVE : 0u) | (tl + 1 == c.nTiles ? TM_LAST : 0u) | (wide ? TM_WIDE : 0u) |
              (heavy ? TM_HEAVY : 0u);
    if (heavy) heavyList[atomicAdd(nWide + 2, 1u)] = t;  // rare
    m.slot = m.sb + m.eb + m.fb + t + (bedTileOff ? bedTileOff[t] : 0u);  // <= records + edges + 1 intervals per tile
    meta[t] = m;
    tileSlot[t] = m.slot;
    if (t + 1 == nTiles) tileSlot[nTiles] = m.slot + m.nS + m.nE + m.nF + 1 + (bedTileOff ? bedTileOff[nTiles] - bedTileOff[t] : 0u);
    }
    // the wide tiles, as a list for k_tile<.., false> (one reservation per wavefront, ascending inside it)
    const u64 wm = __ballot(wide);
    if (wm) {
      u32 base = 0;
      if (lane_id() == 0) base = atomicAdd(nWide, (u32)__popcll(wm));
      base = __shfl(base, 0, 64);
      if (wide) wideList[base + __popcll(wm & ((1ull << lane_id()) - 1ull))] = t;
    }
  }
}

template <bool HALF>
__device__ __forceinline__ int tile_diff(const int* delta, int base) {  // difference at a base of the tile, 1/120 units
  if constexpr (!HALF) return delta[base];
  const int w = delta[base >> 1];
  const int lo = (int)(short)w;
  return ((base & 1) ? (w - lo) >> 16 : lo) * GX_UNIT;
}

// HALF: all tiles except the wide ones.  !HALF: the tiles of wideList (*nWide of them).
template <bool BED, bool HALF>
__global__ __launch_bounds__(TL_NT, 2) void k_tile(TileIn in, u32 nTiles, const u32* __restrict__ wideList,
                                                   const u32* __restrict__ nWide, BedIn bed, TileOut out,
                                                   u32* __restrict__ st) {
  // LDS: the tile's slice of the difference array (one int per base) plus an occupancy bitmap
  // (one bit per base).  Only bases that received a record are ever read back or cleared, so a
  // tile costs O(records) LDS traffic instead of O(TILE): at config 2 ~130 of the 4,096 bases
  // of a tile are touched.  The slice is all-zero whenever a tile starts (each tile clears what it
  // touched).
  extern __shared__ __attribute__((aligned(16))) int lds[];
  constexpr int SLICE = HALF ? TILE / 2 : TILE;
  int* delta = lds;                            // SLICE ints
  int* scr = lds + SLICE;                      // [0..8] sums, [16..24] counts, [32..40] edge counts
  occ_t* occ = reinterpret_cast<occ_t*>(scr + 64);  // one word per thread: bit k = base k of the thread's TL_EPT
  occ_t* eb = occ + TL_NT;                          // -E edge bitmap, same shape
  const int wv = threadIdx.x >> 6;
  u32 bad = 0;
  for (int i = threadIdx.x * 4; i < SLICE; i += TL_NT * 4)
    *reinterpret_cast<int4*>(delta + i) = make_int4(0, 0, 0, 0);
  occ[threadIdx.x] = 0;
  eb[threadIdx.x] = 0;
  __syncthreads();
  // Software pipeline over this workgroup's tiles.  On this hardware loads and stores share one
  // in-order counter (vmcnt), and a wait that follows a data-dependent number of stores can only be
  // "everything": a prefetch consumed after a tile's stores makes every tile wait for its own stores
  // to be acknowledged (measured: the kernel ran at the speed of that round trip).  So all
  // prefetches are issued right after a tile's stores (`issue`) and collected right before the
  // next tile's stores (`collect`): whatever is waited for is at least most of a tile old.
  //   top of tile j:  M(j), K(j), M(j+1) in registers;  K(j+1), M(j+2) [, list(j+4)] in flight
  //   collect (j):    K(j+1), M(j+2) arrived
  //   issue (j):      K(j+2) (needs M(j+2)), M(j+3) [, list(j+5)]
  // (M = descriptor, K = the first TL_NT start / end keys, list = entry of wideList.)
  const u32 G = gridDim.x;
  // neighbouring tiles write neighbouring loose slots: give them to workgroups of the same XCD
  // (workgroups are dealt to the XCDs round-robin), so that the cache lines two tiles share are
  // completed inside one L2
  const u32 lb = xcd_local_block(blockIdx.x, G);
  // When most tiles are wide (multimapped reads everywhere) the 32-bit kernel takes all of them and
  // this one none.
  const u32 nW = *nWide;
  const bool allWide = nW > nTiles / 2;
  if (HALF && allWide) return;
  const u32 nItems = HALF || allWide ? nTiles : nW;
  struct Raw { uint4 a, b, c; };
  auto tileAt = [&](u32 i) -> u32 { return i < nItems ? (HALF || allWide ? i : wideList[i]) : 0u; };
  // (unconditional: past the end tileAt gives tile 0, whose descriptor is loaded and then ignored.  A
  // conditional load would be followed by register copies, i.e. by a wait for it)
  auto loadMeta = [&](u32 tile) -> Raw {
    const uint4* q = reinterpret_cast<const uint4*>(in.meta + tile);
    return Raw{q[0], q[1], q[2]};
  };
  auto uni = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_readfirstlane((int)v); };
  auto cook = [&](const Raw& r, u32 i) -> TileMeta {  // (the whole workgroup loaded the same descriptor)
    TileMeta m;
    m.sb = uni(r.a.x); m.eb = uni(r.a.y); m.fb = uni(r.a.z); m.nS = uni(r.a.w);
    m.nE = uni(r.b.x); m.nF = uni(r.b.y); m.carry = (int)uni(r.b.z); m.ci = uni(r.b.w);
    m.pos0 = uni(r.c.x); m.len = uni(r.c.y); m.flags = uni(r.c.z); m.slot = uni(r.c.w);
    if (i >= nItems) { m.nS = 0; m.nE = 0; m.nF = 0; m.flags = 0; }
    return m;
  };
  // (a wide tile's keys are not for the HALF kernel: it skips the tile)
  struct Keys { u32 v[TL_KPT]; };
  auto keyOf = [&](const uint16_t* K, u32 kb, u32 nk, u32 flags) -> Keys {
    Keys r;
#pragma unroll
    for (int q = 0; q < TL_KPT; q++) {
      const u32 ix = threadIdx.x + q * TL_NT;
      r.v[q] = ix < nk && !(HALF && (flags & TM_WIDE)) ? K[kb + ix] : 0u;
    }
    return r;
  };
  u32 tC = tileAt(lb), tN = tileAt(lb + G), tF = tileAt(lb + 2 * G), tQ = tileAt(lb + 3 * G), tL = tileAt(lb + 4 * G);
  u32 tR = 0;
  TileMeta mC = cook(loadMeta(tC), lb), mN = cook(loadMeta(tN), lb + G), mR{};
  Keys ksC = keyOf(in.S, mC.sb, mC.nS, mC.flags), keC = keyOf(in.E, mC.eb, mC.nE, mC.flags), ksN{}, keN{};
  // (arrived before the loop: a register that is pending on the way in gets a wait at its use inside the
  // loop, which at run time would wait for whatever is in flight in every iteration)
  asm volatile("" : "+v"(ksC.v[0]), "+v"(keC.v[0]), "+v"(tF), "+v"(tQ) :: "memory");
  if constexpr (TL_KPT > 1) asm volatile("" : "+v"(ksC.v[TL_KPT - 1]), "+v"(keC.v[TL_KPT - 1]) :: "memory");
  Keys ksL = keyOf(in.S, mN.sb, mN.nS, mN.flags), keL = keyOf(in.E, mN.eb, mN.nE, mN.flags);  // in flight
  Raw rF = loadMeta(tF);                                                                      // in flight
  for (u32 i = lb; i < nItems; i += G) {
    const u32 t = tC;
    TileMeta m = mC;
    if (HALF && (m.flags & TM_WIDE)) {
      // not this kernel's tile: it passes as an empty, inactive one (one `collect` and one `issue` site
      // keep the registers with loads in flight free of copies, which would be waits); its count of
      // zero intervals is overwritten by the kernel that owns it
      m.nS = 0; m.nE = 0; m.nF = 0; m.flags = 0;
    }
    const Keys ks0 = ksC, ke0 = keC;
    auto collect = [&]() {
      // (the empty asm is the point where the loads in flight must have arrived, and nothing that
      // touches memory moves across it)
      if constexpr (TL_KPT > 1) asm volatile("" : "+v"(ksL.v[TL_KPT - 1]), "+v"(keL.v[TL_KPT - 1]) :: "memory");
      asm volatile("" : "+v"(ksL.v[0]), "+v"(keL.v[0]), "+v"(rF.a.x), "+v"(rF.a.y), "+v"(rF.a.z), "+v"(rF.a.w), "+v"(rF.b.x),
                        "+v"(rF.b.y), "+v"(rF.b.z), "+v"(rF.b.w), "+v"(rF.c.x), "+v"(rF.c.y), "+v"(rF.c.z), "+v"(rF.c.w),
                        "+v"(tL) :: "memory");
      ksN = ksL;
      keN = keL;
      mR = cook(rF, i + 2 * G);
      tR = tF;
    };
    auto issue = [&]() {
      // tile j+1 becomes the current one -- first, so that the registers of what has arrived are free
      // again before the loads are issued (a load into another register would have to be copied at the
      // loop's back edge: a wait)
      mC = mN; tC = tN; ksC = ksN; keC = keN;
      mN = mR; tN = tR;
      ksL = keyOf(in.S, mN.sb, mN.nS, mN.flags);  // K(j+2)
      keL = keyOf(in.E, mN.eb, mN.nE, mN.flags);
      rF = loadMeta(tQ);                          // M(j+3)
      tF = tQ;
      tQ = tL;
      tL = tileAt(i + 5 * G);
    };
    const bool active = m.flags & TM_ACTIVE;
    const u32 pos0 = m.pos0;
    const bool lastTile = (m.flags & TM_LAST) != 0;
    const u32 sb = m.sb, se = m.sb + m.nS, eb0 = m.eb, ee = m.eb + m.nE, fb = m.fb, fe = m.fb + m.nF;
    const int carry = m.carry;
    // (occ / eb words are zero here: each thread clears its word as soon as it has read it)
    // accumulate this tile's endpoints: +1 per start, -1 per end (unit weight = 120), then the
    // fractional records with their own signed weight
    auto add = [&](u32 off, int sign) {
      if constexpr (HALF)
        atomicAdd(&delta[off >> 1], (off & 1) ? sign * 65536 : sign);
      else
        atomicAdd(&delta[off], sign * GX_UNIT);
      atomicOr(&occ[off / TL_EPT], (occ_t)1 << (off % TL_EPT));
    };
#pragma unroll
    for (int q = 0; q < TL_KPT; q++) {
      if (threadIdx.x + q * TL_NT < m.nS) add(ks0.v[q], 1);
      if (threadIdx.x + q * TL_NT < m.nE) add(ke0.v[q], -1);
    }
    for (u32 i = sb + TL_KPT * TL_NT + threadIdx.x; i < se; i += TL_NT) add(in.S[i], 1);
    for (u32 i = eb0 + TL_KPT * TL_NT + threadIdx.x; i < ee; i += TL_NT) add(in.E[i], -1);
    if constexpr (!HALF)
      for (u32 i = fb + threadIdx.x; i < fe; i += TL_NT) {
        u64 r = in.F[i];
        u32 off = (u32)(r >> 8) & (TILE - 1);
        atomicAdd(&delta[off], (int)(int8_t)(r & 0xFF));
        atomicOr(&occ[off / TL_EPT], (occ_t)1 << (off % TL_EPT));
      }
    if (BED)
      for (u32 i = bed.bedTileOff[t] + threadIdx.x; i < bed.bedTileOff[t + 1]; i += TL_NT) {
        u32 off = bed.bedEdge[i];
        atomicOr(&eb[off / TL_EPT], (occ_t)1 << (off % TL_EPT));
      }
    __syncthreads();
    // thread i owns bases [32 i, 32 i + 32): pass 1 over its touched bases (and -E edges)
    const occ_t ow = occ[threadIdx.x];
    const occ_t ew = BED ? eb[threadIdx.x] : (occ_t)0;
    occ[threadIdx.x] = 0;  // own word, next touched after the barrier that ends this tile
    if (BED) eb[threadIdx.x] = 0;
    bool save = true;
    if (BED) {  // `save` state at this thread's first base = tile state ^ parity(edges before it)
      const int pe = occ_popc(ew);
      const int incE = dpp_scan_add(pe);
      if (lane_id() == 63) scr[32 + wv] = incE;
      __syncthreads();
      int preE = incE - pe;
#pragma unroll
      for (int w = 0; w < TL_NW; w++)
        if (w < wv) preE += scr[32 + w];
      save = ((bed.tileSave0[t] != 0) ^ ((preE & 1) != 0));
    }
    const bool save0 = save;
    const int lbase = threadIdx.x * TL_EPT;
    int sum = 0;
    u32 cnt = 0;
    // The first TL_REG touched bases of the thread are fetched from LDS in one batch and kept in
    // registers for both passes (a dependent LDS round trip per base and pass is what this kernel
    // would otherwise wait for); their slots are cleared at once.  Further bases -- rare -- loop.
    int kk[TL_REG], dk[TL_REG];
    occ_t mrest = ow | ew;
#pragma unroll
    for (int j = 0; j < TL_REG; j++) {
      kk[j] = mrest ? occ_ctz(mrest) : -1;
      dk[j] = mrest ? tile_diff<HALF>(delta, lbase + kk[j]) : 0;
      mrest &= mrest - 1;
    }
#define GX_P1_STEP(K, D)                                                        \
  {                                                                            \
    const bool edge = BED && ((ew >> (K)) & 1u);                               \
    sum += (D);                                                                \
    cnt += (edge || (save && (D) != 0)) && (pos0 + lbase + (K) != 0); /* 2241 */ \
    if (edge) save = !save;                                       /* 2258-2263 */ \
  }
    if constexpr (!BED) {
      // Without -E edges the register-held steps need no branches: an absent entry carries d = 0,
      // which adds nothing, counts nothing and needs no clearing.  (Base 0 of a chromosome is the
      // one position whose difference does not close an interval, 2241.)
#pragma unroll
      for (int j = 0; j < TL_REG; j++) {
        const int d = dk[j];
        sum += d;
        cnt += (u32)(d != 0);
        if constexpr (!HALF)
          if (d != 0) delta[lbase + kk[j]] = 0;
      }
      if (pos0 + lbase == 0 && kk[0] == 0 && dk[0] != 0) cnt -= 1;
    } else {
#pragma unroll
      for (int j = 0; j < TL_REG; j++)
        if (kk[j] >= 0) {
          GX_P1_STEP(kk[j], dk[j]);
          if constexpr (!HALF)
            if (dk[j] != 0) delta[lbase + kk[j]] = 0;
        }
    }
    for (occ_t m = mrest; m; m &= m - 1) {
      const int k = occ_ctz(m);
      const int d = tile_diff<HALF>(delta, lbase + k);
      GX_P1_STEP(k, d);
    }
#undef GX_P1_STEP
    if (!active) cnt = 0;
    // fused block scan of (sum, cnt): two DPP wave scans, one cross-wave step
    const int incS = dpp_scan_add(sum);
    const u32 incC = (u32)dpp_scan_add((int)cnt);
    if (lane_id() == 63) { scr[wv] = incS; scr[16 + wv] = (int)incC; }
    __syncthreads();
    int preS = 0;
    u32 preC = 0, totC = 0;
#pragma unroll
    for (int w = 0; w < TL_NW; w++) {  // every thread sums the (few) wave totals it needs
      int ws = scr[w];
      u32 wc = (u32)scr[16 + w];
      if (w < wv) { preS += ws; preC += wc; }
      totC += wc;
    }
    collect();
    // pass 2: emit, and clear what was touched
    int run = carry + preS + (incS - sum);
    const u32 slot = m.slot;
    const u32 o0 = slot + preC + (incC - cnt);
    const u32 totFinal = active ? totC + (lastTile ? 1u : 0u) : 0u;
    u32 o = o0, neg = 0, lastEnd = 0;
    u32 big = threadIdx.x == 0 && carry >= FRAG_FAST_MAXV;
    save = save0;
#define GX_P2_STEP(K, D)                                                \
  {                                                                    \
    const u32 p = pos0 + lbase + (K);                                  \
    const bool edge = BED && ((ew >> (K)) & 1u);                       \
    if (active && (edge || (save && (D) != 0)) && p != 0) {            \
      out.looseEnd[o] = p;                                             \
      out.looseV[o] = save ? run : V_MARK; /* 2244-2248 */             \
      lastEnd = p;                                                     \
      o++;                                                             \
    }                                                                  \
    if (edge) save = !save;                                            \
    run += (D);                                                        \
    neg |= (u32)(run < 0);                                             \
    big |= (u32)(run >= FRAG_FAST_MAXV);                               \
  }
    if constexpr (!BED) {
#pragma unroll
      for (int j = 0; j < TL_REG; j++) {  // (d = 0 for an absent entry: nothing emitted, run unchanged)
        const int d = dk[j];
        const u32 p = pos0 + lbase + kk[j];
        if (active && d != 0 && p != 0) {
          out.looseEnd[o] = p;
          out.looseV[o] = run;
          lastEnd = p;
          o++;
        }
        run += d;
        neg |= (u32)(run < 0);
        big |= (u32)(run >= FRAG_FAST_MAXV);
      }
    } else {
#pragma unroll
      for (int j = 0; j < TL_REG; j++)
        if (kk[j] >= 0) GX_P2_STEP(kk[j], dk[j]);
    }
    for (occ_t m = mrest; m; m &= m - 1) {
      const int k = occ_ctz(m);
      const int d = tile_diff<HALF>(delta, lbase + k);
      GX_P2_STEP(k, d);
      if constexpr (HALF) {  // a word is cleared by the last touched base it holds
        if ((k & 1) || !((m >> (k + 1)) & 1)) delta[(lbase + k) >> 1] = 0;
      } else if (d != 0) {
        delta[lbase + k] = 0;
      }
    }
#undef GX_P2_STEP
    if constexpr (HALF) {  // (only now: a base of the loop above may share its word with a register-held one)
#pragma unroll
      for (int j = 0; j < TL_REG; j++)
        if (dk[j] != 0) delta[(lbase + kk[j]) >> 1] = 0;
    }
    if (active) {  // block-uniform
      if (lastTile && threadIdx.x == TL_NT - 1) {  // closing interval [.., len)
        out.looseEnd[o] = m.len;
        out.looseV[o] = save ? run : V_MARK;
        lastEnd = m.len;
        o++;
      }
      if (o != o0 && o == slot + totFinal) out.tileLastEnd[t] = lastEnd;  // wrote the tile's last interval
      bad |= neg ? ST_NEG_PILE : 0;
      if (big) atomicOr(&out.tileDeep[t], 1u);  // rare
    }
    if (threadIdx.x == 0) out.tileCount[t] = totFinal;
    issue();
    __syncthreads();  // scr, occ and the slice are reused by the next tile
  }
  if (bad) atomicOr(st, bad);
}

// Can any base reach the reference's int16 limits (Genrich.c:2558-2573: an alignment is dropped when
// diff[start] holds INT16_MAX or diff[end] INT16_MIN)?  That takes >= 32,767 unit weights starting,
// or ending, on one base, so only a tile that is wide by count can hold such a base.  Starts and ends
// are summed apart: the running value in the reference's input order can hit a limit while the net
// difference k_tile sees stays small.  The answer is one flag; which alignments are dropped depends
// on their order and is decided on the host (gx_saturate.h), which then rebuilds the sample.
__global__ __launch_bounds__(256) void k_hot_check(TileIn in, const u32* __restrict__ wideList,
                                                   const u32* __restrict__ nWide, u32* __restrict__ hot) {
  __shared__ u32 ws[TILE], we[TILE];
  const u32 nW = *nWide;
  for (u32 i = blockIdx.x; i < nW; i += gridDim.x) {
    const TileMeta m = in.meta[wideList[i]];
    if (!(m.flags & TM_ACTIVE)) continue;  // block-uniform
    if ((u64)m.nS + m.nF < HOT16 / GX_UNIT && (u64)m.nE + m.nF < HOT16 / GX_UNIT) continue;
    for (int j = threadIdx.x; j < TILE; j += 256) { ws[j] = 0; we[j] = 0; }
    __syncthreads();
    for (u32 j = threadIdx.x; j < m.nS; j += 256) atomicAdd(&ws[in.S[m.sb + j]], (u32)GX_UNIT);
    for (u32 j = threadIdx.x; j < m.nE; j += 256) atomicAdd(&we[in.E[m.eb + j]], (u32)GX_UNIT);
    for (u32 j = threadIdx.x; j < m.nF; j += 256) {
      const u64 r = in.F[m.fb + j];
      const int w = (int)(int8_t)(r & 0xFF);
      const u32 off = (u32)(r >> 8) & (TILE - 1);
      if (w > 0) atomicAdd(&ws[off], (u32)w); else atomicAdd(&we[off], (u32)(-w));
    }
    __syncthreads();
    bool h = false;
    for (int j = threadIdx.x; j < TILE; j += 256) h |= ws[j] >= HOT16 || we[j] >= HOT16;
    if (h) atomicOr(hot, 1u);
    __syncthreads();
  }
}

// The difference array of ONE window of a chromosome from the events pushed so far (gx_window_net): net[i] = weight of
// the events that start at pos0 + i minus the weight of those that end there, as saveInterval's `diff` holds them
// (Genrich.c:2576-2583; ends clamped to the chromosome's length as 2536-2544 does, the entry at `len` included).  For a
// caller that reproduces the int16 checks read by read: it asks once per window that comes near the limits.
__global__ __launch_bounds__(256) void k_window_net(const uint4* __restrict__ ev, size_t n, u32 chrom, u32 clen, u32 pos0, u32 nPos,
                                                    unsigned long long* __restrict__ net) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 e = ev[i];
    const u32 cnt = e.w;
    if (e.x != chrom || e.y >= clen || cnt > 10u || !((0x57Eu >> cnt) & 1u)) continue;
    const long long w = (long long)(GX_UNIT / (int)cnt);
    const u32 end = e.z > clen ? clen : e.z;
    if (e.y - pos0 < nPos) atomicAdd(&net[e.y - pos0], (unsigned long long)w);
    if (end - pos0 < nPos) atomicAdd(&net[end - pos0], (unsigned long long)(-w));
  }
}

// ---- heavy tiles: one WORKGROUP per tile, a counter per base (the general chain's tile stage, after k_tile_fast) ----
// The reference's own formulation on one tile (savePileupExpt 2197-2273): difference array -> prefix sum ->
// run-length intervals, with the array (16 KB) in LDS.  Each thread owns four consecutive bases.
constexpr int TH_NT = 1024;
__global__ __launch_bounds__(TH_NT) void k_tile_heavy(TileIn in, const u32* __restrict__ heavyList, const u32* __restrict__ nHeavy,
                                                      TileOut out, u32* __restrict__ st) {
  static_assert(TILE == 4 * TH_NT, "four bases per thread");
  __shared__ int delta[TILE];
  __shared__ u32 scratch[40];
  __shared__ u32 vsRed[2];
  __shared__ u32 sLast;
  const int vsig = (int)__builtin_amdgcn_readfirstlane(loose_vsig(out.ctl, false, vsRed));
  const u32 nH = *nHeavy;
  for (u32 it = blockIdx.x; it < nH; it += gridDim.x) {
    const u32 t = heavyList[it];
    const TileMeta m = in.meta[t];
    __syncthreads();
    *reinterpret_cast<int4*>(delta + 4 * threadIdx.x) = make_int4(0, 0, 0, 0);
    if (threadIdx.x == 0) sLast = 0;
    __syncthreads();
    for (u32 k = threadIdx.x; k < m.nS; k += TH_NT) atomicAdd(&delta[in.S[m.sb + k]], GX_UNIT);
    for (u32 k = threadIdx.x; k < m.nE; k += TH_NT) atomicAdd(&delta[in.E[m.eb + k]], -GX_UNIT);
    for (u32 k = threadIdx.x; k < m.nF; k += TH_NT) {
      const u64 r = in.F[m.fb + k];
      atomicAdd(&delta[(u32)(r >> 8) & (TILE - 1)], (int)(int8_t)(r & 0xFF));
    }
    __syncthreads();
    const bool active = m.flags & TM_ACTIVE, lastTile = (m.flags & TM_LAST) != 0;
    const int4 d4 = *reinterpret_cast<const int4*>(delta + 4 * threadIdx.x);
    const int d[4] = {d4.x, d4.y, d4.z, d4.w};
    u32 nzc = 0;
    int dsum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      dsum += d[k];
      nzc += (u32)(d[k] != 0 && active && (m.pos0 + 4 * threadIdx.x + k != 0));
    }
    // the pileup before the thread's first base, and its output rank
    int dtot;
    const int dex = block_excl_scan<int, TH_NT>(dsum, reinterpret_cast<int*>(scratch), &dtot);
    u32 total;
    const u32 oex = block_excl_scan<u32, TH_NT>(nzc, scratch, &total);
    int run = m.carry + dex;
    u32 o = m.slot + oex;
    u32 neg = 0, big = (u32)(m.carry >= FRAG_FAST_MAXV), last = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 pos = m.pos0 + 4 * threadIdx.x + k;
      if (d[k] != 0 && active && pos != 0) {   // 2241: base 0 closes nothing
        out.looseEnd[o] = pos;
        out.looseV[o] = run;                   // the pileup of the interval that ends here (2244)
        if (run >= vsig) atomicOr((unsigned long long*)&out.sigMask[o >> 6], 1ull << (o & 63));
        o++;
        last = pos;
      }
      run += d[k];
      neg |= (u32)(run < 0);
      big |= (u32)(run >= FRAG_FAST_MAXV);
    }
    if (last) atomicMax(&sLast, last);
    // (__syncthreads_or answers "any non-zero?", not a bitwise or: one call per flag)
    const u32 flagsW = (__syncthreads_or((int)neg) ? 1u : 0u) | (__syncthreads_or((int)big) ? 2u : 0u);
    u32 lastEnd = sLast, all = total;
    if (active) {
      if (lastTile) {  // closing interval [.., len): 2268-2273
        const int runEnd = m.carry + dtot;
        if (threadIdx.x == 0) {
          const u32 oc = m.slot + total;
          out.looseEnd[oc] = m.len;
          out.looseV[oc] = runEnd;
          if (runEnd >= vsig) atomicOr((unsigned long long*)&out.sigMask[oc >> 6], 1ull << (oc & 63));
        }
        lastEnd = m.len;
        all = total + 1;
      }
      if (threadIdx.x == 0) {
        if (all) out.tileLastEnd[t] = lastEnd;
        if (flagsW & 1u) atomicOr(st, ST_NEG_PILE);
        if (flagsW & 2u) {
          atomicOr(&out.tileDeep[t], 1u);
          if (out.ctl) atomicOr(&out.ctl->bad, 1u);
        }
      }
    } else
      all = 0;
    if (threadIdx.x == 0) out.tileCount[t] = all;
    // (unused slots: zero-length intervals behind the last one, for the sweep on the loose slots)
    if (all && vsig != 0x7FFFFFFF) {
      const u32 size = m.nS + m.nE + m.nF + 1;
      for (u32 j = all + threadIdx.x; j < size; j += TH_NT) {
        out.looseEnd[m.slot + j] = lastEnd;
        out.looseV[m.slot + j] = 0;
      }
    }
  }
}

// fragLen.  savePileupExpt adds (float)len * val per interval into a double.  With unit weights
// only (no fractional records, no -E regions) val is an integer, and a product below 2^24 is
// exact, so the sum over such intervals is the number of covered (base, fragment) pairs: the sum
// of the clamped fragment lengths, which k_convert accumulates for free ("closed form").  Only
// intervals with len * val >= 2^24 can round, and they are easy to find:
//   * val >= 2^24 / (2 TILE): k_tile marks the tiles in which the pileup gets that deep;
//   * otherwise len >= 2 TILE: the interval starts before its tile does, so it is the tile's first.
// k_scan_iv looks at every other tile's first interval (and lists the deep tiles), k_frag_walk walks
// the listed ones; both add (rounded product - exact product), an integer, to a correction.
// Everything else (fractional weights, -E, wide records) takes k_frag_walk's general path, which
// walks every interval and accumulates exactly in (integer, fraction * 2^27) form.
struct FragFix {
  u64 fragSum[FRAG_SLOTS];  // closed form partial sums
  u32 slow;                 // general path wanted (bit 1: a fractional weight was seen, FRAG_SLOW_FRAC)
  u32 nList;
  long long corr;
  u32 nF;                   // fractional records appended by k_convert (not fragLen's, but zeroed with it)
  u32 pad_;
};

__device__ __forceinline__ long long frag_corr(u32 len, int v) {
  if (v <= 0) return 0;  // V_MARK never occurs on this path
  const u32 cnt = (u32)v / GX_UNIT;
  const float term = (float)len * (float)cnt;  // what getval returns for a whole pileup
  return (long long)term - (long long)((u64)len * cnt);
}

// tile interval counts -> tight offsets (sum scan) and, per tile, the end of the last interval
// that precedes it on the same chromosome (max scan over (chromosome, end) keys).  Persistent
// multi-workgroup chained scan, 2048 tiles per item; also fills chromIvOff for tiled
// chromosomes and the total.
__device__ __forceinline__ u64 lookback_excl_max(u64* lb, u32 id, u64 aggregate, u32* st) {
  // same protocol as lookback_excl with max instead of +
  u64 excl = 0;
  if (id > 0) {
    if (lane_id() == 0)
      __hip_atomic_store(&lb[id], LB_AGG | (aggregate & LB_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = (int)id - 1;
    u32 spins = 0;
    bool done = false;
    while (!done) {
      int idx = look - lane_id();
      u64 v = idx >= 0 ? __hip_atomic_load(&lb[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_INC;
      u64 flag = v >> 62;
      u64 invalidMask = __ballot(flag == 0);
      u64 incMask = __ballot(flag == 2);
      int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
      int firstInc = incMask ? __builtin_ctzll(incMask) : 64;
      int take = firstInc < firstInvalid ? firstInc + 1 : firstInvalid;
      if (take > 0) {
        u64 m = lane_id() < take ? (v & LB_MASK) : 0ull;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          u64 o = __shfl_xor(m, d, 64);
          m = o > m ? o : m;
        }
        excl = m > excl ? m : excl;
        if (firstInc < firstInvalid) done = true; else look -= firstInvalid;
      } else {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        bool abort_ = spins > LB_SPIN_LIMIT;
        if (!abort_ && (spins & 1023u) == 0)
          abort_ = (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_LOOKBACK) != 0;
        if (abort_) {
          if (lane_id() == 0) atomicOr(st, ST_LOOKBACK);
          done = true;
        }
      }
    }
  }
  u64 inc = aggregate > excl ? aggregate : excl;
  if (lane_id() == 0)
    __hip_atomic_store(&lb[id], LB_INC | (inc & LB_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// "My stores have completed": what a workgroup needs before it takes a ticket that tells another workgroup to read
// them, WHEN those stores were agent-scope atomic stores or went to host memory.  A release fence (__threadfence)
// would also write back the XCD's whole L2 -- tens of microseconds behind a kernel that left it full of dirty lines,
// and that once per workgroup (measured: k_scan_iv 27 -> 80 us, k_peaks 37 -> 104 us).
__device__ __forceinline__ void stores_done() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (the compiler keeps the stores above ...
  __builtin_amdgcn_s_waitcnt(0x0F70);        // ... vmcnt(0) ...
  __atomic_signal_fence(__ATOMIC_SEQ_CST);   // ... and what follows below)
}
__device__ __forceinline__ void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_agent(const u32* p) {
  return __hip_atomic_load(const_cast<u32*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct IvScanOut {
  u32* tileIvOff;   // [nTiles+1]
  u32* tilePrevEnd; // [nTiles] start of the tile's first interval
  u32* chromIvOff;  // [nChrom+1] (entries of chromosomes without tiles are filled by k_fix_chrom_off)
  u32* nIv;
  // for the sweep on the loose slots (LooseCtl): the chromosomes' first loose slots, and the slots of the tiles
  // without intervals filled with zero-length intervals that end where the previous interval ended
  const u32* tileSlot;   // [nTiles+1]
  u32* chromLooseOff;    // [nChrom+1]
  u32* looseEnd;
  int* looseV;
  LooseCtl* ctl;
  // fragLen's correction terms (what k_frag_fix1 did in a launch of its own): the deep tiles go on a list, and of
  // every other tile the first interval is looked at when it can be 2 tiles long -- only when the tile before holds
  // no interval at all
  const u32* tileDeep;
  FragFix* ff;
  u32* fragList;
  // the general fragLen path shared with the tile kernel (TileIn::fragAcc): the tiles' first intervals are added here
  long long* fragAcc = nullptr;
  // the reference's check that a chromosome's pileup returns to zero (savePileupExpt 2283-2289, savePileupCtrl 2152-2158):
  // [nChrom] weight (1/120) of the events that end at the chromosome's length -- all that may still be open in the
  // closing interval [.., len) of its last tile
  const u32* endAtLen = nullptr;
};

__device__ __forceinline__ void scan_iv_body(const u32* __restrict__ tileCount, const u32* __restrict__ tileLastEnd,
                                             const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                             u32 nTiles, u64* __restrict__ lbSum, u64* __restrict__ lbMax,
                                             const IvScanOut& out, u32* __restrict__ st) {
  __shared__ u32 scratch[8];
  __shared__ u64 s64[8];
  __shared__ u64 s_sum, s_max;
  const u32 nChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  const bool slowFrag = out.ff->slow != 0;
  const bool firstTerms = slowFrag && out.fragAcc != nullptr;
  long long fhi = 0, flo = 0;
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 tb = id * STL_CHUNK + threadIdx.x * STL_ITEMS;
    u32 c[STL_ITEMS];
    u64 key[STL_ITEMS];
    u32 cs = 0;
    u64 km = 0;
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      u32 t = tb + k;
      c[k] = t < nTiles ? tileCount[t] : 0;
      // key of a tile with intervals: (chromosome + 1, last end); 0 otherwise
      key[k] = c[k] ? (((u64)(tileChrom[t] + 1) << 32) | tileLastEnd[t]) : 0ull;
      cs += c[k];
      km = key[k] > km ? key[k] : km;
    }
    u32 ctot;
    u32 cex = block_excl_scan<u32, STL_NT>(cs, scratch, &ctot);
    // exclusive max scan across the workgroup (shuffles: 64-bit keys)
    u64 inc = km;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      u64 o = __shfl_up(inc, d, 64);
      if (lane_id() >= d && o > inc) inc = o;
    }
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) s64[wv] = inc;
    __syncthreads();
    u64 kex = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) kex = 0;
    u64 ktot = 0;
    for (int w = 0; w < STL_NT / 64; w++) {
      u64 x = s64[w];
      if (w < wv && x > kex) kex = x;
      if (x > ktot) ktot = x;
    }
    if (threadIdx.x < 64) {
      u64 es = lookback_excl(lbSum, id, (u64)ctot, st);
      u64 em = lookback_excl_max(lbMax, id, ktot, st);
      if (threadIdx.x == 0) {
        s_sum = es;
        s_max = em;
        if (id == nChunks - 1) {
          out.tileIvOff[nTiles] = (u32)es + ctot;
          st_agent(out.nIv, (u32)es + ctot);  // (agent-scope stores: k_scan_iv_close's last workgroup reads these three)
        }
      }
    }
    __syncthreads();
    cex += (u32)s_sum;
    if (s_max > kex) kex = s_max;
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      u32 t = tb + k;
      if (t < nTiles) {
        u32 ci = tileChrom[t];
        out.tileIvOff[t] = cex;
        const u32 prevEnd = (u32)(kex >> 32) == ci + 1 ? (u32)kex : 0u;
        out.tilePrevEnd[t] = prevEnd;
        if (c[k]) {
          if (firstTerms) {  // the tile's first interval: it starts where the last interval before the tile ended
            const u32 slot = out.tileSlot[t];
            frag_term(out.looseEnd[slot] - prevEnd, out.looseV[slot], fhi, flo);
          }
          if (out.tileDeep[t])  // (the list also serves the p-values of the deep tiles: k_pval_deep)
            out.fragList[atomicAdd(&out.ff->nList, 1u)] = t;
          else if (!slowFrag) {
            // the tile's first interval ends inside the tile: it is >= 2 TILE long only if it starts a whole tile earlier
            const u32 pos0 = (t - chroms[ci].tileBase) << TB;
            if (pos0 >= (u32)TILE && prevEnd <= pos0 - (u32)TILE) {
              const u32 slot = out.tileSlot[t];
              const u32 len = out.looseEnd[slot] - prevEnd;
              if (len >= 2u * TILE) {
                const long long cc = frag_corr(len, out.looseV[slot]);
                if (cc) atomicAdd((u64*)&out.ff->corr, (u64)cc);
              }
            }
          }
        }
        if (out.endAtLen && c[k] && t + 1 == chroms[ci].tileBase + chroms[ci].nTiles) {
          // the chromosome's last tile: its last interval is the closing one, [.., len) (2268-2273).  Behind it the
          // reference's difference array must be back at zero (updateVal(diff[len]), 2283-2289); diff[len] holds the ends
          // of the fragments that reach the chromosome's end, which have no record here (k_sort1 / k_sort_a count their
          // weight per chromosome): the closing pileup must be exactly that weight.  (-E: a closing interval inside an
          // excluded region carries no pileup)
          const int vEnd = out.looseV[out.tileSlot[t] + c[k] - 1];
          if (vEnd != V_MARK && vEnd != (int)out.endAtLen[ci]) atomicOr(st, ST_END_PILE);
        }
        if (t == chroms[ci].tileBase) {
          st_agent(&out.chromIvOff[ci], cex);
          st_agent(&out.chromLooseOff[ci], out.tileSlot[t]);
        }
        if (c[k] == 0 && (chroms[ci].flags & CH_SAVE)) {
          // (a tile without intervals has few slots -- one more than it has records, all of which cancel; if it has
          // many, the sweep on the loose slots is called off rather than filled in by one thread)
          const u32 s0 = out.tileSlot[t], s1 = out.tileSlot[t + 1];
          if (s1 - s0 > 64u)
            atomicOr(&out.ctl->bad, 4u);
          else
            for (u32 j = s0; j < s1; j++) {
              out.looseEnd[j] = prevEnd;
              out.looseV[j] = 0;
            }
        }
      }
      cex += c[k];
      if (key[k] > kex) kex = key[k];
    }
    __syncthreads();
  }
  if (firstTerms) {  // block-uniform
    fhi = wave_sum(fhi);
    flo = wave_sum(flo);
    if (lane_id() == 0) {
      if (fhi) atomicAdd((u64*)&out.fragAcc[0], (u64)fhi);
      if (flo) atomicAdd((u64*)&out.fragAcc[1], (u64)flo);
    }
  }
}

__global__ __launch_bounds__(STL_NT) void k_scan_iv(const u32* __restrict__ tileCount, const u32* __restrict__ tileLastEnd,
                                                    const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                                    u32 nTiles, u64* __restrict__ lbSum, u64* __restrict__ lbMax,
                                                    IvScanOut out, u32* __restrict__ st) {
  scan_iv_body(tileCount, tileLastEnd, tileChrom, chroms, nTiles, lbSum, lbMax, out, st);
}

// One wavefront per tile walks the tile's loose slots (read only).  Closed form: the deep tiles of the list,
// adding (rounded - exact) products to the correction.  General path (fractional weights, -E, wide
// records): every tile, accumulating every product exactly in (integer, fraction * 2^27) form.
__global__ __launch_bounds__(256) void k_frag_walk(const u32* __restrict__ looseEnd, const int* __restrict__ looseV,
                                                   const TileMeta* __restrict__ meta, const u32* __restrict__ tileIvOff,
                                                   const u32* __restrict__ tilePrevEnd, u32 nTiles, FragFix* __restrict__ ff,
                                                   const u32* __restrict__ list, long long* __restrict__ acc,
                                                   const u32* __restrict__ heavyList, const u32* __restrict__ nHeavy) {
  // heavyList: the tile stage was k_tile_fast, which adds the general path's terms itself (TileIn::fragAcc) -- but for
  // the heavy tiles (k_tile_heavy's): of those, everything behind the first interval (k_scan_iv's) is added here
  const bool slow = ff->slow != 0, heavyOnly = slow && heavyList != nullptr;
  const u32 nItems = heavyOnly ? *nHeavy : slow ? nTiles : ff->nList;
  long long c = 0, hi = 0, lo = 0;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 it = blockIdx.x * 4 + wv; it < nItems; it += gridDim.x * 4) {
    const u32 t = heavyOnly ? heavyList[it] : slow ? it : list[it];
    const u32 src = meta[t].slot, n = tileIvOff[t + 1] - tileIvOff[t];
    u32 prevEnd = tilePrevEnd[t];
    for (u32 b = 0; b < n; b += 64) {
      const u32 i = b + lane;
      u32 e = 0;
      int v = 0;
      if (i < n) {
        e = looseEnd[src + i];
        v = looseV[src + i];
      }
      u32 s = __shfl_up(e, 1, 64);
      if (lane == 0) s = prevEnd;
      prevEnd = __shfl(e, 63, 64);
      if (i < n && !(heavyOnly && i == 0)) {
        if (slow) frag_term(e - s, v, hi, lo); else c += frag_corr(e - s, v);
      }
    }
  }
  if (slow) {
    hi = wave_sum(hi);
    lo = wave_sum(lo);
    if (lane == 0) {
      if (hi) atomicAdd((u64*)&acc[0], (u64)hi);
      if (lo) atomicAdd((u64*)&acc[1], (u64)lo);
    }
  } else {
    c = wave_sum(c);
    if (lane == 0 && c) atomicAdd((u64*)&ff->corr, (u64)c);
  }
}

// loose slots -> tight arrays (needed when a control pileup is merged against this one)
struct PackIn {
  const u32* looseEnd;
  const int* looseV;
  const TileMeta* meta;    // .slot = first loose slot of the tile
  const u32* tileIvOff;
};

__global__ __launch_bounds__(256) void k_pack(PackIn in, u32 nTiles, u32* __restrict__ ivEnd, int* __restrict__ ivV) {
  const int wv = threadIdx.x >> 6, lane = lane_id();
  // one wavefront per tile (a tile holds a few hundred intervals); four independent pairs of
  // loads per lane and the next tile's header in flight: the shape is latency-bound
  const u32 stride = gridDim.x * 4;
  u32 t = blockIdx.x * 4 + wv;
  u32 src1 = 0, dst1 = 0, n1 = 0;
  if (t < nTiles) {
    src1 = in.meta[t].slot;
    dst1 = in.tileIvOff[t];
    n1 = in.tileIvOff[t + 1] - dst1;
  }
  for (; t < nTiles; t += stride) {
    const u32 src = src1, dst = dst1, n = n1;
    if (t + stride < nTiles) {
      src1 = in.meta[t + stride].slot;
      dst1 = in.tileIvOff[t + stride];
      n1 = in.tileIvOff[t + stride + 1] - dst1;
    }
    for (u32 b = 0; b < n; b += 256) {
      u32 e[4];
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        e[k] = 0;
        v[k] = 0;
        if (i < n) {
          e[k] = in.looseEnd[src + i];
          v[k] = in.looseV[src + i];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        if (i < n) {
          ivEnd[dst + i] = e[k];
          ivV[dst + i] = v[k];
        }
      }
    }
  }
}

// chromosome table epilogue: chromIvOff for chromosomes without tiles (inactive ones get an
// empty range) -- single thread, nChrom is small.
__global__ void k_fix_chrom_off(const DChrom* __restrict__ chroms, u32 nChrom, u32* __restrict__ chromIvOff,
                                const u32* __restrict__ nIv) {
  if (threadIdx.x || blockIdx.x) return;
  u32 next = *nIv;
  chromIvOff[nChrom] = next;
  for (int c = (int)nChrom - 1; c >= 0; c--) {
    if (chroms[c].tileBase == NULL_TILE)
      chromIvOff[c] = next;
    else
      next = chromIvOff[c];
  }
}

// scalars of one replicate, kept on the device so no host round trip sits between kernels
struct Scalars {
  long long fragAcc[2];  // treatment fragLen: integer part, fraction * 2^27
  long long ctrlAcc[2];  // control
  double fragLen;
  double ctrlFrag;
  float lambda;
  float factor;
  u64 genomeLen;
  u64 fracSeen;          // the sample just closed held a fractional weight (count > 1: Genrich's -s)
};

// fragLen / ctrlFrag (exact fixed-point parts, summed over all ranks when `coll` is given) -> lambda, factor
__device__ __forceinline__ void finish_frag(Scalars* s, int isCtrl, u32* st, const long long* __restrict__ coll) {
  long long* acc = isCtrl ? s->ctrlAcc : s->fragAcc;
  if (coll) {  // the sums over all ranks
    acc[0] = coll[0];
    acc[1] = coll[1];
  }
  if (!isCtrl) {
    s->fragLen = (double)s->fragAcc[0] + (double)s->fragAcc[1] * (1.0 / 134217728.0);
    if (s->fragLen == 0.0) atomicOr(st, ST_NO_FRAGS);
  } else {
    s->ctrlFrag = (double)s->ctrlAcc[0] + (double)s->ctrlAcc[1] * (1.0 / 134217728.0);
    s->factor = s->ctrlFrag == 0.0 ? 1.0f : (float)(s->fragLen / s->ctrlFrag);  // calcFactor 2043-2045
  }
  s->lambda = (float)(s->fragLen / (double)s->genomeLen);  // calcLambda 1831
}

// closed form or general path -> the (integer, fraction * 2^27) accumulator pair; with several ranks
// also this rank's contribution to the all-reduce: the pair and its "build this sample again" flags
// (every rank must learn whether any rank has to, before the sums mean anything): +1 when a base can
// reach the reference's int16 limits, +65536 when a level-1 page list overflowed (ST_PT_FULL = 512)
// (acc points into *scal: no __restrict__ on either)
struct FragSelect {
  const FragFix* ff;
  long long* acc;            // (points into *scal: no __restrict__ on either)
  long long* coll;
  const u32* hot;
  u32* st;
  const DChrom* chroms;
  u32 nChrom;
  u32* chromIvOff;
  const u32* nIv;
  Scalars* scal;
  int isCtrl;
  u32* chromLooseOff;
  const u32* tileSlot;
  u32 nTiles;
  LooseCtl* ctl;
  u64* brkLoose;             // the sweep's "first of its chromosome" mask in loose-slot index space (or null)
};

// (ivIn / looseIn: LDS copies of chromIvOff / chromLooseOff to read from, or null)
__device__ __forceinline__ void frag_select_body(const FragSelect& A, const u32* ivIn = nullptr, const u32* looseIn = nullptr) {
  const FragFix* ff = A.ff;
  long long* acc = A.acc;
  long long* coll = A.coll;
  u32* st = A.st;
  Scalars* scal = A.scal;
  LooseCtl* ctl = A.ctl;
  {  // chromosome table epilogue (as k_fix_chrom_off): offsets of the chromosomes without tiles
    u32 next = *A.nIv, nextL = A.tileSlot[A.nTiles];
    A.chromIvOff[A.nChrom] = next;
    A.chromLooseOff[A.nChrom] = nextL;
    for (int c = (int)A.nChrom - 1; c >= 0; c--) {
      if (A.chroms[c].tileBase == NULL_TILE) {
        A.chromIvOff[c] = next;
        A.chromLooseOff[c] = nextL;
      } else {
        next = ivIn ? ivIn[c] : A.chromIvOff[c];
        nextL = looseIn ? looseIn[c] : A.chromLooseOff[c];
        // (one thread: a chain of read-modify-writes; k_close marks the starts with all its lanes instead)
        if (A.brkLoose) A.brkLoose[nextL >> 6] |= 1ull << (nextL & 63);
      }
    }
  }
  const u32* hot = A.hot;
  const int isCtrl = A.isCtrl;
  scal->fracSeen = (ff->slow >> 1) & 1u;
  if (!ff->slow) {
    u64 t = 0;
    for (int i = 0; i < FRAG_SLOTS; i++) t += ff->fragSum[i];
    acc[0] = (long long)t + ff->corr;
    acc[1] = 0;
  }
  if (coll) {
    coll[0] = acc[0];
    coll[1] = acc[1];
    coll[2] = (*hot ? 1 : 0) + ((*st & 512u) ? 65536 : 0) + ((*st & 3072u) ? (1ll << 32) : 0);  // (ST_SB_FULL | ST_SB_FRAC)
  } else
    finish_frag(scal, isCtrl, st, nullptr);  // one rank: the sums are final (otherwise k_finish_frag, after the all-reduce)
  // the sweep may use the loose slots if the tile stage wrote its bits with the lambda that turned out final
  ctl->ok = !coll && ctl->enabled && !ctl->bad && ctl->earlyBits == __float_as_uint(scal->lambda) ? 1u : 0u;
}

__global__ void k_frag_select(FragSelect A) {
  if (threadIdx.x || blockIdx.x) return;
  frag_select_body(A);
}

// a replicate begins: its scalars at zero, the genome length in place
__global__ void k_begin_sample(Scalars* __restrict__ s, u64 genomeLen) {
  static_assert(sizeof(Scalars) % 8 == 0 && sizeof(Scalars) / 8 <= 64, "one wavefront clears the block");
  u64* w = reinterpret_cast<u64*>(s);
  if (threadIdx.x < sizeof(Scalars) / 8) w[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) s->genomeLen = genomeLen;
}

// What a build starts from, in ONE launch (round 3: k_begin_sample, then two fill launches): the replicate's scalars
// (`begin`: gx_sample_begin leaves them to this kernel), the arena of everything that must start at zero, and the
// sweep's masks in loose-slot index space.  Sizes in 16-byte units.
__global__ __launch_bounds__(256) void k_build_init(Scalars* __restrict__ s, int begin, u64 genomeLen, uint4* __restrict__ a,
                                                    size_t nA, uint4* __restrict__ b, size_t nB) {
  if (begin && blockIdx.x == 0) {  // (block-uniform)
    u64* w = reinterpret_cast<u64*>(s);
    if (threadIdx.x < sizeof(Scalars) / 8) w[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) s->genomeLen = genomeLen;
  }
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nA; i += stride) a[i] = z;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nB; i += stride) b[i] = z;
}

// (ctl: several ranks that knew lambda ahead of the tile stage -- was it the final one?  frag_select_body's verdict for
// one rank)
__global__ void k_finish_frag(Scalars* s, int isCtrl, u32* st, const long long* __restrict__ coll, LooseCtl* ctl) {
  if (threadIdx.x || blockIdx.x) return;
  finish_frag(s, isCtrl, st, coll);
  if (ctl) ctl->ok = ctl->enabled && !ctl->bad && ctl->earlyBits == __float_as_uint(s->lambda) ? 1u : 0u;
}

// Everything the host wants to know at a synchronisation point, written by ONE small kernel straight into
// pinned host memory (instead of one copy launch per word): scalars, status, flags, and the list of risky
// p-values (its count and first records).
struct MailOut {
  Scalars* scal;
  u32* status;
  u32* hot;
  u32* nIv;
  long long* again;   // the ranks' "build this sample again" flags (several ranks only)
  u32* extra;         // one more word (interval count of a merge)
  u64* extra64;       // ... and a 64-bit one (the peaks' total length)
  RiskBuf* risk;
  u32* seq;           // written last: the host polls it (mail_sync)
};

__device__ __forceinline__ void mail_body(const Scalars* __restrict__ ds, const u32* __restrict__ st, const u32* __restrict__ hot,
                                          const u32* __restrict__ nIv, const long long* __restrict__ coll,
                                          const u32* __restrict__ extra, const RiskBuf* __restrict__ rb, const MailOut& m, u32 seq,
                                          const u64* __restrict__ extra64 = nullptr) {
  const u32 n = rb->count;
  if (threadIdx.x == 0) {
    if (extra64) *m.extra64 = *extra64;
    if (ds) *m.scal = *ds;
    *m.status = *st;
    if (hot) *m.hot = *hot;
    if (nIv) *m.nIv = *nIv;
    if (coll) *m.again = coll[2];
    if (extra) *m.extra = *extra;
    m.risk->count = n;
  }
  if (threadIdx.x < n && threadIdx.x < RISK_PREFIX) m.risk->rec[threadIdx.x] = rb->rec[threadIdx.x];
  // the sequence number goes last, behind a system-scope fence: a host that sees it sees the mail
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(m.seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_mail(const Scalars* __restrict__ ds, const u32* __restrict__ st, const u32* __restrict__ hot,
                                             const u32* __restrict__ nIv, const long long* __restrict__ coll,
                                             const u32* __restrict__ extra, const RiskBuf* __restrict__ rb, MailOut m,
                                             u32 seq, const u64* __restrict__ extra64) {
  mail_body(ds, st, hot, nIv, coll, extra, rb, m, seq, extra64);
}

// The end of a treatment sample whose lambda -- and with it the table p(V) -- was known before the tile stage (LooseCtl):
// k_frag_select's work and the mail in ONE launch, when nothing else stands between them: no deep tile to walk
// (k_frag_walk), no general fragLen path, and the table built for the lambda that turns out final.  Otherwise
// `*closeState` = 2 (in the mail block) tells the host to run the separate kernels (k_frag_walk, k_frag_select, k_pval_lut, k_mail) after all.
__device__ __forceinline__ void close_body(const FragSelect& A, const u32* nIv, const u32* extra, const RiskBuf* rb,
                                           const MailOut& m, u32* closeState, u32 seq) {
  // (the single thread of k_frag_select walks ~90 dependent loads -- the 64 partial sums, the chromosome table: the
  // wavefront fetches them side by side into LDS first, where the walk costs nothing)
  __shared__ FragFix sff;
  __shared__ DChrom schrom[64];
  __shared__ u32 sIv[64], sLoose[64];
  const bool small = A.nChrom <= 64;
  if (threadIdx.x < FRAG_SLOTS) sff.fragSum[threadIdx.x] = A.ff->fragSum[threadIdx.x];
  if (threadIdx.x == 0) {
    sff.slow = A.ff->slow;
    sff.nList = A.ff->nList;
    sff.corr = A.ff->corr;
  }
  if (small && threadIdx.x < A.nChrom) {
    schrom[threadIdx.x] = A.chroms[threadIdx.x];
    sIv[threadIdx.x] = A.chromIvOff[threadIdx.x];
    sLoose[threadIdx.x] = A.chromLooseOff[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 ok = 0;
    if (!sff.slow && sff.nList == 0 && !A.coll) {
      FragSelect B = A;
      B.ff = &sff;
      B.brkLoose = nullptr;  // (the lanes mark the chromosome starts below)
      if (small) {  // (entries of chromosomes with tiles are only read, the others only written: the copies serve)
        B.chroms = schrom;
        frag_select_body(B, sIv, sLoose);
      } else
        frag_select_body(B, nullptr, nullptr);
      ok = A.ctl->enabled && A.ctl->earlyBits == __float_as_uint(A.scal->lambda);
    }
    *closeState = ok ? 1u : 2u;  // (pinned host memory, ahead of the mail's fence and sequence number)
  }
  // the chromosomes' first loose slots in the sweep's mask (chromLooseOff of a chromosome with tiles: k_scan_iv's)
  if (A.brkLoose)
    for (u32 c = threadIdx.x; c < A.nChrom; c += blockDim.x)
      if ((small ? schrom[c].tileBase : A.chroms[c].tileBase) != NULL_TILE) {
        const u32 a = small ? sLoose[c] : A.chromLooseOff[c];
        atomicOr((unsigned long long*)&A.brkLoose[a >> 6], 1ull << (a & 63));
      }
  __syncthreads();
  mail_body(A.scal, A.st, A.hot, nIv, nullptr, extra, rb, m, seq);
}

// k_scan_iv and k_close in one launch: the workgroup of the scan that finishes LAST closes the sample.
struct CloseArgs {
  FragSelect sel;
  const u32* nIv;
  const u32* extra;
  const RiskBuf* rb;
  MailOut m;
  u32* closeState;
  u32 seq;
  u32* ticket;   // zero before the launch; left at zero
};
__global__ __launch_bounds__(STL_NT) void k_scan_iv_close(const u32* __restrict__ tileCount, const u32* __restrict__ tileLastEnd,
                                                          const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                                          u32 nTiles, u64* __restrict__ lbSum, u64* __restrict__ lbMax,
                                                          IvScanOut out, u32* __restrict__ st, CloseArgs C) {
  __shared__ u32 s_last;
  scan_iv_body(tileCount, tileLastEnd, tileChrom, chroms, nTiles, lbSum, lbMax, out, st);
  // (what the closing workgroup reads of the others: the interval count and the chromosomes' offsets -- agent-scope
  // stores -- and words only ever touched by atomics: fragLen's correction, the list count, LooseCtl::bad, the status)
  stores_done();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(C.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {  // block-uniform
    if (threadIdx.x == 0) *C.ticket = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    close_body(C.sel, C.nIv, C.extra, C.rb, C.m, C.closeState, C.seq);
  }
}

}  // namespace gx
