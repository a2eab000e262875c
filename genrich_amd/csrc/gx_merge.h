// gx_merge.h -- interval merges done tile by tile with LDS bitmaps (gfx950).
//
//  k_merge2 : savePval's two-pointer merge of the treatment and control pileups
//             (Genrich.c:1768-1791) together with savePileupCtrl's value rule
//             net = max(factor * val, lambda) and its "break only where net changes"
//             run-length rule (2107-2127).
//  k_mergeN : combinePval's union of all replicates' breakpoints (612-667) + multPval (567-583).
//
// Both run-length inputs were produced tile by tile, so "the intervals that end inside tile t"
// are a contiguous slice [tileOff[t], tileOff[t+1]).  A tile's breakpoints are set as bits of a
// 2^TB-bit LDS bitmap per input; the union is a bitwise OR; an input's covering interval at
// a union breakpoint j is slice_begin + popcount(bits before j).  Like k_tile, a tile writes into
// its own loose slot and reports a count; k_scan_counts and the pack kernels make the result tight.
#pragma once
#include "gx_stats.h"

namespace gx {

constexpr int MG_WORDS_ = TILE / 32;
constexpr int MG_NT = MG_WORDS_ < 256 ? MG_WORDS_ : 256;
constexpr int MG_WORDS = TILE / 32;        // 512 bitmap words per input
constexpr int MG_WPT = MG_WORDS / MG_NT;   // 2 consecutive words per thread

struct RleIn {   // run-length pileup of a sample
  const u32* end;
  const int* v;
  const u32* tileOff;     // [nTiles + 1] tight offsets (k_scan_iv): a tile's interval count, its share of the output slot
  const TileMeta* meta;   // LOOSE: the sample's own tile descriptors -- `end` / `v` are its loose slots, tile t's intervals
                          // start at meta[t].slot, and the interval that covers the tile's tail has the pileup
                          // meta[t + 1].carry (no breakpoint lies between: it is the next tile's first interval)
};

struct Merge2Out {   // loose slots: tile t writes at A.tileOff[t] + B.tileOff[t] (a union is never longer)
  u32* end;
  int* exptV;        // treatment pileup (1/120 units, V_MARK inside -E regions)
  int* ctrlV;        // control pileup of the covering control interval
  u32* tileCount;    // [nTiles]
  // PV (round 6): p of the interval, looked up while both pileups are in registers -- the two int arrays are then only written
  // for the intervals the tables do not hold (p = MG_P_MISS: a fractional or very deep pileup), whose tiles go on a list
  u32* pBits;
  u32* missList;
  u32* nMiss;
};
constexpr u32 MG_P_MISS = 0x7FC0DEADu;   // (a NaN no evaluation produces: "p still to be computed", k_pairs_missed)

// (expt_val / ctrl_net: the two pileup floats of a p-interval, gx_math.h)

// No inter-workgroup dependency (like k_tile): counts go to k_scan_counts, packing to k_pack_pairs.
// Per tile the work is small (a few hundred breakpoints) and the kernel is bound by the chain of
// dependent loads, so, as in k_tile, tile headers are fetched two tiles ahead and the first MG_NT
// intervals of both inputs one tile ahead; the pileup values of the tile's intervals are staged
// in LDS so that the emit loop does no dependent global gather.
// LOOSE (round 3): the two samples are read where the tile stage left them -- their loose slots -- instead of from
// tight copies (k_pack twice: 0.66 ms and 2.7 GB per step at config 3).  Not with -E regions: there the value of the
// interval behind a tile's last breakpoint can be V_MARK, which no carry says.
constexpr int MG_CAP = 1024;  // intervals per input and tile whose values are staged in LDS
constexpr u32 PT_N = 256, PT_HOT = 64;   // the table of p-values of whole pileup pairs (k_pair_tab2d below) and its corner kept in LDS

struct MergeHdr {
  u32 a0, a1c, a1, b0, b1c, b1, pos0, len, flags;  // flags: 1 active, 2 last tile of its chromosome
  u32 slot;                                         // the tile's first output slot
  int nvA, nvB;                                     // LOOSE: pileups of the intervals that cover the tile's tail
};

template <bool LOOSE>
__device__ __forceinline__ MergeHdr merge_hdr(const RleIn& A, const RleIn& B, const TileMeta* __restrict__ meta, u32 t, u32 nTiles) {
  MergeHdr h;
  const TileMeta m = meta[t];
  h.pos0 = m.pos0;
  h.len = m.len;
  h.flags = m.flags & 3u;
  const u32 ao = A.tileOff[t], bo = B.tileOff[t];
  const u32 na = A.tileOff[t + 1] - ao, nb = B.tileOff[t + 1] - bo;
  h.slot = ao + bo;
  h.nvA = 0;
  h.nvB = 0;
  if (LOOSE) {
    h.a0 = A.meta[t].slot;
    h.b0 = B.meta[t].slot;
    if (t + 1 < nTiles) {
      h.nvA = A.meta[t + 1].carry;
      h.nvB = B.meta[t + 1].carry;
    }
  } else {
    h.a0 = ao;
    h.b0 = bo;
  }
  h.a1 = h.a0 + na;
  h.b1 = h.b0 + nb;
  if (!(h.flags & 1u)) { h.a1 = h.a0; h.b1 = h.b0; }
  const bool lastTile = h.flags & 2u;
  h.a1c = (lastTile && h.a1 > h.a0) ? h.a1 - 1 : h.a1;  // the chromosome-closing interval is handled apart
  h.b1c = (lastTile && h.b1 > h.b0) ? h.b1 - 1 : h.b1;
  return h;
}

// PV: the p-value of every merged interval on the way (k_pack_pairs' table look-ups, the corner of the table in LDS): the loose
// intermediate is (end, p) -- 8 bytes per interval instead of 12 -- and what follows is a copy (k_pack_ep2) instead of k_pack_pairs.
constexpr u32 MG_HOT = 32;   // the corner of the pair table kept in LDS by k_merge2<.., true> (4 KiB: eight workgroups per CU stay)
template <bool LOOSE, bool PV = false>
__global__ __launch_bounds__(MG_NT) void k_merge2(RleIn A, RleIn B, const Scalars* __restrict__ sc,
                                                  const TileMeta* __restrict__ meta, u32 nTiles, Merge2Out out,
                                                  u32* __restrict__ st, const float* __restrict__ p2d) {
  __shared__ u32 bmA[MG_WORDS], bmB[MG_WORDS], bmC[MG_WORDS];
  __shared__ u64 scratch64[8];
  __shared__ int sA[MG_CAP + 1], sC[MG_CAP + 1];
  __shared__ float hotP[PV ? MG_HOT * MG_HOT : 1];
  __shared__ u32 sMiss;
  static_assert(MG_WPT == 1, "one bitmap word per thread");
  const float factor = sc->factor, lambda = sc->lambda;
  u32 neg = 0;
  const u32 G = gridDim.x;
  if (PV) {
    for (u32 i = threadIdx.x; i < MG_HOT * MG_HOT; i += MG_NT) hotP[i] = p2d[(i / MG_HOT) * PT_N + (i % MG_HOT)];
    if (threadIdx.x == 0) sMiss = 0;
  }
  // one merged interval: its two pileups, or (PV) its p-value -- whole pileups below PT_N on both sides come from the table (as in
  // k_pack_pairs), a pair of V_MARKs is SKIP (inside a -E region), anything else is left to k_pairs_missed with its pileups
  auto emitV = [&](u32 o, int va, int vc) {
    if constexpr (PV) {
      u32 bits = MG_P_MISS;
      if (va == V_MARK || vc == V_MARK) {
        if (va == V_MARK && vc == V_MARK) bits = __float_as_uint(GX_SKIPF);
      } else {
        const u32 ec = __umulhi((u32)va, 0x88888889u) >> 6, cc = __umulhi((u32)vc, 0x88888889u) >> 6;
        if (va >= 0 && vc >= 0 && ec < PT_N && cc < PT_N && ec * GX_UNIT == (u32)va && cc * GX_UNIT == (u32)vc)
        {
          // (never `in LDS ? hotP[..] : p2d[..]`: one FLAT load of a selected pointer -- the corner is read in any case, at a
          // clamped index, and the table behind a branch of its own)
          // (... and the corner through an atomic load, which no pass folds into the other one)
          bits = __hip_atomic_load(reinterpret_cast<const u32*>(&hotP[(ec & (MG_HOT - 1)) * MG_HOT + (cc & (MG_HOT - 1))]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          if (ec >= MG_HOT || cc >= MG_HOT) bits = __float_as_uint(p2d[ec * PT_N + cc]);
        }
      }
      out.pBits[o] = bits;
      if (bits == MG_P_MISS) {
        out.exptV[o] = va;
        out.ctrlV[o] = vc;
        sMiss = 1;
      }
    } else {
      out.exptV[o] = va;
      out.ctrlV[o] = vc;
    }
  };
  bmA[threadIdx.x] = 0;
  bmB[threadIdx.x] = 0;
  bmC[threadIdx.x] = 0;
  const u32 lb = xcd_local_block(blockIdx.x, G);  // neighbouring tiles (neighbouring loose slots) on one XCD
  MergeHdr h1{}, h2{};
  if (lb < nTiles) h1 = merge_hdr<LOOSE>(A, B, meta, lb, nTiles);
  if (lb + G < nTiles) h2 = merge_hdr<LOOSE>(A, B, meta, lb + G, nTiles);
  // the pileup of the control interval BEHIND interval j of the tile: the next slot -- unless j is the tile's last
  // interval and the tile not its chromosome's last (LOOSE: the next tile's slots are elsewhere; its carry says it)
  auto nextB = [&](const MergeHdr& h, u32 j) -> int {
    const int v = B.v[h.b0 + j + 1];  // (in bounds: the slot arrays carry slack)
    return LOOSE && h.b0 + j + 1 == h.b1 && !(h.flags & 2u) ? h.nvB : v;
  };
  u32 eA1 = 0, eB1 = 0;
  int vA1 = 0, vB1 = 0, vBn1 = 0;
  if (h1.a0 + threadIdx.x < h1.a1c) { eA1 = A.end[h1.a0 + threadIdx.x]; vA1 = A.v[h1.a0 + threadIdx.x]; }
  if (h1.b0 + threadIdx.x < h1.b1c) {
    eB1 = B.end[h1.b0 + threadIdx.x];
    vB1 = B.v[h1.b0 + threadIdx.x];
    vBn1 = nextB(h1, threadIdx.x);
  }
  __syncthreads();
  for (u32 t = lb; t < nTiles; t += G) {
    const MergeHdr h = h1;
    const u32 eA0 = eA1, eB0 = eB1;
    const int vA0 = vA1, vB0 = vB1, vBn0 = vBn1;
    h1 = h2;
    if (t + G < nTiles) {
      if (h1.a0 + threadIdx.x < h1.a1c) { eA1 = A.end[h1.a0 + threadIdx.x]; vA1 = A.v[h1.a0 + threadIdx.x]; }
      if (h1.b0 + threadIdx.x < h1.b1c) {
        eB1 = B.end[h1.b0 + threadIdx.x];
        vB1 = B.v[h1.b0 + threadIdx.x];
        vBn1 = nextB(h1, threadIdx.x);  // exists: at least the closing interval follows
      }
    }
    if (t + 2 * G < nTiles) h2 = merge_hdr<LOOSE>(A, B, meta, t + 2 * G, nTiles);
    const bool active = h.flags & 1u, lastTile = h.flags & 2u;
    const u32 a0 = h.a0, b0 = h.b0, pos0 = h.pos0;
    const u32 nA = h.a1c - a0, nB = h.b1c - b0;
    const bool staged = nA <= MG_CAP && nB <= MG_CAP;  // block-uniform
    // the pileups of the intervals that cover what follows the tile's last breakpoint
    const bool tailElsewhere = LOOSE && !lastTile;
    // (bitmap words are zero here: each thread clears its words as soon as it has read them)
    // (the prefetched first MG_NT intervals of each input come from registers, the rest -- dense
    // tiles only -- from memory: two pieces of straight code, not a select inside one loop)
    auto addA = [&](u32 j, u32 e, int v) {
      const u32 off = e - pos0;
      atomicOr(&bmA[off >> 5], 1u << (off & 31));
      if (staged) sA[j] = v;
    };
    auto addB = [&](u32 j, u32 e, int v, int vn) {
      const u32 off = e - pos0;
      bool ng1, ng2;
      const float here = ctrl_net(v, factor, lambda, &ng1);
      const float next = ctrl_net(vn, factor, lambda, &ng2);
      neg |= ng1 | ng2;
      atomicOr(&bmC[off >> 5], 1u << (off & 31));
      if (here != next) atomicOr(&bmB[off >> 5], 1u << (off & 31));  // 2122: net != MAX(val, lambda)
      if (staged) sC[j] = v;
    };
    if (threadIdx.x < nA) addA(threadIdx.x, eA0, vA0);
    for (u32 j = threadIdx.x + MG_NT; j < nA; j += MG_NT) addA(j, A.end[a0 + j], A.v[a0 + j]);
    if (threadIdx.x < nB) addB(threadIdx.x, eB0, vB0, vBn0);
    for (u32 j = threadIdx.x + MG_NT; j < nB; j += MG_NT) addB(j, B.end[b0 + j], B.v[b0 + j], nextB(h, j));
    if (active && staged && threadIdx.x == 0) {  // the interval that covers what follows the tile's last breakpoint
      sA[nA] = tailElsewhere ? h.nvA : A.v[h.a1c];
      sC[nB] = tailElsewhere ? h.nvB : B.v[h.b1c];
    }
    __syncthreads();
    const u32 wA = bmA[threadIdx.x], wC = bmC[threadIdx.x], wU = wA | bmB[threadIdx.x];
    bmA[threadIdx.x] = 0;
    bmB[threadIdx.x] = 0;
    bmC[threadIdx.x] = 0;
    const u32 cU = __popc(wU), cA = __popc(wA), cC = __popc(wC);
    // one scan for the three counts (each total <= TILE + 1 < 2^16)
    u64 tot3;
    const u64 ex3 = block_excl_scan<u64, MG_NT>((u64)cU | ((u64)cA << 16) | ((u64)cC << 32), scratch64, &tot3);
    const u32 tU = (u32)tot3 & 0xFFFFu;
    const u32 exU = (u32)ex3 & 0xFFFFu, exA = (u32)(ex3 >> 16) & 0xFFFFu, exC = (u32)(ex3 >> 32) & 0xFFFFu;
    const u32 slot = h.slot;
    if (threadIdx.x == 0) out.tileCount[t] = active ? tU + (lastTile ? 1u : 0u) : 0u;
    if (active) {  // block-uniform
      u32 o = slot + exU;
      if (staged) {  // (two loops, not a select inside one: the choice is per tile)
        for (u32 bits = wU; bits; bits &= bits - 1) {
          const int b = __ffs(bits) - 1;
          const u32 below = (1u << b) - 1;
          out.end[o] = pos0 + threadIdx.x * 32 + b;
          emitV(o, sA[exA + __popc(wA & below)], sC[exC + __popc(wC & below)]);
          o++;
        }
      } else {
        for (u32 bits = wU; bits; bits &= bits - 1) {
          const int b = __ffs(bits) - 1;
          const u32 below = (1u << b) - 1;
          out.end[o] = pos0 + threadIdx.x * 32 + b;
          const u32 ia = exA + __popc(wA & below), ic = exC + __popc(wC & below);
          const int va = A.v[a0 + ia], vc = B.v[b0 + ic];  // (in bounds: slack)
          emitV(o, tailElsewhere && ia == nA ? h.nvA : va, tailElsewhere && ic == nB ? h.nvB : vc);
          o++;
        }
      }
      if (lastTile && threadIdx.x == 0) {  // 1779-1788 at the chromosome end: both pileups close at len
        const u32 oc = slot + tU;
        out.end[oc] = h.len;
        emitV(oc, A.v[h.a1 - 1], B.v[h.b1 - 1]);
      }
    }
    // (block_excl_scan ends with a barrier after the scratch reads; the staged values are read above)
    __syncthreads();
    // (PV: a tile with an interval the tables do not hold goes on k_pairs_missed's list; the flag is cleared before this thread
    // reaches the next tile's barrier, the others set it behind that barrier)
    if (PV && threadIdx.x == 0 && sMiss) {
      out.missList[atomicAdd(out.nMiss, 1u)] = t;
      sMiss = 0;
    }
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
}

// ---- the same merge, ONE WAVEFRONT per tile (round 6) ------------------------------------------------------------------------
// k_merge2 is bound by the chain of its round trips (1.7 ms for 3.3 GB at hg38 / 50 M + 50 M fragments): a tile is a handful of
// dependent loads, two LDS passes and two workgroup barriers for two wavefronts, eight tiles in flight per CU.  Here a wavefront
// owns a tile -- every lane two words of each bitmap, the prefix counts by DPP scans, the phases ordered by the wavefront's own
// program order (no barrier) --, five workgroups of four such wavefronts per CU, and the emission is DENSE: the set bits are
// listed by rank in LDS (a lane lists the bits of its own words: LDS stores only), then lane i takes merged interval i -- its
// position in each input = intervals before its word + set bits below it --, so that the stores are coalesced and (PV) the
// p-value look-ups of 64 intervals run side by side instead of one behind the other in a word's bit loop.
__device__ __forceinline__ void m2w_sync() {  // (LDS operations of one wavefront execute in order: keep the compiler from reordering them)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef GX_M2W_CAP
#define GX_M2W_CAP 384
#endif
#ifndef GX_M2W_HOT
#define GX_M2W_HOT 32
#endif
#ifndef GX_M2W_WGS
#define GX_M2W_WGS 6
#endif
constexpr u32 M2W_HOT = GX_M2W_HOT;  // the corner of the pair table kept in LDS (a power of two)
constexpr int M2W_WGS = GX_M2W_WGS;  // workgroups per CU at most (measured: 24 wavefronts per CU beat 20 and 28)
constexpr int M2W_NW = 4;            // wavefronts (tiles in flight) per workgroup
constexpr int M2W_CAP = GX_M2W_CAP;         // intervals per input and tile whose pileups are staged in LDS
constexpr int M2W_ROUND = 256;       // merged intervals listed per round
struct M2wLds {
  u32 bmA[MG_WORDS], bmB[MG_WORDS], bmC[MG_WORDS];
  u32 preA[MG_WORDS / 2], preC[MG_WORDS / 2];   // (u16 pairs: intervals of the input before each word)
  uint16_t offL[M2W_ROUND];
  int sA[M2W_CAP + 1], sC[M2W_CAP + 1];
};

template <bool LOOSE, bool PV>
__global__ __launch_bounds__(M2W_NW * 64) void k_merge2w(RleIn A, RleIn B, const Scalars* __restrict__ sc,
                                                         const TileMeta* __restrict__ meta, u32 nTiles, Merge2Out out,
                                                         u32* __restrict__ st, const float* __restrict__ p2d) {
  static_assert(MG_WORDS == 128, "two bitmap words per lane");
  __shared__ M2wLds LW[M2W_NW];
  __shared__ float hotP[PV ? M2W_HOT * M2W_HOT : 1];
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: the tile's header by scalar loads)
  M2wLds& L = LW[wv];
  const float factor = sc->factor, lambda = sc->lambda;
  u32 neg = 0;
  if (PV) {
    for (u32 i = threadIdx.x; i < M2W_HOT * M2W_HOT; i += M2W_NW * 64) hotP[i] = p2d[(i / M2W_HOT) * PT_N + (i % M2W_HOT)];
    __syncthreads();
  }
  const uint16_t* preA16 = reinterpret_cast<const uint16_t*>(L.preA);
  const uint16_t* preC16 = reinterpret_cast<const uint16_t*>(L.preC);
  // this lane's two words of every bitmap start empty (and are cleared again by whoever read them)
  *reinterpret_cast<uint2*>(L.bmA + 2 * lane) = make_uint2(0u, 0u);
  *reinterpret_cast<uint2*>(L.bmB + 2 * lane) = make_uint2(0u, 0u);
  *reinterpret_cast<uint2*>(L.bmC + 2 * lane) = make_uint2(0u, 0u);
  const u32 stride = gridDim.x * M2W_NW;
  // neighbouring tiles (neighbouring loose slots) on one XCD, as in k_merge2
  u32 t = xcd_local_block(blockIdx.x, gridDim.x) * M2W_NW + wv;
  MergeHdr h1{};
  if (t < nTiles) h1 = merge_hdr<LOOSE>(A, B, meta, t, nTiles);
  for (; t < nTiles; t += stride) {
    const MergeHdr h = h1;
    if (t + stride < nTiles) h1 = merge_hdr<LOOSE>(A, B, meta, t + stride, nTiles);   // (the next tile's header rides under this tile)
    const bool active = h.flags & 1u, lastTile = h.flags & 2u;
    const u32 a0 = h.a0, b0 = h.b0, pos0 = h.pos0;
    const u32 nA = h.a1c - a0, nB = h.b1c - b0;
    const bool staged = nA <= (u32)M2W_CAP && nB <= (u32)M2W_CAP;  // wave-uniform
    const bool tailElsewhere = LOOSE && !lastTile;
    auto nextB = [&](u32 j, int v) -> int {   // the control pileup BEHIND interval j (v: what lies in the next slot)
      return LOOSE && b0 + j + 1 == h.b1 && !lastTile ? h.nvB : v;
    };
    // ---- 1: breakpoints -> bitmaps; the pileups staged by interval
    for (u32 j0 = 0; j0 < max(nA, nB); j0 += 64) {
      const u32 j = j0 + lane;
      u32 eA = 0, eB = 0;
      int vA = 0, vB = 0, vBn = 0;
      const bool inA = j < nA, inB = j < nB;
      if (inA) { eA = A.end[a0 + j]; vA = A.v[a0 + j]; }
      if (inB) { eB = B.end[b0 + j]; vB = B.v[b0 + j]; vBn = B.v[b0 + j + 1]; }  // (in bounds: the slot arrays carry slack)
      if (inA) {
        const u32 off = eA - pos0;
        atomicOr(&L.bmA[off >> 5], 1u << (off & 31));
        if (staged) L.sA[j] = vA;
      }
      if (inB) {
        const u32 off = eB - pos0;
        bool ng1, ng2;
        const float here = ctrl_net(vB, factor, lambda, &ng1);
        const float next = ctrl_net(nextB(j, vBn), factor, lambda, &ng2);
        neg |= ng1 | ng2;
        atomicOr(&L.bmC[off >> 5], 1u << (off & 31));
        if (here != next) atomicOr(&L.bmB[off >> 5], 1u << (off & 31));  // 2122: net != MAX(val, lambda)
        if (staged) L.sC[j] = vB;
      }
    }
    if (active && staged && lane == 0) {  // the intervals that cover what follows the tile's last breakpoint
      L.sA[nA] = tailElsewhere ? h.nvA : A.v[h.a1c];
      L.sC[nB] = tailElsewhere ? h.nvB : B.v[h.b1c];
    }
    m2w_sync();
    // ---- 2: this lane's words, the inputs' intervals before them, the union's rank
    const uint2 wA2 = *reinterpret_cast<const uint2*>(L.bmA + 2 * lane);
    const uint2 wB2 = *reinterpret_cast<const uint2*>(L.bmB + 2 * lane);
    const uint2 wC2 = *reinterpret_cast<const uint2*>(L.bmC + 2 * lane);
    const u32 wU0 = wA2.x | wB2.x, wU1 = wA2.y | wB2.y;
    {
      const int cA0 = __popc(wA2.x), cA = cA0 + __popc(wA2.y), cC0 = __popc(wC2.x), cC = cC0 + __popc(wC2.y);
      const int inc = dpp_scan_add(cA | (cC << 16));   // (both totals <= TILE: 13 bits each)
      const u32 exA = (u32)(inc & 0xFFFF) - (u32)cA, exC = (u32)(inc >> 16) - (u32)cC;
      L.preA[lane] = exA | ((exA + (u32)cA0) << 16);
      L.preC[lane] = exC | ((exC + (u32)cC0) << 16);
    }
    const int cU = __popc(wU0) + __popc(wU1);
    const int incU = dpp_scan_add(cU);
    const u32 exU = (u32)(incU - cU), tU = (u32)__builtin_amdgcn_readlane(incU, 63);
    if (lane == 0) out.tileCount[t] = active ? tU + (lastTile ? 1u : 0u) : 0u;
    bool missAny = false;
    // one merged interval: its two pileups, or (PV) its p-value (as k_merge2's emitV)
    auto emitV = [&](u32 o, int va, int vc) {
      if constexpr (PV) {
        u32 bits = MG_P_MISS;
        if (va == V_MARK || vc == V_MARK) {
          if (va == V_MARK && vc == V_MARK) bits = __float_as_uint(GX_SKIPF);
        } else {
          const u32 ec = __umulhi((u32)va, 0x88888889u) >> 6, cc = __umulhi((u32)vc, 0x88888889u) >> 6;
          if (va >= 0 && vc >= 0 && ec < PT_N && cc < PT_N && ec * GX_UNIT == (u32)va && cc * GX_UNIT == (u32)vc) {
            bits = __hip_atomic_load(reinterpret_cast<const u32*>(&hotP[(ec & (M2W_HOT - 1)) * M2W_HOT + (cc & (M2W_HOT - 1))]), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ec >= M2W_HOT || cc >= M2W_HOT) bits = __float_as_uint(p2d[ec * PT_N + cc]);
          }
        }
        out.pBits[o] = bits;
        if (bits == MG_P_MISS) {
          out.exptV[o] = va;
          out.ctrlV[o] = vc;
          missAny = true;
        }
      } else {
        out.exptV[o] = va;
        out.ctrlV[o] = vc;
      }
    };
    if (active) {  // wave-uniform
      for (u32 r0 = 0; r0 < tU; r0 += M2W_ROUND) {
        if (r0) m2w_sync();   // (the previous round's list has been read)
        // ---- 3: the merged intervals of this round, by rank (a lane lists the set bits of its own two words)
        {
          u32 rank = exU - r0;  // (unsigned: earlier rounds' ranks wrap far beyond the round)
          for (u32 bits = wU0; bits; bits &= bits - 1, rank++)
            if (rank < (u32)M2W_ROUND) L.offL[rank] = (uint16_t)(lane * 64 + __builtin_ctz(bits));
          for (u32 bits = wU1; bits; bits &= bits - 1, rank++)
            if (rank < (u32)M2W_ROUND) L.offL[rank] = (uint16_t)(lane * 64 + 32 + __builtin_ctz(bits));
        }
        m2w_sync();
        // ---- 4: lane i takes merged interval i
        const u32 nC = min((u32)M2W_ROUND, tU - r0);
        for (u32 i0 = 0; i0 < nC; i0 += 64) {
          const u32 i = i0 + lane;
          if (i < nC) {
            const u32 off = L.offL[i], ww = off >> 5, below = (1u << (off & 31)) - 1u;
            const u32 ia = (u32)preA16[ww] + (u32)__popc(L.bmA[ww] & below), ic = (u32)preC16[ww] + (u32)__popc(L.bmC[ww] & below);
            int va, vc;
            if (staged) {
              va = L.sA[ia];
              vc = L.sC[ic];
            } else {
              const int ga = A.v[a0 + ia], gc = B.v[b0 + ic];  // (in bounds: slack)
              va = tailElsewhere && ia == nA ? h.nvA : ga;
              vc = tailElsewhere && ic == nB ? h.nvB : gc;
            }
            const u32 o = h.slot + r0 + i;
            out.end[o] = pos0 + off;
            emitV(o, va, vc);
          }
        }
      }
      if (lastTile && lane == 0) {  // 1779-1788 at the chromosome end: both pileups close at len
        const u32 oc = h.slot + tU;
        out.end[oc] = h.len;
        emitV(oc, A.v[h.a1 - 1], B.v[h.b1 - 1]);
      }
    }
    m2w_sync();   // (every lane is through with the bitmaps and the staged pileups)
    *reinterpret_cast<uint2*>(L.bmA + 2 * lane) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(L.bmB + 2 * lane) = make_uint2(0u, 0u);
    *reinterpret_cast<uint2*>(L.bmC + 2 * lane) = make_uint2(0u, 0u);
    m2w_sync();
    if (PV && __ballot(missAny) && lane == 0) out.missList[atomicAdd(out.nMiss, 1u)] = t;
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
}

// counts per tile -> offsets (sum scan), chromosome offsets and the total
__global__ __launch_bounds__(STL_NT) void k_scan_counts(const u32* __restrict__ tileCount, const u32* __restrict__ tileChrom,
                                                        const DChrom* __restrict__ chroms, u32 nTiles, u64* __restrict__ lb,
                                                        u32* __restrict__ tileOff, u32* __restrict__ chromOff,
                                                        u32* __restrict__ nOut, u32* __restrict__ st) {
  __shared__ u32 scratch[8];
  __shared__ u32 s_base;
  const u32 nChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 tb = id * STL_CHUNK + threadIdx.x * STL_ITEMS;
    u32 c[STL_ITEMS];
    u32 cs = 0;
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      c[k] = tb + k < nTiles ? tileCount[tb + k] : 0;
      cs += c[k];
    }
    u32 ctot;
    u32 cex = block_excl_scan<u32, STL_NT>(cs, scratch, &ctot);
    if (threadIdx.x < 64) {
      u64 es = lookback_excl(lb, id, (u64)ctot, st);
      if (threadIdx.x == 0) {
        s_base = (u32)es;
        if (id == nChunks - 1) {
          tileOff[nTiles] = (u32)es + ctot;
          *nOut = (u32)es + ctot;
        }
      }
    }
    __syncthreads();
    cex += s_base;
#pragma unroll
    for (int k = 0; k < STL_ITEMS; k++) {
      u32 t = tb + k;
      if (t < nTiles) {
        tileOff[t] = cex;
        if (t == chroms[tileChrom[t]].tileBase) chromOff[tileChrom[t]] = cex;
      }
      cex += c[k];
    }
    __syncthreads();
  }
}

// p-values of (treatment, control) pairs, calcPval (1628-1653), through two tables indexed by the
// exact pileups (both take few distinct values): log(treatment value), and for the control its
// clamped value with the log-normal parameters (1637-1648).  Same double-precision operations
// as the direct evaluation, so the same bits; pileups beyond the tables are computed directly.
constexpr u32 PAIR_LUT = 1u << 16;
struct CtrlEntry { double ml, sl; float net; float pad; };

__global__ __launch_bounds__(256) void k_pair_tabs(const Scalars* __restrict__ sc, double* __restrict__ logE,
                                                   CtrlEntry* __restrict__ ctab) {
  const float factor = sc->factor, lambda = sc->lambda;
  for (u32 v = blockIdx.x * 256 + threadIdx.x; v < PAIR_LUT; v += gridDim.x * 256) {
    bool ng;
    float e = getval((int)v, &ng);
    logE[v] = e > 0.0f ? log((double)e) : 0.0;
    CtrlEntry c;
    c.net = ctrl_net((int)v, factor, lambda, &ng);
    c.ml = 0;
    c.sl = 1;
    c.pad = 0;
    if (c.net > 0.0f) lnorm_params(c.net, &c.ml, &c.sl);
    ctab[v] = c;
  }
}

__device__ __forceinline__ float pval_pair(int ev, int cv, float* exptOut, float* ctrlOut, float factor, float lambda,
                                           const double* __restrict__ logE, const CtrlEntry* __restrict__ ctab, bool* neg,
                                           bool* risky) {
  bool n1 = false, n2 = false;
  const float expt = expt_val(ev, &n1);
  float ctrl;
  double ml, sl;
  if ((u32)cv < PAIR_LUT) {
    CtrlEntry c = ctab[cv];
    ctrl = c.net; ml = c.ml; sl = c.sl;
  } else {
    ctrl = ctrl_net(cv, factor, lambda, &n2);
    ml = 0; sl = 1;
    if (ctrl > 0.0f) lnorm_params(ctrl, &ml, &sl);
  }
  *neg = n1 | n2;
  *exptOut = expt;
  *ctrlOut = ctrl;
  if (ctrl == GX_SKIPF) return GX_SKIPF;
  if (ctrl == 0.0f) return expt == 0.0f ? 0.0f : FLT_MAX;
  if (expt == 0.0f) return 0.0f;
  const double le = (u32)ev < PAIR_LUT ? logE[ev] : log((double)expt);
  return pval_round(pval_double(expt, le, ml, sl), risky);
}

struct PackPairsIn {
  const u32* looseEnd;
  const int* looseE;
  const int* looseC;
  const u32* slotA;    // A.tileOff
  const u32* slotB;    // B.tileOff
  const u32* tileOff;  // tight offsets
};

// Whole pileups (no fractional part) below PT_N on both sides -- nearly every interval of an
// ordinary run -- have their p-value in a PT_N x PT_N table built once per replicate by the same
// routine; the corner PT_HOT x PT_HOT of it and the control's net values sit in LDS.

__global__ __launch_bounds__(256) void k_pair_tab2d(const Scalars* __restrict__ sc, const double* __restrict__ logE,
                                                    const CtrlEntry* __restrict__ ctab, float* __restrict__ p2d,
                                                    RiskBuf* __restrict__ risk) {
  const float factor = sc->factor, lambda = sc->lambda;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < PT_N * PT_N; i += gridDim.x * 256) {
    float e, c;
    bool ng, risky = false;
    p2d[i] = pval_pair((int)((i / PT_N) * GX_UNIT), (int)((i % PT_N) * GX_UNIT), &e, &c, factor, lambda, logE, ctab, &ng,
                       &risky);
    if (risky) risk_add(risk, RK_TAB2D, i, 0, 0, 0.0);
  }
}

// loose slots -> tight (end, expt, ctrl, p); one wavefront per tile.  Table look-ups only: a tile
// holding any other pair of values is put on a list and redone by k_pack_pairs_full (whose
// double-precision path would cost this kernel most of its occupancy).
// (KEEP / MASKS are compile-time for the same reason as in k_pack_pval)
// PONLY (round 6): the pileup floats alone, later, for somebody who asks (gx_get_intervals: -f / -k) -- the step itself writes
// (end, p) and the masks; the merge's loose arrays, the tile offsets and the control's tables are still there then.
template <bool KEEP, bool MASKS, bool PONLY = false>
__global__ __launch_bounds__(256) void k_pack_pairs(PackPairsIn in, u32 nTiles, const CtrlEntry* __restrict__ ctab,
                                                    const float* __restrict__ p2d, u32* __restrict__ end,
                                                    float* __restrict__ expt, float* __restrict__ ctrl,
                                                    float* __restrict__ p, float thr, u64* __restrict__ sigMask,
                                                    u64* __restrict__ skipMask, u32* __restrict__ heavyList,
                                                    u32* __restrict__ nHeavy) {
  __shared__ float hot[PT_HOT * PT_HOT];
  __shared__ float net[PT_N];
  for (int i = threadIdx.x; i < (int)(PT_HOT * PT_HOT); i += 256) hot[i] = p2d[(i / PT_HOT) * PT_N + (i % PT_HOT)];
  for (int i = threadIdx.x; i < (int)PT_N; i += 256) net[i] = ctab[i * GX_UNIT].net;
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const u32 stride = gridDim.x * 4;
  u32 t = blockIdx.x * 4 + wv;
  u32 src1 = 0, dst1 = 0, n1 = 0;
  if (t < nTiles) {
    src1 = in.slotA[t] + in.slotB[t];
    dst1 = in.tileOff[t];
    n1 = in.tileOff[t + 1] - dst1;
  }
  for (; t < nTiles; t += stride) {
    const u32 src = src1, dst = dst1, n = n1;
    if (t + stride < nTiles) {
      src1 = in.slotA[t + stride] + in.slotB[t + stride];
      dst1 = in.tileOff[t + stride];
      n1 = in.tileOff[t + stride + 1] - dst1;
    }
    bool miss = false;
    for (u32 b = 0; b < n; b += 256) {
      u32 e[4];
      int ev[4], cv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        e[k] = 0; ev[k] = 0; cv[k] = 0;
        if (i < n) {
          e[k] = in.looseEnd[src + i];
          ev[k] = in.looseE[src + i];
          cv[k] = in.looseC[src + i];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        float pv = 0.0f;
        if (i < n) {
          float ef = 0.0f, cf = GX_SKIPF;
          if (ev[k] == V_MARK || cv[k] == V_MARK) {
            // inside an excluded region both pileups carry the mark: treatment 0, control and p SKIP
            // (pairs with a single mark do not occur; the full kernel decides them anyway)
            if (ev[k] == V_MARK && cv[k] == V_MARK) pv = GX_SKIPF; else miss = true;
          } else {
            const u32 ec = __umulhi((u32)ev[k], 0x88888889u) >> 6, cc = __umulhi((u32)cv[k], 0x88888889u) >> 6;
            if (ev[k] >= 0 && cv[k] >= 0 && ec < PT_N && cc < PT_N && ec * GX_UNIT == (u32)ev[k] && cc * GX_UNIT == (u32)cv[k]) {
              ef = (float)ec;
              cf = net[cc];
              // (the corner in any case, the table behind a branch of its own: `? hot[..] : p2d[..]` is ONE flat load of a selected pointer)
              pv = hot[(ec & (PT_HOT - 1)) * PT_HOT + (cc & (PT_HOT - 1))];
              if (ec >= PT_HOT || cc >= PT_HOT) pv = p2d[ec * PT_N + cc];
            } else
              miss = true;
          }
          if (!PONLY) end[dst + i] = e[k];
          if (KEEP) {  // (otherwise the pileup floats are not kept, gx_set_keep_pileups)
            expt[dst + i] = ef;
            ctrl[dst + i] = cf;
          }
          if (!PONLY) p[dst + i] = pv;
        }
        if (MASKS) {  // the sweep's masks while p is at hand (pre-zeroed words; a redone tile ORs its own bits in)
          const u64 sg = __ballot(pv > thr), sk = __ballot(pv == GX_SKIPF);
          if ((sg | sk) && lane == 0) {
            const u32 pos = dst + b + k * 64, w = pos >> 6, sh = pos & 63;
            if (sg) {
              atomicOr((unsigned long long*)&sigMask[w], (unsigned long long)(sg << sh));
              if (sh && (sg >> (64 - sh))) atomicOr((unsigned long long*)&sigMask[w + 1], (unsigned long long)(sg >> (64 - sh)));
            }
            if (sk) {
              atomicOr((unsigned long long*)&skipMask[w], (unsigned long long)(sk << sh));
              if (sh && (sk >> (64 - sh))) atomicOr((unsigned long long*)&skipMask[w + 1], (unsigned long long)(sk >> (64 - sh)));
            }
          }
        }
      }
    }
    if (__ballot(miss) && lane == 0) heavyList[atomicAdd(nHeavy, 1u)] = t;
  }
}

// the listed tiles again, every pair evaluated in full (fractional pileups, very deep ones)
__global__ __launch_bounds__(256) void k_pack_pairs_full(PackPairsIn in, const u32* __restrict__ heavyList,
                                                         const u32* __restrict__ nHeavy, const Scalars* __restrict__ sc,
                                                         const double* __restrict__ logE, const CtrlEntry* __restrict__ ctab,
                                                         float* __restrict__ expt, float* __restrict__ ctrl,
                                                         float* __restrict__ p, float thr, u64* __restrict__ sigMask,
                                                         u64* __restrict__ skipMask, u32* __restrict__ st,
                                                         RiskBuf* __restrict__ risk) {
  const float factor = sc->factor, lambda = sc->lambda;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const u32 nH = *nHeavy;
  u32 neg = 0;
  for (u32 li = blockIdx.x * 4 + wv; li < nH; li += gridDim.x * 4) {
    const u32 t = heavyList[li];
    const u32 src = in.slotA[t] + in.slotB[t], dst = in.tileOff[t], n = in.tileOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      bool ng, risky = false;
      float e, c;
      const int ev = in.looseE[src + i], cv = in.looseC[src + i];
      const float pv = pval_pair(ev, cv, &e, &c, factor, lambda, logE, ctab, &ng, &risky);
      if (risky && p) risk_add(risk, RK_PAIR, dst + i, (u32)ev, (u32)cv, 0.0);
      neg |= ng;
      if (expt) {
        expt[dst + i] = e;
        ctrl[dst + i] = c;
      }
      if (!p) continue;   // (the pileup floats alone: p, its risky roundings and the masks were the step's)
      p[dst + i] = pv;
      if (sigMask) {  // the light pass saw p = 0 for the pairs it skipped: only bits to add
        if (pv > thr) atomicOr((unsigned long long*)&sigMask[(dst + i) >> 6], 1ull << ((dst + i) & 63));
        if (pv == GX_SKIPF) atomicOr((unsigned long long*)&skipMask[(dst + i) >> 6], 1ull << ((dst + i) & 63));
      }
    }
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
}

// k_merge2<.., true>'s listed tiles: the intervals it marked (MG_P_MISS) evaluated in full from the pileups it left beside them
// (fractional pileups, very deep ones) -- in the loose slots, before k_pack_ep2 moves them; one wavefront per tile.  A risky
// rounding goes on the host's list under the interval's TIGHT index (tileOff is there: k_scan_counts has run).
__global__ __launch_bounds__(256) void k_pairs_missed(PackPairsIn in, const u32* __restrict__ tileCount, const u32* __restrict__ missList,
                                                      const u32* __restrict__ nMiss, const Scalars* __restrict__ sc,
                                                      const double* __restrict__ logE, const CtrlEntry* __restrict__ ctab,
                                                      u32* __restrict__ pBits, u32* __restrict__ st, RiskBuf* __restrict__ risk) {
  const float factor = sc->factor, lambda = sc->lambda;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const u32 nH = *nMiss;
  u32 neg = 0;
  for (u32 li = blockIdx.x * 4 + wv; li < nH; li += gridDim.x * 4) {
    const u32 t = missList[li];
    const u32 src = in.slotA[t] + in.slotB[t], dst = in.tileOff[t], n = tileCount[t];
    for (u32 i = lane; i < n; i += 64) {
      if (pBits[src + i] != MG_P_MISS) continue;
      bool ng, risky = false;
      float e, c;
      const int ev = in.looseE[src + i], cv = in.looseC[src + i];
      const float pv = pval_pair(ev, cv, &e, &c, factor, lambda, logE, ctab, &ng, &risky);
      if (risky) risk_add(risk, RK_PAIR, dst + i, (u32)ev, (u32)cv, 0.0);
      neg |= ng;
      pBits[src + i] = __float_as_uint(pv);
    }
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
}

// loose (end, p) slots of k_merge2<.., true> -> the tight arrays, the sweep's masks on the way (p mode); one wavefront per tile
template <bool MASKS>
__global__ __launch_bounds__(256) void k_pack_ep2(PackPairsIn in, const float* __restrict__ looseP, u32 nTiles, u32* __restrict__ end,
                                                  float* __restrict__ p, float thr, u64* __restrict__ sigMask, u64* __restrict__ skipMask) {
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const u32 stride = gridDim.x * 4;
  u32 t = blockIdx.x * 4 + wv;
  u32 src1 = 0, dst1 = 0, n1 = 0;
  if (t < nTiles) {
    src1 = in.slotA[t] + in.slotB[t];
    dst1 = in.tileOff[t];
    n1 = in.tileOff[t + 1] - dst1;
  }
  for (; t < nTiles; t += stride) {
    const u32 src = src1, dst = dst1, n = n1;
    if (t + stride < nTiles) {
      src1 = in.slotA[t + stride] + in.slotB[t + stride];
      dst1 = in.tileOff[t + stride];
      n1 = in.tileOff[t + stride + 1] - dst1;
    }
    for (u32 b = 0; b < n; b += 256) {
      u32 e[4];
      float pv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        e[k] = 0; pv[k] = 0.0f;
        if (i < n) {
          e[k] = in.looseEnd[src + i];
          pv[k] = looseP[src + i];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 i = b + k * 64 + lane;
        if (i < n) {
          end[dst + i] = e[k];
          p[dst + i] = pv[k];
        }
        if (MASKS) {  // (pre-zeroed words)
          const u64 sg = __ballot(i < n && pv[k] > thr), sk = __ballot(i < n && pv[k] == GX_SKIPF);
          if ((sg | sk) && lane == 0) {
            const u32 pos = dst + b + k * 64, w = pos >> 6, sh = pos & 63;
            if (sg) {
              atomicOr((unsigned long long*)&sigMask[w], (unsigned long long)(sg << sh));
              if (sh && (sg >> (64 - sh))) atomicOr((unsigned long long*)&sigMask[w + 1], (unsigned long long)(sg >> (64 - sh)));
            }
            if (sk) {
              atomicOr((unsigned long long*)&skipMask[w], (unsigned long long)(sk << sh));
              if (sh && (sk >> (64 - sh))) atomicOr((unsigned long long*)&skipMask[w + 1], (unsigned long long)(sk >> (64 - sh)));
            }
          }
        }
      }
    }
  }
}

// ---- Fisher combination over replicates ---------------------------------------------------------
constexpr int MAX_REPS = 32;

struct RepIn {
  const u32* end;
  const float* p;
  const u32* tileOff;
  const uint8_t* present;  // per chromosome: this replicate has p-values there (pval[j] != NULL)
};
struct RepSet {
  RepIn r[MAX_REPS];
  int n;
};
struct MergeNOut {   // loose slots: tile t writes at sum_r tileOff_r[t]
  u32* end;
  float* p;
  u32* tileCount;
};

// Work layout of k_mergeN.  The union of the replicates' breakpoints comes from per-replicate bitmaps
// (one word per thread, ranks by block scans), as in k_merge2.  Everything per merged interval is DENSE:
//   1  the owner of a bitmap word lists its set bits (offL[rank]): no memory access in the bit loop;
//   2  thread i takes merged interval i: its position in each replicate = intervals before its word
//      (preR) + set bits below it; gathers the replicates' p, sums them in replicate order (multPval
//      570-574) and combines them on the spot: the closed form of the even-df chi-squared tail (gx_math.h
//      fisher_fast; round 6 -- rounds 2-5 kept an LDS cache of results, a device-wide table behind it and a list
//      of misses for pgamma's series).  Risky roundings (gx_math.h) go on the host's list.
// Tiles with more than MN_CAP merged intervals take several rounds.
#ifndef GX_MN_CAP
#define GX_MN_CAP 512
#endif
constexpr int MN_CAP = GX_MN_CAP;                 // merged intervals per round
__host__ __device__ constexpr size_t mergeN_lds_bytes(int n) {
  return (size_t)n * MG_WORDS * 4 + (size_t)n * MG_WORDS * 2 + MN_CAP * 2 /*offL*/ + 64;
}

__global__ __launch_bounds__(MG_NT) void k_mergeN(RepSet S, const u32* __restrict__ tileChrom,
                                                  const DChrom* __restrict__ chroms, u32 nTiles,
                                                  MergeNOut out, u32* __restrict__ st, RiskBuf* __restrict__ risk) {
  static_assert(MG_WPT == 1, "one bitmap word per thread");
  extern __shared__ __attribute__((aligned(16))) u32 dyn[];
  const int n = S.n;
  u32* bm = dyn;                                                        // n bitmaps of MG_WORDS words
  uint16_t* preR = reinterpret_cast<uint16_t*>(bm + n * MG_WORDS);      // [r][w]: intervals of r before word w
  uint16_t* offL = preR + n * MG_WORDS;                                 // offset of merged interval i of the round
  __shared__ u32 scratch[8];
  u32 bad = 0;
  for (u32 t = blockIdx.x; t < nTiles; t += gridDim.x) {  // persistent, round-robin
    __syncthreads();
    for (int i = threadIdx.x; i < n * MG_WORDS; i += MG_NT) bm[i] = 0;
    __syncthreads();
    const u32 ci = tileChrom[t];
    const DChrom c = chroms[ci];
    const u32 tl = t - c.tileBase, pos0 = tl << TB;
    const bool lastTile = tl + 1 == c.nTiles;
    bool any = false;
    u32 slot = 0;
    for (int r = 0; r < n; r++) {
      const u32 a0 = S.r[r].tileOff[t];
      slot += a0;
      if (!S.r[r].present[ci]) continue;
      any = true;
      const u32 a1 = S.r[r].tileOff[t + 1];
      const u32 a1c = (lastTile && a1 > a0) ? a1 - 1 : a1;
      for (u32 i = a0 + threadIdx.x; i < a1c; i += MG_NT) {
        const u32 off = S.r[r].end[i] - pos0;
        atomicOr(&bm[r * MG_WORDS + (off >> 5)], 1u << (off & 31));
      }
    }
    __syncthreads();
    const int w = threadIdx.x;
    u32 wU = 0;
    for (int r = 0; r < n; r++) wU |= bm[r * MG_WORDS + w];
    u32 tU;
    const u32 exU = block_excl_scan<u32, MG_NT>((u32)__popc(wU), scratch, &tU);
    for (int r = 0; r < n; r++) {  // per replicate: intervals ending before each word
      u32 tr;
      preR[r * MG_WORDS + w] = (uint16_t)block_excl_scan<u32, MG_NT>((u32)__popc(bm[r * MG_WORDS + w]), scratch, &tr);
    }
    if (threadIdx.x == 0) out.tileCount[t] = any ? tU + (lastTile ? 1u : 0u) : 0u;
    if (!any) continue;  // block-uniform
    for (u32 r0 = 0; r0 < tU; r0 += MN_CAP) {
      // ---- 1: the merged intervals of this round, by rank ----------------------------------------------------
      {
        u32 rank = exU - r0;  // (unsigned: earlier rounds' ranks wrap far beyond MN_CAP)
        for (u32 bits = wU; bits; bits &= bits - 1, rank++)
          if (rank < (u32)MN_CAP) offL[rank] = (uint16_t)(w * 32 + __builtin_ctz(bits));
      }
      __syncthreads();
      // ---- 2: gather, sum, classify ---------------------------------------------------------------------------
      const u32 nC = min((u32)MN_CAP, tU - r0);
      for (u32 i = threadIdx.x; i < nC; i += MG_NT) {
        const u32 off = offL[i], ww = off >> 5, below = (1u << (off & 31)) - 1u;
        double sum = 0.0;
        int df = 0;
        for (int r = 0; r < n; r++) {  // multPval 570-574, replicate order
          if (!S.r[r].present[ci]) continue;
          const u32 idx = S.r[r].tileOff[t] + preR[r * MG_WORDS + ww] + __popc(bm[r * MG_WORDS + ww] & below);
          const float pv = S.r[r].p[idx];
          if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
        }
        if (df > 400) bad |= ST_BAD_DF;
        const u32 o = slot + r0 + i;
        out.end[o] = pos0 + off;
        bool risky = false;
        out.p[o] = fisher_fast(sum, df, &risky);
        if (risky) risk_add(risk, RK_FISHER, t, r0 + i, (u32)df, sum);
      }
      __syncthreads();
    }
    if (lastTile && threadIdx.x == 0) {  // closing interval [.., len)
      double sum = 0.0;
      int df = 0;
      for (int r = 0; r < n; r++) {
        if (!S.r[r].present[ci]) continue;
        const float pv = S.r[r].p[S.r[r].tileOff[t + 1] - 1];
        if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
      }
      const u32 oc = slot + tU;
      out.end[oc] = c.len;
      bool risky = false;
      out.p[oc] = fisher_combine(sum, df, &risky);
      if (risky) risk_add(risk, RK_FISHER, t, tU, (u32)df, sum);
    }
  }
  if (bad) atomicOr(st, bad);
}

// ---- the same merge, ONE WAVEFRONT per tile -----------------------------------------------------------------------------
// k_mergeN spends its time waiting (SQ counters, config 5: the waves wait 82 % of their cycles, VALU active 6 %; with the
// Fisher evaluation compiled out it still takes 7.5 of its 7.8 ms): a tile is ~20 workgroup barriers and three dependent
// memory round trips, for two wavefronts, four of them per SIMD.  Here a wavefront owns a tile -- every lane two
// bitmap words per replicate, the prefix sums are DPP scans, the phases are ordered by the wavefront's own program
// order (wave_lds_sync: no barrier) -- and a CU holds five workgroups of four such wavefronts, so one tile's round
// trips run under the others' arithmetic.  Work layout per merged interval as in k_mergeN (listed by rank, gathered,
// summed in replicate order, looked up in the caches, the misses evaluated densely); the LDS cache is the wavefront's
// own (128 entries, a 16-byte write per lane: whole entries, no arbitration), behind it the same device-wide table.
constexpr int MNW_NW = 4;          // wavefronts (tiles in flight) per workgroup
#ifndef GX_MNW_CAP
#define GX_MNW_CAP 256
#endif
constexpr int MNW_CAP = GX_MNW_CAP;   // merged intervals per round
constexpr int MNW_MAXREP = 8;      // replicates this kernel takes (beyond: k_mergeN)
#ifndef GX_MNW_SCAP
#define GX_MNW_SCAP 320
#endif
constexpr int MNW_SCAP = GX_MNW_SCAP;  // intervals per replicate and tile whose p-values are staged in LDS (round 6; + the one behind them)
__host__ __device__ constexpr size_t mergeNw_wave_words(int n) {  // LDS words of one wavefront
  return (size_t)n * MG_WORDS + (size_t)n * MG_WORDS / 2 /*preR u16*/ + (size_t)n * (MNW_SCAP + 1) /*staged p*/ + MNW_CAP / 2 /*offL u16*/;
}
__host__ __device__ constexpr size_t mergeNw_lds_bytes(int n) { return MNW_NW * ((mergeNw_wave_words(n) * 4 + 15) / 16 * 16); }

__device__ __forceinline__ void mnw_sync() {  // (LDS operations of one wavefront execute in order: keep the compiler from reordering them)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Round 6: what a tile needs from global memory is asked for ONCE and early -- the next tile's chromosome record rides under
// this tile, a replicate's offsets and its "present" flag are read per tile (they were read per merged interval), and the
// replicates' p-values are staged in LDS with their ends (the gather of the dense phase was three dependent global loads per
// merged interval).
#ifndef GX_MNW_WAVES
#define GX_MNW_WAVES 6
#endif
template <int MAXR>
__global__ __launch_bounds__(MNW_NW * 64, GX_MNW_WAVES) void k_mergeN_w(RepSet S, const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                                          u32 nTiles, MergeNOut out, u32* __restrict__ st, RiskBuf* __restrict__ risk) {
  static_assert(MG_WORDS == 128, "two bitmap words per lane");
  extern __shared__ __attribute__((aligned(16))) u32 dynw[];
  const int n = S.n, lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  u32* bm = dynw + (size_t)wv * ((mergeNw_wave_words(n) * 4 + 15) / 16 * 4);   // n bitmaps of MG_WORDS words
  uint16_t* preR = reinterpret_cast<uint16_t*>(bm + n * MG_WORDS);       // [r][w]: intervals of r before word w
  float* sP = reinterpret_cast<float*>(bm + n * MG_WORDS + n * MG_WORDS / 2);   // [r][MNW_SCAP + 1]: the tile's p-values of r
  uint16_t* offL = reinterpret_cast<uint16_t*>(sP + n * (MNW_SCAP + 1)); // offset of merged interval i of the round
  u32 bad = 0;
  const u32 stride = gridDim.x * MNW_NW;
  u32 t = blockIdx.x * MNW_NW + wv;
  u32 ciN = 0;
  DChrom cN{};
  if (t < nTiles) {
    ciN = tileChrom[t];
    cN = chroms[ciN];
  }
  for (int r = 0; r < n; r++) *reinterpret_cast<uint2*>(bm + r * MG_WORDS + 2 * lane) = make_uint2(0u, 0u);
  for (; t < nTiles; t += stride) {
    mnw_sync();
    const u32 ci = ciN;
    const DChrom c = cN;
    if (t + stride < nTiles) {  // (consumed a tile later)
      ciN = tileChrom[t + stride];
      cN = chroms[ciN];
    }
    const u32 tl = t - c.tileBase, pos0 = tl << TB;
    const bool lastTile = tl + 1 == c.nTiles;
    // the replicates' slices of this tile (wave-uniform: scalar registers)
    u32 a0[MAXR], nR[MAXR], aEnd[MAXR];
    u32 pres = 0, slot = 0, maxN = 0;
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
      a0[r] = 0; nR[r] = 0; aEnd[r] = 0;
      if (r < n) {
        a0[r] = S.r[r].tileOff[t];
        aEnd[r] = S.r[r].tileOff[t + 1];
        slot += a0[r];
        if (S.r[r].present[ci]) {
          pres |= 1u << r;
          const u32 a1c = (lastTile && aEnd[r] > a0[r]) ? aEnd[r] - 1 : aEnd[r];
          nR[r] = a1c - a0[r];
          maxN = max(maxN, nR[r]);
        }
      }
    }
    const bool any = pres != 0;
    const bool staged = maxN <= (u32)MNW_SCAP;   // wave-uniform
    // ---- breakpoints -> bitmaps, p-values -> LDS: every replicate's loads of a step in flight together
    for (u32 j0 = 0; j0 < maxN; j0 += 64) {
      const u32 j = j0 + lane;
      u32 e[MAXR];
      float pv[MAXR];
#pragma unroll
      for (int r = 0; r < MAXR; r++) {
        e[r] = 0; pv[r] = 0.0f;
        if (r < n && j < nR[r]) {
          e[r] = S.r[r].end[a0[r] + j];
          pv[r] = S.r[r].p[a0[r] + j];
        }
      }
#pragma unroll
      for (int r = 0; r < MAXR; r++)
        if (r < n && j < nR[r]) {
          const u32 off = e[r] - pos0;
          atomicOr(&bm[r * MG_WORDS + (off >> 5)], 1u << (off & 31));
          if (staged) sP[r * (MNW_SCAP + 1) + j] = pv[r];
        }
    }
    // (the interval that covers what follows a replicate's last breakpoint in the tile: the next one in its tight array)
    if (staged && lane < n && ((pres >> lane) & 1u)) {
      u32 a0l = 0, nRl = 0;
#pragma unroll
      for (int r = 0; r < MAXR; r++)
        if (r == lane) { a0l = a0[r]; nRl = nR[r]; }
      sP[lane * (MNW_SCAP + 1) + nRl] = S.r[lane].p[a0l + nRl];
    }
    mnw_sync();
    u32 wU0 = 0, wU1 = 0;
    for (int r = 0; r < n; r++) {  // per replicate: intervals ending before each word; the union on the way
      const uint2 w = *reinterpret_cast<const uint2*>(bm + r * MG_WORDS + 2 * lane);
      wU0 |= w.x;
      wU1 |= w.y;
      const int c0 = __popc(w.x), cc = c0 + __popc(w.y);
      const int inc = dpp_scan_add(cc);
      const u32 ex = (u32)(inc - cc);
      *reinterpret_cast<u32*>(preR + r * MG_WORDS + 2 * lane) = ex | ((ex + (u32)c0) << 16);
    }
    const int cU = __popc(wU0) + __popc(wU1);
    const int incU = dpp_scan_add(cU);
    const u32 exU = (u32)(incU - cU), tU = (u32)__builtin_amdgcn_readlane(incU, 63);
    if (lane == 0) out.tileCount[t] = any ? tU + (lastTile ? 1u : 0u) : 0u;
    if (any) {  // wave-uniform
    for (u32 r0 = 0; r0 < tU; r0 += MNW_CAP) {
      mnw_sync();
      // ---- 1: the merged intervals of this round, by rank (a lane lists the set bits of its own two words)
      {
        u32 rank = exU - r0;  // (unsigned: earlier rounds' ranks wrap far beyond the cap)
        for (u32 bits = wU0; bits; bits &= bits - 1, rank++)
          if (rank < (u32)MNW_CAP) offL[rank] = (uint16_t)(lane * 64 + __builtin_ctz(bits));
        for (u32 bits = wU1; bits; bits &= bits - 1, rank++)
          if (rank < (u32)MNW_CAP) offL[rank] = (uint16_t)(lane * 64 + 32 + __builtin_ctz(bits));
      }
      mnw_sync();
      // ---- 2: gather, sum, combine
      const u32 nC = min((u32)MNW_CAP, tU - r0);
      for (u32 i0 = 0; i0 < nC; i0 += 64) {
        const u32 i = i0 + lane;
        double sum = 0.0;
        int df = 0;
        if (i < nC) {
          const u32 off = offL[i], ww = off >> 5, below = (1u << (off & 31)) - 1u;
#pragma unroll
          for (int r = 0; r < MAXR; r++) {  // multPval 570-574, replicate order
            if (r >= n || !((pres >> r) & 1u)) continue;   // wave-uniform
            const u32 idx = (u32)preR[r * MG_WORDS + ww] + (u32)__popc(bm[r * MG_WORDS + ww] & below);
            const float pv = staged ? sP[r * (MNW_SCAP + 1) + idx] : S.r[r].p[a0[r] + idx];
            if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
          }
          if (df > 400) bad |= ST_BAD_DF;
          const u32 o = slot + r0 + i;
          out.end[o] = pos0 + off;
          // (round 6: every interval evaluated on the spot -- the closed form of the even-df tail, gx_math.h fisher_fast -- where
          // rounds 3-5 probed an LDS cache and a device-wide table of results and evaluated the misses by pgamma's series)
          bool risky = false;
#ifdef GX_EXP_NOFISHER   // (measurement: the kernel without the combination's arithmetic)
          out.p[o] = (float)sum;
#else
          out.p[o] = fisher_fast(sum, df, &risky);
#endif
          if (risky) risk_add(risk, RK_FISHER, t, r0 + i, (u32)df, sum);
        }
      }
    }
    if (lastTile && lane == 0) {  // closing interval [.., len)
      double sum = 0.0;
      int df = 0;
#pragma unroll
      for (int r = 0; r < MAXR; r++) {
        if (r >= n || !((pres >> r) & 1u)) continue;
        const float pv = S.r[r].p[aEnd[r] - 1];
        if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
      }
      const u32 oc = slot + tU;
      out.end[oc] = c.len;
      bool risky = false;
      out.p[oc] = fisher_combine(sum, df, &risky);
      if (risky) risk_add(risk, RK_FISHER, t, tU, (u32)df, sum);
    }
    }
    mnw_sync();   // (every lane is through with the bitmaps: this lane's two words of each start the next tile empty)
    for (int r = 0; r < n; r++) *reinterpret_cast<uint2*>(bm + r * MG_WORDS + 2 * lane) = make_uint2(0u, 0u);
  }
  if (bad) atomicOr(st, bad);
}

// loose (end, p) slots of k_mergeN -> tight arrays; one wavefront per tile
__global__ __launch_bounds__(256) void k_pack_ep(RepSet S, const u32* __restrict__ looseEnd, const float* __restrict__ looseP,
                                                 const u32* __restrict__ tileOff, u32 nTiles, u32* __restrict__ end,
                                                 float* __restrict__ p) {
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 t = blockIdx.x * 4 + wv; t < nTiles; t += gridDim.x * 4) {
    u32 src = 0;
    for (int r = 0; r < S.n; r++) src += S.r[r].tileOff[t];
    const u32 dst = tileOff[t], n = tileOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      end[dst + i] = looseEnd[src + i];
      p[dst + i] = looseP[src + i];
    }
  }
}

// scalar device functions exposed for the numerics tests (gx_selftest); risky p-values go on the list
// like everywhere else, and `dbl` (optional) receives the double before its rounding to float
__global__ __launch_bounds__(256) void k_selftest(int what, const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, double* __restrict__ dbl, u32 n,
                                                  RiskBuf* __restrict__ risk) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    bool ng = false, risky = false;
    double d = 0.0;
    switch (what) {
      case 0: out[i] = log10f_host(a[i]); break;
      case 1:
        out[i] = calc_pval(a[i], b[i], &risky);
        if (dbl && a[i] > 0.0f && b[i] > 0.0f) {
          double ml, sl;
          lnorm_params(b[i], &ml, &sl);
          d = pval_double(a[i], log((double)a[i]), ml, sl);
        }
        break;
      case 2: out[i] = getval(__float_as_int(a[i]), &ng); break;
      case 3:
        out[i] = fisher_combine((double)a[i], (int)b[i], &risky);
        if (dbl && (int)b[i] > 2 && a[i] != 0.0f) d = fisher_double((double)a[i], (int)b[i]);
        break;
      case 4:  // the closed form the merge kernels use; a risky rounding is the host's, by the reference's algorithm
        out[i] = fisher_fast((double)a[i], (int)b[i], &risky);
        if (dbl && (int)b[i] > 2 && a[i] != 0.0f) d = fisher_fast_double((double)a[i], (int)b[i]);
        break;
      default: out[i] = 0.0f;
    }
    if (dbl) dbl[i] = d;
    if (risky) risk_add(risk, RK_SELF, i, (u32)what, 0, 0.0);
  }
}

// the host's values for the listed results, written where each kind lives (one workgroup: the list is short)
struct RiskTargets {
  float* lutP;
  float* p2d;
  DeepTab* deep;
  float* pairP;          // RK_PAIR: p-array of the replicate being closed
  u64* sigMask;          // ... and its significance mask (p mode), or nullptr
  float thr;
  float* fisherP;        // RK_FISHER: tight p of the combination
  const u32* fisherTileOff;
  float* selfOut;
};

__global__ __launch_bounds__(256) void k_risk_apply(RiskBuf* __restrict__ rb, const RiskRec* __restrict__ recs, u32 n,
                                                    RiskTargets T) {
  for (u32 i = threadIdx.x; i < n; i += 256) {
    const RiskRec r = recs[i];  // (device copy of a long list, or the host's mapped pinned records)
    switch (r.kind) {
      case RK_LUT:
        T.lutP[r.a] = r.pnew;
        if (r.a % GX_UNIT == 0) T.lutP[PV_LUT + r.a / GX_UNIT] = r.pnew;  // (the compact copy of the whole pileups)
        break;
      case RK_TAB2D: T.p2d[r.a] = r.pnew; break;
      case RK_DEEP: {
        const u32 j = atomicAdd(&T.deep->n, 1u);
        if (j < DEEP_TAB) { T.deep->v[j] = (int)r.a; T.deep->p[j] = r.pnew; }
        break;
      }
      case RK_PAIR:
        T.pairP[r.a] = r.pnew;
        if (T.sigMask) {
          const unsigned long long bit = 1ull << (r.a & 63);
          if (r.pnew > T.thr) atomicOr((unsigned long long*)&T.sigMask[r.a >> 6], bit);
          else atomicAnd((unsigned long long*)&T.sigMask[r.a >> 6], ~bit);
        }
        break;
      case RK_FISHER: T.fisherP[T.fisherTileOff[r.a] + r.b] = r.pnew; break;
      case RK_SELF: T.selfOut[r.a] = r.pnew; break;
      default: break;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) rb->count = 0;  // the list starts empty for the next producer
}

}  // namespace gx
