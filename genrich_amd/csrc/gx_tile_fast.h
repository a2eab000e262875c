// gx_tile_fast.h -- the tile kernel for the common case: narrow tiles (unit-weight records only, fewer
// than 32,767 per stream) of a run without -E regions.  Same contract as k_tile<false, true>
// (gx_kernels.h): LDS difference slice -> prefix sum -> run-length pileup in the tile's loose slot,
// replacing savePileupExpt's two per-base passes (Genrich.c:2197-2273).  Wide tiles and -E runs keep
// the general kernel.
//
// Why another kernel.  k_tile is VALU-issue bound (SQ counters, profiles/r02a_*: 554 VALU instructions
// per tile over its two wavefronts, VALU active on 16 % of the wave cycles x 8 waves per SIMD): every
// thread owns 32 bases and runs unrolled code for four register-held slots whatever the tile holds,
// while a tile of config 2 has ~125 touched bases for its 128 threads.  Here ONE wavefront owns a
// tile and the work is laid out by TOUCHED BASE, not by base range:
//   A  add the tile's records into the LDS slice (ds_add) and mark the occupancy bitmap (ds_or);
//   B  every lane owns two bitmap words: popcounts -> DPP scan -> ranks; the touched positions are
//      written, in position order, as a dense list of 16-bit offsets (LDS);
//   C  64 list entries per step, one per lane: read the difference, give the slot back (subtract),
//      DPP scan for the running pileup, ballot / mbcnt for the output rank, coalesced stores of
//      (end, V120).
// No barrier between wavefronts, no per-thread serial walk over a fixed number of slots, stores
// that fill whole lines; a tile costs ~220 VALU instructions.  LDS per tile: 8 KiB slice (two bases
// per word as signed 16-bit halves of one integer sum, as in k_tile<.., HALF>) + 512 B bitmap +
// 1 KiB list = 9.5 KiB -> 16 tiles in flight per CU with 16 wavefronts.
#pragma once
#include "gx_kernels.h"

namespace gx {

constexpr int TF_LCAP = 512;                                    // list entries per round (a tile rarely holds more)
constexpr int TF_WORDS = TILE / 2 + TILE / 32 + TF_LCAP / 2;    // ints of LDS per workgroup
constexpr int TF_KPL = 2;                                       // prefetched keys per lane and stream (128 per tile)

__global__ __launch_bounds__(64) void k_tile_fast(TileIn in, u32 nTiles, const u32* __restrict__ nWide, TileOut out,
                                                  u32* __restrict__ st) {
  __shared__ __attribute__((aligned(16))) int lds[TF_WORDS];
  int* delta = lds;                                                   // TILE / 2 words: two bases each
  u32* occ = reinterpret_cast<u32*>(lds + TILE / 2);                  // TILE / 32 words: one bit per base
  uint16_t* list = reinterpret_cast<uint16_t*>(lds + TILE / 2 + TILE / 32);  // TF_LCAP touched offsets, ascending
  const int lane = threadIdx.x;
  for (int i = lane * 4; i < TILE / 2 + TILE / 32; i += 64 * 4) *reinterpret_cast<int4*>(lds + i) = make_int4(0, 0, 0, 0);
  __syncthreads();
  if (*nWide > nTiles / 2) return;  // (most tiles are wide: the general kernel takes them all, as k_tile<.., HALF> has it)
  const u32 G = gridDim.x;
  const u32 lb = xcd_local_block(blockIdx.x, G);  // neighbouring tiles (neighbouring loose slots) on one XCD
  u32 bad = 0;
  // Software pipeline over this wavefront's tiles, with k_tile's discipline (loads and stores share the
  // in-order vmcnt): prefetches are issued right after a tile's stores and collected right before the
  // next tile's stores.
  //   top of tile j:  M(j), K(j), M(j+1) in registers;  K(j+1), M(j+2) in flight
  //   collect (j):    K(j+1), M(j+2) arrived
  //   issue (j):      K(j+2) (needs M(j+2)), M(j+3)
  struct Raw { uint4 a, b, c; };
  struct Keys { u32 v[TF_KPL]; };
  auto tileAt = [&](u32 i) -> u32 { return i < nTiles ? i : 0u; };
  auto loadMeta = [&](u32 tile) -> Raw {
    const uint4* q = reinterpret_cast<const uint4*>(in.meta + tile);
    return Raw{q[0], q[1], q[2]};
  };
  auto uni = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_readfirstlane((int)v); };
  auto cook = [&](const Raw& r, u32 i) -> TileMeta {
    TileMeta m;
    m.sb = uni(r.a.x); m.eb = uni(r.a.y); m.fb = uni(r.a.z); m.nS = uni(r.a.w);
    m.nE = uni(r.b.x); m.nF = uni(r.b.y); m.carry = (int)uni(r.b.z); m.ci = uni(r.b.w);
    m.pos0 = uni(r.c.x); m.len = uni(r.c.y); m.flags = uni(r.c.z); m.slot = uni(r.c.w);
    if (i >= nTiles) { m.nS = 0; m.nE = 0; m.nF = 0; m.flags = 0; }
    return m;
  };
  auto keyOf = [&](const uint16_t* K, u32 kb, u32 nk, u32 flags) -> Keys {
    Keys r;
#pragma unroll
    for (int q = 0; q < TF_KPL; q++) {
      const u32 ix = (u32)lane + q * 64;
      r.v[q] = ix < nk && !(flags & TM_WIDE) ? K[kb + ix] : 0u;
    }
    return r;
  };
  u32 tC = tileAt(lb), tN = tileAt(lb + G), tF = tileAt(lb + 2 * G), tQ = tileAt(lb + 3 * G);
  u32 tR = 0;
  TileMeta mC = cook(loadMeta(tC), lb), mN = cook(loadMeta(tN), lb + G), mR{};
  Keys ksC = keyOf(in.S, mC.sb, mC.nS, mC.flags), keC = keyOf(in.E, mC.eb, mC.nE, mC.flags), ksN{}, keN{};
  asm volatile("" : "+v"(ksC.v[0]), "+v"(ksC.v[1]), "+v"(keC.v[0]), "+v"(keC.v[1]) :: "memory");
  Keys ksL = keyOf(in.S, mN.sb, mN.nS, mN.flags), keL = keyOf(in.E, mN.eb, mN.nE, mN.flags);  // in flight
  Raw rF = loadMeta(tF);                                                                      // in flight
  for (u32 i = lb; i < nTiles; i += G) {
    const u32 t = tC;
    TileMeta m = mC;
    const bool mine = !(m.flags & TM_WIDE);  // a wide tile belongs to the general kernel: here it passes as empty
    if (!mine) { m.nS = 0; m.nE = 0; m.flags = 0; }
    const Keys ks0 = ksC, ke0 = keC;
    auto collect = [&]() {
      asm volatile("" : "+v"(ksL.v[0]), "+v"(ksL.v[1]), "+v"(keL.v[0]), "+v"(keL.v[1]), "+v"(rF.a.x), "+v"(rF.a.y),
                        "+v"(rF.a.z), "+v"(rF.a.w), "+v"(rF.b.x), "+v"(rF.b.y), "+v"(rF.b.z), "+v"(rF.b.w), "+v"(rF.c.x),
                        "+v"(rF.c.y), "+v"(rF.c.z), "+v"(rF.c.w) :: "memory");
      ksN = ksL;
      keN = keL;
      mR = cook(rF, i + 2 * G);
      tR = tF;
    };
    auto issue = [&]() {
      mC = mN; tC = tN; ksC = ksN; keC = keN;
      mN = mR; tN = tR;
      ksL = keyOf(in.S, mN.sb, mN.nS, mN.flags);  // K(j+2)
      keL = keyOf(in.E, mN.eb, mN.nE, mN.flags);
      rF = loadMeta(tQ);                          // M(j+3)
      tF = tQ;
      tQ = tileAt(i + 4 * G);
    };
    const bool active = m.flags & TM_ACTIVE;
    const bool lastTile = (m.flags & TM_LAST) != 0;
    const u32 pos0 = m.pos0, slot = m.slot;
    const u32 nS = m.nS, nE = m.nE;
    // ---- A: records -> LDS slice + occupancy bitmap ------------------------------------------------------------
    auto add = [&](u32 off, int sign) {
      atomicAdd(&delta[off >> 1], (off & 1) ? sign * 65536 : sign);
      atomicOr(&occ[off >> 5], 1u << (off & 31));
    };
#pragma unroll
    for (int q = 0; q < TF_KPL; q++) {
      if ((u32)lane + q * 64 < nS) add(ks0.v[q], 1);
      if ((u32)lane + q * 64 < nE) add(ke0.v[q], -1);
    }
    for (u32 k = TF_KPL * 64 + lane; k < nS; k += 64) add(in.S[m.sb + k], 1);
    for (u32 k = TF_KPL * 64 + lane; k < nE; k += 64) add(in.E[m.eb + k], -1);
    __syncthreads();
    // ---- B: bitmap -> ranks -> dense list of touched offsets ---------------------------------------------------
    u32 w0 = 0, w1 = 0;
    if (nS + nE) {  // wave-uniform
      const uint2 ww = *reinterpret_cast<const uint2*>(occ + 2 * lane);
      w0 = ww.x;
      w1 = ww.y;
      *reinterpret_cast<uint2*>(occ + 2 * lane) = make_uint2(0u, 0u);  // own words: the next tile finds them clear
    }
    const int c = __popc(w0) + __popc(w1);
    const int incC = dpp_scan_add(c);
    const u32 exc = (u32)(incC - c);
    const u32 T = (u32)__builtin_amdgcn_readlane(incC, 63);  // touched bases of the tile
    collect();
    // ---- C: 64 touched bases per step --------------------------------------------------------------------------
    int runBase = m.carry;       // pileup (1/120 units) before the first base not yet processed: wave-uniform
    u32 outCount = 0, lastEnd = 0;
    u32 neg = 0, big = (u32)(m.carry >= FRAG_FAST_MAXV);
    for (u32 r0 = 0; r0 < T; r0 += TF_LCAP) {
      {
        u32 rank = exc - r0;  // (unsigned: entries of earlier rounds wrap far beyond TF_LCAP)
        for (u32 b = w0; b; b &= b - 1, rank++)
          if (rank < (u32)TF_LCAP) list[rank] = (uint16_t)(lane * 64 + __builtin_ctz(b));
        for (u32 b = w1; b; b &= b - 1, rank++)
          if (rank < (u32)TF_LCAP) list[rank] = (uint16_t)(lane * 64 + 32 + __builtin_ctz(b));
      }
      __syncthreads();
      const u32 nL = min((u32)TF_LCAP, T - r0);
      for (u32 j0 = 0; j0 < nL; j0 += 64) {
        const u32 j = j0 + lane;
        const bool valid = j < nL;
        u32 p = 0;
        int d = 0;
        if (valid) {
          p = list[j];
          const int w = delta[p >> 1];
          const int lo = (int)(short)w;
          d = (p & 1) ? (w - lo) >> 16 : lo;                              // this base's net count of records
          if (d != 0) atomicAdd(&delta[p >> 1], (p & 1) ? -(d * 65536) : -d);  // its share of the word goes back to zero
        }
        const int d120 = d * GX_UNIT;
        const int incS = dpp_scan_add(d120);
        const int before = runBase + incS - d120;   // the pileup of the interval that ends at this base (2244)
        const int after = before + d120;
        const bool nz = valid && d != 0 && active && (pos0 + p != 0);  // 2241: base 0 closes nothing
        const u64 mask = __ballot(nz);
        if (nz) {
          const u32 o = slot + outCount +
                        __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
          out.looseEnd[o] = pos0 + p;
          out.looseV[o] = before;
        }
        neg |= (u32)(after < 0);
        big |= (u32)(after >= FRAG_FAST_MAXV);
        runBase += __builtin_amdgcn_readlane(incS, 63);
        if (mask) {  // wave-uniform
          outCount += (u32)__popcll(mask);
          lastEnd = pos0 + (u32)__builtin_amdgcn_readlane((int)p, 63 - __builtin_clzll(mask));
        }
      }
      __syncthreads();  // the list is rewritten by the next round
    }
    u32 total = 0;
    if (active) {  // wave-uniform
      total = outCount + (lastTile ? 1u : 0u);
      if (lane == 0) {
        if (lastTile) {  // closing interval [.., len): 2268-2273
          out.looseEnd[slot + outCount] = m.len;
          out.looseV[slot + outCount] = runBase;
          lastEnd = m.len;
        }
        if (total) out.tileLastEnd[t] = lastEnd;
      }
      if (__ballot(neg != 0)) bad |= ST_NEG_PILE;
      if (__ballot(big != 0) && lane == 0) atomicOr(&out.tileDeep[t], 1u);  // rare
    }
    if (lane == 0) out.tileCount[t] = total;  // (0 for a wide tile: overwritten by the kernel that owns it)
    issue();
  }
  if (bad && lane == 0) atomicOr(st, bad);
}

}  // namespace gx
