// gx_tile_fast.h -- the tile kernel of every run without -E regions: unit-weight starts / ends (2-byte
// offsets) and fractional-weight records of multimapped reads (8 bytes, signed weight in 1/120 units) alike.
// Same contract as k_tile (gx_kernels.h): LDS difference slice -> prefix sum -> run-length pileup in the
// tile's loose slot, replacing savePileupExpt's two per-base passes (Genrich.c:2197-2273).  -E runs keep
// the general kernel.
//
// Why another kernel.  k_tile is VALU-issue bound (SQ counters, profiles/r02a_*: 554 VALU instructions
// per tile over its two wavefronts): every thread owns 32 bases and runs unrolled code for four
// register-held slots whatever the tile holds, while a tile of config 2 has ~125 touched bases for its
// 128 threads.  Here ONE wavefront owns a tile, the work is laid out by TOUCHED BASE, and the LDS
// counters are addressed by the touched base's RANK, not by its offset:
//   A1 mark the occupancy bitmap with the tile's records (ds_or) -- nothing else;
//   B  every lane owns two bitmap words: popcounts -> DPP scan -> the number of touched bases before
//      each word (`pre`, 16 bits per word);
//   A2 every record finds its base's rank, pre[word] + popc(word & bits below), adds its sign to
//      cnt[rank] (32-bit sums in 1/120 units: +-120 for a start / an end, the record's weight for a
//      fractional one) and writes its offset to list[rank] (records of one base write the same value): the
//      position-ordered list of touched bases falls out without any per-lane bit loop;
//   C  64 list entries per step, one per lane: the difference is cnt[j] (cleared on the way), DPP scan
//      for the running pileup, ballot / mbcnt for the output rank, coalesced stores of (end, V120).
// No barrier between wavefronts, no per-thread serial walk, stores that fill whole lines: ~190 VALU
// instructions per tile.  LDS per tile: 512 B bitmap + 256 B pre + 2 KiB cnt + 1 KiB list = 3.75 KiB.
// (Round 2's first version kept an 8 KiB slice indexed by offset -- two bases per word -- and built the
// list with per-lane bit loops: 320 VALU instructions per tile, VALU busy ~80 % of the SIMD cycles, 16
// tiles per CU: 0.62 ms for config 2's sample where this one takes 0.47 ms.)
// Workgroups per CU (measured, config 2): 16: 0.59, 20: 0.53, 22: 0.51, 24: 0.47, 26: 0.59, 28: 0.57,
// 32: 0.51 ms -> 24 (TF_WG_PER_CU).
// A tile with more than TR_CAP touched bases takes ceil(T / TR_CAP) rounds of A2 + C.
#pragma once
#include "gx_kernels.h"

namespace gx {

constexpr int TF_KPL = 2;                                       // prefetched keys per lane and stream (128 per tile)
constexpr int TF_FPL = 2;                                       // fractional records per lane kept in registers (128 per tile)
constexpr int TF_WG_PER_CU = 24;                                // one-wavefront workgroups per CU (see above)
constexpr int TR_CAP = 512;
constexpr int TR_WORDS = TILE / 32 + TILE / 64 + TR_CAP + TR_CAP / 2;

static_assert(TILE == 64 * 2 * 32, "k_tile_fast / k_tile_heavy: two 32-bit bitmap words per lane, 64 prefix entries (GX_TB = 12)");
__global__ __launch_bounds__(64) void k_tile_fast(TileIn in, u32 nTiles, const u32* __restrict__ nWide, TileOut out,
                                                  u32* __restrict__ st) {
  __shared__ __attribute__((aligned(16))) int lds[TR_WORDS];
  u32* occ = reinterpret_cast<u32*>(lds);                               // TILE / 32 words: one bit per base
  u32* pre = reinterpret_cast<u32*>(lds + TILE / 32);                   // [lane]: touched bases before word 2 lane | word 2 lane + 1 << 16
  const uint16_t* pre16 = reinterpret_cast<const uint16_t*>(pre);       // [word]
  int* cnt = lds + TILE / 32 + TILE / 64;                               // TR_CAP net record counts, by rank
  uint16_t* list = reinterpret_cast<uint16_t*>(lds + TILE / 32 + TILE / 64 + TR_CAP);  // TR_CAP offsets, by rank
  const int lane = threadIdx.x;
  for (int i = lane * 4; i < TR_WORDS; i += 64 * 4) *reinterpret_cast<int4*>(lds + i) = make_int4(0, 0, 0, 0);
  __syncthreads();
  const u32 G = gridDim.x;
  const u32 lb = xcd_local_block(blockIdx.x, G);
  u32 bad = 0;
  // the general fragLen path (fractional weights): this kernel adds the intervals' exact terms itself (TileIn::fragAcc)
  const bool fragTerms = in.fragAcc != nullptr && (u32)__builtin_amdgcn_readfirstlane((int)in.ff->slow) != 0u;
  long long fhi = 0, flo = 0;
  // pileups from which an interval is significant, when lambda was known before this kernel (LooseCtl)
  __shared__ u32 vsRed[2];
  const int vsig = (int)__builtin_amdgcn_readfirstlane(loose_vsig(out.ctl, blockIdx.x == 0 && lane == 0, vsRed));
  // Software pipeline over this wavefront's tiles, with k_tile's discipline (loads and stores share the
  // in-order vmcnt): prefetches are issued right after a tile's stores and collected right before the
  // next tile's stores.
  //   top of tile j:  M(j), K(j), M(j+1) in registers;  K(j+1), M(j+2) in flight
  //   collect (j):    K(j+1), M(j+2) arrived
  //   issue (j):      K(j+2) (needs M(j+2)), M(j+3)
  // (descriptors through scalar loads -- the constant address space -- were tried: the compiler turns
  // those whose fields feed vector address arithmetic back into vector loads and waits for them on the spot)
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
    m.nE = uni(r.b.x); m.nF = uni(r.b.y); m.carry = (int)uni(r.b.z); m.ci = 0;
    m.pos0 = uni(r.c.x); m.len = uni(r.c.y); m.flags = uni(r.c.z); m.slot = uni(r.c.w);
    if (i >= nTiles) { m.nS = 0; m.nE = 0; m.nF = 0; m.flags = 0; }
    // (a heavy tile is k_tile_heavy's: here it looks empty and inactive -- its count, written as zero, is overwritten there)
    if (m.flags & TM_HEAVY) { m.nS = 0; m.nE = 0; m.nF = 0; m.flags &= ~TM_ACTIVE; }
    return m;
  };
  auto keyOf = [&](const uint16_t* K, u32 kb, u32 nk, u32 flags) -> Keys {
    Keys r;
#pragma unroll
    for (int q = 0; q < TF_KPL; q++) {
      const u32 ix = (u32)lane + q * 64;
      r.v[q] = ix < nk ? K[kb + ix] : 0u;
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
    const Keys ks0 = ksC, ke0 = keC;
    auto collect = [&]() {
      asm volatile("" : "+v"(ksL.v[0]), "+v"(ksL.v[1]), "+v"(keL.v[0]), "+v"(keL.v[1]), "+v"(rF.a.x), "+v"(rF.a.y),
                        "+v"(rF.a.z), "+v"(rF.a.w), "+v"(rF.b.x), "+v"(rF.b.y), "+v"(rF.b.z), "+v"(rF.c.x),
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
    const u32 nS = m.nS, nE = m.nE, nF = m.nF;
    // the tile's fractional records (multimapped reads; none in most runs): the first TF_FPL per lane stay in
    // registers for both passes
    u64 fr[TF_FPL];
    if (nF) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < TF_FPL; q++) fr[q] = (u32)lane + q * 64 < nF ? in.F[m.fb + lane + q * 64] : 0ull;
    }
    auto fOff = [](u64 r) -> u32 { return (u32)(r >> 8) & (TILE - 1); };
    auto fW = [](u64 r) -> int { return (int)(int8_t)(r & 0xFF); };
    // ---- A1: records -> occupancy bitmap -----------------------------------------------------------------------
    auto mark = [&](u32 off) { atomicOr(&occ[off >> 5], 1u << (off & 31)); };
#pragma unroll
    for (int q = 0; q < TF_KPL; q++) {
      if ((u32)lane + q * 64 < nS) mark(ks0.v[q]);
      if ((u32)lane + q * 64 < nE) mark(ke0.v[q]);
    }
    for (u32 k = TF_KPL * 64 + lane; k < nS; k += 64) mark(in.S[m.sb + k]);
    for (u32 k = TF_KPL * 64 + lane; k < nE; k += 64) mark(in.E[m.eb + k]);
    if (nF) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < TF_FPL; q++)
        if ((u32)lane + q * 64 < nF) mark(fOff(fr[q]));
      for (u32 k = TF_FPL * 64 + lane; k < nF; k += 64) mark(fOff(in.F[m.fb + k]));
    }
    __syncthreads();
    // ---- B: bitmap -> touched bases before each word -----------------------------------------------------------
    u32 w0 = 0, w1 = 0;
    if (nS + nE + nF) {  // wave-uniform
      const uint2 ww = *reinterpret_cast<const uint2*>(occ + 2 * lane);
      w0 = ww.x;
      w1 = ww.y;
    }
    const int c0 = __popc(w0), c = c0 + __popc(w1);
    const int incC = dpp_scan_add(c);
    const u32 exc = (u32)(incC - c);
    const u32 T = (u32)__builtin_amdgcn_readlane(incC, 63);  // touched bases of the tile
    if (T) pre[lane] = exc | ((exc + (u32)c0) << 16);
    collect();
    int runBase = m.carry;       // pileup (1/120 units) before the first base not yet processed: wave-uniform
    u32 outCount = 0, lastEnd = 0;
    u64 negM = 0, bigM = m.carry >= FRAG_FAST_MAXV ? ~0ull : 0ull;
    // ---- A2: records -> cnt[rank], list[rank] ------------------------------------------------------------------
    // rank of a record's base = touched bases before its bitmap word + set bits below it in the word.  The
    // ranks of the register-held records are computed once, all LDS reads in one batch (lanes without a
    // record look at offset 0).
    auto rankOf = [&](u32 off) -> u32 {
      const u32 wi = off >> 5;
      return (u32)pre16[wi] + (u32)__popc(occ[wi] & ((1u << (off & 31)) - 1u));
    };
    u32 rs[TF_KPL], re[TF_KPL];
    if (T) {  // wave-uniform
      __syncthreads();  // `pre` is there
#pragma unroll
      for (int q = 0; q < TF_KPL; q++) {
        rs[q] = rankOf(ks0.v[q]);
        re[q] = rankOf(ke0.v[q]);
      }
    }
    for (u32 r0 = 0; r0 < T; r0 += TR_CAP) {
      if (r0) __syncthreads();  // the previous round is through with cnt and list
      auto put = [&](u32 r, u32 off, int w) {  // w: the record's weight in 1/120 units
        r -= r0;
        if (r < (u32)TR_CAP) {
          list[r] = (uint16_t)off;
          atomicAdd(&cnt[r], w);
        }
      };
#pragma unroll
      for (int q = 0; q < TF_KPL; q++) {
        if ((u32)lane + q * 64 < nS) put(rs[q], ks0.v[q], GX_UNIT);
        if ((u32)lane + q * 64 < nE) put(re[q], ke0.v[q], -GX_UNIT);
      }
      for (u32 k = TF_KPL * 64 + lane; k < nS; k += 64) { const u32 off = in.S[m.sb + k]; put(rankOf(off), off, GX_UNIT); }
      for (u32 k = TF_KPL * 64 + lane; k < nE; k += 64) { const u32 off = in.E[m.eb + k]; put(rankOf(off), off, -GX_UNIT); }
      if (nF) {  // wave-uniform
#pragma unroll
        for (int q = 0; q < TF_FPL; q++)
          if ((u32)lane + q * 64 < nF) put(rankOf(fOff(fr[q])), fOff(fr[q]), fW(fr[q]));
        for (u32 k = TF_FPL * 64 + lane; k < nF; k += 64) {
          const u64 r = in.F[m.fb + k];
          put(rankOf(fOff(r)), fOff(r), fW(r));
        }
      }
      __syncthreads();
      // ---- C: 64 touched bases per step ------------------------------------------------------------------------
      const u32 nL = min((u32)TR_CAP, T - r0);
      for (u32 j0 = 0; j0 < nL; j0 += 64) {
        const u32 j = j0 + lane;
        const bool valid = j < nL;
        u32 p = 0;
        int d120 = 0;
        if (valid) {
          p = list[j];
          d120 = cnt[j];       // this base's net weight (1/120 units)
          cnt[j] = 0;          // the next round / tile finds it clear
        }
        const int incS = dpp_scan_add(d120);
        const int after = runBase + incS;
        const int before = after - d120;            // the pileup of the interval that ends at this base (2244)
        const bool nz = d120 != 0 && active && (pos0 + p != 0);  // 2241: base 0 closes nothing
        const u64 mask = __ballot(nz);
        const u32 orank = __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
        if (nz) {
          const u32 o = slot + outCount + orank;
          out.looseEnd[o] = pos0 + p;
          out.looseV[o] = before;
        }
        if (vsig != 0x7FFFFFFF) sig_flush(out.sigMask, slot + outCount, nz && before >= vsig, orank, mask);  // wave-uniform
        negM |= __ballot(after < 0);
        bigM |= __ballot(after >= FRAG_FAST_MAXV);
        runBase += __builtin_amdgcn_readlane(incS, 63);
        if (fragTerms && mask) {  // wave-uniform
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
      }
    }
    if (T) *reinterpret_cast<uint2*>(occ + 2 * lane) = make_uint2(0u, 0u);  // own words: the next tile finds them clear
    u32 total = 0;
    if (active) {  // wave-uniform
      total = outCount + (lastTile ? 1u : 0u);
      if (lane == 0) {
        if (lastTile) {  // closing interval [.., len): 2268-2273
          const u32 o = slot + outCount;
          out.looseEnd[o] = m.len;
          out.looseV[o] = runBase;
          if (runBase >= vsig) atomicOr((unsigned long long*)&out.sigMask[o >> 6], 1ull << (o & 63));
          if (fragTerms && outCount) frag_term(m.len - lastEnd, runBase, fhi, flo);  // (the tile's first interval: k_scan_iv)
          lastEnd = m.len;
        }
        if (total) out.tileLastEnd[t] = lastEnd;
      }
      lastEnd = (u32)__builtin_amdgcn_readfirstlane((int)lastEnd);
      if (negM) bad |= ST_NEG_PILE;
      if (bigM && lane == 0) {  // rare
        atomicOr(&out.tileDeep[t], 1u);
        if (out.ctl) atomicOr(&out.ctl->bad, 1u);  // a pileup beyond (or close to the end of) the table p(V)
      }
    }
    if (lane == 0) out.tileCount[t] = total;
    // the tile's unused slots: zero-length intervals behind its last one, for the sweep on the loose slots (a
    // tile without intervals does not know where the previous one ended: k_scan_iv fills its slots)
    if (total && vsig != 0x7FFFFFFF) {  // wave-uniform
      const u32 size = nS + nE + nF + 1;
      for (u32 j = total + lane; j < size; j += 64) {
        out.looseEnd[slot + j] = lastEnd;
        out.looseV[slot + j] = 0;
      }
    }
    issue();
  }
  if (fragTerms) {  // wave-uniform
    fhi = wave_sum(fhi);
    flo = wave_sum(flo);
    if (lane == 0) {
      if (fhi) atomicAdd((u64*)&in.fragAcc[0], (u64)fhi);
      if (flo) atomicAdd((u64*)&in.fragAcc[1], (u64)flo);
    }
  }
  if (bad && lane == 0) atomicOr(st, bad);
}

}  // namespace gx
