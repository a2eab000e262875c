// gx_stats.h -- p-values, Benjamini-Hochberg q-values and the peak sweep on run-length
// intervals (gfx950).  Interval arrays are laid out chromosome after chromosome; chromOff
// [nChrom+1] gives each chromosome's range, so an interval's start is the previous end or 0.
#pragma once
#include "gx_kernels.h"

namespace gx {

// ---- p-values, no control: savePileupNoCtrl + savePval (Genrich.c:1883-1896, 1720-1794) ----
// The control pileup is the constant lambda, so the p-intervals are the treatment intervals.
// With a constant control the p-value is a function of the exact pileup V alone, so it is
// tabulated once per replicate for V < PV_LUT (pileups up to ~2184x; a 1 MiB table that stays in
// L2) by the very same double-precision routine, and looked up per interval; larger V are
// computed directly.  Same bits either way.
// (PV_LUT = 2^18 entries: gx_kernels.h)

__device__ __forceinline__ float pval_of_v(int v, float lambda, double ml, double sl, float* valOut, bool* neg, bool* risky) {
  float val = getval(v, neg);
  *valOut = val;
  if (lambda == 0.0f) return val == 0.0f ? 0.0f : FLT_MAX;  // calcPval 1631-1632
  return val == 0.0f ? 0.0f : pval_given(val, ml, sl, risky);
}

__device__ __forceinline__ void deep_risky_body(PackIn in, const FragFix* __restrict__ ff, const u32* __restrict__ list,
                                                const Scalars* __restrict__ sc, RiskBuf* __restrict__ risk, u32 block, u32 nBlocks);

// one entry of the table p(V); with `ctl` also the workgroup's slot of LooseCtl (from which pileup on an interval is
// significant).  Call with all 256 threads of the (possibly virtual: a quarter of k_bins_lut's) workgroup `wg`: the slot
// reduction contains barriers.
__device__ __forceinline__ void lut_entry(u32 v, float lambda, double ml, double sl, float* __restrict__ lutP,
                                          RiskBuf* __restrict__ risk, LooseCtl* __restrict__ ctl, float thr, u32* red, u32 tid,
                                          u32 wg) {
  float val;
  bool ng, risky = false;
  const float p = pval_of_v((int)v, lambda, ml, sl, &val, &ng, &risky);
  lutP[v] = p;
  if (v % GX_UNIT == 0) lutP[PV_LUT + v / GX_UNIT] = p;  // the whole pileups once more, compact (the sweep's LDS copy)
  if (risky) risk_add(risk, RK_LUT, v, 0, 0, 0.0);
  if (ctl) {
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    const u64 sg = __ballot(p > thr);
    if (lane_id() == 0) {
      const u32 v0 = v;
      if (sg) atomicMax(&red[0], PV_LUT - (v0 + (u32)__builtin_ctzll(sg)));
      if (~sg) atomicMax(&red[1], v0 + (u32)(63 - __builtin_clzll(~sg)) + 1u);
    }
    __syncthreads();
    if (tid == 0) {
      ctl->sigInv[wg] = red[0];
      ctl->nonP1[wg] = red[1];
    }
  }
}

// blocks [0, PV_LUT / 256): the table; DEEP_BLOCKS more: the deep tiles' risky values (k_deep_risky's work, same launch)
constexpr u32 DEEP_BLOCKS = 32;
// `early` (LooseCtl, gx_kernels.h): the launch ahead of the tile stage, with the lambda of the closed form of fragLen --
// it also finds from which pileup on an interval is significant (p > thr; the tile kernels write the sweep's bits
// with it).  The launch after the tile stage then only rebuilds the table when lambda turned out different.
__global__ __launch_bounds__(256) void k_pval_lut(const Scalars* __restrict__ sc, float* __restrict__ lutP,
                                                  RiskBuf* __restrict__ risk, DeepTab* __restrict__ deep, PackIn in,
                                                  const FragFix* __restrict__ ff, const u32* __restrict__ list,
                                                  LooseCtl* __restrict__ ctl, int early, float thr) {
  if (blockIdx.x == 0 && threadIdx.x == 0) deep->n = 0;  // this sample's host-evaluated deep values come later
  if (blockIdx.x >= PV_LUT / 256) {
    if (in.meta) deep_risky_body(in, ff, list, sc, risk, blockIdx.x - PV_LUT / 256, gridDim.x - PV_LUT / 256);
    return;
  }
  const float lambda = sc->lambda;
  // (early == 2, round 6: the launch after the tile stage of a sample whose lambda only came with its end -- fractional weights,
  // several ranks without the early all-reduce --, which leaves LooseCtl's slots as well: k_loose_late writes the sweep's bits from them)
  if (ctl) {
    if (early == 1 && !ctl->enabled) return;                                                // lambda is not known yet
    if (early != 1 && ctl->enabled && ctl->earlyBits == __float_as_uint(lambda)) return;   // the table is this one already
  }
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  __shared__ u32 red[2];
  for (u32 v = blockIdx.x * 256 + threadIdx.x; v < PV_LUT; v += (PV_LUT / 256) * 256)
    // (the grid covers the table once: v = this wavefront's first entry + lane, one slot per workgroup)
    lut_entry(v, lambda, ml, sl, lutP, risk, early && ctl ? ctl : nullptr, thr, red, threadIdx.x, blockIdx.x);
}

// The sweep on the loose slots for a sample whose lambda was NOT known before the tile stage (round 6: fractional weights take the
// closed form of fragLen away; callPeaks 977-1069 on the intervals where savePileupExpt 2197-2273 left them).  The tile stage wrote
// neither the significance bits nor the zero-length fillers of a tile's unused slots; with the table p(V) and LooseCtl's slots
// there (k_pval_lut, early == 2) one wavefront per tile does both -- a pass over the pileups (4 bytes per interval) instead of
// k_pack_pval's copy of everything into the tight table (16 bytes per interval).  The verdict (LooseCtl::ok: no pileup beyond the table,
// no tile without intervals too long to fill, p(V) > thr a threshold on V) is k_loose_verdict's, with the sample's scalars; the pass
// itself runs when gx_find_peaks finds this replicate to be the run's only one (a further replicate makes the tight tables anyway).
__global__ __launch_bounds__(256) void k_loose_verdict(LooseCtl* __restrict__ ctl) {
  __shared__ u32 red[2];
  const int vsig = loose_vsig(ctl, threadIdx.x == 0, red, true);
  if (threadIdx.x == 0) ctl->ok = vsig != 0x7FFFFFFF && ld_agent(&ctl->bad) == 0u ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_loose_late(const u32* __restrict__ tileSlot, const u32* __restrict__ tileCount,
                                                    const u32* __restrict__ tileLastEnd, u32 nTiles, u32* __restrict__ looseEnd,
                                                    int* __restrict__ looseV, LooseCtl* __restrict__ ctl, u64* __restrict__ sigMask,
                                                    const u32* __restrict__ vq = nullptr, u32* __restrict__ st = nullptr) {
  __shared__ u32 red[2];
  int vsig;
  if (vq) {
    // -q (k_bh_small): vq[0] = the smallest pileup present whose q passes, vq[1] = 1 + the largest present whose q does not
    const u32 v0 = vq[0], v1 = vq[1];
    if (v1 > v0) {  // q is no threshold on the pileup (a table p(V) that is not monotone): the host takes the tight table
      if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(st, ST_Q_LOOSE);
      return;
    }
    vsig = v0 > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)v0;   // (no pileup passes: no bit, the fillers all the same)
  } else {
    vsig = (int)__builtin_amdgcn_readfirstlane(loose_vsig(ctl, false, red, true));
    if (vsig == 0x7FFFFFFF) return;  // (k_loose_verdict said so: not reached)
  }
  const int lane = lane_id();
  // (the fillers of a tile's unused slots are the tile stage's: it writes them whenever LooseCtl is handed to it.  A tile is a few
  // hundred slots: what a wavefront waits for is the chain header -> pileups, so two tiles ahead the header is asked for, one tile
  // ahead the first 256 pileups -- k_pack_pval's pipeline)
  const u32 stride = gridDim.x * 4;
  u32 t = blockIdx.x * 4 + (threadIdx.x >> 6);
  struct Hdr { u32 s0, cnt; };
  auto loadHdr = [&](u32 tt) { return tt < nTiles ? Hdr{tileSlot[tt], tileCount[tt]} : Hdr{0u, 0u}; };
  auto loadV = [&](const Hdr& h, int (&v)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 j = (u32)k * 64 + (u32)lane;
      v[k] = j < h.cnt ? looseV[h.s0 + j] : (int)0x80000000;
    }
  };
  auto bits = [&](u32 base, const int (&v)[4], u32 left) {   // (left: intervals from `base` on)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if ((u32)k * 64 >= left) break;  // wave-uniform
      const u64 m = __ballot((u32)k * 64 + (u32)lane < left && v[k] >= vsig);
      if (m && lane == 0) {
        const u32 b = base + (u32)k * 64, sh = b & 63u;
        atomicOr((unsigned long long*)&sigMask[b >> 6], m << sh);
        if (sh && (m >> (64u - sh))) atomicOr((unsigned long long*)&sigMask[(b >> 6) + 1], m >> (64u - sh));
      }
    }
  };
  Hdr h1 = loadHdr(t), h2 = loadHdr(t + stride);
  int v1[4];
  loadV(h1, v1);
  for (; t < nTiles; t += stride) {
    const Hdr h = h1;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = v1[k];
    h1 = h2;
    loadV(h1, v1);
    h2 = loadHdr(t + 2 * stride);
    bits(h.s0, v, h.cnt);
    for (u32 j0 = 256; j0 < h.cnt; j0 += 256) {   // (a tile with a peak)
      int w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 j = j0 + (u32)k * 64 + (u32)lane;
        w[k] = j < h.cnt ? looseV[h.s0 + j] : (int)0x80000000;
      }
      bits(h.s0 + j0, w, h.cnt - j0);
    }
  }
}

// p-values against a constant control (no control sample), straight from the tile kernel's
// loose slots: one wavefront per tile moves the tile's intervals to their tight position and
// scores them on the way, so the packed (end, V) pair never makes a round trip through HBM.
// Latency, not bandwidth, bounds this shape (a few hundred intervals per tile): every lane keeps
// PP_UNROLL independent (end, V) loads in flight, the next tile's header is fetched while the
// current one is processed, and the hot head of the p-value table lives in LDS (a 64-lane gather
// from L1 costs one cache line per lane).
constexpr int PP_UNROLL = 4;   // 256 intervals of a tile in flight per wavefront
constexpr int PP_HOT = 1024;   // whole pileups below this have their p-value in LDS

// The pileup floats of the intervals (the reference's Pileup.cov, printed by -f / -k only) are NOT written
// here: they are a function of the exact V that stays in the loose slots, and k_piles_from_loose makes them
// when somebody asks (gx_get_intervals, a further sample about to reuse the slots).
// MASKS: whether the sweep masks are written.  Compile-time on purpose: with the same choice as a run-time
// null check on the output pointers the compiler schedules the stores of the hot loop 35 % slower (measured).
// HIST (round 6): the genome-wide "bp at pileup V" histogram on the way -- hashPval (Genrich.c:300-327) for a replicate that turns
// out to be the run's only one, without a control, with -q: p is a function of V, so Benjamini-Hochberg's {p -> bp} table is made
// of ~100 sums (k_bh_from_dense) instead of a hash insertion per interval read back from the tight table (k_bh_hist: 0.25 ms at
// hg38 / 50 M fragments).  Whole pileups inside the table go to an LDS histogram (two copies, by lane parity; 32-bit: a workgroup's
// share of a genome is far below 2^32 bases), fractional ones to `dense` directly; the pileups beyond the table are k_deep_hist's.
// TIGHT = false (round 6, with HIST): the histogram alone -- -q on a replicate whose sweep walks the loose slots makes no tight table.
template <bool MASKS, bool HIST = false, bool TIGHT = true>
__global__ __launch_bounds__(256) void k_pack_pval(PackIn in, u32 nTiles, const Scalars* __restrict__ sc,
                                                   const float* __restrict__ lutP, u32* __restrict__ ivEnd,
                                                   float* __restrict__ pOut, float thr, u64* __restrict__ sigMask,
                                                   u64* __restrict__ skipMask, u32* __restrict__ st,
                                                   const u32* __restrict__ tilePrevEnd = nullptr, u64* __restrict__ dense = nullptr) {
  __shared__ float hot[PP_HOT];  // indexed by the whole pileup c = V / 120 (consecutive banks, unlike V itself)
  __shared__ u32 vh[HIST ? 2 * PV_WHOLE : 1];
  for (int i = threadIdx.x; i < PP_HOT; i += 256) hot[i] = lutP[i * GX_UNIT];
  if (HIST)
    for (u32 i = threadIdx.x; i < 2 * PV_WHOLE; i += 256) vh[i] = 0;
  __syncthreads();
  u32 neg = 0;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const u32 stride = gridDim.x * 4;
  // A wavefront works on a pair of consecutive tiles at a time (their intervals are contiguous in
  // the tight arrays; a 4,096-base tile alone holds ~110 intervals, too few for 64 x PP_UNROLL
  // lanes).  Two-deep software pipeline: headers of pair k+2 and the first 64 * PP_UNROLL (end, V)
  // values of pair k+1 are in flight while pair k is scored.
  struct Hdr { u32 s0, s1, n0, n, dst, pe0, pe1; };   // (pe0 / pe1, HIST: where the two tiles' first intervals start)
  const u32 nUnits = (nTiles + 1) / 2;
  auto loadHdr = [&](u32 u) {
    Hdr h;
    const u32 t0 = 2 * u;
    h.s0 = in.meta[t0].slot;
    h.dst = in.tileIvOff[t0];
    const u32 mid = in.tileIvOff[t0 + 1];
    h.n0 = mid - h.dst;
    h.s1 = 0;
    h.n = h.n0;
    h.pe0 = h.pe1 = 0;
    if (HIST) h.pe0 = tilePrevEnd[t0];
    if (t0 + 1 < nTiles) {
      h.s1 = in.meta[t0 + 1].slot;
      h.n = in.tileIvOff[t0 + 2] - h.dst;
      if (HIST) h.pe1 = tilePrevEnd[t0 + 1];
    }
    return h;
  };
  auto srcOf = [](const Hdr& h, u32 i) { return i < h.n0 ? h.s0 + i : h.s1 + (i - h.n0); };
  u32 u = blockIdx.x * 4 + wv;
  Hdr h1{0, 0, 0, 0, 0, 0, 0}, h2{0, 0, 0, 0, 0, 0, 0};
  u32 e1[PP_UNROLL];
  int v1[PP_UNROLL];
#pragma unroll
  for (int k = 0; k < PP_UNROLL; k++) { e1[k] = 0; v1[k] = 0; }
  if (u < nUnits) {
    h1 = loadHdr(u);
#pragma unroll
    for (int k = 0; k < PP_UNROLL; k++)
      if (k * 64 + lane < h1.n) {
        const u32 si = srcOf(h1, k * 64 + lane);
        e1[k] = in.looseEnd[si];
        v1[k] = in.looseV[si];
      }
  }
  if (u + stride < nUnits) h2 = loadHdr(u + stride);
  for (; u < nUnits; u += stride) {
    const Hdr h = h1;
    const u32 dst = h.dst, n = h.n;
    // (HIST: where the pair's first interval starts, and the second tile's -- the end before it, or 0 on a new chromosome)
    u32 carryPrev = h.pe0;
    const u32 prevT1 = h.pe1;
    u32 e[PP_UNROLL];
    int v[PP_UNROLL];
#pragma unroll
    for (int k = 0; k < PP_UNROLL; k++) { e[k] = e1[k]; v[k] = v1[k]; }
    h1 = h2;
    if (!(u + stride < nUnits)) h1.n = 0;
#pragma unroll
    for (int k = 0; k < PP_UNROLL; k++)
      if (k * 64 + lane < h1.n) {
        const u32 si = srcOf(h1, k * 64 + lane);
        e1[k] = in.looseEnd[si];
        v1[k] = in.looseV[si];
      }
    if (u + 2 * stride < nUnits) h2 = loadHdr(u + 2 * stride);
    // score one batch of 64 * PP_UNROLL intervals starting at b (values already in e / v)
    auto scoreBatch = [&](u32 b, const u32 (&e)[PP_UNROLL], const int (&v)[PP_UNROLL]) {
#pragma unroll
      for (int k = 0; k < PP_UNROLL; k++) {
        const u32 i = b + k * 64 + lane;
        float p = 0.0f;
        if (i < n) {
          if (v[k] == V_MARK) {  // inside an excluded region: treatment 0.0f (2248), control SKIP (1871) -> p SKIP (1629)
            p = GX_SKIPF;
          } else {
            // pileups beyond the table (>= 2184) are scored by k_pval_deep: the double-precision
            // math would cost this kernel half its occupancy
            const u32 c = __umulhi((u32)v[k], 0x88888889u) >> 6;  // V / 120 for V >= 0
            if (v[k] >= 0 && c * GX_UNIT == (u32)v[k] && c < PP_HOT)  // a whole pileup
              p = hot[c];
            else {
              neg |= (u32)getval_neg(v[k]);  // updateVal's ERRPILE (1921)
              p = (u32)v[k] < PV_LUT ? lutP[v[k]] : 0.0f;
            }
          }
          if (TIGHT) {
            ivEnd[dst + i] = e[k];
            pOut[dst + i] = p;
          }
        }
        if (HIST) {  // the interval's length to its pileup's sum (all lanes take part in the shuffles)
          u32 pe = (u32)__shfl_up((int)e[k], 1, 64);
          const u32 first = k == 0 ? carryPrev : (u32)__shfl((int)e[k > 0 ? k - 1 : 0], 63, 64);
          if (lane == 0) pe = first;
          if (i == h.n0) pe = prevT1;
          if (i < n && v[k] != V_MARK && (u32)v[k] < PV_LUT) {
            const u32 len = e[k] - pe;
            const u32 c = __umulhi((u32)v[k], 0x88888889u) >> 6;
            if (c * GX_UNIT == (u32)v[k]) atomicAdd(&vh[(lane & 1) * PV_WHOLE + c], len);
            else {
              atomicAdd((unsigned long long*)&dense[v[k]], (unsigned long long)len);
              if (!TIGHT) atomicOr(st, ST_Q_LOOSE);   // (a pileup that is no whole number: k_bh_small's table is by whole pileup)
            }
          }
          if (k == PP_UNROLL - 1) carryPrev = (u32)__shfl((int)e[k], 63, 64);   // (the next batch's first interval follows lane 63's)
        }
        if (MASKS) {  // the sweep's significance / skip bit masks, while p is at hand (pre-zeroed words)
          const u64 sg = __ballot(p > thr), sk = __ballot(p == GX_SKIPF);
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
    };
    // (the pipelined first batch and the reloaded later ones -- dense tiles only -- are separate
    // pieces of code: a select between register and memory sources inside one loop costs the
    // load / store scheduling of the whole loop)
    scoreBatch(0, e, v);
    for (u32 b = 64 * PP_UNROLL; b < n; b += 64 * PP_UNROLL) {
      u32 e2[PP_UNROLL];
      int v2[PP_UNROLL];
#pragma unroll
      for (int k = 0; k < PP_UNROLL; k++) {
        const u32 i = b + k * 64 + lane;
        e2[k] = 0;
        v2[k] = 0;
        if (i < n) {
          const u32 si = srcOf(h, i);
          e2[k] = in.looseEnd[si];
          v2[k] = in.looseV[si];
        }
      }
      scoreBatch(b, e2, v2);
    }
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
  if (HIST) {
    __syncthreads();
    for (u32 c = threadIdx.x; c < PV_WHOLE; c += 256) {
      const u64 s2 = (u64)vh[c] + (u64)vh[PV_WHOLE + c];
      if (s2) atomicAdd((unsigned long long*)&dense[c * GX_UNIT], (unsigned long long)s2);
    }
  }
}

// the intervals of the deep tiles whose pileup lies beyond the table.  k_deep_risky runs while the
// sample is closed (the host synchronises there): it evaluates the same intervals and lists the
// values that are risky; the host's answers come back in `deep`, where k_pval_deep finds them.
__device__ __forceinline__ void deep_risky_body(PackIn in, const FragFix* __restrict__ ff, const u32* __restrict__ list,
                                                const Scalars* __restrict__ sc, RiskBuf* __restrict__ risk, u32 block, u32 nBlocks) {
  const u32 nList = ff->nList;
  const float lambda = sc->lambda;
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 li = block * 4 + wv; li < nList; li += nBlocks * 4) {
    const u32 t = list[li];
    const u32 src = in.meta[t].slot, n = in.tileIvOff[t + 1] - in.tileIvOff[t];
    for (u32 i = lane; i < n; i += 64) {
      const int v = in.looseV[src + i];
      if (v != V_MARK && (u32)v >= PV_LUT) {
        float val;
        bool ng, risky = false;
        (void)pval_of_v(v, lambda, ml, sl, &val, &ng, &risky);
        if (risky) risk_add(risk, RK_DEEP, (u32)v, 0, 0, 0.0);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_pval_deep(PackIn in, const FragFix* __restrict__ ff, const u32* __restrict__ list,
                                                   const Scalars* __restrict__ sc, const DeepTab* __restrict__ deep,
                                                   float* __restrict__ pOut, float thr, u64* __restrict__ sigMask) {
  const u32 nList = ff->nList;
  const float lambda = sc->lambda;
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 li = blockIdx.x * 4 + wv; li < nList; li += gridDim.x * 4) {
    const u32 t = list[li];
    const u32 src = in.meta[t].slot, dst = in.tileIvOff[t], n = in.tileIvOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      const int v = in.looseV[src + i];
      if (v != V_MARK && (u32)v >= PV_LUT) {
        float val;
        bool ng, risky = false;
        float p = pval_of_v(v, lambda, ml, sl, &val, &ng, &risky);
        if (risky) {  // the host's value (always there: k_deep_risky saw the same interval)
          const u32 nd = min(deep->n, DEEP_TAB);
          for (u32 j = 0; j < nd; j++)
            if (deep->v[j] == v) p = deep->p[j];
        }
        pOut[dst + i] = p;
        if (sigMask && p > thr) atomicOr((unsigned long long*)&sigMask[(dst + i) >> 6], 1ull << ((dst + i) & 63));
      }
    }
  }
}

// The pileup floats of a no-control replicate's intervals, on request: treatment value = getVal of the exact
// pileup (1902-1907; 0.0f inside an excluded region, 2248), control = lambda (SKIP inside one, 1871).  One
// wavefront per tile, from the loose slots to the tight position.
template <bool CTRL>
__global__ __launch_bounds__(256) void k_piles_from_loose(PackIn in, u32 nTiles, const float lambda,
                                                          float* __restrict__ exptOut, float* __restrict__ ctrlOut) {
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 t = blockIdx.x * 4 + wv; t < nTiles; t += gridDim.x * 4) {
    const u32 src = in.meta[t].slot, dst = in.tileIvOff[t], n = in.tileIvOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      const int v = in.looseV[src + i];
      bool ng;
      exptOut[dst + i] = v == V_MARK ? 0.0f : getval(v, &ng);
      if (CTRL) ctrlOut[dst + i] = v == V_MARK ? GX_SKIPF : lambda;
    }
  }
}

// ---- Benjamini-Hochberg: computeQval (352-401) / saveQval (212-250) -----------------------
// hashPval (300-327): genome-wide multiset {distinct float p -> total bp}.  Keys are the
// float's bits (p >= +0, so unsigned bit order == numeric order; -0 is folded into +0 as C's
// == does, 282).  Each workgroup first aggregates into an LDS table so the global table sees
// one atomic per (workgroup, distinct value).
constexpr u32 EMPTY_KEY = 0xFFFFFFFFu;  // a NaN pattern: never a p-value
// (round 6: the table's size, its probes and the workgroup's size are the instance's -- a run whose p-values are Fisher
// combinations holds several thousand distinct values in any stretch of the genome, and a 2048-entry table with 8 probes
// sent most of them to the device-wide table: 4 GB of atomics' lines at hg38 x 3 replicates)

__device__ __forceinline__ u32 bh_hash(u32 k) {
  k *= 2654435761u;
  return k ^ (k >> 15);
}

// the global table plus the list of its occupied slots: whoever claims a slot appends (key, slot),
// so that neither finding the distinct values nor cleaning up afterwards has to scan the table
struct BhTable {
  u32* keys;      // [cap] EMPTY_KEY when free
  u64* lens;      // [cap] bp
  u32 capMask;
  u32* outKeys;   // [<= cap] claimed keys, arbitrary order (sorted afterwards)
  u32* outSlot;
  u32* counter;
};

constexpr u32 BH_MAX_PROBE = 4096;  // an insertion that has not found its slot by then reports the table full (the host grows it)

__device__ __forceinline__ void bh_global_add(const BhTable& T, u32 key, u64 len, u32* st) {
  u32 h = bh_hash(key) & T.capMask;
  for (u32 probe = 0; probe <= T.capMask && probe < BH_MAX_PROBE; probe++) {
    u32 old = T.keys[h];
    if (old != key) {
      if (old != EMPTY_KEY) { h = (h + 1) & T.capMask; continue; }
      old = atomicCAS(&T.keys[h], EMPTY_KEY, key);
      if (old == EMPTY_KEY) {  // this thread claimed the slot
        const u32 j = atomicAdd(T.counter, 1u);
        T.outKeys[j] = key;
        T.outSlot[j] = h;
      } else if (old != key) { h = (h + 1) & T.capMask; continue; }
    }
    atomicAdd(&T.lens[h], len);
    return;
  }
  atomicOr(st, ST_HASH_FULL);
}

// frees the claimed slots again (the table is handed back clean instead of being wiped per call)
__global__ __launch_bounds__(256) void k_bh_clear(BhTable T, u64* __restrict__ kq = nullptr) {
  const u32 n = *T.counter;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 h = T.outSlot[i];
    T.keys[h] = EMPTY_KEY;
    T.lens[h] = 0;
    if (kq) kq[h] = ~0ull;
  }
}

// k_pack_pval<.., HIST>'s companion: the intervals whose pileup lies beyond the table p(V) (deep tiles: a tower) straight into the
// hash table, under the p-value k_pval_deep gave them
__global__ __launch_bounds__(256) void k_deep_hist(PackIn in, const FragFix* __restrict__ ff, const u32* __restrict__ list,
                                                   const u32* __restrict__ tilePrevEnd, const float* __restrict__ pTight, BhTable T,
                                                   u32* __restrict__ st) {
  const u32 nList = ff->nList;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  for (u32 li = blockIdx.x * 4 + wv; li < nList; li += gridDim.x * 4) {
    const u32 t = list[li];
    const u32 src = in.meta[t].slot, dst = in.tileIvOff[t], n = in.tileIvOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      const int v = in.looseV[src + i];
      if (v != V_MARK && (u32)v >= PV_LUT) {
        const u32 pe = i ? in.looseEnd[src + i - 1] : tilePrevEnd[t];
        const float p = pTight[dst + i];
        bh_global_add(T, p == 0.0f ? 0u : __float_as_uint(p), (u64)(in.looseEnd[src + i] - pe), st);
      }
    }
  }
}

// {key, q} of every claimed slot side by side (round 6): k_qlookup's probe of a value that its LDS cache does not hold -- a tenth of
// the intervals of Fisher-combined replicates, whose p-values are nearly all different around the peaks -- was two dependent 128-byte
// fetches (the key, then q of the slot); one now.  Free slots hold ~0 (EMPTY_KEY in the low word).
// (pStar, round 6: the smallest p -- as its bits: the order of non-negative floats -- whose q exceeds the threshold.  q never falls as p
// grows (computeQval's running minimum, 392-399), so "q > thr" (callPeaks 1015) is "p >= pStar": the sweep's significance bits come from
// one compare per interval, k_sig_from_p, and q is looked up where somebody reads it -- inside the candidates, k_q_fill_cands)
__global__ __launch_bounds__(256) void k_kq_build(BhTable T, const float* __restrict__ qOfSlot, u64* __restrict__ kq, float thr,
                                                  u32* __restrict__ pStar) {
  const u32 n = *T.counter;
  u32 mn = 0xFFFFFFFFu;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 h = T.outSlot[i];
    const u32 key = T.keys[h];
    const float q = qOfSlot[h];
    kq[h] = (u64)key | ((u64)__float_as_uint(q) << 32);
    if (q > thr) mn = min(mn, key);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mn = min(mn, (u32)__shfl_xor((int)mn, d, 64));
  if (lane_id() == 0 && mn != 0xFFFFFFFFu) atomicMin(pStar, mn);
}

// a value's q from the {key, q} table (lookup 196-206); the probe ends at a free slot (a value that was never inserted: the caller raises
// the reference's "does not match p-value length")
__device__ __forceinline__ bool kq_probe(const u64* __restrict__ kq, u32 capMask, u32 key, float* q) {
  u32 h = bh_hash(key) & capMask;
  u64 ge;
  while ((u32)(ge = kq[h]) != key && (u32)ge != EMPTY_KEY) h = (h + 1) & capMask;
  *q = __uint_as_float((u32)(ge >> 32));
  return (u32)ge == key;
}

// the sweep's significance / SKIP masks from the p-values and pStar: whole words, four per iteration and wavefront
__global__ __launch_bounds__(256) void k_sig_from_p(const float* __restrict__ p, const u32* __restrict__ nPtr, const u32* __restrict__ pStar,
                                                    u64* __restrict__ sigMask, u64* __restrict__ skipMask) {
  const u32 n = *nPtr, nw = (n + 63) >> 6, ps = *pStar;
  for (u32 w0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; w0 < nw; w0 += gridDim.x * 16) {
    float pv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = ((w0 + k) << 6) + lane_id();
      pv[k] = i < n ? p[i] : GX_SKIPF;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = ((w0 + k) << 6) + lane_id();
      const bool skip = pv[k] == GX_SKIPF;
      const u32 key = pv[k] == 0.0f ? 0u : __float_as_uint(pv[k]);
      const u64 sg = __ballot(i < n && !skip && key >= ps), sk = __ballot(i < n && skip);
      if (lane_id() == 0 && w0 + k < nw) {
        sigMask[w0 + k] = sg;
        skipMask[w0 + k] = sk;
      }
    }
  }
}

// q of the intervals inside the candidates (what updatePeak reads, 943-970): one wavefront per candidate, 64 probes per step.  Everything
// else of q[] stays unwritten until somebody asks for the array (ensure_q: k_qlookup).
__global__ __launch_bounds__(256) void k_q_fill_cands(const uint4* __restrict__ hdr, const u32* __restrict__ nCands, const float* __restrict__ p,
                                                      const u64* __restrict__ kq, u32 capMask, float* __restrict__ q, u32* __restrict__ st) {
  const u32 C = *nCands;
  for (u32 c = blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += gridDim.x * 4) {
    const uint4 h = hdr[c];
    // (four steps in flight: the p-values, then the four first probes side by side -- a probe is a line of a table that no cache holds)
    for (u32 i0 = h.x + lane_id(); i0 <= h.y; i0 += 256) {
      float pv[4];
      u32 key[4], hs[4];
      u64 ge[4];
#pragma unroll
      for (int k = 0; k < 4; k++) pv[k] = i0 + k * 64 <= h.y ? p[i0 + k * 64] : GX_SKIPF;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        key[k] = pv[k] == 0.0f ? 0u : __float_as_uint(pv[k]);
        hs[k] = bh_hash(key[k]) & capMask;
        ge[k] = pv[k] != GX_SKIPF ? kq[hs[k]] : 0ull;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (i0 + k * 64 > h.y) continue;
        float qv = GX_SKIPF;
        if (pv[k] != GX_SKIPF) {
          while ((u32)ge[k] != key[k] && (u32)ge[k] != EMPTY_KEY) {
            hs[k] = (hs[k] + 1) & capMask;
            ge[k] = kq[hs[k]];
          }
          if ((u32)ge[k] == key[k]) qv = __uint_as_float((u32)(ge[k] >> 32));
          else atomicOr(st, ST_BH_LEN);
        }
        q[i0 + k * 64] = qv;
      }
    }
  }
}

__host__ __device__ constexpr size_t bh_hist_lds(int lt) { return (size_t)lt * 12; }
template <int NT, int BH_LT, int BH_LPROBE>
__global__ __launch_bounds__(NT) void k_bh_hist(const u32* __restrict__ end, const float* __restrict__ p,
                                                const u32* __restrict__ chromOff, u32 nChrom,
                                                const u32* __restrict__ nPtr, BhTable T, u32* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bh_raw[];
  u64* ll = reinterpret_cast<u64*>(bh_raw);
  u32* lk = reinterpret_cast<u32*>(bh_raw + (size_t)BH_LT * 8);
  for (int i = threadIdx.x; i < BH_LT; i += NT) { lk[i] = EMPTY_KEY; ll[i] = 0; }
  __syncthreads();
  const u32 n = *nPtr;
  const u32 per = (n + gridDim.x - 1) / gridDim.x;
  const u32 b0 = blockIdx.x * per, b1 = min(n, b0 + per);
  ChromCursor cur;
  for (u32 i0 = b0 + threadIdx.x; i0 < b1; i0 += 4 * NT) {  // four intervals per thread in flight
    float pv4[4];
    u32 e4[4], s4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = i0 + k * NT;
      pv4[k] = GX_SKIPF;
      e4[k] = 0;
      s4[k] = 0;
      if (i < b1) {
        pv4[k] = p[i];
        e4[k] = end[i];
        s4[k] = i ? end[i - 1] : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = i0 + k * NT;
      const float pv = pv4[k];
      if (i < b1 && pv != GX_SKIPF) {  // 319
        cur.seek(chromOff, nChrom, i);
        const u32 s = i == cur.lo ? 0u : s4[k];
        const u64 len = e4[k] - s;
        const u32 key = pv == 0.0f ? 0u : __float_as_uint(pv);
        u32 h = bh_hash(key) & (BH_LT - 1);
        bool placed = false;
        for (int probe = 0; probe < BH_LPROBE && !placed; probe++) {
          u32 old = lk[h];
          if (old != key) {
            if (old == EMPTY_KEY) old = atomicCAS(&lk[h], EMPTY_KEY, key);
            if (old != EMPTY_KEY && old != key) { h = (h + 1) & (BH_LT - 1); continue; }
          }
          atomicAdd(&ll[h], len);
          placed = true;
        }
        if (!placed) bh_global_add(T, key, len, st);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH_LT; i += NT)
    if (lk[i] != EMPTY_KEY) bh_global_add(T, lk[i], ll[i], st);
}

// the exchange format of the multi-GPU BH table: this rank's distinct values as dense records ...
struct BhRec { u32 key, pad; u64 bp; };

__global__ __launch_bounds__(256) void k_bh_pack(const u32* __restrict__ keys, const u32* __restrict__ slots,
                                                 const u64* __restrict__ gLens, u32 n, BhRec* __restrict__ out) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = BhRec{keys[i], 0u, gLens[slots[i]]};
}

// ... and the insertion of every rank's records into a fresh table
__global__ __launch_bounds__(256) void k_bh_insert(const BhRec* __restrict__ recs, u32 n, BhTable T, u32* __restrict__ st) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bh_global_add(T, recs[i].key, recs[i].bp, st);
}

// the same from an all-gathered buffer: rank r's records start at r * stride, counts[r] of them
__global__ __launch_bounds__(256) void k_bh_insert_gathered(const BhRec* __restrict__ recs, const u32* __restrict__ counts,
                                                            u32 world, u32 stride, BhTable T, u32* __restrict__ st) {
  for (u32 r = blockIdx.y; r < world; r += gridDim.y) {
    const u32 n = min(counts[r], stride);
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
      bh_global_add(T, recs[(size_t)r * stride + i].key, recs[(size_t)r * stride + i].bp, st);
  }
}

// ---- the genome-wide p-value histogram of several ranks, dense (no control: p is a function of the pileup V) ---------
// hashPval (300-327) over ALL chromosomes.  Every rank has the same table p(V) (same lambda), so "bp at p" travels as
// "bp at V": one u64 per table entry, summed over the ranks by ONE all-reduce -- no counts to exchange first, no host
// synchronisation inside the exchange (north_star's "single RCCL allreduce for the global p-value histogram").  The few
// values that are not in the table (pileups beyond 2^18 / 120 = 2,184: k_pval_deep's) ride in the same buffer: behind
// the dense part every rank owns a region {count, (key, bp) x BHD_SIDE} that only it writes -- the sum over the ranks is
// the concatenation.  A rank with more such values than fit says so in its count; every rank sees that after the
// all-reduce and all take the general exchange together.
constexpr u32 BHD_SIDE = 4096;                       // values outside the table per rank
constexpr u32 BHD_REGION = 2 + 2 * BHD_SIDE;         // u64 words of a rank's region
__host__ __device__ inline size_t bhd_words(u32 world) { return (size_t)PV_LUT + (size_t)world * BHD_REGION; }

// this rank's distinct values (the claimed slots of its table) -> the dense buffer
__global__ __launch_bounds__(256) void k_bh_dense_fill(const u32* __restrict__ keys, const u32* __restrict__ slots,
                                                       const u64* __restrict__ gLens, const u32* __restrict__ nPtr,
                                                       const float* __restrict__ lutP, u64* __restrict__ dense, u32 rank) {
  const u32 n = *nPtr;
  u64* region = dense + PV_LUT + (size_t)rank * BHD_REGION;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 key = keys[i];
    const u64 bp = gLens[slots[i]];
    // first V with p(V) >= key (p is non-decreasing in V; a table that is not misses, and the value travels in the region)
    u32 lo = 0, hi = PV_LUT;
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      const float pm = lutP[mid];
      if ((pm == 0.0f ? 0u : __float_as_uint(pm)) < key) lo = mid + 1; else hi = mid;
    }
    const float pl = lo < PV_LUT ? lutP[lo] : -1.0f;
    if (lo < PV_LUT && (pl == 0.0f ? 0u : __float_as_uint(pl)) == key)
      atomicAdd(&dense[lo], bp);
    else {
      const u64 j = atomicAdd(&region[0], 1ull);
      if (j < BHD_SIDE) {
        region[2 + 2 * j] = key;
        region[3 + 2 * j] = bp;
      }
    }
  }
}

// the summed buffer -> a fresh table (equal p of different V fall into one entry, as in the reference's hash)
__global__ __launch_bounds__(256) void k_bh_from_dense(const u64* __restrict__ dense, const float* __restrict__ lutP, u32 world,
                                                       BhTable T, u32* __restrict__ overflow, u32* __restrict__ st) {
  for (u32 v = blockIdx.x * 256 + threadIdx.x; v < PV_LUT; v += gridDim.x * 256) {
    const u64 bp = dense[v];
    if (bp) {
      const float p = lutP[v];
      bh_global_add(T, p == 0.0f ? 0u : __float_as_uint(p), bp, st);
    }
  }
  for (u32 r = 0; r < world; r++) {
    const u64* region = dense + PV_LUT + (size_t)r * BHD_REGION;
    u64 n = region[0];
    if (n > BHD_SIDE) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1u;
      n = BHD_SIDE;
    }
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < (u32)n; j += gridDim.x * 256)
      bh_global_add(T, (u32)region[2 + 2 * j], region[3 + 2 * j], st);
  }
}

// float log10 exactly as the host's libm evaluates it (saveQval 221, 226 call log10f).
// glibc 2.35's log10f is the fdlibm formula  z = y*log10_2lo + ivln10*logf(m);  z + y*log10_2hi
// (float ops) around its table-driven logf (16-entry table, cubic in double).  Restated here and
// verified bit-identical to the host's log10f over ALL positive normal floats
// (oracle/check_log10f.c: 2,130,706,432 inputs, 0 mismatches, with or without FMA contraction),
// so the BH table is bit-exact instead of merely within tolerance.  Inputs here are integers
// >= 1 (k, genome length), so the subnormal/negative branches are not needed.
__device__ inline float logf_host(float x) {
  const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0,
                           0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                           0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                           0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
  const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                           -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                           -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                           0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  u32 ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  u32 tmp = ix - 0x3f330000u;
  int i = (tmp >> 19) & 15;
  int k = (int)tmp >> 23;
  u32 iz = ix - (tmp & (0x1ffu << 23));
  double z = (double)__uint_as_float(iz);
  double r = z * invc[i] - 1;
  double y0 = logc[i] + (double)k * Ln2;
  double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

__device__ inline float log10f_host(float x) {
  const float ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
  int hx = (int)__float_as_uint(x);
  int k = (hx >> 23) - 127;
  int i = ((u32)k & 0x80000000u) >> 31;
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  float y = (float)(k + i);
  float m = __uint_as_float((u32)hx);
  float z = y * log10_2lo + ivln10 * logf_host(m);
  return z + y * log10_2hi;
}

// saveQval 219-229 on the sorted table (ascending p): from the most significant value down,
//   raw_i = p_i + logN + log10f(k_i),  k_i = 1 + bp with strictly larger p   (float adds, left to right)
//   q_i = max(min(raw_i, q_{i+1}), 0) = max(min_{j>=i} raw_j, 0)               (suffix minimum: exact)
// Single workgroup; each thread owns a contiguous run of the REVERSED order.
template <typename T, int NT, typename Op>
__device__ __forceinline__ T block_excl_scan_op(T v, T identity, T* scratch, Op op) {
  constexpr int NW = NT / 64;
  T inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(inc, d, 64);
    if (lane_id() >= d) inc = op(o, inc);
  }
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  if (threadIdx.x < 64) {
    T x = threadIdx.x < NW ? scratch[threadIdx.x] : identity;
    T xi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      T o = __shfl_up(xi, d, 64);
      if (lane_id() >= d) xi = op(o, xi);
    }
    T xe = __shfl_up(xi, 1, 64);
    if (threadIdx.x == 0) xe = identity;
    if (threadIdx.x < NW) scratch[threadIdx.x] = xe;
  }
  __syncthreads();
  T prev = __shfl_up(inc, 1, 64);
  if (lane_id() == 0) prev = identity;
  T res = op(scratch[w], prev);
  __syncthreads();
  return res;
}

struct OpAddU64 { __device__ u64 operator()(u64 a, u64 b) const { return a + b; } };
struct OpMinF { __device__ float operator()(float a, float b) const { return a < b ? a : b; } };
struct OpLastU32 { __device__ u32 operator()(u32 a, u32 b) const { return b != 0xFFFFFFFFu ? b : a; } };   // (the later one, if there is one)

__global__ __launch_bounds__(1024) void k_qtable(const u32* __restrict__ keys, const u32* __restrict__ slots,
                                                 const u64* __restrict__ gLens, u32 D, const u64* __restrict__ genomeLenPtr,
                                                 float* __restrict__ qOfSlot, float* __restrict__ raw /* scratch [D] */,
                                                 u32* __restrict__ allOne, u32* __restrict__ lenCheck) {
  __shared__ u64 s64[20];
  __shared__ float sf[20];
  const float logN = -log10f_host((float)*genomeLenPtr);
  const u32 per = (D + 1023) / 1024;
  const u32 r0 = min(D, threadIdx.x * per), r1 = min(D, r0 + per);  // reversed indices [r0, r1)
  u64 sum = 0;
  for (u32 r = r0; r < r1; r++) sum += gLens[slots[D - 1 - r]];
  u64 k = 1 + block_excl_scan_op<u64, 1024>(sum, 0ull, s64, OpAddU64());
  // computeQval 377-382: the lengths collected with the p-values must add up to the genome length (lenCheck: the status
  // word, when the genome length was computed and not given with -L).  With several ranks this is the table AFTER the
  // exchange: a rank whose histogram never arrived shows up here, on every rank.
  if (lenCheck && threadIdx.x == 1023 && k - 1 + sum != *genomeLenPtr) atomicOr(lenCheck, ST_BH_LEN);
  float mn = FLT_MAX;
  for (u32 r = r0; r < r1; r++) {
    u32 i = D - 1 - r;
    float pv = __uint_as_float(keys[i]);
    float rw = pv + logN + log10f_host((float)k);
    raw[i] = rw;
    mn = rw < mn ? rw : mn;
    k += gLens[slots[i]];
  }
  float run = block_excl_scan_op<float, 1024>(mn, FLT_MAX, sf, OpMinF());
  for (u32 r = r0; r < r1; r++) {
    u32 i = D - 1 - r;
    float rw = raw[i];
    run = rw < run ? rw : run;
    float q = run > 0.0f ? run : 0.0f;
    qOfSlot[slots[i]] = q;
    if (r == 0 && allOne) *allOne = q == 0.0f;  // "All q-values are 1" (245)
  }
}

// -q on the loose slots (round 6): ALL of computeQval (352-401) for a table whose values are a function of the whole pileup, in one
// workgroup -- no device-wide insertion, no sort, no round trip to the host for the number of distinct values.  dense[120 c] = bp at
// pileup c (k_pack_pval<.., HIST, false>), p(c) = lutP[120 c].  p never falls as c grows (checked: ST_Q_LOOSE otherwise, the host takes
// the tight table), so the table sorted by p is the pileups in their own order, equal p lying side by side: the most significant
// first, as k_qtable walks its sorted values.  A value's rank k = 1 + bp of the strictly more significant ones, raw = p + logN +
// log10f(k), q = max(min over the values at least as significant, 0).  Of the pileups that share one p the topmost PRESENT one
// stands for the value (everything above it is strictly more significant); a pileup that does not occur takes no part.
// Out: qLut[c] (k_peak_both's LDS table), vq[0] / vq[1] (the smallest present pileup whose q passes, 1 + the largest whose q does
// not, in 1/120 units; preset by the host), {key, q} of every distinct value in the run's table (what k_qlookup probes when
// gx_get_intervals asks for the q array), "all q-values are 1" (245), the length check (377-382).
__global__ __launch_bounds__(1024) void k_bh_small(const u64* __restrict__ dense, const float* __restrict__ lutP, BhTable T,
                                                   u64* __restrict__ kq, const u64* __restrict__ genomeLenPtr, float thr,
                                                   float* __restrict__ qLut, u32* __restrict__ vq, u32* __restrict__ allOne,
                                                   u32* __restrict__ lenCheck, u32* __restrict__ st) {
  __shared__ u64 s64[20];
  __shared__ float sf[20];
  __shared__ u32 su[20];
  __shared__ u32 keyS[PV_WHOLE];
  __shared__ u64 bpS[PV_WHOLE];
  for (u32 c = threadIdx.x; c < PV_WHOLE; c += 1024) {
    const float pv = lutP[c * (u32)GX_UNIT];
    keyS[c] = pv == 0.0f ? 0u : __float_as_uint(pv);
    bpS[c] = dense[c * (u32)GX_UNIT];
  }
  __syncthreads();
  const float logN = -log10f_host((float)*genomeLenPtr);
  constexpr u32 PER = (PV_WHOLE + 1023) / 1024, NONE = 0xFFFFFFFFu;
  const u32 r0 = min((u32)PV_WHOLE, threadIdx.x * PER), r1 = min((u32)PV_WHOLE, r0 + PER);  // reversed indices: c = PV_WHOLE - 1 - r
  // bp of the strictly-above entries, and the nearest present entry above (reversed order: "before")
  u64 sum = 0;
  u32 lastPresent = NONE;   // (the lowest present c of my run = the latest in reversed order)
  for (u32 r = r0; r < r1; r++) {
    const u32 c = PV_WHOLE - 1 - r;
    sum += bpS[c];
    if (bpS[c]) lastPresent = c;
  }
  u64 above = block_excl_scan_op<u64, 1024>(sum, 0ull, s64, OpAddU64());
  u32 presAbove = block_excl_scan_op<u32, 1024>(lastPresent, NONE, su, OpLastU32());
  if (lenCheck && threadIdx.x == 1023 && above + sum != *genomeLenPtr) atomicOr(lenCheck, ST_BH_LEN);
  float raw[PER], mn = FLT_MAX;
  u32 bad = 0, firstIdx = NONE;   // (firstIdx: the genome's most significant value is mine)
  for (u32 r = r0; r < r1; r++) {
    const u32 c = PV_WHOLE - 1 - r, i = r - r0;
    raw[i] = FLT_MAX;
    if (bpS[c]) {
      if (presAbove == NONE) firstIdx = i;
      if (presAbove != NONE && keyS[presAbove] < keyS[c]) bad = 1;   // p falls as the pileup grows
      if (presAbove == NONE || keyS[presAbove] != keyS[c]) raw[i] = __uint_as_float(keyS[c]) + logN + log10f_host((float)(1ull + above));
      presAbove = c;
    }
    above += bpS[c];
    mn = raw[i] < mn ? raw[i] : mn;
  }
  if (bad) atomicOr(st, ST_Q_LOOSE);
  float run = block_excl_scan_op<float, 1024>(mn, FLT_MAX, sf, OpMinF());
  u32 vMin = NONE, vMax = 0;
  for (u32 r = r0; r < r1; r++) {
    const u32 c = PV_WHOLE - 1 - r, i = r - r0;
    const bool top = raw[i] != FLT_MAX;
    run = raw[i] < run ? raw[i] : run;
    const float q = run > 0.0f ? run : 0.0f;
    qLut[c] = bpS[c] ? q : 0.0f;
    if (bpS[c]) {
      if (q > thr) vMin = min(vMin, c * (u32)GX_UNIT);
      else vMax = max(vMax, c * (u32)GX_UNIT + 1u);
    }
    if (top) {
      // the value's entry in the run's table (clean at this point: at most PV_WHOLE insertions into >= 2^22 slots)
      const u32 key = keyS[c];
      u32 h = bh_hash(key) & T.capMask, old;
      while ((old = atomicCAS(&T.keys[h], EMPTY_KEY, key)) != EMPTY_KEY && old != key) h = (h + 1) & T.capMask;
      if (old == EMPTY_KEY) {
        const u32 j = atomicAdd(T.counter, 1u);
        T.outKeys[j] = key;
        T.outSlot[j] = h;
      }
      kq[h] = (u64)key | ((u64)__float_as_uint(q) << 32);
    }
    if (allOne && i == firstIdx) *allOne = q == 0.0f;  // "All q-values are 1" (245)
  }
  if (vMin != NONE) atomicMin(&vq[0], vMin);
  if (vMax) atomicMax(&vq[1], vMax);
}

// The same table for many distinct values (Fisher-combined replicates give millions): three launches over
// chunks of QT_CHUNK entries, in reversed order r = D - 1 - i (most significant first).  A chunk's prefix
// -- base pairs of all more significant values, then the minimum of their raw q -- is the reduction of
// the earlier chunks' aggregates, which every workgroup forms for itself (D / QT_CHUNK values).
constexpr int QT_NT = 256, QT_ITEMS = 8, QT_CHUNK = QT_NT * QT_ITEMS;

// (Dptr: the number of entries when only the device knows it -- the owner's table of the range-partitioned exchange,
// gx_bhx.h: the grid then covers an upper bound and D is read here)
__global__ __launch_bounds__(QT_NT) void k_qt_sums(const u32* __restrict__ slots, const u64* __restrict__ gLens, u32 D,
                                                   u64* __restrict__ dl /* [D] by r */, u64* __restrict__ chunkSum,
                                                   const u32* __restrict__ Dptr) {
  __shared__ u64 red[QT_NT / 64];
  if (Dptr) D = *Dptr;
  const u32 base = blockIdx.x * QT_CHUNK;
  u64 sum = 0;
#pragma unroll
  for (int k = 0; k < QT_ITEMS; k++) {
    const u32 r = base + k * QT_NT + threadIdx.x;
    if (r < D) {
      const u64 v = gLens[slots[D - 1 - r]];
      dl[r] = v;
      sum += v;
    }
  }
  sum = wave_sum(sum);
  if (lane_id() == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int w = 0; w < QT_NT / 64; w++) t += red[w];
    chunkSum[blockIdx.x] = t;
  }
}

// (totals / world / rank: several ranks, this one owns range `rank` of the p axis -- the base pairs of the ranges above
// count as "earlier chunks")
__global__ __launch_bounds__(QT_NT) void k_qt_raw(const u32* __restrict__ keys, const u64* __restrict__ dl, u32 D,
                                                  const u64* __restrict__ genomeLenPtr, const u64* __restrict__ chunkSum,
                                                  float* __restrict__ raw /* [D] by r */, float* __restrict__ chunkMin,
                                                  const u32* __restrict__ Dptr, const u64* __restrict__ totals, u32 world, u32 rank,
                                                  u32* __restrict__ lenCheck) {
  __shared__ u64 s64[QT_NT / 64 + 4];
  __shared__ float redf[QT_NT / 64];
  __shared__ u64 pre64;
  if (Dptr) D = *Dptr;
  // computeQval 377-382 (see k_qtable): several ranks -- the ranges' totals, which every rank holds after their exchange;
  // one rank -- the chunks' totals (<= D / QT_CHUNK words, by the first workgroup)
  if (lenCheck && blockIdx.x == 0) {
    u64 all = 0;
    if (totals)
      for (u32 o = threadIdx.x; o < world; o += QT_NT) all += totals[o];
    else
      for (u32 c = threadIdx.x; c < gridDim.x; c += QT_NT) all += chunkSum[c];
    all = wave_sum(all);
    if (lane_id() == 0) s64[threadIdx.x >> 6] = all;
    __syncthreads();
    if (threadIdx.x == 0) {
      u64 t = 0;
      for (int w = 0; w < QT_NT / 64; w++) t += s64[w];
      if (t != *genomeLenPtr) atomicOr(lenCheck, ST_BH_LEN);
    }
    __syncthreads();
  }
  const float logN = -log10f_host((float)*genomeLenPtr);
  u64 before = 0;  // base pairs of the earlier chunks
  if (totals)
    for (u32 o = rank + 1 + threadIdx.x; o < world; o += QT_NT) before += totals[o];
  for (u32 c = threadIdx.x; c < blockIdx.x; c += QT_NT) before += chunkSum[c];
  before = wave_sum(before);
  if (lane_id() == 0) s64[threadIdx.x >> 6] = before;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int w = 0; w < QT_NT / 64; w++) t += s64[w];
    pre64 = t;
  }
  __syncthreads();
  const u64 chunkBase = pre64;
  __syncthreads();
  const u32 r0 = blockIdx.x * QT_CHUNK + threadIdx.x * QT_ITEMS;  // consecutive entries per thread
  u64 v[QT_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < QT_ITEMS; k++) {
    v[k] = r0 + k < D ? dl[r0 + k] : 0ull;
    sum += v[k];
  }
  u64 kk = 1 + chunkBase + block_excl_scan_op<u64, QT_NT>(sum, 0ull, s64, OpAddU64());
  float mn = FLT_MAX;
#pragma unroll
  for (int k = 0; k < QT_ITEMS; k++) {
    if (r0 + k < D) {
      const float pv = __uint_as_float(keys[D - 1 - (r0 + k)]);
      const float rw = pv + logN + log10f_host((float)kk);
      raw[r0 + k] = rw;
      mn = rw < mn ? rw : mn;
      kk += v[k];
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float o = __shfl_xor(mn, d, 64);
    mn = o < mn ? o : mn;
  }
  if (lane_id() == 0) redf[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = FLT_MAX;
    for (int w = 0; w < QT_NT / 64; w++) t = redf[w] < t ? redf[w] : t;
    chunkMin[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(QT_NT) void k_qt_apply(const u32* __restrict__ slots, const float* __restrict__ raw, u32 D,
                                                    const float* __restrict__ chunkMin, float* __restrict__ qOfSlot,
                                                    u32* __restrict__ allOne, const u32* __restrict__ Dptr,
                                                    const u64* __restrict__ mins /* float bits per range */, u32 world, u32 rank) {
  __shared__ float sf[QT_NT / 64 + 4];
  __shared__ float redf[QT_NT / 64];
  __shared__ float preMin;
  if (Dptr) D = *Dptr;
  float before = FLT_MAX;  // smallest raw q of the earlier chunks
  if (mins)
    for (u32 o = rank + 1 + threadIdx.x; o < world; o += QT_NT) {
      const float m = __uint_as_float((u32)mins[o]);
      before = m < before ? m : before;
    }
  for (u32 c = threadIdx.x; c < blockIdx.x; c += QT_NT) before = chunkMin[c] < before ? chunkMin[c] : before;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float o = __shfl_xor(before, d, 64);
    before = o < before ? o : before;
  }
  if (lane_id() == 0) redf[threadIdx.x >> 6] = before;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = FLT_MAX;
    for (int w = 0; w < QT_NT / 64; w++) t = redf[w] < t ? redf[w] : t;
    preMin = t;
  }
  __syncthreads();
  const float chunkBase = preMin;
  const u32 r0 = blockIdx.x * QT_CHUNK + threadIdx.x * QT_ITEMS;
  float rw[QT_ITEMS], mn = FLT_MAX;
#pragma unroll
  for (int k = 0; k < QT_ITEMS; k++) {
    rw[k] = r0 + k < D ? raw[r0 + k] : FLT_MAX;
    mn = rw[k] < mn ? rw[k] : mn;
  }
  float run = block_excl_scan_op<float, QT_NT>(mn, FLT_MAX, sf, OpMinF());
  run = chunkBase < run ? chunkBase : run;
#pragma unroll
  for (int k = 0; k < QT_ITEMS; k++) {
    if (r0 + k < D) {
      run = rw[k] < run ? rw[k] : run;
      const float q = run > 0.0f ? run : 0.0f;
      qOfSlot[slots[D - 1 - (r0 + k)]] = q;
      if (r0 + k == 0 && allOne) *allOne = q == 0.0f;  // "All q-values are 1" (245)
    }
  }
}

// per interval: q = table[p] (lookup 196-206), SKIP stays SKIP (237-238); the sweep's significance /
// SKIP masks are written on the way (whole words: one wavefront per 64 intervals, four words per
// iteration so that four loads per lane are in flight)
// (a direct-mapped LDS cache of {p bits, q} sits in front of the table: the genome's common values -- the background
// pileups -- are asked for millions of times, and a hit saves the two dependent gathers of the probe.  An entry is
// one 8-byte LDS word, written and read whole; q is a function of p within a run.)
// (round 6: the cache's size and the workgroup's are the instance's, as for k_bh_hist -- 2048 entries thrash on Fisher-combined
// p-values, and every miss is two dependent 128-byte fetches from the table)
template <int NT, int QL_CACHE_LOG>
__global__ __launch_bounds__(NT) void k_qlookup(const float* __restrict__ p, const u32* __restrict__ nPtr,
                                                const u64* __restrict__ kq, u32 capMask, float* __restrict__ q, float thr, u64* __restrict__ sigMask,
                                                u64* __restrict__ skipMask, u32* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ql_raw[];
  u64* lc = reinterpret_cast<u64*>(ql_raw);  // [31:0] key, [63:32] q bits; key EMPTY_KEY: free
  for (int i = threadIdx.x; i < (1 << QL_CACHE_LOG); i += NT) lc[i] = (u64)EMPTY_KEY;
  __syncthreads();
  const u32 n = *nPtr;
  const u32 nw = (n + 63) >> 6;
  constexpr u32 NW = NT / 64;
  for (u32 w0 = (blockIdx.x * NW + (threadIdx.x >> 6)) * 4; w0 < nw; w0 += gridDim.x * NW * 4) {
    float pv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = ((w0 + k) << 6) + lane_id();
      pv[k] = i < n ? p[i] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = ((w0 + k) << 6) + lane_id();
      float qv = 0.0f;
      if (i < n) {
        if (pv[k] == GX_SKIPF)
          qv = GX_SKIPF;
        else {
          const u32 key = pv[k] == 0.0f ? 0u : __float_as_uint(pv[k]);
          const u32 hh = bh_hash(key);
          const u32 hl = (hh >> 7) & ((1u << QL_CACHE_LOG) - 1);
          const u64 c = lc[hl];
          if ((u32)c == key)
            qv = __uint_as_float((u32)(c >> 32));
          else {
            u32 h = hh & capMask;
            // (every p of this rank was inserted -- and if an exchange lost the table, the probe ends at a free slot
            // instead of circling: the run fails with the reference's "does not match p-value length", 377-382)
            u64 ge;
            while ((u32)(ge = kq[h]) != key && (u32)ge != EMPTY_KEY) h = (h + 1) & capMask;
            if ((u32)ge == key) {
              qv = __uint_as_float((u32)(ge >> 32));
              lc[hl] = ge;
            } else
              atomicOr(st, ST_BH_LEN);
          }
        }
        q[i] = qv;
      }
      const u64 sg = __ballot(i < n && qv > thr), sk = __ballot(i < n && qv == GX_SKIPF);
      if (sigMask && lane_id() == 0 && w0 + k < nw) {   // (nullptr: the array alone, for gx_get_intervals -- ensure_q)
        sigMask[w0 + k] = sg;
        skipMask[w0 + k] = sk;
      }
    }
  }
}

// ---- peak sweep: callPeaks (977-1069) ---------------------------------------------------------
// A candidate peak is a maximal chain of significant intervals (pq > thr, strict, 1015) in which
// consecutive members are separated by less than... precisely: two significant intervals belong
// to one candidate iff they are on the same chromosome, no SKIP interval lies between them and
// start_next - end_prev <= maxGap (1031-1032).  More than half of all intervals can be
// significant (peaks are dense in breakpoints), so nothing per-interval is materialised beyond
// three bit masks (significant / SKIP / first-of-chromosome, 1 bit per interval each, L2-resident):
//   k_sig_mask     pq[] -> bit masks                                   (the only full-length read)
//   k_runs         maximal runs of adjacent significant intervals, counted, placed and written in one pass
//   k_cands        a run opens a new candidate unless it links to the previous run (one pass as well)
//   k_cand_hdr     candidate -> {first interval, last interval, start, end}
//   k_peak_both    updatePeak over the ORIGINAL arrays: 16 lanes per candidate (a wavefront for a long one); the
//                  float AUC is summed strictly in interval order (950): lanes form the products in parallel, the
//                  additions are replayed serially through DPP
//   k_peaks        ordered compaction of the candidates passing checkPeak (916-927), into pinned host memory
// List lengths stay on the device (kernels read them through pointers).
constexpr int SW_NT = 256;
constexpr int SW_ITEMS = 8;
constexpr int SW_CHUNK = SW_NT * SW_ITEMS;  // mask words per workgroup in the chunked compactions
// Runs and candidates are few (tens of thousands) and every one costs a chain of dependent loads: one per
// thread, so the chains of a workgroup run side by side instead of eight in a row.
constexpr int RC_CHUNK = SW_NT;

// gx_find_peaks' two scalars for the kernels that read them through pointers (M_NIV, M_GENOME of the misc block)
__global__ void k_set_misc(u32* __restrict__ misc, u32 nivWord, u32 genomeWord, u64 genome, u32 n) {
  misc[nivWord] = n;
  *reinterpret_cast<u64*>(misc + genomeWord) = genome;
}

struct SweepMasks {
  u64* sig;   // bit i: interval i is significant
  u64* skip;  // bit i: interval i is a SKIP (-E) interval
  u64* brk;   // bit i: interval i is the first of its chromosome
  u32 nWords; // (n + 63) / 64, n known on the host
};

// bit i of brk for every non-empty chromosome range start (masks are zeroed by the host)
__global__ void k_brk_mask(const u32* __restrict__ chromOff, u32 nChrom, u64* __restrict__ brk) {
  for (u32 c = blockIdx.x * blockDim.x + threadIdx.x; c < nChrom; c += gridDim.x * blockDim.x) {
    u32 a = chromOff[c], b = chromOff[c + 1];
    if (b > a) atomicOr(&brk[a >> 6], 1ull << (a & 63));
  }
}

__global__ __launch_bounds__(256) void k_sig_mask(const float* __restrict__ p, const float* __restrict__ q,
                                                  const u32* __restrict__ nPtr, float thr, SweepMasks M) {
  const u32 n = *nPtr;
  const float* pq = q ? q : p;
  const u32 nw = (n + 63) >> 6;
  // four words per wave and iteration: a single 4-byte load per lane in flight reaches ~2.5 TB/s
  for (u32 w0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; w0 < nw; w0 += gridDim.x * 16) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = ((w0 + k) << 6) + lane_id();
      v[k] = i < n ? pq[i] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 sg = __ballot(v[k] > thr);
      const u64 sk = __ballot(v[k] == GX_SKIPF);
      if (lane_id() == 0 && w0 + k < nw) {
        M.sig[w0 + k] = sg;
        M.skip[w0 + k] = sk;
      }
    }
  }
}

// run starts / ends of word w: a run starts at a significant interval whose predecessor is not
// significant or which is the first of its chromosome; it ends symmetrically
__device__ __forceinline__ void run_bits(const SweepMasks& M, u32 w, u64* starts, u64* ends) {
  const u64 sg = M.sig[w], bk = M.brk[w];
  const u64 prevBit = w ? M.sig[w - 1] >> 63 : 0ull;
  const u64 nextBit = w + 1 < M.nWords ? M.sig[w + 1] & 1ull : 0ull;
  const u64 nextBrk = w + 1 < M.nWords ? M.brk[w + 1] & 1ull : 0ull;
  *starts = sg & (~((sg << 1) | prevBit) | bk);
  *ends = sg & (~((sg >> 1) | (nextBit << 63)) | ((bk >> 1) | (nextBrk << 63)));
}

// any bit of mask set in the interval-index range [lo, hi) ?
__device__ __forceinline__ bool any_bit(const u64* __restrict__ mask, u32 lo, u32 hi) {
  if (lo >= hi) return false;
  u32 w0 = lo >> 6, w1 = (hi - 1) >> 6;
  for (u32 w = w0; w <= w1; w++) {
    u64 m = mask[w];
    if (w == w0) m &= ~0ull << (lo & 63);
    if (w == w1 && ((hi & 63) != 0)) m &= (1ull << (hi & 63)) - 1;
    if (m) return true;
  }
  return false;
}

// run r opens a new candidate unless it links to run r-1 (same chromosome, no SKIP between,
// start - previous end <= maxGap).  The chromosome test is a search in the (short) offset table, not a
// walk over the mask words between the two runs: with few significant intervals (a -q run with a
// control) consecutive runs lie millions of intervals apart.
__device__ __forceinline__ bool run_is_head(const SweepMasks& M, const u64* __restrict__ skipMask, const u32* __restrict__ end,
                                            const u32* __restrict__ runStart, const u32* __restrict__ runEnd, u32 r,
                                            int maxGap, const u32* __restrict__ chromOff, u32 nChrom) {
  if (r == 0) return true;
  const u32 a = runEnd[r - 1], s = runStart[r];       // a < s
  if ((M.brk[s >> 6] >> (s & 63)) & 1ull) return true;  // first interval of a chromosome
  ChromCursor cur;
  cur.seek(chromOff, nChrom, a);
  if (s >= cur.hi) return true;                         // another chromosome
  const long long gap = (long long)end[s - 1] - (long long)end[a];  // start(s) - end(a) on one chromosome: >= 0
  if (gap != 0 && gap > (long long)maxGap) return true;
  // at most `gap` intervals (each >= 1 bp) lie between linked runs: a short scan
  if (skipMask && any_bit(skipMask, a + 1, s)) return true;  // (no SKIP intervals without -E regions)
  return false;
}

#ifndef GX_PK_SHORT
#define GX_PK_SHORT 1024
#endif
constexpr u32 PK_SHORT = GX_PK_SHORT;   // longer candidates take a wavefront each (k_peak_walk)
#ifndef GX_PK_LOADS
#define GX_PK_LOADS 8
#endif
constexpr int PK_LOADS = GX_PK_LOADS;

// candidate -> {first interval, last interval, start coordinate}: one thread per candidate, so the
// chain candRun -> runStart/runEnd -> end[] is walked by all candidates at once instead of once
// per wavefront in k_peak_walk
__global__ __launch_bounds__(256) void k_cand_hdr(SweepMasks M, const u32* __restrict__ end,
                                                  const u32* __restrict__ runStart, const u32* __restrict__ runEnd,
                                                  const u32* __restrict__ nRuns, const u32* __restrict__ candRun,
                                                  const u32* __restrict__ nCands, uint4* __restrict__ hdr,
                                                  u32* __restrict__ longList, u32* __restrict__ nLong) {
  const u32 R = *nRuns, C = *nCands;
  for (u32 c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    const u32 rFirst = candRun[c], rLast = (c + 1 < C ? candRun[c + 1] : R) - 1;
    const u32 i0 = runStart[rFirst], i1 = runEnd[rLast];  // interval span [i0, i1], both significant
    const bool atChromStart = (M.brk[i0 >> 6] >> (i0 & 63)) & 1ull;
    hdr[c] = make_uint4(i0, i1, atChromStart ? 0u : end[i0 - 1], end[i1]);
    if (i1 - i0 >= PK_SHORT) longList[atomicAdd(nLong, 1u)] = c;  // walked by a whole wavefront (k_peak_walk)
  }
}

__device__ __forceinline__ void peak_finish(u32 c, const uint4 h, float auc, u32 summitPos, float sp, float sq,
                                            float minAUC, int minLen, const u32* __restrict__ chromOff, u32 nChrom,
                                            gx_peak* __restrict__ cand, u32* __restrict__ valid) {
  const u32 peakStart = h.z, peakEnd = h.w;
  const bool ok = auc >= minAUC && (long long)peakEnd - (long long)peakStart >= (long long)minLen;  // checkPeak 916-927
  valid[c] = ok;
  if (ok) {
    ChromCursor cur;
    cur.seek(chromOff, nChrom, h.x);
    gx_peak pk;
    pk.chrom = cur.c;
    pk.start = peakStart;
    pk.end = peakEnd;
    pk.summit = summitPos;
    pk.auc = auc;
    pk.p = sp;
    pk.q = sq;
    cand[c] = pk;
  }
}

// updatePeak (943-970) for one interval [s, e) with p / q values pv / qv
struct PeakAcc {
  float auc = 0.0f, summitVal = -1.0f, sp = -1.0f, sq = -1.0f;
  u32 summitPos = 0, summitLen = 0, s = 0, origin = 0;
  __device__ __forceinline__ void step(u32 e, float pv, float qv, bool useQ, float thr) {
    const float pq = useQ ? qv : pv;
    if (pq > thr) {  // non-significant intervals inside the span only fill gaps
      auc += (float)(e - s) * (pq - thr);  // 949-950: float product, summed in order
      const u32 len = e - s;
      if (pq > summitVal) {               // 956-961
        summitVal = pq;
        sp = pv;
        sq = qv;
        summitPos = (u32)(((u64)e + s) / 2 - origin);
        summitLen = len;
      } else if (pq == summitVal && len > summitLen) {  // 962-968
        summitPos = (u32)(((u64)e + s) / 2 - origin);
        summitLen = len;
      }
    }
    s = e;
  }
};

// DPP inside a row of 16 lanes: lane K's value for the whole row (gfx90a+ row_share / row_newbcast), and the
// all-lanes maximum by four rotations.  Register-to-register: no LDS round trip, unlike __shfl.
template <int K> __device__ __forceinline__ float row_lane(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, false));
}
template <int K> struct RowSeqSum {  // acc + v[lane 0] + v[lane 1] + ... + v[lane K], in this order
  static __device__ __forceinline__ float run(float acc, float v) { return RowSeqSum<K - 1>::run(acc, v) + row_lane<K>(v); }
};
template <> struct RowSeqSum<-1> { static __device__ __forceinline__ float run(float acc, float) { return acc; } };
__device__ __forceinline__ float row_max_f(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));  // row_ror:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ u32 row_max_u(u32 v) {
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
  v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
  return v;
}

// candidates of up to PK_SHORT intervals (all but pathological ones): SIXTEEN LANES each -- a DPP row, four
// candidates per wavefront.  The lanes of a row take 16 consecutive intervals per step: loads, significance
// test, the float product (949-950) and the summit search run in parallel; only the additions into the AUC
// are replayed in interval order -- 16 dependent v_add_f32_dpp per step, lane k's product handed to the row
// by the DPP operand -- so the float sum is updatePeak's.  PK_AHEAD steps of loads are in flight together: a
// candidate of n intervals costs n / (16 PK_AHEAD) memory round trips, where one thread per candidate
// (round 1) paid n / 32 round trips AND ran every updatePeak serially (a 1,000-interval candidate held its
// wavefront for ~100 us, which is what the kernel took).
constexpr int PK_AHEAD = 4;
#ifndef GX_PK_AHEAD_V
#define GX_PK_AHEAD_V 4
#endif
// PV: `p` is the table p(V) and `q` (reinterpreted) the intervals' exact pileups V -- the sweep on the loose slots.
// (that sweep only runs on unit-weight samples whose pileups all lie within the table -- LooseCtl::bad otherwise --, so
// every V is 120 x a whole pileup below PV_WHOLE: the table's compact copy, lut + PV_LUT, sits in LDS)
__device__ __forceinline__ void load_whole_lut(float* __restrict__ hot, const float* __restrict__ lut) {
  for (u32 i = threadIdx.x; i < PV_WHOLE; i += blockDim.x) hot[i] = lut[PV_LUT + i];
  __syncthreads();
}
// (lut: the whole table p(V), V in 1/120 units -- a pileup that is not a whole number, fractional weights, is looked up there)
__device__ __forceinline__ float p_from_v(const float* __restrict__ hot, const float* __restrict__ lut, const int* __restrict__ V, u32 i) {
  const u32 v = (u32)V[i];
  const u32 c = __umulhi(v, 0x88888889u) >> 6;  // V / 120
  if (v != c * (u32)GX_UNIT) return lut[v < PV_LUT ? v : 0u];
  return hot[c < PV_WHOLE ? c : 0u];
}
// (-q on the loose slots: whole pileups within the table only -- unit weights, LooseCtl::bad otherwise)
__device__ __forceinline__ float q_from_v(const float* __restrict__ hotq, const int* __restrict__ V, u32 i) {
  const u32 c = __umulhi((u32)V[i], 0x88888889u) >> 6;  // V / 120
  return hotq[c < PV_WHOLE ? c : 0u];
}
// (PVQ, round 6: -q on the loose slots -- q as well is a function of the whole pileup, from a second LDS table, k_bh_small's)
template <bool USEQ, bool PV, bool PVQ = false>  // compile-time: a run-time test of `q` inside the loop costs the load scheduling
__device__ __forceinline__ void peak_short_body(const u32 blk, const u32 nBlk, const float* __restrict__ hot, const float* __restrict__ hotq,
                                                const uint4* __restrict__ hdr, const u32* __restrict__ end,
                                                const float* __restrict__ p, const float* __restrict__ q,
                                                const u32* __restrict__ chromOff, u32 nChrom,
                                                const u32* __restrict__ nCands, float thr, float minAUC, int minLen,
                                                gx_peak* __restrict__ cand, u32* __restrict__ valid) {
  constexpr int AH = PV ? GX_PK_AHEAD_V : PK_AHEAD;  // steps of loads in flight (the loose-slot walk adds an LDS look-up per step)
  const u32 C = *nCands;
  const int lane = lane_id(), rowBase = lane & 48, rl = lane & 15;
  const u32 rowId = (blk * 4 + (threadIdx.x >> 6)) * 4 + (u32)(lane >> 4);  // 16 rows per workgroup
  for (u32 c0 = 0; c0 < C; c0 += nBlk * 16) {  // (wave-uniform bound: every lane takes part in the shuffles)
    const u32 c = c0 + rowId;
    uint4 h = make_uint4(1u, 0u, 0u, 0u);
    if (c < C) h = hdr[c];
    const u32 i0 = h.x, i1 = h.y, peakStart = h.z;
    const bool mine = c < C && i1 - i0 < PK_SHORT;
    const u32 n = mine ? i1 - i0 + 1 : 0u;
    u32 steps = (n + 15) >> 4;
    steps = max(steps, (u32)__shfl_xor((int)steps, 16, 64));
    steps = max(steps, (u32)__shfl_xor((int)steps, 32, 64));  // the wavefront's longest candidate
    float auc = 0.0f, summitVal = -1.0f, sp = -1.0f, sq = -1.0f;
    u32 summitPos = 0, summitLen = 0;
    for (u32 st0 = 0; st0 < steps; st0 += AH) {
      u32 e[AH], sPrev[AH];
      float pv[AH], qv[AH];
      bool in[AH];
#pragma unroll
      for (int a = 0; a < AH; a++) {
        const u32 i = i0 + (st0 + a) * 16 + rl;
        in[a] = mine && i <= i1;
        e[a] = 0; sPrev[a] = 0; pv[a] = 0.0f; qv[a] = GX_SKIPF;
        if (in[a]) {
          e[a] = end[i];
          sPrev[a] = i == i0 ? peakStart : end[i - 1];
          if (PV) pv[a] = p_from_v(hot, p, reinterpret_cast<const int*>(q), i);
          else pv[a] = p[i];
          if (USEQ) qv[a] = q[i];
          if (PVQ) qv[a] = q_from_v(hotq, reinterpret_cast<const int*>(q), i);
        }
      }
#pragma unroll
      for (int a = 0; a < AH; a++) {
        if (st0 + a >= steps) break;  // wave-uniform
        float pq = USEQ || PVQ ? qv[a] : pv[a];
        const bool sg = in[a] && pq > thr;  // non-significant intervals inside the span only fill gaps
        float term = 0.0f;
        if (sg) term = (float)(e[a] - sPrev[a]) * (pq - thr);  // 949-950: float product ...
        else pq = -2.0f;
        const u64 wSg = __ballot(sg);
        if (wSg) {  // wave-uniform
          // ... summed in interval order: the other lanes hold +0.0f, which changes nothing
          auc = RowSeqSum<15>::run(auc, term);
          // summit of this chunk: maximum pq, earliest lane (956-961); among the lanes at the maximum the first
          // one with the greatest length (962-968)
          const float mx = row_max_f(pq);
          const bool atMax = sg && pq == mx;
          const u32 len = atMax ? e[a] - sPrev[a] : 0u;
          const u32 ml = row_max_u(len);
          const bool rowSg = ((u32)(wSg >> rowBase) & 0xFFFFu) != 0;
          const bool better = rowSg && mx > summitVal, longer = rowSg && mx == summitVal && ml > summitLen;
          if (__ballot(better || longer)) {  // wave-uniform: some row moves its summit (rare in a long candidate)
            const u32 rowMax = (u32)(__ballot(atMax) >> rowBase) & 0xFFFFu;
            const u32 rowLen = (u32)(__ballot(atMax && len == ml) >> rowBase) & 0xFFFFu;
            const int firstMax = rowBase + (rowMax ? __builtin_ctz(rowMax) : 0);
            const int firstLen = rowBase + (rowLen ? __builtin_ctz(rowLen) : 0);
            const u32 myPos = (u32)(((u64)e[a] + sPrev[a]) / 2 - peakStart);
            const u32 cPos = (u32)__shfl((int)myPos, firstLen, 64);
            const float spN = __shfl(pv[a], firstMax, 64), sqN = __shfl(qv[a], firstMax, 64);
            if (better) {
              summitVal = mx;
              sp = spN;
              sq = sqN;
              summitPos = cPos;
              summitLen = ml;
            } else if (longer) {
              summitPos = cPos;
              summitLen = ml;
            }
          }
        }
      }
    }
    if (mine && rl == 0) peak_finish(c, h, auc, summitPos, sp, sq, minAUC, minLen, chromOff, nChrom, cand, valid);
  }
}

// one wavefront per candidate: updatePeak (943-970) over its intervals, then checkPeak (916-927).  64 intervals
// per step, AH steps of loads in flight; the ordered float sum runs row after row (16 dependent DPP adds
// each, the running value carried from row to row by a readlane); maxima by DPP rotations inside the rows and
// readlanes across them: no LDS shuffle anywhere.
template <bool PV, bool PVQ = false>
__device__ __forceinline__ void peak_walk_body(const u32 blk, const u32 nBlk, const float* __restrict__ hot, const float* __restrict__ hotq,
                                               const uint4* __restrict__ hdr, const u32* __restrict__ end,
                                               const float* __restrict__ p, const float* __restrict__ qIn,
                                               const u32* __restrict__ chromOff, u32 nChrom,
                                               const u32* __restrict__ longList, const u32* __restrict__ nLong,
                                               float thr, float minAUC, int minLen,
                                               gx_peak* __restrict__ cand, u32* __restrict__ valid) {
  constexpr int AH = PV ? GX_PK_AHEAD_V : PK_AHEAD;
  const float* __restrict__ q = PV ? nullptr : qIn;
  const u32 L = *nLong;
  const u32 wavesPerGrid = nBlk * 4;
  const int lane = lane_id();
  auto rdf = [](float v, int l) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  for (u32 li = blk * 4 + (threadIdx.x >> 6); li < L; li += wavesPerGrid) {
    const u32 c = longList[li];
    const uint4 h = hdr[c];
    const u32 i0 = h.x, i1 = h.y, peakStart = h.z;
    float auc = 0.0f, summitVal = -1.0f, sp = -1.0f, sq = -1.0f;
    u32 summitPos = 0, summitLen = 0;
    for (u32 base = i0; base <= i1; base += 64 * AH) {
      u32 e[AH], sPrev[AH];
      float pv[AH], qv[AH];
      bool in[AH];
#pragma unroll
      for (int a = 0; a < AH; a++) {
        const u32 i = base + a * 64 + lane;
        in[a] = i <= i1;
        e[a] = 0; sPrev[a] = 0; pv[a] = 0.0f; qv[a] = GX_SKIPF;
        if (in[a]) {
          e[a] = end[i];
          sPrev[a] = i == i0 ? peakStart : end[i - 1];
          if (PV) pv[a] = p_from_v(hot, p, reinterpret_cast<const int*>(qIn), i);
          else pv[a] = p[i];
          if (q) qv[a] = q[i];
          if (PVQ) qv[a] = q_from_v(hotq, reinterpret_cast<const int*>(qIn), i);
        }
      }
#pragma unroll
      for (int a = 0; a < AH; a++) {
        if (base + a * 64 > i1) break;  // wave-uniform
        float pq = q || PVQ ? qv[a] : pv[a];
        const bool sg = in[a] && pq > thr;       // non-significant intervals inside the span only fill gaps
        float term = 0.0f;
        if (sg) term = (float)(e[a] - sPrev[a]) * (pq - thr);  // 949-950: float product ...
        else pq = -2.0f;
        if (__ballot(sg)) {  // wave-uniform
          // ... summed in lane order, row after row; the other lanes hold +0.0f, which changes nothing
          float acc = RowSeqSum<15>::run(auc, term);
          acc = RowSeqSum<15>::run(rdf(acc, 0), term);
          acc = RowSeqSum<15>::run(rdf(acc, 16), term);
          acc = RowSeqSum<15>::run(rdf(acc, 32), term);
          auc = rdf(acc, 48);
          // summit of this chunk: maximum pq, earliest lane (956-961); among the lanes at the maximum
          // the first one with the greatest length (962-968)
          float mx = row_max_f(pq);
          mx = fmaxf(fmaxf(rdf(mx, 0), rdf(mx, 16)), fmaxf(rdf(mx, 32), rdf(mx, 48)));
          const bool atMax = sg && pq == mx;
          const u32 len = atMax ? e[a] - sPrev[a] : 0u;
          u32 ml = row_max_u(len);
          ml = max(max((u32)__builtin_amdgcn_readlane((int)ml, 0), (u32)__builtin_amdgcn_readlane((int)ml, 16)),
                   max((u32)__builtin_amdgcn_readlane((int)ml, 32), (u32)__builtin_amdgcn_readlane((int)ml, 48)));
          if (mx > summitVal || (mx == summitVal && ml > summitLen)) {  // wave-uniform
            const int firstMax = __builtin_ctzll(__ballot(atMax));
            const int firstLen = __builtin_ctzll(__ballot(atMax && len == ml));
            const u32 myPos = (u32)(((u64)e[a] + sPrev[a]) / 2 - peakStart);
            const u32 cPos = (u32)__builtin_amdgcn_readlane((int)myPos, firstLen);
            if (mx > summitVal) {
              summitVal = mx;
              sp = rdf(pv[a], firstMax);
              sq = rdf(qv[a], firstMax);
            }
            summitPos = cPos;
            summitLen = ml;
          }
        }
      }
    }
    if (lane == 0) peak_finish(c, h, auc, summitPos, sp, sq, minAUC, minLen, chromOff, nChrom, cand, valid);
  }
}

// The two walks in ONE launch (they share nothing but the candidate headers): the first workgroups take the
// long candidates, the other nShort the short ones -- the few long candidates (one wavefront each, a chain of dependent
// round trips) run beside the short ones instead of behind them.
template <bool USEQ, bool PV, bool PVQ = false>
__global__ __launch_bounds__(256) void k_peak_both(u32 nShort, const uint4* __restrict__ hdr, const u32* __restrict__ end,
                                                   const float* __restrict__ p, const float* __restrict__ q,
                                                   const u32* __restrict__ chromOff, u32 nChrom,
                                                   const u32* __restrict__ nCands, const u32* __restrict__ longList,
                                                   const u32* __restrict__ nLong, float thr, float minAUC, int minLen,
                                                   gx_peak* __restrict__ cand, u32* __restrict__ valid,
                                                   const float* __restrict__ qLut = nullptr) {
  static_assert(!PVQ || (PV && !USEQ), "q by pileup rides the sweep on the loose slots");
  __shared__ float hot[PV ? PV_WHOLE : 1];
  __shared__ float hotq[PVQ ? PV_WHOLE : 1];
  const u32 nWalk = gridDim.x - nShort;  // (the long candidates' workgroups come first in the grid: they are the long pole)
#ifndef GX_PK_NO_EARLY_EXIT
  if (blockIdx.x < nWalk && blockIdx.x * 4 >= *nLong) return;  // (no long candidate for this workgroup: not even the table)
#endif
  if (PVQ)
    for (u32 i = threadIdx.x; i < PV_WHOLE; i += blockDim.x) hotq[i] = qLut[i];
  if (PV) load_whole_lut(hot, p);
  if (blockIdx.x >= nWalk)
    peak_short_body<USEQ, PV, PVQ>(blockIdx.x - nWalk, nShort, hot, hotq, hdr, end, p, q, chromOff, nChrom, nCands, thr, minAUC, minLen, cand, valid);
  else
    peak_walk_body<PV, PVQ>(blockIdx.x, nWalk, hot, hotq, hdr, end, p, q, chromOff, nChrom, longList, nLong, thr, minAUC, minLen, cand, valid);
}

// ---- the sweep's three ordered compactions, each in ONE pass -------------------------------------------------
// (round 2: count kernel -> k_scan_small -> write kernel, three launches each.)  A chunk counts its items, learns
// how many the chunks before it hold through a decoupled look-back, and writes.  The look-back granules carry a
// GENERATION number (one per gx_find_peaks call), so the arrays are never cleared: an entry of an earlier call
// reads as "nothing yet".   granule = [63:62] flag (1 aggregate, 2 inclusive prefix)  [61:38] generation  [37:0] value
// Forward progress as for lookback_excl: persistent workgroups (grid <= co-resident), chunks taken round-robin.
constexpr int LBG_SHIFT = 38;
__device__ __forceinline__ u64 lookback_gen(u64* lb, u32 id, u64 aggregate, u32 gen, u32* st) {
  const u64 tag = (u64)(gen & 0xFFFFFFu) << LBG_SHIFT, vmask = (1ull << LBG_SHIFT) - 1;
  u64 excl = 0;
  if (id > 0) {
    if (lane_id() == 0)
      __hip_atomic_store(&lb[id], LB_AGG | tag | (aggregate & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int look = (int)id - 1;
    u32 spins = 0;
    bool done = false;
    while (!done) {
      const int idx = look - lane_id();  // lane 0 = nearest predecessor
      u64 v = idx >= 0 ? __hip_atomic_load(&lb[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (LB_INC | tag);
      if (((v >> LBG_SHIFT) & 0xFFFFFFu) != (gen & 0xFFFFFFu)) v = 0;  // another call's entry: nothing yet
      const u64 flag = v >> 62;
      const u64 invalidMask = __ballot(flag == 0), incMask = __ballot(flag == 2);
      const int firstInvalid = invalidMask ? __builtin_ctzll(invalidMask) : 64;
      const int firstInc = incMask ? __builtin_ctzll(incMask) : 64;
      if (firstInc < firstInvalid) {
        excl += wave_sum(lane_id() <= firstInc ? (v & vmask) : 0ull);
        done = true;
      } else if (firstInvalid > 0) {
        excl += wave_sum(lane_id() < firstInvalid ? (v & vmask) : 0ull);
        look -= firstInvalid;
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
  excl &= vmask;
  if (lane_id() == 0)
    __hip_atomic_store(&lb[id], LB_INC | tag | ((excl + aggregate) & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// runs of adjacent significant intervals: starts and ends, counted, placed and written in one pass
__global__ __launch_bounds__(SW_NT) void k_runs(SweepMasks M, u64* __restrict__ lbS, u64* __restrict__ lbE, u32 gen,
                                                u32* __restrict__ runStart, u32* __restrict__ runEnd, u32 cap,
                                                u32* __restrict__ nRuns /* min(total, cap) */, u32* __restrict__ nRunsHost /* total */,
                                                u32* __restrict__ zeroMe, u64* __restrict__ zeroBp, u32* __restrict__ st) {
  __shared__ u32 scratch[8];
  __shared__ u64 s_base[2];
  const u32 nChunks = (M.nWords + SW_CHUNK - 1) / SW_CHUNK;
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 w0 = id * SW_CHUNK + threadIdx.x * SW_ITEMS;
    u64 stb[SW_ITEMS], enb[SW_ITEMS];
    u32 a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < SW_ITEMS; k++) {
      stb[k] = 0;
      enb[k] = 0;
      if (w0 + k < M.nWords) run_bits(M, w0 + k, &stb[k], &enb[k]);
      a += __popcll(stb[k]);
      b += __popcll(enb[k]);
    }
    // ONE scan and one look-back (round 3; two of each before): starts and ends alternate, so the ends before a word
    // are the starts before it minus one if a run is open across its first bit -- that bit significant, but not a start
    u32 totA;
    u32 oa = block_excl_scan<u32, SW_NT>(a, scratch, &totA);
    (void)b;
    if (threadIdx.x < 64) {
      const u64 ea = lookback_gen(lbS, id, totA, gen, st);
      if (threadIdx.x == 0) {
        s_base[0] = ea;
        if (id == nChunks - 1) {
          const u32 total = (u32)ea + totA;
          *nRuns = total > cap ? cap : total;
          *nRunsHost = total;
          *zeroMe = 0;
          *zeroBp = 0;   // (k_peaks' sum of the peaks' lengths)
        }
      }
    }
    __syncthreads();
    oa += (u32)s_base[0];
    u32 ob = oa;
    if (w0 < M.nWords) {
      const u64 sig0 = M.sig[w0];
      ob -= (u32)(sig0 & ~stb[0] & 1ull);
    }
#ifdef GX_RUNS_TWO_SCANS
    {
      u32 totB;
      ob = block_excl_scan<u32, SW_NT>(b, scratch, &totB);
      __shared__ u64 s_b;
      if (threadIdx.x < 64) {
        const u64 eb = lookback_gen(lbE, id, totB, gen, st);
        if (threadIdx.x == 0) s_b = eb;
      }
      __syncthreads();
      ob += (u32)s_b;
    }
#endif
#pragma unroll
    for (int k = 0; k < SW_ITEMS; k++) {
      u64 x = stb[k];
      while (x) {
        const int bit = __builtin_ctzll(x);
        x &= x - 1;
        if (oa < cap) runStart[oa] = ((w0 + k) << 6) + bit;
        oa++;
      }
      x = enb[k];
      while (x) {
        const int bit = __builtin_ctzll(x);
        x &= x - 1;
        if (ob < cap) runEnd[ob] = ((w0 + k) << 6) + bit;
        ob++;
      }
    }
    __syncthreads();
  }
}

// candidates: the runs that open one (run_is_head), counted, placed and listed in one pass
__global__ __launch_bounds__(SW_NT) void k_cands(SweepMasks M, const u64* __restrict__ skipMask, const u32* __restrict__ end,
                                                 const u32* __restrict__ runStart, const u32* __restrict__ runEnd,
                                                 const u32* __restrict__ nRuns, int maxGap, const u32* __restrict__ chromOff,
                                                 u32 nChrom, u64* __restrict__ lb, u32 gen, u32* __restrict__ candRun,
                                                 u32* __restrict__ nCands, u32* __restrict__ st) {
  __shared__ u32 scratch[8];
  __shared__ u64 s_base;
  const u32 R = *nRuns;
  const u32 nChunks = (R + RC_CHUNK - 1) / RC_CHUNK;
  if (nChunks == 0 && blockIdx.x == 0 && threadIdx.x == 0) *nCands = 0;
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 r = id * RC_CHUNK + threadIdx.x;
    const u32 keep = r < R ? (u32)run_is_head(M, skipMask, end, runStart, runEnd, r, maxGap, chromOff, nChrom) : 0u;
    u32 tot;
    u32 o = block_excl_scan<u32, SW_NT>(keep, scratch, &tot);
    if (threadIdx.x < 64) {
      const u64 e = lookback_gen(lb, id, tot, gen, st);
      if (threadIdx.x == 0) {
        s_base = e;
        if (id == nChunks - 1) *nCands = (u32)e + tot;
      }
    }
    __syncthreads();
    if (keep) candRun[o + (u32)s_base] = r;
    __syncthreads();
  }
}

// the candidates that passed checkPeak (916-927), in order, straight into pinned host memory (the workgroup's peaks are packed in LDS and leave as one contiguous run of dwords: 64 consecutive dwords of a wavefront make a few full-size PCIe writes, a 28-byte record per lane seven small ones)
__global__ __launch_bounds__(SW_NT) void k_peaks(const gx_peak* __restrict__ cand, const u32* __restrict__ valid,
                                                 const u32* __restrict__ nHeads, u64* __restrict__ lb, u32 gen,
                                                 gx_peak* __restrict__ peaks /* pinned host memory */, u32* __restrict__ nPeaks,
                                                 u32* __restrict__ nPeaksHost, u32* __restrict__ st,
                                                 u64* __restrict__ bpAcc /* the peaks' total length (callPeaks 925); k_runs left it at zero */) {
  constexpr int PW = sizeof(gx_peak) / 4;
  __shared__ u32 scratch[8];
  __shared__ u32 stage[RC_CHUNK * PW];
  __shared__ u64 s_base;
  const u32 H = *nHeads;
  const u32 nChunks = (H + RC_CHUNK - 1) / RC_CHUNK;
  u64 bp = 0;  // (the host summed the lengths over the pinned records: 1.7 MB of reads behind the kernel)
  if (nChunks == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    *nPeaks = 0;
    *nPeaksHost = 0;
  }
  for (u32 id = blockIdx.x; id < nChunks; id += gridDim.x) {
    const u32 h = id * RC_CHUNK + threadIdx.x;
    const u32 keep = h < H ? valid[h] : 0u;
    u32 tot;
    const u32 r = block_excl_scan<u32, SW_NT>(keep, scratch, &tot);
    if (keep) {
      const u32* src = reinterpret_cast<const u32*>(cand + h);
#pragma unroll
      for (int k = 0; k < PW; k++) stage[r * PW + k] = src[k];
      bp += cand[h].end - cand[h].start;
    }
    if (threadIdx.x < 64) {
      const u64 e = lookback_gen(lb, id, tot, gen, st);
      if (threadIdx.x == 0) {
        s_base = e;
        if (id == nChunks - 1) {
          *nPeaks = (u32)e + tot;
          *nPeaksHost = (u32)e + tot;
        }
      }
    }
    __syncthreads();
    u32* dst = reinterpret_cast<u32*>(peaks + (u32)s_base);
    for (u32 i = threadIdx.x; i < tot * PW; i += SW_NT) dst[i] = stage[i];
    __syncthreads();
  }
  // The peaks' total length (callPeaks 925; the host used to sum it over 1.7 MB of pinned records behind the kernel):
  // one atomic per workgroup that wrote peaks; the mail behind this kernel takes the sum along.
  // (Sending the mail itself from the workgroup that finishes last was tried: the peaks are posted writes to HOST
  // memory from many CUs, and "my stores have completed" on one CU does not order them ahead of another CU's write of
  // the sequence number -- a two-process test saw a stale record once in ten runs.  The end of a kernel does.  A
  // ticket per workgroup, for the last one to hand the sum over, cost 16 us of same-address atomics.)
  __shared__ unsigned long long s_bp;
  if (threadIdx.x == 0) s_bp = 0;
  __syncthreads();
  bp = wave_sum(bp);
  if (lane_id() == 0 && bp) atomicAdd(&s_bp, (unsigned long long)bp);
  __syncthreads();
  if (threadIdx.x == 0 && s_bp) atomicAdd((unsigned long long*)bpAcc, s_bp);
}

}  // namespace gx
