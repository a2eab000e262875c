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
// a union breakpoint j is slice_begin + popcount(bits before j).  Output positions again come
// from the decoupled look-back.
#pragma once
#include "gx_stats.h"

namespace gx {

constexpr int MG_WORDS_ = TILE / 32;
constexpr int MG_NT = MG_WORDS_ < 256 ? MG_WORDS_ : 256;
constexpr int MG_WORDS = TILE / 32;        // 512 bitmap words per input
constexpr int MG_WPT = MG_WORDS / MG_NT;   // 2 consecutive words per thread

struct RleIn {   // run-length pileup from k_tile
  const u32* end;
  const int* v;
  const u32* tileOff;
};

struct Merge2Out {   // loose slots: tile t writes at A.tileOff[t] + B.tileOff[t] (a union is never longer)
  u32* end;
  int* exptV;        // treatment pileup (1/120 units, V_MARK inside -E regions)
  int* ctrlV;        // control pileup of the covering control interval
  u32* tileCount;    // [nTiles]
};

__device__ __forceinline__ float expt_val(int v, bool* neg) {
  if (v == V_MARK) { *neg = false; return 0.0f; }  // excluded region: 2248 / 2273
  return getval(v, neg);
}

__device__ __forceinline__ float ctrl_net(int v, float factor, float lambda, bool* neg) {
  if (v == V_MARK) { *neg = false; return GX_SKIPF; }  // excluded region: 2124 / 2141
  float val = factor * getval(v, neg);  // 2107 / 2118: float product
  return val > lambda ? val : lambda;   // MAX(val, lambda)
}

// No inter-workgroup dependency (like k_tile): counts go to k_scan_counts, packing to k_pack_pairs.
__global__ __launch_bounds__(MG_NT) void k_merge2(RleIn A, RleIn B, const Scalars* __restrict__ sc,
                                                  const u32* __restrict__ tileChrom, const DChrom* __restrict__ chroms,
                                                  u32 nTiles, Merge2Out out, u32* __restrict__ st) {
  __shared__ u32 bmA[MG_WORDS], bmB[MG_WORDS], bmC[MG_WORDS];
  __shared__ u32 scratch[8];
  const float factor = sc->factor, lambda = sc->lambda;
  u32 neg = 0;
  for (u32 t = blockIdx.x; t < nTiles; t += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < MG_WORDS; i += MG_NT) { bmA[i] = 0; bmB[i] = 0; bmC[i] = 0; }
    __syncthreads();
    const u32 ci = tileChrom[t];
    const DChrom c = chroms[ci];
    const bool active = chrom_active(c);
    const u32 tl = t - c.tileBase, pos0 = tl << TB;
    const bool lastTile = tl + 1 == c.nTiles;
    u32 a0 = A.tileOff[t], a1 = A.tileOff[t + 1], b0 = B.tileOff[t], b1 = B.tileOff[t + 1];
    const u32 slot = a0 + b0;
    if (!active) { a1 = a0; b1 = b0; }
    const u32 a1c = (lastTile && a1 > a0) ? a1 - 1 : a1;  // the chromosome-closing interval is handled apart
    const u32 b1c = (lastTile && b1 > b0) ? b1 - 1 : b1;
    for (u32 i = a0 + threadIdx.x; i < a1c; i += MG_NT) {
      u32 off = A.end[i] - pos0;
      atomicOr(&bmA[off >> 5], 1u << (off & 31));
    }
    for (u32 i = b0 + threadIdx.x; i < b1c; i += MG_NT) {
      u32 off = B.end[i] - pos0;
      bool ng1, ng2;
      float here = ctrl_net(B.v[i], factor, lambda, &ng1);
      float next = ctrl_net(B.v[i + 1], factor, lambda, &ng2);  // i + 1 exists: at least the closing interval
      neg |= ng1 | ng2;
      atomicOr(&bmC[off >> 5], 1u << (off & 31));
      if (here != next) atomicOr(&bmB[off >> 5], 1u << (off & 31));  // 2122: net != MAX(val, lambda)
    }
    __syncthreads();
    u32 wU[MG_WPT], wA[MG_WPT], wC[MG_WPT];
    u32 cU = 0, cA = 0, cC = 0;
#pragma unroll
    for (int k = 0; k < MG_WPT; k++) {
      int w = threadIdx.x * MG_WPT + k;
      wA[k] = bmA[w];
      wC[k] = bmC[w];
      wU[k] = wA[k] | bmB[w];
      cU += __popc(wU[k]);
      cA += __popc(wA[k]);
      cC += __popc(wC[k]);
    }
    u32 tU, tA, tC;
    u32 exU = block_excl_scan<u32, MG_NT>(cU, scratch, &tU);
    u32 exA = block_excl_scan<u32, MG_NT>(cA, scratch, &tA);
    u32 exC = block_excl_scan<u32, MG_NT>(cC, scratch, &tC);
    if (threadIdx.x == 0) out.tileCount[t] = active ? tU + (lastTile ? 1u : 0u) : 0u;
    if (active) {  // block-uniform
      u32 o = slot + exU;
#pragma unroll
      for (int k = 0; k < MG_WPT; k++) {
        u32 bits = wU[k];
        while (bits) {
          int b = __ffs(bits) - 1;
          bits &= bits - 1;
          u32 below = (1u << b) - 1;
          out.end[o] = pos0 + (threadIdx.x * MG_WPT + k) * 32 + b;
          out.exptV[o] = A.v[a0 + exA + __popc(wA[k] & below)];
          out.ctrlV[o] = B.v[b0 + exC + __popc(wC[k] & below)];
          o++;
        }
        exA += __popc(wA[k]);
        exC += __popc(wC[k]);
      }
      if (lastTile && threadIdx.x == 0) {  // 1779-1788 at the chromosome end: both pileups close at len
        u32 oc = slot + tU;
        out.end[oc] = c.len;
        out.exptV[oc] = A.v[a1 - 1];
        out.ctrlV[oc] = B.v[b1 - 1];
      }
    }
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
                                           const double* __restrict__ logE, const CtrlEntry* __restrict__ ctab, bool* neg) {
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
  double pv;
  if (sl == 0.0)
    pv = (double)expt < ml ? 0.0 : (double)FLT_MAX;
  else
    pv = -pnorm_upper_log((le - ml) / sl) / 2.30258509299404568402;
  return pv > (double)FLT_MAX ? FLT_MAX : (float)pv;
}

struct PackPairsIn {
  const u32* looseEnd;
  const int* looseE;
  const int* looseC;
  const u32* slotA;    // A.tileOff
  const u32* slotB;    // B.tileOff
  const u32* tileOff;  // tight offsets
};

// loose slots -> tight (end, expt, ctrl, p); one wavefront per tile
__global__ __launch_bounds__(256) void k_pack_pairs(PackPairsIn in, u32 nTiles, const Scalars* __restrict__ sc,
                                                    const double* __restrict__ logE, const CtrlEntry* __restrict__ ctab,
                                                    u32* __restrict__ end, float* __restrict__ expt,
                                                    float* __restrict__ ctrl, float* __restrict__ p,
                                                    u32* __restrict__ st) {
  const float factor = sc->factor, lambda = sc->lambda;
  const int wv = threadIdx.x >> 6, lane = lane_id();
  u32 neg = 0;
  for (u32 t = blockIdx.x * 4 + wv; t < nTiles; t += gridDim.x * 4) {
    const u32 src = in.slotA[t] + in.slotB[t], dst = in.tileOff[t], n = in.tileOff[t + 1] - dst;
    for (u32 i = lane; i < n; i += 64) {
      bool ng;
      float e, c;
      float pv = pval_pair(in.looseE[src + i], in.looseC[src + i], &e, &c, factor, lambda, logE, ctab, &ng);
      neg |= ng;
      end[dst + i] = in.looseEnd[src + i];
      expt[dst + i] = e;
      ctrl[dst + i] = c;
      p[dst + i] = pv;
    }
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
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

__global__ __launch_bounds__(MG_NT) void k_mergeN(RepSet S, const u32* __restrict__ tileChrom,
                                                  const DChrom* __restrict__ chroms, u32 nTiles,
                                                  MergeNOut out, u32* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) u32 bm[];  // S.n bitmaps of MG_WORDS words
  __shared__ u32 scratch[8];
  const int n = S.n;
  for (u32 t = blockIdx.x; t < nTiles; t += gridDim.x) {  // persistent, round-robin
    __syncthreads();
    for (int i = threadIdx.x; i < n * MG_WORDS; i += MG_NT) bm[i] = 0;
    __syncthreads();
    const u32 ci = tileChrom[t];
    const DChrom c = chroms[ci];
    const u32 tl = t - c.tileBase, pos0 = tl << TB;
    const bool lastTile = tl + 1 == c.nTiles;
    bool any = false;
    for (int r = 0; r < n; r++) {
      if (!S.r[r].present[ci]) continue;
      any = true;
      u32 a0 = S.r[r].tileOff[t], a1 = S.r[r].tileOff[t + 1];
      u32 a1c = (lastTile && a1 > a0) ? a1 - 1 : a1;
      for (u32 i = a0 + threadIdx.x; i < a1c; i += MG_NT) {
        u32 off = S.r[r].end[i] - pos0;
        atomicOr(&bm[r * MG_WORDS + (off >> 5)], 1u << (off & 31));
      }
    }
    __syncthreads();
    u32 wU[MG_WPT] = {};
    u32 cU = 0;
#pragma unroll
    for (int k = 0; k < MG_WPT; k++) {
      int w = threadIdx.x * MG_WPT + k;
      for (int r = 0; r < n; r++) wU[k] |= bm[r * MG_WORDS + w];
      cU += __popc(wU[k]);
    }
    u32 tU;
    u32 exU = block_excl_scan<u32, MG_NT>(cU, scratch, &tU);
    // per replicate: intervals ending before this thread's first word (rank base)
    u32 exR[MAX_REPS];
    for (int r = 0; r < n; r++) {
      u32 cr = 0;
#pragma unroll
      for (int k = 0; k < MG_WPT; k++) cr += __popc(bm[r * MG_WORDS + threadIdx.x * MG_WPT + k]);
      u32 tr;
      exR[r] = block_excl_scan<u32, MG_NT>(cr, scratch, &tr);
    }
    u32 slot = 0;
    for (int r = 0; r < n; r++) slot += S.r[r].tileOff[t];
    if (threadIdx.x == 0) out.tileCount[t] = any ? tU + (lastTile ? 1u : 0u) : 0u;
    if (any) {  // block-uniform
    u32 o = slot + exU;
#pragma unroll
    for (int k = 0; k < MG_WPT; k++) {
      const int w = threadIdx.x * MG_WPT + k;
      u32 bits = wU[k];
      while (bits) {
        int b = __ffs(bits) - 1;
        bits &= bits - 1;
        u32 below = (1u << b) - 1;
        double sum = 0.0;
        int df = 0;
        for (int r = 0; r < n; r++) {  // multPval 570-574, replicate order
          if (!S.r[r].present[ci]) continue;
          u32 idx = S.r[r].tileOff[t] + exR[r] + __popc(bm[r * MG_WORDS + w] & below);
          float pv = S.r[r].p[idx];
          if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
        }
        if (df > 400) atomicOr(st, ST_BAD_DF);
        out.end[o] = pos0 + w * 32 + b;
        out.p[o] = fisher_combine(sum, df);
        o++;
      }
      for (int r = 0; r < n; r++) exR[r] += __popc(bm[r * MG_WORDS + w]);
    }
    if (lastTile && threadIdx.x == 0) {
      double sum = 0.0;
      int df = 0;
      for (int r = 0; r < n; r++) {
        if (!S.r[r].present[ci]) continue;
        float pv = S.r[r].p[S.r[r].tileOff[t + 1] - 1];
        if (pv != GX_SKIPF) { sum += (double)pv; df += 2; }
      }
      u32 oc = slot + tU;
      out.end[oc] = c.len;
      out.p[oc] = fisher_combine(sum, df);
    }
    }
  }
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

// scalar device functions exposed for the numerics tests (gx_selftest)
__global__ __launch_bounds__(256) void k_selftest(int what, const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, u32 n) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    bool ng = false;
    switch (what) {
      case 0: out[i] = log10f_host(a[i]); break;
      case 1: out[i] = calc_pval(a[i], b[i]); break;
      case 2: out[i] = getval(__float_as_int(a[i]), &ng); break;
      case 3: out[i] = fisher_combine((double)a[i], (int)b[i]); break;
      default: out[i] = 0.0f;
    }
  }
}

}  // namespace gx
