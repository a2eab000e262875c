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
constexpr u32 PV_LUT = 1u << 18;

__device__ __forceinline__ float pval_of_v(int v, float lambda, double ml, double sl, float* valOut, bool* neg) {
  float val = getval(v, neg);
  *valOut = val;
  if (lambda == 0.0f) return val == 0.0f ? 0.0f : FLT_MAX;  // calcPval 1631-1632
  return val == 0.0f ? 0.0f : pval_given(val, ml, sl);
}

__global__ __launch_bounds__(256) void k_pval_lut(const Scalars* __restrict__ sc, float* __restrict__ lutP) {
  const float lambda = sc->lambda;
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  for (u32 v = blockIdx.x * 256 + threadIdx.x; v < PV_LUT; v += gridDim.x * 256) {
    float val;
    bool ng;
    lutP[v] = pval_of_v((int)v, lambda, ml, sl, &val, &ng);
  }
}

__global__ __launch_bounds__(256) void k_pval_const(const int* __restrict__ ivV, const u32* __restrict__ nIvPtr,
                                                    const Scalars* __restrict__ sc, const float* __restrict__ lutP,
                                                    float* __restrict__ pOut, float* __restrict__ exptOut,
                                                    float* __restrict__ ctrlOut, u32* __restrict__ st) {
  const u32 n = *nIvPtr;
  const float lambda = sc->lambda;
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  u32 neg = 0;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int v = ivV[i];
    bool ng = false;
    float val, p;
    if (v == V_MARK) {  // inside an excluded region: treatment 0.0f (2248), control SKIP (1871) -> p SKIP (1629)
      val = 0.0f;
      p = GX_SKIPF;
    } else if ((u32)v < PV_LUT) {
      val = getval(v, &ng);
      p = lutP[v];
    } else
      p = pval_of_v(v, lambda, ml, sl, &val, &ng);
    neg |= ng;
    pOut[i] = p;
    if (exptOut) exptOut[i] = val;
    if (ctrlOut) ctrlOut[i] = v == V_MARK ? GX_SKIPF : lambda;
  }
  if (neg) atomicOr(st, ST_NEG_PILE);
}

// ---- Benjamini-Hochberg: computeQval (352-401) / saveQval (212-250) -----------------------
// hashPval (300-327): genome-wide multiset {distinct float p -> total bp}.  Keys are the
// float's bits (p >= +0, so unsigned bit order == numeric order; -0 is folded into +0 as C's
// == does, 282).  Each workgroup first aggregates into an LDS table so the global table sees
// one atomic per (workgroup, distinct value).
constexpr u32 EMPTY_KEY = 0xFFFFFFFFu;  // a NaN pattern: never a p-value
constexpr int BH_LT = 2048;             // LDS table entries
constexpr int BH_LPROBE = 8;

__device__ __forceinline__ u32 bh_hash(u32 k) {
  k *= 2654435761u;
  return k ^ (k >> 15);
}

__device__ __forceinline__ void bh_global_add(u32* __restrict__ gKeys, u64* __restrict__ gLens, u32 capMask,
                                              u32 key, u64 len, u32* st) {
  u32 h = bh_hash(key) & capMask;
  for (u32 probe = 0; probe <= capMask; probe++) {
    u32 old = gKeys[h];
    if (old != key) {
      if (old != EMPTY_KEY) { h = (h + 1) & capMask; continue; }
      old = atomicCAS(&gKeys[h], EMPTY_KEY, key);
      if (old != EMPTY_KEY && old != key) { h = (h + 1) & capMask; continue; }
    }
    atomicAdd(&gLens[h], len);
    return;
  }
  atomicOr(st, ST_HASH_FULL);
}

__global__ __launch_bounds__(256) void k_bh_hist(const u32* __restrict__ end, const float* __restrict__ p,
                                                 const u32* __restrict__ chromOff, u32 nChrom,
                                                 const u32* __restrict__ nPtr, u32* __restrict__ gKeys,
                                                 u64* __restrict__ gLens, u32 capMask, u32* __restrict__ st) {
  __shared__ u32 lk[BH_LT];
  __shared__ u64 ll[BH_LT];
  for (int i = threadIdx.x; i < BH_LT; i += 256) { lk[i] = EMPTY_KEY; ll[i] = 0; }
  __syncthreads();
  const u32 n = *nPtr;
  const u32 per = (n + gridDim.x - 1) / gridDim.x;
  const u32 b0 = blockIdx.x * per, b1 = min(n, b0 + per);
  ChromCursor cur;
  for (u32 i = b0 + threadIdx.x; i < b1; i += 256) {
    float pv = p[i];
    if (pv == GX_SKIPF) continue;  // 319
    cur.seek(chromOff, nChrom, i);
    u32 s = i == cur.lo ? 0 : end[i - 1];
    u64 len = end[i] - s;
    u32 key = pv == 0.0f ? 0u : __float_as_uint(pv);
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
    if (!placed) bh_global_add(gKeys, gLens, capMask, key, len, st);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH_LT; i += 256)
    if (lk[i] != EMPTY_KEY) bh_global_add(gKeys, gLens, capMask, lk[i], ll[i], st);
}

// occupied slots -> (key, slot) pairs, arbitrary order (sorted afterwards)
__global__ __launch_bounds__(256) void k_bh_compact(const u32* __restrict__ gKeys, u32 cap, u32* __restrict__ outKeys,
                                                    u32* __restrict__ outSlot, u32* __restrict__ counter) {
  for (u32 s = blockIdx.x * 256 + threadIdx.x; s < cap; s += gridDim.x * 256) {
    u32 k = gKeys[s];
    if (k != EMPTY_KEY) {
      u32 j = atomicAdd(counter, 1u);
      outKeys[j] = k;
      outSlot[j] = s;
    }
  }
}

// insert (key, bp) pairs gathered from other ranks
__global__ __launch_bounds__(256) void k_bh_insert(const u32* __restrict__ keys, const u64* __restrict__ lens, u32 n,
                                                   u32* __restrict__ gKeys, u64* __restrict__ gLens, u32 capMask,
                                                   u32* __restrict__ st) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    bh_global_add(gKeys, gLens, capMask, keys[i], lens[i], st);
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

__global__ __launch_bounds__(1024) void k_qtable(const u32* __restrict__ keys, const u32* __restrict__ slots,
                                                 const u64* __restrict__ gLens, u32 D, const u64* __restrict__ genomeLenPtr,
                                                 float* __restrict__ qOfSlot, float* __restrict__ raw /* scratch [D] */,
                                                 u32* __restrict__ allOne) {
  __shared__ u64 s64[20];
  __shared__ float sf[20];
  const float logN = -log10f_host((float)*genomeLenPtr);
  const u32 per = (D + 1023) / 1024;
  const u32 r0 = min(D, threadIdx.x * per), r1 = min(D, r0 + per);  // reversed indices [r0, r1)
  u64 sum = 0;
  for (u32 r = r0; r < r1; r++) sum += gLens[slots[D - 1 - r]];
  u64 k = 1 + block_excl_scan_op<u64, 1024>(sum, 0ull, s64, OpAddU64());
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

// per interval: q = table[p] (lookup 196-206), SKIP stays SKIP (237-238)
__global__ __launch_bounds__(256) void k_qlookup(const float* __restrict__ p, const u32* __restrict__ nPtr,
                                                 const u32* __restrict__ gKeys, const float* __restrict__ qOfSlot,
                                                 u32 capMask, float* __restrict__ q) {
  const u32 n = *nPtr;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float pv = p[i];
    if (pv == GX_SKIPF) { q[i] = GX_SKIPF; continue; }
    u32 key = pv == 0.0f ? 0u : __float_as_uint(pv);
    u32 h = bh_hash(key) & capMask;
    while (gKeys[h] != key) h = (h + 1) & capMask;  // every p was inserted
    q[i] = qOfSlot[h];
  }
}

// ---- peak sweep: callPeaks (977-1069) ---------------------------------------------------------
// Only significant intervals (pq > thr, strict, 1015) and SKIP intervals matter: two significant
// intervals belong to one candidate iff no SKIP interval lies between them and
// start_next - end_prev <= maxGap (1031-1032).
//   pass 1  k_sweep_count / k_sweep_write : ordered compaction of {significant, SKIP} intervals
//   pass 2  k_heads_count / k_heads_write : ordered list of candidate heads
//   pass 3  k_peak_walk     : one wavefront per candidate; the float AUC is summed strictly in
//                             interval order (950) -- lanes load and form the products in
//                             parallel, the additions are replayed serially through shuffles
//   pass 4  k_peaks_count / k_peaks_write : ordered compaction of the candidates passing checkPeak (916-927)
// Ordered compactions are count -> k_scan_small (chunk counts) -> write; list lengths stay on the
// device (kernels read them through pointers), only the final peak count travels to the host.
struct SweepList {
  u32* chrom;
  u32* start;
  u32* end;
  float* p;
  float* q;
  u32* sig;   // 1 significant, 0 SKIP marker
  u32* count; // number of entries
};

constexpr int SW_NT = 256;
constexpr int SW_ITEMS = 8;
constexpr int SW_CHUNK = SW_NT * SW_ITEMS;

// exclusive scan of a short u32 array (chunk counts) by one workgroup; total -> *total
__global__ __launch_bounds__(1024) void k_scan_small(const u32* __restrict__ in, const u32* __restrict__ nPtr, u32 nMax,
                                                     u32 chunk, u32* __restrict__ out, u32* __restrict__ total) {
  __shared__ u32 scratch[20];
  // n items = ceil(*nPtr / chunk) when nPtr is given (a device-side count), else nMax
  u32 n = nPtr ? (*nPtr + chunk - 1) / chunk : nMax;
  if (n > nMax) n = nMax;
  const u32 per = (n + 1023) / 1024;
  const u32 i0 = min(n, threadIdx.x * per), i1 = min(n, i0 + per);
  u32 sum = 0;
  for (u32 i = i0; i < i1; i++) sum += in[i];
  u32 tot;
  u32 ex = block_excl_scan<u32, 1024>(sum, scratch, &tot);
  for (u32 i = i0; i < i1; i++) {
    u32 v = in[i];
    out[i] = ex;
    ex += v;
  }
  if (threadIdx.x == 0) *total = tot;
}

// pass 1a: how many {significant, SKIP} intervals per chunk of 2048
__global__ __launch_bounds__(SW_NT) void k_sweep_count(const float* __restrict__ p, const float* __restrict__ q,
                                                       const u32* __restrict__ nPtr, float thr, u32* __restrict__ chunkCnt) {
  __shared__ u32 s_cnt[SW_NT / 64];
  const u32 n = *nPtr;
  const u32 i0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  if (blockIdx.x * SW_CHUNK >= n) return;
  const float* pq = q ? q : p;
  u32 cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++) {
    u32 i = i0 + k;
    if (i < n) {
      float v = pq[i];
      cnt += (v > thr || v == GX_SKIPF);
    }
  }
  cnt = wave_sum(cnt);
  if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) chunkCnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// pass 1b: ordered write of those intervals
__global__ __launch_bounds__(SW_NT) void k_sweep_write(const u32* __restrict__ end, const float* __restrict__ p,
                                                       const float* __restrict__ q, const u32* __restrict__ chromOff,
                                                       u32 nChrom, const u32* __restrict__ nPtr, float thr,
                                                       const u32* __restrict__ chunkOff, SweepList out) {
  __shared__ u32 scratch[8];
  const u32 n = *nPtr;
  if (blockIdx.x * SW_CHUNK >= n) return;
  const u32 i0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  float pv[SW_ITEMS], qv[SW_ITEMS];
  u32 keep = 0, cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++) {
    u32 i = i0 + k;
    if (i < n) {
      pv[k] = p[i];
      qv[k] = q ? q[i] : GX_SKIPF;
      float pq = q ? qv[k] : pv[k];
      if (pq > thr || pq == GX_SKIPF) { keep |= 1u << k; cnt++; }
    }
  }
  u32 tot;
  u32 o = chunkOff[blockIdx.x] + block_excl_scan<u32, SW_NT>(cnt, scratch, &tot);
  ChromCursor cur;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++)
    if (keep & (1u << k)) {
      u32 i = i0 + k;
      cur.seek(chromOff, nChrom, i);
      float pq = q ? qv[k] : pv[k];
      out.chrom[o] = cur.c;
      out.start[o] = i == cur.lo ? 0 : end[i - 1];
      out.end[o] = end[i];
      out.p[o] = pv[k];
      out.q[o] = qv[k];
      out.sig[o] = pq == GX_SKIPF ? 0u : 1u;
      o++;
    }
}

// head of a candidate: a significant interval that follows a SKIP marker, a chromosome
// change, or a gap wider than maxGap
__device__ __forceinline__ bool sweep_is_head(const SweepList& L, u32 j, int maxGap) {
  if (!L.sig[j]) return false;
  if (j == 0 || !L.sig[j - 1] || L.chrom[j - 1] != L.chrom[j]) return true;
  long long gap = (long long)L.start[j] - (long long)L.end[j - 1];
  return gap != 0 && gap > (long long)maxGap;
}

// pass 2a / 2b: count and write candidate heads (list positions), chunked like pass 1
__global__ __launch_bounds__(SW_NT) void k_heads_count(SweepList L, int maxGap, u32* __restrict__ chunkCnt) {
  __shared__ u32 s_cnt[SW_NT / 64];
  const u32 M = *L.count;
  if (blockIdx.x * SW_CHUNK >= M) return;
  const u32 j0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  u32 cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++) {
    u32 j = j0 + k;
    if (j < M) cnt += sweep_is_head(L, j, maxGap);
  }
  cnt = wave_sum(cnt);
  if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) chunkCnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(SW_NT) void k_heads_write(SweepList L, int maxGap, const u32* __restrict__ chunkOff,
                                                       u32* __restrict__ headPos) {
  __shared__ u32 scratch[8];
  const u32 M = *L.count;
  if (blockIdx.x * SW_CHUNK >= M) return;
  const u32 j0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  u32 keep = 0, cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++) {
    u32 j = j0 + k;
    if (j < M && sweep_is_head(L, j, maxGap)) { keep |= 1u << k; cnt++; }
  }
  u32 tot;
  u32 o = chunkOff[blockIdx.x] + block_excl_scan<u32, SW_NT>(cnt, scratch, &tot);
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++)
    if (keep & (1u << k)) headPos[o++] = j0 + k;
}

// one wavefront per candidate: updatePeak (943-970) over its intervals, then checkPeak (916-927)
__global__ __launch_bounds__(256) void k_peak_walk(SweepList L, const u32* __restrict__ headPos,
                                                   const u32* __restrict__ nHeads, float thr, float minAUC, int minLen,
                                                   gx_peak* __restrict__ cand, u32* __restrict__ valid) {
  const u32 M = *L.count, H = *nHeads;
  const u32 wavesPerGrid = gridDim.x * 4;
  const int lane = lane_id();
  for (u32 h = blockIdx.x * 4 + (threadIdx.x >> 6); h < H; h += wavesPerGrid) {
    const u32 j0 = headPos[h];
    const u32 jEnd = h + 1 < H ? headPos[h + 1] : M;  // members are [j0, first non-significant or jEnd)
    const bool qOpt = L.q[j0] != GX_SKIPF;
    const u32 peakStart = L.start[j0];
    float auc = 0.0f, summitVal = -1.0f, sp = -1.0f, sq = -1.0f;
    u32 summitPos = 0, summitLen = 0, peakEnd = 0;
    for (u32 base = j0; base < jEnd; base += 64) {
      u32 k = base + lane;
      bool in = k < jEnd && L.sig[k];
      u64 inMask = __ballot(in);
      // members are a prefix: stop at the first lane that is not one
      int nIn = (~inMask) ? __builtin_ctzll(~inMask) : 64;
      float term = 0.0f, pq = -2.0f, pv = 0.0f, qv = 0.0f;
      u32 s = 0, e = 0;
      if (lane < nIn) {
        s = L.start[k];
        e = L.end[k];
        pv = L.p[k];
        qv = L.q[k];
        pq = qOpt ? qv : pv;
        term = (float)(e - s) * (pq - thr);  // 949-950: float product ...
      }
      for (int l = 0; l < nIn; l++) auc += __shfl(term, l, 64);  // ... float running sum, in order
      // summit of this chunk: max pq, earliest lane (956-961)
      float mx = pq;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
      if (nIn > 0) {
        u64 atMax = __ballot(lane < nIn && pq == mx);
        int firstMax = __builtin_ctzll(atMax);
        // among the lanes at the maximum: first one with the greatest length (962-968)
        u32 len = (lane < nIn && pq == mx) ? e - s : 0;
        u32 ml = len;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ml = max(ml, (u32)__shfl_xor((int)ml, d, 64));
        u64 atLen = __ballot(lane < nIn && pq == mx && len == ml);
        int firstLen = __builtin_ctzll(atLen);
        u32 cPos = (u32)(((u64)__shfl((int)e, firstLen, 64) + (u32)__shfl((int)s, firstLen, 64)) / 2 - peakStart);
        if (mx > summitVal) {
          summitVal = mx;
          sp = __shfl(pv, firstMax, 64);
          sq = __shfl(qv, firstMax, 64);
          summitPos = cPos;
          summitLen = ml;
        } else if (mx == summitVal && ml > summitLen) {
          summitPos = cPos;
          summitLen = ml;
        }
        peakEnd = (u32)__shfl((int)e, nIn - 1, 64);
      }
      if (nIn < 64) break;
    }
    if (lane == 0) {
      bool ok = auc >= minAUC && (long long)peakEnd - (long long)peakStart >= (long long)minLen;
      valid[h] = ok;
      if (ok) {
        gx_peak pk;
        pk.chrom = L.chrom[j0];
        pk.start = peakStart;
        pk.end = peakEnd;
        pk.summit = summitPos;
        pk.auc = auc;
        pk.p = sp;
        pk.q = sq;
        cand[h] = pk;
      }
    }
  }
}

// pass 4: ordered compaction of the candidates that passed checkPeak (chunks of 2048 heads)
__global__ __launch_bounds__(SW_NT) void k_peaks_count(const u32* __restrict__ valid, const u32* __restrict__ nHeads,
                                                       u32* __restrict__ chunkCnt) {
  __shared__ u32 s_cnt[SW_NT / 64];
  const u32 H = *nHeads;
  if (blockIdx.x * SW_CHUNK >= H) return;
  const u32 h0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  u32 cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++)
    if (h0 + k < H) cnt += valid[h0 + k];
  cnt = wave_sum(cnt);
  if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) chunkCnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(SW_NT) void k_peaks_write(const gx_peak* __restrict__ cand, const u32* __restrict__ valid,
                                                       const u32* __restrict__ nHeads, const u32* __restrict__ chunkOff,
                                                       gx_peak* __restrict__ peaks, u64* __restrict__ peakBP) {
  __shared__ u32 scratch[8];
  const u32 H = *nHeads;
  if (blockIdx.x * SW_CHUNK >= H) return;
  const u32 h0 = blockIdx.x * SW_CHUNK + threadIdx.x * SW_ITEMS;
  u32 keep = 0, cnt = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++)
    if (h0 + k < H && valid[h0 + k]) { keep |= 1u << k; cnt++; }
  u32 tot;
  u32 o = chunkOff[blockIdx.x] + block_excl_scan<u32, SW_NT>(cnt, scratch, &tot);
  u64 bp = 0;
#pragma unroll
  for (int k = 0; k < SW_ITEMS; k++)
    if (keep & (1u << k)) {
      gx_peak pk = cand[h0 + k];
      peaks[o++] = pk;
      bp += pk.end - pk.start;
    }
  bp = wave_sum(bp);
  if (lane_id() == 0 && bp) atomicAdd(peakBP, bp);
}

}  // namespace gx
