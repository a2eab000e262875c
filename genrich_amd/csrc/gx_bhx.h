// gx_bhx.h -- Benjamini-Hochberg across ranks, general case (a control, replicates): the range-partitioned exchange.
//
// computeQval (Genrich.c:352-401) needs, for every distinct p, the base pairs of ALL chromosomes with a larger p
// (saveQval 212-250: k runs down the sorted table) and the running minimum of the raw values above it.  Chromosomes
// are sharded, so a rank holds a partial table {p -> bp}.  Gathering every rank's table to every rank (rounds 1-3)
// costs each rank the WHOLE table -- 3 x 10^7 distinct values for three Fisher-combined replicates -- however many
// ranks share the work.  Here the p axis is cut into W ranges of about equal numbers of distinct values and rank o
// owns range o:
//   1  every rank sorts its distinct values; 32 evenly spaced samples per rank are exchanged (fixed size, one
//      all-reduce of disjoint regions) and every rank derives the same W - 1 splitters from the W x 32 samples;
//   2  the W x W matrix of record counts is exchanged and read by the host: THE one synchronisation of the
//      exchange (it sizes the buffers and the send / receive calls);
//   3  all-to-all: a rank's {p, bp} records of range o go to rank o (ncclSend / ncclRecv in one group);
//   4  the owner merges what it received (its hash table), sorts the distinct values, and computes -- with two more
//      fixed-size exchanges: the ranges' total bp, then their smallest raw value -- the q of every value of its
//      range exactly as saveQval does: k = 1 + bp of the ranges above + bp above within the range,
//      q = max(min(raw, q of the next larger value), 0);
//   5  the answers travel back along the same counts (4 bytes per record) and land in the sender's own table, from
//      which k_qlookup reads as in a single-rank run.
// A rank sends and receives ~1/W of the distinct values; sorting, suffix sums and minima are over its range only.
#pragma once
#include "gx_stats.h"

namespace gx {

constexpr u32 BHX_SAMPLES = 32;

__global__ __launch_bounds__(64) void k_bhx_samples(const u32* __restrict__ sortedKeys, u32 D, u64* __restrict__ region) {
  const u32 j = threadIdx.x;
  if (j < BHX_SAMPLES) region[j] = D ? (u64)sortedKeys[(u64)(j + 1) * D / (BHX_SAMPLES + 1)] : 0xFFFFFFFFull;
}

// the same splitters on every rank: the samples ranked (ties by position), every BHX_SAMPLES-th of them
__global__ __launch_bounds__(1024) void k_bhx_splitters(const u64* __restrict__ samples, u32 nS, u32 W, u32* __restrict__ spl) {
  __shared__ u32 s[64 * BHX_SAMPLES], sorted[64 * BHX_SAMPLES];
  for (u32 i = threadIdx.x; i < nS; i += 1024) s[i] = (u32)samples[i];
  __syncthreads();
  for (u32 i = threadIdx.x; i < nS; i += 1024) {
    const u32 v = s[i];
    u32 r = 0;
    for (u32 j = 0; j < nS; j++) r += (u32)(s[j] < v || (s[j] == v && j < i));
    sorted[r] = v;
  }
  __syncthreads();
  for (u32 k = threadIdx.x; k <= W; k += 1024) spl[k] = k == 0 ? 0u : k == W ? 0xFFFFFFFFu : sorted[k * BHX_SAMPLES];
}

// where this rank's sorted values are cut (sendOff[o] = first value >= spl[o]) and how many go to each range
__global__ __launch_bounds__(128) void k_bhx_offsets(const u32* __restrict__ sortedKeys, u32 D, const u32* __restrict__ spl, u32 W,
                                                     u32* __restrict__ sendOff, u64* __restrict__ region) {
  __shared__ u32 off[65];
  const u32 o = threadIdx.x;
  if (o <= W) {
    u32 lo = 0, hi = D;
    if (o == W)
      lo = D;
    else {
      const u32 key = spl[o];
      while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (sortedKeys[mid] < key) lo = mid + 1; else hi = mid;
      }
    }
    off[o] = lo;
    sendOff[o] = lo;
  }
  __syncthreads();
  if (o < W) region[o] = off[o + 1] - off[o];
}

__global__ __launch_bounds__(256) void k_bhx_reduce(const u64* __restrict__ chunkSum, const float* __restrict__ chunkMin, u32 nCh,
                                                    u64* __restrict__ outSum, u64* __restrict__ outMin) {
  __shared__ u64 rs[4];
  __shared__ float rm[4];
  u64 sum = 0;
  float mn = FLT_MAX;
  for (u32 c = threadIdx.x; c < nCh; c += 256) {
    if (chunkSum) sum += chunkSum[c];
    if (chunkMin) mn = chunkMin[c] < mn ? chunkMin[c] : mn;
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float o = __shfl_xor(mn, d, 64);
    mn = o < mn ? o : mn;
  }
  if (lane_id() == 0) { rs[threadIdx.x >> 6] = sum; rm[threadIdx.x >> 6] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    float m = FLT_MAX;
    for (int w = 0; w < 4; w++) { t += rs[w]; m = rm[w] < m ? rm[w] : m; }
    if (outSum) *outSum = t;
    if (outMin) *outMin = (u64)__float_as_uint(m);
  }
}

// the owner's answers: q of every record it received, in the order it received them
__global__ __launch_bounds__(256) void k_bhx_answer(const BhRec* __restrict__ recs, u32 n, const u32* __restrict__ keys, u32 capMask,
                                                    const float* __restrict__ qOfSlot, float* __restrict__ out, u32* __restrict__ st) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 key = recs[i].key;
    u32 h = bh_hash(key) & capMask;
    u32 gk;
    while ((gk = keys[h]) != key && gk != EMPTY_KEY) h = (h + 1) & capMask;  // (inserted a moment ago: the probe ends at the key)
    // (a key that is not there -- a record lost on its way: with -L nothing else would notice -- fails the run as k_qlookup's
    // probe does, computeQval 377-382, instead of leaving a q of 0 behind)
    if (gk != key) atomicOr(st, ST_BH_LEN);
    out[i] = gk == key ? qOfSlot[h] : 0.0f;
  }
}

// ... and where they land: the q of this rank's own table, by slot
__global__ __launch_bounds__(256) void k_bhx_scatter(const u32* __restrict__ sortedSlot, const float* __restrict__ q, u32 D,
                                                     float* __restrict__ qOfSlot) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < D; i += gridDim.x * 256) qOfSlot[sortedSlot[i]] = q[i];
}

}  // namespace gx
