// gx_dups.h -- PCR duplicates (-r): "who held this key first?" for every alignment of a run, on the device.
//
// The reference visits the alignment sets of a file from the highest base-quality sum down (sortReads, Genrich.c:3362)
// and keeps a set unless one of its alignments matches one of a set kept before: chained hash tables keyed on the
// fields jenkins_hash_aln hashes (3408-3450: chromosome(s), 5' end(s), strand(s) by alignment type), findDupsPr /
// findDupsSn (3616, 3886), checkAndAdd (3514).  For a set with ONE alignment whose key no multi-alignment set shares
// the greedy rule is "the first holder of a key is kept, every later one is its duplicate" -- a pure function of
// (key, position in the visiting order) that needs no sequential walk: an open-addressing table of record indices,
// every record inserted at once (the slot keeps the smallest index), every record looked up at once.  Sets with
// several alignments can give up ALL their keys when one of them matches, so whatever shares a key with such a set
// ("contested": flagged here, a few per cent) is resolved by the host in the reference's order, as before.
#pragma once
#include "gx_kernels.h"

namespace gx {

constexpr u32 DUP_EMPTY = 0xFFFFFFFFu;
constexpr u32 DUP_CONTESTED = 0x80000000u;

struct DupTab {
  u32* rep;      // [cap] index of a record that holds the slot's key (DUP_EMPTY: free)
  u32* first;    // [cap] smallest index among the records with that key
  u32* multi;    // [cap] non-zero when a record of a multi-alignment set holds the key
  u32 mask;
};

__device__ __forceinline__ u32 dup_hash(const uint4 k) {
  u32 h = k.x * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + k.y * 0x85EBCA6Bu;
  h = (h ^ (h >> 13)) + k.z * 0xC2B2AE35u;
  h = (h ^ (h >> 16)) + k.w * 0x27D4EB2Fu;
  return h ^ (h >> 15);
}
__device__ __forceinline__ bool dup_eq(const uint4 a, const uint4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

// the slot of record i's key (claimed for it when nobody holds the key yet)
__device__ __forceinline__ u32 dup_slot(const uint4* __restrict__ keys, u32 i, const DupTab& T, bool claim) {
  const uint4 key = keys[i];
  u32 h = dup_hash(key) & T.mask;
  for (;;) {
    u32 v = __hip_atomic_load(&T.rep[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == DUP_EMPTY && claim) v = atomicCAS(&T.rep[h], DUP_EMPTY, i);
    if (v == DUP_EMPTY || v == i || dup_eq(keys[v], key)) return h;  // (the table is at most half full: a free slot ends every probe)
    h = (h + 1) & T.mask;
  }
}

__global__ __launch_bounds__(256) void k_dups_insert(const uint4* __restrict__ keys, const uint8_t* __restrict__ multi, u32 n, DupTab T) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 h = dup_slot(keys, i, T, true);
    atomicMin(&T.first[h], i);
    if (multi[i]) T.multi[h] = 1u;  // (every writer writes the same word)
  }
}

__global__ __launch_bounds__(256) void k_dups_lookup(const uint4* __restrict__ keys, u32 n, DupTab T, u32* __restrict__ owner) {
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u32 h = dup_slot(keys, i, T, false);
    owner[i] = T.first[h] | (T.multi[h] ? DUP_CONTESTED : 0u);
  }
}

}  // namespace gx
