// gx_sort.h -- events -> endpoint records grouped by super-bucket, in ONE pass over the events.
//
// Replaces the accumulate step of saveInterval (Genrich.c:2546-2583: diff[start] += w, diff[end] -= w)
// together with level 1 of the bucket sort that groups the endpoint records by tile.  Round 1 did this
// in three passes (k_convert wrote the keys and took per-chunk histograms, k_scan_sb placed the
// super-buckets, k_scatter1 read the keys again and scattered them: 2.15 GB of HBM traffic for 0.4 GB of
// keys).  The histogram is only needed to know where a super-bucket starts -- so the super-buckets are
// not contiguous any more: each is a list of fixed-size PAGES taken from a pool, and a workgroup's run
// of records for a bin is reserved with one atomic add on the bin's cursor, exactly as before.  The
// events are read once (16 B each), the keys written once (4 B each): 1.2 GB.
//
//   page      8192 4-byte keys or 4096 8-byte records (32 KiB)
//   cursor    [NXCD][bin]  records reserved so far in the (XCD class, bin) list -- one list per class of
//             workgroups (blockIdx % 8 = the XCD a workgroup runs on), so the cache lines of a page are
//             completed inside one L2 (tools/bw_probe3.hip: 1.5x the write rate for short runs)
//   pt        [NXCD][bin][jmax]  page id + 1 of the list's j-th page, j >= 1 (0: not allocated yet); a list's
//             first page is fixed (page 1 + list index), so the common reservation needs no table look-up
// A reservation [o, o + c) (c <= one page) touches one or two pages.  Whoever reserves the FIRST slot of
// a page allocates it (one atomic add on the pool counter) and publishes its id in pt; a workgroup that
// needs a page it did not allocate polls pt -- the allocator is a workgroup that is already running and
// publishes right after its own reservation, so the wait is a memory round trip, bounded like every spin
// in this library (ST_LOOKBACK).  A list that needs more than jmax pages, or an exhausted pool, sets
// ST_PT_FULL and the host repeats the sample with a longer table (a pile-up of reads in one spot).
// Level 2 (k_bucket2p, one workgroup per super-bucket) walks the eight lists of its bin.
#pragma once
#include "gx_kernels.h"
#include "gx_stats.h"   // (lut_entry: k_bins_lut)

namespace gx {

constexpr u32 ST_PT_FULL = 512u;  // page table row / page pool exhausted (internal: the host retries)
constexpr u32 ST_SB_FULL = 1024u;  // (internal) a super-bucket does not fit k_sbtile: the host takes the general chain
constexpr u32 ST_SB_FRAC = 2048u;  // (internal) the sample holds fractional-weight records: likewise -- and the context's next
                                   // samples write pair records with a weight class (k_sort_a<true>)

template <typename R> struct PgCfg { static constexpr int SHIFT = 12; };   // 4096 x 8 B
template <> struct PgCfg<u32> { static constexpr int SHIFT = 13; };       // 8192 x 4 B
constexpr u32 PG_BYTES = 32768;

struct PagedStream {
  void* pool;       // pages; page 0 is a sink for writes that have nowhere to go after an overflow
  u32* pt;          // [NXCD][nBins][jmax]
  u32* cursor;      // [NXCD][nBins]
  u32* pagesUsed;   // dynamic pages handed out
  u32 jmax, poolPages;
  u32 nLists;       // NXCD * nBins: pages 1 .. nLists are the lists' first pages, dynamic ones follow
};

__device__ __forceinline__ u32 first_page(u32 list) { return 1u + list; }

// FragFix::slow (gx_kernels.h): bit 0 -- the general fragLen path is wanted; bit 1 -- because a fractional weight was SEEN (the host
// learns it with the sample's scalars: a context that was only told to expect fractions keeps the closed form until then)
constexpr u32 FRAG_SLOW_FRAC = 3u;
struct Sort1Out {
  u64* fragSum;   // [FRAG_SLOTS] sum of the clamped lengths of the fragments kept (closed form of fragLen)
  u32* slowFrag;  // FragFix::slow: FRAG_SLOW_FRAC goes up when a fractional weight is seen (general fragLen path)
  u32* endAtLen;  // [nChrom] weight of the events that end at (or beyond) the chromosome's end
  u32* hot;       // set when a base can reach the reference's int16 limits (see k_hot_check)
};

// one event -> its two endpoint records.  Returns the weight in 1/120 units (0: nothing to add).
struct Endpoints { u32 t0, o0, t1, o1; int w; };

// FX: with the side effects of the first conversion (status bits, covered bases, end-of-chromosome weights).
// Straight-line code (selects, one rare branch): the kernel converts eight events per thread, and as a tree of
// branches -- a switch over the count, one `return` per rejected case -- this function alone was ~180 instructions
// per event, most of them scalar mask bookkeeping.
// `c` = the chromosome's record, loaded by the caller for chromosome min(e.x, nChrom - 1) (so that a batch of
// events has its loads in flight together); `have` = false for the slots past the end of the input.
template <bool FX>
__device__ __forceinline__ Endpoints convert_event(const uint4 e, const DChrom c, bool have, u32 nChrom, const Sort1Out& out,
                                                   u32& bad, u64& covered) {
  // weight 1/count in 1/120 units for count in {1,2,3,4,5,6,8,10} (bit mask 0x57E), else 0: ERRALNS, 2402
  const u32 cnt = e.w;
  const bool cntOk = cnt <= 10u && ((0x57Eu >> cnt) & 1u);
  // (120 / count is an integer for every accepted count: the rounded float quotient is exact)
  const int w0 = cntOk ? (int)(120.5f * __builtin_amdgcn_rcpf((float)cnt)) : 0;
  const bool chromOk = e.x < nChrom;
  const bool posOk = e.y < c.len;               // ERRPOS, 2531
  const u32 end = e.z > c.len ? c.len : e.z;    // 2536-2544
  // (an empty interval adds and removes the same weight.  One that ends before it starts -- the reference's
  // BAM reader makes them from reverse reads without SEQ -- is counted like any other: +w at its start, -w
  // at its end, a negative length towards fragLen)
  const bool live = have && chromOk && w0 != 0 && chrom_active(c) && posOk && end != e.y;
  if (FX) {
    bad |= !have ? 0u
                 : (cntOk ? 0u : ST_BAD_COUNT) | (chromOk ? 0u : ST_BAD_CHROM) |
                       (chromOk && w0 != 0 && chrom_active(c) && !posOk ? ST_BAD_POS : 0u);
    covered += live ? (u64)((long long)end - (long long)e.y) : 0ull;
  }
  Endpoints r;
  r.w = live ? w0 : 0;
  r.t0 = live ? c.tileBase + (e.y >> TB) : NULL_TILE;
  r.o0 = e.y & (TILE - 1);
  const bool hasEnd = live && end < c.len;
  r.t1 = hasEnd ? c.tileBase + (end >> TB) : NULL_TILE;
  r.o1 = end & (TILE - 1);
  if (FX && live && !hasEnd) {  // rare: the fragment reaches the chromosome's end
    // the reference's diff has an entry at `len` too, and its int16 saturates there like anywhere else
    // (2565-2573): these ends have no record, so they are counted here
    if (atomicAdd(&out.endAtLen[e.x], (u32)w0) + (u32)w0 >= HOT16) atomicOr(out.hot, 1u);
  }
  return r;
}

#ifndef GX_S1_NT
#define GX_S1_NT 1024
#endif
constexpr int S1_NT = GX_S1_NT;             // threads per workgroup
#ifndef GX_S1_CHUNK
#define GX_S1_CHUNK 8192
#endif
constexpr int S1_CHUNK = GX_S1_CHUNK;       // events per workgroup
constexpr int S1_ITEMS = S1_CHUNK / S1_NT;
#ifndef GX_S1_BATCH
#define GX_S1_BATCH 8
#endif
constexpr int S1_LCHROM = 256;              // chromosome records kept in LDS (larger tables stay in global memory)
constexpr int S1_BPT = MAX_BINS / S1_NT;    // level-1 bins owned by a thread
static_assert(S1_CHUNK <= (1 << PgCfg<u32>::SHIFT), "a chunk's run for one bin never spans more than two pages");
constexpr u32 S1_STAGE_BYTES = S1_CHUNK * 4;  // a round's records: S1_CHUNK keys, or S1_CHUNK / 2 F records
static_assert(S1_BPT == 2 || S1_BPT == 4, "two or four bins per thread");

// The start keys and the end keys of a chunk go through level 1 TOGETHER (scatter_paged2): one set of barriers, one
// round of cursor reservations in flight for both -- each stream on its own was a chain of eight barrier-separated
// phases with a global atomic's round trip in the middle, in a kernel that runs one workgroup per CU.
//   tabA[s]  per bin: the number of records of stream s, then (same words: the counts are dead once their owners have
//            read them) the record index in the pool of the run's first record
//   tabB[s]  per bin: [15:0] start of the run in the staged chunk, [31:16] records of it in its first page
// (a run that crosses into a second page -- rare: a run is a few records, a page 8192 -- is finished by its OWNER
// thread, so no table of second pages is kept.)  The single-stream rounds of the fractional records use the same
// memory as hist / base0 / startSplit / base1.
struct S1Lds {
  u32 tabA[2][MAX_BINS];
  u32 tabB[2][MAX_BINS];
  __attribute__((aligned(8))) u32 scratch[40];
  u32 count[2];
  __attribute__((aligned(16))) unsigned char stage[2][S1_STAGE_BYTES];
};

// the page whose first slot the caller's run holds: allocate and publish
__device__ __forceinline__ u32 page_alloc(const PagedStream& P, u32* __restrict__ row, u32 j, u32* __restrict__ st) {
  if (j >= P.jmax) {
    atomicOr(st, ST_PT_FULL);
    return 0;
  }
  u32 pid = atomicAdd(P.pagesUsed, 1u) + 1u + P.nLists;
  if (pid >= P.poolPages) {
    atomicOr(st, ST_PT_FULL);
    pid = 0;
  }
  __hip_atomic_store(&row[j], pid + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return pid;
}

// a page somebody else allocates: poll its table entry
__device__ __forceinline__ u32 page_wait(const PagedStream& P, u32* __restrict__ row, u32 j, u32* __restrict__ st) {
  if (j >= P.jmax) return 0;  // (its allocator has raised ST_PT_FULL)
  u32 spins = 0;
  for (;;) {
    const u32 v = __hip_atomic_load(&row[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v) return v - 1u;
    __builtin_amdgcn_s_sleep(1);
    // (every allocator publishes, also after an overflow: only a lost allocator could keep this waiting)
    if (++spins > LB_SPIN_LIMIT ||
        ((spins & 1023u) == 0 && (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_LOOKBACK))) {
      atomicOr(st, ST_LOOKBACK);
      return 0;
    }
  }
}

// What the READERS of a list take as its length: the cursor, but no more than the table row's pages hold.  A list
// that outgrew its row (ST_PT_FULL: the host builds the sample again with longer rows) put the rest in the sink page;
// its readers -- which run before the host has seen the flag -- must neither index the table beyond the row nor take
// keys from the sink (another bin's: their tile offsets would index LDS tables out of range).  Everybody who turns
// cursors into counts clamps the same way, so offsets derived from them agree.
template <typename R> __device__ __forceinline__ u32 list_cap(u32 jmax) {
  return jmax >= (1u << (31 - PgCfg<R>::SHIFT)) ? 0x7FFFFFFFu : jmax << PgCfg<R>::SHIFT;
}
template <typename R> __device__ __forceinline__ u32 list_len(const PagedStream& P, u32 li) {
  return min(P.cursor[li], list_cap<R>(P.jmax));
}

// Scatter NR records per thread (NULL tile = none) of the whole workgroup into their bins' page lists.
template <typename R, int NR>
__device__ __forceinline__ void scatter_paged(const R (&rec)[NR], const PagedStream& P, int sbShift, u32 nBins, S1Lds& L,
                                              u32* __restrict__ st) {
  constexpr int SHIFT = PgCfg<R>::SHIFT;
  constexpr u32 PG = 1u << SHIFT;
  static_assert((size_t)NR * S1_NT * sizeof(R) <= S1_STAGE_BYTES && (u32)NR * S1_NT <= PG, "one page holds a workgroup's records");
  const u32 x = blockIdx.x % NXCD;
  R* stage = reinterpret_cast<R*>(L.stage[0]);
  u32* const Lhist = L.tabA[0];
  u32* const LstartSplit = L.tabB[0];
  u32* const Lbase0 = L.tabA[1];
  u32* const Lbase1 = L.tabB[1];
  for (int i = threadIdx.x; i < (int)nBins; i += S1_NT) Lhist[i] = 0;
  __syncthreads();
  u32 rk[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const u32 t = RecT<R>::tile(rec[k]);
    rk[k] = t != NULL_TILE ? atomicAdd(&Lhist[t >> sbShift], 1u) : 0u;
  }
  __syncthreads();
  {
    // thread t owns bins t, t + NT, t + 2 NT ...
    u32 cq[S1_BPT], liq[S1_BPT], oq[S1_BPT], j0q[S1_BPT], in0q[S1_BPT], split[S1_BPT], base[S1_BPT][2];
    u32* rowq[S1_BPT];
    bool need[S1_BPT];
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 b = threadIdx.x + q * S1_NT;
      cq[q] = b < nBins ? Lhist[b] : 0u;
      liq[q] = x * nBins + b;
      oq[q] = 0; j0q[q] = 0; in0q[q] = 0; split[q] = 0; base[q][0] = 0; base[q][1] = 0; rowq[q] = nullptr; need[q] = false;
    }
    // (all reservations in flight together)
#pragma unroll
    for (int q = 0; q < S1_BPT; q++)
      if (cq[q]) oq[q] = atomicAdd(&P.cursor[liq[q]], cq[q]);
    // Two passes, in this order for every lane of the wavefront: first everything this thread has to
    // PUBLISH (the pages whose first slot its reservations hold), then the waiting for pages that others
    // publish.  As one if / else the compiler may run the waiting lanes of a wavefront before its allocating
    // lanes, and two wavefronts then wait for each other's allocators (seen on MI355X: spin limit).
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 c = cq[q];
      if (c) {
        const u32 li = liq[q], o = oq[q];
        u32* row = P.pt + (size_t)li * P.jmax;
        const u32 j0 = o >> SHIFT, j1 = (o + c - 1) >> SHIFT, in0 = o & (PG - 1);
        rowq[q] = row;
        j0q[q] = j0;
        in0q[q] = in0;
        split[q] = min(c, PG - in0);
        if (j1 != j0) base[q][1] = page_alloc(P, row, j1, st) << SHIFT;
        if (j0 == 0) {  // the list's fixed first page: nothing to look up
          base[q][0] = (first_page(li) << SHIFT) + in0;
          if (j1 == j0) base[q][1] = first_page(li) << SHIFT;
        } else if (in0 == 0) {
          const u32 p0 = page_alloc(P, row, j0, st);
          base[q][0] = p0 << SHIFT;
          if (j1 == j0) base[q][1] = p0 << SHIFT;
        } else
          need[q] = true;
      }
    }
#pragma unroll
    for (int q = 0; q < S1_BPT; q++)
      if (need[q]) {
        const u32 p0 = page_wait(P, rowq[q], j0q[q], st);
        base[q][0] = (p0 << SHIFT) + in0q[q];
      }
    // one block scan for all of a thread's counts (each total <= 8192 < 2^16): 16-bit fields of one word
    u64 packed = 0;
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) packed |= (u64)cq[q] << (16 * q);
    u64 tot;
    const u64 ex = block_excl_scan<u64, S1_NT>(packed, reinterpret_cast<u64*>(L.scratch), &tot);
    u32 before = 0;  // records of the lower bin groups
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 b = threadIdx.x + q * S1_NT;
      if (b < nBins) {
        LstartSplit[b] = (before + (u32)((ex >> (16 * q)) & 0xFFFFu)) | (split[q] << 16);
        Lbase0[b] = base[q][0];
        Lbase1[b] = base[q][1];
      }
      before += (u32)((tot >> (16 * q)) & 0xFFFFu);
    }
    if (threadIdx.x == 0) L.count[0] = before;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const u32 t = RecT<R>::tile(rec[k]);
    if (t != NULL_TILE) stage[(LstartSplit[t >> sbShift] & 0xFFFFu) + rk[k]] = rec[k];
  }
  __syncthreads();
  const u32 cnt = L.count[0];
  R* pool = reinterpret_cast<R*>(P.pool);
  // (a fixed, unrolled trip count: the LDS reads of all NR records are in flight together)
  R v[NR];
  u32 ss[NR], b0[NR], b1[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const u32 i = (u32)k * S1_NT + threadIdx.x;
    v[k] = stage[i < cnt ? i : 0u];
  }
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const u32 i = (u32)k * S1_NT + threadIdx.x;
    const u32 b = i < cnt ? RecT<R>::tile(v[k]) >> sbShift : 0u;
    ss[k] = LstartSplit[b];
    b0[k] = Lbase0[b];
    b1[k] = Lbase1[b];
  }
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const u32 i = (u32)k * S1_NT + threadIdx.x;
    if (i < cnt) {
      const u32 r = i - (ss[k] & 0xFFFFu), sp = ss[k] >> 16;
      pool[r < sp ? b0[k] + r : b1[k] + (r - sp)] = v[k];
    }
  }
  __syncthreads();
}

// The unit-weight start keys and end keys of a chunk, both streams at once (see S1Lds).
template <int NR>
__device__ __forceinline__ void scatter_paged2(const u32 (&recS)[NR], const u32 (&recE)[NR], const PagedStream& PS, const PagedStream& PE,
                                               int sbShift, u32 nBins, S1Lds& L, u32* __restrict__ st) {
  constexpr int SHIFT = PgCfg<u32>::SHIFT;
  constexpr u32 PG = 1u << SHIFT;
  static_assert((size_t)NR * S1_NT * 4 <= S1_STAGE_BYTES && (u32)NR * S1_NT <= PG, "one page holds a workgroup's records");
  const u32 x = blockIdx.x % NXCD;
  for (int i = threadIdx.x; i < (int)nBins; i += S1_NT) {
    L.tabA[0][i] = 0;
    L.tabA[1][i] = 0;
  }
  __syncthreads();
  u32 rkS[NR], rkE[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    rkS[k] = recS[k] != NULL32 ? atomicAdd(&L.tabA[0][(recS[k] >> TB) >> sbShift], 1u) : 0u;
    rkE[k] = recE[k] != NULL32 ? atomicAdd(&L.tabA[1][(recE[k] >> TB) >> sbShift], 1u) : 0u;
  }
  __syncthreads();
  // thread t owns bins t, t + NT, t + 2 NT ... of both streams
  u32 cq[2][S1_BPT], oq[2][S1_BPT];
#pragma unroll
  for (int sI = 0; sI < 2; sI++)
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 b = threadIdx.x + q * S1_NT;
      cq[sI][q] = b < nBins ? L.tabA[sI][b] : 0u;
      oq[sI][q] = 0;
    }
  // (all reservations of both streams in flight together)
#pragma unroll
  for (int sI = 0; sI < 2; sI++) {
    const PagedStream& P = sI ? PE : PS;
#pragma unroll
    for (int q = 0; q < S1_BPT; q++)
      if (cq[sI][q]) oq[sI][q] = atomicAdd(&P.cursor[x * nBins + threadIdx.x + q * S1_NT], cq[sI][q]);
  }
  // block scans of a thread's counts (each total <= 8192 < 2^16): 16-bit fields of one word per stream.  (Every owner
  // has read its counts; the scans' barriers lie between those reads and the writes of the runs' bases into the same
  // words below.)
  static_assert(S1_BPT <= 4, "four 16-bit fields");
#pragma unroll
  for (int sI = 0; sI < 2; sI++) {
    u64 packed = 0;
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) packed |= (u64)cq[sI][q] << (16 * q);
    u64 tot;
    const u64 ex = block_excl_scan<u64, S1_NT>(packed, reinterpret_cast<u64*>(L.scratch), &tot);
    u32 before = 0;  // records of the lower bin groups
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 b = threadIdx.x + q * S1_NT;
      if (b < nBins) {
        const u32 split = min(cq[sI][q], PG - (oq[sI][q] & (PG - 1)));
        L.tabB[sI][b] = (before + (u32)((ex >> (16 * q)) & 0xFFFFu)) | (split << 16);
      }
      before += (u32)((tot >> (16 * q)) & 0xFFFFu);
    }
    if (threadIdx.x == 0) L.count[sI] = before;
  }
  // Where the runs go.  Two passes, in this order for every lane of the wavefront: first everything this thread has to
  // PUBLISH (the pages whose first slot its reservations hold), then the waiting for pages that others publish.  As
  // one if / else the compiler may run the waiting lanes of a wavefront before its allocating lanes, and two
  // wavefronts then wait for each other's allocators (seen on MI355X: spin limit).
  u32 waitMask = 0;
#pragma unroll
  for (int sI = 0; sI < 2; sI++) {
    const PagedStream& P = sI ? PE : PS;
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 c = cq[sI][q];
      if (c) {
        const u32 b = threadIdx.x + q * S1_NT, li = x * nBins + b, o = oq[sI][q], in0 = o & (PG - 1);
        if (o + c <= PG) {  // the list's fixed first page (the common case): nothing to look up
          L.tabA[sI][b] = (first_page(li) << SHIFT) + in0;
        } else {
          const u32 j0 = o >> SHIFT, j1 = (o + c - 1) >> SHIFT;
          u32* row = P.pt + (size_t)li * P.jmax;
          if (j1 != j0) (void)page_alloc(P, row, j1, st);  // (published in the table: its owner finds it there, below)
          if (j0 == 0)
            L.tabA[sI][b] = (first_page(li) << SHIFT) + in0;
          else if (in0 == 0)
            L.tabA[sI][b] = page_alloc(P, row, j0, st) << SHIFT;
          else
            waitMask |= 1u << (sI * S1_BPT + q);
        }
      }
    }
  }
  if (waitMask) {
#pragma unroll
    for (int sI = 0; sI < 2; sI++) {
      const PagedStream& P = sI ? PE : PS;
#pragma unroll
      for (int q = 0; q < S1_BPT; q++)
        if (waitMask & (1u << (sI * S1_BPT + q))) {
          const u32 b = threadIdx.x + q * S1_NT, li = x * nBins + b, o = oq[sI][q];
          L.tabA[sI][b] = (page_wait(P, P.pt + (size_t)li * P.jmax, o >> SHIFT, st) << SHIFT) + (o & (PG - 1));
        }
    }
  }
  __syncthreads();
  u32* stageS = reinterpret_cast<u32*>(L.stage[0]);
  u32* stageE = reinterpret_cast<u32*>(L.stage[1]);
#pragma unroll
  for (int k = 0; k < NR; k++) {
    if (recS[k] != NULL32) stageS[(L.tabB[0][(recS[k] >> TB) >> sbShift] & 0xFFFFu) + rkS[k]] = recS[k];
    if (recE[k] != NULL32) stageE[(L.tabB[1][(recE[k] >> TB) >> sbShift] & 0xFFFFu) + rkE[k]] = recE[k];
  }
  __syncthreads();
#pragma unroll
  for (int sI = 0; sI < 2; sI++) {
    const PagedStream& P = sI ? PE : PS;
    const u32* stage = sI ? stageE : stageS;
    u32* pool = reinterpret_cast<u32*>(P.pool);
    const u32 cnt = L.count[sI];
    // (a fixed, unrolled trip count: the LDS reads of all NR records are in flight together)
    u32 v[NR], ss[NR], b0[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const u32 i = (u32)k * S1_NT + threadIdx.x;
      v[k] = stage[i < cnt ? i : 0u];
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const u32 i = (u32)k * S1_NT + threadIdx.x;
      const u32 b = i < cnt ? (v[k] >> TB) >> sbShift : 0u;
      ss[k] = L.tabB[sI][b];
      b0[k] = L.tabA[sI][b];
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const u32 i = (u32)k * S1_NT + threadIdx.x;
      if (i < cnt) {
        const u32 r = i - (ss[k] & 0xFFFFu), sp = ss[k] >> 16;
        if (r < sp) pool[b0[k] + r] = v[k];  // (the part of a run beyond its first page: its owner, below)
      }
    }
    // the owners of the runs that cross into a second page copy that part
#pragma unroll
    for (int q = 0; q < S1_BPT; q++) {
      const u32 c = cq[sI][q], o = oq[sI][q];
      if (c && ((o + c - 1) >> SHIFT) != (o >> SHIFT)) {  // rare
        const u32 b = threadIdx.x + q * S1_NT, li = x * nBins + b, j1 = (o + c - 1) >> SHIFT;
        const u32 split = PG - (o & (PG - 1)), start = L.tabB[sI][b] & 0xFFFFu;
        // (this thread allocated that page and published it itself; past the table's end, or with the pool exhausted,
        // the records go to the sink page 0 and ST_PT_FULL is up)
        const u32 page = j1 < P.jmax ? __hip_atomic_load(&P.pt[(size_t)li * P.jmax + j1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u : 0u;
        for (u32 k = split; k < c; k++) pool[(page << SHIFT) + (k - split)] = stage[start + k];
      }
    }
  }
  __syncthreads();
}

// UNIT32: tile id + offset fit a 4-byte key, so unit-weight events (the common case) go to the S / E streams as
// bare keys and only multimapped ones (weight 1/k) to F.  Otherwise (a genome beyond 4.29 Gbp) everything is an F record.
template <bool UNIT32>
__global__ __launch_bounds__(S1_NT) void k_sort1(const gx_event* __restrict__ ev, u32 n, const DChrom* __restrict__ chroms,
                                                 u32 nChrom, int sbShift, u32 nBins, PagedStream PS, PagedStream PE,
                                                 PagedStream PF, Sort1Out out, u32* __restrict__ st) {
  __shared__ S1Lds L;
  // the chromosome table in LDS when it fits (hg38: 25 records): the conversion's second, dependent round trip
  // becomes an LDS look-up
  __shared__ DChrom lchrom[S1_LCHROM];
  const bool chromLds = nChrom <= (u32)S1_LCHROM;
  const u32 begin = blockIdx.x * S1_CHUNK;
  u32 bad = 0, frac = 0;
  u64 covered = 0;
  u32 ks[S1_ITEMS], ke[S1_ITEMS];
  // batches of S1_BATCH events: their loads, then their chromosomes' records, are in flight together (one event
  // after the other, each waiting for its two dependent loads, the conversion was a chain of 16 memory round trips)
  constexpr int S1_BATCH = GX_S1_BATCH;
#pragma unroll
  for (int k0 = 0; k0 < S1_ITEMS; k0 += S1_BATCH) {
    uint4 e[S1_BATCH];
    DChrom c[S1_BATCH];
    bool have[S1_BATCH];
#pragma unroll
    for (int q = 0; q < S1_BATCH; q++) {
      const u32 i = begin + (k0 + q) * S1_NT + threadIdx.x;
      have[q] = i < n;
      e[q] = reinterpret_cast<const uint4*>(ev)[have[q] ? i : n - 1];  // chrom, start, end, count
    }
    if (k0 == 0 && chromLds) {  // (behind the first batch of event loads: the table's round trip rides along with theirs)
      for (u32 i = threadIdx.x; i < nChrom; i += S1_NT) lchrom[i] = chroms[i];
      __syncthreads();
    }
    if (chromLds) {  // block-uniform
#pragma unroll
      for (int q = 0; q < S1_BATCH; q++) c[q] = lchrom[min(e[q].x, nChrom - 1)];
    } else {
#pragma unroll
      for (int q = 0; q < S1_BATCH; q++) c[q] = chroms[min(e[q].x, nChrom - 1)];
    }
#pragma unroll
    for (int q = 0; q < S1_BATCH; q++) {
      const Endpoints p = convert_event<true>(e[q], c[q], have[q], nChrom, out, bad, covered);
      const bool unit = UNIT32 && p.w == GX_UNIT;
      ks[k0 + q] = unit ? (p.t0 << TB) | p.o0 : NULL32;
      ke[k0 + q] = unit && p.t1 != NULL_TILE ? (p.t1 << TB) | p.o1 : NULL32;
      frac |= (u32)(p.w != 0 && !unit);
    }
  }
#ifndef GX_EXP_S1   // measurement hook (tools/build_variant.sh -DGX_EXP_S1=n): 1 no scatter at all, 2 the start keys only
#define GX_EXP_S1 0
#endif
  if (GX_EXP_S1 == 1) {
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < S1_ITEMS; k++) x ^= ks[k] ^ ke[k];
    if (x == 0xDEADBEEFu) atomicOr(st, 1u << 30);
  } else if (UNIT32) {
    scatter_paged2<S1_ITEMS>(ks, ke, PS, PE, sbShift, nBins, L, st);
  }
  // fractional (or wide) records: rare, so the events are converted again (they are in L2) instead of being
  // kept in registers; two events = up to four records per thread and round
  if (__syncthreads_or((int)frac)) {
    constexpr int EPR = S1_ITEMS / 4;  // events per thread and round: a round's records (two per event) fill one page at most
    for (int k0 = 0; k0 < S1_ITEMS; k0 += EPR) {
      u64 fr[2 * EPR];
#pragma unroll
      for (int q = 0; q < EPR; q++) {
        const u32 i = begin + (k0 + q) * S1_NT + threadIdx.x;
        fr[2 * q] = (u64)NULL_TILE << 32;
        fr[2 * q + 1] = (u64)NULL_TILE << 32;
        if (i < n) {
          const uint4 e = reinterpret_cast<const uint4*>(ev)[i];
          u32 bad2 = 0;
          u64 cov2 = 0;
          const Endpoints p = convert_event<false>(e, chroms[min(e.x, nChrom - 1)], true, nChrom, out, bad2, cov2);
          if (p.w && !(UNIT32 && p.w == GX_UNIT)) {
            fr[2 * q] = make_rec64(p.t0, p.o0, p.w);
            if (p.t1 != NULL_TILE) fr[2 * q + 1] = make_rec64(p.t1, p.o1, -p.w);
          }
        }
      }
      scatter_paged<u64, 2 * S1_ITEMS / 4>(fr, PF, sbShift, nBins, L, st);
    }
  }
  if (bad) atomicOr(st, bad);
  if (frac) atomicOr(out.slowFrag, FRAG_SLOW_FRAC);
  covered = wave_sum(covered);
  if (lane_id() == 0 && covered) atomicAdd(&out.fragSum[(blockIdx.x * 16 + (threadIdx.x >> 6)) % FRAG_SLOTS], covered);
}

// ---- pair mode: ONE 4-byte record per fragment --------------------------------------------------------------
// For the fused tile stage (k_sbtile<true>, gx_sbtile.h).  A unit-weight fragment whose two ends lie in the SAME
// super-bucket and less than 2^12 bases apart -- practically every fragment -- leaves level 1 as one record
//     [31:12] start, relative to the super-bucket's first base     [11:0] length (1 .. 4095)
// in the bin of its start (stream P: the S stream's pool), instead of a 4-byte start key and a 4-byte end key in two
// streams: half the bytes written here and read by the tile stage, half the runs, one set of LDS tables -- small
// enough for TWO workgroups per CU (512 threads, 16 events each), so that one workgroup's loads run under the other's
// scatter (with one workgroup per CU, k_sort1's phases follow each other on a CU and on the whole chip at once).
// Everything else ("singles": a fragment that crosses into the next super-bucket, one of 4,096 bases or more, one
// that reaches the end of its chromosome and has no end record, one that ends before it starts; also every record of
// fractional weight, which sends the sample to the general chain afterwards) is appended record by record to the
// F stream as 8-byte signed-weight records, straight to global memory: a handful per workgroup.  A pair adds nothing
// to the pileup that a super-bucket hands to the next one; the singles' weights are summed per bin (binNet).
constexpr u32 PAIR_LEN_BITS = 12;
constexpr int SBT_MAXSHIFT = 8;            // tiles per super-bucket (log2) that k_sbtile takes
static_assert(TB + SBT_MAXSHIFT + PAIR_LEN_BITS <= 32, "start within the super-bucket and length share 32 bits");

// one record to the end of its (XCD class, bin) list, by the thread that holds it (lanes without one: have = false).
// The page protocol of scatter_paged with runs of one record; the wavefront's allocations come before any of its waits.
__device__ __forceinline__ void append_single(const PagedStream& P, u32 li, u64 rec, bool have, u32* __restrict__ st) {
  constexpr int SHIFT = PgCfg<u64>::SHIFT;
  constexpr u32 PG = 1u << SHIFT;
  u32 o = 0, page = 0;
  bool wait = false;
  u32* row = P.pt + (size_t)li * P.jmax;
  if (have) {
    o = atomicAdd(&P.cursor[li], 1u);
    const u32 j = o >> SHIFT;
    if (j == 0)
      page = first_page(li);
    else if ((o & (PG - 1)) == 0)
      page = page_alloc(P, row, j, st);
    else
      wait = true;
  }
  __builtin_amdgcn_wave_barrier();  // (keeps the waits below from being merged into the branch above)
  if (wait) page = page_wait(P, row, o >> SHIFT, st);
  if (have) reinterpret_cast<u64*>(P.pool)[((size_t)page << SHIFT) + (o & (PG - 1))] = rec;
}

// ---- pair mode in two passes: 64 coarse bins, then 64 fine bins each --------------------------------------------
// Measured in round 4 on a one-pass pair kernel (k_sort1p, since removed; hg38, 50 M fragments): loads + conversion 0.14 ms,
// with the reservations 0.46 ms, everything 0.59 ms -- what costs is one atomic add per (workgroup, bin) on the bins' cursors,
// 6,104 x 2,946 = 18 M of them onto 12 KB of cursors per XCD class, and what they buy are runs of 2.8 records (11 bytes).
// Two passes with at most 64 bins each instead:
//   k_sort_a  events -> pair records, grouped by COARSE bin (64 level-1 bins = 2^26 bases): 46 reservations per
//             workgroup for hg38, runs of ~180 records.  A record leaves as the 4-byte pair record it will be (start within
//             its fine bin, length) plus one byte, the fine bin's index within the coarse one -- two arrays over one
//             index space (a coarse list's pages hold 8192 x (4 + 1) bytes).
//   k_sort_b  one workgroup per page of a coarse list: the page's records to the 64 fine bins' lists (the ones
//             k_sbtile<true> reads: unchanged), 64 reservations per workgroup, runs of ~128 records.
// 0.7 M atomics instead of 18 M, every run a few full cache lines; the price is 0.25 GB written and read once more.
// Both kernels are the same scatter over <= 64 keys (scatter64): LDS histogram with ranks, the first wavefront reserves
// the runs and finds their pages (the page protocol of scatter_paged: allocations before waits), records staged by
// key, written in staged order so that neighbouring lanes write neighbouring words.
#ifndef GX_S2A_WAVES
#define GX_S2A_WAVES 4   // waves per SIMD k_sort_a is compiled for (two workgroups per CU; at six it spills 24 dwords)
#endif
#ifndef GX_S2B_WAVES
#define GX_S2B_WAVES 6   // ... and k_sort_b (three workgroups per CU)
#endif
#ifndef GX_S2_NT
#define GX_S2_NT 512
#endif
constexpr int S2_NT = GX_S2_NT;
#ifndef GX_S2_BATCH
#define GX_S2_BATCH 4
#endif
constexpr int S2_BATCH = GX_S2_BATCH;   // event loads in flight per thread (k_sort_a)
constexpr int S2_LCHROM = 96;     // chromosome records k_sort_a keeps in LDS (larger tables stay in global memory)
constexpr int S2_ITEMS = 8192 / S2_NT;
static_assert(S2_ITEMS == 16 || S2_ITEMS == 8, "512 or 1024 threads");
constexpr int S2_CHUNK = S2_NT * S2_ITEMS;
constexpr int S2_KEYS = 128;       // keys of one scatter at most (k_sort_a: <= 64 coarse bins; k_sort_b: 64 or 128 fine bins)
constexpr int MAX_BINS_P = 8192;   // level-1 bins in pair mode: 64 coarse x 128 fine (a dense sample takes half-size bins)
__host__ __device__ inline int s2_fine_shift(u32 nBins) { return nBins > 4096u ? 7 : 6; }  // fine bins per coarse bin (log2)
static_assert(S2_CHUNK == (1 << PgCfg<u32>::SHIFT), "k_sort_b: one workgroup per page of a coarse list");
static_assert(S2_CHUNK == S1_CHUNK, "one grid size for the level-1 kernels");

struct S2Lds {
  u32 cnt[S2_KEYS];       // records of this chunk per key
  // per key: x where its run starts in the staged chunk, y records of the run that fit its first page, z pool index of
  // the run's first record, w pool index of the first record in the run's second page (one 16-byte read per record)
  __attribute__((aligned(16))) uint4 run[S2_KEYS];
  u32 total;
  u32 wtot[2];            // records of the keys 0 .. 63 / 64 .. 127 (the two owner wavefronts' totals)
  u32 scratch[24];
  u32 stage[S2_CHUNK];
  uint16_t ka[S2_CHUNK];  // of a staged record: [6:0] its key, [15:8] the byte that travels along (k_sort_a: the fine bin)
};

// The coarse lists' pages hold the 4-byte records at page * 8192 * 4 of `pool` and the bytes at page * 8192 of `aux`.
// AUX: whether the byte array is written (k_sort_a).  `listBase + key` = the list a key's run goes to.
// KEYS: 64, or 128 (k_sort_b on a genome of more than 4096 bins): one owner thread per key -- the first wavefront, or two.
template <bool AUX, int KEYS>
__device__ __forceinline__ void scatter64(const u32 (&rec)[S2_ITEMS], u32 (&ka)[S2_ITEMS], const PagedStream& P,
                                          uint8_t* __restrict__ auxPool, u32 listBase, u32 nKeys, S2Lds& L, u32* __restrict__ st) {
  // ka[k]: [6:0] key, [15:8] the byte that travels along, [31:16] the record's rank among its key's (filled here);
  // NULL32 in rec = no record
  static_assert(KEYS == 64 || KEYS == 128, "one or two owner wavefronts");
  constexpr int SHIFT = PgCfg<u32>::SHIFT;
  constexpr u32 PG = 1u << SHIFT;
  constexpr u32 KM = (u32)KEYS - 1u;
#pragma unroll
  for (int k = 0; k < S2_ITEMS; k++) ka[k] = (ka[k] & 0xFFFFu) | ((rec[k] != NULL32 ? atomicAdd(&L.cnt[ka[k] & KM], 1u) : 0u) << 16);
  __syncthreads();
  // the owners: counts -> reservations on the lists' cursors -> where the runs start in the staged chunk and in the pool.
  // Three steps with workgroup barriers between them (two owner wavefronts: the second needs the first one's total, and
  // every page either of them allocates is published before any of their lanes waits for one)
  const bool owner = threadIdx.x < (u32)KEYS;
  const u32 key = threadIdx.x;
  u32 c = 0, o = 0, inc = 0, li = 0;
  if (owner) {
    c = key < nKeys ? L.cnt[key] : 0u;
    li = listBase + key;
    if (c) o = atomicAdd(&P.cursor[li], c);
    inc = (u32)dpp_scan_add((int)c);
    if ((key & 63u) == 63u) L.wtot[key >> 6] = inc;
  }
  if (KEYS == 64 && threadIdx.x == 0) L.wtot[1] = 0;
  __syncthreads();
  const u32 in0 = o & (PG - 1), j0 = o >> SHIFT, j1 = (o + c - 1) >> SHIFT;
  u32* row = P.pt + (size_t)li * P.jmax;
  u32 p0 = 0, p1 = 0;
  bool wait0 = false;
  if (owner) {
    if (key == 0) L.total = L.wtot[0] + L.wtot[1];
    if (c) {
      if (j1 != j0) p1 = page_alloc(P, row, j1, st);  // (the run holds that page's first slot)
      if (j0 == 0)
        p0 = first_page(li);
      else if (in0 == 0)
        p0 = page_alloc(P, row, j0, st);
      else
        wait0 = true;
    }
  }
  __syncthreads();
  if (owner) {
    if (wait0) p0 = page_wait(P, row, j0, st);
    L.run[key] = make_uint4(inc - c + (key >= 64u ? L.wtot[0] : 0u), min(c, PG - in0), (p0 << SHIFT) + in0, p1 << SHIFT);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < S2_ITEMS; k++)
    if (rec[k] != NULL32) {
      const u32 pos = L.run[ka[k] & KM].x + (ka[k] >> 16);
      L.stage[pos] = rec[k];
      L.ka[pos] = (uint16_t)ka[k];
    }
  __syncthreads();
  u32* pool = reinterpret_cast<u32*>(P.pool);
  const u32 cnt = L.total;
#pragma unroll
  for (int h = 0; h < S2_ITEMS; h += 8) {
    u32 v[8], kk[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {  // (a fixed trip count: the LDS reads of eight records in flight together)
      const u32 i = (u32)(h + k) * S2_NT + threadIdx.x, ii = i < cnt ? i : 0u;
      v[k] = L.stage[ii];
      kk[k] = L.ka[ii];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 i = (u32)(h + k) * S2_NT + threadIdx.x;
      const uint4 rn = L.run[kk[k] & KM];
      const u32 r = i - rn.x;
      const u32 dst = r < rn.y ? rn.z + r : rn.w + (r - rn.y);
      if (i < cnt) {
        pool[dst] = v[k];
        if (AUX) auxPool[dst] = (uint8_t)(kk[k] >> 8);
      }
    }
  }
}

// events -> pair records in the coarse bins' lists (+ the slow events' records straight to the fine F lists, as
// k_sort1p does).  PC: the coarse lists, [NXCD][nCoarse]; its page-table rows hold every page a class can fill.
// FRAC: fractional weights ride along -- [11:9] of a record is the weight class (count 1, 2, 3, 4, 5, 6, 8, 10 -> 0 .. 7),
// the length keeps 9 bits (cut-site intervals and most fragments are shorter than 512 bases; the others are singles).
// A context switches to it once a sample has shown a fractional weight (gx_api.hip: sawFrac).
// PACKED: the events come as 8-byte gx_event8 records (include/genrich_amd.h: start; length, count class, chromosome) -- half the
// bytes of the step's largest stream.  A thread takes them two to a 16-byte load; everything behind the load sees the four
// words of a gx_event.
__device__ __forceinline__ uint4 unpack_event8(u32 start, u32 lcc) {
  // (count of a class: 1, 2, 3, 4, 5, 6, 8, 10 -- a nibble per class)
  return make_uint4(lcc >> 19, start, start + (lcc & 0xFFFFu), (0xA8654321u >> (4u * ((lcc >> 16) & 7u))) & 15u);
}
// gx_event8 -> gx_event (the paths that read events as 16-byte records: the general chain's k_sort1, gx_window_net, the replay of
// the reference's int16 decisions on the host)
__global__ __launch_bounds__(256) void k_unpack_events(const uint2* __restrict__ in, size_t n, uint4* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint2 e = in[i];
    out[i] = unpack_event8(e.x, e.y);
  }
}

template <bool FRAC, bool PACKED = false>
__global__ __launch_bounds__(S2_NT, GX_S2A_WAVES) void k_sort_a(const gx_event* __restrict__ ev, u32 n, const DChrom* __restrict__ chroms,
                                                     u32 nChrom, int sbShift, u32 nBins, u32 nCoarse, PagedStream PC,
                                                     uint8_t* __restrict__ auxPool, PagedStream PF, int* __restrict__ binNet,
                                                     Sort1Out out, u32* __restrict__ st) {
  __shared__ S2Lds L;
  __shared__ DChrom lchrom[S2_LCHROM];
  const int fineShift = s2_fine_shift(nBins);
  // (unit-weight records and a fractional weight somewhere in the input: the sample is going to be built again on the
  // general chain whatever this launch still does -- the workgroups that start after the flag went up leave at once)
  if (!FRAC && (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_SB_FRAC)) return;
  const bool chromLds = nChrom <= (u32)S2_LCHROM;
  const u32 x = blockIdx.x % NXCD;
  const u32 begin = blockIdx.x * S2_CHUNK;
  const u32 binMask = (1u << sbShift) - 1u;
  u32 bad = 0, slow = 0, fracSeen = 0;
  u32 covered32 = 0;  // (sixteen lengths below 2^12)
  u32 rec[S2_ITEMS], ka[S2_ITEMS];
  if (threadIdx.x < S2_KEYS) L.cnt[threadIdx.x] = 0;
  const uint4* __restrict__ evb = reinterpret_cast<const uint4*>(ev) + begin;
  const u32 lastIn = n - 1u - begin;  // (the grid covers the input: begin < n)
  // (the conversion: see k_sort1p)
#pragma unroll
  for (int k0 = 0; k0 < S2_ITEMS; k0 += S2_BATCH) {
    uint4 e[S2_BATCH];
    bool have[S2_BATCH];
    if constexpr (PACKED) {
      static_assert(S2_BATCH % 2 == 0, "two packed events per 16-byte load");
      // (item k of a thread is event 2 ((k / 2) S2_NT + thread) + (k & 1) of the chunk; the array is padded to an even count)
      const uint4* __restrict__ evp = reinterpret_cast<const uint4*>(reinterpret_cast<const uint2*>(ev) + begin);
#pragma unroll
      for (int q = 0; q < S2_BATCH; q += 2) {
        const u32 lp = ((k0 + q) >> 1) * S2_NT + threadIdx.x;
        have[q] = 2u * lp <= lastIn;
        have[q + 1] = 2u * lp + 1u <= lastIn;
        const uint4 two = evp[min(lp, lastIn >> 1)];
        e[q] = unpack_event8(two.x, two.y);
        e[q + 1] = unpack_event8(two.z, two.w);
      }
    } else {
#pragma unroll
    for (int q = 0; q < S2_BATCH; q++) {
      // (a uniform base and a 32-bit offset within the chunk: no 64-bit address per load)
      const u32 li = (k0 + q) * S2_NT + threadIdx.x;
      have[q] = li <= lastIn;
      e[q] = evb[min(li, lastIn)];  // chrom, start, end, count
    }
    }
    if (k0 == 0) {
      if (chromLds)
        for (u32 i = threadIdx.x; i < nChrom; i += S2_NT) lchrom[i] = chroms[i];
      __syncthreads();
    }
    // (two loops, not `chromLds ? lchrom[ci] : chroms[ci]`: the compiler makes ONE load of a selected pointer out of that -- a
    // FLAT load, through the texture path and waiting on both counters, for a record that lies in LDS: round 6, found in the ISA)
    DChrom cq[S2_BATCH];
    if (chromLds) {  // block-uniform
#pragma unroll
      for (int q = 0; q < S2_BATCH; q++) cq[q] = lchrom[min(e[q].x, nChrom - 1)];
    } else {
#pragma unroll
      for (int q = 0; q < S2_BATCH; q++) cq[q] = chroms[min(e[q].x, nChrom - 1)];
    }
#pragma unroll
    for (int q = 0; q < S2_BATCH; q++) {
      const DChrom c = cq[q];
      const u32 cnt = e[q].w;
      const bool cntOk = FRAC ? (cnt <= 10u && ((0x57Eu >> cnt) & 1u)) : cnt == 1u;
      const bool ok1 = have[q] && cntOk && e[q].x < nChrom;
      const bool act = chrom_active(c);
      const u32 len = e[q].z - e[q].y;
      const u32 t0 = c.tileBase + (e[q].y >> TB), t1 = c.tileBase + (e[q].z >> TB);
      const u32 bin = t0 >> sbShift;
      constexpr u32 LENB = FRAC ? 9u : PAIR_LEN_BITS;
      const bool fast = ok1 && act && e[q].z < c.len && len - 1u < (1u << LENB) - 1u && e[q].y < e[q].z && (t1 >> sbShift) == bin;
      const bool nothing = !have[q] || (ok1 && !act);
      // (the weight class of a count: a nibble per count)
      const u32 cls = FRAC ? (u32)((0x70605432100ull >> (4u * (cnt & 15u))) & 7ull) << 9 : 0u;
      fracSeen |= (u32)(FRAC && fast && cnt != 1u);
      rec[k0 + q] = fast ? ((((t0 & binMask) << TB) | (e[q].y & (TILE - 1))) << PAIR_LEN_BITS) | cls | len : NULL32;
      ka[k0 + q] = (bin >> fineShift) | ((bin & ((1u << fineShift) - 1u)) << 8);
      covered32 += fast && cnt == 1u ? len : 0u;
      slow |= (u32)(!fast && !nothing) << (k0 + q);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the next batch's loads stay behind this one's conversion: registers)
  }
  u64 covered = covered32;
  scatter64<true, 64>(rec, ka, PC, auxPool, x * nCoarse, nCoarse, L, st);
  // the slow events: loaded again (they are in L2), converted as k_sort1 converts every event, their records appended
  // one by one
  if (__ballot(slow != 0)) {
#pragma unroll 1
    for (int k = 0; k < S2_ITEMS; k++) {
      bool mine = (slow >> k) & 1u;
      if (!__ballot(mine)) continue;
      u64 r0 = 0, r1 = 0;
      u32 l0 = 0, l1 = 0;
      bool h1 = false;
      if (mine) {
        uint4 e;
        if constexpr (PACKED) {
          const uint2 e8 = reinterpret_cast<const uint2*>(ev)[begin + 2u * ((u32)(k >> 1) * S2_NT + threadIdx.x) + (u32)(k & 1)];
          e = unpack_event8(e8.x, e8.y);
        } else
          e = reinterpret_cast<const uint4*>(ev)[begin + k * S2_NT + threadIdx.x];
        const Endpoints p = convert_event<true>(e, chroms[min(e.x, nChrom - 1)], true, nChrom, out, bad, covered);
        mine = p.w != 0;  // (else: an event that only raised a status bit, or one without effect)
        if (mine) {
          if (p.w != GX_UNIT) {
            atomicOr(out.slowFrag, FRAG_SLOW_FRAC);
            if (!FRAC) {  // (unit-weight pair records: this sample goes to the general chain -- nothing more to append)
              atomicOr(st, ST_SB_FRAC);
              mine = false;
            }
          }
          if (mine) {
          r0 = make_rec64(p.t0, p.o0, p.w);
          l0 = x * nBins + (p.t0 >> sbShift);
          atomicAdd(&binNet[p.t0 >> sbShift], p.w);
          h1 = p.t1 != NULL_TILE;
          }
        }
        if (h1) {
          r1 = make_rec64(p.t1, p.o1, -p.w);
          l1 = x * nBins + (p.t1 >> sbShift);
          atomicAdd(&binNet[p.t1 >> sbShift], -p.w);
        }
      }
      append_single(PF, l0, r0, mine, st);
      append_single(PF, l1, r1, h1, st);
    }
  }
  if (bad) atomicOr(st, bad);
  if (FRAC && __ballot(fracSeen != 0) && lane_id() == 0) atomicOr(out.slowFrag, FRAG_SLOW_FRAC);  // (the closed form of fragLen is off)
  covered = wave_sum(covered);
  if (lane_id() == 0 && covered) atomicAdd(&out.fragSum[(blockIdx.x * 8 + (threadIdx.x >> 6)) % FRAG_SLOTS], covered);
}

// one workgroup per page of a coarse list -> the fine bins' lists (PP: what k_sbtile<true> reads).  The grid is an
// upper bound (NXCD x the most pages a class can hold); a workgroup finds its page from its class's cursors: list
// l = x * nCoarse + cb has ceil(cursor / 8192) pages.
__global__ __launch_bounds__(S2_NT, GX_S2B_WAVES) void k_sort_b(PagedStream PC, const uint8_t* __restrict__ auxPool, u32 nCoarse, u32 nBins,
                                                     PagedStream PP, u32* __restrict__ st) {
  constexpr int SHIFT = PgCfg<u32>::SHIFT;
  __shared__ S2Lds L;
  __shared__ u32 sList, sPage, sCount;
  // workgroup b takes page b / NXCD of XCD class b % NXCD (the class of the workgroups that filled the list: the fine
  // lists' cache lines are completed inside one L2, as in the single-pass kernels)
  if (threadIdx.x < 64) {
    const u32 xc = blockIdx.x % NXCD, q = blockIdx.x / NXCD;
    const u32 len = threadIdx.x < nCoarse ? PC.cursor[xc * nCoarse + threadIdx.x] : 0u;
    const u32 np = (len + S2_CHUNK - 1) >> SHIFT;
    const u32 inc = (u32)dpp_scan_add((int)np), ex = inc - np;
    if (threadIdx.x == 0) sCount = 0;
    L.cnt[threadIdx.x] = 0;
    L.cnt[threadIdx.x + 64] = 0;
    __builtin_amdgcn_wave_barrier();
    if (q >= ex && q < inc) {
      const u32 j = q - ex;
      sList = xc * nCoarse + threadIdx.x;
      sPage = j;
      sCount = min((u32)S2_CHUNK, len - (j << SHIFT));
    }
  }
  __syncthreads();
  const u32 count = sCount;
  if (count == 0) return;  // (beyond the last page)
  const u32 li = sList, j = sPage;
  const u32 x = li / nCoarse, cb = li - x * nCoarse;
  const u32 page = j ? __hip_atomic_load(&PC.pt[(size_t)li * PC.jmax + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u : first_page(li);
  const u32* src = reinterpret_cast<const u32*>(PC.pool) + ((size_t)page << SHIFT);
  const uint8_t* srcA = auxPool + ((size_t)page << SHIFT);
  // (thread t: records 16 t .. 16 t + 15 of the page -- four 16-byte loads of records, one of their bytes)
  u32 rec[S2_ITEMS], ka[S2_ITEMS];
  {
    constexpr int NQ = S2_ITEMS / 4;  // 16-byte loads of records per thread
    uint4 r4[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) r4[q] = reinterpret_cast<const uint4*>(src)[threadIdx.x * NQ + q];
    u32 aw[4] = {0u, 0u, 0u, 0u};
    if (NQ == 4) {
      const uint4 a4 = reinterpret_cast<const uint4*>(srcA)[threadIdx.x];
      aw[0] = a4.x; aw[1] = a4.y; aw[2] = a4.z; aw[3] = a4.w;
    } else {
      const uint2 a2 = reinterpret_cast<const uint2*>(srcA)[threadIdx.x];
      aw[0] = a2.x; aw[1] = a2.y;
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const u32 rw[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u32 i = threadIdx.x * S2_ITEMS + q * 4 + j;
        rec[q * 4 + j] = i < count ? rw[j] : NULL32;
        ka[q * 4 + j] = (aw[q] >> (8 * j)) & 0xFFu;
      }
    }
  }
  // (a pair record is never NULL32: its length is below 2^12 - 1 ... and a page holds only records)
  const int fineShift = s2_fine_shift(nBins);
  const u32 firstBin = cb << fineShift, nk = min(1u << fineShift, nBins - firstBin);
  if (fineShift == 7)  // block-uniform
    scatter64<false, 128>(rec, ka, PP, nullptr, x * nBins + firstBin, nk, L, st);
  else
    scatter64<false, 64>(rec, ka, PP, nullptr, x * nBins + firstBin, nk, L, st);
}

// bin totals (over the XCD classes) -> where each super-bucket's records start after level 2; one workgroup per stream.
// A fourth workgroup prepares what the tile stage needs besides:
//   chromW0[c]  weight of the ends that the chromosomes before c dropped at their own end (no record: endAtLen) --
//               the genome-wide prefix of the tiles' weights at chromosome c's first tile (k_sbtile's carry-in);
//   lambda from the closed form of fragLen (`wantEarly`: one rank, a treatment sample, unit weights so far),
//               so that the table p(V) exists before the tile stage (LooseCtl, gx_kernels.h)
struct BinScan {
  const u32* cursor[3];
  u32 cap[3];          // list_cap of the three streams (what a page-table row holds)
  u32* sbOff[3];
  const u32* endAtLen;
  int* chromW0;
  u32 nChrom;
  const FragFix* ff;
  Scalars* scal;
  LooseCtl* ctl;
  int wantEarly;
  // pair mode (k_sort1p): cursor[0] = the pair records' lists, cursor[2] = the singles'; then
  //   sbOff[0][b] = endpoint keys in the bins before b (two per pair, one per single): k_sbtile's loose-slot base
  //   sbOff[1][b] = weight (1/120) the bins before b hand on: the sum of their singles' weights (a pair hands on nothing)
  int pairMode;
  const int* binNet;
  u32* needPages;      // max over the lists of the pages a list would need (what the host sizes a longer page table by)
  // several ranks: {sum over the ranks of the closed form of fragLen, ranks for which it is not valid} (build_pileup's
  // early all-reduce) instead of this rank's partial sums
  const long long* early;
};

// (block: 0..2 the streams' bin offsets, 3 the chromosome prefix and the early lambda.  `ready`: set -- behind a fence --
// once lambda and LooseCtl's early words are written, for the table blocks of k_bins_lut that wait in the same launch)
__device__ __forceinline__ void scan_bins_body(const BinScan& B, u32 nBins, u32 block, u32* __restrict__ scratch, u32* ready) {
  if (block == 3) {
    if (threadIdx.x < 64 && B.wantEarly) {  // (first: the table blocks wait for it)
      // the wavefront fetches the partial sums side by side (one thread: a chain of FRAG_SLOTS round trips)
      static_assert(FRAG_SLOTS <= 64, "one lane per partial sum");
      u64 t = threadIdx.x < FRAG_SLOTS ? B.ff->fragSum[threadIdx.x] : 0ull;
      t = wave_sum(t);
      bool slow = B.ff->slow != 0;
      if (B.early) {
        t = (u64)B.early[0];
        slow = B.early[1] != 0;
      }
      if (threadIdx.x == 0) {
        if (!slow && t) {  // (as finish_frag will compute it when the correction of k_scan_iv / k_frag_walk is zero)
          const double fragLen = (double)(long long)t + (double)0ll * (1.0 / 134217728.0);
          const float lambda = (float)(fragLen / (double)B.scal->genomeLen);
          // (agent-scope stores, then "they have completed", then the flag: a release fence would write back an L2
          // that k_sort1 left full of dirty lines)
          st_agent(reinterpret_cast<u32*>(&B.scal->lambda), __float_as_uint(lambda));
          st_agent(&B.ctl->earlyBits, __float_as_uint(lambda));
          st_agent(&B.ctl->enabled, 1u);
        }
        if (ready) {
          stores_done();
          st_agent(ready, 1u);
        }
      }
    }
    const u32 per = (B.nChrom + 1023) / 1024;
    const u32 c0 = min(B.nChrom, threadIdx.x * per), c1 = min(B.nChrom, c0 + per);
    u32 sum = 0;
    for (u32 c = c0; c < c1; c++) sum += B.endAtLen[c];
    u32 tot;
    u32 ex = block_excl_scan<u32, 1024>(sum, scratch, &tot);
    for (u32 c = c0; c < c1; c++) {
      B.chromW0[c] = (int)ex;
      ex += B.endAtLen[c];
    }
    return;
  }
  const u32* cursor = B.cursor[block];
  u32* sbOff = B.sbOff[block];
  constexpr int PER = MAX_BINS_P / 1024;
  u32 v[PER], sum = 0;
  {  // the longest list of this stream, in pages (cursors count every reservation, also those beyond the table row)
    const int shift = block == 2 ? PgCfg<u64>::SHIFT : PgCfg<u32>::SHIFT;
    u32 mx = 0;
    for (u32 i = threadIdx.x; i < nBins * NXCD; i += 1024) mx = max(mx, cursor[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (u32)__shfl_xor((int)mx, d, 64));
    if (lane_id() == 0 && mx) atomicMax(B.needPages, (mx >> shift) + 1u);
  }
  // (a thread takes `per` consecutive bins, as few as the bin count asks for: 3 for hg38's 2,946)
  const u32 per = (nBins + 1023u) / 1024u;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 i = threadIdx.x * per + k;
    v[k] = 0;
    if ((u32)k < per && i < nBins) {
      if (B.pairMode && block == 0) {
        for (int x = 0; x < NXCD; x++)
          v[k] += 2u * min(cursor[x * nBins + i], B.cap[0]) + min(B.cursor[2][x * nBins + i], B.cap[2]);
      } else if (B.pairMode && block == 1) {
        v[k] = (u32)B.binNet[i];  // (two's complement: the prefix sums wrap like the ints they are)
      } else {
        for (int x = 0; x < NXCD; x++) v[k] += min(cursor[x * nBins + i], B.cap[block]);  // (list_len)
      }
    }
    sum += v[k];
  }
  u32 tot;
  u32 ex = block_excl_scan<u32, 1024>(sum, scratch, &tot);
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 i = threadIdx.x * per + k;
    if ((u32)k < per && i < nBins) sbOff[i] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) sbOff[nBins] = tot;
}

// this rank's words of the early all-reduce: the closed form of fragLen so far; 1 when it does not hold here
__global__ __launch_bounds__(64) void k_early_words(const FragFix* __restrict__ ff, int invalid, long long* __restrict__ w) {
  u64 t = threadIdx.x < FRAG_SLOTS ? ff->fragSum[threadIdx.x] : 0ull;
  t = wave_sum(t);
  if (threadIdx.x == 0) {
    w[0] = (long long)t;
    w[1] = invalid || ff->slow ? 1 : 0;
    w[2] = 0;
  }
}

__global__ __launch_bounds__(1024) void k_scan_bins(BinScan B, u32 nBins) {
  __shared__ u32 scratch[20];
  scan_bins_body(B, nBins, blockIdx.x, scratch, nullptr);
}

// The launch ahead of the tile stage, folded into the one that scans the level-1 bins (round 3: k_scan_bins, then
// k_pval_lut with `early`, two launches): blocks 0-3 are k_scan_bins', the PV_LUT / 1024 behind them build the table
// p(V) -- each quarter of a block is one of k_pval_lut's workgroups -- as soon as block 3 has published lambda.
// (Block 3 never waits and is dispatched before the table blocks: they cannot starve it.)
__global__ __launch_bounds__(1024) void k_bins_lut(BinScan B, u32 nBins, float* __restrict__ lutP, RiskBuf* __restrict__ risk,
                                                   DeepTab* __restrict__ deep, float thr, u32* __restrict__ st) {
  __shared__ u32 scratch[20];
  __shared__ u32 red[4][2];
  __shared__ u32 s_lambda, s_enabled;
  if (blockIdx.x < 4) {
    scan_bins_body(B, nBins, blockIdx.x, scratch, &B.ctl->ready);
    return;
  }
  if (blockIdx.x == 4 && threadIdx.x == 0) deep->n = 0;  // this sample's host-evaluated deep values come later
  if (threadIdx.x == 0) {
    u32 spins = 0;
    bool timedOut = false;
    while (!ld_agent(&B.ctl->ready)) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > LB_SPIN_LIMIT) {
        // (block 3 of this launch has not published: no table from here -- the launch behind the tile stage builds it, and
        // the sweep takes the tight table; nothing is raised: the run is only slower)
        timedOut = true;
        break;
      }
    }
    s_enabled = timedOut ? 0u : ld_agent(&B.ctl->enabled);
    if (timedOut) atomicOr(&B.ctl->bad, 4u);
    s_lambda = ld_agent(reinterpret_cast<u32*>(&B.scal->lambda));
  }
  __syncthreads();
  if (!s_enabled) return;  // lambda is not known yet: the launch after the tile stage builds the table
  const float lambda = __uint_as_float(s_lambda);
  double ml = 0, sl = 1;
  if (lambda != 0.0f) lnorm_params(lambda, &ml, &sl);
  const u32 v = (blockIdx.x - 4) * 1024 + threadIdx.x;
  lut_entry(v, lambda, ml, sl, lutP, risk, B.ctl, thr, red[threadIdx.x >> 8], threadIdx.x & 255u, v >> 8);
}

// ---- level 2 on paged input: one workgroup per super-bucket ---------------------------------------------------
// As k_bucket2 (gx_kernels.h): per-tile histogram, scan and cursors in LDS, a super-bucket that fits is read
// once into registers and written back as one contiguous stream of 16-bit tile offsets (or whole F records);
// a larger one takes the chunked two-pass path.  The difference is where a record comes from: logical index L
// of the bin = the concatenation of its eight (XCD class) lists, each a chain of pages.
template <typename R>
struct BinSrc {
  const R* pool;
  const u32* pt;   // rows of this bin: pt + (x * nBins + bin) * jmax
  u32 nBins, jmax, bin;
  const u32* pre;  // LDS: pre[x] = records in the lists before list x; pre[NXCD] = total
  __device__ __forceinline__ R at(u32 Lx) const {
    u32 x = 0;
#pragma unroll
    for (int q = 1; q < NXCD; q++) x += (u32)(Lx >= pre[q]);
    const u32 off = Lx - pre[x];
    const u32 j = off >> PgCfg<R>::SHIFT, li = x * nBins + bin;
    const u32 page = j ? pt[(size_t)li * jmax + j] - 1u : first_page(li);
    return pool[((size_t)page << PgCfg<R>::SHIFT) + (off & ((1u << PgCfg<R>::SHIFT) - 1u))];
  }
};

template <typename R>
__device__ __forceinline__ void bucket2p_body(const PagedStream& P, typename B2Out<R>::type* __restrict__ out,
                                              const u32* __restrict__ segOff, u32 nSeg, int sbShift, u32 nTiles,
                                              u32* __restrict__ tileCnt, int* __restrict__ tileWsum) {
  typedef typename B2Out<R>::type O;
  constexpr int ITEMS = ScCfg<R>::ITEMS;
  constexpr int CHUNK = B2_NT * ITEMS;
  constexpr int FITEMS = B2Cfg<R>::FITEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char b2_lds[];
  __shared__ u32 pre[NXCD + 1];
  __shared__ u64 slotPtr[B2Cfg<R>::FITEMS];
  __shared__ u32 slotCnt[B2Cfg<R>::FITEMS];
  const u32 nBins = 1u << sbShift;   // tiles per super-bucket
  constexpr u32 stageBytes = b2_stage_bytes<R>();
  R* stage = reinterpret_cast<R*>(b2_lds);
  O* stageO = reinterpret_cast<O*>(b2_lds);
  u32* hist = reinterpret_cast<u32*>(b2_lds + stageBytes);
  u32* start = hist + nBins;
  u32* cursor = start + nBins;
  u32* base = cursor + nBins;
  u32* scratch = base + nBins;
  auto outOf = [](R r) -> O {
    if constexpr (sizeof(R) == 4) return (O)((u32)r & (TILE - 1)); else return r;
  };
  for (u32 seg = blockIdx.x; seg < nSeg; seg += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 a = 0;
      for (int x = 0; x < NXCD; x++) {
        pre[x] = a;
        a += list_len<R>(P, x * nSeg + seg);
      }
      pre[NXCD] = a;
    }
    __syncthreads();
    const u32 total = pre[NXCD], obase = segOff[seg], segTileBase = seg << sbShift;
    if (total == 0) {  // (tile counts stay zero: the per-sample arena)
      continue;
    }
    const BinSrc<R> src{reinterpret_cast<const R*>(P.pool), P.pt, nSeg, P.jmax, seg, pre};
    // One-pass path: the bin is cut into SLOTS of B2_NT consecutive records of one list (a page is a whole
    // number of slots, so a slot never crosses a page): slot k is read by the whole workgroup, thread i its
    // i-th record, and everything about where the slot lies is the same for all threads -- one descriptor per
    // slot in LDS instead of a page-table walk per record.
    u32 nSlots = 0;
#pragma unroll
    for (int x = 0; x < NXCD; x++) nSlots += (pre[x + 1] - pre[x] + B2_NT - 1) / B2_NT;
    if (nSlots <= (u32)FITEMS) {
      for (int i = threadIdx.x; i < (int)nBins; i += B2_NT) { hist[i] = 0; base[i] = 0; }
      if (threadIdx.x < FITEMS) {
        const u32 k = threadIdx.x;
        u32 x = 0, k0 = 0;
        for (; x < NXCD; x++) {
          const u32 ns = (pre[x + 1] - pre[x] + B2_NT - 1) / B2_NT;
          if (k < k0 + ns) break;
          k0 += ns;
        }
        u64 ptr = 0;
        u32 cnt = 0;
        if (x < NXCD) {
          const u32 off = (k - k0) * B2_NT;
          const u32 jp = off >> PgCfg<R>::SHIFT, li = x * nSeg + seg;
          const u32 page = jp ? P.pt[(size_t)li * P.jmax + jp] - 1u : first_page(li);
          ptr = (u64)(reinterpret_cast<const R*>(P.pool) + ((size_t)page << PgCfg<R>::SHIFT) + (off & ((1u << PgCfg<R>::SHIFT) - 1u)));
          cnt = min((u32)B2_NT, pre[x + 1] - pre[x] - off);
        }
        slotPtr[k] = ptr;
        slotCnt[k] = cnt;
      }
      __syncthreads();
      R r[FITEMS];
      static_assert(FITEMS % 8 == 0, "batches of eight");
#pragma unroll
      for (int k0 = 0; k0 < FITEMS; k0 += 8) {
#pragma unroll
        for (int k = k0; k < k0 + 8; k++)
          if (threadIdx.x < slotCnt[k]) r[k] = reinterpret_cast<const R*>(slotPtr[k])[threadIdx.x];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < FITEMS; k++) {
        if (threadIdx.x < slotCnt[k]) {
          const u32 b = RecT<R>::tile(r[k]) - segTileBase;
          atomicAdd(&hist[b], 1u);
          if (sizeof(R) == 8) atomicAdd(&base[b], (u32)(int)(int8_t)((u64)r[k] & 0xFF));
        }
      }
      __syncthreads();
      u32 carry = 0;
      for (u32 b0 = 0; b0 < nBins; b0 += B2_NT) {
        const u32 b = b0 + threadIdx.x;
        const u32 c = b < nBins ? hist[b] : 0;
        u32 tot;
        const u32 ex = block_excl_scan<u32, B2_NT>(c, scratch, &tot);
        if (b < nBins) {
          cursor[b] = carry + ex;
          if (segTileBase + b < nTiles) {
            tileCnt[segTileBase + b] = c;
            if (sizeof(R) == 8) tileWsum[segTileBase + b] += (int)base[b];
          }
        }
        carry += tot;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < FITEMS; k++)
        if (threadIdx.x < slotCnt[k]) stageO[atomicAdd(&cursor[RecT<R>::tile(r[k]) - segTileBase], 1u)] = outOf(r[k]);
      __syncthreads();
      for (u32 i = threadIdx.x; i < total; i += B2_NT) out[obase + i] = stageO[i];
    } else {
      // chunked two-pass path for a super-bucket that does not fit (a pile-up of records in one place)
      for (int i = threadIdx.x; i < (int)nBins; i += B2_NT) { hist[i] = 0; start[i] = 0; }
      __syncthreads();
      for (u32 i0 = 0; i0 < total; i0 += CHUNK) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u32 idx = i0 + k * B2_NT + threadIdx.x;
          if (idx < total) {
            const R rr = src.at(idx);
            const u32 b = RecT<R>::tile(rr) - segTileBase;
            atomicAdd(&hist[b], 1u);
            if (sizeof(R) == 8) atomicAdd(&start[b], (u32)(int)(int8_t)((u64)rr & 0xFF));
          }
        }
      }
      __syncthreads();
      u32 carry = obase;
      for (u32 b0 = 0; b0 < nBins; b0 += B2_NT) {
        const u32 b = b0 + threadIdx.x;
        const u32 c = b < nBins ? hist[b] : 0;
        u32 tot;
        const u32 ex = block_excl_scan<u32, B2_NT>(c, scratch, &tot);
        if (b < nBins) {
          cursor[b] = carry + ex;
          hist[b] = 0;
          if (segTileBase + b < nTiles) {
            tileCnt[segTileBase + b] = c;
            if (sizeof(R) == 8) tileWsum[segTileBase + b] += (int)start[b];
          }
        }
        carry += tot;
      }
      __syncthreads();
      for (u32 i0 = 0; i0 < total; i0 += CHUNK) {
        const u32 cEnd = min(total, i0 + CHUNK);
        R r[ITEMS];
        u32 rk[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u32 idx = i0 + k * B2_NT + threadIdx.x;
          if (idx < cEnd) r[k] = src.at(idx);
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u32 idx = i0 + k * B2_NT + threadIdx.x;
          if (idx < cEnd) rk[k] = atomicAdd(&hist[RecT<R>::tile(r[k]) - segTileBase], 1u);
        }
        __syncthreads();
        u32 lc = 0;
        for (u32 b0 = 0; b0 < nBins; b0 += B2_NT) {
          const u32 b = b0 + threadIdx.x;
          const u32 c = b < nBins ? hist[b] : 0;
          u32 tot;
          const u32 ex = block_excl_scan<u32, B2_NT>(c, scratch, &tot);
          if (b < nBins) {
            start[b] = lc + ex;
            base[b] = cursor[b];
            cursor[b] += c;
            hist[b] = 0;
          }
          lc += tot;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u32 idx = i0 + k * B2_NT + threadIdx.x;
          if (idx < cEnd) stage[start[RecT<R>::tile(r[k]) - segTileBase] + rk[k]] = r[k];
        }
        __syncthreads();
        const u32 cnt = cEnd - i0;
        for (u32 i = threadIdx.x; i < cnt; i += B2_NT) {
          const R v = stage[i];
          const u32 b = RecT<R>::tile(v) - segTileBase;
          out[base[b] + (i - start[b])] = outOf(v);
        }
        __syncthreads();
      }
    }
  }
}

// The three streams of a sample in ONE launch (blockIdx.y: 0 start keys, 1 end keys, 2 fractional records): the
// workgroups of the next stream start while the last ones of the previous stream finish, and a run without
// fractional records pays no launch for finding every F bin empty.
struct Bucket2Job {
  PagedStream P;
  void* out;
  const u32* segOff;
  u32* tileCnt;
};
struct Bucket2Jobs { Bucket2Job j[3]; };

__global__ __launch_bounds__(B2_NT) void k_bucket2p(Bucket2Jobs J, u32 nSeg, int sbShift, u32 nTiles, int* __restrict__ tileWsum) {
  const Bucket2Job& jb = J.j[blockIdx.y];
  if (blockIdx.y < 2)
    bucket2p_body<u32>(jb.P, static_cast<uint16_t*>(jb.out), jb.segOff, nSeg, sbShift, nTiles, jb.tileCnt, tileWsum);
  else
    bucket2p_body<u64>(jb.P, static_cast<u64*>(jb.out), jb.segOff, nSeg, sbShift, nTiles, jb.tileCnt, tileWsum);
}

}  // namespace gx
