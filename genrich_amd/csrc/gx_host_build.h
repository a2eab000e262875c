// gx_host_build.h -- a sample from its events to its run-length pileup and scalars: build_pileup (sort, tile stage, scans),
// finish_scalars, the retries (general chain, page tables, int16 saturation), the tile layout of a genome.
// (a part of gx_api.hip's translation unit: the kernels are templates and inline functions of the headers it includes;
// split by phase -- context / build / stats / sweep / collectives -- in round 5)
#pragma once
namespace {

uint64_t genome_len_for(const gx_ctx* ctx, const std::vector<uint8_t>& present) {
  // calcLambda 1819-1827 / findPeaks 1091-1101
  uint64_t g = 0;
  for (u32 i = 0; i < ctx->nChrom; i++)
    if (!ctx->skip[i] && present[i]) {
      g += ctx->len[i];
      for (size_t j = 0; j + 1 < ctx->bed[i].size(); j += 2) g -= ctx->bed[i][j + 1] - ctx->bed[i][j];
    }
  return g;
}

int upload_chroms(gx_ctx* ctx, bool force = true) {
  bool changed = force;
  for (u32 i = 0; i < ctx->nChrom; i++) {
    const u32 f = (ctx->skip[i] ? CH_SKIP : 0) | (ctx->save[i] ? CH_SAVE : 0) | (ctx->owned[i] ? CH_OWNED : 0);
    changed |= f != ctx->hChrom[i].flags;
    ctx->hChrom[i].flags = f;
  }
  if (!changed) return GX_OK;  // (the table on the device is this one already: no copy launch per sample)
  // (hChrom may be rewritten by the next call while this copy is in flight: pageable memory is staged by the runtime)
  HIPCHECK(hipMemcpyAsync(ctx->dChrom.p, ctx->hChrom.data(), ctx->nChrom * sizeof(DChrom), hipMemcpyHostToDevice,
                          ctx->stream));
  return GX_OK;
}

// loose slots -> tight (end, V) arrays of a pileup (only needed ahead of a control merge)
int pack_pileup(gx_ctx* ctx, Pileup& P) {
  if (P.packed) return GX_OK;
  hipStream_t s = ctx->stream;
  const u32 nTiles = ctx->nTiles;
  HIPCHECK(pooled(ctx, P.ivV, P.ivEnd.cap));
  PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), P.tileIvOff.as<u32>()};
  hipLaunchKernelGGL(k_pack, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU)))), dim3(256), 0, s, pin, nTiles,
                     P.ivEnd.as<u32>(), P.ivV.as<int>());
  if (int rc__ = dbg_sync(ctx, "k_pack")) return rc__;
  P.packed = true;
  return GX_OK;
}

// gx_sample_begin's clearing of the replicate's scalars, when no k_build_init is going to do it
int flush_begin(gx_ctx* ctx) {
  if (!ctx->beginPending) return GX_OK;
  hipLaunchKernelGGL(k_begin_sample, dim3(1), dim3(64), 0, ctx->stream, ctx->dScal.as<Scalars>(), ctx->beginGenome);
  ctx->beginPending = false;
  return GX_OK;
}

// A sample about to be merged with its control stays in its loose slots (k_merge2<true> reads them there): the
// buffers leave the context -- no copy -- and the context takes others for the next build (pooled).  With -E regions
// the merge needs tight arrays after all (gx_merge.h): pack_pileup.
int stash_or_pack(gx_ctx* ctx, Pileup& P) {
  if (ctx->hasBed) return pack_pileup(ctx, P);
  if (P.inLoose) return GX_OK;
  recycle(ctx, P.looseEnd);
  recycle(ctx, P.looseV);
  recycle(ctx, P.meta);
  P.looseEnd = std::move(ctx->looseEnd);
  P.looseV = std::move(ctx->looseV);
  P.meta = std::move(ctx->tileMeta);
  P.inLoose = true;
  return GX_OK;
}

// events -> tile-bucketed endpoint records -> run-length pileup (loose slots + offsets) and fragLen

// the most level-1 chunks (workgroups of k_sort_a) any XCD class gets: class = blockIdx % NXCD of each piece's launch
template <typename Segs> static u32 class_chunks(const Segs& segs) {
  u32 best = 0;
  for (u32 x = 0; x < (u32)NXCD; x++) {
    u32 c = 0;
    for (auto& sg : segs) {
      const u32 b = (u32)((sg.n + S2_CHUNK - 1) / S2_CHUNK);
      c += b / NXCD + (b % NXCD > x ? 1u : 0u);
    }
    best = std::max(best, c);
  }
  return best;
}

// Packed pieces (gx_event8) -> 16-byte events, in buffers of the library: for everything that reads gx_event records -- the general
// chain's k_sort1, gx_window_net, the host's replay of the reference's int16 decisions -- and (`all` = false) for the pieces that
// k_sort_a<.., PACKED> cannot read in place: a caller's device buffer that is not 16-byte aligned or ends on an odd count.
int unpack_segs(gx_ctx* ctx, bool all) {
  size_t total = 0;
  auto wanted = [&](const gx_ctx::Seg& sg) {
    return sg.packed && sg.n && (all || (reinterpret_cast<uintptr_t>(sg.p) & 15u) || (!sg.ready && (sg.n & 1u)));
  };
  for (auto& sg : ctx->segs)
    if (wanted(sg)) total += sg.n;
  if (!total) return GX_OK;
  if (ctx->unpackUsed == ctx->unpackBufs.size()) ctx->unpackBufs.emplace_back();
  DevBuf& buf = ctx->unpackBufs[ctx->unpackUsed++];
  HIPCHECK(buf.ensure(total * sizeof(gx_event)));
  hipStream_t s = ctx->stream;
  gx_event* at = buf.as<gx_event>();
  for (auto& sg : ctx->segs) {
    if (!wanted(sg)) continue;
    if (sg.ready) HIPCHECK(hipStreamWaitEvent(s, sg.ready, 0));  // (its upload, on the side stream)
    hipLaunchKernelGGL(k_unpack_events, dim3((u32)std::min<size_t>((sg.n + 255) / 256, 8192)), dim3(256), 0, s,
                       reinterpret_cast<const uint2*>(sg.p), sg.n, reinterpret_cast<uint4*>(at));
    sg.p = at;
    sg.ready = nullptr;   // (stream-ordered from here on)
    sg.packed = false;
    at += sg.n;
  }
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

// reuseSort: the sample was built a moment ago and only its tile stage has to be done again on the general chain
// (k_sbtile sent it back): level 1 of the sort -- the pages, the cursors, the closed form of fragLen -- is still
// there, so k_sort1 does not run again and only what the first tile stage and the scans wrote is cleared.
int build_pileup(gx_ctx* ctx, Pileup& out, int isCtrl, bool reuseSort = false) {
  const Knobs& K = ctx->knob;
  // Several ranks: lambda needs every rank's fragLen.  Its closed form (the sum of the fragment lengths, k_sort1) is
  // known BEFORE the tile stage, so the ranks exchange that (`earlyColl`: one all-reduce of three words behind
  // k_sort1; decided by what every rank knows alike) and each of them has the table p(V) and the sweep's bits from the
  // tile stage, as a single rank has.  The all-reduce behind the tile stage (finish_scalars) still carries the exact
  // parts and the ranks' flags; a rank whose lambda came out different there falls back to k_pack_pval as before.
  // (decided ahead of everything that can fail -- the size check, the allocations: a rank that leaves this function
  // early owes the others BOTH all-reduces, and poison_allreduce reads earlyOwed to know)
  const bool multiRank = ctx->world > 1 || ctx->forceColl;
  const bool forceSlowFrag = K.forceSlowFrag != 0, noFused = K.noFused != 0, noLoose = K.noLoose != 0;
  const bool earlyColl = multiRank && !isCtrl && !ctx->par.qval_opt && !ctx->bedGiven && !noLoose && !forceSlowFrag && !K.noEarlyColl;
  ctx->earlyColl = earlyColl;
  ctx->earlyOwed = earlyColl;
  // (host-pushed events sit in the library's device chunks, device-resident segments are used in place)
  const std::vector<gx_ctx::Seg>& segs = ctx->segs;
  size_t n = 0;
  for (auto& sg : segs) n += sg.n;
  if (2 * n >= 0xFFFFFFFFull) {
    ctx->err = "too many events in one sample for 32-bit record offsets";
    return GX_ERR_MEM;
  }
  const u32 nEv = (u32)n;
  const u32 nTiles = ctx->nTiles, nSB = ctx->nSB, nChrom = ctx->nChrom;
  // tile id + offset fit a 4-byte key (GX_FORCE_REC64=1 forces the wide-record path: used by the tests,
  // since only a genome beyond 4.29 Gbp takes it naturally)
  const bool unit32 = nTiles < MAX_TILES32 && !K.forceRec64;
  hipStream_t s = ctx->stream;
  gx_ctx::Stream& SS = ctx->str[0];
  gx_ctx::Stream& SE = ctx->str[1];
  gx_ctx::Stream& SF = ctx->str[2];
  const u32 nL1base = nSB - 1;  // level-1 bins = super-buckets (records without a tile are not scattered at all)
  // ---- what the tile stage will be -------------------------------------------------------------------------
  // k_sbtile (gx_sbtile.h): level 2 of the sort fused with the tile passes -- at most 2^8 tiles per super-bucket, and bins
  // that fit its LDS (a bin that does not raises ST_SB_FULL, a fractional record among unit-weight ones ST_SB_FRAC:
  // finish_scalars then has the sample built again on the general chain).  -E regions ride it since round 6 (pair mode: the
  // tiles with an edge are the second launch's, gx_sbtile.h TM_BEDX).
  // (a sample whose predecessor of the same kind did not fit is not even tried for a while: the same experiment's
  // next replicate, or the next run on the same data, has the same pile-ups)
  const bool backoff = !ctx->fusedOff && ctx->fusedBackoff[isCtrl ? 1 : 0] > 0;
  if (backoff) ctx->fusedBackoff[isCtrl ? 1 : 0]--;
  const bool pairsAllowed = !K.noPairs;
  // (fractional weights ride the pair records -- k_sort_a<true>, k_sbtile<.., true> -- once a sample of this context has
  // shown one; the start / end keys of the other fused variant cannot carry a weight)
  const bool fracOk = pairsAllowed && !K.noFracPairs;
  const bool fracLikely = ctx->sawFrac || ctx->fracHint;
  // (-E regions: unit-weight pair records only -- with a weight class as well the instances would be twelve)
  const bool fused = !backoff && unit32 && (!ctx->hasBed || (pairsAllowed && !K.noBedFused && !fracLikely)) && ctx->sbShift <= SBT_MAXSHIFT && !noFused &&
                     (!fracLikely || fracOk) && !ctx->fusedOff && !forceSlowFrag && (size_t)nEv <= (size_t)std::max(1u, nL1base) * 64000 &&
                     (size_t)2 * nEv + nTiles + ctx->nBedEdges + 64 < ((size_t)1 << 30);  // (k_sbtile's stores use 32-bit byte offsets)
  ctx->fusedUsed = fused;
  // ... and with it level 1: one record per fragment (k_sort_a / k_sort_b) when k_sbtile will read it
  const bool pairs = fused && !reuseSort && pairsAllowed;
  const bool fracPairs = pairs && fracLikely;
  ctx->pairsUsed = pairs;
  ctx->fracPairsUsed = fracPairs;
  ctx->packedUsed = false;
  // (8-byte events: k_sort_a reads them in place; everything else -- and a piece it cannot read in place -- gets 16-byte copies)
  if (!reuseSort)
    if (int rc__ = unpack_segs(ctx, !pairs)) return rc__;

  // A sample so dense that the average bin holds more keys than k_sbtile's key array (ATAC cut sites of a deep library)
  // takes bins of half the size -- level 1 of the pair mode reaches 64 x 128 of them -- so that a bin is one round of
  // the tile kernel again; the general chain (a later fall-back) keeps the context's own bin size.
  int sbS = ctx->sbShift;
  u32 nL1 = nL1base;
  // (not with fractional weights: measured at config 4, the tile passes with weights and the fragLen terms cost more per
  // key than the rounds of full-size bins -- 2.98 against 2.67 ms)
  const bool forceHalf = K.forceHalfBins != 0;  // (tests: the 128-key level 1 on a small input)
  if (pairs && (!fracPairs || forceHalf || K.fracHalfBins) && sbS > 0 &&
      (forceHalf || (size_t)2 * nEv > (size_t)std::max(1u, nL1base) * (SBT_KEYCAP - SBT_KEYCAP / 10)) &&
      ((nTiles + (1u << (sbS - 1)) - 1) >> (sbS - 1)) <= (u32)MAX_BINS_P && !K.noHalfBins) {
    sbS--;
    nL1 = (nTiles + (1u << sbS) - 1) >> sbS;
  }

  // level-2 output: 16-bit tile offsets (S, E) / whole records (F), tile-contiguous
  if (unit32) {
    HIPCHECK(SS.a.ensure((size_t)nEv * 2 + 16));
    HIPCHECK(SE.a.ensure((size_t)nEv * 2 + 16));
  }
  HIPCHECK(SF.a.ensure((size_t)nEv * 16 + 16));  // worst case: every event fractional
  // level-1 page pools (gx_sort.h): every record lands in one page of its (XCD class, bin) list
  const u32 jmax = ctx->ptJmax;
  u32 poolPages[3];
  // (page 0: sink; NXCD * nL1 fixed first pages; at most records / page-size further ones)
  poolPages[0] = poolPages[1] = (u32)(nEv >> PgCfg<u32>::SHIFT) + NXCD * nL1 + 4;
  poolPages[2] = (u32)(((size_t)2 * nEv) >> PgCfg<u64>::SHIFT) + NXCD * nL1 + 4;
  for (int q = 0; q < 3; q++) HIPCHECK(ctx->str[q].pool.ensure((size_t)poolPages[q] * PG_BYTES));
  // lambda ahead of the tile stage (closed form of fragLen; LooseCtl): one rank, a treatment sample, -p
  const bool wantEarly = !isCtrl && (!multiRank || earlyColl) && !ctx->par.qval_opt && !ctx->hasBed && unit32 && !noLoose && !forceSlowFrag &&
                         !ctx->sawFrac;  // (fractional weights: the closed form of fragLen is off, lambda only comes with the sample's end)
  // (round 6) ... and when lambda only comes with the sample's end -- fractional weights: no closed form of fragLen -- the sweep still
  // walks the loose slots: k_loose_late writes the bits and the fillers once the table p(V) is there (finish_scalars), instead of
  // k_pack_pval's copy of every interval into the tight table.  One rank, a treatment sample, -p, no -E regions, the fused tile stage.
  // (-q as well, where q is a function of the pileup: unit weights, no control to come -- gx_find_peaks' qLoose; a control sample
  // that follows only finds the verdict unused)
  const bool qLooseMay = ctx->par.qval_opt && !K.noQLoose && !ctx->qLooseBad && !ctx->sawFrac && !ctx->fracHint && !K.noPackHist;
  const bool wantLate = !wantEarly && !isCtrl && !multiRank && (!ctx->par.qval_opt || qLooseMay) && !ctx->hasBed && unit32 && !noLoose &&
                        !K.noLateLoose && fused && !forceSlowFrag;
  ctx->lateLoose = wantLate;
  const size_t looseCap = (size_t)2 * nEv + nTiles + ctx->nBedEdges + 16;  // slot t: records before + t (+ edges before)
  u64* sigMask = nullptr;
  if (wantEarly || wantLate) {
    // the sweep's masks in loose-slot index space: [significant | first of its chromosome]
    ctx->looseStride = (looseCap + 63) / 64 + 2;
    HIPCHECK(ctx->swMask.ensure(ctx->looseStride * 8 * 3));
    sigMask = ctx->swMask.as<u64>();
    ctx->maskIdx = -1;
  }
  // everything that must start at zero lives in one arena: one launch per sample clears it (k_build_init: with the
  // sweep's masks and the replicate's scalars)
  const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tileBytes = up((size_t)(nTiles + 1) * 4);
    const size_t ffBytes = up(sizeof(FragFix));
    const size_t endBytes = up((size_t)(nChrom + 1) * 4);
    const size_t curBytes = up((size_t)NXCD * nL1 * 4 + 64);          // cursors + (last word) pages handed out
    const size_t ptBytes = up((size_t)NXCD * nL1 * jmax * 4);
    const size_t lbTBytes = up((size_t)3 * (tChunks + 2) * 8), lbIBytes = up((size_t)(2 * tChunks + 4) * 8);
    const size_t ctlBytes = up(sizeof(LooseCtl));
    const size_t netBytes = up((size_t)(MAX_BINS_P + 2) * 4);  // pair mode: the singles' weight per level-1 bin
    // pair mode in two passes (k_sort_a / k_sort_b): the coarse lists' cursors and page tables
    const u32 nCoarse = (std::max(1u, nL1) + (1u << s2_fine_shift(nL1)) - 1) >> s2_fine_shift(nL1);
    const u32 jmaxC = class_chunks(segs) + 3;   // (a class's workgroups cannot fill more pages than that in one list)
    const size_t curCBytes = up((size_t)NXCD * nCoarse * 4 + 64), ptCBytes = up((size_t)NXCD * nCoarse * jmaxC * 4);
    const size_t total = ffBytes + 256 + ctlBytes + endBytes + netBytes + curCBytes + ptCBytes + 3 * (curBytes + ptBytes) + 5 * tileBytes + lbTBytes + lbIBytes;
    HIPCHECK(ctx->zeroArena.ensure(total));
    char* base = ctx->zeroArena.as<char>();
    ctx->fragSum.view(base, ffBytes);
    base += ffBytes;
    ctx->nWide.view(base, 256);
    base += 256;
    ctx->looseCtl.view(base, ctlBytes);
    base += ctlBytes;
    ctx->endAtLen.view(base, endBytes);
    base += endBytes;
    ctx->binNet.view(base, netBytes);
    base += netBytes;
    ctx->curC.view(base, curCBytes);
    base += curCBytes;
    ctx->ptC.view(base, ptCBytes);
    base += ptCBytes;
    for (int q = 0; q < 3; q++) {
      ctx->str[q].cursor.view(base, curBytes);
      base += curBytes;
      ctx->str[q].pt.view(base, ptBytes);
      base += ptBytes;
    }
    for (int q = 0; q < 3; q++, base += tileBytes) ctx->tileCnt[q].view(base, tileBytes);
    ctx->tileWsum.view(base, tileBytes);
    base += tileBytes;
    ctx->tileDeep.view(base, tileBytes);
    base += tileBytes;
    ctx->lb.view(base, lbTBytes);      // k_scan_tiles' three look-back arrays
    base += lbTBytes;
    ctx->lbIv.view(base, lbIBytes);    // k_scan_iv's two
    if (!reuseSort) {
      const size_t nA = total / 16, nB = wantEarly || wantLate ? ctx->looseStride * 8 * 2 / 16 : 0;
      static_assert(sizeof(Scalars) / 8 <= 256, "one workgroup clears the scalars");
      hipLaunchKernelGGL(k_build_init, dim3((u32)std::min<size_t>((nA + nB + 1023) / 1024, 4096)), dim3(256), 0, s,
                         ctx->dScal.as<Scalars>(), ctx->beginPending ? 1 : 0, ctx->beginGenome, ctx->zeroArena.as<uint4>(), nA,
                         wantEarly || wantLate ? ctx->swMask.as<uint4>() : (uint4*)nullptr, nB);
      ctx->beginPending = false;
    } else {
      if (int rc__ = flush_begin(ctx)) return rc__;
      if (wantEarly || wantLate) HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, ctx->looseStride * 8 * 2, s));
      // what the tile stage and the scans of the first attempt left: the per-tile tables and look-back arrays (the
      // arena's tail), the loose-sweep block, the correction words of fragLen and the wide-tile count
      char* tail = ctx->tileCnt[0].as<char>();
      HIPCHECK(hipMemsetAsync(tail, 0, (size_t)(ctx->zeroArena.as<char>() + total - tail), s));
      HIPCHECK(hipMemsetAsync(ctx->looseCtl.p, 0, ctlBytes, s));
      FragFix* f0 = ctx->fragSum.as<FragFix>();
      HIPCHECK(hipMemsetAsync(&f0->nList, 0, 12, s));   // nList, corr (the partial sums and the slow flag stay)
      HIPCHECK(hipMemsetAsync(ctx->nWide.p, 0, 4, s));  // (word 1, the int16 flag of k_sort1, stays)
      HIPCHECK(hipMemsetAsync(ctx->nWide.as<u32>() + 2, 0, 4, s));
    }
  }
  for (int q = 0; q < 3; q++) {
    HIPCHECK(ctx->str[q].sbOff.ensure((MAX_BINS_P + 2) * 4));
    HIPCHECK(ctx->tileOff[q].ensure((size_t)(nTiles + 2) * 4));
  }
  HIPCHECK(ctx->tileCarry.ensure((size_t)(nTiles + 1) * 4));
  // an interval closes at every base with a non-zero difference (<= one per record), at every -E edge,
  // plus one per chromosome
  const size_t ivCap = (size_t)2 * nEv + nChrom + ctx->nBedEdges + 16;
  HIPCHECK(pooled(ctx, out.ivEnd, ivCap * 4));
  HIPCHECK(pooled(ctx, out.tileIvOff, (size_t)(nTiles + 2) * 4));
  HIPCHECK(pooled(ctx, out.chromIvOff, (size_t)(nChrom + 2) * 4));

  phase_begin(ctx, isCtrl ? "c.sort1" : "t.sort1");
  // fragLen: closed form (sum of fragment lengths) unless something sets the slow flag
  HIPCHECK(ctx->fragList.ensure((size_t)(nTiles + 1) * 4));
  FragFix* ff = ctx->fragSum.as<FragFix>();
  u32* slowFrag = &ff->slow;
  // (-E regions: the fused tile stage takes the excluded bases' pileup off the closed form, FragFix::bedExcl; k_tile<BED> does not)
  if ((ctx->hasBed && !fused) || !unit32 || forceSlowFrag) HIPCHECK(hipMemsetAsync(slowFrag, 1, 4, s));
  PagedStream PG3[3];
  for (int q = 0; q < 3; q++) {
    gx_ctx::Stream& st = ctx->str[q];
    PG3[q] = PagedStream{st.pool.p, st.pt.as<u32>(), st.cursor.as<u32>(), st.cursor.as<u32>() + NXCD * nL1, jmax, poolPages[q],
                         NXCD * nL1};
  }
  Sort1Out so1{ff->fragSum, slowFrag, ctx->endAtLen.as<u32>(), ctx->nWide.as<u32>() + 1};
  PagedStream pcLast{};
  u32 ncLast = 0, gridB = 0;
  for (auto& seg : segs) {
    if (!seg.n || reuseSort) continue;
    // (a piece that is still on its way from the host: the main stream waits for that copy only, so the
    // scatter of the pieces that have arrived overlaps the upload of the rest)
    if (seg.ready) HIPCHECK(hipStreamWaitEvent(s, seg.ready, 0));
    const u32 blocks = (u32)((seg.n + S1_CHUNK - 1) / S1_CHUNK);
    if (pairs) {
      // two passes: coarse bins, then the fine ones (gx_sort.h); a piece of 8-byte events by the instance that reads those
      u32 nWG1 = 0;
      for (auto& sg : segs) nWG1 += (u32)((sg.n + S2_CHUNK - 1) / S2_CHUNK);
      const u32 nCoarse = (std::max(1u, nL1) + (1u << s2_fine_shift(nL1)) - 1) >> s2_fine_shift(nL1);
      const u32 perClass = class_chunks(segs), jmaxC = perClass + 3, nListsC = NXCD * nCoarse;
      const u32 pagesC = nWG1 + 2 * nListsC + 8;
      HIPCHECK(ctx->poolC.ensure((size_t)pagesC * PG_BYTES));
      HIPCHECK(ctx->auxC.ensure((size_t)pagesC << PgCfg<u32>::SHIFT));
      PagedStream PC{ctx->poolC.p, ctx->ptC.as<u32>(), ctx->curC.as<u32>(), ctx->curC.as<u32>() + nListsC, jmaxC, pagesC, nListsC};
#define GX_LAUNCH_SORT_A(F, P)                                                                                                 \
  hipLaunchKernelGGL((k_sort_a<F, P>), dim3(blocks), dim3(S2_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom, sbS, \
                     nL1, nCoarse, PC, ctx->auxC.as<uint8_t>(), PG3[2], ctx->binNet.as<int>(), so1, ctx->dStatus.as<u32>())
      if (fracPairs) { if (seg.packed) GX_LAUNCH_SORT_A(true, true); else GX_LAUNCH_SORT_A(true, false); }
      else { if (seg.packed) GX_LAUNCH_SORT_A(false, true); else GX_LAUNCH_SORT_A(false, false); }
#undef GX_LAUNCH_SORT_A
      ctx->packedUsed |= seg.packed;
      pcLast = PC;
      ncLast = nCoarse;
      gridB = NXCD * (perClass + nCoarse);   // (a class's lists hold at most its chunks' + one partly filled page each)
    } else if (unit32)
      hipLaunchKernelGGL(k_sort1<true>, dim3(blocks), dim3(S1_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom,
                         sbS, nL1, PG3[0], PG3[1], PG3[2], so1, ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL(k_sort1<false>, dim3(blocks), dim3(S1_NT), 0, s, seg.p, (u32)seg.n, ctx->dChrom.as<DChrom>(), nChrom,
                         sbS, nL1, PG3[0], PG3[1], PG3[2], so1, ctx->dStatus.as<u32>());
  }
  if (gridB)  // the coarse lists (all pieces' events) -> the fine bins' lists
    hipLaunchKernelGGL(k_sort_b, dim3(gridB), dim3(S2_NT), 0, s, pcLast, (const uint8_t*)ctx->auxC.as<uint8_t>(), ncLast, nL1, PG3[0],
                       ctx->dStatus.as<u32>());
  if (int rc__ = dbg_sync(ctx, "k_sort1")) return rc__;
  phase_end(ctx);
  if (K.fault == 1 && !reuseSort) HIPCHECK(hipMemsetAsync(ctx->endAtLen.p, 0x01, 4, s));  // (tests: ST_END_PILE must catch it)
  long long* earlyWords = nullptr;
  if (earlyColl) {
    // this rank's closed form, whether it is valid here (unit weights so far, no -E regions, 4-byte keys), [2] unused
    earlyWords = ctx->dColl.as<long long>() + 4;
    hipLaunchKernelGGL(k_early_words, dim3(1), dim3(64), 0, s, (const FragFix*)ff, wantEarly ? 0 : 1, earlyWords);
    if (int rc__ = allreduce_words(ctx, earlyWords, 3)) return rc__;
    ctx->earlyOwed = false;
  }

  LooseCtl* ctl = ctx->looseCtl.as<LooseCtl>();
  HIPCHECK(ctx->tileSlot.ensure((size_t)(nTiles + 2) * 4));
  HIPCHECK(ctx->chromW0.ensure((size_t)(nChrom + 1) * 4));
  HIPCHECK(pooled(ctx, ctx->chromLooseOff, (size_t)(nChrom + 2) * 4));  // (moves into the replicate's record: gx_pvalues)
  phase_begin(ctx, isCtrl ? "c.bucket" : "t.bucket");
  {
    auto capOf = [&](int shift) -> u32 { return jmax >= (1u << (31 - shift)) ? 0x7FFFFFFFu : jmax << shift; };  // (list_cap)
    BinScan bs{{SS.cursor.as<u32>(), SE.cursor.as<u32>(), SF.cursor.as<u32>()},
               {capOf(PgCfg<u32>::SHIFT), capOf(PgCfg<u32>::SHIFT), capOf(PgCfg<u64>::SHIFT)},
               {SS.sbOff.as<u32>(), SE.sbOff.as<u32>(), SF.sbOff.as<u32>()},
               ctx->endAtLen.as<u32>(), ctx->chromW0.as<int>(), nChrom, ff, ctx->dScal.as<Scalars>(), ctl, wantEarly ? 1 : 0,
               pairs ? 1 : 0, ctx->binNet.as<int>(), ctx->nWide.as<u32>() + 12, earlyWords};
    static_assert(PV_LUT % 1024 == 0, "k_bins_lut: four of k_pval_lut's workgroups per block");
    if (wantEarly)  // with the table p(V) for that lambda, and from which pileup on an interval is significant
      hipLaunchKernelGGL(k_bins_lut, dim3(4 + PV_LUT / 1024), dim3(1024), 0, s, bs, nL1, ctx->pvLut.as<float>(),
                         ctx->dRisk.as<RiskBuf>(), ctx->dDeep.as<DeepTab>(), ctx->par.thr, ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL(k_scan_bins, dim3(4), dim3(1024), 0, s, bs, nL1);
  }
  if (!fused) {
    // level 2: one workgroup per super-bucket
    const size_t lds2 = std::max(b2_lds_bytes<u32>(1u << sbS), b2_lds_bytes<u64>(1u << sbS));
    if (ctx->b2LdsSet != lds2) {
      HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket2p), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      ctx->b2LdsSet = lds2;
    }
    // (the F stream: multimapped reads, or everything beyond 4.29 Gbp; a run without them finds every bin empty)
    Bucket2Jobs BJ{{{PG3[0], SS.a.p, SS.sbOff.as<u32>(), ctx->tileCnt[0].as<u32>()},
                    {PG3[1], SE.a.p, SE.sbOff.as<u32>(), ctx->tileCnt[1].as<u32>()},
                    {PG3[2], SF.a.p, SF.sbOff.as<u32>(), ctx->tileCnt[2].as<u32>()}}};
    hipLaunchKernelGGL(k_bucket2p, dim3(std::max(1u, nL1), 3), dim3(B2_NT), lds2, s, BJ, nL1, sbS, nTiles,
                       ctx->tileWsum.as<int>());
    if (int rc__ = dbg_sync(ctx, "k_bucket2p")) return rc__;
  }
  TileTabs tt{};
  for (int q = 0; q < 3; q++) {
    tt.cnt[q] = ctx->tileCnt[q].as<u32>();
    tt.off[q] = ctx->tileOff[q].as<u32>();
  }
  tt.wsumF = ctx->tileWsum.as<int>();
  tt.prefW = ctx->tileCarry.as<int>();
  if (!fused) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s, tt, nTiles,
                       ctx->lb.as<u64>(), ctx->lb.as<u64>() + tChunks + 2, ctx->lb.as<u64>() + 2 * (tChunks + 2),
                       ctx->dStatus.as<u32>());
    if (int rc__ = dbg_sync(ctx, "k_scan_tiles")) return rc__;
  }
  HIPCHECK(pooled(ctx, ctx->looseEnd, looseCap * 4));  // (pooled: a sample stashed for its control merge took the last ones along)
  HIPCHECK(pooled(ctx, ctx->looseV, looseCap * 4));
  HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->tileLastEnd.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->tilePrevEnd.ensure((size_t)(nTiles + 1) * 4));
  Scalars* ds = ctx->dScal.as<Scalars>();
  long long* acc = isCtrl ? ds->ctrlAcc : ds->fragAcc;  // zero since gx_sample_begin(treatment)
  TileOut to{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(),
             ctx->tileDeep.as<u32>(), sigMask, wantEarly || wantLate ? ctl : (LooseCtl*)nullptr};   // (late: `enabled` stays 0 -- no bits, no
                                                                                                    // fillers; a pileup beyond the table still says so)
  BedIn bin{ctx->dBedTileOff.as<u32>(), ctx->dBedEdge.as<u32>(), ctx->dTileSave0.as<uint8_t>()};
  HIPCHECK(pooled(ctx, ctx->tileMeta, (size_t)(nTiles + 1) * sizeof(TileMeta)));
  HIPCHECK(ctx->wideList.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(ctx->heavyList.ensure((size_t)(nTiles + 1) * 4));
  if (!fused)
    hipLaunchKernelGGL(k_tile_meta, dim3((nTiles + 255) / 256), dim3(256), 0, s, ctx->tileOff[0].as<u32>(),
                       ctx->tileOff[1].as<u32>(), ctx->tileOff[2].as<u32>(), ctx->tileCarry.as<int>(), ctx->dTileChrom.as<u32>(),
                       ctx->dChrom.as<DChrom>(), ctx->hasBed ? ctx->dBedTileOff.as<u32>() : (const u32*)nullptr, nTiles,
                       ctx->tileMeta.as<TileMeta>(), ctx->wideList.as<u32>(), ctx->nWide.as<u32>(), ctx->tileSlot.as<u32>(),
                       ctx->hasBed ? (u32*)nullptr : ctx->heavyList.as<u32>());
  phase_end(ctx);

  phase_begin(ctx, isCtrl ? "c.tile" : "t.tile");  // k_tile alone: the dominant kernel (bench.py's roofline)
  TileIn tin{SS.a.as<uint16_t>(), SE.a.as<uint16_t>(), SF.a.as<u64>(), ctx->tileMeta.as<TileMeta>()};
  // the tile stage is k_tile_fast (+ k_tile_heavy): the general fragLen path's terms ride in it (TileIn::fragAcc)
  // (-E regions: k_frag_walk's general path walks every interval, on either chain)
  ctx->fragFused = !ctx->hasBed && (!fused || ctx->fracPairsUsed);
  if (ctx->fragFused) {
    tin.ff = ff;
    tin.fragAcc = acc;
  }
  // narrow tiles with 16-bit LDS counters (twice the tiles in flight), then the wide ones from their list
  // (whose length stays on the device: an empty list costs one idle launch)
  const u32* wl = ctx->wideList.as<u32>();
  const u32* nw = ctx->nWide.as<u32>();
  const dim3 gHalf(std::min<u32>(nTiles, (u32)ctx->resTileHalf)), gWide(std::min<u32>(nTiles, (u32)ctx->resTile));
  if (fused) {
    // level 2 of the sort and the tile passes in one kernel, one workgroup per super-bucket (gx_sbtile.h)
    if (!ctx->sbtLdsSet) {
      for (const void* f : {reinterpret_cast<const void*>(k_sbtile<false, false, false>), reinterpret_cast<const void*>(k_sbtile<true, false, false>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, false>), reinterpret_cast<const void*>(k_sbtile<true, false, true>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, true>), reinterpret_cast<const void*>(k_sbtile<true, true, false, SBT_TR_DENSE>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, true, SBT_TR_DENSE>),
                            reinterpret_cast<const void*>(k_sbtile<true, false, false, SBT_TR, true>), reinterpret_cast<const void*>(k_sbtile<true, true, false, SBT_TR, true>),
                            reinterpret_cast<const void*>(k_sbtile<true, true, false, SBT_TR_DENSE, true>)})
        HIPCHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SBT_LDS_BYTES));
      ctx->sbtLdsSet = true;
    }
    HIPCHECK(ctx->bigBins.ensure((size_t)(MAX_BINS_P + 4) * 4));
    SbtIn si{PG3[0], PG3[1], PG3[2], SS.sbOff.as<u32>(), SE.sbOff.as<u32>(), SF.sbOff.as<u32>(), ctx->dTileChrom.as<u32>(),
             ctx->dChrom.as<DChrom>(), ctx->chromW0.as<int>(), nL1, nTiles, sbS,
             ctx->fragFused && ctx->fracPairsUsed ? (const FragFix*)ff : (const FragFix*)nullptr,
             ctx->fragFused && ctx->fracPairsUsed ? acc : (long long*)nullptr, ctx->hasBed ? bin : BedIn{nullptr, nullptr, nullptr},
             ctx->hasBed ? ff->fragSum : (u64*)nullptr};
    SbtOut so2{to, ctx->tileMeta.as<TileMeta>(), ctx->tileSlot.as<u32>(), ctx->nWide.as<u32>() + 1, ctx->nWide.as<u32>() + 13,
               ctx->bigBins.as<u32>(), ctx->heavyList.as<u32>(), ctx->nWide.as<u32>() + 2};
    const dim3 gAll(std::max(1u, nL1)), gBig(std::max(1u, std::min(nL1, (u32)ctx->numCU)));
    // a sample so dense that the average bin already holds more keys than the key array (ATAC cut sites of a deep
    // library): every bin takes the rounds of the second launch, the first one would only find that out bin by bin
    const bool dense = ctx->pairsUsed && (size_t)2 * nEv > (size_t)std::max(1u, nL1) * (SBT_KEYCAP - SBT_KEYCAP / 16);
    if (dense) {
      so2.bigList = nullptr;
      // touched bases per round of the tile passes against keys per round of a bin (they share the LDS, gx_sbtile.h): what
      // costs a dense sample is the number of rounds a bin takes -- each reads the bin's records again --, so: the
      // instance with the smaller scratch when that saves the AVERAGE bin a round.  (No margin for the fuller bins: they
      // take the extra round in either instance.  Config 4, 97.8 K keys per bin against 2 x 50,048: tile stage 2.17 ms
      // with the small scratch, 2.49 with the large one, three runs each.)
      const size_t want = (size_t)2 * nEv / std::max(1u, nL1);
      auto rounds = [&](u32 tr) { return (want + sbt_keycap(tr) - 1) / sbt_keycap(tr); };
      bool small = rounds((u32)SBT_TR_DENSE) < rounds((u32)SBT_TR);
      if (K.sbtTr) small = K.sbtTr == SBT_TR_DENSE;   // (measurements)
      if (ctx->fracPairsUsed) {
        if (small) hipLaunchKernelGGL((k_sbtile<true, true, true, SBT_TR_DENSE>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
        else hipLaunchKernelGGL((k_sbtile<true, true, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      } else if (ctx->hasBed) {
        if (small) hipLaunchKernelGGL((k_sbtile<true, true, false, SBT_TR_DENSE, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
        else hipLaunchKernelGGL((k_sbtile<true, true, false, SBT_TR, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      } else {
        if (small) hipLaunchKernelGGL((k_sbtile<true, true, false, SBT_TR_DENSE>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
        else hipLaunchKernelGGL((k_sbtile<true, true, false>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      }
    } else if (ctx->fracPairsUsed) {
      hipLaunchKernelGGL((k_sbtile<true, false, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      hipLaunchKernelGGL((k_sbtile<true, true, true>), gBig, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
    } else if (ctx->pairsUsed && ctx->hasBed) {
      // (-E regions: the bins with an edge tile are the second launch's as well -- every second bin of hg38 with ~800 regions, so
      // a workgroup per bin of the grid, dealt by the dispatcher; the ones beyond the list leave at once)
      hipLaunchKernelGGL((k_sbtile<true, false, false, SBT_TR, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      hipLaunchKernelGGL((k_sbtile<true, true, false, SBT_TR, true>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
    } else if (ctx->pairsUsed) {
      hipLaunchKernelGGL((k_sbtile<true, false, false>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
      // the bins it left on its list (reads piled up: more keys than the key array holds, a tile with thousands of keys):
      // usually none -- an idle launch
      hipLaunchKernelGGL((k_sbtile<true, true, false>), gBig, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
    } else
      hipLaunchKernelGGL((k_sbtile<false, false, false>), gAll, dim3(SBT_NT), SBT_LDS_BYTES, s, si, so2, ctx->dStatus.as<u32>());
  } else if (ctx->hasBed) {
    hipLaunchKernelGGL((k_tile<true, true>), gHalf, dim3(TL_NT), TL_LDS_HALF * 4, s, tin, nTiles, wl, nw, bin, to,
                       ctx->dStatus.as<u32>());
    hipLaunchKernelGGL((k_tile<true, false>), gWide, dim3(TL_NT), TL_LDS * 4, s, tin, nTiles, wl, nw, bin, to,
                       ctx->dStatus.as<u32>());
  } else {
    // the common case: one wavefront per tile, work laid out by touched base, unit-weight and fractional records
    // alike (gx_tile_fast.h)
    hipLaunchKernelGGL(k_tile_fast, dim3(std::min<u32>(nTiles, (u32)ctx->resTileFast)), dim3(64), 0, s, tin, nTiles, nw, to,
                       ctx->dStatus.as<u32>());
    // the tiles with thousands of records (pile-ups): a workgroup each, a counter per base (usually none: an idle launch)
    hipLaunchKernelGGL(k_tile_heavy, dim3(64), dim3(TH_NT), 0, s, tin, ctx->heavyList.as<u32>(), nw + 2, to, ctx->dStatus.as<u32>());
  }
  if (int rc__ = dbg_sync(ctx, "k_tile")) return rc__;
  phase_end(ctx);
  // (word 1 of the nWide block: the "a base can reach the int16 limits" flag, also set by k_convert)
  // (a bin that fits k_sbtile holds fewer than 32,767 records of a stream: no base of it can reach the limits)
  // (a tile that can hold such a base has >= 32,766 records: it is on the list of the heavy tiles -- walking the list of
  // the WIDE tiles instead cost config 4, where every tile holds fractional records and is "wide", 2.1 ms of header reads)
  if (!fused) {
    const bool haveHeavy = !ctx->hasBed;
    hipLaunchKernelGGL(k_hot_check, dim3(std::min<u32>(nTiles, 256u)), dim3(256), 0, s, tin, haveHeavy ? ctx->heavyList.as<u32>() : wl,
                       haveHeavy ? nw + 2 : nw, ctx->nWide.as<u32>() + 1);
  }
  if (int rc__ = dbg_sync(ctx, "k_hot_check")) return rc__;

  phase_begin(ctx, isCtrl ? "c.pack" : "t.pack");
  const u32 ivChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  IvScanOut so{out.tileIvOff.as<u32>(), ctx->tilePrevEnd.as<u32>(), out.chromIvOff.as<u32>(), ctx->misc.as<u32>() + M_NIV,
               ctx->tileSlot.as<u32>(), ctx->chromLooseOff.as<u32>(), ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctl,
               ctx->tileDeep.as<u32>(), ff, ctx->fragList.as<u32>(), ctx->fragFused ? acc : (long long*)nullptr,
               ctx->endAtLen.as<u32>()};
  const bool closeInScan = wantEarly && !multiRank;  // (k_scan_iv_close, below)
  if (!closeInScan)
    hipLaunchKernelGGL(k_scan_iv, dim3(std::min<u32>(ivChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                       ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->dTileChrom.as<u32>(),
                       ctx->dChrom.as<DChrom>(), nTiles, ctx->lbIv.as<u64>(), ctx->lbIv.as<u64>() + ivChunks + 1, so,
                       ctx->dStatus.as<u32>());
  if (int rc__ = dbg_sync(ctx, "k_scan_iv")) return rc__;
  {
    const u32* lE = ctx->looseEnd.as<u32>();
    const int* lV = ctx->looseV.as<int>();
    const TileMeta* tm = ctx->tileMeta.as<TileMeta>();
    const u32* tOff = out.tileIvOff.as<u32>();
    const u32* tPrev = ctx->tilePrevEnd.as<u32>();
    // (k_frag_fix1's pass over the tiles -- deep-tile list, long first intervals -- rides in k_scan_iv)
    FragSelect fsel{ff, acc, ctx->world > 1 || ctx->forceColl ? ctx->dColl.as<long long>() : (long long*)nullptr,
                    ctx->nWide.as<u32>() + 1, ctx->dStatus.as<u32>(), ctx->dChrom.as<DChrom>(), nChrom, out.chromIvOff.as<u32>(),
                    ctx->misc.as<u32>() + M_NIV, ds, isCtrl, ctx->chromLooseOff.as<u32>(), ctx->tileSlot.as<u32>(), nTiles, ctl,
                    wantEarly || wantLate ? ctx->swMask.as<u64>() + ctx->looseStride : (u64*)nullptr};
    ctx->closeSel = fsel;
    ctx->closeSeq = 0;
    if (closeInScan) {
      // lambda was known before the tile stage: k_frag_select's work and the mail ride in the scan's launch; if a deep tile,
      // the general fragLen path or a changed lambda stands in the way, finish_scalars runs the separate kernels after all
      ctx->closeSeq = ++ctx->mailSeq;
      ctx->mail->nMerged = 0;
      ctx->mail->closeState = 0;
      HostMail* dm = static_cast<HostMail*>(ctx->mailBuf.dp);
      {
        // (the scan's last workgroup closes the sample: one launch)
        CloseArgs ca{fsel, ctx->misc.as<u32>() + M_NIV, (const u32*)&ctl->ok, ctx->dRisk.as<RiskBuf>(), mail_out(ctx),
                     &dm->closeState, ctx->closeSeq, ctx->nWide.as<u32>() + 8};
        hipLaunchKernelGGL(k_scan_iv_close, dim3(std::min<u32>(ivChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                           ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->dTileChrom.as<u32>(),
                           ctx->dChrom.as<DChrom>(), nTiles, ctx->lbIv.as<u64>(), ctx->lbIv.as<u64>() + ivChunks + 1, so,
                           ctx->dStatus.as<u32>(), ca);
      }
      if (int rc__ = dbg_sync(ctx, "k_close")) return rc__;
    } else {
    hipLaunchKernelGGL(k_frag_walk, dim3(std::max(1u, std::min((nTiles + 3) / 4, 4096u))), dim3(256), 0, s, lE, lV, tm, tOff,
                       tPrev, nTiles, ff, ctx->fragList.as<u32>(), acc,
                       ctx->fragFused ? ctx->heavyList.as<u32>() : (const u32*)nullptr, ctx->nWide.as<u32>() + 2);
    // (single thread: chromosome offsets of the chromosomes without tiles, closed form -> accumulator pair,
    // this rank's words of the all-reduce)
    hipLaunchKernelGGL(k_frag_select, dim3(1), dim3(1), 0, s, fsel);
    }
  }
  if (int rc__ = dbg_sync(ctx, "k_frag")) return rc__;
  out.packed = false;
  out.inLoose = false;
  if (isCtrl) {  // a control is always merged against the treatment
    int rc = stash_or_pack(ctx, out);
    if (rc) return rc;
  }
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  ctx->nIvTarget = &out.nIv;  // filled from the mail block once finish_scalars has synchronised
  return GX_OK;
}

constexpr int RETRY_GENERAL = 3;    // (internal) k_sbtile could not take the sample: build it again on the general chain
constexpr int RETRY_SATURATED = 1;  // (internal) finish_scalars: filter the events and build the sample again
constexpr int RETRY_PT = 2;         // (internal) a level-1 page list overflowed: build again with a longer page table
// (the page tables -- NXCD x bins x jmax x 4 bytes, three streams -- at the cap and hg38's 2,946 bins: 18.5 GB, which a
// 288 GB device holds; 2^20, round 2's cap, would have asked for 50 GB per stream.  A list beyond 2^16 pages holds
// more than 5 x 10^8 keys of ONE super-bucket: such a sample fails with "could not be rebuilt")
constexpr u32 PT_JMAX_CAP = 1u << 16;

// fragLen / ctrlFrag partial sums -> (all ranks) -> lambda, factor
int finish_scalars(gx_ctx* ctx, int isCtrl) {
  hipStream_t s = ctx->stream;
  Scalars* ds = ctx->dScal.as<Scalars>();
  const bool multi = ctx->world > 1 || ctx->forceColl;
  long long* dcoll = multi ? ctx->dColl.as<long long>() : nullptr;
  if (multi) {
    // The third word sums the ranks' "build this sample again" flags, so that every rank learns from the one
    // synchronisation below whether the sums are final.
    if (int rc__ = allreduce_words(ctx, dcoll, 3)) return rc__;
    ctx->earlyPending = false;
    // (one rank: k_frag_select has done it).  With lambda known to every rank before the tile stage (the early
    // all-reduce of build_pileup), this is also where a rank learns whether its sweep bits were written with the
    // lambda that turned out final.
    hipLaunchKernelGGL(k_finish_frag, dim3(1), dim3(1), 0, s, ds, isCtrl, ctx->dStatus.as<u32>(), (const long long*)dcoll,
                       !isCtrl && ctx->earlyColl ? ctx->looseCtl.as<LooseCtl>() : (LooseCtl*)nullptr);
    if (int rc__ = dbg_sync(ctx, "k_finish_frag")) return rc__;
  }
  bool closed = false;
  if (ctx->closeSeq) {
    // k_close has sent the mail (build_pileup); only if something stood in its way do the separate kernels run
    if (int rc__ = mail_wait(ctx, ctx->closeSeq)) return rc__;
    ctx->closeSeq = 0;
    closed = ctx->mail->closeState == 1;
    if (!closed) {
      const u32 nTiles = ctx->nTiles;
      hipLaunchKernelGGL(k_frag_walk, dim3(std::max(1u, std::min((nTiles + 3) / 4, 4096u))), dim3(256), 0, s, ctx->looseEnd.as<u32>(),
                         ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), ctx->expt.tileIvOff.as<u32>(),
                         ctx->tilePrevEnd.as<u32>(), nTiles, ctx->fragSum.as<FragFix>(), ctx->fragList.as<u32>(), ctx->closeSel.acc,
                         ctx->fragFused ? ctx->heavyList.as<u32>() : (const u32*)nullptr, ctx->nWide.as<u32>() + 2);
      hipLaunchKernelGGL(k_frag_select, dim3(1), dim3(1), 0, s, ctx->closeSel);
      if (int rc__ = dbg_sync(ctx, "k_frag (after k_close)")) return rc__;
    }
  }
  if (!closed) {
  // lambda (and with a control the factor) is final: build the p-value tables now, so that the values the
  // host has to re-evaluate (risky ones) travel with the synchronisation that returns the scalars
  if (!isCtrl) {
    PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), ctx->expt.tileIvOff.as<u32>()};
    // (when the tile stage had lambda already -- LooseCtl -- and it has not changed, only the deep tiles' part runs)
    hipLaunchKernelGGL(k_pval_lut, dim3(PV_LUT / 256 + DEEP_BLOCKS), dim3(256), 0, s, ds, ctx->pvLut.as<float>(),
                       ctx->dRisk.as<RiskBuf>(), ctx->dDeep.as<DeepTab>(), pin, ctx->fragSum.as<FragFix>(),
                       ctx->fragList.as<u32>(), ctx->looseCtl.as<LooseCtl>(), ctx->lateLoose ? 2 : 0, ctx->par.thr);
    // (lambda came with the sample's end: may the sweep walk the loose slots?  The pass that writes its bits runs when gx_find_peaks
    // finds the replicate to be the run's only one -- gx_stats.h k_loose_late)
    if (ctx->lateLoose) hipLaunchKernelGGL(k_loose_verdict, dim3(1), dim3(256), 0, s, ctx->looseCtl.as<LooseCtl>());
  } else {
    hipLaunchKernelGGL(k_pair_tabs, dim3(PAIR_LUT / 256), dim3(256), 0, s, ds, ctx->pairLogE.as<double>(),
                       ctx->pairCtab.as<CtrlEntry>());
    hipLaunchKernelGGL(k_pair_tab2d, dim3(PT_N * PT_N / 256), dim3(256), 0, s, ds, ctx->pairLogE.as<double>(),
                       ctx->pairCtab.as<CtrlEntry>(), ctx->pairP2d.as<float>(), ctx->dRisk.as<RiskBuf>());
    ctx->pairTabsReady = true;
  }
  if (int rc__ = dbg_sync(ctx, "p-value tables")) return rc__;
  ctx->mail->nMerged = 0;
  if (int rc__ = mail_sync(ctx, ds, ctx->nWide.as<u32>() + 1, ctx->misc.as<u32>() + M_NIV, dcoll,
                           isCtrl ? (const u32*)nullptr : &ctx->looseCtl.as<LooseCtl>()->ok))
    return rc__;
  }
  ctx->hScal = ctx->mail->scal;
  if (ctx->hScal.fracSeen) ctx->sawFrac = true;  // (learned, not hinted: the next sample does without the early lambda)
  ctx->riskNearThr = false;
  const int rcRisk = risk_apply(ctx, RiskTargets{});
  if (!isCtrl) ctx->looseOk = ctx->mail->nMerged != 0 && !ctx->riskNearThr;
  // (with several ranks: if any of them has to rebuild its sample, all go round again with it)
  const long long again = multi ? ctx->mail->coll[2]
                                : (long long)(ctx->mail->hot ? 1 : 0) + ((ctx->mail->status & ST_PT_FULL) ? 65536 : 0) +
                                      ((ctx->mail->status & (ST_SB_FULL | ST_SB_FRAC)) ? (1ll << 32) : 0);
  if (again >> 48) {
    ctx->err = "another rank could not build its sample";
    return GX_ERR_DEVICE;
  }
  if (again >> 32) {
    // k_sbtile could not take some rank's sample (a bin beyond its LDS, or fractional weights): once more, on the
    // general chain
    // (fractional weights in a unit-weight build: its singles may also have overfilled a bin -- that says nothing about
    // the next sample, which writes pair records with a weight class)
    if (ctx->knob.debugRetry) fprintf(stderr, "[gx] sample sent back to the general chain: status %u (fused %d pairs %d frac %d)\n",
                                          ctx->mail->status, (int)ctx->fusedUsed, (int)ctx->pairsUsed, (int)ctx->fracPairsUsed);
    if (ctx->mail->status & ST_SB_FRAC) ctx->sawFrac = true;
    else if (ctx->mail->status & ST_SB_FULL) ctx->fusedBackoff[isCtrl ? 1 : 0] = 8;
    ctx->fusedOff = true;
    ctx->fellBack = true;
    static_cast<RiskBuf*>(ctx->riskHost.p)->count = 0;
    HIPCHECK(hipMemsetAsync(ctx->dRisk.p, 0, 4, s));
    return RETRY_GENERAL;
  }
  if ((again & 0xFFFFFFFFll) >= 65536 && ctx->ptJmax < PT_JMAX_CAP) return RETRY_PT;
  int rc = status_to_rc(ctx, ctx->mail->status);
  if ((again & 0xFFFF) && !ctx->satDone) return RETRY_SATURATED;
  if (ctx->nIvTarget) *ctx->nIvTarget = ctx->mail->nIv;
  ctx->nIvTarget = nullptr;
  return rc ? rc : rcRisk;
}

// The sample holds a base that can reach the reference's int16 limits: bring the events to the host,
// drop the ones saveInterval would drop (gx_saturate.h) and stage what is left for a second build.
int drop_saturated(gx_ctx* ctx, int isCtrl) {
  hipStream_t s = ctx->stream;
  if (int rc__ = unpack_segs(ctx, true)) return rc__;   // (the replay reads gx_event records)
  size_t total = 0;
  for (auto& sg : ctx->segs) total += sg.n;
  std::vector<gx_event> all(total);
  size_t at = 0;
  HIPCHECK(hipStreamSynchronize(ctx->side));  // (the uploads have long arrived: the sample was built once)
  for (auto& sg : ctx->segs) {  // in push order: the replay depends on it
    if (sg.n) HIPCHECK(hipMemcpyAsync(all.data() + at, sg.p, sg.n * sizeof(gx_event), hipMemcpyDeviceToHost, s));
    at += sg.n;
  }
  HIPCHECK(hipStreamSynchronize(s));
  // only the chromosomes this context works on (the others' events are ignored by k_convert too)
  std::vector<uint32_t> len(ctx->nChrom);
  for (u32 i = 0; i < ctx->nChrom; i++) len[i] = ctx->hChrom[i].tileBase == NULL_TILE ? 0u : ctx->len[i];
  std::vector<uint8_t> keep(total);
  const long long dropped = gxsat::filter(all.data(), total, (int)ctx->nChrom, len.data(), keep.data());
  ctx->satDropped = dropped > 0 ? dropped : 0;
  size_t kept = 0;
  if (dropped > 0) {
    for (size_t i = 0; i < total; i++)
      if (keep[i]) all[kept++] = all[i];
  } else
    kept = total;
  // (also when nothing was dropped: the second build must not see the caller's segments twice)
  HIPCHECK(ctx->satBuf.ensure(std::max<size_t>(kept, 1) * sizeof(gx_event)));
  if (kept) HIPCHECK(hipMemcpyAsync(ctx->satBuf.p, all.data(), kept * sizeof(gx_event), hipMemcpyHostToDevice, s));
  HIPCHECK(hipStreamSynchronize(s));  // `all` goes out of scope
  ctx->segs.clear();
  if (kept) ctx->segs.push_back({ctx->satBuf.as<gx_event>(), kept, nullptr});
  ctx->satDone = true;
  // what the first build left behind: its status bits and its contribution to fragLen / ctrlFrag
  Scalars* ds = ctx->dScal.as<Scalars>();
  HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, s));
  HIPCHECK(hipMemsetAsync(isCtrl ? ds->ctrlAcc : ds->fragAcc, 0, 16, s));
  return GX_OK;
}

int close_sample(gx_ctx* ctx, Pileup& P, int isCtrl) {
  ctx->fusedOff = false;
  if (!isCtrl) ctx->looseOk = false;
  auto wipe = [&]() -> int {  // what a build that is repeated left behind: status bits, its part of fragLen / ctrlFrag
    Scalars* ds = ctx->dScal.as<Scalars>();
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
    HIPCHECK(hipMemsetAsync(isCtrl ? ds->ctrlAcc : ds->fragAcc, 0, 16, ctx->stream));
    return GX_OK;
  };
  bool reuseSort = false;
  for (int attempt = 0; attempt < 12; attempt++) {
    int rc = build_pileup(ctx, P, isCtrl, reuseSort);
    reuseSort = false;
    if (rc) {
      // With several ranks the others are about to wait for this one in the fragLen all-reduce: take part in it with a
      // "this rank has failed" word, so that every rank returns an error instead of one returning and the rest hanging.
      const std::string why = ctx->err;
      poison_allreduce(ctx);
      ctx->err = why;
      return rc;
    }
    rc = finish_scalars(ctx, isCtrl);
    if (rc == RETRY_GENERAL) {
      // k_sbtile could not take the sample (finish_scalars has switched it off for this one): the general chain,
      // on the pages level 1 of the sort has already filled
      if (int w = wipe()) return w;
      // (pair records are of no use to the general chain: level 1 runs again as start / end keys)
      reuseSort = !ctx->pairsUsed;
      if (reuseSort) {
        // (k_sort1 does not run again: the status bits IT raised -- bad counts, positions, chromosomes -- must survive)
        // (and only those: what the abandoned tile stage raised -- e.g. "negative pileup" from carries that count the
        // dropped ends of fractional records it never saw -- means nothing)
        ctx->mail->statusKeep = ctx->mail->status & (ST_BAD_CHROM | ST_BAD_POS | ST_BAD_COUNT | ST_PT_FULL | ST_LOOKBACK);
        HIPCHECK(hipMemcpyAsync(ctx->dStatus.p, &ctx->mail->statusKeep, 4, hipMemcpyHostToDevice, ctx->stream));
      }
    } else if (rc == RETRY_PT) {
      // a (XCD class, super-bucket) list needed more pages than its table row holds -- reads piled up in one
      // spot: what the first build left behind goes, the table grows, the sample is built again
      // ... to what the longest list asked for (k_scan_bins: the cursors count every reservation), with a quarter to
      // spare -- not by a blind factor: the table is NXCD x bins x jmax words per stream, cleared for every sample
      u32 need = 0;
      HIPCHECK(hipMemcpy(&need, ctx->nWide.as<u32>() + 12, 4, hipMemcpyDeviceToHost));
      u32 want = std::max(ctx->ptJmax * 2, need + need / 4 + 2);
      ctx->ptJmax = std::min(want, PT_JMAX_CAP);
      ctx->ptGrew = true;
      if (int w = wipe()) return w;
    } else if (rc == RETRY_SATURATED) {
      if ((rc = drop_saturated(ctx, isCtrl))) return rc;  // (sets satDone: finish_scalars asks for this once)
    } else
      return rc;
  }
  ctx->err = "sample could not be rebuilt";
  return GX_ERR_DEVICE;
}

// tile space, super-buckets, -E edge lists and the chromosome table for the chromosomes this
// context works on: not skipped (-e), not empty, and owned by this rank (gx_set_owned)
int layout_tiles(gx_ctx* ctx) {
  const int n = (int)ctx->nChrom;
  const std::vector<uint32_t>& len = ctx->len;
  ctx->hChrom.assign(n, DChrom{});
  std::vector<u32> tileChrom;
  u32 t = 0;
  for (int i = 0; i < n; i++) {
    DChrom& c = ctx->hChrom[i];
    c.len = len[i];
    if (ctx->skip[i] || !ctx->owned[i] || len[i] == 0) {
      c.tileBase = NULL_TILE;
      c.nTiles = 0;
      continue;
    }
    c.tileBase = t;
    c.nTiles = (u32)(((uint64_t)len[i] + TILE - 1) >> TB);
    for (u32 k = 0; k < c.nTiles; k++) tileChrom.push_back((u32)i);
    t += c.nTiles;
  }
  if (t == 0) {
    // a rank that owns nothing still needs a (dormant) tile space: the first analyzable chromosome's
    for (int i = 0; i < n && t == 0; i++)
      if (!ctx->skip[i] && len[i] != 0) {
        DChrom& c = ctx->hChrom[i];
        c.tileBase = 0;
        c.nTiles = (u32)(((uint64_t)len[i] + TILE - 1) >> TB);
        tileChrom.assign(c.nTiles, (u32)i);
        t = c.nTiles;
      }
  }
  ctx->nTiles = t;
  if (t == 0) {
    ctx->err = "No analyzable genome (length=0)";
    return GX_ERR_GEN;
  }
  int lg = 0;
  while ((1u << lg) < t) lg++;
  // tiles per super-bucket: the level-1 scatter wants few bins (long runs per bin and chunk); level 2 wants
  // a super-bucket's keys to fit its one-pass LDS sort (45 K keys: ~2^9 tiles at hg38 / 50 M fragments) and
  // enough super-buckets for every CU; GX_SBSHIFT overrides for experiments
  // (k_sbtile, the fused level 2 + tile kernel, takes super-buckets of up to 2^8 tiles: hg38 = 2,946 bins)
  ctx->sbShift = std::min(SBT_MAXSHIFT, std::max(0, (lg - 1) / 2));
  if (ctx->knob.sbShift >= 0) ctx->sbShift = std::max(0, std::min(11, ctx->knob.sbShift));
  while (((t + (1u << ctx->sbShift) - 1) >> ctx->sbShift) + 1 > (u32)MAX_BINS) ctx->sbShift++;
  if ((1u << ctx->sbShift) > (u32)MAX_BINS) {
    ctx->err = "genome too large for the two-level tile sort";
    return GX_ERR_MEM;
  }
  ctx->nSB = ((t + (1u << ctx->sbShift) - 1) >> ctx->sbShift) + 1;  // + the null bucket
  // -E edges per tile (Genrich.c:2185-2195: a region starting at 0 only flips the initial state)
  {
    std::vector<u32> bedOff(t + 1, 0), edges;
    std::vector<uint8_t> save0(t, 1);
    ctx->hasBed = false;
    for (int i = 0; i < n; i++) {
      const DChrom& c = ctx->hChrom[i];
      if (c.tileBase == NULL_TILE) continue;
      const std::vector<uint32_t>& b = ctx->bed[i];
      if (!b.empty()) ctx->hasBed = true;
      bool state = b.empty() || b[0] != 0;
      size_t k = (!b.empty() && b[0] == 0) ? 1 : 0;
      for (u32 tl = 0; tl < c.nTiles; tl++) {
        const uint64_t lo = (uint64_t)tl << TB, hi = lo + TILE;
        save0[c.tileBase + tl] = state;
        bedOff[c.tileBase + tl] = (u32)edges.size();
        while (k < b.size() && b[k] < hi && b[k] < c.len) {
          edges.push_back((u32)(b[k] - lo));
          state = !state;
          k++;
        }
      }
    }
    bedOff[t] = (u32)edges.size();
    ctx->nBedEdges = edges.size();
    HIPCHECK(ctx->dBedTileOff.ensure((size_t)(t + 1) * 4));
    HIPCHECK(ctx->dBedEdge.ensure(edges.size() * 4 + 16));
    HIPCHECK(ctx->dTileSave0.ensure((size_t)t + 16));
    HIPCHECK(hipMemcpy(ctx->dBedTileOff.p, bedOff.data(), (size_t)(t + 1) * 4, hipMemcpyHostToDevice));
    if (!edges.empty()) HIPCHECK(hipMemcpy(ctx->dBedEdge.p, edges.data(), edges.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ctx->dTileSave0.p, save0.data(), (size_t)t, hipMemcpyHostToDevice));
  }
  HIPCHECK(ctx->dChrom.ensure((size_t)n * sizeof(DChrom)));
  HIPCHECK(ctx->dTileChrom.ensure((size_t)t * 4));
  HIPCHECK(hipMemcpyAsync(ctx->dTileChrom.p, tileChrom.data(), (size_t)t * 4, hipMemcpyHostToDevice, ctx->stream));
  return upload_chroms(ctx);
}

}  // namespace
