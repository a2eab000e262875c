// gx_host_stats.h -- the p-value side: the tight interval table of a replicate without control, its pileup floats on request,
// the merge with a control, the Fisher combination of replicates, Benjamini-Hochberg.
// (a part of gx_api.hip's translation unit: the kernels are templates and inline functions of the headers it includes;
// split by phase -- context / build / stats / sweep / collectives -- in round 5)
#pragma once
#ifndef GX_BH_DEFAULT_VARIANT
#define GX_BH_DEFAULT_VARIANT 1
#endif
namespace {

// Loose slots -> the tight interval table (end, p[, pileups]) of a replicate without control: savePval
// (Genrich.c:1720-1794) against the constant control lambda.  Needs the sample's loose slots, tile tables and
// p(V) table, i.e. must run before the next sample is built (gx_sample_begin sees to that).
// The table of distinct p-values: open addressing, 2^bhCapLog slots.  It starts at 2^22 (16 MiB of keys: L2 /
// Infinity-Cache resident for the per-interval look-ups) and grows by 8x, for good, whenever an insertion
// gives up (ST_HASH_FULL: bh_global_add stops after BH_MAX_PROBE steps instead of crawling through a full
// table) -- the reference's chained hash (recordPval 277-295) has no limit either.
BhTable bh_table_of(gx_ctx* ctx, u32 cap);
// (round 6: the table of the last -q run stays -- the {key, q} pairs answer the q-values nobody has asked for yet, ensure_q -- and is
// freed here, when the next run wants it)
void bh_release_live(gx_ctx* ctx) {
  if (ctx->bhLive && !ctx->bhDirty)
    hipLaunchKernelGGL(k_bh_clear, dim3(256), dim3(256), 0, ctx->stream, bh_table_of(ctx, ctx->bhLiveCap), ctx->bhKQ.as<u64>());
  ctx->bhLive = false;
  ctx->bhLiveIdx = -1;
}
int bh_table_prepare(gx_ctx* ctx, u32 c) {
  hipStream_t s = ctx->stream;
  bh_release_live(ctx);
  const bool fresh = ctx->bhKeys.cap < (size_t)c * 4;
  HIPCHECK(ctx->bhKeys.ensure((size_t)c * 4));
  HIPCHECK(ctx->bhLens.ensure((size_t)c * 8));
  HIPCHECK(ctx->bhQ.ensure((size_t)c * 4));
  const bool freshKQ = ctx->bhKQ.cap < (size_t)c * 8;   // ({key, q} side by side for k_qlookup: free slots hold ~0)
  HIPCHECK(ctx->bhKQ.ensure((size_t)c * 8));
  if (freshKQ || ctx->bhDirty) HIPCHECK(hipMemsetAsync(ctx->bhKQ.p, 0xFF, (size_t)c * 8, s));
  HIPCHECK(ctx->bhOutKeys.ensure((size_t)c * 4));
  HIPCHECK(ctx->bhOutSlot.ensure((size_t)c * 4));
  if (fresh || ctx->bhDirty) {  // normally the table comes back clean from the previous call (k_bh_clear)
    HIPCHECK(hipMemsetAsync(ctx->bhKeys.p, 0xFF, (size_t)c * 4, s));
    HIPCHECK(hipMemsetAsync(ctx->bhLens.p, 0, (size_t)c * 8, s));
  }
  ctx->bhDirty = true;
  return GX_OK;
}
BhTable bh_table_of(gx_ctx* ctx, u32 cap) {
  return BhTable{ctx->bhKeys.as<u32>(), ctx->bhLens.as<u64>(), cap - 1, ctx->bhOutKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                 ctx->misc.as<u32>() + M_BHCOUNT};
}

// hist: the replicate is the run's only one, without a control, and -q follows (gx_find_peaks knows): its "bp at V" histogram is
// made on the way (k_pack_pval<.., HIST>), for bh_qvalues
int materialize_rep(gx_ctx* ctx, int idx, bool hist = false) {
  PArray& pa = ctx->reps[idx];
  if (!pa.loose) return GX_OK;
  hipStream_t s = ctx->stream;
  const u32 n = pa.n;
  HIPCHECK(pooled(ctx, pa.p, (size_t)n * 4 + 16));
  phase_begin(ctx, "pval");
  // (the table p(V) was built when the treatment sample was closed: finish_scalars)
  PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), pa.tileOff.as<u32>()};
  // p-mode: the sweep's significance / skip masks are filled on the way (gx_find_peaks reuses them
  // when this replicate turns out to be the only one)
  u64 *sigM = nullptr, *skipM = nullptr;
  ctx->maskIdx = -1;
  if (!ctx->par.qval_opt) {
    const u32 nWords = (n + 63) / 64;
    HIPCHECK(ctx->swMask.ensure((size_t)(nWords + 2) * 8 * 3));
    HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, (size_t)(nWords + 2) * 8 * 3, s));
    sigM = ctx->swMask.as<u64>();
    skipM = sigM + (nWords + 2);
    ctx->maskIdx = idx;
    ctx->maskN = n;
    ctx->maskStride = nWords + 2;
  }
  {
    const dim3 grid(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
    if (hist) {
      const u32 cap = 1u << ctx->bhCapLog;
      if (int rc = bh_table_prepare(ctx, cap)) return rc;
      HIPCHECK(hipMemsetAsync(ctx->misc.as<u32>() + M_BHCOUNT, 0, 8, s));
      HIPCHECK(ctx->bhDense.ensure(bhd_words(1) * 8));
      HIPCHECK(hipMemsetAsync(ctx->bhDense.p, 0, bhd_words(1) * 8, s));
      // (as many workgroups as the CUs hold at once -- seven, with the histogram's LDS and registers: an eighth would run behind the others)
      int nbH = 0;
      HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbH, k_pack_pval<false, true>, 256, 0));
      const dim3 gridH(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(std::max(1, std::min(nbH, 8)) * ctx->numCU))));
      hipLaunchKernelGGL((k_pack_pval<false, true>), gridH, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                         ctx->pvLut.as<float>(), ctx->expt.ivEnd.as<u32>(), pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                         ctx->dStatus.as<u32>(), (const u32*)ctx->tilePrevEnd.as<u32>(), ctx->bhDense.as<u64>());
    } else if (sigM)
      hipLaunchKernelGGL((k_pack_pval<true>), grid, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                         ctx->pvLut.as<float>(), ctx->expt.ivEnd.as<u32>(), pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                         ctx->dStatus.as<u32>());
    else
      hipLaunchKernelGGL((k_pack_pval<false>), grid, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                         ctx->pvLut.as<float>(), ctx->expt.ivEnd.as<u32>(), pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                         ctx->dStatus.as<u32>());
  }
  hipLaunchKernelGGL(k_pval_deep, dim3(256), dim3(256), 0, s, pin, ctx->fragSum.as<FragFix>(), ctx->fragList.as<u32>(),
                     ctx->dScal.as<Scalars>(), ctx->dDeep.as<DeepTab>(), pa.p.as<float>(), ctx->par.thr, sigM);
  if (hist) {
    hipLaunchKernelGGL(k_deep_hist, dim3(64), dim3(256), 0, s, pin, ctx->fragSum.as<FragFix>(), ctx->fragList.as<u32>(),
                       (const u32*)ctx->tilePrevEnd.as<u32>(), (const float*)pa.p.as<float>(), bh_table_of(ctx, 1u << ctx->bhCapLog),
                       ctx->dStatus.as<u32>());
    ctx->denseHistIdx = idx;
  }
  if (int rc__ = dbg_sync(ctx, "k_pack_pval")) return rc__;
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  pa.end = std::move(ctx->expt.ivEnd);
  pa.hasPiles = false;
  pa.pilesPending = ctx->keepPiles;  // made when somebody asks (ensure_piles), from the exact pileups in the loose slots
  pa.pilesDropped = !ctx->keepPiles;
  pa.loose = false;
  return GX_OK;
}

// The pileup floats of a no-control replicate (Pileup.cov of the reference: only -f / -k print them): made on
// request from the exact pileups, while the sample's loose slots are still there.
int make_pair_piles(gx_ctx* ctx, int idx);
int ensure_piles(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  if (pa.pairPending) return make_pair_piles(ctx, idx);
  if (pa.loose)
    if (int rc = materialize_rep(ctx, idx)) return rc;
  if (!pa.pilesPending) return GX_OK;
  hipStream_t s = ctx->stream;
  HIPCHECK(pooled(ctx, pa.expt, (size_t)pa.n * 4 + 16));
  if (ctx->hasBed) HIPCHECK(pooled(ctx, pa.ctrl, (size_t)pa.n * 4 + 16));
  // (PackIn::end of a kept replicate would be a LATER sample's ends -- the replicate kept its pileups and descriptors only, and
  // k_piles_from_loose reads nothing else: a null pointer says so)
  PackIn pin{pa.keptLoose ? (const u32*)nullptr : ctx->looseEnd.as<u32>(), pa.keptLoose ? pa.keptV.as<int>() : ctx->looseV.as<int>(),
             pa.keptLoose ? pa.keptMeta.as<TileMeta>() : ctx->tileMeta.as<TileMeta>(), pa.tileOff.as<u32>()};
  const dim3 grid(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
  // (the control value of a replicate without control is its lambda: saveLambda 1847-1876)
  if (ctx->hasBed)
    hipLaunchKernelGGL(k_piles_from_loose<true>, grid, dim3(256), 0, s, pin, ctx->nTiles, pa.ctrlConst, pa.expt.as<float>(),
                       pa.ctrl.as<float>());
  else
    hipLaunchKernelGGL(k_piles_from_loose<false>, grid, dim3(256), 0, s, pin, ctx->nTiles, pa.ctrlConst, pa.expt.as<float>(),
                       (float*)nullptr);
  if (int rc__ = dbg_sync(ctx, "k_piles_from_loose")) return rc__;
  pa.hasPiles = true;
  pa.pilesPending = false;
  ctx->pilesMade = true;
  if (pa.keptLoose) {
    recycle(ctx, pa.keptV);
    recycle(ctx, pa.keptMeta);
    pa.keptLoose = false;
  }
  return GX_OK;
}

// The context's loose slots are about to be reused (a further replicate is built, or the Fisher combination writes its
// merged intervals there): a replicate whose pileup floats are still pending keeps what they are made of -- the exact
// pileups (looseV) and the tile descriptors -- instead of having the floats written now for nobody (k_piles_from_loose:
// 0.36 ms and 0.8 GB per replicate at hg38 / 50 M fragments; 0.4 GB of a 288 GB device kept instead).
int keep_loose_for_piles(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  if (pa.pairPending) return make_pair_piles(ctx, idx);   // (with a control: what the floats are made of does not outlive the sample)
  if (pa.loose)
    if (int rc = materialize_rep(ctx, idx)) return rc;
  if (!pa.pilesPending || pa.keptLoose) return GX_OK;
  pa.keptV = std::move(ctx->looseV);
  pa.keptMeta = std::move(ctx->tileMeta);
  pa.keptLoose = true;
  return GX_OK;
}

// k_merge2 on the two samples where they lie (their loose slots, or -- with -E regions -- their tight arrays).  pv: the p-value of
// every merged interval on the way (Merge2Out::pBits; the pileups only of the intervals the tables do not hold)
int launch_merge2(gx_ctx* ctx, const Merge2Out& mo, bool pv) {
  hipStream_t s = ctx->stream;
  const u32 nTiles = ctx->nTiles;
  const bool fromLoose = ctx->expt.inLoose && ctx->ctrl.inLoose;
  if (!fromLoose && (ctx->expt.inLoose || ctx->ctrl.inLoose || !ctx->expt.packed || !ctx->ctrl.packed)) {
    ctx->err = "control merge: the two samples are not in the same form";
    return GX_ERR_ORDER;
  }
  // (pos0 / len / flags of a tile do not depend on the sample: the control build's descriptors serve)
  const dim3 gridM(std::min(nTiles, (u32)(8 * ctx->numCU)));
  const float* p2d = ctx->pairP2d.as<float>();
  if (!ctx->knob.mergeWg) {
    // one wavefront per tile (round 6): as many workgroups as the CUs hold, each wavefront striding over the tiles
#define GX_MERGE2W(LOOSE_, PV_, A_, B_, META_)                                                                                        \
  do {                                                                                                                                \
    int nb = 0;                                                                                                                       \
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_merge2w<LOOSE_, PV_>, M2W_NW * 64, 0));                              \
    const u32 want = (nTiles + M2W_NW - 1) / M2W_NW;                                                                                 \
    hipLaunchKernelGGL((k_merge2w<LOOSE_, PV_>), dim3(std::max(1u, std::min(want, (u32)(std::max(1, std::min(nb, M2W_WGS)) * ctx->numCU)))), \
                       dim3(M2W_NW * 64), 0, s, A_, B_, ctx->dScal.as<Scalars>(), META_, nTiles, mo, ctx->dStatus.as<u32>(), p2d);   \
  } while (0)
    if (fromLoose) {
      RleIn A{ctx->expt.looseEnd.as<u32>(), ctx->expt.looseV.as<int>(), ctx->expt.tileIvOff.as<u32>(), ctx->expt.meta.as<TileMeta>()};
      RleIn Bc{ctx->ctrl.looseEnd.as<u32>(), ctx->ctrl.looseV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), ctx->ctrl.meta.as<TileMeta>()};
      if (pv) GX_MERGE2W(true, true, A, Bc, ctx->ctrl.meta.as<TileMeta>()); else GX_MERGE2W(true, false, A, Bc, ctx->ctrl.meta.as<TileMeta>());
    } else {
      RleIn A{ctx->expt.ivEnd.as<u32>(), ctx->expt.ivV.as<int>(), ctx->expt.tileIvOff.as<u32>(), nullptr};
      RleIn Bc{ctx->ctrl.ivEnd.as<u32>(), ctx->ctrl.ivV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), nullptr};
      if (pv) GX_MERGE2W(false, true, A, Bc, ctx->tileMeta.as<TileMeta>()); else GX_MERGE2W(false, false, A, Bc, ctx->tileMeta.as<TileMeta>());
    }
#undef GX_MERGE2W
    return dbg_sync(ctx, "k_merge2w");
  }
  if (fromLoose) {
    RleIn A{ctx->expt.looseEnd.as<u32>(), ctx->expt.looseV.as<int>(), ctx->expt.tileIvOff.as<u32>(), ctx->expt.meta.as<TileMeta>()};
    RleIn Bc{ctx->ctrl.looseEnd.as<u32>(), ctx->ctrl.looseV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), ctx->ctrl.meta.as<TileMeta>()};
    if (pv)
      hipLaunchKernelGGL((k_merge2<true, true>), gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->ctrl.meta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>(), p2d);
    else
      hipLaunchKernelGGL((k_merge2<true, false>), gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->ctrl.meta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>(), p2d);
  } else {
    RleIn A{ctx->expt.ivEnd.as<u32>(), ctx->expt.ivV.as<int>(), ctx->expt.tileIvOff.as<u32>(), nullptr};
    RleIn Bc{ctx->ctrl.ivEnd.as<u32>(), ctx->ctrl.ivV.as<int>(), ctx->ctrl.tileIvOff.as<u32>(), nullptr};
    if (pv)
      hipLaunchKernelGGL((k_merge2<false, true>), gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->tileMeta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>(), p2d);
    else
      hipLaunchKernelGGL((k_merge2<false, false>), gridM, dim3(MG_NT), 0, s, A, Bc, ctx->dScal.as<Scalars>(), ctx->tileMeta.as<TileMeta>(),
                         nTiles, mo, ctx->dStatus.as<u32>(), p2d);
  }
  return dbg_sync(ctx, "k_merge2");
}

// treatment + control -> the p-value intervals of the replicate (savePval Genrich.c:1720-1794): tile-local union of the
// two samples' breakpoints, p per interval from the control's tables, the sweep's masks on the way (p mode)
int merge_with_control(gx_ctx* ctx, PArray& pa) {
  hipStream_t s = ctx->stream;
  // treatment + control: tile-local union of breakpoints (savePval 1768-1791)
  const u32 nTiles = ctx->nTiles, nChrom = ctx->nChrom;
  const size_t cap = (size_t)ctx->expt.nIv + ctx->ctrl.nIv + 16;
  HIPCHECK(pooled(ctx, pa.end, cap * 4));
  // (the pileup floats -- Pileup.cov of the reference, printed by -f / -k only -- are made when somebody asks: make_pair_piles;
  // round 5 wrote them in every step, 0.9 GB at hg38 / 50 M + 50 M fragments)
  HIPCHECK(pooled(ctx, pa.p, cap * 4));
  HIPCHECK(pooled(ctx, pa.tileOff, (size_t)(nTiles + 2) * 4));
  HIPCHECK(pooled(ctx, pa.chromOff, (size_t)(nChrom + 2) * 4));
  // loose slots: the tile kernel's loose buffers, or others like them (+ one more int array)
  HIPCHECK(pooled(ctx, ctx->looseEnd, cap * 4));
  HIPCHECK(pooled(ctx, ctx->looseV, cap * 4));
  HIPCHECK(ctx->looseC.ensure(cap * 4));
  HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
  u32* misc = ctx->misc.as<u32>();
  phase_begin(ctx, "merge");
  // Round 6: the merge looks every interval's p-value up while both pileups are in registers and leaves (end, p) in its loose
  // slots -- 8 bytes instead of 12 --, the pileups only for the intervals the tables do not hold (their tiles on a list:
  // k_pairs_missed), and what makes the result tight is a copy (k_pack_ep2).  Loose p = the bits in looseV; the listed
  // intervals' pileups in looseC / looseC2.
  const bool mergeP = !ctx->knob.noMergeP;
  ctx->mergePUsed = mergeP;
  HIPCHECK(ctx->fragList.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(hipMemsetAsync(misc + M_TICKET, 0, 4, s));
  if (mergeP) HIPCHECK(ctx->looseC2.ensure(cap * 4));
  Merge2Out mo{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->tileIvCount.as<u32>(), nullptr, nullptr, nullptr};
  if (mergeP)
    mo = Merge2Out{ctx->looseEnd.as<u32>(), ctx->looseC.as<int>(), ctx->looseC2.as<int>(), ctx->tileIvCount.as<u32>(),
                   ctx->looseV.as<u32>(), ctx->fragList.as<u32>(), misc + M_TICKET};
  if (int rc__ = launch_merge2(ctx, mo, mergeP)) return rc__;
  const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  HIPCHECK(hipMemsetAsync(ctx->lb.p, 0, (size_t)(tChunks + 2) * 8, s));
  hipLaunchKernelGGL(k_scan_counts, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                     ctx->tileIvCount.as<u32>(), ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles,
                     ctx->lb.as<u64>(), pa.tileOff.as<u32>(), pa.chromOff.as<u32>(), misc + M_NMERGED,
                     ctx->dStatus.as<u32>());
  hipLaunchKernelGGL(k_fix_chrom_off, dim3(1), dim3(1), 0, s, ctx->dChrom.as<DChrom>(), nChrom, pa.chromOff.as<u32>(),
                     misc + M_NMERGED);
  if (int rc__ = dbg_sync(ctx, "k_scan_counts")) return rc__;
  phase_end(ctx);
  phase_begin(ctx, "pval");
  // (the control's tables -- log(treatment), control parameters, p of whole pileup pairs -- were built when
  // its sample was closed: finish_scalars)
  PackPairsIn ppi{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->expt.tileIvOff.as<u32>(),
                  ctx->ctrl.tileIvOff.as<u32>(), pa.tileOff.as<u32>()};
  if (mergeP) { ppi.looseE = ctx->looseC.as<int>(); ppi.looseC = ctx->looseC2.as<int>(); }   // (the listed intervals' pileups)
  // p-mode: the sweep's masks are filled on the way (the interval count is only bounded here, so the
  // masks are laid out for the bound and gx_find_peaks is told the stride)
  u64 *sigM = nullptr, *skipM = nullptr;
  ctx->maskIdx = -1;
  if (!ctx->par.qval_opt) {
    const size_t stride = (cap + 63) / 64 + 2;
    HIPCHECK(ctx->swMask.ensure(stride * 8 * 3));
    HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, stride * 8 * 3, s));
    sigM = ctx->swMask.as<u64>();
    skipM = sigM + stride;
    ctx->maskIdx = (int)ctx->reps.size();
    ctx->maskStride = stride;
  }
  if (mergeP) {
    hipLaunchKernelGGL(k_pairs_missed, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(4 * ctx->numCU)))), dim3(256), 0, s, ppi,
                       ctx->tileIvCount.as<u32>(), ctx->fragList.as<u32>(), misc + M_TICKET, ctx->dScal.as<Scalars>(),
                       ctx->pairLogE.as<double>(), ctx->pairCtab.as<CtrlEntry>(), ctx->looseV.as<u32>(), ctx->dStatus.as<u32>(),
                       ctx->dRisk.as<RiskBuf>());
    const dim3 grid(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
    if (sigM)
      hipLaunchKernelGGL((k_pack_ep2<true>), grid, dim3(256), 0, s, ppi, ctx->looseV.as<float>(), nTiles, pa.end.as<u32>(), pa.p.as<float>(),
                         ctx->par.thr, sigM, skipM);
    else
      hipLaunchKernelGGL((k_pack_ep2<false>), grid, dim3(256), 0, s, ppi, ctx->looseV.as<float>(), nTiles, pa.end.as<u32>(), pa.p.as<float>(),
                         ctx->par.thr, sigM, skipM);
  } else {
  {
    const dim3 grid(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
    const bool msk = sigM != nullptr;
#define GX_LAUNCH_PACK_PAIRS(K, M)                                                                                     \
hipLaunchKernelGGL((k_pack_pairs<K, M>), grid, dim3(256), 0, s, ppi, nTiles, ctx->pairCtab.as<CtrlEntry>(),           \
                   ctx->pairP2d.as<float>(), pa.end.as<u32>(), pa.expt.as<float>(), pa.ctrl.as<float>(),              \
                   pa.p.as<float>(), ctx->par.thr, sigM, skipM, ctx->fragList.as<u32>(), misc + M_TICKET)
    if (msk) GX_LAUNCH_PACK_PAIRS(false, true); else GX_LAUNCH_PACK_PAIRS(false, false);
#undef GX_LAUNCH_PACK_PAIRS
  }
  hipLaunchKernelGGL(k_pack_pairs_full, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(4 * ctx->numCU)))), dim3(256), 0, s,
                     ppi, ctx->fragList.as<u32>(), misc + M_TICKET, ctx->dScal.as<Scalars>(), ctx->pairLogE.as<double>(),
                     ctx->pairCtab.as<CtrlEntry>(), (float*)nullptr, (float*)nullptr, pa.p.as<float>(), ctx->par.thr, sigM, skipM,
                     ctx->dStatus.as<u32>(), ctx->dRisk.as<RiskBuf>());
  }
  if (int rc__ = dbg_sync(ctx, "k_pack_pairs")) return rc__;
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_NMERGED)) return rc__;
  int rc = status_to_rc(ctx, ctx->mail->status);
  {
    RiskTargets T{};
    T.pairP = pa.p.as<float>();
    T.sigMask = sigM;
    T.thr = ctx->par.thr;
    const int rcRisk = risk_apply(ctx, T);
    if (!rc) rc = rcRisk;
  }
  if (rc) return rc;
  pa.n = ctx->mail->nMerged;
  ctx->maskN = pa.n;
  pa.hasPiles = false;
  pa.pilesPending = pa.pairPending = ctx->keepPiles;
  pa.pilesDropped = !ctx->keepPiles;
  pa.ctrlIsConst = false;
  pa.mergedP = mergeP;
  return GX_OK;
}

// The pileup floats of a replicate WITH a control, on request: k_pack_pairs' look-ups once more, writing the two float arrays only
// (and the listed tiles -- fractional or very deep pileups -- in full).  Everything it reads is the merge's: valid until the next
// sample begins or the Fisher combination takes the loose slots (keep_loose_for_piles sees to it that this has run by then).
int make_pair_piles(gx_ctx* ctx, int idx) {
  PArray& pa = ctx->reps[idx];
  hipStream_t s = ctx->stream;
  const u32 nTiles = ctx->nTiles;
  HIPCHECK(pooled(ctx, pa.expt, (size_t)pa.n * 4 + 64));
  HIPCHECK(pooled(ctx, pa.ctrl, (size_t)pa.n * 4 + 64));
  u32* misc = ctx->misc.as<u32>();
  if (pa.mergedP) {
    // the step's merge left p-values in its loose slots: the pileups' merge now, for the floats (the two samples, the tile tables
    // and the scalars are where the step left them until the next sample begins)
    Merge2Out mo{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->tileIvCount.as<u32>(), nullptr, nullptr, nullptr};
    if (int rc__ = launch_merge2(ctx, mo, false)) return rc__;
    pa.mergedP = false;
  }
  PackPairsIn ppi{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseC.as<int>(), ctx->expt.tileIvOff.as<u32>(),
                  ctx->ctrl.tileIvOff.as<u32>(), pa.tileOff.as<u32>()};
  HIPCHECK(ctx->fragList.ensure((size_t)(nTiles + 1) * 4));
  HIPCHECK(hipMemsetAsync(misc + M_TICKET, 0, 4, s));
  const dim3 grid(std::max(1u, std::min((nTiles + 3) / 4, (u32)(8 * ctx->numCU))));
  hipLaunchKernelGGL((k_pack_pairs<true, false, true>), grid, dim3(256), 0, s, ppi, nTiles, ctx->pairCtab.as<CtrlEntry>(),
                     ctx->pairP2d.as<float>(), (u32*)nullptr, pa.expt.as<float>(), pa.ctrl.as<float>(), (float*)nullptr, ctx->par.thr,
                     (u64*)nullptr, (u64*)nullptr, ctx->fragList.as<u32>(), misc + M_TICKET);
  hipLaunchKernelGGL(k_pack_pairs_full, dim3(std::max(1u, std::min((nTiles + 3) / 4, (u32)(4 * ctx->numCU)))), dim3(256), 0, s,
                     ppi, ctx->fragList.as<u32>(), misc + M_TICKET, ctx->dScal.as<Scalars>(), ctx->pairLogE.as<double>(),
                     ctx->pairCtab.as<CtrlEntry>(), pa.expt.as<float>(), pa.ctrl.as<float>(), (float*)nullptr, ctx->par.thr,
                     (u64*)nullptr, (u64*)nullptr, ctx->dStatus.as<u32>(), ctx->dRisk.as<RiskBuf>());
  if (int rc__ = dbg_sync(ctx, "k_pack_pairs (pileup floats)")) return rc__;
  HIPCHECK(hipGetLastError());
  pa.hasPiles = true;
  pa.pilesPending = pa.pairPending = false;
  ctx->pilesMade = true;
  return GX_OK;
}

// combinePval (Genrich.c:612-667): union of all replicates' breakpoints, Fisher's method per interval; appends the
// combined array to ctx->reps
int combine_replicates(gx_ctx* ctx) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  // combinePval (612-667): union of all replicates' breakpoints, Fisher's method per interval
  const int nr = ctx->sample;
  if (nr > MAX_REPS) {
    ctx->err = "more than 32 replicates are not supported";
    return GX_ERR_DF;
  }
  const u32 nTiles = ctx->nTiles, nChrom = ctx->nChrom;
  PArray comb;
  comb.present.assign(nChrom, 0);
  size_t cap = nChrom + 16;
  RepSet S{};
  S.n = nr;
  for (int r = 0; r < nr; r++) {
    PArray& pa = ctx->reps[r];
    HIPCHECK(pooled(ctx, pa.dPresent, nChrom + 16));
    HIPCHECK(hipMemcpyAsync(pa.dPresent.p, pa.present.data(), nChrom, hipMemcpyHostToDevice, s));
    for (u32 i = 0; i < nChrom; i++) comb.present[i] |= pa.present[i];
    cap += pa.n;
    S.r[r] = RepIn{pa.end.as<u32>(), pa.p.as<float>(), pa.tileOff.as<u32>(), pa.dPresent.as<uint8_t>()};
  }
  HIPCHECK(pooled(ctx, comb.end, cap * 4));
  HIPCHECK(pooled(ctx, comb.p, cap * 4));
  HIPCHECK(pooled(ctx, comb.tileOff, (size_t)(nTiles + 2) * 4));
  HIPCHECK(pooled(ctx, comb.chromOff, (size_t)(nChrom + 2) * 4));
  phase_begin(ctx, "fisher");
  HIPCHECK(pooled(ctx, ctx->looseEnd, cap * 4));
  HIPCHECK(pooled(ctx, ctx->looseV, cap * 4));   // (the last replicate may have kept the previous one: keep_loose_for_piles)
  HIPCHECK(ctx->tileIvCount.ensure((size_t)(nTiles + 1) * 4));
  MergeNOut mo{ctx->looseEnd.as<u32>(), ctx->looseV.as<float>(), ctx->tileIvCount.as<u32>()};
  const size_t lds = mergeN_lds_bytes((int)nr);
  HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mergeN), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  int mnBlocks = 0;
  if (nr <= MNW_MAXREP) {
    // one wavefront per tile (no workgroup barrier in the tile loop, twenty tiles in flight per CU)
    const size_t ldsw = mergeNw_lds_bytes((int)nr);
    const u32 want = (nTiles + MNW_NW - 1) / MNW_NW;
    // (two instances: the per-replicate registers of a tile are unrolled to the instance's bound)
#define GX_MERGEN_W(MAXR)                                                                                                             \
  do {                                                                                                                                \
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mergeN_w<MAXR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw)); \
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&mnBlocks, k_mergeN_w<MAXR>, MNW_NW * 64, ldsw));                           \
    hipLaunchKernelGGL(k_mergeN_w<MAXR>, dim3(std::max(1u, std::min(want, (u32)(std::max(1, mnBlocks) * ctx->numCU)))),               \
                       dim3(MNW_NW * 64), ldsw, s, S, ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles, mo,                \
                       ctx->dStatus.as<u32>(), ctx->dRisk.as<RiskBuf>());                                                             \
  } while (0)
    if (nr <= 4) GX_MERGEN_W(4); else GX_MERGEN_W(MNW_MAXREP);
#undef GX_MERGEN_W
  } else {
  HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&mnBlocks, k_mergeN, MG_NT, lds));
  hipLaunchKernelGGL(k_mergeN, dim3(std::min(nTiles, (u32)(std::max(1, mnBlocks) * ctx->numCU))), dim3(MG_NT), lds, s, S,
                     ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles, mo, ctx->dStatus.as<u32>(),
                     ctx->dRisk.as<RiskBuf>());
  }
  if (int rc__ = dbg_sync(ctx, "k_mergeN")) return rc__;
  const u32 tChunks = (nTiles + STL_CHUNK - 1) / STL_CHUNK;
  HIPCHECK(hipMemsetAsync(ctx->lb.p, 0, (size_t)(tChunks + 2) * 8, s));
  hipLaunchKernelGGL(k_scan_counts, dim3(std::min<u32>(tChunks, (u32)ctx->resSweep)), dim3(STL_NT), 0, s,
                     ctx->tileIvCount.as<u32>(), ctx->dTileChrom.as<u32>(), ctx->dChrom.as<DChrom>(), nTiles,
                     ctx->lb.as<u64>(), comb.tileOff.as<u32>(), comb.chromOff.as<u32>(), misc + M_NMERGED,
                     ctx->dStatus.as<u32>());
  hipLaunchKernelGGL(k_fix_chrom_off, dim3(1), dim3(1), 0, s, ctx->dChrom.as<DChrom>(), nChrom, comb.chromOff.as<u32>(),
                     misc + M_NMERGED);
  hipLaunchKernelGGL(k_pack_ep, dim3(std::max(1u, std::min((nTiles + 3) / 4, 8192u))), dim3(256), 0, s, S,
                     ctx->looseEnd.as<u32>(), ctx->looseV.as<float>(), comb.tileOff.as<u32>(), nTiles, comb.end.as<u32>(),
                     comb.p.as<float>());
  if (int rc__ = dbg_sync(ctx, "k_pack_ep")) return rc__;
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_NMERGED)) return rc__;
  int rc = status_to_rc(ctx, ctx->mail->status);
  {
    RiskTargets T{};
    T.fisherP = comb.p.as<float>();
    T.fisherTileOff = comb.tileOff.as<u32>();
    const int rcRisk = risk_apply(ctx, T);
    if (!rc) rc = rcRisk;
  }
  if (rc) return rc;
  comb.n = ctx->mail->nMerged;
  ctx->reps.push_back(std::move(comb));
  return GX_OK;
}

// q of every interval of `fa` from the {key, q} table of the run (lookup 196-206), with the sweep's masks on the way (or without: nullptr)
int qlookup_all(gx_ctx* ctx, PArray& fa, u32 cap, u64* sigMask, u64* skipMask) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  const int bhv = ctx->knob.bhVariant >= 0 ? ctx->knob.bhVariant : GX_BH_DEFAULT_VARIANT;
  const u32 want = std::max(1u, (fa.n + 4095) / 4096);
#define GX_QLOOKUP(NT, CL, GRID)                                                                                                      \
  do {                                                                                                                                \
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qlookup<NT, CL>), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                 (int)(8u << CL)));                                                                                   \
    hipLaunchKernelGGL((k_qlookup<NT, CL>), dim3(std::min(want, (u32)(GRID))), dim3(NT), (size_t)(8u << CL), s, fa.p.as<float>(),     \
                       misc + M_NIV, ctx->bhKQ.as<u64>(), cap - 1, fa.q.as<float>(), ctx->par.thr,                                  \
                       sigMask, skipMask, ctx->dStatus.as<u32>());                              \
  } while (0)
  if (bhv == 1) GX_QLOOKUP(1024, 14, ctx->numCU);
  else if (bhv == 2) GX_QLOOKUP(512, 13, 2 * ctx->numCU);
  else GX_QLOOKUP(256, 11, 4096);
#undef GX_QLOOKUP
  return GX_OK;
}

// gx_get_intervals' q-values of a -q run whose sweep looked up the candidates' intervals only: the whole array, once
int ensure_q(gx_ctx* ctx, PArray& pa, int idx) {
  if (!pa.qLazy) return GX_OK;
  if (!pa.q.p && ctx->bhLive && ctx->bhLiveIdx == idx && pa.p.p) HIPCHECK(pooled(ctx, pa.q, (size_t)pa.n * 4 + 16));   // (-q on the loose slots)
  if (!ctx->bhLive || ctx->bhLiveIdx != idx || !pa.q.p) {
    ctx->err = "the q-values of this run are gone (another run has taken the table)";
    return GX_ERR_ORDER;
  }
  HIPCHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->misc.as<u32>() + M_NIV), (int)pa.n, 1, ctx->stream));
  if (int rc = qlookup_all(ctx, pa, ctx->bhLiveCap, nullptr, nullptr)) return rc;
  pa.qLazy = false;
  return GX_OK;
}

// -q on the loose slots: the "bp at pileup V" histogram that materialize_rep(.., hist) collects while it writes the tight table, without
// the table (k_pack_pval<.., HIST, false>: reads the loose slots, writes nothing but the sums), then Benjamini-Hochberg on it
int loose_hist(gx_ctx* ctx, PArray& pa, bool genomeOpt) {
  hipStream_t s = ctx->stream;
  phase_begin(ctx, "pval");
  const u32 cap = 1u << ctx->bhCapLog;
  if (int rc = bh_table_prepare(ctx, cap)) return rc;
  HIPCHECK(hipMemsetAsync(ctx->misc.as<u32>() + M_BHCOUNT, 0, 8, s));
  HIPCHECK(ctx->bhDense.ensure(bhd_words(1) * 8));
  HIPCHECK(hipMemsetAsync(ctx->bhDense.p, 0, bhd_words(1) * 8, s));
  PackIn pin{ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->tileMeta.as<TileMeta>(), pa.tileOff.as<u32>()};
  int nbH = 0;
  HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbH, k_pack_pval<false, true, false>, 256, 0));
  const dim3 gridH(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(std::max(1, std::min(nbH, 8)) * ctx->numCU))));
  hipLaunchKernelGGL((k_pack_pval<false, true, false>), gridH, dim3(256), 0, s, pin, ctx->nTiles, ctx->dScal.as<Scalars>(),
                     ctx->pvLut.as<float>(), (u32*)nullptr, (float*)nullptr, ctx->par.thr, (u64*)nullptr, (u64*)nullptr,
                     ctx->dStatus.as<u32>(), (const u32*)ctx->tilePrevEnd.as<u32>(), ctx->bhDense.as<u64>());
  phase_end(ctx);
  // computeQval on ~100 distinct values, whole: one workgroup (k_bh_small) -- q by pileup, from which pileup on it passes, the values'
  // {key, q} in the run's table for whoever asks for the q array later (ensure_q)
  phase_begin(ctx, "bh");
  u32* misc = ctx->misc.as<u32>();
  HIPCHECK(ctx->qLut.ensure((size_t)PV_WHOLE * 4));
  HIPCHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(misc + M_VQ), (int)0xFFFFFFFFu, 1, s));
  // (GX_FAULT=2: "q is no threshold on the pileup")
  HIPCHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(misc + M_VQ + 1), ctx->knob.fault == 2 ? (int)0xFFFFFFFFu : 0, 1, s));
  hipLaunchKernelGGL(k_bh_small, dim3(1), dim3(1024), 0, s, (const u64*)ctx->bhDense.as<u64>(), ctx->pvLut.as<float>(), bh_table_of(ctx, cap),
                     ctx->bhKQ.as<u64>(), reinterpret_cast<const u64*>(misc + M_GENOME), ctx->par.thr, ctx->qLut.as<float>(), misc + M_VQ,
                     misc + M_ALLONE, genomeOpt ? ctx->dStatus.as<u32>() : (u32*)nullptr, ctx->dStatus.as<u32>());
  ctx->denseHistUsed = true;
  pa.qLazy = true;
  ctx->lazyQUsed = true;
  ctx->bhDirty = false;
  ctx->bhLive = true;   // (freed by the next run that wants the table: bh_release_live)
  ctx->bhLiveCap = cap;
  ctx->bhLiveIdx = ctx->finalIdx;
  phase_end(ctx);
  return GX_OK;
}

// computeQval / saveQval (Genrich.c:352-401, 212-250) for the final p-array `fa` of n intervals: the genome-wide table
// {p -> bp} (with several ranks: after the exchange, gx_host_coll.h), its sort and suffix scan, q per interval and the
// sweep's masks on the way.  genomeOpt: the genome length was computed (not -L): the lengths must add up to it
int bh_qvalues(gx_ctx* ctx, PArray& fa, u32 n, bool genomeOpt) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  const u32 nChrom = ctx->nChrom;
  phase_begin(ctx, "bh");
  u32 cap = 1u << ctx->bhCapLog;
  auto bh_table = [&](u32 c) -> int { return bh_table_prepare(ctx, c); };
  auto bh_grow = [&]() -> int {
    if (ctx->bhCapLog >= 28) {
      ctx->err = "p-value table full";
      return GX_ERR_MEM;
    }
    ctx->bhCapLog += 3;
    cap = 1u << ctx->bhCapLog;
    ctx->bhDirty = true;  // (whatever the failed attempt left behind is wiped)
    return GX_OK;
  };
  BhTable T{};
  u32 Dlocal = 0;
  // the instance of k_bh_hist / k_qlookup (gx_stats.h): LDS tables that hold a stretch of the genome's distinct values
  const int bhv = ctx->knob.bhVariant >= 0 ? ctx->knob.bhVariant : GX_BH_DEFAULT_VARIANT;
  auto launch_hist = [&]() -> int {
    const u32 want = std::max(1u, (n + 4095) / 4096);
#define GX_BH_HIST(NT, LT, PR, GRID)                                                                                                  \
  do {                                                                                                                                \
    HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bh_hist<NT, LT, PR>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                 (int)bh_hist_lds(LT)));                                                                              \
    hipLaunchKernelGGL((k_bh_hist<NT, LT, PR>), dim3(std::min(want, (u32)(GRID))), dim3(NT), bh_hist_lds(LT), s, fa.end.as<u32>(),     \
                       fa.p.as<float>(), fa.chromOff.as<u32>(), nChrom, misc + M_NIV, T, ctx->dStatus.as<u32>());                   \
  } while (0)
    if (bhv == 1) GX_BH_HIST(1024, 8192, 16, ctx->numCU);
    else if (bhv == 2) GX_BH_HIST(512, 4096, 16, 3 * ctx->numCU);
    else GX_BH_HIST(256, 2048, 8, 2048);
#undef GX_BH_HIST
    return GX_OK;
  };
  // (one replicate without a control: k_pack_pval<.., HIST> left "bp at V" in bhDense and the deep values in the table)
  bool fromDense = ctx->denseHistIdx >= 0 && ctx->denseHistIdx == ctx->finalIdx && ctx->world <= 1 && !ctx->forceColl;
  ctx->denseHistIdx = -1;
  ctx->denseHistUsed = fromDense;
  for (;;) {
    if (fromDense) {
      T = bh_table_of(ctx, cap);
      HIPCHECK(hipMemsetAsync(misc + M_BHOVF, 0, 4, s));
      hipLaunchKernelGGL(k_bh_from_dense, dim3(256), dim3(256), 0, s, (const u64*)ctx->bhDense.as<u64>(), ctx->pvLut.as<float>(), 1u, T,
                         misc + M_BHOVF, ctx->dStatus.as<u32>());
      if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_BHCOUNT)) return rc__;
      Dlocal = ctx->mail->nMerged;
      if (!(ctx->mail->status & ST_HASH_FULL)) break;
      if (ctx->mail->status != ST_HASH_FULL) return status_to_rc(ctx, ctx->mail->status & ~ST_HASH_FULL);
      HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 4, s));   // (a table too small: once more from the tight intervals, into a larger one)
      if (int rc = bh_grow()) return rc;
      fromDense = false;
      ctx->denseHistUsed = false;
      continue;
    }
    if (int rc = bh_table(cap)) return rc;
    HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 8, s));
    T = bh_table_of(ctx, cap);
    if (int rc__ = launch_hist()) return rc__;
    if (int rc__ = dbg_sync(ctx, "k_bh_hist")) return rc__;
    if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_BHCOUNT)) return rc__;
    Dlocal = ctx->mail->nMerged;
    if (!(ctx->mail->status & ST_HASH_FULL)) break;
    if (ctx->mail->status != ST_HASH_FULL) return status_to_rc(ctx, ctx->mail->status & ~ST_HASH_FULL);
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 4, s));
    if (int rc = bh_grow()) return rc;
  }
  const bool multi = ctx->world > 1 || ctx->forceColl;
  // (computeQval 377-382: checked when the genome length was computed -- not with -L)
  u32* lenCheck = genomeOpt ? ctx->dStatus.as<u32>() : (u32*)nullptr;
  u32 D = 0;
  bool denseDone = false;
  ctx->denseBhUsed = false;
  ctx->rangeBhUsed = false;
  // One sample without a control: p is a function of the pileup, every rank holds the same table p(V), and the
  // genome-wide histogram is ONE all-reduce of a dense "bp at V" array (gx_stats.h: k_bh_dense_fill) -- decided by what
  // every rank knows alike
  if (multi && ctx->sample == 1 && ctx->reps.size() == 1 && fa.ctrlIsConst && !ctx->bedGiven && (u32)std::max(1, ctx->world) <= 64 &&
      !ctx->knob.noDenseBh) {
    const u32 W = (u32)std::max(1, ctx->world);
    const size_t words = bhd_words(W);
    HIPCHECK(ctx->bhDense.ensure(words * 8));
    HIPCHECK(hipMemsetAsync(ctx->bhDense.p, 0, words * 8, s));
    HIPCHECK(hipMemsetAsync(misc + M_BHOVF, 0, 4, s));  // (the "a rank's region overflowed" word)
    hipLaunchKernelGGL(k_bh_dense_fill, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s,
                       ctx->bhOutKeys.as<u32>(), ctx->bhOutSlot.as<u32>(), ctx->bhLens.as<u64>(), misc + M_BHCOUNT,
                       ctx->pvLut.as<float>(), ctx->bhDense.as<u64>(), (u32)ctx->rank);
    if (int rc__ = allreduce_words(ctx, ctx->bhDense.as<long long>(), words)) return rc__;
    hipLaunchKernelGGL(k_bh_clear, dim3(256), dim3(256), 0, s, T);
    HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 4, s));
    hipLaunchKernelGGL(k_bh_from_dense, dim3(256), dim3(256), 0, s, (const u64*)ctx->bhDense.as<u64>(), ctx->pvLut.as<float>(), W, T,
                       misc + M_BHOVF, ctx->dStatus.as<u32>());
    HIPCHECK(hipMemcpyAsync(&ctx->mail->D, misc + M_BHCOUNT, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(&ctx->mail->bhOvf, misc + M_BHOVF, 4, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (ctx->mail->bhOvf == 0) {
      D = ctx->mail->D;
      denseDone = true;
      ctx->denseBhUsed = true;
    } else {
      // some rank holds more values outside the table than its region takes: every rank saw it, all go back to their own
      // tables and take the general exchange
      hipLaunchKernelGGL(k_bh_clear, dim3(256), dim3(256), 0, s, T);
      HIPCHECK(hipMemsetAsync(misc + M_BHCOUNT, 0, 8, s));
      if (int rc__ = launch_hist()) return rc__;
      if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, misc + M_BHCOUNT)) return rc__;
      Dlocal = ctx->mail->nMerged;
    }
  }
  bool rangeDone = false;
  if (denseDone) {
  } else if (multi) {
    // a control / replicates: the range-partitioned exchange (gx_bhx.h) leaves every value's q in this rank's table
    if (int rc = bh_range_exchange(ctx, T, Dlocal, cap)) return rc;
    rangeDone = true;
    D = 0;
  } else
    D = Dlocal;
  if (D && !rangeDone) {
    HIPCHECK(ctx->bhSortKeys.ensure((size_t)D * 4));
    HIPCHECK(ctx->bhSortSlot.ensure((size_t)D * 4));
    HIPCHECK(ctx->bhRaw.ensure((size_t)D * 4));
    size_t tmpBytes = 0;
    HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(),
                                       ctx->bhOutSlot.as<u32>(), ctx->bhSortSlot.as<u32>(), D, 0, 32, s));
    HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
    HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(),
                                       ctx->bhOutSlot.as<u32>(), ctx->bhSortSlot.as<u32>(), D, 0, 32, s));
    if (D <= 16384 && !ctx->knob.qtMulti) {
      hipLaunchKernelGGL(k_qtable, dim3(1), dim3(1024), 0, s, ctx->bhSortKeys.as<u32>(), ctx->bhSortSlot.as<u32>(),
                         ctx->bhLens.as<u64>(), D, reinterpret_cast<const u64*>(misc + M_GENOME), ctx->bhQ.as<float>(),
                         ctx->bhRaw.as<float>(), misc + M_ALLONE, lenCheck);
    } else {  // many distinct values (Fisher-combined replicates): the chunked kernels
      const u32 nCh = (D + QT_CHUNK - 1) / QT_CHUNK;
      HIPCHECK(ctx->bhDl.ensure((size_t)D * 8 + (size_t)nCh * 12 + 64));
      u64* dl = ctx->bhDl.as<u64>();
      u64* chunkSum = dl + D;
      float* chunkMin = reinterpret_cast<float*>(chunkSum + nCh);
      hipLaunchKernelGGL(k_qt_sums, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortSlot.as<u32>(), ctx->bhLens.as<u64>(), D, dl,
                         chunkSum, (const u32*)nullptr);
      hipLaunchKernelGGL(k_qt_raw, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortKeys.as<u32>(), dl, D,
                         reinterpret_cast<const u64*>(misc + M_GENOME), chunkSum, ctx->bhRaw.as<float>(), chunkMin,
                         (const u32*)nullptr, (const u64*)nullptr, 1u, 0u, lenCheck);
      hipLaunchKernelGGL(k_qt_apply, dim3(nCh), dim3(QT_NT), 0, s, ctx->bhSortSlot.as<u32>(), ctx->bhRaw.as<float>(), D,
                         chunkMin, ctx->bhQ.as<float>(), misc + M_ALLONE, (const u32*)nullptr, (const u64*)nullptr, 1u, 0u);
    }
if (int rc__ = dbg_sync(ctx, "k_qtable")) return rc__;
  }
  const bool lazyQ = !ctx->knob.noLazyQ;
  HIPCHECK(pooled(ctx, fa.q, (size_t)n * 4 + 16));
  // q-values and, on the way, the sweep's significance / SKIP masks
  {
    const size_t stride = (size_t)((n + 63) / 64) + 2;
    HIPCHECK(ctx->swMask.ensure(stride * 8 * 3));
    HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, stride * 8 * 3, s));
    ctx->maskIdx = ctx->finalIdx;
    ctx->maskN = n;
    ctx->maskStride = stride;
    HIPCHECK(hipMemsetAsync(misc + M_PSTAR, 0xFF, 4, s));
    hipLaunchKernelGGL(k_kq_build, dim3(256), dim3(256), 0, s, T, ctx->bhQ.as<float>(), ctx->bhKQ.as<u64>(), ctx->par.thr, misc + M_PSTAR);
    if (lazyQ) {
      // q never falls as p grows: the masks from one compare per interval; q itself inside the candidates (run_sweep: k_q_fill_cands)
      // and for whoever asks for the array (ensure_q)
      hipLaunchKernelGGL(k_sig_from_p, dim3(std::max(1u, std::min((n + 1023) / 1024, (u32)(8 * ctx->numCU)))), dim3(256), 0, s, fa.p.as<float>(),
                         misc + M_NIV, misc + M_PSTAR, ctx->swMask.as<u64>(), ctx->swMask.as<u64>() + stride);
    } else {
      if (int rc__ = qlookup_all(ctx, fa, cap, ctx->swMask.as<u64>(), ctx->swMask.as<u64>() + stride)) return rc__;
    }
  }
  fa.qLazy = lazyQ;
  ctx->lazyQUsed = lazyQ;
  ctx->bhDirty = false;
  ctx->bhLive = true;   // (freed by the next run that wants the table: bh_release_live)
  ctx->bhLiveCap = cap;
  ctx->bhLiveIdx = ctx->finalIdx;
  if (!lazyQ) bh_release_live(ctx);
if (int rc__ = dbg_sync(ctx, "k_qlookup")) return rc__;
  phase_end(ctx);
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

}  // namespace
