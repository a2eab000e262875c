// gx_host_sweep.h -- callPeaks (Genrich.c:977-1069) on bit masks: the launches of the peak sweep and its one synchronisation.
// (a part of gx_api.hip's translation unit: the kernels are templates and inline functions of the headers it includes;
// split by phase -- context / build / stats / sweep / collectives -- in round 5)
#pragma once
namespace {

// What the sweep walks: the interval arrays (end, p[, q]) of the final p-array.
struct SweepSrc {
  const u32* end = nullptr;
  const float* p = nullptr;
  const float* q = nullptr;
  // -q with lazy q-values: `q` is written for the candidates' intervals only, by k_q_fill_cands from the run's {key, q} table
  const u64* kq = nullptr;
  u32 kqMask = 0;
  // the sweep on the loose slots (LooseCtl): `end` = the loose ends, p = the table p(V) looked up with the slots' exact
  // pileups `V`; the masks are [significant | first of its chromosome], `mStride` apart, and there are no SKIP intervals
  const int* V = nullptr;
  const float* qLut = nullptr;   // ... with -q: q by whole pileup (k_bh_small) -- the AUC and the summits take q from it
  bool haveMasks = false, hasSkip = true;
  const u32* chromOff = nullptr;
  u32 nChrom = 0, nWords = 0;
  size_t mStride = 0;   // words between the sig / skip / brk masks in swMask
};

// callPeaks (Genrich.c:977-1069) on bit masks: runs of adjacent significant intervals -> candidates -> in-order AUC.
// ONE synchronisation, at the end: the run / candidate arrays are sized by a guess (the largest run count seen so
// far, with headroom), every kernel reads the counts on the device, the true run count comes back with the mail, and
// only when it exceeds the guess is the sweep repeated with arrays that fit.  Counts travel through pinned memory
// written by the kernels themselves, and the peak list is written straight into pinned host memory.
int run_sweep(gx_ctx* ctx, const SweepSrc& S, u32* nPeaksOut) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  HostMail* dm = static_cast<HostMail*>(ctx->mailBuf.dp);
  const u32 nWords = S.nWords, nChrom = S.nChrom;
  const u32 wChunks = (nWords + SW_CHUNK - 1) / SW_CHUNK;
  SweepMasks SM{ctx->swMask.as<u64>(), ctx->swMask.as<u64>() + S.mStride, ctx->swMask.as<u64>() + 2 * S.mStride, nWords};
  if (S.V) SM = SweepMasks{ctx->swMask.as<u64>(), nullptr, ctx->swMask.as<u64>() + S.mStride, nWords};
  u32 R = 0, nPeaks = 0;
  ctx->peakBP = 0;
  ctx->nHostPeaks = 0;
  if (nWords) {
    // (in loose-slot index space the chromosome starts were marked when the sample was closed: k_close / k_frag_select)
    if (!S.V) hipLaunchKernelGGL(k_brk_mask, dim3((nChrom + 255) / 256), dim3(256), 0, s, S.chromOff, nChrom, SM.brk);
    if (!S.haveMasks)
      hipLaunchKernelGGL(k_sig_mask, dim3(std::max(1u, std::min((nWords + 15) / 16, 4096u))), dim3(256), 0, s, S.p, S.q,
                         misc + M_NIV, ctx->par.thr, SM);
    // look-back granules of the three one-pass compactions (generation-tagged: never cleared between calls)
    {
      const size_t need = (size_t)2 * (wChunks + 8) * 8;
      if (ctx->lbSweep.cap < need) {
        HIPCHECK(ctx->lbSweep.ensure(need));
        HIPCHECK(hipMemsetAsync(ctx->lbSweep.p, 0, ctx->lbSweep.cap, s));
      }
    }
    for (int attempt = 0;; attempt++) {
      // arrays for `cap` runs (never more runs than intervals)
      const u64 capMin = ctx->knob.runCapMin > 0 ? (u64)ctx->knob.runCapMin : (u64)1 << 16;  // (tests: a tiny first guess)
      // (first guess: one run per 256 intervals -- several times what a default threshold leaves on a genome --
      // so that a single call does not pay for a second pass)
      const u64 guess = ctx->knob.runCapMin > 0 ? capMin : std::max<u64>(capMin, (u64)nWords / 4);
      const u32 cap = std::min<u64>(std::max<u64>(ctx->runCap, std::max<u64>(guess, 1)), (u64)nWords * 64);
      HIPCHECK(ctx->swStart.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->swEnd.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->headPos.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->cand.ensure((size_t)cap * sizeof(gx_peak)));
      HIPCHECK(ctx->valid.ensure((size_t)cap * 4 + 16));
      HIPCHECK(ctx->hPeaks.ensure((size_t)cap * sizeof(gx_peak) + 16));  // (at most one peak per run)
      HIPCHECK(ctx->candHdr.ensure((size_t)cap * sizeof(uint4)));
      HIPCHECK(ctx->longList.ensure((size_t)cap * 4 + 16));
      const u32 rChunks = (cap + RC_CHUNK - 1) / RC_CHUNK;
      {
        const size_t need = (size_t)2 * (rChunks + 8) * 8;
        if (ctx->lbSweep2.cap < need) {
          HIPCHECK(ctx->lbSweep2.ensure(need));
          HIPCHECK(hipMemsetAsync(ctx->lbSweep2.p, 0, ctx->lbSweep2.cap, s));
        }
      }
      if (++ctx->sweepGen >= (1u << 24)) {  // (the generation field wraps: start over with clean arrays)
        ctx->sweepGen = 1;
        HIPCHECK(hipMemsetAsync(ctx->lbSweep.p, 0, ctx->lbSweep.cap, s));
        HIPCHECK(hipMemsetAsync(ctx->lbSweep2.p, 0, ctx->lbSweep2.cap, s));
      }
      const u32 gen = ctx->sweepGen;
      u64* lbS = ctx->lbSweep.as<u64>();
      u64* lbE = lbS + wChunks + 8;
      u64* lbC = ctx->lbSweep2.as<u64>();
      u64* lbP = lbC + rChunks + 8;
      u32* runStart = ctx->swStart.as<u32>();
      u32* runEnd = ctx->swEnd.as<u32>();
      const u64* skipM = S.hasSkip ? SM.skip : (const u64*)nullptr;
      const u32 gridP = (u32)std::max(1, ctx->resSweep);
      // runs: count, place and write in one pass; the true count goes to the host, at most `cap` to the kernels
      hipLaunchKernelGGL(k_runs, dim3(std::min<u32>(wChunks, gridP)), dim3(SW_NT), 0, s, SM, lbS, lbE, gen, runStart, runEnd, cap,
                         misc + M_SWCOUNT, &dm->R, misc + M_TICKET3, reinterpret_cast<u64*>(misc + M_PEAKBP), ctx->dStatus.as<u32>());
      // candidates (chunks beyond the device-side run count leave at once)
      hipLaunchKernelGGL(k_cands, dim3(std::min<u32>(rChunks, gridP)), dim3(SW_NT), 0, s, SM, skipM, S.end, runStart, runEnd,
                         misc + M_SWCOUNT, ctx->par.max_gap, S.chromOff, nChrom, lbC, gen, ctx->headPos.as<u32>(), misc + M_NHEADS,
                         ctx->dStatus.as<u32>());
      hipLaunchKernelGGL(k_cand_hdr, dim3(std::max(1u, std::min((cap + 255) / 256, 4096u))), dim3(256), 0, s, SM, S.end, runStart,
                         runEnd, misc + M_SWCOUNT, ctx->headPos.as<u32>(), misc + M_NHEADS, ctx->candHdr.as<uint4>(),
                         ctx->longList.as<u32>(), misc + M_TICKET3);
      if (S.kq)
        hipLaunchKernelGGL(k_q_fill_cands, dim3(std::max(1u, std::min((cap + 3) / 4, 32768u))), dim3(256), 0, s,
                           ctx->candHdr.as<uint4>(), misc + M_NHEADS, S.p, S.kq, S.kqMask, const_cast<float*>(S.q), ctx->dStatus.as<u32>());
      {
        const dim3 grid(std::max(1u, std::min((cap + 15) / 16, 16384u)));  // 16 candidates per workgroup and round
        const dim3 gridW(std::max(1u, std::min((cap + 3) / 4, (u32)(8 * ctx->numCU))));
// (k_peak_both: the short candidates' workgroups first, the long candidates' behind them, one launch)
#define GX_LAUNCH_PEAKS(Q, V, NSHORT, QPTR)                                                                               \
  hipLaunchKernelGGL((k_peak_both<Q, V>), dim3((NSHORT) + gridW.x), dim3(256), 0, s, (u32)(NSHORT), ctx->candHdr.as<uint4>(), \
                     S.end, S.p, QPTR, S.chromOff, nChrom, misc + M_NHEADS, ctx->longList.as<u32>(), misc + M_TICKET3,       \
                     ctx->par.thr, ctx->par.min_auc, ctx->par.min_len, ctx->cand.as<gx_peak>(), ctx->valid.as<u32>())
        if (S.V && S.qLut) {  // ... and q from its own table
          const u32 nShortV = std::min<u32>(grid.x, (u32)(8 * ctx->numCU));
          hipLaunchKernelGGL((k_peak_both<false, true, true>), dim3(nShortV + gridW.x), dim3(256), 0, s, nShortV, ctx->candHdr.as<uint4>(),
                             S.end, S.p, reinterpret_cast<const float*>(S.V), S.chromOff, nChrom, misc + M_NHEADS, ctx->longList.as<u32>(),
                             misc + M_TICKET3, ctx->par.thr, ctx->par.min_auc, ctx->par.min_len, ctx->cand.as<gx_peak>(),
                             ctx->valid.as<u32>(), S.qLut);
        } else if (S.V) {  // p from the table p(V) (`q` carries the exact pileups); every workgroup copies the table's compact form to LDS
          const u32 nShortV = std::min<u32>(grid.x, (u32)(8 * ctx->numCU));
          GX_LAUNCH_PEAKS(false, true, nShortV, reinterpret_cast<const float*>(S.V));
        } else if (S.q)
          GX_LAUNCH_PEAKS(true, false, grid.x, S.q);
        else
          GX_LAUNCH_PEAKS(false, false, grid.x, S.q);
#undef GX_LAUNCH_PEAKS
      }
      // the peaks, in order, into pinned host memory; their number with them
      hipLaunchKernelGGL(k_peaks, dim3(std::min<u32>(rChunks, gridP)), dim3(SW_NT), 0, s, ctx->cand.as<gx_peak>(), ctx->valid.as<u32>(),
                         misc + M_NHEADS, lbP, gen, static_cast<gx_peak*>(ctx->hPeaks.dp), misc + M_NPEAKS, &dm->nPeaks,
                         ctx->dStatus.as<u32>(), reinterpret_cast<u64*>(misc + M_PEAKBP));
      if (int rc__ = dbg_sync(ctx, "sweep kernels")) return rc__;
      // the end: status, counts, the peaks' total length (and whatever else is pending) through the mail kernel, one
      // synchronisation
      if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const u64*>(misc + M_PEAKBP)))
        return rc__;
      R = ctx->mail->R;
      ctx->runSeen = R;
      if (R <= cap) break;
      if (attempt >= 2) {
        ctx->err = "peak sweep: run count changed between attempts";
        return GX_ERR_DEVICE;
      }
      ctx->runCap = (u64)R + R / 4 + 1024;  // the guess was too small: once more, with arrays that fit
    }
    ctx->runCap = std::max<u64>(ctx->runCap, (u64)R + R / 4 + 1024);
  } else {
    if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc__;
  }
  if (R) nPeaks = ctx->mail->nPeaks;
  ctx->nHostPeaks = nPeaks;
  ctx->peakBP = R ? ctx->mail->peakBP : 0;  // (callPeaks 925: summed by k_peaks)
  *nPeaksOut = nPeaks;
  return status_to_rc(ctx, ctx->mail->status);
}

}  // namespace
