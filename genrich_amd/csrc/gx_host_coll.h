// gx_host_coll.h -- the exchanges of an N-rank run: the all-reduce of a few words (RCCL on the library's stream, or
// the host program's callback), the failure word, the fixed-size concatenations, the all-to-all and the range-partitioned BH
// exchange (gx_bhx.h).
// (a part of gx_api.hip's translation unit: the kernels are templates and inline functions of the headers it includes;
// split by phase -- context / build / stats / sweep / collectives -- in round 5)
#pragma once
namespace {

// n (<= 4) 64-bit words on the device, summed over all ranks in place: RCCL in stream order (no host hop), or the host
// program's callback (a copy down, a synchronisation, a copy up)
int allreduce_words(gx_ctx* ctx, long long* d, size_t n) {
  hipStream_t s = ctx->stream;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(&ctx->err);
    if (!api) return GX_ERR_DEVICE;
    ncclResult_t r = api->allReduce(d, d, n, ncclInt64, ncclSum, ctx->comm, s);
    if (r != ncclSuccess) {
      ctx->err = std::string("ncclAllReduce: ") + api->getErrorString(r);
      return GX_ERR_DEVICE;
    }
  } else if (ctx->allreduce) {
    long long* acc = ctx->mail->coll;
    // (a big payload -- the dense p-value histogram, the all-to-all buffer of the range exchange -- is timed as a phase of
    // its own, "xfer": the trips to the host are this mode's stand-in for RCCL, not part of the phase they interrupt)
    const bool big = n > 4, wasOpen = ctx->phaseOpen;
    const std::string resume = wasOpen && ctx->nPhases ? ctx->phases[ctx->nPhases - 1].name : std::string();
    if (big) {
      HIPCHECK(ctx->hostRecs.ensure(n * 8));
      acc = static_cast<long long*>(ctx->hostRecs.p);
      phase_end(ctx);
      phase_begin(ctx, "xfer");
    }
    struct Resume {
      gx_ctx* c; std::string nm; bool on;
      ~Resume() { if (on) { phase_end(c); if (!nm.empty()) phase_begin(c, nm.c_str()); } }
    } resumeGuard{ctx, resume, big};
    HIPCHECK(hipMemcpyAsync(acc, d, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (ctx->allreduce(reinterpret_cast<int64_t*>(acc), n, ctx->user)) {
      ctx->err = "allreduce callback failed";
      return GX_ERR_DEVICE;
    }
    HIPCHECK(hipMemcpyAsync(d, acc, n * 8, hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));  // (the pinned words are reused by the next exchange)
  } else {
    ctx->err = "several ranks but no collectives (gx_set_rccl / gx_set_collectives)";
    return GX_ERR_ORDER;
  }
  return GX_OK;
}

constexpr long long COLL_FAILED = 1ll << 48;  // third all-reduce word: some rank could not build its sample

void poison_allreduce(gx_ctx* ctx) {
  if (!(ctx->world > 1 || ctx->forceColl)) return;
  hipStream_t s = ctx->stream;
  long long w[3] = {0, 0, COLL_FAILED};
  // (a build that fails ahead of its early all-reduce -- the closed form of fragLen, build_pileup -- owes the other
  // ranks that one too: they are in it, or about to be)
  const int rounds = ctx->earlyOwed ? 2 : 1;
  ctx->earlyOwed = false;
  for (int r = 0; r < rounds; r++) {
    if (ctx->comm) {
      const gxrccl::Api* api = gxrccl::load(nullptr);
      if (!api || !ctx->dColl.p) return;
      if (hipMemcpyAsync(ctx->dColl.p, w, sizeof w, hipMemcpyHostToDevice, s) != hipSuccess) return;
      (void)api->allReduce(ctx->dColl.p, ctx->dColl.p, 3, ncclInt64, ncclSum, ctx->comm, s);
      (void)hipStreamSynchronize(s);
    } else if (ctx->allreduce) {
      int64_t buf[3] = {w[0], w[1], w[2]};
      (void)ctx->allreduce(buf, 3, ctx->user);
    }
  }
}

// every rank's `per` 64-bit words -- written at [rank * per, ...) of a buffer that is zero elsewhere -- to every rank
// (a sum of disjoint regions is their concatenation: the fixed-size exchanges need no counts and no host)
int coll_concat(gx_ctx* ctx, long long* d, size_t per) { return allreduce_words(ctx, d, per * (size_t)std::max(1, ctx->world)); }

// all-to-all with the counts of M (M[src * W + dst] elements of elemBytes from src to dst; sOff / rOff: this rank's
// send / receive offsets in elements).  RCCL: grouped send / recv on the library's stream.  Host callbacks (the
// validation mode of the tests): one all-reduce of a buffer in which every rank fills its outgoing segments.
int coll_alltoallv(gx_ctx* ctx, const void* dSend, const std::vector<size_t>& sOff, void* dRecv, const std::vector<size_t>& rOff,
                          const std::vector<u32>& M, size_t elemBytes) {
  hipStream_t s = ctx->stream;
  const u32 W = (u32)std::max(1, ctx->world), me = (u32)ctx->rank;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(&ctx->err);
    if (!api) return GX_ERR_DEVICE;
    ncclResult_t r = api->groupStart();
    for (u32 p = 0; p < W && r == ncclSuccess; p++) {
      const size_t ns = (sOff[p + 1] - sOff[p]) * elemBytes, nr = (rOff[p + 1] - rOff[p]) * elemBytes;
      if (ns) r = api->send(static_cast<const char*>(dSend) + sOff[p] * elemBytes, ns, ncclChar, (int)p, ctx->comm, s);
      if (nr && r == ncclSuccess) r = api->recv(static_cast<char*>(dRecv) + rOff[p] * elemBytes, nr, ncclChar, (int)p, ctx->comm, s);
    }
    const ncclResult_t r2 = api->groupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) {
      ctx->err = std::string("ncclSend / ncclRecv: ") + api->getErrorString(r);
      return GX_ERR_DEVICE;
    }
    return GX_OK;
  }
  // (src-major layout of all segments; this rank's outgoing ones are contiguous in it, as in its send buffer)
  std::vector<size_t> segOff((size_t)W * W + 1, 0);
  for (size_t i = 0; i < (size_t)W * W; i++) segOff[i + 1] = segOff[i] + M[i];
  const size_t words = (segOff[(size_t)W * W] * elemBytes + 7) / 8 + 1;
  HIPCHECK(ctx->dGather.ensure(words * 8));
  HIPCHECK(hipMemsetAsync(ctx->dGather.p, 0, words * 8, s));
  const size_t mine = segOff[(size_t)(me + 1) * W] - segOff[(size_t)me * W];
  if (mine)
    HIPCHECK(hipMemcpyAsync(ctx->dGather.as<char>() + segOff[(size_t)me * W] * elemBytes, dSend, mine * elemBytes, hipMemcpyDeviceToDevice, s));
  if (int rc = allreduce_words(ctx, ctx->dGather.as<long long>(), words)) return rc;
  for (u32 src = 0; src < W; src++) {
    const size_t n = M[(size_t)src * W + me];
    if (n)
      HIPCHECK(hipMemcpyAsync(static_cast<char*>(dRecv) + rOff[src] * elemBytes, ctx->dGather.as<char>() + segOff[(size_t)src * W + me] * elemBytes,
                              n * elemBytes, hipMemcpyDeviceToDevice, s));
  }
  return GX_OK;
}

// gx_bhx.h: this rank's table T (Dlocal distinct values, their slots in bhOutKeys / bhOutSlot) -> the q of every one of
// them in ctx->bhQ, by slot
int bh_range_exchange(gx_ctx* ctx, const BhTable& T, u32 Dlocal, u32 capLocal) {
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  const u32 W = (u32)std::max(1, ctx->world), me = (u32)ctx->rank;
  if (W > 64) { ctx->err = "more than 64 ranks"; return GX_ERR_ORDER; }
  (void)capLocal;
  // 1: this rank's distinct values in order
  HIPCHECK(ctx->bhSortKeys.ensure((size_t)std::max(Dlocal, 1u) * 4));
  HIPCHECK(ctx->bhSortSlot.ensure((size_t)std::max(Dlocal, 1u) * 4));
  if (Dlocal) {
    size_t tmpBytes = 0;
    HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                                       ctx->bhSortSlot.as<u32>(), Dlocal, 0, 32, s));
    HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
    HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, ctx->bhOutKeys.as<u32>(), ctx->bhSortKeys.as<u32>(), ctx->bhOutSlot.as<u32>(),
                                       ctx->bhSortSlot.as<u32>(), Dlocal, 0, 32, s));
  }
  // the small fixed-size exchanges share one buffer: samples | counts matrix | range totals | range minima
  const size_t oSamp = 0, oCnt = oSamp + (size_t)W * BHX_SAMPLES, oTot = oCnt + (size_t)W * W, oMin = oTot + W, nSmall = oMin + W;
  HIPCHECK(ctx->bhxSmall.ensure(nSmall * 8 + (size_t)(2 * W + 4) * 4));
  u64* small = ctx->bhxSmall.as<u64>();
  u32* dSpl = reinterpret_cast<u32*>(small + nSmall);
  u32* dSendOff = dSpl + W + 1;
  HIPCHECK(hipMemsetAsync(small, 0, nSmall * 8, s));
  hipLaunchKernelGGL(k_bhx_samples, dim3(1), dim3(64), 0, s, (const u32*)ctx->bhSortKeys.as<u32>(), Dlocal, small + oSamp + (size_t)me * BHX_SAMPLES);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oSamp), BHX_SAMPLES)) return rc;
  hipLaunchKernelGGL(k_bhx_splitters, dim3(1), dim3(1024), 0, s, (const u64*)(small + oSamp), W * BHX_SAMPLES, W, dSpl);
  // 2: the counts, and the one synchronisation
  hipLaunchKernelGGL(k_bhx_offsets, dim3(1), dim3(128), 0, s, (const u32*)ctx->bhSortKeys.as<u32>(), Dlocal, (const u32*)dSpl, W, dSendOff,
                     small + oCnt + (size_t)me * W);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oCnt), W)) return rc;
  std::vector<u64> M64((size_t)W * W);
  HIPCHECK(hipMemcpyAsync(M64.data(), small + oCnt, M64.size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  std::vector<u32> M((size_t)W * W);
  for (size_t i = 0; i < M.size(); i++) M[i] = (u32)M64[i];
  std::vector<size_t> sOff(W + 1, 0), rOff(W + 1, 0);
  for (u32 p = 0; p < W; p++) {
    sOff[p + 1] = sOff[p] + M[(size_t)me * W + p];
    rOff[p + 1] = rOff[p] + M[(size_t)p * W + me];
  }
  if (sOff[W] != Dlocal) { ctx->err = "BH exchange: the counts do not add up"; return GX_ERR_DEVICE; }
  const size_t R = rOff[W];
  if (R >= ((size_t)1 << 31)) { ctx->err = "p-value table full"; return GX_ERR_MEM; }
  // 3: records out, records in
  HIPCHECK(ctx->bhRecs.ensure((size_t)std::max(Dlocal, 1u) * sizeof(BhRec)));
  HIPCHECK(ctx->bhxRecv.ensure(std::max<size_t>(R, 1) * sizeof(BhRec)));
  if (Dlocal)
    hipLaunchKernelGGL(k_bh_pack, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s, ctx->bhSortKeys.as<u32>(),
                       ctx->bhSortSlot.as<u32>(), ctx->bhLens.as<u64>(), Dlocal, ctx->bhRecs.as<BhRec>());
  if (int rc = coll_alltoallv(ctx, ctx->bhRecs.p, sOff, ctx->bhxRecv.p, rOff, M, sizeof(BhRec))) return rc;
  // 4: the owner's table of its range: merged, sorted, scored
  u32 cap2 = 1024;
  while ((size_t)cap2 < 4 * R) cap2 <<= 1;
  const u32 Rb = (u32)std::max<size_t>(R, 1);
  HIPCHECK(ctx->bhxKeys.ensure((size_t)cap2 * 4));
  HIPCHECK(ctx->bhxLens.ensure((size_t)cap2 * 8));
  HIPCHECK(ctx->bhxQ.ensure((size_t)cap2 * 4));
  HIPCHECK(ctx->bhxOut.ensure((size_t)Rb * 16));   // claimed keys | slots | sorted keys | sorted slots
  HIPCHECK(hipMemsetAsync(ctx->bhxKeys.p, 0xFF, (size_t)cap2 * 4, s));
  HIPCHECK(hipMemsetAsync(ctx->bhxLens.p, 0, (size_t)cap2 * 8, s));
  HIPCHECK(hipMemsetAsync(ctx->bhxOut.p, 0xFF, (size_t)Rb * 8, s));  // (unclaimed entries sort behind every value)
  u32* oKeys = ctx->bhxOut.as<u32>();
  u32 *oSlot = oKeys + Rb, *sKeys = oSlot + Rb, *sSlot = sKeys + Rb;
  u32* cnt2 = misc + M_BHOVF;  // (free again: the dense exchange is not this run's)
  HIPCHECK(hipMemsetAsync(cnt2, 0, 4, s));
  BhTable T2{ctx->bhxKeys.as<u32>(), ctx->bhxLens.as<u64>(), cap2 - 1, oKeys, oSlot, cnt2};
  if (R)
    hipLaunchKernelGGL(k_bh_insert, dim3((u32)std::max<size_t>(1, std::min<size_t>((R + 255) / 256, 1024))), dim3(256), 0, s,
                       (const BhRec*)ctx->bhxRecv.as<BhRec>(), (u32)R, T2, ctx->dStatus.as<u32>());
  {
    size_t tmpBytes = 0;
    HIPCHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, oKeys, sKeys, oSlot, sSlot, Rb, 0, 32, s));
    HIPCHECK(ctx->bhTmp.ensure(tmpBytes + 16));
    HIPCHECK(rocprim::radix_sort_pairs(ctx->bhTmp.p, tmpBytes, oKeys, sKeys, oSlot, sSlot, Rb, 0, 32, s));
  }
  const u32 nCh = (Rb + QT_CHUNK - 1) / QT_CHUNK;
  HIPCHECK(ctx->bhDl.ensure((size_t)Rb * 8 + (size_t)nCh * 12 + 64));
  HIPCHECK(ctx->bhRaw.ensure((size_t)Rb * 4));
  u64* dl = ctx->bhDl.as<u64>();
  u64* chunkSum = dl + Rb;
  float* chunkMin = reinterpret_cast<float*>(chunkSum + nCh);
  hipLaunchKernelGGL(k_qt_sums, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sSlot, (const u64*)ctx->bhxLens.as<u64>(), 0u, dl, chunkSum, (const u32*)cnt2);
  hipLaunchKernelGGL(k_bhx_reduce, dim3(1), dim3(256), 0, s, (const u64*)chunkSum, (const float*)nullptr, nCh, small + oTot + me, (u64*)nullptr);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oTot), 1)) return rc;
  hipLaunchKernelGGL(k_qt_raw, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sKeys, (const u64*)dl, 0u, reinterpret_cast<const u64*>(misc + M_GENOME),
                     (const u64*)chunkSum, ctx->bhRaw.as<float>(), chunkMin, (const u32*)cnt2, (const u64*)(small + oTot), W, me,
                     ctx->par.genome_len == 0 ? ctx->dStatus.as<u32>() : (u32*)nullptr);
  hipLaunchKernelGGL(k_bhx_reduce, dim3(1), dim3(256), 0, s, (const u64*)nullptr, (const float*)chunkMin, nCh, (u64*)nullptr, small + oMin + me);
  if (int rc = coll_concat(ctx, reinterpret_cast<long long*>(small + oMin), 1)) return rc;
  hipLaunchKernelGGL(k_qt_apply, dim3(nCh), dim3(QT_NT), 0, s, (const u32*)sSlot, (const float*)ctx->bhRaw.as<float>(), 0u, (const float*)chunkMin,
                     ctx->bhxQ.as<float>(), (u32*)nullptr, (const u32*)cnt2, (const u64*)(small + oMin), W, me);
  // 5: the answers, back along the same counts
  HIPCHECK(ctx->bhxAns.ensure(std::max<size_t>(R, 1) * 4 + (size_t)std::max(Dlocal, 1u) * 4));
  float* ansOut = ctx->bhxAns.as<float>();
  float* ansIn = ansOut + std::max<size_t>(R, 1);
  if (R)
    hipLaunchKernelGGL(k_bhx_answer, dim3((u32)std::max<size_t>(1, std::min<size_t>((R + 255) / 256, 1024))), dim3(256), 0, s,
                       (const BhRec*)ctx->bhxRecv.as<BhRec>(), (u32)R, (const u32*)ctx->bhxKeys.as<u32>(), cap2 - 1, (const float*)ctx->bhxQ.as<float>(), ansOut,
                       ctx->dStatus.as<u32>());
  std::vector<u32> Mt((size_t)W * W);
  for (u32 a = 0; a < W; a++)
    for (u32 b = 0; b < W; b++) Mt[(size_t)a * W + b] = M[(size_t)b * W + a];
  if (int rc = coll_alltoallv(ctx, ansOut, rOff, ansIn, sOff, Mt, 4)) return rc;
  if (Dlocal)
    hipLaunchKernelGGL(k_bhx_scatter, dim3(std::max(1u, std::min((Dlocal + 255) / 256, 1024u))), dim3(256), 0, s, (const u32*)ctx->bhSortSlot.as<u32>(),
                       (const float*)ansIn, Dlocal, ctx->bhQ.as<float>());
  (void)T;
  ctx->rangeBhUsed = true;
  return GX_OK;
}

}  // namespace
