// gx_api.hip -- C ABI (include/genrich_amd.h) over the HIP kernels.  Host code here only
// sizes buffers, launches kernels on one stream and moves scalars; every per-base /
// per-interval computation of the hot path runs on the device.  There is no CPU fallback:
// a missing device or a failed launch is reported as GX_ERR_DEVICE.
#include "gx_host_ctx.h"
#include "gx_host_coll.h"
#include "gx_host_build.h"
#include "gx_host_sweep.h"
#include "gx_host_stats.h"


// ================================ C ABI ==================================================

extern "C" {

const char* gx_strerror(int status) {
  switch (status) {
    case GX_OK: return "";
    case GX_ERR_MEM: return "Cannot allocate memory";
    case GX_ERR_GEN: return "No analyzable genome (length=0)";
    case GX_ERR_EXPT: return "Experimental sample has no analyzable fragments";
    case GX_ERR_PILE: return "Invalid pileup value (< 0)";
    case GX_ERR_POS: return ": read aligned beyond reference end";
    case GX_ERR_ALNS: return "Disallowed number of alignments";
    case GX_ERR_ARR: return "Failure creating experimental pileup";
    case GX_ERR_PVAL: return "Failure collecting p-values";
    case GX_ERR_DF: return "Invalid df in pchisq()";
    case GX_ERR_ORDER: return "API called out of order";
    case GX_ERR_DEVICE: return "HIP device failure";
    default: return "Unknown error";
  }
}

int gx_create(gx_ctx** out, const gx_params* par) {
  if (!out || !par) return GX_ERR_ORDER;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= par->device) {
    fprintf(stderr, "genrich_amd: no HIP device %d (found %d) -- there is no CPU fallback\n", par->device, nd);
    return GX_ERR_DEVICE;
  }
  gx_ctx* ctx = new gx_ctx();
  ctx->par = *par;
  ctx->device = par->device;
  for (const KnobDef& d : KNOBS)
    if (const char* e = getenv(d.name)) set_knob(ctx->knob, d.name, e);
  if (ctx->knob.bhCapLog) ctx->bhCapLog = (u32)std::max(4, std::min(28, ctx->knob.bhCapLog));  // (tests: a tiny first table)
  if (ctx->knob.ptJmax) ctx->ptJmax = (u32)std::max(1, std::min(1 << 16, ctx->knob.ptJmax));   // (tests: short page-table rows)
  *out = ctx;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  HIPCHECK(ctx->mailBuf.ensure(sizeof(HostMail)));
  ctx->mail = static_cast<HostMail*>(ctx->mailBuf.p);
  memset(ctx->mail, 0, sizeof(HostMail));
  HIPCHECK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  HIPCHECK(hipEventCreateWithFlags(&ctx->sideEv, hipEventDisableTiming));
  HIPCHECK(ctx->misc.ensure(M_WORDS * 4));
  HIPCHECK(ctx->dScal.ensure(sizeof(Scalars)));
  HIPCHECK(ctx->dStatus.ensure(64));
  // p-value tables and the list of risky values: fixed sizes, built while a sample is closed
  HIPCHECK(ctx->pvLut.ensure((size_t)(PV_LUT + PV_WHOLE) * 4));  // p(V), and p of the whole pileups once more (compact)
  HIPCHECK(ctx->pairLogE.ensure((size_t)PAIR_LUT * 8));
  HIPCHECK(ctx->pairCtab.ensure((size_t)PAIR_LUT * sizeof(CtrlEntry)));
  HIPCHECK(ctx->pairP2d.ensure((size_t)PT_N * PT_N * 4));
  HIPCHECK(ctx->dColl.ensure(64));
  HIPCHECK(ctx->dRisk.ensure(sizeof(RiskBuf)));
  HIPCHECK(ctx->dDeep.ensure(sizeof(DeepTab)));
  HIPCHECK(ctx->riskHost.ensure(sizeof(RiskBuf)));
  memset(ctx->riskHost.p, 0, 32);
  HIPCHECK(hipMemsetAsync(ctx->dRisk.p, 0, 32, ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dDeep.p, 0, sizeof(DeepTab), ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->misc.p, 0, M_WORDS * 4, ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dScal.p, 0, sizeof(Scalars), ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
  HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS * 4));
  {
    // persistent kernels: the grid must not exceed what is co-resident (look-back forward progress)
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, ctx->device));
    ctx->numCU = prop.multiProcessorCount;
    int nb = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile<true, false>, TL_NT, TL_LDS * 4));
    ctx->resTile = std::max(1, nb) * ctx->numCU;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile<true, true>, TL_NT, TL_LDS_HALF * 4));
    ctx->resTileHalf = std::max(1, nb) * ctx->numCU;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile_fast, 64, 0));
    ctx->resTileFast = std::max(1, std::min(nb, TF_WG_PER_CU)) * ctx->numCU;
    if (ctx->knob.debug)
      fprintf(stderr, "k_tile workgroups: half %d, wide %d, fast %d\n", ctx->resTileHalf, ctx->resTile, ctx->resTileFast);
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_scan_iv, STL_NT, 0));
    ctx->resSweep = std::max(1, std::min(nb, 4)) * ctx->numCU;
  }
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

void gx_destroy(gx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) {
    if (const gxrccl::Api* api = gxrccl::load(nullptr)) (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  for (auto& ph : ctx->phases) {
    (void)hipEventDestroy(ph.a);
    (void)hipEventDestroy(ph.b);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->sideEv) (void)hipEventDestroy(ctx->sideEv);
  for (hipEvent_t e : ctx->evPool) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->stageFree) if (e) (void)hipEventDestroy(e);
  delete ctx;
}

const char* gx_last_error(const gx_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int gx_set_chroms(gx_ctx* ctx, int n, const uint32_t* len, const uint8_t* skip, const uint32_t* const* bed,
                  const int32_t* bed_len) {
  if (!ctx || n <= 0 || !len) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  ctx->nChrom = (u32)n;
  ctx->len.assign(len, len + n);
  ctx->skip.assign(n, 0);
  ctx->save.assign(n, 1);
  ctx->owned.assign(n, 1);
  ctx->bed.assign(n, {});
  ctx->bedGiven = false;
  for (int i = 0; i < n; i++) {
    ctx->skip[i] = skip && skip[i];
    if (bed && bed_len && bed_len[i] > 0 && !ctx->skip[i]) {
      ctx->bed[i].assign(bed[i], bed[i] + bed_len[i]);
      ctx->bedGiven = true;
    }
  }
  int rc = layout_tiles(ctx);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

int gx_set_collectives(gx_ctx* ctx, int rank, int world, gx_allreduce_i64_fn allreduce, void* user) {
  if (!ctx || world < 1 || rank < 0 || rank >= world) return GX_ERR_ORDER;
  if (ctx->comm && allreduce) {  // callbacks replace a communicator of gx_set_rccl
    if (const gxrccl::Api* api = gxrccl::load(nullptr)) (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->rank = rank;
  ctx->world = world;
  ctx->allreduce = allreduce;
  ctx->user = user;
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_rccl_unique_id(void* out, size_t cap) {
  if (!out || cap < sizeof(ncclUniqueId)) return GX_ERR_ORDER;
  std::string err;
  const gxrccl::Api* api = gxrccl::load(&err);
  if (!api) {
    fprintf(stderr, "genrich_amd: %s\n", err.c_str());
    return GX_ERR_DEVICE;
  }
  ncclUniqueId id;
  if (api->getUniqueId(&id) != ncclSuccess) return GX_ERR_DEVICE;
  memcpy(out, &id, sizeof id);
  return GX_OK;
}

int gx_set_rccl(gx_ctx* ctx, int rank, int world, const void* unique_id) {
  if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world || world > 64) return GX_ERR_ORDER;
  const gxrccl::Api* api = gxrccl::load(&ctx->err);
  if (!api) return GX_ERR_DEVICE;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->comm) {
    (void)api->commDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  ncclResult_t r = api->commInitRank(&ctx->comm, world, id, rank);
  if (r != ncclSuccess) {
    ctx->err = std::string("ncclCommInitRank: ") + api->getErrorString(r);
    ctx->comm = nullptr;
    return GX_ERR_DEVICE;
  }
  ctx->rank = rank;
  ctx->world = world;
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_set_keep_pileups(gx_ctx* ctx, int keep) {
  if (!ctx) return GX_ERR_ORDER;
  ctx->keepPiles = keep != 0;
  return GX_OK;
}

int gx_set_owned(gx_ctx* ctx, const uint8_t* owned) {
  if (!ctx || !owned || ctx->nChrom == 0) return GX_ERR_ORDER;
  if (ctx->phase != 0 || ctx->sample != 0) return GX_ERR_ORDER;  // the tile space changes: between runs only
  ctx->owned.assign(owned, owned + ctx->nChrom);
  int rc = layout_tiles(ctx);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  return GX_OK;
}

int gx_reset(gx_ctx* ctx) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  for (Pileup* P : {&ctx->expt, &ctx->ctrl}) {  // (samples stashed for a control merge give their buffers back)
    recycle(ctx, P->looseEnd); recycle(ctx, P->looseV); recycle(ctx, P->meta);
    P->inLoose = false;
  }
  for (auto& pa : ctx->reps) {
    recycle(ctx, pa.end); recycle(ctx, pa.p); recycle(ctx, pa.expt); recycle(ctx, pa.ctrl);
    recycle(ctx, pa.chromOff); recycle(ctx, pa.q); recycle(ctx, pa.tileOff); recycle(ctx, pa.dPresent);
    recycle(ctx, pa.chromLooseOff); recycle(ctx, pa.keptV); recycle(ctx, pa.keptMeta);
  }
  ctx->reps.clear();
  ctx->denseHistIdx = -1;
  ctx->pilesMade = false;
  ctx->sample = 0;
  ctx->phase = 0;
  ctx->finalIdx = -1;
  ctx->segs.clear();
  ctx->unpackUsed = 0;
  ctx->evChunkIdx = ctx->evChunkFill = ctx->evPoolUsed = 0;
  ctx->nHostPeaks = 0;
  if (ctx->statusSeen) {  // (a clean run leaves the status words at zero: no fill launch)
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 64, ctx->stream));
    ctx->statusSeen = 0;
  }
  return GX_OK;
}

int gx_sample_begin(gx_ctx* ctx, int is_ctrl, const uint8_t* save) {
  if (!ctx || ctx->nChrom == 0) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  if (!is_ctrl) {
    if (ctx->phase != 0) return GX_ERR_ORDER;
    // (a further replicate: the previous one's loose slots, tile tables and p(V) table are about to be reused)
    for (size_t r = 0; r < ctx->reps.size(); r++)
      if (ctx->reps[r].loose || ctx->reps[r].pilesPending)
        if (int rc = keep_loose_for_piles(ctx, (int)r)) return rc;
    for (u32 i = 0; i < ctx->nChrom; i++) ctx->save[i] = save ? (save[i] != 0) : 1;
    int rc = upload_chroms(ctx, false);
    if (rc) return rc;
    uint64_t g = ctx->par.genome_len ? ctx->par.genome_len : genome_len_for(ctx, ctx->save);
    if (!g) {
      ctx->err = "No analyzable genome (length=0)";
      return GX_ERR_GEN;  // calcLambda 1828
    }
    Scalars z{};
    z.genomeLen = g;
    ctx->hScal = z;
    // (the device's copy is cleared by the first launch of the build: k_build_init)
    ctx->beginPending = true;
    ctx->beginGenome = (u64)g;
    ctx->nPhases = 0;
    ctx->phase = 1;
  } else {
    if (ctx->phase != 2) return GX_ERR_ORDER;
    int rc = stash_or_pack(ctx, ctx->expt);  // the control's tiles are about to be built: the treatment steps aside
    if (rc) return rc;
    ctx->phase = 3;
  }
  ctx->segs.clear();
  ctx->unpackUsed = 0;
  ctx->evChunkIdx = ctx->evChunkFill = ctx->evPoolUsed = 0;  // (the previous sample's uploads were consumed: gx_sample_end synchronised)
  ctx->satDone = false;
  ctx->satDropped = 0;
  return GX_OK;
}

namespace {

constexpr size_t EV_CHUNK = (size_t)1 << 22;   // events per device chunk (64 MiB)
constexpr size_t EV_STAGE = (size_t)1 << 20;   // events per pinned staging buffer (16 MiB)

hipEvent_t ready_event(gx_ctx* ctx) {
  if (ctx->evPoolUsed == ctx->evPool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    ctx->evPool.push_back(e);
  }
  return ctx->evPool[ctx->evPoolUsed++];
}

// Host events -> the library's device chunks, by asynchronous copies on the side stream.  `pinned`: the
// caller's memory is page-locked and stays untouched until gx_sample_end, so it is the copy's source itself;
// otherwise the events go through two pinned staging buffers (the caller's buffer is free on return, and the
// host keeps parsing while a buffer is in flight).  Nothing here waits for a copy to arrive: every piece
// carries an event that the main stream waits for before the kernel that reads it (build_pileup).
// (`esz`: 16, gx_event, or 8, gx_event8: the chunks are filled in 16-byte units either way, so every piece starts on a 16-byte
// boundary -- k_sort_a<.., PACKED> reads two packed events per load -- and a piece with an odd count leaves 8 bytes unused)
int push_host(gx_ctx* ctx, const void* events_, size_t n, bool pinned, size_t esz) {
  const char* events = static_cast<const char*>(events_);
  const size_t per16 = sizeof(gx_event) / esz;   // events per 16-byte unit
  while (n) {
    if (ctx->evChunkIdx == ctx->evChunks.size()) ctx->evChunks.emplace_back();
    DevBuf& chunk = ctx->evChunks[ctx->evChunkIdx];
    HIPCHECK(chunk.ensure(EV_CHUNK * sizeof(gx_event)));
    size_t take = std::min(n, (EV_CHUNK - ctx->evChunkFill) * per16);
    if (!pinned) take = std::min(take, EV_STAGE * per16);
    char* dst = chunk.as<char>() + ctx->evChunkFill * sizeof(gx_event);
    const char* src = events;
    if (!pinned) {
      const int k = ctx->stageNext;
      ctx->stageNext ^= 1;
      HIPCHECK(ctx->stage[k].ensure(EV_STAGE * sizeof(gx_event)));
      if (!ctx->stageFree[k]) HIPCHECK(hipEventCreateWithFlags(&ctx->stageFree[k], hipEventDisableTiming));
      else HIPCHECK(hipEventSynchronize(ctx->stageFree[k]));  // its previous upload has left the buffer
      memcpy(ctx->stage[k].p, events, take * esz);
      src = static_cast<const char*>(ctx->stage[k].p);
      HIPCHECK(hipMemcpyAsync(dst, src, take * esz, hipMemcpyHostToDevice, ctx->side));
      HIPCHECK(hipEventRecord(ctx->stageFree[k], ctx->side));
    } else
      HIPCHECK(hipMemcpyAsync(dst, src, take * esz, hipMemcpyHostToDevice, ctx->side));
    hipEvent_t ev = ready_event(ctx);
    if (!ev) { ctx->err = "hipEventCreate failed"; return GX_ERR_DEVICE; }
    HIPCHECK(hipEventRecord(ev, ctx->side));
    ctx->segs.push_back({reinterpret_cast<const gx_event*>(dst), take, ev, esz != sizeof(gx_event)});
    ctx->evChunkFill += (take + per16 - 1) / per16;
    if (ctx->evChunkFill == EV_CHUNK) { ctx->evChunkIdx++; ctx->evChunkFill = 0; }
    events += take * esz;
    n -= take;
  }
  return GX_OK;
}

}  // namespace

int gx_push_events(gx_ctx* ctx, const gx_event* events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  return push_host(ctx, events, n, false, sizeof(gx_event));
}

int gx_push_events_pinned(gx_ctx* ctx, const gx_event* events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  return push_host(ctx, events, n, true, sizeof(gx_event));
}

int gx_push_events_packed(gx_ctx* ctx, const gx_event8* events, size_t n, int where) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3) || where < 0 || where > 2) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  if (where == GX_EVENTS_DEVICE) {
    // (used in place; k_sort_a reads two records per 16-byte load: a buffer that does not start on a 16-byte boundary, or an odd
    // count at its end, goes through the 16-byte form instead -- unpack_segs)
    ctx->segs.push_back({reinterpret_cast<const gx_event*>(events), n, nullptr, true});
    return GX_OK;
  }
  return push_host(ctx, events, n, where == GX_EVENTS_PINNED, sizeof(gx_event8));
}

int gx_event8_pack(const gx_event* in, gx_event8* out) {
  if (!in || !out) return 0;
  const uint32_t len = in->end - in->start;
  uint32_t cls = 8;
  for (uint32_t k = 0; k < 8; k++)
    if ((uint32_t)((0xA8654321u >> (4u * k)) & 15u) == in->count) cls = k;
  if (in->end < in->start || len >= 0xFFFFu || cls == 8 || in->chrom >= (1u << 13)) return 0;
  out->start = in->start;
  out->lcc = len | (cls << 16) | (in->chrom << 19);
  return 1;
}

long long gx_filter_saturation(const gx_event* events, size_t n, int n_chrom, const uint32_t* len, uint8_t* keep) {
  if ((!events && n) || (!keep && n) || n_chrom < 0 || (!len && n_chrom)) return GX_ERR_ORDER;
  return gxsat::filter(events, n, n_chrom, len, keep);
}

int gx_push_events_device(gx_ctx* ctx, const gx_event* d_events, size_t n) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3)) return GX_ERR_ORDER;
  if (n) ctx->segs.push_back({d_events, n, nullptr});
  return GX_OK;
}

int gx_sample_end(gx_ctx* ctx, double* frag_len, float* lambda, float* factor) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->phase == 1) {
    int rc = close_sample(ctx, ctx->expt, 0);
    if (rc) return rc;
    ctx->phase = 2;
  } else if (ctx->phase == 3) {
    int rc = close_sample(ctx, ctx->ctrl, 1);
    if (rc) return rc;
    ctx->phase = 4;
  } else
    return GX_ERR_ORDER;
  if (frag_len) *frag_len = ctx->hScal.fragLen;
  if (lambda) *lambda = ctx->hScal.lambda;
  if (factor) *factor = ctx->hScal.factor;
  return GX_OK;
}

int gx_expect_fractional(gx_ctx* ctx, int on) {
  if (!ctx) return GX_ERR_ORDER;
  // (a hint selects kernels; what the library has learned from a sample by itself -- sawFrac -- is not the hint's to clear)
  ctx->fracHint = on != 0;
  return GX_OK;
}

int gx_set_knob(gx_ctx* ctx, const char* name, const char* value) {
  if (!ctx || !name) return GX_ERR_ORDER;
  // (GX_BH_CAPLOG, GX_PT_JMAX and GX_SBSHIFT size tables when the context and its tile layout are made: on a live context
  // they would change nothing, so they are refused instead of accepted without effect)
  for (const char* fixed : {"GX_BH_CAPLOG", "GX_PT_JMAX", "GX_SBSHIFT"})
    if (!strcmp(name, fixed)) {
      ctx->err = std::string(name) + " is read when the context is made (environment); it cannot change afterwards";
      return GX_ERR_ORDER;
    }
  if (!set_knob(ctx->knob, name, value)) {
    ctx->err = std::string("unknown switch ") + name;
    return GX_ERR_ORDER;
  }
  ctx->forceColl = ctx->knob.forceColl != 0;
  return GX_OK;
}

int gx_dups_first(gx_ctx* ctx, const gx_dup_key* keys, const uint8_t* multi, size_t n, uint32_t* owner) {
  // (the table has 2^k >= 2 n slots addressed by 32-bit indices: n <= 2^30)
  if (!ctx || (n && (!keys || !multi || !owner)) || n > ((size_t)1 << 30)) return GX_ERR_ORDER;
  if (!n) return GX_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  u32 cap = 1024;
  while ((size_t)cap < 2 * n) cap <<= 1;   // (<= 2^31: no wrap)
  DevBuf dKeys, dMulti, dOwner, dTab;
  HIPCHECK(dKeys.ensure(n * 16));
  HIPCHECK(dMulti.ensure(n + 16));
  HIPCHECK(dOwner.ensure(n * 4));
  HIPCHECK(dTab.ensure((size_t)cap * 12));
  HIPCHECK(hipMemcpyAsync(dKeys.p, keys, n * 16, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(dMulti.p, multi, n, hipMemcpyHostToDevice, s));
  DupTab T{dTab.as<u32>(), dTab.as<u32>() + cap, dTab.as<u32>() + 2 * (size_t)cap, cap - 1};
  HIPCHECK(hipMemsetAsync(T.rep, 0xFF, (size_t)cap * 8, s));   // rep = free, first = "no index yet"
  HIPCHECK(hipMemsetAsync(T.multi, 0, (size_t)cap * 4, s));
  const u32 blocks = (u32)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192));
  hipLaunchKernelGGL(k_dups_insert, dim3(blocks), dim3(256), 0, s, dKeys.as<uint4>(), (const uint8_t*)dMulti.as<uint8_t>(), (u32)n, T);
  hipLaunchKernelGGL(k_dups_lookup, dim3(blocks), dim3(256), 0, s, dKeys.as<uint4>(), (u32)n, T, dOwner.as<u32>());
  HIPCHECK(hipMemcpyAsync(owner, dOwner.p, n * 4, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

int gx_saturation_dropped(gx_ctx* ctx, long long* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = ctx->satDropped;
  return GX_OK;
}

int gx_window_net(gx_ctx* ctx, uint32_t chrom, uint32_t pos0, uint32_t n, long long* net) {
  if (!ctx || (ctx->phase != 1 && ctx->phase != 3) || !net || !n || n > (1u << 16) || chrom >= ctx->nChrom) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (int rc__ = unpack_segs(ctx, true)) return rc__;
  DevBuf dNet;
  HIPCHECK(dNet.ensure((size_t)n * 8));
  HIPCHECK(hipMemsetAsync(dNet.p, 0, (size_t)n * 8, s));
  for (auto& sg : ctx->segs) {
    if (!sg.n) continue;
    if (sg.ready) HIPCHECK(hipStreamWaitEvent(s, sg.ready, 0));  // (its upload, on the side stream)
    const u32 blocks = (u32)std::min<size_t>((sg.n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_window_net, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(sg.p), sg.n, chrom, ctx->len[chrom],
                       pos0, n, dNet.as<unsigned long long>());
  }
  HIPCHECK(hipMemcpyAsync(net, dNet.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return GX_OK;
}

int gx_sample_no_control(gx_ctx* ctx, float* lambda) {
  if (!ctx || ctx->phase != 2) return GX_ERR_ORDER;
  if (lambda) *lambda = ctx->hScal.lambda;  // computed with fragLen (calcLambda 1831)
  ctx->phase = 5;
  return GX_OK;
}

int gx_pvalues(gx_ctx* ctx) {
  if (!ctx || (ctx->phase != 4 && ctx->phase != 5)) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  PArray pa;
  pa.present.assign(ctx->nChrom, 0);
  for (u32 i = 0; i < ctx->nChrom; i++) pa.present[i] = !ctx->skip[i] && ctx->save[i];
  if (ctx->phase == 5) {
    // no control: the p-intervals are the treatment intervals, and they stay where the tile kernel left them
    // (loose slots) until somebody needs the tight table: gx_find_peaks on a single sample with -p does not
    pa.n = ctx->expt.nIv;
    pa.chromOff = std::move(ctx->expt.chromIvOff);
    pa.tileOff = std::move(ctx->expt.tileIvOff);
    pa.ctrlIsConst = !ctx->hasBed;
    pa.ctrlConst = ctx->hScal.lambda;
    pa.loose = true;
    // (the tile stage wrote the sweep's significance bits, in loose-slot index space: LooseCtl)
    pa.looseSweep = ctx->looseOk && (!ctx->par.qval_opt || ctx->lateLoose) && !ctx->knob.noLoose;   // (-q: gx_find_peaks' qLoose)
    pa.latePending = pa.looseSweep && ctx->lateLoose;   // (its bits are still to be written: k_loose_late, by gx_find_peaks)
    pa.looseStride = ctx->looseStride;
    if (pa.looseSweep) pa.chromLooseOff = std::move(ctx->chromLooseOff);
    ctx->looseOk = false;
    ctx->reps.push_back(std::move(pa));
    ctx->sample++;
    ctx->phase = 0;
    return GX_OK;
  } else {
    if (int rc = merge_with_control(ctx, pa)) return rc;
  }
  ctx->reps.push_back(std::move(pa));
  ctx->sample++;
  ctx->phase = 0;
  return GX_OK;
}

int gx_find_peaks(gx_ctx* ctx, size_t* n_peaks, uint64_t* genome_len, uint64_t* peak_bp) {
  if (!ctx || ctx->phase != 0 || ctx->sample < 1) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  u32* misc = ctx->misc.as<u32>();
  // One replicate without a control, -p, and the tile stage left the sweep's bits on the loose slots (LooseCtl): the
  // sweep walks them where they are -- no tight interval table is made unless somebody asks for it later
  const bool looseFast = ctx->sample == 1 && ctx->reps.size() == 1 && ctx->reps[0].loose && ctx->reps[0].looseSweep &&
                         !ctx->par.qval_opt;
  // (one replicate, no control, unit weights, -q, one rank: the tight table's kernel sums "bp at pileup V" on its way -- BH's table
  // of distinct p-values is made of those sums, not of a hash insertion per interval)
  const bool packHist = ctx->par.qval_opt && ctx->sample == 1 && ctx->reps.size() == 1 && ctx->reps[0].ctrlIsConst && !ctx->sawFrac &&
                        ctx->world <= 1 && !ctx->forceColl && !ctx->knob.noPackHist;
  // (round 6) ... and with -q as well, where q is a function of the pileup like p: BH's histogram is summed from the loose slots, q
  // tabulated by whole pileup (k_bh_small), the bits written from "q passes from this pileup on", the sweep as with -p -- no tight table
  const bool qLoose = packHist && ctx->reps[0].loose && ctx->reps[0].looseSweep && (ctx->reps[0].latePending || ctx->reps[0].lateLoose) &&
                      !ctx->knob.noQLoose && !ctx->qLooseBad && !ctx->hasBed;
  ctx->qLooseUsed = false;
  for (size_t r = 0; r < ctx->reps.size(); r++) {
    if (ctx->reps[r].loose && !looseFast && !qLoose)
      if (int rc = materialize_rep(ctx, (int)r, packHist)) return rc;
    // (the Fisher combination of several replicates reuses the loose slots: the last replicate keeps what its pileup
    // floats, if somebody asks for them, are made of)
    if (ctx->sample > 1 && ctx->reps[r].pilesPending)
      if (int rc = keep_loose_for_piles(ctx, (int)r)) return rc;
  }
  if (ctx->sample > 1 && (int)ctx->reps.size() == ctx->sample) {
    if (int rc = combine_replicates(ctx)) return rc;
  }
  ctx->finalIdx = (int)ctx->reps.size() - 1;
  PArray& fa = ctx->reps[ctx->finalIdx];
  const u32 n = fa.n, nChrom = ctx->nChrom;
  // genome length (findPeaks 1091-1101)
  uint64_t g = ctx->par.genome_len;
  const bool genomeOpt = g == 0;
  if (genomeOpt) g = genome_len_for(ctx, fa.present);
  ctx->genomeLenUsed = g;
  ctx->mail->genome = g;
  ctx->mail->n = n;
  // genome length and interval count for the kernels that read them through pointers (BH, k_sig_mask): one tiny
  // kernel instead of two copy launches, and none at all when the masks came with the p-values
  const bool masksReady = looseFast || (ctx->maskIdx == ctx->finalIdx && ctx->maskN == n);
  if (ctx->par.qval_opt || !masksReady)
    hipLaunchKernelGGL(k_set_misc, dim3(1), dim3(1), 0, s, misc, (u32)M_NIV, (u32)M_GENOME, (u64)g, n);

  ctx->lazyQUsed = false;
  if (qLoose) {
    if (int rc = loose_hist(ctx, fa, genomeOpt)) return rc;
  } else if (ctx->par.qval_opt) {
    if (int rc = bh_qvalues(ctx, fa, n, genomeOpt)) return rc;
  }

  // peak sweep
  phase_begin(ctx, "sweep");
  SweepSrc src{};
  src.nChrom = nChrom;
  if (qLoose) {
    // (q by whole pileup and from which pileup on it passes: k_bh_small, above) the bits on the loose slots
    HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, fa.looseStride * 8, s));   // (the significance words: a run before may have left its own)
    hipLaunchKernelGGL(k_loose_late, dim3(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU)))), dim3(256), 0, s,
                       ctx->tileSlot.as<u32>(), ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->nTiles,
                       ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseCtl.as<LooseCtl>(), ctx->swMask.as<u64>(),
                       (const u32*)(misc + M_VQ), ctx->dStatus.as<u32>());
    fa.latePending = false;
    fa.lateLoose = true;
    ctx->lateLooseUsed = true;
    ctx->qLooseUsed = true;
    src.end = ctx->looseEnd.as<u32>();
    src.V = ctx->looseV.as<int>();
    src.p = ctx->pvLut.as<float>();
    src.qLut = ctx->qLut.as<float>();
    src.chromOff = fa.chromLooseOff.as<u32>();
    src.mStride = fa.looseStride;
    src.nWords = (u32)(fa.looseStride - 2);
    src.haveMasks = true;
    src.hasSkip = false;
  } else if (looseFast) {
    ctx->lateLooseUsed = fa.lateLoose || fa.latePending;
    if (fa.latePending) {
      hipLaunchKernelGGL(k_loose_late, dim3(std::max(1u, std::min((ctx->nTiles + 3) / 4, (u32)(8 * ctx->numCU)))), dim3(256), 0, s,
                         ctx->tileSlot.as<u32>(), ctx->tileIvCount.as<u32>(), ctx->tileLastEnd.as<u32>(), ctx->nTiles,
                         ctx->looseEnd.as<u32>(), ctx->looseV.as<int>(), ctx->looseCtl.as<LooseCtl>(), ctx->swMask.as<u64>());
      fa.latePending = false;
      fa.lateLoose = true;
    }
    src.end = ctx->looseEnd.as<u32>();
    src.V = ctx->looseV.as<int>();
    src.p = ctx->pvLut.as<float>();
    src.chromOff = fa.chromLooseOff.as<u32>();
    src.mStride = fa.looseStride;
    src.nWords = (u32)(fa.looseStride - 2);
    src.haveMasks = true;
    src.hasSkip = false;
  } else {
    src.end = fa.end.as<u32>();
    src.p = fa.p.as<float>();
    src.q = ctx->par.qval_opt ? fa.q.as<float>() : (const float*)nullptr;
    if (ctx->par.qval_opt && fa.qLazy) {
      src.kq = ctx->bhKQ.as<u64>();
      src.kqMask = ctx->bhLiveCap - 1;
    }
    src.chromOff = fa.chromOff.as<u32>();
    src.nWords = (n + 63) / 64;
    // the masks were filled by the pack kernels (p mode, one replicate) or by k_qlookup (q mode)
    src.haveMasks = ctx->maskIdx == ctx->finalIdx && ctx->maskN == n;
    src.mStride = src.haveMasks ? ctx->maskStride : (size_t)src.nWords + 2;
    src.hasSkip = true;
    HIPCHECK(ctx->swMask.ensure(src.mStride * 8 * 3));
    // (whoever filled the sig / skip masks zeroed all three: the chromosome-start mask is still clear)
    if (!src.haveMasks) HIPCHECK(hipMemsetAsync(ctx->swMask.p, 0, src.mStride * 8 * 3, s));
  }
  ctx->maskIdx = -1;
  ctx->looseSwept = looseFast || qLoose;
  u32 nPeaks = 0;
  const int rcSweep = run_sweep(ctx, src, &nPeaks);
  phase_end(ctx);
  if (qLoose && (ctx->mail->status & ST_Q_LOOSE)) {
    // q turned out to be no threshold on the pileup (a table p(V) that is not monotone): once more, on the tight table -- and from now on
    HIPCHECK(hipMemsetAsync(ctx->dStatus.p, 0, 4, s));
    ctx->qLooseBad = true;
    return gx_find_peaks(ctx, n_peaks, genome_len, peak_bp);
  }
  if (rcSweep) return rcSweep;
  if (n_peaks) *n_peaks = nPeaks;
  if (genome_len) *genome_len = g;
  if (peak_bp) *peak_bp = ctx->peakBP;
  return GX_OK;
}

int gx_peak_count(gx_ctx* ctx, size_t* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = ctx->nHostPeaks;
  return GX_OK;
}

int gx_get_peaks(gx_ctx* ctx, gx_peak* out, size_t cap) {
  if (!ctx || !out) return GX_ERR_ORDER;
  size_t n = std::min(cap, ctx->nHostPeaks);
  if (n) memcpy(out, ctx->hPeaks.p, n * sizeof(gx_peak));
  return GX_OK;
}

static const PArray* which_array(gx_ctx* ctx, int which, int chrom, u32* lo, u32* hi) {
  if (chrom < 0 || (u32)chrom >= ctx->nChrom) return nullptr;
  int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
  if (w < 0 || w >= (int)ctx->reps.size()) return nullptr;
  if (ctx->reps[w].loose && materialize_rep(ctx, w) != GX_OK) return nullptr;
  const PArray& pa = ctx->reps[w];
  if (!pa.present[chrom]) return nullptr;
  u32 off[2];
  if (hipMemcpy(off, pa.chromOff.as<u32>() + chrom, 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
  *lo = off[0];
  *hi = off[1];
  return &pa;
}

int gx_interval_count(gx_ctx* ctx, int which, int chrom, size_t* n_iv) {
  if (!ctx || !n_iv) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  u32 lo = 0, hi = 0;
  const PArray* pa = which_array(ctx, which, chrom, &lo, &hi);
  *n_iv = pa ? hi - lo : 0;
  return GX_OK;
}

int gx_get_intervals(gx_ctx* ctx, int which, int chrom, size_t cap, uint32_t* end, float* expt, float* ctrl, float* p,
                     float* q) {
  if (!ctx) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  u32 lo = 0, hi = 0;
  const PArray* pa = which_array(ctx, which, chrom, &lo, &hi);
  if (!pa) return GX_OK;
  size_t n = std::min<size_t>(cap, hi - lo);
  if ((expt || ctrl) && pa->pilesPending) {
    const int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
    if (int rc = ensure_piles(ctx, w)) return rc;
    HIPCHECK(hipStreamSynchronize(ctx->stream));
  }
  if ((expt || ctrl) && pa->pilesDropped) {
    ctx->err = "the pileup values were not kept (gx_set_keep_pileups)";
    return GX_ERR_ORDER;
  }
  if (!n) return GX_OK;
  if (end) HIPCHECK(hipMemcpy(end, pa->end.as<u32>() + lo, n * 4, hipMemcpyDeviceToHost));
  if (p) HIPCHECK(hipMemcpy(p, pa->p.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
  if (expt) {
    if (pa->hasPiles) HIPCHECK(hipMemcpy(expt, pa->expt.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    else std::fill(expt, expt + n, 0.0f);
  }
  if (ctrl) {
    if (pa->hasPiles && !pa->ctrlIsConst) HIPCHECK(hipMemcpy(ctrl, pa->ctrl.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    else std::fill(ctrl, ctrl + n, pa->ctrlIsConst ? pa->ctrlConst : 0.0f);
  }
  if (q) {
    if ((pa->q.p || pa->qLazy) && ctx->par.qval_opt) {
      const int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
      if (int rc = ensure_q(ctx, ctx->reps[w], w)) return rc;
      HIPCHECK(hipStreamSynchronize(ctx->stream));
      HIPCHECK(hipMemcpy(q, pa->q.as<float>() + lo, n * 4, hipMemcpyDeviceToHost));
    }
    if (!((pa->q.p || pa->qLazy) && ctx->par.qval_opt)) std::fill(q, q + n, GX_SKIP);
  }
  return GX_OK;
}

int gx_selftest2(gx_ctx* ctx, int what, const float* a, const float* b, float* out, double* out_double, size_t n,
                 size_t* n_risky) {
  if (!ctx || !a || !out || !n || n > 0xFFFFFFFFull) return GX_ERR_ORDER;
  HIPCHECK(hipSetDevice(ctx->device));
  DevBuf da, db, dout, ddbl;
  HIPCHECK(da.ensure(n * 4));
  HIPCHECK(db.ensure(n * 4));
  HIPCHECK(dout.ensure(n * 4));
  if (out_double) HIPCHECK(ddbl.ensure(n * 8));
  HIPCHECK(hipMemcpy(da.p, a, n * 4, hipMemcpyHostToDevice));
  if (b) HIPCHECK(hipMemcpy(db.p, b, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest, dim3(1024), dim3(256), 0, ctx->stream, what, da.as<float>(), db.as<float>(),
                     dout.as<float>(), ddbl.as<double>(), (u32)n, ctx->dRisk.as<RiskBuf>());
  if (int rc__ = dbg_sync(ctx, "k_selftest")) return rc__;
  if (int rc__ = mail_sync(ctx, nullptr, nullptr, nullptr, nullptr, nullptr)) return rc__;
  if (n_risky) *n_risky = static_cast<RiskBuf*>(ctx->riskHost.p)->count;
  RiskTargets T{};
  T.selfOut = dout.as<float>();
  if (int rc__ = risk_apply(ctx, T, RiskHostIn{a, b})) return rc__;
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  HIPCHECK(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
  if (out_double) HIPCHECK(hipMemcpy(out_double, ddbl.p, n * 8, hipMemcpyDeviceToHost));
  return GX_OK;
}

int gx_selftest(gx_ctx* ctx, int what, const float* a, const float* b, float* out, size_t n) {
  return gx_selftest2(ctx, what, a, b, out, nullptr, n, nullptr);
}

// the same scalar routines compiled for the host (what 1: calcPval, 3: multPval's tail): what the library
// evaluates risky values with, i.e. with this machine's libm; no context, no device
int gx_selftest_host(int what, const float* a, const float* b, float* out, double* out_double, size_t n) {
  if (!a || !b || !out || (what != 1 && what != 3 && what != 4)) return GX_ERR_ORDER;
  for (size_t i = 0; i < n; i++) {
    bool rk = false;
    double d = 0.0;
    if (what == 1) {
      out[i] = calc_pval(a[i], b[i], &rk);
      if (a[i] > 0.0f && b[i] > 0.0f) {
        double ml, sl;
        lnorm_params(b[i], &ml, &sl);
        d = pval_double(a[i], log((double)a[i]), ml, sl);
      }
    } else if (what == 3) {
      out[i] = fisher_combine((double)a[i], (int)b[i], &rk);
      if ((int)b[i] > 2 && a[i] != 0.0f) d = fisher_double((double)a[i], (int)b[i]);
    } else {   // (4: the closed form, host build -- NOT what the library's host side ever rounds: a check of the formula itself)
      out[i] = fisher_fast((double)a[i], (int)b[i], &rk);
      if ((int)b[i] > 2 && a[i] != 0.0f) d = fisher_fast_double((double)a[i], (int)b[i]);
    }
    if (out_double) out_double[i] = d;
  }
  return GX_OK;
}

int gx_total_intervals(gx_ctx* ctx, int which, size_t* n_iv) {
  if (!ctx || !n_iv) return GX_ERR_ORDER;
  int w = which == GX_IV_FINAL ? ctx->finalIdx : which;
  if (w < 0 || w >= (int)ctx->reps.size()) return GX_ERR_ORDER;
  *n_iv = ctx->reps[w].n;
  return GX_OK;
}

int gx_set_phase_filter(gx_ctx* ctx, const char* name) {
  if (!ctx || !name) return GX_ERR_ORDER;
  ctx->phaseFilter = name;
  ctx->phaseLevel = 1;
  return GX_OK;
}

int gx_rccl_nranks(gx_ctx* ctx, int* n) {
  if (!ctx || !n) return GX_ERR_ORDER;
  *n = 0;
  if (ctx->comm) {
    const gxrccl::Api* api = gxrccl::load(nullptr);
    if (api && api->commCount && api->commCount(ctx->comm, n) != ncclSuccess) *n = 0;
  }
  return GX_OK;
}

int gx_path_info(gx_ctx* ctx, unsigned* flags) {
  if (!ctx || !flags) return GX_ERR_ORDER;
  *flags = (ctx->fusedUsed ? GX_PATH_FUSED : 0u) | (ctx->fusedUsed && ctx->pairsUsed ? GX_PATH_PAIRS : 0u) | (ctx->denseBhUsed ? GX_PATH_DENSE_BH : 0u) | (ctx->rangeBhUsed ? GX_PATH_RANGE_BH : 0u) | (ctx->looseSwept ? GX_PATH_LOOSE_SWEEP : 0u) |
           (ctx->fellBack ? GX_PATH_FELL_BACK : 0u) | (ctx->ptGrew ? GX_PATH_PT_GREW : 0u) | (ctx->fusedUsed && ctx->fracPairsUsed ? GX_PATH_FRAC_PAIRS : 0u) |
           (ctx->pilesMade ? GX_PATH_PILES_MADE : 0u) | (ctx->packedUsed ? GX_PATH_PACKED : 0u) | (ctx->mergePUsed ? GX_PATH_MERGE_P : 0u) |
           (ctx->denseHistUsed ? GX_PATH_PACK_HIST : 0u) | (ctx->lazyQUsed ? GX_PATH_LAZY_Q : 0u) | (ctx->looseSwept && ctx->lateLooseUsed ? GX_PATH_LATE_LOOSE : 0u) | (ctx->qLooseUsed ? GX_PATH_Q_LOOSE : 0u);
  return GX_OK;
}

int gx_set_phase_timing(gx_ctx* ctx, int level) {
  if (!ctx || level < 0 || level > 2) return GX_ERR_ORDER;
  ctx->phaseLevel = level;
  return GX_OK;
}

int gx_phase_times(gx_ctx* ctx, const char** names, const float** ms) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->phaseMs.clear();
  ctx->phaseNames.clear();
  for (size_t i = 0; i < ctx->nPhases; i++) {
    Phase& ph = ctx->phases[i];
    float t = 0;
    (void)hipEventElapsedTime(&t, ph.a, ph.b);
    ctx->phaseMs.push_back(t);
    ctx->phaseNames += ph.name;
    ctx->phaseNames.push_back('\0');
  }
  if (names) *names = ctx->phaseNames.c_str();
  if (ms) *ms = ctx->phaseMs.data();
  return (int)ctx->nPhases;
}

}  // extern "C"
