// genrich_amd.cpp -- host program: the Genrich command line over the MI355X hot path.
//
// Everything upstream of the C ABI stays on the CPU, as north_star prescribes: option handling
// (getArgs, Genrich.c:5718-5827), SAM / BAM reading (readSAM 4468, readBAM 4983), pairing of
// alignments into fragments (parseAlign 4141, processAlns 3187), multimap weighting
// (processPair 3122, processSingle 3019), interval geometry (saveFragment 2754, saveFragAtac
// 2728, saveUnpair 2689, processAvgExt 2614) and saveInterval's clamping / -b line (2516-2591).
// Each alignment-derived interval becomes one gx_event; pileups, p/q-values and peaks come
// from libgenrich_amd.so; text output is gx_emit.cpp.  Written from the behaviour described in
// SURVEY.md Appendix A, not transliterated from the reference.
//
// -P (peak calling from a -f log, callPeaksLog 1277-1488) is text processing and runs on the host,
// and so does -r / -R (PCR-duplicate removal, Genrich.c:2776-2977 and 3267-4042).
//
// Extra long options:
//   --threads N     threads that inflate BGZF (BAM, bgzip-ped SAM) input, as many that decode records (SAM lines,
//                   BAM blocks: everything that does not touch the run's state) and as many that run the pairing /
//                   weighting state machine over chunks of whole read-name groups (results merged in file order);
//                   default min(16, cores); 1: one thread does it all, in the reference's order of operations
//   (environment, for tests: GENRICH_BATCH_BYTES, GENRICH_CHUNK_RECS -- where the input is cut; GENRICH_SERIAL_STATE=1 --
//   one thread for the state machine)
//   (environment: GENRICH_HOST_PROF=1 prints the CPU time of the parsing thread after the last input; GENRICH_DUPS_HOST=1
//   keeps -r's tables on the host; GENRICH_NO_INT16=1 leaves saveInterval's int16 checks to the library -- the pileup is the
//   reference's either way, the -v warnings / -b lines / -x average of dropped reads only with the checks on; GENRICH_NO_MMAP=1
//   reads plain files through read() instead of mapping them; GENRICH_ZLIB_INFLATE=1 inflates BGZF blocks with zlib instead of
//   gx_inflate.h; GENRICH_THREADS_REPORT=1 prints how the thread budget was split)
//   --events-only   parse and write the -b file without touching a GPU (diagnostics)
//   --device N      HIP device ordinal (default 0)
//   --devices LIST  several GPUs of one node, e.g. 0-7 or 0,2,5: chromosomes are sharded over them by length, one
//                   library context and one host thread per GPU, RCCL inside the library for the two genome-wide
//                   exchanges, outputs identical to a single-GPU run (a device named twice, e.g. 0,0, runs the
//                   same protocol through host callbacks: the test mode for a single GPU)
#include <getopt.h>
#include <zlib.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <pthread.h>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../include/genrich_amd.h"
#include "bgzf_reader.h"
#include "../csrc/gx_saturate.h"  // (host-only: the canonical int16 part of a difference, the weights)

#define VERSION "0.6.2-amd"
#define MAX_ALNS 128  // Genrich.h:17 (also the length of stored read names)
static const float NOSCORE = -FLT_MAX;

namespace {

// A decoding thread (parallel record decoding, below) must not end the program: an earlier record, still on its way
// through the run's state, may have a warning to print or an error of its own.  With t_capture set, die() keeps the
// message and unwinds to the decoder; the thread that owns the state dies with it when that record's turn comes.
struct DecodeAbort {};
thread_local std::pair<std::string, std::string>* t_capture = nullptr;

[[noreturn]] void die(const std::string& msg, const char* tail) {  // error(), Genrich.c:78-81
  if (t_capture) {
    *t_capture = {msg, tail};
    throw DecodeAbort{};
  }
  fprintf(stderr, "Error! %s%s\n", msg.c_str(), tail);
  // What the reference's exit() does that can be seen from outside is flushing its open streams; the reader, decoder,
  // inflate and state threads of the ingest may still be running here, so nothing else is torn down under them (no
  // static destructors, no atexit handlers -- the HIP runtime's among them -- next to live threads).
  fflush(nullptr);
  _exit(EXIT_FAILURE);
}

int getInt(const char* s) {  // 102-108
  char* end;
  long v = strtol(s, &end, 10);
  if (*end != '\0') die(s, ": cannot convert to int");
  return (int)v;
}
long getLong(const char* s) {
  char* end;
  long v = strtol(s, &end, 10);
  if (*end != '\0') die(s, ": cannot convert to int");
  return v;
}
float getFloat(const char* s) {  // 90-96
  char* end;
  float v = strtof(s, &end);
  if (*end != '\0') die(s, ": cannot convert to float");
  return v;
}

// ---- output files: plain or gzip (-z), printf-style ---------------------------------------
struct Out {
  FILE* f = nullptr;
  gzFile gz = nullptr;
  std::string name;
};

ssize_t gzCookieWrite(void* c, const char* buf, size_t n) { return gzwrite((gzFile)c, buf, (unsigned)n) ? (ssize_t)n : 0; }
int gzCookieClose(void* c) { return gzclose((gzFile)c) == Z_OK ? 0 : -1; }

Out openWrite(const char* path, bool gzOut) {  // openWrite, 5076-5102
  Out o;
  if (path[0] == '-' && strlen(path) > 1) die(path, ": output filename cannot start with '-'");
  bool isStdout = !strcmp(path, "-");
  o.name = path;
  if (gzOut) {
    if (!isStdout && (o.name.size() < 3 || o.name.compare(o.name.size() - 3, 3, ".gz"))) o.name += ".gz";
    gzFile g = isStdout ? gzdopen(fileno(stdout), "wb") : gzopen(o.name.c_str(), "w");
    if (!g) die(o.name, ": cannot open file for writing");
    cookie_io_functions_t io = {nullptr, gzCookieWrite, nullptr, gzCookieClose};
    o.f = fopencookie(g, "w", io);
    o.gz = g;
  } else
    o.f = isStdout ? stdout : fopen(path, "w");
  if (!o.f) die(o.name, ": cannot open file for writing");
  return o;
}
void closeOut(Out& o) {
  if (o.f && o.f != stdout && fclose(o.f)) die(o.name, ": cannot close file");
  if (o.f == stdout) fflush(stdout);
  o.f = nullptr;
}

// ---- chromosome table (saveChrom 4220-4270, saveXBed 1144-1206) -----------------------------
struct Chrom {
  std::string name;
  uint32_t len = 0;
  bool skip = false, save = false;
  std::vector<uint32_t> bed;  // merged start/end pairs
};
struct BedRec { std::string name; uint32_t pos[2]; };

struct Opts {
  const char *inFile = nullptr, *ctrlFile = nullptr, *outFile = nullptr, *logFile = nullptr, *pileFile = nullptr,
             *bedFile = nullptr, *xFile = nullptr, *dupsFile = nullptr, *xchrom = nullptr;
  uint64_t genomeLen = 0;
  int extend = 0, minMapQ = 0, minLen = 0, maxGap = 100, atacLen5 = 100, atacLen3 = 0;
  float asDiff = 0.0f, pqvalue = 0.01f, minAUC = 200.0f;
  bool singleOpt = false, extendOpt = false, avgExtOpt = false, atacOpt = false, atacAdj = true, gzOut = false,
       qvalOpt = false, dupsOpt = false, peaksOpt = true, peaksOnly = false, sortOpt = true, verbose = false,
       eventsOnly = false;
  int device = 0;
  std::vector<int> devices;  // --devices a,b,...: one context per GPU, chromosomes sharded by length
};

// ---- several GPUs of one node -------------------------------------------------------
// One library context per device, chromosomes dealt out by longest-processing-time bin packing; every event
// goes to the context that owns its chromosome; gx_sample_end / gx_pvalues / gx_find_peaks run on one host
// thread per device because they contain the run's collectives (fragLen sums, the BH table): RCCL inside the
// library when the devices are distinct, in-process callbacks when a device is named twice (a test on one GPU).
struct Devs {
  std::vector<gx_ctx*> ctx;
  std::vector<int> owner;                       // chromosome -> index into ctx
  std::vector<std::vector<gx_event>> buf;       // events on their way to each context
  // in-process collectives (callback mode)
  pthread_barrier_t bar;
  bool barInit = false;
  std::vector<std::vector<int64_t>> red;
  struct User { Devs* d; int rank; };
  std::vector<User> users;
  size_t n() const { return ctx.size(); }
};

int devsAllreduce(int64_t* buf, size_t n, void* user) {
  Devs::User* u = static_cast<Devs::User*>(user);
  Devs& D = *u->d;
  D.red[u->rank].assign(buf, buf + n);
  pthread_barrier_wait(&D.bar);
  for (size_t k = 0; k < n; k++) {
    int64_t sum = 0;
    for (size_t r = 0; r < D.n(); r++) sum += D.red[r][k];
    buf[k] = sum;
  }
  pthread_barrier_wait(&D.bar);
  return 0;
}

// ---- saveInterval's int16 checks, read by read (Genrich.c:2558-2573) ----------------------------------------------
// The reference drops an alignment whose start lies on a base where its int16 difference already holds 32,767, or whose
// end lies on one that holds -32,768: a warning with -v, no -b line, length 0 towards the -x average.  The library
// reproduces the effect on the pileup by itself (gx_sample_end); what the reference PRINTS needs the decision when the
// read is saved, in input order.  A base can only get there when its 4096-base window holds 32,766 starts (or ends), so:
// the thread that owns the state counts the weight of starts and of ends per window (two increments per event); a window
// that comes that far is "loaded" -- its exact difference per base so far comes from the device, which has every event
// pushed up to now (gx_window_net) -- and kept on the host from then on.  Reads that touch a loaded window, or bring one
// to the threshold, are decided on exact state; everything else cannot be dropped.  Real data never loads a window.
struct HotWindows {
  static constexpr int WB = 12;
  static constexpr uint32_t WMASK = (1u << WB) - 1u;
  bool on = false;
  std::vector<size_t> base;                                   // first window of a chromosome (position `len` has an entry)
  std::vector<uint32_t> ws, we;                               // weight (1/120) of the sample's starts / ends per window
  std::unordered_map<size_t, std::vector<long long>> win;     // loaded windows: the exact difference per base
  // --events-only -b (no device to ask for a window's state): the sample's events so far, 16 bytes each -- 1.6 GB per 10^8
  // fragments, scanned once per window that comes near the limits (real data: none).  GENRICH_NO_INT16=1 drops the copy
  // (and the read-by-read decisions) for event dumps of that size.
  std::vector<gx_event> all;
  void init(const std::vector<Chrom>& chrom) {
    base.assign(chrom.size() + 1, 0);
    for (size_t c = 0; c < chrom.size(); c++) base[c + 1] = base[c] + ((size_t)chrom[c].len >> WB) + 1;
    ws.assign(base.back(), 0);
    we.assign(base.back(), 0);
    win.clear();
    all.clear();
  }
  // an event of a chunk that the workers prepared: counted; true when it has to be decided on exact state instead
  bool add(const gx_event& e) {
    const uint32_t w = (uint32_t)gxsat::weight_of(e.count);
    const size_t a = base[e.chrom] + (e.start >> WB), b = base[e.chrom] + (e.end >> WB);
    ws[a] += w;
    we[b] += w;
    return ws[a] >= (uint32_t)gxsat::HOT || we[b] >= (uint32_t)gxsat::HOT || (!win.empty() && (win.count(a) || win.count(b)));
  }
  void sub(const gx_event& e) {
    const uint32_t w = (uint32_t)gxsat::weight_of(e.count);
    ws[base[e.chrom] + (e.start >> WB)] -= w;
    we[base[e.chrom] + (e.end >> WB)] -= w;
  }
};

struct State {
  Opts o;
  std::vector<Chrom> chrom;
  std::vector<std::string> xchr;
  std::vector<BedRec> xbed;
  gx_ctx* gx = nullptr;      // the (first) device context; nullptr with --events-only
  Devs devs;                 // all of them
  bool tableFrozen = false;  // the device already holds the chromosome table
  std::unique_ptr<gxhost::Input> stdinIn;  // '-' can be opened once: its header pre-scan is replayed to the real pass
  bool sampleOpen = false;   // gx_sample_begin done for the file being read
  std::vector<gx_event> buf;
  Out bed;
  bool bedOpt = false;
  Out dups;               // -R: log of the reads removed as PCR duplicates
  bool dupsVerb = false;
  bool ctrl = false;
  int sample = 0;
  uint64_t errCount = 0;
  HotWindows hot;            // (with a device: saveInterval's int16 checks)
};

void check(State& S, int rc, gx_ctx* which = nullptr) {
  if (rc == GX_OK) return;
  gx_ctx* g = which ? which : S.gx;
  std::string detail = g ? gx_last_error(g) : "";
  die(detail.empty() ? std::string(gx_strerror(rc)) : detail, "");
}

// f(context index) on every device context: in place for one, one host thread each for several (the calls
// that contain collectives must run side by side); the first failure ends the program like any other,
template <typename F>
void onEachDevice(State& S, F f) {
  Devs& D = S.devs;
  if (D.n() == 1) {
    check(S, f(0), D.ctx[0]);
    return;
  }
  // A context that fails ends the program FROM ITS OWN THREAD: the others may already be waiting for it inside a
  // collective (ncclAllReduce / ncclAllGather, or the barrier of the callback mode), where a join would wait forever.
  std::vector<std::thread> th;
  for (size_t g = 0; g < D.n(); g++)
    th.emplace_back([&, g] {
      const int rc = f((int)g);
      if (rc != GX_OK) {
        const std::string detail = gx_last_error(D.ctx[g]);
        fprintf(stderr, "Error! %s\n", detail.empty() ? gx_strerror(rc) : detail.c_str());
        fflush(nullptr);
        _exit(EXIT_FAILURE);  // (no atexit handlers: the other threads' device work is still in flight)
      }
    });
  for (auto& t : th) t.join();
}

// the events gathered so far go to the contexts that own their chromosomes
void flushEvents(State& S) {
  Devs& D = S.devs;
  if (D.n() == 1) {
    if (!S.buf.empty()) check(S, gx_push_events(S.gx, S.buf.data(), S.buf.size()));
    return;
  }
  for (auto& b : D.buf) b.clear();
  for (const gx_event& e : S.buf) D.buf[D.owner[e.chrom]].push_back(e);
  for (size_t g = 0; g < D.n(); g++)
    if (!D.buf[g].empty()) check(S, gx_push_events(D.ctx[g], D.buf[g].data(), D.buf[g].size()), D.ctx[g]);
}

void mergeBed(Chrom& c, const std::vector<BedRec>& xbed, bool verbose) {
  std::vector<std::pair<uint32_t, uint32_t>> iv;
  for (const BedRec& b : xbed)
    if (b.name == c.name) {
      if (b.pos[0] >= c.len) {
        if (verbose) {
          fprintf(stderr, "Warning! BED interval (%s, %d - %d) ignored\n", b.name.c_str(), b.pos[0], b.pos[1]);
          fprintf(stderr, "  - located off end of reference %s (length %d)\n", c.name.c_str(), c.len);
        }
        continue;
      }
      // insertion before the first interval whose start is >= this start (1165-1175)
      size_t j = 0;
      while (j < iv.size() && b.pos[0] > iv[j].first) j++;
      iv.insert(iv.begin() + j, {b.pos[0], b.pos[1]});
    }
  std::vector<std::pair<uint32_t, uint32_t>> out;
  for (auto& p : iv) {
    if (p.second > c.len) {
      if (verbose) {
        fprintf(stderr, "Warning! BED interval (%s, %d - %d) extends ", c.name.c_str(), p.first, p.second);
        fprintf(stderr, "past end of ref.\n  - edited to (%s, %d - %d)\n", c.name.c_str(), p.first, c.len);
      }
      p.second = c.len;
    }
    if (!out.empty() && p.first <= out.back().second) {
      if (p.second > out.back().second) out.back().second = p.second;
    } else
      out.push_back(p);
  }
  for (auto& p : out) {
    c.bed.push_back(p.first);
    c.bed.push_back(p.second);
  }
}

int saveChrom(State& S, const char* name, uint32_t len) {
  for (size_t i = 0; i < S.chrom.size(); i++)
    if (S.chrom[i].name == name) {
      if (S.chrom[i].len != len) die(name, ": reference sequence has different lengths in BAM/SAM files");
      if (!S.ctrl) S.chrom[i].save = true;
      return (int)i;
    }
  if (S.tableFrozen)  // every header was pre-scanned: this is an @SQ line that follows the first record of its file
    die(name, ": reference sequence first seen after the chromosome table was sent to the device");
  Chrom c;
  c.name = name;
  c.len = len;
  for (auto& x : S.xchr)
    if (x == name) c.skip = true;
  c.save = !S.ctrl;  // do not save if ref in ctrl sample only
  if (!c.skip) mergeBed(c, S.xbed, S.o.verbose);
  S.chrom.push_back(c);
  return (int)S.chrom.size() - 1;
}

// ---- where a read-name group's results go -----------------------------------------------------
// Read-name groups do not depend on each other, only their results have an order: the events (the int16 replay of
// the library goes by it), the -b lines, the -v warnings (and the count that suppresses them after the first 128),
// the additions into the float sum behind the -x average.  A worker thread that takes a chunk of whole groups
// (parallel state, below) has `t_sink` set: everything that would touch the run's state is written there instead and
// merged into the state by its owner, chunk after chunk in file order.  Without a sink (one thread; the tails of
// finishFile) the state is written directly, as before.
struct Sink {
  std::vector<gx_event> ev;
  std::string bed;                                      // -b lines
  std::vector<std::pair<std::string, bool>> warn;       // stderr lines in order; true: counts towards errCount
  std::vector<double> lenTerms;                         // C.totalLen += ... in order (a float sum: not associative)
};
thread_local Sink* t_sink = nullptr;

std::string vformat(const char* fmt, va_list ap) {
  va_list ap2;
  va_copy(ap2, ap);
  const int n = vsnprintf(nullptr, 0, fmt, ap2);
  va_end(ap2);
  std::string out((size_t)std::max(0, n), '\0');
  if (n > 0) vsnprintf(&out[0], (size_t)n + 1, fmt, ap);
  return out;
}
// a warning that takes part in the "(another N warning messages suppressed)" count (saveInterval's two)
void warnCounted(State& S, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (t_sink)
    t_sink->warn.emplace_back(vformat(fmt, ap), true);
  else {
    if (S.errCount < MAX_ALNS) vfprintf(stderr, fmt, ap);
    S.errCount++;
  }
  va_end(ap);
}
// ... and one that does not
void warnPlain(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (t_sink) t_sink->warn.emplace_back(vformat(fmt, ap), false);
  else vfprintf(stderr, fmt, ap);
  va_end(ap);
}
void pushEvent(State& S, const gx_event& e) {  // (state's side: the library takes the events in pieces of 2^20)
  if (!S.gx && S.hot.on) S.hot.all.push_back(e);
  S.buf.push_back(e);
  if (S.buf.size() >= (1u << 20)) {
    if (S.gx && S.sampleOpen) flushEvents(S);
    if (!S.gx || S.sampleOpen) S.buf.clear();  // (--events-only: nothing to send them to; with -b the int16 decisions keep their own copy, hot.all)
  }
}

// A loaded window's exact difference (1/120 units per base): from the device, which is sent what the host still holds first.
std::vector<long long>& hotLoad(State& S, uint32_t ci, size_t gw) {
  auto it = S.hot.win.find(gw);
  if (it != S.hot.win.end()) return it->second;
  if (S.gx && S.sampleOpen && !S.buf.empty()) {
    flushEvents(S);
    S.buf.clear();
  }
  std::vector<long long> net((size_t)1 << HotWindows::WB, 0);
  const uint32_t pos0 = (uint32_t)((gw - S.hot.base[ci]) << HotWindows::WB);
  const uint32_t n = std::min<uint32_t>(1u << HotWindows::WB, S.chrom[ci].len + 1 - pos0);
  if (!S.gx) {  // --events-only: no device to ask; the sample's events were kept for this
    for (const gx_event& e : S.hot.all) {
      if (e.chrom != ci) continue;
      const long long w = gxsat::weight_of(e.count);
      if (e.start - pos0 < n) net[e.start - pos0] += w;
      if (e.end - pos0 < n) net[e.end - pos0] -= w;
    }
  } else if (S.sampleOpen) {
    gx_ctx* g = S.devs.n() == 1 ? S.gx : S.devs.ctx[S.devs.owner[ci]];
    check(S, gx_window_net(g, ci, pos0, n, net.data()), g);
  }
  return S.hot.win.emplace(gw, std::move(net)).first->second;
}
// the state's owner, one event at a time: false when the reference would have dropped it (its warning printed)
bool hotKeeps(State& S, const gx_event& e, const char* qname) {
  HotWindows& H = S.hot;
  const long long w = gxsat::weight_of(e.count);
  if (!w) return true;  // (the library reports the count: ERRALNS)
  const size_t a = H.base[e.chrom] + (e.start >> HotWindows::WB), b = H.base[e.chrom] + (e.end >> HotWindows::WB);
  const bool nearA = H.ws[a] + (uint32_t)w >= (uint32_t)gxsat::HOT, nearB = H.we[b] + (uint32_t)w >= (uint32_t)gxsat::HOT;
  if (nearA || nearB || !H.win.empty()) {
    std::vector<long long>* wa = nearA || H.win.count(a) ? &hotLoad(S, e.chrom, a) : nullptr;
    std::vector<long long>* wb = nearB || H.win.count(b) ? &hotLoad(S, e.chrom, b) : nullptr;
    const Chrom& c = S.chrom[e.chrom];
    if (wa && gxsat::canon_cov((*wa)[e.start & HotWindows::WMASK]) == 32767) {
      if (S.o.verbose) {
        fprintf(stderr, "Warning! Read %s, alignment at (%s, %ld-%ld)", qname, c.name.c_str(), (long)e.start, (long)e.end);
        fprintf(stderr, " skipped due to overflow\n");
      }
      return false;
    }
    if (wb && gxsat::canon_cov((*wb)[e.end & HotWindows::WMASK]) == -32768) {
      if (S.o.verbose) {
        fprintf(stderr, "Warning! Read %s, alignment at (%s, %ld-%ld)", qname, c.name.c_str(), (long)e.start, (long)e.end);
        fprintf(stderr, " skipped due to underflow\n");
      }
      return false;
    }
    if (wa) (*wa)[e.start & HotWindows::WMASK] += w;
    if (wb) (*wb)[e.end & HotWindows::WMASK] -= w;
  }
  H.ws[a] += (uint32_t)w;
  H.we[b] += (uint32_t)w;
  return true;
}

// ---- saveInterval (2516-2591): clamp, event, -b line ----------------------------------------
uint32_t saveInterval(State& S, int ci, int64_t start, int64_t end, const char* qname, uint8_t count) {
  Chrom& c = S.chrom[ci];
  if (start < 0) {
    if (S.o.verbose) warnCounted(S, "Warning! Read %s prevented from extending below 0 on %s\n", qname, c.name.c_str());
    start = 0;
  }
  if (start >= c.len) {
    std::string msg = std::string("Read ") + qname + ", ref. " + c.name;
    if (msg.size() > 511) msg.resize(511);  // (snprintf into char msg[512])
    die(msg, ": read aligned beyond reference end");
  }
  if (end > c.len) {
    if (S.o.verbose) warnCounted(S, "Warning! Read %s prevented from extending past %d on %s\n", qname, c.len, c.name.c_str());
    end = c.len;
  }
  const gx_event e{(uint32_t)ci, (uint32_t)start, (uint32_t)end, count};
  if (t_sink) t_sink->ev.push_back(e);
  else {
    if (S.hot.on && !hotKeeps(S, e, qname)) return 0;  // 2558-2573
    pushEvent(S, e);
  }
  if (S.bedOpt) {
    if (t_sink) {
      char num[96];
      std::string& b = t_sink->bed;
      b += c.name;
      snprintf(num, sizeof num, "\t%ld\t%ld\t", (long)start, (long)end);
      b += num;
      b += qname;
      snprintf(num, sizeof num, "_%d_%c_%d\n", count, S.ctrl ? 'C' : 'E', S.sample);
      b += num;
    } else
      fprintf(S.bed.f, "%s\t%ld\t%ld\t%s_%d_%c_%d\n", c.name.c_str(), (long)start, (long)end, qname, count,
              S.ctrl ? 'C' : 'E', S.sample);
  }
  return (uint32_t)(end - start);
}

// ---- alignments of one read name -------------------------------------------------------------
struct Aln {
  uint32_t pos[2];
  float score;
  bool primary, paired, full, first, strand;
  int chrom;
};
struct Unpair { int chrom; uint32_t pos[2]; bool strand; uint8_t count; std::string name; };

struct Counts {
  uint64_t count = 0, unmapped = 0, paired = 0, single = 0, orphan = 0, pairedPr = 0, singlePr = 0, supp = 0,
           skipped = 0, lowMapQ = 0, secPair = 0, secSingle = 0;
  uint64_t countPr = 0, dupsPr = 0, countDc = 0, dupsDc = 0, countSn = 0, dupsSn = 0;  // -r
  double totalLen = 0.0;
};

// (positions are uint32_t in the reference and its sums with the extension / ATAC lengths wrap in 32 bits
// before they reach saveInterval's int64_t parameters; a position can be "negative" -- wrapped -- when the
// BAM reader made it from a record without SEQ)
uint32_t saveFragment(State& S, const char* qname, const Aln& a, uint8_t count) {  // 2754-2774, 2728-2749
  uint32_t start = a.pos[0], end = a.pos[1];
  if (start > end) std::swap(start, end);
  if (!S.o.atacOpt) return saveInterval(S, a.chrom, start, end, qname, count);
  if (S.o.atacAdj) { start += 5; end += (uint32_t)-5; }
  if (start + S.o.atacLen3 >= (uint32_t)(int)(end - S.o.atacLen3))
    return saveInterval(S, a.chrom, (int)(start - S.o.atacLen5), (uint32_t)(end + S.o.atacLen5), qname, count);
  return saveInterval(S, a.chrom, (int)(start - S.o.atacLen5), (uint32_t)(start + S.o.atacLen3), qname, count) +
         saveInterval(S, a.chrom, (int)(end - S.o.atacLen3), (uint32_t)(end + S.o.atacLen5), qname, count);
}

void saveUnpair(State& S, const char* qname, Aln& a, uint8_t count) {  // 2689-2721
  const Opts& o = S.o;
  if (o.extendOpt) {
    if (a.strand) saveInterval(S, a.chrom, a.pos[0], (uint32_t)(a.pos[0] + o.extend), qname, count);
    else saveInterval(S, a.chrom, (int)(a.pos[1] - o.extend), a.pos[1], qname, count);
  } else if (o.atacOpt) {
    if (a.strand) {
      if (o.atacAdj) a.pos[0] += 5;
      saveInterval(S, a.chrom, (int)(a.pos[0] - o.atacLen5), (uint32_t)(a.pos[0] + o.atacLen3), qname, count);
    } else {
      if (o.atacAdj) a.pos[1] += (uint32_t)-5;
      saveInterval(S, a.chrom, (int)(a.pos[1] - o.atacLen3), (uint32_t)(a.pos[1] + o.atacLen5), qname, count);
    }
  } else
    saveInterval(S, a.chrom, a.pos[0], a.pos[1], qname, count);
}

bool usable(const State& S, const Aln& a) { return S.chrom[a.chrom].save && !S.chrom[a.chrom].skip; }

// new minimum score so that the number of kept alignments is one of 1,2,3,4,5,6,8,10 (2985, 3089)
void subsample(const std::vector<float>& scoresDesc, uint8_t& count, float& score) {
  count = count > 10 ? 10 : count - 1;
  score = scoresDesc[count - 1];
}

int processPair(State& S, const char* qname, std::vector<Aln>& aln, Counts& C, float score) {  // 3122-3176
  if (score != NOSCORE) score -= S.o.asDiff;
  auto ok = [&](const Aln& a) { return a.paired && a.full && a.score >= score && usable(S, a); };
  uint8_t count = 0;
  for (auto& a : aln) count += ok(a);
  if (!count) return 0;
  if (count > 10 || count == 7 || count == 9) {
    std::vector<float> sc;
    for (auto& a : aln)
      if (ok(a)) {  // stable insertion, descending
        size_t j = 0;
        while (j < sc.size() && !(a.score > sc[j])) j++;
        sc.insert(sc.begin() + j, a.score);
      }
    subsample(sc, count, score);
  }
  uint64_t fragLen = 0;
  uint8_t saved = 0;
  for (auto& a : aln)
    if (ok(a)) {
      fragLen += saveFragment(S, qname, a, count);
      if (++saved == count) break;  // in case of AS ties
    }
  if (t_sink) t_sink->lenTerms.push_back((double)fragLen / count);  // (added by the state's owner, in file order)
  else C.totalLen += (double)fragLen / count;
  return 1;
}

int processSingle(State& S, const char* qname, std::vector<Aln>& aln, std::vector<Unpair>& unpair, float score,
                  bool first) {  // 3019-3083
  if (score != NOSCORE) score -= S.o.asDiff;
  auto ok = [&](const Aln& a) { return !a.paired && a.first == first && a.score >= score && usable(S, a); };
  uint8_t count = 0;
  for (auto& a : aln) count += ok(a);
  if (!count) return 0;
  if (count > 10 || count == 7 || count == 9) {
    std::vector<float> sc;
    for (auto& a : aln)
      if (ok(a)) {
        size_t j = 0;
        while (j < sc.size() && !(a.score > sc[j])) j++;
        sc.insert(sc.begin() + j, a.score);
      }
    subsample(sc, count, score);
  }
  uint8_t saved = 0;
  for (auto& a : aln)
    if (ok(a)) {
      if (S.o.avgExtOpt)
        unpair.push_back(Unpair{a.chrom, {a.pos[0], a.pos[1]}, a.strand, count, qname});
      else
        saveUnpair(S, qname, a, count);
      if (++saved == count) break;
    }
  return 1;
}

// ---- -r: alignment sets kept until the end of the file (saveAlns 2942-2977) --------------------
struct DRead {
  std::string name;
  uint16_t qual = 0;            // sum of the base qualities (both mates for pairs / discordant pairs)
  bool first = false;           // singleton: which mate
  float score = NOSCORE, scoreR2 = NOSCORE;
  std::vector<Aln> aln, alnR2;  // alnR2: the R2 alignments of a discordant pair
};
struct DupReads { std::vector<DRead> pr, dc, sn; };

// copyAlns 2815-2851: the unpaired alignments of one mate within the score window
void copyAlns(const State& S, const std::vector<Aln>& aln, float score, bool first, std::vector<Aln>& dest) {
  if (score != NOSCORE) score -= S.o.asDiff;
  for (auto& a : aln)
    if (!a.paired && a.first == first && a.score >= score) dest.push_back(a);
}

void saveAlns(const State& S, const char* qname, const std::vector<Aln>& aln, bool pair, bool singleR1, bool singleR2,
              float scorePr, float scoreR1, float scoreR2, uint16_t qualR1, uint16_t qualR2, DupReads& D) {
  auto sumq = [](uint16_t a, uint16_t b) { return (uint16_t)std::min<int>((int)a + (int)b, UINT16_MAX); };
  if (pair) {  // saveAlnsPair 2890-2935: positions ordered
    DRead r;
    r.name = qname;
    r.qual = sumq(qualR1, qualR2);
    r.score = scorePr;
    float score = scorePr;
    if (score != NOSCORE) score -= S.o.asDiff;
    for (auto& a : aln)
      if (a.paired && a.full && a.score >= score) {
        Aln b = a;
        if (b.pos[0] > b.pos[1]) std::swap(b.pos[0], b.pos[1]);
        r.aln.push_back(b);
      }
    D.pr.push_back(std::move(r));
  } else if (S.o.singleOpt) {
    if (singleR1 && singleR2) {  // both mates aligned, not as a proper pair (saveAlnsDiscord 2874)
      DRead r;
      r.name = qname;
      r.first = true;
      r.score = scoreR1;
      r.scoreR2 = scoreR2;
      copyAlns(S, aln, scoreR1, true, r.aln);
      copyAlns(S, aln, scoreR2, false, r.alnR2);
      r.qual = sumq(qualR1, qualR2);
      D.dc.push_back(std::move(r));
    } else if (singleR1 || singleR2) {  // saveAlnsSingle 2856
      DRead r;
      r.name = qname;
      r.first = singleR1;
      r.score = singleR1 ? scoreR1 : scoreR2;
      r.qual = singleR1 ? qualR1 : qualR2;
      copyAlns(S, aln, r.score, r.first, r.aln);
      D.sn.push_back(std::move(r));
    }
  }
}

void processAlns(State& S, const char* qname, std::vector<Aln>& aln, std::vector<Unpair>& unpair, Counts& C,
                 uint16_t qualR1, uint16_t qualR2, DupReads& D) {  // 3187-3265
  float scorePr = NOSCORE, scoreR1 = NOSCORE, scoreR2 = NOSCORE;
  bool pair = false, singleR1 = false, singleR2 = false;
  for (auto& a : aln) {
    if (a.paired) {
      if (a.full) {
        if (!pair || scorePr < a.score) scorePr = a.score;
        pair = true;
      } else
        C.orphan++;
    } else if (S.o.singleOpt && !pair) {
      if (a.first && scoreR1 <= a.score) { scoreR1 = a.score; singleR1 = true; }
      else if (!a.first && scoreR2 <= a.score) { scoreR2 = a.score; singleR2 = true; }
    }
  }
  if (S.o.dupsOpt)
    saveAlns(S, qname, aln, pair, singleR1, singleR2, scorePr, scoreR1, scoreR2, qualR1, qualR2, D);
  else if (pair)
    C.pairedPr += processPair(S, qname, aln, C, scorePr);
  else if (S.o.singleOpt) {
    if (singleR1) C.singlePr += processSingle(S, qname, aln, unpair, scoreR1, true);
    if (singleR2) C.singlePr += processSingle(S, qname, aln, unpair, scoreR2, false);
  }
}

// ---- -r: PCR duplicates (findDups 3949-4042) ---------------------------------------------------
// Sets are visited from the highest base-quality sum down (a stable order, sortReads 3362); a set
// is a duplicate when any of its alignments matches one already kept: proper pairs by (chrom,
// both 5' ends), discordant pairs by both ends with their strands in either order, singletons by
// (chrom, 5' end, strand) -- against kept singletons and against both ends of every kept pair.
// Only membership matters, so ordered maps stand in for the reference's chained hash tables.
// (hash tables, as the reference's: only "is this key there, and who put it there first" is ever asked -- an ordered
// map of 10^7 keys made findDups the slowest part of a -r run)
struct KeyPr { int chrom; uint32_t p0, p1; bool operator==(const KeyPr& o) const { return chrom == o.chrom && p0 == o.p0 && p1 == o.p1; } };
struct KeySn { int chrom; uint32_t pos; bool strand; bool operator==(const KeySn& o) const { return chrom == o.chrom && pos == o.pos && strand == o.strand; } };
struct KeyDc {
  int c0, c1; uint32_t p0, p1; bool s0, s1;
  bool operator==(const KeyDc& o) const { return c0 == o.c0 && c1 == o.c1 && p0 == o.p0 && p1 == o.p1 && s0 == o.s0 && s1 == o.s1; }
};
inline uint64_t mix64(uint64_t x) {  // (splitmix64's finaliser)
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
struct KeyHash {
  size_t operator()(const KeyPr& k) const { return (size_t)mix64(((uint64_t)k.p0 << 32 | k.p1) ^ mix64((uint64_t)(uint32_t)k.chrom)); }
  size_t operator()(const KeySn& k) const { return (size_t)mix64(((uint64_t)k.pos << 1 | (uint64_t)k.strand) ^ mix64((uint64_t)(uint32_t)k.chrom + 0x9e3779b97f4a7c15ull)); }
  size_t operator()(const KeyDc& k) const {
    return (size_t)mix64(((uint64_t)k.p0 << 32 | k.p1) ^ mix64(((uint64_t)(uint32_t)k.c0 << 32 | (uint32_t)k.c1) ^ ((uint64_t)k.s0 << 1 | (uint64_t)k.s1)));
  }
};

void dupLine(State& S, const char* fmt, ...) {
  char buf[2 * MAX_ALNS + 4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  fputs(buf, S.dups.f);
}

// Who held a key first -- the membership half of the tables -- comes from the device when there is one
// (gx_dups_first, gx_dups.h): every alignment of the file's sets at once instead of one hash probe after the other.
// What it cannot decide (sets with several alignments give up all their keys when one matches; whatever shares a key
// with such a set is "contested") is walked here in the reference's order, on tables that hold the contested keys only.
struct DupDevice {
  bool on = false;
  std::vector<gx_dup_key> keys;
  std::vector<uint8_t> multi;
  std::vector<uint32_t> owner, readOf;   // per record: gx_dups_first's word; the set (or addition) it belongs to
  uint64_t nKeys = 0, nContested = 0, nByTable[4] = {0, 0, 0, 0};   // (by the key's tag: 1 proper pairs, 2 discordant, 3 singletons)
  void clear() { keys.clear(); multi.clear(); owner.clear(); readOf.clear(); }
  void push(uint32_t tag, uint32_t a, uint32_t b, uint32_t c, bool m, uint32_t read) {
    keys.push_back(gx_dup_key{{tag, a, b, c}});
    multi.push_back(m ? 1 : 0);
    readOf.push_back(read);
  }
};
constexpr uint32_t DUP_CONTESTED = 0x80000000u;

void findDups(State& S, DupReads& D, Counts& C) {
  const bool verb = S.dupsVerb;
  std::unordered_map<KeySn, std::string, KeyHash> tabSn;
  const bool useSn = S.o.singleOpt && !D.sn.empty();  // the singleton table exists only when there are singletons
  DupDevice dev;
  dev.on = S.gx && !S.devs.ctx.empty() && !getenv("GENRICH_DUPS_HOST");
  // device mode: what checkAndAdd (3514) puts into the singleton table ahead of the singletons -- both ends of every
  // kept pair -- is a list of records that lead the singletons' own (the table's first holder of a key names the read)
  struct SnAdd { int chrom; uint32_t pos; bool strand; std::string name; };
  std::vector<SnAdd> snAdds;
  auto addSn = [&](int chrom, uint32_t pos, bool strand, const std::string& name) {  // checkAndAdd 3514
    if (dev.on) snAdds.push_back(SnAdd{chrom, pos, strand, verb ? name : std::string()});
    else tabSn.emplace(KeySn{chrom, pos, strand}, verb ? name : std::string());
  };
  auto order = [](const std::vector<DRead>& v) {
    std::vector<uint32_t> o(v.size());
    for (uint32_t i = 0; i < o.size(); i++) o[i] = i;
    std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return v[a].qual > v[b].qual; });
    return o;
  };
  auto runDevice = [&]() {
    dev.owner.assign(dev.keys.size(), 0);
    if (!dev.keys.empty())
      check(S, gx_dups_first(S.devs.ctx[0], dev.keys.data(), dev.multi.data(), dev.keys.size(), dev.owner.data()), S.devs.ctx[0]);
    dev.nKeys += dev.keys.size();
    for (const gx_dup_key& k : dev.keys) dev.nByTable[k.w[0] & 3u]++;   // (per key: a batch need not hold one table's keys)
    for (uint32_t w : dev.owner) dev.nContested += (w & DUP_CONTESTED) != 0;
  };

  {  // properly paired sets (findDupsPr 3616)
    std::unordered_map<KeyPr, std::string, KeyHash> tab;   // (device mode: the contested keys only)
    const std::vector<uint32_t> ord = order(D.pr);
    std::vector<uint32_t> firstRec;
    if (dev.on) {
      firstRec.resize(ord.size() + 1);
      for (size_t q = 0; q < ord.size(); q++) {
        const DRead& r = D.pr[ord[q]];
        firstRec[q] = (uint32_t)dev.keys.size();
        for (auto& a : r.aln) dev.push(1u, (uint32_t)a.chrom, a.pos[0], a.pos[1], r.aln.size() > 1, ord[q]);
      }
      firstRec[ord.size()] = (uint32_t)dev.keys.size();
      runDevice();
    }
    for (size_t q = 0; q < ord.size(); q++) {
      DRead& r = D.pr[ord[q]];
      bool dup = false;
      const bool byDevice = dev.on && r.aln.size() == 1 && !(dev.owner[firstRec[q]] & DUP_CONTESTED);
      if (byDevice) {
        const uint32_t me = firstRec[q], own = dev.owner[me];
        if (own != me) {
          const Aln& a = r.aln[0];
          if (verb) dupLine(S, "%s\t%s:%d-%d\t%s\tpaired\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), a.pos[0], a.pos[1],
                            D.pr[dev.readOf[own]].name.c_str());
          dup = true;
        }
      } else
      for (auto& a : r.aln) {
        auto it = tab.find(KeyPr{a.chrom, a.pos[0], a.pos[1]});
        if (it != tab.end()) {
          if (verb) dupLine(S, "%s\t%s:%d-%d\t%s\tpaired\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), a.pos[0], a.pos[1], it->second.c_str());
          dup = true;
          break;
        }
      }
      if (dup) C.dupsPr++;
      else {
        for (auto& a : r.aln) {
          if (!byDevice) tab.emplace(KeyPr{a.chrom, a.pos[0], a.pos[1]}, verb ? r.name : std::string());
          if (useSn) {
            addSn(a.chrom, a.pos[0], true, r.name);
            addSn(a.chrom, a.pos[1], false, r.name);
          }
        }
        C.pairedPr += processPair(S, r.name.c_str(), r.aln, C, r.score);
      }
      C.countPr++;
    }
    D.pr.clear();
    D.pr.shrink_to_fit();
    dev.clear();
  }
  if (!S.o.singleOpt) {
    if (dev.on && getenv("GENRICH_DUPS_REPORT"))
      fprintf(stderr, "[dups] device: %llu keys, %llu contested (resolved on the host); by table: %llu paired, %llu discordant, %llu single\n",
            (unsigned long long)dev.nKeys, (unsigned long long)dev.nContested, (unsigned long long)dev.nByTable[1],
            (unsigned long long)dev.nByTable[2], (unsigned long long)dev.nByTable[3]);
    return;
  }

  // with -x the average fragment length of the kept pairs becomes the extension (3990-3996)
  const Opts saved = S.o;
  if (S.o.avgExtOpt) {
    int extend = 0;
    if (!C.pairedPr) {
      if (S.o.verbose) {
        fprintf(stderr, "Warning! No paired alignments to calculate avg frag ");
        fprintf(stderr, "length --\n  Printing unpaired alignments \"as is\"\n");
      }
    } else
      extend = (int)(C.totalLen / C.pairedPr + 0.5);
    S.o.extend = extend;
    if (extend) S.o.extendOpt = true;
    S.o.avgExtOpt = false;
  }
  std::vector<Unpair> none;

  {  // discordant sets (findDupsDc 3761): every R1 x R2 combination, either order
    std::unordered_map<KeyDc, std::string, KeyHash> tab;   // (device mode: the contested keys only)
    auto end5 = [](const Aln& a) { return a.strand ? a.pos[0] : a.pos[1]; };
    const std::vector<uint32_t> ord = order(D.dc);
    // Device mode: the reference looks a combination up in both orders and stores it in one (3786-3837), i.e. the table is
    // keyed on the UNORDERED pair of ends -- one record per combination with the two ends in a canonical order.  The
    // chromosomes share a word: genomes of more than 65,536 sequences keep the host's tables.
    const bool dcDev = dev.on && S.chrom.size() <= 65536;
    if (dev.on && !dcDev && !D.dc.empty() && getenv("GENRICH_DUPS_REPORT"))
      fprintf(stderr, "[dups] discordant sets: the host's tables (more than 65,536 sequences: two chromosome indices do not fit a key word)\n");
    std::vector<uint32_t> firstRec;
    auto canon = [&](const Aln& a, const Aln& b, uint32_t w[4]) {
      const uint32_t pa = end5(a), pb = end5(b);
      const bool swap = std::make_tuple(b.chrom, pb, (int)b.strand) < std::make_tuple(a.chrom, pa, (int)a.strand);
      const Aln &x = swap ? b : a, &y = swap ? a : b;
      w[0] = 2u | ((uint32_t)x.strand << 8) | ((uint32_t)y.strand << 9);
      w[1] = (uint32_t)x.chrom | ((uint32_t)y.chrom << 16);
      w[2] = swap ? pb : pa;
      w[3] = swap ? pa : pb;
    };
    if (dcDev) {
      firstRec.resize(ord.size() + 1);
      for (size_t q = 0; q < ord.size(); q++) {
        const DRead& r = D.dc[ord[q]];
        firstRec[q] = (uint32_t)dev.keys.size();
        const bool multi = r.aln.size() * r.alnR2.size() > 1;
        for (auto& a : r.aln)
          for (auto& b : r.alnR2) {
            uint32_t w[4];
            canon(a, b, w);
            dev.keys.push_back(gx_dup_key{{w[0], w[1], w[2], w[3]}});
            dev.multi.push_back(multi ? 1 : 0);
            dev.readOf.push_back(ord[q]);
          }
      }
      firstRec[ord.size()] = (uint32_t)dev.keys.size();
      runDevice();
    }
    for (size_t q = 0; q < ord.size(); q++) {
      DRead& r = D.dc[ord[q]];
      bool dup = false;
      const bool byDevice = dcDev && r.aln.size() == 1 && r.alnR2.size() == 1 && !(dev.owner[firstRec[q]] & DUP_CONTESTED);
      if (byDevice) {
        const uint32_t me = firstRec[q], own = dev.owner[me];
        if (own != me) {
          // (an uncontested key: its first holder is a set of one combination too, stored as (its R1, its R2); the
          // reference finds it in the order of this set's ends first, in the swapped order otherwise -- 3786 / 3812)
          const DRead& hr = D.dc[dev.readOf[own]];
          const Aln &a = r.aln[0], &b = r.alnR2[0], &ha = hr.aln[0], &hb = hr.alnR2[0];
          const uint32_t pos = end5(a), pos1 = end5(b);
          const bool sameOrder = ha.chrom == a.chrom && hb.chrom == b.chrom && end5(ha) == pos && end5(hb) == pos1 &&
                                 ha.strand == a.strand && hb.strand == b.strand;
          if (verb) {
            if (sameOrder)
              dupLine(S, "%s\t%s:%d,%c;%s:%d,%c\t%s\tdiscordant\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), pos,
                      a.strand ? '+' : '-', S.chrom[b.chrom].name.c_str(), pos1, b.strand ? '+' : '-', hr.name.c_str());
            else
              dupLine(S, "%s\t%s:%d,%c;%s:%d,%c\t%s\tdiscordant\n", r.name.c_str(), S.chrom[b.chrom].name.c_str(), pos1,
                      b.strand ? '+' : '-', S.chrom[a.chrom].name.c_str(), pos, a.strand ? '+' : '-', hr.name.c_str());
          }
          dup = true;
        }
      } else
      for (size_t k = 0; k < r.aln.size() && !dup; k++) {
        const Aln& a = r.aln[k];
        const uint32_t pos = end5(a);
        for (size_t j = 0; j < r.alnR2.size() && !dup; j++) {
          const Aln& b = r.alnR2[j];
          const uint32_t pos1 = end5(b);
          auto it = tab.find(KeyDc{a.chrom, b.chrom, pos, pos1, a.strand, b.strand});
          if (it != tab.end()) {
            if (verb) dupLine(S, "%s\t%s:%d,%c;%s:%d,%c\t%s\tdiscordant\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), pos,
                              a.strand ? '+' : '-', S.chrom[b.chrom].name.c_str(), pos1, b.strand ? '+' : '-', it->second.c_str());
            dup = true;
            break;
          }
          it = tab.find(KeyDc{b.chrom, a.chrom, pos1, pos, b.strand, a.strand});
          if (it != tab.end()) {
            if (verb) dupLine(S, "%s\t%s:%d,%c;%s:%d,%c\t%s\tdiscordant\n", r.name.c_str(), S.chrom[b.chrom].name.c_str(), pos1,
                              b.strand ? '+' : '-', S.chrom[a.chrom].name.c_str(), pos, a.strand ? '+' : '-', it->second.c_str());
            dup = true;
          }
        }
      }
      if (dup) C.dupsDc++;
      else {
        for (size_t k = 0; k < r.aln.size(); k++)
          for (size_t j = 0; j < r.alnR2.size(); j++) {
            const Aln &a = r.aln[k], &b = r.alnR2[j];
            if (!byDevice) tab.emplace(KeyDc{a.chrom, b.chrom, end5(a), end5(b), a.strand, b.strand}, verb ? r.name : std::string());
            if (useSn) {
              if (!j) addSn(a.chrom, end5(a), a.strand, r.name);
              if (!k) addSn(b.chrom, end5(b), b.strand, r.name);
            }
          }
        C.singlePr += processSingle(S, r.name.c_str(), r.aln, none, r.score, true);
        C.singlePr += processSingle(S, r.name.c_str(), r.alnR2, none, r.scoreR2, false);
      }
      C.countDc++;
    }
    D.dc.clear();
    D.dc.shrink_to_fit();
    dev.clear();
  }

  {  // singletons (findDupsSn 3886): the table already holds the ends of the kept pairs
    auto end5 = [](const Aln& a) { return a.strand ? a.pos[0] : a.pos[1]; };
    const std::vector<uint32_t> ord = order(D.sn);
    std::vector<uint32_t> firstRec;
    const uint32_t nAdds = (uint32_t)snAdds.size();
    if (dev.on) {
      // the additions first (never duplicates themselves: they only take a key if it is free), then the singletons
      for (uint32_t k = 0; k < nAdds; k++) dev.push(3u, (uint32_t)snAdds[k].chrom, snAdds[k].pos, snAdds[k].strand, false, k);
      firstRec.resize(ord.size() + 1);
      for (size_t q = 0; q < ord.size(); q++) {
        const DRead& r = D.sn[ord[q]];
        firstRec[q] = (uint32_t)dev.keys.size();
        for (auto& a : r.aln) dev.push(3u, (uint32_t)a.chrom, end5(a), a.strand, r.aln.size() > 1, nAdds + ord[q]);
      }
      firstRec[ord.size()] = (uint32_t)dev.keys.size();
      runDevice();
      // the contested keys among the additions enter the host's table, in their order (the first holder stays)
      for (uint32_t k = 0; k < nAdds; k++)
        if (dev.owner[k] & DUP_CONTESTED) tabSn.emplace(KeySn{snAdds[k].chrom, snAdds[k].pos, snAdds[k].strand}, snAdds[k].name);
    }
    auto holder = [&](uint32_t rec) -> const std::string& {  // the read that put record rec's key there
      const uint32_t id = dev.readOf[rec];
      return id < nAdds ? snAdds[id].name : D.sn[id - nAdds].name;
    };
    for (size_t q = 0; q < ord.size(); q++) {
      DRead& r = D.sn[ord[q]];
      bool dup = false;
      const bool byDevice = dev.on && r.aln.size() == 1 && !(dev.owner[firstRec[q]] & DUP_CONTESTED);
      if (byDevice) {
        const uint32_t me = firstRec[q], own = dev.owner[me];
        if (own != me) {
          const Aln& a = r.aln[0];
          if (verb) dupLine(S, "%s\t%s:%d,%c\t%s\tsingle\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), end5(a),
                            a.strand ? '+' : '-', holder(own).c_str());
          dup = true;
        }
      } else
      for (auto& a : r.aln) {
        auto it = tabSn.find(KeySn{a.chrom, end5(a), a.strand});
        if (it != tabSn.end()) {
          if (verb) dupLine(S, "%s\t%s:%d,%c\t%s\tsingle\n", r.name.c_str(), S.chrom[a.chrom].name.c_str(), end5(a),
                            a.strand ? '+' : '-', it->second.c_str());
          dup = true;
          break;
        }
      }
      if (dup) C.dupsSn++;
      else {
        if (!byDevice)
          for (auto& a : r.aln) tabSn.emplace(KeySn{a.chrom, end5(a), a.strand}, verb ? r.name : std::string());
        C.singlePr += processSingle(S, r.name.c_str(), r.aln, none, r.score, r.first);
      }
      C.countSn++;
    }
    D.sn.clear();
    D.sn.shrink_to_fit();
    dev.clear();
  }
  if (dev.on && getenv("GENRICH_DUPS_REPORT"))
    fprintf(stderr, "[dups] device: %llu keys, %llu contested (resolved on the host); by table: %llu paired, %llu discordant, %llu single\n",
            (unsigned long long)dev.nKeys, (unsigned long long)dev.nContested, (unsigned long long)dev.nByTable[1],
            (unsigned long long)dev.nByTable[2], (unsigned long long)dev.nByTable[3]);
  S.o = saved;
}

// parseAlign (4141-4212); returns false when the per-read alignment limit is hit
bool parseAlign(State& S, std::vector<Aln>& aln, uint16_t flag, int ci, uint32_t pos, int length, uint32_t pnext,
                Counts& C, float score) {
  if (flag & 0x1) {
    if ((flag & 0xC0) == 0xC0) die("", "Linear template with >2 reads -- not allowed");
    if (!(flag & 0xC0)) die("", "Unknown index of paired alignment");
  }
  const Chrom& ch = S.chrom[ci];
  const uint32_t end5 = (flag & 0x10) ? pos + length : pos;  // 5' end of a reverse read = pos + refLen
  if ((flag & 0x3) == 0x3) {
    if (ch.skip || !ch.save) C.skipped++;
    else {
      C.paired++;
      if (flag & 0x100) C.secPair++;
    }
    for (auto& a : aln)
      if (a.paired && !a.full && a.chrom == ci &&
          ((flag & 0x40) ? (!a.first && a.pos[0] == pos) : (a.first && a.pos[1] == pos)) &&
          ((flag & 0x100) ? !a.primary : a.primary)) {
        if (flag & 0x40) a.pos[0] = end5; else a.pos[1] = end5;  // updatePairedAln 4049
        if (score == NOSCORE) a.score = NOSCORE;
        else if (a.score != NOSCORE) a.score += score;
        a.full = true;
        return true;
      }
    if (aln.size() == MAX_ALNS) return false;
    Aln a{};
    a.chrom = ci;
    a.score = score;
    a.primary = !(flag & 0x100);
    a.full = false;
    a.paired = true;
    if (flag & 0x40) { a.pos[0] = end5; a.pos[1] = pnext; a.first = true; }
    else { a.pos[0] = pnext; a.pos[1] = end5; a.first = false; }
    aln.push_back(a);
    return true;
  }
  if (ch.skip || !ch.save) C.skipped++;
  else {
    C.single++;
    if (flag & 0x100) C.secSingle++;
  }
  if (S.o.singleOpt) {
    if (aln.size() == MAX_ALNS) return false;
    Aln a{};
    a.chrom = ci;
    a.score = score;
    a.primary = !(flag & 0x100);
    a.paired = false;
    a.strand = !(flag & 0x10);
    a.first = flag & 0x40;
    a.pos[0] = pos;
    a.pos[1] = pos + length;
    aln.push_back(a);
  }
  return true;
}

// ---- input: one gz-transparent byte stream (plain, gzip or BGZF) ------------------------------
// (BGZF members are inflated by a pool of threads, bgzf_reader.h; everything else through zlib's gz*)
typedef gxhost::Input In;
int g_threads = 1;
void openRead(In& in, const char* path) {
  if (!in.open(path, g_threads)) die(path, ": cannot open file for reading");
}
void checkIn(In& in) {
  if (!in.error().empty()) die(in.name(), (": " + in.error()).c_str());
}

// distance to the 3' end from a SAM CIGAR (parseCigar 4408, calcDist 4451)
int calcDist(const char* qname, const char* seq, const char* cigar) {
  int length = strcmp(seq, "*") ? (int)strlen(seq) : 0;
  int offset = 0;
  if (strcmp(cigar, "*")) {
    int len = 0;
    const char* p = cigar;
    while (*p) {
      char* e;
      long n = strtol(p, &e, 10);
      if (e == p && (*p < '0' || *p > '9')) n = 0;  // an op without a number counts as 0
      char op = *e;
      if (!op) break;
      switch (op) {
        case 'M': case '=': case 'X': len += (int)n; break;
        case 'I': case 'S': len += (int)n; offset -= (int)n; break;
        case 'D': offset += (int)n; break;
        case 'N': case 'H': case 'P': break;
        default: {
          char msg[4] = "' '";
          msg[1] = op;
          die(msg, ": unknown Op in CIGAR");
        }
      }
      p = e + 1;
    }
    if (!length) length = len;
    else if (length != len) die(qname, ": mismatch between sequence length and CIGAR");
  } else if (!length)
    die(qname, ": no sequence information (SEQ or CIGAR)");
  return length + offset;
}

float samScore(char* extra) {  // getScore 4383-4402
  // fields on TAB; inside a field the tokens between colons (strtok: a run of colons is one separator).
  // The first field whose first token is "AS" decides: its third token is the score, whatever the second
  // (the type) says; without a third token there is no score.
  if (!extra) return NOSCORE;
  char* endF = nullptr;
  for (char* f = strtok_r(extra, "\t", &endF); f; f = strtok_r(nullptr, "\t", &endF)) {
    char* endT = nullptr;
    char* tag = strtok_r(f, ":", &endT);
    if (tag && !strcmp(tag, "AS")) {
      if (!strtok_r(nullptr, ":", &endT)) return NOSCORE;
      char* v = strtok_r(nullptr, ":", &endT);
      return v ? getFloat(v) : NOSCORE;
    }
  }
  return NOSCORE;
}

void headerLine(State& S, char* line) {  // checkHeader 4307-4342, loadChrom 4275-4299
  // as the reference cuts it up: strtok on TAB, so the last token of the line still carries its '\n' -- a
  // bare "@HD\n" or "@SQ\n" is therefore not recognised -- and only the values of SO: / SN: / LN: are cut at
  // the line feed (a carriage return stays)
  auto cut = [](char* v) {
    if (v) v[strcspn(v, "\n")] = '\0';
  };
  char* tag = strtok(line, "\t");
  if (!tag) return;
  if (!strcmp(tag, "@HD")) {
    char* order = nullptr;
    for (char* f = strtok(nullptr, "\t"); f; f = strtok(nullptr, "\t"))
      if (!strncmp(f, "SO:", 3)) order = f + 3;
    cut(order);
    if (S.o.sortOpt && (!order || strcmp(order, "queryname")))
      die("", "SAM/BAM file not sorted by queryname (samtools sort -n)");
  } else if (!strcmp(tag, "@SQ")) {
    char *name = nullptr, *len = nullptr;
    for (char* f = strtok(nullptr, "\t"); f; f = strtok(nullptr, "\t")) {
      if (!strncmp(f, "SN:", 3)) name = f + 3;
      else if (!strncmp(f, "LN:", 3)) len = f + 3;
    }
    if (!name || !len) return;
    cut(name);
    cut(len);
    saveChrom(S, name, (uint32_t)getInt(len));
  }
}

// the header of the current file is complete: open the sample on the device
void openSample(State& S) {
  if (S.sampleOpen) return;
  S.sampleOpen = true;
  if (!S.gx) {
    if (S.hot.on) S.hot.init(S.chrom);
    return;
  }
  std::vector<uint8_t> save(S.chrom.size());
  for (size_t k = 0; k < S.chrom.size(); k++) save[k] = S.chrom[k].save;
  for (gx_ctx* g : S.devs.ctx) check(S, gx_sample_begin(g, S.ctrl ? 1 : 0, S.ctrl ? nullptr : save.data()), g);
  if (S.hot.on) S.hot.init(S.chrom);  // (the reference's difference arrays start at zero with every file)
}

struct ReadSet {
  std::string name;
  std::vector<Aln> aln;
  std::vector<Unpair> unpair;
  bool have = false;
  uint16_t qualR1 = 0, qualR2 = 0;  // -r: base-quality sums of the two mates of the current read
  DupReads dup;                     // -r: every alignment set of the file
};

void flushSet(State& S, ReadSet& rs, Counts& C) {
  if (rs.have) processAlns(S, rs.name.c_str(), rs.aln, rs.unpair, C, rs.qualR1, rs.qualR2, rs.dup);
  rs.aln.clear();
  rs.qualR1 = rs.qualR2 = 0;
}

// sumQual 4127-4134, bug for bug: `qual[0] == 0xFF` compares a (signed) char with 255 and never
// holds, so a BAM record without qualities (all 0xFF = -1) sums to -len and wraps
uint16_t sumQual(const char* qual, int len, int offset) {
  int sum = 0;
  for (int i = 0; i < len; i++) sum += (int)(signed char)qual[i] - offset;
  return sum > UINT16_MAX ? UINT16_MAX : (uint16_t)sum;
}

// one alignment record, format independent (the tail of readSAM's / parseBAM's loop)
void record(State& S, ReadSet& rs, Counts& C, const char* qname, uint16_t flag, int ci, uint32_t pos, uint8_t mapq,
            int length, uint32_t pnext, float score, const char* qual, int qualLen, int qualOffset) {
  if (mapq < S.o.minMapQ) { C.lowMapQ++; return; }
  if (!t_sink) openSample(S);  // the header is complete once the first record arrives (workers: the state's owner does it)
  if (!rs.have || rs.name != qname) {
    flushSet(S, rs, C);
    rs.have = true;
    rs.name.assign(qname, strnlen(qname, MAX_ALNS));  // strncpy(readName, qname, MAX_ALNS)
  }
  if (S.o.dupsOpt && !(((flag & 0x1) && ((flag & 0xC0) == 0xC0 || !(flag & 0xC0))))) {  // parseAlign 4155-4164
    uint16_t& q = (flag & 0x40) ? rs.qualR1 : rs.qualR2;
    const bool star = qualLen >= 1 && qual[0] == '*' && (qualLen == 1 || qual[1] == '\0');  // strcmp(qual, "*")
    if (!q && !star) q = sumQual(qual, qualLen, qualOffset);
  }
  if (!parseAlign(S, rs.aln, flag, ci, pos, length, pnext, C, score) && S.o.verbose)
    warnPlain("Warning! Read %s has more than %d alignments\n", qname, MAX_ALNS);
}

void finishFile(State& S, ReadSet& rs, Counts& C) {  // the tail of readSAM / parseBAM
  flushSet(S, rs, C);
  if (S.o.dupsOpt) {
    findDups(S, rs.dup, C);
    return;
  }
  if (S.o.avgExtOpt) {  // processAvgExt 2614-2647
    int avgLen = 0;
    if (!C.pairedPr) {
      if (S.o.verbose) {
        fprintf(stderr, "Warning! No paired alignments to calculate avg frag ");
        fprintf(stderr, "length --\n  Printing unpaired alignments \"as is\"\n");
      }
    } else
      avgLen = (int)(C.totalLen / C.pairedPr + 0.5);
    for (auto& u : rs.unpair) {
      if (!avgLen) saveInterval(S, u.chrom, u.pos[0], u.pos[1], u.name.c_str(), u.count);
      else if (u.strand) saveInterval(S, u.chrom, u.pos[0], (uint32_t)(u.pos[0] + avgLen), u.name.c_str(), u.count);
      else saveInterval(S, u.chrom, (int)(u.pos[1] - avgLen), u.pos[1], u.name.c_str(), u.count);
    }
    rs.unpair.clear();
  }
}

// ---- record decoding, apart from the run's state ---------------------------------------------
// What readSAM's / readBAM's loop knows about a record BEFORE it touches the run's state -- the fields cut up and
// converted, the reference sequence looked up, the distance to the 3' end, the alignment score -- depends on nothing
// but the record (and the header, which is complete by then).  That part, three quarters of the parsing thread's
// time on bowtie2-style SAM text, runs on `--threads` decoder threads over batches of records; the rest -- counters,
// read-name groups, pairing, weights, -b lines, events: everything whose order matters -- stays on the one thread
// that owns the state and takes the decoded batches in file order.  An error found while decoding is kept with its
// record and raised when its turn comes, so warnings and errors appear in the order of a sequential run.
struct Decoded {
  enum Kind : uint8_t { REC, UNMAPPED, SUPP, LOWQ, FAIL };
  uint8_t kind = REC, mapq = 0;
  uint16_t flag = 0;
  int ci = 0, length = 0, qualLen = 0;
  uint32_t pos = 0, pnext = 0;
  float score = 0.0f;
  uint32_t qname = 0, qual = 0;   // offsets from the record's first byte
  int fail = -1;                   // FAIL: index into the batch's messages
};

constexpr size_t REC_MAX = 65520;  // (a line is read in pieces of at most 65,520 bytes: readSAM's buffer)
// bytes of records per batch (GENRICH_BATCH_BYTES: tests cut the input into batches of a line or two)
const size_t BATCH_BYTES = getenv("GENRICH_BATCH_BYTES") ? (size_t)std::max(1L, atol(getenv("GENRICH_BATCH_BYTES"))) : (size_t)1 << 20;

struct Batch {
  std::unique_ptr<char[]> bytes;   // the records' text lines (NUL-terminated) or BAM blocks
  size_t used = 0, cap = 0;
  std::vector<uint32_t> off, len;  // one entry per record
  std::vector<Decoded> rec;
  std::vector<std::pair<std::string, std::string>> fails;
  // What the thread that cuts the stream into chunks needs to know about a record, one byte each (made by the decoder
  // that has the records in its cache: the cutting thread reads 64 records per cache line instead of two lines per
  // record): [2:0] the record's kind, MARK_SAME when it has the name of the group the REC record before it IN THIS
  // BATCH belongs to (record()'s comparison with its 128-character quirk); the first REC record of a batch is compared
  // by the cutting thread itself, with the name it carries over (lastName: this batch's contribution to that)
  static constexpr uint8_t MARK_SAME = 0x80;
  std::vector<uint8_t> mark;
  uint32_t firstRec = 0xFFFFFFFFu;  // index of the batch's first REC record
  std::string lastName;             // the name of the group the batch's last REC record belongs to, as record() keeps it
  bool decoded = false;
  // a batch of text that the READER only delimited (a span of a mapped file that ends behind a line feed): the
  // decoder that takes the batch cuts it into lines itself (cutSpan) -- the reader's part is then a memchr per batch
  const char* span = nullptr;
  size_t spanLen = 0;
  // records that stay where they are -- BAM blocks inside an inflated BGZF member, which `holds` keeps alive: off[i] has
  // its top bit set and indexes `ext`
  std::vector<const char*> ext;
  std::vector<std::shared_ptr<const void>> holds;
  size_t extBytes = 0;
  static constexpr uint32_t EXT = 0x80000000u;
  const char* at(size_t i) const { return (off[i] & EXT) ? ext[off[i] & ~EXT] : bytes.get() + off[i]; }
  void addExt(const void* p, size_t n) {
    off.push_back(EXT | (uint32_t)ext.size());
    len.push_back((uint32_t)n);
    ext.push_back(static_cast<const char*>(p));
    extBytes += n;
  }
  bool hasWork() const { return !off.empty() || spanLen; }
  void reset(size_t want) {  // (the reader's thread, before the batch is queued)
    decoded = false;
    clearRecords(want);
  }
  // ... and what a decoder may do to a queued batch: `decoded` belongs to the pipe's mutex (the taker polls it)
  void clearRecords(size_t want) {
    if (cap < want) {
      bytes.reset(new char[want]);
      cap = want;
    }
    used = 0;
    off.clear(); len.clear(); rec.clear(); fails.clear();
    mark.clear();
    firstRec = 0xFFFFFFFFu;
    lastName.clear();
    span = nullptr;
    spanLen = 0;
    ext.clear();
    holds.clear();
    extBytes = 0;
  }
  char* room(size_t n) {  // n more bytes behind the records so far (the batch grows when a record needs it)
    if (used + n > cap) {
      const size_t want = std::max(cap * 2, used + n);
      std::unique_ptr<char[]> nb(new char[want]);
      memcpy(nb.get(), bytes.get(), used);
      bytes = std::move(nb);
      cap = want;
    }
    return bytes.get() + used;
  }
  void add(size_t n) {
    off.push_back((uint32_t)used);
    len.push_back((uint32_t)n);
    used += n;
  }
};

struct ChromIndex {  // findChrom without the linear search (first of equal names, as the search finds it)
  std::unordered_map<std::string, int> at;
  explicit ChromIndex(const State& S) {
    for (size_t i = 0; i < S.chrom.size(); i++) at.emplace(S.chrom[i].name, (int)i);
  }
  int find(const char* rname) const {
    auto it = at.find(rname);
    if (it == at.end()) die(rname, ": cannot find reference sequence name in SAM header");
    return it->second;
  }
};

// one SAM line (readSAM 4524-4560 up to the call of parseAlign)
void decodeSamLine(const Opts& o, const ChromIndex& cx, char* l, Decoded& r) {
  char* const base = l;
  if (l[0] == '@') die(l, ": misplaced SAM header line");
  // QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL [extra], cut up as the reference does
  // (readSAM 4524-4530, loadFields 4350-4377): strtok on TAB -- so a run of tabs is one separator, and
  // the last field of a line keeps its line end -- with the integer fields converted (getInt: the whole
  // token must be a number) as they are met, QUAL included in that order of events.
  char* save = nullptr;
  char* qname = strtok_r(l, "\t", &save);
  if (!qname) die(l, ": poorly formatted SAM/BAM record");
  char* fld[12] = {nullptr};
  fld[0] = qname;
  char* extra = nullptr;
  int nf = 1;
  for (char* f = strtok_r(nullptr, "\t", &save); f; f = strtok_r(nullptr, "\t", &save)) {
    fld[nf] = f;
    switch (nf) {
      case 1: r.flag = (uint16_t)getInt(f); break;
      case 3: r.pos = (uint32_t)(getInt(f) - 1); break;
      case 4: r.mapq = (uint8_t)getInt(f); break;
      case 7: r.pnext = (uint32_t)(getInt(f) - 1); break;
      case 8: (void)getInt(f); break;  // TLEN: converted, then ignored
      default: break;
    }
    if (++nf == 11) {
      extra = strtok_r(nullptr, "\n", &save);
      break;
    }
  }
  if (nf < 11) die(qname, ": poorly formatted SAM/BAM record");
  r.qname = (uint32_t)(qname - base);
  if (r.flag & 0x4) { r.kind = Decoded::UNMAPPED; return; }
  if (!strcmp(qname, "*") || !strcmp(fld[2], "*")) die(qname, ": poorly formatted SAM/BAM record");
  if (r.flag & 0xE00) { r.kind = Decoded::SUPP; return; }
  r.ci = cx.find(fld[2]);
  if (r.mapq < o.minMapQ) { r.kind = Decoded::LOWQ; return; }
  r.length = calcDist(qname, fld[9], fld[5]);
  r.score = samScore(extra);
  r.qual = (uint32_t)(fld[10] - base);
  r.qualLen = (int)strlen(fld[10]);
  r.kind = Decoded::REC;
}

// the state's side of a decoded record (the counters of readSAM / parseBAM, then parseAlign)
inline void applyDecoded(State& S, ReadSet& rs, Counts& C, const Batch& B, size_t i, int qualOffset) {
  const Decoded& r = B.rec[i];
  switch (r.kind) {
    case Decoded::FAIL: die(B.fails[(size_t)r.fail].first, B.fails[(size_t)r.fail].second.c_str());
    case Decoded::UNMAPPED: C.count++; C.unmapped++; return;
    case Decoded::SUPP: C.count++; C.supp++; return;
    case Decoded::LOWQ: C.count++; C.lowMapQ++; return;
    default: break;
  }
  C.count++;
  const char* base = B.at(i);
  record(S, rs, C, base + r.qname, r.flag, r.ci, r.pos, r.mapq, r.length, r.pnext, r.score, base + r.qual, r.qualLen, qualOffset);
}

// A span of text -> the lines gets() would have handed out, one after the other, NUL-terminated, in the batch's own
// memory (the decoders write into them): through the line feed, or REC_MAX - 1 characters of a longer line at a time.
void cutSpan(Batch& B) {
  const char* p = B.span;
  const char* const end = p + B.spanLen;
  B.clearRecords(B.spanLen + B.spanLen / 8 + REC_MAX);   // (clears span / spanLen: p and end are copies)
  while (p < end) {
    const size_t avail = std::min((size_t)(end - p), REC_MAX - 1);
    const void* nl = memchr(p, '\n', avail);
    const size_t k = nl ? (size_t)(static_cast<const char*>(nl) - p) + 1 : avail;
    char* at = B.room(k + 1);
    memcpy(at, p, k);
    at[k] = '\0';
    B.add(k + 1);
    p += k;
  }
}

// decode every record of a batch (any thread); a record that fails ends the batch: nothing behind it will be looked at
template <class F>
void decodeBatch(Batch& B, F one) {
  if (B.spanLen) cutSpan(B);
  B.rec.assign(B.off.size(), Decoded{});
  std::pair<std::string, std::string> msg;
  t_capture = &msg;
  for (size_t i = 0; i < B.off.size(); i++) {
    try {
      one(const_cast<char*>(B.at(i)), B.len[i], B.rec[i]);
    } catch (const DecodeAbort&) {
      B.rec[i].kind = Decoded::FAIL;
      B.rec[i].fail = (int)B.fails.size();
      B.fails.push_back(msg);
      B.rec.resize(i + 1);
      break;
    }
  }
  t_capture = nullptr;
  // the marks (see Batch::mark)
  const size_t n = B.rec.size();
  B.mark.resize(n);
  bool have = false;
  for (size_t i = 0; i < n; i++) {
    const Decoded& r = B.rec[i];
    uint8_t m = r.kind;
    if (r.kind == Decoded::REC) {
      const char* qname = B.at(i) + r.qname;
      if (!have) {
        have = true;
        B.firstRec = (uint32_t)i;
        B.lastName.assign(qname, strnlen(qname, MAX_ALNS));
      } else if (B.lastName == qname)  // (std::string == const char*: the whole name, as record() compares)
        m |= Batch::MARK_SAME;
      else
        B.lastName.assign(qname, strnlen(qname, MAX_ALNS));
    }
    B.mark[i] = m;
  }
}

// The pipeline: a reader thread cuts the input into batches, `nDec` threads decode them, the caller's thread takes them
// in file order.  At most `window` batches exist at a time.
class DecodePipe {
 public:
  typedef std::shared_ptr<Batch> Ptr;
  // fill(B): the reader's step -- append records to B, return false when the input is exhausted (B may hold records)
  template <class Fill, class One>
  DecodePipe(int nDec, Fill fill, One one) : window_((size_t)std::max(4, 3 * nDec)) {
    reader_ = std::thread([this, fill]() {
      for (;;) {
        Ptr b;
        {
          std::lock_guard<std::mutex> lk(m_);
          if (!free_.empty()) {
            b = free_.back();
            free_.pop_back();
          }
        }
        if (!b) b = std::make_shared<Batch>();
        const bool more = fill(*b);
        {
          std::unique_lock<std::mutex> lk(m_);
          if (b->hasWork()) {
            space_.wait(lk, [&] { return order_.size() < window_ || stop_; });
            if (stop_) break;
            order_.push_back(b);
            work_.push_back(b);
          }
          if (!more) {
            eof_ = true;
            avail_.notify_all();
            done_.notify_all();
            if (getenv("GENRICH_HOST_PROF")) {
              struct timespec ts;
              clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
              fprintf(stderr, "reader thread cpu %.3f s\n", ts.tv_sec + ts.tv_nsec * 1e-9);
            }
            break;
          }
        }
        avail_.notify_one();
      }
    });
    for (int i = 0; i < nDec; i++)
      dec_.emplace_back([this, one]() {
        for (;;) {
          Ptr b;
          {
            std::unique_lock<std::mutex> lk(m_);
            avail_.wait(lk, [&] { return !work_.empty() || eof_ || stop_; });
            if (stop_ || (work_.empty() && eof_)) return;
            b = work_.front();
            work_.pop_front();
          }
          decodeBatch(*b, one);
          {
            std::lock_guard<std::mutex> lk(m_);
            b->decoded = true;
          }
          done_.notify_all();
        }
      });
  }
  // the next batch in file order, decoded; nullptr at the end of the input
  Ptr next() {
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return (!order_.empty() && order_.front()->decoded) || (order_.empty() && eof_); });
    if (order_.empty()) return nullptr;
    Ptr b = order_.front();
    order_.pop_front();
    space_.notify_one();
    return b;
  }
  void give(Ptr b) {  // a batch the caller is done with: its memory serves the next one
    std::lock_guard<std::mutex> lk(m_);
    free_.push_back(std::move(b));
  }
  ~DecodePipe() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    space_.notify_all();
    avail_.notify_all();
    if (reader_.joinable()) reader_.join();
    for (auto& t : dec_) t.join();
  }

 private:
  std::mutex m_;
  std::condition_variable avail_, done_, space_;
  std::deque<Ptr> work_, order_;
  std::vector<Ptr> free_;
  size_t window_;
  bool eof_ = false, stop_ = false;
  std::thread reader_;
  std::vector<std::thread> dec_;
};

int g_decoders = 0;  // --threads / GENRICH_THREADS (0 or 1: records are decoded on the parsing thread)
// BAM: inflating the blocks is ~90 % of the ingest's CPU time (binary records decode in no time), so a BAM stream gets
// its own split of the thread budget once it has been recognised (0: as for SAM)
int g_bamInflaters = 0, g_bamDecoders = 0, g_bamState = 0;
int g_stateWorkers = 0;  // (0: as many as decoders)

// reader -> decoders -> the caller's thread, or all three in turn on the caller's thread (one decoder or none)
template <class Fill, class One, class Apply>
void runDecode(int nDec, Fill fill, One one, Apply apply) {
  if (nDec > 1) {
    DecodePipe pipe(nDec, fill, one);
    while (DecodePipe::Ptr b = pipe.next()) {
      for (size_t i = 0; i < b->rec.size(); i++) apply(*b, i);
      pipe.give(std::move(b));
    }
  } else {
    Batch B;
    for (bool more = true; more;) {
      more = fill(B);
      decodeBatch(B, one);
      for (size_t i = 0; i < B.rec.size(); i++) apply(B, i);
    }
  }
}

// ---- parallel state: whole read-name groups to worker threads ---------------------------------------------------
// The decoded records still went through ONE thread's state machine (groups, pairing, weights, intervals): 15-18 M
// records/s.  But a read-name group only depends on its own records; what has an order are its RESULTS (Sink, above).
// So the thread that owns the state only (1) finds where groups begin -- a record starts a group when its name is
// not the current group's, exactly record()'s test, and skipped records (unmapped, supplementary, low MAPQ) never do --,
// (2) cuts the stream into chunks of whole groups right before such a record, and (3) merges the chunks' sinks, in
// order, into the state.  A worker runs the unchanged state machine over its chunk with a ReadSet and Counts of its
// own.  Order of events as in one thread: a chunk's last group is flushed at the chunk's end, i.e. exactly when the
// next chunk's first record -- a group start by construction -- would have flushed it; a record that fails (or a
// die() inside the state machine) ends its chunk there, with the group before it unflushed, as it ends a sequential
// run.
void addCounts(Counts& a, const Counts& b) {
  a.count += b.count; a.unmapped += b.unmapped; a.paired += b.paired; a.single += b.single; a.orphan += b.orphan;
  a.pairedPr += b.pairedPr; a.singlePr += b.singlePr; a.supp += b.supp; a.skipped += b.skipped; a.lowMapQ += b.lowMapQ;
  a.secPair += b.secPair; a.secSingle += b.secSingle;
  a.countPr += b.countPr; a.dupsPr += b.dupsPr; a.countDc += b.countDc; a.dupsDc += b.dupsDc; a.countSn += b.countSn;
  a.dupsSn += b.dupsSn;
}

struct Chunk {
  struct Seg { std::shared_ptr<Batch> b; uint32_t i0, i1; };
  std::vector<Seg> segs;
  size_t nrec = 0;
  // results
  Sink sink;
  Counts C;
  std::vector<Unpair> unpair;
  DupReads dup;
  bool failed = false;
  std::pair<std::string, std::string> fail;
  bool done = false;
};

void processChunk(State& S, Chunk& ch, int qualOffset) {
  std::pair<std::string, std::string> msg;
  t_capture = &msg;
  t_sink = &ch.sink;
  ReadSet rs;
  try {
    for (auto& sg : ch.segs)
      for (uint32_t i = sg.i0; i < sg.i1; i++) applyDecoded(S, rs, ch.C, *sg.b, i, qualOffset);
    flushSet(S, rs, ch.C);
  } catch (const DecodeAbort&) {
    ch.failed = true;
    ch.fail = msg;
  }
  t_sink = nullptr;
  t_capture = nullptr;
  ch.unpair = std::move(rs.unpair);
  ch.dup = std::move(rs.dup);
  if (!S.hot.on) ch.segs.clear();  // (the batches go as soon as nobody needs their bytes; with the int16 checks on, the owner may want them again)
}

// the state's owner takes a chunk's results (in file order)
void mergeChunk(State& S, ReadSet& rs, Counts& C, Chunk& ch, int qualOffset) {
  openSample(S);  // (record() does it at the first record that gets that far; nothing goes to the device before)
  if (S.hot.on) {
    // saveInterval's int16 checks: the chunk's events are counted; one that could be dropped, or that touches a window
    // kept exactly, sends the whole chunk through the state machine again, on this thread and in order, where every
    // read is decided as the reference decides it (the workers' results are not used)
    bool exact = false;
    for (const gx_event& e : ch.sink.ev) exact |= S.hot.add(e);
    if (exact) {
      for (const gx_event& e : ch.sink.ev) S.hot.sub(e);
      ReadSet local;  // (a chunk begins and ends between two read-name groups)
      for (auto& sg : ch.segs)
        for (uint32_t i = sg.i0; i < sg.i1; i++) applyDecoded(S, local, C, *sg.b, i, qualOffset);
      flushSet(S, local, C);
      for (auto& u : local.unpair) rs.unpair.push_back(std::move(u));
      for (auto& d : local.dup.pr) rs.dup.pr.push_back(std::move(d));
      for (auto& d : local.dup.dc) rs.dup.dc.push_back(std::move(d));
      for (auto& d : local.dup.sn) rs.dup.sn.push_back(std::move(d));
      ch.segs.clear();
      return;
    }
    ch.segs.clear();
  }
  for (auto& w : ch.sink.warn) {
    if (w.second) {
      if (S.errCount < MAX_ALNS) fputs(w.first.c_str(), stderr);
      S.errCount++;
    } else
      fputs(w.first.c_str(), stderr);
  }
  if (S.bedOpt && !ch.sink.bed.empty()) fwrite(ch.sink.bed.data(), 1, ch.sink.bed.size(), S.bed.f);
  // (the chunk's events in one piece: pushEvent's bookkeeping once per chunk, not per event)
  if (!ch.sink.ev.empty()) {
    if (!S.gx && S.hot.on) S.hot.all.insert(S.hot.all.end(), ch.sink.ev.begin(), ch.sink.ev.end());
    S.buf.insert(S.buf.end(), ch.sink.ev.begin(), ch.sink.ev.end());
    if (S.buf.size() >= (1u << 20)) {
      if (S.gx && S.sampleOpen) flushEvents(S);
      if (!S.gx || S.sampleOpen) S.buf.clear();  // (--events-only: nothing to send them to; with -b the int16 decisions keep their own copy, hot.all)
    }
  }
  addCounts(C, ch.C);
  for (double x : ch.sink.lenTerms) C.totalLen += x;
  for (auto& u : ch.unpair) rs.unpair.push_back(std::move(u));
  for (auto& d : ch.dup.pr) rs.dup.pr.push_back(std::move(d));
  for (auto& d : ch.dup.dc) rs.dup.dc.push_back(std::move(d));
  for (auto& d : ch.dup.sn) rs.dup.sn.push_back(std::move(d));
  if (ch.failed) die(ch.fail.first, ch.fail.second.c_str());
}

class ChunkPool {
 public:
  typedef std::shared_ptr<Chunk> Ptr;
  ChunkPool(int n, State& S, int qualOffset) {
    for (int i = 0; i < n; i++)
      th_.emplace_back([this, &S, qualOffset]() {
        for (;;) {
          Ptr c;
          {
            std::unique_lock<std::mutex> lk(m_);
            avail_.wait(lk, [&] { return !work_.empty() || stop_; });
            if (work_.empty()) return;
            c = work_.front();
            work_.pop_front();
          }
          processChunk(S, *c, qualOffset);
          {
            std::lock_guard<std::mutex> lk(m_);
            c->done = true;
          }
          done_.notify_all();
        }
      });
  }
  void submit(Ptr c) {
    {
      std::lock_guard<std::mutex> lk(m_);
      work_.push_back(c);
      order_.push_back(c);
    }
    avail_.notify_one();
  }
  // the next finished chunk in file order; with `block` waits for it, otherwise nullptr when it is not ready
  Ptr take(bool block) {
    std::unique_lock<std::mutex> lk(m_);
    if (order_.empty()) return nullptr;
    if (block) done_.wait(lk, [&] { return order_.front()->done; });
    else if (!order_.front()->done) return nullptr;
    Ptr c = order_.front();
    order_.pop_front();
    return c;
  }
  size_t pending() {
    std::lock_guard<std::mutex> lk(m_);
    return order_.size();
  }
  ~ChunkPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      work_.clear();
    }
    avail_.notify_all();
    for (auto& t : th_) t.join();
  }

 private:
  std::mutex m_;
  std::condition_variable avail_, done_;
  std::deque<Ptr> work_, order_;
  bool stop_ = false;
  std::vector<std::thread> th_;
};

// records per chunk (GENRICH_CHUNK_RECS: tests cut a chunk at every group)
const size_t CHUNK_RECS = getenv("GENRICH_CHUNK_RECS") ? (size_t)std::max(1L, atol(getenv("GENRICH_CHUNK_RECS"))) : (size_t)4096;
const bool g_serialState = getenv("GENRICH_SERIAL_STATE") != nullptr;  // (the one-thread state machine of before)

// reader -> decoders -> (this thread: group starts, chunks) -> workers -> (this thread: merge, in order)
template <class Fill, class One>
void runParallelState(State& S, ReadSet& rs, Counts& C, int nThreads, int qualOffset, Fill fill, One one) {
  DecodePipe pipe(nThreads, fill, one);
  ChunkPool pool(g_stateWorkers ? g_stateWorkers : nThreads, S, qualOffset);
  const size_t window = (size_t)std::max(8, 4 * nThreads);
  std::shared_ptr<Chunk> cur = std::make_shared<Chunk>();
  bool have = false;
  std::string name;  // the current group's name as record() keeps it (the first MAX_ALNS characters)
  auto mergeReady = [&](bool block) {
    while (ChunkPool::Ptr c = pool.take(block)) mergeChunk(S, rs, C, *c, qualOffset);
  };
  while (DecodePipe::Ptr b = pipe.next()) {
    uint32_t i0 = 0;
    const uint32_t n = (uint32_t)b->rec.size();
    // A long run of records that never reach the state machine (unmapped, supplementary, low MAPQ: counters only) is
    // counted here and left out of the chunk: a chunk cannot end inside an open group, and a file that goes on for
    // millions of such records would otherwise keep all their batches in memory until the next group starts.
    constexpr uint32_t NONE = 0xFFFFFFFFu, LONG_RUN = 64;
    uint32_t runStart = NONE;
    auto endRun = [&](uint32_t i) {
      if (runStart != NONE && i - runStart >= LONG_RUN) {
        if (runStart > i0) {
          cur->segs.push_back(Chunk::Seg{b, i0, runStart});
          cur->nrec += runStart - i0;
        }
        for (uint32_t k = runStart; k < i; k++) {
          C.count++;
          const uint8_t kind = b->mark[k] & 7u;
          if (kind == Decoded::UNMAPPED) C.unmapped++;
          else if (kind == Decoded::SUPP) C.supp++;
          else C.lowMapQ++;
        }
        i0 = i;
      }
      runStart = NONE;
    };
    const uint8_t* mark = b->mark.data();
    for (uint32_t i = 0; i < n; i++) {
      const uint8_t kind = mark[i] & 7u;
      if (kind == Decoded::UNMAPPED || kind == Decoded::SUPP || kind == Decoded::LOWQ) {
        if (runStart == NONE) runStart = i;
        continue;
      }
      endRun(i);
      if (kind != Decoded::REC) continue;  // (a record that failed keeps its place in the chunk: it ends the run there)
      if (i == b->firstRec) {
        // (the batch's first record that reaches the state machine: against the name carried over from the batches before)
        const char* qname = b->at(i) + b->rec[i].qname;
        if (have && name == qname) continue;  // (std::string == const char*: the whole name, as record() compares)
      } else if (mark[i] & Batch::MARK_SAME)
        continue;
      // a group starts at record i (its name, as record() keeps it, is needed again at the next batch's first record only)
      have = true;
      if (cur->nrec + (i - i0) >= CHUNK_RECS) {
        if (i > i0) cur->segs.push_back(Chunk::Seg{b, i0, i});
        cur->nrec += i - i0;
        pool.submit(cur);
        cur = std::make_shared<Chunk>();
        i0 = i;
      }
    }
    endRun(n);
    if (b->firstRec != 0xFFFFFFFFu) name = b->lastName;  // (have is set: the first REC record either continued a group or began one)
    if (n > i0) {
      cur->segs.push_back(Chunk::Seg{b, i0, n});
      cur->nrec += n - i0;
    }
    mergeReady(false);
    while (pool.pending() > window) {  // (the workers are behind: wait for the oldest chunk rather than queue without bound)
      ChunkPool::Ptr c = pool.take(true);
      if (c) mergeChunk(S, rs, C, *c, qualOffset);
    }
  }
  if (cur->nrec) pool.submit(cur);
  while (pool.pending()) {
    ChunkPool::Ptr c = pool.take(true);
    if (c) mergeChunk(S, rs, C, *c, qualOffset);
  }
}

// a failure of the READER (a truncated BAM record ...) travels as a pseudo-record of length 0 whose bytes are the
// message: the decoder "dies" with it, and it is raised in file order like any other
void readerFailure(Batch& B, const char* tail) {
  const size_t n = strlen(tail) + 1;
  memcpy(B.room(n), tail, n);
  B.off.push_back((uint32_t)B.used);
  B.len.push_back(0u);
  B.used += n;
}

uint64_t readSAM(State& S, In& in, Counts& C) {
  std::vector<char> line(REC_MAX);
  ReadSet rs;
  // the header: on this thread, line by line (it builds the table the decoders look reference names up in)
  char* l;
  while ((l = in.gets(line.data(), (int)line.size())) && l[0] == '@') headerLine(S, l);
  if (l) {
    const ChromIndex cx(S);
    const Opts& o = S.o;
    bool firstPending = true, firstOnly = true;
    auto fill = [&in, &line, &firstPending, &firstOnly](Batch& B) -> bool {
      B.reset(BATCH_BYTES + REC_MAX);
      if (firstPending) {  // (the line that ended the header)
        firstPending = false;
        const size_t n = strlen(line.data()) + 1;
        memcpy(B.room(n), line.data(), n);
        B.add(n);
      }
      if (firstOnly) {  // (the spans start with the next batch)
        firstOnly = false;
        if (!B.off.empty()) return true;
      }
      if (in.plainDirect()) {
        // a plain mapped file: the batch is the next BATCH_BYTES of it, extended to the end of the line they end in;
        // the decoder that gets it cuts it into lines (cutSpan)
        const char* p = reinterpret_cast<const char*>(in.plainPtr());
        const size_t left = in.plainLeft();
        if (!left) return false;
        size_t take = std::min(left, BATCH_BYTES);
        if (take < left) {
          const void* nl = memchr(p + take - 1, '\n', left - take + 1);
          take = nl ? (size_t)(static_cast<const char*>(nl) - p) + 1 : left;
        }
        B.span = p;
        B.spanLen = take;
        in.plainTake(take);
        return take < left;
      }
      while (B.used < BATCH_BYTES) {
        char* at = B.room(REC_MAX);
        if (!in.gets(at, (int)REC_MAX)) return false;
        B.add(strlen(at) + 1);
      }
      return true;
    };
    auto one = [&o, &cx](char* rec, uint32_t, Decoded& r) { decodeSamLine(o, cx, rec, r); };
    if (g_decoders > 1 && !g_serialState)
      runParallelState(S, rs, C, g_decoders, 33, fill, one);
    else
      runDecode(g_decoders, fill, one, [&](const Batch& B, size_t i) { applyDecoded(S, rs, C, B, i, 33); });
  }
  finishFile(S, rs, C);
  return C.count;
}

// ---- BAM (readBAM 4983-5068, parseBAM 4826-4977, getBAMscore 4751) -----------------------------
bool gzReadAll(In& g, void* dst, size_t n) { return n == 0 || g.read(dst, n) == n; }
// readInt32 4628-4640.  Without `must`, the end of the stream -- between records or inside the length
// field -- returns EOF (-1); so does a length field that holds -1, which the reference cannot tell apart.
int32_t rdI32(In& g, bool must) {
  uint8_t b[4];
  int k = (int)g.read(b, 4);
  if (k != 4) {
    if (!must) return -1;
    die("", "Cannot parse BAM file");
  }
  return (int32_t)(b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24));
}

float bamScore(const uint8_t* p, const uint8_t* end) {
  // getBAMscore 4751: the scan stops once fewer than five bytes remain (`while (i < len - 4)`), so a
  // one-byte-valued tag at the very end of the record is never looked at -- kept as is
  const uint8_t* stop = end - 4;
  while (p < stop) {
    const bool isAS = p[0] == 'A' && p[1] == 'S';
    const char ty = (char)p[2];
    p += 3;
    auto need = [&](size_t n) { if (p + n > end) die("", "Poorly formatted BAM auxiliary field"); };
    if (isAS && ty != 'c' && ty != 'C' && ty != 's' && ty != 'S' && ty != 'i' && ty != 'I') {
      char msg[4] = "' '";  // (an alignment score of another type -- 'A', 'f', ... -- is an error, 4782-4786)
      msg[1] = ty;
      die(msg, ": unknown value type in BAM auxiliary field");
    }
    switch (ty) {
      case 'A': case 'c': case 'C': need(1); if (isAS) return ty == 'c' ? (float)(int8_t)p[0] : (float)p[0]; p += 1; break;
      case 's': case 'S': { need(2); uint16_t v = (uint16_t)(p[0] | (p[1] << 8)); if (isAS) return ty == 's' ? (float)(int16_t)v : (float)v; p += 2; break; }
      case 'i': case 'I': { need(4); uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); if (isAS) return ty == 'i' ? (float)(int32_t)v : (float)v; p += 4; break; }
      case 'f': need(4); p += 4; break;
      case 'Z': case 'H': while (p < end && *p) p++; p++; break;
      case 'B': {  // arrayLen 4712-4731
        need(5);
        char st = (char)p[0];
        size_t w;
        switch (st) {
          case 'c': case 'C': w = 1; break;
          case 's': case 'S': w = 2; break;
          case 'i': case 'I': case 'f': w = 4; break;
          default: {
            char msg[4] = "' '";
            msg[1] = st;
            die(msg, ": unknown value type in BAM auxiliary field");
          }
        }
        uint32_t cnt = p[1] | (p[2] << 8) | (p[3] << 16) | ((uint32_t)p[4] << 24);
        p += 5 + (size_t)cnt * w;
        break;
      }
      default: {
        char msg[4] = "' '";
        msg[1] = ty;
        die(msg, ": unknown value type in BAM auxiliary field");
      }
    }
  }
  return NOSCORE;
}

// the text part of a BAM header (readBAM 5007-5031): only its first line is looked at -- it must be an @HD
// line, and with the sort-order check on, one that says SO:queryname
void bamHeaderText(State& S, In& g) {
  int32_t l_text = rdI32(g, true);
  std::vector<char> text((size_t)std::max(0, l_text) + 1, 0);
  if (l_text > 0 && !gzReadAll(g, text.data(), (size_t)l_text)) die("", "Cannot parse BAM file");
  std::string firstLine(text.data(), strcspn(text.data(), "\n"));
  std::vector<char> tmp(firstLine.begin(), firstLine.end());
  tmp.push_back('\0');
  char* tag = strtok(tmp.data(), "\t");
  if (!tag || strcmp(tag, "@HD")) die("", "Cannot parse BAM file");
  const char* order = nullptr;
  for (char* f = strtok(nullptr, "\t"); f; f = strtok(nullptr, "\t"))
    if (!strncmp(f, "SO:", 3)) order = f + 3;
  if (S.o.sortOpt && (!order || strcmp(order, "queryname")))
    die("", "SAM/BAM file not sorted by queryname (samtools sort -n)");
}

// the reference table of a BAM header (readBAM, Genrich.c:5038-5049): n_ref x {l_name, name (NUL-terminated), l_ref};
// one parser for the header pre-scan and the real pass, so neither hands saveChrom an unterminated name
std::vector<int> bamRefTable(State& S, In& g) {
  int32_t n_ref = rdI32(g, true);
  std::vector<int> idx((size_t)std::max(0, n_ref));
  for (int i = 0; i < n_ref; i++) {
    int32_t len = rdI32(g, true);
    if (len < 1 || len > 65520) die("", "Cannot parse BAM file");
    std::vector<char> nm((size_t)len);
    if (!gzReadAll(g, nm.data(), (size_t)len) || nm[len - 1] != '\0') die("", "Cannot parse BAM file");
    idx[i] = saveChrom(S, nm.data(), (uint32_t)rdI32(g, true));
  }
  return idx;
}

// one BAM alignment block (parseBAM 4826-4977 up to the call of parseAlign; loadBAMfields 4660-4688)
void decodeBamBlock(const Opts& o, const std::vector<int>& idx, const char* rec, uint32_t blkLen, Decoded& r) {
  const uint8_t* blk = reinterpret_cast<const uint8_t*>(rec);
  if (blkLen == 0) die("", reinterpret_cast<const char*>(blk));  // (readerFailure)
  const int32_t n_ref = (int32_t)idx.size();
  auto i32 = [&](size_t p) { return (int32_t)(blk[p] | (blk[p + 1] << 8) | (blk[p + 2] << 16) | ((uint32_t)blk[p + 3] << 24)); };
  auto u16 = [&](size_t p) { return (uint16_t)(blk[p] | (blk[p + 1] << 8)); };
  // the name length is a signed byte, and only the end of the last field is checked against the block (4885).
  // (Offsets that leave the block on the way are an error here; the reference reads whatever its line buffer
  // holds there.)
  const int32_t refID = i32(0), pos = i32(4);
  const int l_read_name = (int8_t)blk[8];
  r.mapq = blk[9];
  const uint16_t n_cigar = u16(12);
  r.flag = u16(14);
  const int32_t l_seq = i32(16), next_pos = i32(24);
  const long long nameOff = 32, cigOff = nameOff + l_read_name, seqOff = cigOff + 4LL * n_cigar,
                  qualOff = seqOff + (l_seq + 1) / 2, auxOff = qualOff + l_seq;
  if (auxOff > (long long)blkLen) die("", "Cannot parse BAM file");
  if (cigOff < 32 || seqOff > (long long)blkLen || qualOff < 32 || qualOff > (long long)blkLen || auxOff < 32 ||
      !memchr(blk + nameOff, '\0', blkLen - (size_t)nameOff))
    die("", "Cannot parse BAM file");
  const char* qname = (const char*)&blk[nameOff];
  r.qname = (uint32_t)nameOff;
  if (r.flag & 0x4) { r.kind = Decoded::UNMAPPED; return; }
  if (!strcmp(qname, "*") || refID < 0 || refID >= n_ref || pos < 0) die(qname, ": poorly formatted SAM/BAM record");
  if (r.flag & 0xE00) { r.kind = Decoded::SUPP; return; }
  if (r.mapq < o.minMapQ) { r.kind = Decoded::LOWQ; return; }
  // calcDistBAM 4694-4706: the distance to the 3' end is l_seq - I - S + D, with no check of the CIGAR
  // against the sequence (unlike calcDist for SAM) -- also when SEQ is absent (l_seq = 0)
  int length = l_seq;
  for (int k = 0; k < n_cigar; k++) {
    const uint32_t c = (uint32_t)i32((size_t)cigOff + 4 * (size_t)k);
    const int op = (int)(c & 15);
    if (op == 1 || op == 4) length -= (int)(c >> 4);
    else if (op == 2) length += (int)(c >> 4);
  }
  r.length = length;
  r.score = bamScore(blk + (size_t)auxOff, blk + blkLen);
  r.ci = idx[(size_t)refID];
  r.pos = (uint32_t)pos;
  r.pnext = (uint32_t)next_pos;
  r.qual = (uint32_t)qualOff;
  r.qualLen = l_seq;
  r.kind = Decoded::REC;
}

uint64_t readBAM(State& S, In& in, Counts& C) {
  In& g = in;
  bamHeaderText(S, g);
  const std::vector<int> idx = bamRefTable(S, g);
  ReadSet rs;
  const Opts& o = S.o;
  // the reader: alignment blocks (block_size, then that many bytes), copied out of the inflated stream
  auto fill = [&g](Batch& B) -> bool {
    B.reset(BATCH_BYTES + REC_MAX);
    const void* held = nullptr;
    // (this thread reads four bytes of every alignment block -- its size -- out of data that an inflater on another core
    // has just written: a cache miss per record, 76 ns each, was the BAM pipeline's bound on a host with many cores.
    // The member is walked front to back, so its lines are asked for 2 KiB ahead of the record being delimited.)
    const uint8_t* pf = nullptr;
    while (B.used + B.extBytes < BATCH_BYTES) {
      // (a block that lies inside one inflated BGZF member stays where it is -- the batch keeps the member alive and
      // the decoders read it there: one peek for its size, one for its bytes; one that straddles two members -- or any
      // block, with zlib's reader -- is copied through read())
      if (const uint8_t* p4 = g.peek(4)) {
        const int32_t sz = (int32_t)(p4[0] | (p4[1] << 8) | (p4[2] << 16) | ((uint32_t)p4[3] << 24));
        if (sz >= 32) {
          if (const uint8_t* whole = g.peek(4 + (size_t)sz)) {
            if (g.currentId() != held) {
              held = g.currentId();
              B.holds.push_back(g.holdCurrent());
              pf = whole;
            }
            {
              const uint8_t* const want = whole + std::min<size_t>(g.leftInCurrent(), 2048);
              if (pf < whole) pf = whole;
              for (; pf < want; pf += 64) __builtin_prefetch(pf);
            }
            B.addExt(whole + 4, (size_t)sz);
            g.advance(4 + (size_t)sz);
            continue;
          }
        }
      }
      const int32_t bs = rdI32(g, false);
      if (bs == -1) return false;  // (see rdI32)
      // (a negative size passes the reference's unsigned test and fails its end-of-block test, 4870 / 4885)
      if (bs < 32 || !gzReadAll(g, B.room((size_t)bs), (size_t)bs)) {
        readerFailure(B, "Cannot parse BAM file");
        return false;
      }
      B.add((size_t)bs);
    }
    return true;
  };
  auto one = [&o, &idx](char* rec, uint32_t len, Decoded& r) { decodeBamBlock(o, idx, rec, len, r); };
  if (g_decoders > 1 && !g_serialState)
    runParallelState(S, rs, C, g_decoders, 0, fill, one);
  else
    runDecode(g_decoders, fill, one, [&](const Batch& B, size_t i) { applyDecoded(S, rs, C, B, i, 0); });
  finishFile(S, rs, C);
  return C.count;
}

// ---- -v accounting (logCounts 5295-5374) -------------------------------------------------------
void logCounts(const State& S, const Counts& C, bool bam) {
  const Opts& o = S.o;
  if (S.errCount > MAX_ALNS) fprintf(stderr, "(another %ld warning messages suppressed)\n", (long)(S.errCount - MAX_ALNS));
  double avgLen = C.pairedPr ? C.totalLen / C.pairedPr : 0.0;
  fprintf(stderr, "  %s records analyzed: %11ld\n", bam ? "BAM" : "SAM", (long)C.count);
  if (C.unmapped) fprintf(stderr, "    Unmapped:           %11ld\n", (long)C.unmapped);
  if (C.supp) fprintf(stderr, "    Supp./dups/lowQual: %11ld\n", (long)C.supp);
  if (C.skipped) {
    fprintf(stderr, "    To skipped refs:    %11ld\n", (long)C.skipped);
    fprintf(stderr, "      (");
    bool first = true;
    for (auto& c : S.chrom)
      if (c.skip || !c.save) {
        fprintf(stderr, "%s%s", first ? "" : ",", c.name.c_str());
        first = false;
      }
    fprintf(stderr, ")\n");
  }
  if (C.lowMapQ) fprintf(stderr, "    MAPQ < %-2d:          %11ld\n", o.minMapQ, (long)C.lowMapQ);
  fprintf(stderr, "    Paired alignments:  %11ld\n", (long)C.paired);
  if (C.secPair) fprintf(stderr, "      secondary alns:   %11ld\n", (long)C.secPair);
  if (C.orphan) fprintf(stderr, "      \"orphan\" alns:    %11ld\t** Warning! **\n", (long)C.orphan);
  fprintf(stderr, "    Unpaired alignments:%11ld\n", (long)C.single);
  if (C.secSingle) fprintf(stderr, "      secondary alns:   %11ld\n", (long)C.secSingle);
  if (o.dupsOpt) {
    fprintf(stderr, "  PCR duplicates --\n");
    fprintf(stderr, "    Paired aln sets:    %11ld\n", (long)C.countPr);
    fprintf(stderr, "      duplicates:       %11ld (%.1f%%)\n", (long)C.dupsPr, C.countPr ? 100.0f * C.dupsPr / C.countPr : 0.0f);
    if (o.singleOpt) {
      fprintf(stderr, "    Discordant aln sets:%11ld\n", (long)C.countDc);
      fprintf(stderr, "      duplicates:       %11ld (%.1f%%)\n", (long)C.dupsDc, C.countDc ? 100.0f * C.dupsDc / C.countDc : 0.0f);
      fprintf(stderr, "    Singleton aln sets: %11ld\n", (long)C.countSn);
      fprintf(stderr, "      duplicates:       %11ld (%.1f%%)\n", (long)C.dupsSn, C.countSn ? 100.0f * C.dupsSn / C.countSn : 0.0f);
    }
  }
  fprintf(stderr, "  Fragments analyzed:   %11ld\n", (long)(C.singlePr + C.pairedPr));
  fprintf(stderr, "    Full fragments:     %11ld\n", (long)C.pairedPr);
  if (C.pairedPr && !o.atacOpt) fprintf(stderr, "      (avg. length: %.1fbp)\n", avgLen);
  if (o.singleOpt) {
    fprintf(stderr, "    Half fragments:     %11ld\n", (long)C.singlePr);
    if (C.singlePr) {
      fprintf(stderr, "      (from unpaired alns");
      if (o.extendOpt) fprintf(stderr, ", extended to %dbp", o.extend);
      else if (o.avgExtOpt && C.pairedPr) fprintf(stderr, ", extended to %dbp", (int)(avgLen + 0.5));
      fprintf(stderr, ")\n");
    }
  }
  if (o.atacOpt) {
    fprintf(stderr, "    ATAC-seq cut sites: %11ld\n", (long)(2 * C.pairedPr + C.singlePr));
    fprintf(stderr, "      (expanded to length %dbp)\n", o.atacLen5 + o.atacLen3);
  }
}

// header pre-scan: the chromosome table must be complete (in the order the reference would build
// it: first appearance over t1, c1, t2, c2, ...) before anything is sent to the device
// BAM or SAM?  checkBAM (5107-5125) looks at the first characters of a gzip-compressed input only:
// "BAM\1" is BAM; the end of the stream -- or, `char m = gzgetc()` being signed, a 0xFF byte -- before a
// character that differs from the magic is "cannot open file for reading".  Uncompressed input is SAM.
bool sniffBam(In& in, char magic[4], int& got) {
  got = (int)in.read(magic, 4);
  if (!in.compressed()) return false;
  for (int i = 0; i < 4; i++) {
    if (i >= got || (unsigned char)magic[i] == 0xFF) die("", ": cannot open file for reading");
    if (magic[i] != "BAM\1"[i]) {
      // (the reference pushes the character back with gzungetc, which refuses a negative `char`)
      if ((signed char)magic[i] < 0) die("", "Failure in ungetc() call");
      return false;
    }
  }
  return true;
}

In& openStdin(State& S) {  // openRead 5135-5166 for '-'
  if (!S.stdinIn) {
    S.stdinIn.reset(new In);
    openRead(*S.stdinIn, "-");
    if (S.stdinIn->compressed()) die("", "Cannot pipe in gzip-compressed file (use zcat instead)");
  }
  return *S.stdinIn;
}

void scanHeader(State& S, const char* filename, bool ctrl) {
  const bool isStdin = !strcmp(filename, "-");
  if (isStdin && S.stdinIn) return;  // (named twice: the second reading finds it at its end, as in the reference)
  In file;
  In& in = isStdin ? openStdin(S) : file;
  if (isStdin)
    in.record();
  else if (!in.open(filename, g_threads))
    return;  // reported when the real pass gets to this file, where the reference would find out
  S.ctrl = ctrl;
  char magic[4];
  int got = 0;
  if (sniffBam(in, magic, got)) {
    bamHeaderText(S, in);
    bamRefTable(S, in);
  } else {
    in.unread(magic, (size_t)std::max(0, got));
    std::vector<char> line(65520);
    while (char* l = in.gets(line.data(), (int)line.size())) {
      if (l[0] != '@') break;
      const bool sortSave = S.o.sortOpt;
      S.o.sortOpt = false;  // the sort-order check belongs to the real pass
      headerLine(S, l);
      S.o.sortOpt = sortSave;
    }
  }
  checkIn(in);
  if (isStdin)
    in.replay();
  else
    in.close();
}

void sendChroms(State& S) {
  S.tableFrozen = true;
  if (!S.gx) return;
  size_t n = S.chrom.size();
  if (!n) die("", "No analyzable genome (length=0)");
  std::vector<uint32_t> len(n);
  std::vector<uint8_t> skip(n);
  std::vector<const uint32_t*> bed(n);
  std::vector<int32_t> bedLen(n);
  for (size_t i = 0; i < n; i++) {
    len[i] = S.chrom[i].len;
    skip[i] = S.chrom[i].skip;
    bed[i] = S.chrom[i].bed.data();
    bedLen[i] = (int32_t)S.chrom[i].bed.size();
  }
  Devs& D = S.devs;
  for (gx_ctx* g : D.ctx) check(S, gx_set_chroms(g, (int)n, len.data(), skip.data(), bed.data(), bedLen.data()), g);
  D.owner.assign(n, 0);
  if (D.n() > 1) {
    // longest-processing-time bin packing of the chromosomes by length (the reference's per-chromosome loops,
    // Genrich.c:2172, 1729, 987, are independent): hg38 over 8 GPUs -> heaviest share 400 of 3088 Mbp
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return len[a] > len[b]; });
    std::vector<uint64_t> load(D.n(), 0);
    for (size_t i : order) {
      size_t best = 0;
      for (size_t g = 1; g < D.n(); g++)
        if (load[g] < load[best]) best = g;
      D.owner[i] = (int)best;
      load[best] += len[i];
    }
    for (size_t g = 0; g < D.n(); g++) {
      std::vector<uint8_t> owned(n);
      for (size_t i = 0; i < n; i++) owned[i] = D.owner[i] == (int)g;
      check(S, gx_set_owned(D.ctx[g], owned.data()), D.ctx[g]);
    }
  }
}

void loadBED(State& S, const char* files) {  // loadBED 5187-5238
  std::string list(files);
  std::vector<char> line(65520);
  char* endFn = nullptr;  // (strtok_r as in the reference: the lines are cut up with strtok inside the loop)
  for (char* fn = strtok_r(list.data(), ", ", &endFn); fn; fn = strtok_r(nullptr, ", ", &endFn)) {
    In in;
  openRead(in, fn);
    while (in.gets(line.data(), (int)line.size())) {
      // in the reference's order (5203-5213): a field is converted as soon as it has been cut off, and a
      // missing one is reported with what strtok has left of the line, i.e. the name
      char* name = strtok(line.data(), "\t");
      if (!name) die(line.data(), ": poorly formatted BED record");
      char* a = strtok(nullptr, "\t");
      if (!a) die(line.data(), ": poorly formatted BED record");
      const int p0 = getInt(a);
      char* b = strtok(nullptr, "\t\n");
      if (!b) die(line.data(), ": poorly formatted BED record");
      const int p1 = getInt(b);
      if (p1 <= p0 || p0 < 0 || p1 < 0) {
        char msg[512];
        snprintf(msg, sizeof msg, "%s, %d - %d", name, p0, p1);
        die(msg, ": poorly formatted BED record");
      }
      S.xbed.push_back(BedRec{name, {(uint32_t)p0, (uint32_t)p1}});
    }
    checkIn(in);
  in.close();
  }
}

// ---- -P: call peaks from a bedgraph-ish -f log (findPeaksOnly 5243, callPeaksLog 1277-1488) -------
// Host-only by nature (it re-parses text); the sweep is the same state machine as callPeaks,
// applied to the 6-decimal values of the log, with new -e / -E exclusions cutting intervals.
struct PeakState {
  float auc = 0.0f, summitVal = -1.0f, summitP = -1.0f, summitQ = -1.0f;
  int64_t peakStart = -1, peakEnd = -1;
  uint32_t summitPos = 0, summitLen = 0;
  void reset() { peakStart = -1; summitVal = -1.0f; summitLen = 0; auc = 0.0f; }            // resetVars 932
  void update(uint32_t start, uint32_t end, float pq, float thr, float pv, float qv) {     // updatePeak 943
    uint32_t len = end - start;
    auc += len * (pq - thr);
    if (peakStart == -1) peakStart = start;
    peakEnd = end;
    if (pq > summitVal) {
      summitVal = pq; summitP = pv; summitQ = qv;
      summitPos = (uint32_t)((end + start) / 2 - peakStart);
      summitLen = len;
    } else if (pq == summitVal && len > summitLen) {
      summitPos = (uint32_t)((end + start) / 2 - peakStart);
      summitLen = len;
    }
  }
};

void peaksOnly(State& S, float thr) {
  const Opts& o = S.o;
  In in;
  openRead(in, o.logFile);
  Out out = openWrite(o.outFile, o.gzOut);
  if (o.verbose) fprintf(stderr, "Peak-calling from log file: %s\n", o.logFile);
  std::vector<char> line(65520);
  // header: the LAST -log(p) / -log(q) columns (getIdx 1224-1246)
  if (!in.gets(line.data(), (int)line.size())) die("<header>", ": cannot find field in header of bedgraph-ish log file");
  int idxP = -1, idxQ = -1, nf = 0;
  for (char* f = strtok(line.data(), "\t\n"); f; f = strtok(nullptr, "\t\n"), nf++) {
    if (!strncmp(f, "-log(p)", 7)) idxP = nf;
    else if (!strncmp(f, "-log(q)", 7)) idxQ = nf;
  }
  if (idxP == -1) die("-log(p)", ": cannot find field in header of bedgraph-ish log file");
  if (o.qvalOpt && idxQ == -1) die("-log(q)", ": cannot find field in header of bedgraph-ish log file");
  const int idx = o.qvalOpt ? idxQ : idxP;

  int count = 0;
  uint64_t peakBP = 0, genomeLen = o.genomeLen;
  const bool genomeOpt = genomeLen == 0;
  PeakState P;
  std::string prev, chrName;
  bool skip = false, save = true, warn = false;
  Chrom cur;
  size_t bedIdx = 0;
  uint32_t bedPos = UINT32_MAX;
  auto check = [&](const std::string& name) {  // checkPeak 916 + printPeak 885
    if (P.peakStart != -1 && P.auc >= o.minAUC && P.peakEnd - P.peakStart >= o.minLen) {
      float sc = 1000.0f * P.auc / (P.peakEnd - P.peakStart) + 0.5f;
      unsigned int u = (unsigned int)(long long)sc;
      fprintf(out.f, "%s\t%ld\t%ld\tpeak_%d\t%d\t.\t%f\t%f", name.c_str(), (long)P.peakStart, (long)P.peakEnd, count,
              u < 1000u ? u : 1000u, P.auc, P.summitP);
      if (P.summitQ == GX_SKIP) fprintf(out.f, "\t-1\t%d\n", P.summitPos);
      else fprintf(out.f, "\t%f\t%d\n", P.summitQ, P.summitPos);
      peakBP += (uint64_t)(P.peakEnd - P.peakStart);
      count++;
    }
  };
  auto nextBed = [&]() { bedIdx++; bedPos = bedIdx < cur.bed.size() ? cur.bed[bedIdx] : UINT32_MAX; };
  while (in.gets(line.data(), (int)line.size())) {
    // loadBDG 1252-1272
    char *chr = nullptr, *pStat = nullptr, *qStat = nullptr;
    uint32_t start = 0, end = 0;
    char* f = strtok(line.data(), "\t\n");
    for (int i = 0; i <= idx; i++) {
      if (!f) die("", "Poorly formatted bedgraph-ish log record");
      if (i == 0) chr = f;
      else if (i == 1) start = (uint32_t)getInt(f);
      else if (i == 2) end = (uint32_t)getInt(f);
      else if (i == idxP) pStat = f;
      else if (i == idxQ) qStat = f;
      f = strtok(nullptr, "\t\n");
    }
    if (prev != chr) {  // new chromosome, 1321-1356
      check(prev);
      P.reset();
      skip = false;
      for (auto& x : S.xchr)
        if (x == chr) skip = true;
      if (o.verbose && skip) {
        fprintf(stderr, "Warning! Skipping chromosome %s --\n  ", chr);
        fprintf(stderr, "Reads aligning to it were used in the background");
        fprintf(stderr, " pileup calculation,\n  and its length was included");
        fprintf(stderr, " in the genome length %scalculation\n", o.qvalOpt ? "(and q-value) " : "");
      }
      cur = Chrom();
      cur.name = chr;
      cur.len = UINT32_MAX;  // coordinates cannot be validated here (1344)
      if (!skip) {
        mergeBed(cur, S.xbed, o.verbose);
        bedIdx = 0;
        bedPos = cur.bed.empty() ? UINT32_MAX : cur.bed[0];
        save = true;
      }
      prev = chr;
    }
    chrName = chr;
    if (skip) continue;
    const char* stat = o.qvalOpt ? qStat : pStat;
    if (!strcmp(stat, "NA")) {  // skipped region of the original run
      check(chrName);
      P.reset();
      continue;
    }
    const float pq = getFloat(stat);
    // (with -q the p-value column is converted only where a significant stretch needs it, 1402 / 1452)
    auto pvNow = [&]() { return o.qvalOpt ? getFloat(pStat) : pq; };
    const float qv = o.qvalOpt ? pq : GX_SKIP;
    if (bedPos == start) {  // 1376-1390
      if (save) { check(chrName); P.reset(); }
      save = !save;
      nextBed();
    }
    uint32_t subStart = start;
    while (bedPos > start && bedPos < end) {  // new -E edges inside the interval, 1395-1425
      if (save) {
        if (pq > thr) P.update(subStart, bedPos, pq, thr, pvNow(), qv);
        check(chrName);
        P.reset();
        if (genomeOpt) genomeLen += bedPos - subStart;
      } else
        warn = true;
      subStart = bedPos;
      save = !save;
      nextBed();
    }
    if (!save) { warn = true; continue; }
    start = subStart;
    if (genomeOpt) genomeLen += end - start;
    if (pq > thr)
      P.update(start, end, pq, thr, pvNow(), qv);
    else if ((int64_t)end - P.peakEnd > o.maxGap) {
      check(chrName);
      P.reset();
    }
  }
  check(chrName);
  if (o.verbose) {
    if (warn) {
      fprintf(stderr, "Warning! Skipping given BED regions --\n  ");
      fprintf(stderr, "Reads aligning to them were used in the background");
      fprintf(stderr, " pileup calculation,\n  and the lengths were included");
      fprintf(stderr, " in the genome length %scalculation\n", o.qvalOpt ? "(and q-value) " : "");
    }
    fprintf(stderr, "Peak-calling parameters:\n");
    fprintf(stderr, "  Genome length: %ldbp\n", (long)genomeLen);
    fprintf(stderr, "  Significance threshold: -log(%c) > %.3f\n", o.qvalOpt ? 'q' : 'p', thr);
    fprintf(stderr, "  Min. AUC: %.3f\n", o.minAUC);
    if (o.minLen) fprintf(stderr, "  Min. peak length: %dbp\n", o.minLen);
    fprintf(stderr, "  Max. gap between sites: %dbp\n", o.maxGap);
    fprintf(stderr, "Peaks identified: %d (%ldbp)\n", count, (long)peakBP);
  }
  checkIn(in);
  in.close();
  closeOut(out);
}

void usage() {
  fprintf(stderr,
          "Usage: genrich-amd  -t <file>  -o <file>  [optional arguments]\n"
          "  (same options as Genrich v0.6.2: -t -c -o -f -k -b -z -y -w -x -j -d -D -e -E -m -s\n"
          "   -r -R -p -q -a -l -g -X -P -S -L -v -V)\n"
          "  --device N | --devices 0-7   GPU(s) to use;  --threads N   BGZF inflate threads\n");
  exit(EXIT_FAILURE);
}

}  // namespace

int main(int argc, char** argv) {
  State S;
  Opts& o = S.o;
  static struct option longOpts[] = {{"help", no_argument, nullptr, 'h'},
                                     {"verbose", no_argument, nullptr, 'v'},
                                     {"version", no_argument, nullptr, 'V'},
                                     {"events-only", no_argument, nullptr, 1000},
                                     {"device", required_argument, nullptr, 1001},
                                     {"devices", required_argument, nullptr, 1003},
                                     {"threads", required_argument, nullptr, 1002},
                                     {nullptr, 0, nullptr, 0}};
  {  // BGZF inflate threads and record decoders: --threads N, else GENRICH_THREADS, else up to 16 of the machine's cores
    const char* e = getenv("GENRICH_THREADS");
    unsigned hw = std::thread::hardware_concurrency();
    g_threads = e ? atoi(e) : (int)std::min(16u, hw ? hw : 1u);
  }
  int c;
  while ((c = getopt_long(argc, argv, "ht:c:o:f:k:b:zyw:xjd:De:E:m:s:p:q:a:l:g:rR:XPSL:vV", longOpts, nullptr)) != -1)
    switch (c) {
      case 't': o.inFile = optarg; break;
      case 'c': o.ctrlFile = optarg; break;
      case 'o': o.outFile = optarg; break;
      case 'f': o.logFile = optarg; break;
      case 'k': o.pileFile = optarg; break;
      case 'b': o.bedFile = optarg; break;
      case 'z': o.gzOut = true; break;
      case 'y': o.singleOpt = true; break;
      case 'w': o.extend = getInt(optarg); o.extendOpt = true; break;
      case 'x': o.avgExtOpt = true; break;
      case 'j': o.atacOpt = true; break;
      case 'd': o.atacLen5 = getInt(optarg); break;
      case 'D': o.atacAdj = false; break;
      case 'e': o.xchrom = optarg; break;
      case 'E': o.xFile = optarg; break;
      case 'm': o.minMapQ = getInt(optarg); break;
      case 's': o.asDiff = getFloat(optarg); break;
      case 'p': o.pqvalue = getFloat(optarg); break;
      case 'q': o.pqvalue = getFloat(optarg); o.qvalOpt = true; break;
      case 'a': o.minAUC = getFloat(optarg); break;
      case 'l': o.minLen = getInt(optarg); break;
      case 'g': o.maxGap = getInt(optarg); break;
      case 'r': o.dupsOpt = true; break;
      case 'R': o.dupsFile = optarg; break;
      case 'X': o.peaksOpt = false; break;
      case 'P': o.peaksOnly = true; break;
      case 'S': o.sortOpt = false; break;
      case 'L': o.genomeLen = (uint64_t)getLong(optarg); break;
      case 'v': o.verbose = true; break;
      case 'V': fprintf(stderr, "genrich-amd, version %s (Genrich 0.6.2 hot path on MI355X)\n", VERSION); exit(EXIT_FAILURE);
      case 1000: o.eventsOnly = true; break;
      case 1001: o.device = getInt(optarg); break;
      case 1002: g_threads = getInt(optarg); break;
      case 1003: {  // --devices 0,1,2 or 0-7
        std::string list(optarg);
        for (char* t = strtok(list.data(), ","); t; t = strtok(nullptr, ",")) {
          const char* dash = strchr(t, '-');
          if (dash && dash != t) {
            const int a = atoi(t), b = atoi(dash + 1);
            for (int d = a; d <= b; d++) o.devices.push_back(d);
          } else
            o.devices.push_back(getInt(t));
        }
        break;
      }
      case 'h': usage();
      default: exit(EXIT_FAILURE);
    }
  if (optind < argc) die(argv[optind], ": unknown command-line argument");
  if ((o.peaksOpt && !o.outFile && !o.eventsOnly) || (o.peaksOnly && !o.logFile) || (!o.peaksOnly && !o.inFile)) {
    fprintf(stderr, "Error! Need input/output files\n");
    usage();
  }
  if (o.avgExtOpt) { o.singleOpt = true; o.extendOpt = false; }
  if (o.extendOpt) {
    o.singleOpt = true;
    if (o.extend <= 0) die("", "Extension length must be > 0");
  }
  if (o.atacOpt) {
    o.avgExtOpt = o.extendOpt = false;
    if (o.atacLen5 <= 0) die("", "ATAC-seq interval length must be > 0");
    o.atacLen3 = (int)(o.atacLen5 / 2.0f + 0.5f);
    o.atacLen5 /= 2;
  }
  if (o.minLen < 0) die("", "Minimum peak length must be >= 0");
  if (o.minAUC < 0.0f) die("", "Minimum AUC must be >= 0.0");
  if (o.asDiff < 0.0f) die("", "Secondary alignment score threshold must be >= 0.0");
  if (o.xchrom) {
    std::string list(o.xchrom);
    for (char* t = strtok(list.data(), ", "); t; t = strtok(nullptr, ", ")) S.xchr.push_back(t);
  }
  g_decoders = g_threads;
  {
    // --threads N sizes three pools (BGZF inflaters, record decoders, state workers) next to the reader and the owner:
    // on a host with fewer than 3 N + 2 cores the pools share the budget instead of each taking N
    // (inflate N / 4, decode N / 2, state workers N / 2: the stages' measured shares of the work)
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)(3 * g_threads + 2) > hw && g_threads > 2) {
      const int n = g_threads;
      g_stateWorkers = std::max(1, n / 2);
      g_decoders = std::max(2, n / 2);
      g_threads = std::max(2, n / 4);   // (from here on: the inflaters; two keep the BGZF reader with its block checks)
      g_bamInflaters = std::max(2, n / 2);
      g_bamDecoders = std::max(2, n / 4);
      g_bamState = std::max(1, n / 4);
    } else if (hw && g_threads > 1) {
      // cores to spare: a BAM stream gets twice the inflaters (the other pools keep up with 40 M records/s of SAM text)
      g_bamInflaters = std::min(2 * g_threads, std::max(g_threads, (int)hw - 2 * g_threads - 2));
    }
    if (getenv("GENRICH_THREADS_REPORT"))
      fprintf(stderr, "[threads] inflate %d, decode %d, state %d; BAM: inflate %d, decode %d, state %d\n", g_threads, g_decoders,
              g_stateWorkers ? g_stateWorkers : g_decoders, g_bamInflaters ? g_bamInflaters : g_threads,
              g_bamDecoders ? g_bamDecoders : g_decoders, g_bamState ? g_bamState : (g_stateWorkers ? g_stateWorkers : g_decoders));
  }
  if (o.pqvalue <= 0.0f || o.pqvalue > 1.0f) die("", "p-/q-value must be in (0,1]");
  const float thr = -log10f(o.pqvalue);

  if (o.xFile) loadBED(S, o.xFile);
  if (o.peaksOnly) {  // runProgram 5398-5403
    peaksOnly(S, thr);
    return EXIT_SUCCESS;
  }
  if (o.bedFile) { S.bed = openWrite(o.bedFile, o.gzOut); S.bedOpt = true; }
  if (o.dupsOpt && o.dupsFile) { S.dups = openWrite(o.dupsFile, o.gzOut); S.dupsVerb = true; }  // 5411-5415
  // (--events-only writes nothing but the -b file: without one there is nothing the checks could change)
  if (o.eventsOnly) S.hot.on = o.bedFile != nullptr && getenv("GENRICH_NO_INT16") == nullptr;
  if (!o.eventsOnly) {
    gx_params par{};
    par.thr = thr;
    par.qval_opt = o.qvalOpt;
    par.min_auc = o.minAUC;
    par.min_len = o.minLen;
    par.max_gap = o.maxGap;
    par.device = o.device;
    par.genome_len = o.genomeLen;
    if (o.devices.empty()) o.devices.push_back(o.device);
    if (o.devices.size() > 64) die("", "at most 64 devices");
    Devs& D = S.devs;
    for (int d : o.devices) {
      par.device = d;
      gx_ctx* g = nullptr;
      int rc = gx_create(&g, &par);
      if (rc) die(g ? gx_last_error(g) : gx_strerror(rc), "");
      check(S, gx_set_keep_pileups(g, o.logFile || o.pileFile), g);  // only -f / -k print pileup values
      if (o.asDiff > 0.0f) check(S, gx_expect_fractional(g, 1), g);  // (-s: multimapping reads get weights 1/k)
      D.ctx.push_back(g);
    }
    S.gx = D.ctx[0];
    S.hot.on = getenv("GENRICH_NO_INT16") == nullptr;  // (saveInterval's int16 checks, read by read: HotWindows)
    D.buf.resize(D.n());
    if (D.n() > 1) {
      const int W = (int)D.n();
      bool distinct = true;
      for (int a = 0; a < W; a++)
        for (int b = a + 1; b < W; b++) distinct = distinct && o.devices[a] != o.devices[b];
      if (distinct) {
        // the library's own collectives: one RCCL communicator over the devices (xGMI), joined side by side
        char id[128];
        int rc = gx_rccl_unique_id(id, sizeof id);
        if (rc) die(gx_strerror(rc), "");
        onEachDevice(S, [&](int g) { return gx_set_rccl(D.ctx[g], g, W, id); });
      } else {
        // a device named twice (RCCL wants one rank per GPU): the exchanges go through host callbacks between
        // the threads -- the library's validation mode, here for tests on a single GPU
        pthread_barrier_init(&D.bar, nullptr, (unsigned)W);
        D.barInit = true;
        D.red.resize(W);
        D.users.resize(W);
        for (int g = 0; g < W; g++) {
          D.users[g] = Devs::User{&D, g};
          check(S, gx_set_collectives(D.ctx[g], g, W, devsAllreduce, &D.users[g]), D.ctx[g]);
        }
      }
    }
  }

  // loop over the comma-separated treatment / control lists (runProgram 5455-5585)
  std::string tList(o.inFile), cList(o.ctrlFile ? o.ctrlFile : "");
  std::vector<std::string> tFiles, cFiles;
  for (char* t = strtok(tList.data(), ", "); t; t = strtok(nullptr, ", ")) tFiles.push_back(t);
  for (char* t = strtok(cList.data(), ", "); t; t = strtok(nullptr, ", ")) cFiles.push_back(t);
  for (size_t r = 0; r < tFiles.size(); r++) {  // pre-scan of every header, reference order
    scanHeader(S, tFiles[r].c_str(), false);
    if (o.ctrlFile && r < cFiles.size() && cFiles[r] != "null") scanHeader(S, cFiles[r].c_str(), true);
  }
  sendChroms(S);
  for (size_t r = 0; r < tFiles.size(); r++) {
    for (auto& ch : S.chrom) ch.save = false;
    const char* ctrlName = !o.ctrlFile ? nullptr : (r < cFiles.size() ? cFiles[r].c_str() : nullptr);
    for (int i = 0; i < 2; i++) {
      const char* filename = i ? ctrlName : tFiles[r].c_str();
      if (i && filename && !strcmp(filename, "null")) filename = nullptr;
      if (i && !filename) {
        if (o.verbose) fprintf(stderr, "- control file #%d not provided -\n", S.sample);
        float lambda = 0;
        for (gx_ctx* g : S.devs.ctx) check(S, gx_sample_no_control(g, &lambda), g);  // (lambda is genome-wide: the same everywhere)
        if (o.verbose && S.gx) fprintf(stderr, "  Background pileup value: %f\n", lambda);
        break;
      }
      S.ctrl = i;
      S.sampleOpen = false;
      S.errCount = 0;
      S.buf.clear();
      In file;
      const bool isStdin = !strcmp(filename, "-");
      In& in = isStdin ? openStdin(S) : file;
      if (!isStdin) openRead(in, filename);
      // BAM or SAM?  (checkBAM 5107: the decompressed stream starts with "BAM\1")
      char magic[4] = {0};
      int got = 0;
      const bool bam = sniffBam(in, magic, got);
      if (o.verbose) fprintf(stderr, "Processing %s file #%d: %s\n", i ? "control" : "experimental", S.sample, filename);
      if (S.dupsVerb) fprintf(S.dups.f, "# %s file #%d: %s\n", i ? "control" : "experimental", S.sample, filename);
      Counts C;
      if (bam) {
        const int dec = g_decoders, st = g_stateWorkers;
        if (g_bamInflaters) in.growWorkers(g_bamInflaters);
        if (g_bamDecoders) g_decoders = g_bamDecoders;
        if (g_bamState) g_stateWorkers = g_bamState;
        readBAM(S, in, C);
        g_decoders = dec;
        g_stateWorkers = st;
      } else {
        in.unread(magic, (size_t)std::max(0, got));  // the sniffed bytes are the beginning of the first line
        readSAM(S, in, C);
      }
      checkIn(in);
      if (!isStdin) in.close();
      openSample(S);  // a file without a single usable record still opens (and closes) its sample
      if (S.gx) flushEvents(S);
      S.buf.clear();
      if (o.verbose) logCounts(S, C, bam);
      if (S.gx) {
        double fragLen = 0;
        float lambda = 0, factor = 0;
        onEachDevice(S, [&](int g) {
          double fl = 0;
          float la = 0, fa = 0;
          const int rc = gx_sample_end(S.devs.ctx[g], &fl, &la, &fa);
          if (g == 0) { fragLen = fl; lambda = la; factor = fa; }  // (genome-wide scalars: the same on every device)
          return rc;
        });
        long long dropped = 0;
        for (gx_ctx* g : S.devs.ctx) {
          long long d = 0;
          check(S, gx_saturation_dropped(g, &d), g);
          dropped += d;
        }
        if (dropped)  // saveInterval 2558-2573 warns read by read (with -v) and leaves them out of -b and of the -x average
          fprintf(stderr, "Warning! %lld alignments skipped due to overflow / underflow of the reference's 16-bit counters "
                          "(left out of the pileup as in Genrich; their -b lines and lengths were written before)\n", dropped);
        if (i && o.verbose) {
          fprintf(stderr, "  Background pileup value: %f\n", lambda);
          fprintf(stderr, "  Scaling factor for control pileup: %f\n", factor);
          if (factor > 5.0f) fprintf(stderr, "  ** Warning! Large scaling may mask true signal **\n");
        }
      }
    }
    if (S.gx) onEachDevice(S, [&](int g) { return gx_pvalues(S.devs.ctx[g]); });
    S.sample++;
  }
  if (S.bedOpt) closeOut(S.bed);
  if (S.dupsVerb) closeOut(S.dups);
  if (getenv("GENRICH_HOST_PROF")) { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); fprintf(stderr, "main thread cpu %.3f s\n", ts.tv_sec + ts.tv_nsec * 1e-9); }
  if (o.eventsOnly) return EXIT_SUCCESS;

  size_t nPeaks = 0;
  uint64_t genomeLen = 0, peakBP = 0;
  {
    std::vector<size_t> np(S.devs.n(), 0);
    std::vector<uint64_t> bp(S.devs.n(), 0);
    onEachDevice(S, [&](int g) { return gx_find_peaks(S.devs.ctx[g], &np[g], g == 0 ? &genomeLen : nullptr, &bp[g]); });
    for (size_t g = 0; g < S.devs.n(); g++) { nPeaks += np[g]; peakBP += bp[g]; }
  }
  if (o.verbose) {  // findPeaks 1103-1117, 1130-1132
    if (o.peaksOpt) {
      fprintf(stderr, "Peak-calling parameters:\n");
      fprintf(stderr, "  Genome length: %ldbp\n", (long)genomeLen);
      fprintf(stderr, "  Significance threshold: -log(%c) > %.3f\n", o.qvalOpt ? 'q' : 'p', thr);
      fprintf(stderr, "  Min. AUC: %.3f\n", o.minAUC);
      if (o.minLen) fprintf(stderr, "  Min. peak length: %dbp\n", o.minLen);
      fprintf(stderr, "  Max. gap between sites: %dbp\n", o.maxGap);
    } else {
      fprintf(stderr, "- peak-calling skipped -\n");
      fprintf(stderr, "  Genome length: %ldbp\n", (long)genomeLen);
    }
  }
  std::vector<const char*> names;
  for (auto& ch : S.chrom) names.push_back(ch.name.c_str());
  const int nChrom = (int)names.size();
  if (o.pileFile) {
    Out pile = openWrite(o.pileFile, o.gzOut);
    for (int r = 0; r < S.sample; r++) {
      const char* cn = !o.ctrlFile ? nullptr : ((size_t)r < cFiles.size() ? cFiles[r].c_str() : nullptr);
      check(S, gx_write_pile_group(S.devs.ctx.data(), S.devs.owner.data(), r, names.data(), nChrom, tFiles[r].c_str(), cn, pile.f));
    }
    closeOut(pile);
  }
  if (o.peaksOpt) {
    Out out = openWrite(o.outFile, o.gzOut);
    check(S, gx_write_narrowpeak_group(S.devs.ctx.data(), (int)S.devs.n(), names.data(), out.f));
    closeOut(out);
    if (o.verbose) fprintf(stderr, "Peaks identified: %d (%ldbp)\n", (int)nPeaks, (long)peakBP);
  }
  if (o.logFile) {
    Out log = openWrite(o.logFile, o.gzOut);
    check(S, gx_write_log_group(S.devs.ctx.data(), S.devs.owner.data(), S.sample, names.data(), nChrom, o.qvalOpt, o.peaksOpt, thr,
                                log.f));
    closeOut(log);
  }
  for (gx_ctx* g : S.devs.ctx) gx_destroy(g);
  return EXIT_SUCCESS;
}
