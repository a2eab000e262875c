// bgzf_reader.h -- one gz-transparent input byte stream for the host program.
//
// SAM / BAM tokenising is most of the reference's wall time, and most of that is zlib's inflate
// (SURVEY.md 6.2).  BAM, and bgzip-ped SAM, are BGZF: a concatenation of independent gzip members
// of at most 64 KiB (SAM spec 4.1), so the members are inflated by a pool of threads while the
// single-threaded parser consumes them in order.  Anything else (plain text, ordinary gzip, stdin)
// goes through zlib's gz* layer exactly as the reference reads it (openRead 5125).
#pragma once
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gx_crc32.h"
#include "gx_inflate.h"

namespace gxhost {

class Input {
 public:
  Input() = default;
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;
  ~Input() { close(); }

  // false: cannot open (or empty).  threads <= 1 or not a BGZF file: zlib's own reader
  bool open(const char* path, int threads) {
    name_ = path;
    const bool isStdin = !strcmp(path, "-");
    if (!isStdin) {
      f_ = fopen(path, "rb");
      if (!f_) return false;
      if (fgetc(f_) == EOF) {  // an empty file counts as one that cannot be opened (openRead 5145-5147)
        fclose(f_);
        f_ = nullptr;
        return false;
      }
      rewind(f_);
      if (threads > 1 && looksLikeBgzf()) {
        mapFile();
        startWorkers(threads);
        return true;
      }
      // plain text (not gzip: zlib's reader would hand the bytes through unchanged): served from a mapping --
      // a line is one memchr and one memcpy (gzgets: ~2 GB/s, which the parallel record decoding outruns)
      {
        uint8_t h2[2];
        const size_t k = fread(h2, 1, 2, f_);
        rewind(f_);
        if (!(k == 2 && h2[0] == 0x1f && h2[1] == 0x8b)) {
          mapFile();
          if (map_) {
            pmap_ = map_;
            plen_ = mapLen_;
            map_ = nullptr;
          }
        }
      }
      fclose(f_);
      f_ = nullptr;
      if (pmap_) return true;
    }
    gz_ = isStdin ? gzdopen(fileno(stdin), "rb") : gzopen(path, "rb");
    if (!gz_) return false;
    gzbuffer(gz_, 1 << 20);
    compressed_ = !gzdirect(gz_);
    if (isStdin && !compressed_) {
      int c = gzgetc(gz_);
      if (c < 0) return false;
      gzungetc(c, gz_);
    }
    return true;
  }

  // the stream is gzip-compressed (the reference refuses that on stdin, openRead 5158-5160)
  bool compressed() const { return compressed_ || f_ != nullptr; }

  // Two-pass reading of a stream that cannot be reopened (stdin): everything handed out while
  // recording is kept, and replay() puts it back in front of the stream.
  void record() { recording_ = true; }
  void replay() {
    rec_.append(pre_, prePos_, std::string::npos);
    pre_.swap(rec_);
    rec_.clear();
    prePos_ = 0;
    recording_ = false;
  }

  // hand back the n bytes that the last read() delivered (a format sniff)
  void unread(const void* p, size_t n) {
    pre_.replace(0, prePos_, static_cast<const char*>(p), n);
    prePos_ = 0;
    if (recording_) rec_.resize(rec_.size() - std::min(n, rec_.size()));
  }

  const std::string& name() const { return name_; }
  const std::string& error() const { return err_; }
  bool parallel() const { return f_ != nullptr; }

  // up to n bytes; fewer only at the end of the stream (or on error: error() is then non-empty)
  size_t read(void* dst, size_t n) {
    uint8_t* d = static_cast<uint8_t*>(dst);
    size_t got = 0;
    if (prePos_ < pre_.size()) {
      got = std::min(n, pre_.size() - prePos_);
      memcpy(d, pre_.data() + prePos_, got);
      prePos_ += got;
    }
    if (got < n) got += readStream(d + got, n - got);
    if (recording_) rec_.append(reinterpret_cast<const char*>(d), got);
    return got;
  }

  // like gzgets: at most size-1 characters, through the first newline; nullptr at the end
  char* gets(char* buf, int size) {
    if (size <= 1) return nullptr;
    int got = 0;
    bool done = false;
    if (prePos_ < pre_.size()) {
      const char* p = pre_.data() + prePos_;
      size_t avail = std::min(pre_.size() - prePos_, (size_t)(size - 1));
      const void* nl = memchr(p, '\n', avail);
      size_t k = nl ? (size_t)(static_cast<const char*>(nl) - p) + 1 : avail;
      memcpy(buf, p, k);
      prePos_ += k;
      got = (int)k;
      buf[got] = '\0';
      done = nl || got == size - 1;
    }
    if (!done && !getsStream(buf + got, size - got) && !got) return nullptr;
    if (recording_) rec_.append(buf);
    return buf;
  }

  bool skip(size_t n) {
    if (recording_ || prePos_ < pre_.size()) {
      uint8_t tmp[4096];
      while (n) {
        size_t k = std::min(n, sizeof tmp);
        if (read(tmp, k) != k) return false;
        n -= k;
      }
      return true;
    }
    return skipStream(n);
  }

 private:
  size_t readStream(uint8_t* d, size_t n) {
    if (pmap_) {
      const size_t k = std::min(n, plen_ - ppos_);
      memcpy(d, pmap_ + ppos_, k);
      ppos_ += k;
      return k;
    }
    if (gz_) {
      int k = gzread(gz_, d, (unsigned)n);
      return k < 0 ? 0 : (size_t)k;
    }
    size_t got = 0;
    while (got < n) {
      if (!ensure()) break;
      size_t k = std::min(n - got, cur_->outLen - pos_);
      memcpy(d + got, cur_->out.get() + pos_, k);
      pos_ += k;
      got += k;
    }
    return got;
  }

  char* getsStream(char* buf, int size) {
    if (gz_) return gzgets(gz_, buf, size);
    if (size <= 1) return nullptr;
    if (pmap_) {
      if (ppos_ == plen_) return nullptr;
      const uint8_t* p = pmap_ + ppos_;
      const size_t avail = std::min(plen_ - ppos_, (size_t)(size - 1));
      const void* nl = memchr(p, '\n', avail);
      const size_t k = nl ? (size_t)(static_cast<const uint8_t*>(nl) - p) + 1 : avail;
      memcpy(buf, p, k);
      ppos_ += k;
      buf[k] = '\0';
      return buf;
    }
    int got = 0;
    while (got < size - 1) {
      if (!ensure()) break;
      const uint8_t* p = cur_->out.get() + pos_;
      size_t avail = std::min<size_t>(cur_->outLen - pos_, (size_t)(size - 1 - got));
      const void* nl = memchr(p, '\n', avail);
      size_t k = nl ? (size_t)(static_cast<const uint8_t*>(nl) - p) + 1 : avail;
      memcpy(buf + got, p, k);
      pos_ += k;
      got += (int)k;
      if (nl) break;
    }
    if (!got) return nullptr;
    buf[got] = '\0';
    return buf;
  }

  bool skipStream(size_t n) {
    if (gz_) return gzseek(gz_, (z_off_t)n, SEEK_CUR) != -1;
    if (pmap_) {
      ppos_ += std::min(n, plen_ - ppos_);
      return true;
    }
    while (n) {
      if (!ensure()) return false;
      size_t k = std::min(n, cur_->outLen - pos_);
      pos_ += k;
      n -= k;
    }
    return true;
  }

 public:
  // Zero-copy access for record-structured input: a pointer to the next n bytes when they lie inside
  // the current inflated block (nullptr otherwise, or when zlib's gz* reader is in use: the caller then
  // falls back to read()).  Valid until the bytes have been advance()d over and the next call is made.
  const uint8_t* peek(size_t n) {
    if (gz_ || pmap_ || prePos_ < pre_.size() || recording_ || !ensure()) return nullptr;
    return cur_->outLen - pos_ >= n ? cur_->out.get() + pos_ : nullptr;
  }
  void advance(size_t n) { pos_ += n; }
  // keeps the inflated member that the last peek()'s pointer lies in alive for as long as the handle lives (records
  // decoded by other threads straight out of the inflated data, without a copy)
  std::shared_ptr<const void> holdCurrent() const { return cur_; }
  // (which member that is, without the reference count's atomic: a caller that takes many records out of one member asks
  // for the handle once per member)
  const void* currentId() const { return cur_.get(); }
  // bytes of the current member behind the read position (after a successful peek())
  size_t leftInCurrent() const { return cur_ ? cur_->outLen - pos_ : 0; }

  // A plain mapped file whose replay buffer is exhausted: the rest of the stream as one span of memory, for readers
  // that cut it up themselves (parallel record decoding).  plainTake(n) moves the stream past n bytes of it.
  bool plainDirect() const { return pmap_ && prePos_ >= pre_.size() && !recording_; }
  const uint8_t* plainPtr() const { return pmap_ + ppos_; }
  size_t plainLeft() const { return plen_ - ppos_; }
  void plainTake(size_t n) { ppos_ += std::min(n, plen_ - ppos_); }

  void close() {
    if (gz_) {
      gzclose(gz_);
      gz_ = nullptr;
    }
    if (pmap_) {
      munmap(const_cast<uint8_t*>(pmap_), plen_);
      pmap_ = nullptr;
      plen_ = ppos_ = 0;
    }
    if (f_) {
      {
        std::lock_guard<std::mutex> lk(m_);
        quit_ = true;
      }
      cvWork_.notify_all();
      for (auto& t : workers_) t.join();
      workers_.clear();
      queue_.clear();
      todo_.clear();
      cur_.reset();
      if (map_) munmap(const_cast<uint8_t*>(map_), mapLen_);
      map_ = nullptr;
      fclose(f_);
      f_ = nullptr;
    }
  }

 private:
  struct Block {
    const uint8_t* comp = nullptr;  // deflate payload + crc32 + isize: inside the mapping, or `own`
    size_t compLen = 0;
    std::unique_ptr<uint8_t[]> own;
    std::unique_ptr<uint8_t[]> out;  // (plain new[]: no zero fill of 64 KiB per block)
    size_t outLen = 0;
    bool done = false;
    std::string err;
  };

  // a regular file is mapped, so that the members go to the workers without a copy; anything that
  // cannot be mapped is read with fread
  void mapFile() {
    struct stat st;
    if (fstat(fileno(f_), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return;
    if (getenv("GENRICH_NO_MMAP")) return;  // (a file that may be truncated or rewritten while it is read: the read path reports a short read, a mapping dies of SIGBUS)
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f_), 0);
    if (m == MAP_FAILED) return;
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    map_ = static_cast<const uint8_t*>(m);
    mapLen_ = (size_t)st.st_size;
  }

  bool looksLikeBgzf() {
    uint8_t h[18];
    size_t k = fread(h, 1, sizeof h, f_);
    rewind(f_);
    return k == sizeof h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' &&
           h[14] == 2 && h[15] == 0;
  }

  size_t members_ = 0;  // members handed to the pool so far

  void startWorkers(int threads) {
    depth_ = (size_t)threads * 4 + 4;
    for (int i = 0; i < threads; i++) workers_.emplace_back([this] { work(); });
  }

 public:
  // more inflaters for a stream that turned out to need them (BAM: inflate is ~90 % of its ingest); only when the
  // members already go through the pool
  void growWorkers(int total) {
    if (workers_.empty()) return;
    const int have = (int)workers_.size();
    if (total > have) {
      std::lock_guard<std::mutex> lk(m_);
      depth_ = (size_t)total * 4 + 4;
    }
    for (int i = have; i < total; i++) workers_.emplace_back([this] { work(); });
  }

 private:

  // take the next member of the file and hand it to the pool; false at the end of the file
  bool enqueue() {
    uint8_t hbuf[12];
    const uint8_t* h;
    // zlib's gz* reader, through which the reference reads, stops quietly at bytes that do not begin another
    // gzip member once it has read one (trailing garbage after a valid file): end of stream here too
    if (map_) {
      if (mapPos_ == mapLen_) return false;
      if (mapLen_ - mapPos_ < 12) return members_ ? false : fail("not a BGZF block");
      h = map_ + mapPos_;
    } else {
      size_t k = fread(hbuf, 1, 12, f_);
      if (k == 0) return false;
      if (k != 12) return members_ ? false : fail("not a BGZF block");
      h = hbuf;
    }
    if (h[0] != 0x1f || h[1] != 0x8b) return members_ ? false : fail("not a BGZF block");
    if (h[2] != 8 || !(h[3] & 4)) return fail("not a BGZF block");
    members_++;
    const size_t xlen = h[10] | (h[11] << 8);
    std::vector<uint8_t> xbuf;
    const uint8_t* extra;
    if (map_) {
      if (mapLen_ - mapPos_ - 12 < xlen) return fail("truncated BGZF block");
      extra = h + 12;
    } else {
      xbuf.resize(xlen);
      if (fread(xbuf.data(), 1, xlen, f_) != xlen) return fail("truncated BGZF block");
      extra = xbuf.data();
    }
    long bsize = -1;
    for (size_t o = 0; o + 4 <= xlen;) {
      const size_t slen = extra[o + 2] | (extra[o + 3] << 8);
      if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1;
      o += 4 + slen;
    }
    if (bsize < (long)(12 + xlen + 8)) return fail("BGZF block without a size field");
    auto b = std::make_shared<Block>();
    b->compLen = (size_t)bsize - 12 - xlen;
    if (map_) {
      if (mapLen_ - mapPos_ < (size_t)bsize) return fail("truncated BGZF block");
      b->comp = h + 12 + xlen;
      mapPos_ += (size_t)bsize;
    } else {
      b->own.reset(new uint8_t[b->compLen]);
      if (fread(b->own.get(), 1, b->compLen, f_) != b->compLen) return fail("truncated BGZF block");
      b->comp = b->own.get();
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      queue_.push_back(b);
      todo_.push_back(b);
    }
    cvWork_.notify_one();
    return true;
  }

  bool fail(const char* what) {
    err_ = what;
    eof_ = true;
    return false;
  }

  // make cur_ a block with unread bytes; false at the end of the stream
  bool ensure() {
    for (;;) {
      if (cur_ && pos_ < cur_->outLen) return true;
      while (!eof_ && queue_.size() < depth_)
        if (!enqueue()) eof_ = true;
      if (queue_.empty()) return false;
      std::shared_ptr<Block> b = queue_.front();
      {
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [&] { return b->done; });
        queue_.pop_front();
      }
      if (!b->err.empty()) {
        err_ = b->err;
        queue_.clear();
        eof_ = true;
        cur_.reset();
        return false;
      }
      cur_ = b;
      pos_ = 0;
    }
  }

  void work() {
    for (;;) {
      std::shared_ptr<Block> b;
      {
        std::unique_lock<std::mutex> lk(m_);
        cvWork_.wait(lk, [&] { return quit_ || !todo_.empty(); });
        if (todo_.empty()) return;  // quit
        b = todo_.front();
        todo_.pop_front();
      }
      inflateBlock(*b);
      {
        std::lock_guard<std::mutex> lk(m_);
        b->done = true;
      }
      cvDone_.notify_all();
    }
  }

  static void inflateBlock(Block& b) {
    if (b.compLen < 8) { b.err = "truncated BGZF block"; return; }
    const uint8_t* tail = b.comp + b.compLen - 8;
    const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    if (isize > 65536) { b.err = "BGZF block larger than 64 KiB"; return; }
    b.out.reset(new uint8_t[isize ? isize : 1]);
    // gx_inflate.h: the whole member in, the whole block out, in one call (1.5-2 x zlib's streaming inflate, which was
    // ~90 % of a BAM file's ingest).  Whatever it does not accept -- a damaged stream, or one of the few legal oddities
    // it leaves alone -- goes through zlib, whose verdict stands.
    static const bool zlibOnly = getenv("GENRICH_ZLIB_INFLATE") != nullptr;
    if (zlibOnly || !gxinf::inflate(b.comp, b.compLen - 8, b.out.get(), isize)) {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { b.err = "zlib initialisation failed"; return; }
      zs.next_in = const_cast<Bytef*>(b.comp);
      zs.avail_in = (uInt)(b.compLen - 8);
      zs.next_out = b.out.get();
      zs.avail_out = (uInt)isize;
      int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.total_out != isize) { b.err = "corrupt BGZF block"; return; }
    }
    if (gxcrc::crc32_of(b.out.get(), isize) != crc) { b.err = "BGZF checksum mismatch"; return; }
    b.outLen = isize;
    b.own.reset();
  }

  std::string name_, err_;
  std::string pre_, rec_;  // replay buffer (served first) and the recording that will become one
  size_t prePos_ = 0;
  bool recording_ = false, compressed_ = false;
  gzFile gz_ = nullptr;
  FILE* f_ = nullptr;
  const uint8_t* map_ = nullptr;
  size_t mapLen_ = 0, mapPos_ = 0;
  const uint8_t* pmap_ = nullptr;   // a plain (uncompressed) regular file, mapped
  size_t plen_ = 0, ppos_ = 0;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cvWork_, cvDone_;
  std::deque<std::shared_ptr<Block>> queue_;  // in file order, owned by the consumer
  std::deque<std::shared_ptr<Block>> todo_;   // waiting for a worker (guarded by m_)
  std::shared_ptr<Block> cur_;
  size_t pos_ = 0, depth_ = 0;
  bool eof_ = false, quit_ = false;
};

}  // namespace gxhost
