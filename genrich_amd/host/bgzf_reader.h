// bgzf_reader.h -- one gz-transparent input byte stream for the host program.
//
// SAM / BAM tokenising is most of the reference's wall time, and most of that is zlib's inflate
// (SURVEY.md 6.2).  BAM, and bgzip-ped SAM, are BGZF: a concatenation of independent gzip members
// of at most 64 KiB (SAM spec 4.1), so the members are inflated by a pool of threads while the
// single-threaded parser consumes them in order.  Anything else (plain text, ordinary gzip, stdin)
// goes through zlib's gz* layer exactly as the reference reads it (openRead 5125).
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace gxhost {

class Input {
 public:
  Input() = default;
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;
  ~Input() { close(); }

  // false: cannot open.  threads <= 1 or not a BGZF file: zlib's own reader
  bool open(const char* path, int threads) {
    name_ = path;
    if (strcmp(path, "-") && threads > 1) {
      f_ = fopen(path, "rb");
      if (!f_) return false;
      if (looksLikeBgzf()) {
        startWorkers(threads);
        return true;
      }
      fclose(f_);
      f_ = nullptr;
    }
    gz_ = !strcmp(path, "-") ? gzdopen(fileno(stdin), "rb") : gzopen(path, "rb");
    if (!gz_) return false;
    gzbuffer(gz_, 1 << 20);
    return true;
  }

  const std::string& name() const { return name_; }
  const std::string& error() const { return err_; }
  bool parallel() const { return f_ != nullptr; }

  // up to n bytes; fewer only at the end of the stream (or on error: error() is then non-empty)
  size_t read(void* dst, size_t n) {
    if (gz_) {
      int k = gzread(gz_, dst, (unsigned)n);
      return k < 0 ? 0 : (size_t)k;
    }
    size_t got = 0;
    uint8_t* d = static_cast<uint8_t*>(dst);
    while (got < n) {
      if (!ensure()) break;
      size_t k = std::min(n - got, cur_->outLen - pos_);
      memcpy(d + got, cur_->out.data() + pos_, k);
      pos_ += k;
      got += k;
    }
    return got;
  }

  // like gzgets: at most size-1 characters, through the first newline; nullptr at the end
  char* gets(char* buf, int size) {
    if (gz_) return gzgets(gz_, buf, size);
    if (size <= 1) return nullptr;
    int got = 0;
    while (got < size - 1) {
      if (!ensure()) break;
      const uint8_t* p = cur_->out.data() + pos_;
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

  bool skip(size_t n) {
    if (gz_) return gzseek(gz_, (z_off_t)n, SEEK_CUR) != -1;
    while (n) {
      if (!ensure()) return false;
      size_t k = std::min(n, cur_->outLen - pos_);
      pos_ += k;
      n -= k;
    }
    return true;
  }

  void close() {
    if (gz_) {
      gzclose(gz_);
      gz_ = nullptr;
    }
    if (f_) {
      {
        std::lock_guard<std::mutex> lk(m_);
        quit_ = true;
      }
      cvWork_.notify_all();
      for (auto& t : workers_) t.join();
      workers_.clear();
      fclose(f_);
      f_ = nullptr;
      queue_.clear();
      cur_.reset();
    }
  }

 private:
  struct Block {
    std::vector<uint8_t> comp;  // deflate payload + crc32 + isize
    std::vector<uint8_t> out;
    size_t outLen = 0;
    bool done = false;
    std::string err;
  };

  bool looksLikeBgzf() {
    uint8_t h[18];
    size_t k = fread(h, 1, sizeof h, f_);
    rewind(f_);
    return k == sizeof h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' &&
           h[14] == 2 && h[15] == 0;
  }

  void startWorkers(int threads) {
    depth_ = (size_t)threads * 4 + 4;
    for (int i = 0; i < threads; i++) workers_.emplace_back([this] { work(); });
  }

  // read the next member from the file and hand it to the pool; false at the end of the file
  bool enqueue() {
    uint8_t h[12];
    size_t k = fread(h, 1, 12, f_);
    if (k == 0) return false;
    if (k != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail("not a BGZF block");
    const size_t xlen = h[10] | (h[11] << 8);
    std::vector<uint8_t> extra(xlen);
    if (fread(extra.data(), 1, xlen, f_) != xlen) return fail("truncated BGZF block");
    long bsize = -1;
    for (size_t o = 0; o + 4 <= xlen;) {
      const size_t slen = extra[o + 2] | (extra[o + 3] << 8);
      if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1;
      o += 4 + slen;
    }
    if (bsize < (long)(12 + xlen + 8)) return fail("BGZF block without a size field");
    auto b = std::make_shared<Block>();
    b->comp.resize((size_t)bsize - 12 - xlen);
    if (fread(b->comp.data(), 1, b->comp.size(), f_) != b->comp.size()) return fail("truncated BGZF block");
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
    if (b.comp.size() < 8) { b.err = "truncated BGZF block"; return; }
    const uint8_t* tail = b.comp.data() + b.comp.size() - 8;
    const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    if (isize > 65536) { b.err = "BGZF block larger than 64 KiB"; return; }
    b.out.resize(isize ? isize : 1);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { b.err = "zlib initialisation failed"; return; }
    zs.next_in = const_cast<Bytef*>(b.comp.data());
    zs.avail_in = (uInt)(b.comp.size() - 8);
    zs.next_out = b.out.data();
    zs.avail_out = (uInt)isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != isize) { b.err = "corrupt BGZF block"; return; }
    if (crc32(crc32(0L, Z_NULL, 0), b.out.data(), isize) != crc) { b.err = "BGZF checksum mismatch"; return; }
    b.outLen = isize;
  }

  std::string name_, err_;
  gzFile gz_ = nullptr;
  FILE* f_ = nullptr;
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
