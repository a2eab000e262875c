// gx_inflate.h -- raw DEFLATE (RFC 1951) decoding of ONE complete member into a buffer of known size: what a BGZF block
// is (SAM spec 4.1: at most 64 KiB of data, its length in the trailer).  zlib's streaming inflate is ~90 % of the CPU
// time of BAM ingest (bgzf_reader.h: one call per block on a pool of threads); a decoder that knows the whole input and
// the whole output are in memory does the same work with a 64-bit bit buffer refilled once per match, table entries
// that carry base value and extra-bit count, and 8-byte match copies.  Host only, no dependencies.
//
//   bool gxinf::inflate(in, inLen, out, outLen)
//     true: the stream was well-formed up to and including its final block and produced exactly outLen bytes.
//     false: anything else (the caller falls back to zlib, which then reports the error in its own words) -- never
//     reads outside [in, in + inLen), never writes outside [out, out + outLen).
//
// Layout of a table entry (uint32_t), litlen and offset tables alike:
//   [7:0]    bits to take from the bit buffer for the whole item: codeword + extra bits
//   [15:8]   bits of the codeword alone
//   [30:16]  literal byte | base of the length / offset | first entry of a subtable
//   [31]     HUFF_SUB: the codeword is longer than the primary table's index: look in the subtable
//   kind, in [15:12] (a codeword has at most 15 bits: [11:8] hold its length, [15:12] are free):
//            1 literal, 2 length / offset, 3 end of block, 0 invalid (no codeword has this prefix)
#pragma once
#include <stdint.h>
#include <string.h>
#include <cstddef>

namespace gxinf {

constexpr int LL_BITS = 11, LL_SUB = 15 - LL_BITS;   // primary index bits of the literal/length table; its subtables' index bits
constexpr int OF_BITS = 8, OF_SUB = 15 - OF_BITS;
constexpr int LL_MAX = (1 << LL_BITS) + 286 * (1 << LL_SUB);   // every long codeword could open a subtable of its own
constexpr int OF_MAX = (1 << OF_BITS) + 30 * (1 << OF_SUB);
constexpr uint32_t HUFF_SUB = 0x80000000u;
constexpr uint32_t K_LIT = 1u << 12, K_BASE = 2u << 12, K_END = 3u << 12, K_MASK = 15u << 12;

struct Tables {
  uint32_t ll[LL_MAX];
  uint32_t of[OF_MAX];
};

inline uint32_t rev_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

// Canonical Huffman code of lens[0 .. n) (0: symbol unused) -> decode table with `bits` primary index bits and subtables
// of `sub` bits.  item(sym) gives the entry's [31:16] value, [15:12] kind and the extra bits ([7:0], to which the
// codeword's length is added).  false: over-subscribed, or incomplete in a way DEFLATE does not allow.
template <class Item>
inline bool build(const uint8_t* lens, int n, int bits, int sub, uint32_t* tab, int cap, bool allowIncomplete, Item item) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[lens[i]]++;
  if (count[0] == n) {  // no codeword at all: every lookup is invalid (legal for the offset code of a block without matches)
    if (!allowIncomplete) return false;
    for (int i = 0; i < (1 << bits); i++) tab[i] = 0;
    return true;
  }
  int left = 1;
  for (int len = 1; len <= 15; len++) {
    left = (left << 1) - count[len];
    if (left < 0) return false;  // over-subscribed
  }
  if (left > 0) {
    // incomplete: only a single codeword of one bit is allowed (an offset code with one symbol)
    if (!allowIncomplete || count[1] != 1 || n - count[0] != 1) return false;
  }
  uint32_t next[16];  // first codeword of every length (RFC 1951, 3.2.2)
  {
    uint32_t code = 0;
    for (int len = 1; len <= 15; len++) {
      next[len] = code;
      code = (code + (uint32_t)count[len]) << 1;
    }
  }
  const int prim = 1 << bits;
  for (int i = 0; i < prim; i++) tab[i] = 0;
  int used = prim;
  for (int sym = 0; sym < n; sym++) {
    const int len = lens[sym];
    if (!len) continue;
    const uint32_t code = rev_bits(next[len]++, len);  // as the bits arrive: LSB first
    const uint32_t it = item(sym);
    const uint32_t entry = (it & 0xFFFFF000u) | ((uint32_t)len << 8) | ((it & 0xFFu) + (uint32_t)len);
    if (len <= bits) {
      for (uint32_t i = code; i < (uint32_t)prim; i += 1u << len) tab[i] = entry;
    } else {
      const uint32_t p = code & (uint32_t)(prim - 1);
      if (!(tab[p] & HUFF_SUB)) {
        if (used + (1 << sub) > cap) return false;
        tab[p] = HUFF_SUB | ((uint32_t)used << 16);
        for (int i = 0; i < (1 << sub); i++) tab[used + i] = 0;
        used += 1 << sub;
      }
      const uint32_t start = (tab[p] >> 16) & 0x7FFFu;
      for (uint32_t i = code >> bits; i < (1u << sub); i += 1u << (len - bits)) tab[start + i] = entry;
    }
  }
  return true;
}

struct Bits {
  const uint8_t* in;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;          // valid bits in buf
  // at least `n` (<= 56) bits, or as many as the input still has
  inline void need(int n) {
    while (cnt < n && in < end) {
      buf |= (uint64_t)*in++ << cnt;
      cnt += 8;
    }
  }
  // with at least 8 bytes of input left: top up to 56..63 bits in one load
  inline void fill8() {
    uint64_t w;
    memcpy(&w, in, 8);
    buf |= w << cnt;
    in += (63 - cnt) >> 3;
    cnt |= 56;
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_XTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t OFF_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t OFF_XTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t ll_item(int sym) {
  if (sym < 256) return ((uint32_t)sym << 16) | K_LIT;
  if (sym == 256) return K_END;
  if (sym > 285) return 0;  // 286, 287: in the fixed code, never valid in data
  return ((uint32_t)LEN_BASE[sym - 257] << 16) | K_BASE | LEN_XTRA[sym - 257];
}
inline uint32_t of_item(int sym) {
  if (sym > 29) return 0;
  return ((uint32_t)OFF_BASE[sym] << 16) | K_BASE | OFF_XTRA[sym];
}

inline bool build_block_tables(const uint8_t* llLens, int nll, const uint8_t* ofLens, int nof, Tables& T) {
  return build(llLens, nll, LL_BITS, LL_SUB, T.ll, LL_MAX, false, ll_item) &&
         build(ofLens, nof, OF_BITS, OF_SUB, T.of, OF_MAX, true, of_item);
}

// the lengths of a dynamic block's two codes (RFC 1951, 3.2.7)
inline bool read_dynamic(Bits& B, Tables& T) {
  B.need(14);
  if (B.cnt < 14) return false;
  const int hlit = (int)B.peek(5) + 257; B.drop(5);
  const int hdist = (int)B.peek(5) + 1; B.drop(5);
  const int hclen = (int)B.peek(4) + 4; B.drop(4);
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t pl[19] = {0};
  for (int i = 0; i < hclen; i++) {
    B.need(3);
    if (B.cnt < 3) return false;
    pl[order[i]] = (uint8_t)B.peek(3);
    B.drop(3);
  }
  uint32_t pre[1 << 7];
  if (!build(pl, 19, 7, 0, pre, 1 << 7, false, [](int sym) { return ((uint32_t)sym << 16) | K_LIT; })) {
    // (a precode with a single codeword is accepted by zlib only when complete; leave such streams to it)
    return false;
  }
  uint8_t lens[286 + 30 + 138];
  int n = 0;
  const int total = hlit + hdist;
  while (n < total) {
    B.need(7 + 7);
    const uint32_t e = pre[B.peek(7)];
    if ((e & K_MASK) != K_LIT) return false;
    const int clen = (int)((e >> 8) & 15u);
    if (B.cnt < clen) return false;
    B.drop(clen);
    const int sym = (int)(e >> 16);
    if (sym < 16) {
      lens[n++] = (uint8_t)sym;
      continue;
    }
    int rep, xb, base;
    uint8_t v = 0;
    if (sym == 16) {
      if (n == 0) return false;
      v = lens[n - 1];
      xb = 2; base = 3;
    } else if (sym == 17) {
      xb = 3; base = 3;
    } else {
      xb = 7; base = 11;
    }
    if (B.cnt < xb) return false;
    rep = base + (int)B.peek(xb);
    B.drop(xb);
    if (n + rep > total) return false;
    for (int i = 0; i < rep; i++) lens[n++] = v;
  }
  if (lens[256] == 0) return false;  // no end-of-block codeword
  return build_block_tables(lens, hlit, lens + hlit, hdist, T);
}

inline bool fixed_tables(Tables& T) {
  uint8_t ll[288], of[32];
  for (int i = 0; i < 144; i++) ll[i] = 8;
  for (int i = 144; i < 256; i++) ll[i] = 9;
  for (int i = 256; i < 280; i++) ll[i] = 7;
  for (int i = 280; i < 288; i++) ll[i] = 8;
  for (int i = 0; i < 32; i++) of[i] = 5;
  return build(ll, 288, LL_BITS, LL_SUB, T.ll, LL_MAX, false, ll_item) && build(of, 32, OF_BITS, OF_SUB, T.of, OF_MAX, false, of_item);
}

inline bool inflate(const uint8_t* in, size_t inLen, uint8_t* out, size_t outLen) {
  Bits B{in, in + inLen};
  uint8_t* o = out;
  uint8_t* const oend = out + outLen;
  Tables T;
  for (;;) {
    B.need(3);
    if (B.cnt < 3) return false;
    const uint32_t last = B.peek(1);
    const uint32_t type = (B.peek(3) >> 1);
    B.drop(3);
    if (type == 0) {  // stored: skip to the byte boundary; LEN, ~LEN, bytes
      B.drop(B.cnt & 7);
      B.need(32);
      if (B.cnt < 32) return false;
      const uint32_t len = B.peek(16);
      B.drop(16);
      const uint32_t nlen = B.peek(16);
      B.drop(16);
      if ((len ^ nlen) != 0xFFFFu) return false;
      // whole bytes still in the bit buffer belong to the data
      uint32_t left = len;
      while (left && B.cnt >= 8) {
        if (o == oend) return false;
        *o++ = (uint8_t)B.peek(8);
        B.drop(8);
        left--;
      }
      if (left) {
        if ((size_t)(B.end - B.in) < left || (size_t)(oend - o) < left) return false;
        memcpy(o, B.in, left);
        B.in += left;
        o += left;
        B.buf = 0;  // (what lay above the valid bits was a look-ahead at the bytes just copied)
        B.cnt = 0;
      }
    } else if (type == 1 || type == 2) {
      if (type == 1) {
        // (the fixed code's tables are the same for every block: built once per process, copied -- 43 KB -- instead of
        // rebuilt entry by entry; thread-safe: a function-local static is initialised once)
        static const struct Fixed { Tables T; bool ok; Fixed() { ok = fixed_tables(T); } } fixed;
        if (!fixed.ok) return false;
        T = fixed.T;
      } else if (!read_dynamic(B, T))
        return false;
      // ---- the block's data
      auto lookup = [&]() -> uint32_t {
        uint32_t e = T.ll[B.peek(LL_BITS)];
        if (e & HUFF_SUB) e = T.ll[((e >> 16) & 0x7FFFu) + ((B.buf >> LL_BITS) & ((1u << LL_SUB) - 1u))];
        return e;
      };
      for (;;) {
        // ---- fast: far from both ends.  One refill (>= 56 bits) covers three literals (<= 45 bits) or a length / offset
        // pair (<= 20 + 28 bits); two refills advance the input by at most 14 bytes and read 8 at a time (16 in hand);
        // three literals, a match of 258 bytes and the copy's 8-byte overshoot stay inside the output (274 in hand).
        // Whatever is not a literal or a well-formed match goes to the checked code below, its bits still in the buffer.
        while (B.end - B.in >= 16 && oend - o >= 274) {
          B.fill8();
          uint32_t f = lookup();
          if ((f & K_MASK) == K_LIT) {
            B.drop((int)(f & 0xFFu));
            *o++ = (uint8_t)(f >> 16);
            f = lookup();
            if ((f & K_MASK) == K_LIT) {
              B.drop((int)(f & 0xFFu));
              *o++ = (uint8_t)(f >> 16);
              f = lookup();
              if ((f & K_MASK) == K_LIT) {
                B.drop((int)(f & 0xFFu));
                *o++ = (uint8_t)(f >> 16);
                continue;
              }
            }
            B.fill8();
          }
          if ((f & K_MASK) != K_BASE) break;
          const int ftotal = (int)(f & 0xFFu), fclen = (int)((f >> 8) & 15u);
          const uint32_t flen = ((f >> 16) & 0x7FFFu) + (uint32_t)((B.buf >> fclen) & ((1ull << (ftotal - fclen)) - 1ull));
          uint32_t d = T.of[(uint32_t)(B.buf >> ftotal) & ((1u << OF_BITS) - 1u)];
          if (d & HUFF_SUB) d = T.of[((d >> 16) & 0x7FFFu) + ((B.buf >> (ftotal + OF_BITS)) & ((1u << OF_SUB) - 1u))];
          if ((d & K_MASK) != K_BASE) break;  // (the checked code meets it again and fails)
          B.drop(ftotal);
          const int dtotal = (int)(d & 0xFFu), dclen = (int)((d >> 8) & 15u);
          const uint32_t dist = ((d >> 16) & 0x7FFFu) + (uint32_t)((B.buf >> dclen) & ((1ull << (dtotal - dclen)) - 1ull));
          B.drop(dtotal);
          if (dist > (size_t)(o - out)) return false;
          const uint8_t* s = o - dist;
          uint8_t* const stop = o + flen;
          if (dist >= 8) {
            do {
              uint64_t w;
              memcpy(&w, s, 8);
              memcpy(o, &w, 8);
              s += 8;
              o += 8;
            } while (o < stop);
          } else if (dist == 1) {
            uint64_t w = 0x0101010101010101ull * *s;
            do {
              memcpy(o, &w, 8);
              o += 8;
            } while (o < stop);
          } else {
            do *o++ = *s++; while (o < stop);
          }
          o = stop;
        }
        // ---- checked: near the end of the input or of the output, or an item the fast loop left alone
        B.need(48);
        const uint32_t e = lookup();
        const uint32_t kind = e & K_MASK;
        const int total = (int)(e & 0xFFu), clen = (int)((e >> 8) & 15u);
        if (kind == 0 || B.cnt < total) return false;
        const uint64_t saved = B.buf;
        B.drop(total);
        if (kind == K_LIT) {
          if (o == oend) return false;
          *o++ = (uint8_t)(e >> 16);
          continue;
        }
        if (kind == K_END) break;
        const uint32_t len = ((e >> 16) & 0x7FFFu) + (uint32_t)((saved >> clen) & ((1ull << (total - clen)) - 1ull));
        uint32_t d = T.of[B.peek(OF_BITS)];
        if (d & HUFF_SUB) d = T.of[((d >> 16) & 0x7FFFu) + ((B.buf >> OF_BITS) & ((1u << OF_SUB) - 1u))];
        const int dtotal = (int)(d & 0xFFu), dclen = (int)((d >> 8) & 15u);
        if ((d & K_MASK) != K_BASE || B.cnt < dtotal) return false;
        const uint32_t dist = ((d >> 16) & 0x7FFFu) + (uint32_t)((B.buf >> dclen) & ((1ull << (dtotal - dclen)) - 1ull));
        B.drop(dtotal);
        if (dist > (size_t)(o - out) || len > (size_t)(oend - o)) return false;
        const uint8_t* s = o - dist;
        if (oend - o >= (ptrdiff_t)len + 8 && dist >= 8) {
          uint8_t* const stop = o + len;
          do {
            uint64_t w;
            memcpy(&w, s, 8);
            memcpy(o, &w, 8);
            s += 8;
            o += 8;
          } while (o < stop);
          o = stop;
        } else if (dist == 1) {
          memset(o, *s, len);
          o += len;
        } else {
          for (uint32_t i = 0; i < len; i++) o[i] = s[i];
          o += len;
        }
      }
    } else
      return false;
    if (last) break;
  }
  return o == oend;
}

}  // namespace gxinf
