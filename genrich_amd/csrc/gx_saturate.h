// gx_saturate.h -- host side of the reference's int16 saturation rule (Genrich.c:2558-2573).
//
// saveInterval drops an alignment when the int16 part of its difference array already holds
// INT16_MAX at the alignment's start or INT16_MIN at its end.  Which alignments that hits depends on
// the order of the input, so it cannot be a property of the pileup: it is decided here, by replaying
// the events that touch a position able to saturate at all (>= 32,766 unit weights starting or
// ending on it) in input order.  The device only finds out whether such a position exists
// (k_hot_check); inputs without one -- practically all -- never come here.
#pragma once
#include <stdint.h>

#include <unordered_map>
#include <vector>

#include "../../include/genrich_amd.h"

namespace gxsat {

constexpr long long UNIT = 120;
constexpr long long HOT = 32766 * UNIT;  // a base below this weight of starts (ends) can reach neither limit

// the int16 `cov` of the reference's (cov, eighths, sixths, tenths) state for the exact sum v/120
// (the same canonical form as gx::getval, gx_math.h)
inline long long canon_cov(long long v) {
  long long q = v / UNIT, r = v - q * UNIT;
  if (r < 0) { r += UNIT; q -= 1; }
  if (r == 0) return q;
  const unsigned s = (unsigned)(r * 2) % 3u, t = (unsigned)(r * 3) % 5u;
  const unsigned e = (7u * ((unsigned)r + 4u * (s + t))) & 7u;
  return q - (15 * (long long)e + 20 * (long long)s + 12 * (long long)t - r) / UNIT;
}

inline int weight_of(uint32_t count) {
  switch (count) {
    case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10: return (int)(UNIT / count);
    default: return 0;
  }
}

// keep[i] = 0 for the events the reference would drop; returns how many
inline long long filter(const gx_event* ev, size_t n, int nChrom, const uint32_t* len, uint8_t* keep) {
  for (size_t i = 0; i < n; i++) keep[i] = 1;
  constexpr int TB = 12;
  std::vector<size_t> base((size_t)nChrom + 1, 0);
  for (int c = 0; c < nChrom; c++) base[c + 1] = base[c] + ((size_t)len[c] >> TB) + 1;
  auto usable = [&](const gx_event& e) { return (int)e.chrom < nChrom && e.start < len[e.chrom] && weight_of(e.count); };
  auto endOf = [&](const gx_event& e) { return e.end > len[e.chrom] ? len[e.chrom] : e.end; };
  // pass 0: weight of starts / ends per 2^12-base window
  std::vector<uint64_t> ws(base[nChrom], 0), we(base[nChrom], 0);
  for (size_t i = 0; i < n; i++) {
    const gx_event& e = ev[i];
    if (!usable(e)) continue;
    const int w = weight_of(e.count);
    ws[base[e.chrom] + (e.start >> TB)] += (uint64_t)w;
    we[base[e.chrom] + (endOf(e) >> TB)] += (uint64_t)w;
  }
  bool any = false;
  for (size_t k = 0; k < ws.size() && !any; k++) any = ws[k] >= (uint64_t)HOT || we[k] >= (uint64_t)HOT;
  if (!any) return 0;
  // pass 1: per position, inside those windows only
  struct Tally { uint64_t s = 0, e = 0; };
  std::unordered_map<uint64_t, Tally> tally;
  auto key = [](uint32_t c, uint32_t p) { return ((uint64_t)c << 32) | p; };
  for (size_t i = 0; i < n; i++) {
    const gx_event& e = ev[i];
    if (!usable(e)) continue;
    const int w = weight_of(e.count);
    const uint32_t end = endOf(e);
    if (ws[base[e.chrom] + (e.start >> TB)] >= (uint64_t)HOT) tally[key(e.chrom, e.start)].s += (uint64_t)w;
    if (we[base[e.chrom] + (end >> TB)] >= (uint64_t)HOT) tally[key(e.chrom, end)].e += (uint64_t)w;
  }
  std::unordered_map<uint64_t, long long> run;  // running difference (1/120 units) at the positions that can saturate
  for (auto& kv : tally)
    if (kv.second.s >= (uint64_t)HOT || kv.second.e >= (uint64_t)HOT) run.emplace(kv.first, 0);
  tally.clear();
  if (run.empty()) return 0;
  // pass 2: the reference's decisions, in input order
  long long dropped = 0;
  for (size_t i = 0; i < n; i++) {
    const gx_event& e = ev[i];
    if (!usable(e)) continue;
    auto a = run.find(key(e.chrom, e.start));
    auto b = run.find(key(e.chrom, endOf(e)));
    if (a == run.end() && b == run.end()) continue;
    if ((a != run.end() && canon_cov(a->second) == 32767) || (b != run.end() && canon_cov(b->second) == -32768)) {
      keep[i] = 0;
      dropped++;
      continue;
    }
    const int w = weight_of(e.count);
    if (a != run.end()) a->second += w;
    if (b != run.end()) b->second -= w;
  }
  return dropped;
}

}  // namespace gxsat
