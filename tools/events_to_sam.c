/* events_to_sam -- renders a binary fragment list as queryname-grouped paired SAM text, the way genrich_amd/synth.py's
 * write_sam does (flags 99 / 147, read length 50, names r<i>, AS:i:0), at ~10 M records a second instead of Python's
 * 0.3 M: bench.py's `e2e_cli` leg needs 2 x 10^7 records.  Unit-weight fragments only (one alignment per read).
 *
 *   events_to_sam EVENTS.bin CHROMS.txt OUT.sam
 *     EVENTS.bin  records of four little-endian u32: chrom, start, end, count (include/genrich_amd.h gx_event)
 *     CHROMS.txt  one "name length" line per chromosome, in header order
 * Test / bench tooling: not part of the library, reads nothing of the reference.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char* put_u(char* p, unsigned long v) {
  char t[24];
  int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = t[--n];
  return p;
}
static char* put_s(char* p, const char* s) {
  while (*s) *p++ = *s++;
  return p;
}

int main(int argc, char** argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: events_to_sam EVENTS.bin CHROMS.txt OUT.sam\n");
    return 2;
  }
  FILE* fc = fopen(argv[2], "r");
  if (!fc) { perror(argv[2]); return 1; }
  char (*names)[64] = NULL;
  unsigned long* lens = NULL;
  int nc = 0;
  char nm[64];
  unsigned long ln;
  while (fscanf(fc, "%63s %lu", nm, &ln) == 2) {
    names = realloc(names, (size_t)(nc + 1) * sizeof *names);
    lens = realloc(lens, (size_t)(nc + 1) * sizeof *lens);
    strcpy(names[nc], nm);
    lens[nc++] = ln;
  }
  fclose(fc);
  FILE* fe = fopen(argv[1], "rb");
  if (!fe) { perror(argv[1]); return 1; }
  FILE* fo = fopen(argv[3], "w");
  if (!fo) { perror(argv[3]); return 1; }
  static char obuf[1 << 22];
  setvbuf(fo, obuf, _IOFBF, sizeof obuf);
  fputs("@HD\tVN:1.0\tSO:queryname\n", fo);
  for (int i = 0; i < nc; i++) fprintf(fo, "@SQ\tSN:%s\tLN:%lu\n", names[i], lens[i]);
  enum { BATCH = 1 << 16 };
  static uint32_t ev[BATCH][4];
  static char line[BATCH * 2 * 160];
  unsigned long rid = 0;
  size_t got;
  while ((got = fread(ev, 16, BATCH, fe)) > 0) {
    char* p = line;
    for (size_t i = 0; i < got; i++, rid++) {
      const uint32_t c = ev[i][0], s = ev[i][1], e = ev[i][2];
      if ((int)c >= nc || ev[i][3] != 1 || e <= s) { fprintf(stderr, "events_to_sam: record %lu is not a unit-weight fragment\n", rid); return 1; }
      const unsigned long len = e - s, rl = len < 50 ? len : 50, p1 = (unsigned long)s + 1, p2 = (unsigned long)e - rl + 1;
      for (int mate = 0; mate < 2; mate++) {
        *p++ = 'r'; p = put_u(p, rid);
        p = put_s(p, mate ? "\t147\t" : "\t99\t");
        p = put_s(p, names[c]); *p++ = '\t';
        p = put_u(p, mate ? p2 : p1);
        p = put_s(p, "\t30\t"); p = put_u(p, rl); p = put_s(p, "M\t=\t");
        p = put_u(p, mate ? p1 : p2); *p++ = '\t';
        if (mate) *p++ = '-';
        p = put_u(p, len);
        p = put_s(p, "\t*\t*\tAS:i:0\n");
      }
    }
    if (fwrite(line, 1, (size_t)(p - line), fo) != (size_t)(p - line)) { perror("write"); return 1; }
  }
  if (fclose(fo)) { perror("close"); return 1; }
  fclose(fe);
  return 0;
}
