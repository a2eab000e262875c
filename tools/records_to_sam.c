/* records_to_sam -- renders a binary list of alignment records as SAM text (queryname-grouped as given), fast enough
 * for fixtures of 10^6 .. 10^7 alignments (genrich_amd/synth.py's Python writers manage ~3 x 10^4 lines a second).
 *
 *   records_to_sam RECORDS.bin CHROMS.txt OUT.sam PREFIX
 *     RECORDS.bin  28-byte little-endian records, in output order:
 *                    u32 template (QNAME = PREFIX + decimal)   u16 flag   i16 chrom   i32 pos0 (0-based)
 *                    i16 rnext chrom (-1: '*')   u8 read length   u8 quality (0xFF: QUAL '*', else that Phred value on
 *                    every base)   i32 pnext0 (-1 with rnext -1)   i32 tlen   i8 AS   u8 mapq   u16 pad
 *     CHROMS.txt   one "name length" line per chromosome, in header order
 *   SEQ is 'A' x read length, CIGAR <read length>M, tags NM:i:0 AS:i:<AS>.
 * Test tooling: not part of the library, reads nothing of the reference.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#pragma pack(push, 1)
typedef struct {
  uint32_t tmpl;
  uint16_t flag;
  int16_t chrom;
  int32_t pos;
  int16_t rnext;
  uint8_t rl, qual;
  int32_t pnext, tlen;
  int8_t as;
  uint8_t mapq;
  uint16_t pad;
} Rec;
#pragma pack(pop)

static char* put_u(char* p, unsigned long v) {
  char t[24];
  int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = t[--n];
  return p;
}
static char* put_i(char* p, long v) {
  if (v < 0) { *p++ = '-'; v = -v; }
  return put_u(p, (unsigned long)v);
}
static char* put_s(char* p, const char* s) {
  while (*s) *p++ = *s++;
  return p;
}

int main(int argc, char** argv) {
  if (argc != 5 || sizeof(Rec) != 28) {
    fprintf(stderr, "usage: records_to_sam RECORDS.bin CHROMS.txt OUT.sam PREFIX\n");
    return 2;
  }
  FILE* fc = fopen(argv[2], "r");
  if (!fc) { perror(argv[2]); return 1; }
  char (*names)[64] = NULL;
  int nc = 0;
  char nm[64];
  unsigned long ln;
  FILE* fo = fopen(argv[3], "w");
  if (!fo) { perror(argv[3]); return 1; }
  static char obuf[1 << 22];
  setvbuf(fo, obuf, _IOFBF, sizeof obuf);
  fputs("@HD\tVN:1.0\tSO:queryname\n", fo);
  while (fscanf(fc, "%63s %lu", nm, &ln) == 2) {
    names = realloc(names, (size_t)(nc + 1) * sizeof *names);
    strcpy(names[nc++], nm);
    fprintf(fo, "@SQ\tSN:%s\tLN:%lu\n", nm, ln);
  }
  fclose(fc);
  FILE* fr = fopen(argv[1], "rb");
  if (!fr) { perror(argv[1]); return 1; }
  enum { BATCH = 1 << 15 };
  static Rec rec[BATCH];
  static char line[BATCH * 768];
  size_t got;
  unsigned long at = 0;
  while ((got = fread(rec, sizeof(Rec), BATCH, fr)) > 0) {
    char* p = line;
    for (size_t i = 0; i < got; i++, at++) {
      const Rec* r = rec + i;
      if (r->chrom < 0 || r->chrom >= nc || r->rnext >= nc) { fprintf(stderr, "records_to_sam: record %lu names no chromosome\n", at); return 1; }
      p = put_s(p, argv[4]); p = put_u(p, r->tmpl); *p++ = '\t';
      p = put_u(p, r->flag); *p++ = '\t';
      p = put_s(p, names[r->chrom]); *p++ = '\t';
      p = put_u(p, (unsigned long)r->pos + 1); *p++ = '\t';
      p = put_u(p, r->mapq); *p++ = '\t';
      p = put_u(p, r->rl); p = put_s(p, "M\t");
      if (r->rnext < 0) p = put_s(p, "*\t0\t");
      else {
        p = put_s(p, r->rnext == r->chrom ? "=" : names[r->rnext]); *p++ = '\t';
        p = put_u(p, (unsigned long)r->pnext + 1); *p++ = '\t';
      }
      p = put_i(p, r->tlen); *p++ = '\t';
      memset(p, 'A', r->rl); p += r->rl; *p++ = '\t';
      if (r->qual == 0xFF) *p++ = '*';
      else { memset(p, 33 + r->qual, r->rl); p += r->rl; }
      p = put_s(p, "\tNM:i:0\tAS:i:"); p = put_i(p, r->as); *p++ = '\n';
    }
    if (fwrite(line, 1, (size_t)(p - line), fo) != (size_t)(p - line)) { perror("write"); return 1; }
  }
  if (fclose(fo)) { perror("close"); return 1; }
  fclose(fr);
  return 0;
}
