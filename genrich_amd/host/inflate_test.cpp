// A C entry point around gx_inflate.h for tests/test_inflate.py (ctypes): the decoder against zlib on the same bytes.
#include "gx_inflate.h"
extern "C" int gx_fast_inflate(const unsigned char* in, unsigned long inLen, unsigned char* out, unsigned long outLen) {
  return gxinf::inflate(in, inLen, out, outLen) ? 1 : 0;
}
#include "gx_crc32.h"
extern "C" unsigned int gx_fast_crc32(const unsigned char* p, unsigned long n) { return gxcrc::crc32_of(p, n); }
