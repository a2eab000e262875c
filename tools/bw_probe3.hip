// Does it matter which XCD writes the neighbouring short runs of a scatter?  (k_scatter1's write shape)
// Each workgroup (chunk) writes one run of RUN u32 per bin.  Layout A: runs of consecutive chunks
// are adjacent (chunks c, c+1, ... sit on different XCDs: a 128-byte line is completed by ~6
// workgroups on ~5 XCDs).  Layout B: runs of chunks with equal c % 8 are adjacent (same XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
constexpr int NT = 1024;

template <int LAYOUT>
__global__ __launch_bounds__(NT) void k_runs(const u32* __restrict__ in, u32* __restrict__ out, u32 nChunks, u32 nBins, u32 run) {
  const u32 c = blockIdx.x;
  const u32 perChunk = nBins * run;
  const u32 perBin = nChunks * run;
  for (u32 i = threadIdx.x; i < perChunk; i += NT) {
    const u32 b = i / run, k = i % run;
    const u32 v = in[c * perChunk + i];
    size_t pos;
    if (LAYOUT == 0) pos = (size_t)b * perBin + (size_t)c * run + k;
    else pos = (size_t)b * perBin + (size_t)(c % 8) * (perBin / 8) + (size_t)(c / 8) * run + k;
    out[pos] = v;
  }
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < reps; r++) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}

int main() {
  for (u32 nBins : {738u, 1474u}) for (u32 run : {3u, 6u, 11u, 32u}) {
    const u32 nChunks = 6104;  // multiple of 8
    const size_t n = (size_t)nChunks * nBins * run;
    u32 *in, *out;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(in, 1, n * 4);
    float a = timeit([&] { hipLaunchKernelGGL(k_runs<0>, dim3(nChunks), dim3(NT), 0, 0, in, out, nChunks, nBins, run); });
    float b = timeit([&] { hipLaunchKernelGGL(k_runs<1>, dim3(nChunks), dim3(NT), 0, 0, in, out, nChunks, nBins, run); });
    printf("bins %4u run %2u (%5.1f MB): interleaved XCDs %7.1f us (%5.2f TB/s r+w)   XCD-local %7.1f us (%5.2f TB/s)\n", nBins, run,
           n * 4 / 1e6, a * 1e3, 2.0 * n * 4 / a / 1e9, b * 1e3, 2.0 * n * 4 / b / 1e9);
    (void)hipFree(in); (void)hipFree(out);
  }
  return 0;
}
