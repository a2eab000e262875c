// Which feature of k_pack_pval costs what?  Tile-structured 2-read / 3-write copies on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;

template <int MODE>  // 0 plain, 1 + LDS gather, 2 + int division math
__global__ __launch_bounds__(256) void k_tiles(const u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ slot,
                                               const u32* __restrict__ off, u32 nTiles, u32* x, u32* y, u32* z,
                                               const float* __restrict__ lut) {
  __shared__ float hot[4096];
  if (MODE >= 1) { for (int i = threadIdx.x; i < 4096; i += 256) hot[i] = lut[i]; __syncthreads(); }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (u32 t = blockIdx.x * 4 + wv; t < nTiles; t += gridDim.x * 4) {
    const u32 src = slot[t], dst = off[t], n = off[t + 1] - dst;
    for (u32 base = 0; base < n; base += 256) {
      u32 e[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { u32 i = base + k * 64 + lane; e[k] = 0; v[k] = 0; if (i < n) { e[k] = a[src + i]; v[k] = b[src + i]; } }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u32 i = base + k * 64 + lane;
        if (i < n) {
          float p = MODE >= 1 ? hot[v[k] & 4095] : __uint_as_float(v[k]);
          float val = MODE >= 2 ? (float)(int)(v[k] / 120u) : __uint_as_float(v[k] + 1);
          x[dst + i] = e[k]; y[dst + i] = __float_as_uint(p); z[dst + i] = __float_as_uint(val);
        }
      }
    }
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
  const u32 nTiles = 377000;
  for (int cfg = 0; cfg < 4; cfg++) {
    // cfg 0: n=223, slot stride 265 (k_pack_pval's shape); 1: n=223 dense slots; 2: n=256 aligned dense; 3: n=223, stride 265, dst padded to 256
    std::vector<u32> slot(nTiles), off(nTiles + 1);
    u32 s = 0, d = 0;
    for (u32 t = 0; t < nTiles; t++) {
      u32 n = cfg == 2 ? 256 : 223;
      slot[t] = s; off[t] = d;
      s += (cfg == 0 || cfg == 3) ? 265 : n;
      d += cfg == 3 ? 256 : n;
    }
    off[nTiles] = d;
    size_t NS = s + 1024, ND = d + 1024;
    u32 *A, *B, *X, *Y, *Z, *dSlot, *dOff; float* lut;
    (void)hipMalloc(&A, NS * 4); (void)hipMalloc(&B, NS * 4); (void)hipMalloc(&X, ND * 4); (void)hipMalloc(&Y, ND * 4); (void)hipMalloc(&Z, ND * 4);
    (void)hipMalloc(&dSlot, nTiles * 4); (void)hipMalloc(&dOff, (nTiles + 1) * 4); (void)hipMalloc(&lut, 4096 * 4);
    (void)hipMemset(A, 1, NS * 4); (void)hipMemset(B, 0, NS * 4); (void)hipMemset(lut, 0, 4096 * 4);
    (void)hipMemcpy(dSlot, slot.data(), nTiles * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dOff, off.data(), (nTiles + 1) * 4, hipMemcpyHostToDevice);
    double bytes = 5.0 * 223 * nTiles * 4;
    if (cfg == 2) bytes = 5.0 * 256 * nTiles * 4;
    for (int grid : {2048, 8192}) {
      float t0 = timeit([&] { hipLaunchKernelGGL(k_tiles<0>, dim3(grid), dim3(256), 0, 0, A, B, dSlot, dOff, nTiles, X, Y, Z, lut); });
      float t1 = timeit([&] { hipLaunchKernelGGL(k_tiles<1>, dim3(grid), dim3(256), 0, 0, A, B, dSlot, dOff, nTiles, X, Y, Z, lut); });
      float t2 = timeit([&] { hipLaunchKernelGGL(k_tiles<2>, dim3(grid), dim3(256), 0, 0, A, B, dSlot, dOff, nTiles, X, Y, Z, lut); });
      printf("cfg %d grid %5d plain %7.1f us (%5.2f TB/s)  +lds %7.1f us  +math %7.1f us\n", cfg, grid, t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, t2 * 1e3);
    }
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(X); (void)hipFree(Y); (void)hipFree(Z); (void)hipFree(dSlot); (void)hipFree(dOff); (void)hipFree(lut);
  }
  return 0;
}
