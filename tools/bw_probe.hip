// HBM streaming probe for MI355X: what do simple read / write / mixed patterns reach?
// hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill16(uint4* o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = make_uint4(1, 2, 3, 4);
}
__global__ void k_fill4(u32* o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = 7;
}
__global__ void k_read16(const uint4* a, size_t n, u32* sink) {
  u32 s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
  if (s == 0x12345) *sink = s;
}
__global__ void k_read4(const u32* a, size_t n, u32* sink) {
  u32 s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 0x12345) *sink = s;
}
__global__ void k_copy16(const uint4* a, uint4* o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i];
}
// 2 x 4-byte reads -> 3 x 4-byte writes per element (k_pack_pval's traffic shape), coalesced, aligned
__global__ void k_2r3w(const u32* a, const u32* b, u32* x, u32* y, u32* z, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32 e = a[i], v = b[i];
    x[i] = e; y[i] = v * 3u; z[i] = v + e;
  }
}
// same traffic, 4 elements per thread through 16-byte accesses
__global__ void k_2r3w_v4(const uint4* a, const uint4* b, uint4* x, uint4* y, uint4* z, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    uint4 e = a[i], v = b[i];
    x[i] = e; y[i] = make_uint4(v.x * 3, v.y * 3, v.z * 3, v.w * 3); z[i] = make_uint4(v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w);
  }
}
// 1 x 4-byte read -> 1 bit written (k_sig_mask's shape)
__global__ void k_read4_unroll(const u32* a, size_t n, u32* sink) {
  u32 s = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) { u32 p = a[i], q = a[i + stride], r = a[i + 2 * stride], t = a[i + 3 * stride]; s += p + q + r + t; }
  for (; i < n; i += stride) s += a[i];
  if (s == 0x12345) *sink = s;
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < reps; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}

int main() {
  const size_t N = 84u << 20;  // elements of 4 bytes per stream (~352 MB), like the interval arrays
  u32 *A, *B, *X, *Y, *Z, *sink;
  CK(hipMalloc(&A, N * 4)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&X, N * 4)); CK(hipMalloc(&Y, N * 4)); CK(hipMalloc(&Z, N * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 1, N * 4)); CK(hipMemset(B, 2, N * 4));
  for (int grid : {1024, 2048, 4096, 16384}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, 0, (uint4*)X, N / 4); });
    printf("grid %5d fill16   %7.1f us %6.2f TB/s\n", grid, t * 1e3, N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_fill4, dim3(grid), dim3(256), 0, 0, X, N); });
    printf("grid %5d fill4    %7.1f us %6.2f TB/s\n", grid, t * 1e3, N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)A, N / 4, sink); });
    printf("grid %5d read16   %7.1f us %6.2f TB/s\n", grid, t * 1e3, N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, A, N, sink); });
    printf("grid %5d read4    %7.1f us %6.2f TB/s\n", grid, t * 1e3, N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_read4_unroll, dim3(grid), dim3(256), 0, 0, A, N, sink); });
    printf("grid %5d read4x4  %7.1f us %6.2f TB/s\n", grid, t * 1e3, N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)A, (uint4*)X, N / 4); });
    printf("grid %5d copy16   %7.1f us %6.2f TB/s (r+w)\n", grid, t * 1e3, 2.0 * N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_2r3w, dim3(grid), dim3(256), 0, 0, A, B, X, Y, Z, N); });
    printf("grid %5d 2r3w     %7.1f us %6.2f TB/s (r+w)\n", grid, t * 1e3, 5.0 * N * 4 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_2r3w_v4, dim3(grid), dim3(256), 0, 0, (const uint4*)A, (const uint4*)B, (uint4*)X, (uint4*)Y, (uint4*)Z, N / 4); });
    printf("grid %5d 2r3w_v4  %7.1f us %6.2f TB/s (r+w)\n", grid, t * 1e3, 5.0 * N * 4 / t / 1e9);
  }
  return 0;
}
