// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the blur kernel's access pattern: 2-byte
// per-lane loads (128 contiguous bytes per wave instruction) and 2-byte stores, over a known
// byte count larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_read16(const unsigned short *p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride) acc += p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write16(unsigned short *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (unsigned short)i;
}
int main() {
  const size_t n = 512ull << 20;  // 512 Mi elements = 1 GiB
  unsigned short *p; unsigned *out;
  if (hipMalloc(&p, n * 2) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(p, 1, n * 2);
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(k_read16, dim3(256 * 16), dim3(256), 0, 0, p, n, out);
    hipLaunchKernelGGL(k_write16, dim3(256 * 16), dim3(256), 0, 0, p, n);
  }
  (void)hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", n * 2);
  return 0;
}
