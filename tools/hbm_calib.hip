// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS code's access width (8 B per lane, coalesced
// SoA doubles): MI355X_MICROARCH.md §HBM says FETCH_SIZE reads 1/2 for 16-B-per-lane streams and that other
// widths must be calibrated on a known byte count.  Copies N doubles (known: 8N read, 8N written).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0;
}
int main() {
  const size_t n = (size_t)1 << 27;  // 1 GiB in, 1 GiB out: far beyond the 256 MiB Infinity Cache
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
  (void)hipMemset(a, 0, n * 8);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(copy8, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
  (void)hipDeviceSynchronize();
  printf("copy8: %zu bytes read, %zu bytes written per launch\n", n * 8, n * 8);
  return 0;
}
