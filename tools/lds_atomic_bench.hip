// Developer micro-benchmark: LDS cycles per wave-instruction of ds_add_f64 / ds_read_b64 / ds_read_b128 for the
// lane -> window-slot maps of the tile kernels: slot = bx*sx + by*sy + bz*sz (+ a moving offset), lane = bx + 4 by + 16 bz.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, int sx, int sy, int sz, int shuffle) {
  __shared__ __attribute__((aligned(16))) double acc[2 * 4096];
  for (int i = threadIdx.x; i < 2 * 4096; i += 256) acc[i] = 0.0;
  __syncthreads();
  int lane = threadIdx.x & 63;
  if (shuffle) lane = (lane * 37 + 11) & 63;  // a permutation of the lanes: same slots, scrambled order
  const int bx = lane & 3, by = (lane >> 2) & 3, bz = lane >> 4;
  const int base = bx * sx + by * sy + bz * sz;
  double v = 1.0 + lane;
  const double2* a2 = reinterpret_cast<const double2*>(acc);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int idx = base + ((q * 5 + it) & 31);
      if (MODE == 0) atomicAdd(&acc[idx], v);
      else if (MODE == 1) v += acc[idx];
      else { double2 t = a2[idx]; v += t.x + t.y; }
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x] + v;
}
int main() {
  double* out;
  hipMalloc(&out, 8 * 256 * 4096);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 1000, blocks = 1024;
  const int maps[][3] = {{1, 4, 16}, {1, 8, 68}, {1, 8, 64}, {1, 8, 72}, {1, 9, 68}, {1, 9, 81}, {1, 10, 84}, {1, 12, 100},
                         {1, 8, 66}, {1, 8, 70}, {1, 8, 76}, {1, 8, 80}, {1, 4, 68}, {2, 8, 68}};
  for (auto& m : maps)
    for (int sh = 0; sh < 2; sh++) {
      float ms[3];
      for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(a);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, m[0], m[1], m[2], sh);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, m[0], m[1], m[2], sh);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, m[0], m[1], m[2], sh);
          hipEventRecord(b);
          hipEventSynchronize(b);
        }
        hipEventElapsedTime(&ms[mode], a, b);
      }
      const double wi = (double)blocks * 4 * iters * 16 / 256.0;
      printf("slot = %d bx + %2d by + %3d bz %s: ds_add_f64 %5.1f  ds_read_b64 %5.1f  ds_read_b128 %5.1f cycles/wave-instr/CU @2.4GHz\n",
             m[0], m[1], m[2], sh ? "(scrambled lanes)" : "(lattice order)  ", ms[0] * 1e-3 * 2.4e9 / wi, ms[1] * 1e-3 * 2.4e9 / wi,
             ms[2] * 1e-3 * 2.4e9 / wi);
    }
  return 0;
}
