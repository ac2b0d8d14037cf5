// Developer test of the synchronisation skeleton of k_step_fused (no physics): persistent workgroups, one queue,
// stage-0 items publish a per-tile flag, stage-1 items wait for their neighbours' flags and publish their own.
//   hipcc -O3 --offload-arch=gfx950 -o build/exp/fused_sync_test tools/fused_sync_test.hip && build/exp/fused_sync_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__device__ __forceinline__ bool wait_flag(const unsigned* flag, unsigned seq) {
  const unsigned long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    if (wall_clock64() - t0 > 20000000ull) return false;
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}
__device__ __forceinline__ void publish(unsigned* flag, unsigned seq, int mode) {
  if (mode == 1) return;
  if (mode != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (mode != 5) __syncthreads();
  if (mode == 4 || mode == 5) return;
  if (threadIdx.x == 0) {
    if (mode != 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (mode != 3) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
struct Big {
  unsigned* q;
  unsigned* d2;
  unsigned* d3;
  unsigned seq;
  int n, nstages;
  double pad[40];
  int mode;
};
__device__ __noinline__ int use_big(const Big* b) { return b->n + (int)b->pad[7]; }
#ifndef VAR
#define VAR 0
#endif
__global__ __launch_bounds__(256, 3) void k(unsigned* q, unsigned* d2, unsigned* d3, unsigned seq, int n, int nstages, double* data,
                                         int* status, int mode, Big big) {
  __shared__ int s_item;
#if VAR >= 1
  __shared__ double sh[4044];
  for (int i = threadIdx.x; i < 4044; i += 256) sh[i] = (double)i;
  __syncthreads();
  if (sh[threadIdx.x] < -1.0) atomicOr(status, 2);
#endif
#if VAR >= 2
  n = use_big(&big);
  q = big.q;
  d2 = big.d2;
  d3 = big.d3;
#endif
  while (true) {
    if (threadIdx.x == 0) s_item = (int)atomicAdd(q, 1u);
    __syncthreads();
    const int item = s_item;
    __syncthreads();
    if (item >= nstages * n) break;
    const int stage = item / n, t = item - stage * n;
    if (stage >= 1) {
      if (threadIdx.x < 3) {
        const int nb = t + (int)threadIdx.x - 1;
        if (nb >= 0 && nb < n)
          if (!wait_flag((stage == 1 ? d2 : d3) + nb, seq)) atomicOr(status, 1);
      }
      __syncthreads();
    }
    // some work: atomics on data like a window flush
    for (int r = threadIdx.x; r < 1024; r += 256) atomicAdd(&data[(t * 64 + r) % (n * 64)], 1.0);
    if (stage < 2) publish((stage == 0 ? d2 : d3) + t, seq, mode);
  }
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int n = argc > 2 ? atoi(argv[2]) : 2197;
  const int grid = argc > 3 ? atoi(argv[3]) : 768;
  printf("start\n");
  fflush(stdout);
  unsigned *q, *d;
  double* data;
  int* st;
  hipMalloc(&q, 64);
  hipMalloc(&d, 2 * n * sizeof(unsigned));
  hipMalloc(&data, n * 64 * sizeof(double));
  hipMalloc(&st, 4);
  hipMemset(d, 0, 2 * n * sizeof(unsigned));
  hipMemset(data, 0, n * 64 * sizeof(double));
  hipMemset(st, 0, 4);
  Big big;
  memset(&big, 0, sizeof big);
  big.q = q; big.d2 = d; big.d3 = d + n; big.n = n;
  for (int ns = 1; ns <= (mode ? 1 : 3); ns++)
    for (unsigned seq = 1; seq <= 3; seq++) {
      hipMemset(q, 0, 4);
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, q, d, d + n, seq + 10 * ns, n, ns, data, st, mode, big);
      hipEventRecord(b);
      hipError_t e = hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      int h = 0;
      hipMemcpy(&h, st, 4, hipMemcpyDeviceToHost);
      printf("stages %d seq %u: %s %.3f ms status %d\n", ns, seq, hipGetErrorString(e), ms, h);
      fflush(stdout);
    }
  return 0;
}
