// Cost of the hand-offs a persistent PCG would make (round 5): G co-resident workgroups of 512 threads; per step every
// workgroup publishes P doubles (agent-scope relaxed stores = global_store sc1), drains, sets its flag to the step's epoch;
// then polls ALL G flags (one wave, relaxed agent loads) and reads ALL G x P published doubles (agent loads) - the all-gather a
// CG reduction / coarse-residual exchange needs.  Also a neighbour pattern: 11 KB published per workgroup, 6 KB read.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/persist_sync_cost.hip -o build_exp/persist_sync_cost && build_exp/persist_sync_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__global__ __launch_bounds__(512) void k_sync(double *pay, unsigned *flag, int iters, int G, int P, double *out, unsigned *err) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ double red[8];
  __shared__ int ok_s;
  double acc = 1.0 + b;
  for (int it = 0; it < iters; ++it) {
    const unsigned epoch = it + 1;
    double *mine = pay + ((size_t)(it & 1) * G + b) * P;
    for (int e = threadIdx.x; e < P; e += 512) __hip_atomic_store(mine + e, acc + e, RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag + b, epoch, RLX);
    if (wv == 0) {
      int good = 1;
      unsigned spins = 0;
      for (;;) {
        bool all = true;
        for (int q = lane; q < G; q += 64) all = all && (__hip_atomic_load(flag + q, RLX) >= epoch);
        if (__all(all)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { good = 0; break; }
      }
      if (lane == 0) ok_s = good;
    }
    __syncthreads();
    if (!ok_s) { if (threadIdx.x == 0) atomicExch(err, 1u); return; }
    const double *all = pay + (size_t)(it & 1) * G * P;
    double s = 0.0;
    for (int e = threadIdx.x; e < G * P; e += 512) s += __hip_atomic_load(all + e, RLX);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    s = 0.0;
    for (int q = 0; q < 8; ++q) s += red[q];
    acc = s * 1e-6 + 1.0;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = acc;
}
// neighbour pattern: each workgroup publishes ROWS x 6 doubles, reads HALO x 6 from each of its two ring neighbours
__global__ __launch_bounds__(512) void k_halo(double *pay, unsigned *flag, int iters, int G, int rows, int halo, double *out, unsigned *err) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ int ok_s;
  double acc = 1.0 + b;
  for (int it = 0; it < iters; ++it) {
    const unsigned epoch = it + 1;
    double *mine = pay + ((size_t)(it & 1) * G + b) * rows * 6;
    for (int e = threadIdx.x; e < rows * 6; e += 512) __hip_atomic_store(mine + e, acc + e, RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag + b, epoch, RLX);
    const int nl = (b + G - 1) % G, nr = (b + 1) % G;
    if (wv == 0) {
      int good = 1;
      unsigned spins = 0;
      for (;;) {
        const bool a = __hip_atomic_load(flag + (lane & 1 ? nl : nr), RLX) >= epoch;
        if (__all(a)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { good = 0; break; }
      }
      if (lane == 0) ok_s = good;
    }
    __syncthreads();
    if (!ok_s) { if (threadIdx.x == 0) atomicExch(err, 1u); return; }
    double s = 0.0;
    for (int e = threadIdx.x; e < 2 * halo * 6; e += 512) {
      const int nb = e < halo * 6 ? nl : nr, k = e < halo * 6 ? e : e - halo * 6;
      s += __hip_atomic_load(pay + ((size_t)(it & 1) * G + nb) * rows * 6 + k, RLX);
    }
    acc = s * 1e-9 + 1.0;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = acc;
}
int main() {
  const int iters = 2000;
  for (int G : {64, 128, 256}) {
    for (int P : {1, 16}) {
      double *pay, *out; unsigned *flag, *err;
      hipMalloc(&pay, (size_t)2 * G * P * 8); hipMalloc(&out, G * 8); hipMalloc(&flag, G * 4); hipMalloc(&err, 4);
      hipMemset(pay, 0, (size_t)2 * G * P * 8); hipMemset(flag, 0, G * 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k_sync, dim3(G), dim3(512), 0, 0, pay, flag, 5, G, P, out, err);
      hipDeviceSynchronize(); hipMemset(flag, 0, G * 4);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_sync, dim3(G), dim3(512), 0, 0, pay, flag, iters, G, P, out, err);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("all-gather  G = %3d workgroups, %2d doubles each: %.2f us per step (err %u)\n", G, P, ms * 1e3 / iters, e);
      hipFree(pay); hipFree(out); hipFree(flag); hipFree(err);
    }
    {
      const int rows = 230, halo = 80;
      double *pay, *out; unsigned *flag, *err;
      hipMalloc(&pay, (size_t)2 * G * rows * 6 * 8); hipMalloc(&out, G * 8); hipMalloc(&flag, G * 4); hipMalloc(&err, 4);
      hipMemset(pay, 0, (size_t)2 * G * rows * 6 * 8); hipMemset(flag, 0, G * 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k_halo, dim3(G), dim3(512), 0, 0, pay, flag, 5, G, rows, halo, out, err);
      hipDeviceSynchronize(); hipMemset(flag, 0, G * 4);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_halo, dim3(G), dim3(512), 0, 0, pay, flag, iters, G, rows, halo, out, err);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("halo        G = %3d workgroups, %d rows published, 2 x %d read: %.2f us per step (err %u)\n", G, rows, halo, ms * 1e3 / iters, e);
      hipFree(pay); hipFree(out); hipFree(flag); hipFree(err);
    }
  }
  return 0;
}
