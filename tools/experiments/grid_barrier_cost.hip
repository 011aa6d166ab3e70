// Cost of a grid-wide barrier on gfx950 (8 XCDs): G co-resident workgroups, every iteration each workgroup writes one value
// another workgroup reads after the barrier (so the barrier must make global writes visible across XCDs).  Compared with the
// same work split into dependent kernel launches on one stream.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/grid_barrier_cost.hip -o build_exp/grid_barrier_cost && build_exp/grid_barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fused(double *buf, unsigned *bar, int iters, int G) {
  const int b = blockIdx.x;
  double v = 0.0;
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(buf + (it & 1) * G + b, v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __atomic_thread_fence(__ATOMIC_RELEASE);
      const unsigned target = (unsigned)(it + 1) * (unsigned)G;
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 100000000L) { }
    }
    __syncthreads();
    v = __hip_atomic_load(buf + (it & 1) * G + (b + 1) % G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) buf[2 * G + b] = v;
}
__global__ void k_step(double *buf, int it, int G) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) buf[((it + 1) & 1) * G + b] = buf[(it & 1) * G + (b + 1) % G] + 1.0;
}
int main() {
  const int iters = 2000;
  for (int G : {64, 256, 512, 1024}) {
    double *buf; unsigned *bar;
    hipMalloc(&buf, 3 * G * sizeof(double)); hipMalloc(&bar, 4);
    hipMemset(buf, 0, 3 * G * sizeof(double)); hipMemset(bar, 0, 4);
    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
    hipLaunchKernelGGL(k_fused, dim3(G), dim3(256), 0, 0, buf, bar, 10, G);      // warm-up
    hipDeviceSynchronize();
    hipMemset(bar, 0, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_fused, dim3(G), dim3(256), 0, 0, buf, bar, iters, G);
    hipEventRecord(b2); hipEventSynchronize(b2);
    float ms1; hipEventElapsedTime(&ms1, a, b2);
    hipEventRecord(a);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, 0, buf, it, G);
    hipEventRecord(b2); hipEventSynchronize(b2);
    float ms2; hipEventElapsedTime(&ms2, a, b2);
    double h; hipMemcpy(&h, buf + 2 * G, 8, hipMemcpyDeviceToHost);
    printf("G = %4d workgroups: grid barrier %.2f us per step, dependent kernel launch %.2f us per step (check %.0f)\n", G,
           ms1 * 1e3 / iters, ms2 * 1e3 / iters, h);
    hipFree(buf); hipFree(bar);
  }
  return 0;
}
