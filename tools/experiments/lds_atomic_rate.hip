// LDS atomic throughput on gfx950: ds_add_f64 against ds_add_u64 / ds_add_u32 / plain read-modify-write, same address pattern
// as the K*p accumulator (component-major [6][stride], 64 lanes on consecutive or scattered rows).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/experiments/lds_atomic_rate.hip -o build_exp/lds_atomic_rate && build_exp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int kStride = 153, kIter = 2000;
template <int MODE, bool SCATTER>
__global__ __launch_bounds__(512) void k(double *out) {
  __shared__ double ys[6 * kStride];
  for (int i = threadIdx.x; i < 6 * kStride; i += 512) ys[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x;
  double v = 1.0 + lane * 1e-3;
  for (int it = 0; it < kIter; ++it) {
    int row = SCATTER ? (int)(((unsigned)lane * 2654435761u + it * 40503u) % 152u) : (lane + it) % 152;
#pragma unroll
    for (int k2 = 0; k2 < 6; ++k2) {
      double *p = ys + k2 * kStride + row;
      if (MODE == 0) unsafeAtomicAdd(p, v);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
      else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned int *>(p), (unsigned)lane);
      else *p = *p + v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ys[threadIdx.x];
}
template <int MODE, bool SCATTER>
float run(double *d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, SCATTER>), dim3(1024), dim3(512), 0, 0, d);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, SCATTER>), dim3(1024), dim3(512), 0, 0, d);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  double *d; hipMalloc(&d, 1024 * 8);
  const char *nm[4] = {"ds_add_f64", "ds_add_u64", "ds_add_u32", "plain rmw b64"};
  float t[4][2] = {{run<0, false>(d), run<0, true>(d)}, {run<1, false>(d), run<1, true>(d)}, {run<2, false>(d), run<2, true>(d)},
                   {run<3, false>(d), run<3, true>(d)}};
  // 1024 workgroups of 8 waves on 256 CUs (4 per CU at a time): wave-instructions per CU = 4 * 8 * kIter * 6
  for (int m = 0; m < 4; ++m)
    for (int s = 0; s < 2; ++s) {
      const double instr_per_cu = 4.0 * 8 * kIter * 6;
      printf("%-14s %-10s %7.3f ms  -> %5.1f clk per wave-instruction per CU (2.4 GHz)\n", nm[m], s ? "scattered" : "consecutive",
             t[m][s], t[m][s] * 1e-3 * 2.4e9 / instr_per_cu);
    }
  return 0;
}
