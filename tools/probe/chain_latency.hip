// Dependent-chain cost of v_fma_f64: K independent chains per wavefront (K = 1, 2, 4, 8), 1 / 2 / 4 wavefronts per SIMD on one CU.
// If a dependent v_fma_f64 can issue every 4 cycles the K = 1 line reads ~4 clk per instruction; anything above that is latency
// the recursions of the tile engine (two dependent FMAs per sample and filter) have to hide with other chains or wavefronts.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void k_chain(long long* out, double seed, double c) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
#pragma unroll
      for (int i = 0; i < K; i++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  if (s == 12345.678) out[1000] = 1;
}
template <int K> void run(long long* out) {
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipLaunchKernelGGL(k_chain<K>, dim3(1), dim3(256 * waves), 0, 0, out, 1.25, 0.999); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain<K>, dim3(1), dim3(256 * waves), 0, 0, out, 1.25, 0.999); hipDeviceSynchronize();
    printf("v_fma_f64, %d chain(s), %d wave(s)/SIMD: %6.2f clk per instruction per wave\n", K, waves, (double)out[0] / (256.0 * 32 * K));
  }
}
int main() {
  long long* out; hipMallocManaged(&out, 8 * 2048);
  run<1>(out); run<2>(out); run<4>(out); run<8>(out);
  return 0;
}
