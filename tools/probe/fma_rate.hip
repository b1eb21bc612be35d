// Issue cost of independent v_fma_f64 / v_fma_f32 / v_pk_fma_f32 on one SIMD (cycles of the shader clock per wave-instruction),
// 16 independent accumulators, 1 / 2 / 4 wavefronts per SIMD on one CU; clock64() around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __global__ void k(long long* out, T c, T d, int iters) {
  T a[16];
  for (int i = 0; i < 16; i++) a[i] = c * (T)(i + 1) + (T)threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) a[i] = __builtin_fma(a[i], c, d);
    }
  }
  long long t1 = clock64();
  T s = 0; for (int i = 0; i < 16; i++) s += a[i];
  if (s == (T)12345.678) out[1] = 1;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  long long* out; hipMallocManaged(&out, 64);
  const int iters = 2000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    hipLaunchKernelGGL(k<double>, dim3(1), dim3(256 * waves), 0, 0, out, 0.999999, 1e-9, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<double>, dim3(1), dim3(256 * waves), 0, 0, out, 0.999999, 1e-9, iters); hipDeviceSynchronize();
    printf("v_fma_f64, %d wave(s) per SIMD: %.3f clk per wave-instruction (per wave), %.3f per SIMD issue slot\n", waves, (double)out[0] / (iters * 128.0), (double)out[0] / (iters * 128.0 * waves));
    hipLaunchKernelGGL(k<float>, dim3(1), dim3(256 * waves), 0, 0, out, 0.999999f, 1e-9f, iters); hipDeviceSynchronize();
    printf("v_fma_f32, %d wave(s) per SIMD: %.3f clk per wave-instruction (per wave), %.3f per SIMD issue slot\n", waves, (double)out[0] / (iters * 128.0), (double)out[0] / (iters * 128.0 * waves));
  }
  return 0;
}
