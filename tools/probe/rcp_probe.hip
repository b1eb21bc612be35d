// accuracy of v_rcp_f64 / v_rsq_f64 seeds and of the Newton steps built on them (decides how many steps the fast divisions need)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double d = x[i];
  double y = __builtin_amdgcn_rcp(d);
  y0[i] = y;
  y = fma(fma(-d, y, 1.0), y, y); y1[i] = y;
  y = fma(fma(-d, y, 1.0), y, y); y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  double *x, *a, *b, *c;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&a, n * 8); hipMallocManaged(&b, n * 8); hipMallocManaged(&c, n * 8);
  for (int i = 0; i < n; i++) x[i] = 1.0 + (double)i / n * 3.0 + 1e-9 * (i % 7);
  k<<<n / 256, 256>>>(x, a, b, c, n); hipDeviceSynchronize();
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) { long double t = 1.0L / (long double)x[i]; e0 = fmax(e0, fabsl((a[i] - t) / t)); e1 = fmax(e1, fabsl((b[i] - t) / t)); e2 = fmax(e2, fabsl((c[i] - t) / t)); }
  printf("rcp_f64 max rel err: seed %.3e  1 Newton %.3e  2 Newton %.3e\n", e0, e1, e2);
  return 0;
}
