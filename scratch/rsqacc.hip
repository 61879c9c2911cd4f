// accuracy of v_rsq_f64 and of 1 / 2 Newton steps (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i], hd = -0.5 * d;
  double y = __builtin_amdgcn_rsq(d);
  y0[i] = y;
  y = y * fma(hd * y, y, 1.5);
  y1[i] = y;
  {
    double y0 = __builtin_amdgcn_rsq(d);
    double t = d * y0, h = fma(-t, y0, 1.0), q = fma(0.375, h, 0.5), yh = y0 * h;
    y = fma(yh, q, y0);
  }
  y2[i] = y;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;  // [1,2)
    int e = (int)((s >> 3) % 80) - 40;
    x[i] = ldexp(m, e);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / sqrtl((long double)x[i]);
    e0 = fmax(e0, (double)fabsl((a[i] - t) / t));
    e1 = fmax(e1, (double)fabsl((b[i] - t) / t));
    e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
  }
  printf("max rel err: rsq %.3e, +1 Newton %.3e, cubic %.3e (eps = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
  return 0;
}
