// Accuracy of the hardware v_log_f32 (log2) path vs libm logf, on gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* fast, float* slow, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { fast[i] = __builtin_amdgcn_logf(x[i]) * 0.6931471805599453f; slow[i] = logf(x[i]); }
}
int main() {
  const int n = 1 << 20;
  std::vector<float> x(n), f(n), s(n);
  float *dx, *df, *ds; hipMalloc(&dx, 4 * n); hipMalloc(&df, 4 * n); hipMalloc(&ds, 4 * n);
  const double ranges[4][2] = {{0.9, 1.1}, {0.5, 2.0}, {1e-3, 1e3}, {1e-8, 1e8}};
  for (auto& r : ranges) {
    for (int i = 0; i < n; ++i) x[i] = float(r[0] * std::pow(r[1] / r[0], (i + 0.5) / n));
    hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, df, ds, n);
    hipMemcpy(f.data(), df, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, 4 * n, hipMemcpyDeviceToHost);
    double ea = 0, er = 0, sa = 0, sr = 0;
    for (int i = 0; i < n; ++i) {
      double t = std::log((double)x[i]);
      ea = std::fmax(ea, std::fabs(f[i] - t)); sa = std::fmax(sa, std::fabs(s[i] - t));
      if (std::fabs(t) > 1e-3) { er = std::fmax(er, std::fabs(f[i] - t) / std::fabs(t)); sr = std::fmax(sr, std::fabs(s[i] - t) / std::fabs(t)); }
    }
    printf("x in [%g,%g]: v_log_f32*ln2 max abs err %.3e rel %.3e | logf abs %.3e rel %.3e\n", r[0], r[1], ea, er, sa, sr);
  }
  return 0;
}
