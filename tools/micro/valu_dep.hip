// Issue rate of a fully DEPENDENT fp32 FMA chain (ILP = 1) vs waves per SIMD, gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[ILP];
  for (int u = 0; u < ILP; ++u) a[u] = seed + threadIdx.x + u;
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64 / ILP; ++r)
#pragma unroll
      for (int u = 0; u < ILP; ++u) a[u] = __builtin_fmaf(a[u], m, c);
  }
  float r = 0; for (int u = 0; u < ILP; ++u) r += a[u];
  if (r == 12345.678f) out[0] = r;
}
template <int ILP> void run(int w) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * w;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<ILP><<<blocks, 256>>>(d, 10, 1.f); hipDeviceSynchronize();
  hipEventRecord(a); k<ILP><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("ILP %d waves/SIMD %d: %.2f cycles per wave-instruction per SIMD\n", ILP, w, ms * 1e-3 * 2.4e9 / (double(iters) * 64 * w));
  hipFree(d);
}
int main() { for (int w : {1, 2, 3, 4, 6, 8}) { run<1>(w); run<2>(w); run<4>(w); } return 0; }
