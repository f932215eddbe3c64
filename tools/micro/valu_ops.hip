// Issue rate of v_fma_f32 as a function of how many VGPR source operands it reads (gfx950).
//   MODE 1: a = fma(a, s, s)   one VGPR source (what valu_rate.hip measures: 2.44 cycles)
//   MODE 2: a = fma(a, b, s)   two VGPR sources
//   MODE 3: a = fma(b, c, a)   three VGPR sources (v_fmac-style accumulate)
//   hipcc -O3 --offload-arch=gfx950 valu_ops.hip -o valu_ops && ./valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8], b[8], c[8];
  for (int u = 0; u < 8; ++u) { a[u] = seed + threadIdx.x + u; b[u] = 0.999f + 1e-6f * (threadIdx.x + u); c[u] = 1e-3f * (u + 1) + 1e-7f * threadIdx.x; }
  const float m = 0.999f, cc = 0.001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE == 1) a[u] = __builtin_fmaf(a[u], m, cc);
        else if (MODE == 2) a[u] = __builtin_fmaf(a[u], b[u], cc);
        else a[u] = __builtin_fmaf(b[u], c[u], a[u]);
      }
  }
  float r = 0; for (int u = 0; u < 8; ++u) r += a[u] + b[u] + c[u];
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, int w) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * w;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 10, 1.f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-22s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, w, ms,
         ms * 1e-3 * 2.4e9 / (double(iters) * 64 * w));
  hipFree(d);
}
int main() {
  for (int w : {4, 8}) { run<1>("fma 1 VGPR source", w); run<2>("fma 2 VGPR sources", w); run<3>("fma 3 VGPR sources", w); }
  return 0;
}
