// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_rsq_f32 on gfx950.
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
  const float m = 0.999f, c = 0.001f;
  const v2 mm = {m, m}, cc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
        p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
        p4 = __builtin_elementwise_fma(p4, mm, cc); p5 = __builtin_elementwise_fma(p5, mm, cc);
        p6 = __builtin_elementwise_fma(p6, mm, cc); p7 = __builtin_elementwise_fma(p7, mm, cc);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
        a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
      }
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, int wavesPerSimd) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * wavesPerSimd;  // 256-thread blocks: 1 wave per SIMD each
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 10, 1.f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = double(iters) * 64 * wavesPerSimd;  // wave-instructions issued on each SIMD
  printf("%-12s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, wavesPerSimd, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_rsq_f32", w); }
  return 0;
}
