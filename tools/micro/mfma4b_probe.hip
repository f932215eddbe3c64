// Layout probe of v_mfma_f32_16x16x1_4b_f32 (__builtin_amdgcn_mfma_f32_16x16x1f32): which (A lane, B lane) product lands in
// register k of lane l.  a(l) = p_l, b(l) = q_l with distinct primes-like codes so that every product identifies its pair.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma4b_probe.hip -o tools/micro/mfma4b_probe && tools/micro/mfma4b_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, long long* cyc) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const float a = float(1 + l), b = float(1 + l) * 128.f;
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0);
  for (int k = 0; k < 16; ++k) out[l * 16 + k] = acc[k];
  // issue-rate check: 64 dependent instructions
  f32x16 c;
  for (int k = 0; k < 16; ++k) c[k] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int r = 0; r < 64; ++r) c = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
  asm volatile("s_nop 0" : "+v"(c));
  const long long t1 = __builtin_amdgcn_s_memtime();
  f32x16 d;
  for (int k = 0; k < 16; ++k) d[k] = 0.f;
#pragma unroll
  for (int r = 0; r < 64; ++r) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
  asm volatile("s_nop 0" : "+v"(d));
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  out[64 * 16 + l] = c[0] + d[0];
}
int main() {
  float* d; long long* c;
  hipMalloc(&d, (64 * 16 + 64) * sizeof(float)); hipMalloc(&c, 16);
  probe<<<1, 64>>>(d, c);
  static float h[64 * 16]; long long hc[2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("64 dependent 16x16x1_4b: %lld cycles (%.1f each); 64 dependent 32x32x2: %lld cycles (%.1f each)\n", hc[0], hc[0] / 64.0, hc[1], hc[1] / 64.0);
  // decode: value = (1 + la) * (1 + lb) * 128
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int k = 0; k < 16; ++k) {
      const long long v = llround(h[l * 16 + k] / 128.0);
      int la = -1, lb = -1;
      for (int x = 0; x < 64 && la < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if ((long long)(1 + x) * (1 + y) == v && x / 16 == y / 16) { la = x; lb = y; if (y == l || (y % 16) == (l % 16)) break; }
      printf(" %d:(%d,%d)", k, la, lb);
    }
    printf("\n");
    if (l == 1) { l = 15; }
    else if (l == 17) { l = 31; }
    else if (l == 33) { l = 47; }
    else if (l == 49) break;
  }
  return 0;
}
