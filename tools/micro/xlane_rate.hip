// Issue cost of the cross-lane instructions of the transposing reduction (csrc/smallmat.hpp, wave_reduce_transposed) next to a
// plain v_add_f32, gfx950: eight independent chains per lane, 256-thread workgroups at 1 / 2 / 4 wavefronts per SIMD on every CU.
//   MODE 0 v_add_f32   1 v_permlane32_swap_b32   2 v_permlane16_swap_b32   3 v_add_f32_dpp row_ror:8   4 v_add_f32_dpp quad_perm
//   5 v_mov_b32_dpp row_half_mirror + v_add_f32 (two instructions)   6 ds_bpermute_b32   7 ds_swizzle_b32 (swap 16)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/xlane_rate.hip -o /tmp/xlane_rate && /tmp/xlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8];
  for (int u = 0; u < 8; ++u) a[u] = seed + float(threadIdx.x) * 1e-3f + float(u);
  const int addr = ((threadIdx.x ^ 16) & 63) * 4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 8; u += 2) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a[u + 1])); asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[u + 1]), "+v"(a[u])); }
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 8; u += 2) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a[u + 1])); asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[u + 1]), "+v"(a[u])); }
      } else if constexpr (MODE == 3) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
      } else if constexpr (MODE == 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(a[(u + 1) & 7]));
      } else if constexpr (MODE == 5) {
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          float t;
          asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[u + 1]));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(t));
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          float t;
          asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[u]));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u + 1]) : "v"(t));
        }
      } else if constexpr (MODE == 6) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[u])));
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[u]), 0x401F));
      }
    }
  }
  float r = 0;
  for (int u = 0; u < 8; ++u) r += a[u];
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, int waves) {
  float* d; hipMalloc(&d, 4);
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 2000, blocks = cus * waves;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 10, 1.f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters, 1.f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per_simd = double(iters) * 32 * waves;   // wave-instructions (MODE 5: pairs of instructions) issued on each SIMD
  printf("%-34s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, waves, ms, ms * 1e-3 * 2.4e9 / per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f32", w); run<1>("v_permlane32_swap_b32", w); run<2>("v_permlane16_swap_b32", w); run<3>("v_add_f32_dpp row_ror:8", w);
    run<4>("v_add_f32_dpp quad_perm", w); run<5>("v_mov_dpp half_mirror + v_add (pair)", w); run<6>("ds_bpermute_b32", w); run<7>("ds_swizzle_b32", w);
  }
  return 0;
}
