// Hardware check of the transposing wavefront reduction (csrc/smallmat.hpp: wave_reduce_transposed / reduce_slot):
// every lane must end with the exact wavefront total of the value reduce_slot names, and every value must have
// exactly one writer lane.  Integer-valued inputs, so sums are exact in fp32 and fp64.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 t_reduce.hip -o t_reduce && ./t_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include "../../matrix-manifolds_amd/csrc/smallmat.hpp"

template <typename T, int N> __global__ void k(T* out, int* slot, int* wr) {
  const int lane = threadIdx.x & 63;
  T v[N];
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = T((lane * 7 + q * 13) % 31 - 9 + q);
  const T tot = mm::wave_reduce_transposed<N, T>(v, lane);
  bool w;
  const int s = mm::reduce_slot<N>(lane, w);
  out[lane] = tot; slot[lane] = s; wr[lane] = w ? 1 : 0;
}

template <typename T, int N> int run() {
  T* out; int *slot, *wr;
  hipMalloc(&out, 64 * sizeof(T)); hipMalloc(&slot, 256); hipMalloc(&wr, 256);
  k<T, N><<<1, 64>>>(out, slot, wr);
  T ho[64]; int hs[64], hw[64];
  hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  hipMemcpy(hs, slot, sizeof(hs), hipMemcpyDeviceToHost);
  hipMemcpy(hw, wr, sizeof(hw), hipMemcpyDeviceToHost);
  int bad = 0, writers[64] = {0};
  for (int l = 0; l < 64; ++l) {
    if (hs[l] < 0 || hs[l] >= N) { ++bad; continue; }
    double ref = 0;
    for (int m = 0; m < 64; ++m) ref += double((m * 7 + hs[l] * 13) % 31 - 9 + hs[l]);
    if (double(ho[l]) != ref) ++bad;
    if (hw[l]) ++writers[hs[l]];
  }
  for (int q = 0; q < N; ++q) if (writers[q] != 1) ++bad;
  printf("%s N=%2d: %s\n", std::is_same<T, float>::value ? "f32" : "f64", N, bad ? "FAIL" : "ok");
  hipFree(out); hipFree(slot); hipFree(wr);
  return bad;
}

int main() {
  int bad = 0;
  bad += run<float, 1>(); bad += run<float, 2>(); bad += run<float, 3>(); bad += run<float, 6>(); bad += run<float, 7>();
  bad += run<float, 10>(); bad += run<float, 15>(); bad += run<float, 21>(); bad += run<float, 28>(); bad += run<float, 36>();
  bad += run<float, 45>(); bad += run<float, 64>();
  bad += run<double, 3>(); bad += run<double, 6>(); bad += run<double, 10>(); bad += run<double, 15>(); bad += run<double, 21>();
  bad += run<double, 45>();
  printf(bad ? "FAILED\n" : "all ok\n");
  return bad ? 1 : 0;
}
