// Checks the cross-lane primitives of the transposing reduction (csrc/smallmat.hpp) on hardware:
// lane_xor<0..3> (DPP quad_perm / row_shl,shr:4 / row_ror:8), butterfly16/32 (v_permlane16/32_swap),
// and wave_sum_transposed8/16 in fp32 and fp64 against a host sum.
//   hipcc -O3 --offload-arch=gfx950 -I../../matrix-manifolds_amd/csrc t_perm.hip -o t_perm && ./t_perm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "smallmat.hpp"

using namespace mm;

template <typename T> __global__ void k_xor(const T* in, T* out) {
  const int l = threadIdx.x;
  const T x = in[l];
  out[0 * 64 + l] = lane_xor<0>(x);
  out[1 * 64 + l] = lane_xor<1>(x);
  out[2 * 64 + l] = lane_xor<2>(x);
  out[3 * 64 + l] = lane_xor<3>(x);
  out[4 * 64 + l] = butterfly16(x) - x;
  out[5 * 64 + l] = butterfly32(x) - x;
}

template <typename T, int N> __global__ void k_sum(const T* in, T* out) {
  const int l = threadIdx.x;
  T v[N];
  for (int k = 0; k < N; ++k) v[k] = in[k * 64 + l];
  T tot;
  constexpr int W = N <= 8 ? 8 : 16;
  if constexpr (N <= 8) tot = wave_sum_transposed8<N>(v, l); else tot = wave_sum_transposed16<N>(v, l);
  out[l] = tot;
  out[64 + l] = T(transposed_index<W>(l));
}

template <typename T> int run(const char* name) {
  int bad = 0;
  T *din, *dout;
  hipMalloc(&din, sizeof(T) * 64 * 16);
  hipMalloc(&dout, sizeof(T) * 64 * 16);
  std::vector<T> h(64 * 16), o(64 * 16);
  for (int i = 0; i < 64 * 16; ++i) h[i] = T(1 + (i * 37 % 101)) / T(7);
  hipMemcpy(din, h.data(), sizeof(T) * 64 * 16, hipMemcpyHostToDevice);
  k_xor<T><<<1, 64>>>(din, dout);
  hipMemcpy(o.data(), dout, sizeof(T) * 64 * 16, hipMemcpyDeviceToHost);
  const int xm[6] = {1, 2, 4, 8, 16, 32};
  for (int t = 0; t < 6; ++t)
    for (int l = 0; l < 64; ++l)
      if (t < 4 ? o[t * 64 + l] != h[l ^ xm[t]]  // pure moves: exact; butterflies return x + partner: rounded
                : std::fabs(double(o[t * 64 + l]) - double(h[l ^ xm[t]])) > 1e-5 * std::fabs(double(h[l ^ xm[t]]))) { if (bad < 5) printf("%s xor %d lane %d: got %g want %g\n", name, xm[t], l, double(o[t * 64 + l]), double(h[l ^ xm[t]])); ++bad; }
  auto check_sum = [&](int N, auto kernel) {
    kernel<<<1, 64>>>(din, dout);
    hipMemcpy(o.data(), dout, sizeof(T) * 128, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const int k = int(o[64 + l]);
      double want = 0;
      if (k < N) for (int q = 0; q < 64; ++q) want += double(h[k * 64 + q]);
      if (std::fabs(double(o[l]) - want) > 1e-4 * (1 + std::fabs(want))) { if (bad < 5) printf("%s sum N=%d lane %d idx %d: got %g want %g\n", name, N, l, k, double(o[l]), want); ++bad; }
    }
  };
  check_sum(6, k_sum<T, 6>);
  check_sum(10, k_sum<T, 10>);
  check_sum(15, k_sum<T, 15>);
  printf("%s: %s\n", name, bad ? "FAILED" : "ok");
  hipFree(din); hipFree(dout);
  return bad;
}

int main() { return (run<float>("fp32") + run<double>("fp64")) ? 1 : 0; }
