// Instruction budget of the SPD backward's per-pair phases (review item 2a): every phase of the row loop of
// spd_pdist_bwd_kernel (csrc/spd_pair.hpp) as its own kernel around the SAME device function, compiled with the library's
// flags and counted in the disassembly by tools/phase_budget.py (never run: the numbers are static instruction counts).
// Row operands (L_i^-1, L_i) come through uniform addresses, as in the kernel (scalar loads, SGPR operands); the column
// operands are one record per lane.  `base_*` kernels move the same records without arithmetic: their count is subtracted.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "../../matrix-manifolds_amd/csrc/smallmat.hpp"
#include "../../matrix-manifolds_amd/csrc/spd_ws.hpp"
#include "../../matrix-manifolds_amd/csrc/spd_pair.hpp"

using namespace mm;

template <typename T, int K> __device__ __forceinline__ void ld(const T* p, T (&v)[K]) {
  const T* q = p + (blockIdx.x * 64 + threadIdx.x) * K;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = q[k];
}
template <typename T, int K> __device__ __forceinline__ void st(T* p, const T (&v)[K]) {
  T* q = p + (blockIdx.x * 64 + threadIdx.x) * K;
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = v[k];
}
template <typename T, int K> __device__ __forceinline__ void ldu(const T* p, int row, T (&v)[K]) {   // uniform
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = p[row * K + k];
}

#define PHASE(NAME, T, D) extern "C" __global__ __launch_bounds__(64) void NAME(const T* __restrict__ rowp, int row, const T* __restrict__ in, T* __restrict__ out)

template <typename T, int D> __device__ __forceinline__ void base_np_np(const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T a[NP];
  ld<T, NP>(in, a);
  st<T, NP>(out, a);
}
template <typename T, int D> __device__ __forceinline__ void congr(const T* rowp, int row, const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T li[NP], lj[NP], a[NP];
  ldu<T, NP>(rowp, row, li);
  ld<T, NP>(in, lj);
  congr_chol<T, D>(li, lj, a);
  st<T, NP>(out, a);
}
template <typename T, int D> __device__ __forceinline__ void gate(const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T a[NP];
  ld<T, NP>(in, a);
  a[0] = close_gate<T, D>(a);        // (the other entries pass through: same stores as the base kernel)
  st<T, NP>(out, a);
}
template <typename T, int D, int WHICH> __device__ __forceinline__ void logm(const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T a[NP], m0[NP];
  ld<T, NP>(in, a);
  const T pre = a[0] + a[0];
  if constexpr (WHICH == 0) log_close<T, D>(a, m0, pre);
  else if constexpr (WHICH == 1) { if constexpr (D == 3) log_series3_centred<T>(a, m0, pre); else log_series4_centred<T>(a, m0, pre); }
  else m0[0] += log_cayley<T, D>(a, m0, pre);
  st<T, NP>(out, m0);
}
template <typename T, int D> __device__ __forceinline__ void colcongr(const T* rowp, int row, const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T li[NP], lc[NP], m[NP], acc[D][D];
  ldu<T, NP>(rowp, row, li);
  ldu<T, NP>(rowp, row + 1, lc);
  ld<T, NP>(in, m);
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) acc[r][c] = m[(r * D + c) % NP];   // (accumulators that are alive)
  lt_m_lt_acc<T, D>(li, lc, m, acc);
  T o[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) o[k] = T(0);
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) o[(r * D + c) % NP] += acc[r][c];   // (D*D - NP extra adds: reported with the phase)
  st<T, NP>(out, o);
}
template <typename T, int D> __device__ __forceinline__ void rowred(const T* in, T* out) {
  constexpr int NP = Packed<D>::NP;
  T m[NP];
  ld<T, NP>(in, m);
  m[0] = wave_reduce_transposed<NP, T>(m, int(threadIdx.x));
  st<T, NP>(out, m);
}

#define ALL(T, TN, D)                                                                  \
  PHASE(base_##TN##D, T, D) { base_np_np<T, D>(in, out); }                              \
  PHASE(congr_##TN##D, T, D) { congr<T, D>(rowp, row, in, out); }                       \
  PHASE(gate_##TN##D, T, D) { gate<T, D>(in, out); }                                    \
  PHASE(logclose_##TN##D, T, D) { logm<T, D, 0>(in, out); }                             \
  PHASE(logcayley_##TN##D, T, D) { logm<T, D, 2>(in, out); }                            \
  PHASE(colcongr_##TN##D, T, D) { colcongr<T, D>(rowp, row, in, out); }                 \
  PHASE(rowred_##TN##D, T, D) { rowred<T, D>(in, out); }
ALL(float, f, 3)
ALL(float, f, 4)
ALL(double, d, 3)
ALL(double, d, 4)
PHASE(logcentred_f3, float, 3) { logm<float, 3, 1>(in, out); }
PHASE(logcentred_f4, float, 4) { logm<float, 4, 1>(in, out); }
