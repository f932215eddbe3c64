// Fused objective epilogue shared by the pair kernels (spd.hip, vec.hip): the loss term of one pair
// and its derivative, evaluated in registers — graphembed/graphembed/objectives.py:16-45 on
// m = softplus(scale) * d2 (modules.py:84-88).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/mm_manifolds.h"
#include "smallmat.hpp"

namespace mm {

constexpr int kLossSlots = 256;  // fused-loss partial sums are spread over this many addresses

template <typename T> struct LossArgs {
  const T* scale_raw;  // device scalar: raw scale parameter (softplus applied here); null -> 1
  T alpha, eps;        // quotient loss: target * alpha, 1 / (epoch + 1)
  int terms;           // quotient loss: bit 0 = |m/(a g) - 1|, bit 1 = |a g/(m + eps) - 1|
  T* slots;            // [2][kLossSlots]
  // optional device array {alpha, eps}: when set it overrides the two values above, so that a captured graph
  // of a training step follows the quotient loss's per-epoch eps = 1/(epoch+1) without being re-recorded
  const double* dyn;
};
template <typename T, int LOSS> __device__ __forceinline__ void loss_resolve(LossArgs<T>& la) {
  if constexpr (LOSS == MM_LOSS_QUOTIENT) {
    if (la.dyn) { la.alpha = T(la.dyn[0]); la.eps = T(la.dyn[1]); }
  }
}
template <typename T> __device__ __forceinline__ T softplus_of(const T* raw) {
  if (!raw) return T(1);
  const T v = *raw;  // torch.nn.functional.softplus: beta = 1, threshold = 20
  if (std::is_same<T, float>::value) return v > T(20) ? v : T(::log1pf(::expf(float(v))));
  return v > T(20) ? v : T(::log1p(::exp(double(v))));
}
template <typename T> __device__ __forceinline__ T sign_of(T q) { return q > T(0) ? T(1) : (q < T(0) ? T(-1) : T(0)); }
// the loss term of one pair and (in dldm) d loss / d m
template <typename T, int LOSS>
__device__ __forceinline__ T loss_term(T m, T target, const LossArgs<T>& la, T& dldm) {
  if constexpr (LOSS == MM_LOSS_STRESS) {
    const T r = m - target;
    dldm = r + r;
    return r * r;
  } else {
    const T ag = target * la.alpha;
    T l = T(0);
    dldm = T(0);
    if (la.terms & 1) {
      // (Num<T>::rcp: v_rcp_f32, 1 ulp, in fp32 — an IEEE division is ten instructions per pair; 1 / x in fp64)
      const T inv = Num<T>::rcp(ag), q = m * inv - T(1);
      l += Num<T>::abs(q);
      dldm += sign_of(q) * inv;
    }
    if (la.terms & 2) {
      const T inv = Num<T>::rcp(m + la.eps), q = ag * inv - T(1);
      l += Num<T>::abs(q);
      dldm -= sign_of(q) * ag * inv * inv;
    }
    return l;
  }
}

// Sums the kLossSlots partial sums (fp64), writes loss_out = {loss, d loss / d scale_raw} and leaves the
// slots clean.  Called by the first wavefront of block 0 of a finalize kernel.
template <typename T>
__device__ __forceinline__ void loss_finalize(T* __restrict__ slots, const T* __restrict__ scale_raw,
                                              T* __restrict__ loss_out) {
  double l = 0.0, d = 0.0;
  for (int t = threadIdx.x; t < kLossSlots; t += 64) {
    l += double(slots[t]); d += double(slots[kLossSlots + t]);
    slots[t] = T(0); slots[kLossSlots + t] = T(0);
  }
  l = wave_sum(l); d = wave_sum(d);
  if (threadIdx.x == 0) {
    const double v = scale_raw ? double(*scale_raw) : 0.0;
    loss_out[0] = T(l);
    loss_out[1] = scale_raw ? T(d / (1.0 + ::exp(-v))) : T(0);  // d softplus = sigmoid
  }
}

}  // namespace mm
