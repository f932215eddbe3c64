// Per-pair scalar functions of the vector manifolds, shared by the VALU kernels (vec.hip) and the
// matrix-core kernels (vec_gram.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include <cstdint>

#include "smallmat.hpp"

namespace mm {

constexpr double kEps = 1e-8;  // utils.py:13 (both precisions)

// padded point dimension the pair kernels are instantiated for
constexpr int pad_dim(int m) { return m <= 4 ? 4 : m <= 8 ? 8 : m <= 12 ? 12 : m <= 16 ? 16 : m <= 24 ? 24 : m <= 32 ? 32 : m <= 48 ? 48 : 64; }

// Symmetric VALU backward (vec_sym.hip): prep (padded points, cleared gradient) + pair kernel flushing into the gradient
// `ws` = acc [pad+1][n] | loss slots [2][256] | xpad [n+1][pad].  MM_ERR_UNSUPPORTED outside its range (fp32 m <= 32,
// fp64 m <= 16 — Euclidean: m <= 32 in fp64 too): the caller then takes the ordered-pair kernel.
bool vec_sym_supports(int dtype, int kind, int m);
// *finalized: the gradient (and, with a loss, loss_out) is complete — the pair kernel flushes into `grad` directly.
// mode: VSYM_PLAIN as above.  The training-step forms (vec_step.hpp) leave the sums — already mapped to gradient
// contributions — in the workspace, gacc [n][m] at the head of the accumulator region, for the step's per-point kernel, which
// moves them to the gradient, closes the loss record and clears both; VSYM_STEP_PREPARED also skips the preparation launch
// (that kernel left the padded copy of the new points and clean accumulators behind).
enum { VSYM_PLAIN = 0, VSYM_STEP = 1, VSYM_STEP_PREPARED = 2 };
int vec_sym_backward_pairs(int dtype, int kind, int loss_kind, int squared, const void* x, const void* g, int64_t n, int m,
                           int64_t rb, int64_t re, void* ws, const void* scale_raw, double alpha, double eps, int terms,
                           const double* loss_params, void* grad, void* loss_out, bool* finalized, hipStream_t st,
                           int mode = VSYM_PLAIN);
// the regions of that workspace (pad = pad_dim(m))
template <typename T> struct VecSymWs {
  T* acc;     // [pad + 1][n] elements reserved; the training-step form uses its head as gacc [n][m]
  T* slots;   // [2][kLossSlots]
  T* xpad;    // [n + 1][pad]
  VecSymWs(void* ws, int64_t n, int pad) : acc(static_cast<T*>(ws)), slots(acc + size_t(n) * (pad + 1)), xpad(slots + 2 * 256) {}
};

template <typename T> __device__ __forceinline__ T acos_t(T c);
template <> __device__ __forceinline__ float acos_t<float>(float c) { return ::acosf(c); }
template <> __device__ __forceinline__ double acos_t<double>(double c) { return ::acos(c); }

// Forward value and d(out)/d(inner quantity) for one pair.
//   Euclidean: q = sum (x_i - x_j)^2          out = max(q,eps) [sqrt]      base.py:29-33,56-57
//   Lorentz  : q = -<x_i,x_j>_L               out = max(acosh(max(q,1)),eps)^2   lorentz.py:72-77
//   Sphere   : q = <x_i,x_j>                  out = max(acos(clamp q),eps)^2     sphere.py:68-74
template <typename T, int KIND> struct PairFn {
  static __device__ __forceinline__ T value(T q, int squared) {
    using N = Num<T>;
    if (KIND == MM_EUCLIDEAN) {
      const T s = N::max(q, T(kEps));
      return squared ? s : N::sqrt(s);
    } else if (KIND == MM_LORENTZ) {
      const T t = N::max(q, T(1));
      const T z = N::sqrt(N::fma(t, t, T(-1)));
      const T d = N::max(N::log(t + z), T(kEps));
      return squared ? d * d : d;
    } else {
      const T c = N::min(N::max(q, T(-1 + 1e-16)), T(1 - 1e-16));
      const T th = N::max(acos_t<T>(c), T(kEps));
      return squared ? th * th : th;
    }
  }
  // d(out)/dq  (value clamps are gradient-transparent, as in the reference)
  static __device__ __forceinline__ T dq(T q, int squared) {
    using N = Num<T>;
    if (KIND == MM_EUCLIDEAN) {
      return squared ? T(1) : T(0.5) * N::rsqrt(N::max(q, T(kEps)));
    } else if (KIND == MM_LORENTZ) {
      const T t = N::max(q, T(1));
      const T z = N::sqrt(N::fma(t, t, T(-1)));
      const T d = N::max(N::log(t + z), T(kEps));
      const T dz = N::rcp(N::max(z, T(kEps)));  // lorentz.py:134-138 (this clamp IS in the backward)
      return squared ? (d + d) * dz : dz;
    } else {
      const T c = N::min(N::max(q, T(-1 + 1e-16)), T(1 - 1e-16));
      const T th = N::max(acos_t<T>(c), T(kEps));
      // the reference divides by sqrt(1-c^2) unguarded (inf at c = +-1); we floor it at eps
      const T ds = -N::rcp(N::max(N::sqrt(N::fma(-c, c, T(1))), T(kEps)));
      return squared ? (th + th) * ds : ds;
    }
  }
};

}  // namespace mm
