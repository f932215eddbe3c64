// The maps of the vector manifolds that the optimizers call (egrad2rgrad, norm, exp / retr, transport:
// manifolds/{euclidean,lorentz,sphere}.py, base.py) on a point that is zero-padded to a compile-time width and held in
// registers: every loop unrolls, nothing lives in scratch.  The padding coordinates are zero and stay zero under every map.
// Used by the fused step kernels (vec_step.hip, product_pairs.hip); vec.hip's per-point kernels take any m <= 64 at run time.
#pragma once
#include <hip/hip_runtime.h>

#include "adam.hpp"
#include "smallmat.hpp"
#include "vecfn.hpp"

namespace mm {

template <typename T, int KIND, int MP> struct PadRule {
  using N = Num<T>;
  // <u, v>: Minkowski for the hyperboloid (lorentz.py:101-118), Euclidean otherwise
  static __device__ __forceinline__ T dot(const T (&u)[MP], const T (&v)[MP]) {
    if (KIND == MM_LORENTZ) {
      T s = T(0);
#pragma unroll
      for (int k = 1; k < MP; ++k) s = N::fma(u[k], v[k], s);
      return N::fma(-u[0], v[0], s);
    }
    T s = T(0);
#pragma unroll
    for (int k = 0; k < MP; ++k) s = N::fma(u[k], v[k], s);
    return s;
  }
  static __device__ __forceinline__ T edot(const T (&u)[MP], const T (&v)[MP]) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < MP; ++k) s = N::fma(u[k], v[k], s);
    return s;
  }
  // Manifold.norm (base.py:29-33)
  static __device__ __forceinline__ T norm(const T (&u)[MP]) { return N::sqrt(N::max(dot(u, u), T(kEps))); }
  // egrad2rgrad: lorentz.py:52-57 (flip the time coordinate, then u + <x,u>_L x), sphere.py:41-44, identity
  static __device__ __forceinline__ void rgrad(const T (&xp)[MP], const T (&g)[MP], T (&o)[MP]) {
    if (KIND == MM_EUCLIDEAN) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = g[k];
    } else if (KIND == MM_LORENTZ) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = (k == 0) ? -g[k] : g[k];
      const T d = dot(xp, o);
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(d, xp[k], o[k]);
    } else {
      const T d = edot(xp, g);
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(-d, xp[k], g[k]);
    }
  }
  // exp / retr: lorentz.py:59-62 (retr == exp, base.py:49-50), sphere.py:51-59, x + u
  static __device__ __forceinline__ void step(const T (&xp)[MP], const T (&u)[MP], int exact, T (&o)[MP]) {
    if (KIND == MM_EUCLIDEAN) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = xp[k] + u[k];
    } else if (KIND == MM_LORENTZ) {
      const T un = N::max(N::sqrt(N::max(dot(u, u), T(0))), T(kEps));
      const T ch = ::cosh(un), sh = ::sinh(un) / un;
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(xp[k], ch, sh * u[k]);
    } else {
      const T nu = N::sqrt(N::max(edot(u, u), T(kEps)));
      if (exact && nu > T(kEps)) {
        const T c = ::cos(nu), s = ::sin(nu) / nu;
#pragma unroll
        for (int k = 0; k < MP; ++k) o[k] = N::fma(xp[k], c, s * u[k]);
      } else {
        T nn = T(0);
#pragma unroll
        for (int k = 0; k < MP; ++k) { o[k] = xp[k] + u[k]; nn = N::fma(o[k], o[k], nn); }
        const T inv = T(1) / N::sqrt(N::max(nn, T(kEps)));
#pragma unroll
        for (int k = 0; k < MP; ++k) o[k] *= inv;
      }
    }
  }
  // transport of a tangent b from x to y: lorentz.py:79-82; sphere: proju(y, b) (base.py:65-66); identity
  static __device__ __forceinline__ void transport(const T (&xp)[MP], const T (&y)[MP], T (&b)[MP]) {
    if (KIND == MM_LORENTZ) {
      const T xy = dot(xp, y), uy = dot(b, y);
      const T g = uy / (T(1) - xy);
#pragma unroll
      for (int k = 0; k < MP; ++k) b[k] = N::fma(g, xp[k] + y[k], b[k]);
    } else if (KIND == MM_SPHERE) {
      const T d = edot(y, b);
#pragma unroll
      for (int k = 0; k < MP; ++k) b[k] = N::fma(-d, y[k], b[k]);
    }
  }
};

// zero-padded row p of a [n][m] table (requests from clamped addresses, masked afterwards: a load under `k < m` would sit in
// its own basic block)
template <typename T, int MP> __device__ __forceinline__ void load_padded(const T* t, int64_t p, int m, T (&o)[MP]) {
#pragma unroll
  for (int k = 0; k < MP; ++k) o[k] = t[p * m + min(k, m - 1)];
#pragma unroll
  for (int k = 0; k < MP; ++k) o[k] = k < m ? o[k] : T(0);
}
template <typename T, int MP> __device__ __forceinline__ void store_row(T* t, int64_t p, int m, const T (&v)[MP]) {
#pragma unroll
  for (int k = 0; k < MP; ++k)
    if (k < m) t[p * m + k] = v[k];
}

// ---- the optimizer rules on such a point (optim/rsgd.py:29-82, optim/radam.py:62-98) -----------------------------------
enum { VRULE_RSGD = 0, VRULE_MOMENTUM = 1, VRULE_ADAM = 2 };
template <typename T> struct VecRuleArgs {
  T lr, momentum, dampening, max_grad_norm;   // max_grad_norm <= 0: no clipping
  int exact;
  T* state0; T* state1;                        // momentum buffer / exp_avg, exp_avg_sq: [n][m] tables, updated in place
  AdamArgs<T> adam;
};
// New point `o` of the point xp (row p of its [n][m] table) with Euclidean gradient g.  `in` = the row exists (the state rows
// of a padding thread are read from row 0 and not written).  beta2 / alpha: adam_coeffs of this step (RULE == VRULE_ADAM).
template <typename T, int KIND, int MP, int RULE>
__device__ __forceinline__ void pad_rule_point(const T (&xp)[MP], const T (&g)[MP], int64_t p, int m, bool in,
                                               const VecRuleArgs<T>& a, T beta2, T alpha, T (&o)[MP]) {
  using N = Num<T>;
  using R = PadRule<T, KIND, MP>;
  T r[MP];
  R::rgrad(xp, g, r);
  if constexpr (RULE == VRULE_RSGD) {             // rsgd.py:63-68, 82
    T scale = -a.lr;
    if (a.max_grad_norm > T(0)) scale *= N::min(a.max_grad_norm / R::norm(r), T(1));
#pragma unroll
    for (int k = 0; k < MP; ++k) r[k] *= scale;
    R::step(xp, r, a.exact, o);
  } else if constexpr (RULE == VRULE_MOMENTUM) {  // rsgd.py:70-80
    T b[MP];
    load_padded<T, MP>(a.state0, p, m, b);
    const T clip = a.max_grad_norm > T(0) ? N::min(a.max_grad_norm / R::norm(r), T(1)) : T(1);
#pragma unroll
    for (int k = 0; k < MP; ++k) {
      b[k] = N::fma(a.momentum, b[k], (T(1) - a.dampening) * (r[k] * clip));
      r[k] = -a.lr * b[k];
    }
    R::step(xp, r, a.exact, o);
    R::transport(xp, o, b);
    if (in) store_row<T, MP>(a.state0, p, m, b);
  } else {                                        // radam.py:62-98
    T mo[MP];
    load_padded<T, MP>(a.state0, p, m, mo);
    const T nrm = R::norm(r);
    const T clip = a.adam.max_grad_norm > T(0) ? N::min(a.adam.max_grad_norm / nrm, T(1)) : T(1);
    const T v = N::fma(beta2, a.state1[p * m], (T(1) - beta2) * nrm * nrm);
    const T f = -alpha / (N::sqrt(v) + a.adam.eps);
#pragma unroll
    for (int k = 0; k < MP; ++k) {
      mo[k] = N::fma(a.adam.beta1, mo[k], (T(1) - a.adam.beta1) * (r[k] * clip));
      r[k] = mo[k] * f;
    }
    R::step(xp, r, a.adam.exact, o);
    R::transport(xp, o, mo);
    if (in) {
      store_row<T, MP>(a.state0, p, m, mo);
#pragma unroll
      for (int k = 0; k < MP; ++k)
        if (k < m) a.state1[p * m + k] = v;
    }
  }
}

}  // namespace mm
