// The optimizer rules of an SPD point — packed symmetric X and Euclidean gradient in, new point out — as device functions:
// the bodies of spd.hip's per-point kernels (mm_spd_rsgd_step, ...), of its fused step kernel, and of the product
// embedding's step kernel (product_pairs.hip).  The same arithmetic whichever way a step is issued.
#pragma once
#include <hip/hip_runtime.h>

#include "adam.hpp"
#include "smallmat.hpp"

namespace mm {

// sym(X + U + 1/2 (L^-1 U)^T (L^-1 U)) with U symmetric packed (spd.py:146-154)
template <typename T, int D>
__device__ __forceinline__ void spd_retr(const T (&xs)[Packed<D>::NP], const T (&li)[Packed<D>::NP],
                                         const T (&us)[Packed<D>::NP], T (&out)[Packed<D>::NP]) {
  T b[D][D];  // B = L^-1 U
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k <= r; ++k) acc = Num<T>::fma(li[pidx(r, k)], us[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = Num<T>::fma(b[k][r], b[k][c], acc);
      out[pidx(r, c)] = xs[pidx(r, c)] + us[pidx(r, c)] + T(0.5) * acc;
    }
}

// L f(L^-1 U L^-T) L^T with f = exp or log applied to the eigenvalues
template <typename T, int D, bool IS_LOG>
__device__ __forceinline__ void spd_explog(const T (&l)[Packed<D>::NP], const T (&li)[Packed<D>::NP],
                                           const T (&us)[Packed<D>::NP], T (&out)[Packed<D>::NP]) {
  T a[Packed<D>::NP], v[D][D], f[D], fa[Packed<D>::NP];
  congr_lower<T, D>(li, us, a);
  jacobi_eig<T, D, true>(a, v);
#pragma unroll
  for (int k = 0; k < D; ++k) f[k] = IS_LOG ? Num<T>::log(a[pidx(k, k)]) : Num<T>::exp(a[pidx(k, k)]);
  vdvt<T, D>(v, f, fa);
  congr_lower<T, D>(l, fa, out);
}

// ---- the optimizer rules on one point (packed symmetric X and Euclidean gradient in, new point out) ------------------
// rgrad = X sym(G) X (spd.py:134-135), ||rgrad||_X = ||L^-1 rgrad L^-T||_F (spd.py:113-117), exp / second-order retraction
// (spd.py:137-154).  Each is the body of one per-point kernel below and of the fused step kernel further down: the same
// arithmetic whichever way a step is issued.
template <typename T, int D>
__device__ __forceinline__ T spd_rgrad_setup(const T (&xs)[Packed<D>::NP], const T (&gs)[Packed<D>::NP],
                                             T (&r)[Packed<D>::NP], T (&l)[Packed<D>::NP], T (&li)[Packed<D>::NP],
                                             bool want_norm) {
  constexpr int NP = Packed<D>::NP;
  T xf[D * D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) xf[a * D + c] = xs[pidx(a, c)];
  congr_full<T, D>(xf, gs, r);  // Riemannian gradient X sym(G) X
  cholesky<T, D>(xs, l);
  invert_lower<T, D>(l, li);
  T s = T(0);
  if (want_norm) {
    T a[NP];
    congr_lower<T, D>(li, r, a);
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int c = 0; c <= p; ++c) s += (p == c ? T(1) : T(2)) * a[pidx(p, c)] * a[pidx(p, c)];
  }
  return s;   // ||r||_X^2 (0 if not asked for)
}

// momentum-free RSGD (rsgd.py:63-68, 82)
template <typename T, int D>
__device__ __forceinline__ void spd_rsgd_update(const T (&xs)[Packed<D>::NP], const T (&gs)[Packed<D>::NP], T lr,
                                                T max_grad_norm, int exact, T (&o)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  T r[NP], l[NP], li[NP];
  const T s = spd_rgrad_setup<T, D>(xs, gs, r, l, li, max_grad_norm > T(0));
  T scale = -lr;
  if (max_grad_norm > T(0)) scale *= Num<T>::min(max_grad_norm / Num<T>::sqrt(s), T(1));
#pragma unroll
  for (int q = 0; q < NP; ++q) r[q] *= scale;
  if (exact) spd_explog<T, D, false>(l, li, r, o);
  else spd_retr<T, D>(xs, li, r, o);
}

// heavy-ball variant (rsgd.py:70-80): buf = momentum buf + (1 - dampening) rgrad, x' = exp/retr(x, -lr buf); the
// SPD transport is the identity (spd.py:196-199); buf is kept symmetric and updated in place.
template <typename T, int D>
__device__ __forceinline__ void spd_momentum_update(const T (&xs)[Packed<D>::NP], const T (&gs)[Packed<D>::NP],
                                                    T (&b)[Packed<D>::NP], T lr, T momentum, T dampening, T max_grad_norm,
                                                    int exact, T (&o)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  T r[NP], l[NP], li[NP];
  const T nn = spd_rgrad_setup<T, D>(xs, gs, r, l, li, max_grad_norm > T(0));
  T clip = T(1);
  if (max_grad_norm > T(0)) clip = Num<T>::min(max_grad_norm / Num<T>::sqrt(nn), T(1));
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    b[q] = Num<T>::fma(momentum, b[q], (T(1) - dampening) * (r[q] * clip));
    r[q] = -lr * b[q];
  }
  if (exact) spd_explog<T, D, false>(l, li, r, o);
  else spd_retr<T, D>(xs, li, r, o);
}

// Riemannian Adam (radam.py:62-98) — see vec_radam_step_kernel; the SPD transport is the identity (spd.py:196-199),
// exp_avg is kept symmetric.  vprev / return value: the point's second-moment scalar.
template <typename T, int D>
__device__ __forceinline__ T spd_adam_update(const T (&xs)[Packed<D>::NP], const T (&gs)[Packed<D>::NP],
                                             T (&mo)[Packed<D>::NP], T vprev, const AdamArgs<T>& a, T beta2, T alpha,
                                             T (&o)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  T r[NP], l[NP], li[NP];
  const T nn = spd_rgrad_setup<T, D>(xs, gs, r, l, li, true);   // ||r||_X^2 (no floor, unlike Manifold.norm)
  const T nrm = Num<T>::sqrt(nn);
  const T clip = a.max_grad_norm > T(0) ? Num<T>::min(a.max_grad_norm / nrm, T(1)) : T(1);
  const T v = Num<T>::fma(beta2, vprev, (T(1) - beta2) * nrm * nrm);
  const T f = -alpha / (Num<T>::sqrt(v) + a.eps);
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    mo[q] = Num<T>::fma(a.beta1, mo[q], (T(1) - a.beta1) * (r[q] * clip));
    r[q] = mo[q] * f;
  }
  if (a.exact) spd_explog<T, D, false>(l, li, r, o);
  else spd_retr<T, D>(xs, li, r, o);
  return v;
}

}  // namespace mm
