// Small dense linear algebra in registers for gfx950 (one matrix per lane).
//
// Everything here is templated on the scalar type T (float / double) and the
// compile-time size D, written so that after full unrolling every array index
// is a constant and the matrices live in VGPRs (or SGPRs when wave-uniform).
//
// Conventions
//   * Symmetric matrices are stored packed, lower triangle row-major:
//       idx(r,c) = r(r+1)/2 + c   (r >= c),   NP = D(D+1)/2 entries.
//   * Lower-triangular matrices (Cholesky factors and their inverses) use the
//     same packing.
//   * Full matrices are T[D*D] row-major.
#pragma once
#include <hip/hip_runtime.h>

namespace mm {

template <int D> struct Packed { static constexpr int NP = D * (D + 1) / 2; };

__host__ __device__ constexpr int pidx(int r, int c) {
  return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r;
}

// ---------------------------------------------------------------- scalar math
template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float eps() { return 5.9604645e-8f; }
  static __device__ __forceinline__ float tiny() { return 1e-37f; }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float log(float x) { return ::logf(x); }
  static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
  static __device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
  static __device__ __forceinline__ float max(float a, float b) { return ::fmaxf(a, b); }
  static __device__ __forceinline__ float min(float a, float b) { return ::fminf(a, b); }
  static __device__ __forceinline__ float copysign(float a, float b) { return ::copysignf(a, b); }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static constexpr int kMaxSweeps = 8;
};
template <> struct Num<double> {
  static __device__ __forceinline__ double eps() { return 1.1102230246251565e-16; }
  static __device__ __forceinline__ double tiny() { return 1e-300; }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
  static __device__ __forceinline__ double copysign(double a, double b) { return ::copysign(a, b); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static constexpr int kMaxSweeps = 12;
};

// ---------------------------------------------------------- load / symmetrize
// Packed symmetric part of a full row-major DxD matrix in memory.
template <typename T, int D>
__device__ __forceinline__ void load_sym_packed(const T* __restrict__ m, T (&s)[Packed<D>::NP]) {
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c)
      s[pidx(r, c)] = (r == c) ? m[r * D + c] : T(0.5) * (m[r * D + c] + m[c * D + r]);
}

template <typename T, int D>
__device__ __forceinline__ void store_sym_full(T* __restrict__ m, const T (&s)[Packed<D>::NP]) {
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) m[r * D + c] = s[pidx(r, c)];
}

// ------------------------------------------------------------------ Cholesky
// L (lower, packed) with X = L L^T.  A non-positive pivot yields NaN (sqrt of a
// negative) and `ok` is cleared — the caller reports it (reference: torch's
// cholesky raises, linalg/torch_batch.py:43-48).
template <typename T, int D>
__device__ __forceinline__ bool cholesky(const T (&x)[Packed<D>::NP], T (&l)[Packed<D>::NP]) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T s = x[pidx(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= l[pidx(j, k)] * l[pidx(j, k)];
    ok = ok && (s > T(0));
    T ljj = Num<T>::sqrt(s);
    l[pidx(j, j)] = ljj;
    T inv = T(1) / ljj;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T t = x[pidx(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= l[pidx(i, k)] * l[pidx(j, k)];
      l[pidx(i, j)] = t * inv;
    }
  }
  return ok;
}

// Inverse of a lower-triangular packed matrix.
template <typename T, int D>
__device__ __forceinline__ void invert_lower(const T (&l)[Packed<D>::NP], T (&li)[Packed<D>::NP]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
    li[pidx(j, j)] = T(1) / l[pidx(j, j)];
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T s = T(0);
#pragma unroll
      for (int k = j; k < i; ++k) s += l[pidx(i, k)] * li[pidx(k, j)];
      li[pidx(i, j)] = -s / l[pidx(i, i)];
    }
  }
}

// ------------------------------------------------------------- congruences
// out = Lw S Lw^T  (Lw lower-triangular packed, S symmetric packed).
template <typename T, int D, typename TL>
__device__ __forceinline__ void congr_lower(const TL (&lw)[Packed<D>::NP], const T (&s)[Packed<D>::NP],
                                            T (&out)[Packed<D>::NP]) {
  T b[D][D];  // B = Lw S
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = lw[pidx(r, 0)] * s[pidx(0, c)];
#pragma unroll
      for (int k = 1; k <= r; ++k) acc = Num<T>::fma(lw[pidx(r, k)], s[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = b[r][0] * lw[pidx(c, 0)];
#pragma unroll
      for (int k = 1; k <= c; ++k) acc = Num<T>::fma(b[r][k], lw[pidx(c, k)], acc);
      out[pidx(r, c)] = acc;
    }
}

// out = Lw^T S Lw  (Lw lower-triangular packed, S symmetric packed).
template <typename T, int D, typename TL>
__device__ __forceinline__ void congr_lower_t(const TL (&lw)[Packed<D>::NP], const T (&s)[Packed<D>::NP],
                                              T (&out)[Packed<D>::NP]) {
  T b[D][D];  // B = S Lw : B[r][c] = sum_{k>=c} S[r][k] Lw[k][c]
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = s[pidx(r, c)] * lw[pidx(c, c)];
#pragma unroll
      for (int k = c + 1; k < D; ++k) acc = Num<T>::fma(s[pidx(r, k)], lw[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)  // out = Lw^T B : out[r][c] = sum_{k>=r} Lw[k][r] B[k][c]
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = lw[pidx(r, r)] * b[r][c];
#pragma unroll
      for (int k = r + 1; k < D; ++k) acc = Num<T>::fma(lw[pidx(k, r)], b[k][c], acc);
      out[pidx(r, c)] = acc;
    }
}

// out = F S F^T for a full row-major DxD matrix F (symmetric result, packed).
template <typename T, int D>
__device__ __forceinline__ void congr_full(const T (&f)[D * D], const T (&s)[Packed<D>::NP],
                                           T (&out)[Packed<D>::NP]) {
  T b[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = Num<T>::fma(f[r * D + k], s[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = Num<T>::fma(b[r][k], f[c * D + k], acc);
      out[pidx(r, c)] = acc;
    }
}

// ---------------------------------------------------------- Jacobi eigensolve
// Cyclic Jacobi on a packed symmetric matrix, in place: on exit diag(a) holds
// the eigenvalues and (if WITH_V) the columns of v the eigenvectors,
// A = V diag(w) V^T.
//
// The sweep loop is wave-uniform (runs while any lane of the wavefront is
// unconverged) so there is no divergence; a lane that has converged applies
// exact identity rotations (t = 0 -> c = 1, s = 0), which leave its registers
// bit-for-bit unchanged.  A lane's result therefore depends only on its own
// matrix, never on its wave-mates: results are reproducible under any tiling
// or sharding of the pair list.
//
// Convergence test: off(A)^2 <= eps^2 * ||diag(A)||^2 — the residual coupling
// is then below the rounding error already committed when A was formed.
template <typename T, int D, bool WITH_V>
__device__ __forceinline__ void jacobi_eig(T (&a)[Packed<D>::NP], T (&v)[D][D]) {
  using N = Num<T>;
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) v[r][c] = (r == c) ? T(1) : T(0);
  }
  if (D == 1) return;
  const T tol2 = N::eps() * N::eps();
  for (int sweep = 0; sweep < N::kMaxSweeps; ++sweep) {
    T off2 = T(0), dg2 = T(0);
#pragma unroll
    for (int r = 0; r < D; ++r) {
      dg2 = N::fma(a[pidx(r, r)], a[pidx(r, r)], dg2);
#pragma unroll
      for (int c = 0; c < r; ++c) off2 = N::fma(a[pidx(r, c)], a[pidx(r, c)], off2);
    }
    const bool active = off2 > tol2 * dg2;
    if (!__any(active)) break;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        const T apq = a[pidx(q, p)];
        const T h = a[pidx(q, q)] - a[pidx(p, p)];
        // t = sgn(h) 2 apq / (|h| + sqrt(h^2 + 4 apq^2))  (smaller root)
        const T two_apq = apq + apq;
        const T den = N::abs(h) + N::sqrt(N::fma(h, h, two_apq * two_apq));
        // (sgn(0) := +1 gives the 45-degree rotation when app == aqq; apq == 0
        //  gives t = 0 because den >= tiny.)
        T t = ((h < T(0)) ? -two_apq : two_apq) * N::rcp(N::max(den, N::tiny()));
        t = active ? t : T(0);
        const T c = N::rsqrt(N::fma(t, t, T(1)));
        const T s = t * c;
        a[pidx(p, p)] = N::fma(-t, apq, a[pidx(p, p)]);
        a[pidx(q, q)] = N::fma(t, apq, a[pidx(q, q)]);
        a[pidx(q, p)] = active ? T(0) : apq;
#pragma unroll
        for (int r = 0; r < D; ++r) {
          if (r == p || r == q) continue;
          const T arp = a[pidx(r, p)], arq = a[pidx(r, q)];
          a[pidx(r, p)] = N::fma(c, arp, -s * arq);
          a[pidx(r, q)] = N::fma(s, arp, c * arq);
        }
        if (WITH_V) {
#pragma unroll
          for (int r = 0; r < D; ++r) {
            const T vrp = v[r][p], vrq = v[r][q];
            v[r][p] = N::fma(c, vrp, -s * vrq);
            v[r][q] = N::fma(s, vrp, c * vrq);
          }
        }
      }
    }
  }
}

// out (packed) = V diag(f) V^T
template <typename T, int D>
__device__ __forceinline__ void vdvt(const T (&v)[D][D], const T (&f)[D], T (&out)[Packed<D>::NP]) {
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = Num<T>::fma(v[r][k] * f[k], v[c][k], acc);
      out[pidx(r, c)] = acc;
    }
}

// ------------------------------------------------------------ wave reductions
template <typename T> __device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace mm
