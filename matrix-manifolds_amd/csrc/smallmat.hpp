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

#include <cstdint>
#include <type_traits>
#include <utility>

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
  static __device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
  // v_log_f32 (log2) * ln2: max rel. error 1.6e-7 on gfx950, same as libm logf, ~10x fewer
  // instructions (tools/micro/log_accuracy.hip)
  static __device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
  static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
  static __device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
  static __device__ __forceinline__ float max(float a, float b) { return ::fmaxf(a, b); }
  // max without the two canonicalising `v_max x, x, x` that fmaxf gets for operands the compiler cannot prove quiet
  // (a kernel argument, a value merged from several paths); same result: the non-NaN operand
  // (b wave-uniform: a scalar register operand)
  static __device__ __forceinline__ float max_raw_s(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
  }
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
  // v_rcp_f64 (~26 bits) + two Newton steps: the last bit of a NORMAL, well-scaled argument without the IEEE division's
  // scaling / fix-up instructions (5 against ~12); callers pass quantities bounded away from 0 and infinity
  static __device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = ::fma(::fma(-x, r, 1.0), r, r);
    return ::fma(::fma(-x, r, 1.0), r, r);
  }
  static __device__ __forceinline__ double max_raw_s(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
  }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
  static __device__ __forceinline__ double copysign(double a, double b) { return ::copysign(a, b); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static constexpr int kMaxSweeps = 12;
};

// ------------------------------------------------- compile-time loops, constants
// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The per-column loops of the pair
// kernels are written with it and NOT as `#pragma unroll` loops: a loop is unrolled late, after inlining, and until then
// the per-column arrays are indexed by a variable — they are not split into registers early, and the kernels came out
// with up to twice the vector registers (fp32 SPD(3) backward: 152 instead of 87 for one column).
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// a b + TAB[I] with the double-precision table entry as a SCALAR operand, materialised where it is used.
// The series below carry 14 - 36 coefficients each.  Left to the compiler they are hoisted out of the row loop into
// vector registers, two per coefficient, for the whole kernel: the fp64 SPD(3) backward held 68 registers of constants
// (232 in all: two wavefronts per SIMD), the forward 72 (184) — and round 3 dropped the fp64 recentred series for that
// reason alone.  Handing the compiler the constant in scalar registers does not help: it selects the accumulating form
// v_fmac_f64 and copies the constant into the destination with two v_mov_b32.  So the multiply-add of a Horner step is ONE
// asm statement: two s_mov_b32 with literals into a fixed scalar pair and v_fma_f64 with that pair as its addend — an fp64
// multiply-add keeps the vector pipe busy for twice the issue slot of an instruction, the scalar moves issue in its shadow.
// fp32 constants stay with the compiler (16 registers at most, and the fp32 kernels are bound by instruction issue).
template <uint64_t BITS> __device__ __forceinline__ double fma_sconst64(double a, double b) {
  double r;
  asm("s_mov_b32 s100, %3\n\ts_mov_b32 s101, %4\n\tv_fma_f64 %0, %1, %2, s[100:101]"
      : "=v"(r) : "v"(a), "v"(b), "n"(uint32_t(BITS)), "n"(uint32_t(BITS >> 32)) : "s100", "s101");
  return r;
}
// TAB: a coefficient table type with `static constexpr T at(int)` and kTerms.  The entry is read in a constant expression
// (the index is a template argument): a table reached through a POINTER template argument is an odr-used device variable —
// HIP emits it as an externally initialised __constant__, its loads are not folded, and the whole table was fetched with
// one scalar load, spilled and reloaded in front of every use.
template <typename T, typename TAB, int I> __device__ __forceinline__ T fma_coef(T a, T b) {
  constexpr T c = TAB::at(I);
#ifndef MM_F64_CONST_VGPR   // (A/B builds: the compiler's placement)
  if constexpr (std::is_same<T, double>::value) return fma_sconst64<__builtin_bit_cast(uint64_t, c)>(a, b);
  else
#endif
    return Num<T>::fma(a, b, c);
}
// Horner's rule in the quotient ring R[E]/(chi_E) of a 3x3 matrix, chi_E = x^3 - s1 x^2 + s2 x - s3: on exit
// a0 I + a1 E + a2 E^2 = TAB[0] I + TAB[1] E + ... + TAB[N-1] E^(N-1).  One step, (a0, a1, a2) . E + c I =
// (a2 s3 + c, a0 - a2 s2, a1 + a2 s1), is three multiply-adds — two when E is traceless (s1 = 0).
template <typename T, typename TAB, bool TRACELESS = false>
__device__ __forceinline__ void ring_horner3(T s1, T s2, T s3, T& a0, T& a1, T& a2) {
  constexpr int N = TAB::kTerms;
  static_assert(N >= 3, "");
  constexpr T top0 = TAB::at(N - 3), top1 = TAB::at(N - 2), top2 = TAB::at(N - 1);
  a0 = top0; a1 = top1; a2 = top2;
  static_for<N - 3>([&](auto ic) {
    constexpr int k = N - 4 - decltype(ic)::value;
    const T n0 = fma_coef<T, TAB, k>(a2, s3), n1 = Num<T>::fma(-a2, s2, a0);
    T n2;
    if constexpr (TRACELESS) n2 = a1; else n2 = Num<T>::fma(a2, s1, a1);
    a0 = n0; a1 = n1; a2 = n2;
  });
}
// ... of a 4x4 matrix, chi_E = x^4 - s1 x^3 + s2 x^2 - s3 x + s4: (h0..h3) . E + c I = (c - h3 s4, h0 + h3 s3, h1 - h3 s2, h2 + h3 s1)
template <typename T, typename TAB>
__device__ __forceinline__ void ring_horner4(T s1, T s2, T s3, T s4, T& h0, T& h1, T& h2, T& h3) {
  constexpr int N = TAB::kTerms;
  static_assert(N >= 4, "");
  constexpr T top0 = TAB::at(N - 4), top1 = TAB::at(N - 3), top2 = TAB::at(N - 2), top3 = TAB::at(N - 1);
  h0 = top0; h1 = top1; h2 = top2; h3 = top3;
  static_for<N - 4>([&](auto ic) {
    constexpr int k = N - 5 - decltype(ic)::value;
    const T n0 = fma_coef<T, TAB, k>(-h3, s4), n1 = Num<T>::fma(h3, s3, h0), n2 = Num<T>::fma(-h3, s2, h1),
            n3 = Num<T>::fma(h3, s1, h2);
    h0 = n0; h1 = n1; h2 = n2; h3 = n3;
  });
}

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

// out = (Li Lj)(Li Lj)^T for packed lower-triangular Li, Lj: with Li = chol(X_i)^-1 and Lj = chol(X_j) this
// is L_i^-1 X_j L_i^-T in 2 x D(D+1)(D+2)/6 FMAs (20 for D = 3, 40 for D = 4) instead of 28 / 60 through
// the full product with X_j, and symmetric positive semi-definite by construction.
template <typename T, int D, typename TL>
__device__ __forceinline__ void congr_chol(const TL (&li)[Packed<D>::NP], const T (&lj)[Packed<D>::NP],
                                           T (&out)[Packed<D>::NP]) {
  T b[Packed<D>::NP];  // B = Li Lj (lower): B[r][c] = sum_{c<=k<=r} Li[r][k] Lj[k][c]
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = li[pidx(r, c)] * lj[pidx(c, c)];
#pragma unroll
      for (int k = c + 1; k <= r; ++k) acc = Num<T>::fma(li[pidx(r, k)], lj[pidx(k, c)], acc);
      b[pidx(r, c)] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = b[pidx(r, 0)] * b[pidx(c, 0)];
#pragma unroll
      for (int k = 1; k <= c; ++k) acc = Num<T>::fma(b[pidx(r, k)], b[pidx(c, k)], acc);
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

// out (full DxD, not symmetric) = Li^T M Lc^T  for packed lower Li, Lc and packed symmetric M.
// With Li = L^-1 and Lc = L of the row point this is L^-T M L^T, the column-side gradient of a
// pair up to the right factor X_j^-1 that is common to a whole column (spd.hip).
template <typename T, int D, typename TL>
__device__ __forceinline__ void lt_m_lt(const TL (&li)[Packed<D>::NP], const TL (&lc)[Packed<D>::NP],
                                        const T (&m)[Packed<D>::NP], T (&out)[D][D]) {
  T b[D][D];  // B = Li^T M : B[r][c] = sum_{k>=r} Li[k][r] M[k][c]
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = li[pidx(r, r)] * m[pidx(r, c)];
#pragma unroll
      for (int k = r + 1; k < D; ++k) acc = Num<T>::fma(li[pidx(k, r)], m[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)  // out = B Lc^T : out[r][c] = sum_{k<=c} B[r][k] Lc[c][k]
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T acc = b[r][0] * lc[pidx(c, 0)];
#pragma unroll
      for (int k = 1; k <= c; ++k) acc = Num<T>::fma(b[r][k], lc[pidx(c, k)], acc);
      out[r][c] = acc;
    }
}

// acc += Li^T M Lc^T, the sum carried by the accumulator through the second product's FMA chains (no separate adds)
template <typename T, int D, typename TL>
__device__ __forceinline__ void lt_m_lt_acc(const TL (&li)[Packed<D>::NP], const TL (&lc)[Packed<D>::NP],
                                            const T (&m)[Packed<D>::NP], T (&acc)[D][D]) {
  T b[D][D];  // B = Li^T M : B[r][c] = sum_{k>=r} Li[k][r] M[k][c]
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      T t = li[pidx(r, r)] * m[pidx(r, c)];
#pragma unroll
      for (int k = r + 1; k < D; ++k) t = Num<T>::fma(li[pidx(k, r)], m[pidx(k, c)], t);
      b[r][c] = t;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int k = 0; k <= c; ++k) acc[r][c] = Num<T>::fma(b[r][k], lc[pidx(c, k)], acc[r][c]);
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
// Rotation for the pair (p,q), written to need only two transcendental issues
// (v_rsq_f32 costs ~3.5x a plain VALU op on gfx950, tools/micro/valu_rate.hip):
//   h = aqq - app,  r = rsq(h^2 + 4 apq^2),  cos 2t = |h| r,
//   c = cos t = sqrt((1 + cos 2t)/2) = x rsq(x),   s = sgn(h) apq r / c,   tan t = s / c
// (|t| <= pi/4, the small-angle root).  |h| carries a +1e-15 so that h = apq = 0
// yields exactly the identity (c = 1, s = 0) with no branch; 1e-15 squares to a
// normal float, so rsq never sees a flushed denormal.
//
// The sweep loop is wave-uniform: it runs while any lane of the wavefront has
//   off(A)^2 > tol2 * ||diag(A)||^2         (REL = false: any symmetric matrix), or
//   a_pq^2  > tol2 * |a_pp a_qq| for some pq (REL = true: positive-definite input).
// A lane that has converged keeps rotating by ~eps angles, which is harmless; which
// lanes share a wavefront is fixed by the global column / point index alone (tiles
// are anchored there), so results do not depend on how the pair list is sharded.
// tol2 = eps^2 when eigenvectors are used (the residual coupling is then below the
// rounding already committed in forming A); eigenvalue-only callers may pass eps,
// because symmetric functions of the spectrum are second-order in the residual.
template <typename T, int D, bool WITH_V, bool REL = false>
__device__ __forceinline__ void jacobi_eig(T (&a)[Packed<D>::NP], T (&v)[D][D], T tol2) {
  using N = Num<T>;
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) v[r][c] = (r == c) ? T(1) : T(0);
  }
  if (D == 1) return;
  // Fully unrolled (registers are renamed sweep to sweep instead of being copied around a
  // loop back-edge: -16 % kernel time on MI355X), and no convergence test before the
  // sweeps every 3x3+ matrix needs anyway (one sweep is exact only for D = 2).
  constexpr int kMinSweeps = D >= 3 ? 2 : 1;
  // (D >= 6: the sweeps stay a loop — unrolled, a 9x9 solve with eigenvectors is ~30 000 instructions per call site —
  // and get a few more of them: the quadratic convergence of cyclic Jacobi starts later for larger matrices)
  constexpr int kSweeps = D <= 5 ? N::kMaxSweeps : N::kMaxSweeps + 4;
#ifdef MM_JACOBI_ROLL_F64   // (A/B builds: the fp64 sweeps as a loop — the unrolled fp64 backward is 73 KB of code)
  constexpr int kUnrollSweeps = (D <= 5 && !std::is_same<T, double>::value) ? N::kMaxSweeps : 1;
#else
  constexpr int kUnrollSweeps = D <= 5 ? N::kMaxSweeps : 1;
#endif
#pragma unroll kUnrollSweeps
  for (int sweep = 0; sweep < kSweeps; ++sweep) {
    bool active = sweep < kMinSweeps;
    if (active) {
    } else if (REL) {
      // Demmel-Veselic criterion for positive-definite matrices: every coupling small
      // relative to ITS OWN two diagonal entries, so small eigenvalues of a wide
      // spectrum come out with high relative accuracy (log w needs exactly that).
#pragma unroll
      for (int r = 1; r < D; ++r)
#pragma unroll
        for (int c = 0; c < r; ++c)
          active = active || (a[pidx(r, c)] * a[pidx(r, c)] > tol2 * N::abs(a[pidx(r, r)] * a[pidx(c, c)]));
    } else {
      T off2 = T(0), dg2 = T(0);
#pragma unroll
      for (int r = 0; r < D; ++r) {
        dg2 = N::fma(a[pidx(r, r)], a[pidx(r, r)], dg2);
#pragma unroll
        for (int c = 0; c < r; ++c) off2 = N::fma(a[pidx(r, c)], a[pidx(r, c)], off2);
      }
      active = off2 > tol2 * dg2;
    }
    if (!__any(active)) break;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        const T apq = a[pidx(q, p)];
        const T h = a[pidx(q, q)] - a[pidx(p, p)];
        const T ah = N::abs(h) + T(1e-15);
        const T sa_ = (h < T(0)) ? -apq : apq;         // sgn(h) apq   (h = 0 counts as +)
        const T sa2 = sa_ + sa_;
        const T r = N::rsqrt(N::fma(ah, ah, sa2 * sa2));
        const T x = N::fma(ah * r, T(0.5), T(0.5));     // cos^2 t
        const T ci = N::rsqrt(x);
        const T c = x * ci;
        const T s = (sa_ * r) * ci;
        const T t = s * ci;
        a[pidx(p, p)] = N::fma(-t, apq, a[pidx(p, p)]);
        a[pidx(q, q)] = N::fma(t, apq, a[pidx(q, q)]);
        a[pidx(q, p)] = T(0);
#pragma unroll
        for (int r2 = 0; r2 < D; ++r2) {
          if (r2 == p || r2 == q) continue;
          const T arp = a[pidx(r2, p)], arq = a[pidx(r2, q)];
          a[pidx(r2, p)] = N::fma(c, arp, -s * arq);
          a[pidx(r2, q)] = N::fma(s, arp, c * arq);
        }
        if (WITH_V) {
#pragma unroll
          for (int r2 = 0; r2 < D; ++r2) {
            const T vrp = v[r2][p], vrq = v[r2][q];
            v[r2][p] = N::fma(c, vrp, -s * vrq);
            v[r2][q] = N::fma(s, vrp, c * vrq);
          }
        }
      }
    }
  }
}
template <typename T, int D, bool WITH_V>
__device__ __forceinline__ void jacobi_eig(T (&a)[Packed<D>::NP], T (&v)[D][D]) {
  jacobi_eig<T, D, WITH_V>(a, v, Num<T>::eps() * Num<T>::eps());
}

// One-sided Jacobi (Hestenes) on a D x D matrix g: on exit b = g V has mutually orthogonal columns, i.e. g = U diag(sigma) V^T
// with sigma_k = ||b_k|| and u_k = b_k / sigma_k.  The rotations are those of the two-sided method on g^T g (the formulas of
// jacobi_eig above) but applied to g itself, so the SMALL singular values keep their relative accuracy: through the
// eigenvalues of g^T g anything below sqrt(eps) sigma_max is rounding noise.  Two users (round 4):
//   * Grassmann principal angles (mat.hip): g = x^T y — a cosine next to 0 came out 3e-4 off in fp32;
//   * the SPD pair matrix A = B B^T, B = L_i^-1 L_j (pair_core, spd_pair.hpp): g = B^T, so V = the eigenvectors of A and
//     sigma_k^2 its eigenvalues — in fp32, forming A first costs eps cond(A) of relative accuracy in its small eigenvalues
//     (1.3e-2 of d^2 at cond(X) = 1e4, garbage at 1e6), the rotations on B keep 3e-7 / 3e-6 (emulated with LAPACK's SVD).
// Wave-uniform sweep loop: runs while any lane has a pair of columns with <b_p, b_q>^2 > tol2 ||b_p||^2 ||b_q||^2.
template <typename T, int D, bool WITH_V>
__device__ __forceinline__ void svd_onesided(const T (&g)[D][D], T (&b)[D][D], T (&v)[D][D], T tol2) {
  using N = Num<T>;
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      b[r][c] = g[r][c];
      if (WITH_V) v[r][c] = (r == c) ? T(1) : T(0);
    }
  if constexpr (D == 1) return;
  for (int sweep = 0; sweep < N::kMaxSweeps + 4; ++sweep) {
    bool active = false;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        T app = T(0), aqq = T(0), apq = T(0);
#pragma unroll
        for (int r = 0; r < D; ++r) {
          app = N::fma(b[r][p], b[r][p], app);
          aqq = N::fma(b[r][q], b[r][q], aqq);
          apq = N::fma(b[r][p], b[r][q], apq);
        }
        active = active || (apq * apq > tol2 * (app * aqq));
        const T h = aqq - app;
        const T ah = N::abs(h) + T(1e-15);
        const T sa_ = (h < T(0)) ? -apq : apq;
        const T sa2 = sa_ + sa_;
        const T rr = N::rsqrt(N::fma(ah, ah, sa2 * sa2));
        const T x = N::fma(ah * rr, T(0.5), T(0.5));     // cos^2 t
        const T ci = N::rsqrt(x);
        const T c = x * ci;
        const T sn = (sa_ * rr) * ci;
#pragma unroll
        for (int r = 0; r < D; ++r) {
          const T bp = b[r][p], bq = b[r][q];
          b[r][p] = N::fma(c, bp, -sn * bq);
          b[r][q] = N::fma(sn, bp, c * bq);
          if (WITH_V) {
            const T vp = v[r][p], vq = v[r][q];
            v[r][p] = N::fma(c, vp, -sn * vq);
            v[r][q] = N::fma(sn, vp, c * vq);
          }
        }
      }
    }
    if (!__any(active)) break;
  }
}

// ------------------------------------------ closed-form eigenvalues, 3x3 (fp32)
// Trigonometric solution of the characteristic cubic (the method of the reference's
// fast.symeig3x3, linalg/fast.py:75-91, without its eps fudge terms): q = tr/3,
// B = (A - qI)/p with p^2 = ||A - qI||_F^2 / 6, phi = acos(det(B)/2)/3,
// w = q + p {2cos(phi), -cos(phi) + sqrt3 sin(phi), -cos(phi) - sqrt3 sin(phi)}.
// acos / cos / sin are polynomials (abs err < 5e-8) — no libm calls, ~75 VALU ops
// in all versus ~200 for Jacobi.  Eigenvalues come out with ABSOLUTE error
// ~eps*||A||; callers needing small eigenvalues to high RELATIVE accuracy (wide
// spectra) must fall back to Jacobi — see spd_pdist_fwd_kernel.
__device__ __forceinline__ void eig3_trig(const float (&a)[6], float (&w)[3]) {
  const float q = (a[pidx(0, 0)] + a[pidx(1, 1)] + a[pidx(2, 2)]) * (1.f / 3.f);
  const float b00 = a[pidx(0, 0)] - q, b11 = a[pidx(1, 1)] - q, b22 = a[pidx(2, 2)] - q;
  const float o10 = a[pidx(1, 0)], o20 = a[pidx(2, 0)], o21 = a[pidx(2, 1)];
  float p2 = fmaf(b00, b00, fmaf(b11, b11, b22 * b22));
  p2 = fmaf(2.f, fmaf(o10, o10, fmaf(o20, o20, o21 * o21)), p2) * (1.f / 6.f);
  const float ip = __builtin_amdgcn_rsqf(fmaxf(p2, 1e-30f));
  const float p = p2 * ip;
  const float c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c10 = o10 * ip, c20 = o20 * ip, c21 = o21 * ip;
  float hd = c00 * fmaf(c11, c22, -c21 * c21) - c10 * fmaf(c10, c22, -c21 * c20) + c20 * fmaf(c10, c21, -c11 * c20);
  hd = fminf(fmaxf(0.5f * hd, -1.f), 1.f);
  // acos(|x|) = sqrt(1-|x|) P7(|x|)   (Abramowitz & Stegun 4.4.46, |err| <= 2e-8)
  const float ax = fabsf(hd);
  float pa = fmaf(-0.0012624911f, ax, 0.0066700901f);
  pa = fmaf(pa, ax, -0.0170881256f);
  pa = fmaf(pa, ax, 0.0308918810f);
  pa = fmaf(pa, ax, -0.0501743046f);
  pa = fmaf(pa, ax, 0.0889789874f);
  pa = fmaf(pa, ax, -0.2145988016f);
  pa = fmaf(pa, ax, 1.5707963050f);
  const float ac = __builtin_amdgcn_sqrtf(1.f - ax) * pa;
  const float phi = ((hd < 0.f) ? 3.14159265358979f - ac : ac) * (1.f / 3.f);   // in [0, pi/3]
  const float z = phi * phi;
  float cs = fmaf(z, -2.7557319e-7f, 2.4801587e-5f);
  cs = fmaf(cs, z, -1.3888889e-3f);
  cs = fmaf(cs, z, 4.1666667e-2f);
  cs = fmaf(cs, z, -0.5f);
  cs = fmaf(cs, z, 1.f);
  float sn = fmaf(z, -2.5052108e-8f, 2.7557319e-6f);
  sn = fmaf(sn, z, -1.9841270e-4f);
  sn = fmaf(sn, z, 8.3333333e-3f);
  sn = fmaf(sn, z, -1.6666667e-1f);
  sn = fmaf(sn * z, phi, phi);
  const float r3s = 1.7320508f * sn;
  w[2] = fmaf(p, cs + cs, q);
  w[1] = fmaf(p, r3s - cs, q);
  w[0] = fmaf(p, -r3s - cs, q);
}

// ------------------------------ log(A) near the identity, 3x3 / 4x4 (fp32)
// For ||A - I||_F <= 0.3 (the two points of the pair are closer than ~0.3 — every pair at the
// reference's initialisation) log(A) is evaluated WITHOUT an eigen-decomposition: with E = A - I and an
// economised polynomial log(1+x) = x p(x) on [-0.3,0.3], p(E) is evaluated by HORNER'S RULE IN THE QUOTIENT
// RING R[E]/(chi_E): every intermediate is alpha0 I + alpha1 E + alpha2 E^2, and multiplying by E uses the
// Cayley-Hamilton identity E^3 = s1 E^2 - s2 E + s3 I,
//     (alpha0, alpha1, alpha2) . E + c I = (alpha2 s3 + c,  alpha0 - alpha2 s2,  alpha1 + alpha2 s1),
// three FMAs per coefficient (the power-coefficient recurrence of round 1 needed six), the first two steps
// being constants.  ~60 VALU ops in all, branch-free, no transcendental.  Returns ||E||_F^2 for the gate.
// p of degree 7 interpolated at Chebyshev nodes of |x| <= 0.3 (max error 9.1e-8 |x|,
// tools/design/series_fit.py): the close-pair gate bounds the spectral radius of E by 0.3.
// `pre` multiplies the result (the caller's 2g when it is known before the series: three multiplications
// on the coefficients instead of six on the matrix).
// fp32: p of degree 7 interpolated at Chebyshev nodes of |x| <= 0.3 (max error 9.1e-8 |x|, tools/design/series_fit.py);
// fp64: degree 19 (7.7e-18 |x|, tools/design/series_fit64.py, extended-precision fit).
template <typename T> struct LogSeries;
template <> struct LogSeries<float> {
  static constexpr int kTerms = 8;
  static constexpr float kA[kTerms] = {9.999999337e-01f, -4.999999402e-01f, 3.333568549e-01f, -2.500212282e-01f,
                                       1.987095623e-01f, -1.655023071e-01f, 1.650813480e-01f, -1.450413880e-01f};
  static constexpr float at(int i) { return kA[i]; }
};
template <> struct LogSeries<double> {
  static constexpr int kTerms = 20;
  static constexpr double kA[kTerms] = {
    1.00000000000000000e+00, -4.99999999999999944e-01, 3.33333333333343085e-01, -2.50000000000014155e-01,
    1.99999999996232608e-01, -1.66666666663750795e-01, 1.42857143404437337e-01, -1.25000000386766896e-01,
    1.11111071374215928e-01, -9.99999703591583217e-02, 9.09107326619259803e-02, -8.33346464041358065e-02,
    7.68821017515587124e-02, -7.13938378893102243e-02, 6.72936081364212679e-02, -6.30553787420979894e-02,
    5.31216751520414421e-02, -5.03380364382464973e-02, 8.02156472153527783e-02, -7.58480457039158590e-02};
  static constexpr double at(int i) { return kA[i]; }
};

template <typename T> __device__ __forceinline__ T log_series3(const T (&a)[6], T (&m0)[6], T pre = T(1)) {
  using N = Num<T>;
  using S = LogSeries<T>;
  const T e00 = a[pidx(0, 0)] - T(1), e11 = a[pidx(1, 1)] - T(1), e22 = a[pidx(2, 2)] - T(1);
  const T e10 = a[pidx(1, 0)], e20 = a[pidx(2, 0)], e21 = a[pidx(2, 1)];
  // E^2 (symmetric)
  const T f00 = N::fma(e00, e00, N::fma(e10, e10, e20 * e20));
  const T f11 = N::fma(e10, e10, N::fma(e11, e11, e21 * e21));
  const T f22 = N::fma(e20, e20, N::fma(e21, e21, e22 * e22));
  const T f10 = N::fma(e10, e00, N::fma(e11, e10, e21 * e20));
  const T f20 = N::fma(e20, e00, N::fma(e21, e10, e22 * e20));
  const T f21 = N::fma(e20, e10, N::fma(e21, e11, e22 * e21));
  const T tr2 = f00 + f11 + f22;                      // ||E||_F^2
  const T s1 = e00 + e11 + e22;
  const T s2 = T(0.5) * N::fma(s1, s1, -tr2);
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  // Horner from the top: after the first two steps alpha = (c_{n-3}, c_{n-2}, c_{n-1})
  T a0, a1, a2;
  ring_horner3<T, S>(s1, s2, s3, a0, a1, a2);
  // log(I + E) = E p(E): one more multiplication by E (no constant), then the caller's factor
  const T b0 = (a2 * s3) * pre, b1 = N::fma(-a2, s2, a0) * pre, b2 = N::fma(a2, s1, a1) * pre;
  m0[pidx(0, 0)] = N::fma(b2, f00, N::fma(b1, e00, b0));
  m0[pidx(1, 1)] = N::fma(b2, f11, N::fma(b1, e11, b0));
  m0[pidx(2, 2)] = N::fma(b2, f22, N::fma(b1, e22, b0));
  m0[pidx(1, 0)] = N::fma(b2, f10, b1 * e10);
  m0[pidx(2, 0)] = N::fma(b2, f20, b1 * e20);
  m0[pidx(2, 1)] = N::fma(b2, f21, b1 * e21);
  return tr2;
}

// ---- log(A) of a pair at moderate distance: the RECENTRED series (3x3, fp32) -----------------------------------
// (fp64 was built and dropped: degree 33 / 35 reach 7e-16, but the compiler keeps the 34 - 36 double constants in 68 - 72
// vector registers across the row loop — the fp64 backward went from 218 to 256 registers, two wavefronts per SIMD to one.
// tools/design/series_fit_wide64.py keeps the fit.)
// log A = log(mu) I + log(I + E'),  E' = A / mu - I  with  mu = tr A / 3 (the scalar that minimises ||A - mu I||_F).
// E' is traceless, so its spectral radius is at most sqrt(2/3) ||E'||_F: the gate ||E'||_F^2 <= kCentredGate3 bounds it
// by 0.66, where log(1+x) = x p(x) holds to 4e-8 |x| with p of degree 15 (Chebyshev-interpolated on |x| <= 0.66,
// tools/design/series_fit_wide.py).  Same evaluation as log_series3 — Horner's rule in R[E']/(chi_E') — but with
// s1 = tr E' = 0 a step is TWO multiply-adds: (a0, a1, a2) . E' + c I = (a2 s3 + c, a0 - a2 s2, a1).  ~95 instructions
// with one reciprocal and one logarithm, against ~185 for the Cayley-transform logarithm below: embeddings whose pair
// distances are a few tenths (||log X|| ~ 0.35: training after the first epochs) stay eigen-free AND inverse-free.
// `pre` multiplies the result.
constexpr double kCentredGate3 = 0.6534;   // (2/3) * 0.6534 = 0.66^2
template <typename T> struct LogSeriesWide;
template <> struct LogSeriesWide<float> {
  static constexpr int kTerms = 16;
  static constexpr float kA[kTerms] = {9.999999847e-01f, -4.999999854e-01f, 3.333378158e-01f, -2.500042795e-01f,
                                       1.997873233e-01f, -1.664636323e-01f, 1.466440033e-01f, -1.286147090e-01f,
                                       7.866430935e-02f, -6.903477397e-02f, 2.371349367e-01f, -2.228347962e-01f,
                                       -2.667691364e-01f, 2.562672116e-01f, 4.274908187e-01f, -4.061057313e-01f};
  static constexpr float at(int i) { return kA[i]; }
};
// fp64 (round 5; tools/design/series_fit_wide64.py 36 36): degree 35, 6.5e-17 |x| as a polynomial, 2.8e-16 through Horner's
// rule in fp64.  Used by the MATRIX form only (log_series_mat, SPD(5 .. 9)): its coefficient loop reads the table, whereas the
// unrolled ring forms of SPD(3) / SPD(4) would hold the 36 constants in 72 vector registers (see log_series3_centred).
template <> struct LogSeriesWide<double> {
  static constexpr int kTerms = 36;
  static constexpr double kA[kTerms] = {
    0.999999999999999977, -0.499999999999999978, 0.333333333333367457, -0.250000000000033421,
    0.199999999991592943, -0.166666666658432825, 0.142857143675747047, -0.125000000801733994,
    0.111111069196670012, -0.0999999589496586145, 0.0909103925504135392, -0.0833346081274524372,
    0.0768964148318060428, -0.071402459598029521, 0.0670466571631664423, -0.0628721421125401977,
    0.054920735255657598, -0.0517334464941176604, 0.0821825618685663011, -0.0789393499687729317,
    -0.119544506131984601, 0.11824425305853179, 0.753586818196905653, -0.737031529750978713,
    -2.21838624927452971, 2.17293423563498954, 5.34587266623974133, -5.23378983222008806,
    -8.94716963622682544, 8.7602754349000368, 10.4325804141314891, -10.2127945466777319,
    -7.42237802920880408, 7.26540890455107326, 2.58226980159039129, -2.52672226697783483};
  static constexpr double at(int i) { return kA[i]; }
};
// true if the pair is OUTSIDE the recentred series' range (NaN counts as outside): ||A - mu I||_F^2 > kCentredGate3 mu^2.
// Evaluated from A alone, in the branch that needs it (rows that failed the close-pair gate): the close-pair path keeps
// nothing alive for it.
template <typename T> __device__ __forceinline__ bool centred_far3(const T (&a)[6]) {
  using N = Num<T>;
  const T mu = (a[pidx(0, 0)] + a[pidx(1, 1)] + a[pidx(2, 2)]) * T(1.0 / 3.0);
  const T d0 = a[pidx(0, 0)] - mu, d1 = a[pidx(1, 1)] - mu, d2 = a[pidx(2, 2)] - mu;
  const T off = N::fma(a[pidx(1, 0)], a[pidx(1, 0)], N::fma(a[pidx(2, 0)], a[pidx(2, 0)], a[pidx(2, 1)] * a[pidx(2, 1)]));
  const T dev = N::fma(d0, d0, N::fma(d1, d1, N::fma(d2, d2, off + off)));   // ||A - mu I||_F^2
  return !(dev <= T(kCentredGate3) * (mu * mu)) || !(mu > T(0));
}
template <typename T> __device__ __forceinline__ void log_series3_centred(const T (&a)[6], T (&m0)[6], T pre = T(1)) {
  using N = Num<T>;
  using S = LogSeriesWide<T>;
  const T mu = (a[pidx(0, 0)] + a[pidx(1, 1)] + a[pidx(2, 2)]) * T(1.0 / 3.0);
  const T rmu = N::rcp(mu);
  const T e00 = N::fma(a[pidx(0, 0)], rmu, T(-1)), e11 = N::fma(a[pidx(1, 1)], rmu, T(-1)), e22 = N::fma(a[pidx(2, 2)], rmu, T(-1));
  const T e10 = a[pidx(1, 0)] * rmu, e20 = a[pidx(2, 0)] * rmu, e21 = a[pidx(2, 1)] * rmu;
  const T f00 = N::fma(e00, e00, N::fma(e10, e10, e20 * e20));
  const T f11 = N::fma(e10, e10, N::fma(e11, e11, e21 * e21));
  const T f22 = N::fma(e20, e20, N::fma(e21, e21, e22 * e22));
  const T f10 = N::fma(e10, e00, N::fma(e11, e10, e21 * e20));
  const T f20 = N::fma(e20, e00, N::fma(e21, e10, e22 * e20));
  const T f21 = N::fma(e20, e10, N::fma(e21, e11, e22 * e21));
  const T s2 = T(-0.5) * (f00 + f11 + f22);          // s1 = 0: s2 = -tr(E'^2) / 2
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  T a0, a1, a2;
  ring_horner3<T, S, true>(T(0), s2, s3, a0, a1, a2);
  const T logmu = N::log(mu);
  const T b0 = N::fma(a2, s3, logmu) * pre, b1 = N::fma(-a2, s2, a0) * pre, b2 = a1 * pre;
  m0[pidx(0, 0)] = N::fma(b2, f00, N::fma(b1, e00, b0));
  m0[pidx(1, 1)] = N::fma(b2, f11, N::fma(b1, e11, b0));
  m0[pidx(2, 2)] = N::fma(b2, f22, N::fma(b1, e22, b0));
  m0[pidx(1, 0)] = N::fma(b2, f10, b1 * e10);
  m0[pidx(2, 0)] = N::fma(b2, f20, b1 * e20);
  m0[pidx(2, 1)] = N::fma(b2, f21, b1 * e21);
}

// d^2 = ||log A||_F^2 of a pair at moderate distance, straight from invariants (the forward's counterpart of
// log_series3_centred): with L = log mu, A' = A / mu = I + E' (traceless E'),
//   ||log A||_F^2 = 3 L^2 + 2 L log det A' + tr(E'^2 q(E')),   log^2(1+x) = x^2 q(x),  det A' = 1 + s1 + s2 + s3,
// q of degree 15 on |x| <= 0.66 (3e-7 x^2, tools/design/series_fit_wide.py), Horner in R[E']/(chi_E') with s1 = 0 (two
// multiply-adds per coefficient), tr E'^3 = 3 s3 and tr E'^4 = (tr E'^2)^2 / 2 for a traceless 3 x 3 matrix: no matrix
// product at all, two logarithms and one reciprocal.  ~75 instructions against ~95 for the closed-form eigenvalues.
template <typename T> struct LogSqSeriesWide;
template <> struct LogSqSeriesWide<float> {
  static constexpr int kTerms = 16;
  static constexpr float kQ[kTerms] = {9.999998930e-01f, -9.999998963e-01f, 9.166980138e-01f, -8.333637039e-01f,
                                       7.596244972e-01f, -6.985597554e-01f, 6.746606801e-01f, -6.295878238e-01f,
                                       3.395295792e-01f, -3.133767555e-01f, 1.520582314e+00f, -1.462563750e+00f,
                                       -1.927056128e+00f, 1.871837175e+00f, 2.893810986e+00f, -2.795949233e+00f};
  static constexpr float at(int i) { return kQ[i]; }
};
template <typename T> __device__ __forceinline__ T logsq_series3_centred(const T (&a)[6]) {
  using N = Num<T>;
  using S = LogSqSeriesWide<T>;
  const T mu = (a[pidx(0, 0)] + a[pidx(1, 1)] + a[pidx(2, 2)]) * T(1.0 / 3.0);
  const T rmu = N::rcp(mu);
  const T e00 = N::fma(a[pidx(0, 0)], rmu, T(-1)), e11 = N::fma(a[pidx(1, 1)], rmu, T(-1)), e22 = N::fma(a[pidx(2, 2)], rmu, T(-1));
  const T e10 = a[pidx(1, 0)] * rmu, e20 = a[pidx(2, 0)] * rmu, e21 = a[pidx(2, 1)] * rmu;
  T t2 = N::fma(e00, e00, N::fma(e11, e11, e22 * e22));
  t2 = N::fma(T(2), N::fma(e10, e10, N::fma(e20, e20, e21 * e21)), t2);
  const T s1 = e00 + e11 + e22;                       // rounding only (mu rmu - 1): kept in det A', dropped in the recurrence
  const T s2 = T(-0.5) * t2;
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  T h0, h1, h2;
  ring_horner3<T, S, true>(T(0), s2, s3, h0, h1, h2);
  const T core = N::fma(h2, T(0.5) * t2 * t2, N::fma(h1, T(3) * s3, h0 * t2));
  const T lmu = N::log(mu), ldet = N::log((T(1) + s1) + (s2 + s3));
  return N::fma(lmu, N::fma(T(3), lmu, ldet + ldet), core);
}

// ---- Cayley-transform logarithm: the general eigen-free path (3x3) ------------------------------
// log A = log(mu) I + 2 atanh(Z),  Z = (A - mu I)(A + mu I)^-1  (spectrum z = (l - mu)/(l + mu), |z| < 1
// for every SPD A), atanh(Z) = Z P(Z^2) with P a near-minimax polynomial of atanh(sqrt w)/sqrt w on
// w in [0, 0.36] (degree 6 in fp32: 3.4e-8, degree 13 in fp64: 4e-15; tools/design/cayley_fit.py).
// P(W) is evaluated through the Cayley-Hamilton reduction of W = Z^2 (scalar three-term recurrences),
// the inverse is the adjugate of a 3x3, and mu = 2^k (the power of two next to tr A / 3) so that
// log(mu) = k ln 2 carries no transcendental error and A ~ I gives log(mu) = 0 exactly.  Z is formed
// as (A - mu I) adj(A + mu I) / det — a product of commuting symmetric matrices — which keeps its
// RELATIVE accuracy when A is close to mu I (measured max error 7e-7 of max|log A| in fp32, 4e-13 in
// fp64, spectra of any spread inside the gate).  ~170 ops, one division, no eigensolve, no iteration:
// valid whenever tr(Z^2) <= 0.36, i.e. eigenvalue ratios up to ~16 — every pair of an embedding whose
// distances are O(1).  Returns tr(Z^2) for the caller's (wave-uniform) gate.
constexpr double kCayleyGate = 0.36;
template <typename T> __device__ __forceinline__ T frexp_t(T x, int* k);
template <> __device__ __forceinline__ float frexp_t<float>(float x, int* k) { return ::frexpf(x, k); }
template <> __device__ __forceinline__ double frexp_t<double>(double x, int* k) { return ::frexp(x, k); }
template <typename T> __device__ __forceinline__ T ldexp_t(T x, int k);
template <> __device__ __forceinline__ float ldexp_t<float>(float x, int k) { return ::ldexpf(x, k); }
template <> __device__ __forceinline__ double ldexp_t<double>(double x, int k) { return ::ldexp(x, k); }

template <typename T> __device__ __forceinline__ void sym3_mul(const T (&x)[6], const T (&y)[6], T (&o)[6]) {
  using N = Num<T>;  // product of two COMMUTING symmetric 3x3 (packed 00,10,11,20,21,22)
  o[0] = N::fma(x[0], y[0], N::fma(x[1], y[1], x[3] * y[3]));
  o[1] = N::fma(x[1], y[0], N::fma(x[2], y[1], x[4] * y[3]));
  o[2] = N::fma(x[1], y[1], N::fma(x[2], y[2], x[4] * y[4]));
  o[3] = N::fma(x[3], y[0], N::fma(x[4], y[1], x[5] * y[3]));
  o[4] = N::fma(x[3], y[1], N::fma(x[4], y[2], x[5] * y[4]));
  o[5] = N::fma(x[3], y[3], N::fma(x[4], y[4], x[5] * y[5]));
}

// ---- the same logarithm with the matrix algebra moved into quotient rings (round 4) ------------------------------
// Everything between E^2 and the last combination is scalar arithmetic (tools/design/cayley_ring.py):
//   mu = 2^k,  E = A / mu - I (exact scaling),  s1, s2, s3 = elementary symmetric functions of E's spectrum,
//   Z = E (E + 2I)^-1 = (s3 I + 2 (s1 + 2) E - 2 E^2) / D,  D = det(E + 2I) = 8 + 4 s1 + 2 s2 + s3
//       (adj(B) = B^2 - tr(B) B + e2(B) I for a 3x3, then Cayley-Hamilton for E^3),
//   e_i(Z) = {4 (s1 + s2) + 3 s3,  2 s2 + 3 s3,  s3} / D  (the cubic of z = eps / (2 + eps)),  e_i(W = Z^2) from those,
//   atanh(Z) = Z P(W):  P(W) = c0 + c1 W + c2 W^2 by Horner in R[W]/(chi_W)  (as before: three FMAs per coefficient),
//            = b0 + b1 Z + b2 Z^2      (c0 Z + c1 Z^3 + c2 Z^5 reduced in R[Z]/(chi_Z): three steps of three FMAs),
//            = g0 + g1 E + g2 E^2      (Z, Z^2 written in the E basis by ring arithmetic on their coefficients),
//   log A = log(mu) I + 2 (g0 I + g1 E + g2 E^2).
// Matrix work: E^2 (18 FMAs) and the last line (12) — against the adjugate, three commuting 3x3 products and the assembly
// of P(W) (~105) in log_cayley3_matrix; ~146 operations in all against ~180, six matrix temporaries fewer alive, and the
// gate tr(Z^2) is known before any matrix is formed.  Accuracy (emulated, against a 40-digit eigendecomposition): 5e-7
// of max|log A| in fp32, 7e-15 in fp64, spectra of any spread inside the gate and mu up to 2^+-11.
// `pre` multiplies log A (the caller's 2g when it is known before the logarithm).
template <typename T> struct CayleyP;   // atanh(sqrt w)/sqrt w on [0, 0.36] (tools/design/cayley_fit.py)
template <> struct CayleyP<float> {
  static constexpr int K = 6;
  static constexpr float c[K + 1] = {1.00000002318570891e+00f, 3.33327042495924375e-01f, 2.00274867564608688e-01f,
                                     1.38428695737667723e-01f, 1.44240977093542333e-01f, -3.05379184438951401e-02f,
                                     2.73482843603638170e-01f};
  static constexpr int kTerms = K + 1;
  static constexpr float at(int i) { return c[i]; }
};
template <> struct CayleyP<double> {
  static constexpr int K = 13;
  static constexpr double c[K + 1] = {9.99999999999997002e-01, 3.33333333336093829e-01, 1.99999999512113669e-01,
                                      1.42857176992888746e-01, 1.11109865194520263e-01, 9.09362522505209464e-02,
                                      7.65413194323763535e-02, 7.02835946952955759e-02, 3.51739351378960174e-02,
                                      1.60038305495638411e-01, -2.87332526616183470e-01, 7.35063731671786291e-01,
                                      -8.29196169410508666e-01, 5.70221868872885063e-01};
  static constexpr int kTerms = K + 1;
  static constexpr double at(int i) { return c[i]; }
};
// fp64 only — the narrow tier: the same two functions on [0, 0.16] (tools/design/cayley_sq_fit.py, relative error 2.4e-15 /
// 3.9e-16), four and five coefficients fewer.  tr(Z^2) is known before the polynomial starts, so a wavefront whose pairs
// all have tr(Z^2) <= 0.16 takes the short tables: every pair of the mid-training spread does (max 0.136 at
// ||log X|| = 0.35; mu = 2^k alone contributes up to 0.087), 12 / 15 multiply-adds and twice as many scalar moves fewer.
constexpr double kCayleyNarrow = 0.16;
struct CayleyPn64 {
  static constexpr double c[10] = {0.999999999999997828, 0.333333333336045998, 0.199999999442592788, 0.142857187137109386,
                                   0.111109332105236778, 0.0909499162581166662, 0.0763574513803657068, 0.0714494792571264998,
                                   0.0350458696353453726, 0.113663386241058913};
  static constexpr int kTerms = 10;
  static constexpr double at(int i) { return c[i]; }
};
struct CayleyQn64 {
  static constexpr double c[11] = {1.00000000000000037, 0.666666666666100058, 0.511111111252317187, 0.419047605361109196,
                                   0.357460995535957299, 0.313015598858362064, 0.279652848994267891, 0.248774877825649445,
                                   0.25954346018603201, 0.0906375847139156082, 0.475855998018695552};
  static constexpr int kTerms = 11;
  static constexpr double at(int i) { return c[i]; }
};
// mu = 2^k next to the mean eigenvalue: returns 1 / mu, *logmu = k ln 2 (no transcendental error; A ~ I gives exactly 0)
template <typename T> __device__ __forceinline__ T cayley_scale(T mean, T* logmu) {
  int k;
  const T mant = frexp_t<T>(mean, &k);
  if (mant < T(0.70710678118654752)) k -= 1;
  *logmu = T(k) * T(0.69314718055994531);
  return ldexp_t<T>(T(1), -k);
}
// e_i(Z) (z1, z2, z3), e_i(Z^2) (t1, t2, t3) and 1 / D from the invariants of E
template <typename T>
__device__ __forceinline__ void cayley3_spectrum(T s1, T s2, T s3, T& rD, T& z1, T& z2, T& z3, T& t1, T& t2, T& t3) {
  using N = Num<T>;
  const T s33 = T(3) * s3;
  rD = N::rcp_fast(N::fma(T(4), s1, T(8)) + N::fma(T(2), s2, s3));   // D = prod (1 + lambda_k / mu) > 1
  z1 = N::fma(T(4), s1 + s2, s33) * rD;
  z2 = N::fma(T(2), s2, s33) * rD;
  z3 = s3 * rD;
  t1 = N::fma(z1, z1, T(-2) * z2);
  t2 = N::fma(z2, z2, T(-2) * (z1 * z3));
  t3 = z3 * z3;
}
template <typename T> __device__ __forceinline__ T log_cayley3(const T (&a)[6], T (&m0)[6], T pre = T(1)) {
  using N = Num<T>;
  using P = CayleyP<T>;
  T logmu;
  const T r = cayley_scale<T>((a[0] + a[2] + a[5]) * T(1.0 / 3.0), &logmu);
  const T e00 = N::fma(a[0], r, T(-1)), e11 = N::fma(a[2], r, T(-1)), e22 = N::fma(a[5], r, T(-1));
  const T e10 = a[1] * r, e20 = a[3] * r, e21 = a[4] * r;
  const T f00 = N::fma(e00, e00, N::fma(e10, e10, e20 * e20));
  const T f11 = N::fma(e10, e10, N::fma(e11, e11, e21 * e21));
  const T f22 = N::fma(e20, e20, N::fma(e21, e21, e22 * e22));
  const T f10 = N::fma(e10, e00, N::fma(e11, e10, e21 * e20));
  const T f20 = N::fma(e20, e00, N::fma(e21, e10, e22 * e20));
  const T f21 = N::fma(e20, e10, N::fma(e21, e11, e22 * e21));
  const T s1 = e00 + e11 + e22;
  const T s2 = T(0.5) * N::fma(s1, s1, -(f00 + f11 + f22));
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  T rD, z1, z2, z3, t1, t2, t3;
  cayley3_spectrum<T>(s1, s2, s3, rD, z1, z2, z3, t1, t2, t3);
  // P(W) = c0 + c1 W + c2 W^2 by Horner in R[W]/(chi_W), W^3 = t1 W^2 - t2 W + t3 I
  T c0, c1, c2;
#ifndef MM_CAYLEY_ONE_TIER   // (A/B builds)
  if constexpr (std::is_same<T, double>::value) {
    if (!__any(!(t1 <= T(kCayleyNarrow)))) ring_horner3<T, CayleyPn64>(t1, t2, t3, c0, c1, c2);
    else ring_horner3<T, P>(t1, t2, t3, c0, c1, c2);
  } else
#endif
    ring_horner3<T, P>(t1, t2, t3, c0, c1, c2);
  // Z (c2 Z^4 + c1 Z^2 + c0) in R[Z]/(chi_Z), Z^3 = z1 Z^2 - z2 Z + z3 I: start from c2 Z^2 + c1, then . Z, . Z + c0, . Z
  T b0 = c2 * z3, b1 = N::fma(-c2, z2, c1), b2 = c2 * z1;
  {
    const T n0 = N::fma(b2, z3, c0), n1 = N::fma(-b2, z2, b0), n2 = N::fma(b2, z1, b1);
    b0 = n0; b1 = n1; b2 = n2;
  }
  {
    const T n0 = b2 * z3, n1 = N::fma(-b2, z2, b0), n2 = N::fma(b2, z1, b1);
    b0 = n0; b1 = n1; b2 = n2;
  }
  // Z = y0 + y1 E + y2 E^2 and Z^2 = w0 + w1 E + w2 E^2 (E^3 = s1 E^2 - s2 E + s3, E^4 = h2 E^2 + h1 E + h0)
  const T y0 = z3, y2 = T(-2) * rD, y1 = -(s1 + T(2)) * y2;
  const T q0 = y0 * y0, q1 = T(2) * (y0 * y1), q2 = N::fma(T(2) * y0, y2, y1 * y1), q3 = T(2) * (y1 * y2), q4 = y2 * y2;
  const T h2 = N::fma(s1, s1, -s2), h1 = N::fma(-s1, s2, s3), h0 = s1 * s3;
  const T w0 = N::fma(q4, h0, N::fma(q3, s3, q0));
  const T w1 = N::fma(q4, h1, N::fma(-q3, s2, q1));
  const T w2 = N::fma(q4, h2, N::fma(q3, s1, q2));
  const T p2 = pre + pre;
  const T g0 = N::fma(N::fma(b2, w0, N::fma(b1, y0, b0)), p2, logmu * pre);
  const T g1 = N::fma(b2, w1, b1 * y1) * p2;
  const T g2 = N::fma(b2, w2, b1 * y2) * p2;
  m0[0] = N::fma(g2, f00, N::fma(g1, e00, g0));
  m0[2] = N::fma(g2, f11, N::fma(g1, e11, g0));
  m0[5] = N::fma(g2, f22, N::fma(g1, e22, g0));
  m0[1] = N::fma(g2, f10, g1 * e10);
  m0[3] = N::fma(g2, f20, g1 * e20);
  m0[4] = N::fma(g2, f21, g1 * e21);
  return t1;
}

// d^2 = ||log A||_F^2 of a pair outside the close-pair gate from INVARIANTS ONLY (the forward's counterpart of the ring
// form above; tools/design/cayley_sq_fit.py):
//   sum_k log^2 lambda_k = 2 log(mu) log det A - 3 log^2(mu) + 4 sum_k atanh^2(z_k),   z_k = (lambda_k - mu) / (lambda_k + mu),
// atanh^2(sqrt w) = w q(w), sum_k w_k q(w_k) = tr(W q(W)) = c0 p1 + c1 p2 + c2 p3 with q(W) = c0 + c1 W + c2 W^2 by the same
// Horner recurrence and p_k the power sums of the w_k (Newton's identities on t1, t2, t3).  log det A is NOT computed:
// it is log det X_j - log det X_i from the per-node table (spd_ws.hpp, nodeLd).  No matrix product, no inverse, no
// transcendental: ~100 operations in fp64 against ~190 for the matrix logarithm followed by its Frobenius norm.
// q: 16 coefficients in fp64 (relative error 1.5e-16 on [0, 0.36]), 8 in fp32 (1.1e-8).  *gate receives tr(Z^2).
template <typename T> struct CayleyQ;
template <> struct CayleyQ<float> {
  static constexpr int K = 7;
  static constexpr float c[K + 1] = {0.999999990659938783f, 0.666669978862739941f, 0.510920481665167664f, 0.423174655546735851f,
                                     0.314332175292986808f, 0.551201736862719326f, -0.412567514510121138f, 1.17039095371423926f};
  static constexpr int kTerms = K + 1;
  static constexpr float at(int i) { return c[i]; }
};
template <> struct CayleyQ<double> {
  static constexpr int K = 15;
  static constexpr double c[K + 1] = {0.99999999999999987, 0.666666666666851142, 0.511111111067702619, 0.419047623073864087,
                                      0.35746012197040091, 0.313040831979712641, 0.279195466546191997, 0.254162482860379539,
                                      0.217779908930427374, 0.303497422045376885, -0.242356046115491071, 1.74002823450430709,
                                      -3.70667784316895825, 6.72451540777742607, -6.69607770346690423, 3.64458881905822882};
  static constexpr int kTerms = K + 1;
  static constexpr double at(int i) { return c[i]; }
};
template <typename T> __device__ __forceinline__ T logsq_cayley3(const T (&a)[6], T logdet_a, T* gate) {
  using N = Num<T>;
  using Q = CayleyQ<T>;
  T logmu;
  const T r = cayley_scale<T>((a[0] + a[2] + a[5]) * T(1.0 / 3.0), &logmu);
  const T e00 = N::fma(a[0], r, T(-1)), e11 = N::fma(a[2], r, T(-1)), e22 = N::fma(a[5], r, T(-1));
  const T e10 = a[1] * r, e20 = a[3] * r, e21 = a[4] * r;
  const T s1 = e00 + e11 + e22;
  T tr2 = N::fma(e00, e00, N::fma(e11, e11, e22 * e22));
  tr2 = N::fma(T(2), N::fma(e10, e10, N::fma(e20, e20, e21 * e21)), tr2);
  const T s2 = T(0.5) * N::fma(s1, s1, -tr2);
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  T rD, z1, z2, z3, t1, t2, t3;
  cayley3_spectrum<T>(s1, s2, s3, rD, z1, z2, z3, t1, t2, t3);
  *gate = t1;
  const T p1 = t1, p2 = N::fma(t1, p1, T(-2) * t2), p3 = N::fma(t1, p2, N::fma(-t2, p1, T(3) * t3));
  T c0, c1, c2;
#ifndef MM_CAYLEY_ONE_TIER
  if constexpr (std::is_same<T, double>::value) {
    if (!__any(!(t1 <= T(kCayleyNarrow)))) ring_horner3<T, CayleyQn64>(t1, t2, t3, c0, c1, c2);
    else ring_horner3<T, Q>(t1, t2, t3, c0, c1, c2);
  } else
#endif
    ring_horner3<T, Q>(t1, t2, t3, c0, c1, c2);
  const T s = N::fma(c2, p3, N::fma(c1, p2, c0 * p1));   // sum atanh^2 z_k
  return N::fma(logmu, N::fma(T(-3), logmu, logdet_a + logdet_a), T(4) * s);
}

// (the round-1..3 form, kept for A/B builds: -DMM_CAYLEY_MATRIX)
template <typename T> __device__ __forceinline__ T log_cayley3_matrix(const T (&a)[6], T (&m0)[6]) {
  using N = Num<T>;
  constexpr bool kF32 = std::is_same<T, float>::value;
  constexpr int K = kF32 ? 6 : 13;
  constexpr double kC32[7] = {1.00000002318570891e+00, 3.33327042495924375e-01, 2.00274867564608688e-01,
                              1.38428695737667723e-01, 1.44240977093542333e-01, -3.05379184438951401e-02,
                              2.73482843603638170e-01};
  constexpr double kC64[14] = {9.99999999999997002e-01, 3.33333333336093829e-01, 1.99999999512113669e-01,
                               1.42857176992888746e-01, 1.11109865194520263e-01, 9.09362522505209464e-02,
                               7.65413194323763535e-02, 7.02835946952955759e-02, 3.51739351378960174e-02,
                               1.60038305495638411e-01, -2.87332526616183470e-01, 7.35063731671786291e-01,
                               -8.29196169410508666e-01, 5.70221868872885063e-01};
  int k;
  const T mant = frexp_t<T>((a[0] + a[2] + a[5]) * T(1.0 / 3.0), &k);  // mean eigenvalue = mant 2^k
  if (mant < T(0.70710678118654752)) k -= 1;
  const T mu = ldexp_t<T>(T(1), k), logmu = T(k) * T(0.69314718055994531);
  // adj(B), B = A + mu I
  const T b00 = a[0] + mu, b11 = a[2] + mu, b22 = a[5] + mu, b10 = a[1], b20 = a[3], b21 = a[4];
  T adj[6], e[6], z[6], w[6], w2[6];
  adj[0] = N::fma(b11, b22, -b21 * b21);
  adj[1] = N::fma(b21, b20, -b10 * b22);
  adj[2] = N::fma(b00, b22, -b20 * b20);
  adj[3] = N::fma(b10, b21, -b11 * b20);
  adj[4] = N::fma(b10, b20, -b00 * b21);
  adj[5] = N::fma(b00, b11, -b10 * b10);
  const T rdet = N::rcp(N::fma(b00, adj[0], N::fma(b10, adj[1], b20 * adj[3])));
  e[0] = a[0] - mu; e[1] = a[1]; e[2] = a[2] - mu; e[3] = a[3]; e[4] = a[4]; e[5] = a[5] - mu;
  sym3_mul<T>(e, adj, z);
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] *= rdet;
  sym3_mul<T>(z, z, w);
  sym3_mul<T>(w, w, w2);
  const T t1 = w[0] + w[2] + w[5];
  const T t2 = T(0.5) * N::fma(t1, t1, -(w2[0] + w2[2] + w2[5]));
  const T t3 = w[0] * N::fma(w[2], w[5], -w[4] * w[4]) - w[1] * N::fma(w[1], w[5], -w[4] * w[3]) +
               w[3] * N::fma(w[1], w[4], -w[2] * w[3]);
  auto cf = [&](int i) -> T { return kF32 ? T(kC32[i < 7 ? i : 0]) : T(kC64[i]); };
  // P(W) = sum coef_i W^i by Horner's rule in R[W]/(chi_W) (see log_series3): three FMAs per coefficient
  T c0 = cf(K - 2), c1 = cf(K - 1), c2 = cf(K);
#pragma unroll
  for (int i = K - 3; i >= 0; --i) {
    const T n0 = N::fma(c2, t3, cf(i)), n1 = N::fma(-c2, t2, c0), n2 = N::fma(c2, t1, c1);
    c0 = n0; c1 = n1; c2 = n2;
  }
  T pw[6];
  pw[0] = N::fma(c2, w2[0], N::fma(c1, w[0], c0));
  pw[1] = N::fma(c2, w2[1], c1 * w[1]);
  pw[2] = N::fma(c2, w2[2], N::fma(c1, w[2], c0));
  pw[3] = N::fma(c2, w2[3], c1 * w[3]);
  pw[4] = N::fma(c2, w2[4], c1 * w[4]);
  pw[5] = N::fma(c2, w2[5], N::fma(c1, w[5], c0));
  sym3_mul<T>(z, pw, m0);
  m0[0] = N::fma(T(2), m0[0], logmu);
  m0[1] += m0[1];
  m0[2] = N::fma(T(2), m0[2], logmu);
  m0[3] += m0[3];
  m0[4] += m0[4];
  m0[5] = N::fma(T(2), m0[5], logmu);
  return t1;
}

// product of two COMMUTING symmetric DxD matrices (packed lower), D x (D+1)/2 x D FMAs
template <typename T, int D>
__device__ __forceinline__ void sym_mul(const T (&x)[Packed<D>::NP], const T (&y)[Packed<D>::NP], T (&o)[Packed<D>::NP]) {
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = x[pidx(r, 0)] * y[pidx(0, c)];
#pragma unroll
      for (int k = 1; k < D; ++k) acc = Num<T>::fma(x[pidx(r, k)], y[pidx(k, c)], acc);
      o[pidx(r, c)] = acc;
    }
}

// ---- log(A) without an eigensolve, any D (round 5: SPD(5 .. 9) ran a Jacobi eigensolve with eigenvectors per pair) ----
// log(I + E) = E p(E) with p evaluated in MATRICES by Paterson-Stockmeyer groups of G coefficients:
//   p(E) = sum_j (c_{Gj} I + c_{Gj+1} E + .. + c_{Gj+G-1} E^{G-1}) (E^G)^j,   Horner's rule in E^G over the groups.
// Every operand is a polynomial in E, so the products commute, are symmetric, and sym_mul's packed half is all of them:
// (G - 1) + (K / G - 1) + 1 products of D^2 (D + 1) / 2 multiply-adds — K = 8 (fp32, close pairs): 5; K = 16 (fp32,
// recentred): 7 with G = 4; fp64: 8 for K = 20, 12 for K = 36 — against ~5000 instructions of the 6 x 6 eigensolve + V diag V^T.
// The group loop is a real loop of two steps (ping-pong between two matrices: no copies; coefficients read from the table)
// so that the code holds three products, not K / G.  G = 4 keeps E .. E^4 and two accumulators (6 matrices), G = 2 four.
// S = LogSeries<T> on a spectral radius of 0.3 (close pairs), LogSeriesWide<T> on 0.66 (recentred, below).
template <typename T, int D, typename S, int G>
__device__ __forceinline__ void log_series_mat(const T (&e)[Packed<D>::NP], T (&m0)[Packed<D>::NP]) {
  using N = Num<T>;
  constexpr int NP = Packed<D>::NP, K = S::kTerms, NG = K / G;
  static_assert((G == 2 || G == 4) && K % G == 0 && NG >= 2, "whole groups");
  T pw[NP], e2[G == 4 ? NP : 1], e3[G == 4 ? NP : 1];   // pw = E^G
  if constexpr (G == 4) {
    sym_mul<T, D>(e, e, e2);
    sym_mul<T, D>(e, e2, e3);
    sym_mul<T, D>(e2, e2, pw);
  } else {
    sym_mul<T, D>(e, e, pw);
  }
  // dst += group j (dst holds acc . E^G, or nothing for the first group)
  auto add_group = [&](T (&dst)[NP], int j, bool first) __attribute__((always_inline)) {
    const T c0 = S::kA[G * j], c1 = S::kA[G * j + 1];
    T c2 = T(0), c3 = T(0);
    if constexpr (G == 4) { c2 = S::kA[G * j + 2]; c3 = S::kA[G * j + 3]; }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      T v = first ? c1 * e[k] : N::fma(c1, e[k], dst[k]);
      if constexpr (G == 4) v = N::fma(c3, e3[k], N::fma(c2, e2[k], v));
      dst[k] = v;
    }
#pragma unroll
    for (int r = 0; r < D; ++r) dst[pidx(r, r)] += c0;
  };
  T acc[NP], t[NP];
  add_group(acc, NG - 1, true);
  constexpr int kSteps = NG - 1;
#pragma unroll 1
  for (int j = NG - 2; j >= (kSteps & 1); j -= 2) {
    sym_mul<T, D>(acc, pw, t);
    add_group(t, j, false);
    sym_mul<T, D>(t, pw, acc);
    add_group(acc, j - 1, false);
  }
  if constexpr (kSteps & 1) {
    sym_mul<T, D>(acc, pw, t);
    add_group(t, 0, false);
    sym_mul<T, D>(e, t, m0);
  } else {
    sym_mul<T, D>(e, acc, m0);
  }
}
// groups of four unless they save nothing (K = 8: five products either way) or the six matrices do not fit beside the kernel's
// own state and spill inside the group loop (measured per size and precision: profiles/r05_experiments.md §18)
template <typename T, int D, typename S> constexpr int series_group() {
#ifdef MM_SERIES_GROUP   // (A/B builds)
  return MM_SERIES_GROUP;
#else
  if (S::kTerms <= 8) return 2;
  if (sizeof(T) == 4) return D <= 7 ? 4 : 2;
  return (D == 5 || (D == 6 && S::kTerms <= 20)) ? 4 : 2;
#endif
}
// close pairs (||A - I||_F <= 0.3, the caller's gate)
template <typename T, int D> __device__ __forceinline__ void log_close_mat(const T (&a)[Packed<D>::NP], T (&m0)[Packed<D>::NP]) {
  T e[Packed<D>::NP];
#pragma unroll
  for (int k = 0; k < Packed<D>::NP; ++k) e[k] = a[k];
#pragma unroll
  for (int r = 0; r < D; ++r) e[pidx(r, r)] -= T(1);
  log_series_mat<T, D, LogSeries<T>, series_group<T, D, LogSeries<T>>()>(e, m0);
}
// pairs at moderate distance: log A = log(mu) I + log(I + E'), E' = A / mu - I traceless (mu = tr A / D), whose
// spectral radius is at most sqrt((D - 1) / D) ||E'||_F — the gate bounds it by 0.66 (log_series3_centred's construction)
template <typename T, int D> __device__ __forceinline__ bool centred_far_mat(const T (&a)[Packed<D>::NP]) {
  using N = Num<T>;
  T tr = a[pidx(0, 0)];
#pragma unroll
  for (int r = 1; r < D; ++r) tr += a[pidx(r, r)];
  const T mu = tr * T(1.0 / D);
  T dg = T(0), off = T(0);
#pragma unroll
  for (int r = 0; r < D; ++r) {
    const T d = a[pidx(r, r)] - mu;
    dg = N::fma(d, d, dg);
#pragma unroll
    for (int c = 0; c < r; ++c) off = N::fma(a[pidx(r, c)], a[pidx(r, c)], off);
  }
  const T dev = N::fma(T(2), off, dg);   // ||A - mu I||_F^2
  return !(dev <= T(0.66 * 0.66 * D / (D - 1.0)) * (mu * mu)) || !(mu > T(0));
}
template <typename T, int D> __device__ __forceinline__ void log_centred_mat(const T (&a)[Packed<D>::NP], T (&m0)[Packed<D>::NP]) {
  using N = Num<T>;
  T tr = a[pidx(0, 0)];
#pragma unroll
  for (int r = 1; r < D; ++r) tr += a[pidx(r, r)];
  const T mu = tr * T(1.0 / D), rmu = N::rcp(mu);
  T e[Packed<D>::NP];
#pragma unroll
  for (int k = 0; k < Packed<D>::NP; ++k) e[k] = a[k] * rmu;
#pragma unroll
  for (int r = 0; r < D; ++r) e[pidx(r, r)] -= T(1);
  log_series_mat<T, D, LogSeriesWide<T>, series_group<T, D, LogSeriesWide<T>>()>(e, m0);
  const T lm = N::log(mu);
#pragma unroll
  for (int r = 0; r < D; ++r) m0[pidx(r, r)] += lm;
}

// Cayley-transform logarithm, 4x4 (same construction as log_cayley3): adjugate from the 2x2 minors of
// the row pairs (0,1) and (2,3), W^k = p I + q W + r W^2 + t W^3 with the four invariants of W = Z^2 from
// Newton's identities on tr W^m.  ~400 ops against ~1400 for a 4x4 Jacobi with eigenvectors.
template <typename T> __device__ __forceinline__ T log_cayley4(const T (&a)[10], T (&m0)[10]) {
  using N = Num<T>;
  constexpr bool kF32 = std::is_same<T, float>::value;
  constexpr int K = kF32 ? 6 : 13;
  constexpr double kC32[7] = {1.00000002318570891e+00, 3.33327042495924375e-01, 2.00274867564608688e-01,
                              1.38428695737667723e-01, 1.44240977093542333e-01, -3.05379184438951401e-02,
                              2.73482843603638170e-01};
  constexpr double kC64[14] = {9.99999999999997002e-01, 3.33333333336093829e-01, 1.99999999512113669e-01,
                               1.42857176992888746e-01, 1.11109865194520263e-01, 9.09362522505209464e-02,
                               7.65413194323763535e-02, 7.02835946952955759e-02, 3.51739351378960174e-02,
                               1.60038305495638411e-01, -2.87332526616183470e-01, 7.35063731671786291e-01,
                               -8.29196169410508666e-01, 5.70221868872885063e-01};
  int k;
  const T mant = frexp_t<T>((a[pidx(0, 0)] + a[pidx(1, 1)] + a[pidx(2, 2)] + a[pidx(3, 3)]) * T(0.25), &k);
  if (mant < T(0.70710678118654752)) k -= 1;
  const T mu = ldexp_t<T>(T(1), k), logmu = T(k) * T(0.69314718055994531);
  const T b00 = a[pidx(0, 0)] + mu, b11 = a[pidx(1, 1)] + mu, b22 = a[pidx(2, 2)] + mu, b33 = a[pidx(3, 3)] + mu;
  const T b10 = a[pidx(1, 0)], b20 = a[pidx(2, 0)], b21 = a[pidx(2, 1)], b30 = a[pidx(3, 0)], b31 = a[pidx(3, 1)],
          b32 = a[pidx(3, 2)];
  // 2x2 minors of rows (0,1) and of rows (2,3) (symmetric B: b_rc = b_cr)
  const T s0 = N::fma(b00, b11, -b10 * b10), s1 = N::fma(b00, b21, -b10 * b20), s2 = N::fma(b00, b31, -b10 * b30);
  const T s3 = N::fma(b10, b21, -b11 * b20), s4 = N::fma(b10, b31, -b11 * b30), s5 = N::fma(b20, b31, -b21 * b30);
  const T c5 = N::fma(b22, b33, -b32 * b32), c4 = N::fma(b21, b33, -b31 * b32), c3 = N::fma(b21, b32, -b31 * b22);
  const T c2 = N::fma(b20, b33, -b30 * b32), c1 = N::fma(b20, b32, -b30 * b22), c0 = N::fma(b20, b31, -b30 * b21);
  const T det = N::fma(s0, c5, N::fma(-s1, c4, N::fma(s2, c3, N::fma(s3, c2, N::fma(-s4, c1, s5 * c0)))));
  T adj[10], e[10], z[10], w[10], w2[10], w3[10];
  adj[pidx(0, 0)] = N::fma(b11, c5, N::fma(-b21, c4, b31 * c3));
  adj[pidx(1, 0)] = N::fma(-b10, c5, N::fma(b21, c2, -b31 * c1));
  adj[pidx(1, 1)] = N::fma(b00, c5, N::fma(-b20, c2, b30 * c1));
  adj[pidx(2, 0)] = N::fma(b10, c4, N::fma(-b11, c2, b31 * c0));
  adj[pidx(2, 1)] = N::fma(-b00, c4, N::fma(b10, c2, -b30 * c0));
  adj[pidx(2, 2)] = N::fma(b30, s4, N::fma(-b31, s2, b33 * s0));
  adj[pidx(3, 0)] = N::fma(-b10, c3, N::fma(b11, c1, -b21 * c0));
  adj[pidx(3, 1)] = N::fma(b00, c3, N::fma(-b10, c1, b20 * c0));
  adj[pidx(3, 2)] = N::fma(-b30, s3, N::fma(b31, s1, -b32 * s0));
  adj[pidx(3, 3)] = N::fma(b20, s3, N::fma(-b21, s1, b22 * s0));
  const T rdet = N::rcp(det);
#pragma unroll
  for (int i = 0; i < 10; ++i) e[i] = a[i];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[pidx(r, r)] -= mu;
  sym_mul<T, 4>(e, adj, z);
#pragma unroll
  for (int i = 0; i < 10; ++i) z[i] *= rdet;
  sym_mul<T, 4>(z, z, w);
  sym_mul<T, 4>(w, w, w2);
  sym_mul<T, 4>(w2, w, w3);
  T p1 = T(0), p2 = T(0), p3 = T(0), p4 = T(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p1 += w[pidx(r, r)];
    p2 += w2[pidx(r, r)];
    p3 += w3[pidx(r, r)];
    p4 = N::fma(w2[pidx(r, r)], w2[pidx(r, r)], p4);
#pragma unroll
    for (int c = 0; c < r; ++c) p4 = N::fma(T(2) * w2[pidx(r, c)], w2[pidx(r, c)], p4);
  }
  const T e1 = p1;
  const T e2 = T(0.5) * N::fma(e1, p1, -p2);
  const T e3 = T(1.0 / 3.0) * (N::fma(e2, p1, -e1 * p2) + p3);
  const T e4 = T(0.25) * (N::fma(e3, p1, -e2 * p2) + N::fma(e1, p3, -p4));
  auto cf = [&](int i) -> T { return kF32 ? T(kC32[i < 7 ? i : 0]) : T(kC64[i]); };
  // Horner in R[W]/(chi_W), W^4 = e1 W^3 - e2 W^2 + e3 W - e4 I: four FMAs per coefficient
  T al0 = cf(K - 3), al1 = cf(K - 2), al2 = cf(K - 1), al3 = cf(K);
#pragma unroll
  for (int i = K - 4; i >= 0; --i) {
    const T n0 = N::fma(-al3, e4, cf(i)), n1 = N::fma(al3, e3, al0), n2 = N::fma(-al3, e2, al1),
            n3 = N::fma(al3, e1, al2);
    al0 = n0; al1 = n1; al2 = n2; al3 = n3;
  }
  T pw[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) pw[i] = N::fma(al3, w3[i], N::fma(al2, w2[i], al1 * w[i]));
#pragma unroll
  for (int r2 = 0; r2 < 4; ++r2) pw[pidx(r2, r2)] += al0;
  sym_mul<T, 4>(z, pw, m0);
#pragma unroll
  for (int i = 0; i < 10; ++i) m0[i] += m0[i];
#pragma unroll
  for (int r2 = 0; r2 < 4; ++r2) m0[pidx(r2, r2)] += logmu;
  return p1;
}

// d^2 = ||log A||_F^2 of a close pair (||A - I||_F <= 0.3), 3x3, straight from the invariants of
// E = A - I:  sum_k log^2(1 + e_k) = tr(E^2 q(E)) with log^2(1+x) = x^2 q(x), q an economised polynomial.
// No eigenvalues, no transcendental.  *e2 receives ||E||_F^2.
template <typename T> struct LogSqSeries;
template <> struct LogSqSeries<float> {   // log^2(1+x) = x^2 q(x), q of degree 8 on |x| <= 0.3: max error 6.9e-8 x^2
  static constexpr int kTerms = 9;
  static constexpr float kQ[kTerms] = {9.999999990e-01f, -9.999985182e-01f, 9.166654125e-01f, -8.335515341e-01f,
                                       7.613116690e-01f, -6.914147801e-01f, 6.401787542e-01f, -7.263483416e-01f,
                                       6.811261199e-01f};
  static constexpr float at(int i) { return kQ[i]; }
};
template <> struct LogSqSeries<double> {  // degree 19: max error 4.7e-17 x^2 (tools/design/series_fit64.py)
  static constexpr int kTerms = 20;
  static constexpr double kQ[kTerms] = {
    1.00000000000000000e+00, -9.99999999999999889e-01, 9.16666666666741459e-01, -8.33333333333411308e-01,
    7.61111111083523362e-01, -6.99999999974005638e-01, 6.48214289636532492e-01, -6.03968257620789650e-01,
    5.65793368886221559e-01, -5.32539418364275541e-01, 5.03324479906769540e-01, -4.77427984436160913e-01,
    4.54016324988081366e-01, -4.33266732950779887e-01, 4.19186214916998590e-01, -4.01958766939621959e-01,
    3.42135448468324777e-01, -3.29342696720597350e-01, 5.48091271953506265e-01, -5.29456865029041324e-01};
  static constexpr double at(int i) { return kQ[i]; }
};
// ||A - I||_F^2 with the arithmetic of logsq_series3 (the forward's close-pair gate: shared with the series)
template <typename T> __device__ __forceinline__ T close_dev3(const T (&a)[6]) {
  using N = Num<T>;
  const T e00 = a[pidx(0, 0)] - T(1), e11 = a[pidx(1, 1)] - T(1), e22 = a[pidx(2, 2)] - T(1);
  const T e10 = a[pidx(1, 0)], e20 = a[pidx(2, 0)], e21 = a[pidx(2, 1)];
  T t2 = N::fma(e00, e00, N::fma(e11, e11, e22 * e22));
  return N::fma(T(2), N::fma(e10, e10, N::fma(e20, e20, e21 * e21)), t2);
}
template <typename T> __device__ __forceinline__ T logsq_series3(const T (&a)[6], T* e2) {
  using N = Num<T>;
  using S = LogSqSeries<T>;
  const T e00 = a[pidx(0, 0)] - T(1), e11 = a[pidx(1, 1)] - T(1), e22 = a[pidx(2, 2)] - T(1);
  const T e10 = a[pidx(1, 0)], e20 = a[pidx(2, 0)], e21 = a[pidx(2, 1)];
  T t2 = N::fma(e00, e00, N::fma(e11, e11, e22 * e22));
  t2 = N::fma(T(2), N::fma(e10, e10, N::fma(e20, e20, e21 * e21)), t2);
  const T s1 = e00 + e11 + e22;
  const T s2 = T(0.5) * N::fma(s1, s1, -t2);
  const T s3 = e00 * N::fma(e11, e22, -e21 * e21) - e10 * N::fma(e10, e22, -e21 * e20) +
               e20 * N::fma(e10, e21, -e11 * e20);
  // q(E) = h0 I + h1 E + h2 E^2 by Horner in R[E]/(chi_E) (see log_series3), then
  // tr(E^2 q(E)) = h0 tr E^2 + h1 tr E^3 + h2 tr E^4 with the power sums from Newton's identities
  T h0, h1, h2;
  ring_horner3<T, S>(s1, s2, s3, h0, h1, h2);
  const T t3 = N::fma(T(3), s3, N::fma(s1, t2, -s2 * s1));
  const T t4 = N::fma(s1, t3, N::fma(-s2, t2, s3 * s1));
  *e2 = t2;
  return N::fma(h2, t4, N::fma(h1, t3, h0 * t2));
}

// Same for 4x4: intermediates h0 I + h1 E + h2 E^2 + h3 E^3, multiplication by E through
//   (h0,h1,h2,h3) . E + c I = (c - h3 s4, h0 + h3 s3, h1 - h3 s2, h2 + h3 s1),  E^4 = s1 E^3 - s2 E^2 + s3 E - s4 I,
// s1..s4 from the power sums tr E^m (Newton's identities; tr E^3 = <E^2,E>, tr E^4 = ||E^2||_F^2).
template <typename T, typename S = LogSeries<T>> __device__ __forceinline__ T log_series4(const T (&a)[10], T (&m0)[10], T pre = T(1)) {
  using N = Num<T>;
  T e[10], e2[10], e3[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) e[k] = a[k];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[pidx(r, r)] -= T(1);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = e[pidx(r, 0)] * e[pidx(0, c)];
#pragma unroll
      for (int k = 1; k < 4; ++k) acc = N::fma(e[pidx(r, k)], e[pidx(k, c)], acc);
      e2[pidx(r, c)] = acc;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = e2[pidx(r, 0)] * e[pidx(0, c)];
#pragma unroll
      for (int k = 1; k < 4; ++k) acc = N::fma(e2[pidx(r, k)], e[pidx(k, c)], acc);
      e3[pidx(r, c)] = acc;
    }
  T p1 = T(0), p2 = T(0), p3 = T(0), p4 = T(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p1 += e[pidx(r, r)];
    p2 += e2[pidx(r, r)];
    p3 += e3[pidx(r, r)];
    p4 = N::fma(e2[pidx(r, r)], e2[pidx(r, r)], p4);
#pragma unroll
    for (int c = 0; c < r; ++c) p4 = N::fma(T(2) * e2[pidx(r, c)], e2[pidx(r, c)], p4);
  }
  const T s1 = p1;
  const T s2 = T(0.5) * N::fma(s1, p1, -p2);
  const T s3 = T(1.0 / 3.0) * (N::fma(s2, p1, -s1 * p2) + p3);
  const T s4 = T(0.25) * (N::fma(s3, p1, -s2 * p2) + N::fma(s1, p3, -p4));
  // p(E) by Horner in R[E]/(chi_E), then one more multiplication by E (log(I + E) = E p(E))
  T h0, h1, h2, h3;
  ring_horner4<T, S>(s1, s2, s3, s4, h0, h1, h2, h3);
  const T al[4] = {(-h3 * s4) * pre, N::fma(h3, s3, h0) * pre, N::fma(-h3, s2, h1) * pre, N::fma(h3, s1, h2) * pre};
#pragma unroll
  for (int k = 0; k < 10; ++k) m0[k] = N::fma(al[3], e3[k], N::fma(al[2], e2[k], al[1] * e[k]));
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) m0[pidx(rr, rr)] += al[0];
  return p2;  // ||E||_F^2
}

// d^2 = ||log A||_F^2 of a close 4x4 pair from the invariants (the 4x4 counterpart of logsq_series3): tr(E^2 q(E)) with
// q(E) = h0 I + h1 E + h2 E^2 + h3 E^3 by Horner in R[E]/(chi_E) and the power sums p2..p5 (p5 from Newton's identity).
// Only E^2 is formed: 100 multiply-adds against 180 for log_series4 followed by the Frobenius norm.
template <typename T, typename S = LogSqSeries<T>> __device__ __forceinline__ T logsq_series4(const T (&a)[10], T* det = nullptr) {
  using N = Num<T>;
  T e[10], e2[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) e[k] = a[k];
#pragma unroll
  for (int r = 0; r < 4; ++r) e[pidx(r, r)] -= T(1);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = e[pidx(r, 0)] * e[pidx(0, c)];
#pragma unroll
      for (int k = 1; k < 4; ++k) acc = N::fma(e[pidx(r, k)], e[pidx(k, c)], acc);
      e2[pidx(r, c)] = acc;
    }
  T p1 = T(0), p2 = T(0), p3 = T(0), p4 = T(0), o3 = T(0), o4 = T(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p1 += e[pidx(r, r)];
    p2 += e2[pidx(r, r)];
    p3 = N::fma(e2[pidx(r, r)], e[pidx(r, r)], p3);
    p4 = N::fma(e2[pidx(r, r)], e2[pidx(r, r)], p4);
#pragma unroll
    for (int c = 0; c < r; ++c) {
      o3 = N::fma(e2[pidx(r, c)], e[pidx(r, c)], o3);
      o4 = N::fma(e2[pidx(r, c)], e2[pidx(r, c)], o4);
    }
  }
  p3 = N::fma(T(2), o3, p3);
  p4 = N::fma(T(2), o4, p4);
  const T s1 = p1;
  const T s2 = T(0.5) * N::fma(s1, p1, -p2);
  const T s3 = T(1.0 / 3.0) * (N::fma(s2, p1, -s1 * p2) + p3);
  const T s4 = T(0.25) * (N::fma(s3, p1, -s2 * p2) + N::fma(s1, p3, -p4));
  const T p5 = N::fma(s1, p4, -s2 * p3) + N::fma(s3, p2, -s4 * p1);
  if (det) *det = ((T(1) + s1) + s2) + (s3 + s4);   // det(I + E) (the recentred form needs log det)
  T h0, h1, h2, h3;
  ring_horner4<T, S>(s1, s2, s3, s4, h0, h1, h2, h3);
  return N::fma(h3, p5, N::fma(h2, p4, N::fma(h1, p3, h0 * p2)));
}

// ---- the recentred series for 4 x 4 (fp32): log A = log(mu) I + log(I + E'), E' = A / mu - I, mu = tr A / 4 -----------
// Same polynomials as the 3 x 3 forms (|x| <= 0.66: LogSeriesWide / LogSqSeriesWide); a traceless symmetric 4 x 4 matrix
// has spectral radius at most sqrt(3/4) ||E'||_F, hence the gate ||A - mu I||_F^2 <= kCentredGate4 mu^2.  The general
// 4 x 4 Horner recurrence of log_series4 / logsq_series4 is reused on A / mu (p1 = tr E' is rounding-sized, not assumed
// zero): ~230 instructions against ~400 for the Cayley-transform logarithm.
constexpr double kCentredGate4 = 0.5808;   // (3/4) * 0.5808 = 0.66^2
template <typename T> __device__ __forceinline__ bool centred_far4(const T (&a)[10]) {
  using N = Num<T>;
  const T mu = ((a[pidx(0, 0)] + a[pidx(1, 1)]) + (a[pidx(2, 2)] + a[pidx(3, 3)])) * T(0.25);
  T dg = T(0), off = T(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const T d = a[pidx(r, r)] - mu;
    dg = N::fma(d, d, dg);
#pragma unroll
    for (int c = 0; c < r; ++c) off = N::fma(a[pidx(r, c)], a[pidx(r, c)], off);
  }
  const T dev = N::fma(T(2), off, dg);
  return !(dev <= T(kCentredGate4) * (mu * mu)) || !(mu > T(0));
}
template <typename T> __device__ __forceinline__ void log_series4_centred(const T (&a)[10], T (&m0)[10], T pre = T(1)) {
  using N = Num<T>;
  const T mu = ((a[pidx(0, 0)] + a[pidx(1, 1)]) + (a[pidx(2, 2)] + a[pidx(3, 3)])) * T(0.25);
  const T rmu = N::rcp(mu);
  T as[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) as[k] = a[k] * rmu;
  log_series4<T, LogSeriesWide<T>>(as, m0, pre);
  const T lm = N::log(mu) * pre;
#pragma unroll
  for (int r = 0; r < 4; ++r) m0[pidx(r, r)] += lm;
}
template <typename T> __device__ __forceinline__ T logsq_series4_centred(const T (&a)[10]) {
  using N = Num<T>;
  const T mu = ((a[pidx(0, 0)] + a[pidx(1, 1)]) + (a[pidx(2, 2)] + a[pidx(3, 3)])) * T(0.25);
  const T rmu = N::rcp(mu);
  T as[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) as[k] = a[k] * rmu;
  T det;
  const T core = logsq_series4<T, LogSqSeriesWide<T>>(as, &det);
  const T lmu = N::log(mu), ldet = N::log(det);
  return N::fma(lmu, N::fma(T(4), lmu, ldet + ldet), core);   // sum (L + l_k)^2 = 4 L^2 + 2 L log det A' + sum l_k^2
}

// log of a 2x2 SPD matrix in closed form (round 4; the mixed-manifold pair kernels ran a Jacobi eigensolve with eigenvectors — a
// wave-uniform sweep loop of ~150 instructions — for the SPD(2) factor of BASELINE config 4):
//   lambda+- = mid +- r,  mid = tr A / 2,  r = sqrt(((a00 - a11)/2)^2 + a10^2),  A - mid I = r V diag(1, -1) V^T, hence
//   V diag(l+, l-) V^T = (l+ + l-)/2 I + (l+ - l-)/(2 r) (A - mid I),   l+- = log clamp(lambda+-)  (spd.py:163-169).
// Returns l+^2 + l-^2; mlog = the matrix above (packed 00, 10, 11).  r = 0 gives (l+ - l-) = 0 exactly: no 0/0.
template <typename T> __device__ __forceinline__ T log_spd2(const T (&a)[3], T wmin, T wmax, T (&mlog)[3]) {
  using N = Num<T>;
  const T mid = T(0.5) * (a[0] + a[2]), dh = T(0.5) * (a[0] - a[2]);
  const T r = N::sqrt(N::fma(dh, dh, a[1] * a[1]));
  const T lp = N::log(N::min(N::max(mid + r, wmin), wmax)), lm = N::log(N::min(N::max(mid - r, wmin), wmax));
  const T beta = T(0.5) * (lp - lm) * N::rcp(N::max(r, N::tiny())), alpha = T(0.5) * (lp + lm);
  mlog[0] = N::fma(beta, dh, alpha);
  mlog[1] = beta * a[1];
  mlog[2] = N::fma(-beta, dh, alpha);
  return N::fma(lp, lp, lm * lm);
}

// The same for a PAIR of the all-pairs kernels (round 5: SPD(2) ran the Jacobi eigensolve with eigenvectors in its own pair
// kernels — forward 19.8, backward 32.2 us at n = 5000 against 22.8 / 42.4 for SPD(3), whose matrices have twice the entries).
// A = B B^T with B = L_i^-1 chol(X_j) lower triangular (li, yj packed 00, 10, 11).  lambda+ = mid + r as above; the SMALL
// eigenvalue comes from det A = (b00 b11)^2 — a product, where mid - r cancels: lambda- = det A / lambda+ keeps its relative
// accuracy whatever the condition of the pair (the Jacobi route needed a second, one-sided solve for that in fp32).  For r -> 0
// the two logarithms differ by rounding only; beta multiplies a matrix of norm r, so the error it carries into mlog stays
// eps |log lambda| / 2 (and r = 0 gives a finite beta times exact zeros).  Returns l+^2 + l-^2 (eigenvalues value-clamped as in
// spd.py:163-169); mlog = log A.
template <typename T, typename TL> __device__ __forceinline__ T log_pair2_chol(const TL (&li)[3], const T (&yj)[3], T wmin, T wmax,
                                                                              T (&mlog)[3]) {
  using N = Num<T>;
  const T b00 = li[0] * yj[0], b10 = N::fma(li[2], yj[1], li[1] * yj[0]), b11 = li[2] * yj[2];
  const T a00 = b00 * b00, a10 = b00 * b10, a11 = N::fma(b10, b10, b11 * b11);
  const T mid = T(0.5) * (a00 + a11), dh = T(0.5) * (a00 - a11);
  const T r = N::sqrt(N::fma(dh, dh, a10 * a10));
  const T lam_p = mid + r, sdet = b00 * b11;
  const T lam_m = (sdet * sdet) * N::rcp(lam_p);
  const T cp = N::min(N::max(lam_p, wmin), wmax), cm = N::min(N::max(lam_m, wmin), wmax);
  T lp = N::log(cp), lm = N::log(cm);
  const T s = N::fma(lp, lp, lm * lm);
  // mlog feeds the GRADIENT: where a clamp binds, the reference's autograd leaves log(w_c) w / w_c on the X_i side and
  // log(w_c) / w_c on the X_j side (w.data.clamp_ rewrites what log's backward saved, spd.py:163-169; pair_core's rho in
  // spd_pair.hpp).  The kernels form the X_j side as A^-1 M with the true A, so M = V diag(log(w_c) w / w_c) V^T serves both.
  // A wavefront in which no clamp binds — every one under the default [1e-8, 1e8] — skips it.
  if (__builtin_expect(__any((lam_p != cp) | (lam_m != cm)), 0)) {
    lp *= lam_p * N::rcp(cp);
    lm *= lam_m * N::rcp(cm);
  }
  const T beta = T(0.5) * (lp - lm) * N::rcp(N::max(r, N::tiny())), alpha = T(0.5) * (lp + lm);
  mlog[0] = N::fma(beta, dh, alpha);
  mlog[1] = beta * a10;
  mlog[2] = N::fma(-beta, dh, alpha);
  return s;
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
// ---- transposing reduction, general form: N values per lane -> ONE wavefront total per lane ------------
// Level L = 0..5 works on lane bit 5-L.  A level with several live values pairs them up: of each pair (a, b)
// a lane keeps the value its bit selects and receives the same value from its partner, so the number of live
// values halves; an unpaired value (and the single value left at the end) is a plain butterfly.  The
// cross-lane move of each level is the cheapest the hardware has for that bit:
//   bit 5, 4: v_permlane32_swap / v_permlane16_swap exchange the upper half (odd rows) of a with the lower half
//             (even rows) of b IN ONE INSTRUCTION — swap + add is the whole level for a pair, no select;
//   bit 3   : DPP row_ror:8;   bit 2: DPP row_half_mirror (partner lane ^ 7: opposite bit 2, same bits 5..3 —
//             any perfect matching inside the group serves);   bit 1, 0: DPP quad_perm.  DPP levels cost two
//             selects and one add-with-DPP per pair.
// 6 values: 6 + 5 + 3 + 3 = 17 instructions (the round-1 form, selects at every level and a two-instruction
// xor-4, compiled to 35 with its hazard nops); 10 values: 26.  `reduce_slot<N>(lane, writer)` names the value
// a lane ends up with; lanes with writer == false hold a duplicate.  fp64: the same moves on the two halves.
template <int LEVEL> __device__ __forceinline__ float dpp_partner(float x) {
  static_assert(LEVEL >= 2 && LEVEL <= 5, "levels 0 and 1 are the swap forms");
  constexpr int ctrl = LEVEL == 2 ? 0x128 /* row_ror:8 */ : LEVEL == 3 ? 0x141 /* row_half_mirror */
                     : LEVEL == 4 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* quad_perm [1,0,3,2] */;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xF, 0xF, true));
}
template <int LEVEL> __device__ __forceinline__ double dpp_partner(double x) {
  const long long b = __double_as_longlong(x);
  const float lo = dpp_partner<LEVEL>(__int_as_float(int(b & 0xffffffffLL)));
  const float hi = dpp_partner<LEVEL>(__int_as_float(int(b >> 32)));
  return __longlong_as_double((static_cast<long long>(__float_as_int(hi)) << 32) |
                              static_cast<long long>(static_cast<unsigned int>(__float_as_int(lo))));
}
// lanes whose bit is clear: a + a[partner]; lanes whose bit is set: b + b[partner]   (LEVEL 0: bit 5, 1: bit 4)
template <int LEVEL> __device__ __forceinline__ float swap_sum(float a, float b) {
  if constexpr (LEVEL == 0) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
template <int LEVEL> __device__ __forceinline__ double swap_sum(double a, double b) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const unsigned alo = unsigned(ba & 0xffffffffLL), ahi = unsigned(ba >> 32);
  const unsigned blo = unsigned(bb & 0xffffffffLL), bhi = unsigned(bb >> 32);
  unsigned x0, x1, y0, y1;
  if constexpr (LEVEL == 0) {
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    x0 = l[0]; y0 = l[1]; x1 = h[0]; y1 = h[1];
  } else {
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    x0 = l[0]; y0 = l[1]; x1 = h[0]; y1 = h[1];
  }
  return __longlong_as_double((static_cast<long long>(x1) << 32) | static_cast<long long>(x0)) +
         __longlong_as_double((static_cast<long long>(y1) << 32) | static_cast<long long>(y0));
}
template <int LEVEL, int N, typename T> __device__ __forceinline__ T wave_reduce_level(const T (&v)[N], int lane) {
  if constexpr (LEVEL == 6) {
    static_assert(N == 1, "64 lanes hold at most 64 values");
    return v[0];
  } else {
    constexpr int M = (N + 1) / 2;
    T o[M];
    const bool up = (lane >> (5 - LEVEL)) & 1;
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      if constexpr (LEVEL <= 1) {
        o[k] = swap_sum<LEVEL>(v[2 * k], v[2 * k + 1]);
      } else {
        const T keep = up ? v[2 * k + 1] : v[2 * k];
        const T send = up ? v[2 * k] : v[2 * k + 1];
        o[k] = keep + dpp_partner<LEVEL>(send);
      }
    }
    if constexpr (N & 1) {
      if constexpr (LEVEL <= 1) o[M - 1] = swap_sum<LEVEL>(v[N - 1], v[N - 1]);
      else o[M - 1] = v[N - 1] + dpp_partner<LEVEL>(v[N - 1]);
    }
    return wave_reduce_level<LEVEL + 1, M, T>(o, lane);
  }
}
// The wavefront total of value reduce_slot<N>(lane) lands in every lane.
template <int N, typename T> __device__ __forceinline__ T wave_reduce_transposed(const T (&v)[N], int lane) {
  static_assert(N >= 1 && N <= 64, "");
  return wave_reduce_level<0, N, T>(v, lane);
}
// Index of the value `wave_reduce_transposed<N>` leaves in `lane`; writer = false for lanes that hold a duplicate
// (every value has exactly one writer lane).
template <int N> __device__ __forceinline__ int reduce_slot(int lane, bool& writer) {
  int cnt[7];
  cnt[0] = N;
#pragma unroll
  for (int l = 0; l < 6; ++l) cnt[l + 1] = (cnt[l] + 1) / 2;
  int idx = 0;
  writer = true;
#pragma unroll
  for (int l = 5; l >= 0; --l) {
    const int n = cnt[l], m = cnt[l + 1];
    const int bit = (lane >> (5 - l)) & 1;
    if ((n & 1) && idx == m - 1) { idx = n - 1; writer = writer && !bit; }   // came through a butterfly
    else idx = 2 * idx + bit;
  }
  return idx;
}

template <typename T> __device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace mm
