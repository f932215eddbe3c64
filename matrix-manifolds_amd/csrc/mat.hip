// Grassmann Gr(N,p) and Stiefel St(N,p): projections, retractions, exp/log, the
// principal-angle distance and its gradient — one small N x p matrix per lane, in registers.
//
// Reference arithmetic: graphembed/graphembed/manifolds/grassmann.py:49-96 and
// stiefel.py:40-69, where every QR / SVD is shipped to the CPU (linalg/torch_batch.py:94-121).
// Here:  Q of QR    = Householder with LAPACK's sign convention (so Q matches torch.qr),
//        polar U V^T = Y (Y^T Y)^-1/2           via a p x p Jacobi eigensolve,
//        U f(S) V^T of a thin SVD = Y V f(s)/s V^T  with (s^2, V) = eig(Y^T Y),
//        singular values of x^T y             = sqrt eig((x^T y)^T (x^T y)).
// Points are stored [cnt][N][p] row-major; N is padded to NP in {4,6,9} (zero rows change
// nothing), p in {1,2,3,4} is a template parameter.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include <type_traits>

#include "smallmat.hpp"

namespace mm {
namespace mat {

constexpr double kEps = 1e-8;

template <typename T, int NP, int P> __device__ __forceinline__ void load(const T* __restrict__ p, int N, T (&a)[NP][P]) {
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) a[r][c] = (r < N) ? p[r * P + c] : T(0);
}
template <typename T, int NP, int P> __device__ __forceinline__ void store(T* __restrict__ p, int N, const T (&a)[NP][P]) {
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c)
      if (r < N) p[r * P + c] = a[r][c];
}

// G = A^T B (p x p)
template <typename T, int NP, int P>
__device__ __forceinline__ void gram(const T (&a)[NP][P], const T (&b)[NP][P], T (&g)[P][P]) {
#pragma unroll
  for (int i = 0; i < P; ++i)
#pragma unroll
    for (int j = 0; j < P; ++j) {
      T s = T(0);
#pragma unroll
      for (int r = 0; r < NP; ++r) s = Num<T>::fma(a[r][i], b[r][j], s);
      g[i][j] = s;
    }
}

// out = A * M (N x p times p x p)
template <typename T, int NP, int P>
__device__ __forceinline__ void mulr(const T (&a)[NP][P], const T (&m)[P][P], T (&o)[NP][P]) {
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) {
      T s = T(0);
#pragma unroll
      for (int k = 0; k < P; ++k) s = Num<T>::fma(a[r][k], m[k][c], s);
      o[r][c] = s;
    }
}

// symmetric p x p eigen-decomposition S = V diag(w) V^T
template <typename T, int P> __device__ __forceinline__ void symeig(const T (&s)[P][P], T (&w)[P], T (&v)[P][P]) {
  T a[Packed<P>::NP];
#pragma unroll
  for (int r = 0; r < P; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) a[pidx(r, c)] = T(0.5) * (s[r][c] + s[c][r]);
  jacobi_eig<T, P, true>(a, v);
#pragma unroll
  for (int k = 0; k < P; ++k) w[k] = a[pidx(k, k)];
}

// M = V diag(f) V^T
template <typename T, int P> __device__ __forceinline__ void vfvt(const T (&v)[P][P], const T (&f)[P], T (&m)[P][P]) {
#pragma unroll
  for (int r = 0; r < P; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) {
      T s = T(0);
#pragma unroll
      for (int k = 0; k < P; ++k) s = Num<T>::fma(v[r][k] * f[k], v[c][k], s);
      m[r][c] = s;
    }
}

// polar factor U V^T of Y = U S V^T   (grassmann.py:76-80, stiefel.py:66-69)
template <typename T, int NP, int P> __device__ __forceinline__ void polar(const T (&y)[NP][P], T (&q)[NP][P]) {
  T s[P][P], w[P], v[P][P], m[P][P], f[P];
  gram<T, NP, P>(y, y, s);
  symeig<T, P>(s, w, v);
#pragma unroll
  for (int k = 0; k < P; ++k) f[k] = Num<T>::rsqrt(Num<T>::max(w[k], Num<T>::tiny()));
  vfvt<T, P>(v, f, m);
  mulr<T, NP, P>(y, m, q);
}

// Q of the Householder QR of Y, LAPACK (geqrf/orgqr) sign convention: R_kk = -sgn(a_kk) ||.||.
// SIGNFIX multiplies column k by sgn(R_kk) (stiefel.py:47-50).
template <typename T, int NP, int P, bool SIGNFIX>
__device__ __forceinline__ void qr_q(const T (&y)[NP][P], int N, T (&q)[NP][P]) {
  T a[NP][P], tau[P], rs[P];
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) a[r][c] = y[r][c];
#pragma unroll
  for (int k = 0; k < P; ++k) {
    T xn = T(0);
#pragma unroll
    for (int r = k + 1; r < NP; ++r) xn = Num<T>::fma(a[r][k], a[r][k], xn);
    const T alpha = a[k][k];
    T beta = -Num<T>::copysign(Num<T>::sqrt(Num<T>::fma(alpha, alpha, xn)), alpha);
    const bool trivial = !(xn > T(0));  // LAPACK: H = I when the sub-column is zero
    tau[k] = trivial ? T(0) : (beta - alpha) / beta;
    const T scal = trivial ? T(0) : T(1) / (alpha - beta);
    rs[k] = trivial ? alpha : beta;
#pragma unroll
    for (int r = k + 1; r < NP; ++r) a[r][k] *= scal;  // v below the diagonal (v_k = 1)
#pragma unroll
    for (int c = k + 1; c < P; ++c) {                  // apply H to the trailing columns
      T d = a[k][c];
#pragma unroll
      for (int r = k + 1; r < NP; ++r) d = Num<T>::fma(a[r][k], a[r][c], d);
      d *= tau[k];
      a[k][c] -= d;
#pragma unroll
      for (int r = k + 1; r < NP; ++r) a[r][c] = Num<T>::fma(-d, a[r][k], a[r][c]);
    }
  }
  // Q = H_0 ... H_{p-1} [I_p; 0]
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) q[r][c] = (r == c) ? T(1) : T(0);
#pragma unroll
  for (int k = P - 1; k >= 0; --k) {
#pragma unroll
    for (int c = 0; c < P; ++c) {
      T d = q[k][c];
#pragma unroll
      for (int r = k + 1; r < NP; ++r) d = Num<T>::fma(a[r][k], q[r][c], d);
      d *= tau[k];
      q[k][c] -= d;
#pragma unroll
      for (int r = k + 1; r < NP; ++r) q[r][c] = Num<T>::fma(-d, a[r][k], q[r][c]);
    }
  }
  if (SIGNFIX) {
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T sg = (rs[c] > T(0)) ? T(1) : ((rs[c] < T(0)) ? T(-1) : T(0));
#pragma unroll
      for (int r = 0; r < NP; ++r) q[r][c] *= sg;
    }
  }
  (void)N;
}

// inverse of a p x p matrix (Gauss-Jordan, partial pivoting)
template <typename T, int P> __device__ __forceinline__ void inv_pp(const T (&m)[P][P], T (&inv)[P][P]) {
  T a[P][2 * P];
#pragma unroll
  for (int r = 0; r < P; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) { a[r][c] = m[r][c]; a[r][P + c] = (r == c) ? T(1) : T(0); }
#pragma unroll
  for (int k = 0; k < P; ++k) {
#pragma unroll
    for (int r = k + 1; r < P; ++r) {  // bring the larger pivot up (select-swap: no dynamic indexing)
      const bool sw = Num<T>::abs(a[r][k]) > Num<T>::abs(a[k][k]);
#pragma unroll
      for (int c = 0; c < 2 * P; ++c) { const T x = a[k][c], y = a[r][c]; a[k][c] = sw ? y : x; a[r][c] = sw ? x : y; }
    }
    const T ip = T(1) / a[k][k];
#pragma unroll
    for (int c = 0; c < 2 * P; ++c) a[k][c] *= ip;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      if (r == k) continue;
      const T f = a[r][k];
#pragma unroll
      for (int c = 0; c < 2 * P; ++c) a[r][c] = Num<T>::fma(-f, a[k][c], a[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < P; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) inv[r][c] = a[r][P + c];
}

template <typename T> __device__ __forceinline__ T acos_(T c);
template <> __device__ __forceinline__ float acos_<float>(float c) { return ::acosf(c); }
template <> __device__ __forceinline__ double acos_<double>(double c) { return ::acos(c); }
template <typename T> __device__ __forceinline__ T sincos_(T x, T* c);
template <> __device__ __forceinline__ float sincos_<float>(float x, float* c) { *c = ::cosf(x); return ::sinf(x); }
template <> __device__ __forceinline__ double sincos_<double>(double x, double* c) { *c = ::cos(x); return ::sin(x); }
template <typename T> __device__ __forceinline__ T atan_(T x);
template <> __device__ __forceinline__ float atan_<float>(float x) { return ::atanf(x); }
template <> __device__ __forceinline__ double atan_<double>(double x) { return ::atan(x); }

// Grassmann distance of one pair from G = x^T y:  sum_k acos^2(sigma_k)  (grassmann.py:91-96).
// Returns the value; if WANT_GRAD, dG = d(value)/dG = G V diag(f'(s)/s) V^T with
// f = acos^2 (the reference's NaN at sigma = 1, acos'(1), is replaced by the finite limit -2).
template <typename T> __device__ __forceinline__ T dacos2(T sc, T th) {
  // d acos^2(s)/ds = -2 acos(s)/sqrt(1-s^2); the 0/0 at s = 1 is replaced by its limit -2
  const T om = Num<T>::fma(-sc, sc, T(1));
  return T(-2) * ((om > T(1e-12)) ? th * Num<T>::rsqrt(om) : T(1));
}

template <typename T, int P, bool WANT_GRAD>
__device__ __forceinline__ T grass_pair(const T (&g)[P][P], T (&dg)[P][P]) {
  using N = Num<T>;
  if constexpr (P == 2) {
    // the reference's closed form for 2x2 singular values INCLUDING its eps clamps
    // (linalg/fast.py:138-159; they bias d^2 by ~1e-4 at its own init, so they are part of
    // the specification): S2 = (s1^2-s2^2)^2 >= eps, s_k^2 = (S1 +- sqrt S2)/2 >= eps.
    const T a = g[0][0], b = g[0][1], c = g[1][0], d = g[1][1];
    const T S1 = a * a + b * b + c * c + d * d;
    const T Dd = a * a + b * b - c * c - d * d, E = a * c + b * d;
    const T R = N::sqrt(N::max(N::fma(Dd, Dd, T(4) * E * E), T(kEps)));
    const T s1 = N::sqrt(N::max(T(0.5) * (S1 + R), T(kEps)));
    const T s2 = N::sqrt(N::max(T(0.5) * (S1 - R), T(kEps)));
    const T c1 = N::min(s1, T(1 - 1e-16)), c2 = N::min(s2, T(1 - 1e-16));
    const T t1 = acos_<T>(c1), t2 = acos_<T>(c2);
    if (WANT_GRAD) {
      // value clamps are gradient-transparent: d s_k = d(s_k^2)/(2 s_k), d s_{1,2}^2 = (dS1 +- dR)/2,
      // dR = (2 D dD + 8 E dE)/(2R)
      const T a1 = dacos2<T>(c1, t1) / (T(4) * s1), a2 = dacos2<T>(c2, t2) / (T(4) * s2);
      const T ps = a1 + a2, pr = (a1 - a2) / R;
      dg[0][0] = ps * (a + a) + pr * (Dd * (a + a) + T(4) * E * c);
      dg[0][1] = ps * (b + b) + pr * (Dd * (b + b) + T(4) * E * d);
      dg[1][0] = ps * (c + c) + pr * (-Dd * (c + c) + T(4) * E * a);
      dg[1][1] = ps * (d + d) + pr * (-Dd * (d + d) + T(4) * E * b);
    }
    return N::fma(t1, t1, t2 * t2);
  } else {
    // singular values from a one-sided Jacobi on G itself (not from the eigenvalues of G^T G: see svd_onesided)
    T b[P][P], v[P][P], nn[P];
    // (4 eps: with eps itself the test sits at the rounding of the inner product and some matrices never pass it; emulated
    // in fp32 / fp64 on 400 matrices with cosines from 1 - 1e-8 to 1e-9: at most 6 sweeps, singular values to 4e-7 / 1e-15)
    svd_onesided<T, P, true>(g, b, v, T(16) * N::eps() * N::eps());
    // sigma_k^2: the column norm ||b_k||^2 for the small ones (what the one-sided method is for); for cosines next to 1
    // (small angles: the reference's own initialisation) the Rayleigh quotient v_k^T (G^T G) v_k / v_k^T v_k — the rotations
    // leave ~6 eps of norm drift in b and v, which the quotient cancels and which acos would amplify ~100 x at sigma ~ 1 - 1e-4
    // (emulated, angles ~1e-2: d^2 to 6.7e-3 from the column norms, 2.0e-3 from the quotient, 1.9e-3 through eig(G^T G))
    T sgram[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
      for (int j = 0; j < P; ++j) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < P; ++k) acc = N::fma(g[k][i], g[k][j], acc);
        sgram[i][j] = acc;
      }
    T val = T(0), f[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      nn[k] = T(0);
      T num = T(0), den = T(0);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        nn[k] = N::fma(b[i][k], b[i][k], nn[k]);
        T sv = T(0);
#pragma unroll
        for (int j = 0; j < P; ++j) sv = N::fma(sgram[i][j], v[j][k], sv);
        num = N::fma(v[i][k], sv, num);
        den = N::fma(v[i][k], v[i][k], den);
      }
      const T s2 = nn[k] > T(0.25) ? num / den : nn[k];
      const T sg = N::sqrt(N::max(s2, T(0)));
      const T sc = N::min(sg, T(1 - 1e-16));  // value clamp (grassmann.py:94)
      const T th = acos_<T>(sc);
      val = N::fma(th, th, val);
      if (WANT_GRAD) f[k] = dacos2<T>(sc, th);
    }
    if (WANT_GRAD) {
      // dG = U diag(f') V^T with the left vectors u_k = b_k / ||b_k|| (b = G V).  Normalising by the ACTUAL norm keeps
      // every term bounded when a principal angle is ~pi/2 (sigma_k ~ 0) — tests/fuzz_misc.py once found 1e25-sized
      // gradients in fp32 with a division by sqrt(eigenvalue of G^T G).
#pragma unroll
      for (int k = 0; k < P; ++k) f[k] = f[k] * N::rsqrt(N::max(nn[k], T(1e-30)));
#pragma unroll
      for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) {
          T acc = T(0);
#pragma unroll
          for (int k = 0; k < P; ++k) acc = N::fma(b[i][k] * f[k], v[j][k], acc);
          dg[i][j] = acc;
        }
    }
    return val;
  }
}

// ------------------------------------------------------------ per-point maps
template <typename T, int NP, int P>
__global__ void mat_map_kernel(int kind, int op, const T* __restrict__ x, const T* __restrict__ u, int64_t cnt, int N,
                               T* __restrict__ out) {
  using Nm = Num<T>;
  const int64_t p0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = p0 < cnt;
  const int64_t p = in ? p0 : 0;
  T xa[NP][P], ua[NP][P], o[NP][P];
  load<T, NP, P>(x + p * N * P, N, xa);
  if (op != MM_MAT_PROJX) load<T, NP, P>(u + p * N * P, N, ua);
  if (op == MM_MAT_PROJU) {
    T g[P][P];
    gram<T, NP, P>(xa, ua, g);  // x^T u
    if (kind == MM_STIEFEL) {   // u - x sym(x^T u)   stiefel.py:40-45
#pragma unroll
      for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = i + 1; j < P; ++j) { const T h = T(0.5) * (g[i][j] + g[j][i]); g[i][j] = h; g[j][i] = h; }
    }                           // else u - x x^T u   grassmann.py:49-53
    mulr<T, NP, P>(xa, g, o);
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) o[r][c] = ua[r][c] - o[r][c];
  } else if (op == MM_MAT_PROJX) {  // grassmann.py:55-61 / stiefel.py:47-57
    if (kind == MM_STIEFEL) qr_q<T, NP, P, true>(xa, N, o); else qr_q<T, NP, P, false>(xa, N, o);
  } else if (op == MM_MAT_RETR_SVD || op == MM_MAT_RETR_QR) {
    T y[NP][P];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) y[r][c] = xa[r][c] + ua[r][c];
    if (op == MM_MAT_RETR_SVD) polar<T, NP, P>(y, o);
    else if (kind == MM_STIEFEL) qr_q<T, NP, P, true>(y, N, o);
    else qr_q<T, NP, P, false>(y, N, o);
  } else if (op == MM_MAT_EXP) {  // x V cos(S) V^T + U sin(S) V^T, u = U S V^T  (grassmann.py:63-69)
    T s[P][P], w[P], v[P][P], fc[P], fs[P], mc[P][P], ms[P][P], t1[NP][P], t2[NP][P];
    gram<T, NP, P>(ua, ua, s);
    symeig<T, P>(s, w, v);
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const T sg = Nm::sqrt(Nm::max(w[k], T(0)));
      T c;
      const T sn = sincos_<T>(sg, &c);
      fc[k] = c;
      fs[k] = (sg > T(1e-6)) ? sn / sg : T(1) - sg * sg * T(1.0 / 6.0);
    }
    vfvt<T, P>(v, fc, mc);
    vfvt<T, P>(v, fs, ms);
    mulr<T, NP, P>(xa, mc, t1);
    mulr<T, NP, P>(ua, ms, t2);
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) o[r][c] = t1[r][c] + t2[r][c];
  } else {  // MM_MAT_LOG: log_x(y), y in `u`  (grassmann.py:82-89)
    T ytx[P][P], inv[P][P], b[NP][P];
    gram<T, NP, P>(ua, xa, ytx);  // y^T x
    inv_pp<T, P>(ytx, inv);
    // B = (y - x (y^T x)^T) (y^T x)^-T     (N x p);  B^T = solve(ytx, y^T - ytx x^T)
    T a[NP][P];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) {
        T acc = ua[r][c];
#pragma unroll
        for (int k = 0; k < P; ++k) acc = Nm::fma(-xa[r][k], ytx[c][k], acc);
        a[r][c] = acc;
      }
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < P; ++k) acc = Nm::fma(a[r][k], inv[c][k], acc);
        b[r][c] = acc;
      }
    T s[P][P], w[P], v[P][P], f[P], m[P][P];
    gram<T, NP, P>(b, b, s);
    symeig<T, P>(s, w, v);
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const T sg = Nm::sqrt(Nm::max(w[k], T(0)));
      f[k] = (sg > T(1e-6)) ? atan_<T>(sg) / sg : T(1) - sg * sg * T(1.0 / 3.0);
    }
    vfvt<T, P>(v, f, m);
    mulr<T, NP, P>(b, m, o);
  }
  if (in) store<T, NP, P>(out + p * N * P, N, o);
}

// ------------------------------------------------------- element-wise dist
template <typename T, int NP, int P>
__global__ void grass_dist_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ g,
                                  int64_t cnt, int N, int squared, T* __restrict__ out, T* __restrict__ gx,
                                  T* __restrict__ gy) {
  using Nm = Num<T>;
  const int64_t p0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = p0 < cnt;
  const int64_t p = in ? p0 : 0;
  T xa[NP][P], ya[NP][P], gm[P][P], dg[P][P];
  load<T, NP, P>(x + p * N * P, N, xa);
  load<T, NP, P>(y + p * N * P, N, ya);
  gram<T, NP, P>(xa, ya, gm);
  if (gx == nullptr) {
    const T v = grass_pair<T, P, false>(gm, dg);
    if (in && out) out[p] = squared ? v : Nm::sqrt(v);
    return;
  }
  const T v = grass_pair<T, P, true>(gm, dg);
  if (in && out) out[p] = squared ? v : Nm::sqrt(v);
  T w = g[p];
  if (!squared) w *= T(0.5) * Nm::rsqrt(v);
  T ox[NP][P], oy[NP][P];
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) {
      T ax = T(0), ay = T(0);
#pragma unroll
      for (int k = 0; k < P; ++k) { ax = Nm::fma(ya[r][k], dg[c][k], ax); ay = Nm::fma(xa[r][k], dg[k][c], ay); }
      ox[r][c] = w * ax;  // d/dx = y dG^T
      oy[r][c] = w * ay;  // d/dy = x dG
    }
  if (in) { store<T, NP, P>(gx + p * N * P, N, ox); store<T, NP, P>(gy + p * N * P, N, oy); }
}

// ------------------------------------------------------------------ pdist
constexpr int kBlk = 128;
__host__ __device__ inline int64_t moff(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

template <typename T, int NP, int P, int TI>
__global__ __launch_bounds__(kBlk) void grass_pdist_fwd_kernel(const T* __restrict__ x, int n, int N, int row_begin,
                                                               int row_end, int squared, T* __restrict__ out) {
  const int i0 = row_begin + blockIdx.y * TI, i1 = min(i0 + TI, row_end);
  const int jbase = ((i0 + 1) / kBlk + blockIdx.x) * kBlk;
  if (jbase >= n) return;
  if (jbase + (int(threadIdx.x) & ~63) + 63 <= i0) return;
  const int j = jbase + threadIdx.x;
  const bool jin = j < n;
  T xj[NP][P];
  load<T, NP, P>(x + size_t(jin ? j : 0) * N * P, N, xj);
  const int64_t base = moff(n, row_begin);
  for (int i = i0; i < i1; ++i) {
    T xi[NP][P], gm[P][P], dg[P][P];
    load<T, NP, P>(x + size_t(i) * N * P, N, xi);  // wave-uniform -> scalar loads
    gram<T, NP, P>(xi, xj, gm);
    const T v = grass_pair<T, P, false>(gm, dg);
    if (jin && j > i) out[moff(n, i) - base + (j - i - 1)] = squared ? v : Num<T>::sqrt(v);
  }
}

// every ordered pair: a lane accumulates only its own column's gradient (as in vec.hip)
template <typename T, int NP, int P, int TI>
__global__ __launch_bounds__(kBlk) void grass_pdist_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, int n,
                                                               int N, int row_begin, int row_end, int squared,
                                                               T* __restrict__ acc /* [NP*P][n] */) {
  using Nm = Num<T>;
  const int j = blockIdx.x * kBlk + threadIdx.x;
  const int i0 = blockIdx.y * TI, i1 = min(i0 + TI, n);
  const bool jin = j < n;
  const bool jown = jin && j >= row_begin && j < row_end;
  T xj[NP][P], a[NP][P];
  load<T, NP, P>(x + size_t(jin ? j : 0) * N * P, N, xj);
#pragma unroll
  for (int r = 0; r < NP; ++r)
#pragma unroll
    for (int c = 0; c < P; ++c) a[r][c] = T(0);
  const int64_t base = moff(n, row_begin);
  for (int i = i0; i < i1; ++i) {
    T xi[NP][P], gm[P][P], dg[P][P];
    load<T, NP, P>(x + size_t(i) * N * P, N, xi);
    const bool up = i < j;
    const bool valid = jin && i != j && (up ? (i >= row_begin && i < row_end) : jown);
    const int lo = up ? i : j, hi = up ? j : i;
    T w = T(0);
    if (valid) w = g[moff(n, lo) - base + (hi - lo - 1)];
    gram<T, NP, P>(xi, xj, gm);  // x_i^T x_j ; d/dx_j = x_i dG
    const T v = grass_pair<T, P, true>(gm, dg);
    if (!squared) w *= T(0.5) * Nm::rsqrt(v);
    w = valid ? w : T(0);
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c) {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < P; ++k) s = Nm::fma(xi[r][k], dg[k][c], s);
        a[r][c] = Nm::fma(w, s, a[r][c]);
      }
  }
  if (jin) {
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
      for (int c = 0; c < P; ++c)
        if (r < N) atomic_add(&acc[size_t(r * P + c) * n + j], a[r][c]);
  }
}

template <typename T>
__global__ void grass_finalize_kernel(const T* __restrict__ acc, int n, int np, T* __restrict__ grad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  for (int k = 0; k < np; ++k) grad[size_t(j) * np + k] = acc[size_t(k) * n + j];
}

constexpr int pad_rows(int N) { return N <= 4 ? 4 : N <= 6 ? 6 : 9; }

#define MMM_CHECK()                                    \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) return static_cast<int>(e_); \
  } while (0)

#define MMM_DISPATCH_NP_P(N, p, ...)                                            \
  switch (pad_rows(N) * 10 + (p)) {                                             \
    case 41: { constexpr int NP = 4, P = 1; __VA_ARGS__ }                        \
    case 42: { constexpr int NP = 4, P = 2; __VA_ARGS__ }                        \
    case 43: { constexpr int NP = 4, P = 3; __VA_ARGS__ }                        \
    case 44: { constexpr int NP = 4, P = 4; __VA_ARGS__ }                        \
    case 61: { constexpr int NP = 6, P = 1; __VA_ARGS__ }                        \
    case 62: { constexpr int NP = 6, P = 2; __VA_ARGS__ }                        \
    case 63: { constexpr int NP = 6, P = 3; __VA_ARGS__ }                        \
    case 64: { constexpr int NP = 6, P = 4; __VA_ARGS__ }                        \
    case 91: { constexpr int NP = 9, P = 1; __VA_ARGS__ }                        \
    case 92: { constexpr int NP = 9, P = 2; __VA_ARGS__ }                        \
    case 93: { constexpr int NP = 9, P = 3; __VA_ARGS__ }                        \
    case 94: { constexpr int NP = 9, P = 4; __VA_ARGS__ }                        \
    default: return MM_ERR_UNSUPPORTED;                                         \
  }

#define MMM_DISPATCH_T(dtype, ...)                               \
  if ((dtype) == MM_F32) { using T = float; __VA_ARGS__ }        \
  else if ((dtype) == MM_F64) { using T = double; __VA_ARGS__ }  \
  else return MM_ERR_ARG;

}  // namespace mat
}  // namespace mm

using namespace mm;
using namespace mm::mat;

extern "C" {

int mm_mat_max_rows(void) { return 9; }
int mm_mat_max_cols(void) { return 4; }

size_t mm_grass_pdist_ws_bytes(int dtype, int64_t n, int N, int p) {
  return (dtype == MM_F64 ? 8 : 4) * size_t(n) * size_t(N) * size_t(p);
}

int mm_mat_map(int dtype, int kind, int op, const void* x, const void* u, int64_t cnt, int N, int p, void* out,
               mm_stream_t stream) {
  if (cnt < 0 || N < 1 || p < 1 || p > N || op < 0 || op > MM_MAT_LOG || (kind != MM_GRASSMANN && kind != MM_STIEFEL))
    return MM_ERR_ARG;
  if (cnt > 0 && (!x || !out || (op != MM_MAT_PROJX && !u))) return MM_ERR_ARG;
  if (kind == MM_STIEFEL && (op == MM_MAT_EXP || op == MM_MAT_LOG)) return MM_ERR_UNSUPPORTED;  // stiefel.py:59-75
  if (N > 9 || p > 4) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = unsigned((cnt + 63) / 64);
  MMM_DISPATCH_T(dtype, MMM_DISPATCH_NP_P(N, p, {
    mat_map_kernel<T, NP, P><<<dim3(nb), dim3(64), 0, st>>>(kind, op, static_cast<const T*>(x),
        static_cast<const T*>(u), cnt, N, static_cast<T*>(out));
    MMM_CHECK(); return MM_OK; }))
}

int mm_grass_dist(int dtype, const void* x, const void* y, const void* g, int64_t cnt, int N, int p, int squared,
                  void* out, void* grad_x, void* grad_y, mm_stream_t stream) {
  if (cnt < 0 || N < 1 || p < 1 || p > N || (cnt > 0 && (!x || !y)) || ((grad_x != nullptr) != (grad_y != nullptr)) ||
      (grad_x && !g))
    return MM_ERR_ARG;
  if (N > 9 || p > 4) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = unsigned((cnt + 63) / 64);
  MMM_DISPATCH_T(dtype, MMM_DISPATCH_NP_P(N, p, {
    grass_dist_kernel<T, NP, P><<<dim3(nb), dim3(64), 0, st>>>(static_cast<const T*>(x), static_cast<const T*>(y),
        static_cast<const T*>(g), cnt, N, squared, static_cast<T*>(out), static_cast<T*>(grad_x),
        static_cast<T*>(grad_y));
    MMM_CHECK(); return MM_OK; }))
}

int mm_grass_pdist_fwd(int dtype, const void* x, int64_t n, int N, int p, int64_t row_begin, int64_t row_end,
                       int squared, void* out, mm_stream_t stream) {
  if (!x || n < 0 || N < 1 || p < 1 || p > N || row_begin < 0 || row_end > n || row_begin > row_end || n > (1 << 30))
    return MM_ERR_ARG;
  if (N > 9 || p > 4) return MM_ERR_UNSUPPORTED;
  if (!out && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  if (row_end <= row_begin) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int TI = 16;
  const int gx = int((n + kBlk - 1) / kBlk) - int((row_begin + 1) / kBlk);
  const int gy = int((row_end - row_begin + TI - 1) / TI);
  if (gx <= 0) return MM_OK;
  MMM_DISPATCH_T(dtype, MMM_DISPATCH_NP_P(N, p, {
    grass_pdist_fwd_kernel<T, NP, P, TI><<<dim3(gx, gy), dim3(kBlk), 0, st>>>(static_cast<const T*>(x), int(n), N,
        int(row_begin), int(row_end), squared, static_cast<T*>(out));
    MMM_CHECK(); return MM_OK; }))
}

int mm_grass_pdist_bwd(int dtype, const void* x, const void* g, int64_t n, int N, int p, int64_t row_begin,
                       int64_t row_end, int squared, void* grad_x, void* ws, mm_stream_t stream) {
  if (!x || !grad_x || !ws || n < 1 || N < 1 || p < 1 || p > N || row_begin < 0 || row_end > n ||
      row_begin > row_end || n > (1 << 30))
    return MM_ERR_ARG;
  if (N > 9 || p > 4) return MM_ERR_UNSUPPORTED;
  if (!g && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int TI = 32;
  hipError_t e = hipMemsetAsync(ws, 0, mm_grass_pdist_ws_bytes(dtype, n, N, p), st);
  if (e != hipSuccess) return int(e);
  MMM_DISPATCH_T(dtype, MMM_DISPATCH_NP_P(N, p, {
    if (row_end > row_begin) {
      grass_pdist_bwd_kernel<T, NP, P, TI><<<dim3(int((n + kBlk - 1) / kBlk), int((n + TI - 1) / TI)), dim3(kBlk), 0, st>>>(
          static_cast<const T*>(x), static_cast<const T*>(g), int(n), N, int(row_begin), int(row_end), squared,
          static_cast<T*>(ws));
      MMM_CHECK();
    }
    grass_finalize_kernel<T><<<dim3(int((n + 127) / 128)), dim3(128), 0, st>>>(static_cast<const T*>(ws), int(n),
                                                                              N * p, static_cast<T*>(grad_x));
    MMM_CHECK(); return MM_OK; }))
}

}  // extern "C"
