// graphembed/graphembed/linalg/fast.py — the closed forms for stacks of 2x2 / 3x3 matrices that the SPD and Grassmann
// manifolds are built from (spd.py:35-46, grassmann.py:29) and that the reference's monitor and its tests call by name
// (monitor.py:39-45, tests/test_linalg.py:47-141, tests/test_perf.py:14-81), forward and backward.
//
// The forward of every function is fast.py's arithmetic, operation for operation, including which HALF of a symmetric
// matrix it reads and its `x.data.clamp_(...)` guards (the value is clamped, the derivative passes as if it were not);
// the backward is what torch's autograd makes of that arithmetic — hence gradients that live on the upper triangle only
// (fast.py:5-10 says so), which the callers symmetrise.
//
// One thread per matrix.  The streams are [n][k] with k = 4 or 9 numbers per matrix (k = 1 ... 4 on the output side): a
// workgroup moves its 256 matrices between HBM and LDS as one contiguous, fully coalesced block and every thread picks
// its own matrix out of LDS (stride 9 and stride 3 words are free of bank conflicts; 4-number records go as one 16- or
// 2 x 16-byte access).  HBM-bound: 4 (k_in + k_out) bytes per matrix in fp32.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "smallmat.hpp"

namespace mm {

constexpr int kFastBlock = 256;

template <int OP> struct FastShape;   // numbers per matrix: input, first output, second output (0 = none)
template <> struct FastShape<MM_FAST_SYMEIG2> { static constexpr int kIn = 4, kOut = 2, kOut2 = 0; };
template <> struct FastShape<MM_FAST_SYMEIG3> { static constexpr int kIn = 9, kOut = 3, kOut2 = 0; };
template <> struct FastShape<MM_FAST_CHOLESKY2> { static constexpr int kIn = 4, kOut = 4, kOut2 = 0; };
template <> struct FastShape<MM_FAST_INVCHOLESKY2> { static constexpr int kIn = 4, kOut = 4, kOut2 = 4; };
template <> struct FastShape<MM_FAST_SINGULAR2> { static constexpr int kIn = 4, kOut = 2, kOut2 = 0; };
template <> struct FastShape<MM_FAST_DET2> { static constexpr int kIn = 4, kOut = 1, kOut2 = 0; };
template <> struct FastShape<MM_FAST_DET3> { static constexpr int kIn = 9, kOut = 1, kOut2 = 0; };
template <> struct FastShape<MM_FAST_SYMDET3> { static constexpr int kIn = 9, kOut = 1, kOut2 = 0; };

// `t.data.clamp_(min=lo)` / `clamp_(min=lo, max=hi)`: NaN stays NaN, as in torch
template <typename T> __device__ __forceinline__ T clamp_min(T v, T lo) { return v < lo ? lo : v; }
template <typename T> __device__ __forceinline__ T clamp_both(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T> struct FastFn {
  static __device__ __forceinline__ T sqrt_(T v) { return Num<T>::sqrt(v); }
  static __device__ __forceinline__ T cos_(T v) { if constexpr (std::is_same<T, float>::value) return cosf(v); else return cos(v); }
  static __device__ __forceinline__ T sin_(T v) { if constexpr (std::is_same<T, float>::value) return sinf(v); else return sin(v); }
  static __device__ __forceinline__ T acos_(T v) { if constexpr (std::is_same<T, float>::value) return acosf(v); else return acos(v); }
  static constexpr double kTwoThirdsPi = 2.0 * 3.14159265358979323846 / 3.0;

  // ---- symeig2x2 (fast.py:53-70): reads x00, x11, x01 --------------------------------------------------------------
  template <bool BWD>
  static __device__ __forceinline__ void symeig2(const T (&x)[4], T eps, T (&w)[2], const T (&g)[2], T (&gx)[4]) {
    const T a = x[0], b = x[3], c = x[1];
    const T det = a * b - c * c;
    const T ht = T(0.5) * (a + b);
    const T delta = clamp_min(ht * ht - det, eps);
    const T r = sqrt_(delta);
    w[0] = ht - r;
    w[1] = ht + r;
    if constexpr (BWD) {
      const T g_delta = (g[1] - g[0]) / (r + r);
      const T g_ht = g[0] + g[1] + (ht + ht) * g_delta;
      gx[0] = T(0.5) * g_ht - b * g_delta;
      gx[3] = T(0.5) * g_ht - a * g_delta;
      gx[1] = (c + c) * g_delta;
      gx[2] = T(0);
    }
  }

  // ---- symdet3x3 (fast.py:40-50): upper triangle; gradient of the value into the six entries it reads --------------
  static __device__ __forceinline__ T symdet3(const T (&y)[9]) {
    return y[0] * y[4] * y[8] + T(2) * y[1] * y[2] * y[5] - y[4] * (y[2] * y[2]) - y[0] * (y[5] * y[5]) - y[8] * (y[1] * y[1]);
  }
  static __device__ __forceinline__ void symdet3_grad(const T (&y)[9], T gd, T (&gy)[9]) {
    gy[0] += gd * (y[4] * y[8] - y[5] * y[5]);
    gy[4] += gd * (y[0] * y[8] - y[2] * y[2]);
    gy[8] += gd * (y[0] * y[4] - y[1] * y[1]);
    gy[1] += gd * T(2) * (y[2] * y[5] - y[8] * y[1]);
    gy[2] += gd * T(2) * (y[1] * y[5] - y[4] * y[2]);
    gy[5] += gd * T(2) * (y[1] * y[2] - y[0] * y[5]);
  }

  // ---- symeig3x3 (fast.py:75-91): trigonometric roots; tr(Y^2) over all nine entries, det(Y) over the upper six -----
  template <bool BWD>
  static __device__ __forceinline__ void symeig3(const T (&x)[9], T eps, double eps_d, T (&w)[3], const T (&g)[3], T (&gx)[9]) {
    const T q = (x[0] + x[4] + x[8]) / T(3);
    T y[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) y[k] = x[k];
    y[0] -= q; y[4] -= q; y[8] -= q;
    T ss = T(0);
#pragma unroll
    for (int k = 0; k < 9; ++k) ss += y[k] * y[k];
    const T p = clamp_min(sqrt_(ss / T(6)), eps);
    const T den = T(2) * (p * p * p) + eps;
    const T dy = symdet3(y);
    const T r = clamp_both(dy / den, T(-1.0 + eps_d), T(1.0 - eps_d));
    const T phi = acos_(r) / T(3);
    const T c1 = cos_(phi), c2 = cos_(phi + T(kTwoThirdsPi));
    const T e1 = q + T(2) * p * c1;
    const T e2 = q + T(2) * p * c2;
    const T e3 = T(3) * q - e1 - e2;
    w[0] = e2; w[1] = e3; w[2] = e1;
    if constexpr (BWD) {
      const T g_e2 = g[0] - g[1], g_e1 = g[2] - g[1];
      const T g_q = T(3) * g[1] + g_e1 + g_e2;
      T g_p = T(2) * (c1 * g_e1 + c2 * g_e2);
      const T g_phi = -T(2) * p * (sin_(phi) * g_e1 + sin_(phi + T(kTwoThirdsPi)) * g_e2);
      const T g_r = -(g_phi / T(3)) / sqrt_(T(1) - r * r);
      const T g_dy = g_r / den;
      g_p -= g_r * dy / (den * den) * (T(6) * p * p);
      const T g_ss = g_p / (T(12) * p);
      T gy[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) gy[k] = (y[k] + y[k]) * g_ss;
      symdet3_grad(y, g_dy, gy);
      const T diag = (g_q - (gy[0] + gy[4] + gy[8])) / T(3);   // Y = X - q I, q = tr X / 3
#pragma unroll
      for (int k = 0; k < 9; ++k) gx[k] = gy[k];
      gx[0] += diag; gx[4] += diag; gx[8] += diag;
    }
  }

  // ---- cholesky2x2 / invcholesky2x2 (fast.py:94-135): reads x00, x11, x01 ------------------------------------------
  // L = [[a, 0], [b, c]], a = sqrt(max(x00, eps)), b = x01 / a, c = sqrt(x11 - b^2 + eps); the chain back from (a, b, c)
  static __device__ __forceinline__ void chol2_abc(const T (&x)[4], T eps, T& a, T& b, T& c) {
    a = sqrt_(clamp_min(x[0], eps));
    b = x[1] / a;
    c = sqrt_(x[3] - b * b + eps);
  }
  static __device__ __forceinline__ void chol2_back(T a, T b, T c, T g_a, T g_b, T g_c, T (&gx)[4]) {
    const T g_t = g_c / (c + c);
    g_b -= (b + b) * g_t;
    g_a -= (b / a) * g_b;
    gx[0] = g_a / (a + a);
    gx[1] = g_b / a;
    gx[2] = T(0);
    gx[3] = g_t;
  }
  template <bool BWD>
  static __device__ __forceinline__ void cholesky2(const T (&x)[4], T eps, T (&l)[4], const T (&g)[4], T (&gx)[4]) {
    T a, b, c;
    chol2_abc(x, eps, a, b, c);
    l[0] = a; l[1] = T(0); l[2] = b; l[3] = c;
    if constexpr (BWD) chol2_back(a, b, c, g[0], g[2], g[3], gx);
  }
  // l_inv = [[c, 0], [-b, a]] / max(a c, eps); `g2` = the cotangent of the factor when the caller asked for it, else zeros
  template <bool BWD>
  static __device__ __forceinline__ void invcholesky2(const T (&x)[4], T eps, T (&li)[4], T (&l)[4], const T (&g)[4],
                                                       const T (&g2)[4], T (&gx)[4]) {
    T a, b, c;
    chol2_abc(x, eps, a, b, c);
    const T det = clamp_min(a * c, eps);
    li[0] = c / det; li[1] = T(0) / det; li[2] = -b / det; li[3] = a / det;
    l[0] = a; l[1] = T(0); l[2] = b; l[3] = c;
    if constexpr (BWD) {
      const T g_det = -(g[0] * c - g[2] * b + g[3] * a) / (det * det);
      const T g_a = g[3] / det + c * g_det + g2[0];
      const T g_b = -g[2] / det + g2[2];
      const T g_c = g[0] / det + a * g_det + g2[3];
      chol2_back(a, b, c, g_a, g_b, g_c, gx);
    }
  }

  // ---- singular_values_2x2 (fast.py:138-159) -----------------------------------------------------------------------
  template <bool BWD>
  static __device__ __forceinline__ void singular2(const T (&x)[4], T eps, T (&s)[2], const T (&g)[2], T (&gx)[4]) {
    const T a = x[0], b = x[1], c = x[2], d = x[3];
    const T S1 = a * a + b * b + c * c + d * d;
    const T u = a * a + b * b - c * c - d * d;
    const T v = a * c + b * d;
    const T S2 = sqrt_(clamp_min(u * u + T(4) * (v * v), eps));
    const T s1 = clamp_min(T(0.5) * (S1 + S2), eps);
    const T s2 = clamp_min(T(0.5) * (S1 - S2), eps);
    s[0] = sqrt_(s1);
    s[1] = sqrt_(s2);
    if constexpr (BWD) {
      const T g_s1 = g[0] / (s[0] + s[0]), g_s2 = g[1] / (s[1] + s[1]);
      const T g_S1 = T(0.5) * (g_s1 + g_s2);
      const T g_S2sq = T(0.5) * (g_s1 - g_s2) / (S2 + S2);
      const T g_u = (u + u) * g_S2sq, g_v = T(8) * v * g_S2sq;
      gx[0] = (a + a) * (g_S1 + g_u) + c * g_v;
      gx[1] = (b + b) * (g_S1 + g_u) + d * g_v;
      gx[2] = (c + c) * (g_S1 - g_u) + a * g_v;
      gx[3] = (d + d) * (g_S1 - g_u) + b * g_v;
    }
  }

  // ---- det2x2 / det3x3 (fast.py:25-37): the full matrix --------------------------------------------------------------
  template <bool BWD>
  static __device__ __forceinline__ void det2(const T (&x)[4], T (&o)[1], const T (&g)[1], T (&gx)[4]) {
    o[0] = x[0] * x[3] - x[1] * x[2];
    if constexpr (BWD) { gx[0] = g[0] * x[3]; gx[1] = -g[0] * x[2]; gx[2] = -g[0] * x[1]; gx[3] = g[0] * x[0]; }
  }
  template <bool BWD>
  static __device__ __forceinline__ void det3(const T (&x)[9], T (&o)[1], const T (&g)[1], T (&gx)[9]) {
    const T m1 = x[4] * x[8] - x[5] * x[7];
    const T m2 = x[3] * x[8] - x[5] * x[6];
    const T m3 = x[3] * x[7] - x[4] * x[6];
    o[0] = x[0] * m1 - x[1] * m2 + x[2] * m3;
    if constexpr (BWD) {
      const T s = g[0];
      gx[0] = s * m1; gx[1] = -s * m2; gx[2] = s * m3;
      gx[3] = s * (x[2] * x[7] - x[1] * x[8]);
      gx[4] = s * (x[0] * x[8] - x[2] * x[6]);
      gx[5] = s * (x[1] * x[6] - x[0] * x[7]);
      gx[6] = s * (x[1] * x[5] - x[2] * x[4]);
      gx[7] = s * (x[2] * x[3] - x[0] * x[5]);
      gx[8] = s * (x[0] * x[4] - x[1] * x[3]);
    }
  }
};

// a workgroup's 256 records of K numbers: HBM -> LDS as one contiguous block, then one record per thread (and back)
template <typename T, int K>
__device__ __forceinline__ void tile_load(const T* __restrict__ src, int64_t first, int64_t n, T* lds, T (&v)[K]) {
  const int64_t base = first * K;
  const int64_t left = (n - first) * K;   // numbers of this tile that exist
  __syncthreads();                         // (the previous use of `lds` is over)
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * kFastBlock + int(threadIdx.x);
    if (e < left) lds[e] = src[base + e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = lds[int(threadIdx.x) * K + k];
}
template <typename T, int K>
__device__ __forceinline__ void tile_store(T* __restrict__ dst, int64_t first, int64_t n, T* lds, const T (&v)[K]) {
  const int64_t base = first * K;
  const int64_t left = (n - first) * K;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) lds[int(threadIdx.x) * K + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * kFastBlock + int(threadIdx.x);
    if (e < left) dst[base + e] = lds[e];
  }
}

template <typename T, int OP, bool BWD, bool HAS2>
__global__ __launch_bounds__(kFastBlock) void fast_kernel(const T* __restrict__ x, int64_t n, T eps, double eps_d,
                                                           T* __restrict__ out, T* __restrict__ out2,
                                                           const T* __restrict__ gout, const T* __restrict__ gout2,
                                                           T* __restrict__ gx) {
  using S = FastShape<OP>;
  using F = FastFn<T>;
  constexpr int K2 = S::kOut2 > 0 ? S::kOut2 : 1;
  __shared__ T lds[kFastBlock * S::kIn];
  const int64_t first = int64_t(blockIdx.x) * kFastBlock;
  T v[S::kIn], o[S::kOut], o2[K2], g[S::kOut], g2[K2], gv[S::kIn];
  tile_load<T, S::kIn>(x, first, n, lds, v);
#pragma unroll
  for (int k = 0; k < S::kOut; ++k) g[k] = T(0);
#pragma unroll
  for (int k = 0; k < K2; ++k) g2[k] = T(0);
  if constexpr (BWD) {
    tile_load<T, S::kOut>(gout, first, n, lds, g);
    if constexpr (HAS2) tile_load<T, K2>(gout2, first, n, lds, g2);
  }
  if constexpr (OP == MM_FAST_SYMEIG2) F::template symeig2<BWD>(v, eps, o, g, gv);
  else if constexpr (OP == MM_FAST_SYMEIG3) F::template symeig3<BWD>(v, eps, eps_d, o, g, gv);
  else if constexpr (OP == MM_FAST_CHOLESKY2) F::template cholesky2<BWD>(v, eps, o, g, gv);
  else if constexpr (OP == MM_FAST_INVCHOLESKY2) F::template invcholesky2<BWD>(v, eps, o, o2, g, g2, gv);
  else if constexpr (OP == MM_FAST_SINGULAR2) F::template singular2<BWD>(v, eps, o, g, gv);
  else if constexpr (OP == MM_FAST_DET2) F::template det2<BWD>(v, o, g, gv);
  else if constexpr (OP == MM_FAST_DET3) F::template det3<BWD>(v, o, g, gv);
  else {
    o[0] = F::symdet3(v);
    if constexpr (BWD) {
#pragma unroll
      for (int k = 0; k < 9; ++k) gv[k] = T(0);
      F::symdet3_grad(v, g[0], gv);
    }
  }
  if constexpr (BWD) {
    tile_store<T, S::kIn>(gx, first, n, lds, gv);
  } else {
    tile_store<T, S::kOut>(out, first, n, lds, o);
    if constexpr (HAS2) tile_store<T, K2>(out2, first, n, lds, o2);
  }
}

template <typename T, int OP, bool BWD>
int fast_launch(const void* x, int64_t n, double eps, void* out, void* out2, const void* gout, const void* gout2, void* gx,
                hipStream_t st) {
  const dim3 grid((unsigned)((n + kFastBlock - 1) / kFastBlock)), block(kFastBlock);
  const bool has2 = FastShape<OP>::kOut2 > 0 && (BWD ? gout2 != nullptr : out2 != nullptr);
  auto go = [&](auto kernel) {
    kernel<<<grid, block, 0, st>>>(static_cast<const T*>(x), n, T(eps), eps, static_cast<T*>(out), static_cast<T*>(out2),
                                   static_cast<const T*>(gout), static_cast<const T*>(gout2), static_cast<T*>(gx));
  };
  if constexpr (FastShape<OP>::kOut2 > 0) {
    if (has2) go(fast_kernel<T, OP, BWD, true>); else go(fast_kernel<T, OP, BWD, false>);
  } else {
    go(fast_kernel<T, OP, BWD, false>);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : static_cast<int>(e);
}

template <typename T, bool BWD>
int fast_dispatch(int op, const void* x, int64_t n, double eps, void* out, void* out2, const void* gout, const void* gout2,
                  void* gx, hipStream_t st) {
  switch (op) {
#define MM_FAST_CASE(OP) case OP: return fast_launch<T, OP, BWD>(x, n, eps, out, out2, gout, gout2, gx, st);
    MM_FAST_CASE(MM_FAST_SYMEIG2)
    MM_FAST_CASE(MM_FAST_SYMEIG3)
    MM_FAST_CASE(MM_FAST_CHOLESKY2)
    MM_FAST_CASE(MM_FAST_INVCHOLESKY2)
    MM_FAST_CASE(MM_FAST_SINGULAR2)
    MM_FAST_CASE(MM_FAST_DET2)
    MM_FAST_CASE(MM_FAST_DET3)
    MM_FAST_CASE(MM_FAST_SYMDET3)
#undef MM_FAST_CASE
    default: return MM_ERR_ARG;
  }
}

}  // namespace mm

using namespace mm;

extern "C" {

int mm_fast_fwd(int op, int dtype, const void* x, int64_t n, double eps, void* out, void* out2, mm_stream_t stream) {
  if (n < 0 || op < 0 || op > MM_FAST_SYMDET3 || (n > 0 && (!x || !out))) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32) return fast_dispatch<float, false>(op, x, n, eps, out, out2, nullptr, nullptr, nullptr, st);
  if (dtype == MM_F64) return fast_dispatch<double, false>(op, x, n, eps, out, out2, nullptr, nullptr, nullptr, st);
  return MM_ERR_ARG;
}

int mm_fast_bwd(int op, int dtype, const void* x, const void* grad_out, const void* grad_out2, int64_t n, double eps,
                void* grad_x, mm_stream_t stream) {
  if (n < 0 || op < 0 || op > MM_FAST_SYMDET3 || (n > 0 && (!x || !grad_out || !grad_x))) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32) return fast_dispatch<float, true>(op, x, n, eps, nullptr, nullptr, grad_out, grad_out2, grad_x, st);
  if (dtype == MM_F64) return fast_dispatch<double, true>(op, x, n, eps, nullptr, nullptr, grad_out, grad_out2, grad_x, st);
  return MM_ERR_ARG;
}

}  // extern "C"
