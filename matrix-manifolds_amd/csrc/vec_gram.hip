// Forward pairwise distances for the inner-product manifolds (Lorentz, sphere)
// with the n x m -> n x n Gram on the gfx950 matrix cores.
//
//   G = (X S) X^T,  S = diag(+1,-1,...,-1) for Lorentz (so G_ij = -<x_i,x_j>_L,
//   lorentz.py:72-77,101-118), S = I for the sphere (sphere.py:68-74).
//
// fp32: v_mfma_f32_32x32x2_f32 — exact f32 products/accumulation (bitwise an
// fmaf chain over k, MI355X guide §3), which matters because acosh is evaluated
// next to 1.  One wavefront owns a 32x32 tile of the pair matrix; lane l feeds
// A[i = l&31][k = 2s + (l>>5)] and B[k][j = l&31] for k-step s, both straight
// from the (L2-resident) point table.  The accumulator layout puts one column j
// on a lane and 16 rows in its registers, so register q of lanes 0-31 is 32
// consecutive entries of output row i — the distance map (acosh^2 / acos^2) is
// applied in registers and stored as 128-B segments of the row-major pair vector.
// fp64: v_mfma_f64_16x16x4_f64, 16x16 tiles (its own C layout: row = (l>>4)+4q).
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include "smallmat.hpp"

namespace mm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t gpair_off(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

template <typename T> __device__ __forceinline__ T gacos(T c);
template <> __device__ __forceinline__ float gacos<float>(float c) { return ::acosf(c); }
template <> __device__ __forceinline__ double gacos<double>(double c) { return ::acos(c); }

template <typename T, int KIND> __device__ __forceinline__ T gram_value(T q, int squared) {
  using N = Num<T>;
  if (KIND == MM_LORENTZ) {
    const T t = N::max(q, T(1));
    const T d = N::max(N::log(t + N::sqrt(N::fma(t, t, T(-1)))), T(1e-8));
    return squared ? d * d : d;
  } else {
    const T c = N::min(N::max(q, T(-1 + 1e-16)), T(1 - 1e-16));
    const T th = N::max(gacos<T>(c), T(1e-8));
    return squared ? th * th : th;
  }
}

constexpr int kGramWaves = 4;

// fp32: 32x32 tile per wavefront
template <int KIND>
__global__ __launch_bounds__(64 * kGramWaves) void vec_gram_fwd_f32_kernel(const float* __restrict__ x, int n, int m,
                                                                         int row_begin, int row_end, int squared,
                                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = row_begin + blockIdx.y * 32;
  const int j0 = (((i0 + 1) / 32) + blockIdx.x * kGramWaves + wave) * 32;
  if (j0 >= n) return;  // wave-uniform
  const int ia = i0 + r, jb = j0 + r;
  const bool ia_ok = ia < n, jb_ok = jb < n;
  const float* xa = x + size_t(ia_ok ? ia : 0) * m;
  const float* xb = x + size_t(jb_ok ? jb : 0) * m;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int ksteps = (m + 1) / 2;
  for (int s = 0; s < ksteps; ++s) {
    const int k = 2 * s + h;
    float a = (ia_ok && k < m) ? xa[k] : 0.f;
    const float b = (jb_ok && k < m) ? xb[k] : 0.f;
    if (KIND == MM_LORENTZ && k != 0) a = -a;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const int64_t base = gpair_off(n, row_begin);
  const int j = j0 + r;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;
    const float v = gram_value<float, KIND>(acc[q], squared);
    if (i < row_end && j < n && j > i) out[gpair_off(n, i) - base + (j - i - 1)] = v;
  }
}

// fp64: 16x16 tile per wavefront
template <int KIND>
__global__ __launch_bounds__(64 * kGramWaves) void vec_gram_fwd_f64_kernel(const double* __restrict__ x, int n, int m,
                                                                         int row_begin, int row_end, int squared,
                                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int i0 = row_begin + blockIdx.y * 16;
  const int j0 = (((i0 + 1) / 16) + blockIdx.x * kGramWaves + wave) * 16;
  if (j0 >= n) return;
  const int ia = i0 + r, jb = j0 + r;
  const bool ia_ok = ia < n, jb_ok = jb < n;
  const double* xa = x + size_t(ia_ok ? ia : 0) * m;
  const double* xb = x + size_t(jb_ok ? jb : 0) * m;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int ksteps = (m + 3) / 4;
  for (int s = 0; s < ksteps; ++s) {
    const int k = 4 * s + h;
    double a = (ia_ok && k < m) ? xa[k] : 0.0;
    const double b = (jb_ok && k < m) ? xb[k] : 0.0;
    if (KIND == MM_LORENTZ && k != 0) a = -a;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  const int64_t base = gpair_off(n, row_begin);
  const int j = j0 + r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = i0 + h + 4 * q;  // f64 C/D map: row = (lane>>4) + 4*reg
    const double v = gram_value<double, KIND>(acc[q], squared);
    if (i < row_end && j < n && j > i) out[gpair_off(n, i) - base + (j - i - 1)] = v;
  }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_vec_pdist_fwd_gram(int dtype, int kind, const void* x, int64_t n, int m, int64_t row_begin,
                                     int64_t row_end, int squared, void* out, mm_stream_t stream) {
  if (!x || n < 0 || m < 1 || row_begin < 0 || row_end > n || row_begin > row_end || n > (1 << 30)) return MM_ERR_ARG;
  if (kind != MM_LORENTZ && kind != MM_SPHERE) return MM_ERR_UNSUPPORTED;
  if (!out && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  if (row_end <= row_begin) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tile = dtype == MM_F32 ? 32 : 16;
  const int ntile_cols = int((n + tile - 1) / tile) - int((row_begin + 1) / tile);
  if (ntile_cols <= 0) return MM_OK;
  const dim3 grid((ntile_cols + kGramWaves - 1) / kGramWaves, int((row_end - row_begin + tile - 1) / tile));
  const dim3 block(64 * kGramWaves);
  {
    ProfScope prof(PROF_VEC_FWD, st);
    if (dtype == MM_F32) {
      auto* xp = static_cast<const float*>(x);
      auto* op = static_cast<float*>(out);
      if (kind == MM_LORENTZ)
        vec_gram_fwd_f32_kernel<MM_LORENTZ><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op);
      else
        vec_gram_fwd_f32_kernel<MM_SPHERE><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op);
    } else if (dtype == MM_F64) {
      auto* xp = static_cast<const double*>(x);
      auto* op = static_cast<double*>(out);
      if (kind == MM_LORENTZ)
        vec_gram_fwd_f64_kernel<MM_LORENTZ><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op);
      else
        vec_gram_fwd_f64_kernel<MM_SPHERE><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op);
    } else {
      return MM_ERR_ARG;
    }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}
