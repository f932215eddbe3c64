// fp32 SPD(d) pair kernels on packed math: every lane carries TWO pairs.
//
// Measured on MI355X (profiles/r01_v1_pmc_summary.txt): the v1 kernels are VALU-issue
// bound — a wave64 fp32 instruction occupies the SIMD for 4 cycles (8 for
// rcp/rsq/sqrt/log) and the VALU was 87 % busy.  gfx950 reaches its fp32 vector peak
// only through v_pk_{fma,mul,add}_f32, which process two floats per lane in the same
// 4 cycles.  So here a lane owns two columns (j and j+64: all global accesses of a
// wavefront stay 256-B contiguous) of the same row i, all small-matrix arithmetic is
// written on float2 values, and the Jacobi rotation is reformulated to need two
// (unpackable) rsq instead of sqrt+rcp+rsq:
//     r = rsq(h^2 + 4 apq^2),  cos2t = |h| r,  c = sqrt((1+cos2t)/2) = x rsq(x),
//     s = sgn(h) apq r / c,    t = s / c.
// Everything else (tiling, SGPR-broadcast row operand, recomputation in backward,
// SoA accumulators + per-tile coalesced atomics, finalize) is as in spd.hip, whose
// workspace layout and finalize kernel are shared.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include "smallmat.hpp"

namespace mm {
namespace pk {

typedef float v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2 splat(float s) { return v2{s, s}; }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 rsq2(v2 x) { return v2{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
__device__ __forceinline__ v2 rcp2(v2 x) { return v2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ v2 log2v(v2 x) { return v2{::logf(x.x), ::logf(x.y)}; }
__device__ __forceinline__ v2 clamp2(v2 x, float lo, float hi) {
  return v2{fminf(fmaxf(x.x, lo), hi), fminf(fmaxf(x.y, lo), hi)};
}
__device__ __forceinline__ v2 sign1(v2 h) { return v2{copysignf(1.f, h.x), copysignf(1.f, h.y)}; }

constexpr int kCols = 128;   // columns per wavefront (2 per lane)
constexpr int kWaves = 2;    // wavefronts per workgroup
constexpr int kWgCols = kCols * kWaves;

__host__ __device__ inline int64_t poff(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

// A = Li X Li^T, Li wave-uniform scalars, X packed symmetric float2
template <int D>
__device__ __forceinline__ void congr_lower2(const float (&lw)[Packed<D>::NP], const v2 (&s)[Packed<D>::NP],
                                             v2 (&out)[Packed<D>::NP]) {
  v2 b[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      v2 acc = splat(lw[pidx(r, 0)]) * s[pidx(0, c)];
#pragma unroll
      for (int k = 1; k <= r; ++k) acc = fma2(splat(lw[pidx(r, k)]), s[pidx(k, c)], acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      v2 acc = b[r][0] * splat(lw[pidx(c, 0)]);
#pragma unroll
      for (int k = 1; k <= c; ++k) acc = fma2(b[r][k], splat(lw[pidx(c, k)]), acc);
      out[pidx(r, c)] = acc;
    }
}

// out = Lw^T S Lw
template <int D>
__device__ __forceinline__ void congr_lower_t2(const float (&lw)[Packed<D>::NP], const v2 (&s)[Packed<D>::NP],
                                               v2 (&out)[Packed<D>::NP]) {
  v2 b[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      v2 acc = s[pidx(r, c)] * splat(lw[pidx(c, c)]);
#pragma unroll
      for (int k = c + 1; k < D; ++k) acc = fma2(s[pidx(r, k)], splat(lw[pidx(k, c)]), acc);
      b[r][c] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      v2 acc = splat(lw[pidx(r, r)]) * b[r][c];
#pragma unroll
      for (int k = r + 1; k < D; ++k) acc = fma2(splat(lw[pidx(k, r)]), b[k][c], acc);
      out[pidx(r, c)] = acc;
    }
}

// Cyclic Jacobi on two matrices per lane.  The sweep loop is wave-uniform; unlike
// smallmat.hpp no per-lane masking is applied (a converged lane just performs
// rotations by ~eps angles), which is still reproducible under row sharding because
// the pairs that share a wavefront are fixed by the global column index alone.
// TOL2: stop when off^2 <= TOL2 * diag^2  (eps^2 for eigenvectors, eps for
// eigenvalues only — the value of sum log^2 is second-order in the residual).
template <int D, bool WITH_V>
__device__ __forceinline__ void jacobi2(v2 (&a)[Packed<D>::NP], v2 (&v)[D][D], float tol2) {
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) v[r][c] = splat(r == c ? 1.f : 0.f);
  }
  for (int sweep = 0; sweep < 8; ++sweep) {
    v2 off2 = splat(0.f), dg2 = splat(0.f);
#pragma unroll
    for (int r = 0; r < D; ++r) {
      dg2 = fma2(a[pidx(r, r)], a[pidx(r, r)], dg2);
#pragma unroll
      for (int c = 0; c < r; ++c) off2 = fma2(a[pidx(r, c)], a[pidx(r, c)], off2);
    }
    const v2 lim = dg2 * splat(tol2);
    if (!__any((off2.x > lim.x) || (off2.y > lim.y))) break;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        const v2 apq = a[pidx(q, p)];
        const v2 h = a[pidx(q, q)] - a[pidx(p, p)];
        const v2 sg = sign1(h);
        const v2 ah = fma2(sg, h, splat(1e-15f));   // |h| (+tiny, squares to a NORMAL float: h = apq = 0 gives the identity)
        const v2 sa = sg * apq;
        const v2 sa2 = sa + sa;
        const v2 r = rsq2(fma2(ah, ah, sa2 * sa2));
        const v2 x = fma2(ah * r, splat(0.5f), splat(0.5f));   // cos^2(theta)
        const v2 ci = rsq2(x);
        const v2 c = x * ci;
        const v2 s = (sa * r) * ci;
        const v2 t = s * ci;
        a[pidx(p, p)] = fma2(-t, apq, a[pidx(p, p)]);
        a[pidx(q, q)] = fma2(t, apq, a[pidx(q, q)]);
        a[pidx(q, p)] = splat(0.f);
#pragma unroll
        for (int rr = 0; rr < D; ++rr) {
          if (rr == p || rr == q) continue;
          const v2 arp = a[pidx(rr, p)], arq = a[pidx(rr, q)];
          a[pidx(rr, p)] = fma2(c, arp, -(s * arq));
          a[pidx(rr, q)] = fma2(s, arp, c * arq);
        }
        if (WITH_V) {
#pragma unroll
          for (int rr = 0; rr < D; ++rr) {
            const v2 vrp = v[rr][p], vrq = v[rr][q];
            v[rr][p] = fma2(c, vrp, -(s * vrq));
            v[rr][q] = fma2(s, vrp, c * vrq);
          }
        }
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void vdvt2(const v2 (&v)[D][D], const v2 (&f)[D], v2 (&out)[Packed<D>::NP]) {
  v2 vf[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) vf[r][k] = v[r][k] * f[k];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      v2 acc = vf[r][0] * v[c][0];
#pragma unroll
      for (int k = 1; k < D; ++k) acc = fma2(vf[r][k], v[c][k], acc);
      out[pidx(r, c)] = acc;
    }
}

template <int D>
__device__ __forceinline__ void load_cols(const float* __restrict__ nodeX, int n, int ja, int jb,
                                          v2 (&xj)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  const bool ain = ja < n, bin = jb < n;
#pragma unroll
  for (int k = 0; k < NP; ++k) xj[k] = splat(0.f);
#pragma unroll
  for (int r = 0; r < D; ++r) xj[pidx(r, r)] = splat(1.f);  // out-of-range columns compute on the identity
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (ain) xj[k].x = nodeX[size_t(ja) * NP + k];
    if (bin) xj[k].y = nodeX[size_t(jb) * NP + k];
  }
}

constexpr float kEpsF = 5.9604645e-8f;

// ------------------------------------------------------------------ forward
template <int D, int TI>
__global__ __launch_bounds__(64 * kWaves) void spd_pdist_fwd_pk_kernel(const float* __restrict__ nodeL,
                                                                     const float* __restrict__ nodeX, int n,
                                                                     int row_begin, int row_end, int squared,
                                                                     float wmin, float wmax, float* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  const int i0 = (row_begin / TI + blockIdx.y) * TI;  // tiles anchored at multiples of TI (shard-invariant)
  const int i_lo = max(i0, row_begin), i_hi = min(i0 + TI, row_end);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jw = ((i0 + 1) / kWgCols + blockIdx.x) * kWgCols + wave * kCols;
  if (jw >= n || jw + kCols - 1 <= i_lo) return;  // wave-uniform: out of range / below the diagonal
  const int ja = jw + lane, jb = ja + 64;
  v2 xj[NP];
  load_cols<D>(nodeX, n, ja, jb, xj);
  const int64_t base = poff(n, row_begin);
  for (int i = i_lo; i < i_hi; ++i) {
    float li[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) li[k] = nodeL[size_t(i) * NP + k];
    v2 a[NP], v[D][D];
    congr_lower2<D>(li, xj, a);
    jacobi2<D, false>(a, v, kEpsF);
    v2 s = splat(0.f);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const v2 lw = log2v(clamp2(a[pidx(k, k)], wmin, wmax));
      s = fma2(lw, lw, s);
    }
    s = v2{fmaxf(s.x, wmin), fmaxf(s.y, wmin)};
    if (!squared) s = v2{__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
    float* o = out + (poff(n, i) - base - i - 1);
    if (ja < n && ja > i) o[ja] = s.x;
    if (jb < n && jb > i) o[jb] = s.y;
  }
}

// ------------------------------------------------------------------ backward
template <int D, int TI>
__global__ __launch_bounds__(64 * kWaves) void spd_pdist_bwd_pk_kernel(const float* __restrict__ nodeL,
                                                                     const float* __restrict__ nodeX,
                                                                     const float* __restrict__ g, int n,
                                                                     int row_begin, int row_end, int squared,
                                                                     float wmin, float wmax, float* __restrict__ accM,
                                                                     float* __restrict__ accN) {
  constexpr int NP = Packed<D>::NP;
  __shared__ float redM[kWaves][TI][NP];
  const int i0 = (row_begin / TI + blockIdx.y) * TI;
  const int i_lo = max(i0, row_begin), i_hi = min(i0 + TI, row_end);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jw = ((i0 + 1) / kWgCols + blockIdx.x) * kWgCols + wave * kCols;
  if (((i0 + 1) / kWgCols + blockIdx.x) * kWgCols >= n) return;  // block-uniform
  const bool live = jw < n && jw + kCols - 1 > i_lo;
  const int ja = jw + lane, jb = ja + 64;
  v2 xj[NP], accJ[NP];
  load_cols<D>(nodeX, n, live ? ja : n, live ? jb : n, xj);
#pragma unroll
  for (int k = 0; k < NP; ++k) accJ[k] = splat(0.f);
  const int64_t base = poff(n, row_begin);
  if (live) {
    for (int i = i_lo; i < i_hi; ++i) {
      float li[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) li[k] = nodeL[size_t(i) * NP + k];
      const float* gp = g + (poff(n, i) - base - i - 1);
      v2 gs;
      gs.x = (ja < n && ja > i) ? gp[ja] : 0.f;
      gs.y = (jb < n && jb > i) ? gp[jb] : 0.f;
      v2 a[NP], v[D][D];
      congr_lower2<D>(li, xj, a);
      jacobi2<D, true>(a, v, kEpsF * kEpsF);
      v2 w[D], lw[D], s = splat(0.f);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        w[k] = clamp2(a[pidx(k, k)], wmin, wmax);
        lw[k] = log2v(w[k]);
        s = fma2(lw[k], lw[k], s);
      }
      if (!squared) gs = gs * splat(0.5f) * rsq2(v2{fmaxf(s.x, wmin), fmaxf(s.y, wmin)});
      v2 cm[D], cn[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {
        cm[k] = (gs + gs) * lw[k];
        cn[k] = cm[k] * rcp2(w[k]);
      }
      v2 m[NP], nn[NP], cj[NP];
      vdvt2<D>(v, cm, m);
      vdvt2<D>(v, cn, nn);
      congr_lower_t2<D>(li, nn, cj);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        accJ[k] += cj[k];
        const float r = wave_sum(m[k].x + m[k].y);
        if (lane == 0) redM[wave][i - i0][k] = r;
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < TI * NP; t += 64 * kWaves) {
    const int k = t / TI, il = t % TI, i = i0 + il;
    if (i >= i_lo && i < i_hi) {
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < kWaves; ++wv) {
        const int jwv = ((i0 + 1) / kWgCols + blockIdx.x) * kWgCols + wv * kCols;
        if (jwv < n && jwv + kCols - 1 > i_lo) sum += redM[wv][il][k];
      }
      atomic_add(&accM[size_t(k) * n + i], sum);
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      if (ja < n) atomic_add(&accN[size_t(k) * n + ja], accJ[k].x);
      if (jb < n) atomic_add(&accN[size_t(k) * n + jb], accJ[k].y);
    }
  }
}

template <int D> struct TilePk { static constexpr int TI = 16; };

}  // namespace pk

// Launchers used by spd.hip's dispatch for dtype == MM_F32.
template <int D>
int spd_pk_launch_fwd(const float* nodeL, const float* nodeX, int64_t n, int64_t rb, int64_t re, int squared,
                      double wmin, double wmax, float* out, hipStream_t st) {
  constexpr int TI = pk::TilePk<D>::TI;
  const int gx = int((n + pk::kWgCols - 1) / pk::kWgCols) - int((rb / TI * TI + 1) / pk::kWgCols);
  const int gy = int((re - 1) / TI - rb / TI + 1);
  if (gx <= 0 || gy <= 0) return MM_OK;
  {
    ProfScope prof(PROF_SPD_FWD, st);
    pk::spd_pdist_fwd_pk_kernel<D, TI><<<dim3(gx, gy), dim3(64 * pk::kWaves), 0, st>>>(
        nodeL, nodeX, int(n), int(rb), int(re), squared, float(wmin), float(wmax), out);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

template <int D>
int spd_pk_launch_bwd(const float* nodeL, const float* nodeX, const float* g, int64_t n, int64_t rb, int64_t re,
                      int squared, double wmin, double wmax, float* accM, float* accN, hipStream_t st) {
  constexpr int TI = pk::TilePk<D>::TI;
  const int gx = int((n + pk::kWgCols - 1) / pk::kWgCols) - int((rb / TI * TI + 1) / pk::kWgCols);
  const int gy = int((re - 1) / TI - rb / TI + 1);
  if (gx <= 0 || gy <= 0) return MM_OK;
  {
    ProfScope prof(PROF_SPD_BWD, st);
    pk::spd_pdist_bwd_pk_kernel<D, TI><<<dim3(gx, gy), dim3(64 * pk::kWaves), 0, st>>>(
        nodeL, nodeX, g, int(n), int(rb), int(re), squared, float(wmin), float(wmax), accM, accN);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

#define MM_PK_INST(D)                                                                                            \
  template int spd_pk_launch_fwd<D>(const float*, const float*, int64_t, int64_t, int64_t, int, double, double,  \
                                    float*, hipStream_t);                                                        \
  template int spd_pk_launch_bwd<D>(const float*, const float*, const float*, int64_t, int64_t, int64_t, int,    \
                                    double, double, float*, float*, hipStream_t);
MM_PK_INST(2)
MM_PK_INST(3)
MM_PK_INST(4)
MM_PK_INST(5)

}  // namespace mm
