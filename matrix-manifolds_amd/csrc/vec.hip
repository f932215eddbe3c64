// Vector manifolds — Euclidean R^m, Lorentz (hyperboloid) H^{m-1}, sphere S^{m-1}:
// pairwise distances, their gradients, and the per-point maps RiemannianSGD calls.
//
// Reference arithmetic: graphembed/graphembed/manifolds/{base,euclidean,lorentz,
// sphere}.py (entry-by-entry citations in include/mm_manifolds.h).
//
// Pair kernels (this file, VALU form): lanes own consecutive columns j and keep
// x_j in VGPRs; the row operand x_i is wave-uniform and arrives through scalar
// loads (SGPR broadcast — no LDS).  Forward writes the row-major pair vector with
// lanes on consecutive j (256-B segments).  Backward visits every ORDERED pair
// (i,j), so each lane only ever accumulates into its own column's gradient in
// registers: no cross-lane reduction at all, one coalesced atomic flush per tile
// into structure-of-arrays accumulators, then a per-point finalize.
// The MFMA Gram form of the forward lives in vec_gram.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include "smallmat.hpp"
#include "loss.hpp"
#include "vecfn.hpp"
#include "vec_step.hpp"
#include "vec_rules.hpp"
#include "spd_ws.hpp"
#include "adam.hpp"

namespace mm {

constexpr int kVBlock = 256;
constexpr int kVecMaxDim = 64;
constexpr int kVecSubMaxRows = 64;   // most rows per workgroup of the node-minibatch launch (their points sit in LDS)
__host__ __device__ inline int64_t vpair_off(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

// q for one pair from register/scalar operands
template <typename T, int KIND, int MP, typename TI>
__device__ __forceinline__ T pair_q(const TI (&xi)[MP], const T (&xj)[MP]) {
  using N = Num<T>;
  T q = T(0);
  if (KIND == MM_EUCLIDEAN) {
#pragma unroll
    for (int k = 0; k < MP; ++k) { const T df = xj[k] - xi[k]; q = N::fma(df, df, q); }
  } else if (KIND == MM_LORENTZ) {
#pragma unroll
    for (int k = 1; k < MP; ++k) q = N::fma(xi[k], xj[k], q);
    q = N::fma(xi[0], xj[0], -q);  // -( -x0 y0 + sum x_k y_k )
  } else {
#pragma unroll
    for (int k = 0; k < MP; ++k) q = N::fma(xi[k], xj[k], q);
  }
  return q;
}

template <typename T, int MP>
__device__ __forceinline__ void load_point(const T* __restrict__ x, int64_t idx, int m, T (&r)[MP]) {
#pragma unroll
  for (int k = 0; k < MP; ++k) r[k] = (k < m) ? x[idx * m + k] : T(0);
}

// ------------------------------------------------------------------ forward
// (rows per tile: a launch argument — 32 for a launch that fills the device, fewer for small ones: tree40's 40 rows were ONE
// tile of 32 and one of 8, 11 us of a single wavefront per SIMD walking 32 rows; round 5)
template <typename T, int KIND, int MP>
__global__ __launch_bounds__(kVBlock) void vec_pdist_fwd_kernel(const T* __restrict__ x, int n, int m, int row_begin,
                                                                int row_end, int squared, T* __restrict__ out, int ti) {
  const int i0 = row_begin + blockIdx.y * ti;
  const int i1 = min(i0 + ti, row_end);
  const int jbase = ((i0 + 1) / kVBlock + blockIdx.x) * kVBlock;
  if (jbase >= n) return;
  if (jbase + (int(threadIdx.x) & ~63) + 63 <= i0) return;
  const int j = jbase + threadIdx.x;
  const bool jin = j < n;
  T xj[MP];
  load_point<T, MP>(x, jin ? j : 0, m, xj);
  const int64_t base = vpair_off(n, row_begin);
  for (int i = i0; i < i1; ++i) {
    T xi[MP];
    load_point<T, MP>(x, i, m, xi);  // wave-uniform -> scalar loads
    const T v = PairFn<T, KIND>::value(pair_q<T, KIND, MP>(xi, xj), squared);
    if (jin && j > i) out[vpair_off(n, i) - base + (j - i - 1)] = v;
  }
}

// ------------------------------------------------------------------ backward
// acc[k][j] += sum_i g_ij * dq_ij * (coefficient vector of x_i); finalize applies
// the manifold-specific linear map (2(x_j - .) / -J / identity).
// LOSS != 0 (mm_vec_pdist_loss): `g` holds the TARGET squared distances; the upstream gradient of a
// pair is derived in registers from the loss (loss.hpp) and the loss / scale-gradient sums (counted
// once per unordered pair, in its i < j visit) leave through la.slots.
// SUB: a NODE MINIBATCH (train.py:198-222; modules.py:86 gathers x[idx] first): the n points of the launch are the nodes
// idx[0..n) of an embedding of n_total points — points are read from rows idx[.] of the full table, the target of pair
// (a, b) is dense[idx[a]][idx[b]] (`g` = the dense n_total x n_total matrix, data/dataset.py:19-27) and the sums go to the
// accumulator slots of the NODES ([MP+1][n_total]), from which the finalize kernel over the full embedding writes the
// dense gradient (zero rows outside the batch): no gather and no scatter-add around the kernel, any m the library has.
template <typename T, int KIND, int MP, int TI, int LOSS, bool SUB = false>
__global__ __launch_bounds__(kVBlock) void vec_pdist_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, int n,
                                                                int m, int row_begin, int row_end, int squared,
                                                                T* __restrict__ acc /* [MP+1][n] */, LossArgs<T> la,
                                                                const int64_t* __restrict__ idx = nullptr, int n_total = 0,
                                                                int rows_per_block = TI /* SUB: chosen per launch */) {
  const int j = blockIdx.x * kVBlock + threadIdx.x;
  const int rpb = SUB ? rows_per_block : TI;
  const int i0 = blockIdx.y * rpb, i1 = min(i0 + rpb, n);
  const bool jin = j < n;
  const bool jown = jin && j >= row_begin && j < row_end;  // pairs (j, i>j) belong to this shard
  const int ns = SUB ? n_total : n;                        // stride of the per-node accumulators
  // (node ids of a minibatch are clamped into the table: unvalidated caller data, see spd_pair.hpp batch_node)
  const int jn = SUB ? int(min(uint64_t(idx[jin ? j : 0]), uint64_t(n_total - 1))) : (jin ? j : 0);   // the column's node
  T xj[MP], a[MP];
  load_point<T, MP>(x, jn, m, xj);
#pragma unroll
  for (int k = 0; k < MP; ++k) a[k] = T(0);
  T wsum = T(0), sp = T(1), loss_acc = T(0), ds_acc = T(0);
  loss_resolve<T, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  const int64_t base = vpair_off(n, row_begin);
  // SUB: the tile's rows are staged in LDS — their node ids with one request, all their points with one more.  (Read in
  // the loop below, `x[idx[i] * m + k]` is two DEPENDENT scalar round trips per row: 47 us for the 130 816 pairs of a
  // 512-node Lorentz(24) batch; round 4.)
  constexpr int kSubRows = SUB ? kVecSubMaxRows : 1;
  __shared__ T rowx[kSubRows][SUB ? MP : 1];
  __shared__ int rown[kSubRows];
  if constexpr (SUB) {
    if (int(threadIdx.x) < rpb) rown[threadIdx.x] = int(min(uint64_t(idx[min(i0 + int(threadIdx.x), n - 1)]), uint64_t(n_total - 1)));
    __syncthreads();
    for (int e = threadIdx.x; e < rpb * MP; e += kVBlock) {
      const int r = e / MP, k = e % MP;
      rowx[r][k] = k < m ? x[size_t(rown[r]) * m + k] : T(0);
    }
    __syncthreads();
  }
  // The per-pair arithmetic (~45 VALU ops) is far too short to hide the latency of the load
  // of g it depends on, so the upstream gradients of UNR rows are fetched as one batch first.
  constexpr int UNR = 8;
  for (int ib = i0; ib < i1; ib += UNR) {
    T wv[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = ib + u;
      const bool up = i < j;  // pair (i,j) stored under row i, else under row j
      const bool valid = jin && i < i1 && (up ? (i >= row_begin && i < row_end) : (jown && i > j));
      const int lo = up ? i : j, hi = up ? j : i;
      ok[u] = valid;
      if constexpr (SUB) {   // dense[node(lo)][node(hi)]: the row's node is wave-uniform (a scalar load), the column's is in a register
        const size_t in_ = size_t(rown[min(i, i1 - 1) - i0]);
        wv[u] = valid ? (up ? g[in_ * size_t(n_total) + size_t(jn)] : g[size_t(jn) * size_t(n_total) + in_]) : T(0);
      } else {
        wv[u] = valid ? g[vpair_off(n, lo) - base + (hi - lo - 1)] : T(0);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = min(ib + u, i1 - 1);  // (rows past the tile carry wv = 0)
      T xi[MP];
      if constexpr (SUB) {
#pragma unroll
        for (int k = 0; k < MP; ++k) xi[k] = rowx[i - i0][k];   // (same address in every lane: LDS broadcast)
      } else {
        load_point<T, MP>(x, i, m, xi);
      }
      const T q = pair_q<T, KIND, MP>(xi, xj);
      T w;
      if constexpr (LOSS == MM_LOSS_NONE) {
        w = wv[u] * PairFn<T, KIND>::dq(q, squared);
      } else {
        const T d2 = PairFn<T, KIND>::value(q, 1);
        T dldm;
        const T l = loss_term<T, LOSS>(sp * d2, wv[u], la, dldm);
        const bool once = ok[u] && ib + u < j;
        loss_acc += once ? l : T(0);
        ds_acc += once ? dldm * d2 : T(0);
        w = ok[u] ? dldm * sp * PairFn<T, KIND>::dq(q, 1) : T(0);
      }
      wsum += w;
#pragma unroll
      for (int k = 0; k < MP; ++k) a[k] = Num<T>::fma(w, xi[k], a[k]);
    }
  }
  if (jin) {
#pragma unroll
    for (int k = 0; k < MP; ++k)
      if (k < m) atomic_add(&acc[size_t(k) * ns + jn], a[k]);
    if (KIND == MM_EUCLIDEAN) atomic_add(&acc[size_t(MP) * ns + jn], wsum);
  }
  if constexpr (LOSS != MM_LOSS_NONE) {
    __shared__ T lossW[kVBlock / 64][2];
    const T l = wave_sum(loss_acc), d = wave_sum(ds_acc);
    if ((threadIdx.x & 63) == 0) { lossW[threadIdx.x >> 6][0] = l; lossW[threadIdx.x >> 6][1] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
      T ls = T(0), dsum = T(0);
#pragma unroll
      for (int wv = 0; wv < kVBlock / 64; ++wv) { ls += lossW[wv][0]; dsum += lossW[wv][1]; }
      const int slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kLossSlots - 1);
      atomic_add(&la.slots[slot], ls);
      atomic_add(&la.slots[kLossSlots + slot], dsum);
    }
  }
}

template <typename T, int KIND, int MP>
__global__ void vec_pdist_finalize_kernel(const T* __restrict__ x, const T* __restrict__ acc, int n, int m,
                                          T* __restrict__ grad, T* __restrict__ slots,
                                          const T* __restrict__ scale_raw, T* __restrict__ loss_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (slots && blockIdx.x == 0 && threadIdx.x < 64) loss_finalize<T>(slots, scale_raw, loss_out);
  if (j >= n) return;
  for (int k = 0; k < m; ++k) {
    const T s = acc[size_t(k) * n + j];
    T r;
    if (KIND == MM_EUCLIDEAN) r = T(2) * (acc[size_t(MP) * n + j] * x[size_t(j) * m + k] - s);  // sum w 2(x_j - x_i)
    else if (KIND == MM_LORENTZ) r = (k == 0) ? s : -s;                                           // dq/dx_j = -J x_i
    else r = s;
    grad[size_t(j) * m + k] = r;
  }
}

// ------------------------------------------------------- element-wise dist
template <typename T, int KIND>
__global__ void vec_dist_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ g, int64_t cnt,
                                int m, int squared, T* __restrict__ out, T* __restrict__ gx, T* __restrict__ gy) {
  using N = Num<T>;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= cnt) return;
  const T* xp = x + p * m;
  const T* yp = y + p * m;
  T q = T(0);
  if (KIND == MM_EUCLIDEAN) {
    for (int k = 0; k < m; ++k) { const T df = yp[k] - xp[k]; q = N::fma(df, df, q); }
  } else if (KIND == MM_LORENTZ) {
    for (int k = 1; k < m; ++k) q = N::fma(xp[k], yp[k], q);
    q = N::fma(xp[0], yp[0], -q);
  } else {
    for (int k = 0; k < m; ++k) q = N::fma(xp[k], yp[k], q);
  }
  if (out) out[p] = PairFn<T, KIND>::value(q, squared);
  if (gx) {
    const T w = g[p] * PairFn<T, KIND>::dq(q, squared);
    for (int k = 0; k < m; ++k) {
      T ax, ay;
      if (KIND == MM_EUCLIDEAN) { ax = T(2) * (xp[k] - yp[k]); ay = -ax; }
      else if (KIND == MM_LORENTZ) { ax = (k == 0) ? yp[k] : -yp[k]; ay = (k == 0) ? xp[k] : -xp[k]; }
      else { ax = yp[k]; ay = xp[k]; }
      gx[p * m + k] = w * ax;
      gy[p * m + k] = w * ay;
    }
  }
}

// ------------------------------------------------------ per-point maps
template <typename T> __device__ __forceinline__ T ldot(const T* u, const T* v, int m) {
  T s = T(0);
  for (int k = 1; k < m; ++k) s = Num<T>::fma(u[k], v[k], s);
  return Num<T>::fma(-u[0], v[0], s);
}
template <typename T> __device__ __forceinline__ T edot(const T* u, const T* v, int m) {
  T s = T(0);
  for (int k = 0; k < m; ++k) s = Num<T>::fma(u[k], v[k], s);
  return s;
}

// One thread per point; `loc` = local scratch of the point's tangent (<= kVecMaxDim).
template <typename T, int KIND>
__device__ __forceinline__ void vec_exp_or_retr(const T* xp, const T* u, int m, int exact, T* o) {
  using N = Num<T>;
  if (KIND == MM_EUCLIDEAN) {
    for (int k = 0; k < m; ++k) o[k] = xp[k] + u[k];
  } else if (KIND == MM_LORENTZ) {  // lorentz.py:59-62 (retr == exp, base.py:49-50)
    const T un = N::max(N::sqrt(N::max(ldot(u, u, m), T(0))), T(kEps));
    const T ch = ::cosh(un), sh = ::sinh(un) / un;
    for (int k = 0; k < m; ++k) o[k] = N::fma(xp[k], ch, sh * u[k]);
  } else {  // sphere.py:51-59
    const T nu = N::sqrt(N::max(edot(u, u, m), T(kEps)));
    if (exact && nu > T(kEps)) {
      const T c = ::cos(nu), s = ::sin(nu) / nu;
      for (int k = 0; k < m; ++k) o[k] = N::fma(xp[k], c, s * u[k]);
    } else {
      T nn = T(0);
      for (int k = 0; k < m; ++k) { o[k] = xp[k] + u[k]; nn = N::fma(o[k], o[k], nn); }
      const T inv = T(1) / N::sqrt(N::max(nn, T(kEps)));
      for (int k = 0; k < m; ++k) o[k] *= inv;
    }
  }
}

template <typename T, int KIND>
__device__ __forceinline__ void vec_egrad2rgrad(const T* xp, const T* gp, int m, T* o) {
  using N = Num<T>;
  if (KIND == MM_EUCLIDEAN) {
    for (int k = 0; k < m; ++k) o[k] = gp[k];
  } else if (KIND == MM_LORENTZ) {  // lorentz.py:52-57: flip time coordinate, then u + <x,u>_L x
    for (int k = 0; k < m; ++k) o[k] = (k == 0) ? -gp[k] : gp[k];
    const T d = ldot(xp, o, m);
    for (int k = 0; k < m; ++k) o[k] = N::fma(d, xp[k], o[k]);
  } else {  // sphere.py:41-44
    const T d = edot(xp, gp, m);
    for (int k = 0; k < m; ++k) o[k] = N::fma(-d, xp[k], gp[k]);
  }
}

template <typename T, int KIND> __device__ __forceinline__ T vec_norm(const T* u, int m) {
  const T s = (KIND == MM_LORENTZ) ? ldot(u, u, m) : edot(u, u, m);
  return Num<T>::sqrt(Num<T>::max(s, T(kEps)));  // base.py:29-33
}

template <typename T, int KIND>
__global__ void vec_map_kernel(int op, const T* __restrict__ x, const T* __restrict__ u, const T* __restrict__ y,
                               int64_t cnt, int m, T* __restrict__ out) {
  using N = Num<T>;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= cnt) return;
  const T* xp = x + p * m;
  const T* up = u ? u + p * m : nullptr;
  const T* yp = y ? y + p * m : nullptr;
  T* o = out + p * m;
  T tmp[kVecMaxDim];
  switch (op) {
    case MM_VEC_EGRAD2RGRAD: vec_egrad2rgrad<T, KIND>(xp, up, m, tmp); for (int k = 0; k < m; ++k) o[k] = tmp[k]; break;
    case MM_VEC_PROJU:
      if (KIND == MM_EUCLIDEAN) { for (int k = 0; k < m; ++k) o[k] = up[k]; }
      else if (KIND == MM_LORENTZ) { const T d = ldot(xp, up, m); for (int k = 0; k < m; ++k) o[k] = N::fma(d, xp[k], up[k]); }
      else { const T d = edot(xp, up, m); for (int k = 0; k < m; ++k) o[k] = N::fma(-d, xp[k], up[k]); }
      break;
    case MM_VEC_EXP: vec_exp_or_retr<T, KIND>(xp, up, m, 1, tmp); for (int k = 0; k < m; ++k) o[k] = tmp[k]; break;
    case MM_VEC_RETR: vec_exp_or_retr<T, KIND>(xp, up, m, 0, tmp); for (int k = 0; k < m; ++k) o[k] = tmp[k]; break;
    case MM_VEC_PROJX:
      if (KIND == MM_EUCLIDEAN) { for (int k = 0; k < m; ++k) o[k] = xp[k]; }
      else if (KIND == MM_LORENTZ) {  // lorentz.py:44-50
        T s = T(1);
        for (int k = 1; k < m; ++k) s = N::fma(xp[k], xp[k], s);
        o[0] = N::sqrt(s);
        for (int k = 1; k < m; ++k) o[k] = xp[k];
      } else {
        const T inv = T(1) / vec_norm<T, KIND>(xp, m);
        for (int k = 0; k < m; ++k) o[k] = xp[k] * inv;
      }
      break;
    case MM_VEC_TRANSP:  // transport u from x to y
      if (KIND == MM_EUCLIDEAN) { for (int k = 0; k < m; ++k) o[k] = up[k]; }
      else if (KIND == MM_LORENTZ) {  // lorentz.py:79-82
        const T xy = ldot(xp, yp, m), uy = ldot(up, yp, m);
        const T f = uy / (T(1) - xy);
        for (int k = 0; k < m; ++k) o[k] = N::fma(f, xp[k] + yp[k], up[k]);
      } else {  // base.py:65-66: proju(y, u)
        const T d = edot(yp, up, m);
        for (int k = 0; k < m; ++k) o[k] = N::fma(-d, yp[k], up[k]);
      }
      break;
    case MM_VEC_LOG:  // log_x(y), y passed in `u`
      if (KIND == MM_EUCLIDEAN) { for (int k = 0; k < m; ++k) o[k] = up[k] - xp[k]; }
      else if (KIND == MM_LORENTZ) {  // lorentz.py:64-70
        const T xy = N::min(ldot(xp, up, m), T(-1));
        const T den = N::max(N::sqrt(N::fma(xy, xy, T(-1))), T(kEps));
        const T num = N::max(N::log(-xy + N::sqrt(N::fma(xy, xy, T(-1)))), T(kEps));
        const T f = num / den;
        for (int k = 0; k < m; ++k) tmp[k] = f * N::fma(xy, xp[k], up[k]);
        const T d = ldot(xp, tmp, m);
        for (int k = 0; k < m; ++k) o[k] = N::fma(d, xp[k], tmp[k]);
      } else {  // sphere.py:61-66
        T dmx[kVecMaxDim];
        for (int k = 0; k < m; ++k) dmx[k] = up[k] - xp[k];
        const T d0 = edot(xp, dmx, m);
        for (int k = 0; k < m; ++k) tmp[k] = N::fma(-d0, xp[k], dmx[k]);
        const T c = N::min(N::max(edot(xp, up, m), T(-1 + 1e-16)), T(1 - 1e-16));
        const T th = N::max(acos_t<T>(c), T(kEps));
        const T nu = vec_norm<T, KIND>(tmp, m);
        const T f = (th > T(kEps)) ? th / nu : T(1);
        for (int k = 0; k < m; ++k) o[k] = tmp[k] * f;
      }
      break;
  }
}

template <typename T, int KIND>
__global__ void vec_norm_kernel(const T* __restrict__ u, int64_t cnt, int m, int squared, T* __restrict__ out) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= cnt) return;
  const T nv = vec_norm<T, KIND>(u + p * m, m);
  out[p] = squared ? nv * nv : nv;
}

// fused momentum-free RSGD update (rsgd.py:63-68,82)
// (x and xnew are deliberately not __restrict__: the update may be done in place, xnew == x)
template <typename T, int KIND>
__device__ __forceinline__ void vec_rsgd_point(const T* x, const T* __restrict__ eg, int64_t p, int m, T lr,
                                               T max_grad_norm, int exact, T* xnew) {
  T r[kVecMaxDim], o[kVecMaxDim];
  vec_egrad2rgrad<T, KIND>(x + p * m, eg + p * m, m, r);
  T scale = -lr;
  if (max_grad_norm > T(0)) scale *= Num<T>::min(max_grad_norm / vec_norm<T, KIND>(r, m), T(1));
  for (int k = 0; k < m; ++k) r[k] *= scale;
  vec_exp_or_retr<T, KIND>(x + p * m, r, m, exact, o);
  for (int k = 0; k < m; ++k) xnew[p * m + k] = o[k];
}

// heavy-ball variant (rsgd.py:70-80): buf = momentum buf + (1 - dampening) rgrad; x' = exp/retr(x, -lr buf);
// buf is transported to x' and updated in place.
template <typename T, int KIND>
__device__ __forceinline__ void vec_rsgd_momentum_point(const T* x, const T* __restrict__ eg, T* buf, int64_t p, int m, T lr,
                                                        T momentum, T dampening, T max_grad_norm, int exact, T* xnew) {
  using N = Num<T>;
  T xp[kVecMaxDim], r[kVecMaxDim], b[kVecMaxDim], o[kVecMaxDim];
  for (int k = 0; k < m; ++k) xp[k] = x[p * m + k];
  vec_egrad2rgrad<T, KIND>(xp, eg + p * m, m, r);
  const T clip = max_grad_norm > T(0) ? N::min(max_grad_norm / vec_norm<T, KIND>(r, m), T(1)) : T(1);
  for (int k = 0; k < m; ++k) {
    b[k] = N::fma(momentum, buf[p * m + k], (T(1) - dampening) * (r[k] * clip));
    r[k] = -lr * b[k];
  }
  vec_exp_or_retr<T, KIND>(xp, r, m, exact, o);
  if (KIND == MM_LORENTZ) {  // lorentz.py:79-82
    const T xy = ldot(xp, o, m), uy = ldot(b, o, m);
    const T g = uy / (T(1) - xy);
    for (int k = 0; k < m; ++k) b[k] = N::fma(g, xp[k] + o[k], b[k]);
  } else if (KIND == MM_SPHERE) {  // base.py:65-66: proju(y, u)
    const T d = edot(o, b, m);
    for (int k = 0; k < m; ++k) b[k] = N::fma(-d, o[k], b[k]);
  }
  for (int k = 0; k < m; ++k) {
    xnew[p * m + k] = o[k];
    buf[p * m + k] = b[k];
  }
}
template <typename T, int KIND>
__global__ void vec_rsgd_momentum_kernel(const T* x, const T* __restrict__ eg, T* buf, int64_t cnt, int m, T lr,
                                         T momentum, T dampening, T max_grad_norm, int exact, T* xnew) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= cnt) return;
  vec_rsgd_momentum_point<T, KIND>(x, eg, buf, p, m, lr, momentum, dampening, max_grad_norm, exact, xnew);
}

template <typename T, int KIND>
// (x and xnew are deliberately not __restrict__: the update may be done in place, xnew == x)
__global__ void vec_rsgd_step_kernel(const T* x, const T* __restrict__ eg, int64_t cnt, int m, T lr,
                                     T max_grad_norm, int exact, T* xnew) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= cnt) return;
  vec_rsgd_point<T, KIND>(x, eg, p, m, lr, max_grad_norm, exact, xnew);
}

// fused Riemannian Adam update (radam.py:62-98): Riemannian gradient, its norm (second moment: ONE scalar per
// point, before clipping), clipping, both moments, the bias-corrected step, exp / retr, and the transport of
// the first moment to the new point — one launch instead of ~25.  exp_avg / exp_avg_sq are updated in place.
template <typename T, int KIND>
__device__ __forceinline__ void vec_radam_point(const T* x, const T* __restrict__ eg, T* exp_avg, T* exp_avg_sq,
                                                int64_t p, int m, const AdamArgs<T>& a, T beta2, T alpha, T* xnew) {
  using N = Num<T>;
  T xp[kVecMaxDim], r[kVecMaxDim], mo[kVecMaxDim], o[kVecMaxDim];
  for (int k = 0; k < m; ++k) xp[k] = x[p * m + k];
  vec_egrad2rgrad<T, KIND>(xp, eg + p * m, m, r);
  const T nrm = vec_norm<T, KIND>(r, m);
  const T clip = a.max_grad_norm > T(0) ? N::min(a.max_grad_norm / nrm, T(1)) : T(1);
  const T v = N::fma(beta2, exp_avg_sq[p * m], (T(1) - beta2) * nrm * nrm);
  const T f = -alpha / (N::sqrt(v) + a.eps);
  for (int k = 0; k < m; ++k) {
    mo[k] = N::fma(a.beta1, exp_avg[p * m + k], (T(1) - a.beta1) * (r[k] * clip));
    r[k] = mo[k] * f;  // the step direction
  }
  vec_exp_or_retr<T, KIND>(xp, r, m, a.exact, o);
  // transport of the first moment from x to the new point
  if (KIND == MM_LORENTZ) {  // lorentz.py:79-82
    const T xy = ldot(xp, o, m), uy = ldot(mo, o, m);
    const T g = uy / (T(1) - xy);
    for (int k = 0; k < m; ++k) mo[k] = N::fma(g, xp[k] + o[k], mo[k]);
  } else if (KIND == MM_SPHERE) {  // base.py:65-66: proju(y, u)
    const T d = edot(o, mo, m);
    for (int k = 0; k < m; ++k) mo[k] = N::fma(-d, o[k], mo[k]);
  }
  for (int k = 0; k < m; ++k) {
    xnew[p * m + k] = o[k];
    exp_avg[p * m + k] = mo[k];
    exp_avg_sq[p * m + k] = v;
  }
}

template <typename T, int KIND>
__global__ void vec_radam_step_kernel(const T* x, const T* __restrict__ eg, T* exp_avg, T* exp_avg_sq, int64_t cnt,
                                      int m, AdamArgs<T> a, T* xnew) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  T beta2, alpha;
  adam_coeffs(a, beta2, alpha);
  if (p < cnt) vec_radam_point<T, KIND>(x, eg, exp_avg, exp_avg_sq, p, m, a, beta2, alpha, xnew);
  adam_tick(a.step, a.ticket, gridDim.x);
}

// The same update for several parameters of one optimizer group in ONE launch (blockIdx.y = parameter):
// the points of the vector factors of a product embedding, or its scale parameters — updates that are a
// microsecond of work behind ~3 us of launch each.
template <typename T, int KIND, int RULE>
__device__ __forceinline__ void multi_point_padded(const T* x, const T* eg, T* xnew, int64_t p, int m, const VecRuleArgs<T>& R,
                                                   T beta2, T alpha) {
  T xp[16], g[16], o[16];
  load_padded<T, 16>(x, p, m, xp);
  load_padded<T, 16>(eg, p, m, g);
  pad_rule_point<T, KIND, 16, RULE>(xp, g, p, m, true, R, beta2, alpha, o);
  store_row<T, 16>(xnew, p, m, o);
}
constexpr int kRsgdMultiMax = 8;
template <typename T> struct RsgdMulti {
  const T* x[kRsgdMultiMax];
  const T* eg[kRsgdMultiMax];
  T* xnew[kRsgdMultiMax];
  int64_t cnt[kRsgdMultiMax];
  int m[kRsgdMultiMax], kind[kRsgdMultiMax];
  // hyper-parameters per parameter: the points of a product embedding and its scales sit in different parameter groups
  // (experiments/run_grid.py:29-32) and still leave in one launch
  T lr[kRsgdMultiMax], max_grad_norm[kRsgdMultiMax];
  int exact[kRsgdMultiMax];
};
template <typename T>
__global__ void vec_rsgd_multi_kernel(RsgdMulti<T> a) {
  const int t = blockIdx.y;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= a.cnt[t]) return;
  const int kind = a.kind[t];
  const T lr = a.lr[t], max_grad_norm = a.max_grad_norm[t];
  const int exact = a.exact[t];
  if (a.m[t] <= 16) {   // the register-resident form (vec_rules.hpp; the arithmetic of vec_step.hip's kernels)
    const VecRuleArgs<T> R{lr, T(0), T(0), max_grad_norm, exact, nullptr, nullptr, AdamArgs<T>{}};
    if (kind == MM_EUCLIDEAN) multi_point_padded<T, MM_EUCLIDEAN, VRULE_RSGD>(a.x[t], a.eg[t], a.xnew[t], p, a.m[t], R, T(0), T(0));
    else if (kind == MM_LORENTZ) multi_point_padded<T, MM_LORENTZ, VRULE_RSGD>(a.x[t], a.eg[t], a.xnew[t], p, a.m[t], R, T(0), T(0));
    else multi_point_padded<T, MM_SPHERE, VRULE_RSGD>(a.x[t], a.eg[t], a.xnew[t], p, a.m[t], R, T(0), T(0));
    return;
  }
  if (kind == MM_EUCLIDEAN) vec_rsgd_point<T, MM_EUCLIDEAN>(a.x[t], a.eg[t], p, a.m[t], lr, max_grad_norm, exact, a.xnew[t]);
  else if (kind == MM_LORENTZ) vec_rsgd_point<T, MM_LORENTZ>(a.x[t], a.eg[t], p, a.m[t], lr, max_grad_norm, exact, a.xnew[t]);
  else vec_rsgd_point<T, MM_SPHERE>(a.x[t], a.eg[t], p, a.m[t], lr, max_grad_norm, exact, a.xnew[t]);
}

// ... and the Adam update of several parameters (each with its own moments, step counter, ticket and hyper-parameters)
template <typename T> struct RadamMulti {
  RsgdMulti<T> p;
  T* exp_avg[kRsgdMultiMax];
  T* exp_avg_sq[kRsgdMultiMax];
  double* step[kRsgdMultiMax];
  unsigned* ticket[kRsgdMultiMax];
  T beta1[kRsgdMultiMax], beta2[kRsgdMultiMax], eps[kRsgdMultiMax];
  int nc[kRsgdMultiMax];
};
template <typename T>
__global__ void vec_radam_multi_kernel(RadamMulti<T> s) {
  const int t = blockIdx.y;
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const AdamArgs<T> a{s.p.lr[t], s.beta1[t], s.beta2[t], s.eps[t], s.p.max_grad_norm[t], s.nc[t], s.p.exact[t], s.step[t],
                      s.ticket[t]};
  T beta2, alpha;
  adam_coeffs(a, beta2, alpha);
  if (p < s.p.cnt[t] && s.p.m[t] <= 16) {   // the register-resident form (vec_rules.hpp)
    const int kind = s.p.kind[t], m = s.p.m[t];
    const VecRuleArgs<T> R{a.lr, T(0), T(0), a.max_grad_norm, a.exact, s.exp_avg[t], s.exp_avg_sq[t], a};
    if (kind == MM_EUCLIDEAN) multi_point_padded<T, MM_EUCLIDEAN, VRULE_ADAM>(s.p.x[t], s.p.eg[t], s.p.xnew[t], p, m, R, beta2, alpha);
    else if (kind == MM_LORENTZ) multi_point_padded<T, MM_LORENTZ, VRULE_ADAM>(s.p.x[t], s.p.eg[t], s.p.xnew[t], p, m, R, beta2, alpha);
    else multi_point_padded<T, MM_SPHERE, VRULE_ADAM>(s.p.x[t], s.p.eg[t], s.p.xnew[t], p, m, R, beta2, alpha);
  } else if (p < s.p.cnt[t]) {
    const int kind = s.p.kind[t], m = s.p.m[t];
    if (kind == MM_EUCLIDEAN)
      vec_radam_point<T, MM_EUCLIDEAN>(s.p.x[t], s.p.eg[t], s.exp_avg[t], s.exp_avg_sq[t], p, m, a, beta2, alpha, s.p.xnew[t]);
    else if (kind == MM_LORENTZ)
      vec_radam_point<T, MM_LORENTZ>(s.p.x[t], s.p.eg[t], s.exp_avg[t], s.exp_avg_sq[t], p, m, a, beta2, alpha, s.p.xnew[t]);
    else
      vec_radam_point<T, MM_SPHERE>(s.p.x[t], s.p.eg[t], s.exp_avg[t], s.exp_avg_sq[t], p, m, a, beta2, alpha, s.p.xnew[t]);
  }
  adam_tick(a.step, a.ticket, gridDim.x);
}

// ------------------------------------------------------------------ launchers
#define MMV_CHECK()                                    \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) return static_cast<int>(e_); \
  } while (0)

template <typename T, int KIND, int MP>
int vec_fwd_t(const T* x, int64_t n, int m, int64_t rb, int64_t re, int squared, T* out, hipStream_t st) {
  if (re <= rb) return MM_OK;
  const int nJB = int((n + kVBlock - 1) / kVBlock);
  const int gx = nJB - int((rb + 1) / kVBlock);
  if (gx <= 0) return MM_OK;
  // rows per tile: 32, halved while the launch has fewer than ~4 workgroups per CU (down to 2 rows)
  int ti = 32;
  while (ti > 2 && int64_t(gx) * ((re - rb + ti - 1) / ti) < 4 * int64_t(device_cus())) ti /= 2;
  const int gy = int((re - rb + ti - 1) / ti);
  {
    ProfScope prof(PROF_VEC_FWD, st);
    vec_pdist_fwd_kernel<T, KIND, MP><<<dim3(gx, gy), dim3(kVBlock), 0, st>>>(x, int(n), m, int(rb), int(re), squared, out, ti);
  }
  MMV_CHECK();
  return MM_OK;
}

template <typename T, int KIND, int MP, int LOSS = MM_LOSS_NONE>
int vec_bwd_t(const T* x, const T* g, int64_t n, int m, int64_t rb, int64_t re, int squared, T* grad, void* ws,
              hipStream_t st, LossArgs<T> la = LossArgs<T>{nullptr, T(1), T(0), 0, nullptr}, T* loss_out = nullptr) {
  constexpr int TI = 64;
  T* acc = static_cast<T*>(ws);
  T* slots = acc + size_t(n) * (MP + 1);
  // Symmetric form (vec_sym.hip: every unordered pair once, both endpoints fed, resident balanced grid) where it is
  // instantiated; MM_VEC_BWD_ORDERED=1 forces the ordered-pair kernel below.
  static const bool ordered_env = [] { const char* e = std::getenv("MM_VEC_BWD_ORDERED"); return e && e[0] == '1'; }();
  constexpr int dtype_code = std::is_same<T, float>::value ? MM_F32 : MM_F64;
  bool done = false;
  if (!ordered_env && vec_sym_supports(dtype_code, KIND, m)) {
    bool finalized = false;
    const int rc = vec_sym_backward_pairs(dtype_code, KIND, LOSS, squared, x, g, n, m, rb, re, ws, la.scale_raw, double(la.alpha),
                                          double(la.eps), la.terms, la.dyn, grad, loss_out, &finalized, st);
    if (rc == MM_OK && finalized) return MM_OK;   // (flushed into the gradient by the pair kernel)
    if (rc == MM_OK) done = true;
    else if (rc != MM_ERR_UNSUPPORTED) return rc;
  }
  if (!done) {
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(T) * (size_t(n) * (MP + 1) + 2 * kLossSlots), st);
    if (e != hipSuccess) return int(e);
    la.slots = slots;
    if (re > rb) {
      ProfScope prof(PROF_VEC_BWD, st);
      vec_pdist_bwd_kernel<T, KIND, MP, TI, LOSS>
          <<<dim3(int((n + kVBlock - 1) / kVBlock), int((n + TI - 1) / TI)), dim3(kVBlock), 0, st>>>(
              x, g, int(n), m, int(rb), int(re), squared, acc, la);
    }
    MMV_CHECK();
  }
  vec_pdist_finalize_kernel<T, KIND, MP><<<dim3(int((n + 127) / 128)), dim3(128), 0, st>>>(
      x, acc, int(n), m, grad, LOSS != MM_LOSS_NONE ? slots : static_cast<T*>(nullptr), la.scale_raw, loss_out);
  MMV_CHECK();
  return MM_OK;
}

template <typename T, int KIND, int MP>
int vec_loss_t(int loss_kind, const T* x, const T* target, const T* scale_raw, int64_t n, int m, int64_t rb, int64_t re,
               double alpha, double eps, int terms, const double* loss_params, T* loss_out, T* grad, void* ws, hipStream_t st) {
  LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, nullptr, loss_params};
  if (loss_kind == MM_LOSS_STRESS)
    return vec_bwd_t<T, KIND, MP, MM_LOSS_STRESS>(x, target, n, m, rb, re, 1, grad, ws, st, la, loss_out);
  return vec_bwd_t<T, KIND, MP, MM_LOSS_QUOTIENT>(x, target, n, m, rb, re, 1, grad, ws, st, la, loss_out);
}

// node minibatch: clear the accumulators of ALL nodes, ordered-pair kernel over the batch (SUB), finalize over ALL nodes
template <typename T, int KIND, int MP>
int vec_loss_subset_t(int loss_kind, const T* x, const T* dense, const T* scale_raw, int64_t n_total, int m, const int64_t* idx,
                      int64_t bs, int64_t rb, int64_t re, double alpha, double eps, int terms, const double* loss_params, T* loss_out,
                      T* grad, void* ws, hipStream_t st) {
  // Rows per workgroup: a minibatch of 512 nodes is 2 column blocks — with the 64-row tiles of the full-size launches it
  // would be 16 workgroups on 256 CUs (103 us for the 130 816 pairs of Lorentz(24)); fewer rows mean more workgroups but
  // also more column flushes onto the same accumulator addresses.  MM_VEC_SUBSET_ROWS overrides (multiples of 8).
  constexpr int TI = 8;
  static const int rows_env = [] { const char* e = std::getenv("MM_VEC_SUBSET_ROWS"); return e ? std::atoi(e) : 0; }();
  const int rows = std::min(kVecSubMaxRows, rows_env > 0 ? (rows_env + 7) / 8 * 8 : 16);
  T* acc = static_cast<T*>(ws);
  T* slots = acc + size_t(n_total) * (MP + 1);
  LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, slots, loss_params};
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(T) * (size_t(n_total) * (MP + 1) + 2 * kLossSlots), st);
  if (e != hipSuccess) return int(e);
  if (re > rb && bs > 1) {
    const dim3 grid(int((bs + kVBlock - 1) / kVBlock), int((bs + rows - 1) / rows));
    if (loss_kind == MM_LOSS_STRESS)
      vec_pdist_bwd_kernel<T, KIND, MP, TI, MM_LOSS_STRESS, true><<<grid, dim3(kVBlock), 0, st>>>(
          x, dense, int(bs), m, int(rb), int(re), 1, acc, la, idx, int(n_total), rows);
    else
      vec_pdist_bwd_kernel<T, KIND, MP, TI, MM_LOSS_QUOTIENT, true><<<grid, dim3(kVBlock), 0, st>>>(
          x, dense, int(bs), m, int(rb), int(re), 1, acc, la, idx, int(n_total), rows);
    MMV_CHECK();
  }
  vec_pdist_finalize_kernel<T, KIND, MP><<<dim3(int((n_total + 127) / 128)), dim3(128), 0, st>>>(
      x, acc, int(n_total), m, grad, slots, scale_raw, loss_out);
  MMV_CHECK();
  return MM_OK;
}

#define MMV_DISPATCH_MP(m, ...)                             \
  switch (pad_dim(m)) {                                     \
    case 4: { constexpr int MP = 4; return __VA_ARGS__; }   \
    case 8: { constexpr int MP = 8; return __VA_ARGS__; }   \
    case 12: { constexpr int MP = 12; return __VA_ARGS__; } \
    case 16: { constexpr int MP = 16; return __VA_ARGS__; } \
    case 24: { constexpr int MP = 24; return __VA_ARGS__; } \
    case 32: { constexpr int MP = 32; return __VA_ARGS__; } \
    case 48: { constexpr int MP = 48; return __VA_ARGS__; } \
    default: { constexpr int MP = 64; return __VA_ARGS__; } \
  }

#define MMV_DISPATCH_KIND(kind, ...)                                         \
  switch (kind) {                                                            \
    case MM_EUCLIDEAN: { constexpr int KIND = MM_EUCLIDEAN; __VA_ARGS__ }    \
    case MM_LORENTZ: { constexpr int KIND = MM_LORENTZ; __VA_ARGS__ }        \
    case MM_SPHERE: { constexpr int KIND = MM_SPHERE; __VA_ARGS__ }          \
    default: return MM_ERR_ARG;                                              \
  }

#define MMV_DISPATCH_T(dtype, ...)                               \
  if ((dtype) == MM_F32) { using T = float; __VA_ARGS__ }        \
  else if ((dtype) == MM_F64) { using T = double; __VA_ARGS__ }  \
  else return MM_ERR_ARG;

// matrix-core path of the fused objective (vec_gram.hip)
bool vec_gram_supports(int dtype, int kind, int64_t n, int m);
int vec_gram_loss(int dtype, int kind, int loss_kind, const void* x, const void* target, const void* scale_raw, int64_t n,
                  int m, int64_t row_begin, int64_t row_end, double alpha, double eps, int terms, const double* loss_params, void* loss_out,
                  void* grad, void* slots, hipStream_t st);

// One launch for `count` vector-space parameters that share an optimizer TYPE (vec_step.hpp, VecGroupParam): momentum-free
// RSGD or Riemannian Adam, each parameter with its own hyper-parameters and state.
template <typename T>
int vec_group_step_t(int optimizer, int count, const VecGroupParam* ps, hipStream_t st) {
  RadamMulti<T> s{};
  int64_t most = 0;
  for (int t = 0; t < count; ++t) {
    const VecGroupParam& q = ps[t];
    s.p.x[t] = static_cast<const T*>(q.x);
    s.p.eg[t] = static_cast<const T*>(q.grad);
    s.p.xnew[t] = static_cast<T*>(q.xnew);
    s.p.cnt[t] = q.cnt;
    s.p.m[t] = q.m;
    s.p.kind[t] = q.kind;
    s.p.lr[t] = T(q.lr);
    s.p.max_grad_norm[t] = T(q.max_grad_norm);
    s.p.exact[t] = q.exact;
    s.exp_avg[t] = static_cast<T*>(q.state0);
    s.exp_avg_sq[t] = static_cast<T*>(q.state1);
    s.step[t] = q.step;
    s.ticket[t] = q.ticket;
    s.beta1[t] = T(q.beta1); s.beta2[t] = T(q.beta2); s.eps[t] = T(q.eps);
    s.nc[t] = q.nc;
    most = q.cnt > most ? q.cnt : most;
  }
  if (most == 0) return MM_OK;
  const dim3 grid(unsigned((most + 127) / 128), unsigned(count)), block(128);
  if (optimizer == MM_OPT_RADAM) vec_radam_multi_kernel<T><<<grid, block, 0, st>>>(s);
  else vec_rsgd_multi_kernel<T><<<grid, block, 0, st>>>(s.p);
  MMV_CHECK();
  return MM_OK;
}

int vec_group_step(int dtype, int optimizer, int count, const VecGroupParam* ps, hipStream_t st) {
  if (count < 1 || count > kRsgdMultiMax || !ps) return MM_ERR_ARG;
  if (dtype == MM_F32) return vec_group_step_t<float>(optimizer, count, ps, st);
  if (dtype == MM_F64) return vec_group_step_t<double>(optimizer, count, ps, st);
  return MM_ERR_ARG;
}

}  // namespace mm

using namespace mm;

extern "C" {

int mm_vec_max_dim(void) { return kVecMaxDim; }
int mm_vec_fused_step_supports(int dtype, int kind, int m) { return vec_fused_step_supports(dtype, kind, m) ? 1 : 0; }

size_t mm_vec_pdist_ws_bytes(int dtype, int64_t n, int m) {
  // acc [pad + 1][n] | loss slots [2][256] | zero-padded points [n + 1][pad] (symmetric backward, vec_sym.hip) | the share table of
  // the balanced walk (round 6; spd_ws.hpp WalkShares::of_cached: kShareTabEntries x 32 bytes, 32-byte aligned)
  return vec_ws_tables_end(dtype == MM_F64 ? 8 : 4, n, pad_dim(m)) + size_t(kShareTabEntries) * 32;
}

int mm_vec_pdist_fwd(int dtype, int kind, const void* x, int64_t n, int m, int64_t row_begin, int64_t row_end,
                     int squared, void* out, mm_stream_t stream) {
  if (!x || n < 0 || m < 1 || row_begin < 0 || row_end > n || row_begin > row_end || n > (1 << 30)) return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (!out && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, MMV_DISPATCH_MP(m, (vec_fwd_t<T, KIND, MP>(
      static_cast<const T*>(x), n, m, row_begin, row_end, squared, static_cast<T*>(out), st)))))
}

int mm_vec_pdist_bwd(int dtype, int kind, const void* x, const void* g, int64_t n, int m, int64_t row_begin,
                     int64_t row_end, int squared, void* grad_x, void* ws, mm_stream_t stream) {
  if (!x || !grad_x || !ws || n < 1 || m < 1 || row_begin < 0 || row_end > n || row_begin > row_end || n > (1 << 30))
    return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (!g && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, MMV_DISPATCH_MP(m, (vec_bwd_t<T, KIND, MP>(
      static_cast<const T*>(x), static_cast<const T*>(g), n, m, row_begin, row_end, squared,
      static_cast<T*>(grad_x), ws, st)))))
}

int mm_vec_pdist_loss(int dtype, int kind, int loss_kind, const void* x, const void* target, const void* scale_raw,
                      int64_t n, int m, int64_t row_begin, int64_t row_end, double alpha, double eps, int terms, const double* loss_params,
                      void* loss_out, void* grad_x, void* ws, mm_stream_t stream) {
  if (!x || !grad_x || !ws || !loss_out || n < 1 || m < 1 || row_begin < 0 || row_end > n || row_begin > row_end ||
      n > (1 << 30))
    return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  if (!target && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Inner-product manifolds: the symmetric VALU form (vec_sym.hpp: every pair once, sums flushed straight into the
  // gradient — two launches) up to m = 16 in both precisions (Lorentz(11) n = 4039 training step, fp32: 51.6 us against
  // 63.7 us through the matrix cores; fp64 99.7 against 141); the matrix-core kernel serves fp32 17 <= m <= 32.
  // MM_VEC_LOSS_GRAM=1 / MM_VEC_LOSS_VALU=1 force either.
  static const bool force_gram = [] { const char* e = std::getenv("MM_VEC_LOSS_GRAM"); return e && e[0] == '1'; }();
  static const bool force_valu = [] { const char* e = std::getenv("MM_VEC_LOSS_VALU"); return e && e[0] == '1'; }();
  if (vec_gram_supports(dtype, kind, n, m) && !force_valu && (force_gram || (dtype == MM_F32 && m > 16)))
    return vec_gram_loss(dtype, kind, loss_kind, x, target, scale_raw, n, m, row_begin, row_end, alpha, eps, terms, loss_params,
                         loss_out, grad_x, ws, st);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, MMV_DISPATCH_MP(m, (vec_loss_t<T, KIND, MP>(
      loss_kind, static_cast<const T*>(x), static_cast<const T*>(target), static_cast<const T*>(scale_raw), n, m,
      row_begin, row_end, alpha, eps, terms, loss_params, static_cast<T*>(loss_out), static_cast<T*>(grad_x), ws, st)))))
}

int mm_vec_pdist_loss_subset(int dtype, int kind, int loss_kind, const void* x, const void* dense, const void* scale_raw,
                             int64_t n_total, int m, const int64_t* idx, int64_t bs, int64_t row_begin, int64_t row_end, double alpha,
                             double eps, int terms, const double* loss_params, void* loss_out, void* grad_x, void* ws,
                             mm_stream_t stream) {
  if (!x || !grad_x || !ws || !loss_out || n_total < 1 || n_total > (1 << 30) || m < 1 || bs < 0 || bs > n_total || row_begin < 0 ||
      row_end > bs || row_begin > row_end)
    return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  if ((!dense || !idx) && mm_pair_offset(bs, row_end) > mm_pair_offset(bs, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, MMV_DISPATCH_MP(m, (vec_loss_subset_t<T, KIND, MP>(
      loss_kind, static_cast<const T*>(x), static_cast<const T*>(dense), static_cast<const T*>(scale_raw), n_total, m, idx, bs,
      row_begin, row_end, alpha, eps, terms, loss_params, static_cast<T*>(loss_out), static_cast<T*>(grad_x), ws, st)))))
}

int mm_vec_dist(int dtype, int kind, const void* x, const void* y, const void* g, int64_t cnt, int m, int squared,
                void* out, void* grad_x, void* grad_y, mm_stream_t stream) {
  if (cnt < 0 || m < 1 || (cnt > 0 && (!x || !y)) || ((grad_x != nullptr) != (grad_y != nullptr)) ||
      (grad_x && !g))
    return MM_ERR_ARG;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    vec_dist_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(static_cast<const T*>(x), static_cast<const T*>(y),
        static_cast<const T*>(g), cnt, m, squared, static_cast<T*>(out), static_cast<T*>(grad_x),
        static_cast<T*>(grad_y));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_map(int dtype, int kind, int op, const void* x, const void* u, const void* y, int64_t cnt, int m, void* out,
               mm_stream_t stream) {
  if (cnt < 0 || m < 1 || op < 0 || op > MM_VEC_LOG || (cnt > 0 && (!x || !out))) return MM_ERR_ARG;
  if (cnt > 0 && op != MM_VEC_PROJX && !u) return MM_ERR_ARG;
  if (cnt > 0 && op == MM_VEC_TRANSP && !y) return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    vec_map_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(op, static_cast<const T*>(x), static_cast<const T*>(u),
        static_cast<const T*>(y), cnt, m, static_cast<T*>(out));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_norm(int dtype, int kind, const void* u, int64_t cnt, int m, int squared, void* out, mm_stream_t stream) {
  if (cnt < 0 || m < 1 || (cnt > 0 && (!u || !out))) return MM_ERR_ARG;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    vec_norm_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(static_cast<const T*>(u), cnt, m, squared,
        static_cast<T*>(out));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_rsgd_step(int dtype, int kind, const void* x, const void* egrad, int64_t cnt, int m, double lr,
                     double max_grad_norm, int exact, void* x_new, mm_stream_t stream) {
  if (cnt < 0 || m < 1 || (cnt > 0 && (!x || !egrad || !x_new))) return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  {   // m <= 16: the register-resident form (vec_step.hip)
    const int rc = vec_rule_step(dtype, kind, MM_OPT_RSGD, x, egrad, x_new, cnt, m, lr, 0.0, 0.0, max_grad_norm, exact, nullptr,
                                 nullptr, 0.0, 0.0, 0.0, 0, nullptr, nullptr, st);
    if (rc != MM_ERR_UNSUPPORTED) return rc;
  }
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    vec_rsgd_step_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(static_cast<const T*>(x),
        static_cast<const T*>(egrad), cnt, m, T(lr), T(max_grad_norm), exact, static_cast<T*>(x_new));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_rsgd_momentum_step(int dtype, int kind, const void* x, const void* egrad, void* momentum_buffer, int64_t cnt,
                              int m, double lr, double momentum, double dampening, double max_grad_norm, int exact,
                              void* x_new, mm_stream_t stream) {
  if (cnt < 0 || m < 1 || (cnt > 0 && (!x || !egrad || !momentum_buffer || !x_new))) return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (momentum != 0.0) {   // m <= 16: the register-resident form (vec_step.hip)
    const int rc = vec_rule_step(dtype, kind, MM_OPT_RSGD, x, egrad, x_new, cnt, m, lr, momentum, dampening, max_grad_norm, exact,
                                 momentum_buffer, nullptr, 0.0, 0.0, 0.0, 0, nullptr, nullptr, st);
    if (rc != MM_ERR_UNSUPPORTED) return rc;
  }
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    vec_rsgd_momentum_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(static_cast<const T*>(x),
        static_cast<const T*>(egrad), static_cast<T*>(momentum_buffer), cnt, m, T(lr), T(momentum), T(dampening),
        T(max_grad_norm), exact, static_cast<T*>(x_new));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_radam_step(int dtype, int kind, const void* x, const void* egrad, void* exp_avg, void* exp_avg_sq,
                      double* step, unsigned* ticket, int64_t cnt, int m, double lr, double beta1, double beta2, int nc,
                      double eps, double max_grad_norm, int exact, void* x_new, mm_stream_t stream) {
  if (cnt < 0 || m < 1 || !step || !ticket || (cnt > 0 && (!x || !egrad || !exp_avg || !exp_avg_sq || !x_new)))
    return MM_ERR_ARG;
  if (m > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  if (cnt == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  {   // m <= 16: the register-resident form (vec_step.hip)
    const int rc = vec_rule_step(dtype, kind, MM_OPT_RADAM, x, egrad, x_new, cnt, m, lr, 0.0, 0.0, max_grad_norm, exact, exp_avg,
                                 exp_avg_sq, beta1, beta2, eps, nc, step, ticket, st);
    if (rc != MM_ERR_UNSUPPORTED) return rc;
  }
  const unsigned nb = unsigned((cnt + 127) / 128);
  MMV_DISPATCH_T(dtype, MMV_DISPATCH_KIND(kind, {
    AdamArgs<T> a{T(lr), T(beta1), T(beta2), T(eps), T(max_grad_norm), nc, exact, step, ticket};
    vec_radam_step_kernel<T, KIND><<<dim3(nb), dim3(128), 0, st>>>(static_cast<const T*>(x),
        static_cast<const T*>(egrad), static_cast<T*>(exp_avg), static_cast<T*>(exp_avg_sq), cnt, m, a,
        static_cast<T*>(x_new));
    MMV_CHECK(); return MM_OK; }))
}

int mm_vec_radam_step_multi(int dtype, int count, const int* kinds, const void* const* xs, const void* const* egrads,
                            void* const* exp_avg, void* const* exp_avg_sq, double* const* steps,
                            unsigned* const* tickets, const int64_t* cnts, const int* ms, double lr, double beta1,
                            double beta2, int nc, double eps, double max_grad_norm, int exact, void* const* x_new,
                            mm_stream_t stream) {
  if (count < 1 || !kinds || !xs || !egrads || !exp_avg || !exp_avg_sq || !steps || !tickets || !cnts || !ms || !x_new)
    return MM_ERR_ARG;
  if (count > kRsgdMultiMax) return MM_ERR_UNSUPPORTED;
  for (int t = 0; t < count; ++t) {
    if (cnts[t] < 1 || ms[t] < 1 || !xs[t] || !egrads[t] || !exp_avg[t] || !exp_avg_sq[t] || !steps[t] || !tickets[t] ||
        !x_new[t] || kinds[t] < MM_EUCLIDEAN || kinds[t] > MM_SPHERE)
      return MM_ERR_ARG;
    if (ms[t] > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  }
  VecGroupParam ps[kRsgdMultiMax];
  for (int t = 0; t < count; ++t)
    ps[t] = VecGroupParam{kinds[t], ms[t], cnts[t], xs[t], egrads[t], x_new[t], exp_avg[t], exp_avg_sq[t], steps[t], tickets[t],
                          lr, max_grad_norm, beta1, beta2, eps, nc, exact};
  return vec_group_step(dtype, MM_OPT_RADAM, count, ps, static_cast<hipStream_t>(stream));
}

int mm_vec_rsgd_multi_max(void) { return kRsgdMultiMax; }

int mm_vec_rsgd_step_multi(int dtype, int count, const int* kinds, const void* const* xs, const void* const* egrads,
                           const int64_t* cnts, const int* ms, double lr, double max_grad_norm, int exact,
                           void* const* x_new, mm_stream_t stream) {
  if (count < 1 || !kinds || !xs || !egrads || !cnts || !ms || !x_new) return MM_ERR_ARG;
  if (count > kRsgdMultiMax) return MM_ERR_UNSUPPORTED;
  for (int t = 0; t < count; ++t) {
    if (cnts[t] < 0 || ms[t] < 1 || (cnts[t] > 0 && (!xs[t] || !egrads[t] || !x_new[t]))) return MM_ERR_ARG;
    if (kinds[t] < MM_EUCLIDEAN || kinds[t] > MM_SPHERE) return MM_ERR_ARG;
    if (ms[t] > kVecMaxDim) return MM_ERR_UNSUPPORTED;
  }
  VecGroupParam ps[kRsgdMultiMax];
  for (int t = 0; t < count; ++t)
    ps[t] = VecGroupParam{kinds[t], ms[t], cnts[t], xs[t], egrads[t], x_new[t], nullptr, nullptr, nullptr, nullptr,
                          lr, max_grad_norm, 0.0, 0.0, 0.0, 0, exact};
  return vec_group_step(dtype, MM_OPT_RSGD, count, ps, static_cast<hipStream_t>(stream));
}

}  // extern "C"
