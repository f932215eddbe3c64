// Stein divergence between SPD matrices (included by spd.hip inside namespace mm):
//   S(X, Y) = log det((X + Y) / 2) - (log det X + log det Y) / 2
// graphembed/graphembed/manifolds/spd.py:183-194 (stein_div / stein_pdiv) and 246-295
// (PairwiseSteinDivergence), linalg/torch_batch.py:173-197 (PLogDet).  The reference's backward builds a dense
// (n, n, d, d) tensor; here the pair kernels reuse the tiling, workspace and accumulators of the affine-invariant
// kernels.  Per pair: Cholesky of (X_i + X_j)/2 in registers -> log det from its diagonal; backward: its inverse
// P = ((X_i + X_j)/2)^-1 goes to BOTH endpoints,
//   dS/dX_i = P/2 - X_i^-1/2,      grad_i = sum_j g_ij P_ij / 2 - (sum_j g_ij) X_i^-1 / 2,
// so rows and columns accumulate the same NP + 1 numbers (P packed, and g) — row side through the transposing
// reduction, column side per lane — into accM[0..NP) and accS[0]; finalize applies the X^-1 term per node.
#pragma once

template <typename T, int D>
__device__ __forceinline__ T stein_pair(const T (&xi)[Packed<D>::NP], const T (&xj)[Packed<D>::NP], T ldsum,
                                        T (&l)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  T mid[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) mid[k] = T(0.5) * (xi[k] + xj[k]);
  cholesky<T, D>(mid, l);
  T ld = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) ld += Num<T>::log(l[pidx(k, k)]);
  return (ld + ld) - T(0.5) * ldsum;
}

template <typename T, int D, int TI>
__global__ __launch_bounds__(kBlock) void spd_stein_fwd_kernel(const T* __restrict__ nodeX, const T* __restrict__ nodeLd,
                                                               int n, int row_begin, int row_end, int squared, T wmin,
                                                               T* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  const TileId tile = fold_tile<TI>(n, row_begin, row_end);
  if (!tile.ok) return;
  const int i0 = tile.i0, i1 = min(i0 + TI, row_end), jbase = tile.jbase;
  if (jbase + (int(threadIdx.x) & ~63) + 63 <= i0) return;  // whole wavefront below the diagonal
  const int j = jbase + threadIdx.x;
  const bool jin = j < n;
  T xj[NP], ldj = T(0);
#pragma unroll
  for (int k = 0; k < NP; ++k) xj[k] = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) xj[pidx(k, k)] = T(1);
  if (jin) {
#pragma unroll
    for (int k = 0; k < NP; ++k) xj[k] = nodeX[size_t(j) * NP + k];
    ldj = nodeLd[j];
  }
  const int64_t base = pair_off(n, row_begin);
  for (int i = i0; i < i1; ++i) {
    T xi[NP], l[NP];  // wave-uniform row operand -> scalar loads
#pragma unroll
    for (int k = 0; k < NP; ++k) xi[k] = nodeX[size_t(i) * NP + k];
    T s = Num<T>::max(stein_pair<T, D>(xi, xj, nodeLd[i] + ldj, l), wmin);
    if (!squared) s = Num<T>::sqrt(s);
    if (jin && j > i) out[pair_off(n, i) - base + (j - i - 1)] = s;
  }
}

template <typename T, int D, int TI>
__global__ __launch_bounds__(kBlock) void spd_stein_bwd_kernel(const T* __restrict__ nodeX, const T* __restrict__ nodeLd,
                                                               const T* __restrict__ g, int n, int row_begin,
                                                               int row_end, int squared, T wmin, T* __restrict__ accM,
                                                               T* __restrict__ accS) {
  constexpr int NP = Packed<D>::NP;
  constexpr int NV = NP + 1;  // packed P and the upstream gradient itself
  constexpr int NW = kBlock / 64;
  __shared__ T redR[NW][TI][NV];
  __shared__ T colS[NW][NV][64];
  const TileId tile = fold_tile<NW * TI, 64>(n, row_begin, row_end);
  if (!tile.ok) return;  // block-uniform
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int jbase = tile.jbase;
  const int i0 = tile.i0 + wave * TI, i1 = max(i0, min(i0 + TI, min(tile.i0 + NW * TI, row_end)));
  const bool wave_live = i0 < i1 && jbase + 63 > i0;
  const int j = jbase + lane;
  const bool jin = j < n;
  bool red_writer;
  const int red_slot = reduce_slot<NV>(lane, red_writer);
  T xj[NP], ldj = T(0), accJ[NV];
#pragma unroll
  for (int k = 0; k < NP; ++k) xj[k] = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) xj[pidx(k, k)] = T(1);
#pragma unroll
  for (int k = 0; k < NV; ++k) accJ[k] = T(0);
  if (jin) {
#pragma unroll
    for (int k = 0; k < NP; ++k) xj[k] = nodeX[size_t(j) * NP + k];
    ldj = nodeLd[j];
  }
  const int64_t base = pair_off(n, row_begin);
  if (wave_live) {
    for (int i = i0; i < i1; ++i) {
      T xi[NP], l[NP], li[NP], v[NV];
#pragma unroll
      for (int k = 0; k < NP; ++k) xi[k] = nodeX[size_t(i) * NP + k];
      const bool valid = jin && j > i;
      T gs = pair_row_load<T>(g, n, base, i, j);
      gs = valid ? gs : T(0);
      const T div = stein_pair<T, D>(xi, xj, nodeLd[i] + ldj, l);
      if (!squared) gs *= T(0.5) * Num<T>::rsqrt(Num<T>::max(div, wmin));  // d sqrt(clamp(S)) / dS, clamp transparent
      invert_lower<T, D>(l, li);
      // P = L^-T L^-1 (packed symmetric), scaled by the upstream gradient
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          T acc = T(0);
#pragma unroll
          for (int k = r; k < D; ++k) acc = Num<T>::fma(li[pidx(k, r)], li[pidx(k, c)], acc);
          v[pidx(r, c)] = gs * acc;
        }
      v[NP] = gs;
#pragma unroll
      for (int k = 0; k < NV; ++k) accJ[k] += v[k];
      // row side: transposing reduction, every lane ends with the wavefront total of ONE of the NV values
      const T tot = wave_reduce_transposed<NV, T>(v, lane);
      if (red_writer) redR[wave][i - i0][red_slot] = tot;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < TI * NV; t += 64) {
      const int k = t / TI, il = t % TI;
      if (i0 + il < i1) {
        T* dst = k < NP ? &accM[size_t(k) * n + i0 + il] : &accS[i0 + il];
        atomic_add(dst, redR[wave][il][k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) colS[wave][k][lane] = accJ[k];
  __syncthreads();
  if (jin) {
    for (int k = wave; k < NV; k += NW) {
      T sum = colS[0][k][lane];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) sum += colS[wv][k][lane];
      T* dst = k < NP ? &accM[size_t(k) * n + j] : &accS[j];
      atomic_add(dst, sum);
    }
  }
}

// grad_x[i] = (sum_pairs g P) / 2 - (sum_pairs g) X_i^-1 / 2  (symmetric, full DxD); accumulators left clean
template <typename T, int D>
__global__ void spd_stein_finalize_kernel(const T* __restrict__ nodeL, T* __restrict__ accM, T* __restrict__ accS, int n,
                                          T* __restrict__ grad) {
  constexpr int NP = Packed<D>::NP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T li[NP], out[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) li[k] = nodeL[size_t(i) * NP + k];
  const T gsum = accS[i];
  accS[i] = T(0);
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T xinv = T(0);  // X^-1 = L^-T L^-1
#pragma unroll
      for (int k = r; k < D; ++k) xinv = Num<T>::fma(li[pidx(k, r)], li[pidx(k, c)], xinv);
      const size_t a = size_t(pidx(r, c)) * n + i;
      out[pidx(r, c)] = T(0.5) * (accM[a] - gsum * xinv);
      accM[a] = T(0);
    }
  store_sym_full<T, D>(grad + size_t(i) * D * D, out);
}

// element-wise stein_div(x[k], y[k]) with optional gradients (spd.py:183-189): one thread per pair
template <typename T, int D>
__global__ void spd_stein_div_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ g, int64_t m,
                                     int squared, T wmin, T* __restrict__ out, T* __restrict__ gx, T* __restrict__ gy) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k0 >= m) return;
  T xs[NP], ys[NP], lx[NP], ly[NP], l[NP];
  load_sym_packed<T, D>(x + k0 * D * D, xs);
  load_sym_packed<T, D>(y + k0 * D * D, ys);
  cholesky<T, D>(xs, lx);
  cholesky<T, D>(ys, ly);
  T ldx = T(0), ldy = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) { ldx += Num<T>::log(lx[pidx(k, k)]); ldy += Num<T>::log(ly[pidx(k, k)]); }
  const T div = stein_pair<T, D>(xs, ys, T(2) * (ldx + ldy), l);
  const T s = Num<T>::max(div, wmin);
  if (out) out[k0] = squared ? s : Num<T>::sqrt(s);
  if (gx) {
    T gs = g[k0];
    if (!squared) gs *= T(0.5) * Num<T>::rsqrt(s);
    T li[NP], lix[NP], liy[NP], ox[NP], oy[NP];
    invert_lower<T, D>(l, li);
    invert_lower<T, D>(lx, lix);
    invert_lower<T, D>(ly, liy);
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T p = T(0), ix = T(0), iy = T(0);
#pragma unroll
        for (int k = r; k < D; ++k) {
          p = Num<T>::fma(li[pidx(k, r)], li[pidx(k, c)], p);
          ix = Num<T>::fma(lix[pidx(k, r)], lix[pidx(k, c)], ix);
          iy = Num<T>::fma(liy[pidx(k, r)], liy[pidx(k, c)], iy);
        }
        ox[pidx(r, c)] = T(0.5) * gs * (p - ix);
        oy[pidx(r, c)] = T(0.5) * gs * (p - iy);
      }
    store_sym_full<T, D>(gx + k0 * D * D, ox);
    store_sym_full<T, D>(gy + k0 * D * D, oy);
  }
}

template <typename T, int D>
int spd_stein_fwd_t(const T* x, int64_t n, int64_t rb, int64_t re, int squared, double wmin, T* out, void* wsp, int flags,
                    hipStream_t st) {
  constexpr int TI = 8;
  Ws<T> ws(wsp, n, D);
  int rc = spd_pdist_prepare<T, D>(x, n, ws, flags, st);
  if (rc) return rc;
  if (re <= rb || pair_off(n, re) == pair_off(n, rb)) return MM_OK;
  {
    ProfScope prof(PROF_SPD_FWD, st);
    spd_stein_fwd_kernel<T, D, TI><<<fold_grid<TI>(n, rb, re), dim3(kBlock), 0, st>>>(
        ws.nodeX, ws.nodeLd, int(n), int(rb), int(re), squared, T(wmin), out);
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

template <typename T, int D>
int spd_stein_bwd_t(const T* x, const T* g, int64_t n, int64_t rb, int64_t re, int squared, double wmin, T* grad,
                    void* wsp, int flags, hipStream_t st) {
  constexpr int TI = 8;
  Ws<T> ws(wsp, n, D);
  int rc = spd_pdist_prepare<T, D>(x, n, ws, flags, st);
  if (rc) return rc;
  if (re > rb && pair_off(n, re) > pair_off(n, rb)) {
    ProfScope prof(PROF_SPD_BWD, st);
    spd_stein_bwd_kernel<T, D, TI><<<fold_grid<(kBlock / 64) * TI, 64>(n, rb, re), dim3(kBlock), 0, st>>>(
        ws.nodeX, ws.nodeLd, g, int(n), int(rb), int(re), squared, T(wmin), ws.accM, ws.accS);
  }
  MM_CHECK_LAUNCH();
  spd_stein_finalize_kernel<T, D><<<dim3((n + 127) / 128), dim3(128), 0, st>>>(ws.nodeL, ws.accM, ws.accS, int(n), grad);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
