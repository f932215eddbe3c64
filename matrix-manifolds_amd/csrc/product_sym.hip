// The mixed-manifold pair kernel of product embeddings, SYMMETRIC form: every UNORDERED pair is visited once and feeds both of
// its points (product_pairs.hip visits every ordered pair: twice the arithmetic, no cross-lane step — the better trade at
// csphd's n ~ 1e3, where the launch is latency, and 3 x slower at n = 5000).  The structure is that of the vector manifolds'
// symmetric backward (vec_sym.hpp) with the per-pair work of product_pair_kernel:
//   * a node table [n + 1][W] written by a preparation launch: every vector factor zero-padded to 8 coordinates (Euclidean:
//     coordinate 7 is 1, so that the column sums' and the row reduction's entry 7 is sum w — 2 (x sum w - sum w x_other)
//     needs it), then L^-1 and L of the SPD point (packed);
//   * lanes own consecutive columns j with their points and column sums in registers; the row point is wave-uniform
//     (scalar loads of its table row, one row ahead);
//   * per pair: every factor's squared distance (modules.py:84-88), the weighted sum, the loss term and its derivative
//     (objectives.py:16-45), then per factor w = dl/dm softplus(s_f) d(d2_f)/dq: column side acc_j += w x_i, row side
//     w x_j; SPD factor: M = 2 g log(A) in the row's frame (Jacobi), column side L_i^-T M L_i^T, row side M;
//   * the row side's NV x 8 + NP values are summed across the wavefront by the transposing reduction; per slice they leave
//     as atomics into the SAME accumulators the ordered kernel fills (vector: acc[k][node]; SPD: the row's -L^-T (sum M) L^T,
//     which is what sym(S X^-1) of the finalize needs: -L^-T M L^-1 = sym((-L^-T M L^T) X^-1)), so product_pair_finalize_kernel
//     and product_step_kernel serve both forms;
//   * ONE resident grid with statically balanced shares of the column walk (spd_ws.hpp, ColWalk).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>

#include "../../include/mm_manifolds.h"
#include "loss.hpp"
#include "product_sym.hpp"
#include "smallmat.hpp"
#include "spd_ws.hpp"
#include "vecfn.hpp"

namespace mm {

namespace {

constexpr int kPSWaves = 4, kPSTI = 16;

template <int NV, int SD> struct PSLayout {
  static constexpr int NPS = SD > 0 ? SD * (SD + 1) / 2 : 0;
  static constexpr int VEC = NV * kPSW;
  static constexpr int W = (VEC + 2 * NPS + 3) / 4 * 4;   // a table row: the vector factors, L^-1, L
  static constexpr int NR = VEC + NPS;                    // values of the row-side reduction
};

template <typename T, int NV, int SD>
__global__ void product_sym_prep_kernel(PArgs<T> pa, int n, T* __restrict__ tab) {
  using L = PSLayout<NV, SD>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  T* row = tab + size_t(i) * L::W;
#pragma unroll
  for (int k = 0; k < L::W; ++k) row[k] = T(0);   // (row n: the padding row of the row operand requested one row ahead)
  if (i == n) return;
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    const PVec<T>& F = pa.v[f];
    for (int k = 0; k < F.m; ++k) row[f * kPSW + k] = F.x[size_t(i) * F.m + k];
    if (F.kind == MM_EUCLIDEAN) row[f * kPSW + kPSW - 1] = T(1);
  }
  if constexpr (SD > 0) {
    constexpr int NPS = L::NPS;
    T xs[NPS], l[NPS], li[NPS];
    load_sym_packed<T, SD>(pa.s.x + size_t(i) * SD * SD, xs);
    cholesky<T, SD>(xs, l);
    invert_lower<T, SD>(l, li);
#pragma unroll
    for (int k = 0; k < NPS; ++k) { row[L::VEC + k] = li[k]; row[L::VEC + NPS + k] = l[k]; }
  }
}

template <typename T> __device__ __forceinline__ T q8(int kind, const T (&xi)[kPSW], const T (&xj)[kPSW]) {
  using N = Num<T>;
  T q = T(0);
  if (kind == MM_EUCLIDEAN) {
#pragma unroll
    for (int k = 0; k < kPSW; ++k) { const T df = xj[k] - xi[k]; q = N::fma(df, df, q); }
  } else {
#pragma unroll
    for (int k = 1; k < kPSW; ++k) q = N::fma(xi[k], xj[k], q);
    q = kind == MM_LORENTZ ? N::fma(xi[0], xj[0], -q) : N::fma(xi[0], xj[0], q);
  }
  return q;
}
template <typename T> __device__ __forceinline__ T value_rt(int kind, T q) {
  return kind == MM_EUCLIDEAN ? PairFn<T, MM_EUCLIDEAN>::value(q, 1)
                              : (kind == MM_LORENTZ ? PairFn<T, MM_LORENTZ>::value(q, 1) : PairFn<T, MM_SPHERE>::value(q, 1));
}
template <typename T> __device__ __forceinline__ T dq_rt(int kind, T q) {
  return kind == MM_EUCLIDEAN ? PairFn<T, MM_EUCLIDEAN>::dq(q, 1)
                              : (kind == MM_LORENTZ ? PairFn<T, MM_LORENTZ>::dq(q, 1) : PairFn<T, MM_SPHERE>::dq(q, 1));
}

// KC: the vector factors' kinds as constants (product_args.hpp), -1 = read in the row loop
template <typename T, int NV, int SD, int LOSS, int KC = -1>
__global__ __launch_bounds__(64 * kPSWaves) void product_sym_kernel(PArgs<T> pa, const T* __restrict__ tab, const T* __restrict__ target,
                                                                    int n, int row_begin, int row_end, LossArgs<T> la, WalkShares shares) {
  using L = PSLayout<NV, SD>;
  constexpr int NW = kPSWaves, TI = kPSTI, NPS = L::NPS > 0 ? L::NPS : 1, DS = SD > 0 ? SD : 2, W = L::W, NR = L::NR, VEC = L::VEC;
  loss_resolve<T, LOSS>(la);
  T spv[NV > 0 ? NV : 1], dsv[NV > 0 ? NV : 1], sps = T(1), dss = T(0), loss_acc = T(0);
#pragma unroll
  for (int f = 0; f < NV; ++f) { spv[f] = softplus_of(pa.v[f].scale_raw); dsv[f] = T(0); }
  if constexpr (SD > 0) sps = softplus_of(pa.s.scale_raw);
  __shared__ T redM[NW][TI][NR];
  __shared__ T colS[NW][NR + (SD > 0 ? SD * SD - L::NPS : 0)][64];
  __shared__ T redJunk[NW][64];
  const ColWalk walk(n, row_begin, row_end, 64);
  // this workgroup's share, cut on the host: the block and row it starts at and its budget of units — one per row,
  // shares.cross per block entered (spd_ws.hpp, WalkShares / ColWalk::enter)
  int cb, r, rem;
  shares.of(walk, int(blockIdx.x), cb, r, rem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bool red_writer;
  const int red_slot = reduce_slot<NR>(lane, red_writer);
  const int64_t base = pair_off(n, row_begin);
  T* red_ptr = red_writer ? &redM[wave][0][red_slot] : &redJunk[wave][lane];
  const int red_step = red_writer ? NR : 0;

  while (rem > 0) {   // one pass per column block of this workgroup's share
    const int jbase = cb * 64;
    const int j = jbase + lane;
    const bool jin = j < n;
    const int jv = jin ? j : INT32_MIN;
    const unsigned joff = unsigned(min(j, n - 1)) * unsigned(sizeof(T));
    const T* colp = tab + size_t(min(j, n)) * W;   // (row n: the zero padding row)
    T xj[NV > 0 ? NV : 1][kPSW], accv[NV > 0 ? NV : 1][kPSW], yj[NPS], accS[DS][DS];
#pragma unroll
    for (int f = 0; f < NV; ++f)
#pragma unroll
      for (int k = 0; k < kPSW; ++k) { xj[f][k] = colp[f * kPSW + k]; accv[f][k] = T(0); }
    if constexpr (SD > 0) {
#pragma unroll
      for (int k = 0; k < NPS; ++k) yj[k] = T(0);
#pragma unroll
      for (int k = 0; k < SD; ++k) yj[pidx(k, k)] = T(1);   // (lanes past n: the identity, masked later)
      if (jin) {
#pragma unroll
        for (int k = 0; k < NPS; ++k) yj[k] = colp[VEC + NPS + k];
      }
#pragma unroll
      for (int a = 0; a < SD; ++a)
#pragma unroll
        for (int c = 0; c < SD; ++c) accS[a][c] = T(0);
    }
    const int hi = walk.hi(cb);
    while (rem > 0 && r < hi) {
      const int chunk = min(min(hi - r, NW * TI), rem);
      const int tw = (chunk + NW - 1) / NW;
      const int i0 = r + wave * tw, i1 = min(i0 + tw, r + chunk);
      if (i0 < i1) {
        unsigned roff = unsigned(i0) * unsigned(W * sizeof(T));
        T rowv[2][W];
#pragma unroll
        for (int k = 0; k < W; ++k) rowv[0][k] = tab[size_t(i0) * W + k];
        // the targets of this slice's rows: element (row, j) lives at pair_off(n, row) - base + (j - row - 1) (spd.hip)
        const int glast = min(i1, walk.re) - 1;
        const int64_t gk = glast - i0;
        const unsigned gmax = unsigned((gk * (n - 2) - (int64_t(i0) * gk + gk * (gk - 1) / 2)) * int64_t(sizeof(T)));
        const char* gslice = reinterpret_cast<const char*>(target + (pair_off(n, i0) - base - i0 - 1));
        unsigned goff = 0, gstep = unsigned(n - i0 - 2) * unsigned(sizeof(T));
        unsigned jslice = max(joff, unsigned(i0 + 1) * unsigned(sizeof(T)));
        T gq[2];
        auto request = [&](T& dst) __attribute__((always_inline)) {
          asm volatile("" : "+v"(jslice));
          dst = *reinterpret_cast<const T*>(gslice + goff + jslice);
          goff = min(goff + gstep, gmax);
          gstep -= unsigned(sizeof(T));
        };
        request(gq[0]);
        request(gq[1]);
        for (int ib = i0; ib < i1; ib += 2) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int irow = ib + u;
            const int ieff = (u == 0 || irow < i1) ? irow : INT32_MAX;   // (the slot past an odd slice's last row is masked)
            const T (&ri)[W] = rowv[u];
            roff = min(roff + unsigned(W * sizeof(T)), unsigned(n) * unsigned(W * sizeof(T)));   // (never past the padding row n)
            asm volatile("" : "+s"(roff));
            const T* rowp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(tab) + roff);
#pragma unroll
            for (int k = 0; k < W; ++k) rowv[u ^ 1][k] = rowp[k];
            const bool valid = jv > ieff;
            const T tgt = valid ? gq[u] : T(1);
            request(gq[u]);
            // ---- every factor's squared distance, the weighted sum
            T m = T(0), qv[NV > 0 ? NV : 1], d2v[NV > 0 ? NV : 1], xi[NV > 0 ? NV : 1][kPSW];
#pragma unroll
            for (int f = 0; f < NV; ++f) {
#pragma unroll
              for (int k = 0; k < kPSW; ++k) xi[f][k] = ri[f * kPSW + k];
              qv[f] = q8<T>(MM_PKIND(f), xi[f], xj[f]);
              d2v[f] = value_rt<T>(MM_PKIND(f), qv[f]);
              m = Num<T>::fma(spv[f], d2v[f], m);
            }
            T li[NPS], lc[NPS], lw[DS], vv[DS][DS], mlog[NPS], d2s = T(0);
            if constexpr (SD > 0) {
#pragma unroll
              for (int k = 0; k < NPS; ++k) { li[k] = ri[VEC + k]; lc[k] = ri[VEC + NPS + k]; }
              T a[NPS];
              congr_chol<T, SD>(li, yj, a);
              T s = T(0);
              if constexpr (SD == 2) {   // closed form (smallmat.hpp): no eigensolve for the 2x2 factor
                s = log_spd2<T>(a, pa.s.wmin, pa.s.wmax, mlog);
              } else {
                jacobi_eig<T, SD, true, true>(a, vv, T(64) * Num<T>::eps() * Num<T>::eps());
#pragma unroll
                for (int k = 0; k < SD; ++k) {  // eigenvalue clamp as _norm_log, spd.py:163-169
                  lw[k] = Num<T>::log(Num<T>::min(Num<T>::max(a[pidx(k, k)], pa.s.wmin), pa.s.wmax));
                  s = Num<T>::fma(lw[k], lw[k], s);
                }
              }
              d2s = Num<T>::max(s, pa.s.wmin);  // value clamp (gradient-transparent), spd.py:163-169
              m = Num<T>::fma(sps, d2s, m);
            }
            T dldm;
            const T l = loss_term<T, LOSS>(m, tgt, la, dldm);
            loss_acc += valid ? l : T(0);
            const T coef = valid ? dldm : T(0);
            // ---- gradients: this lane's column, and the row through the reduction
            T rsum[NR];
#pragma unroll
            for (int f = 0; f < NV; ++f) {
              dsv[f] += coef * d2v[f];
              const T w = coef * spv[f] * dq_rt<T>(MM_PKIND(f), qv[f]);
#pragma unroll
              for (int k = 0; k < kPSW; ++k) {
                accv[f][k] = Num<T>::fma(w, xi[f][k], accv[f][k]);
                rsum[f * kPSW + k] = w * xj[f][k];
              }
            }
            if constexpr (SD > 0) {
              dss += coef * d2s;
              const T gs = coef * sps;
              T cm[DS], mm_[NPS], cj[DS][DS];
              if constexpr (SD == 2) {
#pragma unroll
                for (int k = 0; k < NPS; ++k) mm_[k] = (gs + gs) * mlog[k];
              } else {
#pragma unroll
                for (int k = 0; k < SD; ++k) cm[k] = (gs + gs) * lw[k];
                vdvt<T, SD>(vv, cm, mm_);
              }
              lt_m_lt<T, SD>(li, lc, mm_, cj);
#pragma unroll
              for (int a2 = 0; a2 < SD; ++a2)
#pragma unroll
                for (int c = 0; c < SD; ++c) accS[a2][c] += cj[a2][c];
#pragma unroll
              for (int k = 0; k < L::NPS; ++k) rsum[VEC + k] = mm_[k];
            }
            *red_ptr = wave_reduce_transposed<NR, T>(rsum, lane);
            red_ptr += red_step;
          }
        }
        red_ptr -= red_step * ((i1 - i0 + 1) / 2 * 2);
        __builtin_amdgcn_wave_barrier();
        // row side of this slice.  Vector factors: sum w x_j per coordinate (Euclidean: entry 7 = sum w) into acc[k][row]
        for (int t = lane; t < tw * VEC; t += 64) {
          const int c = t / tw, il = t - c * tw;
          const int f = c / kPSW, k = c - f * kPSW;
          PVec<T> F = pa.v[0];
#pragma unroll
          for (int g = 1; g < NV; ++g)
            if (f == g) F = pa.v[g];
          if (i0 + il < i1) {
            if (k < F.m) atomic_add(&F.acc[size_t(k) * n + i0 + il], redM[wave][il][c]);
            else if (k == kPSW - 1 && F.kind == MM_EUCLIDEAN) atomic_add(&F.acc[size_t(kPMP) * n + i0 + il], redM[wave][il][c]);
          }
        }
        if constexpr (SD > 0) {
          // SPD factor: the row's -L^-T (sum M) L^T joins accS (sym(. X^-1) of the finalize gives -L^-T (sum M) L^-1)
          if (lane < i1 - i0) {
            const T* rowt = tab + size_t(i0 + lane) * W;
            T li[NPS], lc[NPS], msum[NPS], out[DS][DS];
#pragma unroll
            for (int k = 0; k < L::NPS; ++k) { li[k] = rowt[VEC + k]; lc[k] = rowt[VEC + L::NPS + k]; msum[k] = redM[wave][lane][VEC + k]; }
            lt_m_lt<T, SD>(li, lc, msum, out);
#pragma unroll
            for (int a2 = 0; a2 < SD; ++a2)
#pragma unroll
              for (int c = 0; c < SD; ++c) atomic_add(&pa.s.accS[size_t(a2 * SD + c) * n + i0 + lane], -out[a2][c]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      r += chunk;
      rem -= chunk;
    }
    // column side of this block: combine the wavefronts, then 256-B contiguous atomics per accumulator row
#pragma unroll
    for (int f = 0; f < NV; ++f)
#pragma unroll
      for (int k = 0; k < kPSW; ++k) colS[wave][f * kPSW + k][lane] = accv[f][k];
    if constexpr (SD > 0) {
#pragma unroll
      for (int a2 = 0; a2 < SD; ++a2)
#pragma unroll
        for (int c = 0; c < SD; ++c) colS[wave][VEC + a2 * SD + c][lane] = accS[a2][c];
    }
    __syncthreads();
    for (int t = wave; t < VEC + (SD > 0 ? SD * SD : 0); t += NW) {   // wave-uniform t
      T sum = colS[0][t][lane];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) sum += colS[wv][t][lane];
      T* dst = nullptr;
      if (t < VEC) {
        const int f = t / kPSW, k = t - f * kPSW;
        PVec<T> F = pa.v[0];
#pragma unroll
        for (int g = 1; g < NV; ++g)
          if (f == g) F = pa.v[g];
        if (k < F.m) dst = F.acc + size_t(k) * n;
        else if (k == kPSW - 1 && F.kind == MM_EUCLIDEAN) dst = F.acc + size_t(kPMP) * n;
      } else {
        if constexpr (SD > 0) dst = pa.s.accS + size_t(t - VEC) * n;
      }
      if (dst && jin) atomic_add(dst + j, sum);
    }
    ++cb;
    r = row_begin;
    rem -= shares.cross;   // (entering the next block is paid for)
    if (rem > 0) __syncthreads();
  }
  // loss and scale-gradient partials: slots [1 + nf][kLossSlots]
  const int slot = (blockIdx.x * NW + wave) & (kLossSlots - 1);
  {
    const T v = wave_sum(loss_acc);
    if (lane == 0) atomic_add(&la.slots[slot], v);
  }
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    const T v = wave_sum(dsv[f]);
    if (lane == 0) atomic_add(&la.slots[size_t(1 + pa.v[f].slot) * kLossSlots + slot], v);
  }
  if constexpr (SD > 0) {
    const T v = wave_sum(dss);
    if (lane == 0) atomic_add(&la.slots[size_t(1 + pa.s.slot) * kLossSlots + slot], v);
  }
}

template <typename T, int NV, int SD>
int launch(int loss_kind, const PArgs<T>& pa, const T* target, int64_t n, int64_t rb, int64_t re, LossArgs<T> la, T* table,
           bool prepared, hipStream_t st) {
  static_assert(PSLayout<NV, SD>::W == (NV * kPSW + (SD > 0 ? SD * (SD + 1) : 0) + 3) / 4 * 4, "product_sym_table_width");
  hipError_t e;
  if (!prepared) {
    product_sym_prep_kernel<T, NV, SD><<<dim3(unsigned((n + 1 + 127) / 128)), dim3(128), 0, st>>>(pa, int(n), table);
    e = hipGetLastError();
    if (e != hipSuccess) return int(e);
  }
  const int64_t units = ColWalk(int(n), int(rb), int(re), 64).total();
  if (units <= 0) return MM_OK;
  const int64_t cus = device_cus();
  // (rows per workgroup as for the other resident-grid kernels: >= ~32 rows of a column block each, whole multiples of the CU
  // count, at most three workgroups per CU)
  int64_t grid = std::min<int64_t>(3 * cus, std::max<int64_t>(cus, units / 32 / cus * cus));
  static const int64_t env_grid = std::getenv("MM_PRODUCT_SYM_GRID") ? std::atoll(std::getenv("MM_PRODUCT_SYM_GRID")) : 0;
  if (env_grid > 0) grid = env_grid;
  grid = std::max<int64_t>(1, std::min<int64_t>(grid, (units + 3) / 4));
  const dim3 g3{unsigned(grid), 1, 1}, b3{unsigned(64 * kPSWaves), 1, 1};
  // rows a share pays for entering a column block (ColWalk::enter; as the SPD backward: spd_pair.hpp) — MM_PRODUCT_SYM_CROSS overrides
  static const int env_cross = std::getenv("MM_PRODUCT_SYM_CROSS") ? std::atoi(std::getenv("MM_PRODUCT_SYM_CROSS")) : (sizeof(T) == 4 ? 16 : 8);
  const ColWalk hw(int(n), int(rb), int(re), 64);
  const int cross = std::min(std::max(env_cross, 0), 1024);
  int kinds_code = 0;
  for (int f = 0; f < NV; ++f) kinds_code |= (pa.v[f].kind & 3) << (2 * f);
  static const bool rt_kinds = [] { const char* e = std::getenv("MM_PRODUCT_RT_KINDS"); return e && e[0] == '1'; }();   // (A/B)
  auto with_kinds = [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (loss_kind == MM_LOSS_STRESS)
      product_sym_kernel<T, NV, SD, MM_LOSS_STRESS, K><<<g3, b3, 0, st>>>(pa, table, target, int(n), int(rb), int(re), la, WalkShares(hw.total_aug(cross), grid, cross));
    else
      product_sym_kernel<T, NV, SD, MM_LOSS_QUOTIENT, K><<<g3, b3, 0, st>>>(pa, table, target, int(n), int(rb), int(re), la, WalkShares(hw.total_aug(cross), grid, cross));
  };
  bool launched = false;
  if constexpr (NV == 1 || NV == 2) launched = !rt_kinds && for_kind_code<NV>(kinds_code, with_kinds);
  if (!launched) with_kinds(std::integral_constant<int, -1>{});
  e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

}  // namespace

template <typename T> bool product_sym_applies(int nv, int sd, const PArgs<T>& pa, int64_t n) {
  // Where it pays (training step of H^5 x S^5 x SPD(2), ordered / symmetric, us — profiles/r03_experiments.md §17): fp32
  // n = 1025 32.4 / 37.7, 1600 43.9 / 42.4, 2000 61.0 / 48.8, 5000 242 / 149; fp64 n = 300 30.4 / 39.5, 700 48.0 / 46.0,
  // 1025 94.4 / 64.8, 5000 1185 / 511: a small launch is latency, and the symmetric form has one wavefront per SIMD there.
  // MM_PRODUCT_ORDERED=1 / MM_PRODUCT_SYM=1 force either.
  static const bool ordered = [] { const char* e = std::getenv("MM_PRODUCT_ORDERED"); return e && e[0] == '1'; }();
  static const bool forced = [] { const char* e = std::getenv("MM_PRODUCT_SYM"); return e && e[0] == '1'; }();
  if (ordered || pa.idx || pa.dense || n > kSpdMaxNodes || (sd != 0 && sd != 2 && sd != 3) || nv < 0 || nv > kPMaxVec ||
      (nv == 0 && sd == 0))
    return false;
  if (!forced && n < (sizeof(T) == 8 ? 640 : 1536)) return false;
  for (int f = 0; f < nv; ++f) {
    const int m = pa.v[f].m;
    if (m > kPSW || (pa.v[f].kind == MM_EUCLIDEAN && m > kPSW - 1)) return false;
  }
  return true;
}
template bool product_sym_applies<float>(int, int, const PArgs<float>&, int64_t);
template bool product_sym_applies<double>(int, int, const PArgs<double>&, int64_t);

template <typename T>
int product_sym_pairs(int loss_kind, int nv, int sd, const PArgs<T>& pa, const T* target, int64_t n, int64_t rb, int64_t re,
                      LossArgs<T> la, T* table, bool prepared, hipStream_t st) {
  if (!product_sym_applies<T>(nv, sd, pa, n)) return MM_ERR_UNSUPPORTED;
#define MM_PS(NV_, SD_) return launch<T, NV_, SD_>(loss_kind, pa, target, n, rb, re, la, table, prepared, st)
  switch (nv * 4 + sd) {
    case 0 * 4 + 2: MM_PS(0, 2);
    case 0 * 4 + 3: MM_PS(0, 3);
    case 1 * 4 + 0: MM_PS(1, 0);
    case 1 * 4 + 2: MM_PS(1, 2);
    case 1 * 4 + 3: MM_PS(1, 3);
    case 2 * 4 + 0: MM_PS(2, 0);
    case 2 * 4 + 2: MM_PS(2, 2);
    case 2 * 4 + 3: MM_PS(2, 3);
    case 3 * 4 + 0: MM_PS(3, 0);
    case 3 * 4 + 2: MM_PS(3, 2);
    case 3 * 4 + 3: MM_PS(3, 3);
    default: return MM_ERR_UNSUPPORTED;
  }
#undef MM_PS
}

template int product_sym_pairs<float>(int, int, int, const PArgs<float>&, const float*, int64_t, int64_t, int64_t, LossArgs<float>,
                                      float*, bool, hipStream_t);
template int product_sym_pairs<double>(int, int, int, const PArgs<double>&, const double*, int64_t, int64_t, int64_t,
                                       LossArgs<double>, double*, bool, hipStream_t);

}  // namespace mm
