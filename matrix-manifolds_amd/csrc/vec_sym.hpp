// Backward of the vector-manifold pdist, SYMMETRIC VALU form (round 3): every UNORDERED pair is visited once and feeds
// both of its points — the structure of the SPD backward (spd.hip) applied to Euclidean / Lorentz / sphere:
//   * lanes own consecutive columns j (NC per lane) and keep x_j and their column sums in registers; the row point x_i is
//     wave-uniform: scalar loads from a zero-padded copy of the points (workspace; [n+1][MP], written by the preparation
//     kernel that also clears the accumulators), one row ahead, into alternating SGPR sets;
//   * per pair: q = <x_i, x_j> (MP multiply-adds with a scalar operand), w = g dout/dq(q) (lorentz.py:72-77,134-138;
//     sphere.py:68-74; base.py:29-33,56-57) or the fused objective's upstream gradient (loss.hpp), column side
//     acc_j += w x_i (MP multiply-adds, scalar operand), row side r = sum over the lane's columns of w x_j (MP);
//   * the row side's MP (+1: sum of w, Euclidean) values are summed across the wavefront by the transposing reduction
//     of smallmat.hpp — one total per lane — and leave per slice with one atomic instruction per 64 values;
//   * ONE resident grid with statically balanced shares of the column walk (spd_ws.hpp, ColWalk), the upstream
//     gradients requested two rows ahead with running scalar offsets, priority lowered along the share.
// Both sides are flushed straight into the gradient, as atomics into contiguous ranges of grad [n][m], the manifold's linear
// map applied on the way (Lorentz: -J; sphere: identity; Euclidean: 2 (x sum w - sum w x_other), x from the padded copy):
// two launches per backward — preparation (padded points, cleared gradient) and pair kernel — and no accumulators.
// Measured (MI355X, profiles/r03_experiments.md §11, §11b): Lorentz(11) n = 4039 fp32 22.9 us in the direct form (31.3 us for
// the matrix-core backward of vec_gram.hip; the accumulator form was 34.0 us + a 4.7-us finalize), 79 vector / 120 total
// instructions per 64 pairs, 0.18 of the HBM peak; fp64 64 us against 94 us.  The default up to m = 16 in both precisions;
// the matrix cores keep fp32 17 <= m <= 32 and the fp32 squared Euclidean distance.  Elsewhere it replaces the ordered-pair
// kernel of vec.hip (2.3 x).
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <type_traits>

#include "loss.hpp"
#include "smallmat.hpp"
#include "spd_ws.hpp"
#include "vecfn.hpp"

namespace mm {

// columns per lane: two where the registers allow it (fp32, MP <= 16)
#ifndef MM_VSYM_NC_SMALL
#define MM_VSYM_NC_SMALL 2   // columns per lane, fp32 m <= 12 (A/B builds: 3, 4)
#endif
template <typename T, int MP> constexpr int vsym_cols() {
  return (sizeof(T) == 4 && MP <= 12) ? MM_VSYM_NC_SMALL : ((sizeof(T) == 4 && MP <= 16) ? 2 : 1);
}
template <typename T, int MP> constexpr int vsym_min_waves() {
  return (sizeof(T) == 4 && MP <= 16) ? (vsym_cols<T, MP>() >= 4 ? 3 : (vsym_cols<T, MP>() == 3 ? 3 : 4)) : (sizeof(T) == 4 ? 2 : 1);
}
#ifndef MM_VSYM_AHEAD
#define MM_VSYM_AHEAD 2   // rows of the pair vector requested ahead (measured, Lorentz(11) n = 4039: 2 -> 34.0 us, 4 -> 36.7 us)
#endif
constexpr int kVSymWaves = 4;
constexpr int kVSymTI = 16;   // rows per wavefront and chunk

// Zero-padded copy of the points ([n + 1][MP]: row n is padding for the row operand requested one row ahead) and clean
// accumulators — one launch in place of the memset of the ordered-pair kernel.
template <typename T, int MP>
__global__ void vec_sym_prep_kernel(const T* __restrict__ x, int n, int m, T* __restrict__ xpad, T* __restrict__ acc,
                                    int acc_count, T* __restrict__ grad /* null: not flushed into directly */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = gridDim.x * blockDim.x;
  for (int e = t; e < (n + 1) * MP; e += stride) {
    const int i = e / MP, k = e - i * MP;
    xpad[e] = (i < n && k < m) ? x[size_t(i) * m + k] : T(0);
  }
  for (int e = t; e < acc_count; e += stride) acc[e] = T(0);
  if (grad)
    for (int e = t; e < n * m; e += stride) grad[e] = T(0);
}

template <typename T, int KIND, int MP, int LOSS, bool SQ>
__global__ __launch_bounds__((64 * kVSymWaves), (vsym_min_waves<T, MP>()))
void vec_pdist_bwd_sym_kernel(const T* __restrict__ xpad /* [n+1][MP] */, const T* __restrict__ g, int n, int m, int row_begin,
                              int row_end, T* __restrict__ acc /* the gradient [n][m] (training-step form: the workspace's gacc) */,
                              LossArgs<T> la, WalkShares shares /* the workgroups' units of the walk, cut on the host */,
                              int* __restrict__ share_tab /* where they start, remembered in the workspace (WalkShares::of_cached) */) {
  __builtin_amdgcn_s_setprio(3);   // (first thing: a wavefront starts at 0 and its prologue would be served after the older workgroups' row loops)
  constexpr int NW = kVSymWaves, TI = kVSymTI;
  constexpr int NC = vsym_cols<T, MP>();
  constexpr bool kEuclid = KIND == MM_EUCLIDEAN;
  // The sums are the gradient up to the manifold's linear map (d q / d x_j = -J x_i, lorentz.py:72-77,101-118; x_i,
  // sphere.py:68-74; 2 (x_j - x_i), base.py:29-33,56-57: 2 (x_j sum w - sum w x_i)), so both sides leave straight into
  // grad [n][m] — contiguous ranges of it, the map applied on the way — and no finalize launch follows.
  constexpr int NR = MP + (kEuclid ? 1 : 0);   // values of the row-side reduction / of a column's sums
  constexpr int squared = SQ ? 1 : 0;
  T sp = T(1), loss_acc = T(0), ds_acc = T(0);
  loss_resolve<T, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  static_assert(TI % MM_VSYM_AHEAD == 0, "the row loop is unrolled kAhead times");
  __shared__ T redM[NW][TI][NR];
  __shared__ T colS[NW][NC][NR][64];
  __shared__ T redJunk[NW][64];
  const ColWalk walk(n, row_begin, row_end, 64 * NC);
  // this workgroup's share, cut on the host: the block and row it starts at and its budget of units — one per row,
  // shares.cross per block entered (spd_ws.hpp, WalkShares / ColWalk::enter)
  int cb, r, rem;
  shares.of_cached(walk, int(blockIdx.x), share_tab, cb, r, rem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bool red_writer;
  const int red_slot = reduce_slot<NR>(lane, red_writer);
  const int64_t base = pair_off(n, row_begin);
  // (priority outranks age in the vector-issue arbiter: whoever is ahead in its share yields — spd.hip)
  const int wave_rows = (rem + NW - 1) / NW;
  // (three equal stretches at 3, 2, 1 and a last one at 0 of a tenth of the rows, at least four: spd_pair.hpp)
  const int prio_last = max(4, wave_rows / 10);
  const int prio_stretch = max(wave_rows - prio_last, 3) / 3;
  int rows_left = prio_stretch;
  int phase = 0;
  T* red_ptr = red_writer ? &redM[wave][0][red_slot] : &redJunk[wave][lane];
  const int red_step = red_writer ? NR : 0;

  while (rem > 0) {   // one pass per column block of this workgroup's share
    const int jbase = cb * (64 * NC);
    int jv[NC];
    unsigned joff[NC];
    T xj[NC][MP], accJ[NC][NR];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int j = jbase + 64 * q + lane;
      const bool jin = j < n;
      jv[q] = jin ? j : INT32_MIN;
      asm volatile("" : "+v"(jv[q]));
      joff[q] = unsigned(min(j, n - 1)) * unsigned(sizeof(T));
      const T* xp = xpad + size_t(min(j, n)) * MP;   // (row n: the zero padding row)
#pragma unroll
      for (int k = 0; k < MP; ++k) xj[q][k] = xp[k];
#pragma unroll
      for (int k = 0; k < NR; ++k) accJ[q][k] = T(0);
    }
    const int hi = walk.hi(cb);
    while (rem > 0 && r < hi) {
      const int chunk = min(min(hi - r, NW * TI), rem);
      const int tw = (chunk + NW - 1) / NW;
      const int i0 = r + wave * tw, i1 = min(i0 + tw, r + chunk);
      if (i0 < i1) {
        constexpr int kAhead = MM_VSYM_AHEAD;
        unsigned roff = unsigned(i0) * unsigned(MP * sizeof(T));   // byte offset of the row's point in the padded table
        T xrow[2][MP];
#pragma unroll
        for (int k = 0; k < MP; ++k) xrow[0][k] = xpad[size_t(i0) * MP + k];
        const int glast = min(i1, walk.re) - 1;
        const int64_t gk = glast - i0;
        const unsigned gmax = unsigned((gk * (n - 2) - (int64_t(i0) * gk + gk * (gk - 1) / 2)) * int64_t(sizeof(T)));
        const char* gslice = reinterpret_cast<const char*>(g + (pair_off(n, i0) - base - i0 - 1));
        unsigned goff = 0, gstep = unsigned(n - i0 - 2) * unsigned(sizeof(T));
        unsigned jslice[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) jslice[q] = max(joff[q], unsigned(i0 + 1) * unsigned(sizeof(T)));
        T gq[kAhead][NC];
        auto request = [&](T (&dst)[NC]) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < NC; ++q) {
            asm volatile("" : "+v"(jslice[q]));
            dst[q] = *reinterpret_cast<const T*>(gslice + goff + jslice[q]);
          }
          goff = min(goff + gstep, gmax);
          gstep -= unsigned(sizeof(T));
        };
#pragma unroll
        for (int u = 0; u < kAhead; ++u) request(gq[u]);
        for (int ib = i0; ib < i1; ib += kAhead) {
#pragma unroll
          for (int u = 0; u < kAhead; ++u) {
            const int irow = ib + u;
            const int ieff = (u == 0 || irow < i1) ? irow : INT32_MAX;   // (slots past the slice's last row are masked)
            const T (&xi)[MP] = xrow[u & 1];
            roff = min(roff + unsigned(MP * sizeof(T)), unsigned(n) * unsigned(MP * sizeof(T)));   // (never past the padding row n)
            asm volatile("" : "+s"(roff));
            const T* rowp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(xpad) + roff);
#pragma unroll
            for (int k = 0; k < MP; ++k) xrow[(u & 1) ^ 1][k] = rowp[k];
            if (__builtin_expect(--rows_left == 0, 0)) {
              ++phase;
              if (phase == 1) { __builtin_amdgcn_s_setprio(2); rows_left = prio_stretch; }
              else if (phase == 2) { __builtin_amdgcn_s_setprio(1); rows_left = max(wave_rows - prio_last - 2 * prio_stretch, 1); }
              else { __builtin_amdgcn_s_setprio(0); rows_left = INT32_MAX; }
            }
            T rsum[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) rsum[k] = T(0);
            T loaded[NC];
            bool valid[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
              valid[q] = jv[q] > ieff;
              loaded[q] = valid[q] ? gq[u][q] : T(0);
            }
            request(gq[u]);
#pragma unroll
            for (int q = 0; q < NC; ++q) {
              T qv = T(0);
              if constexpr (kEuclid) {
#pragma unroll
                for (int k = 0; k < MP; ++k) { const T df = xj[q][k] - xi[k]; qv = Num<T>::fma(df, df, qv); }
              } else if constexpr (KIND == MM_LORENTZ) {
#pragma unroll
                for (int k = 1; k < MP; ++k) qv = Num<T>::fma(xi[k], xj[q][k], qv);
                qv = Num<T>::fma(xi[0], xj[q][0], -qv);
              } else {
#pragma unroll
                for (int k = 0; k < MP; ++k) qv = Num<T>::fma(xi[k], xj[q][k], qv);
              }
              T w;
              if constexpr (LOSS == MM_LOSS_NONE) {
                w = loaded[q] * PairFn<T, KIND>::dq(qv, squared);   // (invalid pairs: loaded = 0; dq is finite)
                w = valid[q] ? w : T(0);
              } else {
                const T d2 = PairFn<T, KIND>::value(qv, 1);
                T dldm;
                const T l = loss_term<T, LOSS>(sp * d2, valid[q] ? loaded[q] : T(1), la, dldm);
                loss_acc += valid[q] ? l : T(0);
                ds_acc += valid[q] ? dldm * d2 : T(0);
                w = valid[q] ? dldm * sp * PairFn<T, KIND>::dq(qv, 1) : T(0);
              }
#pragma unroll
              for (int k = 0; k < MP; ++k) {
                accJ[q][k] = Num<T>::fma(w, xi[k], accJ[q][k]);       // column side: w x_i (scalar operand)
                rsum[k] = Num<T>::fma(w, xj[q][k], rsum[k]);          // row side: w x_j
              }
              if constexpr (kEuclid) { accJ[q][MP] += w; rsum[MP] += w; }
            }
            *red_ptr = wave_reduce_transposed<NR, T>(rsum, lane);
            red_ptr += red_step;
          }
        }
        red_ptr -= red_step * ((i1 - i0 + kAhead - 1) / kAhead * kAhead);
        __builtin_amdgcn_wave_barrier();
        {
          const int cnt = (i1 - i0) * m;             // rows i0 .. i1 - 1 of grad: one contiguous range
          T* gp = acc + size_t(i0) * m;
          for (int t = lane; t < cnt; t += 64) {
            const int il = t / m, k = t - il * m;
            T v = redM[wave][il][k];
            if (KIND == MM_LORENTZ && k != 0) v = -v;
            if constexpr (kEuclid) v = T(2) * Num<T>::fma(xpad[size_t(i0 + il) * MP + k], redM[wave][il][MP], -v);
            atomic_add(&gp[t], v);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      r += chunk;
      rem -= chunk;
    }
    // column side of this block: combine the wavefronts, then 256-B contiguous atomics per coordinate
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int k = 0; k < NR; ++k) colS[wave][q][k][lane] = accJ[q][k];
    __syncthreads();
    {
      const int cols = min(64 * NC, n - jbase);       // columns jbase .. of grad: one contiguous range of cols * m values
      T* gp = acc + size_t(jbase) * m;
      for (int t = threadIdx.x; t < cols * m; t += 64 * NW) {
        const int jl = t / m, k = t - jl * m;
        const int q = jl >> 6, l = jl & 63;
        T sum = colS[0][q][k][l];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) sum += colS[wv][q][k][l];
        if (KIND == MM_LORENTZ && k != 0) sum = -sum;
        if constexpr (kEuclid) {
          T wsum = colS[0][q][MP][l];
#pragma unroll
          for (int wv = 1; wv < NW; ++wv) wsum += colS[wv][q][MP][l];
          sum = T(2) * Num<T>::fma(xpad[size_t(jbase + jl) * MP + k], wsum, -sum);
        }
        atomic_add(&gp[t], sum);
      }
    }
    ++cb;
    r = row_begin;
    rem -= shares.cross;   // (entering the next block is paid for)
    if (rem > 0) __syncthreads();
  }
  if constexpr (LOSS != MM_LOSS_NONE) {
    __shared__ T lossW[NW][2];
    const T l = wave_sum(loss_acc), d = wave_sum(ds_acc);
    if (lane == 0) { lossW[wave][0] = l; lossW[wave][1] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
      T ls = lossW[0][0], dd = lossW[0][1];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) { ls += lossW[wv][0]; dd += lossW[wv][1]; }
      const int slot = blockIdx.x & (kLossSlots - 1);
      atomic_add(&la.slots[slot], ls);
      atomic_add(&la.slots[kLossSlots + slot], dd);
    }
  }
}

}  // namespace mm
