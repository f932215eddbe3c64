// The fused mixed-manifold pair kernel for PRODUCT embeddings (the reference's csphd configuration:
// Lorentz(6) x Sphere(6) x SPD(2), n ~ 1e3; ManifoldEmbedding.compute_dists, modules.py:84-88, with the
// objective of objectives.py:16-45 and what loss.backward() computes, train.py:213-217).
//
// One launch evaluates, for every pair, the squared distance of EVERY factor, their softplus-weighted sum,
// the loss term and its derivative, and accumulates the gradients of all factors' points and scales.  At the
// sizes of these configurations (5e5 pairs) the arithmetic is microseconds; what the reference — and a
// kernel-per-factor design — pays for is launches (~95 resp. ~15 per step) and serialised memory round
// trips.  So the kernel is organised around latency, not the last flop:
//  * every ORDERED pair is visited: a lane (= column j) accumulates only into its own column, no factor needs
//    a cross-lane reduction; the row point i is wave-uniform and read from LDS, where each wavefront stages
//    its rows once, zero-padded to 16 coordinates (no load sits under a `k < m` branch);
//  * the SPD factor's Cholesky factors are formed in the kernel (no per-node tables / preparation launch),
//    its eigen-decomposition is the Jacobi path;
//  * 4 wavefronts per workgroup share 64 columns and combine their column sums in LDS before the atomics;
//  * the finalize kernel leaves all accumulators zero (MM_WS_CLEAN: the workspace is cleared once, ever);
//  * node minibatches are addressed through an index vector inside the kernels (rows, dense targets, gradient
//    rows): no gather / scatter launches around it.
// Supported per launch: up to 3 vector factors (Euclidean / Lorentz / sphere, m <= 16, kinds chosen at run
// time) and at most one SPD(2) or SPD(3) factor; anything else takes the per-factor kernels around
// mm_product_loss.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>

#include "../../include/mm_manifolds.h"
#include "loss.hpp"
#include "product_args.hpp"
#include "product_step.hpp"
#include "product_sym.hpp"
#include "smallmat.hpp"
#include "spd_rules.hpp"
#include "stamp.hpp"
#include "vec_rules.hpp"
#include "vecfn.hpp"

namespace mm {

// Pins a loaded value (per-lane / wave-uniform) so that the load is issued where it is written: without
// it the compiler sinks each load under the condition that masks its result — one basic block and one
// full memory round trip per element.
__device__ __forceinline__ void pin_v(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_v(double& v) { asm volatile("" : "+v"(v)); }

template <typename T, int PW> __device__ __forceinline__ T vec_q_rt(int kind, const T (&xi)[PW], const T (&xj)[PW]) {
  using N = Num<T>;
  T q = T(0);
  if (kind == MM_EUCLIDEAN) {
#pragma unroll
    for (int k = 0; k < PW; ++k) { const T df = xj[k] - xi[k]; q = N::fma(df, df, q); }
  } else {
#pragma unroll
    for (int k = 1; k < PW; ++k) q = N::fma(xi[k], xj[k], q);
    q = kind == MM_LORENTZ ? N::fma(xi[0], xj[0], -q) : N::fma(xi[0], xj[0], q);
  }
  return q;
}
template <typename T> __device__ __forceinline__ T vec_value_rt(int kind, T q) {
  return kind == MM_EUCLIDEAN ? PairFn<T, MM_EUCLIDEAN>::value(q, 1)
                              : (kind == MM_LORENTZ ? PairFn<T, MM_LORENTZ>::value(q, 1) : PairFn<T, MM_SPHERE>::value(q, 1));
}
template <typename T> __device__ __forceinline__ T vec_dq_rt(int kind, T q) {
  return kind == MM_EUCLIDEAN ? PairFn<T, MM_EUCLIDEAN>::dq(q, 1)
                              : (kind == MM_LORENTZ ? PairFn<T, MM_LORENTZ>::dq(q, 1) : PairFn<T, MM_SPHERE>::dq(q, 1));
}

// PW: padded width of a vector factor inside the kernel — 8 when every vector factor has at most 8 coordinates (csphd:
// 6 + 6), else kPMP = 16: the row point's LDS reads, the inner product and the column accumulation are PW wide.
// IDX: a node minibatch — columns and rows address the factors' points, the dense targets and the accumulators through
// pa.idx.  A template argument, not `pa.idx ? pa.idx[j] : j` in front of every dependent load: that form put a conditional
// index load and an UNconditional s_waitcnt vmcnt(0) in front of each of the six groups of prologue loads — with or without
// an index vector, every workgroup started with six serial memory round trips (9 k of its 27 k cycles; round 4).  Now the
// node ids (this lane's column, and the tile's rows — lane r holds row i0 + r) are one request, everything else follows
// in one batch, and the rows' ids reach the lanes that stage them through ds_bpermute / v_readlane.
template <typename T, int NV, int SD, int LOSS, int PW, bool IDX, int KC = -1>
__global__ __launch_bounds__(kPCols * kPWaves) void product_pair_kernel(PArgs<T> pa, const T* __restrict__ target, int n,
                                                               int row_begin, int row_end, int ti, LossArgs<T> la) {
  constexpr int NPS = SD > 0 ? Packed<(SD > 0 ? SD : 2)>::NP : 1;
  constexpr int DS = SD > 0 ? SD : 2;
  MM_PSTAMP_BEGIN();
  loss_resolve<T, LOSS>(la);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware tile order: the eight XCDs have private L2s and workgroups are dealt to them round-robin, so the workgroups
  // that share an XCD (orig % 8: a group label, not the XCD's id) take a CONTIGUOUS run of tiles in row-major order — a band
  // of row tiles — and the band's targets (its rows of the pair vector, and its columns of the rows above) are fetched into
  // one L2 instead of all eight (bijective for any grid size).  -DMM_PRODUCT_NO_XCD: the plain order (A/B builds).
#ifdef MM_PRODUCT_NO_XCD
  const unsigned bx = blockIdx.x, by = blockIdx.y;
#else
  const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
  const unsigned bx = tile % gridDim.x, by = tile / gridDim.x;
#endif
  const int j = bx * kPCols + lane;
  const int i0 = min(int((by * kPWaves + wave) * ti), n), i1 = min(i0 + ti, n);
  const bool jin = j < n;
  const bool jown = jin && j >= row_begin && j < row_end;
  int64_t jnode = jin ? j : n - 1;  // lanes past n: clamped, masked later
  int rnode = min(i0 + min(lane, ti - 1), n - 1);      // lane r < ti: the node of row i0 + r
  if constexpr (IDX) {
    // (clamped into the tables: the index vector is unvalidated caller data — spd_pair.hpp, batch_node)
    jnode = int64_t(min(uint64_t(pa.idx[jnode]), uint64_t(pa.dense_n - 1)));
    rnode = int(min(uint64_t(pa.idx[rnode]), uint64_t(pa.dense_n - 1)));
  }
  auto node_of_row = [&](int r) -> int {       // r per lane
    if constexpr (IDX) return __shfl(rnode, r); else return min(i0 + r, n - 1);
  };
  auto node_of_row_uniform = [&](int r) -> int {   // r wave-uniform
    if constexpr (IDX) return __builtin_amdgcn_readlane(rnode, r); else return min(i0 + r, n - 1);
  };
  // per-lane column data
  T xj[NV > 0 ? NV : 1][PW], accv[NV > 0 ? NV : 1][PW], wsum[NV > 0 ? NV : 1], spv[NV > 0 ? NV : 1],
      dsv[NV > 0 ? NV : 1];
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    const PVec<T>& F = pa.v[f];
#pragma unroll
    for (int k = 0; k < PW; ++k) xj[f][k] = F.x[size_t(jnode) * F.m + min(k, F.m - 1)];  // clamped, masked below
#pragma unroll
    for (int k = 0; k < PW; ++k) accv[f][k] = T(0);
    wsum[f] = T(0);
    dsv[f] = T(0);
    spv[f] = *F.scale_raw;
  }
  T sps = T(1), dss = T(0);
  if constexpr (SD > 0) sps = *pa.s.scale_raw;

  // SPD factor: the column point AND this lane's row point (lanes < ti hold one row of the tile each) are requested here,
  // with every other load of the prologue; their Cholesky factors are formed once all requests are out (round 4: the row
  // points were requested behind the column point's factorisation — a second memory round trip in every workgroup's life)
  T yj[NPS], accS[DS][DS], xs_col[NPS], xs_row[NPS];
  if constexpr (SD > 0) {
    load_sym_packed<T, SD>(pa.s.x + size_t(jnode) * SD * SD, xs_col);
    load_sym_packed<T, SD>(pa.s.x + size_t(rnode) * SD * SD, xs_row);
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c < SD; ++c) accS[r][c] = T(0);
  }
  const int64_t base = int64_t(row_begin) * (2 * int64_t(n) - row_begin - 1) / 2;
  auto pair_ok = [&](int i) {  // pair (i, j) is stored under row min(i, j)
    return jin && i != j && (i < j ? (i >= row_begin && i < row_end) : jown);
  };
  auto target_at = [&](int i) -> T {  // unconditional load from a clamped address, one row ahead of its use
    // (a node minibatch reads dense[idx[i]][idx[j]]; the two come together: mm_product_pairs_loss_subset requires both)
    if constexpr (IDX) return pa.dense[int64_t(node_of_row_uniform(min(i, n - 1) - min(i0, n - 1))) * pa.dense_n + jnode];
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return target[pair_ok(i) ? int64_t(lo) * (2 * int64_t(n) - lo - 1) / 2 - base + (hi - lo - 1) : int64_t(0)];
  };
  T tnext = target_at(min(i0, n - 1));
  // the row points of the tile, zero-padded to kPMP, staged once (coalesced); a load under `k < m` in the
  // row loop would sit in its own basic block and serialise 16 memory round trips per factor and row
  __shared__ T rowpt[NV > 0 ? NV : 1][kPWaves][kPMaxTI][PW];
  // (round 4: the rows' node ids are taken per lane, node_of(i0 + r), not through an LDS table filled first — that put the
  // row points' loads one LDS round trip and, for a node minibatch, one memory round trip BEHIND the column data's; the
  // workgroup timeline showed 4.7 k + 3.8 k cycles of prologue, two serial latencies: tools/product_timeline.py)
  // (requested unconditionally from clamped addresses for ALL factors first, masked when they are written to LDS further
  // down: as `valid ? F.x[..] : 0` inside the loop, every factor's rows cost one serial memory round trip)
  constexpr int kStageIts = kPMaxTI * PW / 64;
  T staged[NV > 0 ? NV : 1][kStageIts];
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    const PVec<T>& F = pa.v[f];
#pragma unroll
    for (int it = 0; it < kStageIts; ++it) {
      const int e = lane + 64 * it, r = min(e / PW, ti - 1), k = e % PW;
      staged[f][it] = F.x[size_t(node_of_row(r)) * F.m + min(k, F.m - 1)];
    }
  }
  MM_PSTAMP(1);
  __shared__ T rowL[kPWaves][kPMaxTI][2 * NPS];  // L_i^-1 and L_i of the tile's rows
  if constexpr (SD > 0) {
#pragma unroll
    for (int k = 0; k < NPS; ++k) { pin_v(xs_col[k]); pin_v(xs_row[k]); }
    // (no per-node tables, no preparation launch: the factorisations are ~1 % of a row's arithmetic)
    cholesky<T, SD>(xs_col, yj);
    if (lane < ti) {
      T l[NPS], li_[NPS];
      cholesky<T, SD>(xs_row, l);
      invert_lower<T, SD>(l, li_);
#pragma unroll
      for (int k = 0; k < NPS; ++k) { rowL[wave][lane][k] = li_[k]; rowL[wave][lane][NPS + k] = l[k]; }
    }
  }
  // every load of the preamble is in flight by now; pin them here so that none is sunk below its mask
#pragma unroll
  for (int f = 0; f < NV; ++f) {
#pragma unroll
    for (int it = 0; it < kStageIts; ++it) {
      pin_v(staged[f][it]);
      const int e = lane + 64 * it, r = e / PW, k = e % PW;
      if (e < ti * PW) rowpt[f][wave][r][k] = (k < pa.v[f].m && i0 + r < n) ? staged[f][it] : T(0);
    }
  }
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    pin_v(spv[f]);
#pragma unroll
    for (int k = 0; k < PW; ++k) pin_v(xj[f][k]);
  }
  pin_v(tnext);
#pragma unroll
  for (int f = 0; f < NV; ++f) {
#pragma unroll
    for (int k = 0; k < PW; ++k) xj[f][k] = (jin && k < pa.v[f].m) ? xj[f][k] : T(0);
    spv[f] = softplus_of(&spv[f]);
  }
  if constexpr (SD > 0) sps = softplus_of(&sps);
  __syncthreads();
  MM_PSTAMP(2);
  T loss_acc = T(0);
  for (int i = i0; i < i1; ++i) {  // wave-uniform row
    const bool up = i < j;
    const bool valid = pair_ok(i);
    const T tgt = tnext;
    tnext = target_at(min(i + 1, i1 - 1));
    // ---- phase 1: every factor's squared distance, the weighted sum
    T m = T(0), qv[NV > 0 ? NV : 1], d2v[NV > 0 ? NV : 1], xi[NV > 0 ? NV : 1][PW];
#pragma unroll
    for (int f = 0; f < NV; ++f) {
      const PVec<T>& F = pa.v[f];
#pragma unroll
      for (int k = 0; k < PW; ++k) xi[f][k] = rowpt[f][wave][i - i0][k];  // same address in every lane: LDS broadcast
      qv[f] = vec_q_rt<T, PW>(MM_PKIND(f), xi[f], xj[f]);
      d2v[f] = vec_value_rt<T>(MM_PKIND(f), qv[f]);
      m = Num<T>::fma(spv[f], d2v[f], m);
    }
    T li[NPS], lc[NPS], lw[DS], vv[DS][DS], mlog[NPS], d2s = T(0);
    if constexpr (SD > 0) {
#pragma unroll
      for (int k = 0; k < NPS; ++k) { li[k] = rowL[wave][i - i0][k]; lc[k] = rowL[wave][i - i0][NPS + k]; }
      T a[NPS];
      congr_chol<T, SD>(li, yj, a);
      T s = T(0);
      if constexpr (SD == 2) {
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
    const T l = loss_term<T, LOSS>(m, valid ? tgt : T(1), la, dldm);
    const bool once = valid && up;
    loss_acc += once ? l : T(0);
    const T coef = valid ? dldm : T(0);
    // ---- phase 2: gradients, into this lane's column only
#pragma unroll
    for (int f = 0; f < NV; ++f) {
      dsv[f] += once ? dldm * d2v[f] : T(0);
      const T w = coef * spv[f] * vec_dq_rt<T>(MM_PKIND(f), qv[f]);
      wsum[f] += w;
#pragma unroll
      for (int k = 0; k < PW; ++k) accv[f][k] = Num<T>::fma(w, xi[f][k], accv[f][k]);
    }
    if constexpr (SD > 0) {
      dss += once ? dldm * d2s : T(0);
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
      for (int r = 0; r < SD; ++r)
#pragma unroll
        for (int c = 0; c < SD; ++c) accS[r][c] += cj[r][c];
    }
  }
  MM_PSTAMP(3);
  // ---- flush: the workgroup's wavefronts hold partial sums of the same columns; combine them in LDS, then
  // one coalesced atomic per accumulator row (float atomics cost ~60 ns of CU time per wave instruction)
  constexpr int kRedRows = (PW + 1 > DS * DS) ? PW + 1 : DS * DS;
  __shared__ T red[kPWaves][kRedRows][64];
  auto combine_and_add = [&](int rows, auto&& dst_of) {  // red[w][k][lane], k < rows -> atomics on dst_of(k)
    __syncthreads();
    for (int k = wave; k < rows; k += kPWaves) {  // wave-uniform k
      T* dst = dst_of(k);
      if (!dst) continue;
      T sum = red[0][k][lane];
#pragma unroll
      for (int w = 1; w < kPWaves; ++w) sum += red[w][k][lane];
      if (jin) atomic_add(dst + j, sum);
    }
    __syncthreads();
  };
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    const PVec<T>& F = pa.v[f];
#pragma unroll
    for (int k = 0; k < PW; ++k) red[wave][k][lane] = accv[f][k];
    red[wave][PW][lane] = wsum[f];
    combine_and_add(PW + 1, [&](int k) -> T* {   // (the sum of the weights lives in accumulator row kPMP whatever PW is)
      return k < F.m ? F.acc + size_t(k) * n : ((k == PW && F.kind == MM_EUCLIDEAN) ? F.acc + size_t(kPMP) * n : nullptr);
    });
  }
  if constexpr (SD > 0) {
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c < SD; ++c) red[wave][r * SD + c][lane] = accS[r][c];
    combine_and_add(SD * SD, [&](int k) -> T* { return pa.s.accS + size_t(k) * n; });
  }
  MM_PSTAMP(4);
  // loss and scale-gradient partials: slots [1 + nf][kLossSlots].  ONE transposing reduction for all of them (round 4: four
  // butterfly sums through ds_bpermute, one after the other, were 2.5 k cycles of every workgroup's 29 k)
  const int slot = (bx + (by * kPWaves + wave) * gridDim.x) & (kLossSlots - 1);
  {
    constexpr int NS = 1 + NV + (SD > 0 ? 1 : 0);
    T sums[NS];
    sums[0] = loss_acc;
#pragma unroll
    for (int f = 0; f < NV; ++f) sums[1 + f] = dsv[f];
    if constexpr (SD > 0) sums[1 + NV] = dss;
    bool writer;
    const int which = reduce_slot<NS>(lane, writer);
    const T total = wave_reduce_transposed<NS, T>(sums, lane);
    if (writer) {
      int dst = 0;   // the caller's slot of sum `which`: 0 = loss, 1 + k = d loss / d scale of factor k
#pragma unroll
      for (int f = 0; f < NV; ++f)
        if (which == 1 + f) dst = 1 + pa.v[f].slot;
      if (SD > 0 && which == 1 + NV) dst = 1 + pa.s.slot;
      atomic_add(&la.slots[size_t(dst) * kLossSlots + slot], total);
    }
  }
  MM_PSTAMP(5);
}

// Accumulators -> gradients (the manifold-specific map of each factor) and the loss sums.
// blockIdx.y selects the job: vector factor y (one thread per coordinate), the SPD factor (one thread per
// node), or the loss / scale-gradient sums (one block per output).
template <typename T, int NV, int SD>
__global__ __launch_bounds__(256) void product_pair_finalize_kernel(PArgs<T> pa, int n, T* __restrict__ slots,
                                                                    T* __restrict__ loss_out) {
  constexpr int NPS = SD > 0 ? Packed<(SD > 0 ? SD : 2)>::NP : 1;
  const int job = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (job == NV + (SD > 0 ? 1 : 0)) {
    const int q = blockIdx.x;  // 0: loss, 1 + k: d loss / d scale_raw of the caller's factor k
    if (q > pa.nf) return;
    __shared__ double part[4];
    static_assert(kLossSlots == 256, "one slot per thread");
    const double v = wave_sum(double(slots[size_t(q) * kLossSlots + threadIdx.x]));
    slots[size_t(q) * kLossSlots + threadIdx.x] = T(0);  // every accumulator is left clean (MM_WS_CLEAN)
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double tot = part[0] + part[1] + part[2] + part[3];
      const T* raw = nullptr;
#pragma unroll
      for (int f = 0; f < NV; ++f)
        if (pa.v[f].slot == q - 1) raw = pa.v[f].scale_raw;
      if (SD > 0 && pa.s.slot == q - 1) raw = pa.s.scale_raw;
      loss_out[q] = q == 0 ? T(tot) : (raw ? T(tot / (1.0 + ::exp(-double(*raw)))) : T(0));  // d softplus = sigmoid
    }
    return;
  }
  if (job < NV) {
    PVec<T> F = pa.v[0];
#pragma unroll
    for (int f = 1; f < NV; ++f)
      if (job == f) F = pa.v[f];
    const int j = t / kPMP, k = t % kPMP;
    if (j >= n || k >= F.m) return;
    const int64_t jn = node_of(pa, j);
    const T s = F.acc[size_t(k) * n + j];
    F.acc[size_t(k) * n + j] = T(0);
    T r = s;
    if (F.kind == MM_EUCLIDEAN) {
      const T ws = F.acc[size_t(kPMP) * n + j];
      r = T(2) * (ws * F.x[size_t(jn) * F.m + k] - s);
      // the 16 threads of a node share a wavefront (64 % kPMP == 0): all of them have read ws by now
      if (k == 0) F.acc[size_t(kPMP) * n + j] = T(0);
    } else if (F.kind == MM_LORENTZ) {
      r = (k == 0) ? s : -s;
    }
    F.grad[size_t(jn) * F.m + k] = r;
    return;
  }
  const int j = t;
  if (j >= n) return;
  if constexpr (SD > 0) {
    T li[NPS], xinv[NPS], sc[SD][SD], gi[NPS];
    {
      T xs[NPS], l[NPS];
      load_sym_packed<T, SD>(pa.s.x + size_t(node_of(pa, j)) * SD * SD, xs);
      cholesky<T, SD>(xs, l);
      invert_lower<T, SD>(l, li);
    }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c < SD; ++c) {
        sc[r][c] = pa.s.accS[size_t(r * SD + c) * n + j];
        pa.s.accS[size_t(r * SD + c) * n + j] = T(0);
      }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T acc = T(0);  // X^-1 = L^-T L^-1
#pragma unroll
        for (int k = r; k < SD; ++k) acc = Num<T>::fma(li[pidx(k, r)], li[pidx(k, c)], acc);
        xinv[pidx(r, c)] = acc;
      }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T a = T(0), b = T(0);
#pragma unroll
        for (int k = 0; k < SD; ++k) {
          a = Num<T>::fma(sc[r][k], xinv[pidx(k, c)], a);
          b = Num<T>::fma(sc[c][k], xinv[pidx(k, r)], b);
        }
        gi[pidx(r, c)] = T(0.5) * (a + b);  // ordered pairs: both roles of a node arrive as "column side"
      }
    store_sym_full<T, SD>(pa.s.grad + size_t(node_of(pa, j)) * SD * SD, gi);
  }
}

// ---- the step kernel of a product embedding (product_step.hpp): product_pair_finalize_kernel's work — sums -> gradients,
// loss record — followed, per point, by the optimizer rule of its factor (vec_rules.hpp / spd_rules.hpp: the arithmetic of
// the per-parameter kernels mm_vec_*_step / mm_spd_*_step) and, per scale, by its momentum-free RSGD update.
// blockIdx.y = job: vector factor y (one thread per node), the SPD factor (one thread per node), the loss / scale sums (one
// block per output).  A training step of the product is then TWO launches.
template <typename T> struct PStep {
  VecRuleArgs<T> v[kPMaxVec]; int vrule[kPMaxVec];
  VecRuleArgs<T> s; int srule;
  T* scale_x[4]; T scale_lr[4], scale_clip[4];   // by the caller's factor position; null: stepped elsewhere / frozen
};
template <typename T, int KIND>
__device__ __forceinline__ void product_vec_rule(int rule, const T (&xp)[kPMP], const T (&g)[kPMP], int64_t j, int m, bool in,
                                                 const VecRuleArgs<T>& R, T beta2, T alpha, T (&o)[kPMP]) {
  if (rule == VRULE_ADAM) pad_rule_point<T, KIND, kPMP, VRULE_ADAM>(xp, g, j, m, in, R, beta2, alpha, o);
  else if (rule == VRULE_MOMENTUM) pad_rule_point<T, KIND, kPMP, VRULE_MOMENTUM>(xp, g, j, m, in, R, beta2, alpha, o);
  else pad_rule_point<T, KIND, kPMP, VRULE_RSGD>(xp, g, j, m, in, R, beta2, alpha, o);
}
template <typename T, int SD>
__global__ __launch_bounds__(128) void product_step_kernel(PArgs<T> pa, PStep<T> ps, int nv, int n, T* __restrict__ slots,
                                                           T* __restrict__ loss_out, T* __restrict__ tab /* node table of the
                                                           symmetric pair kernel, rows of width tabw: the rows of the NEW points
                                                           are written for the next step (null: not kept) */, int tabw) {
  constexpr int NPS = SD > 0 ? Packed<(SD > 0 ? SD : 2)>::NP : 1;
  constexpr int DS = SD > 0 ? SD : 2;
  using N = Num<T>;
  const int job = blockIdx.y;
  if (job == nv + (SD > 0 ? 1 : 0)) {
    const int q = blockIdx.x;  // 0: loss, 1 + k: d loss / d scale_raw of the caller's factor k
    if (q > pa.nf) return;
    __shared__ double part[2];
    static_assert(kLossSlots == 256, "two slots per thread");
    T* sl = slots + size_t(q) * kLossSlots;
    const double v = wave_sum(double(sl[threadIdx.x]) + double(sl[128 + threadIdx.x]));
    sl[threadIdx.x] = T(0);  // every accumulator is left clean (MM_WS_CLEAN)
    sl[128 + threadIdx.x] = T(0);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double tot = part[0] + part[1];
      const T* raw = nullptr;
      for (int f = 0; f < nv; ++f)
        if (pa.v[f].slot == q - 1) raw = pa.v[f].scale_raw;
      if (SD > 0 && pa.s.slot == q - 1) raw = pa.s.scale_raw;
      const T out = q == 0 ? T(tot) : (raw ? T(tot / (1.0 + ::exp(-double(*raw)))) : T(0));  // d softplus = sigmoid
      loss_out[q] = out;
      if (q > 0 && ps.scale_x[q - 1]) {
        // the momentum-free RSGD rule for one scalar (Euclidean(1)): r = g, ||r|| = sqrt(max(g^2, 1e-8)), x' = x - lr clip r
        T scale = -ps.scale_lr[q - 1];
        const T clip = ps.scale_clip[q - 1];
        if (clip > T(0)) scale *= N::min(clip / N::sqrt(N::max(out * out, T(1e-8))), T(1));
        *ps.scale_x[q - 1] = *ps.scale_x[q - 1] + out * scale;
      }
    }
    return;
  }
  const int j0 = blockIdx.x * 128 + threadIdx.x;
  const bool in = j0 < n;
  const int64_t j = in ? j0 : 0;
  if (job < nv) {
    PVec<T> F = pa.v[0];
    VecRuleArgs<T> R = ps.v[0];
    int rule = ps.vrule[0];
#pragma unroll
    for (int f = 1; f < kPMaxVec; ++f)
      if (job == f) { F = pa.v[f]; R = ps.v[f]; rule = ps.vrule[f]; }
    const int m = F.m;
    T xp[kPMP], g[kPMP], o[kPMP];
    load_padded<T, kPMP>(F.x, j, m, xp);
    const T wsum = F.acc[size_t(kPMP) * n + j];
#pragma unroll
    for (int k = 0; k < kPMP; ++k) g[k] = F.acc[size_t(min(k, m - 1)) * n + j];
#pragma unroll
    for (int k = 0; k < kPMP; ++k) {
      T r = g[k];
      if (F.kind == MM_EUCLIDEAN) r = T(2) * (wsum * xp[k] - r);   // sum w 2 (x_j - x_i)
      else if (F.kind == MM_LORENTZ) r = (k == 0) ? r : -r;         // dq / dx_j = -J x_i
      g[k] = k < m ? r : T(0);
    }
    if (in) {
#pragma unroll
      for (int k = 0; k < kPMP; ++k)
        if (k < m) F.acc[size_t(k) * n + j] = T(0);
      if (F.kind == MM_EUCLIDEAN) F.acc[size_t(kPMP) * n + j] = T(0);
      store_row<T, kPMP>(F.grad, j, m, g);
    }
    T beta2 = T(0), alpha = T(0);
    if (rule == VRULE_ADAM) adam_coeffs(R.adam, beta2, alpha);
    if (F.kind == MM_EUCLIDEAN) product_vec_rule<T, MM_EUCLIDEAN>(rule, xp, g, j, m, in, R, beta2, alpha, o);
    else if (F.kind == MM_LORENTZ) product_vec_rule<T, MM_LORENTZ>(rule, xp, g, j, m, in, R, beta2, alpha, o);
    else product_vec_rule<T, MM_SPHERE>(rule, xp, g, j, m, in, R, beta2, alpha, o);
    if (in) {
      store_row<T, kPMP>(const_cast<T*>(F.x), j, m, o);
      if (tab) {   // (product_sym.hip: the factor zero-padded to kPSW coordinates; Euclidean: the last one is 1)
        T* row = tab + size_t(j) * tabw + job * kPSW;
#pragma unroll
        for (int k = 0; k < kPSW; ++k) row[k] = o[k];
        if (F.kind == MM_EUCLIDEAN) row[kPSW - 1] = T(1);
      }
    }
    if (rule == VRULE_ADAM) adam_tick(R.adam.step, R.adam.ticket, gridDim.x);   // (block-uniform)
    return;
  }
  if constexpr (SD > 0) {
    T xs[NPS], li[NPS], xinv[NPS], sc[DS][DS], gi[NPS], o[NPS];
    load_sym_packed<T, SD>(pa.s.x + size_t(j) * SD * SD, xs);
    {
      T l[NPS];
      cholesky<T, SD>(xs, l);
      invert_lower<T, SD>(l, li);
    }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c < SD; ++c) {
        sc[r][c] = pa.s.accS[size_t(r * SD + c) * n + j];
        if (in) pa.s.accS[size_t(r * SD + c) * n + j] = T(0);
      }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T acc = T(0);  // X^-1 = L^-T L^-1
#pragma unroll
        for (int k = r; k < SD; ++k) acc = N::fma(li[pidx(k, r)], li[pidx(k, c)], acc);
        xinv[pidx(r, c)] = acc;
      }
#pragma unroll
    for (int r = 0; r < SD; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        T a = T(0), b = T(0);
#pragma unroll
        for (int k = 0; k < SD; ++k) {
          a = N::fma(sc[r][k], xinv[pidx(k, c)], a);
          b = N::fma(sc[c][k], xinv[pidx(k, r)], b);
        }
        gi[pidx(r, c)] = T(0.5) * (a + b);  // ordered pairs: both roles of a node arrive as "column side"
      }
    if (in) store_sym_full<T, SD>(pa.s.grad + size_t(j) * SD * SD, gi);
    const VecRuleArgs<T>& R = ps.s;
    T* x = const_cast<T*>(pa.s.x);
    if (ps.srule == VRULE_RSGD) {
      spd_rsgd_update<T, SD>(xs, gi, R.lr, R.max_grad_norm, R.exact, o);
    } else if (ps.srule == VRULE_MOMENTUM) {
      T b[NPS];
      load_sym_packed<T, SD>(R.state0 + size_t(j) * SD * SD, b);
      spd_momentum_update<T, SD>(xs, gi, b, R.lr, R.momentum, R.dampening, R.max_grad_norm, R.exact, o);
      if (in) store_sym_full<T, SD>(R.state0 + size_t(j) * SD * SD, b);
    } else {
      T beta2, alpha, mo[NPS];
      adam_coeffs(R.adam, beta2, alpha);
      load_sym_packed<T, SD>(R.state0 + size_t(j) * SD * SD, mo);
      const T v = spd_adam_update<T, SD>(xs, gi, mo, R.state1[size_t(j) * SD * SD], R.adam, beta2, alpha, o);
      if (in) {
        store_sym_full<T, SD>(R.state0 + size_t(j) * SD * SD, mo);
#pragma unroll
        for (int q = 0; q < SD * SD; ++q) R.state1[size_t(j) * SD * SD + q] = v;
      }
    }
    if (in) {
      store_sym_full<T, SD>(x + size_t(j) * SD * SD, o);
      if (tab) {   // L^-1 and L of the new point
        T l[NPS], linv[NPS];
        cholesky<T, SD>(o, l);
        invert_lower<T, SD>(l, linv);
        T* row = tab + size_t(j) * tabw + nv * kPSW;
#pragma unroll
        for (int k = 0; k < NPS; ++k) { row[k] = linv[k]; row[NPS + k] = l[k]; }
      }
    }
    if (ps.srule == VRULE_ADAM) adam_tick(R.adam.step, R.adam.ticket, gridDim.x);
  }
}

inline int64_t device_cus_product() {
  static const int64_t cus = [] {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
    return int64_t(c);
  }();
  return cus;
}

template <typename T, int NV, int SD>
int product_pairs_launch(int loss_kind, PArgs<T> pa, const T* target, int64_t n, int64_t rb, int64_t re, LossArgs<T> la,
                         T* loss_out, hipStream_t st, const PStep<T>* ps, bool pairs_done, T* table) {
  if (!pairs_done && mm_pair_offset(n, re) > mm_pair_offset(n, rb)) {
    // rows per wavefront: short enough that the launch has ~2 wavefronts per SIMD (the work of a small
    // product embedding is latency, not throughput; measured at n = 1025: 2 -> 123, 4 -> 103, 8 -> 94,
    // 16 -> 97 us per training step), at most 32
    const int64_t cols = (n + kPCols - 1) / kPCols;
    static const int ti_env = [] { const char* e = std::getenv("MM_PRODUCT_TI"); return e ? std::atoi(e) : 0; }();
    // Round 4: the launch is as long as its FULLEST compute unit — ceil(workgroups / CUs) workgroups of 4 wavefronts x ti rows —
    // so ti minimises that product, the smaller ti (more wavefronts to hide latency behind) on ties.  n = 1025, 256 CUs:
    // ti = 6 (731 workgroups, 3 x 6 = 18 rows on the fullest CU) 16.6 us, ti = 9 (493, 2 x 9) 17.5, ti = 8 (561, 3 x 8 = 24:
    // the round-3 choice) 18.5, ti = 12 (374, 2 x 12) 20.1 (profiles/r04_experiments.md).  Large n: the old rule's cap.
    int ti = int(std::min<int64_t>(kPMaxTI, std::max<int64_t>(4, cols * n / 2048)));
    if (ti <= 12) {   // (the sizes the rule was measured at: csphd; larger launches are throughput, and have the symmetric form)
      const int64_t cus = device_cus_product();
      int64_t best = INT64_MAX;
      for (int t = 4; t <= kPMaxTI; ++t) {
        const int64_t wgs = cols * ((n + kPWaves * t - 1) / (kPWaves * t));
        const int64_t cost = ((wgs + cus - 1) / cus) * t;
        if (wgs >= cus && cost < best) { best = cost; ti = t; }
      }
    }
    if (ti_env > 0 && ti_env <= kPMaxTI) ti = ti_env;
    const dim3 grid(unsigned(cols), unsigned((n + kPWaves * ti - 1) / (kPWaves * ti)));
    int widest = 0;
    for (int f = 0; f < NV; ++f) widest = std::max(widest, pa.v[f].m);
    static const bool wide_env = [] { const char* e = std::getenv("MM_PRODUCT_PW16"); return e && e[0] == '1'; }();   // (A/B: always 16 wide)
    const dim3 block(kPCols * kPWaves);
    auto launch = [&](auto kernel) { kernel<<<grid, block, 0, st>>>(pa, target, int(n), int(rb), int(re), ti, la); };
    int kinds_code = 0;
    for (int f = 0; f < NV; ++f) kinds_code |= (pa.v[f].kind & 3) << (2 * f);
    static const bool rt_kinds = [] { const char* e = std::getenv("MM_PRODUCT_RT_KINDS"); return e && e[0] == '1'; }();   // (A/B)
    auto by_loss = [&](auto pw, auto idx) {
      constexpr int W = decltype(pw)::value;
      constexpr bool I = decltype(idx)::value;
      auto with_kinds = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if (loss_kind == MM_LOSS_STRESS) launch(product_pair_kernel<T, NV, SD, MM_LOSS_STRESS, W, I, K>);
        else launch(product_pair_kernel<T, NV, SD, MM_LOSS_QUOTIENT, W, I, K>);
      };
      if constexpr (W == 8 && (NV == 1 || NV == 2)) {
        if (!rt_kinds && for_kind_code<NV>(kinds_code, with_kinds)) return;
      }
      with_kinds(std::integral_constant<int, -1>{});
    };
    auto by_idx = [&](auto pw) {
      if (pa.idx) by_loss(pw, std::true_type{}); else by_loss(pw, std::false_type{});
    };
    if (widest <= 8 && !wide_env) by_idx(std::integral_constant<int, 8>{});
    else by_idx(std::integral_constant<int, kPMP>{});
  }
  if (ps) {   // training step: gradients, loss record, optimizer rules and scales in one launch
    const dim3 sgrid(unsigned(std::max<int64_t>((n + 127) / 128, 1 + pa.nf)), unsigned(NV + (SD > 0 ? 1 : 0) + 1));
    // (the symmetric pair kernel's node table follows the points: its next launch skips the preparation — MM_WS_PREPARED)
    T* tab = product_sym_applies<T>(NV, SD, pa, n) ? table : static_cast<T*>(nullptr);
    product_step_kernel<T, SD><<<sgrid, dim3(128), 0, st>>>(pa, *ps, NV, int(n), la.slots, loss_out, tab,
                                                            product_sym_table_width(NV, SD));
  } else {
    const dim3 fgrid(unsigned(std::max<int64_t>((n * kPMP + 255) / 256, 8)), unsigned(NV + (SD > 0 ? 1 : 0) + 1));
    product_pair_finalize_kernel<T, NV, SD><<<fgrid, dim3(256), 0, st>>>(pa, int(n), la.slots, loss_out);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

template <typename T>
int product_pairs_t(int loss_kind, int nf, const int* kinds, const int* dims, const void* const* xs,
                    const void* const* scale_raw, const void* target, int64_t n, int64_t rb, int64_t re, double alpha,
                    double eps, int terms, const double* loss_params, double wmin, double wmax, void* const* grads, void* loss_out, void* wsp,
                    int flags, hipStream_t st, const int64_t* idx = nullptr, const void* dense = nullptr,
                    int64_t dense_n = 0, const mm_train_step* step = nullptr, bool* scale_stepped = nullptr) {
  PArgs<T> pa{};
  PStep<T> ps{};
  auto rule_of = [](const mm_step_param& p, VecRuleArgs<T>& r) {
    r = VecRuleArgs<T>{T(p.lr), T(p.momentum), T(p.dampening), T(p.max_grad_norm), p.exact, static_cast<T*>(p.state0),
                       static_cast<T*>(p.state1),
                       AdamArgs<T>{T(p.lr), T(p.beta1), T(p.beta2), T(p.adam_eps), T(p.max_grad_norm), p.nc, p.exact, p.step, p.ticket}};
    return p.optimizer == MM_OPT_RADAM ? int(VRULE_ADAM) : (p.momentum != 0.0 ? int(VRULE_MOMENTUM) : int(VRULE_RSGD));
  };
  pa.nf = nf;
  pa.idx = idx;
  pa.dense = static_cast<const T*>(dense);
  pa.dense_n = dense_n;
  if ((idx == nullptr) != (dense == nullptr)) return MM_ERR_ARG;   // (a node minibatch is an index vector AND the dense targets)
  int nv = 0, sd = 0;
  // workspace: [64 bytes reserved] [slots (1+nf) x 256] [accumulators]
  char* wsb = static_cast<char*>(wsp);
  T* slots = reinterpret_cast<T*>(wsb + 64);
  T* accp = slots + size_t(1 + nf) * kLossSlots;
  size_t acc_elems = 0;
  for (int k = 0; k < nf; ++k) {
    if (kinds[k] == MM_FACTOR_SPD) {
      if (sd != 0 || (dims[k] != 2 && dims[k] != 3)) return MM_ERR_UNSUPPORTED;
      // (eigenvalue clamps that could bind: the per-factor kernels — the gradient here lacks what a binding clamp leaves in
      // the reference's, spd_pair.hpp pair_core)
      if (!(wmin <= 1e-6 && wmax >= 1e6)) return MM_ERR_UNSUPPORTED;
      sd = dims[k];
      pa.s.x = static_cast<const T*>(xs[k]);
      pa.s.scale_raw = static_cast<const T*>(scale_raw[k]);
      pa.s.accS = accp + acc_elems;
      pa.s.wmin = T(wmin);
      pa.s.wmax = T(wmax);
      pa.s.slot = k;
      pa.s.grad = static_cast<T*>(grads[k]);
      if (step) ps.srule = rule_of(step->points[k], ps.s);
      acc_elems += size_t(sd) * sd * n;
    } else {
      if (nv == kPMaxVec || dims[k] < 1 || dims[k] > kPMP || kinds[k] < MM_EUCLIDEAN || kinds[k] > MM_SPHERE)
        return MM_ERR_UNSUPPORTED;
      PVec<T>& F = pa.v[nv++];
      F.x = static_cast<const T*>(xs[k]);
      F.scale_raw = static_cast<const T*>(scale_raw[k]);
      F.acc = accp + acc_elems;
      F.m = dims[k];
      F.kind = kinds[k];
      F.slot = k;
      F.grad = static_cast<T*>(grads[k]);
      if (step) ps.vrule[nv - 1] = rule_of(step->points[k], ps.v[nv - 1]);
      acc_elems += size_t(kPMP + 1) * n;
    }
  }
  if (!(flags & MM_WS_CLEAN)) {
    hipError_t e = hipMemsetAsync(slots, 0, sizeof(T) * (size_t(1 + nf) * kLossSlots + acc_elems), st);
    if (e != hipSuccess) return int(e);
  }
  if (step)
    for (int k = 0; k < nf; ++k) {
      const mm_step_param& q = step->scales[k];
      const bool fuse = q.x && q.optimizer == MM_OPT_RSGD && q.momentum == 0.0;
      ps.scale_x[k] = fuse ? static_cast<T*>(q.x) : nullptr;
      ps.scale_lr[k] = T(q.lr);
      ps.scale_clip[k] = T(q.max_grad_norm);
      if (scale_stepped) scale_stepped[k] = fuse;
    }
  LossArgs<T> la{nullptr, T(alpha), T(eps), terms, slots, loss_params};
  const T* tg = static_cast<const T*>(target);
  T* lo = static_cast<T*>(loss_out);
  const PStep<T>* psp = step ? &ps : nullptr;
  // the symmetric pair kernel (product_sym.hip: every unordered pair once) where it applies — full batches, vector factors
  // of at most 8 coordinates; the ordered kernel below otherwise.  Both fill the same accumulators.
  bool pairs_done = false;
  if (mm_pair_offset(n, re) > mm_pair_offset(n, rb)) {
    const int rc = product_sym_pairs<T>(loss_kind, nv, sd, pa, tg, n, rb, re, la, accp + acc_elems,
                                        (flags & MM_WS_PREPARED) != 0, st);
    if (rc == MM_OK) pairs_done = true;
    else if (rc != MM_ERR_UNSUPPORTED) return rc;
  }
#define MM_PP(NV_, SD_) return product_pairs_launch<T, NV_, SD_>(loss_kind, pa, tg, n, rb, re, la, lo, st, psp, pairs_done, accp + acc_elems)
  switch (nv * 4 + sd) {
    case 0 * 4 + 2: MM_PP(0, 2);
    case 0 * 4 + 3: MM_PP(0, 3);
    case 1 * 4 + 0: MM_PP(1, 0);
    case 1 * 4 + 2: MM_PP(1, 2);
    case 1 * 4 + 3: MM_PP(1, 3);
    case 2 * 4 + 0: MM_PP(2, 0);
    case 2 * 4 + 2: MM_PP(2, 2);
    case 2 * 4 + 3: MM_PP(2, 3);
    case 3 * 4 + 0: MM_PP(3, 0);
    case 3 * 4 + 2: MM_PP(3, 2);
    case 3 * 4 + 3: MM_PP(3, 3);
    default: return MM_ERR_UNSUPPORTED;
  }
#undef MM_PP
}

#ifdef MM_PRODUCT_STAMP
extern "C" int mm_dbg_read_product_stamps(void* host, size_t bytes) {   // (diagnostic builds: tools/product_timeline.py)
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_product_stamps), bytes));
}
#endif
// ---- fused training step (product_step.hpp) --------------------------------------------------------------------
bool product_step_fusable(const mm_train_step* s) {
  static const bool off = [] { const char* e = std::getenv("MM_PRODUCT_STEP_UNFUSED"); return e && e[0] == '1'; }();
  if (off || s->nf < 2) return false;
  for (int k = 0; k < s->nf; ++k) {
    const mm_step_param& p = s->points[k];
    if (!p.x || !p.grad || p.count != s->n) return false;
    if (p.optimizer == MM_OPT_RSGD) { if (p.momentum != 0.0 && !p.state0) return false; }
    else if (p.optimizer == MM_OPT_RADAM) { if (!p.state0 || !p.state1 || !p.step || !p.ticket) return false; }
    else return false;
  }
  return true;
}

int product_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, hipStream_t st, bool* scale_stepped) {
  int kinds[4], dims[4];
  const void* xs[4];
  const void* sc[4];
  void* grads[4];
  for (int k = 0; k < s->nf; ++k) {
    kinds[k] = s->points[k].kind; dims[k] = s->points[k].dim; xs[k] = s->points[k].x; sc[k] = s->scales[k].x;
    grads[k] = s->points[k].grad;
    scale_stepped[k] = false;
    if (!sc[k]) return MM_ERR_ARG;
  }
  if (s->dtype == MM_F32)
    return product_pairs_t<float>(s->loss_kind, s->nf, kinds, dims, xs, sc, s->target, s->n, rb, re, s->alpha, s->eps, s->terms,
                                  s->loss_params, s->wmin, s->wmax, grads, s->loss_out, s->ws, s->ws_flags, st, nullptr, nullptr,
                                  0, s, scale_stepped);
  return product_pairs_t<double>(s->loss_kind, s->nf, kinds, dims, xs, sc, s->target, s->n, rb, re, s->alpha, s->eps, s->terms,
                                 s->loss_params, s->wmin, s->wmax, grads, s->loss_out, s->ws, s->ws_flags, st, nullptr, nullptr,
                                 0, s, scale_stepped);
}

}  // namespace mm

using namespace mm;

extern "C" {

size_t mm_product_pairs_ws_bytes(int dtype, int nf, const int* kinds, const int* dims, int64_t n) {
  const size_t es = dtype == MM_F64 ? 8 : 4;
  size_t elems = size_t(1 + nf) * kLossSlots;
  for (int k = 0; k < nf; ++k)
    elems += kinds[k] == MM_FACTOR_SPD ? size_t(dims[k]) * dims[k] * n : size_t(kPMP + 1) * n;
  elems += product_sym_table_elems(n);   // node table of the symmetric pair kernel (product_sym.hip)
  return 64 + es * elems;
}

int mm_product_pairs_loss(int dtype, int loss_kind, int nf, const int* kinds, const int* dims, const void* const* xs,
                          const void* const* scale_raw, const void* target, int64_t n, int64_t row_begin,
                          int64_t row_end, double alpha, double eps, int terms, const double* loss_params, double wmin, double wmax,
                          void* const* grads, void* loss_out, void* ws, int flags, mm_stream_t stream) {
  if (nf < 1 || nf > 4 || !kinds || !dims || !xs || !scale_raw || !grads || !loss_out || !ws || n < 1 ||
      n > (1 << 30) || row_begin < 0 || row_end > n || row_begin > row_end)
    return MM_ERR_ARG;
  for (int k = 0; k < nf; ++k)
    if (!xs[k] || !scale_raw[k] || !grads[k]) return MM_ERR_ARG;
  if (!target && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32)
    return product_pairs_t<float>(loss_kind, nf, kinds, dims, xs, scale_raw, target, n, row_begin, row_end, alpha, eps,
                                  terms, loss_params, wmin, wmax, grads, loss_out, ws, flags, st);
  if (dtype == MM_F64)
    return product_pairs_t<double>(loss_kind, nf, kinds, dims, xs, scale_raw, target, n, row_begin, row_end, alpha, eps,
                                   terms, loss_params, wmin, wmax, grads, loss_out, ws, flags, st);
  return MM_ERR_ARG;
}

int mm_product_pairs_loss_subset(int dtype, int loss_kind, int nf, const int* kinds, const int* dims,
                                 const void* const* xs, const void* const* scale_raw, const void* dense,
                                 int64_t n_total, const int64_t* idx, int64_t bs, int64_t row_begin, int64_t row_end,
                                 double alpha, double eps, int terms, const double* loss_params, double wmin, double wmax, void* const* grads,
                                 void* loss_out, void* ws, int flags, mm_stream_t stream) {
  if (nf < 1 || nf > 4 || !kinds || !dims || !xs || !scale_raw || !grads || !loss_out || !ws || !idx || !dense ||
      bs < 1 || bs > n_total || n_total >= (int64_t(1) << 31) || row_begin < 0 || row_end > bs || row_begin > row_end)
    return MM_ERR_ARG;
  for (int k = 0; k < nf; ++k)
    if (!xs[k] || !scale_raw[k] || !grads[k]) return MM_ERR_ARG;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32)
    return product_pairs_t<float>(loss_kind, nf, kinds, dims, xs, scale_raw, nullptr, bs, row_begin, row_end, alpha, eps,
                                  terms, loss_params, wmin, wmax, grads, loss_out, ws, flags, st, idx, dense, n_total);
  if (dtype == MM_F64)
    return product_pairs_t<double>(loss_kind, nf, kinds, dims, xs, scale_raw, nullptr, bs, row_begin, row_end, alpha,
                                   eps, terms, loss_params, wmin, wmax, grads, loss_out, ws, flags, st, idx, dense, n_total);
  return MM_ERR_ARG;
}

}  // extern "C"
