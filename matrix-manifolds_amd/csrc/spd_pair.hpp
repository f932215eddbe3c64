// SPD(d) pair kernels and their launchers as templates, shared by the translation units that instantiate them:
//   spd.hip        forward, backward (upstream gradient), element-wise dist, per-point maps, optimizer kernels, Stein divergence
//   spd_loss.hip   fused objective (loss + gradients in one pass) and the fused training step of a single SPD factor
//   spd_subset.hip node minibatches inside the pair kernel (index vector -> table rows, dense targets, gradient rows)
// (one file until round 4: 3 m 45 s of single-threaded compilation; the split compiles in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "smallmat.hpp"
#include "loss.hpp"
#include "adam.hpp"
#include "spd_rules.hpp"

#include "spd_ws.hpp"
#include "spd_step.hpp"
#include "stamp.hpp"

namespace mm {

// ------------------------------------------------------------------ prep
// The per-node tables of ONE point from its packed symmetric part: Cholesky factor, its inverse, log det (the part of
// _lult, manifolds/spd.py:108-111, that depends on one point only).  Shared by spd_prep_kernel and by the fused optimizer
// kernels, which write the tables of the NEW point in the same pass (no preparation launch in steady state).
template <typename T, int D>
__device__ __forceinline__ void node_tables(const T (&xs)[Packed<D>::NP], int i, T* __restrict__ nodeL, T* __restrict__ nodeX,
                                            T* __restrict__ nodeC, int* __restrict__ bad, T* __restrict__ nodeLd,
                                            T* __restrict__ nodeLC) {
  constexpr int NP = Packed<D>::NP;
  T l[NP], li[NP];
  const bool ok = cholesky<T, D>(xs, l);
  invert_lower<T, D>(l, li);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    nodeL[size_t(i) * NP + k] = li[k];
    nodeX[size_t(i) * NP + k] = xs[k];
    nodeC[size_t(i) * NP + k] = l[k];
    nodeLC[size_t(i) * 2 * NP + k] = li[k];
    nodeLC[size_t(i) * 2 * NP + NP + k] = l[k];
  }
  bad[i] = ok ? 0 : 1;
  T ld = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) ld += Num<T>::log(l[pidx(k, k)]);
  nodeLd[i] = ld + ld;
}

template <typename T, int D>
__global__ void spd_prep_kernel(const T* __restrict__ x, int n, T* __restrict__ nodeL, T* __restrict__ nodeX,
                                T* __restrict__ nodeC, T* __restrict__ accM, T* __restrict__ accS,
                                T* __restrict__ loss, int* __restrict__ bad, T* __restrict__ nodeLd,
                                T* __restrict__ nodeLC) {
  constexpr int NP = Packed<D>::NP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < 2 * kLossSlots; t += blockDim.x) loss[t] = T(0);
  if (i >= n) return;
  T xs[NP];
  load_sym_packed<T, D>(x + size_t(i) * D * D, xs);
  node_tables<T, D>(xs, i, nodeL, nodeX, nodeC, bad, nodeLd, nodeLC);
#pragma unroll
  for (int k = 0; k < NP; ++k) accM[size_t(k) * n + i] = T(0);
#pragma unroll
  for (int k = 0; k < D * D; ++k) accS[size_t(k) * n + i] = T(0);
}


// A = Li X Li^T, eigen-decompose, return s = sum log^2 clamp(w).
// The second operand of a pair is X_j (CHOL = false: element-wise kernels) or its Cholesky factor
// (CHOL = true: the all-pairs kernels, which have it in the node tables — cheaper congruence).
template <typename T, int D, bool CHOL, typename TL>
__device__ __forceinline__ void pair_a(const TL (&li)[Packed<D>::NP], const T (&yj)[Packed<D>::NP],
                                       T (&a)[Packed<D>::NP]) {
  if constexpr (CHOL) congr_chol<T, D>(li, yj, a); else congr_lower<T, D>(li, yj, a);
}

// SECOND: the second (one-sided) solve for ill-conditioned fp32 pairs — 0: none, 1: inline, 2: out of line (below)
// The second, one-sided solve of pair_core OUT OF LINE (SECOND == 2): for the kernel that cannot afford it inline — the
// two-columns-per-lane SPD(4) backward sits at its register cap (168: three wavefronts per SIMD) with the hot paths alone.
// Round 6, same box, n = 16384, us (pdist backward at the reference init / fused QuotientLoss step / pdist backward at the
// mid-training spread; profiles/r06_experiments.md section 2):
//   no second solve (rounds 4 - 5: the gradient's accuracy on ill-conditioned points depended on the launch size)  918 / 1030 / 1176
//   inlined in registers: 108 - 510 spilled registers, placed on the hot paths                                       944 / 3620 / 1390
//   inlined with B^T and V in LDS (the combine buffer), rotation loops rolled, a dozen temporaries                   947 / 1240 / 1558
//   THIS: a real call — operands through a scratch record, the callee allocates its own registers                   935 / 1058 / 1169
// Any code added to that kernel moves its register assignment; the call moves it least (+1.4 ... 2.7 %, against +4 ... 5 % for one
// column per lane at every size), and what it costs is paid for an accuracy that no longer depends on how a problem is sharded.
template <typename T, int D>
__device__ __attribute__((noinline)) void second_solve_ool(const T* __restrict__ in /* li[NP], xj[NP] */, T* __restrict__ out /* ev[D], v[D][D] */, T tol2) {
  constexpr int NP = Packed<D>::NP;
  T g[D][D], b[D][D], v[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      if (c > r) { g[c][r] = T(0); continue; }        // g = B^T: g[c][r] = B[r][c], B = L_i^-1 L_j
      T acc = in[pidx(r, c)] * in[NP + pidx(c, c)];
#pragma unroll
      for (int k = c + 1; k <= r; ++k) acc = Num<T>::fma(in[pidx(r, k)], in[NP + pidx(k, c)], acc);
      g[c][r] = acc;
    }
  svd_onesided<T, D, true>(g, b, v, tol2);
#pragma unroll
  for (int k = 0; k < D; ++k) {
    T e = T(0);
#pragma unroll
    for (int r = 0; r < D; ++r) e = Num<T>::fma(b[r][k], b[r][k], e);
    out[k] = e;
  }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) out[D + r * D + c] = v[r][c];
}

template <typename T, int D, bool WITH_V, bool CHOL = false, int SECOND = 1, typename TL>
__device__ __forceinline__ T pair_core(const TL (&li)[Packed<D>::NP], const T (&xj)[Packed<D>::NP], T wmin, T wmax,
                                       T (&w)[D], T (&lw)[D], T (&v)[D][D], T* rho = nullptr) {
  // eigenvalues only: sum log^2 w is second-order in the residual coupling -> tol2 = eps;
  // with eigenvectors: residual coupling <= 8 eps relative (gradient error ~1e-6, a quarter of
  // the wavefronts at the reference init would otherwise run a 4th sweep for the last bit)
  const T tol2 = WITH_V ? T(64) * Num<T>::eps() * Num<T>::eps() : Num<T>::eps();
  T ev[D];
  {
    T a[Packed<D>::NP];
    pair_a<T, D, CHOL>(li, xj, a);
    jacobi_eig<T, D, WITH_V, true>(a, v, tol2);
#pragma unroll
    for (int k = 0; k < D; ++k) ev[k] = a[pidx(k, k)];
  }
#ifndef MM_SPD_JACOBI_TWO_SIDED   // (A/B builds: the round-1..3 route only)
  if constexpr (SECOND && CHOL && D <= 4 && std::is_same<T, float>::value) {
    // fp32, ill-conditioned pairs: forming A = B B^T (B = L_i^-1 L_j) costs eps cond(A) of relative accuracy in A's small
    // eigenvalues whatever solves it afterwards — 1.2 (!) of d^2 at cond(X) = 1e4, where fp64 is fine.  A wavefront that
    // holds a pair with lambda_max > 256 lambda_min (or a non-positive / NaN spectrum) solves again, by a one-sided Jacobi on
    // B^T: V = eigenvectors of A, squared column norms = its eigenvalues, with the relative accuracy of B's entries
    // (smallmat.hpp, svd_onesided).  Measured (tools/illcond_probe.py, max relative error of d^2 against fp64 on the same
    // inputs): SPD(3) cond(X) = 1e4 1.2 -> 1.3e-4, 1e6 1.2 -> 6e-3; SPD(4) 7.6e-2 -> 8.9e-5, 0.94 -> 4.4e-3.  Always taking this
    // route would cost the whole Jacobi regime 30 - 75 % (profiles/r04_experiments.md): it is a second solve for rows that need it.
    T mn = ev[0], mx = ev[0];
#pragma unroll
    for (int k = 1; k < D; ++k) { mn = Num<T>::min(mn, ev[k]); mx = Num<T>::max(mx, ev[k]); }
    if constexpr (SECOND == 2) {
      static_assert(WITH_V, "the out-of-line second solve returns eigenvectors");
      if (__builtin_expect(__any(!(mx <= T(256) * mn)), 0)) {
        T rec_in[2 * Packed<D>::NP], rec_out[D + D * D];   // (escape into the call: they live in scratch, on this path only)
#pragma unroll
        for (int k = 0; k < Packed<D>::NP; ++k) { rec_in[k] = T(li[k]); rec_in[Packed<D>::NP + k] = xj[k]; }
        second_solve_ool<T, D>(rec_in, rec_out, tol2);
#pragma unroll
        for (int k = 0; k < D; ++k) ev[k] = rec_out[k];
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
          for (int c = 0; c < D; ++c) v[r][c] = rec_out[D + r * D + c];
      }
    } else
    if (__builtin_expect(__any(!(mx <= T(256) * mn)), 0)) {   // (cold: laid out behind the loop)
      T g[D][D], b[D][D];
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          if (c > r) { g[c][r] = T(0); continue; }        // g = B^T: g[c][r] = B[r][c]
          T acc = li[pidx(r, c)] * xj[pidx(c, c)];
#pragma unroll
          for (int k = c + 1; k <= r; ++k) acc = Num<T>::fma(li[pidx(r, k)], xj[pidx(k, c)], acc);
          g[c][r] = acc;
        }
      svd_onesided<T, D, WITH_V>(g, b, v, tol2);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        ev[k] = T(0);
#pragma unroll
        for (int r = 0; r < D; ++r) ev[k] = Num<T>::fma(b[r][k], b[r][k], ev[k]);
      }
    }
  }
#endif
  T s = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) {
    w[k] = Num<T>::min(Num<T>::max(ev[k], wmin), wmax);
    lw[k] = Num<T>::log(w[k]);
    s = Num<T>::fma(lw[k], lw[k], s);
    // rho = true / clamped eigenvalue, exactly 1 unless a clamp binds: the reference clamps the VALUES in place (w.data.clamp_,
    // spd.py:163-169), so log's backward divides by the clamped eigenvalue while the decomposition's backward runs on the true A —
    // the X_j side of the gradient carries log(w_c) / w_c, the X_i side log(w_c) w / w_c (tests/golden/gen_golden_clamps.py).
    // Asked for by the element-wise backward only: the pair kernels run under clamps that cannot bind (spd_clamps_supported).
    if (rho) rho[k] = w[k] == ev[k] ? T(1) : ev[k] * Num<T>::rcp(w[k]);
  }
  return s;
}

// ---- eigen-free matrix logarithm, D = 3, 4 (smallmat.hpp) -----------------------------------
// ||M||_F^2 of a packed symmetric matrix
template <typename T, int D> __device__ __forceinline__ T frob2(const T (&m)[Packed<D>::NP]) {
  T s = T(0);
#pragma unroll
  for (int r = 0; r < D; ++r) {
    s = Num<T>::fma(m[pidx(r, r)], m[pidx(r, r)], s);
#pragma unroll
    for (int c = 0; c < r; ++c) s = Num<T>::fma(T(2) * m[pidx(r, c)], m[pidx(r, c)], s);
  }
  return s;
}
// ||A - I||_F^2 with the arithmetic of log_series3/4 (so the gate shares it with the series)
template <typename T, int D> __device__ __forceinline__ T close_gate(const T (&a)[Packed<D>::NP]) {
  using N = Num<T>;
  T e[Packed<D>::NP];
#pragma unroll
  for (int k = 0; k < Packed<D>::NP; ++k) e[k] = a[k];
#pragma unroll
  for (int r = 0; r < D; ++r) e[pidx(r, r)] -= T(1);
  if constexpr (D == 3) {
    return N::fma(e[0], e[0], N::fma(e[1], e[1], e[3] * e[3])) + N::fma(e[1], e[1], N::fma(e[2], e[2], e[4] * e[4])) +
           N::fma(e[3], e[3], N::fma(e[4], e[4], e[5] * e[5]));
  } else {
    T p2 = T(0);
#pragma unroll
    for (int r = 0; r < D; ++r) {
      T acc = e[pidx(r, 0)] * e[pidx(0, r)];
#pragma unroll
      for (int k = 1; k < D; ++k) acc = N::fma(e[pidx(r, k)], e[pidx(k, r)], acc);
      p2 += acc;
    }
    return p2;
  }
}
// close pairs: ||A - I||_F <= 0.3 (the series' polynomials are fitted on a spectral radius of 0.3)
constexpr double kCloseGate = 0.09;
template <typename T, int D> __device__ __forceinline__ void log_close(const T (&a)[Packed<D>::NP], T (&m0)[Packed<D>::NP],
                                                                     T pre = T(1)) {
  if constexpr (D == 3) log_series3<T>(a, m0, pre); else log_series4<T>(a, m0, pre);
}
// `pre` multiplies log A (folded into three scalars by the 3x3 ring form; 4x4: applied to the matrix)
template <typename T, int D> __device__ __forceinline__ T log_cayley(const T (&a)[Packed<D>::NP], T (&m0)[Packed<D>::NP], T pre = T(1)) {
#ifdef MM_CAYLEY_MATRIX   // (A/B builds: the round-3 form with the adjugate and three matrix products)
  constexpr bool kRing = false;
#else
  constexpr bool kRing = D == 3;
#endif
  if constexpr (kRing) {
    return log_cayley3<T>(a, m0, pre);
  } else {
    T gate;
    if constexpr (D == 3) gate = log_cayley3_matrix<T>(a, m0); else gate = log_cayley4<T>(a, m0);
#pragma unroll
    for (int k = 0; k < Packed<D>::NP; ++k) m0[k] *= pre;
    return gate;
  }
}
// the forward's far path takes the invariants-only form where log det A is at hand (all-pairs kernels: per-node table)
template <typename T, int D> constexpr bool fwd_uses_logdet() {
#ifdef MM_CAYLEY_MATRIX
  return false;
#else
  return D == 3 && std::is_same<T, double>::value;
#endif
}

// The reference clamps the eigenvalues of A to [wmin, wmax] (value only, spd.py:29-30, 163-169; defaults 1e-8 / 1e8).  The
// eigen-free paths of SPD(3 .. 9) never see eigenvalues and the fused objectives skip the clamp of d^2 at wmin: the pair kernels run
// only under clamps at least as wide as [1e-6, 1e6] (inside a gate a pair's eigenvalues are within a factor 16 of tr A / D, so a
// clamp could bind there only for tr A / D beyond 6e4 resp. below 1.6e-5 of the window's end — no embedding has such pairs) and
// their launchers return MM_ERR_UNSUPPORTED otherwise.  Narrower windows: SPD(2) `pdist` in closed form (log_pair2_chol), every
// size through the element-wise kernels below (eigensolve, with what the clamp leaves in the gradient: pair_core's rho) —
// graphembed.manifolds.spd routes `pdist` there, as the reference's own Manifold.pdist does (base.py:59-63).
__host__ __device__ inline bool spd_clamps_wide(double wmin, double wmax) { return wmin <= 1e-6 && wmax >= 1e6; }
template <int D, int LOSS> inline bool spd_clamps_supported(double wmin, double wmax) {
  return (D <= 2 && LOSS == MM_LOSS_NONE) || spd_clamps_wide(wmin, wmax);
}
#ifdef MM_SPD_NO_SERIES_MAT   // (A/B builds: SPD(5 .. 9) on the Jacobi route of rounds 1-4)
constexpr bool kSeriesMat = false;
#else
constexpr bool kSeriesMat = true;
#endif
// Forward-only value of one pair.  SPD(3) in fp32 takes the closed-form (trigonometric)
// eigenvalues; a wavefront in which any pair has a wide spectrum (w_max > 32 w_min, where
// the closed form's absolute error would show in log w_min) re-solves with Jacobi.
// HAS_LD: `logdet_a` = log det A = log det X_j - log det X_i is known (per-node table).
template <typename T, int D, bool CHOL = false, bool HAS_LD = false, typename TL>
__device__ __forceinline__ T pair_value(const TL (&li)[Packed<D>::NP], const T (&xj)[Packed<D>::NP], T wmin, T wmax,
                                        T logdet_a = T(0)) {
  if constexpr (D == 3 && std::is_same<T, float>::value) {
    float a[6], w[3], v[3][3];
    pair_a<float, 3, CHOL>(li, xj, a);
    // close pairs (whole wavefront within ||A - I||_F <= 0.3): invariants-only series (its arithmetic shares the gate's)
    if (__builtin_expect(!__any(!(close_dev3<float>(a) <= float(kCloseGate))), 1)) {
      float e2;
      return logsq_series3<float>(a, &e2);
    }
#ifndef MM_NO_CENTRED
    // pairs at moderate distance (training after the first epochs): the recentred invariants-only series
    if (!__any(centred_far3<float>(a))) return logsq_series3_centred<float>(a);
#endif
    eig3_trig(a, w);
    const bool wide = !(w[0] * 32.f > w[2]);  // true for NaN / non-positive spectra too
    if (__any(wide)) {
      if constexpr (CHOL) {   // (wide spectra = ill-conditioned pairs: the eigenvalues from B = L_i^-1 L_j itself, pair_core)
        float lw[3];
        return pair_core<float, 3, false, true>(li, xj, wmin, wmax, w, lw, v);
      } else {
        jacobi_eig<float, 3, false, true>(a, v, Num<float>::eps());
        w[0] = a[pidx(0, 0)]; w[1] = a[pidx(1, 1)]; w[2] = a[pidx(2, 2)];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float l = Num<float>::log(fminf(fmaxf(w[k], wmin), wmax));
      s = fmaf(l, l, s);
    }
    return s;
  } else if constexpr (D == 3 || D == 4) {
    // SPD(4), and SPD(3) in fp64: ||log A||_F^2 without an eigensolve — close-pair series (SPD(3) fp64: straight
    // from the invariants, like fp32), then the Cayley-transform logarithm; Jacobi only if a pair of the wavefront
    // has a very wide spectrum
    constexpr int NP = Packed<D>::NP;
    T a[NP], m0[NP];
    pair_a<T, D, CHOL>(li, xj, a);
    if constexpr (D == 3) {
      // (the gate FIRST, as in fp32: with the series in front of it — sharing its arithmetic — a far row paid for 20
      // coefficients it threw away: 60 multiply-adds per pair of the mid-training regime)
      if (__builtin_expect(!__any(!(close_dev3<T>(a) <= T(kCloseGate))), 1)) {
        T e2;
        return logsq_series3<T>(a, &e2);
      }
      if constexpr (HAS_LD && fwd_uses_logdet<T, D>()) {   // eigenvalue ratios up to ~16: invariants only (smallmat.hpp, logsq_cayley3)
        T gate;
        const T sc = logsq_cayley3<T>(a, logdet_a, &gate);
        if (__builtin_expect(!__any(!(gate <= T(kCayleyGate))), 1)) return sc;
        T w[D], lw[D], v[D][D];   // (the matrix form below has the same gate: straight to the eigensolve)
        return pair_core<T, D, false, CHOL>(li, xj, wmin, wmax, w, lw, v);
      }
    } else {
      if (__builtin_expect(!__any(!(close_gate<T, D>(a) <= T(kCloseGate))), 1)) return logsq_series4<T>(a);
#ifndef MM_NO_CENTRED
      if constexpr (std::is_same<T, float>::value) {   // pairs at moderate distance: the recentred invariants-only series
        if (!__any(centred_far4<T>(a))) return logsq_series4_centred<T>(a);
      }
#endif
    }
    const T gate = log_cayley<T, D>(a, m0);
    if (__builtin_expect(!__any(!(gate <= T(kCayleyGate))), 1)) return frob2<T, D>(m0);
    T w[D], lw[D], v[D][D];
    return pair_core<T, D, false, CHOL>(li, xj, wmin, wmax, w, lw, v);
#ifndef MM_SPD2_JACOBI   // (A/B builds: the Jacobi route of rounds 1-4)
  } else if constexpr (D == 2 && CHOL) {
    T mlog[3];   // (closed form, the small eigenvalue from the determinant: smallmat.hpp; the matrix is not needed here)
    return log_pair2_chol<T>(li, xj, wmin, wmax, mlog);
#endif
  } else if constexpr (D >= 5 && kSeriesMat) {
    // SPD(5 .. 9): the matrix-Horner series (smallmat.hpp, log_series_mat) for wavefronts of close pairs and of pairs at
    // moderate distance; the eigensolve for the rest
    constexpr int NP = Packed<D>::NP;
    T a[NP], m0[NP];
    pair_a<T, D, CHOL>(li, xj, a);
    if (__builtin_expect(!__any(!(close_gate<T, D>(a) <= T(kCloseGate))), 1)) {
      log_close_mat<T, D>(a, m0);
      return frob2<T, D>(m0);
    }
    if (!__any(centred_far_mat<T, D>(a))) {
      log_centred_mat<T, D>(a, m0);
      return frob2<T, D>(m0);
    }
    T w[D], lw[D], v[D][D];
    return pair_core<T, D, false, CHOL>(li, xj, wmin, wmax, w, lw, v);
  } else {
    T w[D], lw[D], v[D][D];
    return pair_core<T, D, false, CHOL>(li, xj, wmin, wmax, w, lw, v);
  }
}

// ------------------------------------------------------------------ forward
// Columns per lane of the pair kernels: two for fp32 SPD(2), SPD(3) — the second pair of a row shares the row operand's
// scalar loads and the loop's scalar bookkeeping (every instruction of a wavefront, scalar ones included, takes an issue
// slot of its SIMD) and, in the backward, the row-side reduction (one reduction of M_a + M_b); wider matrices and fp64
// do not have the registers for it.
#ifndef MM_SPD2_FWD_NC   // (A/B builds: columns per lane of the SPD(2) kernels)
#define MM_SPD2_FWD_NC 2
#endif
#ifndef MM_SPD2_BWD_NC
#define MM_SPD2_BWD_NC 2
#endif
template <typename T, int D> constexpr int pair_cols() { return (sizeof(T) == 4 && D == 2) ? MM_SPD2_FWD_NC : (sizeof(T) == 4 && D <= 3) ? 2 : ((sizeof(T) == 4 && D == 4) ? MM_SPD4_FWD_NC : 1); }
template <typename T, int D> constexpr int pair_cols_bwd() { return (sizeof(T) == 4 && D == 3) ? MM_SPD3_BWD_NC : (sizeof(T) == 4 && D == 2) ? MM_SPD2_BWD_NC : ((sizeof(T) == 4 && D == 4) ? MM_SPD4_BWD_NC : 1); }

// (Round 5: the preparation launch in front of this kernel — spd_prep_kernel, 5 us — cannot be fused into it at a profit.  The row
// operand must reach the arithmetic as a SCALAR operand: handed over through v_readlane, every vector instruction that reads a
// scalar register a vector instruction has just written stalls its SIMD ~20 cycles (forward + 40 ... 55 %); as wave-uniform values
// in VECTOR registers (LDS broadcast) the multiply-adds read three vector operands instead of two (+ 25 % at SPD(4) n = 16384).
// Scalar loads from a table that another launch wrote are the cheap way.  profiles/r05_experiments.md, section 12.)
// Tile: TI rows x (256 x NC) columns per workgroup; lane l of wavefront w owns the columns jbase + 64 (NC w + q) + l.
// The row loop is unrolled twice with two alternating scalar register sets for the row operand L_i^-1 (no copies), the
// output row is a running scalar pointer (row i + 1 starts n - i - 2 elements after row i) plus a fixed lane offset:
// `global_store_dword v_off, v, s[ptr]`, lanes on consecutive j -> 256-B coalesced segments of the row-major pair vector.
template <typename T, int D, int TI, bool SQ>
__global__ __launch_bounds__(kBlock) void spd_pdist_fwd_kernel(const T* __restrict__ nodeL,
                                                               const T* __restrict__ nodeY /* column operand: chol(X_j) */,
                                                               const T* __restrict__ nodeLd /* log det X */, int n, int row_begin,
                                                               int row_end, T wmin, T wmax, T* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  constexpr int NC = pair_cols<T, D>();
  constexpr bool kLd = fwd_uses_logdet<T, D>();   // log det A = nodeLd[j] - nodeLd[i] feeds the far path
  static_assert(TI % 2 == 0, "the row loop is unrolled twice");
  const TileId tile = fold_tile<TI, kBlock * NC>(n, row_begin, row_end);
  if (!tile.ok) return;
  const int i0 = tile.i0, i1 = min(i0 + TI, row_end);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave_j0 = tile.jbase + wave * (64 * NC);
  if (wave_j0 + 64 * NC - 1 <= i0) return;  // whole wavefront below the diagonal
  int jv[NC];          // column for the validity test (never above a row for lanes beyond n)
  unsigned joff[NC];   // byte offset of the column in a row of the pair vector
  T xj[NC][NP], ldj[NC];
  static_for<NC>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int j = wave_j0 + 64 * q + lane;
    const bool jin = j < n;
    jv[q] = jin ? j : INT32_MIN;
    asm volatile("" : "+v"(jv[q]));   // (or the select is undone into `jin && j > row`: a scalar AND per row and column)
    joff[q] = unsigned(j) * unsigned(sizeof(T));
#pragma unroll
    for (int k = 0; k < NP; ++k) xj[q][k] = T(0);
#pragma unroll
    for (int k = 0; k < D; ++k) xj[q][pidx(k, k)] = T(1);
    ldj[q] = T(0);
    if (jin) {
#pragma unroll
      for (int k = 0; k < NP; ++k) xj[q][k] = nodeY[size_t(j) * NP + k];
      if constexpr (kLd) ldj[q] = nodeLd[j];
    }
  });
  const int64_t base = pair_off(n, row_begin);
  char* op = reinterpret_cast<char*>(out + (pair_off(n, i0) - base - i0 - 1));   // element (i0, j) lives at op + j sizeof(T)
  unsigned ostep = unsigned(n - i0 - 2) * unsigned(sizeof(T));                    // bytes from row i to row i + 1
  // row operand: wave-uniform -> scalar loads, issued one row ahead (the row after the tile's last is read too: inside the
  // workspace — nodeL is followed by nodeX — and never used)
  unsigned roff = unsigned(i0) * unsigned(NP * sizeof(T));
  T lrow[2][NP], ldrow[2] = {T(0), T(0)};
#pragma unroll
  for (int k = 0; k < NP; ++k) lrow[0][k] = nodeL[size_t(i0) * NP + k];
  if constexpr (kLd) ldrow[0] = nodeLd[i0];
  for (int ib = i0; ib < i1; ib += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int irow = ib + u;
      const int ieff = (u == 0 || irow < i1) ? irow : INT32_MAX;   // (the second slot of an odd last pair of rows stores nothing)
      roff += unsigned(NP * sizeof(T));
      asm volatile("" : "+s"(roff));   // (pinned: scalar loads at small positive offsets of the running offset)
      const T* rowp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(nodeL) + roff);
#pragma unroll
      for (int k = 0; k < NP; ++k) lrow[u ^ 1][k] = rowp[k];
      // (row irow + 1 <= n - 1 of the log-det table whenever the slot's result is used; the masked slot past the range reads
      // at most nodeLd[n + 1], inside the table that follows it in the workspace)
      if constexpr (kLd) ldrow[u ^ 1] = nodeLd[irow + 1];
      const T (&li)[NP] = lrow[u];
      static_for<NC>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        T s = Num<T>::max_raw_s(pair_value<T, D, true, kLd>(li, xj[q], wmin, wmax, ldj[q] - ldrow[u]), wmin);
        if constexpr (!SQ) s = Num<T>::sqrt(s);
        // (re-defined in this block: a zero-extension hoisted out of the loop hides from instruction selection that the
        // lane offset is 32 bits wide, and the store gets a 64-bit vector address instead of `v_off, s[ptr]`)
        asm volatile("" : "+v"(joff[q]));
        if (jv[q] > ieff) *reinterpret_cast<T*>(op + joff[q]) = s;
      });
#ifndef MM_DIAG_STATIC_ROWS   // (diagnostic builds: every row of a tile is stored over its first — the kernel without its HBM stream; wrong results)
      op += ostep;
      ostep -= unsigned(sizeof(T));
#endif
    }
  }
}

// The value loaded from the pair vector is the upstream gradient (LOSS == 0: of d2, or of d when
// !squared) or the loss target; either way this returns d loss / d (d2) of the pair.
template <typename T, int LOSS>
__device__ __forceinline__ T upstream_of(T loaded, T dsq, bool valid, int squared, T wmin, T sp, const LossArgs<T>& la,
                                         T& loss_acc, T& ds_acc) {
  if constexpr (LOSS == MM_LOSS_NONE) {
    if (__builtin_expect(!squared, 0)) loaded *= T(0.5) * Num<T>::rsqrt(Num<T>::max(dsq, wmin));  // d -> d2
    return loaded;
  } else {
    T dldm;
    const T l = loss_term<T, LOSS>(sp * dsq, loaded, la, dldm);
    loss_acc += valid ? l : T(0);
    ds_acc += valid ? dldm * dsq : T(0);
    return valid ? dldm * sp : T(0);
  }
}

// ------------------------------------------------------------------ backward
// Node of batch position k of a node minibatch: the low word of the 64-bit index (node ids are < 2^22), CLAMPED into the
// per-node tables — the index vector is caller data the kernels cannot validate (host-side tensors are checked by the
// Python layer, graphembed.modules.distinct_in_range; device-side ones by nobody): a negative or too large index then
// yields wrong numbers for that batch, not a read or an atomic outside the workspace.  One s_min_u32 / v_min_u32 per load.
__device__ __forceinline__ int batch_node(const int* __restrict__ idx32, size_t k, int n_total) {
  return int(min(unsigned(idx32[2 * k]), unsigned(n_total - 1)));
}

// One row of the pair vector for this lane: element (row, j) lives at pair_off(n, row) - base + (j - row - 1).
// The row's base is wave-uniform (scalar registers), the lane part is a 32-bit byte offset, so the load is
// `global_load_dword v, v_off, s[base]` with ONE vector instruction of address arithmetic (a clamp).  Lanes at or
// below the diagonal and beyond n read the row's first / last element instead — unconditional loads: a predicated
// one is an exec-masked branch behind which the compiler waits for vmcnt(0), exposing the HBM latency — and
// their value is masked at use.
template <typename T>
__device__ __forceinline__ T pair_row_load(const T* __restrict__ g, int n, int64_t base, int row, int j) {
  const char* grow = reinterpret_cast<const char*>(g + (pair_off(n, row) - base - row - 1));   // wave-uniform
  const unsigned off = unsigned(min(max(j, row + 1), n - 1)) * unsigned(sizeof(T));
  return *reinterpret_cast<const T*>(grow + off);
}

// Backward: a launch of (at most) as many workgroups as the device holds at once; workgroup w walks its share of the
// balanced column walk (spd_ws.hpp, ColWalk / WalkShares): down one block of 64 NC columns — each wavefront ONE contiguous slice of
// the block's rows of the share, its lanes the block's columns, the row sums leaving every TI rows — then on to the next block.
// NCX != 0: that many columns per lane instead of pair_cols_bwd<T, D>() (fp32 SPD(4): two for large launches, below).
// SUB: a NODE MINIBATCH (train.py:198-222 with batch_size set; modules.py:86 gathers x[idx] first): the n points of the launch
// are the nodes idx[0..n) of an embedding of n_total points.  The pair list (rows, column blocks, shares) is that of the n
// batch positions; everything that lives per NODE goes through the index vector — the row operands nodeLC[idx[a]], the
// column operands nodeY[idx[b]], the target dense[idx[a]][idx[b]] (`g` is the dense n_total x n_total matrix of
// GraphDataset, data/dataset.py:19-27, instead of a pair vector) and the accumulators accM / accS [.][n_total], which the
// per-node kernels that follow turn into rows idx[.] of the FULL gradient (every other row comes out zero: autograd's
// index backward).  No x[idx] gather, no target gather, no scatter-add around the kernel.
template <typename T, int D, int TI, int LOSS, bool SQ, int NCX = 0, bool SUB = false>
__global__ __launch_bounds__((64 * bwd_waves<T, D>()), (bwd_min_waves_nc<T, D, NCX>())) void spd_pdist_bwd_kernel(const T* __restrict__ nodeLC /* {L_i^-1, L_i} */,
                                                               const T* __restrict__ nodeY /* chol(X_j) */,
                                                               const T* __restrict__ g, int n, int row_begin,
                                                               int row_end, T wmin, T wmax,
                                                               T* __restrict__ accM, T* __restrict__ accS,
                                                               LossArgs<T> la,
                                                               const int64_t* __restrict__ idx, int n_total,
                                                               WalkShares shares, int* __restrict__ share_tab) {
  static_assert(!SUB || LOSS != MM_LOSS_NONE, "node minibatches exist for the fused objective only");
  const int ns = SUB ? n_total : n;                      // stride of the per-node accumulators
  const int* idx32 = reinterpret_cast<const int*>(idx);  // (little-endian low words: node ids are < 2^22)
  constexpr int NP = Packed<D>::NP;
  constexpr int NW = bwd_waves<T, D>();
  constexpr int NC = NCX ? NCX : pair_cols_bwd<T, D>();   // lane l owns the columns jbase + 64 q + l, q < NC
  // LOSS != 0: `g` holds the TARGET (graph) squared distances; the upstream gradient of each pair is
  // derived in registers from the loss, and the loss / scale-gradient sums leave through la.slots.
  constexpr int squared = SQ ? 1 : 0;   // (a template parameter: as a run-time flag it cost two vector instructions per row)
  T sp = T(1), loss_acc = T(0), ds_acc = T(0);
  loss_resolve<T, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  // The wavefronts of a workgroup share the columns: their column-side partial sums are combined through LDS
  // and leave with ONE set of atomics per column block (float atomics are a per-CU serial resource, ~50 ns per
  // wave instruction).
  static_assert(TI % 2 == 0, "the row loop is unrolled twice");
  __shared__ T redM[NW][TI][NP];
  __shared__ T colS[NW][NC][D * D][64];
  MM_SPD_STAMP_BEGIN();
#ifndef MM_DIAG_NO_PRIO
  // (first thing: a wavefront starts at priority 0, and with the older workgroups of its CU in their row loops at 3 the prologue
  // of a late arrival would be served last)
  __builtin_amdgcn_s_setprio(3);
#endif
  const ColWalk walk(n, row_begin, row_end, 64 * NC);
  // this workgroup's share of the walk, cut on the host (WalkShares): the column block and row it starts at and its budget of
  // units — one per row, shares.cross per block entered (ColWalk::enter); block-uniform
  int cb, r, rem;
#ifdef MM_NO_SHARE_TAB   // (A/B builds: the start computed by every workgroup of every launch, as in rounds 5 and before)
  shares.of(walk, int(blockIdx.x), cb, r, rem);
#else
  shares.of_cached(walk, int(blockIdx.x), share_tab, cb, r, rem);
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> row operands stay scalar loads
  bool red_writer;
  const int red_slot = reduce_slot<NP>(lane, red_writer);  // which entry of M this lane holds after the row reduction
  const int64_t base = pair_off(n, row_begin);
  MM_SPD_STAMP_MARK(0);
  // Vector issue is arbitrated oldest wavefront first: of equal shares started together, the oldest workgroup of a
  // CU finishes when the youngest is barely half way, and the rest of the launch runs at one or two wavefronts per
  // SIMD (measured: five completion steps of 256 workgroups, the last 20 us at <= 40 % residency).  Priority outranks
  // age, so every wavefront LOWERS its priority as it advances through its share (3, 2, 1 over three equal stretches, 0 for the
  // last tenth): whoever is ahead yields to whoever is behind, and all of them enter the last stretch together.
  // Progress is counted in rows of THIS wavefront (a quarter of the share's), one scalar compare per row.
  const int wave_rows = (rem + NW - 1) / NW;
  // three equal stretches at 3, 2, 1 and a LAST one at 0 of a tenth of the rows, at least four: what follows the last step runs
  // in dispatch order — oldest wavefront first — and its length is the spread of the finishing times.  (Rounds 2-4 stepped at
  // 40 / 70 / 90 % + 1 row each: with the ~25 rows a wavefront has in the headline launch the third step fell on the last row
  // and the final stretch was the six rows behind the second.  Headline 43.3 -> 42.2 us, SPD(4) n = 2274 27.4 -> 25.6; a fixed
  // four rows for every size cost the 500-row shares of n = 16384 2 %: profiles/r05_experiments.md.)
#ifndef MM_SPD_PRIO_LAST
#define MM_SPD_PRIO_LAST 4
#endif
  const int prio_last = max(MM_SPD_PRIO_LAST, wave_rows / 10);
  const int prio_stretch = max(wave_rows - prio_last, 3) / 3;
  int rows_left = prio_stretch;   // rows until the next priority step
  int phase = 0;
#ifndef MM_SPD_PRIO_LOOP   // (A/B builds: priority at which the row loop starts; the prologue always runs at 3)
#define MM_SPD_PRIO_LOOP 3
#endif
  // The row reduction leaves one total per lane; NP of the lanes hold distinct entries, the others duplicates.  All
  // lanes store (an exec-masked store costs two scalar instructions per row): writers into redM, advancing by one row
  // per row, the others into a slot of their own that does not move.
  __shared__ T redJunk[NW][64];
  T* red_ptr = red_writer ? &redM[wave][0][red_slot] : &redJunk[wave][lane];
  const int red_step = red_writer ? NP : 0;

  while (rem > 0) {   // one pass per column block of this workgroup's share
    const int jbase = cb * (64 * NC);
    int jv[NC];          // column for the validity test (never above a row for lanes beyond n)
    unsigned joff[NC];   // byte offset of min(column, n - 1) in a row of the pair vector
    T xj[NC][NP], accJ[NC][D][D];
    static_for<NC>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int j = jbase + 64 * q + lane;
      const bool jin = j < n;
      jv[q] = jin ? j : INT32_MIN;
      asm volatile("" : "+v"(jv[q]));   // (or the select is undone into `jin && j > row`: a scalar AND per row and column)
      joff[q] = unsigned(min(j, n - 1)) * unsigned(sizeof(T));
      int jn = j;                       // the column's node
      if constexpr (SUB) {
        jn = batch_node(idx32, size_t(min(j, n - 1)), n_total);
        joff[q] = unsigned(jn) * unsigned(sizeof(T));   // its offset in a row of the dense target matrix
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) xj[q][k] = T(0);
#pragma unroll
      for (int k = 0; k < D; ++k) xj[q][pidx(k, k)] = T(1);
      if (jin) {
#pragma unroll
        for (int k = 0; k < NP; ++k) xj[q][k] = nodeY[size_t(jn) * NP + k];
      }
#pragma unroll
      for (int rr = 0; rr < D; ++rr)
#pragma unroll
        for (int c = 0; c < D; ++c) accJ[q][rr][c] = T(0);
    });
    const int hi = walk.hi(cb);
    // The rows of this block that belong to the share are cut into NW CONTIGUOUS slices, one per wavefront, and a wavefront
    // walks its slice in one go: the requests for the pair vector run kAhead rows ahead across the whole slice — the row sums
    // leave every TI rows (segments), without restarting the request pipeline.  (Rounds 2-4 cut the rows into chunks of NW x TI
    // and gave every wavefront TI rows of each chunk: every 16 rows a wavefront set its slice up again (~100 scalar
    // instructions), issued its first requests right in front of their use and — vmcnt counts in order — waited behind the
    // atomics of the flush it had just issued: the memory latency was exposed once per 16 rows.  MM_SPD_BWD_CHUNKED restores
    // that form for A/B builds; profiles/r05_experiments.md.)  A slice's byte offsets are 32-bit: rows x n x sizeof(T) < 2^31.
#ifdef MM_SPD_BWD_CHUNKED
    const int slice_cap = TI;
#else
    const int slice_cap = max(TI, int(min(int64_t(1) << 20, (int64_t(1) << 31) / (int64_t(n) * int64_t(sizeof(T))))));
#endif
    while (rem > 0 && r < hi) {   // (one pass, unless the 32-bit cap cuts the block's rows)
      const int chunk = int(min(int64_t(min(hi - r, rem)), int64_t(NW) * slice_cap));
      const int tw = (chunk + NW - 1) / NW;
      const int i0 = min(r + wave * tw, r + chunk), i1 = min(i0 + tw, r + chunk);
      if (i0 < i1) {
        // EVERY instruction of a wavefront — scalar ones included — takes an issue slot of its SIMD (these kernels run
        // at 2.0 - 2.4 cycles per instruction of any kind), so the row loop is written for the smallest TOTAL:
        // * row operands {L_i^-1, L_i}: one interleaved table, one running scalar pointer, two scalar loads per row
        //   into one of two register sets that alternate with the unrolled slots (no copies);
        // * the pair vector: element (row, j) lives at pair_off(n, row) - base + (j - row - 1).  Requests are
        //   `global_load_dword v, v_off, s[ptr]`: a slice base (64-bit, once per slice) plus a 32-bit running row offset
        //   that stops at the slice's last row (requests past it — the unrolled loop issues kAhead - 1 of them — read
        //   that row again: rows beyond the launch's range may lie outside the caller's buffer), and a lane offset that
        //   is FIXED for the slice, clamp(j, i0 + 1, n - 1) elements: for a later row of the slice a lane at or below
        //   the diagonal then reads an element of an earlier row, still inside the buffer, and its value is masked at
        //   use.  Requests are unconditional (a predicated one is an exec-masked branch behind which the compiler waits
        //   for vmcnt(0)) and run kAhead rows ahead into registers that rotate by RENAMING (the loop is unrolled kAhead
        //   times): rotating with moves would wait for the NEWEST request at every row.
#ifndef MM_SPD_BWD_AHEAD
#define MM_SPD_BWD_AHEAD 2
#endif
        constexpr int kAhead = MM_SPD_BWD_AHEAD;   // (A/B builds: -DMM_SPD_BWD_AHEAD=4)
        // SUB: the node of a row comes from the index vector — scalar loads issued one row ahead of their use (the operand
        // table) resp. right behind the previous request (the target row)
        const int first_node = SUB ? __builtin_amdgcn_readfirstlane(batch_node(idx32, size_t(i0), n_total)) : i0;
        int next_node = SUB ? __builtin_amdgcn_readfirstlane(batch_node(idx32, size_t(min(i0 + 1, n - 1)), n_total)) : 0;   // node of row i0 + 1
        unsigned roff = unsigned(first_node) * unsigned(2 * NP * sizeof(T));   // byte offset of the row's operands (the table is < 4 GB)
        T lrow[kAhead][2 * NP];   // (register sets that alternate with the unrolled slots)
#pragma unroll
        for (int k = 0; k < 2 * NP; ++k) lrow[0][k] = nodeLC[size_t(first_node) * (2 * NP) + k];
        const int glast = min(i1, walk.re) - 1;
        const int64_t gk = glast - i0;
        const unsigned gmax = unsigned((gk * (n - 2) - (int64_t(i0) * gk + gk * (gk - 1) / 2)) * int64_t(sizeof(T)));
        const char* gslice = reinterpret_cast<const char*>(g + (pair_off(n, i0) - base - i0 - 1));
        unsigned goff = 0, gstep = unsigned(n - i0 - 2) * unsigned(sizeof(T));   // bytes from the current row to the next
        unsigned jslice[NC];
        static_for<NC>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          jslice[q] = max(joff[q], unsigned(i0 + 1) * unsigned(sizeof(T)));
        });
        T gq[kAhead][NC];
        int req_row = i0, req_node = first_node;   // SUB: the row the next request is for, and its node
        auto request = [&](T (&dst)[NC]) __attribute__((always_inline)) {
          if constexpr (SUB) {
            // target of (row, column) = dense[node(row)][node(column)]: a gather, one row of the dense matrix per request
            const char* grow = reinterpret_cast<const char*>(g) + size_t(req_node) * size_t(n_total) * sizeof(T);
            static_for<NC>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              dst[q] = *reinterpret_cast<const T*>(grow + joff[q]);
            });
            req_row = min(req_row + 1, glast);   // (requests past the slice read its last row again; never used)
            req_node = __builtin_amdgcn_readfirstlane(batch_node(idx32, size_t(req_row), n_total));
          } else {
            static_for<NC>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              asm volatile("" : "+v"(jslice[q]));   // (see the forward's store: keeps the `v_off, s[ptr]` form inside the loop)
              dst[q] = *reinterpret_cast<const T*>(gslice + goff + jslice[q]);
            });
#ifndef MM_DIAG_STATIC_ROWS   // (diagnostic builds: every row of a slice reads the slice's first row of g)
            goff = min(goff + gstep, gmax);
            gstep -= unsigned(sizeof(T));
#endif
          }
        };
#ifndef MM_DIAG_NO_G
#pragma unroll
        for (int u = 0; u < kAhead; ++u) request(gq[u]);
#endif
        MM_SPD_STAMP_MARK(1);
#if MM_SPD_PRIO_LOOP != 3
        if (phase == 0) __builtin_amdgcn_s_setprio(MM_SPD_PRIO_LOOP);   // (the prologue ran at 3: a workgroup that arrives outranks the row loops of the older ones)
#endif
        for (int s0 = i0; s0 < i1; s0 += TI) {   // segments of TI rows: their row sums are staged in LDS and leave together
        const int s1 = min(s0 + TI, i1);
        for (int ib = s0; ib < s1; ib += kAhead) {
#pragma unroll
         for (int u = 0; u < kAhead; ++u) {
          // (a slice with an odd number of rows runs its last unrolled slot on a masked row: an early exit here would
          // make the number of outstanding requests path-dependent and the compiler falls back to vmcnt(0))
          const int irow = ib + u;
          const int ieff = (u == 0 || irow < s1) ? irow : INT32_MAX;   // scalar; only the slot after the first can be past the slice (its last segment)
          const T (&lcur)[2 * NP] = lrow[u];
          // the row after the slice is inside the table (i1 <= n - 1), and the row after THAT — requested by the masked last
          // slot of a slice with an odd row count — is at most the table's padding row n (spd_ws.hpp).  The running offset is pinned: otherwise the loop
          // runs on a pointer one iteration ahead and every scalar load pays a 64-bit add for its negative offset
          if constexpr (SUB) {
            roff = unsigned(next_node) * unsigned(2 * NP * sizeof(T));
            next_node = __builtin_amdgcn_readfirstlane(batch_node(idx32, size_t(min(irow + 2, n - 1)), n_total));
          } else {
            roff += unsigned(2 * NP * sizeof(T));
            if constexpr (kAhead > 2) roff = min(roff, unsigned(n) * unsigned(2 * NP * sizeof(T)));   // (more masked slots than the one padding row covers)
          }
          asm volatile("" : "+s"(roff));
          const T* rowp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(nodeLC) + roff);
#pragma unroll
          for (int k = 0; k < 2 * NP; ++k) lrow[(u + 1) % kAhead][k] = rowp[k];
          T li[NP], lc[NP];
#pragma unroll
          for (int k = 0; k < NP; ++k) { li[k] = lcur[k]; lc[k] = lcur[NP + k]; }
#ifdef MM_DIAG_NO_PRIO
          if (false) {
#else
          if (__builtin_expect(--rows_left == 0, 0)) {   // wave-uniform
#endif
            ++phase;
#if MM_SPD_PRIO_LOOP == 3
            if (phase == 1) { __builtin_amdgcn_s_setprio(2); rows_left = prio_stretch; }
            else if (phase == 2) { __builtin_amdgcn_s_setprio(1); rows_left = max(wave_rows - prio_last - 2 * prio_stretch, 1); }
            else { __builtin_amdgcn_s_setprio(0); rows_left = INT32_MAX; }
#else   // (A/B builds: the prologue at 3, the row loop from 2 down — two steps)
            if (phase == 1) { __builtin_amdgcn_s_setprio(1); rows_left = max(wave_rows - prio_last - 2 * prio_stretch, 1) + prio_stretch; }
            else { __builtin_amdgcn_s_setprio(0); rows_left = INT32_MAX; }
#endif
          }
          bool valid[NC];
          T gs[NC], m[NC][NP];
          static_for<NC>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            valid[q] = jv[q] > ieff;
#ifdef MM_DIAG_NO_G
            gs[q] = valid[q] ? T(1) : T(0);
#else
            gs[q] = valid[q] ? gq[u][q] : T(0);  // upstream gradient (or target) of this row
#endif
          });
#ifndef MM_DIAG_NO_G
          request(gq[u]);
#endif
          // the upstream gradient is known before log(A) unless it depends on the distance (fused loss, d instead of d^2)
          constexpr bool g_first = LOSS == MM_LOSS_NONE && SQ;
          auto jacobi_path = [&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
#ifdef MM_DIAG_NO_COLD   // (diagnostic builds, wrong results beyond the gates: what do the cold paths cost the hot one in registers and code?)
#pragma unroll
            for (int k = 0; k < NP; ++k) m[q][k] = T(0);
            return;
#endif
            T w[D], lw[D], v[D][D];
#ifdef MM_SPD4_TWO_COL_NO_SECOND   // (A/B builds: the round-4/5 form — no second, one-sided solve in the two-column SPD(4) backward)
            constexpr int kSecond = (D == 4 && NC == 2) ? 0 : 1;
#elif defined(MM_SPD4_TWO_COL_INLINE_SECOND)
            constexpr int kSecond = 1;
#else
            constexpr int kSecond = (D == 4 && NC == 2) ? 2 : 1;   // two columns: out of line (second_solve_ool)
#endif
            const T s = pair_core<T, D, true, true, kSecond>(li, xj[q], wmin, wmax, w, lw, v);
            gs[q] = upstream_of<T, LOSS>(gs[q], s, valid[q], squared, wmin, sp, la, loss_acc, ds_acc);
            T cm[D];
#pragma unroll
            for (int k = 0; k < D; ++k) cm[k] = (gs[q] + gs[q]) * lw[k];
            vdvt<T, D>(v, cm, m[q]);
          };
          auto finish = [&](auto qc, const T (&m0)[NP], bool scaled) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if (scaled) {
#pragma unroll
              for (int k = 0; k < NP; ++k) m[q][k] = m0[k];
              return;
            }
            T s = T(0);
            if (LOSS != MM_LOSS_NONE || !SQ) s = frob2<T, D>(m0);
            gs[q] = upstream_of<T, LOSS>(gs[q], s, valid[q], squared, wmin, sp, la, loss_acc, ds_acc);
            const T g2 = gs[q] + gs[q];
#pragma unroll
            for (int k = 0; k < NP; ++k) m[q][k] = g2 * m0[k];
          };
          if constexpr (D == 3 || D == 4) {
            // Eigen-free paths, chosen per wavefront: close pairs (||A - I||_F <= 0.3) take the
            // Cayley-Hamilton series of log(I + E) (degree 7 in fp32, 19 in fp64); anything with tr(Z^2) <= 0.36 (eigenvalue ratios up
            // to ~16: every pair of an embedding with O(1) distances) the Cayley-transform logarithm.
            // What is left (very wide spectra, NaN, non-PD) goes to the Jacobi path.
            T a[NC][NP];
            bool far = false;    // some pair of this row is outside the close-pair gate
#ifdef MM_NO_CENTRED   // (A/B builds: tools/snap_make.sh nocentred -DMM_NO_CENTRED)
            constexpr bool kCentred = false;
#else
            constexpr bool kCentred = (D == 3 || D == 4) && std::is_same<T, float>::value;
#endif
            static_for<NC>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              congr_chol<T, D>(li, xj[q], a[q]);
              far = far | !(close_gate<T, D>(a[q]) <= T(kCloseGate));   // (| : no exec-masked short circuit)
            });
            bool far2 = true;    // some pair is outside the recentred series' range (SPD(3) fp32), decided only for far rows
            if (__builtin_expect(!__any(far), 1)) {
              static_for<NC>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                T m0[NP];
                log_close<T, D>(a[q], m0, g_first ? gs[q] + gs[q] : T(1));
                finish(qc, m0, g_first);
              });
              far2 = false;
            } else if constexpr (kCentred) {
              far2 = false;
              static_for<NC>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (D == 3) far2 = far2 | centred_far3<T>(a[q]); else far2 = far2 | centred_far4<T>(a[q]);
              });
              far2 = __any(far2);
              if (!far2) {
                // pairs at moderate distance (spectral radius of A / mu - I up to 0.66): the recentred series — no
                // eigensolve, no inverse (smallmat.hpp, log_series3_centred / log_series4_centred)
                static_for<NC>([&](auto qc) {
                  constexpr int q = decltype(qc)::value;
                  T ac[NP], m0[NP];
#pragma unroll
                  for (int k = 0; k < NP; ++k) {
                    ac[k] = a[q][k];
                    // (the series' arithmetic starts HERE: nothing of it is hoisted above the gates.  The copies are real — six / ten
                    // v_mov per pair of this path; pinning A in place instead costs MORE copies: 12 + 12 -> 19 + 12 + 12, the
                    // other paths then keep their own)
                    asm volatile("" : "+v"(ac[k]));
                  }
                  if constexpr (D == 3) log_series3_centred<T>(ac, m0, g_first ? gs[q] + gs[q] : T(1));
                  else log_series4_centred<T>(ac, m0, g_first ? gs[q] + gs[q] : T(1));
                  finish(qc, m0, g_first);
                });
              }
            }
            if (far2) {
              static_for<NC>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                T m0[NP];
#ifdef MM_DIAG_NO_CAYLEY
                jacobi_path(qc);
                return;
#endif
                const T gate = log_cayley<T, D>(a[q], m0, g_first ? gs[q] + gs[q] : T(1));
                if (__builtin_expect(!__any(!(gate <= T(kCayleyGate))), 1)) finish(qc, m0, g_first); else jacobi_path(qc);
              });
            }
#ifndef MM_SPD2_JACOBI
          } else if constexpr (D == 2) {
            // SPD(2): eigenvalues and logarithm in closed form (smallmat.hpp, log_pair2_chol) — no eigensolve
            static_for<NC>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              T m0[NP];
              const T s = log_pair2_chol<T>(li, xj[q], wmin, wmax, m0);
              gs[q] = upstream_of<T, LOSS>(gs[q], s, valid[q], squared, wmin, sp, la, loss_acc, ds_acc);
              const T g2 = gs[q] + gs[q];
#pragma unroll
              for (int k = 0; k < NP; ++k) m[q][k] = g2 * m0[k];
            });
#endif
          } else if constexpr (D >= 5 && kSeriesMat) {
            // SPD(5 .. 9): matrix-Horner series per wavefront (close pairs, pairs at moderate distance), else Jacobi
            static_assert(NC == 1, "one column per lane for D >= 5");
            // (the fp64 SPD(8) backward keeps the eigensolve at moderate distance: 19 products of four spilling 36-entry double
            // matrices next to the 64 column accumulators took 2383 us against 2002, profiles/r05_experiments.md section 18)
            constexpr bool kWideBwd = !(std::is_same<T, double>::value && D == 8);
            T a[NP], m0[NP];
            congr_chol<T, D>(li, xj[0], a);
            bool done = false;
            if (__builtin_expect(!__any(!(close_gate<T, D>(a) <= T(kCloseGate))), 1)) {
              log_close_mat<T, D>(a, m0);
              done = true;
            } else if (kWideBwd && !__any(centred_far_mat<T, D>(a))) {
              log_centred_mat<T, D>(a, m0);
              done = true;
            }
            if (done) finish(std::integral_constant<int, 0>{}, m0, false); else jacobi_path(std::integral_constant<int, 0>{});
          } else {
            static_for<NC>([&](auto qc) { jacobi_path(qc); });
          }
          // (one shared tail for all paths.  Issuing it separately inside the close-pair branch — so that the paths' results
          // need not merge in front of it — was measured: no gain for SPD(3), +3.7 % for SPD(4); profiles/r03_experiments.md)
          static_for<NC>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            lt_m_lt_acc<T, D>(li, lc, m[q], accJ[q]);   // (the sum rides the FMA chains: 6 adds fewer per pair, -0.8 us)
#pragma unroll
            for (int rr = 0; rr < D; ++rr)
#pragma unroll
              for (int c = 0; c < D; ++c) {
                // pinned here: the column side must have consumed M before the reduction below, whose first levels
                // (v_permlane*_swap) overwrite their operands — otherwise the compiler sinks the congruence behind the
                // reduction and pays a register copy per entry of M to keep them alive
                asm volatile("" : "+v"(accJ[q][rr][c]));
              }
          });
          // row side: ONE transposing reduction of the lane's NC matrices added up — every lane ends up with the
          // wavefront total of one entry of M
#ifndef MM_RED_IN_PLACE   // (A/B builds: the round-2..4 form, the second column added into the first one's registers)
          // The columns' sum goes into FRESH registers: added into m[0] in place, every v_permlane*_swap of the reduction's first
          // level got two register copies in front of it (the swap overwrites both operands and the compiler kept m[0] alive) —
          // 12 copies per two rows; headline backward 41.5 -> 40.5 us, mid-training 54.2 -> 51.7, SPD(4) n = 16 384 934 -> 892
          // (profiles/r05_experiments.md section 20).  SPD(2) keeps the in-place form (26.5 against 26.8 us).
          constexpr bool kFreshSums = NC >= 2 && D >= 3;
#else
          constexpr bool kFreshSums = false;
#endif
          if constexpr (kFreshSums) {
          T ms[NP];
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            ms[k] = m[0][k];
            static_for<NC - 1>([&](auto qc) { ms[k] += m[decltype(qc)::value + 1][k]; });
          }
          *red_ptr = wave_reduce_transposed<NP, T>(ms, lane);
          } else {
          static_for<NC - 1>([&](auto qc) {
            constexpr int q = decltype(qc)::value + 1;
#pragma unroll
            for (int k = 0; k < NP; ++k) m[0][k] += m[q][k];
          });
#ifdef MM_DIAG_NO_RED
          *red_ptr = m[0][0] + m[0][NP - 1];
#else
          *red_ptr = wave_reduce_transposed<NP, T>(m[0], lane);
#endif
          }
          red_ptr += red_step;
         }
        }
        // row side: each wavefront owns the rows of its slice
        const int sw = s1 - s0;
        red_ptr -= red_step * ((sw + kAhead - 1) / kAhead * kAhead);
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < sw * NP; t += 64) {
          const int k = t / sw, il = t - k * sw;
          {
            const int node = SUB ? batch_node(idx32, size_t(s0 + il), n_total) : s0 + il;
#ifndef MM_DIAG_NO_ROW_ATOMICS   // (diagnostic builds, wrong results: what does each part of the row loop's surroundings cost?)
            atomic_add(&accM[size_t(k) * ns + node], redM[wave][il][k]);
#endif
          }
        }
        __builtin_amdgcn_wave_barrier();   // redM is rewritten by the next segment
        }
        MM_SPD_STAMP_MARK(2);
      }
      r += chunk;
      rem -= chunk;
    }
    // column side of this block: combine the wavefronts, then 256-B contiguous atomics per entry
    static_for<NC>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int rr = 0; rr < D; ++rr)
#pragma unroll
        for (int c = 0; c < D; ++c) colS[wave][q][rr * D + c][lane] = accJ[q][rr][c];
    });
    __syncthreads();
    for (int t = wave; t < NC * D * D; t += NW) {
      const int q = t / (D * D), k = t - q * (D * D);
      const int j = jbase + 64 * q + lane;
      T sum = colS[0][q][k][lane];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) sum += colS[wv][q][k][lane];
      if (j < n) {
        const int node = SUB ? batch_node(idx32, size_t(j), n_total) : j;
#ifndef MM_DIAG_NO_COL_ATOMICS
        atomic_add(&accS[size_t(k) * ns + node], sum);
#else
        if (sum == T(12345.678)) accS[0] = sum;
#endif
      }
    }
    ++cb;
    r = row_begin;
    rem -= shares.cross;            // (entering the next block is paid for: ColWalk::enter)
    if (rem > 0) __syncthreads();   // colS is rewritten by the next block
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
  MM_SPD_STAMP_END();
}

// grad_x[i] = sym(accS_i X_i^-1) - L_i^-T accM_i L_i^-1   (symmetric; packed in gi).  Reads the accumulators of node i and
// leaves them zero for the next backward.  Shared by spd_pdist_finalize_kernel and the fused optimizer kernels.
template <typename T, int D>
__device__ __forceinline__ void node_gradient(const T* __restrict__ nodeL, T* __restrict__ accM, T* __restrict__ accS, int n,
                                              int i, T (&gi)[Packed<D>::NP]) {
  constexpr int NP = Packed<D>::NP;
  T li[NP], m[NP], xinv[NP], sc[D][D];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    li[k] = nodeL[size_t(i) * NP + k];
    m[k] = accM[size_t(k) * n + i];
    accM[size_t(k) * n + i] = T(0);  // leave the accumulators clean for the next backward
  }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      sc[r][c] = accS[size_t(r * D + c) * n + i];
      accS[size_t(r * D + c) * n + i] = T(0);
    }
  congr_lower_t<T, D>(li, m, gi);   // L^-T M L^-1
#pragma unroll
  for (int r = 0; r < D; ++r)       // X^-1 = L^-T L^-1
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T acc = T(0);
#pragma unroll
      for (int k = r; k < D; ++k) acc = Num<T>::fma(li[pidx(k, r)], li[pidx(k, c)], acc);
      xinv[pidx(r, c)] = acc;
    }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      T a = T(0), b = T(0);         // (S X^-1)[r][c] and (S X^-1)[c][r]
#pragma unroll
      for (int k = 0; k < D; ++k) {
        a = Num<T>::fma(sc[r][k], xinv[pidx(k, c)], a);
        b = Num<T>::fma(sc[c][k], xinv[pidx(k, r)], b);
      }
      gi[pidx(r, c)] = T(0.5) * (a + b) - gi[pidx(r, c)];
    }
}

template <typename T, int D>
__global__ void spd_pdist_finalize_kernel(const T* __restrict__ nodeL, T* __restrict__ accM,
                                          T* __restrict__ accS, int n, T* __restrict__ grad,
                                          T* __restrict__ slots, const T* __restrict__ scale_raw,
                                          T* __restrict__ loss_out) {
  constexpr int NP = Packed<D>::NP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (slots && blockIdx.x == 0 && threadIdx.x < 64) loss_finalize<T>(slots, scale_raw, loss_out);
  if (i >= n) return;
  T gi[NP];
  node_gradient<T, D>(nodeL, accM, accS, n, i, gi);
  store_sym_full<T, D>(grad + size_t(i) * D * D, gi);
}

// ------------------------------------------------------- element-wise dist
template <typename T, int D>
__global__ void spd_dist_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y, int64_t m, int squared, T wmin,
                                    T wmax, T* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], ys[NP], l[NP], li[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(y + k * D * D, ys);
  cholesky<T, D>(xs, l);
  invert_lower<T, D>(l, li);
  T s;
  if (spd_clamps_wide(wmin, wmax)) {
    s = pair_value<T, D>(li, ys, wmin, wmax);
  } else {   // (the eigen-free paths of pair_value do not see eigenvalues: a window that could bind takes the eigensolve)
    T w[D], lw[D], v[D][D];
    s = pair_core<T, D, false>(li, ys, wmin, wmax, w, lw, v);
  }
  s = Num<T>::max(s, wmin);
  if (!squared) s = Num<T>::sqrt(s);
  if (in) out[k] = s;
}

template <typename T, int D>
__global__ void spd_dist_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ g,
                                    int64_t m, int squared, T wmin, T wmax, T* __restrict__ gx,
                                    T* __restrict__ gy) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], ys[NP], l[NP], li[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(y + k * D * D, ys);
  cholesky<T, D>(xs, l);
  invert_lower<T, D>(l, li);
  T w[D], lw[D], v[D][D], rho[D];
  const T s = pair_core<T, D, true>(li, ys, wmin, wmax, w, lw, v, rho);
  T gs = g[k];
  if (!squared) gs *= T(0.5) * Num<T>::rsqrt(Num<T>::max(s, wmin));
  T cm[D], cn[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    cm[q] = -(gs + gs) * lw[q];
    cn[q] = -cm[q] / w[q];
    cm[q] *= rho[q];   // (x side: log(w_c) w / w_c, see pair_core)
  }
  T mm_[NP], nn[NP], o[NP];
  vdvt<T, D>(v, cm, mm_);
  vdvt<T, D>(v, cn, nn);
  if (in) {
    congr_lower_t<T, D>(li, mm_, o);
    store_sym_full<T, D>(gx + k * D * D, o);
    congr_lower_t<T, D>(li, nn, o);
    store_sym_full<T, D>(gy + k * D * D, o);
  }
}

// ------------------------------------------------------ per-point maps
template <typename T, int D> __device__ __forceinline__ void load_full(const T* __restrict__ p, T (&f)[D * D]) {
#pragma unroll
  for (int k = 0; k < D * D; ++k) f[k] = p[k];
}

// expand packed lower-triangular to a full row-major matrix
template <typename T, int D> __device__ __forceinline__ void lower_to_full(const T (&l)[Packed<D>::NP], T (&f)[D * D]) {
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) f[r * D + c] = (c <= r) ? l[pidx(r, c)] : T(0);
}

template <typename T, int D>
__global__ void spd_map_kernel(int op, const T* __restrict__ x, const T* __restrict__ u, int64_t m, T wmin, T wmax,
                               T* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], us[NP], o[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  if (op != MM_SPD_PROJX) load_sym_packed<T, D>(u + k * D * D, us);
  if (op == MM_SPD_EGRAD2RGRAD) {
    T xf[D * D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) xf[r * D + c] = xs[pidx(r, c)];
    congr_full<T, D>(xf, us, o);
  } else if (op == MM_SPD_PROJU) {
#pragma unroll
    for (int q = 0; q < NP; ++q) o[q] = us[q];
  } else if (op == MM_SPD_PROJX) {
    T v[D][D], f[D];
    jacobi_eig<T, D, true>(xs, v);
#pragma unroll
    for (int q = 0; q < D; ++q) f[q] = Num<T>::min(Num<T>::max(xs[pidx(q, q)], wmin), wmax);
    vdvt<T, D>(v, f, o);
  } else {
    T l[NP], li[NP];
    cholesky<T, D>(xs, l);
    invert_lower<T, D>(l, li);
    if (op == MM_SPD_RETR) spd_retr<T, D>(xs, li, us, o);
    else if (op == MM_SPD_EXP) spd_explog<T, D, false>(l, li, us, o);
    else spd_explog<T, D, true>(l, li, us, o);
  }
  if (in) store_sym_full<T, D>(out + k * D * D, o);
}

template <typename T, int D>
__global__ void spd_norm_kernel(const T* __restrict__ x, const T* __restrict__ u, int64_t m, int squared,
                                T* __restrict__ out) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], us[NP], l[NP], li[NP], a[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(u + k * D * D, us);
  cholesky<T, D>(xs, l);
  invert_lower<T, D>(l, li);
  congr_lower<T, D>(li, us, a);
  T s = T(0);
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) s += (r == c ? T(1) : T(2)) * a[pidx(r, c)] * a[pidx(r, c)];
  if (in) out[k] = squared ? s : Num<T>::sqrt(s);
}

// Eigenvalues of sym(X), ascending: SymmetricPositiveDefinite.symeig (spd.py:35-41, 63-64; fast.symeig2x2 / symeig3x3 for
// n = 2, 3 in the reference, LAPACK on the CPU otherwise) — one symmetric matrix per lane, cyclic Jacobi to machine precision.
template <typename T, int D>
__global__ void spd_eigvalsh_kernel(const T* __restrict__ x, int64_t m, T* __restrict__ w) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T a[NP], v[D][D];
  load_sym_packed<T, D>(x + k * D * D, a);
  jacobi_eig<T, D, false>(a, v);
  T e[D];
#pragma unroll
  for (int r = 0; r < D; ++r) e[r] = a[pidx(r, r)];
#pragma unroll
  for (int i = 0; i < D; ++i)          // (odd-even transposition sort: branch-free compare-exchanges)
#pragma unroll
    for (int j = (i & 1); j + 1 < D; j += 2) {
      const T lo = Num<T>::min(e[j], e[j + 1]), hi = Num<T>::max(e[j], e[j + 1]);
      e[j] = lo; e[j + 1] = hi;
    }
  if (in) {
#pragma unroll
    for (int r = 0; r < D; ++r) w[k * D + r] = e[r];
  }
}

template <typename T, int D>
// (x and xnew are deliberately not __restrict__: the update may be done in place, xnew == x — every thread
// reads its whole point before it writes it)
__global__ void spd_rsgd_step_kernel(const T* x, const T* __restrict__ eg, int64_t m, T lr,
                                     T max_grad_norm, int exact, T* xnew) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], gs[NP], o[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(eg + k * D * D, gs);
  spd_rsgd_update<T, D>(xs, gs, lr, max_grad_norm, exact, o);
  if (in) store_sym_full<T, D>(xnew + k * D * D, o);
}

template <typename T, int D>
__global__ void spd_rsgd_momentum_kernel(const T* x, const T* __restrict__ eg, T* buf, int64_t m, T lr, T momentum,
                                         T dampening, T max_grad_norm, int exact, T* xnew) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T xs[NP], gs[NP], o[NP], b[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(eg + k * D * D, gs);
  load_sym_packed<T, D>(buf + k * D * D, b);
  spd_momentum_update<T, D>(xs, gs, b, lr, momentum, dampening, max_grad_norm, exact, o);
  if (in) {
    store_sym_full<T, D>(xnew + k * D * D, o);
    store_sym_full<T, D>(buf + k * D * D, b);
  }
}

template <typename T, int D>
__global__ void spd_radam_step_kernel(const T* x, const T* __restrict__ eg, T* exp_avg, T* exp_avg_sq, int64_t m,
                                      AdamArgs<T> a, T* xnew) {
  constexpr int NP = Packed<D>::NP;
  const int64_t k0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = k0 < m;
  const int64_t k = in ? k0 : 0;
  T beta2, alpha;
  adam_coeffs(a, beta2, alpha);
  T xs[NP], gs[NP], o[NP], mo[NP];
  load_sym_packed<T, D>(x + k * D * D, xs);
  load_sym_packed<T, D>(eg + k * D * D, gs);
  load_sym_packed<T, D>(exp_avg + k * D * D, mo);
  const T v = spd_adam_update<T, D>(xs, gs, mo, exp_avg_sq[k * D * D], a, beta2, alpha, o);
  if (in) {
    store_sym_full<T, D>(xnew + k * D * D, o);
    store_sym_full<T, D>(exp_avg + k * D * D, mo);
#pragma unroll
    for (int q = 0; q < D * D; ++q) exp_avg_sq[k * D * D + q] = v;
  }
  adam_tick(a.step, a.ticket, gridDim.x);
}

// ---- the fused training-step kernel: gradient of a point from the pair kernel's accumulators (what
// spd_pdist_finalize_kernel does) -> optimizer rule -> new point -> ITS per-node tables for the next step's pair
// kernels (what spd_prep_kernel does), one thread per point; block 0 also closes the loss record and, when asked,
// applies the scale parameter's momentum-free RSGD update (Euclidean(1): rsgd.py:56-68 with base.py:29-33's norm).
// A step is then TWO launches — the pair kernel and this one — instead of prep + pair + finalize + point update +
// scale update (train.py:198-222 is ~60 framework launches in the reference).
enum { RULE_RSGD = 0, RULE_MOMENTUM = 1, RULE_ADAM = 2 };
template <typename T> struct StepRule {
  T lr, momentum, dampening, max_grad_norm;
  int exact;
  T* state0;          // momentum buffer / exp_avg
  T* state1;          // exp_avg_sq
  AdamArgs<T> adam;
};
template <typename T> struct StepFuse {
  // gradient source: FIN -> the accumulators (finalize arithmetic here; the Euclidean gradient is also stored to grad);
  // else the finished gradient is read from grad
  const T* nodeL; T* accM; T* accS;
  T* grad;
  T* slots; const T* scale_raw; T* loss_out;     // FIN: the loss record, closed by block 0
  T* scale_x; T scale_lr, scale_clip;            // FIN: the scale's own RSGD update (null: stepped elsewhere / frozen)
  // PREP: tables of the new points
  T* tabL; T* tabX; T* tabC; int* bad; T* tabLd; T* tabLC;
};
template <typename T, int D, int RULE, bool FIN, bool PREP>
__global__ void spd_fused_step_kernel(T* x, int n, StepRule<T> rule, StepFuse<T> f) {
  constexpr int NP = Packed<D>::NP;
  if constexpr (FIN) {
    if (f.slots && blockIdx.x == 0 && threadIdx.x < 64) {
      loss_finalize<T>(f.slots, f.scale_raw, f.loss_out);
      if (f.scale_x && threadIdx.x == 0) {
        // vec_rsgd_point<T, MM_EUCLIDEAN> for one scalar: r = g, ||r|| = sqrt(max(g^2, 1e-8)), x' = x - lr clip r
        const T g = f.loss_out[1];
        T scale = -f.scale_lr;
        if (f.scale_clip > T(0)) scale *= Num<T>::min(f.scale_clip / Num<T>::sqrt(Num<T>::max(g * g, T(1e-8))), T(1));
        *f.scale_x = *f.scale_x + g * scale;
      }
    }
  }
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i0 < n;
  const int i = in ? i0 : 0;
  T beta2 = T(0), alpha = T(0);
  if constexpr (RULE == RULE_ADAM) adam_coeffs(rule.adam, beta2, alpha);
  T xs[NP], gs[NP], o[NP];
  load_sym_packed<T, D>(x + size_t(i) * D * D, xs);
  if constexpr (FIN) {
    if (in) {
      node_gradient<T, D>(f.nodeL, f.accM, f.accS, n, i, gs);
      store_sym_full<T, D>(f.grad + size_t(i) * D * D, gs);
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) gs[k] = T(0);
    }
  } else {
    load_sym_packed<T, D>(f.grad + size_t(i) * D * D, gs);
  }
  if constexpr (RULE == RULE_RSGD) {
    spd_rsgd_update<T, D>(xs, gs, rule.lr, rule.max_grad_norm, rule.exact, o);
  } else if constexpr (RULE == RULE_MOMENTUM) {
    T b[NP];
    load_sym_packed<T, D>(rule.state0 + size_t(i) * D * D, b);
    spd_momentum_update<T, D>(xs, gs, b, rule.lr, rule.momentum, rule.dampening, rule.max_grad_norm, rule.exact, o);
    if (in) store_sym_full<T, D>(rule.state0 + size_t(i) * D * D, b);
  } else {
    T mo[NP];
    load_sym_packed<T, D>(rule.state0 + size_t(i) * D * D, mo);
    const T v = spd_adam_update<T, D>(xs, gs, mo, rule.state1[size_t(i) * D * D], rule.adam, beta2, alpha, o);
    if (in) {
      store_sym_full<T, D>(rule.state0 + size_t(i) * D * D, mo);
#pragma unroll
      for (int q = 0; q < D * D; ++q) rule.state1[size_t(i) * D * D + q] = v;
    }
  }
  if (in) {
    store_sym_full<T, D>(x + size_t(i) * D * D, o);
    if constexpr (PREP) node_tables<T, D>(o, i, f.tabL, f.tabX, f.tabC, f.bad, f.tabLd, f.tabLC);
  }
  if constexpr (RULE == RULE_ADAM) adam_tick(rule.adam.step, rule.adam.ticket, gridDim.x);
}

// ------------------------------------------------------------------ launchers
#define MM_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return static_cast<int>(e_); \
  } while (0)

// Threads per workgroup of the per-node kernels (prep, finalize): n = 5000 nodes are 40 workgroups of 128 or 79 of 64
#ifndef MM_NODE_BLOCK
#define MM_NODE_BLOCK 128
#endif
constexpr int kNodeBlock = MM_NODE_BLOCK;
template <typename T, int D>
int spd_pdist_prepare(const T* x, int64_t n, Ws<T>& ws, int flags, hipStream_t st) {
  if (!(flags & MM_WS_PREPARED)) {
    spd_prep_kernel<T, D><<<dim3((n + kNodeBlock - 1) / kNodeBlock), dim3(kNodeBlock), 0, st>>>(x, int(n), ws.nodeL, ws.nodeX, ws.nodeC,
                                                                       ws.accM, ws.accS, ws.loss, ws.bad, ws.nodeLd, ws.nodeLC);
    MM_CHECK_LAUNCH();
  }
  return MM_OK;
}

#ifndef MM_FWD_TI   // (A/B builds)
#define MM_FWD_TI 8
#endif
constexpr int kFwdTI = MM_FWD_TI;   // rows of a forward tile (sweep on MI355X, SPD(3) fp32, n = 5000: 8 / 16 / 32 rows -> 28.8 / 30.1 / 33.0 us)
template <typename T, int D, int TI>
int spd_pdist_fwd_launch(const T* nl, const T* nc, const T* nld, int64_t n, int64_t rb, int64_t re, int squared, double wmin, double wmax,
                         T* out, hipStream_t st) {
  if (!spd_clamps_supported<D, MM_LOSS_NONE>(wmin, wmax)) return MM_ERR_UNSUPPORTED;
  const dim3 grid = fold_grid<TI, kBlock * pair_cols<T, D>()>(n, rb, re);
  if (squared)
    launch_timed(PROF_SPD_FWD, spd_pdist_fwd_kernel<T, D, TI, true>, grid, dim3(kBlock), st, nl, nc, nld, int(n), int(rb), int(re),
                 T(wmin), T(wmax), out);
  else
    launch_timed(PROF_SPD_FWD, spd_pdist_fwd_kernel<T, D, TI, false>, grid, dim3(kBlock), st, nl, nc, nld, int(n), int(rb), int(re),
                 T(wmin), T(wmax), out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
template <typename T, int D>
int spd_pdist_fwd_t(const T* x, int64_t n, int64_t rb, int64_t re, int squared, double wmin, double wmax, T* out,
                    void* wsp, int flags, hipStream_t st) {
  Ws<T> ws(wsp, n, D);
  int rc = spd_pdist_prepare<T, D>(x, n, ws, flags, st);
  if (rc) return rc;
  if (re <= rb || pair_off(n, re) == pair_off(n, rb)) return MM_OK;
  const T* nl = ws.nodeL;
  const T* nc = ws.nodeC;
  // A small launch (a rank's shard of a small problem, a small graph) is as long as ONE workgroup's tile: with fewer
  // than ~4 workgroups per CU at 8 rows, tiles of 2 rows make it four times shorter (one eighth of the 5000-node problem:
  // 420 workgroups of 8 rows, 7 us, against 3 us pro rata).  D <= 4 only: the instantiations are not free to compile.
  if constexpr (D <= 4) {
    const dim3 g8 = fold_grid<kFwdTI, kBlock * pair_cols<T, D>()>(n, rb, re);
    if (int64_t(g8.x) * g8.y < 4 * int64_t(device_cus()))
      return spd_pdist_fwd_launch<T, D, 2>(nl, nc, ws.nodeLd, n, rb, re, squared, wmin, wmax, out, st);
  }
  return spd_pdist_fwd_launch<T, D, kFwdTI>(nl, nc, ws.nodeLd, n, rb, re, squared, wmin, wmax, out, st);
}

// One launch of (at most) the resident capacity; fewer workgroups when the row range is small (>= 8 rows of a column
// block per workgroup, two per wavefront).
constexpr int kBwdTI = 16;   // rows per wavefront and chunk
// Rows a share pays for entering a column block (ColWalk::enter).  Sweep on the headline backward (fp32 SPD(3), n = 5000, us):
// 0 / 4 / 8 / 12 / 16 / 24 rows -> 45.4 / 45.3 / 44.3 / 43.7 / 43.6 / 43.5; the launch's span 43.0 -> 41.2, its last workgroups
// 42.8 - 43.0 -> 40.8 - 41.2 (profiles/r05_experiments.md).  An fp64 row takes twice the time: half the rows.
template <typename T> constexpr int bwd_cross_units() { return sizeof(T) == 4 ? 16 : 8; }
// fp32 SPD(4): one column per lane by default (128 registers, four wavefronts per SIMD); two columns — scalar bookkeeping,
// row-operand loads and the transposing reduction paid once per 128 pairs, three wavefronts per SIMD — win on large launches
// only (fused QuotientLoss step, us, one / two columns: n = 2274 58.6 / 71.6, 4096 109 / 108, 5793 170 / 174-197,
// 8192 301 / 295, 11585 571 / 549, 16384 1082-1096 / 1039-1041; profiles/r03_experiments.md §15): from 30 M pairs per launch
// on.  MM_SPD4_BWD_TWO_COLS=0 / 1 forces either.
constexpr int64_t kSpd4TwoColPairs = 30000000;
constexpr int64_t kSpd4TwoColBandPairs = 12000000;
// SUB (node minibatch): `ws` is the workspace of the FULL embedding (n_total points), n the batch size, g the dense target matrix.
template <typename T, int D, int LOSS, bool SQ, int NCX = 0, bool SUB = false>
int spd_pdist_bwd_launch_sq(Ws<T>& ws, const T* g, int64_t n, int64_t rb, int64_t re, double wmin, double wmax,
                            hipStream_t st, LossArgs<T> la, const int64_t* idx = nullptr, int64_t n_total = 0) {
  if (!spd_clamps_supported<D, LOSS>(wmin, wmax)) return MM_ERR_UNSUPPORTED;
  constexpr int kThreads = 64 * bwd_waves<T, D>();
  constexpr int kCols = NCX ? NCX : pair_cols_bwd<T, D>();
  if constexpr (!SUB && NCX == 0 && sizeof(T) == 4 && D == 4 && pair_cols_bwd<T, D>() == 1) {
    static const int two = [] { const char* e = std::getenv("MM_SPD4_BWD_TWO_COLS"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    // (a rank's BAND of rows of a larger problem — every column block as tall as the band — takes two columns from 12 M
    // pairs on: the N = 8 shards of n = 16384, 16.8 M pairs each, 182 - 199 -> 173 - 179 us; the last rank's TRIANGLE is like
    // a full problem of its size and keeps one: 198 / 200 us; profiles/r04_experiments.md)
    const int64_t pairs = pair_off(n, re) - pair_off(n, rb);
    if (two == 1 || (two < 0 && (pairs >= kSpd4TwoColPairs || (pairs >= kSpd4TwoColBandPairs && re + 2048 <= n))))
      return spd_pdist_bwd_launch_sq<T, D, LOSS, SQ, 2>(ws, g, n, rb, re, wmin, wmax, st, la);
  }
  auto kernel = spd_pdist_bwd_kernel<T, D, kBwdTI, LOSS, SQ, NCX, SUB>;
  const int64_t units = ColWalk(int(n), int(rb), int(re), 64 * kCols).total();
  if (units <= 0) return MM_OK;
  int64_t grid = resident_workgroups<spd_pdist_bwd_kernel<T, D, kBwdTI, LOSS, SQ, NCX, SUB>>(kThreads);
  // Small launches (a rank's shard, small n): a workgroup flushes its column-side sums once per column block, so it needs
  // enough rows to pay for that — with too few rows of a column block per workgroup the launch is made of prologues and flushes
  // (round 3, one eighth of the headline problem, 13 k units: 1024 workgroups 23.8 us, 256 workgroups 17.1 us; the rule then
  // was 48 rows and whole multiples of the CU count; tools/gpu_shard_grid.sh).  At least one workgroup per CU.
  {
    const int64_t cus = device_cus();
    // at most four workgroups per CU: the kernels with few registers (SPD(2): 55, seven workgroups per CU resident) are no
    // faster for more, shorter shares — every share pays its prologue and flush (SPD(2) n = 5000 backward, 1792 / 1280 / 1024 /
    // 768 workgroups: 29.7 / 27.9 / 26.5 / 27.1 us fp32, 115 / 112 / 111 / 112 fp64; profiles/r05_experiments.md)
    grid = std::min<int64_t>(grid, 4 * cus);
    // (round 5, after the prologue diet: ~32 rows per workgroup for the two-column kernels, ~48 for the others, to the nearest
    // half multiple of the CU count — SPD(3) n = 2000, 256 / 384 / 512 / 768 / 1024 workgroups 16.8 / 15.9 / 14.9 / 15.7 / 18.6 us;
    // one eighth of the headline problem 16.6 / 14.8 / 15.0 / 16.5 / 18.5; SPD(4) n = 2274 38.8 / 34.0 / 28.0 / 25.7 / 26.6)
    // (the one-column kernels keep whole multiples at 48 rows: SPD(4) n = 2274 at 896 workgroups 27.0 us)
    const int64_t half = std::max<int64_t>(1, cus / 2);
    const int64_t by_rows = kCols >= 2 ? (units / 32 + half / 2) / half * half : units / 48 / cus * cus;
    if (by_rows < grid) grid = std::max<int64_t>(cus, by_rows);
  }
  static const int64_t env_grid = std::getenv("MM_SPD_BWD_GRID") ? std::atoll(std::getenv("MM_SPD_BWD_GRID")) : 0;
  if (env_grid > 0) grid = env_grid;   // (experiments: over- / under-subscription of the device)
  grid = std::max<int64_t>(1, std::min<int64_t>(grid, (units + 7) / 8));
  dim3 g3{unsigned(grid), 1, 1};
  const T* nlc = ws.nodeLC;
  const T* nc = ws.nodeC;
  // block-entry cost of the shares (ColWalk::enter), in rows of a block: MM_SPD_BWD_CROSS overrides (0 = equal shares)
  static const int env_cross = std::getenv("MM_SPD_BWD_CROSS") ? std::atoi(std::getenv("MM_SPD_BWD_CROSS")) : bwd_cross_units<T>();
  const ColWalk hw(int(n), int(rb), int(re), 64 * kCols);
  const int cross = std::min(std::max(env_cross, 0), 1024);
  const WalkShares shares(hw.total_aug(cross), grid, cross);
  launch_timed(PROF_SPD_BWD, kernel, g3, dim3(kThreads), st, nlc, nc, g, int(n), int(rb), int(re), T(wmin), T(wmax), ws.accM, ws.accS, la,
               idx, int(n_total), shares, ws.shareTab);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
template <typename T, int D, int LOSS = MM_LOSS_NONE>
int spd_pdist_bwd_launch(Ws<T>& ws, const T* g, int64_t n, int64_t rb, int64_t re, int squared, double wmin, double wmax,
                         hipStream_t st, LossArgs<T> la = LossArgs<T>{nullptr, T(1), T(0), 0, nullptr}) {
  if constexpr (LOSS != MM_LOSS_NONE) return spd_pdist_bwd_launch_sq<T, D, LOSS, true>(ws, g, n, rb, re, wmin, wmax, st, la);
  else return squared ? spd_pdist_bwd_launch_sq<T, D, LOSS, true>(ws, g, n, rb, re, wmin, wmax, st, la)
                      : spd_pdist_bwd_launch_sq<T, D, LOSS, false>(ws, g, n, rb, re, wmin, wmax, st, la);
}

// loss + gradients in one pass over the pairs (no pair vector of distances is ever written)
template <typename T, int D>
int spd_pdist_loss_t(int kind, const T* x, const T* target, const T* scale_raw, int64_t n, int64_t rb, int64_t re,
                     double alpha, double eps, int terms, const double* loss_params, double wmin, double wmax, T* loss_out, T* grad, void* wsp,
                     int flags, hipStream_t st) {
  Ws<T> ws(wsp, n, D);
  int rc = spd_pdist_prepare<T, D>(x, n, ws, flags, st);
  if (rc) return rc;
  if (re > rb && pair_off(n, re) > pair_off(n, rb)) {
    LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, ws.loss, loss_params};
    if (kind == MM_LOSS_STRESS) rc = spd_pdist_bwd_launch<T, D, MM_LOSS_STRESS>(ws, target, n, rb, re, 1, wmin, wmax, st, la);
    else rc = spd_pdist_bwd_launch<T, D, MM_LOSS_QUOTIENT>(ws, target, n, rb, re, 1, wmin, wmax, st, la);
    if (rc) return rc;
  }
  spd_pdist_finalize_kernel<T, D><<<dim3((n + kNodeBlock - 1) / kNodeBlock), dim3(kNodeBlock), 0, st>>>(ws.nodeL, ws.accM, ws.accS, int(n),
                                                                               grad, ws.loss, scale_raw, loss_out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// ... of a NODE MINIBATCH: the pair kernel in its SUB form over the `bs` batch positions, prepare and finalize over the
// FULL embedding — `grad` [n_total, D, D] comes out complete (zero rows for the nodes outside the batch), which is what
// autograd's index backward of x[idx] produces in the reference (modules.py:86, train.py:206-213).
template <typename T, int D>
int spd_pdist_loss_subset_t(int kind, const T* x, const T* dense, const T* scale_raw, int64_t n_total, const int64_t* idx, int64_t bs,
                            int64_t rb, int64_t re, double alpha, double eps, int terms, const double* loss_params, double wmin,
                            double wmax, T* loss_out, T* grad, void* wsp, int flags, hipStream_t st) {
  Ws<T> ws(wsp, n_total, D);
  int rc = spd_pdist_prepare<T, D>(x, n_total, ws, flags, st);
  if (rc) return rc;
  if (re > rb && pair_off(bs, re) > pair_off(bs, rb)) {
    LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, ws.loss, loss_params};
    if (kind == MM_LOSS_STRESS)
      rc = spd_pdist_bwd_launch_sq<T, D, MM_LOSS_STRESS, true, 0, true>(ws, dense, bs, rb, re, wmin, wmax, st, la, idx, n_total);
    else
      rc = spd_pdist_bwd_launch_sq<T, D, MM_LOSS_QUOTIENT, true, 0, true>(ws, dense, bs, rb, re, wmin, wmax, st, la, idx, n_total);
    if (rc) return rc;
  }
  spd_pdist_finalize_kernel<T, D><<<dim3((n_total + kNodeBlock - 1) / kNodeBlock), dim3(kNodeBlock), 0, st>>>(ws.nodeL, ws.accM, ws.accS, int(n_total),
                                                                                     grad, ws.loss, scale_raw, loss_out);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

template <typename T, int D>
int spd_pdist_bwd_t(const T* x, const T* g, int64_t n, int64_t rb, int64_t re, int squared, double wmin, double wmax,
                    T* grad, void* wsp, int flags, hipStream_t st) {
  constexpr int NP = Packed<D>::NP;
  Ws<T> ws(wsp, n, D);
  int rc = spd_pdist_prepare<T, D>(x, n, ws, flags, st);
  if (rc) return rc;
  if (re > rb && pair_off(n, re) > pair_off(n, rb)) {
    rc = spd_pdist_bwd_launch<T, D>(ws, g, n, rb, re, squared, wmin, wmax, st);
    if (rc) return rc;
  }
  spd_pdist_finalize_kernel<T, D><<<dim3((n + kNodeBlock - 1) / kNodeBlock), dim3(kNodeBlock), 0, st>>>(
      ws.nodeL, ws.accM, ws.accS, int(n), grad, static_cast<T*>(nullptr), static_cast<const T*>(nullptr),
      static_cast<T*>(nullptr));
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// ---- fused training step (spd_step.hpp) ------------------------------------------------------------------------
template <typename T, int D, bool FIN>
int spd_fused_step_launch(const mm_train_step* s, Ws<T>& ws, hipStream_t st, bool fuse_scale) {
  const mm_step_param& p = s->points[0];
  const mm_step_param& q = s->scales[0];
  const int n = int(s->n);
  StepRule<T> rule{T(p.lr), T(p.momentum), T(p.dampening), T(p.max_grad_norm), p.exact, static_cast<T*>(p.state0),
                   static_cast<T*>(p.state1),
                   AdamArgs<T>{T(p.lr), T(p.beta1), T(p.beta2), T(p.adam_eps), T(p.max_grad_norm), p.nc, p.exact, p.step, p.ticket}};
  StepFuse<T> f{ws.nodeL, ws.accM, ws.accS, static_cast<T*>(p.grad),
                ws.loss, static_cast<const T*>(q.x), static_cast<T*>(s->loss_out),
                fuse_scale ? static_cast<T*>(q.x) : nullptr, T(q.lr), T(q.max_grad_norm),
                ws.nodeL, ws.nodeX, ws.nodeC, ws.bad, ws.nodeLd, ws.nodeLC};
  const dim3 grid((n + 127) / 128), block(128);
  T* x = static_cast<T*>(p.x);
  if (p.optimizer == MM_OPT_RADAM) spd_fused_step_kernel<T, D, RULE_ADAM, FIN, true><<<grid, block, 0, st>>>(x, n, rule, f);
  else if (p.momentum != 0.0) spd_fused_step_kernel<T, D, RULE_MOMENTUM, FIN, true><<<grid, block, 0, st>>>(x, n, rule, f);
  else spd_fused_step_kernel<T, D, RULE_RSGD, FIN, true><<<grid, block, 0, st>>>(x, n, rule, f);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// SUB: a node minibatch (s->batch_idx): the pair kernel runs over the batch, the per-point kernel over ALL points — the
// nodes outside the batch have zero gradients, and an optimizer with state (momentum, Adam) still moves them, exactly as
// the reference's dense x.grad makes its optimizers do (train.py:218-222).
template <typename T, int D, bool SUB = false>
int spd_fused_train_step_t(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st,
                           bool* scale_stepped) {
  const int64_t n = s->n;
  Ws<T> ws(s->ws, n, D);
  *scale_stepped = false;
  if (!with_objective) return spd_fused_step_launch<T, D, false>(s, ws, st, false);
  int rc = spd_pdist_prepare<T, D>(static_cast<const T*>(s->points[0].x), n, ws, s->ws_flags, st);
  if (rc) return rc;
  const int64_t np = SUB ? s->batch : n;   // points of the pair list
  if (re > rb && pair_off(np, re) > pair_off(np, rb)) {
    LossArgs<T> la{static_cast<const T*>(s->scales[0].x), T(s->alpha), T(s->eps), s->terms, ws.loss, s->loss_params};
    const T* target = static_cast<const T*>(s->target);
    if constexpr (SUB) {
      if (s->loss_kind == MM_LOSS_STRESS)
        rc = spd_pdist_bwd_launch_sq<T, D, MM_LOSS_STRESS, true, 0, true>(ws, target, np, rb, re, s->wmin, s->wmax, st, la, s->batch_idx, n);
      else
        rc = spd_pdist_bwd_launch_sq<T, D, MM_LOSS_QUOTIENT, true, 0, true>(ws, target, np, rb, re, s->wmin, s->wmax, st, la, s->batch_idx, n);
    } else {
      if (s->loss_kind == MM_LOSS_STRESS)
        rc = spd_pdist_bwd_launch<T, D, MM_LOSS_STRESS>(ws, target, n, rb, re, 1, s->wmin, s->wmax, st, la);
      else
        rc = spd_pdist_bwd_launch<T, D, MM_LOSS_QUOTIENT>(ws, target, n, rb, re, 1, s->wmin, s->wmax, st, la);
    }
    if (rc) return rc;
  }
  const mm_step_param& q = s->scales[0];
  const bool fuse_scale = q.x && q.optimizer == MM_OPT_RSGD && q.momentum == 0.0;
  rc = spd_fused_step_launch<T, D, true>(s, ws, st, fuse_scale);
  if (rc == MM_OK) *scale_stepped = fuse_scale;
  return rc;
}

// dtype x D dispatch ---------------------------------------------------------
#if MM_SPD_MAX_D >= 9
#define MM_DISPATCH_D_HI(CALL)                    \
    case 6: { constexpr int D = 6; return CALL; } \
    case 7: { constexpr int D = 7; return CALL; } \
    case 8: { constexpr int D = 8; return CALL; } \
    case 9: { constexpr int D = 9; return CALL; }
#else
#define MM_DISPATCH_D_HI(CALL)
#endif
#define MM_DISPATCH_D(T, d, CALL)                \
  switch (d) {                                   \
    case 2: { constexpr int D = 2; return CALL; } \
    case 3: { constexpr int D = 3; return CALL; } \
    case 4: { constexpr int D = 4; return CALL; } \
    case 5: { constexpr int D = 5; return CALL; } \
    MM_DISPATCH_D_HI(CALL)                       \
    default: return MM_ERR_UNSUPPORTED;          \
  }

#define MM_DISPATCH(dtype, d, CALL_T)                                 \
  do {                                                                \
    if ((dtype) == MM_F32) { using T = float; MM_DISPATCH_D(T, d, CALL_T) } \
    if ((dtype) == MM_F64) { using T = double; MM_DISPATCH_D(T, d, CALL_T) } \
    return MM_ERR_ARG;                                                \
  } while (0)

}  // namespace mm
