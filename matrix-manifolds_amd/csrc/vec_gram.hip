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

#include <cstdlib>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include "stamp.hpp"
#include "smallmat.hpp"
#include "vecfn.hpp"
#include "loss.hpp"

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

// fp32: 32x32 tiles, kGramFwdTiles consecutive column tiles per wavefront (the row-block operand stays in
// registers; one tile per wavefront made the kernel wave-launch-bound: 8 k wavefronts of ~1 us each)
// row of a 32x32 accumulator tile held by register q of lane half h
__device__ __forceinline__ int mfma_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }
constexpr int kGramFwdTiles = 4;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // four floats at any 4-byte offset (rows of the pair vector)
// Operand loads are UNCONDITIONAL, from clamped addresses, and masked after they arrive: written as `valid ? x[..] : 0`
// every load is sunk under its lane condition — an exec-masked branch with `s_waitcnt vmcnt(0)` behind it — and the
// kernel walked through 6 serial memory round trips per tile (load, wait, MFMA, load, wait, MFMA ...).  The next tile's B
// operand is requested before the current tile's epilogue (acosh + 16 stores).  KS = MFMA k-steps (ceil(m / 2), rounded up
// to a dispatch class) is a template parameter so that the operand arrays stay in registers.
template <int KIND, int KS>
__global__ __launch_bounds__(64 * kGramWaves) void vec_gram_fwd_f32_kernel(const float* __restrict__ x, int n, int m,
                                                                         int row_begin, int row_end, int squared,
                                                                         float* __restrict__ out) {
  using u32 = unsigned int;
  __shared__ __attribute__((aligned(16))) float sO[kGramWaves][32][36];   // output tile per wavefront (rows of 144 B: 16-B aligned quads)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // grid: x = row tile, y = group of column tiles.  (The other way round, a launch whose x extent is 8 — n = 4039: 127
  // column tiles in groups of 16 — puts column group k of EVERY row tile on XCD k, workgroups being dealt round-robin over
  // the 8 XCDs in linear order: the triangle's long rows all start in group 0, so one XCD ran 127 live workgroups and the
  // last one 15.)
  const int i0 = row_begin + blockIdx.x * 32;
  const int jt0 = ((i0 + 1) / 32) + (blockIdx.y * kGramWaves + wave) * kGramFwdTiles;
  if (jt0 * 32 >= n) return;  // wave-uniform
  MM_FSTAMP(0);
  const int ia = i0 + r;
  const bool ia_ok = ia < n;
  const u32 xao = u32(ia_ok ? ia : n - 1) * u32(m);
  float av[KS];  // A operand: x[i0 + r][2 s + h] (Lorentz: space-like part negated)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + h;
    av[s] = x[xao + u32(k < m ? k : m - 1)];
  }
  auto request_b = [&](int t, float (&bv)[KS]) __attribute__((always_inline)) {   // B operand of tile t: x[j0 + r][2 s + h]
    const int jb = (jt0 + t) * 32 + r;
    const u32 xbo = u32(jb < n ? jb : n - 1) * u32(m);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + h;
      bv[s] = x[xbo + u32(k < m ? k : m - 1)];
    }
  };
  float b_next[KS];
  request_b(0, b_next);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    asm volatile("" : "+v"(av[s]));
    const int k = 2 * s + h;
    float a = (ia_ok && k < m) ? av[s] : 0.f;
    if (KIND == MM_LORENTZ && k != 0) a = -a;
    av[s] = a;
  }
  const int64_t base = gpair_off(n, row_begin);
  MM_FSTAMP(1);
  for (int t = 0; t < kGramFwdTiles; ++t) {
    const int j0 = (jt0 + t) * 32;
    if (j0 >= n) break;  // wave-uniform
    const bool jb_ok = j0 + r < n;
    float bv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bv[s] = b_next[s];
      asm volatile("" : "+v"(bv[s]));
    }
    MM_FSTAMP(2 + 3 * t);
    if (t + 1 < kGramFwdTiles) request_b(t + 1, b_next);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + h;
      const float b = (jb_ok && k < m) ? bv[s] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b, acc, 0, 0, 0);
    }
    const int j = j0 + r;
    MM_STAMP_PIN(acc);
    MM_FSTAMP(3 + 3 * t);
    if (j0 > i0 + 31 && j0 + 32 <= n && i0 + 32 <= row_end) {
      // Interior tile (all but the diagonal and the ragged edges): the kernel is bound by the ISSUE of its stores
      // (tools/gram_timeline.py fwd: the 16 dword stores of a tile took 2 - 4 k cycles to issue), so the tile is turned
      // through LDS — a lane then holds 4 consecutive columns of a row — and leaves as 4 `global_store_dwordx4`.
#pragma unroll
      for (int q = 0; q < 16; ++q) sO[wave][mfma_row(q, h)][r] = gram_value<float, KIND>(acc[q], squared);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int c4 = 4 * (lane & 7);
      int row = i0 + (lane >> 3);
      int64_t o = gpair_off(n, row) - base + (j0 + c4 - row - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(&sO[wave][(lane >> 3) + 8 * k][c4]);
        *reinterpret_cast<f32x4u*>(out + o) = v;
        o += 8 * int64_t(n - row - 2) - 28;   // eight rows down: sum of (n - row' - 2) over row' = row .. row + 7
        row += 8;
      }
      __builtin_amdgcn_wave_barrier();   // the tile is rewritten by the next pass
    } else {
      // offsets advance by additions: rows in register order are i0 + 4h + {0,1,2,3, 8,...}, and
      // off(row + 1) = off(row) + n - row - 2 (a 64-bit multiply per element cost more than the acosh)
      int row = i0 + 4 * h;
      int64_t o = gpair_off(n, row) - base + (j - row - 1);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = gram_value<float, KIND>(acc[q], squared);
        if (row < row_end && j < n && j > row) out[o] = v;
        const int step = (q & 3) == 3 ? 5 : 1;
#pragma unroll
        for (int d = 0; d < 5; ++d)
          if (d < step) { o += n - row - 2; ++row; }
      }
    }
    MM_FSTAMP(4 + 3 * t);
  }
  MM_STAMP_WAIT_VM();
  MM_FSTAMP(14);
  MM_FSTAMP_END();
}

// fp64: 16x16 tile per wavefront
template <int KIND>
__global__ __launch_bounds__(64 * kGramWaves) void vec_gram_fwd_f64_kernel(const double* __restrict__ x, int n, int m,
                                                                         int row_begin, int row_end, int squared,
                                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int i0 = row_begin + blockIdx.x * 16;   // grid: x = row tile, y = group of column tiles (see the fp32 kernel)
  const int j0 = (((i0 + 1) / 16) + blockIdx.y * kGramWaves + wave) * 16;
  if (j0 >= n) return;
  const int ia = i0 + r, jb = j0 + r;
  const bool ia_ok = ia < n, jb_ok = jb < n;
  const double* xa = x + size_t(ia_ok ? ia : 0) * m;
  const double* xb = x + size_t(jb_ok ? jb : 0) * m;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int ksteps = (m + 3) / 4;
  for (int s = 0; s < ksteps; ++s) {
    const int k = 4 * s + h;
    const int kc = k < m ? k : m - 1;
    double a = xa[kc], b = xb[kc];        // unconditional, pinned, masked after (see the fp32 kernel)
    asm volatile("" : "+v"(a), "+v"(b));
    a = (ia_ok && k < m) ? a : 0.0;
    b = (jb_ok && k < m) ? b : 0.0;
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


// ------------------------------------------------------------------ backward on the matrix cores
// grad_j = sum_i w_ij * (d q_ij / d x_j) with w_ij = g_ij * dout/dq(q_ij): with W the (symmetric, zero-
// diagonal) n x n matrix of the w_ij this is  ACC = W X  followed by the manifold's sign pattern
// (Lorentz: -J, lorentz.py:72-77,134-138; sphere: identity, sphere.py:68-74) — two chained GEMMs with an
// element-wise map between them, never materialising W:
//   1. Q tile (32 i x 32 j) = (X_I S) X_J^T            v_mfma_f32_32x32x2_f32, K = m
//   2. w = g * dout/dq(Q)                              on the accumulator registers (16 per lane)
//   3. ACC_J (32 j x m) += W_IJ^T X_I                  v_mfma_f32_32x32x2_f32, K = 32 — the A operand is
//      the accumulator of step 1 AS IT LIES: lane (j, h) holds W[i(s,h)][j] in register s, which is exactly
//      A^T[j][k] for the k-order (s, h); the B operand is loaded in the same row order.  No shuffle, no LDS.
// A wavefront owns one 32-column block J, keeps ACC_J in 16 accumulator registers and walks 32-row blocks
// I; every ORDERED tile is visited (the mirrored tile feeds the other block's wavefront), so there is no
// cross-lane reduction and a single flush per workgroup (4 wavefronts = 4 row sets, combined in LDS).
// The upstream gradient of a tile is read in 128-B row segments of the pair vector: directly for tiles
// above the diagonal (row i fixed per register), through a 32x33 LDS transpose for tiles below it (stored
// under row j: contiguous in i).  Against the VALU kernel: per pair 2 x (m + m) FMAs and the strided
// second read of g go away; what is left per pair is dout/dq (~15 VALU ops) and 4 B of HBM traffic.
constexpr int kGramBwdWaves = 4;
#ifndef MM_GRAM_BWD_MIN_WAVES
#define MM_GRAM_BWD_MIN_WAVES 2   // measured: 2-3 wavefronts per SIMD (156 VGPRs) 39 us, 4 (128 VGPRs) 47 us
#endif

// LOSS != 0 (mm_vec_pdist_loss): `g` holds the TARGET squared distances; the loss term, its derivative and
// the upstream gradient of each pair are formed on the accumulator registers (loss.hpp).  Entries outside
// the matrix / the shard carry a NaN sentinel instead of 0 through the layout change.
template <int KIND, int KS, int LOSS>  // KS = MFMA k-steps of the Gram = ceil(m / 2) rounded up to a dispatch class
__global__ __launch_bounds__(64 * kGramBwdWaves) __attribute__((amdgpu_waves_per_eu(MM_GRAM_BWD_MIN_WAVES)))
void vec_gram_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ g, int n, int m, int row_begin,
                             int row_end, int squared, int tiles_per_wave, float* __restrict__ grad,
                             LossArgs<float> la) {
  __shared__ float sT[kGramBwdWaves][32][33];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nT = (n + 31) / 32;
  const int jb = blockIdx.x, J = jb * 32;
  // All loads are issued unconditionally from clamped addresses and masked afterwards: predicated loads
  // compile to one exec-masked branch each and serialise the memory latency.  Element offsets are 32-bit
  // (the launcher bounds n), so every load is base-SGPR + 32-bit VGPR offset.
  using u32 = unsigned int;
  const int rc = r < m ? r : m - 1;
  // Euclidean (squared distances only): dout/dq = 1, so W = g and no Gram is needed at all — the kernel is a
  // symmetric packed matrix times X; a ones column appended to X delivers the row sums of W that the
  // gradient 2 (x_j sum_i w_ij - sum_i w_ij x_i) needs.
  constexpr bool kEuclid = KIND == MM_EUCLIDEAN;
  float bJ[KS];  // B operand of the Gram: x[J + r][2 s + h]
  if constexpr (!kEuclid) {
    const int jr = J + r;
    const u32 xb = u32(jr < n ? jr : n - 1) * u32(m);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + h;
      bJ[s] = x[xb + u32(k < m ? k : m - 1)];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {   // (pinned, then masked: `valid ? x[..] : 0` serialises the loads — see the forward kernel)
      asm volatile("" : "+v"(bJ[s]));
      bJ[s] = (jr < n && 2 * s + h < m) ? bJ[s] : 0.f;
    }
  }
  f32x16 accJ;
#pragma unroll
  for (int q = 0; q < 16; ++q) accJ[q] = 0.f;
  float sp = 1.f, loss_acc = 0.f, ds_acc = 0.f;
  loss_resolve<float, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  const float kInvalid = LOSS != MM_LOSS_NONE ? __builtin_nanf("") : 0.f;
  const u32 base = u32(gpair_off(n, row_begin));
  const u32 gmax = u32(gpair_off(n, row_end)) - base - 1u;  // last valid index of this shard's slice
  auto goff = [&](int lo, int hi) -> u32 {  // pair (lo < hi) -> offset in this shard's slice, clamped
    const u32 o = u32(lo) * u32(2 * n - lo - 1) / 2u - base + u32(hi - lo - 1);
    return o > gmax ? gmax : o;   // (also catches the wrap-around of rows below row_begin)
  };
  const int j = J + r;
  // Loads of a tile (operands of both GEMMs and the upstream gradients, in LOAD layout: for a tile below
  // the diagonal the gradients arrive transposed), all issued before anything waits on them.
  struct TileLoads { float xa[KS], bI[16], gr[16]; };
  auto tile_block = [&](int t) { return (blockIdx.y * tiles_per_wave + t) * kGramBwdWaves + wave; };
  // Offsets advance by additions: the rows of a tile in register order are row0 + {0,1,2,3, 8,...} (accumulator
  // layout) or row0 + {0,2,4,...} (transposed load), and  off(row + 1) = off(row) + n - row - 2.
  auto issue_xa = [&](int ib, TileLoads& L) {  // A operand of the Gram: rows of X_I, lane = row
    const int ia = ib * 32 + r;
    const u32 xo = u32(ia < n ? ia : n - 1) * u32(m);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + h;
      L.xa[s] = x[xo + u32(k < m ? k : m - 1)];
    }
  };
  auto issue_bI = [&](int ib, TileLoads& L) {  // B operand of W^T X_I: rows I + mfma_row(s, h), lane = feature
    const u32 xlast = u32(n - 1) * u32(m) + u32(rc);
    u32 xo = u32(ib * 32 + 4 * h) * u32(m) + u32(rc);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      L.bI[s] = x[xo < xlast ? xo : xlast];
      xo += u32((s & 3) == 3 ? 5 : 1) * u32(m);
    }
  };
  auto issue_g = [&](int ib, TileLoads& L) {
    const int I = ib * 32;
    if (ib != jb) {
      // above the diagonal (ib < jb): pair (i, j) lies in row i of the pair vector, contiguous in j — loaded
      // straight into accumulator layout.  Below (ib > jb): pair (j', i) lies in row j', contiguous in i —
      // loaded as the transposed tile (lane = i, register = j') and turned through LDS at use.
      const bool upper = ib < jb;
      const int A0 = upper ? I : J, B0 = upper ? J : I;  // row block / column block of the stored pairs
      const int col = B0 + r < n ? B0 + r : n - 1;
      int row = A0 + (upper ? 4 * h : h);
      u32 o = u32(row) * u32(2 * n - row - 1) / 2u - base + u32(col - row - 1);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        L.gr[s] = g[o > gmax ? gmax : o];  // (clamp: ragged edges and rows outside the shard, masked at use)
        const int step = upper ? ((s & 3) == 3 ? 5 : 1) : 2;
#pragma unroll
        for (int d = 0; d < 5; ++d)
          if (d < step) { o += u32(n - row - 2); ++row; }
      }
    } else {  // diagonal block: both orientations, element-wise gather (1 tile in nT)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int i = I + mfma_row(s, h);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        L.gr[s] = g[goff(lo, hi < n ? (hi > lo ? hi : lo + 1) : n - 1)];
      }
    }
  };
  // (Requesting the next tile's operands while the current tile computes was measured SLOWER in both
  // forms tried — a second register set: 240 VGPRs, 50 us; reusing each set right after its last use:
  // 195 VGPRs, 42 us — against 37 us for this plain order at 162 VGPRs / 3 wavefronts per SIMD.)
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int ib = tile_block(t);
    if (ib >= nT) break;  // wave-uniform
    const int I = ib * 32;
    TileLoads cur;
    issue_g(ib, cur);
    if constexpr (!kEuclid) issue_xa(ib, cur);
    issue_bI(ib, cur);
    // upstream gradients in accumulator layout: element (i = I + row(s,h), j = J + r).  Tiles that lie
    // entirely inside the matrix and the shard (all but the ragged edges) need no masking at all.
    const bool upper = ib < jb;
    const int A0 = upper ? I : J;
    const bool interior = ib != jb && I + 32 <= n && J + 32 <= n && A0 >= row_begin && A0 + 32 <= row_end;
    float gv[16];
    if (__builtin_expect(interior, 1)) {
#pragma unroll
      for (int s = 0; s < 16; ++s) gv[s] = cur.gr[s];
    } else if (ib != jb) {
      const int col = (upper ? J : I) + r;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = A0 + (upper ? mfma_row(s, h) : 2 * s + h);
        const bool valid = col < n && row >= row_begin && row < row_end;  // row < col always
        gv[s] = valid ? cur.gr[s] : kInvalid;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int i = I + mfma_row(s, h);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const bool valid = i != j && hi < n && lo >= row_begin && lo < row_end;
        gv[s] = valid ? cur.gr[s] : kInvalid;
      }
    }
    if (ib > jb) {  // transposed load -> accumulator layout
#pragma unroll
      for (int u = 0; u < 16; ++u) sT[wave][2 * u + h][r] = gv[u];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int s = 0; s < 16; ++s) gv[s] = sT[wave][r][mfma_row(s, h)];
      __builtin_amdgcn_wave_barrier();
    }
    // 1. Gram tile
    f32x16 q;
#pragma unroll
    for (int k = 0; k < 16; ++k) q[k] = 0.f;
    if constexpr (!kEuclid) {
      const bool ia_ok = I + r < n;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + h;
        float a = (ia_ok && k < m) ? cur.xa[s] : 0.f;
        if (KIND == MM_LORENTZ && k != 0) a = -a;
        q = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bJ[s], q, 0, 0, 0);
      }
    }
    // 2.-3. w = g * dout/dq(Q) on the accumulator registers; ACC_J += W^T X_I
    const bool rows_in = I + 32 <= n;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float w;
      if constexpr (kEuclid) {
        w = gv[s];
      } else if constexpr (LOSS == MM_LOSS_NONE) {
        w = gv[s] * PairFn<float, KIND>::dq(q[s], squared);
      } else {
        const bool ok = gv[s] == gv[s];
        const float d2 = PairFn<float, KIND>::value(q[s], 1);
        float dldm;
        const float l = loss_term<float, LOSS>(sp * d2, gv[s], la, dldm);
        const bool once = ok && (ib < jb || (ib == jb && I + mfma_row(s, h) < j));  // each unordered pair once
        loss_acc += once ? l : 0.f;
        ds_acc += once ? dldm * d2 : 0.f;
        w = ok ? dldm * sp * PairFn<float, KIND>::dq(q[s], 1) : 0.f;
      }
      float b = ((rows_in || I + mfma_row(s, h) < n) && r < m) ? cur.bI[s] : 0.f;
      if (kEuclid && r == m) b = 1.f;  // ones column: accumulates sum_i w_ij (w is 0 for rows outside the matrix)
      accJ = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, accJ, 0, 0, 0);
    }
  }
  // combine the workgroup's wavefronts (same J, different rows) and flush once
#pragma unroll
  for (int q = 0; q < 16; ++q) sT[wave][mfma_row(q, h)][r] = accJ[q];
  __shared__ float lossW[kGramBwdWaves][2];
  if constexpr (LOSS != MM_LOSS_NONE) {
    const float l = wave_sum(loss_acc), d = wave_sum(ds_acc);
    if (lane == 0) { lossW[wave][0] = l; lossW[wave][1] = d; }
  }
  __syncthreads();
  if constexpr (LOSS != MM_LOSS_NONE) {
    if (threadIdx.x == 0) {
      float l = 0.f, d = 0.f;
#pragma unroll
      for (int w = 0; w < kGramBwdWaves; ++w) { l += lossW[w][0]; d += lossW[w][1]; }
      const int slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kLossSlots - 1);
      atomic_add(&la.slots[slot], l);
      atomic_add(&la.slots[kLossSlots + slot], d);
    }
  }
  for (int e = threadIdx.x; e < 32 * 32; e += 64 * kGramBwdWaves) {
    const int jj = e >> 5, c = e & 31;
    if (c < m && J + jj < n) {
      float sum = sT[0][jj][c];
#pragma unroll
      for (int w = 1; w < kGramBwdWaves; ++w) sum += sT[w][jj][c];
      if (KIND == MM_LORENTZ && c != 0) sum = -sum;  // d q / d x_j = -J x_i
      if constexpr (kEuclid) {                        // d q / d x_j = 2 (x_j - x_i)
        float wsum = sT[0][jj][m];
#pragma unroll
        for (int w = 1; w < kGramBwdWaves; ++w) wsum += sT[w][jj][m];
        sum = 2.f * fmaf(wsum, x[size_t(J + jj) * m + c], -sum);
      }
      atomic_add(&grad[size_t(J + jj) * m + c], sum);
    }
  }
}


// ------------------------------------------------------------------ backward, symmetric tiles (fp32)
// Every UNORDERED 32 x 32 tile of the pair matrix is visited once: its W = g * dout/dq(Q) feeds BOTH gradient
// blocks, ACC_J += W^T X_I (the accumulator is the A operand as it lies, as above) and ACC_I += W X_J (W turned
// through a 32 x 33 LDS tile into the A-operand layout).  Against the ordered-tile kernel above this halves the reads
// of g (its mirrored tile re-read every upstream gradient, transposed through LDS), the Gram products and — the
// vector-ALU bulk of the kernel — the evaluations of dout/dq (acosh / acos derivatives: sqrt, log, rcp per pair).
// One of the two output blocks of a tile can stay in accumulator registers along a walk, the other cannot, so a
// workgroup owns a SUPER-TILE of 4 x 4 tiles (128 rows x 128 columns): wavefront w keeps ACC_J of column block w in
// registers, the four row blocks' ACC_I live in LDS (16 KB), and at step t wavefront w works on tile
// (row block (w + t) mod 4, column block w) — the four wavefronts always add into four different row blocks, so
// the LDS accumulation needs no atomics, only a barrier between steps.  Both accumulators leave once per
// super-tile as contiguous 32 x m float atomics.  Super-tiles on the diagonal run their 10 upper tiles.
// Variants measured and dropped (n = 4039, m = 11, this kernel 36.8 us): requesting the next step's operands before the
// current step computes (176 VGPRs, 2 wavefronts per SIMD: 43.5 us); 2 x 4 super-tiles with a private LDS accumulator
// set per wavefront and no barrier (twice the workgroups, half the tiles each: 53 us); LDS float atomics on one shared
// accumulator set (100 us).
template <int KIND, int KS, int LOSS>
__global__ __launch_bounds__(64 * kGramBwdWaves) __attribute__((amdgpu_waves_per_eu(MM_GRAM_BWD_MIN_WAVES)))
void vec_gram_bwd_sym_f32_kernel(const float* __restrict__ x, const float* __restrict__ g, int n, int m, int row_begin,
                                 int row_end, int squared, int parts, float* __restrict__ grad, LossArgs<float> la) {
  __shared__ float sT[kGramBwdWaves][32][33];     // per-wavefront transpose tile
  __shared__ float accI[kGramBwdWaves][32][32];   // row-side accumulators of the super-tile's four row blocks
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  using u32 = unsigned int;
  MM_GSTAMP(0);
  // super-tile (A, B), A <= B, from the linear workgroup index.  OFF-DIAGONAL super-tiles first (id = B (B - 1) / 2 + A,
  // A < B: 16 tiles each), the diagonal ones (10 tiles) last: workgroups are placed in index order, breadth first over
  // the CUs, and a launch of slightly more than two workgroups per CU (n = 4039: 528 on 256 CUs) leaves a third one on a
  // few CUs, which then decide the kernel's duration (measured: wavefront life 80 k cycles there against 56 k on the
  // others, tools/gram_timeline.py) — so the last workgroups placed are the lightest.
  const int nT = (n + 31) / 32;
  const int nS = (nT + 3) / 4;
  // `parts` (1, 2 or 4) workgroups share a super-tile: each takes 4 / parts of its four steps and flushes its own sums
  const int id = blockIdx.x / parts, part = blockIdx.x - id * parts, nOff = nS * (nS - 1) / 2;
  const int steps = 4 / parts, t_begin = part * steps, t_end = t_begin + steps;
  int As, Bs;
  if (id < nOff) {
    Bs = int((__builtin_sqrtf(8.f * float(id) + 1.f) + 1.f) * 0.5f);
    while (Bs * (Bs - 1) / 2 > id) --Bs;
    while ((Bs + 1) * Bs / 2 <= id) ++Bs;
    As = id - Bs * (Bs - 1) / 2;
  } else {
    As = Bs = id - nOff;
  }
  if (As * 128 >= row_end || As * 128 + 127 < row_begin) return;   // no pair of this super-tile has its row in the shard
  constexpr bool kEuclid = KIND == MM_EUCLIDEAN;
  const int rc = r < m ? r : m - 1;
  const int jb = Bs * 4 + wave, J = jb * 32;   // this wavefront's column block
  const bool j_live = jb < nT;                  // (ragged last super-tile)
  const u32 base = u32(gpair_off(n, row_begin));
  const u32 gmax = u32(gpair_off(n, row_end)) - base - 1u;  // last valid index of this shard's slice
  // Upstream gradients (accumulator layout) and the Gram A operand (rows of X_I) of step t.  Issued ONE STEP AHEAD into a
  // second register set (the first request before the prologue's barrier): a wavefront spent a quarter of its life
  // waiting for these loads at the top of every step (tools/gram_timeline.py).  Unconditional, from clamped offsets: for a
  // step this wavefront skips they read some valid element that nobody uses.
  auto request = [&](int t, float (&gr)[16], float (&xa)[KS]) __attribute__((always_inline)) {
    const int ib = As * 4 + ((wave + t) & 3), I = ib * 32;
    if (ib != jb) {   // pair (i, j) lies in row i of the pair vector, contiguous in j
      const int col = J + r < n ? J + r : n - 1;
      int row = I + 4 * h;
      u32 o = u32(row) * u32(2 * n - row - 1) / 2u - base + u32(col - row - 1);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        gr[s] = g[o > gmax ? gmax : o];   // (clamp: ragged edges and rows outside the shard, masked at use)
        const int step = (s & 3) == 3 ? 5 : 1;
#pragma unroll
        for (int d = 0; d < 5; ++d)
          if (d < step) { o += u32(n - row - 2); ++row; }
      }
    } else {
      const int j = J + r;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int i = I + mfma_row(s, h);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int hc = hi < n ? (hi > lo ? hi : lo + 1) : n - 1;
        const u32 o = u32(lo) * u32(2 * n - lo - 1) / 2u - base + u32(hc - lo - 1);
        gr[s] = g[o > gmax ? gmax : o];
      }
    }
    if constexpr (!kEuclid) {
      const int ia = I + r;
      const u32 xo = u32(ia < n ? ia : n - 1) * u32(m);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + h;
        xa[s] = x[xo + u32(k < m ? k : m - 1)];
      }
    }
  };
  float bJ[KS];      // B operand of the Gram: x[J + r][2 s + h]
  float xJ[16];      // B operand of W X_J: rows J + 2 s + h, lane = feature
  // (the fused-loss instantiations need 185 - 233 registers with the second set: two wavefronts per SIMD, i.e. fewer
  // workgroup slots than a 4039-node launch has workgroups — they request at the top of the step)
  constexpr bool kAhead = LOSS == MM_LOSS_NONE;
  float gr_next[16], xa_next[KS];
  {
    // All loads first, from clamped addresses, PINNED below the last of them, masks afterwards: written as
    // `valid ? x[..] : 0` each load is sunk under its condition — an exec-masked branch with a `vmcnt(0)` behind every
    // one of the 22 loads (measured: a 7 k-cycle prologue, tools/gram_timeline.py).
    const int jr = J + r;
    const u32 xb = u32(jr < n ? jr : n - 1) * u32(m);
    if constexpr (!kEuclid) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + h;
        bJ[s] = x[xb + u32(k < m ? k : m - 1)];
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int jj = J + 2 * s + h;
      xJ[s] = x[u32(jj < n ? jj : n - 1) * u32(m) + u32(rc)];
    }
    if constexpr (kAhead) request(t_begin, gr_next, xa_next);   // the first step's operands, in flight together with the column block's
    if constexpr (!kEuclid) {
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bJ[s]));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(xJ[s]));
    if constexpr (!kEuclid) {
#pragma unroll
      for (int s = 0; s < KS; ++s) bJ[s] = (jr < n && 2 * s + h < m) ? bJ[s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int jj = J + 2 * s + h;
      xJ[s] = (jj < n && r < m) ? xJ[s] : 0.f;
      if (kEuclid && r == m) xJ[s] = jj < n ? 1.f : 0.f;   // ones column: row sums of W
    }
  }
  f32x16 accJ;
#pragma unroll
  for (int q = 0; q < 16; ++q) accJ[q] = 0.f;
  for (int e = threadIdx.x; e < kGramBwdWaves * 32 * 32; e += 64 * kGramBwdWaves) (&accI[0][0][0])[e] = 0.f;
  float sp = 1.f, loss_acc = 0.f, ds_acc = 0.f;
  loss_resolve<float, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  const float kInvalid = LOSS != MM_LOSS_NONE ? __builtin_nanf("") : 0.f;
  __syncthreads();
  MM_GSTAMP(1);
  for (int t = t_begin; t < t_end; ++t) {
    const int a = (wave + t) & 3;          // row block of this step (distinct per wavefront)
    const int ib = As * 4 + a, I = ib * 32;
    const bool live = j_live && ib < nT && ib <= jb;   // wave-uniform; below-diagonal tiles belong to their mirror
    float gr[16], xa[KS];
    if constexpr (kAhead) {
#pragma unroll
      for (int s = 0; s < 16; ++s) gr[s] = gr_next[s];
#pragma unroll
      for (int s = 0; s < KS; ++s) xa[s] = xa_next[s];
      if (t + 1 < t_end) request(t + 1, gr_next, xa_next);
    }
    if (live) {
      if constexpr (!kAhead) request(t, gr, xa);
      const bool diag = ib == jb;
      float bI[16];   // B operand of W^T X_I
      {
        const u32 xlast = u32(n - 1) * u32(m) + u32(rc);
        u32 xo = u32(I + 4 * h) * u32(m) + u32(rc);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          bI[s] = x[xo < xlast ? xo : xlast];
          xo += u32((s & 3) == 3 ? 5 : 1) * u32(m);
        }
      }
      MM_STAMP_WAIT_VM();
      MM_GSTAMP(2 + 5 * t);
      // ---- validity: each unordered pair once (i < j), inside the matrix, row inside the shard
      const bool interior = !diag && I + 32 <= n && J + 32 <= n && I >= row_begin && I + 32 <= row_end;
      float gv[16];
      if (__builtin_expect(interior, 1)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) gv[s] = gr[s];
      } else {
        const int j = J + r;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int i = I + mfma_row(s, h);
          const bool valid = i < j && j < n && i >= row_begin && i < row_end;
          gv[s] = valid ? gr[s] : kInvalid;
        }
      }
      // ---- 1. Gram tile
      f32x16 q;
#pragma unroll
      for (int k = 0; k < 16; ++k) q[k] = 0.f;
      if constexpr (!kEuclid) {
        const bool ia_ok = I + r < n;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int k = 2 * s + h;
          float av = (ia_ok && k < m) ? xa[s] : 0.f;
          if (KIND == MM_LORENTZ && k != 0) av = -av;
          q = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bJ[s], q, 0, 0, 0);
        }
      }
      // ---- 2. w = g * dout/dq(Q) on the accumulator registers
      float w[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (kEuclid) {
          w[s] = gv[s];
        } else if constexpr (LOSS == MM_LOSS_NONE) {
          w[s] = gv[s] * PairFn<float, KIND>::dq(q[s], squared);
        } else {
          const bool ok = gv[s] == gv[s];
          const float d2 = PairFn<float, KIND>::value(q[s], 1);
          float dldm;
          const float l = loss_term<float, LOSS>(sp * d2, gv[s], la, dldm);
          loss_acc += ok ? l : 0.f;
          ds_acc += ok ? dldm * d2 : 0.f;
          w[s] = ok ? dldm * sp * PairFn<float, KIND>::dq(q[s], 1) : 0.f;
        }
      }
      MM_STAMP_PIN(w[15]);
      MM_GSTAMP(3 + 5 * t);
      // ---- 3. ACC_J += W^T X_I (the accumulator as the A operand) ...
      // The B operand needs no masking: rows outside the matrix were loaded from the last row (finite) and meet w = 0;
      // lanes r >= m hold a copy of column m - 1 and only feed output columns >= m, which are never flushed.  (The masks
      // used to cost a dozen scalar branch instructions per element of the tile.)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float b = bI[s];
        if (kEuclid && r == m) b = 1.f;  // ones column (w is 0 for rows outside the matrix)
        accJ = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], b, accJ, 0, 0, 0);
      }
      // ---- 4. ... and P = W X_J: W[i][j] from accumulator layout (lane = j) to A-operand layout (lane = i)
#pragma unroll
      for (int s = 0; s < 16; ++s) sT[wave][mfma_row(s, h)][r] = w[s];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MM_GSTAMP(4 + 5 * t);
      f32x16 pacc;
#pragma unroll
      for (int k = 0; k < 16; ++k) pacc[k] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[wave][r][2 * s + h], xJ[s], pacc, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
      // row block a of the super-tile belongs to this wavefront during this step
#pragma unroll
      for (int k = 0; k < 16; ++k) accI[a][mfma_row(k, h)][r] += pacc[k];
      MM_STAMP_WAIT_LGKM();
      MM_GSTAMP(5 + 5 * t);
    }
    __syncthreads();
    MM_GSTAMP(6 + 5 * t);
  }
  // ---- flush: wavefront w writes its column block's ACC_J and row block w's ACC_I, 32 x m contiguous floats each
  if constexpr (LOSS != MM_LOSS_NONE) {
    __shared__ float lossW[kGramBwdWaves][2];
    const float l = wave_sum(loss_acc), d = wave_sum(ds_acc);
    if (lane == 0) { lossW[wave][0] = l; lossW[wave][1] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ls = 0.f, dd = 0.f;
#pragma unroll
      for (int wv = 0; wv < kGramBwdWaves; ++wv) { ls += lossW[wv][0]; dd += lossW[wv][1]; }
      const int slot = blockIdx.x & (kLossSlots - 1);
      atomic_add(&la.slots[slot], ls);
      atomic_add(&la.slots[kLossSlots + slot], dd);
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) sT[wave][mfma_row(k, h)][r] = accJ[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  auto flush = [&](const float (&acc)[32][33], int R0) {   // acc[row][c] -> grad[(R0 + row) * m + c]
    if (R0 >= n) return;
    const int cnt = min(32, n - R0) * m;
    for (int e = lane; e < cnt; e += 64) {
      const int row = e / m, c = e - row * m;
      float sum = acc[row][c];
      if (KIND == MM_LORENTZ && c != 0) sum = -sum;                            // d q / d x = -J x'
      if constexpr (kEuclid) sum = 2.f * fmaf(acc[row][m], x[size_t(R0) * m + e], -sum);   // 2 (x sum w - sum w x')
      atomic_add(&grad[size_t(R0) * m + e], sum);
    }
  };
  if (j_live) flush(sT[wave], J);
  __builtin_amdgcn_wave_barrier();
  {   // row block `wave` of the super-tile (accI is 32 wide: through the transpose tile for one flush routine)
    const int I = (As * 4 + wave) * 32;
#pragma unroll
    for (int k = 0; k < 16; ++k) sT[wave][mfma_row(k, h)][r] = accI[wave][mfma_row(k, h)][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    flush(sT[wave], I);
  }
  MM_GSTAMP(22);
  MM_GSTAMP_END();
}

#include "vec_gram_bwd64.hpp"

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
  const int per_wg = kGramWaves * (dtype == MM_F32 ? kGramFwdTiles : 1);
  const dim3 grid(int((row_end - row_begin + tile - 1) / tile), (ntile_cols + per_wg - 1) / per_wg);   // x = row tiles, y = column groups
  const dim3 block(64 * kGramWaves);
  {
    ProfScope prof(PROF_VEC_FWD, st);
    if (dtype == MM_F32) {
      auto* xp = static_cast<const float*>(x);
      auto* op = static_cast<float*>(out);
      if (m > 32) return MM_ERR_UNSUPPORTED;   // 16 k-steps of 2
      const int ks = (m + 1) / 2;
#define MM_GRAM_FWD_LAUNCH(KS_)                                                                                          \
  do {                                                                                                                   \
    if (kind == MM_LORENTZ)                                                                                              \
      vec_gram_fwd_f32_kernel<MM_LORENTZ, KS_><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op); \
    else                                                                                                                 \
      vec_gram_fwd_f32_kernel<MM_SPHERE, KS_><<<grid, block, 0, st>>>(xp, int(n), m, int(row_begin), int(row_end), squared, op);  \
  } while (0)
      if (ks <= 2) MM_GRAM_FWD_LAUNCH(2);
      else if (ks <= 4) MM_GRAM_FWD_LAUNCH(4);
      else if (ks <= 6) MM_GRAM_FWD_LAUNCH(6);
      else if (ks <= 8) MM_GRAM_FWD_LAUNCH(8);
      else if (ks <= 12) MM_GRAM_FWD_LAUNCH(12);
      else MM_GRAM_FWD_LAUNCH(16);
#undef MM_GRAM_FWD_LAUNCH
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

namespace mm {

template <typename T>
__global__ void gram_loss_finalize_kernel(T* __restrict__ slots, const T* __restrict__ scale_raw, T* __restrict__ loss_out) {
  loss_finalize<T>(slots, scale_raw, loss_out);
}

// Shared launchers of the matrix-core backward: plain (loss_kind = MM_LOSS_NONE, `g` = upstream gradients)
// or fused objective (`g` = targets; la.slots must be zeroed [2][kLossSlots]).
int vec_gram_bwd_launch(int kind, int loss_kind, const float* xp, const float* gp, int64_t n, int m, int64_t row_begin,
                        int64_t row_end, int squared, float* op, LossArgs<float> la, hipStream_t st) {
  hipError_t e = hipMemsetAsync(op, 0, sizeof(float) * size_t(n) * m, st);
  if (e != hipSuccess) return int(e);
  if (mm_pair_offset(n, row_end) == mm_pair_offset(n, row_begin)) return MM_OK;
  const int nT = int((n + 31) / 32);
  // 16 row blocks per workgroup (4 per wavefront) unless that leaves the chip under-filled
  static const int tpw_env = [] { const char* e = std::getenv("MM_GRAM_BWD_TPW"); return e ? std::atoi(e) : 0; }();
  int tpw = tpw_env > 0 ? tpw_env : 4;
  while (tpw_env <= 0 && tpw > 1 && int64_t(nT) * ((nT + kGramBwdWaves * tpw - 1) / (kGramBwdWaves * tpw)) < 768) tpw >>= 1;
  const dim3 grid(nT, (nT + kGramBwdWaves * tpw - 1) / (kGramBwdWaves * tpw));
  const dim3 block(64 * kGramBwdWaves);
  // Symmetric tiles for the inner-product manifolds (half the reads of g, the Gram products and the dout/dq evaluations;
  // measured at n = 4039, m = 11: 36.8 us and 45 MB fetched against 35.7 us and 96 MB for the ordered kernel — both are
  // bound by the per-wavefront chain load -> MFMA -> dout/dq -> MFMA at <= 4 wavefronts per SIMD, not by bandwidth; the
  // halved traffic pays where g does not sit in the Infinity Cache).  The squared Euclidean distance has neither Gram nor
  // dout/dq to halve and keeps the ordered kernel (24 us against 30 us).  MM_GRAM_BWD_ORDERED = 1 / 0 forces one.
  static const int forced = [] { const char* e = std::getenv("MM_GRAM_BWD_ORDERED"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  const bool ordered = forced >= 0 ? forced == 1 : kind == MM_EUCLIDEAN;
  const int nS = (nT + 3) / 4;                       // super-tiles of 4 x 4 tiles per side
  // Workgroups per super-tile.  All workgroups of these launches are resident at once, so the CU with the most of them
  // decides the duration: 528 super-tiles (n = 4039) are two per CU on 240 CUs and three on 16 (tools/gram_timeline.py).
  // When the fullest CU carries more than 1.2 x the average, every super-tile is split between two workgroups (two of
  // its four steps each; both flush their own sums): n = 4039 Lorentz 33.9 -> 32.7 us, n = 5000 sphere 46.8 -> 43.6 us on
  // one box; four parts: 39.5 / 48.4 us (prologue and flush per workgroup outweigh the balance).  MM_GRAM_BWD_PARTS forces.
  static const int parts_env = [] { const char* e = std::getenv("MM_GRAM_BWD_PARTS"); return e ? std::atoi(e) : 0; }();
  static const int cus = [] {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
    return c;
  }();
  const int64_t nsuper = int64_t(nS) * (nS + 1) / 2;
  int parts = ((nsuper + cus - 1) / cus) * cus * 5 > nsuper * 6 ? 2 : 1;
  if (parts_env == 1 || parts_env == 2 || parts_env == 4) parts = parts_env;
  const dim3 sgrid(unsigned(nS * (nS + 1) / 2) * unsigned(parts));     // unordered pairs of super-tile indices
  {
    ProfScope prof(PROF_VEC_BWD, st);
#define MM_GRAM_BWD(KIND_, KS_, LOSS_)                                                                        \
  do {                                                                                                        \
    if (ordered)                                                                                              \
      vec_gram_bwd_f32_kernel<KIND_, KS_, LOSS_><<<grid, block, 0, st>>>(xp, gp, int(n), m, int(row_begin), \
                                                                        int(row_end), squared, tpw, op, la);  \
    else                                                                                                      \
      vec_gram_bwd_sym_f32_kernel<KIND_, KS_, LOSS_><<<sgrid, block, 0, st>>>(xp, gp, int(n), m,            \
                                                                        int(row_begin), int(row_end), squared, parts, op, la); \
  } while (0)
#define MM_GRAM_BWD_KS(KIND_, LOSS_)                        \
  do {                                                      \
    const int ks = (m + 1) / 2;                             \
    if (ks <= 2) MM_GRAM_BWD(KIND_, 2, LOSS_);              \
    else if (ks <= 4) MM_GRAM_BWD(KIND_, 4, LOSS_);         \
    else if (ks <= 6) MM_GRAM_BWD(KIND_, 6, LOSS_);         \
    else if (ks <= 8) MM_GRAM_BWD(KIND_, 8, LOSS_);         \
    else if (ks <= 12) MM_GRAM_BWD(KIND_, 12, LOSS_);       \
    else MM_GRAM_BWD(KIND_, 16, LOSS_);                     \
  } while (0)
#define MM_GRAM_BWD_LOSS(KIND_)                                                  \
  do {                                                                           \
    if (loss_kind == MM_LOSS_NONE) MM_GRAM_BWD_KS(KIND_, MM_LOSS_NONE);          \
    else if (loss_kind == MM_LOSS_STRESS) MM_GRAM_BWD_KS(KIND_, MM_LOSS_STRESS); \
    else MM_GRAM_BWD_KS(KIND_, MM_LOSS_QUOTIENT);                                \
  } while (0)
    if (kind == MM_LORENTZ) MM_GRAM_BWD_LOSS(MM_LORENTZ);
    else if (kind == MM_SPHERE) MM_GRAM_BWD_LOSS(MM_SPHERE);
    else MM_GRAM_BWD(MM_EUCLIDEAN, 2, MM_LOSS_NONE);
#undef MM_GRAM_BWD_LOSS
#undef MM_GRAM_BWD_KS
#undef MM_GRAM_BWD
  }
  e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

int vec_gram_bwd_launch(int kind, int loss_kind, const double* xp, const double* gp, int64_t n, int m,
                        int64_t row_begin, int64_t row_end, int squared, double* op, LossArgs<double> la,
                        hipStream_t st) {
  hipError_t e = hipMemsetAsync(op, 0, sizeof(double) * size_t(n) * m, st);
  if (e != hipSuccess) return int(e);
  if (mm_pair_offset(n, row_end) == mm_pair_offset(n, row_begin)) return MM_OK;
  const int nT = int((n + 15) / 16);
  int tpw = 8;  // 32 row blocks of 16 per workgroup
  while (tpw > 1 && int64_t(nT) * ((nT + kGramBwdWaves * tpw - 1) / (kGramBwdWaves * tpw)) < 768) tpw >>= 1;
  const dim3 grid(nT, (nT + kGramBwdWaves * tpw - 1) / (kGramBwdWaves * tpw));
  const dim3 block(64 * kGramBwdWaves);
  {
    ProfScope prof(PROF_VEC_BWD, st);
#define MM_GRAM_BWD64(KIND_, KS_, LOSS_)                                                                      \
  vec_gram_bwd_f64_kernel<KIND_, KS_, LOSS_><<<grid, block, 0, st>>>(xp, gp, int(n), m, int(row_begin),     \
                                                                    int(row_end), squared, tpw, op, la)
#define MM_GRAM_BWD64_KS(KIND_, LOSS_)                      \
  do {                                                      \
    const int ks = (m + 3) / 4;                             \
    if (ks <= 1) MM_GRAM_BWD64(KIND_, 1, LOSS_);            \
    else if (ks <= 2) MM_GRAM_BWD64(KIND_, 2, LOSS_);       \
    else if (ks <= 3) MM_GRAM_BWD64(KIND_, 3, LOSS_);       \
    else MM_GRAM_BWD64(KIND_, 4, LOSS_);                    \
  } while (0)
#define MM_GRAM_BWD64_LOSS(KIND_)                                                  \
  do {                                                                             \
    if (loss_kind == MM_LOSS_NONE) MM_GRAM_BWD64_KS(KIND_, MM_LOSS_NONE);          \
    else if (loss_kind == MM_LOSS_STRESS) MM_GRAM_BWD64_KS(KIND_, MM_LOSS_STRESS); \
    else MM_GRAM_BWD64_KS(KIND_, MM_LOSS_QUOTIENT);                                \
  } while (0)
    if (kind == MM_LORENTZ) MM_GRAM_BWD64_LOSS(MM_LORENTZ); else MM_GRAM_BWD64_LOSS(MM_SPHERE);
#undef MM_GRAM_BWD64_LOSS
#undef MM_GRAM_BWD64_KS
#undef MM_GRAM_BWD64
  }
  e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

// which configurations the matrix-core kernels take (32-bit element offsets: n <= 32768)
bool vec_gram_supports(int dtype, int kind, int64_t n, int m) {
  if (kind != MM_LORENTZ && kind != MM_SPHERE) return false;
  if (n > 32768) return false;
  return (dtype == MM_F32 && m <= 32) || (dtype == MM_F64 && m <= 16);
}
// plain backward only: also the Euclidean squared distance in fp32 (no Gram, one spare column for the row sums)
bool vec_gram_bwd_supports(int dtype, int kind, int64_t n, int m, int squared) {
  return vec_gram_supports(dtype, kind, n, m) ||
         (dtype == MM_F32 && kind == MM_EUCLIDEAN && squared && m <= 31 && n <= 32768);
}
template <typename T>
int vec_gram_loss_t(int kind, int loss_kind, const T* x, const T* target, const T* scale_raw, int64_t n, int m,
                    int64_t row_begin, int64_t row_end, double alpha, double eps, int terms, const double* loss_params, T* loss_out, T* grad,
                    T* slots, hipStream_t st) {
  hipError_t e = hipMemsetAsync(slots, 0, sizeof(T) * 2 * kLossSlots, st);
  if (e != hipSuccess) return int(e);
  LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, slots, loss_params};
  const int rc = vec_gram_bwd_launch(kind, loss_kind, x, target, n, m, row_begin, row_end, 1, grad, la, st);
  if (rc) return rc;
  gram_loss_finalize_kernel<T><<<dim3(1), dim3(64), 0, st>>>(slots, scale_raw, loss_out);
  e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}
// mm_vec_pdist_loss on the matrix cores (called from vec.hip when the configuration qualifies)
int vec_gram_loss(int dtype, int kind, int loss_kind, const void* x, const void* target, const void* scale_raw, int64_t n,
                  int m, int64_t row_begin, int64_t row_end, double alpha, double eps, int terms, const double* loss_params, void* loss_out,
                  void* grad, void* slots, hipStream_t st) {
  if (dtype == MM_F32)
    return vec_gram_loss_t<float>(kind, loss_kind, static_cast<const float*>(x), static_cast<const float*>(target),
                                  static_cast<const float*>(scale_raw), n, m, row_begin, row_end, alpha, eps, terms, loss_params,
                                  static_cast<float*>(loss_out), static_cast<float*>(grad), static_cast<float*>(slots), st);
  return vec_gram_loss_t<double>(kind, loss_kind, static_cast<const double*>(x), static_cast<const double*>(target),
                                 static_cast<const double*>(scale_raw), n, m, row_begin, row_end, alpha, eps, terms, loss_params,
                                 static_cast<double*>(loss_out), static_cast<double*>(grad), static_cast<double*>(slots),
                                 st);
}

}  // namespace mm

#ifdef MM_GRAM_STAMP
extern "C" int mm_dbg_read_gramf_stamps(void* host, size_t bytes) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(mm::g_gramf_stamps), bytes));
}
extern "C" int mm_dbg_read_gram_stamps(void* host, size_t bytes) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(mm::g_gram_stamps), bytes));
}
#endif

extern "C" int mm_vec_pdist_bwd_gram(int dtype, int kind, const void* x, const void* g, int64_t n, int m,
                                     int64_t row_begin, int64_t row_end, int squared, void* grad_x,
                                     mm_stream_t stream) {
  if (!x || !grad_x || n < 1 || m < 1 || row_begin < 0 || row_end > n || row_begin > row_end || n > (1 << 30))
    return MM_ERR_ARG;
  // 32-bit element offsets inside the kernel: n (n - 1) / 2 pairs and the row products must fit
  if (!vec_gram_bwd_supports(dtype, kind, n, m, squared)) return MM_ERR_UNSUPPORTED;
  if (!g && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F64)
    return vec_gram_bwd_launch(kind, MM_LOSS_NONE, static_cast<const double*>(x), static_cast<const double*>(g), n, m,
                               row_begin, row_end, squared, static_cast<double*>(grad_x),
                               LossArgs<double>{nullptr, 1.0, 0.0, 0, nullptr}, st);
  return vec_gram_bwd_launch(kind, MM_LOSS_NONE, static_cast<const float*>(x), static_cast<const float*>(g), n, m,
                             row_begin, row_end, squared, static_cast<float*>(grad_x),
                             LossArgs<float>{nullptr, 1.f, 0.f, 0, nullptr}, st);
}
