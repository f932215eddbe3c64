// Workspace layout, launch constants and the folded triangular tile map shared by the SPD pair kernels
// (spd.hip: affine-invariant distance; spd_stein.hip: Stein divergence).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mm_manifolds.h"
#include "loss.hpp"
#include "smallmat.hpp"

namespace mm {

constexpr int kBlock = 256;  // 4 wavefronts
// Tile-shape sweep on MI355X (SPD(3) fp32, n = 5000, backward): rows per wavefront 4 / 6 / 8 / 12 -> 90.7 / 81.5 /
// 70.0 / 73.5 us; wavefronts per workgroup 8 -> +1.5 us.  Persistent workgroups (one launch-filling grid that
// loops over tiles) were measured slower too: 118 us with a global atomic tile counter (same-address returning
// atomics serialise at ~18 ns each), 75-81 us with a static round-robin (78 VGPRs -> 6 wavefronts per SIMD).
#ifndef MM_BWD_G_AHEAD
#define MM_BWD_G_AHEAD 3   // rows of the pair vector requested ahead of their use in the backward
#endif
#ifndef MM_BWD_WAVES
#define MM_BWD_WAVES 4
#endif
// backward: wavefronts that share one 64-column tile (one column-side atomic flush per workgroup);
// 4 where the LDS combine buffer of 8 would not fit (fp64, D = 5)
template <typename T, int D> constexpr int bwd_waves() { return (sizeof(T) == 4 && D <= 4) ? MM_BWD_WAVES : 4; }
constexpr int kSpdMaxD = 5;

__host__ __device__ inline int64_t pair_off(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

// Workspace layout (T = element type, NP = d(d+1)/2):
//   [0,64)            word 0 = number of non-PD points (written by mm_spd_status' reduction)
//   bad    int[n]     per-point "Cholesky failed" flag, written unconditionally by prep
//   nodeL  T[n][NP]   packed lower L_i^-1
//   nodeX  T[n][NP]   packed sym(X_i)
//   nodeC  T[n][NP]   packed lower Cholesky factor L_i
//   accM   T[NP][n]   row-side accumulators     sum_j M_ij,            M_ij = 2 g log(A_ij)
//   accS   T[D*D][n]  column-side accumulators  sum_i L_i^-T M_ij L_i^T
//   loss   T[2][256]  fused-loss partial sums (loss, d loss / d softplus(scale)), spread over 256 slots
//   nodeLd T[n]       log det X_i = 2 sum log diag L_i (Stein divergence, spd_stein.hip)
// The gradient w.r.t. the column point is L_i^-T [M A^-1] L_i^-1 with A^-1 = L_i^T X_j^-1 L_i, i.e.
// (L_i^-T M L_i^T) X_j^-1: the factor X_j^-1 is common to the whole column, so only M is formed per
// pair and X_j^-1 is applied once per point in finalize.
// No memset is ever needed: prep zeroes the accumulators of its point and finalize zeroes
// them again after reading (the fill kernels cost more than prep itself at n = 5000).
template <typename T> struct Ws {
  int* status;
  int* bad;
  T* nodeL;
  T* nodeX;
  T* nodeC;
  T* accM;
  T* accS;
  T* loss;
  T* nodeLd;  // log det X_i (Stein divergence)
  static size_t bad_bytes(int64_t n) { return (size_t(n) * sizeof(int) + 63) / 64 * 64; }
  static size_t bytes(int64_t n, int d) {
    const int np = d * (d + 1) / 2;
    return 64 + bad_bytes(n) + sizeof(T) * (size_t(n) * (4 * np + d * d + 1) + 2 * kLossSlots);
  }
  Ws(void* base, int64_t n, int d) {
    const int np = d * (d + 1) / 2;
    char* p = static_cast<char*>(base);
    status = reinterpret_cast<int*>(p);
    bad = reinterpret_cast<int*>(p + 64);
    nodeL = reinterpret_cast<T*>(p + 64 + bad_bytes(n));
    nodeX = nodeL + n * np;
    nodeC = nodeX + n * np;
    accM = nodeC + n * np;
    accS = accM + n * np;
    loss = accS + n * d * d;
    nodeLd = loss + 2 * kLossSlots;
  }
};

// Folded triangular grid.  Row tile y needs the column blocks from the one holding its
// first super-diagonal entry to the last; pairing tile y with tile gy-1-y makes every
// grid row about equally long, so (almost) no launched workgroup is empty.
struct TileId { int i0, jbase; bool ok; };
template <int TI, int BW = kBlock> __device__ __forceinline__ TileId fold_tile(int n, int row_begin, int row_end) {
  const int gy = (row_end - row_begin + TI - 1) / TI;
  const int nJB = (n + BW - 1) / BW;
  int y = blockIdx.y, x = blockIdx.x;
  int cb0 = (row_begin + y * TI + 1) / BW;
  int cnt = nJB - cb0;
  if (x >= cnt) {
    x -= cnt;
    const int y2 = gy - 1 - y;
    if (y2 <= y) return {0, 0, false};
    y = y2;
    cb0 = (row_begin + y * TI + 1) / BW;
    cnt = nJB - cb0;
    if (x >= cnt) return {0, 0, false};
  }
  return {row_begin + y * TI, (cb0 + x) * BW, true};
}
template <int TI, int BW = kBlock> inline dim3 fold_grid(int64_t n, int64_t rb, int64_t re) {
  const int gy = int((re - rb + TI - 1) / TI);
  const int nJB = int((n + BW - 1) / BW);
  int gx = 0;
  for (int y = 0; y < (gy + 1) / 2; ++y) {
    const int y2 = gy - 1 - y;
    int c = nJB - int((rb + int64_t(y) * TI + 1) / BW);
    if (y2 > y) c += nJB - int((rb + int64_t(y2) * TI + 1) / BW);
    gx = c > gx ? c : gx;
  }
  return dim3(gx > 0 ? gx : 1, (gy + 1) / 2 > 0 ? (gy + 1) / 2 : 1);
}

}  // namespace mm
