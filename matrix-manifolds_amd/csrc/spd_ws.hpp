// Workspace layout, launch constants and the folded triangular tile map shared by the SPD pair kernels
// (spd.hip: affine-invariant distance; spd_stein.hip: Stein divergence).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "../../include/mm_manifolds.h"
#include "loss.hpp"
#include "smallmat.hpp"

namespace mm {

constexpr int kBlock = 256;  // 4 wavefronts
// Launch shape of the backward (MI355X, SPD(3) fp32, n = 5000).  Round 1 dispatched tiles (rows per wavefront 4 / 6 / 8 / 12:
// 90.7 / 81.5 / 70.0 / 73.5 us); a persistent grid with a global atomic tile counter ran 118 us (same-address returning
// atomics serialise at ~18 ns each).  Round 2: ONE resident grid with statically balanced shares (ColWalk below), 50 us.
// Most nodes of one call: the pair kernels address the node tables and a row of the pair vector with 32-bit byte offsets
// (2 NP sizeof(T) <= 720 bytes of row operands per node).  4 M nodes are 8.8e12 pairs — 35 TB of fp32 distances.
constexpr int64_t kSpdMaxNodes = int64_t(1) << 22;
// Wavefronts per SIMD the pair kernels are compiled for (the second __launch_bounds__ argument).  The fp32 SPD(3) / SPD(4)
// backward needs 121 - 138 vector registers left alone — for several instantiations one allocation granule above the 128
// that let four wavefronts share a SIMD instead of three (SPD(4), n = 16384: 1157 -> 1069 us; the few spills land in the
// Jacobi fallback).  fp64 SPD(3) (forward 138, backward 217 registers) capped at 128 / 168 was measured and NOT adopted:
// reference init -4 % but the mid-training spread +15 % (spills in the Cayley path).
#ifndef MM_SPD4_BWD_WAVES
#define MM_SPD4_BWD_WAVES 4
#endif
#ifndef MM_SPD4_BWD_NC
#define MM_SPD4_BWD_NC 1
#endif
#ifndef MM_SPD4_FWD_NC
#define MM_SPD4_FWD_NC 1
#endif
// fp32 SPD(3) backward: columns per lane / wavefronts per SIMD (A/B builds: tools/snap_make.sh nc3 "-DMM_SPD3_BWD_NC=3 -DMM_SPD3_BWD_WAVES=3")
#ifndef MM_SPD3_BWD_NC
#define MM_SPD3_BWD_NC 2
#endif
#ifndef MM_SPD3_BWD_WAVES
#define MM_SPD3_BWD_WAVES 4
#endif
// fp64 SPD(3) backward: three wavefronts per SIMD (168 registers) since its series constants are scalar operands
// (smallmat.hpp, fma_sconst64) and the Cayley logarithm is the ring form — 187 registers left alone; the values spilled
// for the cap land in the Jacobi fallback.  Same box, n = 5000: reference init 96.4 -> 91.9 us, mid-training spread
// 127.8 -> 122.4, spectra beyond the Cayley gate (||log X|| = 0.6) 339 -> 353 (profiles/r04_experiments.md).
#ifndef MM_SPD3_F64_BWD_WAVES
#define MM_SPD3_F64_BWD_WAVES 3
#endif
template <typename T, int D> constexpr int bwd_min_waves() {
  return (sizeof(T) == 4 && D == 3) ? MM_SPD3_BWD_WAVES : ((sizeof(T) == 4 && D == 4) ? MM_SPD4_BWD_WAVES : ((sizeof(T) == 8 && D == 3) ? MM_SPD3_F64_BWD_WAVES : 1));
}
// ... with NCX columns per lane forced (0: the default of pair_cols_bwd): two columns of fp32 SPD(4) need ~185 registers —
// three wavefronts per SIMD (168 registers, the rest spilled) measured best (profiles/r03_experiments.md §2)
#ifndef MM_SPD4_BWD2_WAVES
#define MM_SPD4_BWD2_WAVES 3
#endif
template <typename T, int D, int NCX> constexpr int bwd_min_waves_nc() {
  return (NCX == 2 && sizeof(T) == 4 && D == 4) ? MM_SPD4_BWD2_WAVES : bwd_min_waves<T, D>();
}
// Wavefronts of a backward workgroup: they share one column block and flush its column-side sums once
template <typename T, int D> constexpr int bwd_waves() {
  // (the column-side combine buffer is D^2 x 64 values per wavefront: 4 wavefronts up to 64 KB of it, else 2, else 1)
  return int(sizeof(T)) * D * D * 64 * 4 <= 65536 ? 4 : (int(sizeof(T)) * D * D * 64 * 2 <= 65536 ? 2 : 1);
}
#ifndef MM_SPD_MAX_D
#define MM_SPD_MAX_D 9   // (development builds: -DMM_SPD_MAX_D=5 compiles spd.hip three times faster)
#endif
constexpr int kSpdMaxD = MM_SPD_MAX_D;   // the reference's tests go to 9 (tests/test_spd.py:15,26,62,70); D >= 6 spills to scratch

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
//   nodeLC T[n+1][2NP] {L_i^-1, L_i} interleaved: the backward's row operand, ONE scalar pointer and two scalar loads per row.
//                     Row n is padding: the backward requests the operands of the NEXT row unconditionally, and the masked
//                     last slot of a slice with an odd row count that ends at row n - 1 asks for row n (never used).
// The gradient w.r.t. the column point is L_i^-T [M A^-1] L_i^-1 with A^-1 = L_i^T X_j^-1 L_i, i.e.
// (L_i^-T M L_i^T) X_j^-1: the factor X_j^-1 is common to the whole column, so only M is formed per
// pair and X_j^-1 is applied once per point in finalize.
// No memset is ever needed: prep zeroes the accumulators of its point and finalize zeroes
// them again after reading (the fill kernels cost more than prep itself at n = 5000).
// Where the workgroups of the balanced walk start, remembered in the workspace (round 6; WalkShares::of_cached): 32 bytes per
// workgroup — the launch's key and the start it leads to.  Launches are at most four workgroups per CU.
constexpr int kShareTabEntries = 2048;
// (the vector manifolds' workspace — mm_vec_pdist_ws_bytes — carries the same table behind its accumulators [mp + 1][n], loss
// slots and zero-padded points [n + 1][mp]: bytes up to it, 32-byte aligned)
inline size_t vec_ws_tables_end(size_t esz, int64_t n, int mp) {
  return (esz * (size_t(n) * (mp + 1) + 2 * kLossSlots + size_t(n + 1) * mp) + 31) / 32 * 32;
}
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
  T* nodeLC;  // {L_i^-1, L_i} per node (backward row operand)
  int* shareTab;  // kShareTabEntries x 8 ints: where a workgroup of the balanced walk starts (WalkShares::of_cached)
  static size_t bad_bytes(int64_t n) { return (size_t(n) * sizeof(int) + 63) / 64 * 64; }
  static size_t tables_end(int64_t n, int d) {   // bytes up to the end of nodeLC
    const int np = d * (d + 1) / 2;
    return 64 + bad_bytes(n) + sizeof(T) * (size_t(n) * (6 * np + d * d + 1) + 2 * kLossSlots + 2 * np);
  }
  static size_t bytes(int64_t n, int d) { return (tables_end(n, d) + 31) / 32 * 32 + size_t(kShareTabEntries) * 32; }
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
    nodeLC = nodeLd + n;
    shareTab = reinterpret_cast<int*>(p + (tables_end(n, d) + 31) / 32 * 32);
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


// ---- balanced walk of the upper triangle (backward kernels) ---------------------------------------------------------
// Work is measured in ROWS OF COLUMN BLOCKS of bw = 64 or 128 columns: block c (columns bw c .. bw c + bw - 1) owns the
// rows [rb, hi(c)), hi(c) = min(re, bw c + bw - 1) (row i has a pair in the block iff i < bw c + bw - 1; re = min(row_end, n-1)).  The blocks' row
// ranges, concatenated block after block, form a line of W units, and workgroup w of G takes the units
// [start(w), start(w+1)), start(w) = floor(W/G) w + min(w, W mod G): every workgroup gets the same amount of arithmetic
// to within one row, so a launch of exactly the resident capacity has no tail (tiles handed out by the dispatcher never
// filled the machine and ended below half occupancy — profiles/r02_timeline_tiles64.txt), and a workgroup walks DOWN
// a column block, so the column-side sums stay in registers until the block changes.
// counts: 0 for c < c0 = (rb+1)/bw; bw c + bw - 1 - rb for c0 <= c < c1 = max(c0, re/bw); R = re - rb from c1 on.
struct ColWalk {
  int rb, re, ncb, c0, c1, bw;   // bw = columns per block (64 per column a lane owns)
  __host__ __device__ ColWalk(int n, int row_begin, int row_end, int block_width = 64) {
    bw = block_width;
    rb = row_begin;
    re = row_end < n - 1 ? row_end : n - 1;
    if (re < rb) re = rb;
    ncb = (n + bw - 1) / bw;
    c0 = (rb + 1) / bw;
    c1 = re / bw > c0 ? re / bw : c0;
    if (c1 > ncb) c1 = ncb;
    if (c0 > ncb) c0 = ncb;
  }
  __host__ __device__ int hi(int c) const { const int d = bw * c + bw - 1; return d < re ? d : re; }
  __host__ __device__ int64_t prefix(int c) const {   // units before column block c
    if (c <= c0 || re == rb) return 0;
    const int64_t m = c < c1 ? c : c1;
    int64_t s = (bw / 2) * (m * (m - 1) - int64_t(c0) * (c0 - 1)) + (m - c0) * int64_t(bw - 1 - rb);
    if (c > c1) s += int64_t(c - c1) * (re - rb);
    return s;
  }
  __host__ __device__ int64_t total() const { return prefix(ncb); }
  __host__ __device__ int find(int64_t p) const {     // the column block holding unit p (0 <= p < total())
    int lo = c0, hi_ = ncb - 1;
    while (lo < hi_) {
      const int mid = (lo + hi_ + 1) / 2;
      if (prefix(mid) <= p) lo = mid; else hi_ = mid - 1;
    }
    return lo;
  }
  // total = q g + r, 0 <= r < g, for total < 2^52 and g < 2^31 WITHOUT a 64-bit integer division: on the device that is a
  // software routine of several hundred instructions — two of them sat in the prologue of every workgroup of every kernel
  // that walks the triangle (round 5: ~1 us each at four wavefronts per SIMD).  One fp64 division and a correction step.
  static __host__ __device__ void divmod_small(int64_t total, int64_t g, int64_t& q, int64_t& r) {
    q = int64_t(double(total) / double(g));
    r = total - q * g;
    if (r < 0) { --q; r += g; } else if (r >= g) { ++q; r -= g; }
  }
  static __host__ __device__ int64_t share_begin(int64_t total, int64_t w, int64_t g) {
    int64_t q, r;
    divmod_small(total, g, q, r);
    return q * w + (w < r ? w : r);
  }
  // The block holding unit p in closed form (round 5): the binary search above is ~7 evaluations of prefix() — with the two
  // 64-bit divisions of share_begin, ~1500 scalar instructions in front of every workgroup's first load (2.4 us of the headline
  // backward's 37 at the median, up to 11 us for the workgroups that also miss the instruction cache there: the stragglers
  // that end the launch; tools/stamp_timeline.py, profiles/r05_experiments.md).  Triangular part c0 <= c <= c1:
  //   prefix(c0 + u) = A u^2 + Bs u  -> the positive root, then a step or two either way against prefix() itself; from c1 on the
  // blocks are R = re - rb units each.
  __host__ __device__ int find_fast(int64_t p, int cross = 0) const {
    int64_t off;
    return find_fast_off(p, cross, &off);
  }
  // ... and the unit's offset inside its block, *off = p - aprefix(c) (what enter() needs: one evaluation of aprefix() less).
  // `capped` (host checks): set when a stepping loop used up its four steps — tools/micro/walk_check.hip requires it never to be.
  __host__ __device__ int find_fast_off(int64_t p, int cross, int64_t* off, bool* capped = nullptr) const {
    // (cross > 0: the same search in the line that has `cross` extra units in front of every block's rows — aprefix())
    // The guess in fp32 (one v_sqrt_f32 / one v_rcp_f32: an fp64 square root and two fp64 divisions were 1.3 us of every
    // workgroup's prologue), made exact by stepping against the line itself — the guess is within a block of the answer for
    // every n the entry points accept (2^16 blocks: 24 bits resolve 0.004 of a block).
    const int last = ncb - 1;
    if (capped) *capped = false;
    if (last <= c0) { *off = p; return c0; }
    int c;
    const int64_t pc1 = aprefix(c1, cross);
    if (p >= pc1) {
      const float R = float(re - rb + cross);
      c = c1 + int(float(p - pc1) / (R > 0.f ? R : 1.f));
    } else {
      // in u = c - c0: aprefix = A u^2 + Bs u,  A = bw / 2,  Bs = bw / 2 - 1 + (bw c0 - rb) + cross  (|bw c0 - rb| < bw: no
      // large terms cancel, whatever the row shard)
      const float A = 0.5f * float(bw), Bs = A - 1.0f + float(bw * c0 - rb + cross);
      c = c0 + int((-Bs + sqrtf(Bs * Bs + 4.0f * A * float(p))) / (2.0f * A));
    }
    c = c < c0 ? c0 : (c > last ? last : c);
    // Stepped exact: ONE evaluation of aprefix() at the guess, then block lengths added or taken away (block c holds
    // hi(c) - rb rows and `cross` entry units: aprefix(c + 1) = aprefix(c) + hi(c) - rb + cross) — round 5 evaluated aprefix()
    // for every comparison and once more in enter(): five evaluations of ~25 scalar instructions in front of every workgroup's
    // first load, now two (this one and pc1).  At most four steps either way as COUNTED loops (round 6: the same loops written
    // `while (cond)` came out of the compiler as 276 straight-line scalar instructions and 16 spilled scalar registers — the
    // headline backward 41.8 -> 43.3 us, profiles/r06_experiments.md section 5), and exact by CONSTRUCTION: a loop that uses up
    // its four steps hands over to a bisection (never taken: tools/micro/walk_check.hip requires the guess to be within four
    // blocks at every block boundary +- 2 of every n, shard and cross up to n = 2^22).
    int64_t P = aprefix(c, cross);
    int ku = 0, kd = 0;
    for (; ku < 4 && c < last; ++ku) {
      const int64_t L = int64_t(hi(c) - rb + cross);
      if (P + L > p) break;
      P += L;
      ++c;
    }
    for (; kd < 4 && c > c0 && P > p; ++kd) {
      --c;
      P -= int64_t(hi(c) - rb + cross);
    }
    if (__builtin_expect(ku == 4 || kd == 4, 0)) {
      if (capped) *capped = true;
      int lo = c0, hi_ = last;
#pragma nounroll
      while (lo < hi_) {
        const int mid = (lo + hi_ + 1) / 2;
        if (aprefix(mid, cross) <= p) lo = mid; else hi_ = mid - 1;
      }
      c = lo;
      P = aprefix(c, cross);
    }
    *off = p - P;
    return c;
  }
  // (host checks: the guess was within four blocks, i.e. the stepping loops did not hand over to the bisection — a drifting guess
  // would cost every workgroup's prologue and must be SEEN, not silently absorbed)
  __host__ __device__ bool find_fast_converges(int64_t p, int cross = 0) const {
    int64_t off;
    bool capped;
    find_fast_off(p, cross, &off, &capped);
    return !capped;
  }
  // Shares that PAY for entering a column block (round 5).  A workgroup whose share spans a block boundary flushes the column
  // sums of the block it leaves (LDS, barrier, d^2 NC atomics per lane), requests the operands of the next block's columns and
  // starts its request pipeline again: +2.0 ... 3.2 us of 35 on the headline backward — and since the four workgroups of a CU
  // finish in dispatch order (oldest first), a boundary in the share of a YOUNGEST workgroup was what ended the launch: the six
  // slowest workgroups of 1024 were all of this kind, 42.3 us against 39.5 for the slowest of the others
  // (tools/stamp_timeline.py, profiles/r05_experiments.md).  The line of units therefore gets `cross` extra units in front of
  // every block's rows, the shares are equal in THAT line (WalkShares on total_aug), and enter() maps a share's start back:
  // a share that holds a block start holds `cross` rows less.  aprefix(c) = prefix(c) + cross (c - c0).
  __host__ __device__ int64_t aprefix(int c, int cross) const { return prefix(c) + int64_t(cross) * (c > c0 ? c - c0 : 0); }
  __host__ __device__ int64_t total_aug(int cross) const { return total() + int64_t(cross) * (ncb > c0 ? ncb - c0 : 0); }
  // A share [a, a + count) of the augmented line, entered: the block that holds a, the first row of it to process and the units
  // the share has left for rows (a start inside a block's entry zone pays the rest of the zone and starts at the block's
  // first row).  The walker then spends one unit per row and `cross` units whenever it moves on to the next block.
  __host__ __device__ void enter(int64_t a, int count, int cross, int* block, int* row, int* left) const {
    int64_t off;
    const int c = find_fast_off(a, cross, &off);
    *block = c;
    if (off < cross) { *row = rb; *left = count - int(cross - off); }
    else { *row = rb + int(off - cross); *left = count; }
  }
};

// Shares of the balanced walk, cut on the HOST (kernel argument, by value): workgroup w of `grid` takes the units
// [q w + min(w, r), ...), count q + (w < r).  A 64-bit division is a software routine on the device — two of them, with the
// binary search for the share's first block, were ~1500 scalar instructions in front of every workgroup's first load: 2.4 us of
// the headline backward's 37 per workgroup at the median and up to 11 us for the stragglers that end the launch (round 5,
// tools/stamp_timeline.py phase marks; now ColWalk::find_fast and this: 1.3 us / 6.7 us).
// (Also measured and NOT kept: shares tilted by dispatch slot — of the four workgroups a CU hosts the youngest finishes ~3 us
// after the oldest, vector issue being arbitrated oldest first — +-1 .. 6 % per slot: no gain either way; `cross` units charged
// per column-block start: none.  profiles/r05_experiments.md)
struct WalkShares {
  int64_t q;
  int r;
  int cross;   // units charged per column-block start (ColWalk::enter); 0: `units` is the plain line
  WalkShares() = default;
  WalkShares(int64_t units, int64_t grid, int cross_ = 0) : q(units / grid), r(int(units % grid)), cross(cross_) {}
  __device__ __forceinline__ void of(int w, int64_t& begin, int& count) const {
    begin = q * int64_t(w) + (w < r ? w : r);
    count = int(q) + (w < r ? 1 : 0);
  }
  // ... entered (ColWalk::enter): the block and row the workgroup starts at and its budget of units
  __device__ __forceinline__ void of(const ColWalk& walk, int w, int& block, int& row, int& left) const {
    int64_t a;
    int cnt;
    of(w, a, cnt);
    walk.enter(a, cnt, cross, &block, &row, &left);
  }
  // ... REMEMBERED in the workspace (round 6).  The start of workgroup w is a pure function of the walk (rb, re, ncb, bw), the cut
  // (q, r, cross) and w — ~200 scalar instructions of 64-bit products, a square root and a stepped search that sixteen wavefronts
  // per CU execute at once at the head of every launch: 1.3 us at the median and up to 4.5 us for the workgroups that then end
  // the launch (tools/stamp_timeline.py, profiles/r06_timeline_warm.txt).  A training loop — and a replayed graph — launches
  // the same walk again and again: entry w of `tab` holds {rb, re, ncb | w, q, r | cross | bw} and the start they lead to, 32
  // bytes, one scalar load.  An entry is its own proof: it is used only if ALL FIVE key words match (whatever the memory held
  // before — an uninitialised workspace, another embedding's walk — cannot match the key and carry a wrong start, because a
  // start stored under a key is the start of that key), and a workgroup that misses computes as before and stores its entry.
  // No ordering between workgroups, launches or streams is relied on: every workgroup reads and writes only its own entry.
  __device__ __forceinline__ void of_cached(const ColWalk& walk, int w, int* __restrict__ tab, int& block, int& row, int& left) const {
    const bool usable = tab != nullptr && w < kShareTabEntries && r < (1 << 18) && q < (int64_t(1) << 31) && walk.ncb < (1 << 16);
    const int k4 = (r << 13) | (cross << 2) | ((walk.bw >> 6) - 1);   // (cross <= 1024, bw in {64, 128, 192, 256})
    // (w is part of the key: the table's place in a workspace depends on n, so one buffer used for two embeddings whose walks have
    // the same key — row shards of n = 257 and n = 300 have — shows the second launch the first one's entries SHIFTED)
    const int k2 = walk.ncb | (w << 16);
    if (usable) {
      const int* e = tab + 8 * w;
      const int e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4], e5 = e[5], e6 = e[6], e7 = e[7];
      if (e0 == walk.rb && e1 == walk.re && e2 == k2 && e3 == int(q) && e4 == k4) {
        block = e5;
        row = e6;
        left = e7;
        return;
      }
    }
    of(walk, w, block, row, left);
    if (usable && threadIdx.x == 0) {
      int* e = tab + 8 * w;
      e[5] = block; e[6] = row; e[7] = left;
      e[0] = walk.rb; e[1] = walk.re; e[2] = k2; e[3] = int(q); e[4] = k4;
    }
  }
};

// Compute units of the current device (cached)
inline int device_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cache[dev] = cus;
  }
  return cache[dev];
}

// Workgroups of `Kernel` that are resident at once on the current device (occupancy x compute units), cached per
// kernel INSTANTIATION and device: the kernel is a non-type template parameter, so every <T, D, LOSS, SQ> variant has its
// own cache (keyed on the function-pointer TYPE, all variants of one element type shared the entry of whichever ran
// first).  The cache entries are atomics: concurrent first calls compute the same value twice, nothing worse.
// hipOccupancyMaxActiveBlocksPerMultiprocessor can report one block too many per CU for kernels
// with 81-96 scalar registers (MI355X_MICROARCH.md, Residency): capped at 7 blocks of 256 threads.
template <auto Kernel> inline int resident_workgroups(int block_threads) {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cache[dev].load(std::memory_order_relaxed) == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kernel, block_threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int cap = (7 * 256) / block_threads > 0 ? (7 * 256) / block_threads : 1;
    if (per_cu > cap) per_cu = cap;
    cache[dev].store(per_cu * cus, std::memory_order_relaxed);
  }
  return cache[dev].load(std::memory_order_relaxed);
}

}  // namespace mm
