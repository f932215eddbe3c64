// Argument structures of the mixed-manifold pair kernels (product_pairs.hip: every ORDERED pair, any vector width <= 16,
// node minibatches; product_sym.hip: every UNORDERED pair once, vector factors of width <= 8, full batches) — they fill
// the same accumulators, so the finalize / step kernels of product_pairs.hip serve both.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mm {

constexpr int kPMaxVec = 3;
constexpr int kPMP = 16;   // padded dimension of a vector factor
constexpr int kPMaxTI = 16;  // most rows per wavefront
constexpr int kPCols = 64;   // columns per workgroup (n = 1025 wastes 6 % of the lanes; 256 columns would waste 20 %)
constexpr int kPWaves = 4;   // wavefronts per workgroup: same columns, consecutive row tiles; their column
                             // sums are combined in LDS before the flush (a quarter of the atomics)

template <typename T> struct PVec {
  const T* x;          // [n][m]
  const T* scale_raw;  // device scalar
  T* acc;              // [(kPMP + 1)][n]: sum_i w x_i, and sum_i w (Euclidean)
  T* grad;             // [n][m]
  int m, kind, slot;   // slot = position in the caller's factor list (for loss_out)
};
template <typename T> struct PSpd {
  const T* x;          // [n][D][D]
  const T* scale_raw;
  T* accS;             // [D*D][n]
  T* grad;             // [n][D][D]
  T wmin, wmax;
  int slot;
};
template <typename T> struct PArgs {
  PVec<T> v[kPMaxVec];
  PSpd<T> s;
  int nf;
  // node minibatch (train.py:198-222): the n points of this call are rows idx[0..n) of the factors' full
  // tables, targets come from the dense matrix, gradients go to rows idx[.] of full-size buffers
  const int64_t* idx;  // null: all nodes, in order
  const T* dense;      // [dense_n][dense_n] targets (with idx); null: `target` is the pair vector
  int64_t dense_n;
};
template <typename T> __device__ __forceinline__ int64_t node_of(const PArgs<T>& pa, int j) {
  // (a minibatch comes with its dense target matrix, dense_n = the tables' row count: ids are clamped into them — the index
  // vector is caller data no kernel validates; a bad index yields wrong numbers, not an access outside the buffers)
  return pa.idx ? int64_t(min(uint64_t(pa.idx[j]), uint64_t(pa.dense_n - 1))) : int64_t(j);
}

// KC: the vector factors' kinds, two bits each (factor f: (KC >> 2 f) & 3), or -1 = read pa.v[f].kind in the row loop.
// With kinds known at run time only, every row pays ~20 scalar branches around the three places the kinds differ (inner
// product, distance function, its derivative): the csphd pair kernel 15.3 -> 13.2 us with them as constants (round 4,
// profiles/r04_experiments.md).  Instantiated for products of one or two narrow vector factors (3 + 9 combinations; three
// factors would be 27: they keep the run-time form).
template <int KC, typename T> __device__ __forceinline__ int pkind_of(const PArgs<T>& pa, int f) {
  if constexpr (KC >= 0) return (KC >> (2 * f)) & 3; else return pa.v[f].kind;
}
#define MM_PKIND(f) (pkind_of<KC>(pa, (f)))

// f(integral_constant<int, C>) for the compile-time C that equals `code` (two bits per factor, every digit a valid kind)
template <int NV, int C = 0, typename F> bool for_kind_code(int code, F&& f) {
  if constexpr (C >= (1 << (2 * NV))) {
    return false;
  } else {
    constexpr bool valid = (C & 3) <= MM_SPHERE && ((C >> 2) & 3) <= MM_SPHERE;
    if constexpr (valid) {
      if (code == C) { f(std::integral_constant<int, C>{}); return true; }
    }
    return for_kind_code<NV, C + 1>(code, f);
  }
}

}  // namespace mm
