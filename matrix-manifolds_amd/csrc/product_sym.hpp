// Internal interface of the SYMMETRIC mixed-manifold pair kernel (product_sym.hip) for product_pairs.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "loss.hpp"
#include "product_args.hpp"

namespace mm {

constexpr int kPSW = 8;   // padded width of a vector factor in the symmetric kernel's node table
// elements of the node table [n + 1][W] appended to the product workspace (upper bound over the supported layouts)
inline size_t product_sym_table_elems(int64_t n) { return size_t(n + 1) * (kPMaxVec * kPSW + 12); }

// True if the symmetric kernel serves this layout and size (what product_sym_pairs checks); the width of a node-table row.
template <typename T> bool product_sym_applies(int nv, int sd, const PArgs<T>& pa, int64_t n);
inline int product_sym_table_width(int nv, int sd) { return (nv * kPSW + (sd > 0 ? sd * (sd + 1) : 0) + 3) / 4 * 4; }

// [Preparation (node table) unless `prepared`: the table holds the rows of the CURRENT points — product_step_kernel wrote
// them with the previous step's update] + pair kernel over rows [rb, re): fills the accumulators of `pa` and the loss slots
// exactly as product_pair_kernel does.  MM_ERR_UNSUPPORTED — the caller takes the ordered kernel — for node minibatches, a
// vector factor wider than 8 (Euclidean: 7), an SPD factor other than SPD(2) / SPD(3), small n (fp32 n < 1536, fp64 n < 640:
// the ordered kernel is faster there) unless MM_PRODUCT_SYM=1, or MM_PRODUCT_ORDERED=1.
template <typename T>
int product_sym_pairs(int loss_kind, int nv, int sd, const PArgs<T>& pa, const T* target, int64_t n, int64_t rb, int64_t re,
                      LossArgs<T> la, T* table, bool prepared, hipStream_t st);

}  // namespace mm
