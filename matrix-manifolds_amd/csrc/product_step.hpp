// Internal interface between step.hip (mm_train_step_run) and product_pairs.hip: the fused training step of a product
// embedding — the counterpart of spd_step.hpp / vec_step.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"

namespace mm {

// True if every factor's points carry an optimizer rule with its state (momentum-free or heavy-ball RSGD, Riemannian Adam).
bool product_step_fusable(const mm_train_step* s);

// The mixed-manifold pair kernel over rows [rb, re), then ONE kernel: sums -> gradients (stored to points[k].grad), the loss
// record, the optimizer rule of every factor's points, and the update of every scale that is a momentum-free RSGD parameter
// (scale_stepped[k] = true) — instead of finalize + one launch per SPD factor + one per group of vector-space parameters.
int product_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, hipStream_t st, bool* scale_stepped /* [nf] */);

}  // namespace mm
