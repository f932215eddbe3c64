// Internal interface between step.hip (mm_train_step_run) and spd.hip: the fused SPD training step.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"

namespace mm {

// True if points[0] of `s` (a single SPD factor) can take the fused step kernel: an optimizer rule with its state present.
bool spd_step_fusable(const mm_train_step* s);

// with_objective: [prep unless s->ws_flags has MM_WS_PREPARED] -> pair kernel (loss + accumulators) over rows [rb, re) ->
//   ONE kernel: finalize + optimizer rule of points[0] + tables of the new points (+ loss record; + the scale's own update
//   when it is a momentum-free RSGD parameter: *scale_stepped = true).
// !with_objective: the gradient in points[0].grad is final (sharded step, after the all-reduce): optimizer rule + tables.
// Either way the workspace holds the tables of the NEW points afterwards: the next call may pass MM_WS_PREPARED.
int spd_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st, bool* scale_stepped);
// The same with the objective over a node minibatch (s->batch_idx, s->batch; rows [rb, re) of the BATCH's pair list): spd_subset.hip
int spd_fused_train_step_subset(const mm_train_step* s, int64_t rb, int64_t re, hipStream_t st, bool* scale_stepped);

}  // namespace mm
