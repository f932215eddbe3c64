// Internal interface between step.hip (mm_train_step_run) and vec.hip: the fused training step of a single vector-manifold
// factor (Euclidean / Lorentz / sphere) — the counterpart of spd_step.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"

namespace mm {

// The sizes the fused step serves: where the symmetric VALU pair kernel (vec_sym.hpp) is the fused objective's default.
bool vec_fused_step_supports(int dtype, int kind, int m);

// True if points[0] of `s` (a single vector factor) can take the fused step: a supported size and an optimizer rule with its
// state present.
bool vec_step_fusable(const mm_train_step* s);

// with_objective: [preparation unless s->ws_flags has MM_WS_PREPARED] -> pair kernel (loss + gradient sums, left in the
//   workspace) over rows [rb, re) -> ONE kernel: gradient from the sums (stored to points[0].grad) + loss record + optimizer
//   rule of points[0] + the padded copy of the new points for the next step (+ the scale's own update when it is a
//   momentum-free RSGD parameter: *scale_stepped = true).  Afterwards the workspace is prepared for the next call.
// !with_objective: the gradient in points[0].grad is final (sharded step, after the all-reduce): optimizer rule
//   (+ the scale's, as above, from loss_out[1]).
int vec_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st, bool* scale_stepped);

// The per-parameter optimizer rules (mm_vec_rsgd_step / _momentum_step / mm_vec_radam_step) on a register-resident padded
// point: m <= 16; MM_ERR_UNSUPPORTED otherwise (the caller then takes vec.hip's run-time-m kernels).
int vec_rule_step(int dtype, int kind, int optimizer, const void* x, const void* grad, void* xnew, int64_t cnt, int m,
                  double lr, double momentum, double dampening, double max_grad_norm, int exact, void* state0, void* state1,
                  double beta1, double beta2, double eps, int nc, double* step, unsigned* ticket, hipStream_t st);

// One launch for several vector-space parameters (points of vector factors, scales) that share an optimizer TYPE — the
// momentum-free RSGD or Riemannian Adam — each with its own hyper-parameters and state (vec.hip, vec_*_multi_kernel).
struct VecGroupParam {
  int kind, m;
  int64_t cnt;
  const void* x; const void* grad; void* xnew;
  void* state0; void* state1; double* step; unsigned* ticket;     // Adam: exp_avg, exp_avg_sq, step counter, ticket
  double lr, max_grad_norm, beta1, beta2, eps;
  int nc, exact;
};
int vec_group_step(int dtype, int optimizer, int count, const VecGroupParam* ps, hipStream_t st);

}  // namespace mm
