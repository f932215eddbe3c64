// Fused objective of a single SPD(d) factor (loss + both gradients in one pass over the pairs: objectives.py:16-45 on
// spd.py:175-181) and its fused training step (spd_step.hpp): the LOSS instantiations of the backward pair kernel and the
// per-point finalize / optimizer / tables kernel (spd_pair.hpp).  A translation unit of its own: compiled in parallel with spd.hip.
#include "spd_pair.hpp"

namespace mm {

constexpr int kFusedStepMaxD = 5;
bool spd_step_fusable(const mm_train_step* s) {
  if (!s || s->nf != 1 || s->n < 1 || s->n > kSpdMaxNodes || !s->ws) return false;
  const mm_step_param& p = s->points[0];
  // (SPD(2..5): the sizes the paper grid and the BASELINE configurations train; for d >= 6 a step is dominated by the Jacobi
  // pair kernel — hundreds of microseconds — and the 48 fused instantiations would add minutes to the build)
  if (p.kind != MM_FACTOR_SPD || p.dim < 2 || p.dim > kFusedStepMaxD || !p.x || !p.grad || p.count != s->n) return false;
  if (s->loss_kind != MM_LOSS_STRESS && s->loss_kind != MM_LOSS_QUOTIENT) return false;
  if (s->loss_kind == MM_LOSS_QUOTIENT && !(s->terms & 3)) return false;
  if (p.optimizer == MM_OPT_RSGD) return p.momentum == 0.0 || p.state0 != nullptr;
  if (p.optimizer == MM_OPT_RADAM) return p.state0 && p.state1 && p.step && p.ticket;
  return false;
}

int spd_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st,
                         bool* scale_stepped) {
  const int d = s->points[0].dim;
  if (s->dtype != MM_F32 && s->dtype != MM_F64) return MM_ERR_ARG;
#define MM_FUSED_D(T_, d_)                                                                                           \
  switch (d_) {                                                                                                      \
    case 2: return spd_fused_train_step_t<T_, 2>(s, rb, re, with_objective, st, scale_stepped);                      \
    case 3: return spd_fused_train_step_t<T_, 3>(s, rb, re, with_objective, st, scale_stepped);                      \
    case 4: return spd_fused_train_step_t<T_, 4>(s, rb, re, with_objective, st, scale_stepped);                      \
    case 5: return spd_fused_train_step_t<T_, 5>(s, rb, re, with_objective, st, scale_stepped);                      \
    default: return MM_ERR_UNSUPPORTED;                                                                              \
  }
  if (s->dtype == MM_F32) { MM_FUSED_D(float, d) }
  MM_FUSED_D(double, d)
#undef MM_FUSED_D
}

}  // namespace mm

using namespace mm;

extern "C" {

int mm_spd_fused_step_max_dim(void) { return kFusedStepMaxD < kSpdMaxD ? kFusedStepMaxD : kSpdMaxD; }

int mm_spd_pdist_loss(int dtype, int loss_kind, const void* x, const void* target, const void* scale_raw, int64_t n,
                      int d, int64_t row_begin, int64_t row_end, double alpha, double eps, int terms, const double* loss_params, double wmin,
                      double wmax, void* loss_out, void* grad_x, void* ws, int flags, mm_stream_t stream) {
  if (!x || !ws || !grad_x || !loss_out || n < 0 || row_begin < 0 || row_end > n || row_begin > row_end ||
      n > kSpdMaxNodes)
    return MM_ERR_ARG;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  if (!target && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    const size_t es = dtype == MM_F64 ? 8 : 4;
    hipError_t e = hipMemsetAsync(loss_out, 0, 2 * es, st);
    return e == hipSuccess ? MM_OK : int(e);
  }
  MM_DISPATCH(dtype, d,
              (spd_pdist_loss_t<T, D>(loss_kind, static_cast<const T*>(x), static_cast<const T*>(target),
                                      static_cast<const T*>(scale_raw), n, row_begin, row_end, alpha, eps, terms, loss_params, wmin,
                                      wmax, static_cast<T*>(loss_out), static_cast<T*>(grad_x), ws, flags, st)));
}

}  // extern "C"
