// The fused training step of a single vector-manifold factor (vec_step.hpp; mm_train_step_run): the pair kernel of
// vec_sym.hpp leaves its sums in the workspace, and ONE per-point kernel finishes the step — gradient, loss record,
// optimizer rule (optim/rsgd.py:29-82, optim/radam.py:62-98 with the maps of manifolds/{euclidean,lorentz,sphere}.py), the
// zero-padded copy of the new points for the next pair kernel, and a momentum-free RSGD scale.  A step is then TWO launches
// instead of preparation + pair kernel + loss record + point update + scale update.
//
// The rules are those of vec.hip's per-point kernels, written here on a point that is zero-padded to a compile-time width
// (4 / 8 / 12 / 16 / 24 / 32) and held in registers: every loop unrolls and nothing lives in scratch (vec.hip's kernels
// take any m <= 64 at run time: arrays of 64 in scratch, 9 us for 4039 points where this kernel needs 5).  The padding
// coordinates are zero and stay zero under every map of the three manifolds.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/mm_manifolds.h"
#include "adam.hpp"
#include "loss.hpp"
#include "smallmat.hpp"
#include "spd_ws.hpp"
#include "vec_rules.hpp"
#include "vec_step.hpp"
#include "vecfn.hpp"

namespace mm {

namespace {

template <typename T> struct VecStep {
  const T* x; T* xnew; int n, m;                 // (xnew == x: in place — every thread reads its whole point before it writes it)
  T* acc;                                        // gradient source: gacc [n][m] in the workspace (cleared here); null: `grad` is final
  T* grad;
  T* slots; const T* scale_raw; T* loss_out;     // the loss record, closed by block 0 (slots null: already closed)
  T* scale_x; T scale_lr, scale_clip;            // the scale's own RSGD update (null: stepped elsewhere / frozen)
  T* xpad;                                       // padded copy of the new points (null: not kept)
  VecRuleArgs<T> rule;
};

template <typename T, int KIND, int MP, int RULE>
__global__ __launch_bounds__(128) void vec_fused_step_kernel(VecStep<T> a) {
  using N = Num<T>;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    if (a.slots) loss_finalize<T>(a.slots, a.scale_raw, a.loss_out);
    if (a.scale_x && threadIdx.x == 0) {
      // the momentum-free RSGD rule for one scalar (Euclidean(1)): r = g, ||r|| = sqrt(max(g^2, 1e-8)), x' = x - lr clip r
      const T g = a.loss_out[1];
      T scale = -a.scale_lr;
      if (a.scale_clip > T(0)) scale *= N::min(a.scale_clip / N::sqrt(N::max(g * g, T(1e-8))), T(1));
      *a.scale_x = *a.scale_x + g * scale;
    }
  }
  const int64_t p0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int m = a.m, n = a.n;
  const bool in = p0 < n;
  const int64_t p = in ? p0 : 0;
  T beta2 = T(0), alpha = T(0);
  if constexpr (RULE == VRULE_ADAM) adam_coeffs(a.rule.adam, beta2, alpha);
  T xp[MP], g[MP], o[MP];
  load_padded<T, MP>(a.x, p, m, xp);
  if (a.acc) {   // the gradient from the pair kernel's sums
    load_padded<T, MP>(a.acc, p, m, g);   // gacc [n][m]: the pair kernel applied the manifold's map on the way
    if (in) {
#pragma unroll
      for (int k = 0; k < MP; ++k)
        if (k < m) a.acc[p * m + k] = T(0);
    }
    if (in) store_row<T, MP>(a.grad, p, m, g);
  } else {
    load_padded<T, MP>(a.grad, p, m, g);
  }
  pad_rule_point<T, KIND, MP, RULE>(xp, g, p, m, in, a.rule, beta2, alpha, o);
  if (in) {
    store_row<T, MP>(a.xnew, p, m, o);
    if (a.xpad) {
#pragma unroll
      for (int k = 0; k < MP; ++k) a.xpad[p * MP + k] = o[k];   // (the padding of o is zero)
    }
  }
  if constexpr (RULE == VRULE_ADAM) adam_tick(a.rule.adam.step, a.rule.adam.ticket, gridDim.x);
}

template <typename T, int KIND, int MP>
int launch_rule(const VecStep<T>& a, int rule, hipStream_t st) {
  const dim3 grid(unsigned((a.n + 127) / 128)), block(128);
  if (rule == VRULE_ADAM) vec_fused_step_kernel<T, KIND, MP, VRULE_ADAM><<<grid, block, 0, st>>>(a);
  else if (rule == VRULE_MOMENTUM) vec_fused_step_kernel<T, KIND, MP, VRULE_MOMENTUM><<<grid, block, 0, st>>>(a);
  else vec_fused_step_kernel<T, KIND, MP, VRULE_RSGD><<<grid, block, 0, st>>>(a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

template <typename T, int KIND>
int launch_pad(const VecStep<T>& a, int rule, hipStream_t st) {
  switch (pad_dim(a.m)) {
    case 4: return launch_rule<T, KIND, 4>(a, rule, st);
    case 8: return launch_rule<T, KIND, 8>(a, rule, st);
    case 12: return launch_rule<T, KIND, 12>(a, rule, st);
    case 16: return launch_rule<T, KIND, 16>(a, rule, st);
    default: break;
  }
  if constexpr (KIND == MM_EUCLIDEAN) {   // (the symmetric pair kernel's range for this kind; Lorentz / sphere stop at 16)
    if (pad_dim(a.m) == 24) return launch_rule<T, KIND, 24>(a, rule, st);
    if (pad_dim(a.m) == 32) return launch_rule<T, KIND, 32>(a, rule, st);
  }
  return MM_ERR_UNSUPPORTED;
}

template <typename T>
int vec_fused_step_t(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st, bool* scale_stepped) {
  const mm_step_param& p = s->points[0];
  const mm_step_param& q = s->scales[0];
  const int64_t n = s->n;
  const int m = p.dim, pad = pad_dim(m);
  VecSymWs<T> ws(s->ws, n, pad);
  *scale_stepped = false;
  if (with_objective) {
    bool finalized = false;
    const int rc = vec_sym_backward_pairs(s->dtype, p.kind, s->loss_kind, 1, p.x, s->target, n, m, rb, re, s->ws, q.x, s->alpha,
                                          s->eps, s->terms, s->loss_params, p.grad, s->loss_out, &finalized, st,
                                          (s->ws_flags & MM_WS_PREPARED) ? VSYM_STEP_PREPARED : VSYM_STEP);
    if (rc != MM_OK) return rc;
  }
  const bool fuse_scale = q.x && q.optimizer == MM_OPT_RSGD && q.momentum == 0.0;
  VecStep<T> a{static_cast<const T*>(p.x), static_cast<T*>(p.x), int(n), m,
               with_objective ? ws.acc : static_cast<T*>(nullptr), static_cast<T*>(p.grad),
               with_objective ? ws.slots : static_cast<T*>(nullptr), static_cast<const T*>(q.x), static_cast<T*>(s->loss_out),
               fuse_scale ? static_cast<T*>(q.x) : static_cast<T*>(nullptr), T(q.lr), T(q.max_grad_norm),
               with_objective ? ws.xpad : static_cast<T*>(nullptr),
               VecRuleArgs<T>{T(p.lr), T(p.momentum), T(p.dampening), T(p.max_grad_norm), p.exact,
                              static_cast<T*>(p.state0), static_cast<T*>(p.state1),
                              AdamArgs<T>{T(p.lr), T(p.beta1), T(p.beta2), T(p.adam_eps), T(p.max_grad_norm), p.nc, p.exact,
                                          p.step, p.ticket}}};
  const int rule = p.optimizer == MM_OPT_RADAM ? VRULE_ADAM : (p.momentum != 0.0 ? VRULE_MOMENTUM : VRULE_RSGD);
  int rc;
  if (p.kind == MM_EUCLIDEAN) rc = launch_pad<T, MM_EUCLIDEAN>(a, rule, st);
  else if (p.kind == MM_LORENTZ) rc = launch_pad<T, MM_LORENTZ>(a, rule, st);
  else rc = launch_pad<T, MM_SPHERE>(a, rule, st);
  if (rc == MM_OK) *scale_stepped = fuse_scale;
  return rc;
}

}  // namespace

// The per-parameter optimizer kernels (mm_vec_rsgd_step, ...) for m <= 16: the same kernel without a gradient source, a loss
// record or a padded copy.
int vec_rule_step(int dtype, int kind, int optimizer, const void* x, const void* grad, void* xnew, int64_t cnt, int m,
                  double lr, double momentum, double dampening, double max_grad_norm, int exact, void* state0, void* state1,
                  double beta1, double beta2, double eps, int nc, double* step, unsigned* ticket, hipStream_t st) {
  static const bool off = [] { const char* e = std::getenv("MM_VEC_RULE_GENERIC"); return e && e[0] == '1'; }();
  if (off || m > 16 || cnt >= (int64_t(1) << 31) || kind < MM_EUCLIDEAN || kind > MM_SPHERE) return MM_ERR_UNSUPPORTED;
  const int rule = optimizer == MM_OPT_RADAM ? VRULE_ADAM : (momentum != 0.0 ? VRULE_MOMENTUM : VRULE_RSGD);
  auto run = [&](auto tag) {
    using T = decltype(tag);
    VecStep<T> a{static_cast<const T*>(x), static_cast<T*>(xnew), int(cnt), m, nullptr, const_cast<T*>(static_cast<const T*>(grad)),
                 nullptr, nullptr, nullptr, nullptr, T(0), T(0), nullptr,
                 VecRuleArgs<T>{T(lr), T(momentum), T(dampening), T(max_grad_norm), exact, static_cast<T*>(state0),
                                static_cast<T*>(state1),
                                AdamArgs<T>{T(lr), T(beta1), T(beta2), T(eps), T(max_grad_norm), nc, exact, step, ticket}}};
    if (kind == MM_EUCLIDEAN) return launch_pad<T, MM_EUCLIDEAN>(a, rule, st);
    if (kind == MM_LORENTZ) return launch_pad<T, MM_LORENTZ>(a, rule, st);
    return launch_pad<T, MM_SPHERE>(a, rule, st);
  };
  if (dtype == MM_F32) return run(float{});
  if (dtype == MM_F64) return run(double{});
  return MM_ERR_ARG;
}

bool vec_fused_step_supports(int dtype, int kind, int m) {
  static const bool off = [] { const char* e = std::getenv("MM_VEC_STEP_UNFUSED"); return e && e[0] == '1'; }();
  static const bool gram = [] { const char* e = std::getenv("MM_VEC_LOSS_GRAM"); return e && e[0] == '1'; }();
  if (off || gram || !vec_sym_supports(dtype, kind, m)) return false;
  if (kind == MM_EUCLIDEAN) return true;
  return (kind == MM_LORENTZ || kind == MM_SPHERE) && m <= 16;   // (fp32 17 <= m <= 32: the matrix-core objective)
}

bool vec_step_fusable(const mm_train_step* s) {
  const mm_step_param& p = s->points[0];
  if (p.kind == MM_FACTOR_SPD || !p.x || !p.grad || p.count != s->n || s->n < 1 || s->n > kSpdMaxNodes || !s->ws) return false;
  if (!vec_fused_step_supports(s->dtype, p.kind, p.dim)) return false;
  // (anything else takes the unfused path, where mm_vec_pdist_loss reports MM_ERR_UNSUPPORTED / MM_ERR_ARG as before: the
  // fused pair kernel would read an unknown loss kind as "no loss" and step the points on the targets)
  if (s->loss_kind != MM_LOSS_STRESS && s->loss_kind != MM_LOSS_QUOTIENT) return false;
  if (s->loss_kind == MM_LOSS_QUOTIENT && !(s->terms & 3)) return false;
  if (p.optimizer == MM_OPT_RSGD) return p.momentum == 0.0 || p.state0;
  if (p.optimizer == MM_OPT_RADAM) return p.state0 && p.state1 && p.step && p.ticket;
  return false;
}

int vec_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st, bool* scale_stepped) {
  if (s->dtype == MM_F32) return vec_fused_step_t<float>(s, rb, re, with_objective, st, scale_stepped);
  return vec_fused_step_t<double>(s, rb, re, with_objective, st, scale_stepped);
}

}  // namespace mm
