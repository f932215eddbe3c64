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
#include "vec_step.hpp"
#include "vecfn.hpp"

namespace mm {

namespace {

template <typename T, int KIND, int MP> struct PadRule {
  using N = Num<T>;
  // <u, v>: Minkowski for the hyperboloid (lorentz.py:101-118), Euclidean otherwise
  static __device__ __forceinline__ T dot(const T (&u)[MP], const T (&v)[MP]) {
    if (KIND == MM_LORENTZ) {
      T s = T(0);
#pragma unroll
      for (int k = 1; k < MP; ++k) s = N::fma(u[k], v[k], s);
      return N::fma(-u[0], v[0], s);
    }
    T s = T(0);
#pragma unroll
    for (int k = 0; k < MP; ++k) s = N::fma(u[k], v[k], s);
    return s;
  }
  static __device__ __forceinline__ T edot(const T (&u)[MP], const T (&v)[MP]) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < MP; ++k) s = N::fma(u[k], v[k], s);
    return s;
  }
  // Manifold.norm (base.py:29-33)
  static __device__ __forceinline__ T norm(const T (&u)[MP]) { return N::sqrt(N::max(dot(u, u), T(kEps))); }
  // egrad2rgrad: lorentz.py:52-57 (flip the time coordinate, then u + <x,u>_L x), sphere.py:41-44, identity
  static __device__ __forceinline__ void rgrad(const T (&xp)[MP], const T (&g)[MP], T (&o)[MP]) {
    if (KIND == MM_EUCLIDEAN) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = g[k];
    } else if (KIND == MM_LORENTZ) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = (k == 0) ? -g[k] : g[k];
      const T d = dot(xp, o);
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(d, xp[k], o[k]);
    } else {
      const T d = edot(xp, g);
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(-d, xp[k], g[k]);
    }
  }
  // exp / retr: lorentz.py:59-62 (retr == exp, base.py:49-50), sphere.py:51-59, x + u
  static __device__ __forceinline__ void step(const T (&xp)[MP], const T (&u)[MP], int exact, T (&o)[MP]) {
    if (KIND == MM_EUCLIDEAN) {
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = xp[k] + u[k];
    } else if (KIND == MM_LORENTZ) {
      const T un = N::max(N::sqrt(N::max(dot(u, u), T(0))), T(kEps));
      const T ch = ::cosh(un), sh = ::sinh(un) / un;
#pragma unroll
      for (int k = 0; k < MP; ++k) o[k] = N::fma(xp[k], ch, sh * u[k]);
    } else {
      const T nu = N::sqrt(N::max(edot(u, u), T(kEps)));
      if (exact && nu > T(kEps)) {
        const T c = ::cos(nu), s = ::sin(nu) / nu;
#pragma unroll
        for (int k = 0; k < MP; ++k) o[k] = N::fma(xp[k], c, s * u[k]);
      } else {
        T nn = T(0);
#pragma unroll
        for (int k = 0; k < MP; ++k) { o[k] = xp[k] + u[k]; nn = N::fma(o[k], o[k], nn); }
        const T inv = T(1) / N::sqrt(N::max(nn, T(kEps)));
#pragma unroll
        for (int k = 0; k < MP; ++k) o[k] *= inv;
      }
    }
  }
  // transport of a tangent b from x to y: lorentz.py:79-82; sphere: proju(y, b) (base.py:65-66); identity
  static __device__ __forceinline__ void transport(const T (&xp)[MP], const T (&y)[MP], T (&b)[MP]) {
    if (KIND == MM_LORENTZ) {
      const T xy = dot(xp, y), uy = dot(b, y);
      const T g = uy / (T(1) - xy);
#pragma unroll
      for (int k = 0; k < MP; ++k) b[k] = N::fma(g, xp[k] + y[k], b[k]);
    } else if (KIND == MM_SPHERE) {
      const T d = edot(y, b);
#pragma unroll
      for (int k = 0; k < MP; ++k) b[k] = N::fma(-d, y[k], b[k]);
    }
  }
};

enum { VRULE_RSGD = 0, VRULE_MOMENTUM = 1, VRULE_ADAM = 2 };
template <typename T> struct VecStep {
  T* x; int n, m;
  T* acc;                                        // gradient source: the workspace sums (cleared here); null: `grad` is final
  T* grad;
  T* slots; const T* scale_raw; T* loss_out;     // the loss record, closed by block 0 (slots null: already closed)
  T* scale_x; T scale_lr, scale_clip;            // the scale's own RSGD update (null: stepped elsewhere / frozen)
  T* xpad;                                       // padded copy of the new points (null: not kept)
  T lr, momentum, dampening, max_grad_norm; int exact;
  T* state0; T* state1;                          // momentum buffer / exp_avg, exp_avg_sq
  AdamArgs<T> adam;
};

// zero-padded row p of a [n][m] table (requests from clamped addresses, masked afterwards: a load under `k < m` would sit in
// its own basic block)
template <typename T, int MP> __device__ __forceinline__ void load_padded(const T* t, int64_t p, int m, T (&o)[MP]) {
#pragma unroll
  for (int k = 0; k < MP; ++k) o[k] = t[p * m + min(k, m - 1)];
#pragma unroll
  for (int k = 0; k < MP; ++k) o[k] = k < m ? o[k] : T(0);
}
template <typename T, int MP> __device__ __forceinline__ void store_row(T* t, int64_t p, int m, const T (&v)[MP]) {
#pragma unroll
  for (int k = 0; k < MP; ++k)
    if (k < m) t[p * m + k] = v[k];
}

template <typename T, int KIND, int MP, int RULE>
__global__ __launch_bounds__(128) void vec_fused_step_kernel(VecStep<T> a) {
  using N = Num<T>;
  using R = PadRule<T, KIND, MP>;
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
  if constexpr (RULE == VRULE_ADAM) adam_coeffs(a.adam, beta2, alpha);
  T xp[MP], g[MP], r[MP], o[MP];
  load_padded<T, MP>(a.x, p, m, xp);
  if (a.acc) {   // the gradient from the pair kernel's sums
    if (KIND == MM_EUCLIDEAN) {   // acc [MP + 1][n]: sum w x_i per coordinate, then sum w:  sum w 2 (x_j - x_i)
      const T wsum = a.acc[size_t(MP) * n + p];
#pragma unroll
      for (int k = 0; k < MP; ++k) g[k] = a.acc[size_t(k) * n + p];
#pragma unroll
      for (int k = 0; k < MP; ++k) g[k] = k < m ? T(2) * (wsum * xp[k] - g[k]) : T(0);
      if (in) {
#pragma unroll
        for (int k = 0; k <= MP; ++k) a.acc[size_t(k) * n + p] = T(0);
      }
    } else {                      // gacc [n][m]: the sums are the gradient (sign pattern applied by the pair kernel)
      load_padded<T, MP>(a.acc, p, m, g);
      if (in) {
#pragma unroll
        for (int k = 0; k < MP; ++k)
          if (k < m) a.acc[p * m + k] = T(0);
      }
    }
    if (in) store_row<T, MP>(a.grad, p, m, g);
  } else {
    load_padded<T, MP>(a.grad, p, m, g);
  }
  R::rgrad(xp, g, r);
  if constexpr (RULE == VRULE_RSGD) {             // rsgd.py:63-68, 82
    T scale = -a.lr;
    if (a.max_grad_norm > T(0)) scale *= N::min(a.max_grad_norm / R::norm(r), T(1));
#pragma unroll
    for (int k = 0; k < MP; ++k) r[k] *= scale;
    R::step(xp, r, a.exact, o);
  } else if constexpr (RULE == VRULE_MOMENTUM) {  // rsgd.py:70-80
    T b[MP];
    load_padded<T, MP>(a.state0, p, m, b);
    const T clip = a.max_grad_norm > T(0) ? N::min(a.max_grad_norm / R::norm(r), T(1)) : T(1);
#pragma unroll
    for (int k = 0; k < MP; ++k) {
      b[k] = N::fma(a.momentum, b[k], (T(1) - a.dampening) * (r[k] * clip));
      r[k] = -a.lr * b[k];
    }
    R::step(xp, r, a.exact, o);
    R::transport(xp, o, b);
    if (in) store_row<T, MP>(a.state0, p, m, b);
  } else {                                        // radam.py:62-98
    T mo[MP];
    load_padded<T, MP>(a.state0, p, m, mo);
    const T nrm = R::norm(r);
    const T clip = a.adam.max_grad_norm > T(0) ? N::min(a.adam.max_grad_norm / nrm, T(1)) : T(1);
    const T v = N::fma(beta2, a.state1[p * m], (T(1) - beta2) * nrm * nrm);
    const T f = -alpha / (N::sqrt(v) + a.adam.eps);
#pragma unroll
    for (int k = 0; k < MP; ++k) {
      mo[k] = N::fma(a.adam.beta1, mo[k], (T(1) - a.adam.beta1) * (r[k] * clip));
      r[k] = mo[k] * f;
    }
    R::step(xp, r, a.adam.exact, o);
    R::transport(xp, o, mo);
    if (in) {
      store_row<T, MP>(a.state0, p, m, mo);
#pragma unroll
      for (int k = 0; k < MP; ++k)
        if (k < m) a.state1[p * m + k] = v;
    }
  }
  if (in) {
    store_row<T, MP>(a.x, p, m, o);
    if (a.xpad) {
#pragma unroll
      for (int k = 0; k < MP; ++k) a.xpad[p * MP + k] = o[k];   // (the padding of o is zero)
    }
  }
  if constexpr (RULE == VRULE_ADAM) adam_tick(a.adam.step, a.adam.ticket, gridDim.x);
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
  if constexpr (KIND == MM_EUCLIDEAN && sizeof(T) == 4) {   // (the symmetric pair kernel's fp32 range; Lorentz / sphere stop at 16)
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
  VecStep<T> a{static_cast<T*>(p.x), int(n), m,
               with_objective ? ws.acc : static_cast<T*>(nullptr), static_cast<T*>(p.grad),
               with_objective ? ws.slots : static_cast<T*>(nullptr), static_cast<const T*>(q.x), static_cast<T*>(s->loss_out),
               fuse_scale ? static_cast<T*>(q.x) : static_cast<T*>(nullptr), T(q.lr), T(q.max_grad_norm),
               with_objective ? ws.xpad : static_cast<T*>(nullptr),
               T(p.lr), T(p.momentum), T(p.dampening), T(p.max_grad_norm), p.exact,
               static_cast<T*>(p.state0), static_cast<T*>(p.state1),
               AdamArgs<T>{T(p.lr), T(p.beta1), T(p.beta2), T(p.adam_eps), T(p.max_grad_norm), p.nc, p.exact, p.step, p.ticket}};
  const int rule = p.optimizer == MM_OPT_RADAM ? VRULE_ADAM : (p.momentum != 0.0 ? VRULE_MOMENTUM : VRULE_RSGD);
  int rc;
  if (p.kind == MM_EUCLIDEAN) rc = launch_pad<T, MM_EUCLIDEAN>(a, rule, st);
  else if (p.kind == MM_LORENTZ) rc = launch_pad<T, MM_LORENTZ>(a, rule, st);
  else rc = launch_pad<T, MM_SPHERE>(a, rule, st);
  if (rc == MM_OK) *scale_stepped = fuse_scale;
  return rc;
}

}  // namespace

bool vec_fused_step_supports(int dtype, int kind, int m) {
  static const bool off = [] { const char* e = std::getenv("MM_VEC_STEP_UNFUSED"); return e && e[0] == '1'; }();
  static const bool gram = [] { const char* e = std::getenv("MM_VEC_LOSS_GRAM"); return e && e[0] == '1'; }();
  if (off || gram || !vec_sym_supports(dtype, m)) return false;
  if (kind == MM_EUCLIDEAN) return true;
  return (kind == MM_LORENTZ || kind == MM_SPHERE) && m <= 16;   // (fp32 17 <= m <= 32: the matrix-core objective)
}

bool vec_step_fusable(const mm_train_step* s) {
  const mm_step_param& p = s->points[0];
  if (p.kind == MM_FACTOR_SPD || !p.x || !p.grad || p.count != s->n || s->n < 1 || s->n > kSpdMaxNodes || !s->ws) return false;
  if (!vec_fused_step_supports(s->dtype, p.kind, p.dim)) return false;
  if (p.optimizer == MM_OPT_RSGD) return p.momentum == 0.0 || p.state0;
  if (p.optimizer == MM_OPT_RADAM) return p.state0 && p.state1 && p.step && p.ticket;
  return false;
}

int vec_fused_train_step(const mm_train_step* s, int64_t rb, int64_t re, bool with_objective, hipStream_t st, bool* scale_stepped) {
  if (s->dtype == MM_F32) return vec_fused_step_t<float>(s, rb, re, with_objective, st, scale_stepped);
  return vec_fused_step_t<double>(s, rb, re, with_objective, st, scale_stepped);
}

}  // namespace mm
