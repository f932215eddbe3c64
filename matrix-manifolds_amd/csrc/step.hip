// One training step per call: the objective kernel of the embedding and the optimizer kernels of its parameters,
// issued back to back from C++ (include/mm_manifolds.h, mm_train_step_run).  The reference's loop body
// (graphembed/graphembed/train.py:198-222) is ~60 framework launches driven from Python; captured in a HIP graph
// (graphembed/graphed.py) this package replays it as one launch, and for callers that cannot capture — the
// reference's own eager loop — this entry point removes the ~15 foreign-function calls, tensor wrappers and
// autograd nodes a step otherwise costs on the host (130-400 us of Python for 100 us of kernels).
#include <cstddef>
#include <cstring>
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "spd_step.hpp"
#include "vec_step.hpp"
#include "product_step.hpp"

namespace {

inline size_t esize(int dtype) { return dtype == MM_F64 ? 8 : 4; }

int optimizer_step(int dtype, const mm_step_param& p, const void* grad, mm_stream_t st) {
  if (!p.x || !grad || p.count <= 0) return p.count == 0 ? MM_OK : MM_ERR_ARG;
  const bool spd = p.kind == MM_FACTOR_SPD;
  if (p.optimizer == MM_OPT_RSGD) {
    if (p.momentum == 0.0)
      return spd ? mm_spd_rsgd_step(dtype, p.x, grad, p.count, p.dim, p.lr, p.max_grad_norm, p.exact, p.x, st)
                 : mm_vec_rsgd_step(dtype, p.kind, p.x, grad, p.count, p.dim, p.lr, p.max_grad_norm, p.exact, p.x, st);
    if (!p.state0) return MM_ERR_ARG;
    return spd ? mm_spd_rsgd_momentum_step(dtype, p.x, grad, p.state0, p.count, p.dim, p.lr, p.momentum, p.dampening,
                                           p.max_grad_norm, p.exact, p.x, st)
               : mm_vec_rsgd_momentum_step(dtype, p.kind, p.x, grad, p.state0, p.count, p.dim, p.lr, p.momentum,
                                           p.dampening, p.max_grad_norm, p.exact, p.x, st);
  }
  if (p.optimizer == MM_OPT_RADAM) {
    if (!p.state0 || !p.state1 || !p.step || !p.ticket) return MM_ERR_ARG;
    return spd ? mm_spd_radam_step(dtype, p.x, grad, p.state0, p.state1, p.step, p.ticket, p.count, p.dim, p.lr, p.beta1,
                                   p.beta2, p.nc, p.adam_eps, p.max_grad_norm, p.exact, p.x, st)
               : mm_vec_radam_step(dtype, p.kind, p.x, grad, p.state0, p.state1, p.step, p.ticket, p.count, p.dim, p.lr,
                                   p.beta1, p.beta2, p.nc, p.adam_eps, p.max_grad_norm, p.exact, p.x, st);
  }
  return MM_ERR_ARG;
}

// parameters that can share one multi-parameter launch: the same optimizer type (each brings its own hyper-parameters);
// there is no multi-parameter heavy-ball kernel
bool same_rule(const mm_step_param& a, const mm_step_param& b) {
  return a.optimizer == b.optimizer && (a.optimizer == MM_OPT_RADAM || (a.momentum == 0.0 && b.momentum == 0.0));
}

// all vector-space parameters of `ps` (with gradients gs) that share a rule go out in one launch
int vector_group_step(int dtype, const mm_step_param* const* ps, const void* const* gs, int count, mm_stream_t st) {
  if (count == 1) return optimizer_step(dtype, *ps[0], gs[0], st);
  mm::VecGroupParam g[8];
  for (int t = 0; t < count; ++t) {
    const mm_step_param& p = *ps[t];
    if (!p.x || !gs[t] || p.count < 0 || p.dim < 1 || p.kind < MM_EUCLIDEAN || p.kind > MM_SPHERE) return MM_ERR_ARG;
    if (p.dim > mm_vec_max_dim()) return MM_ERR_UNSUPPORTED;
    if (p.optimizer == MM_OPT_RADAM && (!p.state0 || !p.state1 || !p.step || !p.ticket)) return MM_ERR_ARG;
    g[t] = mm::VecGroupParam{p.kind, p.dim, p.count, p.x, gs[t], p.x, p.state0, p.state1, p.step, p.ticket,
                             p.lr, p.max_grad_norm, p.beta1, p.beta2, p.adam_eps, p.nc, p.exact};
  }
  return mm::vec_group_step(dtype, ps[0]->optimizer, count, g, static_cast<hipStream_t>(st));
}

// [p, p + bytes) lies inside [base, base + size)
bool inside(const void* p, size_t bytes, const void* base, size_t size) {
  const char* a = static_cast<const char*>(p);
  const char* b = static_cast<const char*>(base);
  return a >= b && a + bytes <= b + size;
}

}  // namespace

static int train_step_run(const mm_train_step* s, mm_stream_t st);

// The struct is versioned by its size (struct_size, first member, ABI 4): the caller's bytes are copied into a zeroed struct
// of THIS library's layout — fields the caller's header did not have yet read as zero (= feature off), and a longer struct of a
// newer header is accepted as long as the tail this library does not know is all zero.
extern "C" int mm_train_step_run(const mm_train_step* step, mm_stream_t st) {
  if (!step) return MM_ERR_ARG;
  constexpr size_t kBase = offsetof(mm_train_step, batch) + sizeof(step->batch);   // the layout ABI 4 started with
  static_assert(kBase <= sizeof(mm_train_step), "fields are appended behind `batch`");
  const size_t have = step->struct_size;
  // (an ABI <= 3 caller has `dtype` and `loss_kind` where struct_size is: 0 ... 2^33, never a size in this window)
  if (have < kBase || have > 16 * sizeof(mm_train_step)) return MM_ERR_ARG;
  mm_train_step local;
  std::memset(&local, 0, sizeof local);
  std::memcpy(&local, step, have < sizeof local ? have : sizeof local);
  if (have > sizeof local) {
    const unsigned char* tail = reinterpret_cast<const unsigned char*>(step) + sizeof local;
    for (size_t k = 0; k < have - sizeof local; ++k)
      if (tail[k]) return MM_ERR_UNSUPPORTED;
  }
  local.struct_size = sizeof local;
  return train_step_run(&local, st);
}

static int train_step_run(const mm_train_step* s, mm_stream_t st) {
  if (!s || s->nf < 1 || s->nf > 4 || !s->loss_out || s->n < 0) return MM_ERR_ARG;
  if (s->dtype != MM_F32 && s->dtype != MM_F64) return MM_ERR_ARG;
  const int nf = s->nf;
  // a node minibatch (train.py:198-222 with batch_size set): the pair list is that of the batch's nodes
  const bool batched = s->batch_idx != nullptr;
  if (batched && (s->batch < 0 || s->batch > s->n)) return MM_ERR_ARG;
  if (!batched && s->batch != 0) return MM_ERR_ARG;   // (a batch size without an index vector: not silently a full batch)
  if (batched && nf != 1) return MM_ERR_UNSUPPORTED;   // (products: mm_product_pairs_loss_subset + the optimizer entry points)
  const int64_t np = batched ? s->batch : s->n;        // points of the pair list
  // this rank's rows of the pair list (all of them on one GPU)
  const int64_t rb = s->row_begin, re = s->row_end <= 0 ? np : s->row_end;
  if (rb < 0 || re > np || rb > re) return MM_ERR_ARG;
  if (!s->target && mm_pair_offset(np, re) > mm_pair_offset(np, rb)) return MM_ERR_ARG;
  if (s->comm) {
    // the message of the collective is ONE buffer holding every gradient and the loss record
    if (!s->reduce_buf || s->reduce_count <= 0) return MM_ERR_ARG;
    const size_t size = size_t(s->reduce_count) * esize(s->dtype);
    if (!inside(s->loss_out, size_t(1 + nf) * esize(s->dtype), s->reduce_buf, size)) return MM_ERR_ARG;
    for (int k = 0; k < nf; ++k) {
      const mm_step_param& p = s->points[k];
      const size_t pt = p.kind == MM_FACTOR_SPD ? size_t(p.dim) * p.dim : size_t(p.dim);
      if (!p.grad || !inside(p.grad, size_t(p.count) * pt * esize(s->dtype), s->reduce_buf, size)) return MM_ERR_ARG;
    }
  }
  int rc;
  // ---- a single SPD factor takes the fused form: pair kernel -> ONE kernel for finalize + optimizer rule + the tables of
  // the new points (sharded: pair kernel -> finalize -> all-reduce -> optimizer rule + tables)
  const bool spd_fused = nf == 1 && mm::spd_step_fusable(s);
  // ---- ... and so does a single vector factor where the symmetric VALU pair kernel is the objective (vec_step.hpp)
  const bool vec_fused = nf == 1 && !spd_fused && !batched && mm::vec_step_fusable(s);
  // ---- ... and a product embedding on one GPU: the mixed-manifold pair kernel + ONE kernel for everything else
  const bool product_fused = nf > 1 && !s->comm && mm::product_step_fusable(s);
  bool points_done = false;
  bool scale_done[4] = {false, false, false, false};
  if (spd_fused && !s->comm) {
    rc = batched ? mm::spd_fused_train_step_subset(s, rb, re, static_cast<hipStream_t>(st), &scale_done[0])
                 : mm::spd_fused_train_step(s, rb, re, true, static_cast<hipStream_t>(st), &scale_done[0]);
    if (rc != MM_OK) return rc;
    points_done = true;
  } else if (vec_fused && !s->comm) {
    rc = mm::vec_fused_train_step(s, rb, re, true, static_cast<hipStream_t>(st), &scale_done[0]);
    if (rc != MM_OK) return rc;
    points_done = true;
  } else if (product_fused) {
    for (int k = 0; k < nf; ++k)
      if (!s->scales[k].x) return MM_ERR_ARG;
    rc = mm::product_fused_train_step(s, rb, re, static_cast<hipStream_t>(st), scale_done);
    if (rc != MM_OK) return rc;
    points_done = true;
  } else
  // ---- objective and gradients
  if (nf == 1) {
    const mm_step_param& p = s->points[0];
    if (!p.x || !p.grad) return MM_ERR_ARG;
    if (p.kind == MM_FACTOR_SPD) {
      // MM_WS_PREPARED ("the workspace holds the tables of the current points") is honoured only where this call keeps that
      // promise for the NEXT step: the sharded fused step (its optimizer kernel rewrites the tables, below) or frozen points.
      // On the plain unfused path the per-point optimizer kernels do not touch the tables, so a caller that derives the
      // flag from the dimension alone would otherwise run its second step on stale Cholesky factors.
      const bool tables_kept = spd_fused || p.optimizer == MM_OPT_NONE;
      const int flags = tables_kept ? (s->ws_flags & MM_WS_PREPARED) : 0;
      if (batched)
        rc = mm_spd_pdist_loss_subset(s->dtype, s->loss_kind, p.x, s->target, s->scales[0].x, s->n, p.dim, s->batch_idx, s->batch,
                                      rb, re, s->alpha, s->eps, s->terms, s->loss_params, s->wmin, s->wmax, s->loss_out, p.grad,
                                      s->ws, flags, st);
      else
        rc = mm_spd_pdist_loss(s->dtype, s->loss_kind, p.x, s->target, s->scales[0].x, s->n, p.dim, rb, re, s->alpha, s->eps,
                               s->terms, s->loss_params, s->wmin, s->wmax, s->loss_out, p.grad, s->ws, flags, st);
    } else if (batched) {
      rc = mm_vec_pdist_loss_subset(s->dtype, p.kind, s->loss_kind, p.x, s->target, s->scales[0].x, s->n, p.dim, s->batch_idx,
                                    s->batch, rb, re, s->alpha, s->eps, s->terms, s->loss_params, s->loss_out, p.grad, s->ws, st);
    } else
      rc = mm_vec_pdist_loss(s->dtype, p.kind, s->loss_kind, p.x, s->target, s->scales[0].x, s->n, p.dim, rb, re, s->alpha,
                             s->eps, s->terms, s->loss_params, s->loss_out, p.grad, s->ws, st);
  } else {
    int kinds[4], dims[4];
    const void* xs[4];
    const void* sc[4];
    void* grads[4];
    for (int k = 0; k < nf; ++k) {
      kinds[k] = s->points[k].kind; dims[k] = s->points[k].dim; xs[k] = s->points[k].x; sc[k] = s->scales[k].x;
      grads[k] = s->points[k].grad;
      if (!xs[k] || !grads[k]) return MM_ERR_ARG;
    }
    // (MM_WS_PREPARED is the fused step's promise about the symmetric pair kernel's node table: not kept on this path)
    rc = mm_product_pairs_loss(s->dtype, s->loss_kind, nf, kinds, dims, xs, sc, s->target, s->n, rb, re, s->alpha, s->eps,
                               s->terms, s->loss_params, s->wmin, s->wmax, grads, s->loss_out, s->ws,
                               s->ws_flags & ~MM_WS_PREPARED, st);
  }
  if (rc != MM_OK) return rc;
  // ---- the one collective of a sharded step: {gradients, loss, scale gradients} summed over the ranks, in place, on the
  // same stream (train.py:107-109 is a broadcast + gather + reduce-add per step in the reference)
  if (s->comm) {
    rc = mm_allreduce_sum(s->comm, s->dtype, s->reduce_buf, s->reduce_count, st);
    if (rc != MM_OK) return rc;
  }
  // ---- optimizer: SPD points one launch each; vector-space parameters (points and scales) grouped by rule
  const mm_step_param* vec[8];
  const void* vgrad[8];
  int nv = 0;
  if (spd_fused && s->comm) {
    rc = mm::spd_fused_train_step(s, rb, re, false, static_cast<hipStream_t>(st), &scale_done[0]);
    if (rc != MM_OK) return rc;
    points_done = true;
  } else if (vec_fused && s->comm) {
    rc = mm::vec_fused_train_step(s, rb, re, false, static_cast<hipStream_t>(st), &scale_done[0]);
    if (rc != MM_OK) return rc;
    points_done = true;
  }
  for (int k = 0; k < nf && !points_done; ++k) {
    const mm_step_param& p = s->points[k];
    if (p.optimizer == MM_OPT_NONE) continue;   // frozen
    if (p.kind == MM_FACTOR_SPD) {
      rc = optimizer_step(s->dtype, p, p.grad, st);
      if (rc != MM_OK) return rc;
    } else {
      vec[nv] = &p; vgrad[nv++] = p.grad;
    }
  }
  for (int k = 0; k < nf; ++k) {
    const mm_step_param& q = s->scales[k];
    if (!q.x || q.optimizer == MM_OPT_NONE) continue;   // a factor without a scale / a frozen one (burn-in): read, not stepped
    if (scale_done[k]) continue;                        // updated by the fused step kernel
    vec[nv] = &q;
    vgrad[nv++] = static_cast<const char*>(s->loss_out) + size_t(1 + k) * esize(s->dtype);
  }
  bool done[8] = {false, false, false, false, false, false, false, false};
  const int most = mm_vec_rsgd_multi_max();
  for (int a = 0; a < nv; ++a) {
    if (done[a]) continue;
    const mm_step_param* grp[8];
    const void* gg[8];
    int cnt = 0;
    for (int b = a; b < nv && cnt < most; ++b)
      if (!done[b] && (b == a || same_rule(*vec[a], *vec[b]))) {
        grp[cnt] = vec[b]; gg[cnt++] = vgrad[b]; done[b] = true;
      }
    rc = vector_group_step(s->dtype, grp, gg, cnt, st);
    if (rc != MM_OK) return rc;
  }
  return MM_OK;
}
