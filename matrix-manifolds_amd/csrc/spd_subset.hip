// Node minibatches of a single SPD(d) factor INSIDE the pair kernel, every d the library is built for.
//
// The reference trains on node minibatches (graphembed/graphembed/train.py:198-222, batch_size = 512 in the paper grid,
// experiments/run_grid.py:131): ManifoldEmbedding.compute_dists(idx) gathers x[idx] (modules.py:86), the dataset gathers
// dense[idx][:, idx] (data/dataset.py:19-27) and autograd's index backward scatters the gradient rows back.  Here the index
// vector goes into the pair kernel (spd_pair.hpp, spd_pdist_bwd_kernel<..., SUB = true>): table rows, targets and
// accumulator slots are addressed through it, and the per-node kernels over the FULL embedding produce the dense gradient.
// A translation unit of its own: the SUB instantiations compile next to spd.hip / spd_loss.hip.
#include "spd_pair.hpp"

namespace mm {

int spd_fused_train_step_subset(const mm_train_step* s, int64_t rb, int64_t re, hipStream_t st, bool* scale_stepped) {
  const int d = s->points[0].dim;
  if (s->dtype != MM_F32 && s->dtype != MM_F64) return MM_ERR_ARG;
#define MM_FUSED_D(T_, d_)                                                                                   \
  switch (d_) {                                                                                              \
    case 2: return spd_fused_train_step_t<T_, 2, true>(s, rb, re, true, st, scale_stepped);                  \
    case 3: return spd_fused_train_step_t<T_, 3, true>(s, rb, re, true, st, scale_stepped);                  \
    case 4: return spd_fused_train_step_t<T_, 4, true>(s, rb, re, true, st, scale_stepped);                  \
    case 5: return spd_fused_train_step_t<T_, 5, true>(s, rb, re, true, st, scale_stepped);                  \
    default: return MM_ERR_UNSUPPORTED;                                                                      \
  }
  if (s->dtype == MM_F32) { MM_FUSED_D(float, d) }
  MM_FUSED_D(double, d)
#undef MM_FUSED_D
}

}  // namespace mm

using namespace mm;

extern "C" {

int mm_spd_pdist_loss_subset(int dtype, int loss_kind, const void* x, const void* dense, const void* scale_raw, int64_t n_total,
                             int d, const int64_t* idx, int64_t bs, int64_t row_begin, int64_t row_end, double alpha, double eps,
                             int terms, const double* loss_params, double wmin, double wmax, void* loss_out, void* grad_x, void* ws,
                             int flags, mm_stream_t stream) {
  if (!x || !ws || !grad_x || !loss_out || n_total < 0 || n_total > kSpdMaxNodes || bs < 0 || bs > n_total || row_begin < 0 ||
      row_end > bs || row_begin > row_end)
    return MM_ERR_ARG;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  if ((!dense || !idx) && mm_pair_offset(bs, row_end) > mm_pair_offset(bs, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_total == 0) {
    const size_t es = dtype == MM_F64 ? 8 : 4;
    hipError_t e = hipMemsetAsync(loss_out, 0, 2 * es, st);
    return e == hipSuccess ? MM_OK : int(e);
  }
  MM_DISPATCH(dtype, d,
              (spd_pdist_loss_subset_t<T, D>(loss_kind, static_cast<const T*>(x), static_cast<const T*>(dense),
                                             static_cast<const T*>(scale_raw), n_total, idx, bs, row_begin, row_end, alpha, eps, terms,
                                             loss_params, wmin, wmax, static_cast<T*>(loss_out), static_cast<T*>(grad_x), ws, flags, st)));
}

}  // extern "C"
