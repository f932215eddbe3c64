/* mm_manifolds.h — C ABI of libmm_manifolds.so (MI355X / gfx950).
 *
 * Drop-in boundary for graphembed's pairwise manifold-distance training path.
 * The reference has no FFI of its own: the interface these entry points
 * replace is the Python plugin class `Manifold`
 * (graphembed/graphembed/manifolds/base.py:7-81) and its callers
 * (optim/rsgd.py:40-82, modules.py:84-88).  Each entry point below cites the
 * reference method whose arithmetic it performs; INTEGRATION.md shows the
 * ctypes binding that plugs them back into that class.
 *
 * Rules of the ABI
 *   - plain C: pointers, sizes, scalars.  No torch / C++ types.
 *   - every `const void*` / `void*` data pointer is DEVICE memory (HBM),
 *     contiguous, row-major, batch-first; element type given by `dtype`.
 *   - the caller owns all buffers, including the workspace `ws`
 *     (size from the matching *_ws_bytes()); nothing is allocated inside.
 *   - kernels are enqueued on `stream` (a hipStream_t) and never synchronise;
 *     no call blocks the host.
 *   - return value: MM_OK, a negative MM_ERR_* for argument errors, or a
 *     positive hipError_t from the launch.  No exceptions cross the ABI.
 *   - pair order everywhere: k <-> (i,j), i<j, row-major upper triangle
 *     (torch.triu_indices(n,n,1); base.py:62, spd.py:179).
 *   - a row range [row_begin,row_end) selects the pairs (i,j) with
 *     row_begin <= i < row_end: a contiguous slice of the pair list starting at
 *     mm_pair_offset(n,row_begin).  This is the multi-GPU sharding unit.
 */
#ifndef MM_MANIFOLDS_H
#define MM_MANIFOLDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mm_stream_t; /* hipStream_t */

enum { MM_F32 = 0, MM_F64 = 1 };
enum { MM_OK = 0, MM_ERR_ARG = -1, MM_ERR_UNSUPPORTED = -2,
       MM_ERR_COMM = -3 /* RCCL missing or an RCCL call failed: text in mm_comm_last_error() */ };

/* vector-manifold kinds for mm_vec_* */
enum { MM_EUCLIDEAN = 0, MM_LORENTZ = 1, MM_SPHERE = 2 };

/* ---- library info -------------------------------------------------------- */
int mm_abi_version(void);
const char* mm_target_arch(void); /* "gfx950" */

/* ---- optional live kernel timing (HIP events on the launch stream) ---------- */
enum { MM_PROF_SPD_FWD = 0, MM_PROF_SPD_BWD = 1, MM_PROF_VEC_FWD = 2, MM_PROF_VEC_BWD = 3 };
/* When on, each pair-kernel launch is bracketed by hipEventRecord on its stream. */
int mm_prof_enable(int on);
/* Waits for the recorded events of kernel `id`; returns their count and summed
 * duration since the previous collect. */
int mm_prof_collect(int id, int64_t* launches, double* total_ms);
/* Shader clock at this moment of the stream: one wavefront runs `iters` dependent multiply-adds between two readings of
 * s_memtime (shader cycles) and s_memrealtime (the constant 100 MHz reference); out (device, 2 x uint64) receives the two
 * differences — clock [MHz] = 100 * out[0] / out[1].  Enqueued behind the kernels whose clock regime is to be read
 * (bench.py: behind a burst of replays of the timed step), so that an instruction budget can be priced in time. */
int mm_prof_clock_probe(void* out, int iters, mm_stream_t stream);

/* ---- pair-list geometry (host-side helpers, no GPU work) ----------------- */
/* number of pairs in rows < row:  row*(2n-row-1)/2 */
int64_t mm_pair_offset(int64_t n, int64_t row);
/* contiguous row range of shard `rank` of `world`, balanced by pair count */
int mm_shard_rows(int64_t n, int world, int rank, int64_t* row_begin, int64_t* row_end);

/* ---- SPD(d), affine-invariant metric ------------------------------------- */
/* Largest d the pairwise / per-node SPD kernels are instantiated for. */
int mm_spd_max_dim(void);

/* Workspace bytes for mm_spd_pdist_fwd/bwd on n points.  The workspace is the caller's and may hold anything when it is first
 * handed over (no memset is needed, ever): the per-node tables are written by the preparation, the accumulators are cleared by
 * it and by finalize, and the 64 KB behind them — where the backward's workgroups remember the start of their share of the
 * pair walk from one launch to the next (one 32-byte entry per workgroup: the walk's key and the start it leads to) — are
 * self-validating: an entry is used only if its whole key matches the launch, so stale or uninitialised contents cost a
 * recomputation, never a wrong result.  Keeping ONE workspace per embedding across steps (as the Python layer and bench.py
 * do) is what makes those entries hit; results do not depend on it.  One stream at a time may use a workspace. */
size_t mm_spd_pdist_ws_bytes(int dtype, int64_t n, int d);

/* flags for the pdist calls */
enum { MM_WS_PREPARED = 1 /* ws already holds the per-node factors of this x */ };

/* Fills ws with the per-node factors of x (Cholesky factor, its inverse, log det; the part of
 * _lult, manifolds/spd.py:108-111, that depends on one point only) and clears its accumulators and
 * status word; later pdist calls on the same x may then pass MM_WS_PREPARED. */
int mm_spd_prepare(int dtype, const void* x, int64_t n, int d, void* ws, mm_stream_t stream);

/* SymmetricPositiveDefinite.pdist — manifolds/spd.py:175-181 (+ _norm_log
 * 163-169, _lult 108-111).  out[k] = sum_m log^2 lambda_m(L_i^-1 X_j L_i^-T)
 * (sqrt of it if !squared), eigenvalues value-clamped to [wmin,wmax], result
 * value-clamped >= wmin.  d >= 3 (and the fused objectives of every d): windows narrower than
 * [1e-6, 1e6] return MM_ERR_UNSUPPORTED — the element-wise mm_spd_dist_fwd / _bwd take any window
 * (DESIGN.md section 6, item 11).
 *   x    [n,d,d]   out  [mm_pair_offset(n,row_end)-mm_pair_offset(n,row_begin)]
 * A non-positive-definite x[i] sets the status word (mm_spd_status).
 * n <= 2^22 in every SPD entry point (32-bit byte offsets into the node tables; 2^22 nodes are 8.8e12 pairs):
 * MM_ERR_ARG beyond, as for null pointers and row ranges outside [0, n]. */
int mm_spd_pdist_fwd(int dtype, const void* x, int64_t n, int d, int64_t row_begin,
                     int64_t row_end, int squared, double wmin, double wmax, void* out,
                     void* ws, int flags, mm_stream_t stream);

/* Backward of the above (what autograd computes in the reference; symmetric
 * part — SURVEY.md §8 a5).  g has the layout of `out`.  grad_x [n,d,d] is
 * OVERWRITTEN with this shard's partial gradient (full shape; sum the shards).
 * fp32 on ILL-CONDITIONED points: pairs whose matrix L_i^-1 X_j L_i^-T has lambda_max > 256 lambda_min are solved a second
 * time by a one-sided Jacobi on L_i^-1 L_j (d = 2..4) in EVERY form of the kernel — since round 6 also in the
 * two-columns-per-lane form of the SPD(4) backward that launches of >= 30 M pairs (n >= 7747) and row bands of >= 12 M pairs
 * take (there through an out-of-line call), so the gradient's accuracy does not depend on the launch size or on how a problem
 * is sharded: 8e-5 of its largest entry at cond(X) = 1e4, 1.5e-6 at 1e2 (measured, tools/illcond_probe.py; pinned by
 * tests/test_spd_gpu.py::test_ill_conditioned_points_fp32 in both forms).  MM_SPD4_BWD_TWO_COLS=0 / 1 in the environment
 * forces the one- / two-column form at every size (A/B builds only); fp64 needs no second solve. */
int mm_spd_pdist_bwd(int dtype, const void* x, const void* g, int64_t n, int d,
                     int64_t row_begin, int64_t row_end, int squared, double wmin,
                     double wmax, void* grad_x, void* ws, int flags, mm_stream_t stream);

/* Fused objective + gradients for a single-factor SPD embedding: ONE pass over the pairs
 * evaluates  m_k = softplus(*scale_raw) * d2_k  (ManifoldEmbedding.compute_dists,
 * modules.py:84-88), the loss against target[k] (objectives.py:16-45), and both gradients —
 * what `loss = objective_fn(gdists, compute_dists()); loss.backward()` computes in the reference
 * (train.py:213-217) without ever writing the pair vector of distances (4 B/pair of HBM traffic
 * instead of >= 40 B/pair through the framework's element-wise ops).
 *   loss_kind  MM_LOSS_STRESS    sum (m - target)^2                         objectives.py:39-45
 *              MM_LOSS_QUOTIENT  terms bit0: sum |m/(alpha target) - 1|
 *                                terms bit1: sum |alpha target/(m + eps) - 1|,  eps = 1/(epoch+1)
 *                                                                            objectives.py:16-36
 *   target     pair vector (layout of mm_spd_pdist_fwd's `out`) of squared graph distances
 *   scale_raw  device scalar (the raw scale parameter) or NULL for scale 1
 *   loss_params  NULL, or a device fp64 array {alpha, eps} that overrides the two by-value arguments
 *              (quotient loss): a captured graph of a step then follows eps = 1/(epoch+1) across epochs
 *              by updating that array, without being re-recorded.  Same argument in every *_loss entry.
 *   loss_out   device [2]: { loss of this shard, d loss / d scale_raw of this shard }
 *   grad_x     [n,d,d], OVERWRITTEN with this shard's partial d loss / d x. */
enum { MM_LOSS_NONE = 0, MM_LOSS_STRESS = 1, MM_LOSS_QUOTIENT = 2 };
int mm_spd_pdist_loss(int dtype, int loss_kind, const void* x, const void* target,
                      const void* scale_raw, int64_t n, int d, int64_t row_begin, int64_t row_end,
                      double alpha, double eps, int terms, const double* loss_params, double wmin, double wmax,
                      void* loss_out, void* grad_x, void* ws, int flags, mm_stream_t stream);

/* The same objective for a NODE MINIBATCH, with no gather or scatter launch around it (train.py:198-222 draws
 * idx = randperm(n)[a:b]; ManifoldEmbedding.compute_dists(idx) gathers x[idx], modules.py:86; GraphDataset.__getitem__
 * gathers dense[idx][:, idx], data/dataset.py:19-27; autograd's index backward scatters the gradient rows):
 *   x        the FULL table [n_total, d, d]; the `bs` points of the step are its rows idx[0..bs) (device int64, DISTINCT)
 *   dense    the dataset's dense n_total x n_total matrix of squared graph distances: target of pair (a, b) = dense[idx[a]][idx[b]]
 *   grad_x   [n_total, d, d], OVERWRITTEN: rows idx[.] with the gradient, every other row with zero — the dense
 *            gradient the reference's optimizers see
 *   ws       mm_spd_pdist_ws_bytes(dtype, n_total, d): the per-node tables are those of the FULL embedding (flags =
 *            MM_WS_PREPARED skips their preparation when they are current); row_begin / row_end shard the pair list of
 *            the BATCH (0 .. bs).  Every d of mm_spd_max_dim().
 *   idx      is caller data the kernels cannot validate: they read the low 32 bits of each index and CLAMP the node id into
 *            [0, n_total) — an out-of-range or negative index yields wrong numbers for that batch, never an access outside
 *            the buffers; repeated indices break the "each gradient row is written once" contract silently.  The Python
 *            layer checks host-side index tensors (graphembed.modules.distinct_in_range); device-side ones are the
 *            caller's responsibility.  The same holds for mm_vec_pdist_loss_subset, mm_product_pairs_loss_subset,
 *            mm_pair_gather and mm_train_step.batch_idx. */
int mm_spd_pdist_loss_subset(int dtype, int loss_kind, const void* x, const void* dense, const void* scale_raw,
                             int64_t n_total, int d, const int64_t* idx, int64_t bs, int64_t row_begin, int64_t row_end,
                             double alpha, double eps, int terms, const double* loss_params, double wmin, double wmax,
                             void* loss_out, void* grad_x, void* ws, int flags, mm_stream_t stream);

/* Stein divergence S(X,Y) = log det((X+Y)/2) - (log det X + log det Y)/2 — the second SPD "distance" of the
 * reference (SymmetricPositiveDefinite(use_stein_div=True): spd.py:183-194 stein_div / stein_pdiv,
 * 246-295 PairwiseSteinDivergence, linalg/torch_batch.py:173-197 PLogDet).  Value clamped >= wmin
 * (gradient-transparent), sqrt of it if !squared.  Layouts, workspace (mm_spd_pdist_ws_bytes), row ranges and
 * flags as mm_spd_pdist_fwd / mm_spd_pdist_bwd; mm_spd_stein_div is the element-wise form over m pairs
 * (out and/or both gradients may be requested; g is the upstream gradient of out). */
int mm_spd_stein_pdiv_fwd(int dtype, const void* x, int64_t n, int d, int64_t row_begin,
                          int64_t row_end, int squared, double wmin, void* out, void* ws, int flags,
                          mm_stream_t stream);
int mm_spd_stein_pdiv_bwd(int dtype, const void* x, const void* g, int64_t n, int d,
                          int64_t row_begin, int64_t row_end, int squared, double wmin,
                          void* grad_x, void* ws, int flags, mm_stream_t stream);
int mm_spd_stein_div(int dtype, const void* x, const void* y, const void* g, int64_t m, int d,
                     int squared, double wmin, void* out, void* grad_x, void* grad_y,
                     mm_stream_t stream);

/* Counts the points whose Cholesky factorisation failed in the last prepare of `ws`
 * (n = the point count it was prepared for) into *host_status (0 = all succeeded).
 * Synchronises `stream` — the only blocking call of the ABI. */
int mm_spd_status(void* ws, int64_t n, int* host_status, mm_stream_t stream);

/* SymmetricPositiveDefinite.dist — spd.py:171-173, element-wise over m pairs
 * (x[k], y[k]).  x,y [m,d,d]; out [m]. */
int mm_spd_dist_fwd(int dtype, const void* x, const void* y, int64_t m, int d, int squared,
                    double wmin, double wmax, void* out, mm_stream_t stream);
int mm_spd_dist_bwd(int dtype, const void* x, const void* y, const void* g, int64_t m, int d,
                    int squared, double wmin, double wmax, void* grad_x, void* grad_y,
                    mm_stream_t stream);

/* Per-point maps used by RiemannianSGD (optim/rsgd.py:56-82); all [m,d,d]. */
enum {
  MM_SPD_EGRAD2RGRAD = 0, /* X sym(U) X                         spd.py:134-135 */
  MM_SPD_EXP = 1,         /* L expm(L^-1 U L^-T) L^T            spd.py:137-144 */
  MM_SPD_RETR = 2,        /* sym(X + U + 1/2 U X^-1 U)          spd.py:146-154 */
  MM_SPD_LOG = 3,         /* L logm(L^-1 Y L^-T) L^T  (U = Y)   spd.py:156-161 */
  MM_SPD_PROJX = 4,       /* V clamp(w) V^T of sym(X) (U unused) spd.py:126-132 */
  MM_SPD_PROJU = 5        /* sym(U)                             spd.py:119-124 */
};
int mm_spd_map(int dtype, int op, const void* x, const void* u, int64_t m, int d, double wmin,
               double wmax, void* out, mm_stream_t stream);
/* SymmetricPositiveDefinite.symeig — spd.py:35-41, 63-64 (linalg/fast.py:53-91 for n = 2, 3; LAPACK on the CPU
 * otherwise): eigenvalues of sym(x[k]), ascending.   x [m,d,d] -> w [m,d] */
int mm_spd_eigvalsh(int dtype, const void* x, int64_t m, int d, void* w, mm_stream_t stream);
/* graphembed/linalg/fast.py:25-159 by name — the closed forms for stacks of 2x2 / 3x3 matrices (spd.py:35-46 and
 * grassmann.py:29 pick them; monitor.py:39-45, tests/test_linalg.py:47-141 and tests/test_perf.py:14-81 call them directly),
 * with the derivative torch's autograd gives the reference's arithmetic (upper-triangle gradients, fast.py:5-10; the
 * `.data.clamp_` guards clamp the value and pass the derivative).  x [n,2,2] or [n,3,3]:
 *   op                      out              out2 (optional)     reference
 *   MM_FAST_SYMEIG2         w [n,2] asc.                         fast.py:53-70
 *   MM_FAST_SYMEIG3         w [n,3] asc.                         fast.py:75-91
 *   MM_FAST_CHOLESKY2       L [n,2,2]                            fast.py:94-107
 *   MM_FAST_INVCHOLESKY2    L^-1 [n,2,2]     L [n,2,2] or NULL   fast.py:110-135
 *   MM_FAST_SINGULAR2       s [n,2] desc.                        fast.py:138-159
 *   MM_FAST_DET2 / DET3 / SYMDET3   det [n]                      fast.py:25-50
 * mm_fast_bwd: grad_x [n,k,k] from the cotangents of out (and of out2, or NULL).  `eps` is the functions' `eps` argument. */
enum { MM_FAST_SYMEIG2 = 0, MM_FAST_SYMEIG3 = 1, MM_FAST_CHOLESKY2 = 2, MM_FAST_INVCHOLESKY2 = 3, MM_FAST_SINGULAR2 = 4,
       MM_FAST_DET2 = 5, MM_FAST_DET3 = 6, MM_FAST_SYMDET3 = 7 };
int mm_fast_fwd(int op, int dtype, const void* x, int64_t n, double eps, void* out, void* out2, mm_stream_t stream);
int mm_fast_bwd(int op, int dtype, const void* x, const void* grad_out, const void* grad_out2, int64_t n, double eps,
                void* grad_x, mm_stream_t stream);
/* SymmetricPositiveDefinite.norm — spd.py:113-117: ||L^-1 U L^-T||_F  -> out [m] */
int mm_spd_norm(int dtype, const void* x, const void* u, int64_t m, int d, int squared, void* out,
                mm_stream_t stream);
/* One fused RiemannianSGD update without momentum (rsgd.py:63-68, 82):
 *   r = X sym(G) X; r *= min(max_grad_norm/||r||_X, 1) (skipped if
 *   max_grad_norm <= 0); x_new = (exact ? exp : retr)(X, -lr r). */
int mm_spd_rsgd_step(int dtype, const void* x, const void* egrad, int64_t m, int d, double lr,
                     double max_grad_norm, int exact, void* x_new, mm_stream_t stream);

/* ---- vector manifolds: Euclidean R^m, Lorentz H^{m-1}, sphere S^{m-1} -------- */
/* `kind` is MM_EUCLIDEAN / MM_LORENTZ / MM_SPHERE; points are rows of x [n,m]. */
int mm_vec_max_dim(void);
/* (accumulators, loss slots, zero-padded points and — as for SPD — 64 KB of self-validating workgroup starts of the backward's walk) */
size_t mm_vec_pdist_ws_bytes(int dtype, int64_t n, int m);

/* Manifold.pdist default = gather + dist — manifolds/base.py:59-63 with
 *   Euclidean: max(sum (y-x)^2, 1e-8) [sqrt]                 base.py:29-33,56-57; euclidean.py:35-48
 *   Lorentz  : max(acosh(max(-<x,y>_L,1)),1e-8)^2            lorentz.py:72-77,101-138
 *   Sphere   : max(acos(clamp <x,y>),1e-8)^2                 sphere.py:68-74
 * out has the row-range layout described at the top. */
int mm_vec_pdist_fwd(int dtype, int kind, const void* x, int64_t n, int m, int64_t row_begin,
                     int64_t row_end, int squared, void* out, mm_stream_t stream);
/* Same, forward only, with the Gram matrix X J X^T formed on the matrix cores
 * (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64) and the distance map fused
 * into the accumulator epilogue.  Lorentz and sphere (the reference evaluates
 * both from inner products); Euclidean uses the difference form above. */
int mm_vec_pdist_fwd_gram(int dtype, int kind, const void* x, int64_t n, int m, int64_t row_begin,
                          int64_t row_end, int squared, void* out, mm_stream_t stream);
/* Backward: grad_x [n,m] OVERWRITTEN with this shard's partial gradient.
 * (Sphere: the reference's 1/sqrt(1-c^2) is floored at 1e-8 instead of inf.) */
int mm_vec_pdist_bwd(int dtype, int kind, const void* x, const void* g, int64_t n, int m,
                     int64_t row_begin, int64_t row_end, int squared, void* grad_x, void* ws,
                     mm_stream_t stream);

/* Backward of mm_vec_pdist_fwd(_gram) on the matrix cores (fp32; Lorentz, sphere, and the SQUARED
 * Euclidean distance): W^T X with W = g * dout/dq(Gram) formed tile by tile in the MFMA accumulators
 * (csrc/vec_gram.hip; Euclidean: W = g, no Gram).  Same result as mm_vec_pdist_bwd; needs no
 * workspace.  MM_ERR_UNSUPPORTED for other dtypes / kinds / sizes (n > 32768, m > 32). */
int mm_vec_pdist_bwd_gram(int dtype, int kind, const void* x, const void* g, int64_t n, int m,
                          int64_t row_begin, int64_t row_end, int squared, void* grad_x,
                          mm_stream_t stream);

/* Fused objective + gradients for a single-factor vector-manifold embedding — the counterpart of
 * mm_spd_pdist_loss (same loss kinds, arguments and outputs; squared distances). */
int mm_vec_pdist_loss(int dtype, int kind, int loss_kind, const void* x, const void* target,
                      const void* scale_raw, int64_t n, int m, int64_t row_begin, int64_t row_end,
                      double alpha, double eps, int terms, const double* loss_params, void* loss_out, void* grad_x, void* ws,
                      mm_stream_t stream);
/* ... for a NODE MINIBATCH, the counterpart of mm_spd_pdist_loss_subset (same arguments: the FULL table x [n_total, m], the dense
 * target matrix, idx int64[bs] distinct, grad_x [n_total, m] OVERWRITTEN — zero rows outside the batch; ws as
 * mm_vec_pdist_ws_bytes(dtype, n_total, m)); every kind, every m <= mm_vec_max_dim(). */
int mm_vec_pdist_loss_subset(int dtype, int kind, int loss_kind, const void* x, const void* dense, const void* scale_raw,
                             int64_t n_total, int m, const int64_t* idx, int64_t bs, int64_t row_begin, int64_t row_end,
                             double alpha, double eps, int terms, const double* loss_params, void* loss_out, void* grad_x,
                             void* ws, mm_stream_t stream);
/* Element-wise dist over cnt pairs (x[k],y[k]).  out may be NULL (backward only);
 * grad_x/grad_y may both be NULL (forward only), else g [cnt] is required. */
int mm_vec_dist(int dtype, int kind, const void* x, const void* y, const void* g, int64_t cnt, int m,
                int squared, void* out, void* grad_x, void* grad_y, mm_stream_t stream);

/* Per-point maps (optim/rsgd.py:56-82 call sites); x,u,y,out are [cnt,m]. */
enum {
  MM_VEC_EGRAD2RGRAD = 0, /* lorentz.py:52-57, sphere.py:41-44, identity for Euclidean      */
  MM_VEC_PROJU = 1,       /* lorentz.py:39-42, sphere.py:41-44                              */
  MM_VEC_EXP = 2,         /* lorentz.py:59-62, sphere.py:51-56, x+u                         */
  MM_VEC_RETR = 3,        /* = exp for Lorentz/Euclidean (base.py:49-50); sphere.py:58-59   */
  MM_VEC_PROJX = 4,       /* lorentz.py:44-50, sphere.py:46-49 (u unused)                   */
  MM_VEC_TRANSP = 5,      /* u from x to y: lorentz.py:79-82; proju(y,u) (base.py:65-66)    */
  MM_VEC_LOG = 6          /* log_x(u): lorentz.py:64-70, sphere.py:61-66, u-x               */
};
int mm_vec_map(int dtype, int kind, int op, const void* x, const void* u, const void* y, int64_t cnt,
               int m, void* out, mm_stream_t stream);
/* Manifold.norm — base.py:29-33: sqrt(max(<u,u>, 1e-8)) with the manifold's inner product */
int mm_vec_norm(int dtype, int kind, const void* u, int64_t cnt, int m, int squared, void* out,
                mm_stream_t stream);
/* Fused momentum-free RiemannianSGD update (rsgd.py:63-68,82). */
int mm_vec_rsgd_step(int dtype, int kind, const void* x, const void* egrad, int64_t cnt, int m,
                     double lr, double max_grad_norm, int exact, void* x_new, mm_stream_t stream);
/* The same update for `count` (<= mm_vec_rsgd_multi_max()) parameters of one optimizer group in one
 * launch — the loop over group['params'] of rsgd.py:52-82 for parameters that share lr / max_grad_norm /
 * exact.  kinds, xs, egrads, cnts, ms, x_new are HOST arrays; x_new[t] may equal xs[t]. */
int mm_vec_rsgd_multi_max(void);
int mm_vec_rsgd_step_multi(int dtype, int count, const int* kinds, const void* const* xs,
                           const void* const* egrads, const int64_t* cnts, const int* ms, double lr,
                           double max_grad_norm, int exact, void* const* x_new, mm_stream_t stream);

/* The heavy-ball variant of the RSGD update (rsgd.py:70-80): momentum_buffer = momentum * momentum_buffer +
 * (1 - dampening) * rgrad (clipped), x_new = exp/retr(x, -lr * momentum_buffer), and the buffer is transported
 * to x_new — updated IN PLACE (SPD: identity transport, the buffer is kept symmetric).  x_new may equal x. */
int mm_vec_rsgd_momentum_step(int dtype, int kind, const void* x, const void* egrad, void* momentum_buffer,
                              int64_t cnt, int m, double lr, double momentum, double dampening,
                              double max_grad_norm, int exact, void* x_new, mm_stream_t stream);
int mm_spd_rsgd_momentum_step(int dtype, const void* x, const void* egrad, void* momentum_buffer, int64_t m,
                              int d, double lr, double momentum, double dampening, double max_grad_norm,
                              int exact, void* x_new, mm_stream_t stream);

/* One fused RiemannianAdam update (radam.py:62-98): Riemannian gradient, second moment from its norm
 * BEFORE clipping (one scalar per point, stored broadcast over the point as the reference does), clipping,
 * first moment, step size lr sqrt(1-beta2^t)/(1-beta1^t) (nc != 0: beta2 = 1 - 1/t), exp (exact) or
 * retr, transport of the first moment to the new point.  exp_avg / exp_avg_sq are updated in place;
 * x_new may equal x.  `step` is state['step'] as a DEVICE fp64 scalar (>= 1), advanced by the kernel;
 * `ticket` is a device uint32 that is zero between calls (the last block to finish advances `step`), so a
 * captured graph of a training step keeps counting when replayed. */
int mm_vec_radam_step(int dtype, int kind, const void* x, const void* egrad, void* exp_avg,
                      void* exp_avg_sq, double* step, unsigned* ticket, int64_t cnt, int m, double lr,
                      double beta1, double beta2, int nc, double eps, double max_grad_norm, int exact,
                      void* x_new, mm_stream_t stream);
/* ... for `count` (<= mm_vec_rsgd_multi_max()) vector-space parameters of one group in one launch; every
 * parameter has its own moments, step counter and ticket (HOST arrays of device pointers; cnts[t] >= 1). */
int mm_vec_radam_step_multi(int dtype, int count, const int* kinds, const void* const* xs,
                            const void* const* egrads, void* const* exp_avg, void* const* exp_avg_sq,
                            double* const* steps, unsigned* const* tickets, const int64_t* cnts,
                            const int* ms, double lr, double beta1, double beta2, int nc, double eps,
                            double max_grad_norm, int exact, void* const* x_new, mm_stream_t stream);
int mm_spd_radam_step(int dtype, const void* x, const void* egrad, void* exp_avg, void* exp_avg_sq,
                      double* step, unsigned* ticket, int64_t m, int d, double lr, double beta1,
                      double beta2, int nc, double eps, double max_grad_norm, int exact, void* x_new,
                      mm_stream_t stream);

/* ---- Grassmann Gr(N,p) / Stiefel St(N,p): points are [cnt,N,p], N <= 9, p <= 4 ---- */
enum { MM_GRASSMANN = 0, MM_STIEFEL = 1 };
enum {
  MM_MAT_PROJU = 0,    /* u - x x^T u (grassmann.py:49-53) / u - x sym(x^T u) (stiefel.py:40-45)   */
  MM_MAT_PROJX = 1,    /* Q of QR(x) (grassmann.py:55-61); Stiefel: sign-fixed (stiefel.py:47-57)  */
  MM_MAT_RETR_SVD = 2, /* polar factor U V^T of x+u (grassmann.py:76-80, stiefel.py:66-69)         */
  MM_MAT_RETR_QR = 3,  /* Q of QR(x+u) (grassmann.py:71-74); Stiefel sign-fixed (stiefel.py:62-63) */
  MM_MAT_EXP = 4,      /* x V cos(S) V^T + U sin(S) V^T, u = U S V^T (grassmann.py:63-69)          */
  MM_MAT_LOG = 5       /* log_x(u) (grassmann.py:82-89)                                            */
};
int mm_mat_max_rows(void);
int mm_mat_max_cols(void);
int mm_mat_map(int dtype, int kind, int op, const void* x, const void* u, int64_t cnt, int N, int p,
               void* out, mm_stream_t stream);
/* Grassmann.dist — grassmann.py:91-96: sum_k acos^2 sigma_k(x^T y) [sqrt]; element-wise over
 * cnt pairs; out / (grad_x,grad_y) optional as in mm_vec_dist.  (acos'(1), where the
 * reference yields NaN, is replaced by its finite limit.) */
int mm_grass_dist(int dtype, const void* x, const void* y, const void* g, int64_t cnt, int N, int p,
                  int squared, void* out, void* grad_x, void* grad_y, mm_stream_t stream);
/* Manifold.pdist default (base.py:59-63) with Grassmann.dist, row-range layout as above. */
size_t mm_grass_pdist_ws_bytes(int dtype, int64_t n, int N, int p);
int mm_grass_pdist_fwd(int dtype, const void* x, int64_t n, int N, int p, int64_t row_begin,
                       int64_t row_end, int squared, void* out, mm_stream_t stream);
int mm_grass_pdist_bwd(int dtype, const void* x, const void* g, int64_t n, int N, int p,
                       int64_t row_begin, int64_t row_end, int squared, void* grad_x, void* ws,
                       mm_stream_t stream);

/* ---- product embeddings ---------------------------------------------------- */
/* Objective of a product embedding in one pass over the pair vectors (the element-wise part of
 * train.py:213-217 for several factors): with d2[k] the squared pair distances of factor k,
 *   m = sum_k softplus(*scale_raw[k]) * d2[k]      (modules.py:84-88)
 *   loss = objective(target, m)                    (objectives.py:16-45; kinds as mm_spd_pdist_loss)
 * writes g_out[k] = dloss/dm * softplus(scale_k) (the upstream gradient of factor k's pdist backward)
 * and loss_out = { loss, dloss/dscale_raw[0], ..., dloss/dscale_raw[nf-1] }.
 * d2, scale_raw, g_out are HOST arrays of nf device pointers (nf <= mm_product_max_factors()). */
int mm_product_max_factors(void);
size_t mm_product_loss_ws_bytes(int dtype, int nf);
int mm_product_loss(int dtype, int loss_kind, int nf, const void* const* d2, const void* target,
                    const void* const* scale_raw, int64_t npairs, double alpha, double eps, int terms, const double* loss_params,
                    void* const* g_out, void* loss_out, void* ws, mm_stream_t stream);

/* The same objective in ONE pair kernel over all factors (the csphd configuration: Lorentz x sphere x
 * SPD(2) — one launch computes every factor's squared distance, the weighted sum, the loss term and all
 * gradients; ManifoldEmbedding.compute_dists modules.py:84-88 + objectives.py:16-45 + their backward).
 *   kinds[k]   MM_EUCLIDEAN / MM_LORENTZ / MM_SPHERE (xs[k] = [n, dims[k]], dims[k] <= 16) or
 *              MM_FACTOR_SPD (xs[k] = [n, d, d] with d = dims[k] in {2, 3}; the Cholesky factors are
 *              formed inside the pair kernel — no workspace, no preparation launch)
 *   grads[k]   Euclidean gradient of the loss w.r.t. xs[k] (same shape), rows [row_begin,row_end) of the
 *              pair list only — summed over shards it is the full gradient
 *   loss_out   { loss, dloss/dscale_raw[0..nf-1] }
 *   ws         mm_product_pairs_ws_bytes; every successful call leaves its accumulators zero, so a
 *              workspace that is reused for the same (dtype, factor list, n) may be passed with
 *              flags = MM_WS_CLEAN from the second call on (saves the clearing launches); | MM_WS_PREPARED only when the
 *              workspace's node table (symmetric pair kernel, large n) holds the CURRENT points — mm_train_step_run leaves
 *              it so after a fused product step; a caller that does not know passes neither and the table is rebuilt
 * At most 3 vector factors and one SPD factor; otherwise MM_ERR_UNSUPPORTED (use mm_product_loss around
 * the per-factor kernels).  kinds, dims, xs, scale_raw, grads are HOST arrays. */
enum { MM_FACTOR_SPD = 16 };
enum { MM_WS_CLEAN = 2 /* flag: the workspace's accumulators are already zero */ };
size_t mm_product_pairs_ws_bytes(int dtype, int nf, const int* kinds, const int* dims, int64_t n);
int mm_product_pairs_loss(int dtype, int loss_kind, int nf, const int* kinds, const int* dims,
                          const void* const* xs, const void* const* scale_raw, const void* target,
                          int64_t n, int64_t row_begin, int64_t row_end, double alpha, double eps,
                          int terms, const double* loss_params, double wmin, double wmax, void* const* grads, void* loss_out,
                          void* ws, int flags, mm_stream_t stream);

/* The same for a node minibatch (train.py:198-222, batch_size = 512 in the paper grid) without any gather or
 * scatter launch around it: the `bs` points of the step are rows idx[0..bs) (device int64) of the factors'
 * FULL tables xs[k] = [n_total, ...]; the target of pair (a, b) is dense[idx[a]][idx[b]] (the dense
 * n_total x n_total matrix of GraphDataset, data/dataset.py:19-27); gradients are written to rows idx[.] of
 * the full-size grads[k] — the other rows are NOT touched: pass zero-filled buffers.  row_begin/row_end
 * shard the pair list of the batch; ws as mm_product_pairs_ws_bytes(..., n = bs). */
int mm_product_pairs_loss_subset(int dtype, int loss_kind, int nf, const int* kinds, const int* dims,
                                 const void* const* xs, const void* const* scale_raw, const void* dense,
                                 int64_t n_total, const int64_t* idx, int64_t bs, int64_t row_begin,
                                 int64_t row_end, double alpha, double eps, int terms, const double* loss_params, double wmin,
                                 double wmax, void* const* grads, void* loss_out, void* ws, int flags,
                                 mm_stream_t stream);

/* Targets of a node minibatch: out[pair (a,b), a<b] = dense[idx[a]][idx[b]] in pair-vector order
 * (GraphDataset.__getitem__, data/dataset.py:19-27).  dense [n,n]; idx int64[bs] (device); out [bs(bs-1)/2]. */
int mm_pair_gather(int dtype, const void* dense, int64_t n, const int64_t* idx, int64_t bs, void* out,
                   mm_stream_t stream);

/* ---- graph-reconstruction metric --------------------------------------------- */
/* Per-node average precision of the embedding's neighbour ranking — what
 * FastPrecision::MeanAveragePrecision (pyx/impl/precision.cpp) and py_mean_average_precision
 * (metrics.py:61-96) average into the MAP score.
 *   dist          [n,n] dense symmetric embedding distances (zeros on the diagonal)
 *   indptr/indices  CSR adjacency of the (unweighted) graph, int32, device memory
 *   rank_scratch  int[nnz] device scratch
 *   ap_out        [n]: AP(u) = 1/deg(u) sum_{v in N(u)} (#neighbours ranked <= v) / rank(v); ties by node index. */
int mm_graph_average_precision(int dtype, const void* dist, int64_t n, const int* indptr,
                               const int* indices, int* rank_scratch, void* ap_out,
                               mm_stream_t stream);

/* Layer-wise F1 scores of the embedding's ordering against the shortest-path trees of the graph —
 * FastPrecision::LayerMeanF1Scores / LayerMeanAverageF1Scores (pyx/impl/precision.cpp:300-429), unweighted
 * AND weighted graphs.  order [n,n] int32: row u = all nodes sorted by embedding distance to u (stable: ties by node id);
 * hops [n,n] int32: the LAYER of v in the shortest-path tree rooted at u = the dense rank of the graph distance d(u,v)
 * among the distinct distances from u (precision.cpp:150-166; for unweighted graphs that is the hop distance); indptr CSR row pointers (degrees, for the min/max filter);
 * num_layers = diameter + 1 (<= 2048).  ACCUMULATES into the device arrays m1, m2, counts (double[num_layers-1],
 * zeroed by the caller): per layer the sum of F1, of F1^2 and the number of terms (per_tree_average = 0), or
 * the same over per-tree layer means (per_tree_average = 1). */
int mm_graph_layer_f1(const int* order, const int* hops, int64_t n, const int* indptr, int min_degree,
                      int max_degree, int per_tree_average, int num_layers, double* m1, double* m2,
                      double* counts, mm_stream_t stream);

/* Row-wise stable argsort of the dense embedding-distance matrix: order[u][k] = the node with the k-th smallest
 * distance to u, ties in node order (SortNodeDists, pyx/impl/precision.cpp:107-121) — the input of mm_graph_layer_f1.
 * dist [n,n] (device), order int32 [n,n], ws of mm_graph_sort_rows_ws_bytes bytes; n <= 46340. */
size_t mm_graph_sort_rows_ws_bytes(int dtype, int64_t n);
int mm_graph_sort_rows(int dtype, const void* dist, int64_t n, int* order, void* ws, size_t ws_bytes,
                       mm_stream_t stream);

/* ---- the collective of the sharded path ----------------------------------------- */
/* One process per GPU; the embedding is replicated, the pair list is cut into row ranges (mm_shard_rows) and every
 * step ends with ONE all-reduce(sum) of {point gradients, loss, scale gradients} over a process-lifetime RCCL
 * communicator (xGMI).  Replaces the reference's only parallel call site, torch.nn.DataParallel around
 * BatchedObjective (graphembed/graphembed/train.py:107-109: broadcast + gather + reduce-add per step); SURVEY.md §8b
 * "allreduce_grad(buf)".  RCCL is bound at run time (dlopen; MM_RCCL_LIB overrides the search), so the library loads
 * without it; every entry below then returns MM_ERR_COMM.
 *   rendezvous: rank 0 calls mm_comm_unique_id and hands the MM_COMM_ID_BYTES token to the other ranks through any
 *               host channel (file, socket, MPI, a torch.distributed store); every rank then calls mm_comm_init with
 *               the same token — collective, blocks until all `world` ranks have arrived.  `device` = HIP device
 *               ordinal of this rank (made current).  world = 1 is valid (a one-rank communicator).
 *   mm_allreduce_sum: in place, buf [count] of `dtype` in device memory, enqueued on `stream`, never synchronises —
 *               capturable into a HIP graph together with the kernels around it (after one uncaptured call).
 *   mm_comm_last_error: text of this thread's last MM_ERR_COMM. */
typedef struct mm_comm* mm_comm_t;
enum { MM_COMM_ID_BYTES = 128 };
int mm_comm_available(void);          /* 1 if RCCL could be bound */
int mm_comm_rccl_version(void);       /* NCCL_VERSION_CODE of the bound RCCL, 0 if none */
int mm_comm_unique_id(void* id_out /* host, MM_COMM_ID_BYTES */);
int mm_comm_init(mm_comm_t* comm, int rank, int world, const void* unique_id, int device);
int mm_comm_rank(mm_comm_t comm);
int mm_comm_world(mm_comm_t comm);
int mm_allreduce_sum(mm_comm_t comm, int dtype, void* buf, int64_t count, mm_stream_t stream);
int mm_comm_destroy(mm_comm_t comm);
const char* mm_comm_last_error(void);

/* ---- one training step per call ------------------------------------------------ */
/* The body of the reference's training loop for one full batch (graphembed/graphembed/train.py:198-222:
 * objective(dataset[idx], embedding.compute_dists(idx)) -> backward -> optimizer.step() for the point and the
 * scale parameter groups), issued by ONE call: the fused objective kernel of the embedding (mm_spd_pdist_loss /
 * mm_vec_pdist_loss / mm_product_pairs_loss) followed by the fused optimizer kernels of every parameter
 * (mm_*_rsgd_step, mm_*_rsgd_momentum_step, mm_*_radam_step[_multi]).  Nothing is allocated, nothing synchronises;
 * a caller whose loop is not captured in a HIP graph pays one foreign-function call per step instead of ~15.
 * Parameters are updated IN PLACE; gradients are left in the `grad` buffers (the scales' gradients in
 * loss_out[1 + k]); loss_out[0] is the loss BEFORE the update, as the reference logs it.
 * A single SPD factor is issued as TWO launches: the pair kernel (loss + gradient sums) and one per-point kernel that
 * finishes the gradient, applies the optimizer rule, writes the new point and its tables for the next step, closes the
 * loss record and updates a momentum-free RSGD scale — instead of prepare + pair + finalize + point update + scale update.
 * So is a single vector factor where mm_vec_fused_step_supports says so (the per-point kernel there moves the pair kernel's
 * sums into the gradient, steps the points and writes their zero-padded copy for the next pair kernel).
 * Multi-GPU: with a row range and a communicator the same call issues objective (this rank's pairs) -> one
 * all-reduce -> optimizer, still without touching the host in between (capturable as one HIP graph). */
enum { MM_OPT_NONE = -1 /* frozen: read by the objective, never stepped (a scale during burn-in) */,
       MM_OPT_RSGD = 0, MM_OPT_RADAM = 1 };
typedef struct mm_step_param {
  int kind;              /* MM_EUCLIDEAN / MM_LORENTZ / MM_SPHERE, or MM_FACTOR_SPD; flat parameters: MM_EUCLIDEAN */
  int dim;               /* m of a vector point, d of an SPD(d) point, 1 for a scalar                              */
  int64_t count;         /* points                                                                                */
  void* x;               /* the parameter (device), updated in place                                              */
  void* grad;            /* its Euclidean gradient (device): written by the objective, read by the optimizer       */
  int optimizer;         /* MM_OPT_RSGD (optim/rsgd.py:10-82) or MM_OPT_RADAM (optim/radam.py:12-98)               */
  double lr, momentum, dampening, max_grad_norm /* <= 0: no clipping */, beta1, beta2, adam_eps;
  int nc, exact;
  void* state0;          /* RSGD: momentum buffer (NULL when momentum == 0); Adam: exp_avg                         */
  void* state1;          /* Adam: exp_avg_sq                                                                       */
  double* step;          /* Adam: state['step'], device fp64                                                       */
  unsigned* ticket;      /* Adam: device counter, zero between calls                                               */
} mm_step_param;
typedef struct mm_train_step {
  size_t struct_size;            /* sizeof(mm_train_step) of the CALLER's header (mm_abi_version() >= 4): the library reads
                                    exactly that many bytes and takes every later field as zero, so a caller compiled against an
                                    older header of this layout keeps working when fields are appended; a value that is not a
                                    plausible size of this struct (below the ABI-4 base, or a caller of ABI <= 3 whose first
                                    member was `dtype`) is MM_ERR_ARG, and a LONGER struct whose extra tail is not all zero —
                                    a feature this library does not have — MM_ERR_UNSUPPORTED.  `mm_train_step s = {0};
                                    s.struct_size = sizeof s;` */
  int dtype, loss_kind, terms;
  double alpha, eps;             /* quotient loss: target scale and 1 / (epoch + 1)                                */
  const double* loss_params;     /* optional device {alpha, eps} overriding the two values above                   */
  double wmin, wmax;             /* SPD eigenvalue clamps (spd.py:29-30)                                           */
  int64_t n;                     /* points per factor                                                              */
  int nf;                        /* factors of the product manifold, 1..4                                          */
  mm_step_param points[4];       /* one per factor                                                                 */
  mm_step_param scales[4];       /* the factors' raw scale parameters (kind MM_EUCLIDEAN, dim 1, count 1;
                                    .grad is ignored: the gradient is loss_out[1 + k])                             */
  const void* target;            /* squared graph distances, pair-vector order [n(n-1)/2]                          */
  void* loss_out;                /* [1 + nf]                                                                       */
  void* ws;                      /* workspace of the embedding's objective kernel (mm_*_ws_bytes)                  */
  int ws_flags;                  /* MM_WS_CLEAN when a product workspace is known to be clean; MM_WS_PREPARED when a
                                    single SPD factor's workspace holds the tables of the CURRENT points: a step with an
                                    optimizer on the SPD points writes the tables of the new points itself (the optimizer
                                    kernel does what mm_spd_prepare does), so from the second consecutive step on the
                                    caller passes MM_WS_PREPARED — unless it changed the points in between.  The same
                                    holds for a single vector factor that takes the two-launch form
                                    (mm_vec_fused_step_supports) and its padded copy of the points, and for a product
                                    embedding on one GPU (the node table of its symmetric pair kernel); ignored elsewhere.
                                    A node-minibatch step (batch_idx) of a VECTOR factor does not rewrite the padded copy:
                                    the next full-batch step must not pass MM_WS_PREPARED (an SPD factor's does rewrite its tables) */
  /* -- sharded step (mm_abi_version() >= 2); all zero = the whole pair list on one GPU ------------------------- */
  int64_t row_begin, row_end;    /* this rank's rows of the pair list (mm_shard_rows); row_end <= 0 means n.  `target`
                                    is then this rank's SLICE: the targets of the pairs from mm_pair_offset(n,row_begin) on */
  mm_comm_t comm;                /* NULL: no collective.  Else ONE mm_allreduce_sum of reduce_buf between the objective
                                    and the optimizer kernels, on the same stream: afterwards every rank holds the full
                                    gradients and loss and applies the identical update — replicas stay equal, no broadcast */
  void* reduce_buf;              /* [reduce_count] of `dtype`: ONE allocation that contains every points[k].grad and
                                    loss_out (MM_ERR_ARG otherwise) — the message of the collective                 */
  int64_t reduce_count;
  /* -- node minibatch (mm_abi_version() >= 3); NULL / 0 = full batch -------------------------------------------------
     train.py:198-222 with batch_size set: the step's pair list is that of the `batch` nodes batch_idx[.] (device int64,
     distinct); `target` is then the DENSE n x n matrix of squared graph distances (GraphDataset.pdists), row_begin /
     row_end shard the batch's pair list, and the optimizer still steps all n points — those outside the batch with a
     zero gradient (the reference's dense x.grad; momentum and Adam state keep moving them).  Single SPD(d) or vector
     factor; MM_ERR_UNSUPPORTED for products (mm_product_pairs_loss_subset + the optimizer entry points serve those). */
  const int64_t* batch_idx;
  int64_t batch;
} mm_train_step;
int mm_train_step_run(const mm_train_step* step, mm_stream_t stream);
/* Largest d for which a single SPD(d) factor takes the two-launch form above (and therefore leaves the workspace holding
 * the tables of the new points: MM_WS_PREPARED on the next call); wider matrices are stepped by separate launches. */
int mm_spd_fused_step_max_dim(void);
/* 1 if a single vector factor (kind, dimension m) takes the two-launch form: the sizes at which the symmetric VALU pair
 * kernel is the fused objective (every Euclidean size it is built for; Lorentz / sphere up to m = 16). */
int mm_vec_fused_step_supports(int dtype, int kind, int m);

#ifdef __cplusplus
}
#endif
#endif /* MM_MANIFOLDS_H */
