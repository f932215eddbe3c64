/* A torch-free C client of the C ABI: plain C (gcc), hipMalloc'ed buffers, the HIP stream API — what a cgo / JNI /
 * ctypes binding of the reference's Manifold plugin would do.  SPD(3): the pair distances of the
 * points X_k = diag(e^{a_k}, e^{2 a_k}, e^{-a_k}) have the closed form d^2_ij = 6 (a_i - a_j)^2, and the gradient of
 * sum_ij d^2_ij w.r.t. X_i is diagonal with entries 2 c_m s_i / x_m, s_i = sum_j (a_i - a_j), c = (1, 2, -1).
 * Also a vector manifold (Euclidean), the fused RSGD step, the RCCL collective and the one-call training step.
 * Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mm_manifolds.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "hip error %d line %d\n", (int)r_, __LINE__); return 2; } } while (0)
#define CHECK_MM(e) do { int r_ = (e); if (r_ != MM_OK) { fprintf(stderr, "mm error %d line %d\n", r_, __LINE__); return 3; } } while (0)

int main(void) {
  const int64_t n = 37;
  const int d = 3;
  const int64_t P = n * (n - 1) / 2;
  if (mm_abi_version() < 2) return 4;
  if (mm_pair_offset(n, n) != P) return 4;
  double* a = malloc(sizeof(double) * n);
  double* x = calloc((size_t)n * d * d, sizeof(double));
  for (int64_t k = 0; k < n; ++k) {
    a[k] = 0.02 * (double)k - 0.3;
    x[k * 9 + 0] = exp(a[k]);
    x[k * 9 + 4] = exp(2 * a[k]);
    x[k * 9 + 8] = exp(-a[k]);
  }
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  void *dx, *dout, *dg, *dgrad, *ws;
  const size_t wsb = mm_spd_pdist_ws_bytes(MM_F64, n, d);
  CHECK_HIP(hipMalloc(&dx, sizeof(double) * n * 9));
  CHECK_HIP(hipMalloc(&dout, sizeof(double) * P));
  CHECK_HIP(hipMalloc(&dg, sizeof(double) * P));
  CHECK_HIP(hipMalloc(&dgrad, sizeof(double) * n * 9));
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_HIP(hipMemcpyAsync(dx, x, sizeof(double) * n * 9, hipMemcpyHostToDevice, st));
  CHECK_MM(mm_spd_pdist_fwd(MM_F64, dx, n, d, 0, n, 1, 1e-8, 1e8, dout, ws, 0, st));
  double* out = malloc(sizeof(double) * P);
  double* ones = malloc(sizeof(double) * P);
  for (int64_t k = 0; k < P; ++k) ones[k] = 1.0;
  CHECK_HIP(hipMemcpyAsync(dg, ones, sizeof(double) * P, hipMemcpyHostToDevice, st));
  CHECK_MM(mm_spd_pdist_bwd(MM_F64, dx, dg, n, d, 0, n, 1, 1e-8, 1e8, dgrad, ws, MM_WS_PREPARED, st));
  double* grad = malloc(sizeof(double) * n * 9);
  CHECK_HIP(hipMemcpyAsync(out, dout, sizeof(double) * P, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(grad, dgrad, sizeof(double) * n * 9, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  int status = -1;
  CHECK_MM(mm_spd_status(ws, n, &status, st));
  if (status != 0) { fprintf(stderr, "status %d\n", status); return 5; }
  double worst = 0.0, worst_g = 0.0;
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = i + 1; j < n; ++j, ++k) {
      const double want = fmax(6.0 * (a[i] - a[j]) * (a[i] - a[j]), 1e-8);
      worst = fmax(worst, fabs(out[k] - want));
    }
  const double c[3] = {1.0, 2.0, -1.0};  /* log x_m = c_m a */
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t j = 0; j < n; ++j) s += a[i] - a[j];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        const double want = r == q ? 2.0 * c[r] * s / x[i * 9 + r * 4] : 0.0;
        worst_g = fmax(worst_g, fabs(grad[i * 9 + r * 3 + q] - want));
      }
  }
  printf("spd(3) n=%ld: max |d2 - closed form| = %.3e, max |grad - closed form| = %.3e\n", (long)n, worst, worst_g);
  if (!(worst < 1e-10 && worst_g < 1e-8)) return 6;

  /* Euclidean(4): d2 = |y - x|^2; one RSGD step x' = x - lr g */
  const int m = 4;
  float xe[3 * 4] = {0, 0, 0, 0, 1, 2, 2, 0, -1, 0, 0, 1}, ge[3 * 4], oute[3], xn[3 * 4];
  for (int t = 0; t < 12; ++t) ge[t] = 0.5f * (float)(t % 3);
  void *dxe, *doute, *dge, *dxn;
  CHECK_HIP(hipMalloc(&dxe, sizeof xe)); CHECK_HIP(hipMalloc(&doute, sizeof oute));
  CHECK_HIP(hipMalloc(&dge, sizeof ge)); CHECK_HIP(hipMalloc(&dxn, sizeof xn));
  CHECK_HIP(hipMemcpyAsync(dxe, xe, sizeof xe, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(dge, ge, sizeof ge, hipMemcpyHostToDevice, st));
  CHECK_MM(mm_vec_pdist_fwd(MM_F32, MM_EUCLIDEAN, dxe, 3, m, 0, 3, 1, doute, st));
  CHECK_MM(mm_vec_rsgd_step(MM_F32, MM_EUCLIDEAN, dxe, dge, 3, m, 0.1, -1.0, 0, dxn, st));
  CHECK_HIP(hipMemcpyAsync(oute, doute, sizeof oute, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(xn, dxn, sizeof xn, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  if (fabsf(oute[0] - 9.f) > 1e-5f || fabsf(oute[1] - 2.f) > 1e-5f || fabsf(oute[2] - 13.f) > 1e-5f) return 7;
  for (int t = 0; t < 12; ++t)
    if (fabsf(xn[t] - (xe[t] - 0.1f * ge[t])) > 1e-6f) return 8;
  printf("euclidean(4): pdist and RSGD step ok\n");
  /* the collective of the sharded path (replaces torch.nn.DataParallel, train.py:107-109): a one-rank RCCL communicator —
   * rendezvous token, init, an in-place all-reduce(sum) on the stream (one rank: the buffer must come back unchanged),
   * and the sharded training-step descriptor's argument checks */
  if (mm_comm_available()) {
    char token[MM_COMM_ID_BYTES];
    mm_comm_t comm = NULL;
    CHECK_MM(mm_comm_unique_id(token));
    int dev = 0;
    CHECK_HIP(hipGetDevice(&dev));
    CHECK_MM(mm_comm_init(&comm, 0, 1, token, dev));
    if (mm_comm_world(comm) != 1 || mm_comm_rank(comm) != 0) return 10;
    CHECK_MM(mm_allreduce_sum(comm, MM_F32, dxe, 12, st));
    CHECK_MM(mm_allreduce_sum(comm, MM_F64, dgrad, n * 9, st));
    float back[12];
    double* gback = malloc(sizeof(double) * n * 9);
    CHECK_HIP(hipMemcpyAsync(back, dxe, sizeof back, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(gback, dgrad, sizeof(double) * n * 9, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < 12; ++t) if (back[t] != xe[t]) return 11;
    for (int64_t t = 0; t < n * 9; ++t) if (gback[t] != grad[t]) return 11;
    if (mm_allreduce_sum(NULL, MM_F32, dxe, 12, st) != MM_ERR_ARG) return 12;
    if (mm_allreduce_sum(comm, 7, dxe, 12, st) != MM_ERR_ARG) return 12;
    CHECK_MM(mm_comm_destroy(comm));
    printf("mm_comm: one-rank RCCL communicator (RCCL %d), all-reduce in place ok\n", mm_comm_rccl_version());
  } else {
    printf("mm_comm: RCCL not bound (%s)\n", mm_comm_last_error());
    return 13;
  }
  /* one training step per call (train.py:198-222 for one full batch): Euclidean(3), n = 5, StressLoss, RSGD on the points and
   * on the scale — two consecutive mm_train_step_run calls (the second with MM_WS_PREPARED: the first left the workspace
   * prepared) against the same two steps done here in double precision:
   *   m_ij = softplus(s) |x_i - x_j|^2,  loss = sum (m_ij - t_ij)^2,  dL/dx_j = sum_i 4 softplus(s) (m_ij - t_ij) (x_j - x_i),
   *   dL/ds = sigmoid(s) sum 2 (m_ij - t_ij) |x_i - x_j|^2,  x' = x - lr g,  s' = s - lr_s dL/ds                               */
  {
    enum { TN = 5, TM = 3, TP = TN * (TN - 1) / 2 };
    double tx[TN * TM], tt[TP], ts = 0.3, hx[TN * TM], hs = 0.3, lout[2];
    for (int t = 0; t < TN * TM; ++t) hx[t] = tx[t] = 0.1 * (double)((t * 7) % 11) - 0.4;
    for (int t = 0; t < TP; ++t) tt[t] = 0.05 + 0.03 * (double)t;
    const double lr = 0.01, lr_s = 0.001;
    if (!mm_vec_fused_step_supports(MM_F64, MM_EUCLIDEAN, TM)) return 14;
    void *dtx, *dtg, *dtt, *dts, *dlo, *tws;
    CHECK_HIP(hipMalloc(&dtx, sizeof tx)); CHECK_HIP(hipMalloc(&dtg, sizeof tx)); CHECK_HIP(hipMalloc(&dtt, sizeof tt));
    CHECK_HIP(hipMalloc(&dts, sizeof ts)); CHECK_HIP(hipMalloc(&dlo, sizeof lout));
    const size_t twb = mm_vec_pdist_ws_bytes(MM_F64, TN, TM);
    CHECK_HIP(hipMalloc(&tws, twb));
    CHECK_HIP(hipMemsetAsync(tws, 0, twb, st));
    CHECK_HIP(hipMemcpyAsync(dtx, tx, sizeof tx, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(dtt, tt, sizeof tt, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(dts, &ts, sizeof ts, hipMemcpyHostToDevice, st));
    mm_train_step step;
    memset(&step, 0, sizeof step);
    /* an unversioned struct (struct_size 0, or an ABI <= 3 caller whose first member was dtype) is refused, not misread */
    if (mm_train_step_run(&step, st) != MM_ERR_ARG) return 16;
    step.struct_size = sizeof step;
    step.dtype = MM_F64; step.loss_kind = MM_LOSS_STRESS; step.n = TN; step.nf = 1;
    step.wmin = 1e-8; step.wmax = 1e8;
    step.points[0].kind = MM_EUCLIDEAN; step.points[0].dim = TM; step.points[0].count = TN;
    step.points[0].x = dtx; step.points[0].grad = dtg; step.points[0].optimizer = MM_OPT_RSGD;
    step.points[0].lr = lr; step.points[0].max_grad_norm = -1.0; step.points[0].exact = 1;
    step.scales[0].kind = MM_EUCLIDEAN; step.scales[0].dim = 1; step.scales[0].count = 1; step.scales[0].x = dts;
    step.scales[0].optimizer = MM_OPT_RSGD; step.scales[0].lr = lr_s; step.scales[0].max_grad_norm = -1.0;
    step.target = dtt; step.loss_out = dlo; step.ws = tws;
    double want_loss[2];
    for (int it = 0; it < 2; ++it) {
      step.ws_flags = it == 0 ? 0 : MM_WS_PREPARED;
      CHECK_MM(mm_train_step_run(&step, st));
      /* the same step on the host */
      const double sp = log1p(exp(hs)), sg = 1.0 / (1.0 + exp(-hs));
      double g[TN * TM] = {0}, gs = 0.0, loss = 0.0;
      int pk = 0;
      for (int i = 0; i < TN; ++i)
        for (int j = i + 1; j < TN; ++j, ++pk) {
          double d2 = 0.0;
          for (int q = 0; q < TM; ++q) d2 += (hx[i * TM + q] - hx[j * TM + q]) * (hx[i * TM + q] - hx[j * TM + q]);
          const double r = sp * d2 - tt[pk];
          loss += r * r;
          gs += 2.0 * r * d2 * sg;
          for (int q = 0; q < TM; ++q) {
            const double df = hx[j * TM + q] - hx[i * TM + q];
            g[j * TM + q] += 4.0 * sp * r * df;
            g[i * TM + q] -= 4.0 * sp * r * df;
          }
        }
      for (int t = 0; t < TN * TM; ++t) hx[t] -= lr * g[t];
      hs -= lr_s * gs;
      want_loss[it] = loss;
    }
    CHECK_HIP(hipMemcpyAsync(tx, dtx, sizeof tx, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(&ts, dts, sizeof ts, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(lout, dlo, sizeof lout, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    double wx = 0.0;
    for (int t = 0; t < TN * TM; ++t) wx = fmax(wx, fabs(tx[t] - hx[t]));
    printf("mm_train_step_run (Euclidean(3), 2 steps): max |x - host| = %.3e, |s - host| = %.3e, loss %.6f (host %.6f)\n", wx,
           fabs(ts - hs), lout[0], want_loss[1]);
    if (!(wx < 1e-12 && fabs(ts - hs) < 1e-12 && fabs(lout[0] - want_loss[1]) < 1e-10 * fmax(1.0, want_loss[1]))) return 15;
  }
  /* argument errors are return codes, never exceptions */
  if (mm_spd_pdist_fwd(MM_F64, NULL, n, d, 0, n, 1, 1e-8, 1e8, dout, ws, 0, st) != MM_ERR_ARG) return 9;
  if (mm_spd_pdist_fwd(MM_F64, dx, n, 10, 0, n, 1, 1e-8, 1e8, dout, ws, 0, st) != MM_ERR_UNSUPPORTED) return 9;   /* SPD(2..9) */
  printf("C ABI smoke: ok\n");
  return 0;
}
