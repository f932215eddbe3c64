// Launchers of the symmetric VALU backward of the vector manifolds (vec_sym.hpp): its own translation unit so that the
// ~100 instantiations compile beside vec.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"
#include "vec_sym.hpp"

namespace mm {

bool vec_sym_supports(int dtype, int kind, int m) {
  // (fp64, 17 <= m <= 32: 2 x 32 doubles of column point and sums per lane — built for the Euclidean factor only, whose
  // training step it completes: round 4)
  return (dtype == MM_F32 && m >= 1 && m <= 32) || (dtype == MM_F64 && m >= 1 && m <= (kind == MM_EUCLIDEAN ? 32 : 16));
}

namespace {

constexpr int kPlain = VSYM_PLAIN, kStepPrepared = VSYM_STEP_PREPARED;

template <typename T, int KIND, int MP, int LOSS, bool SQ>
int launch_pairs(const T* xpad, const T* g, int64_t n, int m, int64_t rb, int64_t re, T* acc, LossArgs<T> la, int* share_tab, hipStream_t st) {
  constexpr int kThreads = 64 * kVSymWaves;
  const int64_t units = ColWalk(int(n), int(rb), int(re), 64 * vsym_cols<T, MP>()).total();
  if (units <= 0) return MM_OK;
  int64_t grid = resident_workgroups<vec_pdist_bwd_sym_kernel<T, KIND, MP, LOSS, SQ>>(kThreads);
  // (measured: Lorentz(11) n = 4039, 512 / 768 / 1024 workgroups 23.8 / 22.0 / 22.7 us; narrow points — sphere(6) n = 5000, MP = 8 —
  // 30.5 / 26.9 / 25.0: profiles/r05_experiments.md)
  grid = std::min<int64_t>(grid, (MP <= 8 ? 4 : 3) * int64_t(device_cus()));
  {   // small launches: enough rows per workgroup to pay for its column flush (as the SPD backward, spd.hip)
    const int64_t cus = device_cus();
    // (as the SPD backward after round 5's prologue diet: ~32 rows per workgroup for the two-column kernels, to the nearest half
    // multiple of the CU count — Lorentz(11) n = 2000, 256 / 384 / 512 / 768 workgroups 11.8 / 11.2 / 10.9 / 11.7 us)
    const int64_t half = std::max<int64_t>(1, cus / 2);
    const int64_t by_rows = vsym_cols<T, MP>() >= 2 ? (units / 32 + half / 2) / half * half : units / 48 / cus * cus;
    if (by_rows < grid) grid = std::max<int64_t>(cus, by_rows);
  }
  static const int64_t env_grid = std::getenv("MM_VEC_BWD_GRID") ? std::atoll(std::getenv("MM_VEC_BWD_GRID")) : 0;
  if (env_grid > 0) grid = env_grid;
  grid = std::max<int64_t>(1, std::min<int64_t>(grid, (units + 7) / 8));
  // rows a share pays for entering a column block (ColWalk::enter; as the SPD backward: spd_pair.hpp) — MM_VEC_BWD_CROSS overrides
  static const int env_cross = std::getenv("MM_VEC_BWD_CROSS") ? std::atoi(std::getenv("MM_VEC_BWD_CROSS")) : (sizeof(T) == 4 ? 16 : 8);
  const ColWalk hw(int(n), int(rb), int(re), 64 * vsym_cols<T, MP>());
  const int cross = std::min(std::max(env_cross, 0), 1024);
  {
    ProfScope prof(PROF_VEC_BWD, st);
    vec_pdist_bwd_sym_kernel<T, KIND, MP, LOSS, SQ><<<dim3(unsigned(grid)), dim3(kThreads), 0, st>>>(xpad, g, int(n), m, int(rb),
                                                                                                   int(re), acc, la, WalkShares(hw.total_aug(cross), grid, cross), share_tab);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

template <typename T>
__global__ void vec_sym_loss_finalize_kernel(T* __restrict__ slots, const T* __restrict__ scale_raw, T* __restrict__ loss_out) {
  loss_finalize<T>(slots, scale_raw, loss_out);
}

template <typename T, int KIND, int MP>
int pairs_t(int loss_kind, int squared, const T* x, const T* g, int64_t n, int m, int64_t rb, int64_t re, void* ws,
            const T* scale_raw, double alpha, double eps, int terms, const double* loss_params, T* grad, T* loss_out,
            bool* finalized, int mode, hipStream_t st) {
  // The pair kernel flushes into the gradient itself — or, in the training-step form (mode != kPlain), into the head of the
  // accumulator region, gacc [n][m], which the step's per-point kernel moves to the gradient and clears
  const bool direct = mode == kPlain;
  T* acc = static_cast<T*>(ws);
  T* slots = acc + size_t(n) * (MP + 1);
  T* xpad = slots + 2 * kLossSlots;
  int* share_tab = reinterpret_cast<int*>(static_cast<char*>(ws) + vec_ws_tables_end(sizeof(T), n, MP));
  hipError_t e;
  if (mode != kStepPrepared) {
    // (direct form: only the loss slots of the accumulator region are used — and cleared)
    T* clear = direct ? slots : acc;
    const int clear_count = direct ? 2 * kLossSlots : int(n * (MP + 1) + 2 * kLossSlots);
    const int64_t work = std::max<int64_t>(std::max<int64_t>((n + 1) * MP, clear_count), direct ? n * m : 0);
    vec_sym_prep_kernel<T, MP><<<dim3(unsigned(std::min<int64_t>(1024, (work + 255) / 256))), dim3(256), 0, st>>>(
        x, int(n), m, xpad, clear, clear_count, direct ? grad : static_cast<T*>(nullptr));
    e = hipGetLastError();
    if (e != hipSuccess) return int(e);
  }
  *finalized = direct;
  int rc = MM_OK;
  if (re > rb && pair_off(n, re) > pair_off(n, rb)) {
    LossArgs<T> la{scale_raw, T(alpha), T(eps), terms, slots, loss_params};
    T* out = direct ? grad : acc;
    if (loss_kind == MM_LOSS_STRESS) rc = launch_pairs<T, KIND, MP, MM_LOSS_STRESS, true>(xpad, g, n, m, rb, re, out, la, share_tab, st);
    else if (loss_kind == MM_LOSS_QUOTIENT) rc = launch_pairs<T, KIND, MP, MM_LOSS_QUOTIENT, true>(xpad, g, n, m, rb, re, out, la, share_tab, st);
    else if (squared) rc = launch_pairs<T, KIND, MP, MM_LOSS_NONE, true>(xpad, g, n, m, rb, re, out, la, share_tab, st);
    else rc = launch_pairs<T, KIND, MP, MM_LOSS_NONE, false>(xpad, g, n, m, rb, re, out, la, share_tab, st);
    if (rc != MM_OK) return rc;
  }
  if (direct && loss_kind != MM_LOSS_NONE) {   // the loss record
    vec_sym_loss_finalize_kernel<T><<<dim3(1), dim3(64), 0, st>>>(slots, scale_raw, loss_out);
    e = hipGetLastError();
    if (e != hipSuccess) return int(e);
  }
  return MM_OK;
}

template <typename T, int KIND>
int pairs_mp(int loss_kind, int squared, const T* x, const T* g, int64_t n, int m, int64_t rb, int64_t re, void* ws,
             const T* scale_raw, double alpha, double eps, int terms, const double* loss_params, T* grad, T* loss_out,
             bool* finalized, int mode, hipStream_t st) {
#define MM_VSYM_CASE(MP_) case MP_: return pairs_t<T, KIND, MP_>(loss_kind, squared, x, g, n, m, rb, re, ws, scale_raw, alpha, eps, terms, loss_params, grad, loss_out, finalized, mode, st)
  switch (pad_dim(m)) {
    MM_VSYM_CASE(4);
    MM_VSYM_CASE(8);
    MM_VSYM_CASE(12);
    MM_VSYM_CASE(16);
    default: break;
  }
  if constexpr (sizeof(T) == 4 || KIND == MM_EUCLIDEAN) {
    switch (pad_dim(m)) {
      MM_VSYM_CASE(24);
      MM_VSYM_CASE(32);
      default: break;
    }
  }
#undef MM_VSYM_CASE
  return MM_ERR_UNSUPPORTED;
}

template <typename T>
int pairs_kind(int kind, int loss_kind, int squared, const T* x, const T* g, int64_t n, int m, int64_t rb, int64_t re, void* ws,
               const T* scale_raw, double alpha, double eps, int terms, const double* loss_params, T* grad, T* loss_out,
               bool* finalized, int mode, hipStream_t st) {
  switch (kind) {
    case MM_EUCLIDEAN: return pairs_mp<T, MM_EUCLIDEAN>(loss_kind, squared, x, g, n, m, rb, re, ws, scale_raw, alpha, eps, terms, loss_params, grad, loss_out, finalized, mode, st);
    case MM_LORENTZ: return pairs_mp<T, MM_LORENTZ>(loss_kind, squared, x, g, n, m, rb, re, ws, scale_raw, alpha, eps, terms, loss_params, grad, loss_out, finalized, mode, st);
    case MM_SPHERE: return pairs_mp<T, MM_SPHERE>(loss_kind, squared, x, g, n, m, rb, re, ws, scale_raw, alpha, eps, terms, loss_params, grad, loss_out, finalized, mode, st);
    default: return MM_ERR_ARG;
  }
}

}  // namespace

int vec_sym_backward_pairs(int dtype, int kind, int loss_kind, int squared, const void* x, const void* g, int64_t n, int m,
                           int64_t rb, int64_t re, void* ws, const void* scale_raw, double alpha, double eps, int terms,
                           const double* loss_params, void* grad, void* loss_out, bool* finalized, hipStream_t st, int mode) {
  *finalized = false;
  if (!vec_sym_supports(dtype, kind, m) || n > kSpdMaxNodes) return MM_ERR_UNSUPPORTED;
  if (dtype == MM_F32)
    return pairs_kind<float>(kind, loss_kind, squared, static_cast<const float*>(x), static_cast<const float*>(g), n, m, rb, re, ws,
                             static_cast<const float*>(scale_raw), alpha, eps, terms, loss_params, static_cast<float*>(grad),
                             static_cast<float*>(loss_out), finalized, mode, st);
  return pairs_kind<double>(kind, loss_kind, squared, static_cast<const double*>(x), static_cast<const double*>(g), n, m, rb, re,
                            ws, static_cast<const double*>(scale_raw), alpha, eps, terms, loss_params, static_cast<double*>(grad),
                            static_cast<double*>(loss_out), finalized, mode, st);
}

}  // namespace mm
