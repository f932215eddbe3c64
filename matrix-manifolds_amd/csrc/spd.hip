// SPD(d) affine-invariant pairwise distance + gradient, and the per-point maps
// RiemannianSGD needs, as hand-written gfx950 kernels.
//
// Reference arithmetic: graphembed/graphembed/manifolds/spd.py:108-181 and
// linalg/{fast,torch_batch}.py (see include/mm_manifolds.h for the per-entry
// citations).  Design notes: DESIGN.md §3.
//
// Pair kernels: lanes own consecutive columns j of the upper triangle and keep the Cholesky factor
// L_j of their column (packed) in VGPRs for a whole tile; the row operand L_i^-1 is wave-uniform, so
// it is fetched with scalar loads into SGPRs (no LDS, no VGPRs) and A = (L_i^-1 L_j)(L_i^-1 L_j)^T
// = L_i^-1 X_j L_i^-T costs 20 FMAs with one scalar operand each (D = 3).
// log(A) comes, per wavefront, from the cheapest method that holds for all 64 pairs: the
// Cayley-Hamilton series of log(I + E) for close pairs (fp32), the Cayley-transform logarithm
// (atanh of (A - mu)(A + mu)^-1) for eigenvalue ratios up to ~16, cyclic Jacobi in registers for the
// rest (smallmat.hpp).  Forward stores d^2 with lanes on consecutive j -> 256-B coalesced segments
// of the row-major pair vector.  Backward recomputes log(A) (cheaper than a 28-48 B/pair round trip
// through HBM), accumulates the column side per lane in registers (64-column tiles shared by the
// wavefronts of a workgroup, combined in LDS), reduces the row side across the wavefront with a
// transposing DPP/permlane reduction, and flushes both once per tile with float atomics into
// self-cleaning structure-of-arrays accumulators; the fused objective (loss.hpp) rides on the same
// kernel.  spd_stein.hpp: the Stein divergence on the same tiling and workspace.
#include "spd_pair.hpp"

namespace mm {

__global__ void spd_count_bad_kernel(const int* __restrict__ bad, int n, int* __restrict__ status) {
  int c = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) c += bad[i];
  c = wave_sum(c);
  __shared__ int part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) t += part[w];
    status[0] = t;
  }
}

#include "spd_stein.hpp"

// dtype x D dispatch ---------------------------------------------------------
template <typename T, int D> int spd_prepare_t(const T* x, int64_t n, void* wsp, hipStream_t st) {
  Ws<T> ws(wsp, n, D);
  return spd_pdist_prepare<T, D>(x, n, ws, 0, st);
}

template <typename T, int D, typename K, typename... A>
int launch_pointwise(K kernel, int64_t m, hipStream_t st, A... args) {
  if (m <= 0) return MM_OK;
  const int bs = 128;
  kernel<<<dim3((unsigned)((m + bs - 1) / bs)), dim3(bs), 0, st>>>(args...);
  MM_CHECK_LAUNCH();
  return MM_OK;
}

}  // namespace mm

using namespace mm;

extern "C" {

#ifdef MM_BWD_STAMP
int mm_dbg_read_bwd_stamps(void* host, size_t bytes) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(mm::g_bwd_stamps), bytes));
}
int mm_dbg_read_bwd_marks(void* host, size_t bytes) {
  return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(mm::g_bwd_marks), bytes));
}
#endif

int mm_spd_max_dim(void) { return kSpdMaxD; }

size_t mm_spd_pdist_ws_bytes(int dtype, int64_t n, int d) {
  return dtype == MM_F64 ? Ws<double>::bytes(n, d) : Ws<float>::bytes(n, d);
}

int mm_spd_pdist_fwd(int dtype, const void* x, int64_t n, int d, int64_t row_begin, int64_t row_end, int squared,
                     double wmin, double wmax, void* out, void* ws, int flags, mm_stream_t stream) {
  if (!x || !ws || n < 0 || row_begin < 0 || row_end > n || row_begin > row_end || n > kSpdMaxNodes) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!out && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (spd_pdist_fwd_t<T, D>(static_cast<const T*>(x), n, row_begin, row_end, squared, wmin, wmax,
                                     static_cast<T*>(out), ws, flags, st)));
}

int mm_spd_prepare(int dtype, const void* x, int64_t n, int d, void* ws, mm_stream_t stream) {
  if (!x || !ws || n < 0 || n > kSpdMaxNodes) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d, (spd_prepare_t<T, D>(static_cast<const T*>(x), n, ws, st)));
}

int mm_spd_pdist_bwd(int dtype, const void* x, const void* g, int64_t n, int d, int64_t row_begin, int64_t row_end,
                     int squared, double wmin, double wmax, void* grad_x, void* ws, int flags, mm_stream_t stream) {
  if (!x || !ws || !grad_x || n < 0 || row_begin < 0 || row_end > n || row_begin > row_end || n > kSpdMaxNodes)
    return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!g && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (spd_pdist_bwd_t<T, D>(static_cast<const T*>(x), static_cast<const T*>(g), n, row_begin, row_end,
                                     squared, wmin, wmax, static_cast<T*>(grad_x), ws, flags, st)));
}

int mm_spd_status(void* ws, int64_t n, int* host_status, mm_stream_t stream) {
  if (!ws || !host_status || n < 0) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int* status = static_cast<int*>(ws);
  const int* bad = reinterpret_cast<const int*>(static_cast<char*>(ws) + 64);
  spd_count_bad_kernel<<<dim3(1), dim3(256), 0, st>>>(bad, int(n), status);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return int(e);
  e = hipMemcpyAsync(host_status, status, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return int(e);
  e = hipStreamSynchronize(st);
  return e == hipSuccess ? MM_OK : int(e);
}

int mm_spd_dist_fwd(int dtype, const void* x, const void* y, int64_t m, int d, int squared, double wmin, double wmax,
                    void* out, mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !y || !out))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_dist_fwd_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(y), m, squared, T(wmin), T(wmax), static_cast<T*>(out))));
}

int mm_spd_dist_bwd(int dtype, const void* x, const void* y, const void* g, int64_t m, int d, int squared,
                    double wmin, double wmax, void* grad_x, void* grad_y, mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !y || !g || !grad_x || !grad_y))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_dist_bwd_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(y), static_cast<const T*>(g), m, squared, T(wmin),
                                      T(wmax), static_cast<T*>(grad_x), static_cast<T*>(grad_y))));
}

int mm_spd_map(int dtype, int op, const void* x, const void* u, int64_t m, int d, double wmin, double wmax, void* out,
               mm_stream_t stream) {
  if (m < 0 || op < 0 || op > MM_SPD_PROJU || (m > 0 && (!x || !out))) return MM_ERR_ARG;
  if (m > 0 && op != MM_SPD_PROJX && !u) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_map_kernel<T, D>, m, st, op, static_cast<const T*>(x),
                                      static_cast<const T*>(u), m, T(wmin), T(wmax), static_cast<T*>(out))));
}

int mm_spd_norm(int dtype, const void* x, const void* u, int64_t m, int d, int squared, void* out,
                mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !u || !out))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_norm_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(u), m, squared, static_cast<T*>(out))));
}

int mm_spd_eigvalsh(int dtype, const void* x, int64_t m, int d, void* w, mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !w))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_eigvalsh_kernel<T, D>, m, st, static_cast<const T*>(x), m, static_cast<T*>(w))));
}

int mm_spd_rsgd_step(int dtype, const void* x, const void* egrad, int64_t m, int d, double lr, double max_grad_norm,
                     int exact, void* x_new, mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !egrad || !x_new))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_rsgd_step_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(egrad), m, T(lr), T(max_grad_norm), exact,
                                      static_cast<T*>(x_new))));
}

int mm_spd_rsgd_momentum_step(int dtype, const void* x, const void* egrad, void* momentum_buffer, int64_t m, int d,
                              double lr, double momentum, double dampening, double max_grad_norm, int exact, void* x_new,
                              mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !egrad || !momentum_buffer || !x_new))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_rsgd_momentum_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(egrad), static_cast<T*>(momentum_buffer), m, T(lr),
                                      T(momentum), T(dampening), T(max_grad_norm), exact, static_cast<T*>(x_new))));
}

int mm_spd_radam_step(int dtype, const void* x, const void* egrad, void* exp_avg, void* exp_avg_sq, double* step,
                      unsigned* ticket, int64_t m, int d, double lr, double beta1, double beta2, int nc, double eps,
                      double max_grad_norm, int exact, void* x_new, mm_stream_t stream) {
  if (m < 0 || !step || !ticket || (m > 0 && (!x || !egrad || !exp_avg || !exp_avg_sq || !x_new))) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_radam_step_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(egrad), static_cast<T*>(exp_avg),
                                      static_cast<T*>(exp_avg_sq), m,
                                      AdamArgs<T>{T(lr), T(beta1), T(beta2), T(eps), T(max_grad_norm), nc, exact, step,
                                                  ticket},
                                      static_cast<T*>(x_new))));
}

int mm_spd_stein_pdiv_fwd(int dtype, const void* x, int64_t n, int d, int64_t row_begin, int64_t row_end, int squared,
                          double wmin, void* out, void* ws, int flags, mm_stream_t stream) {
  if (!x || !ws || n < 0 || row_begin < 0 || row_end > n || row_begin > row_end || n > kSpdMaxNodes) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!out && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (spd_stein_fwd_t<T, D>(static_cast<const T*>(x), n, row_begin, row_end, squared, wmin,
                                     static_cast<T*>(out), ws, flags, st)));
}

int mm_spd_stein_pdiv_bwd(int dtype, const void* x, const void* g, int64_t n, int d, int64_t row_begin, int64_t row_end,
                          int squared, double wmin, void* grad_x, void* ws, int flags, mm_stream_t stream) {
  if (!x || !ws || !grad_x || n < 0 || row_begin < 0 || row_end > n || row_begin > row_end || n > kSpdMaxNodes)
    return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  if (!g && mm_pair_offset(n, row_end) > mm_pair_offset(n, row_begin)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (spd_stein_bwd_t<T, D>(static_cast<const T*>(x), static_cast<const T*>(g), n, row_begin, row_end, squared,
                                     wmin, static_cast<T*>(grad_x), ws, flags, st)));
}

int mm_spd_stein_div(int dtype, const void* x, const void* y, const void* g, int64_t m, int d, int squared, double wmin,
                     void* out, void* grad_x, void* grad_y, mm_stream_t stream) {
  if (m < 0 || (m > 0 && (!x || !y)) || ((grad_x != nullptr) != (grad_y != nullptr)) || (grad_x && !g) ||
      (!out && !grad_x))
    return MM_ERR_ARG;
  if (m == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MM_DISPATCH(dtype, d,
              (launch_pointwise<T, D>(spd_stein_div_kernel<T, D>, m, st, static_cast<const T*>(x),
                                      static_cast<const T*>(y), static_cast<const T*>(g), m, squared, T(wmin),
                                      static_cast<T*>(out), static_cast<T*>(grad_x), static_cast<T*>(grad_y))));
}

}  // extern "C"
