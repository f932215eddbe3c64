// Objective of a PRODUCT embedding in one pass over the pair vectors: the reference evaluates
//   m = sum_k softplus(s_k) * d2_k            (ManifoldEmbedding.compute_dists, modules.py:84-88)
//   loss = objective(target, m)               (objectives.py:16-45)
// and its backward through ~25 element-wise framework kernels per step (each a launch of a few
// microseconds: the reference's product configurations are launch-bound).  Here one kernel reads the
// K per-factor squared-distance vectors and the targets, and writes the loss partials, the per-factor
// upstream gradients g_k = dloss/dm * softplus(s_k) (what each factor's backward kernel consumes) and
// the partials of dloss/ds_k; a one-wavefront kernel finishes the sums.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "loss.hpp"
#include "smallmat.hpp"

namespace mm {

constexpr int kMaxFactors = 8;

template <typename T> struct ProductArgs {
  const T* d2[kMaxFactors];
  const T* scale_raw[kMaxFactors];
  T* g[kMaxFactors];
  int nf;
};

template <typename T, int LOSS>
__global__ __launch_bounds__(256) void product_loss_kernel(ProductArgs<T> pa, const T* __restrict__ target,
                                                           int64_t npairs, LossArgs<T> la) {
  T sp[kMaxFactors], ds[kMaxFactors];
  loss_resolve<T, LOSS>(la);
#pragma unroll
  for (int k = 0; k < kMaxFactors; ++k) { sp[k] = k < pa.nf ? softplus_of(pa.scale_raw[k]) : T(0); ds[k] = T(0); }
  T loss = T(0);
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < npairs; p += int64_t(gridDim.x) * blockDim.x) {
    T d[kMaxFactors], m = T(0);
#pragma unroll
    for (int k = 0; k < kMaxFactors; ++k)
      if (k < pa.nf) { d[k] = pa.d2[k][p]; m = Num<T>::fma(sp[k], d[k], m); }
    T dldm;
    loss += loss_term<T, LOSS>(m, target[p], la, dldm);
#pragma unroll
    for (int k = 0; k < kMaxFactors; ++k)
      if (k < pa.nf) { pa.g[k][p] = dldm * sp[k]; ds[k] = Num<T>::fma(dldm, d[k], ds[k]); }
  }
  // slots: [1 + nf][kLossSlots]
  __shared__ T red[4][1 + kMaxFactors];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T l = wave_sum(loss);
  if (lane == 0) red[wave][0] = l;
#pragma unroll
  for (int k = 0; k < kMaxFactors; ++k)
    if (k < pa.nf) {
      const T v = wave_sum(ds[k]);
      if (lane == 0) red[wave][1 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x <= unsigned(pa.nf)) {
    const T v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomic_add(&la.slots[size_t(threadIdx.x) * kLossSlots + (blockIdx.x & (kLossSlots - 1))], v);
  }
}

// loss_out[0] = loss, loss_out[1 + k] = d loss / d scale_raw_k; slots left clean
template <typename T>
__global__ void product_loss_finalize_kernel(ProductArgs<T> pa, T* __restrict__ slots, T* __restrict__ loss_out) {
  for (int q = 0; q <= pa.nf; ++q) {
    double v = 0.0;
    for (int t = threadIdx.x; t < kLossSlots; t += 64) { v += double(slots[size_t(q) * kLossSlots + t]); slots[size_t(q) * kLossSlots + t] = T(0); }
    v = wave_sum(v);
    if (threadIdx.x == 0) {
      if (q == 0) loss_out[0] = T(v);
      else {
        const double s = double(*pa.scale_raw[q - 1]);
        loss_out[q] = T(v / (1.0 + ::exp(-s)));  // d softplus = sigmoid
      }
    }
  }
}

template <typename T>
int product_loss_t(int loss_kind, int nf, const void* const* d2, const void* target, const void* const* scale_raw,
                   int64_t npairs, double alpha, double eps, int terms, const double* loss_params, void* const* g_out, void* loss_out, void* ws,
                   hipStream_t st) {
  ProductArgs<T> pa;
  pa.nf = nf;
  for (int k = 0; k < kMaxFactors; ++k) {
    pa.d2[k] = k < nf ? static_cast<const T*>(d2[k]) : nullptr;
    pa.scale_raw[k] = k < nf ? static_cast<const T*>(scale_raw[k]) : nullptr;
    pa.g[k] = k < nf ? static_cast<T*>(g_out[k]) : nullptr;
  }
  T* slots = static_cast<T*>(ws);
  hipError_t e = hipMemsetAsync(slots, 0, sizeof(T) * size_t(1 + nf) * kLossSlots, st);
  if (e != hipSuccess) return int(e);
  LossArgs<T> la{nullptr, T(alpha), T(eps), terms, slots, loss_params};
  if (npairs > 0) {
    const int blocks = int(npairs / 1024 < 1 ? 1 : (npairs / 1024 > 2048 ? 2048 : npairs / 1024));
    if (loss_kind == MM_LOSS_STRESS)
      product_loss_kernel<T, MM_LOSS_STRESS><<<dim3(blocks), dim3(256), 0, st>>>(pa, static_cast<const T*>(target), npairs, la);
    else
      product_loss_kernel<T, MM_LOSS_QUOTIENT><<<dim3(blocks), dim3(256), 0, st>>>(pa, static_cast<const T*>(target), npairs, la);
  }
  product_loss_finalize_kernel<T><<<dim3(1), dim3(64), 0, st>>>(pa, slots, static_cast<T*>(loss_out));
  e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

// Targets of a node minibatch straight from the dense target matrix (GraphDataset.__getitem__,
// data/dataset.py:19-27; train.py:203-213): out[pair (a, b)] = dense[idx[a]][idx[b]] for a < b in the
// pair-vector order — one launch instead of two row/column gathers, a triu_indices and a masked gather.
template <typename T>
__global__ __launch_bounds__(256) void pair_gather_kernel(const T* __restrict__ dense, int64_t n,
                                                          const int64_t* __restrict__ idx, int bs, T* __restrict__ out) {
  const int a = blockIdx.y;
  const int b = a + 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  // (clamped into the matrix: the indices are caller data; the host layer raises IndexError for host-side tensors)
  const int64_t ia = int64_t(min(uint64_t(idx[a]), uint64_t(n - 1))), ibb = int64_t(min(uint64_t(idx[b]), uint64_t(n - 1)));
  out[int64_t(a) * (2 * int64_t(bs) - a - 1) / 2 + (b - a - 1)] = dense[ia * n + ibb];
}

}  // namespace mm

using namespace mm;

extern "C" {

int mm_product_max_factors(void) { return kMaxFactors; }

size_t mm_product_loss_ws_bytes(int dtype, int nf) { return (dtype == MM_F64 ? 8 : 4) * size_t(1 + nf) * kLossSlots; }

int mm_product_loss(int dtype, int loss_kind, int nf, const void* const* d2, const void* target,
                    const void* const* scale_raw, int64_t npairs, double alpha, double eps, int terms, const double* loss_params,
                    void* const* g_out, void* loss_out, void* ws, mm_stream_t stream) {
  if (nf < 1 || nf > kMaxFactors || !d2 || !scale_raw || !g_out || !loss_out || !ws || npairs < 0 || (npairs > 0 && !target))
    return MM_ERR_ARG;
  for (int k = 0; k < nf; ++k)
    if (!scale_raw[k] || (npairs > 0 && (!d2[k] || !g_out[k]))) return MM_ERR_ARG;
  if (loss_kind != MM_LOSS_STRESS && loss_kind != MM_LOSS_QUOTIENT) return MM_ERR_UNSUPPORTED;
  if (loss_kind == MM_LOSS_QUOTIENT && !(terms & 3)) return MM_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32)
    return product_loss_t<float>(loss_kind, nf, d2, target, scale_raw, npairs, alpha, eps, terms, loss_params, g_out, loss_out, ws, st);
  if (dtype == MM_F64)
    return product_loss_t<double>(loss_kind, nf, d2, target, scale_raw, npairs, alpha, eps, terms, loss_params, g_out, loss_out, ws, st);
  return MM_ERR_ARG;
}

int mm_pair_gather(int dtype, const void* dense, int64_t n, const int64_t* idx, int64_t bs, void* out,
                   mm_stream_t stream) {
  if (n < 0 || bs < 0 || bs > (1 << 30) || (bs > 1 && (!dense || !idx || !out))) return MM_ERR_ARG;
  if (bs < 2) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(unsigned((bs - 1 + 255) / 256), unsigned(bs - 1));
  if (dtype == MM_F32)
    pair_gather_kernel<float><<<grid, dim3(256), 0, st>>>(static_cast<const float*>(dense), n, idx, int(bs),
                                                         static_cast<float*>(out));
  else if (dtype == MM_F64)
    pair_gather_kernel<double><<<grid, dim3(256), 0, st>>>(static_cast<const double*>(dense), n, idx, int(bs),
                                                          static_cast<double*>(out));
  else
    return MM_ERR_ARG;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}

}  // extern "C"
