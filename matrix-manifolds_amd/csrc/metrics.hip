// Graph-reconstruction metric on the GPU: per-node average precision of the embedding's neighbour ranking —
// the quantity graphembed/graphembed/pyx/impl/precision.cpp (FastPrecision::MeanAveragePrecision) and
// metrics.py:61-96 (py_mean_average_precision) average into the MAP score.
//
//   AP(u) = 1/deg(u) * sum over neighbours v of u of  k_v / r_v,
//   r_v = rank of v among all other nodes by embedding distance to u (1 = closest),
//   k_v = number of neighbours of u with rank <= r_v.
//
// No sort is needed: only the ranks of the deg(u) neighbours matter, and a rank is a count.  One workgroup
// per node: for each neighbour the workgroup counts the nodes that are closer (ties broken by node index,
// i.e. a stable argsort), then the neighbours' ranks are ranked among themselves.  Cost 2|E| n comparisons
// against the n^2 log n of sorting every row; rows of the dense distance matrix are read coalesced and stay
// in L2 for the deg(u) passes.
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"
#include "smallmat.hpp"

namespace mm {

constexpr int kApBlock = 256;

template <typename T>
__global__ __launch_bounds__(kApBlock) void average_precision_kernel(const T* __restrict__ dist /* [n][n] */, int n,
                                                                     const int* __restrict__ indptr,
                                                                     const int* __restrict__ indices,
                                                                     int* __restrict__ rank /* [nnz] scratch */,
                                                                     T* __restrict__ ap /* [n] */) {
  __shared__ int part[kApBlock / 64];
  __shared__ double partd[kApBlock / 64];
  const int u = blockIdx.x;
  const int e0 = indptr[u], e1 = indptr[u + 1];
  const T* row = dist + size_t(u) * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = e0; e < e1; ++e) {  // block-uniform
    const int v = indices[e];
    const T dv = row[v];
    int c = 0;
    for (int w = threadIdx.x; w < n; w += kApBlock) {
      const T dw = row[w];
      c += (w != u && w != v && (dw < dv || (dw == dv && w < v))) ? 1 : 0;
    }
    c = wave_sum(c);
    if (lane == 0) part[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 1;
      for (int k = 0; k < kApBlock / 64; ++k) t += part[k];
      rank[e] = t;
    }
    __syncthreads();
  }
  // k_v / r_v summed over the neighbours
  double acc = 0.0;
  for (int e = e0 + threadIdx.x; e < e1; e += kApBlock) {
    const int r = rank[e];
    int k = 0;
    for (int f = e0; f < e1; ++f) k += rank[f] <= r ? 1 : 0;
    acc += double(k) / double(r);
  }
  acc = wave_sum(acc);
  if (lane == 0) partd[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kApBlock / 64; ++k) t += partd[k];
    ap[u] = e1 > e0 ? T(t / double(e1 - e0)) : T(0);
  }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_graph_average_precision(int dtype, const void* dist, int64_t n, const int* indptr, const int* indices,
                                          int* rank_scratch, void* ap_out, mm_stream_t stream) {
  if (n < 0 || n > (1 << 30) || (n > 0 && (!dist || !indptr || !ap_out))) return MM_ERR_ARG;
  if (n == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32)
    average_precision_kernel<float><<<dim3(unsigned(n)), dim3(kApBlock), 0, st>>>(
        static_cast<const float*>(dist), int(n), indptr, indices, rank_scratch, static_cast<float*>(ap_out));
  else if (dtype == MM_F64)
    average_precision_kernel<double><<<dim3(unsigned(n)), dim3(kApBlock), 0, st>>>(
        static_cast<const double*>(dist), int(n), indptr, indices, rank_scratch, static_cast<double*>(ap_out));
  else
    return MM_ERR_ARG;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}
