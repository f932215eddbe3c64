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

// ---- layer-wise F1 scores (precision.cpp:300-429) ---------------------------------------------------
// For the shortest-path tree rooted at u every other node v gets an F1 score that compares the embedding's
// ordering around u with the tree's layers:
//   i(v)  = rank of v by embedding distance to u (1 = closest; ties by node index)
//   nb(v) = 1 + #{w ranked before v with layer(w) <= layer(v)}          ("before v in BOTH orderings")
//   precision = nb / i,   recall = nb / (#{layer < layer(v)} + #{w ranked before v on v's layer} + 1)
// and the scores are averaged per layer over all trees (LayerMeanF1Scores) or first per tree, then over
// the trees (LayerMeanAverageF1Scores).  The reference walks each sorted row with an ordered multiset
// (O(n) per insertion).  Here the rows arrive sorted (`order`: node ids by embedding distance, a stable
// segmented sort done by the caller) and one workgroup per tree walks its row in chunks of 256 positions:
// running per-layer counts of the nodes already passed live in LDS, the positions inside a chunk are
// resolved by a 256 x 256 comparison — O(n L + 256 n) per tree instead of O(n^2).
// (kMaxLayers = layer capacity of the instantiation: 256 for real graphs, 2048 for long paths / rings)
template <int kMaxLayers>
__global__ __launch_bounds__(kApBlock) void layer_f1_kernel(const int* __restrict__ order /* [n][n] */,
                                                            const int* __restrict__ hops /* [n][n] */, int n,
                                                            const int* __restrict__ indptr, int min_degree,
                                                            int max_degree, int per_tree_average, int num_layers,
                                                            double* __restrict__ m1, double* __restrict__ m2,
                                                            double* __restrict__ counts) {
  __shared__ int hist[kMaxLayers], strict_before[kMaxLayers], seen[kMaxLayers], seen_le[kMaxLayers], lcnt[kMaxLayers];
  __shared__ double lm1[kMaxLayers], lm2[kMaxLayers];
  __shared__ int chunk_layer[kApBlock];
  const int u = blockIdx.x;
  const int deg = indptr[u + 1] - indptr[u];
  if (deg < min_degree || deg > max_degree) return;  // block-uniform
  const int* orow = order + size_t(u) * n;
  const int* hrow = hops + size_t(u) * n;
  for (int l = threadIdx.x; l < kMaxLayers; l += kApBlock) { hist[l] = 0; seen[l] = 0; lcnt[l] = 0; lm1[l] = 0.0; lm2[l] = 0.0; }
  __syncthreads();
  for (int w = threadIdx.x; w < n; w += kApBlock)
    if (w != u) atomicAdd(&hist[min(hrow[w], kMaxLayers - 1)], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;  // nodes on layers 1 .. l-1 (the root's layer 0 is not counted)
    strict_before[0] = 0;
    for (int l = 1; l < kMaxLayers; ++l) { strict_before[l] = run; run += hist[l]; }
  }
  int passed = 0;  // positions of the sorted row already consumed, the root excluded (block-uniform)
  for (int p0 = 0; p0 < n; p0 += kApBlock) {
    __syncthreads();
    if (threadIdx.x == 0) {  // seen_le[l] = nodes already passed on layers <= l
      int run = 0;
      for (int l = 0; l < kMaxLayers; ++l) { run += seen[l]; seen_le[l] = run; }
    }
    const int p = p0 + threadIdx.x;
    const int v = p < n ? orow[p] : u;
    const bool live = p < n && v != u;
    const int hv = live ? min(hrow[v], kMaxLayers - 1) : -1;
    chunk_layer[threadIdx.x] = hv;
    __syncthreads();
    int in_le = 0, in_eq = 0, in_before = 0;  // live positions of this chunk in front of mine
    for (int t = 0; t < int(threadIdx.x); ++t) {
      const int ht = chunk_layer[t];
      in_before += ht >= 0 ? 1 : 0;
      in_le += (ht >= 0 && ht <= hv) ? 1 : 0;
      in_eq += ht == hv ? 1 : 0;
    }
    if (live) {
      const double nb = double(seen_le[hv] + in_le + 1);
      const double precision = nb / double(passed + in_before + 1);
      const double recall = nb / double(strict_before[hv] + seen[hv] + in_eq + 1);
      const double f1 = 2.0 * precision * recall / (precision + recall);
      if (hv >= 1) {
        atomicAdd(&lm1[hv - 1], f1);
        atomicAdd(&lm2[hv - 1], f1 * f1);
        atomicAdd(&lcnt[hv - 1], 1);
      }
    }
    __syncthreads();  // everyone has read seen / seen_le
    if (live) atomicAdd(&seen[hv], 1);
    int chunk_live = 0;  // block-uniform count of live positions in this chunk
    for (int t = 0; t < kApBlock; ++t) chunk_live += chunk_layer[t] >= 0 ? 1 : 0;
    passed += chunk_live;
  }
  __syncthreads();
  for (int l = threadIdx.x; l < num_layers - 1 && l < kMaxLayers; l += kApBlock) {
    if (lcnt[l] == 0) continue;
    if (per_tree_average) {
      const double mean = lm1[l] / double(lcnt[l]);
      atomicAdd(&m1[l], mean);
      atomicAdd(&m2[l], mean * mean);
      atomicAdd(&counts[l], 1.0);
    } else {
      atomicAdd(&m1[l], lm1[l]);
      atomicAdd(&m2[l], lm2[l]);
      atomicAdd(&counts[l], double(lcnt[l]));
    }
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

extern "C" int mm_graph_layer_f1(const int* order, const int* hops, int64_t n, const int* indptr, int min_degree,
                                 int max_degree, int per_tree_average, int num_layers, double* m1, double* m2,
                                 double* counts, mm_stream_t stream) {
  if (n < 0 || n > (1 << 30) || num_layers < 1 || (n > 0 && (!order || !hops || !indptr || !m1 || !m2 || !counts)))
    return MM_ERR_ARG;
  if (num_layers > 2048) return MM_ERR_UNSUPPORTED;  // graph diameter >= 2048
  if (n == 0) return MM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (num_layers <= 256)
    layer_f1_kernel<256><<<dim3(unsigned(n)), dim3(kApBlock), 0, st>>>(order, hops, int(n), indptr, min_degree,
                                                                   max_degree, per_tree_average, num_layers, m1, m2, counts);
  else
    layer_f1_kernel<2048><<<dim3(unsigned(n)), dim3(kApBlock), 0, st>>>(order, hops, int(n), indptr, min_degree,
                                                                    max_degree, per_tree_average, num_layers, m1, m2, counts);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : int(e);
}
