// Stable argsort of every row of the dense embedding-distance matrix — the "manifold-sorted neighbours" of
// FastPrecision::LayerF1Scores (graphembed/graphembed/pyx/impl/precision.cpp:107-121, SortNodeDists; ties keep
// node order, as the evaluator's walk of tied distances is defined by it).  rocPRIM's segmented radix sort (one segment
// per row, LSD radix = stable) on (distance, node) pairs; the library call replaces the framework sort the evaluator
// used in round 1.  Everything lives in the caller's workspace:
//   [keys_out n^2][values_in n^2 int][offsets n+1 int][rocPRIM temporary storage]
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "../../include/mm_manifolds.h"

namespace mm {
namespace {

__global__ void sort_iota_kernel(int* __restrict__ values, int* __restrict__ offsets, int n) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = int64_t(n) * n;
  if (t < total) values[t] = int(t % n);
  if (t <= n) offsets[t] = int(t * n);
}

inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }

template <typename T> size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  const unsigned total = unsigned(n * n);
  (void)rocprim::segmented_radix_sort_pairs(nullptr, bytes, static_cast<const T*>(nullptr), static_cast<T*>(nullptr),
                                            static_cast<const int*>(nullptr), static_cast<int*>(nullptr), total, unsigned(n),
                                            static_cast<const int*>(nullptr), static_cast<const int*>(nullptr), 0,
                                            unsigned(8 * sizeof(T)), hipStream_t(nullptr));
  return bytes;
}

template <typename T> size_t sort_ws_bytes(int64_t n) {
  const size_t nn = size_t(n) * n;
  return align_up(nn * sizeof(T)) + align_up(nn * sizeof(int)) + align_up((n + 1) * sizeof(int)) + align_up(sort_temp_bytes<T>(n));
}

template <typename T>
int sort_rows(const T* dist, int64_t n, int* order, void* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < sort_ws_bytes<T>(n)) return MM_ERR_ARG;
  const size_t nn = size_t(n) * n;
  char* p = static_cast<char*>(ws);
  T* keys_out = reinterpret_cast<T*>(p); p += align_up(nn * sizeof(T));
  int* values_in = reinterpret_cast<int*>(p); p += align_up(nn * sizeof(int));
  int* offsets = reinterpret_cast<int*>(p); p += align_up((n + 1) * sizeof(int));
  size_t temp = sort_temp_bytes<T>(n);
  sort_iota_kernel<<<dim3(unsigned((nn + 255) / 256)), dim3(256), 0, st>>>(values_in, offsets, int(n));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return int(e);
  e = rocprim::segmented_radix_sort_pairs(p, temp, dist, keys_out, values_in, order, unsigned(nn), unsigned(n), offsets,
                                          offsets + 1, 0, unsigned(8 * sizeof(T)), st);
  return e == hipSuccess ? MM_OK : int(e);
}

}  // namespace
}  // namespace mm

extern "C" size_t mm_graph_sort_rows_ws_bytes(int dtype, int64_t n) {
  if (n <= 0 || n > 46340) return 0;   // n^2 must fit the segmented sort's 32-bit offsets
  return dtype == MM_F64 ? mm::sort_ws_bytes<double>(n) : mm::sort_ws_bytes<float>(n);
}

extern "C" int mm_graph_sort_rows(int dtype, const void* dist, int64_t n, int* order, void* ws, size_t ws_bytes,
                                  mm_stream_t stream) {
  if (!dist || !order || !ws || n <= 0) return MM_ERR_ARG;
  if (n > 46340) return MM_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MM_F32) return mm::sort_rows<float>(static_cast<const float*>(dist), n, order, ws, ws_bytes, st);
  if (dtype == MM_F64) return mm::sort_rows<double>(static_cast<const double*>(dist), n, order, ws, ws_bytes, st);
  return MM_ERR_ARG;
}
