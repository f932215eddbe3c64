// Are workgroup-scope float atomics executed in the XCD-local L2 (and much cheaper than agent-scope
// ones, which go to the memory side on a multi-XCD part)?  Every workgroup adds into accumulators that
// are private to ITS XCD (index from HW_REG_XCC_ID), so L2-local atomicity is sufficient; the copies
// are summed afterwards.  Checks the totals and times both scopes.
//   hipcc -O3 --offload-arch=gfx950 xcd_atomic.hip -o xcd_atomic && ./xcd_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}

template <int SCOPE>  // 0: agent scope, one copy; 1: workgroup scope, XCD-private copies
__global__ void k_atomic(float* acc, int ncols, int reps, int* xcd_hist) {
  const int x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xcd_hist[x & 15], 1);
  const int col = (blockIdx.x * 64 + (threadIdx.x & 63)) % ncols;
  float* base = SCOPE ? acc + size_t(x) * 9 * ncols : acc;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float* p = base + size_t(k) * ncols + col;
      if (SCOPE) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  const int ncols = 5056, nblocks = 6400, reps = 4, ncopies = 16;
  float* acc; int* hist;
  hipMalloc(&acc, sizeof(float) * ncopies * 9 * ncols);
  hipMalloc(&hist, sizeof(int) * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int scope = 0; scope < 2; ++scope) {
    for (int it = 0; it < 3; ++it) {
      hipMemset(acc, 0, sizeof(float) * ncopies * 9 * ncols);
      hipMemset(hist, 0, sizeof(int) * 16);
      hipDeviceSynchronize();
      hipEventRecord(a);
      if (scope) k_atomic<1><<<nblocks, 256>>>(acc, ncols, reps, hist);
      else k_atomic<0><<<nblocks, 256>>>(acc, ncols, reps, hist);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<float> h(size_t(ncopies) * 9 * ncols);
      std::vector<int> hh(16);
      hipMemcpy(h.data(), acc, h.size() * sizeof(float), hipMemcpyDeviceToHost);
      hipMemcpy(hh.data(), hist, 64, hipMemcpyDeviceToHost);
      double tot = 0; for (float v : h) tot += v;
      const double expect = double(nblocks) * 256 * reps * 9;
      printf("scope=%s  %.1f us  total %.0f expected %.0f %s  xcd hist:", scope ? "workgroup/XCD-private" : "agent", ms * 1e3, tot, expect,
             tot == expect ? "OK" : "MISMATCH");
      for (int i = 0; i < 16; ++i) printf(" %d", hh[i]);
      printf("\n");
    }
  }
  return 0;
}
