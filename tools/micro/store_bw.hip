// Achievable HBM write bandwidth for (a) a linear stream and (b) the row-major upper-triangle pair vector
// written the way the forward kernels do (a workgroup = 256 consecutive columns j of rows i0..i0+7).
//   hipcc -O3 --offload-arch=gfx950 store_bw.hip -o store_bw && ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_linear(float* out, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = float(i);
}
__host__ __device__ inline long long poff(long long n, long long r) { return r * (2 * n - r - 1) / 2; }
__global__ void k_pairs(float* out, int n, int TI) {
  const int i0 = blockIdx.y * TI;
  const int j = ((i0 + 1) / 256 + blockIdx.x) * 256 + threadIdx.x;
  if (j >= n) return;
  for (int i = i0; i < i0 + TI && i < n; ++i)
    if (j > i) out[poff(n, i) + (j - i - 1)] = float(j);
}

int main() {
  const int n = 5000; const size_t P = size_t(n) * (n - 1) / 2;
  float* out; hipMalloc(&out, P * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a); k_linear<<<4096, 256>>>(out, P); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("linear : %.1f us  %.2f TB/s\n", ms * 1e3, P * 4.0 / (ms * 1e-3) / 1e12);
    for (int TI : {8, 32}) {
      dim3 grid((n + 255) / 256, (n + TI - 1) / TI);
      hipEventRecord(a); k_pairs<<<grid, 256>>>(out, n, TI); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      printf("pairs TI=%d: %.1f us  %.2f TB/s\n", TI, ms * 1e3, P * 4.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
