// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the SPD pair kernels: a known byte
// count is read (a) 16 B per lane, linear; (b) 4 B per lane, linear, 256-B aligned wave segments; (c) 4 B per lane
// in the backward kernel's pattern — 64 consecutive columns j of consecutive rows i of the row-major upper-triangle
// pair vector, whose 256-B wave segments start at arbitrary 4-B offsets.
//   hipcc -O3 --offload-arch=gfx950 load_bw.hip -o load_bw
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- ./load_bw
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_linear16(const float4* in, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x) {
    const float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_linear4(const float* in, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += in[i];
  if (acc == 12345.678f) sink[0] = acc;
}
__host__ __device__ inline long long poff(long long n, long long r) { return r * (2 * n - r - 1) / 2; }
// one workgroup = 64 columns x 64 rows (4 wavefronts x 16 rows), as spd_pdist_bwd_kernel reads g
__global__ void k_pairs(const float* in, int n, float* sink) {
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
  const int j = ((blockIdx.y * 64 + 1) / 64 + blockIdx.x) * 64 + (threadIdx.x & 63);
  float acc = 0.f;
  if (j < n)
    for (int i = i0; i < i0 + 16 && i < n; ++i)
      if (j > i) acc += in[poff(n, i) + (j - i - 1)];
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int n = 5000; const size_t P = size_t(n) * (n - 1) / 2;   // 49.99 MB: the headline pair vector
  float *in, *sink; hipMalloc(&in, P * sizeof(float) + 64); hipMalloc(&sink, 64);
  hipMemset(in, 0, P * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a); k_linear16<<<4096, 256>>>(reinterpret_cast<const float4*>(in), P / 4, sink); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("linear16: %.1f us  %.2f TB/s  (%zu bytes)\n", ms * 1e3, P * 4.0 / (ms * 1e-3) / 1e12, P / 4 * 16);
    hipEventRecord(a); k_linear4<<<4096, 256>>>(in, P, sink); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("linear4 : %.1f us  %.2f TB/s  (%zu bytes)\n", ms * 1e3, P * 4.0 / (ms * 1e-3) / 1e12, P * 4);
    dim3 grid((n + 63) / 64, (n + 63) / 64);
    hipEventRecord(a); k_pairs<<<grid, 256>>>(in, n, sink); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("pairs   : %.1f us  %.2f TB/s  (%zu bytes)\n", ms * 1e3, P * 4.0 / (ms * 1e-3) / 1e12, P * 4);
  }
  return 0;
}
