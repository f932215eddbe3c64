// What the misaligned row segments of the triangular pair vector cost, and what the proposed cures would buy — the access
// patterns of the fp32 SPD(3) pair kernels (two columns per lane) without their arithmetic.
// Row i of the pair vector starts at element pair_off(n, i) — an arbitrary 4-byte offset — and a wavefront touches the 128
// consecutive elements (i, jb .. jb + 127) of it: 512 bytes that straddle FIVE 128-byte lines instead of four.
//   split   : lane l handles columns jb + l and jb + 64 + l, one dword access each (the kernels as they are: two 256-byte
//             spans per row, three lines each)
//   pair    : lane l handles columns jb + 2l, jb + 2l + 1 with ONE 8-byte access (4-byte aligned every other row)
//   aligned : the same rows cut at 512-byte boundaries of the ADDRESS space instead of at column boundaries — what a
//             kernel would reach that rotated its values across lanes / wavefronts so that every access is line-aligned
//             (upper bound of that cure; head and tail of a row segment masked)
// Loads walk 16 rows per wavefront down a 128-column block (the backward: 4 wavefronts x 16 rows per workgroup);
// stores write 8 rows of 512 columns per workgroup (the forward).
//   hipcc -O3 --offload-arch=gfx950 pair_align.hip -o pair_align && ./pair_align
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- ./pair_align      (separate pass: WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>

__host__ __device__ inline long long poff(long long n, long long r) { return r * (2 * n - r - 1) / 2; }

// ---- loads --------------------------------------------------------------------------------------------------------
template <int MODE> __global__ void k_load(const float* __restrict__ in, int n, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = blockIdx.y * 64 + wave * 16;
  const int jb = ((blockIdx.y * 64 + 1) / 128 + blockIdx.x) * 128;
  float acc = 0.f;
  for (int i = i0; i < i0 + 16 && i < n - 1; ++i) {
    const long long row = poff(n, i) - i - 1;   // element (i, j) lives at row + j
    if (MODE == 0) {
      const int j0 = jb + lane, j1 = jb + 64 + lane;
      if (j0 > i && j0 < n) acc += in[row + j0];
      if (j1 > i && j1 < n) acc += in[row + j1];
    } else if (MODE == 1) {
      const int j0 = jb + 2 * lane;
      if (j0 > i && j0 + 1 < n) { const float2 v = *reinterpret_cast<const float2*>(in + row + j0); acc += v.x + v.y; }
      else { if (j0 > i && j0 < n) acc += in[row + j0]; if (j0 + 1 > i && j0 + 1 < n) acc += in[row + j0 + 1]; }
    } else {
      // chunk k of row i = the 128 elements from the 512-byte boundary at or below the row's first element, + 128 k; this
      // wavefront takes the chunk whose index is its column block's (same work distribution as the other two modes)
      const long long a0 = (row + i + 1) & ~127LL;
      const long long e0 = a0 + 128LL * (jb / 128 - (i + 1) / 128) + 2 * lane;
      const long long lo = row + i + 1, hi = row + n;
      if (e0 >= lo && e0 + 1 < hi) { const float2 v = *reinterpret_cast<const float2*>(in + e0); acc += v.x + v.y; }
      else { if (e0 >= lo && e0 < hi) acc += in[e0]; if (e0 + 1 >= lo && e0 + 1 < hi) acc += in[e0 + 1]; }
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

// ---- stores -------------------------------------------------------------------------------------------------------
template <int MODE> __global__ void k_store(float* __restrict__ out, int n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = blockIdx.y * 8;
  const int jb = ((i0 + 1) / 512 + blockIdx.x) * 512 + wave * 128;
  for (int i = i0; i < i0 + 8 && i < n - 1; ++i) {
    const long long row = poff(n, i) - i - 1;
    if (MODE == 0) {
      const int j0 = jb + lane, j1 = jb + 64 + lane;
      if (j0 > i && j0 < n) out[row + j0] = float(j0);
      if (j1 > i && j1 < n) out[row + j1] = float(j1);
    } else if (MODE == 1) {
      const int j0 = jb + 2 * lane;
      if (j0 > i && j0 + 1 < n) *reinterpret_cast<float2*>(out + row + j0) = make_float2(float(j0), float(j0 + 1));
      else { if (j0 > i && j0 < n) out[row + j0] = float(j0); if (j0 + 1 > i && j0 + 1 < n) out[row + j0 + 1] = float(j0 + 1); }
    } else {
      const long long a0 = (row + i + 1) & ~127LL;
      const long long e0 = a0 + 128LL * (jb / 128 - (i + 1) / 128) + 2 * lane;
      const long long lo = row + i + 1, hi = row + n;
      if (e0 >= lo && e0 + 1 < hi) *reinterpret_cast<float2*>(out + e0) = make_float2(1.f, 2.f);
      else { if (e0 >= lo && e0 < hi) out[e0] = 1.f; if (e0 + 1 >= lo && e0 + 1 < hi) out[e0 + 1] = 2.f; }
    }
  }
}

template <typename F> float time_us(F&& launch, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int n = 5000; const size_t P = size_t(n) * (n - 1) / 2;
  float *buf, *sink; hipMalloc(&buf, P * sizeof(float) + 4096); hipMalloc(&sink, 64);
  hipMemset(buf, 0, P * sizeof(float) + 4096);
  const dim3 lgrid((n + 127) / 128, (n + 63) / 64), sgrid((n + 511) / 512, (n + 7) / 8);
  const char* names[3] = {"split (two dword accesses, columns l and 64 + l)", "pair (one 8-byte access, columns 2l and 2l + 1)",
                          "aligned (512-byte boundaries of the address space: upper bound of a rotation)"};
  const double mb = P * 4.0 / 1e6;
  float t;
  t = time_us([&] { k_load<0><<<lgrid, 256>>>(buf, n, sink); }); printf("load  %-82s %6.1f us  %.2f TB/s\n", names[0], t, mb / t);
  t = time_us([&] { k_load<1><<<lgrid, 256>>>(buf, n, sink); }); printf("load  %-82s %6.1f us  %.2f TB/s\n", names[1], t, mb / t);
  t = time_us([&] { k_load<2><<<lgrid, 256>>>(buf, n, sink); }); printf("load  %-82s %6.1f us  %.2f TB/s\n", names[2], t, mb / t);
  t = time_us([&] { k_store<0><<<sgrid, 256>>>(buf, n); }); printf("store %-82s %6.1f us  %.2f TB/s\n", names[0], t, mb / t);
  t = time_us([&] { k_store<1><<<sgrid, 256>>>(buf, n); }); printf("store %-82s %6.1f us  %.2f TB/s\n", names[1], t, mb / t);
  t = time_us([&] { k_store<2><<<sgrid, 256>>>(buf, n); }); printf("store %-82s %6.1f us  %.2f TB/s\n", names[2], t, mb / t);
  return 0;
}
