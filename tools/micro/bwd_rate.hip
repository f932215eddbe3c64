// Issue rate of the SPD(3) fp32 backward's row arithmetic WITHOUT its memory streams (round 5): the per-row work of
// spd_pdist_bwd_kernel (csrc/spd_pair.hpp) — congruence, close-pair gate, ring-Horner logarithm, column-side congruence into
// register accumulators, transposing row reduction — in a loop on register data, row operands through scalar loads of a
// 64-row table as in the kernel, two columns per lane, 256-thread workgroups at 1 .. 4 wavefronts per SIMD on every CU.
// PHASES is a bit mask (1 congruence+gate, 2 logarithm, 4 column side, 8 row reduction, 16 LDS store of the reduced value).
// Prints cycles per loop iteration and per wavefront (s_memtime) and the wall time; tools/devasm.sh on the binary's code
// object gives the instruction count of the loop body, hence cycles per instruction.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 tools/micro/bwd_rate.hip -o /tmp/bwd_rate && /tmp/bwd_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../include/mm_manifolds.h"
#include "../../matrix-manifolds_amd/csrc/smallmat.hpp"
#include "../../matrix-manifolds_amd/csrc/spd_ws.hpp"
#include "../../matrix-manifolds_amd/csrc/spd_pair.hpp"

using namespace mm;

#ifndef RATE_D   // (-DRATE_D=4: the two-column SPD(4) backward's row arithmetic; 1 .. 4 wavefronts per SIMD)
#define RATE_D 3
#endif
constexpr int kNP = RATE_D * (RATE_D + 1) / 2;
template <int PHASES, int WAVES>
__global__ __launch_bounds__(256, WAVES) void rate_kernel(const float* __restrict__ rowtab /* [64][2 NP] */, const float* __restrict__ col,
                                                          int iters, float* __restrict__ out, long long* __restrict__ cyc) {
  constexpr int D = RATE_D, NP = D * (D + 1) / 2, NC = 2;
  __shared__ float red[4][16][NP];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float xj[NC][NP], accJ[NC][D][D], gsum = 0.f;
#pragma unroll
  for (int q = 0; q < NC; ++q) {
#pragma unroll
    for (int k = 0; k < NP; ++k) xj[q][k] = col[(size_t(blockIdx.x) * 256 + threadIdx.x) * NP * NC + q * NP + k];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) accJ[q][r][c] = 0.f;
  }
  bool writer;
  const int slot = reduce_slot<NP>(lane, writer);
  float* rp = &red[wave][0][writer ? slot : 0];
  const long long t0 = __builtin_readcyclecounter();
  unsigned roff = 0;
  for (int it = 0; it < iters; ++it) {
    roff = roff + unsigned(8 * kNP) >= 64u * unsigned(8 * kNP) ? 0u : roff + unsigned(8 * kNP);
    asm volatile("" : "+s"(roff));
    const float* rowp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rowtab) + roff);
    float li[NP], lc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { li[k] = rowp[k]; lc[k] = rowp[NP + k]; }
    float m[NC][NP];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      float a[NP];
      if constexpr (PHASES & 1) {
        congr_chol<float, D>(li, xj[q], a);
        gsum += close_gate<float, D>(a) <= float(kCloseGate) ? 0.f : 1.f;
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) a[k] = xj[q][k] + li[k];
      }
      if constexpr (PHASES & 2) {
        log_close<float, D>(a, m[q], a[0] + a[0]);
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) m[q][k] = a[k];
      }
      if constexpr (PHASES & 4) {
        lt_m_lt_acc<float, D>(li, lc, m[q], accJ[q]);
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) accJ[q][0][0] += m[q][k];
      }
#pragma unroll
      for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) asm volatile("" : "+v"(accJ[q][r][c]));
    }
    if constexpr (PHASES & 8) {
#pragma unroll
      for (int k = 0; k < NP; ++k) m[0][k] += m[1][k];
      const float tot = wave_reduce_transposed<NP, float>(m[0], lane);
      if constexpr (PHASES & 16) rp[(it & 15) * NP] = tot; else gsum += tot;
    } else {
      gsum += m[0][0] + m[1][0];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = gsum;
#pragma unroll
  for (int q = 0; q < NC; ++q)
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) s += accJ[q][r][c];
  if constexpr (PHASES & 16) s += red[wave][lane & 15][lane % NP];
  out[size_t(blockIdx.x) * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PHASES, int WAVES> void run(const char* name, const float* rowtab, const float* col, float* out, long long* cyc, int cus) {
  const int iters = 4000, blocks = cus * WAVES;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) rate_kernel<PHASES, WAVES><<<blocks, 256>>>(rowtab, col, iters, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(a);
  rate_kernel<PHASES, WAVES><<<blocks, 256>>>(rowtab, col, iters, out, cyc);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(size_t(blocks) * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : h) mean += double(v);
  mean /= double(h.size());
  // a SIMD hosts WAVES wavefronts: cycles per iteration PER SIMD = wave cycles / iters / WAVES
  printf("%-34s waves/SIMD %d: %8.3f ms; %8.1f clk per iteration and wavefront -> %7.1f clk of its SIMD per iteration (128 pairs)\n", name,
         WAVES, ms, mean / iters, mean / iters / WAVES);
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  constexpr int NP = kNP;
  std::vector<float> rt(64 * 2 * NP), cl(size_t(cus) * 4 * 256 * 2 * NP);
  auto fill = [&](float* l, float e, float sgn) {   // packed lower triangle of a matrix near the identity (||log X|| ~ 0.1)
    int k = 0;
    for (int r = 0; r < RATE_D; ++r)
      for (int c = 0; c <= r; ++c, ++k) l[k] = c == r ? 1.f + sgn * e * float(1 + (r & 1) - (r >> 1)) : 0.01f * float((r + 2 * c) % 3) - 0.01f;
  };
  for (int r = 0; r < 64; ++r) {
    float l[NP];
    fill(l, 0.01f * float(r % 7), 1.f);
    for (int k = 0; k < NP; ++k) { rt[r * 2 * NP + k] = l[k]; rt[r * 2 * NP + NP + k] = l[k]; }
  }
  for (size_t t = 0; t < cl.size() / NP; ++t) {
    float l[NP];
    fill(l, 0.002f * float(t % 11), -1.f);
    for (int k = 0; k < NP; ++k) cl[t * NP + k] = l[k];
  }
  float *drt, *dcl, *dout;
  long long* dcyc;
  hipMalloc(&drt, rt.size() * 4); hipMalloc(&dcl, cl.size() * 4); hipMalloc(&dout, size_t(cus) * 4 * 256 * 4); hipMalloc(&dcyc, size_t(cus) * 4 * 4 * 8);
  hipMemcpy(drt, rt.data(), rt.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dcl, cl.data(), cl.size() * 4, hipMemcpyHostToDevice);
#define RUN(P, NAME) run<P, 1>(NAME, drt, dcl, dout, dcyc, cus); run<P, 2>(NAME, drt, dcl, dout, dcyc, cus); run<P, 3>(NAME, drt, dcl, dout, dcyc, cus); run<P, 4>(NAME, drt, dcl, dout, dcyc, cus);
  RUN(31, "all phases + LDS store")
  RUN(15, "all phases")
  RUN(7, "no row reduction")
  RUN(11, "no column side")
  RUN(13, "no logarithm")
  RUN(14, "no congruence / gate")
  RUN(0, "loop + scalar loads only")
  return 0;
}
