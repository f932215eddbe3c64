// A/B of the two data layouts for the SPD(3) pair arithmetic (forward: d^2 = sum log^2 eig(L_i^-1 X_j L_i^-T)), with the
// SAME mathematics in both — Cholesky-whitened congruence and three fixed cyclic-Jacobi sweeps — so that only the layout
// differs:
//   A  one PAIR PER LANE, the matrix in registers, the row operand wave-uniform (what csrc/spd.hip does);
//   B1 one WAVEFRONT PER MATRIX: nine lanes hold the nine entries, operands staged in LDS, every Jacobi rotation
//      exchanges entries with __shfl and the final sum is a __shfl reduction (the layout BASELINE.json's north_star
//      names), 55 of 64 lanes idle;
//   B2 the same with SEVEN matrices per wavefront (7 x 9 = 63 lanes busy) — the most favourable packing of layout B.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 layout_ab.hip -o layout_ab && ./layout_ab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void rot(float app, float aqq, float apq, float& c, float& s) {
  const float h = aqq - app, ah = fabsf(h) + 1e-15f, sa = h < 0.f ? -apq : apq, sa2 = sa + sa;
  const float r = __builtin_amdgcn_rsqf(fmaf(ah, ah, sa2 * sa2));
  const float x = fmaf(ah * r, 0.5f, 0.5f), ci = __builtin_amdgcn_rsqf(x);
  c = x * ci; s = (sa * r) * ci;
}

// ---- A: one pair per lane --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lane(const float* __restrict__ li /* [n][9] full lower */, const float* __restrict__ x /* [n][9] */,
                                              int n, float* __restrict__ out /* [n][n] */) {
  const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 8;
  if (j >= n) return;
  float X[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) X[k] = x[j * 9 + k];
  for (int i = i0; i < i0 + 8 && i < n; ++i) {
    float L[9], T[9], A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) L[k] = li[i * 9 + k];   // wave-uniform -> scalar loads
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[r * 3 + c] = L[r * 3] * X[c] + L[r * 3 + 1] * X[3 + c] + L[r * 3 + 2] * X[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) A[r * 3 + c] = T[r * 3] * L[c * 3] + T[r * 3 + 1] * L[c * 3 + 1] + T[r * 3 + 2] * L[c * 3 + 2];
#pragma unroll
    for (int sw = 0; sw < 3; ++sw)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = p + 1; q < 3; ++q) {
          float c, s;
          rot(A[p * 3 + p], A[q * 3 + q], A[p * 3 + q], c, s);
#pragma unroll
          for (int r = 0; r < 3; ++r) {   // columns p, q
            const float ap = A[r * 3 + p], aq = A[r * 3 + q];
            A[r * 3 + p] = c * ap - s * aq; A[r * 3 + q] = s * ap + c * aq;
          }
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {   // rows p, q
            const float ap = A[p * 3 + cc], aq = A[q * 3 + cc];
            A[p * 3 + cc] = c * ap - s * aq; A[q * 3 + cc] = s * ap + c * aq;
          }
        }
    const float l0 = __logf(A[0]), l1 = __logf(A[4]), l2 = __logf(A[8]);
    out[size_t(i) * n + j] = l0 * l0 + l1 * l1 + l2 * l2;
  }
}

// ---- B: nine lanes per matrix, G matrices per wavefront -----------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void k_wave(const float* __restrict__ li, const float* __restrict__ x, int n,
                                              float* __restrict__ out, int pairs_per_wave) {
  __shared__ float sL[4][G][9], sX[4][G][9], sT[4][G][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / 9, e = lane % 9, r = e / 3, c = e % 3;
  const bool on = grp < G;
  const int g0 = on ? grp : 0;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  const int base = g0 * 9;   // first lane of this lane's matrix
  for (int t = 0; t < pairs_per_wave; ++t) {
    const long long pid = (wid * pairs_per_wave + t) * G + g0;
    const int i = int(pid / n), j = int(pid % n);
    const bool live = on && i < n;
    if (live) { sL[wave][g0][e] = li[i * 9 + e]; sX[wave][g0][e] = x[j * 9 + e]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float tv = 0.f;
    if (live) tv = sL[wave][g0][r * 3] * sX[wave][g0][c] + sL[wave][g0][r * 3 + 1] * sX[wave][g0][3 + c] + sL[wave][g0][r * 3 + 2] * sX[wave][g0][6 + c];
    if (on) sT[wave][g0][e] = tv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a = 0.f;
    if (live) a = sT[wave][g0][r * 3] * sL[wave][g0][c * 3] + sT[wave][g0][r * 3 + 1] * sL[wave][g0][c * 3 + 1] + sT[wave][g0][r * 3 + 2] * sL[wave][g0][c * 3 + 2];
#pragma unroll
    for (int sw = 0; sw < 3; ++sw)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = p + 1; q < 3; ++q) {
          const float app = __shfl(a, base + p * 3 + p), aqq = __shfl(a, base + q * 3 + q), apq = __shfl(a, base + p * 3 + q);
          float cs, sn;
          rot(app, aqq, apq, cs, sn);
          // columns p, q: entry (r, c) with c in {p, q} needs A[r][p] and A[r][q]
          const float ap = __shfl(a, base + r * 3 + p), aq = __shfl(a, base + r * 3 + q);
          if (c == p) a = cs * ap - sn * aq; else if (c == q) a = sn * ap + cs * aq;
          // rows p, q
          const float bp = __shfl(a, base + p * 3 + c), bq = __shfl(a, base + q * 3 + c);
          if (r == p) a = cs * bp - sn * bq; else if (r == q) a = sn * bp + cs * bq;
        }
    float l = (r == c) ? __logf(a) : 0.f;
    l *= l;
    const float s = __shfl(l, base) + __shfl(l, base + 4) + __shfl(l, base + 8);   // __shfl reduction over the diagonal lanes
    if (live && e == 0) out[size_t(i) * n + j] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

int main() {
  const int n = 2048;
  std::vector<float> hl(n * 9), hx(n * 9);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float a[9];
    for (int k = 0; k < 9; ++k) a[k] = 0.1f * (rand() / float(RAND_MAX) - 0.5f);
    float X[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { X[r * 3 + c] = (r == c); for (int k = 0; k < 3; ++k) X[r * 3 + c] += a[r * 3 + k] * a[c * 3 + k] + 0.05f * (r == c); }
    // Cholesky and its inverse (full 3x3 lower storage)
    float l00 = sqrtf(X[0]), l10 = X[3] / l00, l20 = X[6] / l00, l11 = sqrtf(X[4] - l10 * l10), l21 = (X[7] - l20 * l10) / l11,
          l22 = sqrtf(X[8] - l20 * l20 - l21 * l21);
    float i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22, i10 = -l10 * i00 / l11, i21 = -l21 * i11 / l22, i20 = -(l20 * i00 + l21 * i10) / l22;
    float L[9] = {i00, 0, 0, i10, i11, 0, i20, i21, i22};
    for (int k = 0; k < 9; ++k) { hl[i * 9 + k] = L[k]; hx[i * 9 + k] = X[k]; }
  }
  float *dl, *dx, *oa, *ob;
  hipMalloc(&dl, n * 9 * 4); hipMalloc(&dx, n * 9 * 4); hipMalloc(&oa, size_t(n) * n * 4); hipMalloc(&ob, size_t(n) * n * 4);
  hipMemcpy(dl, hl.data(), n * 9 * 4, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), n * 9 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double pairs = double(n) * n;
  auto time = [&](auto launch, const char* name) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int k = 0; k < 5; ++k) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %9.1f us  %8.2f G pairs/s\n", name, ms * 1e3, pairs / (ms * 1e-3) / 1e9);
    return ms;
  };
  const float ta = time([&] { k_lane<<<dim3(n / 256, n / 8), 256>>>(dl, dx, n, oa); }, "A  one pair per lane (registers, SGPR rows)");
  hipMemset(ob, 0, size_t(n) * n * 4);
  const int ppw = 64;
  const float t1 = time([&] { k_wave<1><<<dim3(unsigned(pairs / (4 * ppw))), 256>>>(dl, dx, n, ob, ppw); }, "B1 one wavefront per matrix (LDS + __shfl)");
  std::vector<float> ha(size_t(n) * n), hb(size_t(n) * n);
  hipMemcpy(ha.data(), oa, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), ob, hb.size() * 4, hipMemcpyDeviceToHost);
  double md = 0; for (size_t k = 0; k < ha.size(); ++k) md = fmax(md, fabs(ha[k] - hb[k]));
  const float t7 = time([&] { k_wave<7><<<dim3(unsigned((pairs + 4 * ppw * 7 - 1) / (4 * ppw * 7))), 256>>>(dl, dx, n, ob, ppw); }, "B2 seven matrices per wavefront (63 lanes)");
  hipMemcpy(hb.data(), ob, hb.size() * 4, hipMemcpyDeviceToHost);
  double md7 = 0; for (size_t k = 0; k < ha.size(); ++k) md7 = fmax(md7, fabs(ha[k] - hb[k]));
  printf("max |d2_A - d2_B1| = %.2e, |d2_A - d2_B2| = %.2e;  B1 / A = %.1fx, B2 / A = %.1fx slower\n", md, md7, t1 / ta, t7 / ta);
  return 0;
}
