// fp64 form of the matrix-core backward (included by vec_gram.hip inside namespace mm):
// v_mfma_f64_16x16x4_f64, 16 x 16 tiles, m <= 16.
//
// Same construction as the fp32 kernel with the layouts of the f64 instruction: A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], C/D[row = (l >> 4) + 4 q][col = l & 15].  The accumulator of the Gram again IS
// the A operand of W^T X_I: lane (j, h) holds W[h + 4 q][j] in register q = A^T[j][k = h] of k-step q.
// (graphembed/run.py:32-35 forces float64: this is the path a drop-in run of the reference's scripts takes.)
#pragma once

template <int KIND, int KS, int LOSS>  // KS = ceil(m / 4) rounded up to a dispatch class
__global__ __launch_bounds__(64 * kGramBwdWaves) void vec_gram_bwd_f64_kernel(const double* __restrict__ x,
                                                                            const double* __restrict__ g, int n, int m,
                                                                            int row_begin, int row_end, int squared,
                                                                            int tiles_per_wave, double* __restrict__ grad,
                                                                            LossArgs<double> la) {
  __shared__ double sT[kGramBwdWaves][16][17];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, h = lane >> 4;
  const int nT = (n + 15) / 16;
  const int jb = blockIdx.x, J = jb * 16;
  using u32 = unsigned int;
  const int rc = r < m ? r : m - 1;
  double bJ[KS];  // B operand of the Gram: x[J + r][4 s + h]
  {
    const int jr = J + r;
    const u32 xb = u32(jr < n ? jr : n - 1) * u32(m);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + h;
      bJ[s] = x[xb + u32(k < m ? k : m - 1)];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {   // (pinned, then masked: `valid ? x[..] : 0` would serialise the loads)
      asm volatile("" : "+v"(bJ[s]));
      bJ[s] = (jr < n && 4 * s + h < m) ? bJ[s] : 0.0;
    }
  }
  f64x4 accJ = {0.0, 0.0, 0.0, 0.0};
  double sp = 1.0, loss_acc = 0.0, ds_acc = 0.0;
  loss_resolve<double, LOSS>(la);
  if constexpr (LOSS != MM_LOSS_NONE) sp = softplus_of(la.scale_raw);
  const double kInvalid = LOSS != MM_LOSS_NONE ? __builtin_nan("") : 0.0;
  const u32 base = u32(gpair_off(n, row_begin));
  const u32 gmax = u32(gpair_off(n, row_end)) - base - 1u;
  auto goff = [&](int lo, int hi) -> u32 {  // pair (lo < hi) -> offset in this shard's slice, clamped
    const u32 o = u32(lo) * u32(2 * n - lo - 1) / 2u - base + u32(hi - lo - 1);
    return o > gmax ? gmax : o;
  };
  const int j = J + r;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int ib = (blockIdx.y * tiles_per_wave + t) * kGramBwdWaves + wave;
    if (ib >= nT) break;  // wave-uniform
    const int I = ib * 16;
    // ---- all loads of the tile first (clamped addresses, masked at use)
    double xa[KS], bI[4], gr[4];
    {
      const int ia = I + r;
      const u32 xo = u32(ia < n ? ia : n - 1) * u32(m);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + h;
        xa[s] = x[xo + u32(k < m ? k : m - 1)];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int irow = I + h + 4 * q;
      bI[q] = x[u32(irow < n ? irow : n - 1) * u32(m) + u32(rc)];
    }
    const bool upper = ib < jb;
    if (ib != jb) {
      // above the diagonal: pair (i, j) in row i, contiguous in j -> accumulator layout directly; below: pair
      // (j', i) in row j', contiguous in i -> loaded transposed (lane = i, register = j' = J + h + 4 q)
      const int A0 = upper ? I : J, B0 = upper ? J : I;
      const int col = B0 + r < n ? B0 + r : n - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = A0 + h + 4 * q;
        gr[q] = g[goff(row < n - 1 ? row : n - 2, col > row ? col : row + 1)];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = I + h + 4 * q;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        gr[q] = g[goff(lo < n - 1 ? lo : n - 2, hi < n ? (hi > lo ? hi : lo + 1) : n - 1)];
      }
    }
    double gv[4];
    if (ib != jb) {
      const int A0 = upper ? I : J;
      const int col = (upper ? J : I) + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = A0 + h + 4 * q;
        const bool valid = col < n && row < n && row >= row_begin && row < row_end;  // row < col always
        gv[q] = valid ? gr[q] : kInvalid;
      }
      if (!upper) {  // transposed load -> accumulator layout through LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[wave][h + 4 * q][r] = gv[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = sT[wave][r][h + 4 * q];
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = I + h + 4 * q;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const bool valid = i != j && hi < n && lo >= row_begin && lo < row_end;
        gv[q] = valid ? gr[q] : kInvalid;
      }
    }
    // 1. Gram tile
    f64x4 qv = {0.0, 0.0, 0.0, 0.0};
    {
      const bool ia_ok = I + r < n;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + h;
        double a = (ia_ok && k < m) ? xa[s] : 0.0;
        if (KIND == MM_LORENTZ && k != 0) a = -a;
        qv = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bJ[s], qv, 0, 0, 0);
      }
    }
    // 2.-3. w on the accumulator registers; ACC_J += W^T X_I
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double w;
      if constexpr (LOSS == MM_LOSS_NONE) {
        w = gv[q] * PairFn<double, KIND>::dq(qv[q], squared);
      } else {
        const bool ok = gv[q] == gv[q];
        const double d2 = PairFn<double, KIND>::value(qv[q], 1);
        double dldm;
        const double l = loss_term<double, LOSS>(sp * d2, gv[q], la, dldm);
        const bool once = ok && (ib < jb || (ib == jb && I + h + 4 * q < j));  // each unordered pair once
        loss_acc += once ? l : 0.0;
        ds_acc += once ? dldm * d2 : 0.0;
        w = ok ? dldm * sp * PairFn<double, KIND>::dq(qv[q], 1) : 0.0;
      }
      const double b = (I + h + 4 * q < n && r < m) ? bI[q] : 0.0;
      accJ = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b, accJ, 0, 0, 0);
    }
  }
  // combine the workgroup's wavefronts and flush once: accJ[q] = ACC[j = h + 4 q][c = r]
#pragma unroll
  for (int q = 0; q < 4; ++q) sT[wave][h + 4 * q][r] = accJ[q];
  __shared__ double lossW[kGramBwdWaves][2];
  if constexpr (LOSS != MM_LOSS_NONE) {
    const double l = wave_sum(loss_acc), d = wave_sum(ds_acc);
    if (lane == 0) { lossW[wave][0] = l; lossW[wave][1] = d; }
  }
  __syncthreads();
  if constexpr (LOSS != MM_LOSS_NONE) {
    if (threadIdx.x == 0) {
      double l = 0.0, d = 0.0;
#pragma unroll
      for (int w = 0; w < kGramBwdWaves; ++w) { l += lossW[w][0]; d += lossW[w][1]; }
      const int slot = (blockIdx.x + blockIdx.y * gridDim.x) & (kLossSlots - 1);
      atomic_add(&la.slots[slot], l);
      atomic_add(&la.slots[kLossSlots + slot], d);
    }
  }
  {
    const int e = threadIdx.x;  // 256 threads = the 16 x 16 entries of the column block
    const int jj = e >> 4, c = e & 15;
    if (c < m && J + jj < n) {
      double sum = sT[0][jj][c];
#pragma unroll
      for (int w = 1; w < kGramBwdWaves; ++w) sum += sT[w][jj][c];
      if (KIND == MM_LORENTZ && c != 0) sum = -sum;  // d q / d x_j = -J x_i
      atomic_add(&grad[size_t(J + jj) * m + c], sum);
    }
  }
}
