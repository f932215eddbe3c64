// Diagnostic clocks of the pair kernels — compiled out of the product.  Variant builds define MM_BWD_STAMP (SPD backward:
// per-workgroup start / end / placement, tools/stamp_timeline.py) or MM_GRAM_STAMP (matrix-core vector kernels: per-wavefront
// phase clocks, tools/gram_timeline.py); the kernels carry one-line hooks that expand to nothing otherwise.
#pragma once
#include <hip/hip_runtime.h>

namespace mm {

__device__ __forceinline__ unsigned long long stamp_hw_id() {   // HW_ID << 32 | XCC_ID
  return (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))) << 32) |
         __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
}

#ifdef MM_BWD_STAMP
__device__ unsigned long long g_bwd_stamps[4 * 16384];
__device__ unsigned long long g_bwd_marks[3 * 16384];   // cycles from the start to the phase marks
#define MM_SPD_STAMP_BEGIN()                                              \
  const unsigned long long stamp0_ = __builtin_amdgcn_s_memrealtime();    \
  const unsigned long long stampc0_ = __builtin_amdgcn_s_memtime();       \
  unsigned long long stampm_[3] = {0, 0, 0}
// phase marks (round 5): 0 = shares cut and first block found, 1 = first row about to start (operands landed), 2 = last row done
// (k = 1 keeps its FIRST value, k = 2 its last); cycles since the start, read through mm_dbg_read_bwd_marks
#define MM_SPD_STAMP_MARK(k)                                                                   \
  do {                                                                                         \
    if ((k) != 1 || stampm_[1] == 0) stampm_[(k)] = __builtin_amdgcn_s_memtime() - stampc0_;   \
  } while (0)
#define MM_SPD_STAMP_END()                                                        \
  do {                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                 \
      unsigned long long* o_ = &g_bwd_stamps[4 * blockIdx.x];                     \
      o_[0] = stamp0_;                                                            \
      o_[1] = __builtin_amdgcn_s_memrealtime();                                   \
      o_[2] = stamp_hw_id();                                                      \
      o_[3] = __builtin_amdgcn_s_memtime() - stampc0_;                            \
      g_bwd_marks[3 * blockIdx.x] = stampm_[0];                                   \
      g_bwd_marks[3 * blockIdx.x + 1] = stampm_[1];                               \
      g_bwd_marks[3 * blockIdx.x + 2] = stampm_[2];                               \
    }                                                                             \
  } while (0)
#else
#define MM_SPD_STAMP_BEGIN() do {} while (0)
#define MM_SPD_STAMP_MARK(k) do {} while (0)
#define MM_SPD_STAMP_END() do {} while (0)
#endif

#ifdef MM_PRODUCT_STAMP
// ordered mixed-manifold pair kernel (product_pairs.hip): 8 slots per workgroup, written by lane 0 of wavefront 0
//   0 start (s_memtime)  1 column data + row staging requested  2 staged rows in LDS, barrier passed  3 row loop done
//   4 flushes done  5 end (s_memtime)  6 start (s_memrealtime, 100 MHz)  7 HW_ID << 32 | XCC_ID
__device__ unsigned long long g_product_stamps[8 * 4096];
#define MM_PSTAMP(k)                                                                                          \
  do {                                                                                                        \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                 \
    if (threadIdx.x == 0 && wg_ < 4096) g_product_stamps[wg_ * 8 + (k)] = __builtin_amdgcn_s_memtime();       \
  } while (0)
#define MM_PSTAMP_BEGIN()                                                                                     \
  do {                                                                                                        \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                 \
    if (threadIdx.x == 0 && wg_ < 4096) {                                                                     \
      g_product_stamps[wg_ * 8 + 0] = __builtin_amdgcn_s_memtime();                                           \
      g_product_stamps[wg_ * 8 + 6] = __builtin_amdgcn_s_memrealtime();                                       \
      g_product_stamps[wg_ * 8 + 7] = stamp_hw_id();                                                          \
    }                                                                                                         \
  } while (0)
#else
#define MM_PSTAMP(k) do {} while (0)
#define MM_PSTAMP_BEGIN() do {} while (0)
#endif

#ifdef MM_GRAM_STAMP
__device__ unsigned long long g_gram_stamps[1024 * 4 * 26];    // symmetric backward: 26 slots per wavefront
__device__ unsigned long long g_gramf_stamps[2048 * 4 * 16];   // forward: 16 slots per wavefront
#define MM_GSTAMP(k)                                                                                                  \
  do {                                                                                                                \
    if (lane == 0 && blockIdx.x < 1024) g_gram_stamps[(blockIdx.x * 4 + wave) * 26 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define MM_GSTAMP_END()                                                           \
  do {                                                                            \
    if (lane == 0 && blockIdx.x < 1024) {                                         \
      unsigned long long* o_ = &g_gram_stamps[(blockIdx.x * 4 + wave) * 26];      \
      o_[23] = __builtin_amdgcn_s_memrealtime();                                  \
      o_[24] = stamp_hw_id();                                                     \
    }                                                                             \
  } while (0)
#define MM_FSTAMP(k)                                                                                               \
  do {                                                                                                             \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                      \
    if (lane == 0 && wg_ < 2048) g_gramf_stamps[(wg_ * 4 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#define MM_FSTAMP_END()                                                                                            \
  do {                                                                                                             \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                      \
    if (lane == 0 && wg_ < 2048) g_gramf_stamps[(wg_ * 4 + wave) * 16 + 15] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#define MM_STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")      // a phase ends when its loads have landed
#define MM_STAMP_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define MM_STAMP_PIN(v) asm volatile("" : "+v"(v))                              // ... or when a value exists
#else
#define MM_GSTAMP(k) do {} while (0)
#define MM_GSTAMP_END() do {} while (0)
#define MM_FSTAMP(k) do {} while (0)
#define MM_FSTAMP_END() do {} while (0)
#define MM_STAMP_WAIT_VM() do {} while (0)
#define MM_STAMP_WAIT_LGKM() do {} while (0)
#define MM_STAMP_PIN(v) do {} while (0)
#endif

}  // namespace mm
