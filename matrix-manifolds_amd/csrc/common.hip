// Library info and pair-list geometry helpers (host side; no GPU work).
#include <hip/hip_runtime.h>

#include "../../include/mm_manifolds.h"

extern "C" {

int mm_abi_version(void) { return 1; }

const char* mm_target_arch(void) { return "gfx950"; }

int64_t mm_pair_offset(int64_t n, int64_t row) { return row * (2 * n - row - 1) / 2; }

// Shard r owns the rows whose first pair index falls in [P*r/world, P*(r+1)/world):
// contiguous, disjoint, covering, and balanced to within one row (< n pairs).
static int64_t first_row_at_or_after(int64_t n, int64_t target) {
  int64_t lo = 0, hi = n;  // smallest row with pair_offset(row) >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) / 2;
    if (mm_pair_offset(n, mid) >= target) hi = mid; else lo = mid + 1;
  }
  return lo;
}

int mm_shard_rows(int64_t n, int world, int rank, int64_t* row_begin, int64_t* row_end) {
  if (n < 0 || world <= 0 || rank < 0 || rank >= world || !row_begin || !row_end) return MM_ERR_ARG;
  const int64_t P = n * (n - 1) / 2;
  const __int128 p = P;
  *row_begin = rank == 0 ? 0 : first_row_at_or_after(n, (int64_t)(p * rank / world));
  *row_end = rank == world - 1 ? n : first_row_at_or_after(n, (int64_t)(p * (rank + 1) / world));
  return MM_OK;
}

}  // extern "C"
