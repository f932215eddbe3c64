// Library info and pair-list geometry helpers (host side; no GPU work).
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/mm_manifolds.h"
#include "prof.hpp"

namespace mm {
namespace {
struct Span { hipEvent_t a, b; };
bool g_on = false;
std::vector<Span> g_spans[PROF_COUNT];
std::vector<Span> g_pool;
hipEvent_t g_open[PROF_COUNT];
}  // namespace
bool prof_on() { return g_on; }
void prof_begin(int id, hipStream_t st) {
  Span s;
  if (!g_pool.empty()) { s = g_pool.back(); g_pool.pop_back(); }
  else { (void)hipEventCreate(&s.a); (void)hipEventCreate(&s.b); }
  (void)hipEventRecord(s.a, st);
  g_spans[id].push_back(s);
}
void prof_end(int id, hipStream_t st) { (void)hipEventRecord(g_spans[id].back().b, st); }
void prof_span(int id, hipEvent_t* start, hipEvent_t* stop) {
  Span s;
  if (!g_pool.empty()) { s = g_pool.back(); g_pool.pop_back(); }
  else { (void)hipEventCreate(&s.a); (void)hipEventCreate(&s.b); }
  g_spans[id].push_back(s);
  *start = s.a; *stop = s.b;
}
}  // namespace mm

extern "C" {

int mm_abi_version(void) { return 2; }   // 2: mm_comm_*, sharded mm_train_step (row range + communicator), MM_OPT_NONE

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

int mm_prof_enable(int on) { mm::g_on = on != 0; return MM_OK; }

// Synchronises the recorded events; returns launches and total milliseconds of
// kernel `id` since the last collect, then clears them.
int mm_prof_collect(int id, int64_t* launches, double* total_ms) {
  if (id < 0 || id >= mm::PROF_COUNT || !launches || !total_ms) return MM_ERR_ARG;
  *launches = 0; *total_ms = 0.0;
  for (auto& s : mm::g_spans[id]) {
    hipError_t e = hipEventSynchronize(s.b);
    if (e != hipSuccess) return int(e);
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, s.a, s.b);
    if (e != hipSuccess) return int(e);
    *total_ms += ms; ++*launches;
    mm::g_pool.push_back(s);
  }
  mm::g_spans[id].clear();
  return MM_OK;
}

}  // extern "C"
