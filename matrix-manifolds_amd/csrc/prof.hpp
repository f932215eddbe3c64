// Optional live kernel timing with HIP events on the launch stream.
// Off by default (zero cost: one predictable branch per launch).  bench.py turns
// it on to obtain the dominant kernel's average duration inside the timed region
// (the number the committed rocprofv3 summary must agree with).
#pragma once
#include <hip/hip_runtime.h>

namespace mm {
enum ProfId { PROF_SPD_FWD = 0, PROF_SPD_BWD = 1, PROF_VEC_FWD = 2, PROF_VEC_BWD = 3, PROF_COUNT = 4 };
bool prof_on();
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
struct ProfScope {
  int id; hipStream_t st; bool on;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on(prof_on()) { if (on) prof_begin(id, st); }
  ~ProfScope() { if (on) prof_end(id, st); }
};
}  // namespace mm
