// Optional live kernel timing with HIP events on the launch stream.
// Off by default (zero cost: one predictable branch per launch).  bench.py turns
// it on to obtain the dominant kernel's average duration inside the timed region
// (the number the committed rocprofv3 summary must agree with).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace mm {
enum ProfId { PROF_SPD_FWD = 0, PROF_SPD_BWD = 1, PROF_VEC_FWD = 2, PROF_VEC_BWD = 3, PROF_COUNT = 4 };
bool prof_on();
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
// Events for one launch of kernel `id`, attached to the kernel's own dispatch by hipExtLaunchKernelGGL: their elapsed time is
// the kernel's execution time as the profiler sees it.  (Events recorded on the stream before and after a launch read
// 2 - 3 us more: command-processor time between the event packets and the dispatch.)
void prof_span(int id, hipEvent_t* start, hipEvent_t* stop);
// `kernel<<<grid, block, 0, st>>>(args...)`, timed when profiling is on
template <typename... Args, typename F = void (*)(Args...)>
inline void launch_timed(int id, F kernel, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (prof_on() && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {   // (never inside a graph)
    hipEvent_t a, b;
    prof_span(id, &a, &b);
    hipExtLaunchKernelGGL(kernel, grid, block, 0, st, a, b, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
  }
}
struct ProfScope {
  int id; hipStream_t st; bool on;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), on(prof_on()) { if (on) prof_begin(id, st); }
  ~ProfScope() { if (on) prof_end(id, st); }
};
}  // namespace mm
