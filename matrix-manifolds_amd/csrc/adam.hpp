// Shared pieces of the fused Riemannian Adam kernels (vec.hip, spd.hip) — graphembed/optim/radam.py:62-98.
#pragma once
#include <hip/hip_runtime.h>

namespace mm {

template <typename T> struct AdamArgs {
  T lr, beta1, beta2, eps, max_grad_norm;  // max_grad_norm <= 0: no clipping
  int nc, exact;
  double* step;      // device scalar t >= 1 (state['step']); advanced by the kernel
  unsigned* ticket;  // device counter, zero between launches
};

// beta2 of this step (AdamNc: 1 - 1/t, radam.py:81-82) and the bias-corrected step size
// alpha = lr sqrt(1 - beta2^t) / (1 - beta1^t) (radam.py:88), from the step counter in device memory.
template <typename T> __device__ __forceinline__ void adam_coeffs(const AdamArgs<T>& a, T& beta2, T& alpha) {
  const double t = *a.step;
  const double b2 = a.nc ? 1.0 - 1.0 / t : double(a.beta2);
  alpha = T(double(a.lr) * ::sqrt(1.0 - ::pow(b2, t)) / (1.0 - ::pow(double(a.beta1), t)));
  beta2 = T(b2);
}

// Called by every block as its last action: the block that arrives last — every other block has read the
// counter by then — advances it and re-arms the ticket.  (No separate `step += 1` launch, and replaying a
// captured graph keeps counting.)
// `total` = number of blocks that work on this parameter (and call this).
__device__ __forceinline__ void adam_tick(double* step, unsigned* ticket, unsigned total) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1u) == total - 1) {
      *ticket = 0;
      *step += 1.0;
    }
  }
}

}  // namespace mm
