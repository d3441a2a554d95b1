// Host stand-in for <hip/hip_runtime.h>, used ONLY by oracle/cpu_port (the CPU baseline / sanitizer build of the
// pair-physics templates in mythos_amd/csrc/oxdna_math.h, oxdna_pair.h and philox.h).  TEST INFRASTRUCTURE: nothing
// under mythos_amd/ sees this file; the product is compiled by hipcc against the real header.
#pragma once
#include <cmath>
#include <cstdint>

#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// the device fast-math intrinsics (glibc declares functions of these names itself: map them by macro)
#define __expf expf
#define __logf logf
#define __sinf sinf
#define __cosf cosf
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
// (named by a branch of the templates that only the GPU's dU/d(sequence distribution) instantiation compiles)
static inline double atomicAdd(double* p, double v) {
  const double old = *p;
  *p += v;
  return old;
}
