// Counter-based random numbers for the thermostats: Philox4x32-10 keyed by the run's seed, counter =
// (particle, step, stream); Box-Muller normals.  Restated on the host by oracle/langevin_oracle.py.
#ifndef MYTHOS_PHILOX_H
#define MYTHOS_PHILOX_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mythos {

// ------------------------------------------------------------------ Philox4x32-10 (counter RNG)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0;
  c[1] = n1;
  c[2] = n2;
  c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// two standard normals from two 32-bit words (Box-Muller); the fp64 build evaluates it in
// double so a host restatement (oracle/langevin_oracle.py) reproduces the stream to round-off
__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, float& z0, float& z1) {
  const float a = (float(u0) + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
  const float b = float(u1) * 2.3283064365386963e-10f;
  const float r = sqrtf(-2.0f * __logf(a));
  const float s = __sinf(6.283185307179586f * b), c = __cosf(6.283185307179586f * b);
  z0 = r * c;
  z1 = r * s;
}
__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, double& z0, double& z1) {
  const double a = (double(u0) + 1.0) * 2.3283064365386963e-10;  // (0, 1]
  const double b = double(u1) * 2.3283064365386963e-10;
  const double r = sqrt(-2.0 * log(a));
  double s, c;
  sincos(6.283185307179586 * b, &s, &c);
  z0 = r * c;
  z1 = r * s;
}
// six normals for (particle, step)
template <typename R>
__device__ __forceinline__ void normals6(uint64_t seed, uint32_t particle, uint64_t step, uint32_t stream, R* z) {
  uint32_t c[4] = {particle, uint32_t(step), uint32_t(step >> 32), stream};
  philox4x32(c, uint32_t(seed), uint32_t(seed >> 32));
  box_muller(c[0], c[1], z[0], z[1]);
  box_muller(c[2], c[3], z[2], z[3]);
  uint32_t d[4] = {particle, uint32_t(step), uint32_t(step >> 32), stream + 1u};
  philox4x32(d, uint32_t(seed), uint32_t(seed >> 32));
  box_muller(d[0], d[1], z[4], z[5]);
}

}  // namespace mythos

#endif  // MYTHOS_PHILOX_H
