// Rigid-body Langevin MD for oxDNA: one fused kernel per time step.
//
// Replaces the hot loop of the reference, jax.lax.scan(step_fn) with
// step_fn = jax_md.simulate.nvt_langevin on RigidBody states
// (mythos/simulators/jax_md/jaxmd.py:73-94).  jax_md (third party, not in the reference tree)
// advances one step as  B(dt/2) A(dt/2) O(dt) A(dt/2) [force] B(dt/2):
//   B  p += h F,  Pi += h F_q            (F_q = -dU/dq, quaternion conjugate momentum Pi)
//   A  x += h p/m, free rotor by the NO_SQUISH splitting R3(h/2) R2(h/2) R1(h) R2(h/2) R3(h/2)
//   O  p = c1 p + c2 sqrt(m) xi,  body angular momentum L = c1 L + c2 sqrt(I) xi,
//      c1 = exp(-gamma dt), c2 = sqrt(kT (1 - c1^2))
// Here the rotational state is the body-frame angular momentum L_k = 1/2 (P_k q).Pi, for which
// the kick is the body torque and the free rotor is a rotation about a principal axis; the two
// forms are the same map for a unit quaternion.
//
// Fusion: the kernel that evaluates F(x_k) first closes step k-1 (second half kick), optionally
// emits the snapshot / energies of x_k, then opens step k (half kick, A, O, A) and writes
// x_{k+1} to the other buffer of a ping-pong pair (other workgroups are still reading x_k).
// One launch per MD step; a run of K steps costs K+1 force evaluations.
//
// Per nucleotide per step (fp32): read + write {pos 4, quat 4, p 4, L 4} words, read the
// neighbour row and the neighbours' pos/quat through L2.  Algorithmic HBM bytes are stated in
// DESIGN.md; the working set of a 12 kbp duplex (~3 MB) is L2 / Infinity-Cache resident.
#include <algorithm>
#include <cmath>

#include "oxdna_gather.h"

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
  float s, c;
  __sincosf(6.283185307179586f * b, &s, &c);
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

template <typename R>
struct LangevinConst {
  R dt, half_dt;
  R inv_mass;
  R inv_inertia[3];
  R c1_t, c2_t;     // translational OU: p = c1 p + c2 xi   (c2 includes sqrt(m))
  R c1_r, c2_r[3];  // rotational OU per principal axis     (c2 includes sqrt(I_k))
  R skin_half_sq;   // (skin/2)^2 for the displacement check, <= 0 disables
};

// rotation about body axis K by angle phi = h L_K / I_K  (one NO_SQUISH factor)
template <int K, typename R>
__device__ __forceinline__ void free_rotor(R* q, R* L, R h, const R* inv_I) {
  const R phi = h * L[K] * inv_I[K];
  R s, c;
  if constexpr (sizeof(R) == 4) {
    __sincosf(R(0.5) * phi, &s, &c);
  } else {
    sincos(R(0.5) * phi, &s, &c);
  }
  // q <- q (x) (c, s e_K) = c q + s P_K q
  const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  if constexpr (K == 0) {
    q[0] = c * q0 - s * q1;
    q[1] = c * q1 + s * q0;
    q[2] = c * q2 + s * q3;
    q[3] = c * q3 - s * q2;
  } else if constexpr (K == 1) {
    q[0] = c * q0 - s * q2;
    q[1] = c * q1 - s * q3;
    q[2] = c * q2 + s * q0;
    q[3] = c * q3 + s * q1;
  } else {
    q[0] = c * q0 - s * q3;
    q[1] = c * q1 + s * q2;
    q[2] = c * q2 - s * q1;
    q[3] = c * q3 + s * q0;
  }
  // body components of the (lab-fixed) angular momentum rotate by -phi about e_K
  const R cf = c * c - s * s, sf = R(2) * s * c;
  constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
  const R la = L[A], lb = L[B];
  L[A] = cf * la + sf * lb;
  L[B] = -sf * la + cf * lb;
}

template <typename R>
__device__ __forceinline__ void drift(R* x, R* q, const R* p, R* L, R h, const LangevinConst<R>& K) {
  x[0] += h * p[0] * K.inv_mass;
  x[1] += h * p[1] * K.inv_mass;
  x[2] += h * p[2] * K.inv_mass;
  free_rotor<2>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<1>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<0>(q, L, h, K.inv_inertia);
  free_rotor<1>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<2>(q, L, R(0.5) * h, K.inv_inertia);
}

constexpr int kMdBlock = 256;
constexpr int kTraceWidth = T_COUNT + 2;  // 8 energy terms + KE_trans + KE_rot

// One MD step (see file header).  kick_close: multiple of dt*F that closes the previous step
// (0 for the first kernel of a run, 1/2 otherwise); do_step = 0 for the closing-only kernel.
template <typename R, int MODEL, int G, bool SAVE>
__global__ __launch_bounds__(kMdBlock, (sizeof(R) == 4 ? 4 : 2)) void md_step_kernel(
    const OxParams<R> P, const BoxT<R> box, const LangevinConst<R> K, int n,
    const typename Vec4T<R>::type* __restrict__ pos_in, const typename Vec4T<R>::type* __restrict__ quat_in,
    typename Vec4T<R>::type* __restrict__ pos_out, typename Vec4T<R>::type* __restrict__ quat_out,
    typename Vec4T<R>::type* __restrict__ mom, typename Vec4T<R>::type* __restrict__ ang,
    const int* __restrict__ meta, const int* __restrict__ rows, const int* __restrict__ row_len, int row_stride,
    R kick_close, int do_step, uint64_t seed, uint64_t step, const typename Vec4T<R>::type* __restrict__ ref_pos,
    int* __restrict__ flags, R* __restrict__ traj_c, R* __restrict__ traj_q, double* __restrict__ e_part) {
  using V4 = typename Vec4T<R>::type;
  constexpr int PPB = kMdBlock / G;
  __shared__ double e_lds[SAVE ? PPB : 1][kTraceWidth];
  const int grp = threadIdx.x / G;
  const int lane = threadIdx.x % G;
  const int i = blockIdx.x * PPB + grp;

  R e[T_COUNT];
#pragma unroll
  for (int k = 0; k < T_COUNT; ++k) e[k] = R(0);
  SelfGrad<R> sg;
  sg.dc = sg.g1 = sg.g2 = sg.g3 = V3<R>{R(0), R(0), R(0)};
  Nuc<R> self;
  R qs[4] = {R(1), R(0), R(0), R(0)};
  if (i < n) {
    Vec4Loader<R> ld{pos_in, quat_in, meta};
    ld.load(i, self, qs);
    NoPG pg;
    gather_row<R, MODEL, true, NoPG, G>(P, ld, box, rows, row_stride, row_len[i], i, self, lane, e, sg, pg);
  }
  group_reduce<G, R, true>(e, sg);

  double ke_t = 0.0, ke_r = 0.0;
  if (lane == 0 && i < n) {
    // force and body-frame torque at x_k
    const V3<R> F = -sg.dc;
    const V3<R> tl = axes_grad_to_torque(self, sg);
    const R tb[3] = {dot(self.a1, tl), dot(self.a2, tl), dot(self.a3, tl)};
    V4 pm = mom[i], lm = ang[i];
    R p[3] = {pm.x, pm.y, pm.z}, L[3] = {lm.x, lm.y, lm.z};
    const R kc = kick_close * K.dt;
    p[0] += kc * F.x;
    p[1] += kc * F.y;
    p[2] += kc * F.z;
    L[0] += kc * tb[0];
    L[1] += kc * tb[1];
    L[2] += kc * tb[2];
    if constexpr (SAVE) {
      ke_t = 0.5 * double(K.inv_mass) * (double(p[0]) * p[0] + double(p[1]) * p[1] + double(p[2]) * p[2]);
      ke_r = 0.5 * (double(K.inv_inertia[0]) * L[0] * L[0] + double(K.inv_inertia[1]) * L[1] * L[1] +
                    double(K.inv_inertia[2]) * L[2] * L[2]);
      if (traj_c) {
        traj_c[3 * i + 0] = self.c.x;
        traj_c[3 * i + 1] = self.c.y;
        traj_c[3 * i + 2] = self.c.z;
      }
      if (traj_q) {
        traj_q[4 * i + 0] = qs[0];
        traj_q[4 * i + 1] = qs[1];
        traj_q[4 * i + 2] = qs[2];
        traj_q[4 * i + 3] = qs[3];
      }
    }
    R x[3] = {self.c.x, self.c.y, self.c.z};
    if (do_step) {
      p[0] += K.half_dt * F.x;
      p[1] += K.half_dt * F.y;
      p[2] += K.half_dt * F.z;
      L[0] += K.half_dt * tb[0];
      L[1] += K.half_dt * tb[1];
      L[2] += K.half_dt * tb[2];
      drift(x, qs, p, L, K.half_dt, K);
      R z[6];
      normals6(seed, (uint32_t)i, step, 0u, z);
      p[0] = K.c1_t * p[0] + K.c2_t * z[0];
      p[1] = K.c1_t * p[1] + K.c2_t * z[1];
      p[2] = K.c1_t * p[2] + K.c2_t * z[2];
      L[0] = K.c1_r * L[0] + K.c2_r[0] * z[3];
      L[1] = K.c1_r * L[1] + K.c2_r[1] * z[4];
      L[2] = K.c1_r * L[2] + K.c2_r[2] * z[5];
      drift(x, qs, p, L, K.half_dt, K);
      // keep the quaternion on the unit sphere (fp32 round-off)
      const R inv = m_rsqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
      qs[0] *= inv;
      qs[1] *= inv;
      qs[2] *= inv;
      qs[3] *= inv;
      if (K.skin_half_sq > R(0)) {
        const V4 r0 = ref_pos[i];
        const R dx = x[0] - r0.x, dy = x[1] - r0.y, dz = x[2] - r0.z;
        if (dx * dx + dy * dy + dz * dz > K.skin_half_sq) atomicOr(flags, 1);
      }
      if (!(x[0] == x[0]) || !(qs[0] == qs[0])) atomicOr(flags, 2);
    }
    pos_out[i] = V4{x[0], x[1], x[2], R(0)};
    quat_out[i] = V4{qs[0], qs[1], qs[2], qs[3]};
    mom[i] = V4{p[0], p[1], p[2], R(0)};
    ang[i] = V4{L[0], L[1], L[2], R(0)};
  }
  if constexpr (SAVE) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < T_COUNT; ++k) e_lds[grp][k] = (i < n) ? double(e[k]) : 0.0;
      e_lds[grp][T_COUNT] = ke_t;
      e_lds[grp][T_COUNT + 1] = ke_r;
    }
    __syncthreads();
    if (threadIdx.x < kTraceWidth) {
      double s = 0.0;
      for (int g = 0; g < PPB; ++g) s += e_lds[g][threadIdx.x];
      e_part[(size_t)blockIdx.x * kTraceWidth + threadIdx.x] = s;
    }
  }
}

__global__ void reduce_trace_kernel(const double* __restrict__ part, int n_blocks, double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= kTraceWidth) return;
  double s = 0.0;
  for (int b = 0; b < n_blocks; ++b) s += part[(size_t)b * kTraceWidth + k];
  if (out) out[k] = s;
}

// ------------------------------------------------------------------ packed (N,3)/(N,4) <-> vec4
template <typename R>
__global__ void pack_state_kernel(int n, const R* __restrict__ c, const R* __restrict__ q, const R* __restrict__ p,
                                  const R* __restrict__ l, typename Vec4T<R>::type* pos,
                                  typename Vec4T<R>::type* quat, typename Vec4T<R>::type* mom,
                                  typename Vec4T<R>::type* ang) {
  using V4 = typename Vec4T<R>::type;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pos[i] = V4{c[3 * i], c[3 * i + 1], c[3 * i + 2], R(0)};
  // the kernels assume unit quaternions (torque form); normalise on entry
  R q0 = q[4 * i], q1 = q[4 * i + 1], q2 = q[4 * i + 2], q3 = q[4 * i + 3];
  const R inv = m_rsqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  quat[i] = V4{q0 * inv, q1 * inv, q2 * inv, q3 * inv};
  mom[i] = V4{p[3 * i], p[3 * i + 1], p[3 * i + 2], R(0)};
  ang[i] = V4{l[3 * i], l[3 * i + 1], l[3 * i + 2], R(0)};
}
template <typename R>
__global__ void unpack_state_kernel(int n, const typename Vec4T<R>::type* pos, const typename Vec4T<R>::type* quat,
                                    const typename Vec4T<R>::type* mom, const typename Vec4T<R>::type* ang,
                                    R* __restrict__ c, R* __restrict__ q, R* __restrict__ p, R* __restrict__ l) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto a = pos[i];
  const auto b = quat[i];
  const auto m = mom[i];
  const auto w = ang[i];
  c[3 * i] = a.x, c[3 * i + 1] = a.y, c[3 * i + 2] = a.z;
  q[4 * i] = b.x, q[4 * i + 1] = b.y, q[4 * i + 2] = b.z, q[4 * i + 3] = b.w;
  p[3 * i] = m.x, p[3 * i + 1] = m.y, p[3 * i + 2] = m.z;
  l[3 * i] = w.x, l[3 * i + 1] = w.y, l[3 * i + 2] = w.z;
}

// Maxwell-Boltzmann momenta; the centre-of-mass momentum is removed (jax_md initialize_momenta
// with center_velocity=True).  Single block: n is at most a few 10^4 and this runs once.
template <typename R>
__global__ void init_momenta_kernel(int n, R sd_t, R sd_r0, R sd_r1, R sd_r2, uint64_t seed, R* __restrict__ p,
                                    R* __restrict__ l) {
  __shared__ double sum[3][256];
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    R z[6];
    normals6(seed, (uint32_t)i, 0xFFFFFFFFFFFFFFFFull, 7u, z);
    p[3 * i] = sd_t * z[0], p[3 * i + 1] = sd_t * z[1], p[3 * i + 2] = sd_t * z[2];
    l[3 * i] = sd_r0 * z[3], l[3 * i + 1] = sd_r1 * z[4], l[3 * i + 2] = sd_r2 * z[5];
    s0 += p[3 * i], s1 += p[3 * i + 1], s2 += p[3 * i + 2];
  }
  sum[0][threadIdx.x] = s0, sum[1][threadIdx.x] = s1, sum[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; ++k) sum[k][threadIdx.x] += sum[k][threadIdx.x + o];
    __syncthreads();
  }
  const R m0 = R(sum[0][0] / n), m1 = R(sum[1][0] / n), m2 = R(sum[2][0] / n);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    p[3 * i] -= m0, p[3 * i + 1] -= m1, p[3 * i + 2] -= m2;
  }
}

}  // namespace mythos

using namespace mythos;

struct mythos_sim {
  mythos_system* sys = nullptr;
  double dt = 0, kT = 0, gamma_t = 0, gamma_r = 0, mass = 1, inertia[3] = {1, 1, 1};
  uint64_t seed = 0;
  int64_t step = 0;
  // neighbour policy
  double r_cut = 0, skin = 0;
  int rebuild_every = 0;
  // device state (vec4, ping-pong positions / quaternions)
  void *pos[2] = {nullptr, nullptr}, *quat[2] = {nullptr, nullptr}, *mom = nullptr, *ang = nullptr;
  int* d_flags = nullptr;
  double* d_epart = nullptr;
  int epart_blocks = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // sampled per-launch timing: every kSampleStride-th step launch is bracketed by its own event pair
  static constexpr int kMaxSamples = 64;
  hipEvent_t sa[kMaxSamples] = {}, sb[kMaxSamples] = {};
  double last_avg_ms = 0;     // (ev1 - ev0) / launches: includes rebuilds and inter-kernel gaps
  double last_kernel_ms = 0;  // mean over the sampled single-launch intervals
  int last_launches = 0;
  int last_samples = 0;
};

namespace mythos {

template <typename R>
static LangevinConst<R> make_const(const mythos_sim* s) {
  LangevinConst<R> K;
  K.dt = R(s->dt);
  K.half_dt = R(0.5 * s->dt);
  K.inv_mass = R(1.0 / s->mass);
  const double c1t = std::exp(-s->gamma_t * s->dt), c1r = std::exp(-s->gamma_r * s->dt);
  K.c1_t = R(c1t);
  K.c2_t = R(std::sqrt(s->kT * (1.0 - c1t * c1t) * s->mass));
  K.c1_r = R(c1r);
  for (int k = 0; k < 3; ++k) {
    K.inv_inertia[k] = R(1.0 / s->inertia[k]);
    K.c2_r[k] = R(std::sqrt(s->kT * (1.0 - c1r * c1r) * s->inertia[k]));
  }
  K.skin_half_sq = R(s->rebuild_every > 0 ? 0.25 * s->skin * s->skin : -1.0);
  return K;
}

template <typename R, int MODEL>
static int run_typed(mythos_sim* sim, R* center, R* quat, R* p_lin, R* p_ang, int n_steps, int save_every,
                     R* traj_center, R* traj_quat, double* e_trace, hipStream_t st) {
  using V4 = typename Vec4T<R>::type;
  constexpr int G = 16;
  constexpr int PPB = kMdBlock / G;
  mythos_system* sys = sim->sys;
  const int n = sys->n;
  const int blocks = (n + PPB - 1) / PPB;
  const int tb = (n + 255) / 256;
  const OxParams<R>& P = params_of<R>(sys);
  const BoxT<R> box = make_box<R>(sys);
  const LangevinConst<R> K = make_const<R>(sim);
  V4* pos[2] = {(V4*)sim->pos[0], (V4*)sim->pos[1]};
  V4* qt[2] = {(V4*)sim->quat[0], (V4*)sim->quat[1]};
  V4* mom = (V4*)sim->mom;
  V4* ang = (V4*)sim->ang;
  MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags, 0, sizeof(int), st));
  hipLaunchKernelGGL(pack_state_kernel<R>, dim3(tb), dim3(256), 0, st, n, center, quat, p_lin, p_ang, pos[0], qt[0],
                     mom, ang);
  int cur = 0;
  const bool dynamic_list = sim->rebuild_every > 0;
  auto rebuild = [&](int buf) -> int {
    if (int rc = rows_build_device(sys, pos[buf], true, sim->r_cut, sim->skin, st)) return rc;
    MYTHOS_HIP_TRY(hipMemcpyAsync(sys->d_ref_pos, pos[buf], (size_t)n * sizeof(V4), hipMemcpyDeviceToDevice, st));
    return 0;
  };
  if (dynamic_list)
    if (int rc = rebuild(cur)) return rc;
  MYTHOS_HIP_TRY(hipEventRecord(sim->ev0, st));
  int launches = 0, samples = 0;
  const int sample_stride = std::max(1, (n_steps + 1) / mythos_sim::kMaxSamples);
  for (int k = 0; k <= n_steps; ++k) {
    const bool last = (k == n_steps);
    const bool save = save_every > 0 && k > 0 && (k % save_every == 0);
    const int sidx = save ? (k / save_every - 1) : 0;
    if (dynamic_list && k > 0 && !last && (k % sim->rebuild_every == 0))
      if (int rc = rebuild(cur)) return rc;
    const R kick_close = (k == 0) ? R(0) : R(0.5);
    const int do_step = last ? 0 : 1;
    R* tc = (save && traj_center) ? traj_center + (size_t)sidx * n * 3 : nullptr;
    R* tq = (save && traj_quat) ? traj_quat + (size_t)sidx * n * 4 : nullptr;
    const V4* ref = (const V4*)sys->d_ref_pos;
    const bool sampled = !save && (k % sample_stride == sample_stride / 2) && samples < mythos_sim::kMaxSamples;
    if (sampled) MYTHOS_HIP_TRY(hipEventRecord(sim->sa[samples], st));
    if (save) {
      hipLaunchKernelGGL((md_step_kernel<R, MODEL, G, true>), dim3(blocks), dim3(kMdBlock), 0, st, P, box, K, n,
                         pos[cur], qt[cur], pos[cur ^ 1], qt[cur ^ 1], mom, ang, sys->d_meta, sys->d_rows,
                         sys->d_row_len, sys->row_stride, kick_close, do_step, sim->seed, (uint64_t)(sim->step + k),
                         ref, sim->d_flags, tc, tq, sim->d_epart);
      hipLaunchKernelGGL(reduce_trace_kernel, dim3(1), dim3(64), 0, st, sim->d_epart, blocks,
                         e_trace ? e_trace + (size_t)sidx * kTraceWidth : nullptr);
    } else {
      hipLaunchKernelGGL((md_step_kernel<R, MODEL, G, false>), dim3(blocks), dim3(kMdBlock), 0, st, P, box, K, n,
                         pos[cur], qt[cur], pos[cur ^ 1], qt[cur ^ 1], mom, ang, sys->d_meta, sys->d_rows,
                         sys->d_row_len, sys->row_stride, kick_close, do_step, sim->seed, (uint64_t)(sim->step + k),
                         ref, sim->d_flags, tc, tq, sim->d_epart);
    }
    if (sampled) MYTHOS_HIP_TRY(hipEventRecord(sim->sb[samples++], st));
    ++launches;
    cur ^= 1;
  }
  MYTHOS_HIP_TRY(hipEventRecord(sim->ev1, st));
  MYTHOS_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(unpack_state_kernel<R>, dim3(tb), dim3(256), 0, st, n, pos[cur], qt[cur], mom, ang, center,
                     quat, p_lin, p_ang);
  int flags = 0, ov = 0;
  MYTHOS_HIP_TRY(hipMemcpyAsync(&flags, sim->d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(&ov, sys->d_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
  MYTHOS_HIP_TRY(hipStreamSynchronize(st));
  float ms = 0;
  MYTHOS_HIP_TRY(hipEventElapsedTime(&ms, sim->ev0, sim->ev1));
  sim->last_avg_ms = launches ? double(ms) / launches : 0.0;
  sim->last_launches = launches;
  double acc = 0;
  for (int k = 0; k < samples; ++k) {
    float t = 0;
    MYTHOS_HIP_TRY(hipEventElapsedTime(&t, sim->sa[k], sim->sb[k]));
    acc += t;
  }
  sim->last_kernel_ms = samples ? acc / samples : 0.0;
  sim->last_samples = samples;
  sim->step += n_steps;
  if (flags & 2) {
    set_error("mythos_langevin_run: NaN in the state (time step too large or overlapping start configuration)");
    return MYTHOS_ERR_NUMERIC;
  }
  if (ov != 0) {
    set_error("mythos_langevin_run: neighbour row capacity exceeded (" + std::to_string(ov) + " > " +
              std::to_string(sys->row_stride) + "); rebuild with mythos_oxdna_build_neighbors first");
    return MYTHOS_ERR_OVERFLOW;
  }
  if (flags & 1) {
    set_error("mythos_langevin_run: a nucleotide moved more than skin/2 between neighbour rebuilds");
    return MYTHOS_ERR_OVERFLOW;
  }
  return MYTHOS_OK;
}

}  // namespace mythos

extern "C" {

mythos_sim_t* mythos_langevin_create(mythos_system_t* sys, double dt, double kT, double gamma_t, double gamma_r,
                                     double mass, const double* inertia, uint64_t seed) {
  if (!sys || !(dt > 0) || !(kT >= 0) || gamma_t < 0 || gamma_r < 0 || !(mass > 0)) {
    set_error("mythos_langevin_create: invalid argument");
    return nullptr;
  }
  if (hipSetDevice(sys->device) != hipSuccess) {
    set_error("mythos_langevin_create: hipSetDevice failed");
    return nullptr;
  }
  auto* s = new mythos_sim();
  s->sys = sys;
  s->dt = dt;
  s->kT = kT;
  s->gamma_t = gamma_t;
  s->gamma_r = gamma_r;
  s->mass = mass;
  for (int k = 0; k < 3; ++k) s->inertia[k] = inertia ? inertia[k] : 1.0;
  s->seed = seed;
  const size_t v4 = (sys->dtype == MYTHOS_F32 ? sizeof(float4) : sizeof(double4)) * (size_t)sys->n;
  constexpr int PPB = kMdBlock / 16;
  s->epart_blocks = (sys->n + PPB - 1) / PPB;
  bool ok = true;
  for (int k = 0; k < 2; ++k) {
    ok = ok && hipMalloc(&s->pos[k], v4) == hipSuccess && hipMalloc(&s->quat[k], v4) == hipSuccess;
  }
  ok = ok && hipMalloc(&s->mom, v4) == hipSuccess && hipMalloc(&s->ang, v4) == hipSuccess &&
       hipMalloc((void**)&s->d_flags, sizeof(int)) == hipSuccess &&
       hipMalloc((void**)&s->d_epart, (size_t)s->epart_blocks * kTraceWidth * sizeof(double)) == hipSuccess &&
       hipEventCreate(&s->ev0) == hipSuccess && hipEventCreate(&s->ev1) == hipSuccess;
  for (int k = 0; ok && k < mythos_sim::kMaxSamples; ++k)
    ok = hipEventCreate(&s->sa[k]) == hipSuccess && hipEventCreate(&s->sb[k]) == hipSuccess;
  if (ok && !sys->d_ref_pos) ok = hipMalloc(&sys->d_ref_pos, v4) == hipSuccess;
  if (!ok) {
    set_error("mythos_langevin_create: device allocation failed");
    mythos_langevin_destroy(s);
    return nullptr;
  }
  return s;
}

void mythos_langevin_destroy(mythos_sim_t* s) {
  if (!s) return;
  for (int k = 0; k < 2; ++k) {
    if (s->pos[k]) (void)hipFree(s->pos[k]);
    if (s->quat[k]) (void)hipFree(s->quat[k]);
  }
  if (s->mom) (void)hipFree(s->mom);
  if (s->ang) (void)hipFree(s->ang);
  if (s->d_flags) (void)hipFree(s->d_flags);
  if (s->d_epart) (void)hipFree(s->d_epart);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  for (int k = 0; k < mythos_sim::kMaxSamples; ++k) {
    if (s->sa[k]) (void)hipEventDestroy(s->sa[k]);
    if (s->sb[k]) (void)hipEventDestroy(s->sb[k]);
  }
  delete s;
}

int mythos_langevin_set_neighbor_policy(mythos_sim_t* s, double r_cut, double skin, int every) {
  if (!s || (every > 0 && (!(r_cut > 0) || !(skin > 0)))) {
    set_error("mythos_langevin_set_neighbor_policy: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->r_cut = r_cut;
  s->skin = skin;
  s->rebuild_every = every;
  return MYTHOS_OK;
}

int mythos_langevin_init_momenta(mythos_sim_t* s, void* p_lin, void* p_ang, mythos_stream_t stream) {
  if (!s || !p_lin || !p_ang) {
    set_error("mythos_langevin_init_momenta: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  const double sd_t = std::sqrt(s->mass * s->kT);
  double sd_r[3];
  for (int k = 0; k < 3; ++k) sd_r[k] = std::sqrt(s->inertia[k] * s->kT);
  if (s->sys->dtype == MYTHOS_F32)
    hipLaunchKernelGGL(init_momenta_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, s->sys->n, float(sd_t),
                       float(sd_r[0]), float(sd_r[1]), float(sd_r[2]), s->seed, (float*)p_lin, (float*)p_ang);
  else
    hipLaunchKernelGGL(init_momenta_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, s->sys->n, sd_t,
                       sd_r[0], sd_r[1], sd_r[2], s->seed, (double*)p_lin, (double*)p_ang);
  MYTHOS_HIP_TRY(hipGetLastError());
  return MYTHOS_OK;
}

int mythos_langevin_run(mythos_sim_t* s, void* center, void* quat, void* p_lin, void* p_ang, int n_steps,
                        int save_every, void* traj_center, void* traj_quat, double* e_trace,
                        mythos_stream_t stream) {
  if (!s || !center || !quat || !p_lin || !p_ang || n_steps < 0 || save_every < 0) {
    set_error("mythos_langevin_run: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  mythos_system* sys = s->sys;
  if (!sys->params_set || (!sys->nbrs_set && s->rebuild_every <= 0)) {
    set_error("mythos_langevin_run: parameters and neighbours (or a neighbour policy) must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(sys->device));
  if (s->rebuild_every > 0 && sys->row_stride == 0)
    if (int rc = rows_reserve(sys, 64)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (sys->dtype == MYTHOS_F32) {
    if (sys->model == 1)
      return run_typed<float, 1>(s, (float*)center, (float*)quat, (float*)p_lin, (float*)p_ang, n_steps, save_every,
                                 (float*)traj_center, (float*)traj_quat, e_trace, st);
    return run_typed<float, 2>(s, (float*)center, (float*)quat, (float*)p_lin, (float*)p_ang, n_steps, save_every,
                               (float*)traj_center, (float*)traj_quat, e_trace, st);
  }
  if (sys->model == 1)
    return run_typed<double, 1>(s, (double*)center, (double*)quat, (double*)p_lin, (double*)p_ang, n_steps,
                                save_every, (double*)traj_center, (double*)traj_quat, e_trace, st);
  return run_typed<double, 2>(s, (double*)center, (double*)quat, (double*)p_lin, (double*)p_ang, n_steps, save_every,
                              (double*)traj_center, (double*)traj_quat, e_trace, st);
}

int64_t mythos_langevin_get_step(const mythos_sim_t* s) { return s ? s->step : -1; }

int mythos_langevin_set_step(mythos_sim_t* s, int64_t step) {
  if (!s || step < 0) {
    set_error("mythos_langevin_set_step: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->step = step;
  return MYTHOS_OK;
}

int mythos_langevin_last_kernel_ms(const mythos_sim_t* s, double* kernel_ms, double* loop_ms_per_launch,
                                   int* launches, int* samples) {
  if (!s) {
    set_error("mythos_langevin_last_kernel_ms: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (kernel_ms) *kernel_ms = s->last_kernel_ms;
  if (loop_ms_per_launch) *loop_ms_per_launch = s->last_avg_ms;
  if (launches) *launches = s->last_launches;
  if (samples) *samples = s->last_samples;
  return MYTHOS_OK;
}

}  // extern "C"
