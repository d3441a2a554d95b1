// Langevin MD of a MARTINI system: one fused kernel per time step (BASELINE configs[2]).
//
// The reference has no MARTINI integrator of its own - it drives GROMACS as an external process
// (mythos/simulators/gromacs/) and only re-evaluates energies (mythos/energy/martini/m2/*.py).  This file
// is the device-resident counterpart for the same force field: shifted-cut-off Lennard-Jones over a Verlet
// list, harmonic bonds, G96 / harmonic angles, and the BAOAB Langevin splitting used for oxDNA
// (langevin_core.inc) specialised to point particles:
//   B  v += h F / m      A  x += h v      O  v = c1 v + sqrt(kT (1 - c1^2) / m) xi,  c1 = exp(-gamma dt)
// Units are GROMACS': nm, ps, amu, kJ/mol (1 kJ/mol = 1 amu nm^2 / ps^2), kT = 0.0083144626 T.
//
// Work decomposition (as md_step_kernel): 16 lanes per bead (kMmG), 16 beads per 256-thread workgroup; the lanes stride
// over the bead's neighbour row (partners inside r_c + skin, bonded partners already excluded), then over the
// bead's bonds and angles; DPP fold; one wavefront integrates the workgroup's beads.  The kernel that
// evaluates F(x_k) closes step k-1 and opens step k; frames ping-pong.  No atomics in the force path.
// Roofline: as for oxDNA the state is cache resident; algorithmic bytes per bead per step =
// 2 x (pos 3 + vel 3) words + 4 B type + 4 B x nbar.
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "cell_list.h"
#include "martini_internal.h"
#include "philox.h"
#include "wave_ops.h"

namespace mythos {

// (threads per workgroup, scanned in round 4 at 20 480 beads: 128 / 256 / 512 / 1 024 -> 81.4 k / 84.5 k / 83.2 k / 76.6 k steps/s)
#ifndef MYTHOS_MM_BLOCK
#define MYTHOS_MM_BLOCK 256
#endif
constexpr int kMmBlock = MYTHOS_MM_BLOCK;
// Lanes per bead.  Sixteen since round 4: the kernel needs 28 VGPRs, so twice the wavefronts (1 280 workgroups of 16 beads
// for 20 480 beads, five per CU) fit without any trade, and a lane's chain through the row is half as long - the kernel
// was latency-bound at 2.5 wavefronts per SIMD (65 % of its wave cycles waiting).  20 480 beads, events, loop rate
// (scripts/exp_martini_lanes_r04.sh, two alternations): 8 lanes 12.05 us / 68.7 k steps/s; 16 lanes with 4 / 3 / 2 row
// entries per lane in flight 10.6 / 11.1 / 11.1 us, 76.0 k / 75.6 k / 74.7 k; 32 lanes 14.3 us / 59.7 k.
#ifndef MYTHOS_MM_G
#define MYTHOS_MM_G 16
#endif
#ifndef MYTHOS_MM_BATCH
#define MYTHOS_MM_BATCH 4
#endif
constexpr int kMmG = MYTHOS_MM_G;
constexpr int kMmPPB = kMmBlock / kMmG;
constexpr int kMmTrace = 4;  // lj, bond, angle, kinetic

template <typename R>
struct Real4;
template <>
struct Real4<float> {
  using type = float4;
};
template <>
struct Real4<double> {
  using type = double4;
};

template <typename R>
struct MmConst {
  R lx, ly, lz, ilx, ily, ilz;
  R rc2;
  R dt, half_dt, c1, kT;
  R skin_half_sq;  // (skin/2)^2, <= 0 disables the displacement check
  R rin2;          // (r_c + inner margin)^2: range of the pruned rows (EMIT); <= 0: no pruned rows
  R step_max_sq;   // a bead may move (margin / 2) / (inner_every - 1) per step while pruned rows are in use; <= 0: no check
  int n_types, angle_kind;
};

// Fused multiply-add spelled out.  The squared distance of the row walk decides which entries the pruned rows keep: every
// instantiation of the step kernel has to round it the same way, whatever contraction the compiler would choose for it
// (the energy-trace and the plain instantiation disagreed in the last bit of r^2 in fp64, kept different entries at
// the edge of the pruned range, and the sums behind that entry fell into other lanes).
__device__ __forceinline__ float mm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double mm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename R>
__device__ __forceinline__ R mm_wrap(R d, R l, R il) {
  return mm_fma(-l, m_rint(d * il), d);
}

// optimisation barrier on a register value: whatever produced it stays before this point, its uses after
template <typename T>
__device__ __forceinline__ void mm_pin(T& v) {
  asm volatile("" : "+v"(v));
}

// Wave priority by phase (see md_step_kernel, langevin_core.inc): MYTHOS_MM_PRIO_MAP = three decimal digits, the s_setprio
// level of the row loop, the bonded lists, and everything behind the barrier (0 = no s_setprio)
// (20 480 beads: 64.7 k -> 65.3 k steps/s with 310)
#ifndef MYTHOS_MM_PRIO_MAP
#define MYTHOS_MM_PRIO_MAP 310
#endif
#if MYTHOS_MM_PRIO_MAP != 0
constexpr int mm_prio_digit(int phase) {
  int v = MYTHOS_MM_PRIO_MAP;
  for (int k = 2; k > phase; --k) v /= 10;
  return v % 10;
}
#define MM_PRIO(phase) __builtin_amdgcn_s_setprio(mm_prio_digit(phase))
#else
#define MM_PRIO(phase) do { } while (0)
#endif

// EMIT (pruned rows, round 4): the launch walks the Verlet rows (range r_c + skin) and, as a by-product of the distances it
// computes anyway, writes for every bead the entries inside r_c + margin to a second set of rows, in row order.  The next
// inner_every - 1 launches walk those instead - 78 entries for 135 at the bilayer's policy - while no bead moves more
// than margin / 2 in total (checked per step, conservatively; a violation halts like a bead that leaves its skin, and
// the recovery rebuilds both lists).  GROMACS prunes its pair list the same way between searches ("dynamic pruning").
// Which list a launch walks is a function of the steps since the Verlet rows were built, so split calls stay bitwise equal.
template <typename R, bool SAVE, bool EMIT = false>
__global__ __launch_bounds__(kMmBlock, 1024 / kMmBlock) void martini_md_step_kernel(
    int n, const MmConst<R> K, const typename Real4<R>::type* __restrict__ in, typename Real4<R>::type* __restrict__ out,
    typename Real4<R>::type* __restrict__ vel, const int* __restrict__ rows, const int* __restrict__ row_len,
    int row_stride, const R* __restrict__ sigma, const R* __restrict__ eps, const int* __restrict__ bead_bonds,
    const int* __restrict__ bead_angles, const int* __restrict__ bonds, const R* __restrict__ bond_k,
    const R* __restrict__ bond_r0, const int* __restrict__ angles, const R* __restrict__ angle_k,
    const R* __restrict__ angle_t0, R kick_close, int do_step, uint64_t seed, uint64_t step,
    const typename Real4<R>::type* __restrict__ ref_pos, int* __restrict__ flags, R* __restrict__ traj,
    double* __restrict__ e_part, const int* __restrict__ list_overflow, int k_index, int* __restrict__ emit_rows,
    int* __restrict__ emit_len, const int* __restrict__ bb_partner, const int2* __restrict__ ba_partner) {
  using V4 = typename Real4<R>::type;
  constexpr int G = kMmG, PPB = kMmPPB;
  extern __shared__ unsigned char smem_raw[];
  R* s_sig2 = reinterpret_cast<R*>(smem_raw);
  R* s_eps = s_sig2 + K.n_types * K.n_types;
  __shared__ R s_f[PPB][4];
  __shared__ double s_e[SAVE ? PPB : 1][kMmTrace];

  const int n_blocks = (n + PPB - 1) / PPB;
  const int bid = (int)(blockIdx.x & 7) * ((n_blocks + 7) >> 3) + (int)(blockIdx.x >> 3);  // XCD-aware order
  if (bid >= n_blocks) return;
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int i = bid * PPB + grp;
  const bool valid = i < n;
  const int ii = valid ? i : n - 1;
  // Halt word (see mythos_langevin_run in langevin_core.inc): set by the step that moved a bead out of its skin, or by a
  // rebuild that overflowed.  One lane requests it here; everybody looks at it behind the barrier in front of the
  // integration, before which nothing is written to global memory.
  __shared__ int s_halt;
  int halt_word = 0;
  // the word holds the index of the first launch that must not run: late workgroups of the launch that set it go on
  if (threadIdx.x == 0) {
    const int hw = flags[1];
    halt_word = ((hw != 0 && hw <= k_index) ? 1 : 0) | (list_overflow ? (list_overflow[0] | list_overflow[1]) : 0);
  }

  const int tt = K.n_types * K.n_types;
  for (int k = threadIdx.x; k < tt; k += kMmBlock) {  // (sigma: squared already, mm compact tables)
    s_sig2[k] = sigma[k];
    s_eps[k] = eps[k];
  }
  const V4 me = in[ii];
  const int type_i = (int)me.w * K.n_types;  // the bead type travels as an integer-valued real in .w
  const int* __restrict__ row = rows + (size_t)ii * row_stride;
  const int len = valid ? row_len[ii] : 0;
  // bonds and angles: the lane's incidence entry and the partner beads it names are requested here, in front of the row
  // walk, so that behind it ONE round of gathers (the partners' positions) is left.  (Until round 4: entry -> bond /
  // angle record -> positions, three dependent reads behind the walk, 2.2 us of a 10.6 us kernel.)
  int ent_b0 = -1, par_b0 = -1, ent_a0 = -1;
  int2 par_a0{-1, -1};
  if (valid) {
    if (lane < kMaxBeadBonds) ent_b0 = bead_bonds[(size_t)i * kMaxBeadBonds + lane], par_b0 = bb_partner[(size_t)i * kMaxBeadBonds + lane];
    if (lane < kMaxBeadAngles) ent_a0 = bead_angles[(size_t)i * kMaxBeadAngles + lane], par_a0 = ba_partner[(size_t)i * kMaxBeadAngles + lane];
  }
  __syncthreads();

  MM_PRIO(0);
  R gx = 0, gy = 0, gz = 0;  // dU/dx_i
  R e_lj = 0, e_b = 0, e_a = 0;
  const R irc2 = R(1) / K.rc2;
  // ---- Lennard-Jones over the row, kLjBatch entries per lane at a time: their neighbour positions are requested
  //      together and the next batch's row entries before this batch is evaluated.  20 480 beads x 8 lanes are 2.5
  //      wavefronts per SIMD, too few to hide a gather per iteration (one-ahead prefetch: 14 exposed round trips per
  //      row of 112; batches of four: four).  The order of the sums over a lane's entries is unchanged.
  {
    constexpr int kLjBatch = MYTHOS_MM_BATCH;
    int jn[kLjBatch];
    int n_in = 0;  // (EMIT) entries of the pruned row so far: the same number in every lane of the group
    int* __restrict__ emit_row = EMIT ? emit_rows + (size_t)ii * row_stride : nullptr;
    const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned int kGroupMask = (G >= 32) ? 0xffffffffu : ((1u << (G & 31)) - 1u);
    (void)n_in, (void)emit_row, (void)gshift;
#pragma unroll
    for (int u = 0; u < kLjBatch; ++u) jn[u] = (u * G + lane < len) ? row[u * G + lane] : -1;
#pragma unroll 1
    for (int s0 = 0; s0 < len; s0 += kLjBatch * G) {
      int j[kLjBatch];
      V4 o[kLjBatch];
#pragma unroll
      for (int u = 0; u < kLjBatch; ++u) {
        j[u] = jn[u];
        o[u] = in[j[u] >= 0 ? j[u] : ii];
      }
#pragma unroll
      for (int u = 0; u < kLjBatch; ++u) {
        const int idx = s0 + (kLjBatch + u) * G + lane;
        jn[u] = (idx < len) ? row[idx] : -1;
      }
#pragma unroll
      for (int u = 0; u < kLjBatch; ++u) {
        const bool have = j[u] >= 0;
        const R dx = mm_wrap(me.x - o[u].x, K.lx, K.ilx), dy = mm_wrap(me.y - o[u].y, K.ly, K.ily), dz = mm_wrap(me.z - o[u].z, K.lz, K.ilz);
        const R r2 = mm_fma(dz, dz, mm_fma(dy, dy, dx * dx));
        if constexpr (EMIT) {
          const bool keep = have && r2 < K.rin2;
          const unsigned int gm = (unsigned int)(__ballot(keep) >> gshift) & kGroupMask;
          if (keep) emit_row[n_in + __popc(gm & ((1u << lane) - 1u))] = j[u];
          n_in += __popc(gm);
        }
        if (have && r2 < K.rc2) {
          const int tp = type_i + (int)o[u].w;
          const R ir2 = R(1) / r2;
          const R s2 = s_sig2[tp] * ir2, s6 = s2 * s2 * s2, s12 = s6 * s6;
          const R ep = s_eps[tp];
          const R g = R(-24) * ep * (R(2) * s12 - s6) * ir2;  // (dV/dr) / r
          gx += g * dx, gy += g * dy, gz += g * dz;
          if constexpr (SAVE) {
            const R c2 = s_sig2[tp] * irc2, c6 = c2 * c2 * c2;
            e_lj += R(0.5) * R(4) * ep * ((s12 - s6) - (c6 * c6 - c6));
          }
        }
      }
    }
    if constexpr (EMIT)
      if (valid && lane == 0) emit_len[i] = n_in;
  }
  // ---- bonds and angles of this bead (incidence lists, one entry per lane)
  MM_PRIO(1);
  if (valid) {
    for (int s = lane; s < kMaxBeadBonds; s += G) {
      const int ent = (s == lane) ? ent_b0 : bead_bonds[(size_t)i * kMaxBeadBonds + s];
      if (ent < 0) continue;
      const int partner = (s == lane) ? par_b0 : bb_partner[(size_t)i * kMaxBeadBonds + s];
      const int b = ent >> 1, side = ent & 1;
      const V4 o = in[partner];
      const R dx = wrap(me.x - o.x, K.lx, K.ilx), dy = wrap(me.y - o.y, K.ly, K.ily), dz = wrap(me.z - o.z, K.lz, K.ilz);
      const R r = m_sqrt(dx * dx + dy * dy + dz * dz), x = r - bond_r0[b];
      const R c = bond_k[b] * x / r;
      gx += c * dx, gy += c * dy, gz += c * dz;
      if constexpr (SAVE)
        if (side == 0) e_b += R(0.5) * bond_k[b] * x * x;
    }
    for (int s = lane; s < kMaxBeadAngles; s += G) {
      const int ent = (s == lane) ? ent_a0 : bead_angles[(size_t)i * kMaxBeadAngles + s];
      if (ent < 0) continue;
      const int2 others = (s == lane) ? par_a0 : ba_partner[(size_t)i * kMaxBeadAngles + s];
      const int a = ent >> 2, role = ent & 3;  // 0: first bead, 1: centre, 2: last bead
      // (the other two beads of the angle, in its order; this bead's own position is in registers)
      const V4 q0 = in[others.x], q1 = in[others.y];
      const V4 pi = role == 0 ? me : q0, pj = role == 0 ? q0 : (role == 1 ? me : q1), pk = role == 2 ? me : q1;
      const R u[3] = {wrap(pi.x - pj.x, K.lx, K.ilx), wrap(pi.y - pj.y, K.ly, K.ily), wrap(pi.z - pj.z, K.lz, K.ilz)};
      const R v[3] = {wrap(pk.x - pj.x, K.lx, K.ilx), wrap(pk.y - pj.y, K.ly, K.ily), wrap(pk.z - pj.z, K.lz, K.ilz)};
      const R u2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2], v2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      const R uv = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
      const R iu = R(1) / m_sqrt(u2), iv = R(1) / m_sqrt(v2);
      const R c = uv * iu * iv;
      R dEdc, en;
      if (K.angle_kind == 0) {
        const R x = c - angle_t0[a];  // (G96: the array holds cos(theta0), mm_angle_ref)
        dEdc = angle_k[a] * x;
        en = R(0.5) * angle_k[a] * x * x;
      } else {
        const R cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const R sn = m_sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) * iu * iv;
        R th;
        if constexpr (sizeof(R) == 4) th = atan2f(sn, c); else th = atan2(sn, c);
        const R x = th - angle_t0[a];
        dEdc = (sn > R(1e-6)) ? -angle_k[a] * x / sn : angle_k[a];
        en = R(0.5) * angle_k[a] * x * x;
      }
      if constexpr (SAVE)
        if (role == 0) e_a += en;
      R gk[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const R du = (v[k] * iv - c * u[k] * iu) * iu, dv = (u[k] * iu - c * v[k] * iv) * iv;
        gk[k] = (role == 0) ? du : ((role == 2) ? dv : -(du + dv));
      }
      gx += dEdc * gk[0], gy += dEdc * gk[1], gz += dEdc * gk[2];
    }
  }
  gx = group_sum<G>(gx);
  gy = group_sum<G>(gy);
  gz = group_sum<G>(gz);
  if constexpr (SAVE) {
    e_lj = group_sum<G>(e_lj);
    e_b = group_sum<G>(e_b);
    e_a = group_sum<G>(e_a);
  }
  if (lane == 0) {
    s_f[grp][0] = gx, s_f[grp][1] = gy, s_f[grp][2] = gz;
    if constexpr (SAVE) {
      s_e[grp][0] = valid ? double(e_lj) : 0.0;
      s_e[grp][1] = valid ? double(e_b) : 0.0;
      s_e[grp][2] = valid ? double(e_a) : 0.0;
      s_e[grp][3] = 0.0;
    }
  }
  // ---- integrator prologue, before the barrier: the integrating wavefront draws its thermostat noise and fetches
  //      position, velocity and list reference here, so the tail of the kernel behind the barrier is arithmetic only
  //      (the oxDNA step kernel's arrangement, langevin_core.inc).  mm_pin keeps the values on this side of the barrier.
  const int int_wave = (bid >> 2) & (kMmBlock / 64 - 1) & 3;
  const int il = threadIdx.x & 63;
  const int ib = bid * PPB + il;
  const bool integrates = (int)(threadIdx.x >> 6) == int_wave && il < PPB && ib < n;
  R z[6] = {R(0), R(0), R(0), R(0), R(0), R(0)};
  V4 x0{}, vv{}, r0{};
  if (integrates) {
    x0 = in[ib];
    vv = vel[ib];
    if (do_step && K.skin_half_sq > R(0)) r0 = ref_pos[ib];
    if (do_step) normals6(seed, (uint32_t)ib, step, 0u, z);
    mm_pin(z[0]), mm_pin(z[1]), mm_pin(z[2]);
    mm_pin(x0.x), mm_pin(x0.y), mm_pin(x0.z), mm_pin(vv.x), mm_pin(vv.y), mm_pin(vv.z), mm_pin(vv.w);
  }
  if (threadIdx.x == 0) s_halt = halt_word;
  __syncthreads();
  if (s_halt != 0) return;  // halted: the state stays at the last valid step
  MM_PRIO(2);
  if (bid == 0 && threadIdx.x == 0) flags[2] = k_index + 1;
  // ---- one wavefront integrates the beads of the workgroup, one per lane
  if (integrates) {
    const R im = vv.w;  // inverse mass
    const R F[3] = {-s_f[il][0], -s_f[il][1], -s_f[il][2]};
    R x[3] = {x0.x, x0.y, x0.z}, v[3] = {vv.x, vv.y, vv.z};
    const R kc = kick_close * K.dt * im;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] += kc * F[k];
    if constexpr (SAVE) {
      s_e[il][3] = 0.5 * (double(v[0]) * v[0] + double(v[1]) * v[1] + double(v[2]) * v[2]) / double(im);
      if (traj) traj[3 * (size_t)ib] = x[0], traj[3 * (size_t)ib + 1] = x[1], traj[3 * (size_t)ib + 2] = x[2];
    }
    if (do_step) {
      const R hk = K.half_dt * im;
      const R c2 = m_sqrt(K.kT * im * (R(1) - K.c1 * K.c1));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v[k] += hk * F[k];
        x[k] += K.half_dt * v[k];
        v[k] = K.c1 * v[k] + c2 * z[k];
        x[k] += K.half_dt * v[k];
      }
      if (K.skin_half_sq > R(0)) {
        const R dx = x[0] - r0.x, dy = x[1] - r0.y, dz = x[2] - r0.z;
        if (dx * dx + dy * dy + dz * dz > K.skin_half_sq) atomicMax(flags + 1, k_index + 1);  // stale for the NEXT forces: launch k + 1 halts
      }
      if (K.step_max_sq > R(0)) {  // pruned rows in use: see EMIT
        const R dx = x[0] - x0.x, dy = x[1] - x0.y, dz = x[2] - x0.z;
        if (dx * dx + dy * dy + dz * dz > K.step_max_sq) atomicMax(flags + 1, k_index + 1);
      }
      if (!(x[0] == x[0]) || !(v[0] == v[0])) atomicOr(flags, 2);
    }
    out[ib] = V4{x[0], x[1], x[2], x0.w};
    vel[ib] = V4{v[0], v[1], v[2], im};
    if constexpr (!SAVE) {
      // positions-only trajectory (e_trace == NULL): the launch that produces x_{k+1} writes it to the caller's row too
      // (md_step_kernel does the same, langevin_core.inc) - no energy-trace instantiation, no reduction, no closing launch
      if (traj) traj[3 * (size_t)ib] = x[0], traj[3 * (size_t)ib + 1] = x[1], traj[3 * (size_t)ib + 2] = x[2];
    }
  }
  if constexpr (SAVE) {
    __syncthreads();
    if (threadIdx.x < kMmTrace) {
      double s = 0.0;
      for (int g = 0; g < PPB; ++g) s += s_e[g][threadIdx.x];
      e_part[(size_t)bid * kMmTrace + threadIdx.x] = s;
    }
  }
}

// 256 threads = 16 columns x 16 groups of workgroup partials, the group sums added in a fixed order (as langevin_core.inc's)
__global__ __launch_bounds__(256) void mm_reduce_trace_kernel(const double* __restrict__ part, int n_blocks, double* __restrict__ out) {
  static_assert(kMmTrace <= 16, "one column per trace entry");
  __shared__ double acc[16][17];
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  double s = 0.0;
  if (k < kMmTrace)
    for (int b = g; b < n_blocks; b += 16) s += part[(size_t)b * kMmTrace + k];
  acc[g][k] = s;
  __syncthreads();
  if (g == 0 && k < kMmTrace && out) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[j][k];
    out[k] = t;
  }
}

template <typename R>
__global__ void mm_pack_kernel(int n, const R* __restrict__ pos, const R* __restrict__ v, const int* __restrict__ types,
                               const R* __restrict__ inv_mass, typename Real4<R>::type* __restrict__ frame,
                               typename Real4<R>::type* __restrict__ vel) {
  using V4 = typename Real4<R>::type;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  frame[i] = V4{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], R(types[i])};
  vel[i] = V4{v[3 * i], v[3 * i + 1], v[3 * i + 2], inv_mass[i]};
}

template <typename R>
__global__ void mm_unpack_kernel(int n, const typename Real4<R>::type* __restrict__ frame,
                                 const typename Real4<R>::type* __restrict__ vel, R* __restrict__ pos, R* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pos[3 * i] = frame[i].x, pos[3 * i + 1] = frame[i].y, pos[3 * i + 2] = frame[i].z;
  v[3 * i] = vel[i].x, v[3 * i + 1] = vel[i].y, v[3 * i + 2] = vel[i].z;
}

// Maxwell-Boltzmann velocities with the centre-of-mass momentum removed (single workgroup, runs once)
template <typename R>
__global__ void mm_init_velocities_kernel(int n, R kT, const R* __restrict__ inv_mass, uint64_t seed, R* __restrict__ v) {
  __shared__ double sum[4][256];
  double s0 = 0, s1 = 0, s2 = 0, sm = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    R z[6];
    normals6(seed, (uint32_t)i, 0xFFFFFFFFFFFFFFFFull, 7u, z);
    const R sd = m_sqrt(kT * inv_mass[i]);
    v[3 * i] = sd * z[0], v[3 * i + 1] = sd * z[1], v[3 * i + 2] = sd * z[2];
    const double m = 1.0 / double(inv_mass[i]);
    s0 += m * v[3 * i], s1 += m * v[3 * i + 1], s2 += m * v[3 * i + 2], sm += m;
  }
  sum[0][threadIdx.x] = s0, sum[1][threadIdx.x] = s1, sum[2][threadIdx.x] = s2, sum[3][threadIdx.x] = sm;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 4; ++k) sum[k][threadIdx.x] += sum[k][threadIdx.x + o];
    __syncthreads();
  }
  const R c0 = R(sum[0][0] / sum[3][0]), c1 = R(sum[1][0] / sum[3][0]), c2 = R(sum[2][0] / sum[3][0]);
  for (int i = threadIdx.x; i < n; i += blockDim.x) v[3 * i] -= c0, v[3 * i + 1] -= c1, v[3 * i + 2] -= c2;
}

// ------------------------------------------------------------------------------------------------
// Verlet rows for point particles in a periodic orthorhombic box.  One wavefront per bead; candidates come
// from the 27 surrounding cells of the hashed cell list (cell_list.h) or, for boxes under three cells per edge
// and tiny systems, from a sweep over all beads.  Directly bonded partners are left out (the reference masks
// them, mythos/energy/martini/m2/lj.py:137-157).  Rows are in ascending (cell, index) order: reproducible.
// ------------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ bool mm_excluded(const int* __restrict__ ex, int j) {
  bool hit = false;
#pragma unroll
  for (int q = 0; q < kMaxExcl; ++q) hit = hit || (ex[q] == j);
  return hit;
}

// G lanes per bead (kMmRowG): the rows of 64 / G beads are built side by side in one wavefront.  The kernel is
// VALU-bound (SQ_ACTIVE_INST_VALU x 4 clocks / SIMDs = 75 % of its time): ~2.2 instructions per candidate, but a bead
// has 400-750 candidates in its 27 cells of which ~112 are inside the list range (cells of the full range: the
// sphere is 15 % of the 27-cell volume).  16, 32 and 64 lanes per bead take the same time; 8 is slower.
#ifndef MYTHOS_MM_ROW_G
#define MYTHOS_MM_ROW_G 16
#endif
constexpr int kMmRowG = MYTHOS_MM_ROW_G;

// (Round 4 tried cells of HALF the list range again - 125 around a bead, those whose nearest point lies beyond the range
// dropped before the sweep: 2.6 times fewer candidates, and 41.9 us against 36.1 us for this kernel at 20 480 beads
// (rocprofv3, scripts/exp_martini_r04.sh; the loop 66.0 k against 68.6 k steps/s).  A sweep of 16 candidates then spans
// four or five cells and every lane walks the prefix table through them - five dependent LDS reads per sweep where the
// 27-cell grid needs one every other sweep - and 125 table look-ups per bead replace 27.  The kernel is not bound by its
// candidate count alone; the variant is not in the source.)
template <typename R, int G>
__global__ __launch_bounds__(256) void mm_build_rows_cells_kernel(
    int n, const typename Real4<R>::type* __restrict__ pos, const MmConst<R> K, const CellGrid<R> g, R rl2,
    const int* __restrict__ excl, const int* __restrict__ cell_cnt,
    const typename CellPlace<R>::type* __restrict__ place, int cell_cap, const int* __restrict__ spill, int cell_H,
    int* __restrict__ rows, int* __restrict__ row_len, int row_stride, int* __restrict__ overflow,
    typename Real4<R>::type* __restrict__ ref_pos) {
  constexpr int NG = 256 / G;  // beads per workgroup
  __shared__ int s_pre[NG][29], s_st[NG][28], s_c[NG][28][3];
  const int grp = threadIdx.x / G, l = threadIdx.x % G;
  const int gshift = (threadIdx.x & 63) & ~(G - 1);  // first lane of this group inside its wavefront
  const int i = blockIdx.x * NG + grp;
  if (i >= n) return;
  const auto pi = pos[i];
  int cx, cy, cz;
  cell_of(g, pi.x, pi.y, pi.z, cx, cy, cz);
  for (int k = l; k < 28; k += G) {
    int cnt;
    if (k < 27) {
      int c[3] = {cx + k % 3 - 1, cy + (k / 3) % 3 - 1, cz + k / 9 - 1};
#pragma unroll
      for (int a = 0; a < 3; ++a) c[a] = (c[a] + g.nc[a]) % g.nc[a];
      const int h = cell_slot(g, c[0], c[1], c[2]);
      cnt = min(cell_cnt[h], cell_cap);
      s_st[grp][k] = h * cell_cap;
      s_c[grp][k][0] = c[0], s_c[grp][k][1] = c[1], s_c[grp][k][2] = c[2];
    } else {  // the spill list: particles whose bucket was full, candidates for every row
      cnt = min(cell_cnt[cell_H], kCellSpill);
    }
    s_pre[grp][k + 1] = cnt;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (l == 0) {
    int run = 0;
    s_pre[grp][0] = 0;
    for (int k = 1; k <= 28; ++k) {
      run += s_pre[grp][k];
      s_pre[grp][k] = run;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int total = s_pre[grp][28];
  int ex[kMaxExcl];
#pragma unroll
  for (int q = 0; q < kMaxExcl; ++q) ex[q] = excl[(size_t)i * kMaxExcl + q];
  // excluded partners are beads of the same molecule, whose indices lie in a short window around i: the eight
  // comparisons run only in sweeps where some candidate falls inside it
  int ex_lo = 0x7fffffff, ex_hi = -1;
#pragma unroll
  for (int q = 0; q < kMaxExcl; ++q)
    if (ex[q] >= 0) ex_lo = min(ex_lo, ex[q]), ex_hi = max(ex_hi, ex[q]);
  int* row = rows + (size_t)i * row_stride;
  constexpr unsigned int kGroupMask = (G == 32) ? 0xffffffffu : ((1u << G) - 1u);
  const unsigned int below = (1u << l) - 1u;
  int out = 0;
  int lo = 0;  // cell of this lane's candidate: only ever advances, t grows by G per sweep
  // The sweep body is written without nested branches (clamped reads, predicates folded into `hit`): as nested ifs it
  // compiled to a dozen exec-mask branches per sweep, and the kernel is bound by the instructions of its ~164 k
  // wave-sweeps (10.5 M candidates / 64), not by their memory traffic.
  const int n_direct = s_pre[grp][27];  // candidates that come from buckets; the rest is the spill list
  // kSweeps sweeps per iteration: their (clamped, unconditional) reads are issued together and waited for once.  With
  // four beads per wavefront a dense cell is ~47 sweeps of 16 for its group, and a group's chain of sweeps - each a
  // table look-up and a read away from its test - is what the kernel's duration follows.
#ifndef MYTHOS_MM_SWEEPS
#define MYTHOS_MM_SWEEPS 4
#endif
  constexpr int kSweeps = MYTHOS_MM_SWEEPS;
  for (int t0 = 0; t0 < total; t0 += kSweeps * G) {
    typename CellPlace<R>::type pj[kSweeps];
    int cell[kSweeps], tcs[kSweeps];
#pragma unroll
    for (int u = 0; u < kSweeps; ++u) {
      const int t = t0 + u * G + l;
      const int tc = t < total ? t : total - 1;
      while (s_pre[grp][lo + 1] <= tc) ++lo;
      cell[u] = lo;
      tcs[u] = tc;
      // position and index of the candidate arrive together, a contiguous stream per cell
      pj[u] = place[tc < n_direct ? s_st[grp][lo] + (tc - s_pre[grp][lo]) : 0];
    }
#pragma unroll
    for (int u = 0; u < kSweeps; ++u) {
      const int t = t0 + u * G + l;
      const bool live = t < total;
      const bool from_bucket = tcs[u] < n_direct;
      int j = cell_index_of(pj[u].w);
      if (!from_bucket) {  // spill list (rare): index, then a gather
        j = spill[tcs[u] - n_direct];
        const auto q = pos[j];
        pj[u].x = q.x, pj[u].y = q.y, pj[u].z = q.z;
      }
      bool keep = live & (j != i);
      if (j >= ex_lo && j <= ex_hi) keep = keep & !mm_excluded<R>(ex, j);  // one or two sweeps of a row
      if (!g.direct) {  // hashed table: a bucket may mix cells that collide, a candidate counts only for the cell it lies in
        int jx, jy, jz;
        cell_of(g, pj[u].x, pj[u].y, pj[u].z, jx, jy, jz);
        const int c = cell[u];
        keep = keep & (!from_bucket | (jx == s_c[grp][c][0] & jy == s_c[grp][c][1] & jz == s_c[grp][c][2]));
      }
      const R dx = wrap(pj[u].x - pi.x, K.lx, K.ilx), dy = wrap(pj[u].y - pi.y, K.ly, K.ily), dz = wrap(pj[u].z - pi.z, K.lz, K.ilz);
      const bool hit = keep & (dx * dx + dy * dy + dz * dz < rl2);
      const unsigned int m = (unsigned int)(__ballot(hit) >> gshift) & kGroupMask;
      const int slot = out + __popc(m & below);
      if (hit & (slot < row_stride)) row[slot] = j;
      out += __popc(m);
    }
  }
  if (l == 0) {
    if (out > row_stride) {
      atomicMax(overflow, out);
      out = row_stride;
    }
    row_len[i] = out;
    ref_pos[i] = pi;  // what the displacement check of the step kernel compares against
  }
}

template <typename R>
__global__ __launch_bounds__(256) void mm_build_rows_allpairs_kernel(
    int n, const typename Real4<R>::type* __restrict__ pos, const MmConst<R> K, R rl2, const int* __restrict__ excl,
    int* __restrict__ rows, int* __restrict__ row_len, int row_stride, int* __restrict__ overflow,
    typename Real4<R>::type* __restrict__ ref_pos) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + w;
  if (i >= n) return;
  const auto pi = pos[i];
  int ex[kMaxExcl];
#pragma unroll
  for (int q = 0; q < kMaxExcl; ++q) ex[q] = excl[(size_t)i * kMaxExcl + q];
  int* row = rows + (size_t)i * row_stride;
  int out = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < n && j != i && !mm_excluded<R>(ex, j)) {
      const auto pj = pos[j];
      const R dx = wrap(pj.x - pi.x, K.lx, K.ilx), dy = wrap(pj.y - pi.y, K.ly, K.ily), dz = wrap(pj.z - pi.z, K.lz, K.ilz);
      hit = dx * dx + dy * dy + dz * dz < rl2;
    }
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int slot = out + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < row_stride) row[slot] = j;
    }
    out += __popcll(m);
  }
  if (lane == 0) {
    if (out > row_stride) {
      atomicMax(overflow, out);
      out = row_stride;
    }
    row_len[i] = out;
    ref_pos[i] = pi;  // what the displacement check of the step kernel compares against
  }
}

}  // namespace mythos

using namespace mythos;

struct mythos_martini_sim {
  mythos_martini* sys = nullptr;
  double dt = 0.02, kT = 2.27, gamma = 1.0;
  uint64_t seed = 0;
  long long step = 0;
  double skin = 0.2;
  int rebuild_every = 10;
  void* frame[2] = {nullptr, nullptr};
  void *vel = nullptr, *ref_pos = nullptr, *d_inv_mass = nullptr;
  // Type tables of the step kernel: only the types that occur, renumbered 0 .. n_ctypes - 1 (the DMPC bilayer uses 4 of
  // the force field's 37: every workgroup filled 2 x 37^2 LDS entries per launch, 0.9 us of a 10.6 us kernel), sigma
  // already squared.  d_ctypes: the compact type of every bead (what the frames carry in .w).
  int n_ctypes = 0;
  int* d_ctypes = nullptr;
  void *d_csig2 = nullptr, *d_ceps = nullptr;
  // per incidence slot of a bead: the partner of the bond, the two other beads of the angle (see the step kernel)
  int* d_bb_partner = nullptr;
  int2* d_ba_partner = nullptr;
  void* d_angle_ref = nullptr;  // per angle: cos(theta0) for the G96 form (once, instead of a cosine per lane and step), theta0 for the harmonic one
  int *d_rows = nullptr, *d_row_len = nullptr, *d_cell = nullptr, *d_flags = nullptr, *d_overflow = nullptr;
  size_t cell_cap = 0;       // ints allocated at d_cell (cell_list.h CellBins: counters [2][H], buckets [H][cap])
  int cell_H = 0, cell_alloc_bucket_cap = 0, cell_bucket_cap = 64, cell_phase = 0;
  int row_stride = 256;
  // pruned rows (martini_md_step_kernel, EMIT): entries of the Verlet rows inside r_c + inner_margin, rewritten every
  // inner_every steps by the step launch itself; inner_margin <= 0 or inner_every < 2: not used
  int *d_rows_in = nullptr, *d_row_len_in = nullptr;
  int rows_in_stride = 0;
  double inner_margin = 0.0;  // off until mythos_martini_langevin_set_inner_list asks for them
  int inner_every = 4;
  double* d_epart = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  static constexpr int kMaxSamples = 16;
  int timing_samples = 0;  // dispatches per run timed with their own event pair (set_timing; ~8 us each)
  hipEvent_t sa[kMaxSamples] = {}, sb[kMaxSamples] = {};
  double last_kernel_ms = 0, last_avg_ms = 0;
  int last_launches = 0, last_samples = 0, last_max_row = 0, last_recoveries = 0;
  bool list_fitted = false;  // a synchronising, growing build has sized rows and buckets for this integrator
  // resident state (mythos_martini_langevin_load / advance / store): the frames hold a state between calls
  double box[3] = {0, 0, 0};
  bool resident = false;    // frame[cur] + vel hold a state
  bool list_valid = false;  // the rows were built from this state's history and the rebuild schedule continues
  bool open = false;        // x_n with velocities short of the closing half kick of step n (advance)
  int cur = 0;
  int since_build = 0;      // steps taken since the rows were built
  int last_rebuilds = 0;    // scheduled rebuilds inside the last advance
  int* h_ctl = nullptr;     // pinned: [0..3] d_flags, [4..6] d_overflow as the device published them
  int* d_ctl = nullptr;     // device address of h_ctl
};

namespace mythos {

template <typename R>
static bool upload_real_vec(void** dst, const std::vector<double>& src) {
  std::vector<R> tmp(src.size());
  for (size_t k = 0; k < src.size(); ++k) tmp[k] = R(src[k]);
  return hipMalloc(dst, std::max<size_t>(tmp.size(), 1) * sizeof(R)) == hipSuccess &&
         hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(R), hipMemcpyHostToDevice) == hipSuccess;
}

template <typename R>
static int mm_rebuild(mythos_martini_sim* sim, const typename Real4<R>::type* pos, const MmConst<R>& K, const double box[3],
                      hipStream_t st) {
  mythos_martini* m = sim->sys;
  const int n = m->n;
  const double rl = m->r_cut + sim->skin;
  CellGrid<R> g;
  bool cells_ok = n >= 512;
  for (int k = 0; k < 3; ++k) {
    const int nc = (int)std::floor(box[k] / rl);
    if (nc < 3) cells_ok = false;
    g.nc[k] = std::max(nc, 1);
    g.ibox[k] = R(1.0 / box[k]);
    g.inv[k] = R(g.nc[k] / box[k]);
  }
  const int wb = (n + 3) / 4;
  if (!cells_ok) {
    hipLaunchKernelGGL(mm_build_rows_allpairs_kernel<R>, dim3(wb), dim3(256), 0, st, n, pos, K, R(rl * rl), m->d_excl,
                       sim->d_rows, sim->d_row_len, sim->row_stride, sim->d_overflow, (typename Real4<R>::type*)sim->ref_pos);
  } else {
    // periodic grid: one table slot per cell (no hashing) whenever the grid is not much larger than the system
    const long long n_cells = (long long)g.nc[0] * g.nc[1] * g.nc[2];
    g.direct = n_cells <= 8LL * n ? 1 : 0;
    const int H = g.direct ? (int)n_cells : next_pow2(2 * n);
    if (cell_cap_override()) sim->cell_bucket_cap = cell_cap_override();
    const int cap = sim->cell_bucket_cap;
    const size_t need = CellBins::ints(H, cap, sizeof(R));
    if (need > sim->cell_cap || H != sim->cell_H || cap != sim->cell_alloc_bucket_cap) {
      if (sim->d_cell) (void)hipFree(sim->d_cell);
      sim->d_cell = nullptr;
      sim->cell_cap = 0;
      MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_cell, need * sizeof(int)));
      MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_cell + CellBins::zero_offset(H, cap, sizeof(R)), 0, CellBins::zero_ints(H) * sizeof(int), st));
      sim->cell_cap = need;
      sim->cell_H = H;
      sim->cell_alloc_bucket_cap = cap;
      sim->cell_phase = 0;
    }
    const CellBins bins(sim->d_cell, H, cap, sizeof(R), sim->cell_phase);
    sim->cell_phase ^= 1;
    // buckets sorted by bead index: the row builder copies candidates in bucket order, and rows must not depend on
    // the order in which the binning atomics landed
    cell_bins_build<R, true>(n, reinterpret_cast<const R*>(pos), g, bins, sim->d_overflow, true, st);
hipLaunchKernelGGL((mm_build_rows_cells_kernel<R, kMmRowG>), dim3((n + 256 / kMmRowG - 1) / (256 / kMmRowG)), dim3(256), 0, st, n, pos, K, g, R(rl * rl), m->d_excl,
                       bins.cnt_cur, (const typename CellPlace<R>::type*)bins.place, bins.cap, bins.spill, bins.H, sim->d_rows, sim->d_row_len, sim->row_stride, sim->d_overflow, (typename Real4<R>::type*)sim->ref_pos);
  }
  return 0;
}

// End of a segment: hand the control words to the host (pinned memory), as the oxDNA integrator does (langevin_core.inc).
static __global__ void mm_publish_ctl_kernel(const int* __restrict__ flags, const int* __restrict__ overflow, int* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = flags[0], out[1] = flags[1], out[2] = flags[2], out[3] = flags[3];
  out[4] = overflow[0], out[5] = overflow[1], out[6] = overflow[2];
}

// pruned rows make sense inside the skin and when at least one launch walks them between two prunings
static bool mm_inner_on(const mythos_martini_sim* sim) {
  return sim->inner_margin > 0 && sim->inner_margin < sim->skin && sim->inner_every >= 2;
}

template <typename R>
static MmConst<R> mm_const(const mythos_martini_sim* sim) {
  const mythos_martini* m = sim->sys;
  const double* box = sim->box;
  MmConst<R> K;
  K.lx = R(box[0]), K.ly = R(box[1]), K.lz = R(box[2]);
  K.ilx = R(1.0 / box[0]), K.ily = R(1.0 / box[1]), K.ilz = R(1.0 / box[2]);
  K.rc2 = R(m->r_cut * m->r_cut);
  K.dt = R(sim->dt), K.half_dt = R(0.5 * sim->dt), K.c1 = R(std::exp(-sim->gamma * sim->dt)), K.kT = R(sim->kT);
  K.skin_half_sq = R(0.25 * sim->skin * sim->skin);
  const bool inner = mm_inner_on(sim);
  const double rin = m->r_cut + sim->inner_margin, smax = 0.5 * sim->inner_margin / std::max(1, sim->inner_every - 1);
  K.rin2 = inner ? R(rin * rin) : R(0);
  K.step_max_sq = inner ? R(smax * smax) : R(0);
  K.n_types = sim->n_ctypes, K.angle_kind = m->angle_kind;
  return K;
}

// caller's (n, 3) arrays -> the resident frame; the list of a previous state does not carry over
template <typename R>
static int mm_load_typed(mythos_martini_sim* sim, const R* pos, const R* v, const double box[3], hipStream_t st) {
  using V4 = typename Real4<R>::type;
  mythos_martini* m = sim->sys;
  const int n = m->n, tb = (n + 255) / 256;
  for (int k = 0; k < 3; ++k) sim->box[k] = box[k];
  hipLaunchKernelGGL(mm_pack_kernel<R>, dim3(tb), dim3(256), 0, st, n, pos, v, sim->d_ctypes, (const R*)sim->d_inv_mass,
                     (V4*)sim->frame[0], (V4*)sim->vel);
  MYTHOS_HIP_TRY(hipGetLastError());
  sim->cur = 0;
  sim->resident = true;
  sim->list_valid = false;
  sim->open = false;
  sim->since_build = 0;
  return MYTHOS_OK;
}

// n_steps on the resident state (the protocol of advance_typed in langevin_core.inc).  Launch k evaluates the forces at
// x_k, closes step k - 1 with them (second half kick) and takes step k up to its drift.  close = true: n_steps + 1
// launches, the last one only closes.  close = false (mythos_martini_langevin_advance): n_steps launches, the frame is
// left OPEN and whoever comes next supplies the closing half kick with the force evaluation it needs anyway - the next
// advance in its first launch (advance(a); advance(b) is advance(a + b) launch for launch), store through a zero-step
// closing call.  Rows with energies (e_trace != NULL) come from the energy-trace instantiation at x_k plus a reduction
// launch; rows without are written by the launch that produces the saved state.
template <typename R>
static int mm_advance_typed(mythos_martini_sim* sim, int n_steps, int save_every, bool close, R* traj_pos, double* e_trace,
                            hipStream_t st) {
  using V4 = typename Real4<R>::type;
  mythos_martini* m = sim->sys;
  const int n = m->n;
  const int blocks = (n + kMmPPB - 1) / kMmPPB, grid = 8 * ((blocks + 7) / 8);
  const MmConst<R> K = mm_const<R>(sim);
  const double* box = sim->box;
  V4* fr[2] = {(V4*)sim->frame[0], (V4*)sim->frame[1]};
  V4* vel = (V4*)sim->vel;
  const bool energy_rows = save_every > 0 && e_trace != nullptr;
  const bool plain_rows = save_every > 0 && e_trace == nullptr && traj_pos != nullptr;
  const bool closes = close || (energy_rows && n_steps > 0 && n_steps % save_every == 0);
  const int n_launch = closes ? n_steps + 1 : n_steps;
  if (n_launch == 0) return MYTHOS_OK;
  const bool was_open = sim->open;
  const int cur0 = sim->cur;
  int cur = cur0;
  MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags, 0, 4 * sizeof(int), st));
  // A build that may grow: rows until the longest fits with a quarter of headroom, buckets until none is more than
  // half full (fuller ones work, through the spill list, but slowly).  Used for the first build of an integrator and to
  // recover from a halt (see below); the scheduled builds inside a call cannot stop to grow.
  auto build_until_fit = [&](int buf) -> int {
    for (int attempt = 0;; ++attempt) {
      MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_overflow, 0, 3 * sizeof(int), st));
      if (int rc = mm_rebuild<R>(sim, fr[buf], K, box, st)) return rc;
      int ov0[3] = {0, 0, 0};
      MYTHOS_HIP_TRY(hipMemcpyAsync(ov0, sim->d_overflow, sizeof(ov0), hipMemcpyDeviceToHost, st));
      MYTHOS_HIP_TRY(hipStreamSynchronize(st));
      if (ov0[1] > 0) {
        set_error("mythos_martini_langevin_run: more than " + std::to_string(kCellSpill) +
                  " beads did not fit the buckets of their cells during a neighbour rebuild");
        return MYTHOS_ERR_OVERFLOW;
      }
      const int demand = cell_cap_override() ? 0 : ov0[2];  // a bucket more than half full: double the places
      if (ov0[0] == 0 && demand == 0) {
        if (ov0[2] > 0) MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_overflow + 2, 0, sizeof(int), st));
        return 0;
      }
      if (attempt == 5) {
        set_error("mythos_martini_langevin_run: neighbour rows or cell buckets keep overflowing");
        return MYTHOS_ERR_OVERFLOW;
      }
      if (ov0[0] > 0) {
        const int stride = ((ov0[0] + ov0[0] / 4 + 15) / 16) * 16;
        (void)hipFree(sim->d_rows);
        sim->d_rows = nullptr;
        MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_rows, (size_t)n * stride * sizeof(int)));
        sim->row_stride = stride;
      }
      if (demand > 0) sim->cell_bucket_cap = ((2 * demand + 15) / 16) * 16;
    }
  };
  // k index at which the rows in use were built (negative: so many steps before this call)
  int built_at = 0;
  if (!sim->list_fitted) {
    if (int rc = build_until_fit(cur)) return rc;
    sim->list_fitted = true;
  } else if (!sim->list_valid) {
    MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_overflow, 0, 3 * sizeof(int), st));
    if (int rc = mm_rebuild<R>(sim, fr[cur], K, box, st)) return rc;
  } else {
    built_at = -sim->since_build;
  }
  sim->list_valid = true;
  const bool inner_on = mm_inner_on(sim);
  // the pruned rows share the Verlet rows' stride (a pruned row is a subset of its row: it cannot overflow)
  auto ensure_inner = [&]() -> int {
    if (!inner_on) return 0;
    if (!sim->d_row_len_in) MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_row_len_in, (size_t)n * sizeof(int)));
    if (!sim->d_rows_in || sim->rows_in_stride != sim->row_stride) {
      if (sim->d_rows_in) (void)hipFree(sim->d_rows_in);
      sim->d_rows_in = nullptr;
      MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_rows_in, (size_t)n * sim->row_stride * sizeof(int)));
      sim->rows_in_stride = sim->row_stride;
    }
    return 0;
  };
  if (int rc = ensure_inner()) return rc;
  const size_t lds = (size_t)2 * sim->n_ctypes * sim->n_ctypes * sizeof(R);
  const bool timing = sim->timing_samples > 0;
  if (timing) MYTHOS_HIP_TRY(hipEventRecord(sim->ev0, st));
  int launches = 0, samples = 0, recoveries = 0, scheduled = 0;
  const int max_samples = std::min(sim->timing_samples, (int)mythos_martini_sim::kMaxSamples);  // 0: no dispatch is bracketed
  const int sample_stride = std::max(1, n_launch / std::max(1, max_samples));
  // Segments of kSegment launches; a halted segment (a bead left its skin before the scheduled rebuild, or a rebuild
  // overflowed: the launches behind it return at once) is followed by a growing rebuild at the last valid state and
  // a resume there - the protocol of mythos_langevin_run (langevin_core.inc).
  constexpr int kMaxRecoveries = 64;
  const long long dbg_seg = debug_value(MYTHOS_DEBUG_MD_SEGMENT);
  const int kSegment = dbg_seg > 0 ? (int)std::min<long long>(dbg_seg, 1 << 20) : 8192;
  int k = 0, seg_len = kSegment;  // a call that has halted once looks more often: less queued behind the next halt
  int err_bits = 0, ovw[3] = {0, 0, 0};
  while (k < n_launch) {
    const int seg_end = std::min(n_launch - 1, k + seg_len - 1);
    for (; k <= seg_end; ++k) {
      const bool last = (k == n_steps);  // (reached only by a call that closes)
      const bool save = energy_rows && k > 0 && (k % save_every == 0);
      const bool save_next = plain_rows && !last && ((k + 1) % save_every == 0);  // this launch's OUTPUT is a saved state
      const int sidx = save ? (k / save_every - 1) : (save_next ? ((k + 1) / save_every - 1) : 0);
      // (a closing-only launch rebuilds too when the schedule says so - as advance_typed in langevin_core.inc)
      if (k - built_at >= sim->rebuild_every) {
        if (int rc = mm_rebuild<R>(sim, fr[cur], K, box, st)) return rc;
        built_at = k;
        ++scheduled;
      }
      const R kick_close = (k == 0 && !was_open) ? R(0) : R(0.5);
      const int do_step = last ? 0 : 1;
      R* tp = ((save || save_next) && traj_pos) ? traj_pos + (size_t)sidx * n * 3 : nullptr;
      const bool sampled = !save && (k % sample_stride == sample_stride / 2) && samples < max_samples;
      // pruned rows: the launch d steps after the Verlet rows were built prunes when d is a multiple of inner_every (so
      // the first launch on new rows does) and walks the pruned rows otherwise
      const bool emit = inner_on && ((k - built_at) % sim->inner_every == 0);
      const int* walk_rows = (inner_on && !emit) ? sim->d_rows_in : sim->d_rows;
      const int* walk_len = (inner_on && !emit) ? sim->d_row_len_in : sim->d_row_len;
      int* emit_rows = emit ? sim->d_rows_in : nullptr;
      int* emit_len = emit ? sim->d_row_len_in : nullptr;
#define MM_ARGS                                                                                                    \
  n, K, (const V4*)fr[cur], fr[cur ^ 1], vel, walk_rows, walk_len, sim->row_stride, (const R*)sim->d_csig2,          \
      (const R*)sim->d_ceps, m->d_bead_bonds, m->d_bead_angles, m->d_bonds, (const R*)m->d_bond_k,                      \
      (const R*)m->d_bond_r0, m->d_angles, (const R*)m->d_angle_k, (const R*)sim->d_angle_ref, kick_close, do_step,  \
      sim->seed, (uint64_t)(sim->step + k), (const V4*)sim->ref_pos, sim->d_flags, tp, sim->d_epart, sim->d_overflow, k, \
      emit_rows, emit_len, sim->d_bb_partner, sim->d_ba_partner
      auto go = [&](auto save_tag, auto emit_tag) {
        constexpr bool SV = decltype(save_tag)::value, EM = decltype(emit_tag)::value;
        if (sampled) {
          hipExtLaunchKernelGGL((martini_md_step_kernel<R, SV, EM>), dim3(grid), dim3(kMmBlock), lds, st, sim->sa[samples],
                                sim->sb[samples], 0, MM_ARGS);
        } else {
          hipLaunchKernelGGL((martini_md_step_kernel<R, SV, EM>), dim3(grid), dim3(kMmBlock), lds, st, MM_ARGS);
        }
      };
      if (save) {
        if (emit) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{});
        hipLaunchKernelGGL(mm_reduce_trace_kernel, dim3(1), dim3(256), 0, st, sim->d_epart, blocks,
                           e_trace + (size_t)sidx * kMmTrace);
      } else {
        if (emit) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{});
        if (sampled) ++samples;
      }
#undef MM_ARGS
      ++launches;
      cur ^= 1;
    }
    if (timing && k >= n_launch) MYTHOS_HIP_TRY(hipEventRecord(sim->ev1, st));
    hipLaunchKernelGGL(mm_publish_ctl_kernel, dim3(1), dim3(1), 0, st, (const int*)sim->d_flags, (const int*)sim->d_overflow, sim->d_ctl);
    MYTHOS_HIP_TRY(hipGetLastError());
    MYTHOS_HIP_TRY(hipStreamSynchronize(st));
    const int* ctl = sim->h_ctl;
    err_bits |= ctl[0];
    for (int w = 0; w < 3; ++w) ovw[w] = ctl[4 + w];
    if ((err_bits & 2) != 0) break;                               // NaN: reported below
    if (ctl[1] == 0 && ovw[0] == 0 && ovw[1] == 0) continue;      // nothing halted
    const int ran = ctl[2];  // kernels 0 .. ran-1 ran; the state they left is in the frame kernel `ran` reads
    if (++recoveries > kMaxRecoveries) {
      sim->cur = cur0 ^ (ran & 1);
      sim->step += ran;
      sim->open = was_open || ran > 0;
      sim->list_valid = false;
      (void)hipMemsetAsync(sim->d_flags, 0, 4 * sizeof(int), st);
      set_error("mythos_martini_langevin_run: the neighbour list had to be rebuilt out of turn more than " +
                std::to_string(kMaxRecoveries) + " times in one run: the skin (" + std::to_string(sim->skin) +
                ") is too small for a rebuild every " + std::to_string(sim->rebuild_every) + " steps" +
                (mm_inner_on(sim) ? ", or the margin of the pruned rows (" + std::to_string(sim->inner_margin) +
                                        ") for one pruning in " + std::to_string(sim->inner_every) + " steps"
                                  : std::string()));
      return MYTHOS_ERR_OVERFLOW;
    }
    k = ran;
    cur = cur0 ^ (k & 1);
    seg_len = std::max(std::min(256, kSegment), seg_len / 4);
    MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags + 1, 0, sizeof(int), st));
    if (int rc = build_until_fit(cur)) return rc;
    if (int rc = ensure_inner()) return rc;  // (the rows may have grown)
    built_at = k;
    ovw[0] = ovw[1] = 0;
  }
  sim->last_recoveries = recoveries;
  sim->last_rebuilds = scheduled;
  sim->cur = cur;
  sim->open = !closes;
  sim->since_build = n_steps - built_at;
  if (timing) {
    float ms = 0;
    MYTHOS_HIP_TRY(hipEventElapsedTime(&ms, sim->ev0, sim->ev1));
    sim->last_avg_ms = launches ? double(ms) / launches : 0.0;
    double acc = 0;
    for (int q = 0; q < samples; ++q) {
      float t = 0;
      MYTHOS_HIP_TRY(hipEventElapsedTime(&t, sim->sa[q], sim->sb[q]));
      acc += t;
    }
    sim->last_kernel_ms = samples ? acc / samples : 0.0;
  } else {
    sim->last_avg_ms = sim->last_kernel_ms = 0.0;
  }
  sim->last_launches = launches;
  sim->last_samples = samples;
  sim->step += n_steps;
  if (err_bits & 2) {
    sim->resident = false;
    set_error("mythos_martini_langevin_run: NaN in the state (time step too large or overlapping start configuration)");
    return MYTHOS_ERR_NUMERIC;
  }
  if (ovw[0] != 0) {
    set_error("mythos_martini_langevin_run: neighbour row capacity exceeded (" + std::to_string(ovw[0]) + " > " +
              std::to_string(sim->row_stride) + ")");
    return MYTHOS_ERR_OVERFLOW;
  }
  if (ovw[1] != 0) {
    set_error("mythos_martini_langevin_run: more than " + std::to_string(kCellSpill) +
              " beads did not fit the buckets of their cells during a neighbour rebuild");
    return MYTHOS_ERR_OVERFLOW;
  }
  return MYTHOS_OK;
}

// the resident frame -> caller's arrays (asynchronous on st); an open frame gets its closing half kick first
template <typename R>
static int mm_store_typed(mythos_martini_sim* sim, R* pos, R* v, hipStream_t st) {
  using V4 = typename Real4<R>::type;
  const int n = sim->sys->n;
  const int rc = sim->open ? mm_advance_typed<R>(sim, 0, 0, true, nullptr, nullptr, st) : MYTHOS_OK;
  hipLaunchKernelGGL(mm_unpack_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, st, n, (const V4*)sim->frame[sim->cur], (const V4*)sim->vel, pos, v);
  MYTHOS_HIP_TRY(hipGetLastError());
  return rc;
}

}  // namespace mythos

extern "C" {

void mythos_martini_langevin_destroy(mythos_martini_sim_t* s) {
  if (!s) return;
  (void)hipSetDevice(s->sys->device);
  void* ptrs[] = {s->frame[0], s->frame[1], s->vel, s->ref_pos, s->d_inv_mass, s->d_rows, s->d_row_len,
                  s->d_cell,   s->d_flags,  s->d_overflow, s->d_epart, s->d_rows_in, s->d_row_len_in, s->d_angle_ref, s->d_ctypes, s->d_csig2, s->d_ceps, s->d_bb_partner, s->d_ba_partner};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (s->h_ctl) (void)hipHostFree(s->h_ctl);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  for (int k = 0; k < mythos_martini_sim::kMaxSamples; ++k) {
    if (s->sa[k]) (void)hipEventDestroy(s->sa[k]);
    if (s->sb[k]) (void)hipEventDestroy(s->sb[k]);
  }
  delete s;
}

mythos_martini_sim_t* mythos_martini_langevin_create(mythos_martini_t* sys, double dt, double kT, double gamma,
                                                     const double* mass, uint64_t seed) {
  if (!sys || !(dt > 0) || !(kT > 0) || !(gamma >= 0)) {
    set_error("mythos_martini_langevin_create: invalid argument");
    return nullptr;
  }
  if (hipSetDevice(sys->device) != hipSuccess) {
    set_error("mythos_martini_langevin_create: hipSetDevice failed");
    return nullptr;
  }
  auto* s = new mythos_martini_sim();
  s->sys = sys, s->dt = dt, s->kT = kT, s->gamma = gamma, s->seed = seed;
  const int n = sys->n;
  const size_t w = sys->dtype == MYTHOS_F32 ? sizeof(float) : sizeof(double);
  std::vector<double> im(n);
  for (int i = 0; i < n; ++i) {
    const double mi = mass ? mass[i] : 72.0;  // MARTINI's standard bead mass (amu)
    if (!(mi > 0)) {
      set_error("mythos_martini_langevin_create: masses must be positive");
      delete s;
      return nullptr;
    }
    im[i] = 1.0 / mi;
  }
  const int blocks = (n + kMmPPB - 1) / kMmPPB;
  bool ok = hipMalloc(&s->frame[0], (size_t)n * 4 * w) == hipSuccess && hipMalloc(&s->frame[1], (size_t)n * 4 * w) == hipSuccess &&
            hipMalloc(&s->vel, (size_t)n * 4 * w) == hipSuccess && hipMalloc(&s->ref_pos, (size_t)n * 4 * w) == hipSuccess &&
            hipMalloc((void**)&s->d_rows, (size_t)n * s->row_stride * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_row_len, (size_t)n * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_flags, 4 * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_overflow, 3 * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_epart, (size_t)blocks * kMmTrace * sizeof(double)) == hipSuccess;
  ok = ok && (sys->dtype == MYTHOS_F32 ? upload_real_vec<float>(&s->d_inv_mass, im) : upload_real_vec<double>(&s->d_inv_mass, im));
  if (ok) {  // partners per incidence slot
    std::vector<int> bb((size_t)n * kMaxBeadBonds), ba((size_t)n * kMaxBeadAngles), bonds((size_t)2 * std::max(sys->n_bonds, 0)),
        angles((size_t)3 * std::max(sys->n_angles, 0));
    ok = hipMemcpy(bb.data(), sys->d_bead_bonds, bb.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(ba.data(), sys->d_bead_angles, ba.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
         (bonds.empty() || hipMemcpy(bonds.data(), sys->d_bonds, bonds.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) &&
         (angles.empty() || hipMemcpy(angles.data(), sys->d_angles, angles.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess);
    std::vector<int> pb(bb.size(), -1);
    std::vector<int2> pa(ba.size(), int2{-1, -1});
    for (size_t k = 0; ok && k < bb.size(); ++k)
      if (bb[k] >= 0) pb[k] = bonds[2 * (size_t)(bb[k] >> 1) + (1 - (bb[k] & 1))];
    for (size_t k = 0; ok && k < ba.size(); ++k)
      if (ba[k] >= 0) {
        const int* t = &angles[3 * (size_t)(ba[k] >> 2)];
        const int role = ba[k] & 3;
        pa[k] = role == 0 ? int2{t[1], t[2]} : (role == 1 ? int2{t[0], t[2]} : int2{t[0], t[1]});
      }
    ok = ok && hipMalloc((void**)&s->d_bb_partner, std::max<size_t>(pb.size(), 1) * sizeof(int)) == hipSuccess &&
         hipMalloc((void**)&s->d_ba_partner, std::max<size_t>(pa.size(), 1) * sizeof(int2)) == hipSuccess &&
         hipMemcpy(s->d_bb_partner, pb.data(), pb.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(s->d_ba_partner, pa.data(), pa.size() * sizeof(int2), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok) {  // the compact type tables
    const int T = sys->n_types;
    std::vector<int> types(n), cmap(T, -1), used;
    ok = hipMemcpy(types.data(), sys->d_types, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    std::vector<double> sig((size_t)T * T), ep((size_t)T * T);
    if (sys->dtype == MYTHOS_F32) {
      std::vector<float> a((size_t)T * T), b((size_t)T * T);
      ok = ok && hipMemcpy(a.data(), sys->d_sigma, a.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(b.data(), sys->d_eps, b.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
      for (size_t k = 0; k < a.size(); ++k) sig[k] = a[k], ep[k] = b[k];
    } else {
      ok = ok && hipMemcpy(sig.data(), sys->d_sigma, sig.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(ep.data(), sys->d_eps, ep.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (ok) {
      for (int t : types)
        if (t >= 0 && t < T && cmap[t] < 0) cmap[t] = 0;
      for (int t = 0; t < T; ++t)
        if (cmap[t] == 0) cmap[t] = (int)used.size(), used.push_back(t);
      const int C = std::max<int>(1, (int)used.size());
      std::vector<double> cs((size_t)C * C, 0.0), ce((size_t)C * C, 0.0);
      for (size_t p = 0; p < used.size(); ++p)
        for (size_t q = 0; q < used.size(); ++q) {
          // (squared in the kernel's own precision, as the kernel did)
          const double sg = sig[(size_t)used[p] * T + used[q]];
          cs[p * C + q] = sys->dtype == MYTHOS_F32 ? double(float(sg) * float(sg)) : sg * sg;
          ce[p * C + q] = ep[(size_t)used[p] * T + used[q]];
        }
      for (int& t : types) t = (t >= 0 && t < T) ? cmap[t] : 0;
      s->n_ctypes = C;
      ok = hipMalloc((void**)&s->d_ctypes, (size_t)n * sizeof(int)) == hipSuccess &&
           hipMemcpy(s->d_ctypes, types.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
           (sys->dtype == MYTHOS_F32 ? upload_real_vec<float>(&s->d_csig2, cs) && upload_real_vec<float>(&s->d_ceps, ce)
                                     : upload_real_vec<double>(&s->d_csig2, cs) && upload_real_vec<double>(&s->d_ceps, ce));
    }
  }
  {
    std::vector<double> ref((size_t)std::max(sys->n_angles, 0));
    if (ok && sys->n_angles > 0) {
      if (sys->dtype == MYTHOS_F32) {
        std::vector<float> t0(ref.size());
        ok = hipMemcpy(t0.data(), sys->d_angle_t0, t0.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
        for (size_t a = 0; a < ref.size(); ++a) ref[a] = sys->angle_kind == 0 ? std::cos(double(t0[a])) : double(t0[a]);
      } else {
        ok = hipMemcpy(ref.data(), sys->d_angle_t0, ref.size() * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        if (sys->angle_kind == 0)
          for (double& v : ref) v = std::cos(v);
      }
    }
    ok = ok && (sys->dtype == MYTHOS_F32 ? upload_real_vec<float>(&s->d_angle_ref, ref) : upload_real_vec<double>(&s->d_angle_ref, ref));
  }
  ok = ok && hipEventCreate(&s->ev0) == hipSuccess && hipEventCreate(&s->ev1) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&s->h_ctl, 8 * sizeof(int), hipHostMallocDefault) == hipSuccess &&
       hipHostGetDevicePointer((void**)&s->d_ctl, s->h_ctl, 0) == hipSuccess && hipMemset(s->d_overflow, 0, 3 * sizeof(int)) == hipSuccess;
  if (ok) std::fill(s->h_ctl, s->h_ctl + 8, 0);
  for (int k = 0; ok && k < mythos_martini_sim::kMaxSamples; ++k)
    ok = hipEventCreate(&s->sa[k]) == hipSuccess && hipEventCreate(&s->sb[k]) == hipSuccess;
  if (!ok) {
    set_error("mythos_martini_langevin_create: device allocation failed");
    mythos_martini_langevin_destroy(s);
    return nullptr;
  }
  return s;
}

int mythos_martini_langevin_set_neighbor_policy(mythos_martini_sim_t* s, double skin, int rebuild_every) {
  if (!s || !(skin > 0) || rebuild_every < 1) {
    set_error("mythos_martini_langevin_set_neighbor_policy: skin > 0 and rebuild_every >= 1 required");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->skin = skin;
  s->rebuild_every = rebuild_every;
  s->list_fitted = false;  // another list range: size rows and buckets again at the next run
  s->list_valid = false;
  return MYTHOS_OK;
}

int mythos_martini_langevin_set_inner_list(mythos_martini_sim_t* s, double margin, int every) {
  if (!s || every < 0) {
    set_error("mythos_martini_langevin_set_inner_list: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->inner_margin = margin;
  s->inner_every = every;
  s->list_valid = false;  // the schedule of pruning starts again with the next build
  return MYTHOS_OK;
}

int mythos_martini_langevin_init_velocities(mythos_martini_sim_t* s, void* vel, mythos_stream_t stream) {
  if (!s || !vel) {
    set_error("mythos_martini_langevin_init_velocities: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  if (s->sys->dtype == MYTHOS_F32)
    hipLaunchKernelGGL(mm_init_velocities_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, s->sys->n, float(s->kT),
                       (const float*)s->d_inv_mass, s->seed, (float*)vel);
  else
    hipLaunchKernelGGL(mm_init_velocities_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, s->sys->n, s->kT,
                       (const double*)s->d_inv_mass, s->seed, (double*)vel);
  MYTHOS_HIP_TRY(hipGetLastError());
  return MYTHOS_OK;
}

namespace {

int mm_check_box(const mythos_martini_sim_t* s, const double* box, const char* who) {
  if (!box || !(box[0] > 0) || !(box[1] > 0) || !(box[2] > 0)) {
    set_error(std::string(who) + ": invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  const double rl = s->sys->r_cut + s->skin;
  if (2.0 * rl > std::min(box[0], std::min(box[1], box[2]))) {
    set_error(std::string(who) + ": the box is smaller than twice (r_cut + skin): minimum image breaks down");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  return MYTHOS_OK;
}

int mm_load(mythos_martini_sim_t* s, const void* pos, const void* vel, const double* box, hipStream_t st) {
  return s->sys->dtype == MYTHOS_F32 ? mm_load_typed<float>(s, (const float*)pos, (const float*)vel, box, st)
                                     : mm_load_typed<double>(s, (const double*)pos, (const double*)vel, box, st);
}
int mm_advance(mythos_martini_sim_t* s, int n_steps, int save_every, bool close, void* traj, double* e_trace, hipStream_t st) {
  return s->sys->dtype == MYTHOS_F32 ? mm_advance_typed<float>(s, n_steps, save_every, close, (float*)traj, e_trace, st)
                                     : mm_advance_typed<double>(s, n_steps, save_every, close, (double*)traj, e_trace, st);
}
int mm_store(mythos_martini_sim_t* s, void* pos, void* vel, hipStream_t st) {
  return s->sys->dtype == MYTHOS_F32 ? mm_store_typed<float>(s, (float*)pos, (float*)vel, st) : mm_store_typed<double>(s, (double*)pos, (double*)vel, st);
}

}  // namespace

int mythos_martini_langevin_run(mythos_martini_sim_t* s, void* pos, void* vel, const double* box, int n_steps,
                                int save_every, void* traj_pos, double* e_trace, mythos_stream_t stream) {
  if (!s || !pos || !vel || n_steps < 0 || save_every < 0) {
    set_error("mythos_martini_langevin_run: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (int rc = mm_check_box(s, box, "mythos_martini_langevin_run")) return rc;
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  hipStream_t st = (hipStream_t)stream;
  if (int rc = mm_load(s, pos, vel, box, st)) return rc;
  const int rc = mm_advance(s, n_steps, save_every, true, traj_pos, e_trace, st);
  // the state of the last valid step goes back to the caller whatever the run reported
  if (s->resident) {
    if (int rs = mm_store(s, pos, vel, st)) return rc ? rc : rs;
    MYTHOS_HIP_TRY(hipStreamSynchronize(st));
  }
  return rc;
}

int mythos_martini_langevin_load(mythos_martini_sim_t* s, const void* pos, const void* vel, const double* box, mythos_stream_t stream) {
  if (!s || !pos || !vel) {
    set_error("mythos_martini_langevin_load: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (int rc = mm_check_box(s, box, "mythos_martini_langevin_load")) return rc;
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  return mm_load(s, pos, vel, box, (hipStream_t)stream);
}

int mythos_martini_langevin_advance(mythos_martini_sim_t* s, int n_steps, int save_every, void* traj_pos, double* e_trace,
                                    mythos_stream_t stream) {
  if (!s || n_steps < 0 || save_every < 0) {
    set_error("mythos_martini_langevin_advance: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->resident) {
    set_error("mythos_martini_langevin_advance: no resident state (call mythos_martini_langevin_load first; a run that ended in a "
              "numeric error drops its state)");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  return mm_advance(s, n_steps, save_every, false, traj_pos, e_trace, (hipStream_t)stream);
}

int mythos_martini_langevin_store(mythos_martini_sim_t* s, void* pos, void* vel, mythos_stream_t stream) {
  if (!s || !pos || !vel) {
    set_error("mythos_martini_langevin_store: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->resident) {
    set_error("mythos_martini_langevin_store: no resident state");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  return mm_store(s, pos, vel, (hipStream_t)stream);
}

int64_t mythos_martini_langevin_get_step(const mythos_martini_sim_t* s) { return s ? (int64_t)s->step : -1; }

int mythos_martini_langevin_last_rebuilds(const mythos_martini_sim_t* s, int* scheduled) {
  if (!s || !scheduled) {
    set_error("mythos_martini_langevin_last_rebuilds: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  *scheduled = s->last_rebuilds;
  return MYTHOS_OK;
}

int mythos_martini_langevin_last_kernel_ms(const mythos_martini_sim_t* s, double* kernel_ms, double* loop_ms_per_launch,
                                           int* launches, int* samples) {
  if (!s) {
    set_error("mythos_martini_langevin_last_kernel_ms: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (kernel_ms) *kernel_ms = s->last_kernel_ms;
  if (loop_ms_per_launch) *loop_ms_per_launch = s->last_avg_ms;
  if (launches) *launches = s->last_launches;
  if (samples) *samples = s->last_samples;
  return MYTHOS_OK;
}

int mythos_martini_langevin_neighbor_stats(const mythos_martini_sim_t* s, int* max_row, double* mean_row) {
  if (!s) {
    set_error("mythos_martini_langevin_neighbor_stats: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  std::vector<int> len(s->sys->n);
  MYTHOS_HIP_TRY(hipMemcpy(len.data(), s->d_row_len, len.size() * sizeof(int), hipMemcpyDeviceToHost));
  long long tot = 0;
  int mx = 0;
  for (int v : len) tot += v, mx = std::max(mx, v);
  if (max_row) *max_row = mx;
  if (mean_row) *mean_row = double(tot) / std::max<size_t>(1, len.size());
  return MYTHOS_OK;
}

int mythos_martini_langevin_get_rows(const mythos_martini_sim_t* s, int which, int32_t* rows, int32_t* row_len, int* stride) {
  if (!s || (which != 0 && which != 1)) {
    set_error("mythos_martini_langevin_get_rows: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  const int* d_rows = which == 0 ? s->d_rows : s->d_rows_in;
  const int* d_len = which == 0 ? s->d_row_len : s->d_row_len_in;
  if (stride) *stride = s->row_stride;
  if (!rows && !row_len) return MYTHOS_OK;
  if (!d_rows || !d_len || !s->list_valid) {
    set_error(which == 0 ? "mythos_martini_langevin_get_rows: no list has been built"
                         : "mythos_martini_langevin_get_rows: no pruned rows (switched off, or no step taken yet)");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  MYTHOS_HIP_TRY(hipDeviceSynchronize());
  if (rows) MYTHOS_HIP_TRY(hipMemcpy(rows, d_rows, (size_t)s->sys->n * s->row_stride * sizeof(int), hipMemcpyDeviceToHost));
  if (row_len) MYTHOS_HIP_TRY(hipMemcpy(row_len, d_len, (size_t)s->sys->n * sizeof(int), hipMemcpyDeviceToHost));
  return MYTHOS_OK;
}

int mythos_martini_langevin_last_recoveries(const mythos_martini_sim_t* s, int* recoveries) {
  if (!s || !recoveries) {
    set_error("mythos_martini_langevin_last_recoveries: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  *recoveries = s->last_recoveries;
  return MYTHOS_OK;
}

int mythos_martini_langevin_set_timing(mythos_martini_sim_t* s, int samples) {
  if (!s || samples < 0) {
    set_error("mythos_martini_langevin_set_timing: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->timing_samples = std::min(samples, (int)mythos_martini_sim::kMaxSamples);
  return MYTHOS_OK;
}

}  // extern "C"
