// Per-nucleotide gather of all pair interactions: the work decomposition of every oxDNA kernel.
//
// A group of G consecutive lanes (G = 8..64, a divisor of the 64-wide wavefront) owns one
// nucleotide.  Each lane walks the nucleotide's neighbour row with stride G, evaluates the
// ordered pair, keeps the half that belongs to the owner, and the group folds the partial
// force / axis gradients / energies with DPP shuffles.  No atomics, results are bitwise
// reproducible, rows are read as G consecutive ints (coalesced per group) and neighbour state
// comes through L2 (a 12 kbp duplex is 0.7 MB of fp32 state).  Each pair is evaluated from
// both ends (energies and parameter partials are therefore weighted 1/2).
#pragma once
#include "mythos_internal.h"
#include "oxdna_pair.h"
#include "wave_ops.h"

namespace mythos {
inline namespace MYTHOS_MATH_NS {  // (see oxdna_math.h)

template <typename R>
struct Vec4T;
template <>
struct Vec4T<float> {
  using type = float4;
};
template <>
struct Vec4T<double> {
  using type = double4;
};

// state as the reference lays it out: center (N,3), quaternion (N,4), one frame
template <typename R>
struct PackedLoader {
  const R* __restrict__ center;
  const R* __restrict__ quat;
  const int* __restrict__ meta;
  __device__ __forceinline__ V3<R> centre(int j) const { return V3<R>{center[3 * j + 0], center[3 * j + 1], center[3 * j + 2]}; }
  __device__ __forceinline__ void load(int j, Nuc<R>& o, R* q4) const {
    o.c = {center[3 * j + 0], center[3 * j + 1], center[3 * j + 2]};
    q4[0] = quat[4 * j + 0];
    q4[1] = quat[4 * j + 1];
    q4[2] = quat[4 * j + 2];
    q4[3] = quat[4 * j + 3];
    quat_axes(q4[0], q4[1], q4[2], q4[3], o.a1, o.a2, o.a3);
    const int m = meta[j];
    o.seq = m & 3;
    o.is_end = (m >> 2) & 1;
    o.rna = (m >> 3) & 1;
    o.idx = j;
  }
};

// MD state: 16/32-byte aligned vec4 position (w unused) and vec4 quaternion
template <typename R>
struct Vec4Loader {
  using V4 = typename Vec4T<R>::type;
  const V4* __restrict__ pos;
  const V4* __restrict__ quat;
  const int* __restrict__ meta;
  __device__ __forceinline__ V3<R> centre(int j) const {
    const V4 c = pos[j];
    return V3<R>{c.x, c.y, c.z};
  }
  __device__ __forceinline__ void load(int j, Nuc<R>& o, R* q4) const {
    const V4 c = pos[j];
    const V4 q = quat[j];
    o.c = {c.x, c.y, c.z};
    q4[0] = q.x;
    q4[1] = q.y;
    q4[2] = q.z;
    q4[3] = q.w;
    quat_axes(q4[0], q4[1], q4[2], q4[3], o.a1, o.a2, o.a3);
    const int m = meta[j];
    o.seq = m & 3;
    o.is_end = (m >> 2) & 1;
    o.rna = (m >> 3) & 1;
    o.idx = j;
  }
};

template <typename R>
__device__ __forceinline__ V3<R> min_image(V3<R> d, const BoxT<R>& box) {
  if (box.on) {
    d.x -= box.l[0] * m_rint(d.x * box.il[0]);
    d.y -= box.l[1] * m_rint(d.y * box.il[1]);
    d.z -= box.l[2] * m_rint(d.z * box.il[2]);
  }
  return d;
}

template <int G, typename R>
__device__ __forceinline__ void group_reduce_v3(V3<R>& v) {
  v.x = group_sum<G>(v.x);
  v.y = group_sum<G>(v.y);
  v.z = group_sum<G>(v.z);
}

template <int G, typename R, bool GRAD>
__device__ __forceinline__ void group_reduce(R* __restrict__ e, SelfGrad<R>& sg) {
#pragma unroll
  for (int k = 0; k < T_COUNT; ++k) e[k] = group_sum<G>(e[k]);
  if constexpr (GRAD) {
    group_reduce_v3<G>(sg.dc);
    group_reduce_v3<G>(sg.g1);
    group_reduce_v3<G>(sg.g2);
    group_reduce_v3<G>(sg.g3);
  }
}

// Walk row i with the G lanes of a group.  On return every lane holds its PARTIAL sums;
// call group_reduce() to fold them.  Every pair is visited from both ends, so energies (and
// parameter partials, inside the sink) carry weight 1/2.
// A row is walked in stages, each dense over the G lanes of the group:
//   1. bonded slots (one lane each) - unless the caller evaluates the bonded pairs elsewhere (BONDED = false: the
//      energy kernel gathers the 64 bonded slots of its workgroup in one wavefront);
//   2. every unbonded entry: centre distance only (12 bytes, no quaternion algebra) against rnear2, beyond which no
//      site pair of the two nucleotides can be inside any cut-off; the near entries are compacted into an LDS list;
//   3. the near list: full neighbour state, the radial terms, and a second compaction of the entries whose angular
//      terms can act;
//   4. that list: the angular terms.
// With the reference's all-pairs list (63 entries per nucleotide in a 32-bp duplex, ~8 near, ~5 with angular
// support) a single fused loop ran ~200 instructions of radial terms for every entry and the ~1 000-instruction
// angular code in every iteration for a lane or two; the kernel is VALU-bound, so instructions are its time.
// items: this group's LDS, two lists of list_cap ints (near entries, angular entries of the segment being walked).
// PARK (the energy kernel's fp64 gradient instantiations): the sums of the radial stages do not ride through the angular
// stage in registers.  In front of stage 4 they are folded over the group and added to the group's row `park` in LDS
// (kParkWidth words, zeroed by the caller), the accumulators start again from zero, and every angular item adds its own
// contribution to that row with LDS atomics - all lanes of a group are lanes of ONE wavefront, so what a row receives
// comes from one instruction stream in program order (colliding lanes of an instruction are served in lane order): the
// sums are reproducible bit for bit like the folded ones.  The caller reads the row back after the walk.  What it buys:
// 12 gradient + 8 energy accumulators (40 registers in fp64) are not live across the ~3 000 instructions of an angular
// item with parameter partials, which is where that instantiation spilled.
constexpr int kParkWidth = T_COUNT + 12;
template <typename R, int MODEL, bool GRAD, class PG, int G, bool BONDED = true, bool SEGMENTED = false, bool PARK = false, class Loader, class PT>
__device__ __forceinline__ void gather_row(const PT& P, const Loader& ld, const BoxT<R>& box,
                                           const int* __restrict__ rows, int row_stride, int len, int i,
                                           const Nuc<R>& self, int lane, R* __restrict__ e, SelfGrad<R>& sg,
                                           PG& pg, int* __restrict__ items, R rnear2, int list_cap, R* __restrict__ park = nullptr) {
  static_assert(G <= 32, "group masks below are 32-bit");
  const int* __restrict__ row = rows + (size_t)i * row_stride;
  const int gshift = (threadIdx.x & 63) & ~(G - 1);
  constexpr unsigned int kGroupMask = (G == 32) ? 0xffffffffu : ((1u << G) - 1u);
  const unsigned int below = (1u << lane) - 1u;
  int* __restrict__ near_list = items;
  int* __restrict__ ang_list = items + list_cap;
  // 1. bonded slots
  if (BONDED && lane < ROW_BONDED_SLOTS && lane < len) {
    const int entry = row[lane];
    if (entry >= 0) {
      Nuc<R> other;
      R q4[4];
      ld.load(entry & ROW_INDEX_MASK, other, q4);
      bonded_pair<R, MODEL, GRAD, PG>(P, self, other, min_image(other.c - self.c, box), (lane & 1) == 1, R(0.5), e, sg, pg);
    }
  }
  // Stages 2 - 4 run per SEGMENT of list_cap row entries, so the two LDS lists have a fixed size whatever the row
  // length: the reference's all-pairs lists (mythos/simulators/jax_md/utils.py:49-67) put n - 3 entries in every row,
  // 997 for the 1 000-nt persistence-length system, and a list sized by the row would not fit the LDS.  Rows up to
  // list_cap entries (every Verlet list, and all-pairs lists of the DiffTRe-sized systems) are one segment.
  // (SEGMENTED = false: the row fits the lists; one pass, and the compiler sees no outer loop - with the loop the
  // fp64 energy-only instantiation ran 50 % longer on rows that never needed a second segment)
  for (int seg0 = ROW_BONDED_SLOTS; seg0 < len; seg0 = SEGMENTED ? seg0 + list_cap : len) {
    const int seg_end = SEGMENTED ? min(len, seg0 + list_cap) : len;
    // 2. centre-distance filter
    int n_near = 0;
    for (int s0 = seg0; s0 < seg_end; s0 += G) {
      const int s = s0 + lane;
      const int entry = (s < seg_end) ? row[s] : -1;
      bool near = false;
      if (entry >= 0) {
        const V3<R> d = min_image(ld.centre(entry & ROW_INDEX_MASK) - self.c, box);
        near = dot(d, d) < rnear2;
      }
      const unsigned int m = (unsigned int)(__ballot(near) >> gshift) & kGroupMask;
      if (near) near_list[n_near + __popc(m & below)] = entry;
      n_near += __popc(m);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 3. radial terms of the near entries
    int n_ang = 0;
    for (int k0 = 0; k0 < n_near; k0 += G) {
      const int k = k0 + lane;
      bool flag = false;
      int entry = -1;
      if (k < n_near) {
        entry = near_list[k];
        Nuc<R> other;
        R q4[4];
        ld.load(entry & ROW_INDEX_MASK, other, q4);
        flag = unbonded_radial<R, MODEL, GRAD, PG>(P, self, other, min_image(other.c - self.c, box), (entry & ROW_ROLE_Q) == 0, R(0.5),
                                                   e, sg, pg);
      }
      const unsigned int m = (unsigned int)(__ballot(flag) >> gshift) & kGroupMask;
      if (flag) ang_list[n_ang + __popc(m & below)] = entry;
      n_ang += __popc(m);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 4. angular terms
    if constexpr (PARK) {
      group_reduce<G, R, GRAD>(e, sg);
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < T_COUNT; ++t) park[t] += e[t];
        if constexpr (GRAD) {
          park[T_COUNT + 0] += sg.dc.x, park[T_COUNT + 1] += sg.dc.y, park[T_COUNT + 2] += sg.dc.z;
          park[T_COUNT + 3] += sg.g1.x, park[T_COUNT + 4] += sg.g1.y, park[T_COUNT + 5] += sg.g1.z;
          park[T_COUNT + 6] += sg.g2.x, park[T_COUNT + 7] += sg.g2.y, park[T_COUNT + 8] += sg.g2.z;
          park[T_COUNT + 9] += sg.g3.x, park[T_COUNT + 10] += sg.g3.y, park[T_COUNT + 11] += sg.g3.z;
        }
      }
#pragma unroll
      for (int t = 0; t < T_COUNT; ++t) e[t] = R(0);
      sg.dc = sg.g1 = sg.g2 = sg.g3 = V3<R>{R(0), R(0), R(0)};
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int k = lane; k < n_ang; k += G) {
        const int entry = ang_list[k];
        Nuc<R> other;
        R q4[4];
        ld.load(entry & ROW_INDEX_MASK, other, q4);
        R ei[T_COUNT];
#pragma unroll
        for (int t = 0; t < T_COUNT; ++t) ei[t] = R(0);
        SelfGrad<R> si;
        si.dc = si.g1 = si.g2 = si.g3 = V3<R>{R(0), R(0), R(0)};
        unbonded_angular<R, MODEL, GRAD, PG>(P, self, other, min_image(other.c - self.c, box), (entry & ROW_ROLE_Q) == 0, R(0.5), ei, si,
                                             pg);
        // (the angular terms write three of the energies: hydrogen bonding, cross-stacking, coaxial stacking)
        atomicAdd(&park[T_HB], ei[T_HB]), atomicAdd(&park[T_CRST], ei[T_CRST]), atomicAdd(&park[T_CXST], ei[T_CXST]);
        if constexpr (GRAD) {
          atomicAdd(&park[T_COUNT + 0], si.dc.x), atomicAdd(&park[T_COUNT + 1], si.dc.y), atomicAdd(&park[T_COUNT + 2], si.dc.z);
          atomicAdd(&park[T_COUNT + 3], si.g1.x), atomicAdd(&park[T_COUNT + 4], si.g1.y), atomicAdd(&park[T_COUNT + 5], si.g1.z);
          atomicAdd(&park[T_COUNT + 6], si.g2.x), atomicAdd(&park[T_COUNT + 7], si.g2.y), atomicAdd(&park[T_COUNT + 8], si.g2.z);
          atomicAdd(&park[T_COUNT + 9], si.g3.x), atomicAdd(&park[T_COUNT + 10], si.g3.y), atomicAdd(&park[T_COUNT + 11], si.g3.z);
        }
      }
    } else {
    for (int k = lane; k < n_ang; k += G) {
      const int entry = ang_list[k];
      Nuc<R> other;
      R q4[4];
      ld.load(entry & ROW_INDEX_MASK, other, q4);
      unbonded_angular<R, MODEL, GRAD, PG>(P, self, other, min_image(other.c - self.c, box), (entry & ROW_ROLE_Q) == 0, R(0.5), e, sg,
                                           pg);
    }
    }
    __builtin_amdgcn_wave_barrier();  // the lists are rewritten by the next segment
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// dU/dq (4) from the axis gradients; a_k(q) as in mythos/energy/utils.py:18-36
template <typename R>
__device__ __forceinline__ void axes_grad_to_quat_grad(const R* q, const SelfGrad<R>& sg, R* dq) {
  const R q0 = R(2) * q[0], q1 = R(2) * q[1], q2 = R(2) * q[2], q3 = R(2) * q[3];
  const V3<R>&g1 = sg.g1, &g2 = sg.g2, &g3 = sg.g3;
  dq[0] = (q0 * g1.x + q3 * g1.y - q2 * g1.z) + (-q3 * g2.x + q0 * g2.y + q1 * g2.z) + (q2 * g3.x - q1 * g3.y + q0 * g3.z);
  dq[1] = (q1 * g1.x + q2 * g1.y + q3 * g1.z) + (q2 * g2.x - q1 * g2.y + q0 * g2.z) + (q3 * g3.x - q0 * g3.y - q1 * g3.z);
  dq[2] = (-q2 * g1.x + q1 * g1.y - q0 * g1.z) + (q1 * g2.x + q2 * g2.y + q3 * g2.z) + (q0 * g3.x + q3 * g3.y - q2 * g3.z);
  dq[3] = (-q3 * g1.x + q0 * g1.y + q1 * g1.z) + (-q0 * g2.x - q3 * g2.y + q2 * g2.z) + (q1 * g3.x + q2 * g3.y + q3 * g3.z);
}

// lab-frame torque  -sum_k a_k x dU/da_k
template <typename R>
__device__ __forceinline__ V3<R> axes_grad_to_torque(const Nuc<R>& s, const SelfGrad<R>& sg) {
  V3<R> t = cross(s.a1, sg.g1) + cross(s.a2, sg.g2) + cross(s.a3, sg.g3);
  return -t;
}

}  // inline namespace MYTHOS_MATH_NS
}  // namespace mythos
