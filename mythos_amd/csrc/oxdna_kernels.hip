// Energy / gradient evaluation over frames: the kernel behind mythos_oxdna_energy().
//
// Replaces ComposedEnergyFunction.compute_terms + EnergyFunction.map
// (mythos/energy/base.py:312-319, 90-93) and jax.grad / jax.value_and_grad of them
// (mythos/optimization/objective.py:235): one launch covers (frame, nucleotide-group) and
// returns the 8 term energies per frame, dU/dcenter, dU/dquat and dU/dparam.
//
// Layout: grid = (ceil(N / PPB), frames); a 256-thread block owns PPB = 256/G nucleotides of
// one frame.  Energies: group -> LDS (fixed order) -> per-block partial in HBM -> a second
// tiny kernel sums the partials in a fixed order, so e_terms are run-to-run reproducible.
// Roofline: HBM; algorithmic bytes per frame = (7 s + 4) N + 4 (nbar + 2) N  read,
// 7 s N + 8*8 + 8 K written (s = sizeof(real)).
#include <cstdlib>

#include "observables.h"
#include "oxdna_gather.h"

namespace mythos {

constexpr int kBlock = 256;
// entries of a neighbour row walked at a time: 32 nucleotides x 2 lists x 192 ints = 48 KB of LDS per workgroup
constexpr int kEnergyListCap = 192;

// Parameter-partial sink: fp64 LDS atomics into one of kPgCopies private copies of the accumulator, so the 64 lanes
// of a wave instruction that add to the SAME parameter - the common case: the index is a compile-time constant at
// most call sites - do not all hit one address.  A copy belongs to ONE wavefront (two copies per wavefront, by lane
// parity): everything added to it comes from that wavefront's instruction stream, in program order, and lanes that
// collide inside one instruction are served in lane order - so the sum a copy holds does not depend on how the four
// wavefronts of the workgroup interleave, and dU/dtheta is reproducible bit for bit like energies and forces (copies
// chosen by lane alone were shared by the wavefronts and were not).
// Eight copies, not sixteen: the LDS atomics are a few per cent of the kernel's time either way, but 35 KB of copies
// left room for two workgroups per CU where the registers allow three (dU/dtheta call -30 % in fp32).
constexpr int kPgCopies = 8;
#ifdef MYTHOS_EN_OLD_PGCOPY  // (dev A/B: copies chosen by lane, shared by the wavefronts)
__device__ __forceinline__ int pg_copy_of(unsigned int tid) { return (int)(tid % 8u); }
#else
__device__ __forceinline__ int pg_copy_of(unsigned int tid) { return (int)((tid >> 6) * 2u + (tid & 1u)); }
#endif
static_assert(kPgCopies == 2 * (256 / 64), "two accumulator copies per wavefront of the 256-thread workgroup");
template <int MODEL>
constexpr int pg_stride() {  // one past the count (242 doubles for oxDNA): the copies of one parameter land in different bank pairs
  return oxp_used<MODEL>() + 1;
}
struct LdsPG {
  static constexpr bool on = true;
  double* acc;  // this lane's copy
  template <typename R>
  __device__ __forceinline__ void add(int idx, R v) const {
    atomicAdd(&acc[idx], 0.5 * double(v));
  }
};

// workgroups per CU the register allocator makes room for: the gradient modes need ~170 (fp32) registers to stay
// out of scratch, the energy-only mode fits in 128
#ifndef EN_LB
#define EN_LB 3
#endif
template <typename R, int MODE, int MODEL = 2>
constexpr int energy_blocks_per_cu() {
  // oxNA with parameter partials: eight accumulator copies of three vectors are 50 KB of LDS - two workgroups per CU
  if (MODEL == 4 && MODE >= 2) return 2;
  // fp64: the energy-only mode runs faster at three workgroups per CU with 100 B of scratch than at two without
  // (0.58 -> 0.49 ms on the DiffTRe shape); the gradient modes spill too much for that (0.72 -> 1.43 ms)
  // and do not want ONE either (no spills, but one wavefront per SIMD: 0.72 -> 1.16 ms forces, 1.18 -> 1.44 ms dU/dtheta)
  // fp32 forces mode: 128 registers without scratch, so four fit (at three the allocator takes 138 and the call is 18 % slower)
#ifndef MYTHOS_EN_F64_GRAD_BLOCKS  // (dev A/B)
#define MYTHOS_EN_F64_GRAD_BLOCKS 2
#endif
  return sizeof(R) == 4 ? (MODE <= 1 ? 4 : EN_LB) : (MODE == 0 ? 3 : MYTHOS_EN_F64_GRAD_BLOCKS);
}

// SEG: rows longer than the LDS lists are walked in segments (gather_row)
// MODE 0 energy, 1 + gradients, 2 + parameter partials, 3 + dU/d(sequence distribution) (mythos_oxdna_energy_dpseq: its
// own instantiation - as run-time branches of MODE 2 the atomics cost the ordinary dU/dtheta call 1.3 - 1.5 % in fp64)
template <typename R, int MODEL, int MODE, int G, bool SEG>
__global__ __launch_bounds__(kBlock, (energy_blocks_per_cu<R, MODE, MODEL>())) void oxdna_energy_kernel(
    const R* __restrict__ Pg, const BoxT<R> box, int n, const R* __restrict__ center, const R* __restrict__ quat,
    const int* __restrict__ meta, const int* __restrict__ rows, const int* __restrict__ row_len, int row_stride,
    double* __restrict__ e_part, R* __restrict__ dU_dcenter, R* __restrict__ dU_dquat,
    double* __restrict__ pg_part, R rnear2, const PseqView<R> pseq, int list_cap) {
  constexpr int PPB = kBlock / G;
  constexpr bool GRAD = MODE >= 1;
  __shared__ double e_lds[PPB][T_COUNT];
  extern __shared__ int item_lds[];  // [PPB][2][list_cap]: per group, the near and the angular entries of a row segment (gather_row)
  constexpr int kPgStride = pg_stride<MODEL>(), kPgUsed = oxp_used<MODEL>();
  __shared__ double pg_lds[MODE >= 2 ? kPgCopies * kPgStride : 1];
  // parameters through the constant address space: scalar loads at the point of use (langevin_core.inc has the
  // measurements: by value in the kernel-argument segment they were spilled to scratch, from LDS they cost VGPRs)
  // (+ the probabilistic sequence, if one is set: a uniform branch at the two sequence-weight lookups)
  // (oxNA, MODEL 4: three vectors one after the other - oxDNA2, oxRNA2, hybrid; a probabilistic sequence reaches the
  //  weight look-ups through each of them - the reference threads it through its na1 hydrogen-bonding term,
  //  na1/hydrogen_bonding.py:243-304)
  const auto P = [&] {
    if constexpr (MODEL == 4) {
      // (MODE 3: dU/d(distribution) through the hydrogen-bonding weight of whichever vector a pair takes,
      //  na1/hydrogen_bonding.py:127-128, 243-304: all three views add into the same two buffers of this frame)
      using CP = ConstParams<R, true, MODE == 3>;
      PseqView<R> ps = pseq;
      if constexpr (MODE == 3) ps.gmarg += (size_t)blockIdx.y * n * 4, ps.gbp += (size_t)blockIdx.y * ps.bp_rows * 4;
      return Na1Params<CP>{CP(Pg, ps), CP(Pg + OXP_COUNT, ps), CP(Pg + 2 * OXP_COUNT, ps)};
    } else {
#ifdef MYTHOS_EN_NO_PSEQ  // (dev A/B)
      return ConstParams<R, false>(Pg);
#else
      PseqView<R> ps = pseq;
      if constexpr (MODE == 3) {  // the gradient buffers of this frame
        ps.gmarg += (size_t)blockIdx.y * n * 4, ps.gbp += (size_t)blockIdx.y * ps.bp_rows * 4;
      }
      return ConstParams<R, true, MODE == 3>(Pg, ps);
#endif
    }
  }();

  const int frame = blockIdx.y;
  const int grp = threadIdx.x / G;
  const int lane = threadIdx.x % G;
  const int i = blockIdx.x * PPB + grp;
  const size_t fo = (size_t)frame * n;

  if constexpr (MODE >= 2) {
    for (int k = threadIdx.x; k < kPgCopies * kPgStride; k += kBlock) pg_lds[k] = 0.0;
  }
  __syncthreads();

  R e[T_COUNT];
#pragma unroll
  for (int k = 0; k < T_COUNT; ++k) e[k] = R(0);
  SelfGrad<R> sg;
  sg.dc = sg.g1 = sg.g2 = sg.g3 = V3<R>{R(0), R(0), R(0)};
  R qs[4] = {R(1), R(0), R(0), R(0)};
  const PackedLoader<R> ld{center + fo * 3, quat + fo * 4, meta};

  // ---- bonded pairs: the 16 x 4 bonded slots of the workgroup are the 64 lanes of ONE wavefront (which one rotates
  //      with the workgroup), results handed to the owners through LDS.  Left inside the row walk, every wavefront ran
  //      the ~600 bonded instructions for the 4 of 16 lanes of its groups that hold a bonded slot: a quarter of the
  //      kernel's instructions, and the kernel is VALU-bound.
  constexpr int kBondedWaves = (PPB * ROW_BONDED_SLOTS + 63) / 64;
  static_assert(kBondedWaves <= kBlock / 64, "the bonded slots of a workgroup fit its wavefronts");
  constexpr int kBondedWidth = T_COUNT + 12;
  __shared__ R bonded_lds[PPB][ROW_BONDED_SLOTS][kBondedWidth];
  const int bonded_wave = ((int)(threadIdx.x >> 6) - (int)(blockIdx.x & 3)) & 3;  // 0 .. kBondedWaves-1: a bonded wavefront
  if (bonded_wave < kBondedWaves) {
    const int wl = bonded_wave * 64 + (threadIdx.x & 63), p = wl / ROW_BONDED_SLOTS, slot = wl % ROW_BONDED_SLOTS;
    const bool slot_ok = wl < PPB * ROW_BONDED_SLOTS;
    const int ip = blockIdx.x * PPB + p;
    R eb[T_COUNT];
#pragma unroll
    for (int k = 0; k < T_COUNT; ++k) eb[k] = R(0);
    SelfGrad<R> sb;
    sb.dc = sb.g1 = sb.g2 = sb.g3 = V3<R>{R(0), R(0), R(0)};
    if (slot_ok && ip < n && slot < row_len[ip]) {
      const int entry = rows[(size_t)ip * row_stride + slot];
      if (entry >= 0) {
        Nuc<R> sp, other;
        R qp[4], q4[4];
        ld.load(ip, sp, qp);
        ld.load(entry & ROW_INDEX_MASK, other, q4);
        const V3<R> dco = min_image(other.c - sp.c, box);
        if constexpr (MODE >= 2) {
          LdsPG pg{pg_lds + pg_copy_of(threadIdx.x) * kPgStride};
          bonded_pair<R, MODEL, GRAD, LdsPG>(P, sp, other, dco, (slot & 1) == 1, R(0.5), eb, sb, pg);
        } else {
          NoPG pg;
          bonded_pair<R, MODEL, GRAD, NoPG>(P, sp, other, dco, (slot & 1) == 1, R(0.5), eb, sb, pg);
        }
      }
    }
    R* br = bonded_lds[slot_ok ? p : 0][slot_ok ? slot : 0];
    if (slot_ok) {
#pragma unroll
    for (int k = 0; k < T_COUNT; ++k) br[k] = eb[k];
    br[T_COUNT + 0] = sb.dc.x, br[T_COUNT + 1] = sb.dc.y, br[T_COUNT + 2] = sb.dc.z;
    br[T_COUNT + 3] = sb.g1.x, br[T_COUNT + 4] = sb.g1.y, br[T_COUNT + 5] = sb.g1.z;
    br[T_COUNT + 6] = sb.g2.x, br[T_COUNT + 7] = sb.g2.y, br[T_COUNT + 8] = sb.g2.z;
    br[T_COUNT + 9] = sb.g3.x, br[T_COUNT + 10] = sb.g3.y, br[T_COUNT + 11] = sb.g3.z;
    }
  }

  if (i < n) {
    Nuc<R> self;
    ld.load(i, self, qs);
    if constexpr (MODE >= 2) {
      LdsPG pg{pg_lds + pg_copy_of(threadIdx.x) * kPgStride};
      gather_row<R, MODEL, GRAD, LdsPG, G, false, SEG>(P, ld, box, rows, row_stride, row_len[i], i, self, lane, e, sg, pg, item_lds + (size_t)grp * 2 * list_cap, rnear2, list_cap);
    } else {
      NoPG pg;
      gather_row<R, MODEL, GRAD, NoPG, G, false, SEG>(P, ld, box, rows, row_stride, row_len[i], i, self, lane, e, sg, pg, item_lds + (size_t)grp * 2 * list_cap, rnear2, list_cap);
    }
  }
  __syncthreads();  // the bonded results are in LDS
  if (i < n && lane < ROW_BONDED_SLOTS) {
    const R* br = bonded_lds[grp][lane];
#pragma unroll
    for (int k = 0; k < T_COUNT; ++k) e[k] += br[k];
    if constexpr (GRAD) {
      sg.dc = sg.dc + V3<R>{br[T_COUNT + 0], br[T_COUNT + 1], br[T_COUNT + 2]};
      sg.g1 = sg.g1 + V3<R>{br[T_COUNT + 3], br[T_COUNT + 4], br[T_COUNT + 5]};
      sg.g2 = sg.g2 + V3<R>{br[T_COUNT + 6], br[T_COUNT + 7], br[T_COUNT + 8]};
      sg.g3 = sg.g3 + V3<R>{br[T_COUNT + 9], br[T_COUNT + 10], br[T_COUNT + 11]};
    }
  }
  group_reduce<G, R, GRAD>(e, sg);

  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < T_COUNT; ++k) e_lds[grp][k] = (i < n) ? double(e[k]) : 0.0;
    if constexpr (GRAD) {
      if (i < n) {
        if (dU_dcenter) {
          R* o = dU_dcenter + (fo + i) * 3;
          o[0] = sg.dc.x;
          o[1] = sg.dc.y;
          o[2] = sg.dc.z;
        }
        if (dU_dquat) {
          R dq[4];
          axes_grad_to_quat_grad(qs, sg, dq);
          R* o = dU_dquat + (fo + i) * 4;
          o[0] = dq[0];
          o[1] = dq[1];
          o[2] = dq[2];
          o[3] = dq[3];
        }
      }
    }
  }
  __syncthreads();
  const size_t bo = (size_t)frame * gridDim.x + blockIdx.x;
  if (threadIdx.x < T_COUNT) {
    double s = 0.0;
    for (int g = 0; g < PPB; ++g) s += e_lds[g][threadIdx.x];
    e_part[bo * T_COUNT + threadIdx.x] = s;
  }
  if constexpr (MODE >= 2) {
    for (int k = threadIdx.x; k < kPgUsed; k += kBlock) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < kPgCopies; ++c) s += pg_lds[c * kPgStride + k];  // fixed order
      pg_part[bo * kPgUsed + k] = s;
    }
  }
}

// out[frame][k] = sum_b part[frame][b][k] for k < width; 0 for width <= k < out_width.  Grid (frames, ceil(out_width / 16)),
// 256 threads = 16 columns x 16 groups of workgroup partials: every thread adds the partials b = group, group + 16, ...
// of its column, then the 16 group sums are added in a fixed order.  (One thread per column walking all n_blocks
// partials - the first version - is a chain of n_blocks dependent loads: 110 us for the 750 workgroups of a 24 000-nt
// frame, four times the energy kernel itself; harmless only in the DiffTRe shape of many frames and two workgroups.)
constexpr int kReduceCols = 16, kReduceGroups = 16;
__global__ __launch_bounds__(kReduceCols * kReduceGroups) void reduce_partials_kernel(const double* __restrict__ part, int n_blocks,
                                                                                  int width, double* __restrict__ out, int out_width) {
  __shared__ double acc[kReduceGroups][kReduceCols + 1];
  const int frame = blockIdx.x;
  const int kk = threadIdx.x % kReduceCols, g = threadIdx.x / kReduceCols;
  const int k = blockIdx.y * kReduceCols + kk;
  double s = 0.0;
  if (k < width) {
    const double* p = part + (size_t)frame * n_blocks * width + k;
    for (int b = g; b < n_blocks; b += kReduceGroups) s += p[(size_t)b * width];
  }
  acc[g][kk] = s;
  __syncthreads();
  if (g == 0 && k < out_width) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < kReduceGroups; ++j) t += acc[j][kk];
    out[(size_t)frame * out_width + k] = t;
  }
}

// The same sums for few workgroups per frame (the DiffTRe shape: thousands of frames, two workgroups each): one thread
// per column, the partials added in index order - there the grouped kernel above would launch sixteen workgroups per
// frame to add two numbers each (dU/dtheta call +2 % fp64, +7 % fp32).
__global__ void reduce_partials_few_kernel(const double* __restrict__ part, int n_blocks, int width, double* __restrict__ out,
                                           int out_width) {
  const int frame = blockIdx.x;
  for (int k = threadIdx.x; k < out_width; k += blockDim.x) {
    double s = 0.0;
    if (k < width) {
      const double* p = part + (size_t)frame * n_blocks * width + k;
      for (int b = 0; b < n_blocks; ++b) s += p[(size_t)b * width];
    }
    out[(size_t)frame * out_width + k] = s;
  }
}

static void reduce_partials_launch(const double* part, int n_frames, int n_blocks, int width, double* out, int out_width, hipStream_t stream) {
  if (n_blocks <= kReduceGroups)
    hipLaunchKernelGGL(reduce_partials_few_kernel, dim3(n_frames), dim3(out_width <= 64 ? 64 : 256), 0, stream, part, n_blocks, width, out,
                       out_width);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_frames, (out_width + kReduceCols - 1) / kReduceCols), dim3(kReduceCols * kReduceGroups),
                       0, stream, part, n_blocks, width, out, out_width);
}

template <typename T>
static int ensure(T*& ptr, size_t& cap, size_t need) {
  if (need <= cap) return 0;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  MYTHOS_HIP_TRY(hipMalloc((void**)&ptr, need * sizeof(T)));
  cap = need;
  return 0;
}

template <typename R, int MODEL, int G>
static int launch_typed(mythos_system* sys, const R* center, const R* quat, int n_frames, double* e_terms,
                        R* dU_dcenter, R* dU_dquat, double* dU_dparams, mythos_obs* oset, double* obs_out, hipStream_t stream) {
  constexpr int PPB = kBlock / G;
  const int n = sys->n;
  const int blocks = (n + PPB - 1) / PPB;
  const bool dpseq = dU_dparams && sys->pseq_terms != 0 && sys->ps_gmarg != nullptr;  // mythos_oxdna_energy_dpseq
  const int mode = dU_dparams ? (dpseq ? 3 : 2) : ((dU_dcenter || dU_dquat) ? 1 : 0);
  // frames per chunk bounded by scratch (<= 256 MB of parameter partials) and the 65535 grid.y limit
  const int n_out = sys->param_sets() * (int)OXP_COUNT;  // width of a dU/dparams row
  size_t per_frame = (size_t)blocks * (mode >= 2 ? n_out : T_COUNT) * sizeof(double);
  int chunk = (int)std::min<size_t>(65535, std::max<size_t>(1, (size_t(256) << 20) / per_frame));
  chunk = std::min(chunk, n_frames);
  if (int rc = ensure(sys->d_epart, sys->epart_cap, (size_t)chunk * blocks * T_COUNT)) return rc;
  if (mode >= 2)
    if (int rc = ensure(sys->d_pgpart, sys->pgpart_cap, (size_t)chunk * blocks * n_out)) return rc;
  const R* P = device_params_of<R>(sys);
  const BoxT<R> box = make_box<R>(sys);
  // centre distance beyond which no site pair of two nucleotides is inside any cut-off: the longest range of a term
  // plus twice the farthest site from the centre (with a margin for the rounding of the comparison)
  // (oxNA: the largest over its three vectors; the oxRNA2 stacking sites are bonded-only and do not enter)
  double range = 0.0, reach = 0.0;
  for (int k = 0; k < sys->param_sets(); ++k) {
    const double* Pd = sys->param_sets() == 1 ? sys->pd.v : sys->pd_sets.data() + (size_t)k * OXP_COUNT;
    range = std::max({range, Pd[NEXC_BACKBONE_RC], Pd[NEXC_BASE_RC], Pd[NEXC_BACK_BASE_RC], Pd[NEXC_BASE_BACK_RC], Pd[HYDR_RCHIGH],
                      Pd[CRST_RCHIGH], Pd[CXST_RCHIGH]});
    if (MODEL >= 2) range = std::max(range, Pd[DH_RCUT]);
    reach = std::max({reach, std::hypot(Pd[GEO_BACK_A1], MODEL >= 2 ? Pd[GEO_BACK_A2] : 0.0), std::fabs(Pd[GEO_BASE]),
                      std::fabs(Pd[GEO_STACK])});
  }
  const double rnear = (range + 2.0 * reach) * (1.0 + 1e-4) + 1e-4;
  const R rnear2 = R(rnear * rnear);
  PseqView<R> pseq;
  if (sys->pseq_terms != 0) {
    pseq.marg = (const R*)sys->d_ps_marg, pseq.unit = sys->d_ps_unit, pseq.bp = (const R*)sys->d_ps_bp, pseq.terms = sys->pseq_terms;
  }
  // LDS lists of the row walk: a row is walked in segments of list_cap entries (gather_row)
  int list_cap = std::min(sys->row_stride, kEnergyListCap);
  if (const long long v = debug_value(MYTHOS_DEBUG_ENERGY_LIST_CAP)) {  // test hook: short segments on small systems
    if (v >= 8 && v <= kEnergyListCap) list_cap = std::min(list_cap, (int)v);
  }
  ObsView obs;  // width 0: no observables asked for
  if (oset && obs_out) {
    if (int rc = obs_view_for(oset, n_frames, &obs)) return rc;
  }
  for (int f0 = 0; f0 < n_frames; f0 += chunk) {
    const int nf = std::min(chunk, n_frames - f0);
    dim3 grid(blocks, nf);
    if (obs.width > 0) obs.axis = oset->d_axis + (size_t)f0 * obs.n_q * 3;
    const R* c = center + (size_t)f0 * n * 3;
    const R* q = quat + (size_t)f0 * n * 4;
    R* gc = dU_dcenter ? dU_dcenter + (size_t)f0 * n * 3 : nullptr;
    R* gq = dU_dquat ? dU_dquat + (size_t)f0 * n * 4 : nullptr;
    if (mode == 3) {
      pseq.bp_rows = std::max(sys->ps_n_bp, 1);
      pseq.gmarg = sys->ps_gmarg + (size_t)f0 * n * 4;
      pseq.gbp = sys->ps_gbp + (size_t)f0 * pseq.bp_rows * 4;
      MYTHOS_HIP_TRY(hipMemsetAsync(pseq.gmarg, 0, (size_t)nf * n * 4 * sizeof(double), stream));
      MYTHOS_HIP_TRY(hipMemsetAsync(pseq.gbp, 0, (size_t)nf * pseq.bp_rows * 4 * sizeof(double), stream));
    }
    auto launch = [&](auto mode_tag, auto seg_tag) {
      hipLaunchKernelGGL((oxdna_energy_kernel<R, MODEL, decltype(mode_tag)::value, G, decltype(seg_tag)::value>),
                         grid, dim3(kBlock), (size_t)PPB * 2 * list_cap * sizeof(int), stream, P, box, n, c, q, sys->d_meta, sys->d_rows,
                         sys->d_row_len, sys->row_stride, sys->d_epart, gc, gq, sys->d_pgpart, rnear2, pseq, list_cap);
    };
    auto by_seg = [&](auto mode_tag) {
      if (sys->row_stride > list_cap) launch(mode_tag, std::true_type{}); else launch(mode_tag, std::false_type{});
    };
    if (mode == 0) by_seg(std::integral_constant<int, 0>{});
    else if (mode == 1) by_seg(std::integral_constant<int, 1>{});
    else if (mode == 3) by_seg(std::integral_constant<int, 3>{});
    else by_seg(std::integral_constant<int, 2>{});
    // The observables of the same frames: the stand-alone kernel queued right behind the energy launch (the frames it
    // reads were just read: L2).  Through round 3 they rode in an epilogue of the energy kernel (OBS instantiations, the
    // first workgroup of every frame); measured against this form on the DiffTRe shape the epilogue was 3 - 10 % SLOWER
    // per call (its fp64 site algebra cost every workgroup of the launch registers and occupancy).
    if (obs.width > 0) {
      MYTHOS_HIP_TRY(hipGetLastError());
      if (int rc = observables_launch(oset, obs, c, q, nf, obs_out + (size_t)f0 * obs.width, stream)) return rc;
    }
    MYTHOS_HIP_TRY(hipGetLastError());
    reduce_partials_launch(sys->d_epart, nf, blocks, (int)T_COUNT, e_terms + (size_t)f0 * T_COUNT, (int)T_COUNT, stream);
    if (mode >= 2) reduce_partials_launch(sys->d_pgpart, nf, blocks, oxp_used<MODEL>(), dU_dparams + (size_t)f0 * n_out, n_out, stream);
    MYTHOS_HIP_TRY(hipGetLastError());
  }
  return 0;
}

int oxdna_energy_launch(mythos_system* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                        void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_obs* oset, double* obs_out, hipStream_t stream) {
  // 8 lanes per nucleotide = 32 nucleotides per workgroup (half the wavefronts of 16 lanes for the same rows, and the
  // short angular lists fill 8 lanes better than 16).  Rows of any length: the walk is segmented (gather_row), so the
  // reference's all-pairs lists of a 1 000-nt system (997 entries per row) go through the same kernel.
  if (sys->dtype == MYTHOS_F32) {
    if (sys->model == 1)
      return launch_typed<float, 1, 8>(sys, (const float*)center, (const float*)quat, n_frames, e_terms,
                                       (float*)dU_dcenter, (float*)dU_dquat, dU_dparams, oset, obs_out, stream);
    if (sys->model == 3)
      return launch_typed<float, 3, 8>(sys, (const float*)center, (const float*)quat, n_frames, e_terms,
                                       (float*)dU_dcenter, (float*)dU_dquat, dU_dparams, oset, obs_out, stream);
    if (sys->model == 4)
      return launch_typed<float, 4, 8>(sys, (const float*)center, (const float*)quat, n_frames, e_terms,
                                       (float*)dU_dcenter, (float*)dU_dquat, dU_dparams, oset, obs_out, stream);
    return launch_typed<float, 2, 8>(sys, (const float*)center, (const float*)quat, n_frames, e_terms,
                                     (float*)dU_dcenter, (float*)dU_dquat, dU_dparams, oset, obs_out, stream);
  }
  if (sys->model == 1)
    return launch_typed<double, 1, 8>(sys, (const double*)center, (const double*)quat, n_frames, e_terms,
                                      (double*)dU_dcenter, (double*)dU_dquat, dU_dparams, oset, obs_out, stream);
  if (sys->model == 4)
    return launch_typed<double, 4, 8>(sys, (const double*)center, (const double*)quat, n_frames, e_terms,
                                      (double*)dU_dcenter, (double*)dU_dquat, dU_dparams, oset, obs_out, stream);
  if (sys->model == 3)
    return launch_typed<double, 3, 8>(sys, (const double*)center, (const double*)quat, n_frames, e_terms,
                                      (double*)dU_dcenter, (double*)dU_dquat, dU_dparams, oset, obs_out, stream);
  return launch_typed<double, 2, 8>(sys, (const double*)center, (const double*)quat, n_frames, e_terms,
                                    (double*)dU_dcenter, (double*)dU_dquat, dU_dparams, oset, obs_out, stream);
}

}  // namespace mythos
