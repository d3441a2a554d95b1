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
// Per nucleotide per step (fp32): read + write the expanded frame (centre hi + lo, a1, a3, backbone
// offset, quaternion) and the momenta, read the neighbour row and the neighbours' frames through L2.
// Algorithmic HBM bytes are stated in DESIGN.md; the working set of a 12 kbp duplex (a few MB) is
// L2 / Infinity-Cache resident, the kernel is bound by VALU issue and latency, not by bytes.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_ext.h>

// The fp32 stepping kernels evaluate the piecewise modulation functions (f1, f2, f4, f5) in their branch-free forms
// and read the sequence weights with one indexed load (oxdna_math.h: MYTHOS_LEAN_MATH).  Measured on MI355X (round 3,
// A/B of one build against the other on one box): 12 kbp 61.4 k -> 63.9 k steps/s, 100 kbp 10.7 k -> 11.3 k, 256
// replicas of 64 nt 62.7 k -> 65.4 k; the base-pair item of the angular pass 949 VALU + 434 SALU -> 744 + 151.  fp64
// keeps the branchy forms: with the branch-free ones the three-per-CU instantiation spilled more (29.4 k -> 25.7 k).
#ifndef MYTHOS_LEAN_MATH
#define MYTHOS_LEAN_MATH 1
#endif

#include "chunk_order.h"
#include "oxdna_gather.h"
#include "philox.h"

namespace mythos {

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
    // native v_sin / v_cos: |phi| is a few 1e-2, the precise sincosf (argument reduction, private
    // out-pointers) costs an order of magnitude more instructions for digits fp32 MD cannot use
    s = __sinf(R(0.5) * phi);
    c = __cosf(R(0.5) * phi);
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
__device__ __forceinline__ void drift(R* x, R* q, const R* p, R* L, R h, const LangevinConst<R>& K, bool rotate = true) {
  x[0] += h * p[0] * K.inv_mass;
  x[1] += h * p[1] * K.inv_mass;
  x[2] += h * p[2] * K.inv_mass;
  if (!rotate) return;
  free_rotor<2>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<1>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<0>(q, L, h, K.inv_inertia);
  free_rotor<1>(q, L, R(0.5) * h, K.inv_inertia);
  free_rotor<2>(q, L, R(0.5) * h, K.inv_inertia);
}

constexpr int kMdBlock = 256;
constexpr int kMdG = 8;                    // lanes per nucleotide
constexpr int kMdPPB = kMdBlock / kMdG;    // nucleotides per workgroup
#ifndef MYTHOS_MD_ITEMS  // (dev A/B: scripts/build_variant.sh)
#define MYTHOS_MD_ITEMS 16
#endif
#ifndef MYTHOS_MD_F64_BLOCKS  // workgroups per CU the register allocator is asked to make room for, by variant
#define MYTHOS_MD_F64_BLOCKS 3
#endif
#ifndef MYTHOS_MD_F64S_BLOCKS
#define MYTHOS_MD_F64S_BLOCKS 2
#endif
#ifndef MYTHOS_MD_F32_BLOCKS
#define MYTHOS_MD_F32_BLOCKS 3
#endif
#ifndef MYTHOS_MD_F32S_BLOCKS
#define MYTHOS_MD_F32S_BLOCKS 2
#endif
constexpr int kMdItems = MYTHOS_MD_ITEMS;  // flagged unbonded neighbours per nucleotide (phase 2) the stepping kernel has room for
// ... and the variant a run falls back to when a nucleotide has more (see md_step_kernel): 32, or what the 160 KB of LDS
// leave for the fp64 energy-trace instantiation, whose result rows carry the 8 term energies as well
template <typename R, bool SAVE>
constexpr int md_items_big() {
  return (sizeof(R) == 8 && SAVE) ? 22 : 32;
}
// Result rows (one per evaluated bonded slot / angular item) come out of ONE pool per workgroup, handed out by a prefix
// sum over the 32 nucleotides' counts: a duplex uses 2 + ~5 rows per nucleotide, a fixed 4 + 16 per nucleotide was two
// thirds empty and its 33 KB (fp32) / 66 KB (fp64) of LDS decided how many workgroups a CU holds.  A workgroup whose
// nucleotides need more rows than the pool has aborts the launch like one whose work lists are too short (ITEMS), and
// the run goes on with the big instantiation, whose pool is the full 32 x (4 + ITEMS).
#ifndef MYTHOS_MD_POOL  // (dev A/B)
#define MYTHOS_MD_POOL 320
#endif
constexpr int kMdPool = MYTHOS_MD_POOL;
constexpr int kTraceWidth = T_COUNT + 2;   // 8 energy terms + KE_trans + KE_rot

// Expanded per-nucleotide state of one time level ("frame"), written by the kernel that
// produced the positions so that neighbour visits never redo the quaternion -> axes algebra:
//   p0 = (centre, meta)   p1 = (a1, 0)   p2 = (a3, 0)   p3 = (backbone offset k1 a1 + k2 a2, 0)
//   pl = (centre_lo, 0)   fp32 only: the centre is the unevaluated sum p0.xyz + pl.xyz (|lo| <= ulp(hi)/2), which
//        keeps ~48 bits of position however large the coordinates are (a 12 kbp duplex is 4 800 length units long,
//        where a bare fp32 coordinate resolves 5e-4).  Differences of nearby centres are then exact to fp32
//        round-off of the DIFFERENCE: (hi_j - hi_i) is exact (Sterbenz), (lo_j - lo_i) is tiny.
//   q  = quaternion
//   mom = (p, 0), ang = (L_body, 0): the momenta of the same time level.  They ping-pong with the positions, so a
//        launch never modifies the state it read: whatever it discovers on the way (a work list that does not fit),
//        the host can discard what it wrote and run that step again from intact inputs.
template <typename R>
struct Frame {
  typename Vec4T<R>::type *p0, *p1, *p2, *p3, *q, *pl, *mom, *ang;
};

template <typename R>
constexpr bool kHiLo = sizeof(R) == 4;

// centre(o) - centre(s) from the hi (and, in fp32, lo) parts
template <typename R>
__device__ __forceinline__ V3<R> centre_diff(const typename Vec4T<R>::type& o_hi, const typename Vec4T<R>::type& o_lo,
                                             const V3<R>& s_hi, const V3<R>& s_lo) {
  V3<R> d{o_hi.x - s_hi.x, o_hi.y - s_hi.y, o_hi.z - s_hi.z};
  if constexpr (kHiLo<R>) {
    d.x += o_lo.x - s_lo.x;
    d.y += o_lo.y - s_lo.y;
    d.z += o_lo.z - s_lo.z;
  }
  return d;
}

// squared cut-offs of the radial pass, derived on the host from the parameter vector
template <typename R>
struct MdCut {
  R rbb2;    // backbone-backbone: max(Debye r_cut, excluded-volume r_c)^2
  R rcom2;   // centre-centre distance below which the base / stack site terms can act
  // squared supports of the angular terms' radial factors (base-base for H-bond and cross-stacking,
  // stack-stack for coaxial stacking): the radial pass flags a neighbour without taking a square root
  R hb_lo2, hb_hi2, cr_lo2, cr_hi2, cx_lo2, cx_hi2;
  // bit (4 * seq_p + seq_q) set where the H-bond weight table is non-zero (only complementary pairs by default):
  // the radial pass tests one bit instead of walking the 16-entry table
  unsigned int hb_mask;
};

template <typename R>
__device__ __forceinline__ V3<R> xyz(const typename Vec4T<R>::type& v) {
  return V3<R>{v.x, v.y, v.z};
}

#ifdef MYTHOS_MD_EXP_RADIAL_STUB  // (dev experiment, WRONG physics: the radial functions at the price of two multiplications -
// the step's cost if tabulating f3 / Debye-Hueckel made them free; bounds what such tables can save)
#define MD_RAD_F3(r, eps, fp) FD<R>{(r) * R(1e-4), (r) * R(-1e-4)}
#define MD_RAD_DH(r, dhp) FD<R>{(r) * R(1e-5), (r) * R(-1e-5)}
#define MD_RAD_SQRT(r2) (r2)
#define MD_RAD_OVER(a, r) ((a))
#else
#define MD_RAD_F3(r, eps, fp) f3_eval(r, eps, fp)
#define MD_RAD_DH(r, dhp) debye_eval(r, dhp)
#define MD_RAD_SQRT(r2) m_sqrt(r2)
#define MD_RAD_OVER(a, r) ((a) / (r))
#endif

// radial f3 from r^2: returns the energy and, in coef, tw * V'(r) / r (0 outside the support)
template <typename R>
__device__ __forceinline__ R f3_coef(R eps, R tw, const F3P<R>& fp, R r2, R& coef) {
  coef = R(0);
#ifndef MYTHOS_MD_EXP_RADIAL_STUB
  if (r2 >= fp.rc * fp.rc) return R(0);
#endif
  const R r = MD_RAD_SQRT(r2);
  const FD<R> v = MD_RAD_F3(r, eps, fp);
  coef = MD_RAD_OVER(tw * v.d, r);
  return v.f;
}

template <typename R>
__device__ __forceinline__ R f3_radial(R eps, R tw, const F3P<R>& fp, V3<R> d, R r2, V3<R>& g) {
#ifndef MYTHOS_MD_EXP_RADIAL_STUB
  if (r2 >= fp.rc * fp.rc) return R(0);
#endif
  const R r = MD_RAD_SQRT(r2);
  const FD<R> v = MD_RAD_F3(r, eps, fp);
  axpy(g, MD_RAD_OVER(tw * v.d, r), d);
  return v.f;
}

// Diagnostics are compiled in only with -DMYTHOS_MD_DIAG (make DIAG=1): in the product build no stamp executes and
// the ablation word is a compile-time zero.
#ifdef MYTHOS_MD_DIAG
// Diagnostic stamps (ablate bit 7): lane 0 of every wavefront records s_memtime (bit 8: the 100 MHz
// s_memrealtime instead) at the phase boundaries
// into the (otherwise unused) energy scratch; no output value depends on them.
#define MD_STAMP(k)                                                                                   \
  do {                                                                                                \
    if ((ablate & 128) && (threadIdx.x & 63) == 0)                                                     \
      reinterpret_cast<unsigned long long*>(e_part)[(((step & 1) * n_blocks + (size_t)bid) * 4 + (threadIdx.x >> 6)) * 8 + (k)] = \
          (ablate & 256) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();           \
  } while (0)
#define MD_ABLATE(x) (x)
#else
#define MD_STAMP(k) do { } while (0)
#define MD_ABLATE(x) 0
#endif

// Optimisation barrier on a register value: whatever produced it stays before this point, its uses after.
template <typename T>
__device__ __forceinline__ void md_pin(T& v) {
  asm volatile("" : "+v"(v));
}

// One MD step (see file header).  kick_close: multiple of dt*F that closes the previous step
// (0 for the first kernel of a run, 1/2 otherwise); do_step = 0 for the closing-only kernel.
//
// Work decomposition: 8 lanes per nucleotide, 32 nucleotides per 256-thread workgroup.
//   phase 1 (radial): the lanes stride over the nucleotide's unbonded row, close segment then far segment;
//           per neighbour they read the centre (hi, lo), the backbone offset and - in the close segment -
//           a1, evaluate Debye-Hueckel and the excluded-volume site pairs with early-outs on squared
//           distances, and flag the few neighbours whose base-base / stack-stack distance lies in the
//           support of an angular term (two LDS lists per nucleotide: base-pair terms, coaxial stacking);
//   phase 2 (angular): work items of the whole workgroup, one code path per wavefront: the bonded
//           neighbours (FENE, bonded excluded volume, stacking), the two halves of the base-pair list
//           (H-bond + cross-stacking evaluated together), the coaxial list; results go to LDS rows;
//   fold:   each group sums its rows (DPP reductions over the 8 lanes);
//   integrate: one wavefront advances the 32 nucleotides of the workgroup and writes the next frame.
// workgroups per CU the register allocator is asked to make room for: what the LDS footprint of the
// variant allows (fp32 stepping 43 KB; the trace and fp64 variants carry wider result rows)
// DENSE (fp64 stepping only): the grid is larger than two workgroups per CU - ask for three (168 VGPRs + 116 B of scratch
// instead of 224 without: all 750 workgroups of 12 kbp resident at once, +11.6 %); a grid that fits anyway keeps the
// spill-free allocation (1 kbp: 49.5 k steps/s against 44.7 k with the tighter bound)
template <typename R, bool SAVE, int ITEMS, bool DENSE = false>
constexpr int md_blocks_per_cu() {
  if (ITEMS > kMdItems) return sizeof(R) == 4 ? (SAVE ? 1 : 2) : 1;  // a pool of 32 x 36 rows: 64 - 100 KB (fp32), 125 - 150 KB (fp64) of LDS
  return sizeof(R) == 4 ? (SAVE ? MYTHOS_MD_F32S_BLOCKS : MYTHOS_MD_F32_BLOCKS) : (SAVE ? MYTHOS_MD_F64S_BLOCKS : (DENSE ? MYTHOS_MD_F64_BLOCKS : 2));
}

// What the radial pass reads of the parameters, gathered so that the oxNA instantiation (MODEL 4) can hold one set per
// kind of pair - DNA-DNA, RNA-RNA, hybrid - and choose per row entry; every other instantiation has ONE set, built from
// the values it always used (scalar registers; the compiler sees the same operands as before).
template <typename R>
struct RadSet {
  F3P<R> f_bb, f_base, f_bkba, f_babk;
  R eps_n, tw_n, tw_dh;
  DebyeP<R> dhp;
  bool half_ends;
  R rbb2, hb_lo2, hb_hi2, cr_lo2, cr_hi2, cx_lo2, cx_hi2;
  unsigned int hb_mask;
};
template <typename R, int MODEL, class PT>
__device__ __forceinline__ RadSet<R> radset_from(const PT& P, const MdCut<R>& cut) {
  RadSet<R> s;
  s.f_bb = f3_params<R>(P, NEXC_BACKBONE_RSTAR), s.f_base = f3_params<R>(P, NEXC_BASE_RSTAR);
  s.f_bkba = f3_params<R>(P, NEXC_BACK_BASE_RSTAR), s.f_babk = f3_params<R>(P, NEXC_BASE_BACK_RSTAR);
  s.eps_n = P[NEXC_EPS];
  s.tw_n = P[TW_NEXC], s.tw_dh = (MODEL >= 2) ? P[TW_DH] : R(0);
  s.half_ends = (MODEL >= 2) && (P[DH_HALF_CHARGED_ENDS] != R(0));
  s.dhp = (MODEL >= 2) ? debye_params<R>(P) : DebyeP<R>{};
  s.rbb2 = cut.rbb2, s.hb_lo2 = cut.hb_lo2, s.hb_hi2 = cut.hb_hi2, s.cr_lo2 = cut.cr_lo2, s.cr_hi2 = cut.cr_hi2;
  s.cx_lo2 = cut.cx_lo2, s.cx_hi2 = cut.cx_hi2, s.hb_mask = cut.hb_mask;
  return s;
}
// oxNA: the supports of one parameter vector, derived on the device (scalar arithmetic, once per workgroup) the way
// make_cut derives them on the host for the single-vector models
template <typename R, class PT>
__device__ __forceinline__ MdCut<R> cut_from(const PT& P, R rcom2) {
  MdCut<R> c;
  const R rbb = fmax(P[NEXC_BACKBONE_RC], P[DH_RCUT]);
  c.rbb2 = rbb * rbb, c.rcom2 = rcom2;
  c.hb_lo2 = P[HYDR_RCLOW] * P[HYDR_RCLOW], c.hb_hi2 = P[HYDR_RCHIGH] * P[HYDR_RCHIGH];
  c.cr_lo2 = P[CRST_RCLOW] * P[CRST_RCLOW], c.cr_hi2 = P[CRST_RCHIGH] * P[CRST_RCHIGH];
  c.cx_lo2 = P[CXST_RCLOW] * P[CXST_RCLOW], c.cx_hi2 = P[CXST_RCHIGH] * P[CXST_RCHIGH];
  c.hb_mask = 0u;
#pragma unroll
  for (int k = 0; k < 16; ++k) c.hb_mask |= (P[HYDR_EPS_00 + k] != R(0)) ? (1u << k) : 0u;
  return c;
}
template <typename R>
__device__ __forceinline__ F3P<R> pick3(const F3P<R>& a, const F3P<R>& b, const F3P<R>& c, int k) {
  return {k == 0 ? a.rstar : (k == 1 ? b.rstar : c.rstar), k == 0 ? a.sigma : (k == 1 ? b.sigma : c.sigma),
          k == 0 ? a.b : (k == 1 ? b.b : c.b), k == 0 ? a.rc : (k == 1 ? b.rc : c.rc), a.base};
}
#define MD_PICK3(f) (k == 0 ? a.f : (k == 1 ? b.f : c.f))
template <typename R>
__device__ __forceinline__ RadSet<R> pick3(const RadSet<R>& a, const RadSet<R>& b, const RadSet<R>& c, int k) {
  RadSet<R> s;
  s.f_bb = pick3(a.f_bb, b.f_bb, c.f_bb, k), s.f_base = pick3(a.f_base, b.f_base, c.f_base, k);
  s.f_bkba = pick3(a.f_bkba, b.f_bkba, c.f_bkba, k), s.f_babk = pick3(a.f_babk, b.f_babk, c.f_babk, k);
  s.eps_n = MD_PICK3(eps_n), s.tw_n = MD_PICK3(tw_n), s.tw_dh = MD_PICK3(tw_dh);
  s.dhp = {MD_PICK3(dhp.rcut), MD_PICK3(dhp.rhigh), MD_PICK3(dhp.kappa), MD_PICK3(dhp.prefactor), MD_PICK3(dhp.bsmooth)};
  s.half_ends = a.half_ends;  // one switch for the whole system (na1/debye.py:25)
  s.rbb2 = MD_PICK3(rbb2), s.hb_lo2 = MD_PICK3(hb_lo2), s.hb_hi2 = MD_PICK3(hb_hi2), s.cr_lo2 = MD_PICK3(cr_lo2);
  s.cr_hi2 = MD_PICK3(cr_hi2), s.cx_lo2 = MD_PICK3(cx_lo2), s.cx_hi2 = MD_PICK3(cx_hi2), s.hb_mask = MD_PICK3(hb_mask);
  return s;
}
#undef MD_PICK3

// the parameter set of the row entry being evaluated: the one set of the model, or (oxNA) the set of the pair's kind -
// 0 DNA-DNA, 1 RNA-RNA, 2 hybrid - from the type bits of the two meta words
#define MD_RADSET_OF_ENTRY                                                                                         \
  const int md_kind = (MODEL == 4) ? na1_kind(self.rna, ((int)o0.w >> 3) & 1) : 0;                                 \
  const RadSet<R> rs_picked = (MODEL == 4) ? pick3(rs0, rs1, rs2, md_kind) : rs0;                                  \
  const RadSet<R>& rs = (MODEL == 4) ? rs_picked : rs0;
__device__ __forceinline__ int na1_kind(int self_rna, int other_rna) { return (self_rna && other_rna) ? 1 : ((self_rna || other_rna) ? 2 : 0); }

// ITEMS: result rows per nucleotide for the angular work lists.  16 is enough for any duplex, junction or origami
// at physical density (a base has 3 - 5 partners inside the range of an angular term); a nucleotide with more makes
// the launch ABORT: it raises flags[3], the host discards what that launch wrote (its inputs are intact: frames and
// momenta ping-pong) and runs the step again with the ITEMS = 32 instantiation, which stays in use for the rest of
// the run.  More than 32 is reported as an error (sterically that takes overlapping bases).
// PSEQ: the system carries a probabilistic sequence (mythos_oxdna_set_pseq): the two sequence-weight look-ups of the
// angular pass are expectations (ConstParams<R, true>, as in the energy kernel) and the radial pass flags every pair
// inside the hydrogen-bonding range, whatever the discrete sequence says.  Its own instantiations (with the wide work
// lists only): the plain ones keep their registers and instruction counts.
template <typename R, int MODEL, bool SAVE, int ITEMS, bool DENSE = false, bool PSEQ = false>
__global__ __launch_bounds__(kMdBlock, (md_blocks_per_cu<R, SAVE, ITEMS, DENSE>())) void md_step_kernel(
    const R* __restrict__ Pg, const BoxT<R> box, const LangevinConst<R> K, const MdCut<R> cut, int n, const Frame<R> in,
    const Frame<R> out,
    const int* __restrict__ rows, const int* __restrict__ row_len, const int* __restrict__ row_close, int row_stride,
    int extra_bonds, R kick_close, int do_step, uint64_t seed, uint64_t step, const typename Vec4T<R>::type* __restrict__ ref_pos,
    const typename Vec4T<R>::type* __restrict__ ref_off, const typename Vec4T<R>::type* __restrict__ ref_a1,
    int* __restrict__ flags,
    R* __restrict__ traj_c, R* __restrict__ traj_q, double* __restrict__ e_part, const int* __restrict__ chunk_order,
    const int* __restrict__ list_overflow, int k_index, int ablate_arg, const PseqView<R> pseq) {
  using V4 = typename Vec4T<R>::type;
  const int ablate = MD_ABLATE(ablate_arg);
  constexpr int G = kMdG, PPB = kMdPPB;
  constexpr int RW = (SAVE ? 12 + T_COUNT : 12) + 1;  // result row: dc, g1, g2, g3 (+ energies), padded to odd
  constexpr int kSlots = ROW_BONDED_SLOTS + ITEMS;
  // two work lists per nucleotide: 0 = base-pair terms (H-bond and / or cross-stacking: they share the base-base
  // vector and all six angles, so one evaluation serves both), 1 = coaxial stacking
  __shared__ int items[2][PPB][ITEMS];  // the flagged row ENTRIES (index | role bit), not their slots
  __shared__ int item_cnt[2][PPB];
  __shared__ int item_pre[4][PPB + 1];  // per WAVEFRONT: the prefix of the list that wavefront will walk
  __shared__ R self_lds[PPB][13];
  __shared__ R rad_lds[PPB][7];  // radial-pass site gradients (backbone, base) of each nucleotide
  // result rows, [nucleotide][slot][RW] with the nucleotide stride padded to an odd word count: the 32
  // nucleotides' rows then start in 32 different banks (20 x 13 = 260 words would alias p and p + 8)
  // fp64: rows out of the workgroup's pool (see kMdPool).  fp32 keeps a fixed block of kSlots rows per nucleotide: its
  // LDS never decided the residency (2.9 workgroups per CU at 12 kbp), and the pool's second prefix scan and base-row
  // look-ups cost it 1.2 % (61.3 k against 62.1 k steps/s, A/B on one box).
  constexpr bool kPooled = sizeof(R) == 8;
  constexpr int kPool = (ITEMS > kMdItems || !kPooled) ? PPB * kSlots : kMdPool;
  static_assert(kPool >= PPB * ROW_BONDED_SLOTS, "the pool holds at least the bonded rows");
  __shared__ R res_flat[kPool * RW + (kPooled ? 0 : PPB)];
  __shared__ int row_base[kPooled ? 4 : 1][PPB + 1];  // per WAVEFRONT (like item_pre): first pool row of every nucleotide
  // fixed layout: the nucleotide stride padded to an odd word count, so the 32 nucleotides' blocks start in 32 banks
  constexpr int kFixedStride = (kSlots * RW) | 1;
  auto pool_row = [&](int row) -> R* { return res_flat + row * RW; };
  auto fixed_row = [&](int pp, int slot) -> R* { return res_flat + pp * kFixedStride + slot * RW; };
  __shared__ double e_lds[SAVE ? PPB : 1][kTraceWidth];
  using CP = ConstParams<R, PSEQ>;
  const auto make_cp = [&](const R* g) {
    if constexpr (PSEQ) return CP(g, pseq); else return CP(g);
  };
  const CP P = make_cp(Pg);  // scalar loads at the point of use; an LDS copy was measured 2.4x slower
  const int grp = threadIdx.x / G;
  const int lane = threadIdx.x % G;
  // XCD-aware order: the hardware deals consecutive workgroups round-robin to the 8 XCDs, so workgroup b
  // takes chunk (b % 8) * ceil(n_blocks / 8) + b / 8 - every XCD then owns one contiguous eighth of the
  // nucleotide index range and neighbouring chunks (same strand, adjacent cells) share its L2.
  const int n_blocks = (n + PPB - 1) / PPB;
  const int vb = (int)(blockIdx.x & 7) * ((n_blocks + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (vb >= n_blocks) return;  // grid is padded to a multiple of 8; whole workgroup leaves together
  // Halted (flags[1], set by the previous step when a site left its skin; or a rebuild overflowed its rows or spill
  // list): this and every later launch of the segment do nothing, the state stays at the last valid step, and the
  // host rebuilds and resumes from flags[2] (kernel index after the last one that ran).
  // One lane requests the words here; everybody looks at them behind the first barrier (LDS), before which the
  // kernel writes nothing to global memory.  (Every thread loading and testing them up front cost 1.7 % of the step.)
  __shared__ int s_halt;
  int halt_word = 0;  // requested now, parked in LDS just before the barrier: nobody waits for it on the way
  // The halt word carries the index of the first launch that must not run (set by launch k: k + 1): a workgroup of
  // the SAME launch that starts after the word was set keeps going - on a grid larger than what is resident at once
  // the late workgroups of launch k would otherwise skip a step the early ones took.
  if (threadIdx.x == 0) {
    const int hw = flags[1], aw = flags[3];  // aw: an earlier launch aborted (work lists too short, see ITEMS)
    halt_word = ((hw != 0 && hw <= k_index) ? 1 : 0) | ((aw != 0 && aw <= k_index) ? 1 : 0) |
                (list_overflow ? (list_overflow[0] | list_overflow[1]) : 0);
  }
  // chunk_order (host, from the positions at the start of a run): the chunks of 32 nucleotides in spatial order, so
  // the contiguous eighth an XCD works on is also contiguous in space - in a duplex the two complementary
  // stretches of the strands, which are far apart in index, land on the same XCD and share its L2
  const int bid = chunk_order ? chunk_order[vb] : vb;
  const int i = bid * PPB + grp;
  const bool valid = i < n;
  const int ii = valid ? i : n - 1;  // out-of-range groups shadow the last nucleotide and discard

  const R g_ba = P[GEO_BASE], g_st = P[GEO_STACK];
  // oxNA: the oxRNA2 vector (sites of an RNA nucleotide) and the hybrid one; P itself is the oxDNA2 vector there
  const CP Prna = make_cp(Pg + ((MODEL == 4) ? OXP_COUNT : 0)), Pdrh = make_cp(Pg + ((MODEL == 4) ? 2 * OXP_COUNT : 0));
  const Na1Params<CP> P4{P, Prna, Pdrh};

  // ---- owner state (also parked in LDS for the block-wide angular pass)
  Nuc<R> self;
  V3<R> offb_s, self_lo{R(0), R(0), R(0)};
  {
    const V4 s0 = in.p0[ii], s1 = in.p1[ii], s2 = in.p2[ii], s3 = in.p3[ii];
    if constexpr (kHiLo<R>) self_lo = xyz<R>(in.pl[ii]);
    self.c = xyz<R>(s0);
    self.a1 = xyz<R>(s1);
    self.a3 = xyz<R>(s2);
    self.a2 = cross(self.a3, self.a1);
    offb_s = xyz<R>(s3);
    const int m = (int)s0.w;
    self.seq = m & 3;
    self.is_end = (m >> 2) & 1;
    self.rna = (m >> 3) & 1;
    if (lane == 0) {
      R* sl = self_lds[grp];
      sl[0] = s0.x, sl[1] = s0.y, sl[2] = s0.z, sl[3] = s1.x, sl[4] = s1.y, sl[5] = s1.z;
      sl[6] = s2.x, sl[7] = s2.y, sl[8] = s2.z, sl[9] = s0.w;
      sl[10] = self_lo.x, sl[11] = self_lo.y, sl[12] = self_lo.z;
    }
  }
  const int* __restrict__ row = rows + (size_t)ii * row_stride;
  const int len = (valid && !(ablate & 1)) ? row_len[ii] : 0;
  const int close_end = min(len, row_close[ii]);  // [2, close_end): any term may act; [close_end, len): backbone only

  R e[T_COUNT];
#pragma unroll
  for (int k = 0; k < T_COUNT; ++k) e[k] = R(0);
  V3<R> gbk{R(0), R(0), R(0)}, gba{R(0), R(0), R(0)};  // sum of dV/dd acting on self's backbone / base site

  MD_STAMP(0);
  // ---- phase 1: radial pass over the unbonded slots
  const RadSet<R> rs0 = (MODEL == 4) ? radset_from<R, 2>(P, cut_from<R>(P, cut.rcom2)) : radset_from<R, MODEL>(P, cut);
  // (oxNA: the oxRNA2 and the hybrid vector follow the oxDNA2 one; the other models never read rs1 / rs2)
  const RadSet<R> rs1 = (MODEL == 4) ? radset_from<R, 2>(Prna, cut_from<R>(Prna, cut.rcom2)) : rs0;
  const RadSet<R> rs2 = (MODEL == 4) ? radset_from<R, 2>(Pdrh, cut_from<R>(Pdrh, cut.rcom2)) : rs0;
  int n_items[2] = {0, 0};
  const int lane64 = threadIdx.x & 63;
  const int gshift = lane64 & ~(G - 1);
  // Software pipeline: the lane's row entries are fetched kEnt at a time, and the neighbour
  // state (centre, backbone offset) of entry k+1 is requested before entry k is evaluated, so the
  // L2 / Infinity-Cache round trips overlap the arithmetic instead of serialising with it.
  // (rolled: keeping the body once in the instruction stream matters more than unrolling - the whole
  // kernel has to stay inside the instruction cache that two CUs share)
  // (the prefetches are unconditional: an entry past the end of the segment is read from a clamped slot and replaced by
  // -1, a missing neighbour's state is read from nucleotide 0 and never used - a load behind a lane-dependent branch
  // made the compiler wait for ALL outstanding loads at the join, the one just issued included)
  const int last_slot = row_stride - 1;
  auto row_at = [&](int s, int end) -> int {
    const int v = row[min(s, last_slot)];
    return s < end ? v : -1;
  };
  {
    int e_cur = -1, e_nxt = -1;
    V4 n0{}, n3{}, n1{}, nl{};
    {
      const int s = ROW_BONDED_SLOTS + lane;
      e_cur = row_at(s, close_end);
      e_nxt = row_at(s + G, close_end);
      const int j = max(e_cur, 0) & ROW_INDEX_MASK;
      n0 = in.p0[j];
      n3 = in.p3[j];
      n1 = in.p1[j];
      if constexpr (kHiLo<R>) nl = in.pl[j];
    }
#pragma unroll 1
    for (int s0 = ROW_BONDED_SLOTS; s0 < close_end; s0 += G) {
      const int s = s0 + lane;
      const int entry = e_cur;
      const V4 o0 = n0, o3 = n3, o1 = n1, ol = nl;
      e_cur = e_nxt;
      e_nxt = row_at(s + 2 * G, close_end);
      {  // the close segment reads a1 as well: nearly all of its entries need it
        const int jn = max(e_cur, 0) & ROW_INDEX_MASK;
        n0 = in.p0[jn];
        n3 = in.p3[jn];
        n1 = in.p1[jn];
        if constexpr (kHiLo<R>) nl = in.pl[jn];
      }
      bool flag[2] = {false, false};
      if (entry >= 0) {
        const bool role_p = (entry & ROW_ROLE_Q) == 0;
        MD_RADSET_OF_ENTRY
        const bool o_rna = (MODEL == 4) && ((((int)o0.w) >> 3) & 1);
        const R gba_s = (MODEL == 4 && self.rna) ? Prna[GEO_BASE] : g_ba, gba_o = o_rna ? Prna[GEO_BASE] : g_ba;
        const R gst_s = (MODEL == 4 && self.rna) ? Prna[GEO_STACK] : g_st, gst_o = o_rna ? Prna[GEO_STACK] : g_st;
        (void)gst_s, (void)gst_o;
        const V3<R> dco = min_image(centre_diff<R>(o0, ol, self.c, self_lo), box);
        const V3<R> offb_o = xyz<R>(o3);
        const bool close = dot(dco, dco) < cut.rcom2;
        // backbone - backbone: excluded volume + Debye-Hueckel
        {
          const V3<R> d = dco + offb_o - offb_s;
          const R r2 = dot(d, d);
          if (r2 < rs.rbb2) {
            const R r = MD_RAD_SQRT(r2);
            const FD<R> v = MD_RAD_F3(r, rs.eps_n, rs.f_bb);
            R dVdr = rs.tw_n * v.d;
            R en = v.f;
            if constexpr (MODEL >= 2) {
              const FD<R> dh = MD_RAD_DH(r, rs.dhp);
              R mult = R(1);
              if (rs.half_ends) {
                const int mo = (int)o0.w;
                mult = (self.is_end ? R(0.5) : R(1)) * (((mo >> 2) & 1) ? R(0.5) : R(1));
              }
              dVdr += rs.tw_dh * mult * dh.d;
              if constexpr (SAVE) e[T_DH] += R(0.5) * mult * dh.f;
            }
            if constexpr (SAVE) e[T_NEXC] += R(0.5) * en;
            axpy(gbk, MD_RAD_OVER(dVdr, r), d);
          }
        }
        if (close) {
          const V3<R> a1o = xyz<R>(o1);
          R en = R(0);
          // self backbone - other base and self base - other backbone: which of the two is the reference's
          // "back_p - base_q" / "base_p - back_q" depends on the role; the squared distances are routed by
          // role so both parameter blocks stay scalar operands
          {
            V3<R> dA = dco - offb_s;
            axpy(dA, gba_o, a1o);
            V3<R> dB = dco + offb_o;
            axpy(dB, -gba_s, self.a1);
            const R ra2 = dot(dA, dA), rb2 = dot(dB, dB);
            R c1, c2;
            en += f3_coef(rs.eps_n, rs.tw_n, rs.f_bkba, role_p ? ra2 : rb2, c1);
            en += f3_coef(rs.eps_n, rs.tw_n, rs.f_babk, role_p ? rb2 : ra2, c2);
            axpy(gbk, role_p ? c1 : c2, dA);
            axpy(gba, role_p ? c2 : c1, dB);
          }
          const V3<R> da = a1o - self.a1;
          {
            V3<R> d = dco;
            if constexpr (MODEL == 4) {  // each nucleotide's base site at the offset of its own type
              axpy(d, gba_o, a1o);
              axpy(d, -gba_s, self.a1);
            } else {
              axpy(d, g_ba, da);
            }
            const R r2 = dot(d, d);
            en += f3_radial(rs.eps_n, rs.tw_n, rs.f_base, d, r2, gba);
            flag[0] = rs.cr_lo2 < r2 && r2 < rs.cr_hi2;
            if (!flag[0] && rs.hb_lo2 < r2 && r2 < rs.hb_hi2) {  // H-bond only for pairs with a non-zero weight
              const int so = (int)o0.w & 3;
              if (PSEQ && (pseq.terms & 2) != 0)
                flag[0] = rs.hb_mask != 0u;  // the weight is an expectation over both bases: any non-zero table entry may count
              else
                flag[0] = (rs.hb_mask >> (role_p ? (self.seq * 4 + so) : (so * 4 + self.seq))) & 1u;
            }
          }
          {
            V3<R> d = dco;
            if constexpr (MODEL == 4) {
              axpy(d, gst_o, a1o);
              axpy(d, -gst_s, self.a1);
            } else {
              axpy(d, g_st, da);
            }
            const R r2 = dot(d, d);
            flag[1] = rs.cx_lo2 < r2 && r2 < rs.cx_hi2;
          }
          if constexpr (SAVE) e[T_NEXC] += R(0.5) * en;
        }
      }
      // append the flagged slots of this group to its two LDS lists, in slot order
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned long long bal = __ballot(flag[t]);
        const unsigned int gm = (unsigned int)(bal >> gshift) & ((1u << G) - 1u);
        if (flag[t]) {
          const int pos = n_items[t] + __popc(gm & ((1u << lane) - 1u));
          if (pos < ITEMS) items[t][grp][pos] = entry;
        }
        n_items[t] += __popc(gm);
      }
    }
  }
  // far segment: only the backbone-backbone terms (excluded volume + Debye-Hueckel) can act
  {
    int e_cur = -1, e_nxt = -1;
    V4 n0{}, n3{}, nl{};
    {
      const int s = close_end + lane;
      e_cur = row_at(s, len);
      e_nxt = row_at(s + G, len);
      const int j = max(e_cur, 0) & ROW_INDEX_MASK;
      n0 = in.p0[j];
      n3 = in.p3[j];
      if constexpr (kHiLo<R>) nl = in.pl[j];
    }
#pragma unroll 1
    for (int s0 = close_end; s0 < len; s0 += G) {
      const int s = s0 + lane;
      const int entry = e_cur;
      const V4 o0 = n0, o3 = n3, ol = nl;
      e_cur = e_nxt;
      e_nxt = row_at(s + 2 * G, len);
      {
        const int jn = max(e_cur, 0) & ROW_INDEX_MASK;
        n0 = in.p0[jn];
        n3 = in.p3[jn];
        if constexpr (kHiLo<R>) nl = in.pl[jn];
      }
      if (entry >= 0) {
        MD_RADSET_OF_ENTRY
        const V3<R> dco = min_image(centre_diff<R>(o0, ol, self.c, self_lo), box);
        const V3<R> d = dco + xyz<R>(o3) - offb_s;
        const R r2 = dot(d, d);
        if (r2 < rs.rbb2) {
          const R r = MD_RAD_SQRT(r2);
          const FD<R> v = MD_RAD_F3(r, rs.eps_n, rs.f_bb);
          R dVdr = rs.tw_n * v.d;
          if constexpr (MODEL >= 2) {
            const FD<R> dh = MD_RAD_DH(r, rs.dhp);
            R mult = R(1);
            if (rs.half_ends) {
              const int mo = (int)o0.w;
              mult = (self.is_end ? R(0.5) : R(1)) * (((mo >> 2) & 1) ? R(0.5) : R(1));
            }
            dVdr += rs.tw_dh * mult * dh.d;
            if constexpr (SAVE) e[T_DH] += R(0.5) * mult * dh.f;
          }
          if constexpr (SAVE) e[T_NEXC] += R(0.5) * v.f;
          axpy(gbk, MD_RAD_OVER(dVdr, r), d);
        }
      }
    }
  }
  if (n_items[0] + n_items[1] > ITEMS) {  // result rows of one nucleotide exhausted: this launch does not count
    if (lane == 0) atomicMax(flags + 3, k_index + 1);
    n_items[0] = n_items[1] = 0;
  }
  // The radial sums are folded over the group now and parked in LDS: nothing computed so far stays in
  // registers across the angular pass (whose pair functions need the whole register budget).
  group_reduce_v3<G>(gbk);
  group_reduce_v3<G>(gba);
  if (lane == 0) {
    R* rl = rad_lds[grp];
    rl[0] = gbk.x, rl[1] = gbk.y, rl[2] = gbk.z, rl[3] = gba.x, rl[4] = gba.y, rl[5] = gba.z;
#pragma unroll
    for (int t = 0; t < 2; ++t) item_cnt[t][grp] = valid ? n_items[t] : 0;
  }
  if (threadIdx.x == 0) s_halt = halt_word;
  MD_STAMP(1);
  __syncthreads();  // self_lds, rad_lds and item_cnt are visible
  MD_STAMP(2);
  if (s_halt != 0) return;  // halted: nothing has been written to global memory yet
  if (vb == 0 && threadIdx.x == 0) flags[2] = k_index + 1;

  // ---- phase 2: angular pass, work items spread over the whole workgroup so that every wavefront
  //      runs ONE code path (roles below).  Results go to the owner's result rows in LDS.
  {
    NoPG pg;
    // role of this wavefront: 0 bonded, 1 and 2 the two halves of the base-pair list (~100 items per workgroup
    // in a duplex: one sweep of 64 each instead of two sweeps on one wavefront), 3 coaxial list; rotated with the
    // workgroup index so the heavy and the light roles spread over the four SIMDs of a CU
    const int wave = ((threadIdx.x >> 6) + bid) & 3;
    const bool bonded_wave = wave == 0;
    const int lst = wave == 3 ? 1 : 0;
    // exclusive prefix of the 32 per-nucleotide counts of this wavefront's list, so the list is dense over the
    // workgroup; every wavefront scans for itself (5 DPP-free shuffle steps) instead of meeting at a second barrier
    const int pw = threadIdx.x >> 6;
    // rows 2, 3 (second-bond slots) exist only in systems with circular strands
    const int n_bonded_rows = extra_bonds ? ROW_BONDED_SLOTS : 2;
    {
      const int l = threadIdx.x & 63;
      int inc = (l < PPB) ? item_cnt[lst][l] : 0;
      // ... and (pooled rows) of the rows every nucleotide takes from the result pool: its bonded slots, then its two lists
      int rows_inc = (kPooled && l < PPB) ? n_bonded_rows + item_cnt[0][l] + item_cnt[1][l] : 0;
#pragma unroll
      for (int o = 1; o < PPB; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (l >= o) inc += u;
        if constexpr (kPooled) {
          const int v = __shfl_up(rows_inc, o, 64);
          if (l >= o) rows_inc += v;
        }
      }
      if (l < PPB) item_pre[pw][l + 1] = inc;
      if (l == 0) item_pre[pw][0] = 0;
      if constexpr (kPooled) {
        if (l < PPB) row_base[pw][l + 1] = rows_inc;
        if (l == 0) row_base[pw][0] = 0;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // more rows than the pool holds (every wavefront computes the same number): nothing of the angular pass is
    // evaluated or folded, the launch is marked as not counting and the host goes on with the big instantiation
    const bool pool_over = kPooled && row_base[kPooled ? pw : 0][PPB] > kPool;
    if (pool_over && threadIdx.x == 0) atomicMax(flags + 3, k_index + 1);
    const int n_list = pool_over ? 0 : item_pre[pw][PPB];
    const int half = (n_list + 1) >> 1;
    const int q_lo = wave == 2 ? half : 0;                      // this wavefront's slice [q_lo, q_hi) of the list
    const int q_hi = wave == 1 ? half : n_list;
    const int n_total = q_hi - q_lo;
    const int n_sweeps = (n_total + 63) / 64;
    // bonded wave: one sweep over slots 0 / 1 of the 32 nucleotides, and a second over slots 2 / 3 only in
    // systems with circular strands (a ring's two ends carry a second bond in one role)
    const int my_sweeps = bonded_wave ? (((ablate & 16) || pool_over) ? 0 : (extra_bonds ? 2 : 1)) : ((ablate & 8) ? 0 : n_sweeps);
    for (int sweep = 0; sweep < ((ablate & 2) ? 0 : my_sweeps); ++sweep) {
      int p, idx, sl;
      bool active;
      if (bonded_wave) {
        p = (threadIdx.x & 63) >> 1;
        idx = (threadIdx.x & 1) + 2 * sweep;
        sl = idx;
        active = true;
      } else {
        const int q = q_lo + sweep * 64 + (threadIdx.x & 63);
        active = q < q_hi;
        int lo = 0, hi = PPB;  // owner: largest p with item_pre[pw][p] <= q
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (item_pre[pw][mid] <= q) lo = mid; else hi = mid;
        }
        p = lo;
        const int k = q - item_pre[pw][lo];
        sl = active ? items[lst][p][k] : -1;  // for these waves sl carries the row entry itself
        // result row: the bonded slots, then the nucleotide's H-bond, cross-stacking and coaxial items
        idx = n_bonded_rows + k + (lst >= 1 ? item_cnt[0][p] : 0);
      }
      const int ip = bid * PPB + p;
      if (!active || ip >= n) continue;
      const int entry = bonded_wave ? rows[(size_t)ip * row_stride + sl] : sl;
      R* out_r = kPooled ? pool_row(row_base[kPooled ? pw : 0][p] + idx) : fixed_row(p, idx);
      SelfGrad<R> g;
      g.dc = g.g1 = g.g2 = g.g3 = V3<R>{R(0), R(0), R(0)};
      R ee[T_COUNT];
#pragma unroll
      for (int k = 0; k < T_COUNT; ++k) ee[k] = R(0);
      if (entry >= 0) {
        const int j = entry & ROW_INDEX_MASK;
        const bool role_p = bonded_wave ? ((sl & 1) == 1) : ((entry & ROW_ROLE_Q) == 0);
        Nuc<R> me, o;
        const R* ms = self_lds[p];
        me.c = V3<R>{ms[0], ms[1], ms[2]};
        me.a1 = V3<R>{ms[3], ms[4], ms[5]};
        me.a3 = V3<R>{ms[6], ms[7], ms[8]};
        me.a2 = cross(me.a3, me.a1);
        const int mm = (int)ms[9];
        me.seq = mm & 3;
        me.is_end = (mm >> 2) & 1;
        me.rna = (mm >> 3) & 1;
        me.idx = ip, o.idx = j;  // (read only by the expectation of a probabilistic sequence)
        const V4 o0 = in.p0[j], o1 = in.p1[j], o2 = in.p2[j];
        V4 ol{};
        if constexpr (kHiLo<R>) ol = in.pl[j];
        o.c = xyz<R>(o0);
        o.a1 = xyz<R>(o1);
        o.a3 = xyz<R>(o2);
        o.a2 = cross(o.a3, o.a1);
        const int mo = (int)o0.w;
        o.seq = mo & 3;
        o.is_end = (mo >> 2) & 1;
        o.rna = (mo >> 3) & 1;
        const V3<R> dco = min_image(centre_diff<R>(o0, ol, me.c, V3<R>{ms[10], ms[11], ms[12]}), box);
        // (oxNA: the pair templates pick vector, form and sites by the kind of the pair from the three vectors)
        const auto& PP = [&]() -> const auto& {
          if constexpr (MODEL == 4) return P4; else return P;
        }();
        if (wave == 0) {
          bonded_pair<R, MODEL, true, NoPG>(PP, me, o, dco, role_p, R(0.5), ee, g, pg);
        } else if (wave != 3) {
#ifdef MYTHOS_MD_EXP_HALF_ITEMS  // (dev experiment, WRONG physics: the step's cost if ONE wavefront's sweep covered the base-pair list)
          if (wave == 1)
#endif
          unbonded_angular<R, MODEL, true, NoPG, 3>(PP, me, o, dco, role_p, R(0.5), ee, g, pg);
        } else {
          unbonded_angular<R, MODEL, true, NoPG, 4>(PP, me, o, dco, role_p, R(0.5), ee, g, pg);
        }
      }
      out_r[0] = g.dc.x, out_r[1] = g.dc.y, out_r[2] = g.dc.z;
      out_r[3] = g.g1.x, out_r[4] = g.g1.y, out_r[5] = g.g1.z;
      out_r[6] = g.g2.x, out_r[7] = g.g2.y, out_r[8] = g.g2.z;
      out_r[9] = g.g3.x, out_r[10] = g.g3.y, out_r[11] = g.g3.z;
      if constexpr (SAVE) {
#pragma unroll
        for (int k = 0; k < T_COUNT; ++k) out_r[12 + k] = ee[k];
      }
    }
  }
  // ---- integrator prologue, early: the wavefront with the coaxial role is the first to leave the angular pass
  //      (few items) and would idle at the barrier; it is also the one that integrates below, so it draws the
  //      thermostat noise and fetches momenta, quaternion and list-reference rows here, off the tail of the kernel
  //      where nothing else is left to hide their latency.  md_pin keeps the values on this side of the barriers.
  const int int_wave = (3 - bid) & 3;  // the wavefront whose role above was 3
  const int il = threadIdx.x & 63;     // nucleotide of this lane in the integrating wave
  const int i_int = bid * PPB + il;
  const bool integrates = (int)(threadIdx.x >> 6) == int_wave && il < PPB && i_int < n;
  R z[6] = {R(0), R(0), R(0), R(0), R(0), R(0)};
  V4 pm{}, lm{}, qv{}, r0{}, f0{}, a0{};
  if (integrates) {
    pm = in.mom[i_int], lm = in.ang[i_int], qv = in.q[i_int];
    if (do_step && K.skin_half_sq > R(0)) r0 = ref_pos[i_int], f0 = ref_off[i_int], a0 = ref_a1[i_int];
    if (do_step && !MD_ABLATE(ablate & (4 | 32))) normals6(seed, (uint32_t)i_int, step, 0u, z);
#pragma unroll
    for (int k = 0; k < 6; ++k) md_pin(z[k]);
    md_pin(pm.x), md_pin(pm.y), md_pin(pm.z);
    md_pin(lm.x), md_pin(lm.y), md_pin(lm.z);
    md_pin(qv.x), md_pin(qv.y), md_pin(qv.z), md_pin(qv.w);
  }
  MD_STAMP(3);
  __syncthreads();
  MD_STAMP(4);

  // ---- fold: each group gathers its owner's result rows (one per lane), adds the radial-pass
  //      site gradients, and reduces over its 8 lanes in a fixed order
  SelfGrad<R> sg;
  sg.dc = sg.g1 = sg.g2 = sg.g3 = V3<R>{R(0), R(0), R(0)};
  // (any wavefront's copy of row_base: they are identical, and complete since the barrier above)
  const int fw = kPooled ? (int)(threadIdx.x >> 6) : 0;
  const int rb = kPooled ? row_base[fw][grp] : 0;
  const bool pool_ok = !kPooled || row_base[fw][PPB] <= kPool;
  const int n_bonded_fold = extra_bonds ? ROW_BONDED_SLOTS : 2;  // rows 2, 3 exist only in systems with circular strands
  if (valid && pool_ok) {
    const int total = kPooled ? row_base[fw][grp + 1] - rb : n_bonded_fold + item_cnt[0][grp] + item_cnt[1][grp];
    for (int u = lane; u < total; u += G) {
      const R* rr = kPooled ? pool_row(rb + u) : fixed_row(grp, u);
      sg.dc = sg.dc + V3<R>{rr[0], rr[1], rr[2]};
      sg.g1 = sg.g1 + V3<R>{rr[3], rr[4], rr[5]};
      sg.g2 = sg.g2 + V3<R>{rr[6], rr[7], rr[8]};
      sg.g3 = sg.g3 + V3<R>{rr[9], rr[10], rr[11]};
      if constexpr (SAVE) {
#pragma unroll
        for (int k = 0; k < T_COUNT; ++k) e[k] += rr[12 + k];
      }
    }
  }
  if (lane == 0) {  // radial-pass sums (already folded over the group)
    const R* rl = rad_lds[grp];
    const V3<R> rbk{rl[0], rl[1], rl[2]}, rba{rl[3], rl[4], rl[5]};
    sg.dc = sg.dc - (rbk + rba);
    if constexpr (MODEL == 4) {  // the sites of this nucleotide's own type
      const bool r = self.rna != 0;
      axpy(sg.g1, -(r ? Prna[GEO_BACK_A1] : P[GEO_BACK_A1]), rbk);
      axpy(sg.g1, -(r ? Prna[GEO_BASE] : P[GEO_BASE]), rba);
      axpy(sg.g2, r ? R(0) : -P[GEO_BACK_A2], rbk);
      axpy(sg.g3, r ? -Prna[GEO_BACK_A2] : R(0), rbk);
    } else {
    axpy(sg.g1, -P[GEO_BACK_A1], rbk);
    axpy(sg.g1, -P[GEO_BASE], rba);
    if constexpr (MODEL == 2) axpy(sg.g2, -P[GEO_BACK_A2], rbk);
    if constexpr (MODEL == 3) axpy(sg.g3, -P[GEO_BACK_A2], rbk);  // oxRNA2: the backbone site's second axis is a3
    }
  }
  if constexpr (SAVE) {
    group_reduce<G, R, true>(e, sg);
  } else {
    group_reduce_v3<G>(sg.dc);
    group_reduce_v3<G>(sg.g1);
    group_reduce_v3<G>(sg.g2);
    group_reduce_v3<G>(sg.g3);
  }

  // the folded gradient of every nucleotide goes back to LDS (row 0 of its own result block, which only
  // this group has read) so that ONE wavefront integrates all 32 nucleotides of the workgroup, one per
  // lane: the integrator is ~0.6 k instructions per lane whatever the lane count, and run by lane 0 of
  // every group it occupied all four SIMDs at 1/8 lane use
  if (lane == 0) {
    R* fr = kPooled ? pool_row(pool_ok ? rb : grp * 2) : fixed_row(grp, 0);  // (pool exhausted: the launch does not count; any free row will do)
    fr[0] = sg.dc.x, fr[1] = sg.dc.y, fr[2] = sg.dc.z;
    fr[3] = sg.g1.x, fr[4] = sg.g1.y, fr[5] = sg.g1.z;
    fr[6] = sg.g2.x, fr[7] = sg.g2.y, fr[8] = sg.g2.z;
    fr[9] = sg.g3.x, fr[10] = sg.g3.y, fr[11] = sg.g3.z;
  }
  MD_STAMP(5);
  __syncthreads();
  MD_STAMP(6);
  double ke_t = 0.0, ke_r = 0.0;
  if (integrates) {
    const int i = i_int;
    Nuc<R> self;
    SelfGrad<R> sg;
    {
      const R* ms = self_lds[il];
      self.c = V3<R>{ms[0], ms[1], ms[2]};
      self.a1 = V3<R>{ms[3], ms[4], ms[5]};
      self.a3 = V3<R>{ms[6], ms[7], ms[8]};
      self.a2 = cross(self.a3, self.a1);
      const R* fr = kPooled ? pool_row(row_base[fw][PPB] <= kPool ? row_base[fw][il] : il * 2) : fixed_row(il, 0);
      sg.dc = V3<R>{fr[0], fr[1], fr[2]};
      sg.g1 = V3<R>{fr[3], fr[4], fr[5]};
      sg.g2 = V3<R>{fr[6], fr[7], fr[8]};
      sg.g3 = V3<R>{fr[9], fr[10], fr[11]};
    }
    const bool int_rna = (MODEL == 4) && ((((int)self_lds[il][9]) >> 3) & 1);  // oxNA: this nucleotide's own geometry
    const R g_k1 = int_rna ? Prna[GEO_BACK_A1] : P[GEO_BACK_A1];
    const R g_k2 = (MODEL >= 2) ? (int_rna ? Prna[GEO_BACK_A2] : P[GEO_BACK_A2]) : R(0);
    const V3<R> F = -sg.dc;
    const V3<R> tl = axes_grad_to_torque(self, sg);
    const R tb[3] = {dot(self.a1, tl), dot(self.a2, tl), dot(self.a3, tl)};
    R p[3] = {pm.x, pm.y, pm.z}, L[3] = {lm.x, lm.y, lm.z};
    R qs[4] = {qv.x, qv.y, qv.z, qv.w};
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
    R xl[3] = {self_lds[il][10], self_lds[il][11], self_lds[il][12]};  // low part of the centre (fp32 runs)
    R dxa[3] = {R(0), R(0), R(0)};                                      // this step's displacement
    R* const xd = kHiLo<R> ? dxa : x;
    V3<R> n1 = self.a1, n2 = self.a2, n3 = self.a3;
    V3<R> nbk = (MODEL == 3 || int_rna) ? n3 : n2;  // second axis of the backbone site (a2; a3 in oxRNA2)
    if (do_step && !(ablate & 4)) {
      p[0] += K.half_dt * F.x;
      p[1] += K.half_dt * F.y;
      p[2] += K.half_dt * F.z;
      L[0] += K.half_dt * tb[0];
      L[1] += K.half_dt * tb[1];
      L[2] += K.half_dt * tb[2];
      drift(xd, qs, p, L, K.half_dt, K, !(ablate & 64));
      p[0] = K.c1_t * p[0] + K.c2_t * z[0];
      p[1] = K.c1_t * p[1] + K.c2_t * z[1];
      p[2] = K.c1_t * p[2] + K.c2_t * z[2];
      L[0] = K.c1_r * L[0] + K.c2_r[0] * z[3];
      L[1] = K.c1_r * L[1] + K.c2_r[1] * z[4];
      L[2] = K.c1_r * L[2] + K.c2_r[2] * z[5];
      drift(xd, qs, p, L, K.half_dt, K, !(ablate & 64));
      if constexpr (kHiLo<R>) {
        // centre += displacement in (hi, lo) form: the displacement goes to the low part, then one fast two-sum
        // re-normalises (|hi| >= |lo + d| always holds here)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const R sdl = xl[k] + dxa[k];
          const R t = x[k] + sdl;
          xl[k] = sdl - (t - x[k]);
          x[k] = t;
        }
      }
      // keep the quaternion on the unit sphere (fp32 round-off)
      const R inv = m_rsqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
      qs[0] *= inv;
      qs[1] *= inv;
      qs[2] *= inv;
      qs[3] *= inv;
      if (!(x[0] == x[0]) || !(qs[0] == qs[0])) atomicOr(flags, 2);
      quat_axes(qs[0], qs[1], qs[2], qs[3], n1, n2, n3);
      nbk = (MODEL == 3 || int_rna) ? n3 : n2;
      if (K.skin_half_sq > R(0)) {
        // the list is valid while neither the centre nor the backbone and base sites (the segments are selected
        // by site distances, and a rotation moves the sites) have travelled more than skin / 2 since the build
        const R dx = x[0] - r0.x, dy = x[1] - r0.y, dz = x[2] - r0.z;
        const R bx = dx + (g_k1 * n1.x + g_k2 * nbk.x - f0.x), by = dy + (g_k1 * n1.y + g_k2 * nbk.y - f0.y),
                bz = dz + (g_k1 * n1.z + g_k2 * nbk.z - f0.z);
        // base site c + g_base a1 (the stacking site lies between it and the centre)
        const R gb = int_rna ? Prna[GEO_BASE] : P[GEO_BASE];
        const R sx = dx + gb * (n1.x - a0.x), sy = dy + gb * (n1.y - a0.y), sz = dz + gb * (n1.z - a0.z);
        if (dx * dx + dy * dy + dz * dz > K.skin_half_sq || bx * bx + by * by + bz * bz > K.skin_half_sq ||
            sx * sx + sy * sy + sz * sz > K.skin_half_sq)
          atomicMax(flags + 1, k_index + 1);  // the list is stale for the NEXT force evaluation: launch k + 1 halts
      }
    }
    out.p0[i] = V4{x[0], x[1], x[2], self_lds[il][9]};
    if constexpr (kHiLo<R>) out.pl[i] = V4{xl[0], xl[1], xl[2], R(0)};
    out.p1[i] = V4{n1.x, n1.y, n1.z, R(0)};
    out.p2[i] = V4{n3.x, n3.y, n3.z, R(0)};
    // (a closing-only launch hands the frame on unchanged, bit for bit: the offset is copied, not re-derived from
    // axes whose cross product may round differently - advance(a); advance(b) then equals advance(a + b) exactly)
    out.p3[i] = do_step ? V4{g_k1 * n1.x + g_k2 * nbk.x, g_k1 * n1.y + g_k2 * nbk.y, g_k1 * n1.z + g_k2 * nbk.z, R(0)} : in.p3[i];
    out.q[i] = V4{qs[0], qs[1], qs[2], qs[3]};
    out.mom[i] = V4{p[0], p[1], p[2], R(0)};
    out.ang[i] = V4{L[0], L[1], L[2], R(0)};
  }
  MD_STAMP(7);
  if constexpr (SAVE) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < T_COUNT; ++k) e_lds[grp][k] = valid ? double(e[k]) : 0.0;
    }
    if ((int)(threadIdx.x >> 6) == int_wave && il < PPB) {
      e_lds[il][T_COUNT] = ke_t;
      e_lds[il][T_COUNT + 1] = ke_r;
    }
    __syncthreads();
    if (threadIdx.x < kTraceWidth) {
      double s = 0.0;
      for (int g = 0; g < PPB; ++g) s += e_lds[g][threadIdx.x];
      e_part[(size_t)bid * kTraceWidth + threadIdx.x] = s;
    }
  }
}

// 256 threads = 16 columns (10 used) x 16 groups of workgroup partials, the group sums added in a fixed order (one thread
// per column was a chain of n_blocks dependent loads: 110 us per saved step at 12 kbp)
__global__ __launch_bounds__(256) void reduce_trace_kernel(const double* __restrict__ part, int n_blocks, double* __restrict__ out) {
  __shared__ double acc[16][17];
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;
  double s = 0.0;
  if (k < kTraceWidth)
    for (int b = g; b < n_blocks; b += 16) s += part[(size_t)b * kTraceWidth + k];
  acc[g][k] = s;
  __syncthreads();
  if (g == 0 && k < kTraceWidth && out) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += acc[j][k];
    out[k] = t;
  }
}

// ------------------------------------------------------------------ packed (N,3)/(N,4) <-> frame
// BX: the axis of the second backbone coefficient (2: a2, 3: a3), or 0 = by the nucleotide's type (oxNA: g_* for DNA on
// a1 / a2, r_* for RNA on a1 / a3)
template <typename R, int BX>
__global__ void pack_state_kernel(int n, R g_k1, R g_k2, R r_k1, R r_k2, const R* __restrict__ c, const R* __restrict__ q,
                                  const R* __restrict__ p, const R* __restrict__ l, const int* __restrict__ meta,
                                  const Frame<R> f, const R* __restrict__ keep_hi, const R* __restrict__ keep_lo) {
  using V4 = typename Vec4T<R>::type;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (kHiLo<R>) {
    // the caller holds fp32 centres; where they are still the values the last run handed out, the low parts that
    // run kept are restored, so a trajectory advanced in several run() calls loses nothing at the seams
    R lo[3] = {R(0), R(0), R(0)};
    if (keep_hi) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (c[3 * i + k] == keep_hi[3 * i + k]) lo[k] = keep_lo[3 * i + k];
    }
    f.pl[i] = V4{lo[0], lo[1], lo[2], R(0)};
  }
  // the kernels assume unit quaternions (torque form); normalise on entry
  R q0 = q[4 * i], q1 = q[4 * i + 1], q2 = q[4 * i + 2], q3 = q[4 * i + 3];
  const R inv = m_rsqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  q0 *= inv, q1 *= inv, q2 *= inv, q3 *= inv;
  V3<R> a1, a2, a3;
  quat_axes(q0, q1, q2, q3, a1, a2, a3);
  f.p0[i] = V4{c[3 * i], c[3 * i + 1], c[3 * i + 2], R(meta[i])};
  f.p1[i] = V4{a1.x, a1.y, a1.z, R(0)};
  f.p2[i] = V4{a3.x, a3.y, a3.z, R(0)};
  const bool rna = BX == 0 && ((meta[i] >> 3) & 1);
  const V3<R> ab = (BX == 3 || rna) ? a3 : a2;  // second axis of the backbone site
  const R k1 = rna ? r_k1 : g_k1, k2 = rna ? r_k2 : g_k2;
  f.p3[i] = V4{k1 * a1.x + k2 * ab.x, k1 * a1.y + k2 * ab.y, k1 * a1.z + k2 * ab.z, R(0)};
  f.q[i] = V4{q0, q1, q2, q3};
  f.mom[i] = V4{p[3 * i], p[3 * i + 1], p[3 * i + 2], R(0)};
  f.ang[i] = V4{l[3 * i], l[3 * i + 1], l[3 * i + 2], R(0)};
}
template <typename R>
__global__ void unpack_state_kernel(int n, const Frame<R> f, R* __restrict__ c, R* __restrict__ q,
                                    R* __restrict__ p, R* __restrict__ l, R* __restrict__ keep_hi,
                                    R* __restrict__ keep_lo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const auto a = f.p0[i];
  if constexpr (kHiLo<R>) {
    const auto lo = f.pl[i];
    keep_hi[3 * i] = a.x, keep_hi[3 * i + 1] = a.y, keep_hi[3 * i + 2] = a.z;
    keep_lo[3 * i] = lo.x, keep_lo[3 * i + 1] = lo.y, keep_lo[3 * i + 2] = lo.z;
  }
  const auto b = f.q[i];
  const auto m = f.mom[i];
  const auto w = f.ang[i];
  c[3 * i] = a.x, c[3 * i + 1] = a.y, c[3 * i + 2] = a.z;
  q[4 * i] = b.x, q[4 * i + 1] = b.y, q[4 * i + 2] = b.z, q[4 * i + 3] = b.w;
  p[3 * i] = m.x, p[3 * i + 1] = m.y, p[3 * i + 2] = m.z;
  l[3 * i] = w.x, l[3 * i + 1] = w.y, l[3 * i + 2] = w.z;
}

// Parameters (site geometry) or nucleotide types were replaced while a state is resident: the words of the frame that
// were derived from them - the meta word and the backbone offset - are derived again from the quaternion.
template <typename R, int BX>
__global__ void rederive_frame_kernel(int n, R g_k1, R g_k2, R r_k1, R r_k2, const int* __restrict__ meta, const Frame<R> f) {
  using V4 = typename Vec4T<R>::type;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V4 q = f.q[i];
  V3<R> a1, a2, a3;
  quat_axes(q.x, q.y, q.z, q.w, a1, a2, a3);
  V4 c = f.p0[i];
  c.w = R(meta[i]);
  f.p0[i] = c;
  const bool rna = BX == 0 && ((meta[i] >> 3) & 1);
  const V3<R> ab = (BX == 3 || rna) ? a3 : a2;
  const R k1 = rna ? r_k1 : g_k1, k2 = rna ? r_k2 : g_k2;
  f.p3[i] = V4{k1 * a1.x + k2 * ab.x, k1 * a1.y + k2 * ab.y, k1 * a1.z + k2 * ab.z, R(0)};
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
  // device state: two ping-pong frames of 8 vec4 arrays each (p0, p1, p2, p3, q, pl, mom, ang)
  static constexpr int kFrameArrays = 8;
  void* frame[2][kFrameArrays] = {};
  int cur = 0;             // the frame that holds the current state
  bool resident = false;   // the frames hold a state (mythos_langevin_load, or the last run)
  bool list_valid = false; // the rows were built from this state's history and the rebuild schedule continues
  int since_build = 0;     // steps taken since the rows were built
  int builds = 0;          // scheduled rebuilds so far (the chunk order is refreshed every 64th)
  int list_epoch = 0;      // sys->list_epoch the rows in use belong to
  // centres as the last run handed them out (hi) and the low parts that went with them (fp32 systems)
  void *keep_hi = nullptr, *keep_lo = nullptr;
  bool keep_valid = false;
  static constexpr int kCtlWords = 4;  // [0] error bits (2 NaN), [1] halt, [2] progress, [3] aborted launch + 1
  bool items_big = false;              // the ITEMS = 32 instantiation is in use (a launch of this load found 16 too few)
  bool want_unfused = false;           // mythos_langevin_set_option(MYTHOS_LANGEVIN_UNFUSED): takes effect at the next load
  bool unfused = false;                // the resident state lives in the unfused path's buffers (decided by load)
  int param_epoch = 0;                 // sys->param_epoch the packed site offsets of the resident frames were derived from
  int* d_flags = nullptr;
  // control words as the device published them at the end of a segment: [0..3] d_flags, [4..6] the list builder's
  // overflow words.  Pinned host memory the publishing kernel writes directly: one stream synchronisation per
  // segment and no copy commands.
  int* h_ctl = nullptr;
  int* d_ctl = nullptr;          // device address of h_ctl
  int last_recoveries = 0;       // halts of the last run that were rebuilt and resumed
  int last_rebuilds = 0;         // scheduled list rebuilds inside the last advance (the first build of a list not counted)
  bool list_fitted = false;      // a synchronising, growing build has sized rows and buckets for this integrator
  int* d_chunk_order = nullptr;  // [blocks] spatial order of the 32-nucleotide chunks (null: index order)
  unsigned long long* d_chunk_keys = nullptr;
  double* d_epart = nullptr;
  int epart_blocks = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // sampled per-launch timing: every kSampleStride-th step launch is bracketed by its own event pair
  static constexpr int kMaxSamples = 16;
  int timing_samples = 0;  // dispatches per run timed with their own event pair (set_timing; ~8 us each)
  hipEvent_t sa[kMaxSamples] = {}, sb[kMaxSamples] = {};
  // oxNA (model 4), unfused path: packed state + gradients of the energy kernel + list reference (see unfused_*)
  void *u_c = nullptr, *u_q = nullptr, *u_p = nullptr, *u_l = nullptr, *u_gc = nullptr, *u_gq = nullptr, *u_ref = nullptr;
  double* u_e = nullptr;  // [8] term energies of the last force evaluation + [2] kinetic energies
  bool u_forces_valid = false;
  double last_avg_ms = 0;     // (ev1 - ev0) / launches: includes rebuilds and inter-kernel gaps
  double last_kernel_ms = 0;  // mean over the sampled single-launch intervals
  int last_launches = 0;
  int last_samples = 0;
};

namespace mythos {

// End of a segment: hand the control words to the host (pinned memory) and clear the ones a later segment starts
// from, so that neither a copy command nor a memset sits between two runs.
__global__ void publish_ctl_kernel(int* __restrict__ flags, int* __restrict__ overflow, int* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = flags[0], out[1] = flags[1], out[2] = flags[2], out[3] = flags[3];
  out[4] = overflow ? overflow[0] : 0, out[5] = overflow ? overflow[1] : 0, out[6] = overflow ? overflow[2] : 0;
  flags[0] = 0, flags[2] = 0;  // the halt word stays until the host has recovered (later launches must see it)
}

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

template <typename R>
static MdCut<R> make_cut(const mythos_system* sys) {
  const OxParams<double>& P = sys->pd;
  // (oxNA: rbb2 and rcom2 - the coarse tests - cover all three vectors; the kernel derives the supports of each vector itself)
  double rbb = oxdna_param_max(sys, NEXC_BACKBONE_RC);
  if (sys->model >= 2) rbb = std::max(rbb, oxdna_param_max(sys, DH_RCUT));
  const double rcom = oxdna_close_range(sys);
  MdCut<R> c;
  c.rbb2 = R(rbb * rbb);
  c.rcom2 = R(rcom * rcom);
  auto sq = [](double v) { return R(v * v); };
  c.hb_lo2 = sq(P[HYDR_RCLOW]), c.hb_hi2 = sq(P[HYDR_RCHIGH]);
  c.cr_lo2 = sq(P[CRST_RCLOW]), c.cr_hi2 = sq(P[CRST_RCHIGH]);
  c.cx_lo2 = sq(P[CXST_RCLOW]), c.cx_hi2 = sq(P[CXST_RCHIGH]);
  c.hb_mask = 0;
  for (int k = 0; k < 16; ++k)
    if (P[HYDR_EPS_00 + k] != 0.0) c.hb_mask |= 1u << k;
  return c;
}

template <typename R>
static Frame<R> frame_of(const mythos_sim* sim, int k) {
  using V4 = typename Vec4T<R>::type;
  return Frame<R>{(V4*)sim->frame[k][0], (V4*)sim->frame[k][1], (V4*)sim->frame[k][2], (V4*)sim->frame[k][3],
                  (V4*)sim->frame[k][4], (V4*)sim->frame[k][5], (V4*)sim->frame[k][6], (V4*)sim->frame[k][7]};
}

// Spatial order of the workgroups' chunks (chunk_order.h): two small kernels on the run's stream, at every load and
// every 64th scheduled list rebuild (molecules drift slowly, and a stale order costs speed, not correctness).
// Systems under 64 chunks do not bother.
template <typename R>
static int update_chunk_order(mythos_sim* sim, const typename Vec4T<R>::type* p0, int blocks, hipStream_t st) {
  if (blocks < 64) return 0;
  if (!sim->d_chunk_keys) MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_chunk_keys, (size_t)blocks * sizeof(unsigned long long)));
  if (!sim->d_chunk_order) MYTHOS_HIP_TRY(hipMalloc((void**)&sim->d_chunk_order, (size_t)blocks * sizeof(int)));
  MYTHOS_HIP_TRY(chunk_order_device(p0, blocks, kMdPPB, std::max(1.0, sim->r_cut > 0 ? sim->r_cut : 4.0), sim->d_chunk_keys,
                                    sim->d_chunk_order, st));
  return 0;
}

// Caller's (N,3)/(N,4) arrays -> the resident frames.  The list of a previous state does not carry over.
template <typename R, int MODEL>
static int load_typed(mythos_sim* sim, const R* center, const R* quat, const R* p_lin, const R* p_ang, hipStream_t st) {
  mythos_system* sys = sim->sys;
  const int n = sys->n;
  const int tb = (n + 255) / 256;
  const OxParams<R>& P = params_of<R>(sys);
  const R g_k1 = P[GEO_BACK_A1], g_k2 = (MODEL >= 2) ? P[GEO_BACK_A2] : R(0);
  sim->cur = 0;
  const Frame<R> f0 = frame_of<R>(sim, 0);
  // (oxNA: the RNA nucleotides take the oxRNA2 vector's backbone site)
  const double* Prna = oxdna_param_set(sys, sys->param_sets() == 1 ? 0 : 1);
  hipLaunchKernelGGL((pack_state_kernel<R, (MODEL == 4 ? 0 : back_axis<MODEL>())>), dim3(tb), dim3(256), 0, st, n, g_k1, g_k2,
                     R(Prna[GEO_BACK_A1]), R(Prna[GEO_BACK_A2]), center, quat, p_lin, p_ang,
                     sys->d_meta, f0, sim->keep_valid ? (const R*)sim->keep_hi : nullptr,
                     (const R*)sim->keep_lo);
  MYTHOS_HIP_TRY(hipGetLastError());
  sim->resident = true;
  sim->list_valid = false;
  sim->since_build = 0;
  sim->items_big = false;
  sim->param_epoch = sys->param_epoch;
  return update_chunk_order<R>(sim, f0.p0, (n + kMdPPB - 1) / kMdPPB, st);
}

// The resident frames -> caller's arrays (asynchronous on st; the state stays resident).
template <typename R>
static int store_typed(mythos_sim* sim, R* center, R* quat, R* p_lin, R* p_ang, hipStream_t st) {
  const int n = sim->sys->n;
  hipLaunchKernelGGL(unpack_state_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, st, n, frame_of<R>(sim, sim->cur),
                     center, quat, p_lin, p_ang, (R*)sim->keep_hi, (R*)sim->keep_lo);
  MYTHOS_HIP_TRY(hipGetLastError());
  sim->keep_valid = true;
  return 0;
}

// n_steps on the resident state: n_steps + 1 launches (the last one closes the final half kick) and ONE stream
// synchronisation per segment of kSegment launches - the host has to see the halt word before it can say the steps
// were taken.  Nothing else is between two calls: the list and its rebuild schedule carry over, the control words
// are published and cleared by a one-thread kernel, events are recorded only when timing was asked for.
template <typename R, int MODEL>
static int advance_typed(mythos_sim* sim, int n_steps, int save_every, R* traj_center, R* traj_quat, double* e_trace,
                         hipStream_t st) {
  using V4 = typename Vec4T<R>::type;
  mythos_system* sys = sim->sys;
  const int n = sys->n;
  const int blocks = (n + kMdPPB - 1) / kMdPPB;
  const int grid = 8 * ((blocks + 7) / 8);  // padded for the kernel's XCD-aware workgroup order
  int sim_cus = 256;  // compute units of the device: decides between the two fp64 register allocations (md_blocks_per_cu)
  (void)hipDeviceGetAttribute(&sim_cus, hipDeviceAttributeMultiprocessorCount, sys->device);
  const R* Pdev = device_params_of<R>(sys);
  const BoxT<R> box = make_box<R>(sys);
  const LangevinConst<R> K = make_const<R>(sim);
  const MdCut<R> cut = make_cut<R>(sys);
  const Frame<R> fr[2] = {frame_of<R>(sim, 0), frame_of<R>(sim, 1)};
  int cur = sim->cur;
  if (sim->param_epoch != sys->param_epoch) {  // mythos_oxdna_set_params / set_nucleotide_types since the load
    const OxParams<R>& Ph = params_of<R>(sys);
    const double* Prna = oxdna_param_set(sys, sys->param_sets() == 1 ? 0 : 1);
    hipLaunchKernelGGL((rederive_frame_kernel<R, (MODEL == 4 ? 0 : back_axis<MODEL>())>), dim3((n + 255) / 256), dim3(256), 0, st, n,
                       Ph[GEO_BACK_A1], (MODEL >= 2) ? Ph[GEO_BACK_A2] : R(0), R(Prna[GEO_BACK_A1]), R(Prna[GEO_BACK_A2]),
                       (const int*)sys->d_meta, fr[cur]);
    MYTHOS_HIP_TRY(hipGetLastError());
    sim->param_epoch = sys->param_epoch;
  }
#ifdef MYTHOS_MD_DIAG
  const char* abl = getenv("MYTHOS_MD_ABLATE");  // profiling aid: bit 0/1/2 skip radial / angular / integrate
  const int ablate = abl ? atoi(abl) : 0;
#else
  const int ablate = 0;
#endif
  const bool dynamic_list = sim->rebuild_every > 0;
  const bool timing = sim->timing_samples > 0;
  // a probabilistic sequence (mythos_oxdna_set_pseq): the PSEQ instantiations, which exist with the wide work lists only
  PseqView<R> pseq;
  const bool use_pseq = sys->pseq_terms != 0;
  if (use_pseq) {
    pseq.marg = (const R*)sys->d_ps_marg, pseq.unit = sys->d_ps_unit, pseq.bp = (const R*)sys->d_ps_bp, pseq.terms = sys->pseq_terms;
    sim->items_big = true;
  }
  auto rebuild = [&](int buf) -> int {
    if ((++sim->builds & 63) == 0)
      if (int rc = update_chunk_order<R>(sim, fr[buf].p0, blocks, st)) return rc;
    return rows_build_device(sys, fr[buf].p0, true, sim->r_cut, sim->skin, fr[buf].p3, fr[buf].p1, true, st);
  };
  // k index at which the rows in use were built (negative: so many steps before this call)
  int built_at = 0;
  if (dynamic_list) {
    if (!sim->list_fitted) {
      // the first build of this integrator sizes rows (a quarter of headroom) and cell buckets (none more than half
      // full) with a synchronising build; later ones just rebuild - should that overflow, the next step kernel
      // halts and the recovery below grows what is needed
      if (int rc = rows_build_until_fit(sys, fr[cur].p0, true, sim->r_cut, sim->skin, fr[cur].p3, fr[cur].p1, true, true, st))
        return rc;
      sim->list_fitted = true;
    } else if (!sim->list_valid) {
      if (int rc = rebuild(cur)) return rc;
    } else {
      built_at = -sim->since_build;
    }
    sim->list_valid = true;
  }
  if (timing) MYTHOS_HIP_TRY(hipEventRecord(sim->ev0, st));
  int launches = 0, samples = 0, recoveries = 0, scheduled_rebuilds = 0;
  const int max_samples = std::min(sim->timing_samples, (int)mythos_sim::kMaxSamples);  // 0: no dispatch is bracketed
  const int sample_stride = std::max(1, (n_steps + 1) / std::max(1, max_samples));
  int* halt_words = dynamic_list ? sys->d_overflow : nullptr;
  // The kernels of a run are queued in segments of kSegment; after each the host looks at the halt word.  A step that
  // moves a site out of its skin, or a rebuild that overflows its rows or spill list, halts the launches behind it
  // (they return at once); the host then rebuilds at the last valid state - growing what overflowed - and resumes
  // there.  A run never integrates on a stale or truncated list, and neither condition is an error any more; what
  // it costs is the empty launches behind the halt (at most a segment) and a synchronisation.
  constexpr int kMaxRecoveries = 64;
  const long long dbg_seg = debug_value(MYTHOS_DEBUG_MD_SEGMENT);
  const int kSegment = dbg_seg > 0 ? (int)std::min<long long>(dbg_seg, 1 << 20) : 8192;
  int k = 0, seg_len = kSegment;  // a run that has halted once looks more often: less queued behind the next halt
  int err_bits = 0, ovw[kOverflowWords] = {0, 0, 0};
  while (k <= n_steps) {
    const int seg_end = std::min(n_steps, k + seg_len - 1);
    // The device's progress word (flags[2], cleared by publish_ctl_kernel after every segment) says nothing when the
    // FIRST launch of a segment halts before writing it (a scheduled rebuild in front of it overflowed): the launches
    // of the earlier segments count all the same.
    const int seg_start = k;
    for (; k <= seg_end; ++k) {
      const bool last = (k == n_steps);
      const bool save = save_every > 0 && k > 0 && (k % save_every == 0);
      const int sidx = save ? (k / save_every - 1) : 0;
      if (dynamic_list && !last && k - built_at >= sim->rebuild_every) {
        if (int rc = rebuild(cur)) return rc;
        built_at = k;
        ++scheduled_rebuilds;
        if (debug_value(MYTHOS_DEBUG_MD_OVERFLOW_AT) == k + 1) {  // test hook: this build claims a row did not fit
          debug_clear(MYTHOS_DEBUG_MD_OVERFLOW_AT);
          MYTHOS_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)sys->d_overflow, sys->row_stride + 1, 1, st));
        }
      }
      const R kick_close = (k == 0) ? R(0) : R(0.5);
      const int do_step = last ? 0 : 1;
      R* tc = (save && traj_center) ? traj_center + (size_t)sidx * n * 3 : nullptr;
      R* tq = (save && traj_quat) ? traj_quat + (size_t)sidx * n * 4 : nullptr;
      const V4* ref = (const V4*)sys->d_ref_pos;
      const V4* ref_off = (const V4*)sys->d_ref_off;
      const V4* ref_a1 = (const V4*)sys->d_ref_a1;
      const bool sampled = !save && (k % sample_stride == sample_stride / 2) && samples < max_samples;
      auto launch_pseq = [&](auto save_tag, hipEvent_t ea, hipEvent_t eb) {  // (the wide work lists only, see md_step_kernel)
        constexpr bool SV = decltype(save_tag)::value;
        hipExtLaunchKernelGGL((md_step_kernel<R, MODEL, SV, md_items_big<R, SV>(), false, true>), dim3(grid), dim3(kMdBlock), 0, st, ea, eb, 0,
                              Pdev, box, K, cut, n, fr[cur], fr[cur ^ 1], sys->d_rows, sys->d_row_len, row_close_of(sys),
                              sys->row_stride, sys->extra_bonds ? 1 : 0, kick_close, do_step, sim->seed, (uint64_t)(sim->step + k), ref,
                              ref_off, ref_a1, sim->d_flags, tc, tq, sim->d_epart, sim->d_chunk_order, halt_words, k, ablate, pseq);
      };
      auto launch = [&](auto save_tag, auto items_tag, hipEvent_t ea, hipEvent_t eb) {
        constexpr bool SV = decltype(save_tag)::value;
        constexpr int IT = decltype(items_tag)::value;
        // with events: the pair receives the begin / end time stamps of THIS dispatch (the same stamps a profiler's
        // kernel trace reports), not the time between two markers in the queue
        auto go = [&](auto dense_tag) {
#ifdef MYTHOS_MD_PLAIN_LAUNCH  // (dev A/B: the host cost of the two launch calls)
          if (!ea && !eb) {
            hipLaunchKernelGGL((md_step_kernel<R, MODEL, SV, IT, decltype(dense_tag)::value>), dim3(grid), dim3(kMdBlock), 0, st,
                               Pdev, box, K, cut, n, fr[cur], fr[cur ^ 1], sys->d_rows, sys->d_row_len, row_close_of(sys),
                               sys->row_stride, sys->extra_bonds ? 1 : 0, kick_close, do_step, sim->seed, (uint64_t)(sim->step + k), ref,
                               ref_off, ref_a1, sim->d_flags, tc, tq, sim->d_epart, sim->d_chunk_order, halt_words, k, ablate, PseqView<R>{});
            return;
          }
#endif
          hipExtLaunchKernelGGL((md_step_kernel<R, MODEL, SV, IT, decltype(dense_tag)::value>), dim3(grid), dim3(kMdBlock), 0, st, ea, eb, 0,
                                Pdev, box, K, cut, n, fr[cur], fr[cur ^ 1], sys->d_rows, sys->d_row_len, row_close_of(sys),
                                sys->row_stride, sys->extra_bonds ? 1 : 0, kick_close, do_step, sim->seed, (uint64_t)(sim->step + k), ref,
                                ref_off, ref_a1, sim->d_flags, tc, tq, sim->d_epart, sim->d_chunk_order, halt_words, k, ablate, PseqView<R>{});
        };
        if constexpr (sizeof(R) == 8 && !SV && IT == kMdItems) {
          if (grid > 2 * sim_cus) go(std::true_type{}); else go(std::false_type{});
        } else {
          go(std::false_type{});
        }
      };
      using T = std::true_type;
      using F = std::false_type;
      using Small = std::integral_constant<int, kMdItems>;
      using BigS = std::integral_constant<int, md_items_big<R, true>()>;
      using BigN = std::integral_constant<int, md_items_big<R, false>()>;
      hipEvent_t ea = nullptr, eb = nullptr;
      if (sampled) ea = sim->sa[samples], eb = sim->sb[samples], ++samples;
      if (use_pseq) {
        if (save) launch_pseq(T{}, ea, eb); else launch_pseq(F{}, ea, eb);
        if (save)
          hipLaunchKernelGGL(reduce_trace_kernel, dim3(1), dim3(256), 0, st, sim->d_epart, blocks,
                             e_trace ? e_trace + (size_t)sidx * kTraceWidth : nullptr);
      } else if (save) {
        if (sim->items_big) launch(T{}, BigS{}, ea, eb); else launch(T{}, Small{}, ea, eb);
        hipLaunchKernelGGL(reduce_trace_kernel, dim3(1), dim3(256), 0, st, sim->d_epart, blocks,
                           e_trace ? e_trace + (size_t)sidx * kTraceWidth : nullptr);
      } else {
        if (sim->items_big) launch(F{}, BigN{}, ea, eb); else launch(F{}, Small{}, ea, eb);
      }
      ++launches;
      cur ^= 1;
    }
    if (timing && k > n_steps) MYTHOS_HIP_TRY(hipEventRecord(sim->ev1, st));
    hipLaunchKernelGGL(publish_ctl_kernel, dim3(1), dim3(1), 0, st, sim->d_flags, halt_words, sim->d_ctl);
    MYTHOS_HIP_TRY(hipGetLastError());
    MYTHOS_HIP_TRY(hipStreamSynchronize(st));
    const int* ctl = sim->h_ctl;
    err_bits |= ctl[0];
    for (int w = 0; w < kOverflowWords; ++w) ovw[w] = ctl[4 + w];
    if ((err_bits & 2) != 0) break;                              // NaN: reported below
    const int aborted = ctl[3];  // launch index + 1 whose angular work lists were too short (its output does not count)
    if (ctl[1] == 0 && ovw[0] == 0 && ovw[1] == 0 && aborted == 0) continue;  // nothing halted
    if (aborted != 0) {
      if (sim->items_big) {
        // what is handed back: the positions after the last step that counted, momenta short of its closing half kick
        sim->cur ^= ((aborted - 1) & 1);
        sim->step += aborted - 1;
        set_error("mythos_langevin_run: more than " + std::to_string(md_items_big<R, false>()) + " (" + std::to_string(md_items_big<R, true>()) +
                  " on steps that save energies)"
                  " neighbours of one nucleotide are inside the range of an angular term (overlapping bases?)");
        return MYTHOS_ERR_OVERFLOW;
      }
      sim->items_big = true;  // run that step again, and the rest of the run, with the wider instantiation
    } else if (!dynamic_list) {
      break;  // (a static list cannot halt; defensive)
    }
    // kernels 0 .. ran-1 count; the state they left is in the frame kernel `ran` reads (an aborted launch and
    // everything behind it do not count: their inputs are untouched)
    const int progressed = std::max(ctl[2], seg_start);
    const int ran = aborted != 0 ? std::min(progressed, aborted - 1) : progressed;
    if (++recoveries > kMaxRecoveries) {
      sim->cur ^= (ran & 1);  // positions after the last step that counted, momenta short of its closing half kick
      sim->step += ran;
      set_error("mythos_langevin_run: the neighbour list had to be rebuilt out of turn more than " + std::to_string(kMaxRecoveries) +
                " times in one run: the skin (" + std::to_string(sim->skin) + ") is too small for a rebuild every " +
                std::to_string(sim->rebuild_every) + " steps");
      return MYTHOS_ERR_OVERFLOW;
    }
    cur = sim->cur ^ (ran & 1);
    k = ran;
    seg_len = std::max(std::min(256, kSegment), seg_len / 4);
    MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags + 1, 0, 3 * sizeof(int), st));
    if (dynamic_list) {
      if (int rc = rows_build_until_fit(sys, fr[cur].p0, true, sim->r_cut, sim->skin, fr[cur].p3, fr[cur].p1, true, true, st))
        return rc;
      built_at = k;
    }
    ovw[0] = ovw[1] = 0;
  }
  sim->last_recoveries = recoveries;
  sim->last_rebuilds = scheduled_rebuilds;
  sim->cur = cur;
  sim->since_build = n_steps - built_at;
  if (timing) {
    float ms = 0;
    MYTHOS_HIP_TRY(hipEventElapsedTime(&ms, sim->ev0, sim->ev1));
    sim->last_avg_ms = launches ? double(ms) / launches : 0.0;
    double acc = 0;
    for (int s = 0; s < samples; ++s) {
      float t = 0;
      MYTHOS_HIP_TRY(hipEventElapsedTime(&t, sim->sa[s], sim->sb[s]));
      acc += t;
    }
    sim->last_kernel_ms = samples ? acc / samples : 0.0;
  } else {
    sim->last_avg_ms = sim->last_kernel_ms = 0.0;
  }
  sim->last_launches = launches;
  sim->last_samples = samples;
  if (ablate & 128) {  // diagnostic: dump the cycle stamps of the last launch
    if (const char* path = getenv("MYTHOS_MD_STAMPS")) {
      std::vector<unsigned long long> h((size_t)blocks * 64);  // two launches: even step | odd step
      MYTHOS_HIP_TRY(hipMemcpy(h.data(), sim->d_epart, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) {
        fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        fclose(f);
      }
    }
  }
  sim->step += n_steps;
  if (err_bits & 2) {
    sim->resident = false;
    set_error("mythos_langevin_run: NaN in the state (time step too large or overlapping start configuration)");
    return MYTHOS_ERR_NUMERIC;
  }
  if (dynamic_list && ovw[0] != 0) {
    set_error("mythos_langevin_run: neighbour row capacity exceeded (" + std::to_string(ovw[0]) + " > " +
              std::to_string(sys->row_stride) + "); rebuild with mythos_oxdna_build_neighbors first");
    return MYTHOS_ERR_OVERFLOW;
  }
  if (dynamic_list && ovw[1] != 0) {
    set_error("mythos_langevin_run: too many nucleotides (" + std::to_string(ovw[1]) +
              ") did not fit the buckets of their cells during a neighbour rebuild");
    return MYTHOS_ERR_OVERFLOW;
  }
  return MYTHOS_OK;
}

// ------------------------------------------------------------------------------------------------
// oxNA (model 4): the UNFUSED path, behind MYTHOS_NA1_UNFUSED=1 - a second implementation of a hybrid system's dynamics
// that the tests hold md_step_kernel<R, 4, ...> to (it came first and stayed as the cross-check).  Two launches per step:
// the energy kernel's forces instantiation (dU/dcentre, dU/dquaternion of the packed state), then this integrator
// kernel, one thread per nucleotide: the same B A O A | B map, Philox stream and free-rotor drift as md_step_kernel's
// integrator (shared device functions), so a trajectory is held to the same oracle.
// The list: static rows (mythos_oxdna_set_neighbors), or the integrator's policy - rows of range r_cut + skin rebuilt
// every rebuild_every steps from the centres; the host looks at the skin flag at every rebuild (it synchronises there
// anyway) and a violation is an error (no halt-and-resume on this path): shorten the interval or widen the skin.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ void unfused_integrate_kernel(int n, const LangevinConst<R> K, R* __restrict__ c, R* __restrict__ q, R* __restrict__ p,
                                         R* __restrict__ L, const R* __restrict__ gc, const R* __restrict__ gq, R kick_close,
                                         int do_step, uint64_t seed, uint64_t step, const R* __restrict__ ref, R site_reach,
                                         int* __restrict__ flags, R* __restrict__ traj_c, R* __restrict__ traj_q,
                                         double* __restrict__ ke /* [2], atomics; null: not wanted */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double ke_t = 0.0, ke_r = 0.0;
  if (i < n) {
    R x[3] = {c[3 * i], c[3 * i + 1], c[3 * i + 2]};
    R qs[4] = {q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]};
    R pp[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}, LL[3] = {L[3 * i], L[3 * i + 1], L[3 * i + 2]};
    const R F[3] = {-gc[3 * i], -gc[3 * i + 1], -gc[3 * i + 2]};
    const R g0 = gq[4 * i], g1 = gq[4 * i + 1], g2 = gq[4 * i + 2], g3 = gq[4 * i + 3];
    // body torque from the quaternion gradient: tau_k = -1/2 (P_k q) . dU/dq (NO_SQUISH permutations)
    const R tb[3] = {R(-0.5) * (-qs[1] * g0 + qs[0] * g1 + qs[3] * g2 - qs[2] * g3),
                     R(-0.5) * (-qs[2] * g0 - qs[3] * g1 + qs[0] * g2 + qs[1] * g3),
                     R(-0.5) * (-qs[3] * g0 + qs[2] * g1 - qs[1] * g2 + qs[0] * g3)};
    const R kc = kick_close * K.dt;
#pragma unroll
    for (int k = 0; k < 3; ++k) pp[k] += kc * F[k], LL[k] += kc * tb[k];
    if (ke) {
      ke_t = 0.5 * double(K.inv_mass) * (double(pp[0]) * pp[0] + double(pp[1]) * pp[1] + double(pp[2]) * pp[2]);
      ke_r = 0.5 * (double(K.inv_inertia[0]) * LL[0] * LL[0] + double(K.inv_inertia[1]) * LL[1] * LL[1] +
                    double(K.inv_inertia[2]) * LL[2] * LL[2]);
    }
    if (traj_c) traj_c[3 * i] = x[0], traj_c[3 * i + 1] = x[1], traj_c[3 * i + 2] = x[2];
    if (traj_q) traj_q[4 * i] = qs[0], traj_q[4 * i + 1] = qs[1], traj_q[4 * i + 2] = qs[2], traj_q[4 * i + 3] = qs[3];
    if (do_step) {
      R z[6];
      normals6(seed, (uint32_t)i, step, 0u, z);
#pragma unroll
      for (int k = 0; k < 3; ++k) pp[k] += K.half_dt * F[k], LL[k] += K.half_dt * tb[k];
      drift(x, qs, pp, LL, K.half_dt, K);
#pragma unroll
      for (int k = 0; k < 3; ++k) pp[k] = K.c1_t * pp[k] + K.c2_t * z[k], LL[k] = K.c1_r * LL[k] + K.c2_r[k] * z[3 + k];
      drift(x, qs, pp, LL, K.half_dt, K);
      const R inv = m_rsqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) qs[k] *= inv;
      if (!(x[0] == x[0]) || !(qs[0] == qs[0])) atomicOr(flags, 2);
      if (ref != nullptr) {
        // no site may have moved more than skin / 2 since the build: |d site| <= |d centre| + sum_k |coef_k| |d a_k|
        // (site_reach bounds the sum of the offset coefficients of any site in either geometry)
        V3<R> a1, a2, a3;
        quat_axes(qs[0], qs[1], qs[2], qs[3], a1, a2, a3);
        const R* rr = ref + 12 * (size_t)i;
        const V3<R> dx{x[0] - rr[0], x[1] - rr[1], x[2] - rr[2]};
        const V3<R> d1{a1.x - rr[3], a1.y - rr[4], a1.z - rr[5]}, d2{a2.x - rr[6], a2.y - rr[7], a2.z - rr[8]},
            d3{a3.x - rr[9], a3.y - rr[10], a3.z - rr[11]};
        const R da = m_sqrt(fmax(dot(d1, d1), fmax(dot(d2, d2), dot(d3, d3))));
        const R moved = m_sqrt(dot(dx, dx)) + site_reach * da;
        if (moved * moved > K.skin_half_sq) atomicOr(flags + 1, 1);
      }
      c[3 * i] = x[0], c[3 * i + 1] = x[1], c[3 * i + 2] = x[2];
      q[4 * i] = qs[0], q[4 * i + 1] = qs[1], q[4 * i + 2] = qs[2], q[4 * i + 3] = qs[3];
    }
    p[3 * i] = pp[0], p[3 * i + 1] = pp[1], p[3 * i + 2] = pp[2];
    L[3 * i] = LL[0], L[3 * i + 1] = LL[1], L[3 * i + 2] = LL[2];
  }
  if (ke) {  // one atomic pair per wavefront
    for (int o = 32; o > 0; o >>= 1) ke_t += __shfl_down(ke_t, o, 64), ke_r += __shfl_down(ke_r, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(ke, ke_t), atomicAdd(ke + 1, ke_r);
  }
}

// list reference of the unfused path: centre and the three axes at build time
template <typename R>
__global__ void unfused_ref_kernel(int n, const R* __restrict__ c, const R* __restrict__ q, R* __restrict__ ref) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3<R> a1, a2, a3;
  quat_axes(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3], a1, a2, a3);
  R* rr = ref + 12 * (size_t)i;
  rr[0] = c[3 * i], rr[1] = c[3 * i + 1], rr[2] = c[3 * i + 2];
  rr[3] = a1.x, rr[4] = a1.y, rr[5] = a1.z, rr[6] = a2.x, rr[7] = a2.y, rr[8] = a2.z, rr[9] = a3.x, rr[10] = a3.y, rr[11] = a3.z;
}

__global__ void unfused_trace_kernel(const double* __restrict__ e, double* __restrict__ row) {
  if (threadIdx.x < kTraceWidth) row[threadIdx.x] = e[threadIdx.x];
}

template <typename R>
static int unfused_alloc(mythos_sim* sim) {
  if (sim->u_c) return 0;
  const size_t n = (size_t)sim->sys->n;
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_c, 3 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_q, 4 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_p, 3 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_l, 3 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_gc, 3 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_gq, 4 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc(&sim->u_ref, 12 * n * sizeof(R)));
  MYTHOS_HIP_TRY(hipMalloc((void**)&sim->u_e, kTraceWidth * sizeof(double)));
  return 0;
}

template <typename R>
static int unfused_load(mythos_sim* sim, const R* c, const R* q, const R* p, const R* l, hipStream_t st) {
  if (int rc = unfused_alloc<R>(sim)) return rc;
  const size_t n = (size_t)sim->sys->n;
  MYTHOS_HIP_TRY(hipMemcpyAsync(sim->u_c, c, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(sim->u_q, q, 4 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(sim->u_p, p, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(sim->u_l, l, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  sim->resident = true;
  sim->list_valid = false;
  sim->since_build = 0;
  sim->u_forces_valid = false;
  return 0;
}

template <typename R>
static int unfused_store(mythos_sim* sim, R* c, R* q, R* p, R* l, hipStream_t st) {
  const size_t n = (size_t)sim->sys->n;
  MYTHOS_HIP_TRY(hipMemcpyAsync(c, sim->u_c, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(q, sim->u_q, 4 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(p, sim->u_p, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  MYTHOS_HIP_TRY(hipMemcpyAsync(l, sim->u_l, 3 * n * sizeof(R), hipMemcpyDeviceToDevice, st));
  return 0;
}

template <typename R>
static int unfused_advance(mythos_sim* sim, int n_steps, int save_every, R* traj_c, R* traj_q, double* e_trace, hipStream_t st) {
  mythos_system* sys = sim->sys;
  const int n = sys->n, tb = (n + 255) / 256;
  const LangevinConst<R> K = make_const<R>(sim);
  R *c = (R*)sim->u_c, *q = (R*)sim->u_q, *p = (R*)sim->u_p, *l = (R*)sim->u_l, *gc = (R*)sim->u_gc, *gq = (R*)sim->u_gq;
  R* ref = (R*)sim->u_ref;
  const bool dynamic = sim->rebuild_every > 0;
  sim->last_recoveries = 0;
  // the sum of the offset coefficients of the farthest site, over both geometries (bounds a site's motion under rotation)
  double reach = 0.0;
  for (int k = 0; k < 2; ++k) {
    const double* P = sys->pd_sets.data() + (size_t)k * OXP_COUNT;
    reach = std::max({reach, std::fabs(P[GEO_BACK_A1]) + std::fabs(P[GEO_BACK_A2]), std::fabs(P[GEO_BASE]), std::fabs(P[GEO_STACK]),
                      std::fabs(P[GEO_STACK3_A1]) + std::fabs(P[GEO_STACK3_A2]), std::fabs(P[GEO_STACK5_A1]) + std::fabs(P[GEO_STACK5_A2])});
  }
  auto build = [&]() -> int {
    if (int rc = rows_build_until_fit(sys, c, false, sim->r_cut, sim->skin, nullptr, nullptr, false, true, st)) return rc;
    hipLaunchKernelGGL(unfused_ref_kernel<R>, dim3(tb), dim3(256), 0, st, n, (const R*)c, (const R*)q, ref);
    sim->since_build = 0;
    sim->list_valid = true;
    sim->list_epoch = ++sys->list_epoch;
    return 0;
  };
  auto forces = [&]() -> int {
    return oxdna_energy_launch(sys, c, q, 1, sim->u_e, gc, gq, nullptr, nullptr, nullptr, st);
  };
  auto check_flags = [&](const char* when) -> int {
    int fl[2] = {0, 0};
    MYTHOS_HIP_TRY(hipMemcpyAsync(fl, sim->d_flags, sizeof(fl), hipMemcpyDeviceToHost, st));
    MYTHOS_HIP_TRY(hipStreamSynchronize(st));
    MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags, 0, 2 * sizeof(int), st));
    if (fl[0] & 2) {
      sim->resident = false;
      set_error(std::string("mythos_langevin_run (oxNA, unfused): NaN in the state ") + when);
      return MYTHOS_ERR_NUMERIC;
    }
    if (fl[1] != 0) {
      sim->resident = false;
      set_error("mythos_langevin_run (oxNA, unfused): a site moved more than skin / 2 between two list rebuilds; shorten "
                "rebuild_every or widen the skin (this path does not halt and resume)");
      return MYTHOS_ERR_OVERFLOW;
    }
    return 0;
  };
  MYTHOS_HIP_TRY(hipMemsetAsync(sim->d_flags, 0, mythos_sim::kCtlWords * sizeof(int), st));
  if (!sim->list_valid) sim->u_forces_valid = false;  // parameters or rows were replaced since the last force evaluation
  if (dynamic && !sim->list_valid)
    if (int rc = build()) return rc;
  if (!sim->u_forces_valid) {
    if (int rc = forces()) return rc;
    sim->u_forces_valid = true;
  }
  int saved = 0;
  for (int k = 0; k <= n_steps; ++k) {
    // launch k: close the kick of step k - 1 (the forces at x_k are in gc / gq), record x_k, then step k -> k + 1
    const bool do_step = k < n_steps;
    const bool save = save_every > 0 && k > 0 && k % save_every == 0;
    if (k == 0 && !do_step) break;  // zero steps: nothing to close
    R* tc = (save && traj_c) ? traj_c + (size_t)saved * n * 3 : nullptr;
    R* tq = (save && traj_q) ? traj_q + (size_t)saved * n * 4 : nullptr;
    double* ke = save ? sim->u_e + T_COUNT : nullptr;
    if (save) MYTHOS_HIP_TRY(hipMemsetAsync(sim->u_e + T_COUNT, 0, 2 * sizeof(double), st));
    hipLaunchKernelGGL(unfused_integrate_kernel<R>, dim3(tb), dim3(256), 0, st, n, K, c, q, p, l, (const R*)gc, (const R*)gq,
                       R(k > 0 ? 0.5 : 0.0), do_step ? 1 : 0, sim->seed, (uint64_t)(sim->step + k), dynamic ? (const R*)ref : nullptr,
                       R(reach), sim->d_flags, tc, tq, ke);
    if (save) {
      if (e_trace) hipLaunchKernelGGL(unfused_trace_kernel, dim3(1), dim3(64), 0, st, (const double*)sim->u_e, e_trace + (size_t)saved * kTraceWidth);
      ++saved;
    }
    if (!do_step) break;
    ++sim->since_build;
    if (dynamic && sim->since_build >= sim->rebuild_every) {
      if (int rc = check_flags("before a list rebuild")) return rc;
      if (int rc = build()) return rc;
    }
    if (int rc = forces()) return rc;
  }
  MYTHOS_HIP_TRY(hipGetLastError());
  if (int rc = check_flags("at the end of the run")) return rc;
  sim->step += n_steps;
  return 0;
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
  s->epart_blocks = (sys->n + kMdPPB - 1) / kMdPPB;
  bool ok = true;
  for (int k = 0; k < 2; ++k)
    for (int a = 0; a < mythos_sim::kFrameArrays; ++a) ok = ok && hipMalloc(&s->frame[k][a], v4) == hipSuccess;
  ok = ok && hipMalloc(&s->keep_hi, v4) == hipSuccess && hipMalloc(&s->keep_lo, v4) == hipSuccess;
  ok = ok && hipMalloc((void**)&s->d_flags, mythos_sim::kCtlWords * sizeof(int)) == hipSuccess &&
       hipMalloc((void**)&s->d_epart, (size_t)s->epart_blocks * 64 * sizeof(double)) == hipSuccess &&
       hipEventCreate(&s->ev0) == hipSuccess && hipEventCreate(&s->ev1) == hipSuccess &&
       hipHostMalloc((void**)&s->h_ctl, 8 * sizeof(int), hipHostMallocDefault) == hipSuccess &&
       hipHostGetDevicePointer((void**)&s->d_ctl, s->h_ctl, 0) == hipSuccess &&
       hipMemset(s->d_flags, 0, mythos_sim::kCtlWords * sizeof(int)) == hipSuccess;
  if (ok) std::fill(s->h_ctl, s->h_ctl + 8, 0);
  for (int k = 0; ok && k < mythos_sim::kMaxSamples; ++k)
    ok = hipEventCreate(&s->sa[k]) == hipSuccess && hipEventCreate(&s->sb[k]) == hipSuccess;
  if (ok && !sys->d_ref_pos) ok = hipMalloc(&sys->d_ref_pos, v4) == hipSuccess;
  if (ok && !sys->d_ref_off) ok = hipMalloc(&sys->d_ref_off, v4) == hipSuccess;
  if (ok && !sys->d_ref_a1) ok = hipMalloc(&sys->d_ref_a1, v4) == hipSuccess;
  if (!ok) {
    set_error("mythos_langevin_create: device allocation failed");
    mythos_langevin_destroy(s);
    return nullptr;
  }
  return s;
}

void mythos_langevin_destroy(mythos_sim_t* s) {
  if (!s) return;
  for (int k = 0; k < 2; ++k)
    for (int a = 0; a < mythos_sim::kFrameArrays; ++a)
      if (s->frame[k][a]) (void)hipFree(s->frame[k][a]);
  for (void* u : {s->u_c, s->u_q, s->u_p, s->u_l, s->u_gc, s->u_gq, s->u_ref, (void*)s->u_e})
    if (u) (void)hipFree(u);
  if (s->keep_hi) (void)hipFree(s->keep_hi);
  if (s->keep_lo) (void)hipFree(s->keep_lo);
  if (s->d_flags) (void)hipFree(s->d_flags);
  if (s->d_chunk_order) (void)hipFree(s->d_chunk_order);
  if (s->d_chunk_keys) (void)hipFree(s->d_chunk_keys);
  if (s->h_ctl) (void)hipHostFree(s->h_ctl);
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
  s->list_fitted = false;  // another list range: size rows and buckets again at the next run
  s->list_valid = false;
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

namespace {

// what a state needs before it can be packed into frames: the site geometry (parameters) and, for oxNA, the types that
// choose between the two geometries
int md_ready_state(mythos_sim_t* s, const char* who) {
  mythos_system* sys = s->sys;
  if (!sys->params_set) {
    set_error(std::string(who) + ": parameters must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  if (sys->model == 4 && !sys->types_set) {
    set_error(std::string(who) + ": an oxNA system needs its nucleotide types (mythos_oxdna_set_nucleotide_types)");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(sys->device));
  return MYTHOS_OK;
}

// what every entry that launches step kernels checks first
int md_ready(mythos_sim_t* s, const char* who) {
  mythos_system* sys = s->sys;
  if (int rc = md_ready_state(s, who)) return rc;
  if (!sys->nbrs_set && s->rebuild_every <= 0) {
    set_error(std::string(who) + ": parameters and neighbours (or a neighbour policy) must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  if (s->rebuild_every > 0 && sys->row_stride == 0)
    if (int rc = rows_reserve(sys, 64)) return rc;
  if (s->list_epoch != sys->list_epoch) {  // parameters or rows were replaced behind the integrator's back
    s->list_valid = false;
    s->list_epoch = sys->list_epoch;
  }
  return MYTHOS_OK;
}

// oxNA systems step through the fused kernel's MODEL 4 instantiation; mythos_langevin_set_option(MYTHOS_LANGEVIN_UNFUSED)
// selects the two-launch path (the energy kernel's forces + unfused_integrate_kernel) instead - a second implementation
// the tests hold the first to.  The choice is made when a state is loaded and holds while that state is resident.

int md_load(mythos_sim_t* s, void* c, void* q, void* p, void* l, hipStream_t st) {
  mythos_system* sys = s->sys;
  s->unfused = s->want_unfused && sys->model == 4;
  if (s->unfused)
    return sys->dtype == MYTHOS_F32 ? unfused_load<float>(s, (float*)c, (float*)q, (float*)p, (float*)l, st)
                                    : unfused_load<double>(s, (double*)c, (double*)q, (double*)p, (double*)l, st);
  if (sys->dtype == MYTHOS_F32)
    return sys->model == 1   ? load_typed<float, 1>(s, (float*)c, (float*)q, (float*)p, (float*)l, st)
           : sys->model == 2 ? load_typed<float, 2>(s, (float*)c, (float*)q, (float*)p, (float*)l, st)
           : sys->model == 3 ? load_typed<float, 3>(s, (float*)c, (float*)q, (float*)p, (float*)l, st)
                             : load_typed<float, 4>(s, (float*)c, (float*)q, (float*)p, (float*)l, st);
  return sys->model == 1   ? load_typed<double, 1>(s, (double*)c, (double*)q, (double*)p, (double*)l, st)
         : sys->model == 2 ? load_typed<double, 2>(s, (double*)c, (double*)q, (double*)p, (double*)l, st)
         : sys->model == 3 ? load_typed<double, 3>(s, (double*)c, (double*)q, (double*)p, (double*)l, st)
                           : load_typed<double, 4>(s, (double*)c, (double*)q, (double*)p, (double*)l, st);
}

int md_advance(mythos_sim_t* s, int n_steps, int save_every, void* tc, void* tq, double* e_trace, hipStream_t st) {
  mythos_system* sys = s->sys;
  if (s->unfused)
    return sys->dtype == MYTHOS_F32 ? unfused_advance<float>(s, n_steps, save_every, (float*)tc, (float*)tq, e_trace, st)
                                    : unfused_advance<double>(s, n_steps, save_every, (double*)tc, (double*)tq, e_trace, st);
  if (sys->dtype == MYTHOS_F32)
    return sys->model == 1   ? advance_typed<float, 1>(s, n_steps, save_every, (float*)tc, (float*)tq, e_trace, st)
           : sys->model == 2 ? advance_typed<float, 2>(s, n_steps, save_every, (float*)tc, (float*)tq, e_trace, st)
           : sys->model == 3 ? advance_typed<float, 3>(s, n_steps, save_every, (float*)tc, (float*)tq, e_trace, st)
                             : advance_typed<float, 4>(s, n_steps, save_every, (float*)tc, (float*)tq, e_trace, st);
  return sys->model == 1   ? advance_typed<double, 1>(s, n_steps, save_every, (double*)tc, (double*)tq, e_trace, st)
         : sys->model == 2 ? advance_typed<double, 2>(s, n_steps, save_every, (double*)tc, (double*)tq, e_trace, st)
         : sys->model == 3 ? advance_typed<double, 3>(s, n_steps, save_every, (double*)tc, (double*)tq, e_trace, st)
                           : advance_typed<double, 4>(s, n_steps, save_every, (double*)tc, (double*)tq, e_trace, st);
}

int md_store(mythos_sim_t* s, void* c, void* q, void* p, void* l, hipStream_t st) {
  if (s->unfused)
    return s->sys->dtype == MYTHOS_F32 ? unfused_store<float>(s, (float*)c, (float*)q, (float*)p, (float*)l, st)
                                       : unfused_store<double>(s, (double*)c, (double*)q, (double*)p, (double*)l, st);
  if (s->sys->dtype == MYTHOS_F32) return store_typed<float>(s, (float*)c, (float*)q, (float*)p, (float*)l, st);
  return store_typed<double>(s, (double*)c, (double*)q, (double*)p, (double*)l, st);
}

}  // namespace

int mythos_langevin_run(mythos_sim_t* s, void* center, void* quat, void* p_lin, void* p_ang, int n_steps,
                        int save_every, void* traj_center, void* traj_quat, double* e_trace,
                        mythos_stream_t stream) {
  if (!s || !center || !quat || !p_lin || !p_ang || n_steps < 0 || save_every < 0) {
    set_error("mythos_langevin_run: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (int rc = md_ready(s, "mythos_langevin_run")) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = md_load(s, center, quat, p_lin, p_ang, st)) return rc;
  const int rc = md_advance(s, n_steps, save_every, traj_center, traj_quat, e_trace, st);
  // the state of the last valid step goes back to the caller whatever the run reported
  if (int rs = md_store(s, center, quat, p_lin, p_ang, st)) return rc ? rc : rs;
  return rc;
}

int mythos_langevin_load(mythos_sim_t* s, const void* center, const void* quat, const void* p_lin, const void* p_ang,
                         mythos_stream_t stream) {
  if (!s || !center || !quat || !p_lin || !p_ang) {
    set_error("mythos_langevin_load: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (int rc = md_ready_state(s, "mythos_langevin_load")) return rc;
  return md_load(s, (void*)center, (void*)quat, (void*)p_lin, (void*)p_ang, (hipStream_t)stream);
}

int mythos_langevin_advance(mythos_sim_t* s, int n_steps, int save_every, void* traj_center, void* traj_quat,
                            double* e_trace, mythos_stream_t stream) {
  if (!s || n_steps < 0 || save_every < 0) {
    set_error("mythos_langevin_advance: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->resident) {
    set_error("mythos_langevin_advance: no resident state (call mythos_langevin_load first; a run that ended in a numeric "
              "error drops its state)");
    return MYTHOS_ERR_NOT_READY;
  }
  if (int rc = md_ready(s, "mythos_langevin_advance")) return rc;
  return md_advance(s, n_steps, save_every, traj_center, traj_quat, e_trace, (hipStream_t)stream);
}

int mythos_langevin_store(mythos_sim_t* s, void* center, void* quat, void* p_lin, void* p_ang, mythos_stream_t stream) {
  if (!s || !center || !quat || !p_lin || !p_ang) {
    set_error("mythos_langevin_store: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->resident) {
    set_error("mythos_langevin_store: no resident state");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  return md_store(s, center, quat, p_lin, p_ang, (hipStream_t)stream);
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

int mythos_langevin_last_recoveries(const mythos_sim_t* s, int* recoveries) {
  if (!s || !recoveries) {
    set_error("mythos_langevin_last_recoveries: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  *recoveries = s->last_recoveries;
  return MYTHOS_OK;
}

int mythos_langevin_last_rebuilds(const mythos_sim_t* s, int* scheduled) {
  if (!s || !scheduled) {
    set_error("mythos_langevin_last_rebuilds: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  *scheduled = s->last_rebuilds;
  return MYTHOS_OK;
}

int mythos_langevin_set_option(mythos_sim_t* s, int option, int64_t value) {
  if (!s || option != MYTHOS_LANGEVIN_UNFUSED || (value != 0 && value != 1)) {
    set_error("mythos_langevin_set_option: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (value == 1 && s->sys->model != 4) {
    set_error("mythos_langevin_set_option: the unfused path exists for oxNA systems (model 4) only");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->want_unfused = value == 1;
  return MYTHOS_OK;
}

int mythos_langevin_set_timing(mythos_sim_t* s, int samples) {
  if (!s || samples < 0) {
    set_error("mythos_langevin_set_timing: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->timing_samples = std::min(samples, (int)mythos_sim::kMaxSamples);
  return MYTHOS_OK;
}

}  // extern "C"
