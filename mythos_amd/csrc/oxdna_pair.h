// Pair interactions of oxDNA1 / oxDNA2 with analytic gradients, written from the point of view
// of ONE nucleotide ("self") interacting with a neighbour ("other").
//
// The reference evaluates every ordered pair (p, q) once and lets jax.grad scatter the
// derivatives to both members.  On the GPU each nucleotide gathers its own row instead, so a
// pair function only needs the OWNER's derivatives:
//   dU/dc_self and dU/da1, dU/da2, dU/da3 of self (axes treated as independent vectors),
// from which the caller forms the lab torque  -sum_k a_k x dU/da_k  (MD) or the quaternion
// gradient jax.grad returns (energy API).  Which reference role self plays (p = op_i / nn_i or
// q = op_j / nn_j) only changes signs and which parameter block a cosine uses; those choices
// are per-lane selects, so no pair is ever re-ordered in registers.  Optionally every
// evaluation also emits dU/dparam for the flat parameter vector.
//
// Geometry and role conventions follow the reference term by term:
//   bonded  (p = nn_i, q = nn_j):  dna1/fene.py:37-56, dna1/bonded_excluded_volume.py:84-114,
//                                  dna1/stacking.py:192-289, dna2/stacking.py:19-39
//   unbonded (p = op_i, q = op_j): dna1/unbonded_excluded_volume.py:105-146,
//       dna1/hydrogen_bonding.py:232-306, dna1/cross_stacking.py:192-266,
//       dna1/coaxial_stacking.py:181-260, dna2/coaxial_stacking.py:138-201, dna2/debye.py:82-110
// All displacement vectors below are d = site_other - site_self (minimum image on the centres).
#pragma once
#include "oxdna_math.h"

namespace mythos {
inline namespace MYTHOS_MATH_NS {  // (see oxdna_math.h)

enum OxTerm : int {
  T_FENE = 0,
  T_BEXC = 1,
  T_STCK = 2,
  T_NEXC = 3,
  T_HB = 4,
  T_CRST = 5,
  T_CXST = 6,
  T_DH = 7,
  T_COUNT = 8
};

template <typename R>
struct Nuc {
  V3<R> c, a1, a2, a3;
  int seq;
  int is_end;
  int idx;  // nucleotide index: read only where the parameter accessor carries a probabilistic sequence
  int rna;  // oxNA (MODEL 4): the nucleotide is RNA (topology nt_type); not read by the other instantiations
};

// axes from an (un-normalised) quaternion, mythos/energy/utils.py:18-36
template <typename R>
__device__ __forceinline__ void quat_axes(R q0, R q1, R q2, R q3, V3<R>& a1, V3<R>& a2, V3<R>& a3) {
  a1 = {q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, R(2) * (q1 * q2 + q0 * q3), R(2) * (q1 * q3 - q0 * q2)};
  a2 = {R(2) * (q1 * q2 - q0 * q3), q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, R(2) * (q2 * q3 + q0 * q1)};
  a3 = {R(2) * (q1 * q3 + q0 * q2), R(2) * (q2 * q3 - q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3};
}

// 4x4 sequence-weight lookup: an indexed read of an LDS-staged vector, or a select chain over scalar
// registers for the kernel-argument copy (a lane-varying index into kernel arguments would force the
// whole block into scratch memory)
template <typename R, class PT>
__device__ __forceinline__ auto weight_lookup(const PT& P, int base, int k) {
  if constexpr (PT::indexed || kLeanMath<R>) {
    // (fp32 stepping kernels, MYTHOS_LEAN_MATH: ONE vector load at a lane-varying address; the select chain below is
    // what the compiler makes of it anyway - fifteen compares and selects that compute the offset of that same load)
    return P[base + k];
  } else {
    auto w = P[base];
#pragma unroll
    for (int t = 1; t < 16; ++t) w = (k == t) ? P[base + t] : w;
    return w;
  }
}

// What the sequence-weight look-ups read of a nucleotide, BY VALUE.  The callers choose p and q by the role of the owner;
// choosing between two `const Nuc&` made the select one of ADDRESSES, and where the compiler could not see through it
// (the oxNA instantiations: three inlined copies of every pair function) both nucleotides - 112 B each in fp64 - were
// kept in scratch memory and read back through a lane-varying address: that was the 232 ... 720 B of scratch of every
// oxNA kernel through round 3, not register pressure.
struct SeqId {
  int seq, idx;
};
template <typename R>
__device__ __forceinline__ SeqId seq_id(bool first, const Nuc<R>& a, const Nuc<R>& b) {
  return SeqId{first ? a.seq : b.seq, first ? a.idx : b.idx};
}

// Weight of the ordered pair (p, q) of a sequence-dependent term: table[seq_p][seq_q] (dna1/stacking.py:287,
// dna1/hydrogen_bonding.py:333), or - under a probabilistic sequence - its expectation (energy/utils.py:45-132):
// sum_ab P(p = a, q = b) table[a][b] with P the product of the marginals for nucleotides of different units and
// sum_t P(type t) [a, b = members of type t] for the two members of one constrained base pair.
template <typename R, class PT>
__device__ __forceinline__ R seq_weight(const PT& P, int base, int term_bit, const SeqId p, const SeqId q) {
  if constexpr (PT::has_pseq) {
    if (P.ps.marg != nullptr && (P.ps.terms & term_bit) != 0) {
      const int up = P.ps.unit[p.idx], uq = P.ps.unit[q.idx];
      if (up >= 0 && uq >= 0 && (up >> 1) == (uq >> 1)) {
        // types AT, TA, GC, CG = (A,T), (T,A), (G,C), (C,G) for (member 0, member 1); A, C, G, T = 0 .. 3
        const R* b = P.ps.bp + 4 * (up >> 1);
        const bool fwd = (up & 1) == 0;  // p is member 0
        return b[0] * (fwd ? P[base + 3] : P[base + 12]) + b[1] * (fwd ? P[base + 12] : P[base + 3]) +
               b[2] * (fwd ? P[base + 9] : P[base + 6]) + b[3] * (fwd ? P[base + 6] : P[base + 9]);
      }
      const R* mp = P.ps.marg + 4 * p.idx;
      const R* mq = P.ps.marg + 4 * q.idx;
      const R q0 = mq[0], q1 = mq[1], q2 = mq[2], q3 = mq[3];
      R w = R(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        w += mp[a] * (q0 * P[base + 4 * a] + q1 * P[base + 4 * a + 1] + q2 * P[base + 4 * a + 2] + q3 * P[base + 4 * a + 3]);
      return w;
    }
  }
  return weight_lookup<R>(P, base, p.seq * 4 + q.seq);
}

// dU/dtable[a][b] of the same weight: scale * P(p = a, q = b)
template <typename R, class PG, class PT>
__device__ __forceinline__ void seq_weight_pgrad(const PT& P, int base, int term_bit, const SeqId p, const SeqId q,
                                                 R scale, PG& pg) {
  if constexpr (PT::has_pseq) {
    if (P.ps.marg != nullptr && (P.ps.terms & term_bit) != 0) {
      const int up = P.ps.unit[p.idx], uq = P.ps.unit[q.idx];
      if (up >= 0 && uq >= 0 && (up >> 1) == (uq >> 1)) {
        const R* b = P.ps.bp + 4 * (up >> 1);
        const bool fwd = (up & 1) == 0;
        pg.add(base + (fwd ? 3 : 12), scale * b[0]);
        pg.add(base + (fwd ? 12 : 3), scale * b[1]);
        pg.add(base + (fwd ? 9 : 6), scale * b[2]);
        pg.add(base + (fwd ? 6 : 9), scale * b[3]);
        if constexpr (PT::has_pseq_grad) {  // dU/d(type probabilities): the weight is linear in them; half per visit of the pair
          double* g = P.ps.gbp + 4 * (up >> 1);
          const double h = 0.5 * double(scale);
          atomicAdd(g + 0, h * double(fwd ? P[base + 3] : P[base + 12]));
          atomicAdd(g + 1, h * double(fwd ? P[base + 12] : P[base + 3]));
          atomicAdd(g + 2, h * double(fwd ? P[base + 9] : P[base + 6]));
          atomicAdd(g + 3, h * double(fwd ? P[base + 6] : P[base + 9]));
        }
        return;
      }
      const R* mp = P.ps.marg + 4 * p.idx;
      const R* mq = P.ps.marg + 4 * q.idx;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) pg.add(base + 4 * a + b, scale * mp[a] * mq[b]);
      if constexpr (PT::has_pseq_grad) {  // dU/d(marginals): w = sum_ab mp[a] mq[b] T[a][b] is bilinear; half per visit of the pair
        const double h = 0.5 * double(scale);
        double* gp = P.ps.gmarg + 4 * p.idx;
        double* gq = P.ps.gmarg + 4 * q.idx;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          R sp = R(0), sq = R(0);
#pragma unroll
          for (int b = 0; b < 4; ++b) sp += mq[b] * P[base + 4 * a + b], sq += mp[b] * P[base + 4 * b + a];
          atomicAdd(gp + a, h * double(sp));
          atomicAdd(gq + a, h * double(sq));
        }
      }
      return;
    }
  }
  pg.add(base + p.seq * 4 + q.seq, scale);
}

template <typename R>
struct SelfGrad {  // gradient of U with respect to the owner's centre and axes
  V3<R> dc, g1, g2, g3;
};

// Which axis the second coefficient of a site offset multiplies: a2 for oxDNA (dna2/nucleotide.py:44-49: the
// groove backbone site; the stacking sites of oxRNA2 too), a3 for the oxRNA2 backbone site (rna2/nucleotide.py:56).
template <int MODEL>
constexpr int back_axis() {
  return MODEL == 3 ? 3 : 2;
}

// gd = dV/dd with d = site_other - site_self and site_self = c + al*a1 + be*a_AX
template <int AX = 2, typename R>
__device__ __forceinline__ void acc_self_site(SelfGrad<R>& sg, V3<R> gd, R al, R be) {
  axpy(sg.dc, R(-1), gd);
  axpy(sg.g1, -al, gd);
  if constexpr (AX == 3) axpy(sg.g3, -be, gd); else axpy(sg.g2, -be, gd);
}

template <int AX = 2, typename R>
__device__ __forceinline__ V3<R> site_disp(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o, R als, R bes, R alo, R beo) {
  V3<R> d = dco;
  axpy(d, alo, o.a1);
  axpy(d, beo, AX == 3 ? o.a3 : o.a2);
  axpy(d, -als, s.a1);
  axpy(d, -bes, AX == 3 ? s.a3 : s.a2);
  return d;
}

// cosine c = sg * (u . n) with n = d / r:  dc/dd = (sg*u - c*n) / r
template <typename R>
__device__ __forceinline__ void acc_dir(V3<R>& gd, R coef, R sg, R c, V3<R> u, V3<R> n, R inv_r) {
  axpy(gd, coef * sg * inv_r, u);
  axpy(gd, -coef * c * inv_r, n);
}

// one radial f3 site pair: energy, self gradient, parameter partials
template <typename R, bool GRAD, class PG, int AX = 2, class PT>
__device__ __forceinline__ R f3_site_pair(const PT& P, int ie, const F3P<R>& fp, V3<R> d, R als, R bes,
                                          R tw, SelfGrad<R>& sg, PG& pg) {
  const R r = m_sqrt(dot(d, d));
  const FD<R> v = f3_eval(r, P[ie], fp);
  if constexpr (GRAD)
    if (v.d != R(0)) acc_self_site<AX>(sg, (tw * v.d / r) * d, als, bes);
  f3_pgrad(r, P[ie], ie, fp, tw, pg);
  return v.f;
}

// ------------------------------------------------------------------------------------------------
// Where the sites of the two nucleotides of an UNBONDED pair are.  UniGeo: both nucleotides have the geometry of the
// one model (what every instantiation but oxNA uses; it compiles to the site algebra written out by hand).
// HybGeo: oxNA's DNA-RNA pairs (mythos/energy/na1/nucleotide.py:12-78 + the (nucleotide.dna, nucleotide.rna) argument
// pairs of the hybrid branches, e.g. na1/hydrogen_bonding.py:336-348): each nucleotide has the sites of its OWN type -
// oxDNA2 offsets with the backbone on a1 / a2, or oxRNA2 offsets with the backbone on a1 / a3 - chosen per lane.
// ------------------------------------------------------------------------------------------------
template <typename R, int MODEL>
struct UniGeo {
  static constexpr int BX = back_axis<MODEL>();
  R st, ba, k1, k2;
  template <class PT>
  __device__ __forceinline__ explicit UniGeo(const PT& P)
      : st(P[GEO_STACK]), ba(P[GEO_BASE]), k1(P[GEO_BACK_A1]), k2((MODEL >= 2) ? P[GEO_BACK_A2] : R(0)) {}
  __device__ __forceinline__ V3<R> back_back(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return site_disp<BX>(dco, s, o, k1, k2, k1, k2); }
  __device__ __forceinline__ V3<R> back_base(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return site_disp<BX>(dco, s, o, k1, k2, ba, R(0)); }
  __device__ __forceinline__ V3<R> base_back(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return site_disp<BX>(dco, s, o, ba, R(0), k1, k2); }
  __device__ __forceinline__ V3<R> base_base(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return site_disp(dco, s, o, ba, R(0), ba, R(0)); }
  __device__ __forceinline__ V3<R> stack_stack(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return site_disp(dco, s, o, st, R(0), st, R(0)); }
  __device__ __forceinline__ void acc_back(SelfGrad<R>& sg, V3<R> gd) const { acc_self_site<BX>(sg, gd, k1, k2); }
  __device__ __forceinline__ void acc_base(SelfGrad<R>& sg, V3<R> gd) const { acc_self_site(sg, gd, ba, R(0)); }
  __device__ __forceinline__ void acc_stack(SelfGrad<R>& sg, V3<R> gd) const { acc_self_site(sg, gd, st, R(0)); }
};

template <typename R>
struct HybGeo {
  R st_s, ba_s, k1_s, k2_s, st_o, ba_o, k1_o, k2_o;
  bool s3, o3;  // the second backbone coefficient multiplies a3 (RNA) instead of a2 (DNA)
  template <class PT>
  __device__ __forceinline__ HybGeo(const PT& Pd, const PT& Pr, bool s_rna, bool o_rna) : s3(s_rna), o3(o_rna) {
    const R st_d = Pd[GEO_STACK], ba_d = Pd[GEO_BASE], k1_d = Pd[GEO_BACK_A1], k2_d = Pd[GEO_BACK_A2];
    const R st_r = Pr[GEO_STACK], ba_r = Pr[GEO_BASE], k1_r = Pr[GEO_BACK_A1], k2_r = Pr[GEO_BACK_A2];
    st_s = s_rna ? st_r : st_d, ba_s = s_rna ? ba_r : ba_d, k1_s = s_rna ? k1_r : k1_d, k2_s = s_rna ? k2_r : k2_d;
    st_o = o_rna ? st_r : st_d, ba_o = o_rna ? ba_r : ba_d, k1_o = o_rna ? k1_r : k1_d, k2_o = o_rna ? k2_r : k2_d;
  }
  __device__ __forceinline__ V3<R> disp(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o, R als, R bes, R alo, R beo) const {
    // (the second axis is selected BY VALUE, component by component: `o3 ? o.a3 : o.a2` is a select between two ADDRESSES
    // inside the nucleotides, and that kept both of them - 112 B each in fp64 - in scratch memory in every oxNA kernel)
    const V3<R> ob{o3 ? o.a3.x : o.a2.x, o3 ? o.a3.y : o.a2.y, o3 ? o.a3.z : o.a2.z};
    const V3<R> sb{s3 ? s.a3.x : s.a2.x, s3 ? s.a3.y : s.a2.y, s3 ? s.a3.z : s.a2.z};
    V3<R> d = dco;
    axpy(d, alo, o.a1);
    axpy(d, beo, ob);
    axpy(d, -als, s.a1);
    axpy(d, -bes, sb);
    return d;
  }
  __device__ __forceinline__ V3<R> back_back(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return disp(dco, s, o, k1_s, k2_s, k1_o, k2_o); }
  __device__ __forceinline__ V3<R> back_base(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return disp(dco, s, o, k1_s, k2_s, ba_o, R(0)); }
  __device__ __forceinline__ V3<R> base_back(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return disp(dco, s, o, ba_s, R(0), k1_o, k2_o); }
  __device__ __forceinline__ V3<R> base_base(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return disp(dco, s, o, ba_s, R(0), ba_o, R(0)); }
  __device__ __forceinline__ V3<R> stack_stack(V3<R> dco, const Nuc<R>& s, const Nuc<R>& o) const { return disp(dco, s, o, st_s, R(0), st_o, R(0)); }
  __device__ __forceinline__ void acc_back(SelfGrad<R>& sg, V3<R> gd) const {
    axpy(sg.dc, R(-1), gd);
    axpy(sg.g1, -k1_s, gd);
    axpy(sg.g2, s3 ? R(0) : -k2_s, gd);
    axpy(sg.g3, s3 ? -k2_s : R(0), gd);
  }
  __device__ __forceinline__ void acc_base(SelfGrad<R>& sg, V3<R> gd) const { acc_self_site(sg, gd, ba_s, R(0)); }
  __device__ __forceinline__ void acc_stack(SelfGrad<R>& sg, V3<R> gd) const { acc_self_site(sg, gd, st_s, R(0)); }
};

// one radial f3 site pair of an unbonded pair; SITE: which of self's sites d starts from (0 backbone, 1 base)
template <typename R, bool GRAD, class PG, int SITE, class PT, class GEO>
__device__ __forceinline__ R f3_site_pair_geo(const PT& P, int ie, const F3P<R>& fp, V3<R> d, R tw, SelfGrad<R>& sg, PG& pg,
                                              const GEO& geo) {
  const R r = m_sqrt(dot(d, d));
  const FD<R> v = f3_eval(r, P[ie], fp);
  if constexpr (GRAD)
    if (v.d != R(0)) {
      if constexpr (SITE == 0) geo.acc_back(sg, (tw * v.d / r) * d); else geo.acc_base(sg, (tw * v.d / r) * d);
    }
  f3_pgrad(r, P[ie], ie, fp, tw, pg);
  return v.f;
}

// oxNA: the three parameter vectors of a hybrid system, one after the other in device memory (DNA-DNA pairs: the oxDNA2
// vector; RNA-RNA: the oxRNA2 vector; DNA-RNA: the hybrid numbers in the oxDNA1 forms), and a parameter-gradient sink
// that files a partial under its vector.
template <class PT>
struct Na1Params {
  static constexpr bool indexed = false;
  static constexpr bool has_pseq = PT::has_pseq;
  static constexpr bool has_pseq_grad = false;
  PT dna, rna, drh;
};
template <class PG>
struct OffsetPG {
  static constexpr bool on = PG::on;
  PG& pg;
  int base;
  template <typename R>
  __device__ __forceinline__ void add(int idx, R v) const {
    pg.add(base + idx, v);
  }
};

// ------------------------------------------------------------------------------------------------
// bonded pair: FENE + bonded excluded volume + stacking.   role_p: self is nn_i of the bond.
// ------------------------------------------------------------------------------------------------
template <typename R, int MODEL, bool GRAD, class PG, class PT>
__device__ __forceinline__ void bonded_pair(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco,
                                            bool role_p, R wgt, R* __restrict__ e, SelfGrad<R>& sg, PG& pg) {
  if constexpr (MODEL == 4) {
    // oxNA (na1/fene.py:89-106, bonded_excluded_volume.py:96-116, stacking.py:193-217): a bond between two RNA
    // nucleotides is an oxRNA2 bond; any other bond is evaluated as an oxDNA2 bond on the oxDNA2 sites
    if (s.rna && o.rna) {
      OffsetPG<PG> opg{pg, (int)OXP_COUNT};
      bonded_pair<R, 3, GRAD, OffsetPG<PG>>(P.rna, s, o, dco, role_p, wgt, e, sg, opg);
    } else {
      OffsetPG<PG> opg{pg, 0};
      bonded_pair<R, 2, GRAD, OffsetPG<PG>>(P.dna, s, o, dco, role_p, wgt, e, sg, opg);
    }
  } else {
  constexpr int BX = back_axis<MODEL>();
  const R g_st = P[GEO_STACK], g_ba = P[GEO_BASE], g_k1 = P[GEO_BACK_A1];
  const R g_k2 = (MODEL >= 2) ? P[GEO_BACK_A2] : R(0);
  const R g_d1 = (MODEL == 2) ? P[GEO_BACK_DNA1] : g_k1;
  (void)g_st, (void)g_d1;

  // ---- FENE on the backbone sites (symmetric)
  {
    const V3<R> d = site_disp<BX>(dco, s, o, g_k1, g_k2, g_k1, g_k2);
    const R r = m_sqrt(dot(d, d));
    const FD<R> v = fene_eval(r, P);
    e[T_FENE] += wgt * v.f;
    if constexpr (GRAD) acc_self_site<BX>(sg, (P[TW_FENE] * v.d / r) * d, g_k1, g_k2);
    fene_pgrad(r, P, v.d, P[TW_FENE], pg);
  }
  // ---- bonded excluded volume: base-base, back_p-base_q, base_p-back_q
  {
    R eb = f3_site_pair<R, GRAD, PG>(P, BEXC_EPS, f3_params<R>(P, BEXC_BASE_RSTAR),
                                     site_disp(dco, s, o, g_ba, R(0), g_ba, R(0)), g_ba, R(0), P[TW_BEXC], sg, pg);
    // self backbone - other base: "back_p - base_q" if self is p, else "base_p - back_q"
    eb += f3_site_pair<R, GRAD, PG, BX>(P, BEXC_EPS, f3_params_sel<R>(P, role_p, BEXC_BACK_BASE_RSTAR, BEXC_BASE_BACK_RSTAR),
                                        site_disp<BX>(dco, s, o, g_k1, g_k2, g_ba, R(0)), g_k1, g_k2, P[TW_BEXC], sg, pg);
    eb += f3_site_pair<R, GRAD, PG>(P, BEXC_EPS, f3_params_sel<R>(P, role_p, BEXC_BASE_BACK_RSTAR, BEXC_BACK_BASE_RSTAR),
                                    site_disp<BX>(dco, s, o, g_ba, R(0), g_k1, g_k2), g_ba, R(0), P[TW_BEXC], sg, pg);
    e[T_BEXC] += wgt * eb;
  }
  // ---- oxRNA2 stacking (rna2/stacking.py:186-292, rna2/interactions.py:15-135): the 5' stacking site of nn_i against the
  //      3' site of nn_j, no theta4, and theta9 = acos(-p3_q . dr_b / r_b), theta10 = acos(-p5_p . dr_b / r_b) with the
  //      body-fixed vectors p3 / p5 and dr_b = back_p - back_q on the oxRNA2 backbone sites.  With d = other - self,
  //      dr = sgm * d (sgm = -1 if self is p): self's own vector gives theta10 if self is p, theta9 if self is q.
  if constexpr (MODEL == 3) {
    const R s5a = P[GEO_STACK5_A1], s5b = P[GEO_STACK5_A2], s3a = P[GEO_STACK3_A1], s3b = P[GEO_STACK3_A2];
    const R als = role_p ? s5a : s3a, bes = role_p ? s5b : s3b, alo = role_p ? s3a : s5a, beo = role_p ? s3b : s5b;
    const V3<R> ds = site_disp(dco, s, o, als, bes, alo, beo);
    const R rs = m_sqrt(dot(ds, ds));
    const FD<R> F1 = f1_eval(rs, P, STCK_RLOW);
    if (F1.f == R(0) && F1.d == R(0)) return;
    const R sgm = role_p ? R(-1) : R(1);
    const R irs = R(1) / rs;
    const V3<R> ns = irs * ds;
    const R cs = sgm * dot(s.a3, ns);  // theta6 if p, theta5 if q
    const R co = sgm * dot(o.a3, ns);  // theta5 if p, theta6 if q
    FD<R> ts = acos_clamped(cs), to = acos_clamped(co);
    ts.f = R(kPi) - ts.f;
    ts.d = -ts.d;
    to.f = R(kPi) - to.f;
    to.d = -to.d;
    const F4P<R> ps = f4_params_sel<R>(P, role_p, STCK_TH6_T0, STCK_TH5_T0);
    const F4P<R> po = f4_params_sel<R>(P, role_p, STCK_TH5_T0, STCK_TH6_T0);
    const FD<R> As = f4_eval(ts.f, ps);
    if (As.f == R(0)) return;
    const FD<R> Ao = f4_eval(to.f, po);
    if (Ao.f == R(0)) return;
    const V3<R> db = site_disp<3>(dco, s, o, g_k1, g_k2, g_k1, g_k2);
    const R irb = m_rsqrt(dot(db, db));
    const V3<R> nb = irb * db;
    // self's vector: p5 if self is p, p3 if self is q; other's the opposite
    const R vsx = role_p ? P[GEO_P5_X] : P[GEO_P3_X], vsy = role_p ? P[GEO_P5_Y] : P[GEO_P3_Y], vsz = role_p ? P[GEO_P5_Z] : P[GEO_P3_Z];
    const R vox = role_p ? P[GEO_P3_X] : P[GEO_P5_X], voy = role_p ? P[GEO_P3_Y] : P[GEO_P5_Y], voz = role_p ? P[GEO_P3_Z] : P[GEO_P5_Z];
    V3<R> vs = vsx * s.a1, vo = vox * o.a1;
    axpy(vs, vsy, s.a2), axpy(vs, vsz, s.a3);
    axpy(vo, voy, o.a2), axpy(vo, voz, o.a3);
    const R cvs = -sgm * dot(vs, nb);  // cos(theta10) if p, cos(theta9) if q
    const R cvo = -sgm * dot(vo, nb);  // cos(theta9) if p, cos(theta10) if q
    const FD<R> tvs = acos_clamped(cvs), tvo = acos_clamped(cvo);
    const F4P<R> pvs = f4_params_sel<R>(P, role_p, STCK_TH10_T0, STCK_TH9_T0);
    const F4P<R> pvo = f4_params_sel<R>(P, role_p, STCK_TH9_T0, STCK_TH10_T0);
    const FD<R> Avs = f4_eval(tvs.f, pvs);
    if (Avs.f == R(0)) return;
    const FD<R> Avo = f4_eval(tvo.f, pvo);
    if (Avo.f == R(0)) return;
    const R xs = sgm * dot(s.a2, nb);  // -cos(phi1) if p, -cos(phi2) if q
    const R xo = sgm * dot(o.a2, nb);
    const F5P<R> qs = f5_params_sel<R>(P, role_p, STCK_PHI1_XS, STCK_PHI2_XS);
    const F5P<R> qo = f5_params_sel<R>(P, role_p, STCK_PHI2_XS, STCK_PHI1_XS);
    const FD<R> Bs = f5_eval(xs, qs);
    if (Bs.f == R(0)) return;
    const FD<R> Bo = f5_eval(xo, qo);
    if (Bo.f == R(0)) return;
    const R wseq = seq_weight<R>(P, STCK_EPS_00, 1, seq_id(role_p, s, o), seq_id(role_p, o, s));
    const R ang = As.f * Ao.f * Avs.f * Avo.f;
    const R phi = Bs.f * Bo.f;
    const R v = F1.f * ang * phi;
    e[T_STCK] += wgt * wseq * v;
    const R w = wseq * P[TW_STCK];
    if constexpr (PG::on) {
      seq_weight_pgrad<R>(P, STCK_EPS_00, 1, seq_id(role_p, s, o), seq_id(role_p, o, s), P[TW_STCK] * v, pg);
      f1_pgrad(rs, P, STCK_RLOW, w * ang * phi, pg);
      f4_pgrad(ts.f, ps, w * F1.f * Ao.f * Avs.f * Avo.f * phi, pg);
      f4_pgrad(to.f, po, w * F1.f * As.f * Avs.f * Avo.f * phi, pg);
      f4_pgrad(tvs.f, pvs, w * F1.f * As.f * Ao.f * Avo.f * phi, pg);
      f4_pgrad(tvo.f, pvo, w * F1.f * As.f * Ao.f * Avs.f * phi, pg);
      f5_pgrad(xs, qs, w * F1.f * ang * Bo.f, pg);
      f5_pgrad(xo, qo, w * F1.f * ang * Bs.f, pg);
    }
    if constexpr (GRAD) {
      const R wf = w * F1.f;
      V3<R> gds{R(0), R(0), R(0)}, gdb{R(0), R(0), R(0)};
      axpy(gds, w * F1.d * ang * phi, ns);
      const R ks = wf * Ao.f * Avs.f * Avo.f * phi * As.d * ts.d;
      axpy(sg.g3, ks * sgm, ns);
      acc_dir(gds, ks, sgm, cs, s.a3, ns, irs);
      const R ko = wf * As.f * Avs.f * Avo.f * phi * Ao.d * to.d;
      acc_dir(gds, ko, sgm, co, o.a3, ns, irs);
      // cvs = -sgm (vs . nb), vs = vsx a1 + vsy a2 + vsz a3 of self
      const R kvs = wf * As.f * Ao.f * Avo.f * phi * Avs.d * tvs.d;
      axpy(sg.g1, -kvs * sgm * vsx, nb);
      axpy(sg.g2, -kvs * sgm * vsy, nb);
      axpy(sg.g3, -kvs * sgm * vsz, nb);
      acc_dir(gdb, kvs, -sgm, cvs, vs, nb, irb);
      const R kvo = wf * As.f * Ao.f * Avs.f * phi * Avo.d * tvo.d;
      acc_dir(gdb, kvo, -sgm, cvo, vo, nb, irb);
      const R kxs = wf * ang * Bo.f * Bs.d;
      axpy(sg.g2, kxs * sgm, nb);
      acc_dir(gdb, kxs, sgm, xs, s.a2, nb, irb);
      const R kxo = wf * ang * Bs.f * Bo.d;
      acc_dir(gdb, kxo, sgm, xo, o.a2, nb, irb);
      acc_self_site(sg, gds, als, bes);
      acc_self_site<3>(sg, gdb, g_k1, g_k2);
    }
  } else
  // ---- stacking.  Reference: dr = site_p - site_q, theta5 = pi - acos(dr.a3_q / r),
  //      theta6 = pi - acos(a3_p.dr / r), -cos(phi1) = a2_p.dr_b / r_b, -cos(phi2) = a2_q.dr_b / r_b.
  //      With d = other - self:  dr = -d if self is p, +d if self is q  ->  sign sgm.
  {
    const V3<R> ds = site_disp(dco, s, o, g_st, R(0), g_st, R(0));
    const R rs = m_sqrt(dot(ds, ds));
    const FD<R> F1 = f1_eval(rs, P, STCK_RLOW);
    if (F1.f == R(0) && F1.d == R(0)) return;
    const R sgm = role_p ? R(-1) : R(1);
    const R irs = R(1) / rs;
    const V3<R> ns = irs * ds;
    const R c4 = dot(s.a3, o.a3);
    const R cs = sgm * dot(s.a3, ns);  // cosine built from self's normal : theta6 if p, theta5 if q
    const R co = sgm * dot(o.a3, ns);  // cosine built from other's normal: theta5 if p, theta6 if q
    const FD<R> t4 = acos_clamped(c4);
    FD<R> ts = acos_clamped(cs), to = acos_clamped(co);
    ts.f = R(kPi) - ts.f;
    ts.d = -ts.d;
    to.f = R(kPi) - to.f;
    to.d = -to.d;
    const F4P<R> p4 = f4_params<R>(P, STCK_TH4_T0);
    const F4P<R> ps = f4_params_sel<R>(P, role_p, STCK_TH6_T0, STCK_TH5_T0);
    const F4P<R> po = f4_params_sel<R>(P, role_p, STCK_TH5_T0, STCK_TH6_T0);
    const FD<R> A4 = f4_eval(t4.f, p4);
    if (A4.f == R(0)) return;
    const FD<R> As = f4_eval(ts.f, ps);
    if (As.f == R(0)) return;
    const FD<R> Ao = f4_eval(to.f, po);
    if (Ao.f == R(0)) return;
    const V3<R> db = site_disp(dco, s, o, g_d1, R(0), g_d1, R(0));
    const R irb = m_rsqrt(dot(db, db));
    const V3<R> nb = irb * db;
    const R xs = sgm * dot(s.a2, nb);  // -cos(phi1) if p, -cos(phi2) if q
    const R xo = sgm * dot(o.a2, nb);
    const F5P<R> qs = f5_params_sel<R>(P, role_p, STCK_PHI1_XS, STCK_PHI2_XS);
    const F5P<R> qo = f5_params_sel<R>(P, role_p, STCK_PHI2_XS, STCK_PHI1_XS);
    const FD<R> Bs = f5_eval(xs, qs);
    if (Bs.f == R(0)) return;
    const FD<R> Bo = f5_eval(xo, qo);
    if (Bo.f == R(0)) return;
    const R wseq = seq_weight<R>(P, STCK_EPS_00, 1, seq_id(role_p, s, o), seq_id(role_p, o, s));
    const R ang = A4.f * As.f * Ao.f;
    const R phi = Bs.f * Bo.f;
    const R v = F1.f * ang * phi;
    e[T_STCK] += wgt * wseq * v;
    const R w = wseq * P[TW_STCK];  // gradients carry the term weight
    if constexpr (PG::on) {
      seq_weight_pgrad<R>(P, STCK_EPS_00, 1, seq_id(role_p, s, o), seq_id(role_p, o, s), P[TW_STCK] * v, pg);
      f1_pgrad(rs, P, STCK_RLOW, w * ang * phi, pg);
      f4_pgrad(t4.f, p4, w * F1.f * As.f * Ao.f * phi, pg);
      f4_pgrad(ts.f, ps, w * F1.f * A4.f * Ao.f * phi, pg);
      f4_pgrad(to.f, po, w * F1.f * A4.f * As.f * phi, pg);
      f5_pgrad(xs, qs, w * F1.f * ang * Bo.f, pg);
      f5_pgrad(xo, qo, w * F1.f * ang * Bs.f, pg);
    }
    if constexpr (GRAD) {
      const R wf = w * F1.f;
      V3<R> gds{R(0), R(0), R(0)}, gdb{R(0), R(0), R(0)};
      axpy(gds, w * F1.d * ang * phi, ns);
      axpy(sg.g3, wf * As.f * Ao.f * phi * A4.d * t4.d, o.a3);
      const R ks = wf * A4.f * Ao.f * phi * As.d * ts.d;
      axpy(sg.g3, ks * sgm, ns);
      acc_dir(gds, ks, sgm, cs, s.a3, ns, irs);
      const R ko = wf * A4.f * As.f * phi * Ao.d * to.d;
      acc_dir(gds, ko, sgm, co, o.a3, ns, irs);
      const R kxs = wf * ang * Bo.f * Bs.d;
      axpy(sg.g2, kxs * sgm, nb);
      acc_dir(gdb, kxs, sgm, xs, s.a2, nb, irb);
      const R kxo = wf * ang * Bs.f * Bo.d;
      acc_dir(gdb, kxo, sgm, xo, o.a2, nb, irb);
      acc_self_site(sg, gds, g_st, R(0));
      acc_self_site(sg, gdb, g_d1, R(0));
    }
  }
  }  // MODEL != 4
}

// radial supports of the angular unbonded terms: H-bond / cross-stacking act on the base-base
// distance, coaxial stacking on the stack-stack distance
template <typename R, class PT>
__device__ __forceinline__ bool hb_crst_support(const PT& P, R r) {
  return (P[HYDR_RCLOW] < r && r < P[HYDR_RCHIGH]) || (P[CRST_RCLOW] < r && r < P[CRST_RCHIGH]);
}
template <typename R, class PT>
__device__ __forceinline__ bool cxst_support(const PT& P, R r) {
  return P[CXST_RCLOW] < r && r < P[CXST_RCHIGH];
}

// ------------------------------------------------------------------------------------------------
// unbonded pair, radial part: excluded volume (4 x f3) + Debye.  Returns true if any of the
// angular terms (H-bond, cross-stacking, coaxial stacking) can be non-zero for this pair.
// role_p: self is op_i of the ordered pair.
// ------------------------------------------------------------------------------------------------
template <typename R, int MODEL, bool GRAD, class PG, class PT, class GEO>
__device__ __forceinline__ bool unbonded_radial_geo(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco, bool role_p, R wgt,
                                                    R* __restrict__ e, SelfGrad<R>& sg, PG& pg, const GEO& geo) {
  // ---- backbone-backbone: excluded volume and (dna2) Debye-Hueckel share the distance
  {
    const V3<R> d = geo.back_back(dco, s, o);
    const R r = m_sqrt(dot(d, d));
    const F3P<R> fp = f3_params<R>(P, NEXC_BACKBONE_RSTAR);
    const FD<R> v = f3_eval(r, P[NEXC_EPS], fp);
    e[T_NEXC] += wgt * v.f;
    R dVdr = P[TW_NEXC] * v.d;
    f3_pgrad(r, P[NEXC_EPS], NEXC_EPS, fp, P[TW_NEXC], pg);
    if constexpr (MODEL >= 2) {  // oxDNA2, oxRNA2 and the oxNA hybrid pairs carry the Debye-Hueckel term
      const FD<R> dh = debye_eval(r, P);
      R mult = R(1);
      if (P[DH_HALF_CHARGED_ENDS] != R(0)) mult = (s.is_end ? R(0.5) : R(1)) * (o.is_end ? R(0.5) : R(1));
      e[T_DH] += wgt * mult * dh.f;
      dVdr += P[TW_DH] * mult * dh.d;
      debye_pgrad(r, P, P[TW_DH] * mult, pg);
    }
    if constexpr (GRAD)
      if (dVdr != R(0)) geo.acc_back(sg, (dVdr / r) * d);
  }
  // ---- self backbone - other base ("back_p - base_q" if self is p) and self base - other backbone
  {
    R en = f3_site_pair_geo<R, GRAD, PG, 0>(P, NEXC_EPS, f3_params_sel<R>(P, role_p, NEXC_BACK_BASE_RSTAR, NEXC_BASE_BACK_RSTAR),
                                            geo.back_base(dco, s, o), P[TW_NEXC], sg, pg, geo);
    en += f3_site_pair_geo<R, GRAD, PG, 1>(P, NEXC_EPS, f3_params_sel<R>(P, role_p, NEXC_BASE_BACK_RSTAR, NEXC_BACK_BASE_RSTAR),
                                           geo.base_back(dco, s, o), P[TW_NEXC], sg, pg, geo);
    e[T_NEXC] += wgt * en;
  }
  // ---- base-base excluded volume
  bool angular;
  {
    const V3<R> d = geo.base_base(dco, s, o);
    const R r = m_sqrt(dot(d, d));
    const F3P<R> fp = f3_params<R>(P, NEXC_BASE_RSTAR);
    const FD<R> v = f3_eval(r, P[NEXC_EPS], fp);
    e[T_NEXC] += wgt * v.f;
    f3_pgrad(r, P[NEXC_EPS], NEXC_EPS, fp, P[TW_NEXC], pg);
    if constexpr (GRAD)
      if (v.d != R(0)) geo.acc_base(sg, (P[TW_NEXC] * v.d / r) * d);
    angular = hb_crst_support(P, r);
  }
  {
    const V3<R> d = geo.stack_stack(dco, s, o);
    angular = angular || cxst_support(P, m_sqrt(dot(d, d)));
  }
  return angular;
}

// Which parameter vector and functional form an unbonded oxNA pair takes (na1/unbonded_excluded_volume.py:140-174 and
// the other unbonded terms): both RNA -> oxRNA2, both DNA -> oxDNA2, one of each -> the hybrid numbers in the oxDNA1
// forms (cross-stacking with theta4, coaxial stacking with f5 of cos phi3 / phi4) plus Debye-Hueckel: "MODEL 4" below.
// (an if / else-if / else chain of plain calls whose result lands in RESULT: through round 3 this was three `return FN(...)`
// inside a lambda that captured e, sg, pg by reference, and the closure kept them addressable)
#define MYTHOS_NA1_UNBONDED(RESULT, FN, ...)                                                                 \
  if (s.rna && o.rna) {                                                                                      \
    OffsetPG<PG> opg{pg, (int)OXP_COUNT};                                                                    \
    const UniGeo<R, 3> geo(P.rna);                                                                           \
    RESULT = FN<R, 3, GRAD, OffsetPG<PG> __VA_ARGS__>(P.rna, s, o, dco, role_p, wgt, e, sg, opg, geo);       \
  } else if (!s.rna && !o.rna) {                                                                             \
    OffsetPG<PG> opg{pg, 0};                                                                                 \
    const UniGeo<R, 2> geo(P.dna);                                                                           \
    RESULT = FN<R, 2, GRAD, OffsetPG<PG> __VA_ARGS__>(P.dna, s, o, dco, role_p, wgt, e, sg, opg, geo);       \
  } else {                                                                                                   \
    OffsetPG<PG> opg{pg, 2 * (int)OXP_COUNT};                                                                \
    const HybGeo<R> geo(P.dna, P.rna, s.rna != 0, o.rna != 0);                                               \
    RESULT = FN<R, 4, GRAD, OffsetPG<PG> __VA_ARGS__>(P.drh, s, o, dco, role_p, wgt, e, sg, opg, geo);       \
  }

template <typename R, int MODEL, bool GRAD, class PG, class PT>
__device__ __forceinline__ bool unbonded_radial(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco,
                                                bool role_p, R wgt, R* __restrict__ e, SelfGrad<R>& sg, PG& pg) {
  if constexpr (MODEL == 4) {
    bool angular;
    MYTHOS_NA1_UNBONDED(angular, unbonded_radial_geo)
    return angular;
  } else {
    const UniGeo<R, MODEL> geo(P);
    return unbonded_radial_geo<R, MODEL, GRAD, PG>(P, s, o, dco, role_p, wgt, e, sg, pg, geo);
  }
}

// ------------------------------------------------------------------------------------------------
// unbonded pair, angular part: H-bond + cross-stacking (base-base vector) and coaxial stacking.
//      Reference: dr = base_q - base_p (= +d if self is p, -d if q);
//      theta1 = acos(-a1p.a1q) theta2 = acos(-a1q.n) theta3 = acos(a1p.n)
//      theta4 = acos(a3p.a3q)  theta7 = acos(-a3q.n) theta8 = pi - acos(a3p.n)
//      In self/other form the cosines are  a1s.n, -a1o.n, a3s.n, -a3o.n  for either role; the role
//      decides which angle (and parameter block) each of them is.
// ------------------------------------------------------------------------------------------------
// TERMS selects which terms are compiled in: 1 = H-bond, 2 = cross-stacking, 4 = coaxial stacking.  The MD
// kernel instantiates the three separately so that each wavefront of its angular pass runs one term on a
// homogeneous work list; the energy path uses all three (7).
template <typename R, int MODEL, bool GRAD, class PG, int TERMS = 7, class PT, class GEO>
__device__ __forceinline__ bool unbonded_angular_geo(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco, bool role_p, R wgt,
                                                     R* __restrict__ e, SelfGrad<R>& sg, PG& pg, const GEO& geo) {
  // dna1-style coaxial term, f5(cos phi3) f5(cos phi4): oxDNA1, oxRNA2 (rna2/tests/test_integration.py:258-287) and the
  // oxNA hybrid pairs (na1/coaxial_stacking.py:272-284)
  constexpr bool kCoaxF5 = MODEL != 2;
  if constexpr ((TERMS & 3) != 0) {
    const V3<R> d = geo.base_base(dco, s, o);
    const R r = m_sqrt(dot(d, d));
    V3<R> gd{R(0), R(0), R(0)};
    bool any = false;
    const R whb = seq_weight<R>(P, HYDR_EPS_00, 2, seq_id(role_p, s, o), seq_id(role_p, o, s));
    const FD<R> F1 = ((TERMS & 1) && (whb != R(0) || PG::on)) ? f1_eval(r, P, HYDR_RLOW) : FD<R>{R(0), R(0)};
    const FD<R> F2 = (TERMS & 2) ? f2_eval(r, P, CRST_RLOW) : FD<R>{R(0), R(0)};
    const bool hb_on = (F1.f != R(0) || F1.d != R(0));
    const bool cr_on = (F2.f != R(0) || F2.d != R(0));
    if (hb_on || cr_on) {
      const R ir = R(1) / r;
      const V3<R> n = ir * d;
      const R c1 = -dot(s.a1, o.a1);
      const R c4 = dot(s.a3, o.a3);
      const R cs1 = dot(s.a1, n);   // theta3 if p, theta2 if q
      const R co1 = -dot(o.a1, n);  // theta2 if p, theta3 if q
      const R cs3 = dot(s.a3, n);   // theta8 (pi - acos) if p, theta7 if q
      const R co3 = -dot(o.a3, n);  // theta7 if p, theta8 (pi - acos) if q
      const FD<R> t1 = acos_clamped(c1), t4 = acos_clamped(c4);
      const FD<R> ts1 = acos_clamped(cs1), to1 = acos_clamped(co1);
      FD<R> ts3 = acos_clamped(cs3), to3 = acos_clamped(co3);
      if (role_p) {
        ts3.f = R(kPi) - ts3.f;
        ts3.d = -ts3.d;
      } else {
        to3.f = R(kPi) - to3.f;
        to3.d = -to3.d;
      }
      R k1 = R(0), k4 = R(0), ks1 = R(0), ko1 = R(0), ks3 = R(0), ko3 = R(0), krad = R(0);
      if (hb_on) {
        const F4P<R> p1 = f4_params<R>(P, HYDR_TH1_T0), p4 = f4_params<R>(P, HYDR_TH4_T0);
        const F4P<R> ps1 = f4_params_sel<R>(P, role_p, HYDR_TH3_T0, HYDR_TH2_T0);
        const F4P<R> po1 = f4_params_sel<R>(P, role_p, HYDR_TH2_T0, HYDR_TH3_T0);
        const F4P<R> ps3 = f4_params_sel<R>(P, role_p, HYDR_TH8_T0, HYDR_TH7_T0);
        const F4P<R> po3 = f4_params_sel<R>(P, role_p, HYDR_TH7_T0, HYDR_TH8_T0);
        const FD<R> A1 = f4_eval(t1.f, p1), A4 = f4_eval(t4.f, p4);
        const FD<R> As1 = f4_eval(ts1.f, ps1), Ao1 = f4_eval(to1.f, po1);
        const FD<R> As3 = f4_eval(ts3.f, ps3), Ao3 = f4_eval(to3.f, po3);
        const R ang = A1.f * A4.f * As1.f * Ao1.f * As3.f * Ao3.f;
        if (ang != R(0)) {
          const R vhb = F1.f * ang;
          e[T_HB] += wgt * whb * vhb;
          const R o1 = A4.f * As1.f * Ao1.f * As3.f * Ao3.f, o4 = A1.f * As1.f * Ao1.f * As3.f * Ao3.f;
          const R os1 = A1.f * A4.f * Ao1.f * As3.f * Ao3.f, oo1 = A1.f * A4.f * As1.f * As3.f * Ao3.f;
          const R os3 = A1.f * A4.f * As1.f * Ao1.f * Ao3.f, oo3 = A1.f * A4.f * As1.f * Ao1.f * As3.f;
          const R whg = whb * P[TW_HB];
          const R wf = whg * F1.f;
          if constexpr (PG::on) {
            seq_weight_pgrad<R>(P, HYDR_EPS_00, 2, seq_id(role_p, s, o), seq_id(role_p, o, s), P[TW_HB] * vhb, pg);
            f1_pgrad(r, P, HYDR_RLOW, whg * ang, pg);
            f4_pgrad(t1.f, p1, wf * o1, pg);
            f4_pgrad(t4.f, p4, wf * o4, pg);
            f4_pgrad(ts1.f, ps1, wf * os1, pg);
            f4_pgrad(to1.f, po1, wf * oo1, pg);
            f4_pgrad(ts3.f, ps3, wf * os3, pg);
            f4_pgrad(to3.f, po3, wf * oo3, pg);
          }
          if constexpr (GRAD) {
            krad += whg * F1.d * ang;
            k1 += wf * o1 * A1.d * t1.d;
            k4 += wf * o4 * A4.d * t4.d;
            ks1 += wf * os1 * As1.d * ts1.d;
            ko1 += wf * oo1 * Ao1.d * to1.d;
            ks3 += wf * os3 * As3.d * ts3.d;
            ko3 += wf * oo3 * Ao3.d * to3.d;
          }
        }
      }
      if (cr_on) {
        const F4P<R> p1 = f4_params<R>(P, CRST_TH1_T0);
        const F4P<R> ps1 = f4_params_sel<R>(P, role_p, CRST_TH3_T0, CRST_TH2_T0);
        const F4P<R> po1 = f4_params_sel<R>(P, role_p, CRST_TH2_T0, CRST_TH3_T0);
        const FD<R> A1 = f4_eval(t1.f, p1), As1 = f4_eval(ts1.f, ps1), Ao1 = f4_eval(to1.f, po1);
        const R a123 = A1.f * As1.f * Ao1.f;
        if (a123 != R(0)) {
          const F4P<R> p4 = f4_params<R>(P, CRST_TH4_T0);
          const F4P<R> ps3 = f4_params_sel<R>(P, role_p, CRST_TH8_T0, CRST_TH7_T0);
          const F4P<R> po3 = f4_params_sel<R>(P, role_p, CRST_TH7_T0, CRST_TH8_T0);
          // (oxRNA2's cross-stacking has no theta4 factor, rna2/interactions.py:238-256: H4 = 1)
          const FD<R> A4a = (MODEL == 3) ? FD<R>{R(1), R(0)} : f4_eval(t4.f, p4);
          const FD<R> A4b = (MODEL == 3) ? FD<R>{R(0), R(0)} : f4_eval(R(kPi) - t4.f, p4);
          const FD<R> Sa = f4_eval(ts3.f, ps3), Sb = f4_eval(R(kPi) - ts3.f, ps3);
          const FD<R> Oa = f4_eval(to3.f, po3), Ob = f4_eval(R(kPi) - to3.f, po3);
          const R H4 = A4a.f + A4b.f, Hs = Sa.f + Sb.f, Ho = Oa.f + Ob.f;
          const R ang = a123 * H4 * Hs * Ho;
          if (ang != R(0)) {
            e[T_CRST] += wgt * F2.f * ang;
            const FD<R> F2w{P[TW_CRST] * F2.f, P[TW_CRST] * F2.d};
            const R hh = H4 * Hs * Ho;
            const R o1 = As1.f * Ao1.f * hh, os1 = A1.f * Ao1.f * hh, oo1 = A1.f * As1.f * hh;
            const R o4 = a123 * Hs * Ho, os3 = a123 * H4 * Ho, oo3 = a123 * H4 * Hs;
            if constexpr (PG::on) {
              f2_pgrad(r, P, CRST_RLOW, P[TW_CRST] * ang, pg);
              f4_pgrad(t1.f, p1, F2w.f * o1, pg);
              f4_pgrad(ts1.f, ps1, F2w.f * os1, pg);
              f4_pgrad(to1.f, po1, F2w.f * oo1, pg);
              if constexpr (MODEL != 3) {
                f4_pgrad(t4.f, p4, F2w.f * o4, pg);
                f4_pgrad(R(kPi) - t4.f, p4, F2w.f * o4, pg);
              }
              f4_pgrad(ts3.f, ps3, F2w.f * os3, pg);
              f4_pgrad(R(kPi) - ts3.f, ps3, F2w.f * os3, pg);
              f4_pgrad(to3.f, po3, F2w.f * oo3, pg);
              f4_pgrad(R(kPi) - to3.f, po3, F2w.f * oo3, pg);
            }
            if constexpr (GRAD) {
              krad += F2w.d * ang;
              k1 += F2w.f * o1 * A1.d * t1.d;
              ks1 += F2w.f * os1 * As1.d * ts1.d;
              ko1 += F2w.f * oo1 * Ao1.d * to1.d;
              k4 += F2w.f * o4 * (A4a.d - A4b.d) * t4.d;
              ks3 += F2w.f * os3 * (Sa.d - Sb.d) * ts3.d;
              ko3 += F2w.f * oo3 * (Oa.d - Ob.d) * to3.d;
            }
          }
        }
      }
      if constexpr (GRAD) {
        if (krad != R(0) || k1 != R(0) || k4 != R(0) || ks1 != R(0) || ko1 != R(0) || ks3 != R(0) || ko3 != R(0)) {
          any = true;
          axpy(gd, krad, n);
          axpy(sg.g1, -k1, o.a1);
          axpy(sg.g3, k4, o.a3);
          axpy(sg.g1, ks1, n);
          acc_dir(gd, ks1, R(1), cs1, s.a1, n, ir);
          acc_dir(gd, ko1, R(-1), co1, o.a1, n, ir);
          axpy(sg.g3, ks3, n);
          acc_dir(gd, ks3, R(1), cs3, s.a3, n, ir);
          acc_dir(gd, ko3, R(-1), co3, o.a3, n, ir);
        }
      }
    }
    if constexpr (GRAD)
      if (any) geo.acc_base(sg, gd);
  }
  // ---- coaxial stacking on the stacking sites.  Reference: dr = stack_q - stack_p,
  //      theta5 = acos(a3p.n), theta6 = acos(-a3q.n); self/other cosines a3s.n and -a3o.n.
  if constexpr ((TERMS & 4) != 0) {
    const V3<R> d = geo.stack_stack(dco, s, o);
    const R r = m_sqrt(dot(d, d));
    const FD<R> F2 = f2_eval(r, P, CXST_RLOW);
    if (F2.f == R(0) && F2.d == R(0)) return true;
    const R ir = R(1) / r;
    const V3<R> n = ir * d;
    const R c4 = dot(s.a3, o.a3);
    const R c1 = -dot(s.a1, o.a1);
    const R cs = dot(s.a3, n);   // theta5 if p, theta6 if q
    const R co = -dot(o.a3, n);  // theta6 if p, theta5 if q
    const FD<R> t4 = acos_clamped(c4), t1 = acos_clamped(c1), ts = acos_clamped(cs), to = acos_clamped(co);
    const F4P<R> p4 = f4_params<R>(P, CXST_TH4_T0), p1 = f4_params<R>(P, CXST_TH1_T0);
    const F4P<R> ps = f4_params_sel<R>(P, role_p, CXST_TH5_T0, CXST_TH6_T0);
    const F4P<R> po = f4_params_sel<R>(P, role_p, CXST_TH6_T0, CXST_TH5_T0);
    const FD<R> A4 = f4_eval(t4.f, p4);
    if (A4.f == R(0)) return true;
    const FD<R> Sa = f4_eval(ts.f, ps), Sb = f4_eval(R(kPi) - ts.f, ps);
    const R Hs = Sa.f + Sb.f;
    if (Hs == R(0)) return true;
    const FD<R> Oa = f4_eval(to.f, po), Ob = f4_eval(R(kPi) - to.f, po);
    const R Ho = Oa.f + Ob.f;
    if (Ho == R(0)) return true;
    const FD<R> A1a = f4_eval(t1.f, p1);
    FD<R> A1b;
    R dH1;
    if constexpr (kCoaxF5) {
      A1b = f4_eval(R(2 * kPi) - t1.f, p1);
      dH1 = A1a.d - A1b.d;
    } else {
      A1b = f6_eval(t1.f, P, CXST_F6_A);
      dH1 = A1a.d + A1b.d;
    }
    const R H1 = A1a.f + A1b.f;
    if (H1 == R(0)) return true;
    R phi = R(1);
    FD<R> Bs{R(1), R(0)}, Bo{R(1), R(0)};
    R xs = R(0), xo = R(0), irb = R(0);
    V3<R> nb{R(0), R(0), R(0)};
    F5P<R> qs{}, qo{};
    if constexpr (kCoaxF5) {
      // cos(phi3) = n.(nb x a1q), cos(phi4) = n.(nb x a1p): both vectors flip with the role, the
      // triple product does not.
      const V3<R> db = geo.back_back(dco, s, o);
      irb = m_rsqrt(dot(db, db));
      nb = irb * db;
      xs = dot(n, cross(nb, s.a1));
      xo = dot(n, cross(nb, o.a1));
      qs = f5_params_sel<R>(P, role_p, CXST_PHI4_XS, CXST_PHI3_XS);
      qo = f5_params_sel<R>(P, role_p, CXST_PHI3_XS, CXST_PHI4_XS);
      Bs = f5_eval(xs, qs);
      Bo = f5_eval(xo, qo);
      phi = Bs.f * Bo.f;
      if (phi == R(0)) return true;
    }
    const R ang = A4.f * H1 * Hs * Ho;
    e[T_CXST] += wgt * F2.f * ang * phi;
    const FD<R> F2w{P[TW_CXST] * F2.f, P[TW_CXST] * F2.d};
    if constexpr (PG::on) {
      f2_pgrad(r, P, CXST_RLOW, P[TW_CXST] * ang * phi, pg);
      const R s4 = F2w.f * H1 * Hs * Ho * phi, s1 = F2w.f * A4.f * Hs * Ho * phi;
      const R ss = F2w.f * A4.f * H1 * Ho * phi, so = F2w.f * A4.f * H1 * Hs * phi;
      f4_pgrad(t4.f, p4, s4, pg);
      f4_pgrad(t1.f, p1, s1, pg);
      if constexpr (kCoaxF5) {
        f4_pgrad(R(2 * kPi) - t1.f, p1, s1, pg);
        f5_pgrad(xs, qs, F2w.f * ang * Bo.f, pg);
        f5_pgrad(xo, qo, F2w.f * ang * Bs.f, pg);
      } else {
        f6_pgrad(t1.f, P, CXST_F6_A, s1, pg);
      }
      f4_pgrad(ts.f, ps, ss, pg);
      f4_pgrad(R(kPi) - ts.f, ps, ss, pg);
      f4_pgrad(to.f, po, so, pg);
      f4_pgrad(R(kPi) - to.f, po, so, pg);
    }
    if constexpr (GRAD) {
      V3<R> gd{R(0), R(0), R(0)};
      axpy(gd, F2w.d * ang * phi, n);
      axpy(sg.g3, F2w.f * H1 * Hs * Ho * phi * A4.d * t4.d, o.a3);
      axpy(sg.g1, -(F2w.f * A4.f * Hs * Ho * phi * dH1 * t1.d), o.a1);
      const R ks = F2w.f * A4.f * H1 * Ho * phi * (Sa.d - Sb.d) * ts.d;
      axpy(sg.g3, ks, n);
      acc_dir(gd, ks, R(1), cs, s.a3, n, ir);
      const R ko = F2w.f * A4.f * H1 * Hs * phi * (Oa.d - Ob.d) * to.d;
      acc_dir(gd, ko, R(-1), co, o.a3, n, ir);
      if constexpr (kCoaxF5) {
        // t = n.(nb x a):  dt/da = n x nb,  dt/dn = nb x a,  dt/dnb = a x n
        V3<R> gn{R(0), R(0), R(0)}, gnb{R(0), R(0), R(0)};
        const R kxs = F2w.f * ang * Bo.f * Bs.d, kxo = F2w.f * ang * Bs.f * Bo.d;
        axpy(sg.g1, kxs, cross(n, nb));
        axpy(gn, kxs, cross(nb, s.a1));
        axpy(gnb, kxs, cross(s.a1, n));
        axpy(gn, kxo, cross(nb, o.a1));
        axpy(gnb, kxo, cross(o.a1, n));
        axpy(gd, ir, gn);
        axpy(gd, -ir * dot(gn, n), n);
        V3<R> gdb{R(0), R(0), R(0)};
        axpy(gdb, irb, gnb);
        axpy(gdb, -irb * dot(gnb, nb), nb);
        geo.acc_back(sg, gdb);
      }
      geo.acc_stack(sg, gd);
    }
  }
  return true;
}

template <typename R, int MODEL, bool GRAD, class PG, int TERMS = 7, class PT>
__device__ __forceinline__ void unbonded_angular(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco,
                                                 bool role_p, R wgt, R* __restrict__ e, SelfGrad<R>& sg, PG& pg) {
  if constexpr (MODEL == 4) {
    bool done;
    MYTHOS_NA1_UNBONDED(done, unbonded_angular_geo, , TERMS)
    (void)done;
  } else {
    const UniGeo<R, MODEL> geo(P);
    unbonded_angular_geo<R, MODEL, GRAD, PG, TERMS>(P, s, o, dco, role_p, wgt, e, sg, pg, geo);
  }
}

// whole unbonded pair (energy API path)
template <typename R, int MODEL, bool GRAD, class PG, class PT>
__device__ __forceinline__ void unbonded_pair(const PT& P, const Nuc<R>& s, const Nuc<R>& o, V3<R> dco,
                                              bool role_p, R wgt, R* __restrict__ e, SelfGrad<R>& sg, PG& pg) {
  if (unbonded_radial<R, MODEL, GRAD, PG>(P, s, o, dco, role_p, wgt, e, sg, pg))
    unbonded_angular<R, MODEL, GRAD, PG>(P, s, o, dco, role_p, wgt, e, sg, pg);
}

}  // inline namespace MYTHOS_MATH_NS
}  // namespace mythos
