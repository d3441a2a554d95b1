// Scalar building blocks of the oxDNA force field for gfx950: value AND analytic derivative
// of f1..f6 / FENE / Debye, plus their partials with respect to every parameter.
//
// What each function computes follows the reference's jnp graph
//   mythos/energy/potentials.py:11-70, mythos/energy/dna1/base_functions.py:13-129,
//   mythos/energy/dna2/base_functions.py:13-17, mythos/energy/dna1/interactions.py:16-41,
//   mythos/energy/dna2/interactions.py:15-28, mythos/utils/math.py:68-81
// with the same strict inequalities on every branch.  The reference obtains derivatives by
// jax.grad; here they are written out so one pass yields energy, force and dU/dtheta.
#pragma once
#include <hip/hip_runtime.h>

namespace mythos {

// ------------------------------------------------------------------ parameter index enum
enum OxParamIndex : int {
#define OXP(name) name,
#include "oxdna_param_list.inc"
#undef OXP
  OXP_COUNT
};
// The entries only oxRNA2 reads are at the end of the list: the oxDNA instantiations accumulate, reduce and write
// parameter partials for the entries before them only (the rest of a dU/dparams row is zero).  With all 261 the
// per-workgroup reduction of the partials took two passes of the 256 threads where 241 take one (+2 - 4 % per call).
constexpr int OXP_COUNT_DNA = GEO_STACK3_A1;
template <int MODEL>
constexpr int oxp_used() {
  return MODEL == 4 ? 3 * (int)OXP_COUNT : (MODEL == 3 ? (int)OXP_COUNT : OXP_COUNT_DNA);  // oxNA: three vectors
}

// The flat parameter vector on the host (and, for the Debye / cut-off helpers, as a plain array).
template <typename R>
struct OxParams {
  static constexpr bool indexed = false;
  static constexpr bool has_pseq = false;
  static constexpr bool has_pseq_grad = false;
  R v[OXP_COUNT];
  __host__ __device__ __forceinline__ R operator[](int i) const { return v[i]; }
};

// How the kernels read it: the vector lives in device memory and is read through the constant address space,
// so every access with a compile-time index is one s_load at the point of use (scalar cache) and no parameter
// has to stay live in a register between uses.  Two alternatives were measured and dropped: passed by value in
// the kernel-argument segment the compiler loads all used entries in the entry block and spills them (SGPR ->
// VGPR -> scratch, 0.5-1.4 KB per lane); staged in LDS the loads are hoisted and held in VGPRs (256 registers,
// one wavefront per SIMD).  `indexed` = true would mean "lane-varying indices are cheap" (an LDS or global
// gather); the role-dependent parameter blocks are instead chosen by per-value selects (the *_sel helpers).
//
// Probabilistic sequence as the energy kernel sees it (mythos/energy/utils.py:45-132, restated in terms of
// per-nucleotide marginals by mythos_amd/input/sequence_constraints.py kernel_tables): two nucleotides of different
// units are independent, the two members of one constrained base pair are tied through its type.
template <typename R>
struct PseqView {
  const R* marg = nullptr;   // [n][4] probability of A, C, G, T per nucleotide; null: discrete sequence
  const int* unit = nullptr; // [n] 2 * base pair + position inside it, or -1 for an unpaired nucleotide
  const R* bp = nullptr;     // [n_bp][4] probability of the types AT, TA, GC, CG per constrained base pair
  int terms = 0;             // bit 0: stacking weights are expectations, bit 1: hydrogen-bonding weights are
  // dU/d(distribution), accumulated with the parameter partials when asked for (mythos_oxdna_energy_dpseq): one frame's
  // [n][4] and [bp_rows][4] doubles, fp64 atomics (the order of the additions is not fixed)
  double* gmarg = nullptr;
  double* gbp = nullptr;
  int bp_rows = 0;
};
template <typename R, bool PSEQ = false, bool PSEQ_GRAD = false>
struct ConstParams {
  static constexpr bool indexed = false;
  static constexpr bool has_pseq = PSEQ;  // the MD kernel compiles without the branch (and its registers)
  static constexpr bool has_pseq_grad = PSEQ_GRAD;  // dU/d(distribution) accumulated too (PseqView::gmarg / gbp are set)
  typedef const R __attribute__((address_space(4))) * cptr;
  cptr p;
  PseqView<R> ps;
  __device__ __forceinline__ explicit ConstParams(const R* g) : p((cptr)g) {}
  __device__ __forceinline__ ConstParams(const R* g, const PseqView<R>& v) : p((cptr)g), ps(v) {}
  __device__ __forceinline__ R operator[](int i) const { return p[i]; }
};

// ------------------------------------------------------------------ tiny vector algebra
template <typename R>
struct V3 {
  R x, y, z;
};
template <typename R>
__host__ __device__ __forceinline__ V3<R> mk(R x, R y, R z) { return V3<R>{x, y, z}; }
template <typename R>
__host__ __device__ __forceinline__ V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R>
__host__ __device__ __forceinline__ V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R>
__host__ __device__ __forceinline__ V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R>
__host__ __device__ __forceinline__ V3<R> operator*(R s, V3<R> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename R>
__host__ __device__ __forceinline__ R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R>
__host__ __device__ __forceinline__ V3<R> cross(V3<R> a, V3<R> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename R>
__host__ __device__ __forceinline__ void axpy(V3<R>& y, R a, V3<R> x) {
  y.x += a * x.x;
  y.y += a * x.y;
  y.z += a * x.z;
}

__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
// fp64 square root (round 4): the compiler's expansion of sqrt(double) is v_rsq_f64 and two and a half coupled Newton steps
// - the ten instructions below - wrapped in a scaling of inputs under 2^-767 (compare, select, two ldexp) and a class test that
// puts 0, inf and NaN back (20 instructions in all, 44 square roots in md_step_kernel<double, 2>).  The arguments here are
// squared lengths and 1 - c^2: never that small, never infinite; zero is possible and is put back by one select.  Same
// arithmetic, same bits for every argument in range.  (The host build of these templates - oracle/cpu_port - takes libm's.)
__device__ __forceinline__ double m_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
#else
  return sqrt(x);
#endif
}
__device__ __forceinline__ float m_rsqrt(float x) { return rsqrtf(x); }
// (the library's rsqrt - v_rsq_f64 + refinement - measured 1.4 % SLOWER per fp64 MD step than this form)
__device__ __forceinline__ double m_rsqrt(double x) { return 1.0 / m_sqrt(x); }
__device__ __forceinline__ float m_exp(float x) { return __expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_log(float x) { return __logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
// acos on [-1, 1], Abramowitz & Stegun 4.4.46 (|error| <= 2e-8, below fp32 resolution of the
// angle): sqrt(1 - |x|) * P7(|x|), reflected for x < 0.  A third of the instructions of acosf.
__device__ __forceinline__ float m_acos(float x) {
  const float ax = fabsf(x);
  float p = -0.0012624911f;
  p = fmaf(p, ax, 0.0066700901f);
  p = fmaf(p, ax, -0.0170881256f);
  p = fmaf(p, ax, 0.0308918810f);
  p = fmaf(p, ax, -0.0501743046f);
  p = fmaf(p, ax, 0.0889789874f);
  p = fmaf(p, ax, -0.2145988016f);
  p = fmaf(p, ax, 1.5707963050f);
  const float r = sqrtf(1.0f - ax) * p;
  return x < 0.0f ? 3.14159265358979f - r : r;
}
__device__ __forceinline__ double m_acos(double x) { return acos(x); }
__device__ __forceinline__ float m_rint(float x) { return rintf(x); }
__device__ __forceinline__ double m_rint(double x) { return rint(x); }

// MYTHOS_LEAN_MATH (set by langevin_core.inc for the stepping kernels: 1 = the fp32 instantiations, 2 = both precisions):
// branch-free forms of the piecewise modulation functions - all pieces computed, then selected - instead of divergent
// three-way branches (every piece is executed anyway as soon as the lanes of a wavefront disagree, plus the exec-mask
// traffic around it).  Equal to the branchy forms except AT the breakpoints, where the reference's strict inequalities
// give 0 (a removable discontinuity) and these forms give the continuous value.
#ifndef MYTHOS_LEAN_MATH
#define MYTHOS_LEAN_MATH 0
#endif
// Everything below - and oxdna_pair.h, oxdna_gather.h on top of it - lives in an inline namespace named after the switch:
// the stepping kernels' translation units (MYTHOS_LEAN_MATH = 1) and the others (0) compile different bodies for f1, f2,
// f4, f5 and the weight look-up, and under one name that is a violation of the one-definition rule (harmless while device
// code is inlined per translation unit, undefined as soon as anything links device code across units).  Distinct
// entities have no such problem; unqualified names resolve as before.
#if MYTHOS_LEAN_MATH == 0
#define MYTHOS_MATH_NS exact_math
#elif MYTHOS_LEAN_MATH == 1
#define MYTHOS_MATH_NS lean_math_f32
#else
#define MYTHOS_MATH_NS lean_math_all
#endif
inline namespace MYTHOS_MATH_NS {

template <typename R>
struct FD {  // value and derivative with respect to the argument
  R f, d;
};

constexpr double kPi = 3.14159265358979323846;

template <typename R>
constexpr bool kLeanMath = (MYTHOS_LEAN_MATH == 2) || ((MYTHOS_LEAN_MATH == 1) && sizeof(R) == 4);

// No-op / real sinks for parameter partials.  A sink receives (index, dV/dparam).
struct NoPG {
  static constexpr bool on = false;
  template <typename R>
  __device__ __forceinline__ void add(int, R) const {}
};

// acos(clamp(c)) with d(theta)/dc; the clamp has zero slope where it clips (math.py:78-81)
template <typename R>
__device__ __forceinline__ FD<R> acos_clamped(R c) {
  const bool clipped = (c >= R(1)) || (c <= R(-1));
  const R cc = c >= R(1) ? R(1) : (c <= R(-1) ? R(-1) : c);
  FD<R> o;
  o.f = m_acos(cc);
  o.d = clipped ? R(0) : -m_rsqrt(R(1) - cc * cc);
  return o;
}

// ------------------------------------------------------------------ f1 (base_functions.py:13-37), eps = 1
// parameter block layout: RLOW,RHIGH,RCLOW,RCHIGH,A,R0,RC,BLOW,BHIGH,SHIFT
template <typename R, class PT>
__device__ __forceinline__ FD<R> f1_eval(R r, const PT& P, int b) {
  const R rlow = P[b + 0], rhigh = P[b + 1], rclow = P[b + 2], rchigh = P[b + 3];
  if constexpr (kLeanMath<R>) {  // all three pieces, then selects (see f4_eval_lean; equal to the branchy form off the breakpoints)
    const R a = P[b + 4];
    const R e = m_exp(-a * (r - P[b + 5]));
    const bool lo = r < rlow;
    const R t = (lo ? rclow : rchigh) - r, bq = lo ? P[b + 7] : P[b + 8];
    const bool core = rlow < r && r < rhigh, tail = (rclow < r && lo) || (rhigh < r && r < rchigh);
    FD<R> o;
    o.f = core ? (R(1) - e) * (R(1) - e) - P[b + 9] : (tail ? bq * t * t : R(0));
    o.d = core ? R(2) * a * e * (R(1) - e) : (tail ? R(-2) * bq * t : R(0));
    return o;
  }
  FD<R> o{R(0), R(0)};
  if (rlow < r && r < rhigh) {
    const R a = P[b + 4];
    const R e = m_exp(-a * (r - P[b + 5]));
    o.f = (R(1) - e) * (R(1) - e) - P[b + 9];
    o.d = R(2) * a * e * (R(1) - e);
  } else if (rclow < r && r < rlow) {
    const R t = rclow - r;
    o.f = P[b + 7] * t * t;
    o.d = R(-2) * P[b + 7] * t;
  } else if (rhigh < r && r < rchigh) {
    const R t = rchigh - r;
    o.f = P[b + 8] * t * t;
    o.d = R(-2) * P[b + 8] * t;
  }
  return o;
}
template <typename R, class PG, class PT>
__device__ __forceinline__ void f1_pgrad(R r, const PT& P, int b, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  const R rlow = P[b + 0], rhigh = P[b + 1], rclow = P[b + 2], rchigh = P[b + 3];
  if (rlow < r && r < rhigh) {
    const R a = P[b + 4], x = r - P[b + 5];
    const R e = m_exp(-a * x);
    pg.add(b + 4, scale * R(2) * (R(1) - e) * e * x);
    pg.add(b + 5, scale * R(-2) * a * e * (R(1) - e));
    pg.add(b + 9, -scale);
  } else if (rclow < r && r < rlow) {
    const R t = rclow - r;
    pg.add(b + 7, scale * t * t);
    pg.add(b + 2, scale * R(2) * P[b + 7] * t);
  } else if (rhigh < r && r < rchigh) {
    const R t = rchigh - r;
    pg.add(b + 8, scale * t * t);
    pg.add(b + 3, scale * R(2) * P[b + 8] * t);
  }
}

// ------------------------------------------------------------------ f2 (base_functions.py:40-63)
// layout: RLOW,RHIGH,RCLOW,RCHIGH,K,R0,RC,BLOW,BHIGH,SHIFT ; SHIFT = (rc-r0)^2/2
template <typename R, class PT>
__device__ __forceinline__ FD<R> f2_eval(R r, const PT& P, int b) {
  const R rlow = P[b + 0], rhigh = P[b + 1], rclow = P[b + 2], rchigh = P[b + 3], k = P[b + 4];
  if constexpr (kLeanMath<R>) {
    const R x = r - P[b + 5];
    const bool lo = r < rlow;
    const R t = (lo ? rclow : rchigh) - r, kb = k * (lo ? P[b + 7] : P[b + 8]);
    const bool core = rlow < r && r < rhigh, tail = (rclow < r && lo) || (rhigh < r && r < rchigh);
    FD<R> o;
    o.f = core ? k * (R(0.5) * x * x - P[b + 9]) : (tail ? kb * t * t : R(0));
    o.d = core ? k * x : (tail ? R(-2) * kb * t : R(0));
    return o;
  }
  FD<R> o{R(0), R(0)};
  if (rlow < r && r < rhigh) {
    const R x = r - P[b + 5];
    o.f = k * (R(0.5) * x * x - P[b + 9]);
    o.d = k * x;
  } else if (rclow < r && r < rlow) {
    const R t = rclow - r;
    o.f = k * P[b + 7] * t * t;
    o.d = R(-2) * k * P[b + 7] * t;
  } else if (rhigh < r && r < rchigh) {
    const R t = rchigh - r;
    o.f = k * P[b + 8] * t * t;
    o.d = R(-2) * k * P[b + 8] * t;
  }
  return o;
}
template <typename R, class PG, class PT>
__device__ __forceinline__ void f2_pgrad(R r, const PT& P, int b, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  const R rlow = P[b + 0], rhigh = P[b + 1], rclow = P[b + 2], rchigh = P[b + 3], k = P[b + 4];
  if (rlow < r && r < rhigh) {
    const R x = r - P[b + 5];
    pg.add(b + 4, scale * (R(0.5) * x * x - P[b + 9]));
    pg.add(b + 5, -scale * k * x);
    pg.add(b + 9, -scale * k);
  } else if (rclow < r && r < rlow) {
    const R t = rclow - r;
    pg.add(b + 4, scale * P[b + 7] * t * t);
    pg.add(b + 7, scale * k * t * t);
    pg.add(b + 2, scale * R(2) * k * P[b + 7] * t);
  } else if (rhigh < r && r < rchigh) {
    const R t = rchigh - r;
    pg.add(b + 4, scale * P[b + 8] * t * t);
    pg.add(b + 8, scale * k * t * t);
    pg.add(b + 3, scale * R(2) * k * P[b + 8] * t);
  }
}

// ------------------------------------------------------------------ f3 (base_functions.py:66-79)
// per-site-pair block layout: RSTAR,SIGMA,B,RC ; eps is shared by the term
template <typename R>
struct F3P {
  R rstar, sigma, b, rc;
  int base;  // index of RSTAR in the flat vector (for parameter partials)
};
template <typename R, class PT>
__device__ __forceinline__ F3P<R> f3_params(const PT& P, int b) {
  return {P[b + 0], P[b + 1], P[b + 2], P[b + 3], b};
}
// role-dependent choice between two blocks (compile-time indices -> one v_cndmask per value)
// role-dependent choice between two blocks: an indexed read for the LDS copy, one v_cndmask per value
// for the kernel-argument copy (a lane-varying index into kernel arguments would spill the whole block)
template <typename R, class PT>
__device__ __forceinline__ F3P<R> f3_params_sel(const PT& P, bool first, int ba, int bb) {
  if constexpr (PT::indexed) return f3_params<R>(P, first ? ba : bb);
  return {first ? P[ba + 0] : P[bb + 0], first ? P[ba + 1] : P[bb + 1], first ? P[ba + 2] : P[bb + 2],
          first ? P[ba + 3] : P[bb + 3], first ? ba : bb};
}
template <typename R>
__device__ __forceinline__ FD<R> f3_eval(R r, R eps, const F3P<R>& p) {
  FD<R> o{R(0), R(0)};
  if (r < p.rstar) {
    const R ir = R(1) / r;
    const R s = p.sigma * ir;
    const R s2 = s * s;
    const R s6 = s2 * s2 * s2;
    const R s12 = s6 * s6;
    o.f = R(4) * eps * (s12 - s6);
    o.d = R(-24) * eps * (R(2) * s12 - s6) * ir;
  } else if (p.rstar < r && r < p.rc) {
    const R t = p.rc - r;
    o.f = eps * p.b * t * t;
    o.d = R(-2) * eps * p.b * t;
  }
  return o;
}
template <typename R, class PG>
__device__ __forceinline__ void f3_pgrad(R r, R eps, int ie, const F3P<R>& p, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  if (r < p.rstar) {
    const R s = p.sigma / r;
    const R s2 = s * s;
    const R s6 = s2 * s2 * s2;
    const R s12 = s6 * s6;
    pg.add(ie, scale * R(4) * (s12 - s6));
    pg.add(p.base + 1, scale * R(4) * eps * (R(12) * s12 - R(6) * s6) / p.sigma);
  } else if (p.rstar < r && r < p.rc) {
    const R t = p.rc - r;
    pg.add(ie, scale * p.b * t * t);
    pg.add(p.base + 2, scale * eps * t * t);
    pg.add(p.base + 3, scale * R(2) * eps * p.b * t);
  }
}

// ------------------------------------------------------------------ f4 (base_functions.py:82-107)
// block layout: T0,TS,TC,A,B
template <typename R>
struct F4P {
  R t0, ts, tc, a, b;
  int base;
};
template <typename R, class PT>
__device__ __forceinline__ F4P<R> f4_params(const PT& P, int b) {
  return {P[b + 0], P[b + 1], P[b + 2], P[b + 3], P[b + 4], b};
}
template <typename R, class PT>
__device__ __forceinline__ F4P<R> f4_params_sel(const PT& P, bool first, int ba, int bb) {
  if constexpr (PT::indexed) return f4_params<R>(P, first ? ba : bb);
  return {first ? P[ba + 0] : P[bb + 0], first ? P[ba + 1] : P[bb + 1], first ? P[ba + 2] : P[bb + 2],
          first ? P[ba + 3] : P[bb + 3], first ? P[ba + 4] : P[bb + 4], first ? ba : bb};
}
// Branch-free form for the fp32 stepping kernels (MYTHOS_LEAN_MATH, set by langevin_core.inc): f4 is symmetric about t0, so
// one |x| serves both tails; the three pieces are computed and selected (15 VALU, no exec-mask traffic) instead of the
// three-way divergent branch.  The only difference from the branchy form: at the exact breakpoints |x| == ts, where the
// reference's strict inequalities give 0 (a removable discontinuity), this form gives the continuous value.
template <typename R>
__device__ __forceinline__ FD<R> f4_eval_lean(R th, const F4P<R>& p) {
  const R x = th - p.t0;
  const R ax = x < R(0) ? -x : x;
  const R t = p.tc - ax;
  const R bt = p.b * t;
  const bool core = ax < p.ts, tail = ax < p.tc;
  const R dtail = x < R(0) ? R(2) * bt : R(-2) * bt;
  FD<R> o;
  o.f = core ? R(1) - p.a * ax * ax : (tail ? bt * t : R(0));
  o.d = core ? R(-2) * p.a * x : (tail ? dtail : R(0));
  return o;
}
template <typename R>
__device__ __forceinline__ FD<R> f4_eval(R th, const F4P<R>& p) {
  if constexpr (kLeanMath<R>) return f4_eval_lean(th, p);
  FD<R> o{R(0), R(0)};
  if (p.t0 - p.ts < th && th < p.t0 + p.ts) {
    const R x = th - p.t0;
    o.f = R(1) - p.a * x * x;
    o.d = R(-2) * p.a * x;
  } else if (p.t0 - p.tc < th && th < p.t0 - p.ts) {
    const R t = p.t0 - p.tc - th;
    o.f = p.b * t * t;
    o.d = R(-2) * p.b * t;
  } else if (p.t0 + p.ts < th && th < p.t0 + p.tc) {
    const R t = p.t0 + p.tc - th;
    o.f = p.b * t * t;
    o.d = R(-2) * p.b * t;
  }
  return o;
}
template <typename R, class PG>
__device__ __forceinline__ void f4_pgrad(R th, const F4P<R>& p, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  if (p.t0 - p.ts < th && th < p.t0 + p.ts) {
    const R x = th - p.t0;
    pg.add(p.base + 3, -scale * x * x);
    pg.add(p.base + 0, scale * R(2) * p.a * x);
  } else if (p.t0 - p.tc < th && th < p.t0 - p.ts) {
    const R t = p.t0 - p.tc - th;
    pg.add(p.base + 4, scale * t * t);
    pg.add(p.base + 0, scale * R(2) * p.b * t);
    pg.add(p.base + 2, -scale * R(2) * p.b * t);
  } else if (p.t0 + p.ts < th && th < p.t0 + p.tc) {
    const R t = p.t0 + p.tc - th;
    pg.add(p.base + 4, scale * t * t);
    pg.add(p.base + 0, scale * R(2) * p.b * t);
    pg.add(p.base + 2, scale * R(2) * p.b * t);
  }
}

// ------------------------------------------------------------------ f5 (base_functions.py:110-129)
// block layout: XS,XC,A,B
template <typename R>
struct F5P {
  R xs, xc, a, b;
  int base;
};
template <typename R, class PT>
__device__ __forceinline__ F5P<R> f5_params(const PT& P, int b) {
  return {P[b + 0], P[b + 1], P[b + 2], P[b + 3], b};
}
template <typename R, class PT>
__device__ __forceinline__ F5P<R> f5_params_sel(const PT& P, bool first, int ba, int bb) {
  if constexpr (PT::indexed) {
    const int b = first ? ba : bb;
    return {P[b + 0], P[b + 1], P[b + 2], P[b + 3], b};
  }
  return {first ? P[ba + 0] : P[bb + 0], first ? P[ba + 1] : P[bb + 1], first ? P[ba + 2] : P[bb + 2],
          first ? P[ba + 3] : P[bb + 3], first ? ba : bb};
}
template <typename R>
__device__ __forceinline__ FD<R> f5_eval(R x, const F5P<R>& p) {
  if constexpr (kLeanMath<R>) {  // (as f4_eval_lean; differs from the branchy form only at x == 0 and x == xs exactly)
    const R t = p.xc - x;
    const R bt = p.b * t;
    const bool one = x > R(0), core = x > p.xs, tail = x > p.xc;
    FD<R> o;
    o.f = one ? R(1) : (core ? R(1) - p.a * x * x : (tail ? bt * t : R(0)));
    o.d = one ? R(0) : (core ? R(-2) * p.a * x : (tail ? R(-2) * bt : R(0)));
    return o;
  }
  FD<R> o{R(0), R(0)};
  if (x > R(0)) {
    o.f = R(1);
  } else if (p.xs < x && x < R(0)) {
    o.f = R(1) - p.a * x * x;
    o.d = R(-2) * p.a * x;
  } else if (p.xc < x && x < p.xs) {
    const R t = p.xc - x;
    o.f = p.b * t * t;
    o.d = R(-2) * p.b * t;
  }
  return o;
}
template <typename R, class PG>
__device__ __forceinline__ void f5_pgrad(R x, const F5P<R>& p, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  if (x > R(0)) {
  } else if (p.xs < x && x < R(0)) {
    pg.add(p.base + 2, -scale * x * x);
  } else if (p.xc < x && x < p.xs) {
    const R t = p.xc - x;
    pg.add(p.base + 3, scale * t * t);
    pg.add(p.base + 1, scale * R(2) * p.b * t);
  }
}

// ------------------------------------------------------------------ f6 (dna2/base_functions.py:13-17)
// layout: A,B
template <typename R, class PT>
__device__ __forceinline__ FD<R> f6_eval(R th, const PT& P, int b) {
  FD<R> o{R(0), R(0)};
  if (th >= P[b + 1]) {
    const R t = th - P[b + 1];
    o.f = R(0.5) * P[b + 0] * t * t;
    o.d = P[b + 0] * t;
  }
  return o;
}
template <typename R, class PG, class PT>
__device__ __forceinline__ void f6_pgrad(R th, const PT& P, int b, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  if (th >= P[b + 1]) {
    const R t = th - P[b + 1];
    pg.add(b + 0, scale * R(0.5) * t * t);
    pg.add(b + 1, -scale * P[b + 0] * t);
  }
}

// ------------------------------------------------------------------ smoothed FENE (interactions.py:16-41)
template <typename R, class PT>
__device__ __forceinline__ FD<R> fene_eval(R r, const PT& P) {
  const R eps = P[FENE_EPS], x = r - P[FENE_R0], delta = P[FENE_DELTA];
  const R diff = m_sqrt(x * x + R(1e-10));
  FD<R> o;
  if (diff > P[FENE_XMAX]) {
    const R c = (P[FENE_FMAX] - P[FENE_FINF]) * P[FENE_XMAX];
    o.f = c * m_log(diff) + P[FENE_FINF] * diff + P[FENE_CONST];
    o.d = (c / diff + P[FENE_FINF]) * x / diff;
  } else {
    const R d2 = delta * delta;
    o.f = R(-0.5) * eps * m_log(R(1) - x * x / d2);
    o.d = eps * x / (d2 - x * x);
  }
  return o;
}
template <typename R, class PG, class PT>
__device__ __forceinline__ void fene_pgrad(R r, const PT& P, R dVdr, R scale, PG& pg) {
  if constexpr (!PG::on) return;
  const R eps = P[FENE_EPS], x = r - P[FENE_R0], delta = P[FENE_DELTA];
  const R diff = m_sqrt(x * x + R(1e-10));
  pg.add(FENE_R0, -scale * dVdr);
  if (diff > P[FENE_XMAX]) {
    const R ld = m_log(diff);
    pg.add(FENE_FMAX, scale * P[FENE_XMAX] * ld);
    pg.add(FENE_FINF, scale * (-P[FENE_XMAX] * ld + diff));
    pg.add(FENE_XMAX, scale * (P[FENE_FMAX] - P[FENE_FINF]) * ld);
    pg.add(FENE_CONST, scale);
  } else {
    const R d2 = delta * delta;
    pg.add(FENE_EPS, scale * R(-0.5) * m_log(R(1) - x * x / d2));
    pg.add(FENE_DELTA, -scale * eps * x * x / (delta * (d2 - x * x)));
  }
}

// ------------------------------------------------------------------ Debye-Hueckel (dna2/interactions.py:15-28)
template <typename R, class PT>
__device__ __forceinline__ FD<R> debye_eval(R r, const PT& P) {
  FD<R> o{R(0), R(0)};
  if (r < P[DH_RCUT]) {
    if (r < P[DH_RHIGH]) {
      const R ir = R(1) / r;
      o.f = m_exp(-P[DH_KAPPA] * r) * P[DH_PREFACTOR] * ir;
      o.d = -o.f * (P[DH_KAPPA] + ir);
    } else {
      const R t = r - P[DH_RCUT];
      o.f = P[DH_BSMOOTH] * t * t;
      o.d = R(2) * P[DH_BSMOOTH] * t;
    }
  }
  return o;
}
// the five Debye-Hueckel scalars read once (a kernel that evaluates the term in a loop keeps them in SGPRs
// instead of issuing a scalar load and a wait at every use)
template <typename R>
struct DebyeP {
  R rcut, rhigh, kappa, prefactor, bsmooth;
};
template <typename R, class PT>
__device__ __forceinline__ DebyeP<R> debye_params(const PT& P) {
  return {P[DH_RCUT], P[DH_RHIGH], P[DH_KAPPA], P[DH_PREFACTOR], P[DH_BSMOOTH]};
}
template <typename R>
__device__ __forceinline__ FD<R> debye_eval(R r, const DebyeP<R>& p) {
  FD<R> o{R(0), R(0)};
  if (r < p.rcut) {
    if (r < p.rhigh) {
      const R ir = R(1) / r;
      o.f = m_exp(-p.kappa * r) * p.prefactor * ir;
      o.d = -o.f * (p.kappa + ir);
    } else {
      const R t = r - p.rcut;
      o.f = p.bsmooth * t * t;
      o.d = R(2) * p.bsmooth * t;
    }
  }
  return o;
}
template <typename R, class PG, class PT>
__device__ __forceinline__ void debye_pgrad(R r, const PT& P, R mult, PG& pg) {
  if constexpr (!PG::on) return;
  if (r < P[DH_RCUT]) {
    if (r < P[DH_RHIGH]) {
      const R e = m_exp(-P[DH_KAPPA] * r) / r;
      pg.add(DH_KAPPA, -mult * r * e * P[DH_PREFACTOR]);
      pg.add(DH_PREFACTOR, mult * e);
    } else {
      const R t = r - P[DH_RCUT];
      pg.add(DH_BSMOOTH, mult * t * t);
      pg.add(DH_RCUT, -mult * R(2) * P[DH_BSMOOTH] * t);
    }
  }
}

}  // inline namespace MYTHOS_MATH_NS
}  // namespace mythos
