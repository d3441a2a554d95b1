// Per-frame structural observables of oxDNA duplex trajectories, evaluated by ONE workgroup per frame:
// propeller twist, helical rise, pitch angle and the persistence-length partials (mean base-pair spacing and the
// autocorrelation of the local helical axes).  Launched from two places: mythos_observables_eval, and
// mythos_oxdna_energy_obs (behind its energy launch, while the frames are still in L2), so that a DiffTRe evaluation -
// energies, dU/dtheta and the observable it reweights - is one call.
//
// What is computed follows the reference function by function:
//   propeller twist  mythos/observables/propeller.py:19-71   mean over the listed base pairs of 180 - acos(a3_i . a3_j) [deg]
//   local axis       mythos/observables/base.py:24-45        unit vector between the base-site midpoints of two adjacent pairs
//   rise             mythos/observables/rise.py:21-39        (midpoint displacement) . axis, in Angstrom
//   pitch angle      mythos/observables/pitch.py:33-59       angle between the backbone-backbone vectors of the two pairs
//                                                            after projecting out the axis [rad]
//   persistence      mythos/observables/persistence_length.py:47-91  C(d) = mean_i l_i . l_(i+d), <l0>; skip_ends drops two
//                                                            quartets at either end
// Arithmetic in fp64 whatever the state's precision (a few hundred flops per base pair); fixed-order reductions.
//
// Output row of a frame, `width` = 4 + n_corr doubles:
//   [0] propeller twist (deg)  [1] rise (Angstrom)  [2] pitch angle (rad)  [3] <l0> (oxDNA length units)  [4 + d] C(d)
#pragma once
#include <hip/hip_runtime.h>

#include "oxdna_math.h"

namespace mythos {

constexpr double kAngstromPerOxdnaLength = 8.518;  // mythos/utils/units.py:5-8

struct ObsView {
  const int* bps = nullptr;       // [n_bp][2] hydrogen-bonded pairs of the propeller twist
  const int* quartets = nullptr;  // [n_q][2][2] adjacent base pairs ((a1, b1), (a2, b2))
  double* axis = nullptr;         // scratch [frames][n_q][3]: the local axes of a frame (autocorrelation input)
  int n_bp = 0, n_q = 0;
  int skip = 0;                   // quartets dropped at either end for the persistence-length partials (0 or 2)
  int n_corr = 0;                 // n_q - 2 * skip (>= 0)
  int width = 0;                  // 4 + n_corr; 0 = no observables
  int model = 2;
  double g_hb = 0, g_k1 = 0, g_k2 = 0;  // base site c + g_hb a1; backbone site c + g_k1 a1 + g_k2 a2 (model 3: g_k2 a3)
  int box_on = 0;
  double box[3] = {1, 1, 1};
};

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ D3 obs_min_image(D3 d, const ObsView& v) {
  if (v.box_on) {
    d.x -= v.box[0] * rint(d.x / v.box[0]);
    d.y -= v.box[1] * rint(d.y / v.box[1]);
    d.z -= v.box[2] * rint(d.z / v.box[2]);
  }
  return d;
}

__device__ __forceinline__ double obs_clamp(double c) { return c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c); }

// sum over the workgroup in a fixed order: wavefront shuffles, then the wavefronts' partials in order.
// red: shared scratch of blockDim.x / 64 doubles.  Every thread of the workgroup must call it.
__device__ __forceinline__ double obs_block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();  // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// center [n][3], quat [n][4] of ONE frame; out [width]; axis scratch [n_q][3] of this frame.
template <typename R>
__device__ __forceinline__ void frame_observables(const ObsView& v, const R* __restrict__ center, const R* __restrict__ quat,
                                                  double* __restrict__ out, double* __restrict__ axis, double* red) {
  auto centre = [&](int i) { return D3{(double)center[3 * i], (double)center[3 * i + 1], (double)center[3 * i + 2]}; };
  auto axes = [&](int i, D3& a1, D3& a2, D3& a3) {
    const double q0 = quat[4 * i], q1 = quat[4 * i + 1], q2 = quat[4 * i + 2], q3 = quat[4 * i + 3];
    a1 = {q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)};
    a2 = {2 * (q1 * q2 - q0 * q3), q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2 * (q2 * q3 + q0 * q1)};
    a3 = {2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3};
  };
  // ---- propeller twist
  double pt = 0.0;
  for (int k = threadIdx.x; k < v.n_bp; k += blockDim.x) {
    D3 a1, a2, n1, n2;
    axes(v.bps[2 * k], a1, a2, n1);
    axes(v.bps[2 * k + 1], a1, a2, n2);
    pt += 180.0 - acos(obs_clamp(ddot(n1, n2))) * (180.0 / kPi);
  }
  pt = obs_block_sum(pt, red);
  // ---- quartets: axis, rise, pitch angle, spacing
  double rise = 0.0, ang = 0.0, l0 = 0.0;
  for (int k = threadIdx.x; k < v.n_q; k += blockDim.x) {
    const int ia1 = v.quartets[4 * k], ib1 = v.quartets[4 * k + 1], ia2 = v.quartets[4 * k + 2], ib2 = v.quartets[4 * k + 3];
    D3 x1, y1, z1, x2, y2, z2, x3, y3, z3, x4, y4, z4;
    axes(ia1, x1, y1, z1), axes(ib1, x2, y2, z2), axes(ia2, x3, y3, z3), axes(ib2, x4, y4, z4);
    const D3 c1 = centre(ia1), c2 = centre(ib1), c3 = centre(ia2), c4 = centre(ib2);
    const D3 m1 = 0.5 * ((c1 + v.g_hb * x1) + (c2 + v.g_hb * x2));
    const D3 m2 = 0.5 * ((c3 + v.g_hb * x3) + (c4 + v.g_hb * x4));
    const D3 dr = obs_min_image(m2 - m1, v);
    const double norm = sqrt(ddot(dr, dr));
    const D3 ax = (1.0 / norm) * dr;
    axis[3 * k] = ax.x, axis[3 * k + 1] = ax.y, axis[3 * k + 2] = ax.z;
    rise += ddot(dr, ax) * kAngstromPerOxdnaLength;
    if (k >= v.skip && k < v.n_q - v.skip) l0 += norm;
    // backbone-backbone vectors of the two pairs, helical component removed
    auto back = [&](D3 c, D3 a1, D3 ab) { return c + v.g_k1 * a1 + v.g_k2 * ab; };
    const bool on_a3 = v.model == 3;  // oxRNA2 backbone site: second coefficient on a3 (rna2/nucleotide.py:56)
    D3 bb1 = obs_min_image(back(c2, x2, on_a3 ? z2 : y2) - back(c1, x1, on_a3 ? z1 : y1), v);
    D3 bb2 = obs_min_image(back(c4, x4, on_a3 ? z4 : y4) - back(c3, x3, on_a3 ? z3 : y3), v);
    bb1 = obs_min_image(bb1 - ddot(ax, bb1) * ax, v);
    bb2 = obs_min_image(bb2 - ddot(ax, bb2) * ax, v);
    const double c = ddot(bb1, bb2) / sqrt(ddot(bb1, bb1) * ddot(bb2, bb2));
    ang += acos(obs_clamp(c));
  }
  rise = obs_block_sum(rise, red);
  ang = obs_block_sum(ang, red);
  l0 = obs_block_sum(l0, red);  // (its barriers also publish the axes of this frame to the whole workgroup)
  __threadfence_block();
  if (threadIdx.x == 0) {
    out[0] = v.n_bp > 0 ? pt / v.n_bp : 0.0;
    out[1] = v.n_q > 0 ? rise / v.n_q : 0.0;
    out[2] = v.n_q > 0 ? ang / v.n_q : 0.0;
    out[3] = v.n_corr > 0 ? l0 / v.n_corr : 0.0;
  }
  // ---- autocorrelation of the kept axes: C(d) = mean over the n_corr - d pairs at lag d
  const double* a = axis + 3 * v.skip;
  for (int d = threadIdx.x; d < v.n_corr; d += blockDim.x) {
    double s = 0.0;
    for (int i = 0; i + d < v.n_corr; ++i)
      s += a[3 * i] * a[3 * (i + d)] + a[3 * i + 1] * a[3 * (i + d) + 1] + a[3 * i + 2] * a[3 * (i + d) + 2];
    out[4 + d] = s / (v.n_corr - d);
  }
}

}  // namespace mythos

// host side of an observable set (observables.hip)
struct mythos_obs {
  int n = 0, dtype = 0, device = 0;
  mythos::ObsView view;     // device pointers filled in; view.axis is (re)allocated per call
  int* d_bps = nullptr;
  int* d_quartets = nullptr;
  double* d_axis = nullptr;
  size_t axis_cap = 0;      // doubles
};

namespace mythos {
// makes sure the axis scratch covers n_frames and returns the view to pass to a kernel
int obs_view_for(mythos_obs* o, int n_frames, ObsView* out);
// the stand-alone kernel on n_frames frames (view: from obs_view_for, its axis pointer already at the first frame)
int observables_launch(mythos_obs* o, const ObsView& view, const void* center, const void* quat, int n_frames, double* out,
                       hipStream_t stream);
}  // namespace mythos
