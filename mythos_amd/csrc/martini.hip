// MARTINI 2/3 energy path: shifted-cut-off Lennard-Jones over all non-bonded bead pairs, harmonic
// bonds, G96 (MARTINI 2) or harmonic (MARTINI 3) angles, per frame with that frame's periodic box.
// Replaces mythos/energy/martini/m2/lj.py:55-88,137-157, m2/bond.py:34-71, m2/angle.py:35-129 and
// m3/angle.py:8-11 (all evaluated by jax.vmap over an (M, M) pair mask in the reference).
//
// LJ: the classic LDS-tiled all-pairs sweep.  A 256-thread workgroup owns 256 beads; tiles of 256
// partner beads (position + type) are staged in LDS and every thread walks the tile, so each
// global position is read once per workgroup instead of once per pair.  The partner range is split
// over blockIdx.y so small systems still fill 256 CUs; partial forces / energies are combined by a
// fixed-order reduction (no atomics, reproducible).  Bonded pairs are excluded through a short
// per-bead list that is consulted only inside the cut-off.  Every pair is visited from both ends
// (energy weight 1/2).  Bonds and angles are gathered per bead from incidence lists.
// Roofline: the sweep is ALU/LDS bound, not HBM bound: 16 B per bead per tile pass from L2/HBM versus
// ~20 flops per pair per lane.
#include <algorithm>
#include <cmath>

#include "martini_internal.h"

namespace mythos {


template <typename R>
__global__ __launch_bounds__(kLjBlock) void martini_lj_kernel(
    int n, const R* __restrict__ pos, const R* __restrict__ box, const int* __restrict__ types,
    const R* __restrict__ sigma, const R* __restrict__ eps, const int* __restrict__ excl, MartiniConst<R> K,
    int n_tiles, R* __restrict__ f_part, double* __restrict__ e_part) {
  extern __shared__ unsigned char smem_raw[];
  R* s_sig = reinterpret_cast<R*>(smem_raw);
  R* s_eps = s_sig + K.n_types * K.n_types;
  R* s_shift = s_eps + K.n_types * K.n_types;  // V(r_c) per type pair
  R* s_x = s_shift + K.n_types * K.n_types;
  R* s_y = s_x + kLjBlock;
  R* s_z = s_y + kLjBlock;
  int* s_t = reinterpret_cast<int*>(s_z + kLjBlock);
  __shared__ double s_e[kLjBlock / 64];

  const int frame = blockIdx.z;
  const int js = blockIdx.y, n_js = gridDim.y;
  const int i = blockIdx.x * kLjBlock + threadIdx.x;
  const R* __restrict__ p = pos + (size_t)frame * n * 3;
  const R lx = box[frame * 3], ly = box[frame * 3 + 1], lz = box[frame * 3 + 2];
  const R ilx = R(1) / lx, ily = R(1) / ly, ilz = R(1) / lz;
  const int tt = K.n_types * K.n_types;
  const R irc2 = R(1) / K.rc2;
  for (int k = threadIdx.x; k < tt; k += kLjBlock) {
    const R sg = sigma[k], ep = eps[k];
    s_sig[k] = sg * sg;  // sigma^2
    s_eps[k] = ep;
    const R s2 = sg * sg * irc2, s6 = s2 * s2 * s2;
    s_shift[k] = R(4) * ep * (s6 * s6 - s6);
  }
  R xi = 0, yi = 0, zi = 0;
  int ti = 0;
  int ex[kMaxExcl];
#pragma unroll
  for (int k = 0; k < kMaxExcl; ++k) ex[k] = -1;
  if (i < n) {
    xi = p[3 * i], yi = p[3 * i + 1], zi = p[3 * i + 2];
    ti = types[i] * K.n_types;
#pragma unroll
    for (int k = 0; k < kMaxExcl; ++k) ex[k] = excl[(size_t)i * kMaxExcl + k];
  }
  R fx = 0, fy = 0, fz = 0;  // dU/dx_i
  double e = 0.0;
  for (int tile = js; tile < n_tiles; tile += n_js) {
    __syncthreads();
    const int j0 = tile * kLjBlock, jl = j0 + threadIdx.x;
    if (jl < n) {
      s_x[threadIdx.x] = p[3 * jl], s_y[threadIdx.x] = p[3 * jl + 1], s_z[threadIdx.x] = p[3 * jl + 2];
      s_t[threadIdx.x] = types[jl];
    }
    __syncthreads();
    const int cnt = min(kLjBlock, n - j0);
    if (i < n) {
      R et = 0;
      for (int k = 0; k < cnt; ++k) {
        const R dx = wrap(xi - s_x[k], lx, ilx), dy = wrap(yi - s_y[k], ly, ily), dz = wrap(zi - s_z[k], lz, ilz);
        const R r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < K.rc2) {
          const int j = j0 + k;
          bool skip = (j == i);
#pragma unroll
          for (int q = 0; q < kMaxExcl; ++q) skip = skip || (ex[q] == j);
          if (!skip) {
            const int tp = ti + s_t[k];
            const R ir2 = R(1) / r2;
            const R s2 = s_sig[tp] * ir2, s6 = s2 * s2 * s2, s12 = s6 * s6;
            const R ep = s_eps[tp];
            et += R(4) * ep * (s12 - s6) - s_shift[tp];
            const R g = R(-24) * ep * (R(2) * s12 - s6) * ir2;  // (dV/dr) / r
            fx += g * dx, fy += g * dy, fz += g * dz;
          }
        }
      }
      e += double(et);
    }
  }
  if (i < n) {
    R* o = f_part + (((size_t)frame * (n_js + 1) + js) * n + i) * 3;
    o[0] = fx, o[1] = fy, o[2] = fz;
  }
  // workgroup energy (pairs are double counted)
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
  if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < kLjBlock / 64; ++w) s += s_e[w];
    e_part[((size_t)frame * n_js + js) * gridDim.x + blockIdx.x] = 0.5 * s;
  }
}

// bonds and angles gathered per bead; slot n_js of f_part receives the gradient
template <typename R>
__global__ __launch_bounds__(256) void martini_bonded_kernel(
    int n, const R* __restrict__ pos, const R* __restrict__ box, const int* __restrict__ bead_bonds,
    const int* __restrict__ bead_angles, const int* __restrict__ bonds, const R* __restrict__ bond_k,
    const R* __restrict__ bond_r0, const int* __restrict__ angles, const R* __restrict__ angle_k,
    const R* __restrict__ angle_t0, int angle_kind, int n_js, R* __restrict__ f_part, double* __restrict__ eb_part) {
  __shared__ double s_e[2][4];
  const int frame = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const R* __restrict__ p = pos + (size_t)frame * n * 3;
  const R l[3] = {box[frame * 3], box[frame * 3 + 1], box[frame * 3 + 2]};
  const R il[3] = {R(1) / l[0], R(1) / l[1], R(1) / l[2]};
  double eb = 0.0, ea = 0.0;
  if (i < n) {
    R g[3] = {0, 0, 0};
    for (int s = 0; s < kMaxBeadBonds; ++s) {
      const int ent = bead_bonds[(size_t)i * kMaxBeadBonds + s];
      if (ent < 0) break;
      const int b = ent >> 1, side = ent & 1;
      const int o = bonds[2 * b + (1 - side)];
      R d[3], r2 = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d[k] = wrap(p[3 * i + k] - p[3 * o + k], l[k], il[k]);
        r2 += d[k] * d[k];
      }
      const R r = m_sqrt(r2), x = r - bond_r0[b];
      const R c = bond_k[b] * x / r;
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] += c * d[k];
      if (side == 0) eb += 0.5 * double(bond_k[b]) * double(x) * double(x);
    }
    for (int s = 0; s < kMaxBeadAngles; ++s) {
      const int ent = bead_angles[(size_t)i * kMaxBeadAngles + s];
      if (ent < 0) break;
      const int a = ent >> 2, role = ent & 3;  // role 0: first bead, 1: centre, 2: last bead
      const int bi = angles[3 * a], bj = angles[3 * a + 1], bk = angles[3 * a + 2];
      R u[3], v[3], u2 = 0, v2 = 0, uv = 0;  // u = r_i - r_j, v = r_k - r_j
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        u[k] = wrap(p[3 * bi + k] - p[3 * bj + k], l[k], il[k]);
        v[k] = wrap(p[3 * bk + k] - p[3 * bj + k], l[k], il[k]);
        u2 += u[k] * u[k], v2 += v[k] * v[k], uv += u[k] * v[k];
      }
      const R iu = R(1) / m_sqrt(u2), iv = R(1) / m_sqrt(v2);
      const R c = uv * iu * iv;  // cos(theta)
      // |uhat x vhat| for the atan2 form of the reference (m2/angle.py:49-58)
      R cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const R sn = m_sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) * iu * iv;
      R dEdc;
      double en;
      if (angle_kind == 0) {
        R c0;
        if constexpr (sizeof(R) == 4) c0 = cosf(angle_t0[a]); else c0 = cos(angle_t0[a]);
        const R x = c - c0;
        dEdc = angle_k[a] * x;
        en = 0.5 * double(angle_k[a]) * double(x) * double(x);
      } else {
        R th;
        if constexpr (sizeof(R) == 4) th = atan2f(sn, c); else th = atan2(sn, c);
        const R x = th - angle_t0[a];
        // d(theta)/d(cos) = -1/sin; (theta - pi)/sin(theta) -> -1 at theta = pi
        dEdc = (sn > R(1e-6)) ? -angle_k[a] * x / sn : angle_k[a];
        en = 0.5 * double(angle_k[a]) * double(x) * double(x);
      }
      if (role == 0) ea += en;
      // dc/du = (vhat - c uhat)/|u|, dc/dv = (uhat - c vhat)/|v|
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const R du = (v[k] * iv - c * u[k] * iu) * iu, dv = (u[k] * iu - c * v[k] * iv) * iv;
        const R gk = (role == 0) ? du : ((role == 2) ? dv : -(du + dv));
        g[k] += dEdc * gk;
      }
    }
    R* o = f_part + (((size_t)frame * (n_js + 1) + n_js) * n + i) * 3;
    o[0] = g[0], o[1] = g[1], o[2] = g[2];
  }
  for (int o = 32; o > 0; o >>= 1) {
    eb += __shfl_xor(eb, o, 64);
    ea += __shfl_xor(ea, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_e[0][threadIdx.x >> 6] = eb;
    s_e[1][threadIdx.x >> 6] = ea;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) s += s_e[threadIdx.x][w];
    eb_part[((size_t)frame * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Parameter gradients (the reference obtains them by jax.grad through the same functions).
// LJ: dU/dsigma[a][b] and dU/deps[a][b] per frame for the ORDERED type pair (type of the owner, type of the
// partner); every pair is visited from both ends with weight 1/2, so the derivative with respect to a
// symmetric table entry is out[a][b] + out[b][a] - which is what differentiating through the host's
// symmetric table construction yields.
// Reproducible bit for bit (round 4; until then the tables were one LDS copy shared by the four wavefronts and fp64
// global atomics per touched entry): every WAVEFRONT accumulates into its own LDS copy of the two tables - what a copy
// receives comes from one instruction stream in program order, colliding lanes of an instruction are served in lane
// order -, the four copies are added in a fixed order into the workgroup's partial tables in HBM, and a second kernel
// adds the workgroups' partials of a frame in index order.  No atomic touches global memory.
// ------------------------------------------------------------------------------------------------
constexpr int kLjWaves = kLjBlock / 64;
template <typename R>
__global__ __launch_bounds__(kLjBlock) void martini_lj_pgrad_kernel(
    int n, const R* __restrict__ pos, const R* __restrict__ box, const int* __restrict__ types,
    const R* __restrict__ sigma, const R* __restrict__ eps, const int* __restrict__ excl, MartiniConst<R> K,
    int n_tiles, double* __restrict__ part /* [frame][workgroup][2 tt] */) {
  extern __shared__ unsigned char smem_raw[];
  const int tt = K.n_types * K.n_types;
  double* s_tab = reinterpret_cast<double*>(smem_raw);  // [kLjWaves][2 tt]: dU/dsigma | dU/deps, one copy per wavefront
  double* s_ds = s_tab + (size_t)(threadIdx.x >> 6) * 2 * tt;
  double* s_de = s_ds + tt;
  R* s_sig = reinterpret_cast<R*>(s_tab + (size_t)kLjWaves * 2 * tt);
  R* s_eps = s_sig + K.n_types * K.n_types;
  R* s_x = s_eps + K.n_types * K.n_types;
  R* s_y = s_x + kLjBlock;
  R* s_z = s_y + kLjBlock;
  int* s_t = reinterpret_cast<int*>(s_z + kLjBlock);

  const int frame = blockIdx.z;
  const int js = blockIdx.y, n_js = gridDim.y;
  const int i = blockIdx.x * kLjBlock + threadIdx.x;
  const R* __restrict__ p = pos + (size_t)frame * n * 3;
  const R lx = box[frame * 3], ly = box[frame * 3 + 1], lz = box[frame * 3 + 2];
  const R ilx = R(1) / lx, ily = R(1) / ly, ilz = R(1) / lz;
  const R irc2 = R(1) / K.rc2;
  for (int k = threadIdx.x; k < tt; k += kLjBlock) {
    s_sig[k] = sigma[k];
    s_eps[k] = eps[k];
  }
  for (int k = threadIdx.x; k < kLjWaves * 2 * tt; k += kLjBlock) s_tab[k] = 0.0;
  R xi = 0, yi = 0, zi = 0;
  int ti = 0;
  int ex[kMaxExcl];
#pragma unroll
  for (int k = 0; k < kMaxExcl; ++k) ex[k] = -1;
  if (i < n) {
    xi = p[3 * i], yi = p[3 * i + 1], zi = p[3 * i + 2];
    ti = types[i] * K.n_types;
#pragma unroll
    for (int k = 0; k < kMaxExcl; ++k) ex[k] = excl[(size_t)i * kMaxExcl + k];
  }
  for (int tile = js; tile < n_tiles; tile += n_js) {
    __syncthreads();
    const int j0 = tile * kLjBlock, jl = j0 + threadIdx.x;
    if (jl < n) {
      s_x[threadIdx.x] = p[3 * jl], s_y[threadIdx.x] = p[3 * jl + 1], s_z[threadIdx.x] = p[3 * jl + 2];
      s_t[threadIdx.x] = types[jl];
    }
    __syncthreads();
    const int cnt = min(kLjBlock, n - j0);
    if (i < n) {
      for (int k = 0; k < cnt; ++k) {
        const R dx = wrap(xi - s_x[k], lx, ilx), dy = wrap(yi - s_y[k], ly, ily), dz = wrap(zi - s_z[k], lz, ilz);
        const R r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < K.rc2) {
          const int j = j0 + k;
          bool skip = (j == i);
#pragma unroll
          for (int q = 0; q < kMaxExcl; ++q) skip = skip || (ex[q] == j);
          if (!skip) {
            const int tp = ti + s_t[k];
            const R sg = s_sig[tp], ep = s_eps[tp];
            const R ir2 = R(1) / r2;
            const R s2 = sg * sg * ir2, s6 = s2 * s2 * s2, s12 = s6 * s6;
            const R c2 = sg * sg * irc2, c6 = c2 * c2 * c2, c12 = c6 * c6;
            // V = 4 eps [(s12 - s6) - (c12 - c6)];  dV/dsigma = 4 eps [12 s12 - 6 s6 - 12 c12 + 6 c6] / sigma
            atomicAdd(&s_de[tp], 0.5 * double(R(4) * ((s12 - s6) - (c12 - c6))));
            atomicAdd(&s_ds[tp], 0.5 * double(R(4) * ep * (R(12) * (s12 - c12) - R(6) * (s6 - c6)) / sg));
          }
        }
      }
    }
  }
  __syncthreads();
  // the workgroup's partial tables: the wavefronts' copies added in wavefront order
  double* __restrict__ out = part + ((size_t)frame * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * tt;
  for (int k = threadIdx.x; k < 2 * tt; k += kLjBlock) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kLjWaves; ++w) acc += s_tab[(size_t)w * 2 * tt + k];
    out[k] = acc;
  }
}

// d_sigma[frame][k], d_eps[frame][k] = the sum over the frame's workgroups, in index order, of their partial tables
__global__ __launch_bounds__(256) void martini_lj_pgrad_reduce_kernel(const double* __restrict__ part, int n_wg, int tt,
                                                                      double* __restrict__ d_sigma, double* __restrict__ d_eps) {
  const int frame = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 2 * tt) return;
  const double* p = part + (size_t)frame * n_wg * 2 * tt + k;
  double acc = 0.0;
  for (int w = 0; w < n_wg; ++w) acc += p[(size_t)w * 2 * tt];
  if (k < tt) d_sigma[(size_t)frame * tt + k] = acc; else d_eps[(size_t)frame * tt + (k - tt)] = acc;
}

// one thread per bond / per angle: dE/dk and dE/dr0 (dE/dtheta0), E as in martini_bonded_kernel
template <typename R>
__global__ void martini_bonded_pgrad_kernel(int n, const R* __restrict__ pos, const R* __restrict__ box, int n_bonds,
                                            const int* __restrict__ bonds, const R* __restrict__ bond_k,
                                            const R* __restrict__ bond_r0, int n_angles,
                                            const int* __restrict__ angles, const R* __restrict__ angle_k,
                                            const R* __restrict__ angle_t0, int angle_kind,
                                            double* __restrict__ d_bk, double* __restrict__ d_br,
                                            double* __restrict__ d_ak, double* __restrict__ d_at) {
  const int frame = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const R* __restrict__ p = pos + (size_t)frame * n * 3;
  const R l[3] = {box[frame * 3], box[frame * 3 + 1], box[frame * 3 + 2]};
  const R il[3] = {R(1) / l[0], R(1) / l[1], R(1) / l[2]};
  if (t < n_bonds) {
    const int i = bonds[2 * t], o = bonds[2 * t + 1];
    R r2 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const R d = wrap(p[3 * i + k] - p[3 * o + k], l[k], il[k]);
      r2 += d * d;
    }
    const double x = double(m_sqrt(r2)) - double(bond_r0[t]);
    if (d_bk) d_bk[(size_t)frame * n_bonds + t] = 0.5 * x * x;
    if (d_br) d_br[(size_t)frame * n_bonds + t] = -double(bond_k[t]) * x;
  }
  if (t < n_angles) {
    const int bi = angles[3 * t], bj = angles[3 * t + 1], bk = angles[3 * t + 2];
    R u[3], v[3], u2 = 0, v2 = 0, uv = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u[k] = wrap(p[3 * bi + k] - p[3 * bj + k], l[k], il[k]);
      v[k] = wrap(p[3 * bk + k] - p[3 * bj + k], l[k], il[k]);
      u2 += u[k] * u[k], v2 += v[k] * v[k], uv += u[k] * v[k];
    }
    const R iu = R(1) / m_sqrt(u2), iv = R(1) / m_sqrt(v2);
    const R c = uv * iu * iv;
    const double k0 = double(angle_k[t]), t0 = double(angle_t0[t]);
    double dk, dt0;
    if (angle_kind == 0) {
      const double x = double(c) - cos(t0);
      dk = 0.5 * x * x;
      dt0 = k0 * x * sin(t0);  // d/dtheta0 of 1/2 k (cos theta - cos theta0)^2
    } else {
      R cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const R sn = m_sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) * iu * iv;
      const double x = atan2(double(sn), double(c)) - t0;
      dk = 0.5 * x * x;
      dt0 = -k0 * x;
    }
    if (d_ak) d_ak[(size_t)frame * n_angles + t] = dk;
    if (d_at) d_at[(size_t)frame * n_angles + t] = dt0;
  }
}

template <typename R>
__global__ void martini_reduce_kernel(int n, int n_slots, const R* __restrict__ f_part, R* __restrict__ dU,
                                      const double* __restrict__ e_part, int n_e, const double* __restrict__ eb_part,
                                      int n_eb, double* __restrict__ e_terms) {
  const int frame = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (dU && t < n * 3) {
    R s = 0;
    for (int k = 0; k < n_slots; ++k) s += f_part[((size_t)frame * n_slots + k) * n * 3 + t];
    dU[(size_t)frame * n * 3 + t] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < 3) {
    double s = 0;
    if (threadIdx.x == 0) {
      for (int k = 0; k < n_e; ++k) s += e_part[(size_t)frame * n_e + k];
    } else {
      for (int k = 0; k < n_eb; ++k) s += eb_part[((size_t)frame * n_eb + k) * 2 + (threadIdx.x - 1)];
    }
    e_terms[(size_t)frame * 3 + threadIdx.x] = s;
  }
}

}  // namespace mythos

using namespace mythos;


namespace mythos {

template <typename R>
static bool upload_real(void** dst, const double* src, size_t count) {
  std::vector<R> tmp(std::max<size_t>(count, 1));
  for (size_t k = 0; k < count; ++k) tmp[k] = R(src[k]);
  return hipMalloc(dst, tmp.size() * sizeof(R)) == hipSuccess &&
         hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(R), hipMemcpyHostToDevice) == hipSuccess;
}

static bool upload_int(int** dst, const std::vector<int>& v) {
  const size_t c = std::max<size_t>(v.size(), 1);
  return hipMalloc((void**)dst, c * sizeof(int)) == hipSuccess &&
         (v.empty() || hipMemcpy(*dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess);
}

template <typename T>
static int grow(T*& ptr, size_t& cap, size_t need) {
  if (need <= cap) return 0;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  MYTHOS_HIP_TRY(hipMalloc((void**)&ptr, need * sizeof(T)));
  cap = need;
  return 0;
}

template <typename R>
static int martini_energy_typed(mythos_martini* m, const R* pos, const R* box, int n_frames, double* e_terms, R* dU,
                                hipStream_t st) {
  const int n = m->n;
  const int nbx = (n + kLjBlock - 1) / kLjBlock;
  const int n_tiles = nbx;
  const int nbb = (n + 255) / 256;
  // frames per launch: grid.z <= 65535 and partial-force scratch <= 512 MB
  for (int f0 = 0; f0 < n_frames;) {
    int n_js = std::min(n_tiles, std::max(1, 768 / std::max(1, nbx)));
    const size_t per_frame = (size_t)(n_js + 1) * n * 3 * sizeof(R);
    int nf = (int)std::min<size_t>(std::min(n_frames - f0, 4096), std::max<size_t>(1, (size_t(512) << 20) / per_frame));
    if (nf > 16) n_js = std::max(1, n_js / 4);  // many frames already fill the GPU
    char* fp = (char*)m->d_fpart;
    if (int rc = grow(fp, m->fpart_cap, (size_t)nf * (n_js + 1) * n * 3 * sizeof(R))) return rc;
    m->d_fpart = fp;
    if (int rc = grow(m->d_epart, m->epart_cap, (size_t)nf * n_js * nbx)) return rc;
    if (int rc = grow(m->d_ebpart, m->ebpart_cap, (size_t)nf * nbb * 2)) return rc;
    MartiniConst<R> K{R(m->r_cut * m->r_cut), m->n_types, m->angle_kind};
    const size_t lds = (size_t)3 * m->n_types * m->n_types * sizeof(R) + 3 * kLjBlock * sizeof(R) + kLjBlock * sizeof(int);
    const R* p = pos + (size_t)f0 * n * 3;
    const R* b = box + (size_t)f0 * 3;
    MYTHOS_HIP_TRY(hipFuncSetAttribute((const void*)martini_lj_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(martini_lj_kernel<R>, dim3(nbx, n_js, nf), dim3(kLjBlock), lds, st, n, p, b, m->d_types,
                       (const R*)m->d_sigma, (const R*)m->d_eps, m->d_excl, K, n_tiles, (R*)m->d_fpart, m->d_epart);
    hipLaunchKernelGGL(martini_bonded_kernel<R>, dim3(nbb, nf), dim3(256), 0, st, n, p, b, m->d_bead_bonds,
                       m->d_bead_angles, m->d_bonds, (const R*)m->d_bond_k, (const R*)m->d_bond_r0, m->d_angles,
                       (const R*)m->d_angle_k, (const R*)m->d_angle_t0, m->angle_kind, n_js, (R*)m->d_fpart,
                       m->d_ebpart);
    hipLaunchKernelGGL(martini_reduce_kernel<R>, dim3((n * 3 + 255) / 256, nf), dim3(256), 0, st, n, n_js + 1,
                       (const R*)m->d_fpart, dU ? dU + (size_t)f0 * n * 3 : nullptr, m->d_epart, n_js * nbx,
                       m->d_ebpart, nbb, e_terms + (size_t)f0 * 3);
    MYTHOS_HIP_TRY(hipGetLastError());
    f0 += nf;
  }
  return 0;
}

template <typename R>
static int martini_pgrad_typed(mythos_martini* m, const R* pos, const R* box, int n_frames, double* d_sigma,
                               double* d_eps, double* d_bk, double* d_br, double* d_ak, double* d_at, hipStream_t st) {
  const int n = m->n;
  const int nbx = (n + kLjBlock - 1) / kLjBlock;
  const size_t tt = (size_t)m->n_types * m->n_types;
  MartiniConst<R> K{R(m->r_cut * m->r_cut), m->n_types, m->angle_kind};
  if (d_sigma || d_eps) {
    // both tables are produced together; a caller that wants one still passes scratch for the other
    if (!d_sigma || !d_eps) {
      set_error("mythos_martini_param_grads: d_sigma and d_eps come as a pair");
      return MYTHOS_ERR_INVALID_ARGUMENT;
    }
    const size_t lds = (size_t)kLjWaves * 2 * tt * sizeof(double) + 2 * tt * sizeof(R) + 3 * kLjBlock * sizeof(R) + kLjBlock * sizeof(int);
    if (lds > 160 * 1024) {
      set_error("mythos_martini_param_grads: too many bead types for the per-wavefront gradient tables (" + std::to_string(m->n_types) + ")");
      return MYTHOS_ERR_INVALID_ARGUMENT;
    }
    MYTHOS_HIP_TRY(hipFuncSetAttribute((const void*)martini_lj_pgrad_kernel<R>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int n_js_all = std::min(nbx, std::max(1, 768 / std::max(1, nbx)));
    const int n_js = n_frames > 16 ? std::max(1, n_js_all / 4) : n_js_all;
    const int n_wg = nbx * n_js;  // workgroups, and partial tables, per frame
    // frames per chunk: at most 256 MB of partial tables
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(4096, (size_t(256) << 20) / ((size_t)n_wg * 2 * tt * sizeof(double))));
    if (int rc = grow(m->d_ljpart, m->ljpart_cap, (size_t)std::min(chunk, n_frames) * n_wg * 2 * tt)) return rc;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
      const int nf = std::min(n_frames - f0, chunk);
      hipLaunchKernelGGL(martini_lj_pgrad_kernel<R>, dim3(nbx, n_js, nf),
                         dim3(kLjBlock), lds, st, n, pos + (size_t)f0 * n * 3, box + (size_t)f0 * 3, m->d_types,
                         (const R*)m->d_sigma, (const R*)m->d_eps, m->d_excl, K, nbx, m->d_ljpart);
      hipLaunchKernelGGL(martini_lj_pgrad_reduce_kernel, dim3((unsigned)((2 * tt + 255) / 256), nf), dim3(256), 0, st,
                         (const double*)m->d_ljpart, n_wg, (int)tt, d_sigma + (size_t)f0 * tt, d_eps + (size_t)f0 * tt);
    }
  }
  if (d_bk || d_br || d_ak || d_at) {
    const int cnt = std::max(m->n_bonds, m->n_angles);
    for (int f0 = 0; cnt > 0 && f0 < n_frames; f0 += 4096) {
      const int nf = std::min(n_frames - f0, 4096);
      auto off = [&](double* ptr, int per) { return ptr ? ptr + (size_t)f0 * per : nullptr; };
      hipLaunchKernelGGL(martini_bonded_pgrad_kernel<R>, dim3((cnt + 255) / 256, nf), dim3(256), 0, st, n,
                         pos + (size_t)f0 * n * 3, box + (size_t)f0 * 3, m->n_bonds, m->d_bonds,
                         (const R*)m->d_bond_k, (const R*)m->d_bond_r0, m->n_angles, m->d_angles,
                         (const R*)m->d_angle_k, (const R*)m->d_angle_t0, m->angle_kind, off(d_bk, m->n_bonds),
                         off(d_br, m->n_bonds), off(d_ak, m->n_angles), off(d_at, m->n_angles));
    }
  }
  MYTHOS_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace mythos

extern "C" {

mythos_martini_t* mythos_martini_create(int n, const int32_t* types, int n_types, const double* sigma,
                                        const double* eps, int n_bonds, const int32_t* bonds, const double* bond_k,
                                        const double* bond_r0, int n_angles, const int32_t* angles,
                                        const double* angle_k, const double* angle_t0, int angle_kind, double r_cut,
                                        int dtype, int device) {
  if (n < 1 || !types || n_types < 1 || n_types > kMaxTypes || !sigma || !eps || n_bonds < 0 || n_angles < 0 ||
      (n_bonds > 0 && (!bonds || !bond_k || !bond_r0)) || (n_angles > 0 && (!angles || !angle_k || !angle_t0)) ||
      (angle_kind != 0 && angle_kind != 1) || !(r_cut > 0) || (dtype != MYTHOS_F32 && dtype != MYTHOS_F64)) {
    set_error("mythos_martini_create: invalid argument (1 <= n_types <= 64, angle_kind 0|1)");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0 || device >= ndev ||
      hipSetDevice(device) != hipSuccess) {
    set_error("mythos_martini_create: no usable HIP device (the HIP path has no CPU fallback)");
    return nullptr;
  }
  std::vector<int> excl((size_t)n * kMaxExcl, -1), bb((size_t)n * kMaxBeadBonds, -1), ba((size_t)n * kMaxBeadAngles, -1);
  std::vector<int> ne(n, 0), nb(n, 0), na(n, 0);
  std::vector<int> hb(bonds, bonds + 2 * (size_t)n_bonds), ha(angles, angles + 3 * (size_t)n_angles), ht(types, types + n);
  for (int i = 0; i < n; ++i)
    if (types[i] < 0 || types[i] >= n_types) {
      set_error("mythos_martini_create: bead type out of range");
      return nullptr;
    }
  for (int b = 0; b < n_bonds; ++b) {
    const int i = bonds[2 * b], j = bonds[2 * b + 1];
    if (i < 0 || j < 0 || i >= n || j >= n || i == j || ne[i] >= kMaxExcl || ne[j] >= kMaxExcl ||
        nb[i] >= kMaxBeadBonds || nb[j] >= kMaxBeadBonds) {
      set_error("mythos_martini_create: bad bond (index range, or more than 8 bonds on one bead)");
      return nullptr;
    }
    excl[(size_t)i * kMaxExcl + ne[i]++] = j;
    excl[(size_t)j * kMaxExcl + ne[j]++] = i;
    bb[(size_t)i * kMaxBeadBonds + nb[i]++] = 2 * b + 0;
    bb[(size_t)j * kMaxBeadBonds + nb[j]++] = 2 * b + 1;
  }
  for (int a = 0; a < n_angles; ++a)
    for (int r = 0; r < 3; ++r) {
      const int i = angles[3 * a + r];
      if (i < 0 || i >= n || na[i] >= kMaxBeadAngles) {
        set_error("mythos_martini_create: bad angle (index range, or more than 12 angles on one bead)");
        return nullptr;
      }
      ba[(size_t)i * kMaxBeadAngles + na[i]++] = 4 * a + r;
    }
  auto* m = new mythos_martini();
  m->n = n, m->n_types = n_types, m->n_bonds = n_bonds, m->n_angles = n_angles, m->angle_kind = angle_kind;
  m->dtype = dtype, m->device = device, m->r_cut = r_cut;
  bool ok = upload_int(&m->d_types, ht) && upload_int(&m->d_excl, excl) && upload_int(&m->d_bead_bonds, bb) &&
            upload_int(&m->d_bead_angles, ba) && upload_int(&m->d_bonds, hb) && upload_int(&m->d_angles, ha);
  const size_t tt = (size_t)n_types * n_types;
  if (dtype == MYTHOS_F32)
    ok = ok && upload_real<float>(&m->d_sigma, sigma, tt) && upload_real<float>(&m->d_eps, eps, tt) &&
         upload_real<float>(&m->d_bond_k, bond_k, n_bonds) && upload_real<float>(&m->d_bond_r0, bond_r0, n_bonds) &&
         upload_real<float>(&m->d_angle_k, angle_k, n_angles) && upload_real<float>(&m->d_angle_t0, angle_t0, n_angles);
  else
    ok = ok && upload_real<double>(&m->d_sigma, sigma, tt) && upload_real<double>(&m->d_eps, eps, tt) &&
         upload_real<double>(&m->d_bond_k, bond_k, n_bonds) && upload_real<double>(&m->d_bond_r0, bond_r0, n_bonds) &&
         upload_real<double>(&m->d_angle_k, angle_k, n_angles) && upload_real<double>(&m->d_angle_t0, angle_t0, n_angles);
  if (!ok) {
    set_error("mythos_martini_create: device allocation failed");
    mythos_martini_destroy(m);
    return nullptr;
  }
  return m;
}

void mythos_martini_destroy(mythos_martini_t* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  void* ptrs[] = {m->d_types, m->d_excl,   m->d_bead_bonds, m->d_bead_angles, m->d_bonds,  m->d_angles, m->d_sigma,
                  m->d_eps,   m->d_bond_k, m->d_bond_r0,    m->d_angle_k,     m->d_angle_t0, m->d_fpart, m->d_epart,
                  m->d_ebpart, m->d_ljpart};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete m;
}

int mythos_martini_energy(mythos_martini_t* m, const void* pos, const void* box, int n_frames, double* e_terms,
                          void* dU_dpos, mythos_stream_t stream) {
  if (!m || !pos || !box || !e_terms || n_frames < 0) {
    set_error("mythos_martini_energy: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (n_frames == 0) return MYTHOS_OK;
  MYTHOS_HIP_TRY(hipSetDevice(m->device));
  if (m->dtype == MYTHOS_F32)
    return martini_energy_typed<float>(m, (const float*)pos, (const float*)box, n_frames, e_terms, (float*)dU_dpos,
                                       (hipStream_t)stream);
  return martini_energy_typed<double>(m, (const double*)pos, (const double*)box, n_frames, e_terms, (double*)dU_dpos,
                                      (hipStream_t)stream);
}

int mythos_martini_param_grads(mythos_martini_t* m, const void* pos, const void* box, int n_frames, double* d_sigma,
                               double* d_eps, double* d_bond_k, double* d_bond_r0, double* d_angle_k,
                               double* d_angle_t0, mythos_stream_t stream) {
  if (!m || !pos || !box || n_frames < 0) {
    set_error("mythos_martini_param_grads: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (n_frames == 0) return MYTHOS_OK;
  MYTHOS_HIP_TRY(hipSetDevice(m->device));
  if (m->dtype == MYTHOS_F32)
    return martini_pgrad_typed<float>(m, (const float*)pos, (const float*)box, n_frames, d_sigma, d_eps, d_bond_k,
                                      d_bond_r0, d_angle_k, d_angle_t0, (hipStream_t)stream);
  return martini_pgrad_typed<double>(m, (const double*)pos, (const double*)box, n_frames, d_sigma, d_eps, d_bond_k,
                                     d_bond_r0, d_angle_k, d_angle_t0, (hipStream_t)stream);
}

}  // extern "C"
