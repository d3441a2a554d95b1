// C++/OpenMP port of the MARTINI force evaluation + Langevin step  --  TEST INFRASTRUCTURE ONLY.
//
// What it is for: bench.py's cpu_baseline leg of BASELINE configs[2] (20 480-bead bilayer; the torch oracle,
// oracle/martini_oracle.py, sums all M (M - 1) / 2 pairs and cannot step a system of that size), and a third
// implementation of the step for tests/test_cpu_port.py, which holds it to oracle/martini_oracle.py (pinned to the
// GROMACS energies the reference ships) and oracle/martini_langevin_oracle.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it; nothing under mythos_amd/ does.
//
// Algorithm (the reference's energy terms, restated): shifted-cut-off Lennard-Jones over the pairs inside r_c that are
// not directly bonded (mythos/energy/martini/m2/lj.py:55-88,137-157), harmonic bonds (m2/bond.py:34-40), G96 cosine
// angles (m2/angle.py:35-93) or harmonic angles (m3/angle.py:8-11), minimum image in an orthorhombic box
// (martini/base.py:15-17); BAOAB Langevin for point particles as oracle/martini_langevin_oracle.py restates it (the
// reference runs MARTINI dynamics in GROMACS, so the integrator is unpinned).  fp64.  Decomposition: every bead gathers
// its own Verlet row (each pair visited from both ends, as on the GPU: no scatter, no atomics, results independent of
// the thread count); OpenMP over beads; the rows come from a cell list rebuilt every `rebuild_every` steps.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include <omp.h>

#include "philox.h"

using namespace mythos;

namespace {

struct MSys {
  int n = 0, n_types = 0, angle_kind = 0;
  double r_cut = 1.1;
  std::vector<int> types;
  std::vector<double> sigma, eps;  // [T][T]
  std::vector<int> bonds, angles;  // [nb][2], [na][3]
  std::vector<double> bk, br0, ak, at0, inv_mass;
  std::vector<int> excl_ptr, excl;       // CSR of directly bonded partners
  std::vector<int> bb_ptr, bb;           // per bead: bond index << 1 | side
  std::vector<int> ba_ptr, ba;           // per bead: angle index << 2 | role
  std::vector<int> row_ptr, row;         // Verlet rows
};

inline double wrap(double d, double l) { return d - l * std::rint(d / l); }

void build_rows(MSys& s, const double* x, const double* box, double r_list) {
  const int n = s.n;
  int nc[3];
  for (int k = 0; k < 3; ++k) nc[k] = std::max(1, (int)std::floor(box[k] / r_list));
  const bool cells = nc[0] >= 3 && nc[1] >= 3 && nc[2] >= 3;
  const double rl2 = r_list * r_list;
  std::vector<int> cell_of(n), head, order(n);
  if (cells) {
    const int ncell = nc[0] * nc[1] * nc[2];
    head.assign(ncell + 1, 0);
    for (int i = 0; i < n; ++i) {
      int c[3];
      for (int k = 0; k < 3; ++k) {
        double f = x[3 * i + k] / box[k];
        f -= std::floor(f);
        c[k] = std::min(nc[k] - 1, (int)(f * nc[k]));
      }
      cell_of[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
      ++head[cell_of[i] + 1];
    }
    for (int c = 0; c < ncell; ++c) head[c + 1] += head[c];
    std::vector<int> fill(head.begin(), head.end() - 1);
    for (int i = 0; i < n; ++i) order[fill[cell_of[i]]++] = i;  // ascending bead index inside a cell
  }
  std::vector<std::vector<int>> rows(n);
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n; ++i) {
    std::vector<int>& r = rows[i];
    auto consider = [&](int j) {
      if (j == i) return;
      for (int t = s.excl_ptr[i]; t < s.excl_ptr[i + 1]; ++t)
        if (s.excl[t] == j) return;
      const double dx = wrap(x[3 * j] - x[3 * i], box[0]), dy = wrap(x[3 * j + 1] - x[3 * i + 1], box[1]),
                   dz = wrap(x[3 * j + 2] - x[3 * i + 2], box[2]);
      if (dx * dx + dy * dy + dz * dz < rl2) r.push_back(j);
    };
    if (!cells) {
      for (int j = 0; j < n; ++j) consider(j);
    } else {
      const int c = cell_of[i], cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int ox = (cx + dx + nc[0]) % nc[0], oy = (cy + dy + nc[1]) % nc[1], oz = (cz + dz + nc[2]) % nc[2];
            const int oc = (oz * nc[1] + oy) * nc[0] + ox;
            for (int t = head[oc]; t < head[oc + 1]; ++t) consider(order[t]);
          }
      std::sort(r.begin(), r.end());
    }
  }
  s.row_ptr.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) s.row_ptr[i + 1] = s.row_ptr[i] + (int)rows[i].size();
  s.row.resize(s.row_ptr[n]);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) std::copy(rows[i].begin(), rows[i].end(), s.row.begin() + s.row_ptr[i]);
}

// dU/dx per bead and the three energies (each pair / bond / angle counted once)
void forces(const MSys& s, const double* x, const double* box, double* g, double* e3) {
  const int n = s.n, T = s.n_types;
  const double rc2 = s.r_cut * s.r_cut, irc2 = 1.0 / rc2;
  const int nt = omp_get_max_threads();
  std::vector<double> e_thr((size_t)nt * 3, 0.0);
#pragma omp parallel
  {
    double* et = e_thr.data() + (size_t)omp_get_thread_num() * 3;
#pragma omp for schedule(static)
    for (int i = 0; i < n; ++i) {
      double gx = 0, gy = 0, gz = 0, e_lj = 0, e_b = 0, e_a = 0;
      const double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
      const int ti = s.types[i] * T;
      for (int t = s.row_ptr[i]; t < s.row_ptr[i + 1]; ++t) {
        const int j = s.row[t];
        const double dx = wrap(xi - x[3 * j], box[0]), dy = wrap(yi - x[3 * j + 1], box[1]), dz = wrap(zi - x[3 * j + 2], box[2]);
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 >= rc2) continue;
        const int tp = ti + s.types[j];
        const double sg = s.sigma[tp], ep = s.eps[tp];
        const double ir2 = 1.0 / r2, s2 = sg * sg * ir2, s6 = s2 * s2 * s2, s12 = s6 * s6;
        const double c = -24.0 * ep * (2.0 * s12 - s6) * ir2;  // (dV/dr) / r
        gx += c * dx, gy += c * dy, gz += c * dz;
        const double c2 = sg * sg * irc2, c6 = c2 * c2 * c2;
        e_lj += 0.5 * 4.0 * ep * ((s12 - s6) - (c6 * c6 - c6));
      }
      for (int t = s.bb_ptr[i]; t < s.bb_ptr[i + 1]; ++t) {
        const int b = s.bb[t] >> 1, side = s.bb[t] & 1, o = s.bonds[2 * b + (1 - side)];
        const double dx = wrap(xi - x[3 * o], box[0]), dy = wrap(yi - x[3 * o + 1], box[1]), dz = wrap(zi - x[3 * o + 2], box[2]);
        const double r = std::sqrt(dx * dx + dy * dy + dz * dz), d = r - s.br0[b];
        const double c = s.bk[b] * d / r;
        gx += c * dx, gy += c * dy, gz += c * dz;
        if (side == 0) e_b += 0.5 * s.bk[b] * d * d;
      }
      for (int t = s.ba_ptr[i]; t < s.ba_ptr[i + 1]; ++t) {
        const int a = s.ba[t] >> 2, role = s.ba[t] & 3;
        const int pi = s.angles[3 * a], pj = s.angles[3 * a + 1], pk = s.angles[3 * a + 2];
        double u[3], v[3];
        for (int k = 0; k < 3; ++k) u[k] = wrap(x[3 * pi + k] - x[3 * pj + k], box[k]), v[k] = wrap(x[3 * pk + k] - x[3 * pj + k], box[k]);
        const double u2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2], v2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const double uv = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
        const double iu = 1.0 / std::sqrt(u2), iv = 1.0 / std::sqrt(v2), c = uv * iu * iv;
        double dEdc, en;
        if (s.angle_kind == 0) {
          const double d = c - std::cos(s.at0[a]);
          dEdc = s.ak[a] * d, en = 0.5 * s.ak[a] * d * d;
        } else {
          const double cr[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
          const double sn = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) * iu * iv;
          const double d = std::atan2(sn, c) - s.at0[a];
          dEdc = sn > 1e-12 ? -s.ak[a] * d / sn : s.ak[a], en = 0.5 * s.ak[a] * d * d;
        }
        if (role == 0) e_a += en;
        for (int k = 0; k < 3; ++k) {
          const double du = (v[k] * iv - c * u[k] * iu) * iu, dv = (u[k] * iu - c * v[k] * iv) * iv;
          const double gk = role == 0 ? du : (role == 2 ? dv : -(du + dv));
          (k == 0 ? gx : (k == 1 ? gy : gz)) += dEdc * gk;
        }
      }
      g[3 * i] = gx, g[3 * i + 1] = gy, g[3 * i + 2] = gz;
      et[0] += e_lj, et[1] += e_b, et[2] += e_a;
    }
  }
  for (int k = 0; k < 3; ++k) {
    double t = 0;
    for (int r = 0; r < nt; ++r) t += e_thr[(size_t)r * 3 + k];
    e3[k] = t;
  }
}

}  // namespace

extern "C" {

void* mythos_cpu_martini_create(int n, const int32_t* types, int n_types, const double* sigma, const double* eps, int n_bonds,
                                const int32_t* bonds, const double* bk, const double* br0, int n_angles, const int32_t* angles,
                                const double* ak, const double* at0, int angle_kind, double r_cut, const double* mass) {
  if (n <= 0 || !types || n_types <= 0 || !sigma || !eps) return nullptr;
  auto* s = new MSys();
  s->n = n, s->n_types = n_types, s->angle_kind = angle_kind, s->r_cut = r_cut;
  s->types.assign(types, types + n);
  s->sigma.assign(sigma, sigma + (size_t)n_types * n_types);
  s->eps.assign(eps, eps + (size_t)n_types * n_types);
  s->bonds.assign(bonds, bonds + 2 * (size_t)n_bonds);
  s->bk.assign(bk, bk + n_bonds), s->br0.assign(br0, br0 + n_bonds);
  s->angles.assign(angles, angles + 3 * (size_t)n_angles);
  s->ak.assign(ak, ak + n_angles), s->at0.assign(at0, at0 + n_angles);
  s->inv_mass.resize(n);
  for (int i = 0; i < n; ++i) s->inv_mass[i] = 1.0 / (mass ? mass[i] : 72.0);
  std::vector<std::vector<int>> ex(n), bb(n), ba(n);
  for (int b = 0; b < n_bonds; ++b) {
    const int i = bonds[2 * b], j = bonds[2 * b + 1];
    ex[i].push_back(j), ex[j].push_back(i);
    bb[i].push_back(b << 1), bb[j].push_back((b << 1) | 1);
  }
  for (int a = 0; a < n_angles; ++a)
    for (int r = 0; r < 3; ++r) ba[angles[3 * a + r]].push_back((a << 2) | r);
  auto csr = [n](const std::vector<std::vector<int>>& v, std::vector<int>& ptr, std::vector<int>& out) {
    ptr.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + (int)v[i].size();
    out.clear();
    for (int i = 0; i < n; ++i) out.insert(out.end(), v[i].begin(), v[i].end());
  };
  csr(ex, s->excl_ptr, s->excl), csr(bb, s->bb_ptr, s->bb), csr(ba, s->ba_ptr, s->ba);
  return s;
}

void mythos_cpu_martini_destroy(void* h) { delete static_cast<MSys*>(h); }

// [lj, bond, angle] and dU/dx at x, on the list of range r_cut built here
void mythos_cpu_martini_energy(void* h, const double* x, const double* box, double* e3, double* g) {
  MSys& s = *static_cast<MSys*>(h);
  build_rows(s, x, box, s.r_cut);
  forces(s, x, box, g, e3);
}

// n_steps of BAOAB Langevin in place; returns the number of list builds; e4 = lj, bond, angle, kinetic of the final state
int mythos_cpu_martini_run(void* h, double* x, double* v, const double* box, int n_steps, double dt, double kT, double gamma,
                           uint64_t seed, int64_t step0, double skin, int rebuild_every, double* e4) {
  MSys& s = *static_cast<MSys*>(h);
  const int n = s.n;
  const double hdt = 0.5 * dt, c1 = std::exp(-gamma * dt);
  std::vector<double> g(3 * (size_t)n);
  double e3[3];
  int builds = 0;
  build_rows(s, x, box, s.r_cut + skin), ++builds;
  forces(s, x, box, g.data(), e3);
  for (int k = 0; k < n_steps; ++k) {
    if (k > 0 && rebuild_every > 0 && k % rebuild_every == 0) build_rows(s, x, box, s.r_cut + skin), ++builds;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      double z[6];
      normals6(seed, (uint32_t)i, (uint64_t)(step0 + k), 0u, z);
      const double im = s.inv_mass[i], c2 = std::sqrt(kT * im * (1.0 - c1 * c1));
      for (int a = 0; a < 3; ++a) {
        double vv = v[3 * i + a] - hdt * im * g[3 * i + a];
        double xx = x[3 * i + a] + hdt * vv;
        vv = c1 * vv + c2 * z[a];
        xx += hdt * vv;
        v[3 * i + a] = vv, x[3 * i + a] = xx;
      }
    }
    forces(s, x, box, g.data(), e3);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) v[3 * i + a] -= hdt * s.inv_mass[i] * g[3 * i + a];
  }
  if (e4) {
    double ke = 0;
    for (int i = 0; i < n; ++i) ke += 0.5 * (v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]) / s.inv_mass[i];
    e4[0] = e3[0], e4[1] = e3[1], e4[2] = e3[2], e4[3] = ke;
  }
  return builds;
}

}  // extern "C"
