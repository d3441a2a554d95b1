// Shared by martini.hip (energy path) and martini_md.hip (Langevin MD): limits, constants, the system struct.
#ifndef MYTHOS_MARTINI_INTERNAL_H
#define MYTHOS_MARTINI_INTERNAL_H

#include "mythos_internal.h"

namespace mythos {

constexpr int kLjBlock = 256;
constexpr int kMaxExcl = 8;
constexpr int kMaxBeadBonds = 8;
constexpr int kMaxBeadAngles = 12;
constexpr int kMaxTypes = 64;

template <typename R>
struct MartiniConst {
  R rc2;
  int n_types;
  int angle_kind;  // 0 = G96 cosine, 1 = harmonic
};

template <typename R>
__device__ __forceinline__ R wrap(R d, R l, R il) {
  return d - l * m_rint(d * il);
}

}  // namespace mythos

struct mythos_martini {
  int n = 0, n_types = 0, n_bonds = 0, n_angles = 0, angle_kind = 0, dtype = 0, device = 0;
  double r_cut = 1.1;
  int *d_types = nullptr, *d_excl = nullptr, *d_bead_bonds = nullptr, *d_bead_angles = nullptr, *d_bonds = nullptr,
      *d_angles = nullptr;
  void *d_sigma = nullptr, *d_eps = nullptr, *d_bond_k = nullptr, *d_bond_r0 = nullptr, *d_angle_k = nullptr,
       *d_angle_t0 = nullptr;
  void* d_fpart = nullptr;
  size_t fpart_cap = 0;
  double *d_epart = nullptr, *d_ebpart = nullptr, *d_ljpart = nullptr;  // d_ljpart: per-workgroup dU/dsigma | dU/deps tables
  size_t epart_cap = 0, ebpart_cap = 0, ljpart_cap = 0;
};

#endif  // MYTHOS_MARTINI_INTERNAL_H
