// Internal definitions shared by the translation units of libmythos_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mythos_hip.h"
#include "oxdna_math.h"

namespace mythos {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);
// test / diagnostic switches (mythos_debug_set): process-wide, 0 = default
long long debug_value(int key);
void debug_clear(int key);

#define MYTHOS_HIP_TRY(expr)                                   \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) return ::mythos::hip_fail(_e, #expr); \
  } while (0)

// Row entry encoding of the per-nucleotide neighbour rows:
//   slot 0 : bonded partner on the 3' side  (bond (j, self): self plays nn_j)   or -1
//   slot 1 : bonded partner on the 5' side  (bond (self, j): self plays nn_i)   or -1
//   slot 2, 3: a second partner in the nn_j / nn_i role, or -1.  Only the two ends of a circular strand have one:
//           the reference closes a ring with the pair (first, last) in that order (mythos/input/topology.py:178-180),
//           so `first` is nn_i of two bonds and `last` is nn_j of two.  Even slots: self is nn_j; odd: self is nn_i.
//   slot>=4: unbonded neighbour index | ROLE_Q if self plays op_j of the ordered pair
constexpr int ROW_ROLE_Q = 1 << 30;
constexpr int ROW_INDEX_MASK = ROW_ROLE_Q - 1;
constexpr int ROW_BONDED_SLOTS = 4;

template <typename R>
struct BoxT {
  R l[3];
  R il[3];
  int on;
};

}  // namespace mythos

struct mythos_obs;

struct mythos_system {
  int model = 0;
  int n = 0;
  int dtype = 0;
  int device = 0;
  int n_bonded = 0;
  bool has_box = false;
  double box[3] = {0, 0, 0};

  // topology (device)
  int* d_meta = nullptr;  // [n] seq | is_end << 2

  // neighbour rows (device)
  int* d_rows = nullptr;     // [n][row_stride]
  int* d_row_len = nullptr;  // [n] used slots (>= 2) | [2n] bonded partners | [n] end of the "close" segment
  int row_stride = 0;
  size_t rows_cap = 0;  // allocated ints in d_rows
  bool nbrs_set = false;
  int list_epoch = 0;  // bumped when parameters or rows are replaced through the ABI (integrators re-validate their list)
  int param_epoch = 0; // bumped when parameters or nucleotide types are replaced (integrators re-derive the site offsets they carry)
  // host copy of the bonded partners, [n][2]
  std::vector<int> h_partners;
  bool extra_bonds = false;  // some nucleotide uses slot 2 or 3 (circular strands)

  // Verlet build scratch
  int* d_overflow = nullptr;  // [kOverflowWords] largest demands seen since last cleared (see rows_build_until_fit)
  int* d_cell = nullptr;      // cell table (cell_list.h CellBins): counters [2][H], buckets [H][cell_bucket_cap]
  size_t cell_cap = 0;        // ints allocated at d_cell
  int cell_H = 0;             // table slots of the current allocation
  int cell_alloc_bucket_cap = 0;  // places per slot the allocation was laid out for
  bool cell_sites = false;    // the allocation carries the two site streams (cell_list.h CellBins::sites)
  int cell_bucket_cap = 32;   // places per slot
  int cell_phase = 0;         // which counter half the next build counts into
  void* d_ref_pos = nullptr;  // [n] real4 positions at the last build (MD displacement check)
  void* d_ref_off = nullptr;  // [n] real4 backbone offsets at the last build
  void* d_ref_a1 = nullptr;   // [n] real4 base vectors at the last build

  // parameters
  bool params_set = false;
  mythos::OxParams<float> pf;
  mythos::OxParams<double> pd;
  // oxNA (model 4): three vectors - oxDNA2 (also in pf / pd), oxRNA2, hybrid - one after the other, host and device
  std::vector<double> pd_sets;
  int param_sets() const { return model == 4 ? 3 : 1; }
  bool types_set = false;  // model 4: mythos_oxdna_set_nucleotide_types has run
  std::vector<int> h_meta;  // seq | is_end << 2 | is_rna << 3, as uploaded to d_meta
  float* d_pf = nullptr;   // the same vectors in device memory (read by the MD kernel as scalar loads)
  double* d_pd = nullptr;

  // probabilistic sequence (mythos_oxdna_set_pseq); pseq_terms == 0: discrete sequence
  void* d_ps_marg = nullptr;  // [n][4] real: marginal base probabilities
  int* d_ps_unit = nullptr;   // [n] 2 * base pair + member, or -1
  void* d_ps_bp = nullptr;    // [max(n_bp, 1)][4] real: base-pair type probabilities
  int ps_bp_cap = 0;
  int pseq_terms = 0;         // bit 0 stacking, bit 1 hydrogen bonding
  int ps_n_bp = 0;            // constrained base pairs of the distribution
  double* ps_gmarg = nullptr; // caller's dU/d(marginals) and dU/d(type probabilities) buffers, set for the duration of
  double* ps_gbp = nullptr;   // one mythos_oxdna_energy_dpseq call

  // energy-pass scratch
  double* d_epart = nullptr;  // [frames_chunk][blocks][8]
  size_t epart_cap = 0;
  double* d_pgpart = nullptr;  // [frames_chunk][blocks][OXP_COUNT]
  size_t pgpart_cap = 0;
};

namespace mythos {

template <typename R>
BoxT<R> make_box(const mythos_system* s) {
  BoxT<R> b;
  b.on = s->has_box ? 1 : 0;
  for (int k = 0; k < 3; ++k) {
    b.l[k] = R(s->has_box ? s->box[k] : 1.0);
    b.il[k] = R(s->has_box ? 1.0 / s->box[k] : 1.0);
  }
  return b;
}

template <typename R>
const OxParams<R>& params_of(const mythos_system* s);
template <>
inline const OxParams<float>& params_of<float>(const mythos_system* s) { return s->pf; }
template <>
inline const OxParams<double>& params_of<double>(const mythos_system* s) { return s->pd; }
template <typename R>
const R* device_params_of(const mythos_system* s);
template <>
inline const float* device_params_of<float>(const mythos_system* s) { return s->d_pf; }
template <>
inline const double* device_params_of<double>(const mythos_system* s) { return s->d_pd; }

// d_row_len holds three arrays back to back: row length [n] | bonded partners [n][ROW_BONDED_SLOTS] | end of the
// "close" segment of the row [n]
static_assert(ROW_BONDED_SLOTS == 4, "the row builders copy four partner slots");
inline int* row_close_of(const mythos_system* sys) { return sys->d_row_len + (size_t)(1 + ROW_BONDED_SLOTS) * sys->n; }

// Largest centre-centre distance at which anything other than the backbone-backbone terms (excluded
// volume between base / backbone sites, H-bond, cross- and coaxial stacking) can act.  Rows keep the
// neighbours inside this range (+ skin) in a leading "close" segment so the MD kernel's radial pass
// runs its heavy and its Debye-only code on homogeneous wavefronts.
// One flat vector of the system: the only one, or (oxNA) vector k of oxDNA2, oxRNA2, hybrid.
inline const double* oxdna_param_set(const mythos_system* sys, int k) {
  return sys->param_sets() == 1 ? sys->pd.v : sys->pd_sets.data() + (size_t)k * OXP_COUNT;
}
// max over the system's vectors of one entry, or of a function of a vector
inline double oxdna_param_max(const mythos_system* sys, int idx) {
  double m = oxdna_param_set(sys, 0)[idx];
  for (int k = 1; k < sys->param_sets(); ++k) m = std::max(m, oxdna_param_set(sys, k)[idx]);
  return m;
}

inline double oxdna_close_range(const mythos_system* sys) {
  // (oxNA: ranges from whichever vector is largest, site offsets from whichever geometry reaches farthest)
  double off_back = 0.0, off_base = 0.0, off_stack = 0.0;
  for (int k = 0; k < std::min(sys->param_sets(), 2); ++k) {  // the hybrid vector carries no geometry of its own
    const double* P = oxdna_param_set(sys, k);
    off_back = std::max(off_back, std::sqrt(P[GEO_BACK_A1] * P[GEO_BACK_A1] + (sys->model >= 2 ? P[GEO_BACK_A2] * P[GEO_BACK_A2] : 0.0)));
    off_base = std::max(off_base, std::fabs(P[GEO_BASE]));
    off_stack = std::max(off_stack, std::fabs(P[GEO_STACK]));
  }
  auto mx = [&](int idx) { return oxdna_param_max(sys, idx); };
  double rcom = std::max(mx(NEXC_BACK_BASE_RC), mx(NEXC_BASE_BACK_RC)) + off_back + off_base;
  rcom = std::max(rcom, mx(NEXC_BASE_RC) + 2 * off_base);
  rcom = std::max(rcom, std::max(mx(HYDR_RCHIGH), mx(CRST_RCHIGH)) + 2 * off_base);
  rcom = std::max(rcom, mx(CXST_RCHIGH) + 2 * off_stack);
  return rcom * (1.0 + 1e-6);
}

// oxdna_kernels.hip
int oxdna_energy_launch(mythos_system* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                        void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_obs* obs, double* obs_out,
                        hipStream_t stream);
// neighbors.hip
int rows_from_pairs(mythos_system* sys, const int32_t* pairs, int n_pairs);
// backbone_offsets, base_vectors: real4 per nucleotide (MD frames) to select the segments by site distances, or null
int rows_build_device(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                      const void* backbone_offsets, const void* base_vectors, bool write_refs, hipStream_t stream);
// write_refs: also store the positions / backbone offsets / base vectors the list was built from in
// d_ref_pos / d_ref_off / d_ref_a1 (the MD kernel's displacement check compares against them)
int rows_reserve(mythos_system* sys, int stride);
// d_overflow words: [0] longest row if over row_stride, [1] fullest bucket if over its capacity, [2] fullest
// bucket if over half its capacity (headroom hint, not an error)
constexpr int kOverflowWords = 3;
int rows_build_until_fit(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                         const void* backbone_offsets, const void* base_vectors, bool write_refs, bool headroom,
                         hipStream_t stream);

}  // namespace mythos
