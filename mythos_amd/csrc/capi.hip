// C-ABI entry points of libmythos_hip.so (declared in include/mythos_hip.h).
#include <atomic>
#include <cmath>
#include <cstring>

#include "mythos_internal.h"
#include "observables.h"

namespace mythos {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
  return MYTHOS_ERR_HIP;
}

static std::atomic<long long> g_debug[MYTHOS_DEBUG_KEYS];

long long debug_value(int key) { return (key >= 0 && key < MYTHOS_DEBUG_KEYS) ? g_debug[key].load(std::memory_order_relaxed) : 0; }
void debug_clear(int key) {
  if (key >= 0 && key < MYTHOS_DEBUG_KEYS) g_debug[key].store(0, std::memory_order_relaxed);
}

static const char* const kParamNames[] = {
#define OXP(name) #name,
#include "oxdna_param_list.inc"
#undef OXP
};

}  // namespace mythos

using namespace mythos;

extern "C" {

const char* mythos_version(void) { return "mythos_amd 0.1 (gfx950)"; }

const char* mythos_last_error(void) { return g_last_error.c_str(); }

int mythos_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mythos_debug_set(int key, int64_t value) {
  if (key < 0 || key >= MYTHOS_DEBUG_KEYS || value < 0) {
    set_error("mythos_debug_set: unknown key or negative value");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  g_debug[key].store((long long)value, std::memory_order_relaxed);
  return MYTHOS_OK;
}

int64_t mythos_debug_get(int key) { return (int64_t)debug_value(key); }

int mythos_oxdna_param_count(void) { return (int)OXP_COUNT; }

const char* mythos_oxdna_param_name(int index) {
  if (index < 0 || index >= (int)OXP_COUNT) return nullptr;
  return kParamNames[index];
}

mythos_system_t* mythos_oxdna_create(int model, int n, const int32_t* seq, const uint8_t* is_end, int n_bonded,
                                     const int32_t* bonded, const double* box, int dtype, int device) {
  g_last_error.clear();
  if ((model < 1 || model > 4) || n < 1 || !seq || n_bonded < 0 || (n_bonded > 0 && !bonded) ||
      (dtype != MYTHOS_F32 && dtype != MYTHOS_F64) || n >= ROW_ROLE_Q) {
    set_error("mythos_oxdna_create: invalid argument");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("mythos_oxdna_create: no HIP device visible (the HIP path has no CPU fallback)");
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    set_error("mythos_oxdna_create: device index out of range");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("mythos_oxdna_create: hipSetDevice failed");
    return nullptr;
  }
  auto* s = new mythos_system();
  s->model = model;
  s->n = n;
  s->dtype = dtype;
  s->device = device;
  s->n_bonded = n_bonded;
  if (box) {
    s->has_box = true;
    for (int k = 0; k < 3; ++k) s->box[k] = box[k];
    if (!(box[0] > 0 && box[1] > 0 && box[2] > 0)) {
      set_error("mythos_oxdna_create: box lengths must be positive");
      delete s;
      return nullptr;
    }
  }
  std::vector<int> meta(n);
  for (int i = 0; i < n; ++i) {
    if (seq[i] < 0 || seq[i] > 3) {
      set_error("mythos_oxdna_create: sequence entries must be 0..3");
      delete s;
      return nullptr;
    }
    meta[i] = seq[i] | ((is_end && is_end[i]) ? 4 : 0);
  }
  s->h_meta = meta;
  s->h_partners.assign((size_t)ROW_BONDED_SLOTS * n, -1);
  for (int b = 0; b < n_bonded; ++b) {
    const int i = bonded[2 * b], j = bonded[2 * b + 1];
    if (i < 0 || j < 0 || i >= n || j >= n || i == j) {
      set_error("mythos_oxdna_create: bonded index out of range");
      delete s;
      return nullptr;
    }
    int* pi = &s->h_partners[(size_t)ROW_BONDED_SLOTS * i];
    int* pj = &s->h_partners[(size_t)ROW_BONDED_SLOTS * j];
    const int si = pi[1] == -1 ? 1 : 3;  // i plays nn_i: odd slots
    const int sj = pj[0] == -1 ? 0 : 2;  // j plays nn_j: even slots
    if (pi[si] != -1 || pj[sj] != -1) {
      set_error("mythos_oxdna_create: a nucleotide has more than two bonded partners in one role");
      delete s;
      return nullptr;
    }
    pi[si] = j;
    pj[sj] = i;
    if (si == 3 || sj == 2) s->extra_bonds = true;
  }
  bool ok = hipMalloc((void**)&s->d_meta, n * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_row_len, (size_t)(2 + ROW_BONDED_SLOTS) * n * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&s->d_overflow, kOverflowWords * sizeof(int)) == hipSuccess &&
            hipMemcpy(s->d_meta, meta.data(), n * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(s->d_row_len + n, s->h_partners.data(), (size_t)ROW_BONDED_SLOTS * n * sizeof(int), hipMemcpyHostToDevice) ==
                hipSuccess &&
            hipMemset(s->d_overflow, 0, kOverflowWords * sizeof(int)) == hipSuccess;
  if (!ok) {
    set_error("mythos_oxdna_create: device allocation failed");
    mythos_oxdna_destroy(s);
    return nullptr;
  }
  return s;
}

void mythos_oxdna_destroy(mythos_system_t* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->d_meta) (void)hipFree(s->d_meta);
  if (s->d_rows) (void)hipFree(s->d_rows);
  if (s->d_row_len) (void)hipFree(s->d_row_len);
  if (s->d_overflow) (void)hipFree(s->d_overflow);
  if (s->d_cell) (void)hipFree(s->d_cell);
  if (s->d_ref_pos) (void)hipFree(s->d_ref_pos);
  if (s->d_ref_off) (void)hipFree(s->d_ref_off);
  if (s->d_ref_a1) (void)hipFree(s->d_ref_a1);
  if (s->d_pf) (void)hipFree(s->d_pf);
  if (s->d_pd) (void)hipFree(s->d_pd);
  if (s->d_epart) (void)hipFree(s->d_epart);
  if (s->d_pgpart) (void)hipFree(s->d_pgpart);
  if (s->d_ps_marg) (void)hipFree(s->d_ps_marg);
  if (s->d_ps_unit) (void)hipFree(s->d_ps_unit);
  if (s->d_ps_bp) (void)hipFree(s->d_ps_bp);
  delete s;
}

int mythos_oxdna_set_params(mythos_system_t* s, const double* flat, int n_params) {
  const int sets = s ? s->param_sets() : 1;
  const int total = sets * (int)OXP_COUNT;
  if (!s || !flat || n_params != total) {
    set_error("mythos_oxdna_set_params: expected " + std::to_string(total) + " parameters" +
              (sets == 3 ? " (oxNA: the oxDNA2, oxRNA2 and hybrid vectors one after the other)" : ""));
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  for (int k = 0; k < n_params; ++k)
    if (!std::isfinite(flat[k])) {
      set_error(std::string("mythos_oxdna_set_params: non-finite value for ") + kParamNames[k % (int)OXP_COUNT]);
      return MYTHOS_ERR_NUMERIC;
    }
  for (int k = 0; k < (int)OXP_COUNT; ++k) {
    s->pd.v[k] = flat[k];
    s->pf.v[k] = (float)flat[k];
  }
  s->pd_sets.assign(flat, flat + total);
  std::vector<float> pf_sets(flat, flat + total);
  // device copies for the MD kernel; a blocking copy after a device-wide sync, so no kernel in flight
  // sees a half-written vector
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  if (!s->d_pf) MYTHOS_HIP_TRY(hipMalloc((void**)&s->d_pf, total * sizeof(float)));
  if (!s->d_pd) MYTHOS_HIP_TRY(hipMalloc((void**)&s->d_pd, total * sizeof(double)));
  MYTHOS_HIP_TRY(hipDeviceSynchronize());
  MYTHOS_HIP_TRY(hipMemcpy(s->d_pf, pf_sets.data(), total * sizeof(float), hipMemcpyHostToDevice));
  MYTHOS_HIP_TRY(hipMemcpy(s->d_pd, s->pd_sets.data(), total * sizeof(double), hipMemcpyHostToDevice));
  s->params_set = true;
  ++s->list_epoch;  // cut-offs may have moved: integrators rebuild their list
  ++s->param_epoch; // ... and re-derive the site offsets their resident frames carry
  return MYTHOS_OK;
}

int mythos_oxdna_set_nucleotide_types(mythos_system_t* s, const uint8_t* is_rna) {
  if (!s || !is_rna) {
    set_error("mythos_oxdna_set_nucleotide_types: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (s->model != 4) {
    set_error("mythos_oxdna_set_nucleotide_types: only an oxNA system (model 4) has nucleotide types");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < s->n; ++i) s->h_meta[i] = (s->h_meta[i] & 7) | (is_rna[i] ? 8 : 0);
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  MYTHOS_HIP_TRY(hipDeviceSynchronize());
  MYTHOS_HIP_TRY(hipMemcpy(s->d_meta, s->h_meta.data(), s->n * sizeof(int), hipMemcpyHostToDevice));
  s->types_set = true;
  ++s->list_epoch;
  ++s->param_epoch;
  return MYTHOS_OK;
}

int mythos_oxdna_set_pseq(mythos_system_t* s, const double* marginals, const int32_t* unit, int n_bp, const double* bp_probs,
                          int terms) {
  if (!s || terms < 0 || terms > 3 || n_bp < 0 || (terms != 0 && (!marginals || !unit || (n_bp > 0 && !bp_probs)))) {
    set_error("mythos_oxdna_set_pseq: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (s->model == 4 && (terms & 1) != 0) {
    set_error("mythos_oxdna_set_pseq: an oxNA system (model 4) takes a probabilistic sequence for hydrogen bonding only "
              "(terms = 2), as the reference's na1 terms do");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  MYTHOS_HIP_TRY(hipDeviceSynchronize());  // no kernel in flight reads a half-written table
  if (terms == 0) {
    s->pseq_terms = 0;
    ++s->list_epoch;  // integrators drop forces cached from the old distribution (the unfused oxNA path keeps some)
    return MYTHOS_OK;
  }
  const int n = s->n;
  for (int i = 0; i < n; ++i) {
    const int u = unit[i];
    if (u < -1 || u >= 2 * n_bp) {
      set_error("mythos_oxdna_set_pseq: unit of nucleotide " + std::to_string(i) + " is outside the " + std::to_string(n_bp) + " base pairs");
      return MYTHOS_ERR_INVALID_ARGUMENT;
    }
  }
  for (size_t k = 0; k < 4 * (size_t)n; ++k)
    if (!std::isfinite(marginals[k]) || marginals[k] < 0.0) {
      set_error("mythos_oxdna_set_pseq: non-finite or negative probability");
      return MYTHOS_ERR_NUMERIC;
    }
  for (size_t k = 0; k < 4 * (size_t)n_bp; ++k)
    if (!std::isfinite(bp_probs[k]) || bp_probs[k] < 0.0) {
      set_error("mythos_oxdna_set_pseq: non-finite or negative base-pair type probability");
      return MYTHOS_ERR_NUMERIC;
    }
  {
    // every constrained base pair has exactly one member 0 and one member 1 (the kernels split a pair's contribution
    // half and half between its two visits), and a member's marginal is what its pair's type distribution implies:
    // types AT, TA, GC, CG give member 0 the bases A, T, G, C and member 1 the bases T, A, C, G (A, C, G, T = 0 .. 3)
    std::vector<int> owner(2 * (size_t)n_bp, -1);
    for (int i = 0; i < n; ++i) {
      const int u = unit[i];
      if (u < 0) continue;
      if (owner[u] >= 0) {
        set_error("mythos_oxdna_set_pseq: nucleotides " + std::to_string(owner[u]) + " and " + std::to_string(i) +
                  " claim the same place of base pair " + std::to_string(u >> 1));
        return MYTHOS_ERR_INVALID_ARGUMENT;
      }
      owner[u] = i;
    }
    static const int kBase[2][4] = {{0, 3, 2, 1}, {3, 0, 1, 2}};
    for (int u = 0; u < 2 * n_bp; ++u) {
      if (owner[u] < 0) {
        set_error("mythos_oxdna_set_pseq: base pair " + std::to_string(u >> 1) + " has no member " + std::to_string(u & 1));
        return MYTHOS_ERR_INVALID_ARGUMENT;
      }
      double implied[4] = {0, 0, 0, 0};
      for (int t = 0; t < 4; ++t) implied[kBase[u & 1][t]] += bp_probs[4 * (size_t)(u >> 1) + t];
      for (int b = 0; b < 4; ++b)
        if (std::fabs(implied[b] - marginals[4 * (size_t)owner[u] + b]) > 1e-6) {
          set_error("mythos_oxdna_set_pseq: the marginal of nucleotide " + std::to_string(owner[u]) +
                    " is not the one its base pair's type probabilities imply");
          return MYTHOS_ERR_INVALID_ARGUMENT;
        }
    }
  }
  const size_t word = s->dtype == MYTHOS_F32 ? sizeof(float) : sizeof(double);
  const int nb = std::max(n_bp, 1);
  if (!s->d_ps_marg) MYTHOS_HIP_TRY(hipMalloc(&s->d_ps_marg, 4 * (size_t)n * word));
  if (!s->d_ps_unit) MYTHOS_HIP_TRY(hipMalloc((void**)&s->d_ps_unit, (size_t)n * sizeof(int)));
  if (nb > s->ps_bp_cap) {
    if (s->d_ps_bp) (void)hipFree(s->d_ps_bp);
    s->d_ps_bp = nullptr;
    s->ps_bp_cap = 0;
    MYTHOS_HIP_TRY(hipMalloc(&s->d_ps_bp, 4 * (size_t)nb * word));
    s->ps_bp_cap = nb;
  }
  std::vector<double> bp(4 * (size_t)nb, 0.0);
  if (n_bp > 0) std::copy(bp_probs, bp_probs + 4 * (size_t)n_bp, bp.begin());
  if (s->dtype == MYTHOS_F32) {
    std::vector<float> mf(marginals, marginals + 4 * (size_t)n), bf(bp.begin(), bp.end());
    MYTHOS_HIP_TRY(hipMemcpy(s->d_ps_marg, mf.data(), mf.size() * sizeof(float), hipMemcpyHostToDevice));
    MYTHOS_HIP_TRY(hipMemcpy(s->d_ps_bp, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
  } else {
    MYTHOS_HIP_TRY(hipMemcpy(s->d_ps_marg, marginals, 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    MYTHOS_HIP_TRY(hipMemcpy(s->d_ps_bp, bp.data(), bp.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  MYTHOS_HIP_TRY(hipMemcpy(s->d_ps_unit, unit, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  s->pseq_terms = terms;
  s->ps_n_bp = n_bp;
  ++s->list_epoch;
  return MYTHOS_OK;
}

int mythos_oxdna_set_neighbors(mythos_system_t* s, const int32_t* pairs, int n_pairs) {
  if (!s || n_pairs < 0 || (n_pairs > 0 && !pairs)) {
    set_error("mythos_oxdna_set_neighbors: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  ++s->list_epoch;
  return rows_from_pairs(s, pairs, n_pairs);
}

int mythos_oxdna_build_neighbors(mythos_system_t* s, const void* center, double r_cut, double skin,
                                 mythos_stream_t stream) {
  if (!s || !center || !(r_cut > 0) || skin < 0) {
    set_error("mythos_oxdna_build_neighbors: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  ++s->list_epoch;
  return rows_build_until_fit(s, center, false, r_cut, skin, nullptr, nullptr, false, false, st);
}

int mythos_oxdna_neighbor_stats(mythos_system_t* s, int* max_row, double* mean_row) {
  if (!s || !s->nbrs_set) {
    set_error("mythos_oxdna_neighbor_stats: no neighbour list");
    return MYTHOS_ERR_NOT_READY;
  }
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  std::vector<int> len(s->n);
  MYTHOS_HIP_TRY(hipMemcpy(len.data(), s->d_row_len, s->n * sizeof(int), hipMemcpyDeviceToHost));
  long long tot = 0;
  int mx = 0;
  for (int v : len) {
    tot += v - ROW_BONDED_SLOTS;
    mx = std::max(mx, v - ROW_BONDED_SLOTS);
  }
  if (max_row) *max_row = mx;
  if (mean_row) *mean_row = double(tot) / s->n;
  return MYTHOS_OK;
}

int mythos_oxdna_energy(mythos_system_t* s, const void* center, const void* quat, int n_frames, double* e_terms,
                        void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_stream_t stream) {
  if (!s || n_frames < 0 || (n_frames > 0 && (!center || !quat || !e_terms))) {
    set_error("mythos_oxdna_energy: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->params_set || !s->nbrs_set || (s->model == 4 && !s->types_set)) {
    set_error("mythos_oxdna_energy: parameters and neighbours (and, for oxNA, nucleotide types) must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  if (n_frames == 0) return MYTHOS_OK;  // an empty batch (its buffers may be null) is not an error
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  return oxdna_energy_launch(s, center, quat, n_frames, e_terms, dU_dcenter, dU_dquat, dU_dparams, nullptr, nullptr,
                             (hipStream_t)stream);
}

int mythos_oxdna_energy_dpseq(mythos_system_t* s, const void* center, const void* quat, int n_frames, double* e_terms,
                              void* dU_dcenter, void* dU_dquat, double* dU_dparams, double* dU_dmarginals, double* dU_dbp,
                              mythos_stream_t stream) {
  if (!s || n_frames < 0 || (n_frames > 0 && (!center || !quat || !e_terms || !dU_dparams || !dU_dmarginals || !dU_dbp))) {
    set_error("mythos_oxdna_energy_dpseq: invalid argument (dU_dparams and both distribution-gradient buffers are required)");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->params_set || !s->nbrs_set || s->pseq_terms == 0 || (s->model == 4 && !s->types_set)) {
    set_error("mythos_oxdna_energy_dpseq: parameters, neighbours and a probabilistic sequence (mythos_oxdna_set_pseq) must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  if (n_frames == 0) return MYTHOS_OK;
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  s->ps_gmarg = dU_dmarginals, s->ps_gbp = dU_dbp;
  const int rc = oxdna_energy_launch(s, center, quat, n_frames, e_terms, dU_dcenter, dU_dquat, dU_dparams, nullptr, nullptr,
                                     (hipStream_t)stream);
  s->ps_gmarg = s->ps_gbp = nullptr;
  return rc;
}

int mythos_oxdna_energy_obs(mythos_system_t* s, const void* center, const void* quat, int n_frames, double* e_terms,
                            void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_obs_t* obs, double* obs_out,
                            mythos_stream_t stream) {
  if (!s || n_frames < 0 || (n_frames > 0 && (!center || !quat || !e_terms)) || (obs && n_frames > 0 && !obs_out)) {
    set_error("mythos_oxdna_energy_obs: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (obs && (obs->n != s->n || obs->dtype != s->dtype || obs->device != s->device)) {
    set_error("mythos_oxdna_energy_obs: the observable set was made for another system size, precision or device");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (!s->params_set || !s->nbrs_set || (s->model == 4 && !s->types_set)) {
    set_error("mythos_oxdna_energy_obs: parameters and neighbours (and, for oxNA, nucleotide types) must be set first");
    return MYTHOS_ERR_NOT_READY;
  }
  if (obs && s->model == 4) {
    set_error("mythos_oxdna_energy_obs: the structural observables take one site geometry; an oxNA system has two");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (n_frames == 0) return MYTHOS_OK;
  MYTHOS_HIP_TRY(hipSetDevice(s->device));
  return oxdna_energy_launch(s, center, quat, n_frames, e_terms, dU_dcenter, dU_dquat, dU_dparams, obs, obs_out,
                             (hipStream_t)stream);
}

}  // extern "C"
