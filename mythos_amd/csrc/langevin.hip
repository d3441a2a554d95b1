// Langevin integrator, translation unit 1 of 2: the fp32 instantiations of langevin_core.inc and the C entry points
// (mythos_langevin_*).  The fp64 instantiations are in langevin_f64.hip.
#include "langevin_core.inc"

MYTHOS_MD_DEFINE_PRECISION(float)

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
  return s->sys->dtype == MYTHOS_F32 ? mythos_md_init_momenta<float>(s, p_lin, p_ang, (hipStream_t)stream)
                                     : mythos_md_init_momenta<double>(s, p_lin, p_ang, (hipStream_t)stream);
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
  s->unfused = s->want_unfused && s->sys->model == 4;
  return s->sys->dtype == MYTHOS_F32 ? mythos_md_load<float>(s, c, q, p, l, st) : mythos_md_load<double>(s, c, q, p, l, st);
}

int md_advance(mythos_sim_t* s, int n_steps, int save_every, bool close, void* tc, void* tq, double* e_trace, hipStream_t st) {
  // quaternion rows are written as one 4-vector per nucleotide
  const size_t q_align = s->sys->dtype == MYTHOS_F32 ? sizeof(float4) : sizeof(double4);
  if (save_every > 0 && tq && (reinterpret_cast<uintptr_t>(tq) % q_align) != 0) {
    set_error("mythos_langevin_run / advance: traj_quat must be aligned to 4 elements (" + std::to_string(q_align) + " bytes)");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  return s->sys->dtype == MYTHOS_F32 ? mythos_md_advance<float>(s, n_steps, save_every, close, tc, tq, e_trace, st)
                                     : mythos_md_advance<double>(s, n_steps, save_every, close, tc, tq, e_trace, st);
}

int md_store(mythos_sim_t* s, void* c, void* q, void* p, void* l, hipStream_t st) {
  return s->sys->dtype == MYTHOS_F32 ? mythos_md_store<float>(s, c, q, p, l, st) : mythos_md_store<double>(s, c, q, p, l, st);
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
  const int rc = md_advance(s, n_steps, save_every, true, traj_center, traj_quat, e_trace, st);
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
  return md_advance(s, n_steps, save_every, false, traj_center, traj_quat, e_trace, (hipStream_t)stream);
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
  if (s->open) {  // one more launch closes the frame: what a launch needs has to be there (it was, for the advance before)
    if (int rc = md_ready(s, "mythos_langevin_store")) return rc;
  } else {  // a copy: load; store round-trips a state whether or not neighbours were ever set
    MYTHOS_HIP_TRY(hipSetDevice(s->sys->device));
  }
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

int mythos_langevin_set_seed(mythos_sim_t* s, uint64_t seed) {
  if (!s) {
    set_error("mythos_langevin_set_seed: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  s->seed = seed;
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
