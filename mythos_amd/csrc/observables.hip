// Observable sets behind the C ABI (mythos_observables_*): see observables.h for what is computed and where the
// reference defines it.  One workgroup per frame; mythos_oxdna_energy_obs (oxdna_kernels.hip) queues the same kernel behind its energy launch.
#include "observables.h"

#include "mythos_internal.h"

namespace mythos {

template <typename R>
__global__ __launch_bounds__(256) void observables_kernel(const ObsView v, int n, const R* __restrict__ center,
                                                          const R* __restrict__ quat, double* __restrict__ out) {
  __shared__ double red[4];
  const size_t f = blockIdx.x;
  frame_observables<R>(v, center + f * n * 3, quat + f * n * 4, out + f * v.width, v.axis + f * (size_t)v.n_q * 3, red);
}

int obs_view_for(mythos_obs* o, int n_frames, ObsView* out) {
  const size_t need = (size_t)std::max(n_frames, 1) * std::max(o->view.n_q, 1) * 3;
  if (need > o->axis_cap) {
    if (o->d_axis) (void)hipFree(o->d_axis);
    o->d_axis = nullptr;
    o->axis_cap = 0;
    MYTHOS_HIP_TRY(hipMalloc((void**)&o->d_axis, need * sizeof(double)));
    o->axis_cap = need;
  }
  o->view.axis = o->d_axis;
  *out = o->view;
  return 0;
}

int observables_launch(mythos_obs* o, const ObsView& v, const void* center, const void* quat, int n_frames, double* out,
                       hipStream_t st) {
  // one workgroup per frame; a frame whose lists fit one wavefront (the DiffTRe systems: 30 base pairs, 31 quartets) gets
  // a workgroup of one - its sums are the first wavefront's sums of the wider workgroup bit for bit (the other partials
  // are zeros), its barriers cost nothing, and four times as many frames are in flight
  const int threads = (v.n_bp <= 64 && v.n_q <= 64) ? 64 : 256;
  if (o->dtype == MYTHOS_F32)
    hipLaunchKernelGGL(observables_kernel<float>, dim3(n_frames), dim3(threads), 0, st, v, o->n, (const float*)center,
                       (const float*)quat, out);
  else
    hipLaunchKernelGGL(observables_kernel<double>, dim3(n_frames), dim3(threads), 0, st, v, o->n, (const double*)center,
                       (const double*)quat, out);
  MYTHOS_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace mythos

using namespace mythos;

extern "C" {

mythos_obs_t* mythos_observables_create(int model, int n, const double* geometry, const double* box, int n_bp,
                                        const int32_t* base_pairs, int n_quartets, const int32_t* quartets, int skip_ends,
                                        int dtype, int device) {
  if ((model < 1 || model > 3) || n <= 0 || !geometry || n_bp < 0 || n_quartets < 0 || (n_bp > 0 && !base_pairs) ||
      (n_quartets > 0 && !quartets) || (dtype != MYTHOS_F32 && dtype != MYTHOS_F64)) {
    set_error("mythos_observables_create: invalid argument");
    return nullptr;
  }
  for (int k = 0; k < 2 * n_bp; ++k)
    if (base_pairs[k] < 0 || base_pairs[k] >= n) {
      set_error("mythos_observables_create: base-pair index out of range");
      return nullptr;
    }
  for (int k = 0; k < 4 * n_quartets; ++k)
    if (quartets[k] < 0 || quartets[k] >= n) {
      set_error("mythos_observables_create: quartet index out of range");
      return nullptr;
    }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("mythos_observables_create: hipSetDevice failed");
    return nullptr;
  }
  auto* o = new mythos_obs();
  o->n = n, o->dtype = dtype, o->device = device;
  ObsView& v = o->view;
  v.n_bp = n_bp, v.n_q = n_quartets;
  v.skip = skip_ends ? 2 : 0;
  v.n_corr = std::max(0, n_quartets - 2 * v.skip);
  v.width = 4 + v.n_corr;
  v.model = model;
  v.g_hb = geometry[0], v.g_k1 = geometry[1], v.g_k2 = model >= 2 ? geometry[2] : 0.0;
  if (box) {
    v.box_on = 1;
    for (int k = 0; k < 3; ++k) v.box[k] = box[k];
  }
  bool ok = true;
  if (n_bp > 0) {
    ok = ok && hipMalloc((void**)&o->d_bps, 2 * (size_t)n_bp * sizeof(int)) == hipSuccess &&
         hipMemcpy(o->d_bps, base_pairs, 2 * (size_t)n_bp * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok && n_quartets > 0) {
    ok = hipMalloc((void**)&o->d_quartets, 4 * (size_t)n_quartets * sizeof(int)) == hipSuccess &&
         hipMemcpy(o->d_quartets, quartets, 4 * (size_t)n_quartets * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) {
    set_error("mythos_observables_create: device allocation failed");
    mythos_observables_destroy(o);
    return nullptr;
  }
  v.bps = o->d_bps, v.quartets = o->d_quartets;
  return o;
}

void mythos_observables_destroy(mythos_obs_t* o) {
  if (!o) return;
  (void)hipSetDevice(o->device);
  if (o->d_bps) (void)hipFree(o->d_bps);
  if (o->d_quartets) (void)hipFree(o->d_quartets);
  if (o->d_axis) (void)hipFree(o->d_axis);
  delete o;
}

int mythos_observables_width(const mythos_obs_t* o) { return o ? o->view.width : -1; }

int mythos_observables_eval(mythos_obs_t* o, const void* center, const void* quat, int n_frames, double* out,
                            mythos_stream_t stream) {
  if (!o || n_frames < 0 || (n_frames > 0 && (!center || !quat || !out))) {
    set_error("mythos_observables_eval: invalid argument");
    return MYTHOS_ERR_INVALID_ARGUMENT;
  }
  if (n_frames == 0) return MYTHOS_OK;
  MYTHOS_HIP_TRY(hipSetDevice(o->device));
  ObsView v;
  if (int rc = obs_view_for(o, n_frames, &v)) return rc;
  return observables_launch(o, v, center, quat, n_frames, out, (hipStream_t)stream);
}

}  // extern "C"
