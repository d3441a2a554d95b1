// Energy kernel, translation unit 1 of 2: the fp32 instantiations of oxdna_energy_core.inc and the dispatch by
// precision.  The fp64 instantiations are in oxdna_kernels_f64.hip.
#include "oxdna_energy_core.inc"

namespace mythos {

MYTHOS_ENERGY_DEFINE_PRECISION(float)

int oxdna_energy_launch(mythos_system* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                        void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_obs* oset, double* obs_out, hipStream_t stream) {
  return sys->dtype == MYTHOS_F32
             ? oxdna_energy_launch_typed<float>(sys, center, quat, n_frames, e_terms, dU_dcenter, dU_dquat, dU_dparams, oset, obs_out, stream)
             : oxdna_energy_launch_typed<double>(sys, center, quat, n_frames, e_terms, dU_dcenter, dU_dquat, dU_dparams, oset, obs_out, stream);
}

}  // namespace mythos
