// Energy kernel, translation unit 2 of 2: the fp64 instantiations of oxdna_energy_core.inc (the reference's precision,
// and the one its DiffTRe gradients are validated in).  Compiled without machine LICM (Makefile: ENERGY_F64_FLAGS),
// which the fp32 unit keeps.
#include "oxdna_energy_core.inc"

namespace mythos {

MYTHOS_ENERGY_DEFINE_PRECISION(double)

}  // namespace mythos
