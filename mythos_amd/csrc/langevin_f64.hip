// Langevin integrator, translation unit 2 of 2: the fp64 instantiations of langevin_core.inc (the reference's
// precision).  Compiled without machine LICM (Makefile: LANGEVIN_F64_FLAGS), which the fp32 unit keeps.
#include "langevin_core.inc"

MYTHOS_MD_DEFINE_PRECISION(double)
