// MARTINI 2/3 energy path (shifted-cut-off Lennard-Jones over all non-bonded bead pairs, harmonic
// bonds, G96 / harmonic angles).  Replaces mythos/energy/martini/m2/{lj,bond,angle}.py and
// m3/angle.py.  PLACEHOLDER: entry points exist so the ABI is complete; the kernels land next.
#include "mythos_internal.h"

using namespace mythos;

struct mythos_martini {
  int n = 0;
};

extern "C" {

mythos_martini_t* mythos_martini_create(int, const int32_t*, int, const double*, const double*, int, const int32_t*,
                                        const double*, const double*, int, const int32_t*, const double*,
                                        const double*, int, double, int, int) {
  set_error("mythos_martini_create: MARTINI kernels are not implemented yet");
  return nullptr;
}

void mythos_martini_destroy(mythos_martini_t* m) { delete m; }

int mythos_martini_energy(mythos_martini_t*, const void*, const void*, int, double*, void*, mythos_stream_t) {
  set_error("mythos_martini_energy: MARTINI kernels are not implemented yet");
  return MYTHOS_ERR_NOT_READY;
}

}  // extern "C"
