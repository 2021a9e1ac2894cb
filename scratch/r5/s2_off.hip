// scratch A/B: the library without the stride-2 weight-gradient kernel (those launches fall back to the round-2 tiles)
#include "common.h"
#include "conv_igemm.h"
namespace dynmm {
bool wgrad_s2_shape_ok(const dynmm_conv_geom*) { return false; }
void launch_wgrad_s2(const WgradArgs&, const WgradGroup&, dim3, hipStream_t) {}
}
