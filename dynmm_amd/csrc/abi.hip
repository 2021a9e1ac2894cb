#include "common.h"

extern "C" int dynmm_abi_version(void) { return 4; }

extern "C" const char* dynmm_build_info(void) {
    return "libdynmm_hip gfx950 fp32-MFMA(v_mfma_f32_32x32x2_f32) built " __DATE__ " " __TIME__;
}
