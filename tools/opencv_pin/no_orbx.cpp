// opencv_pin built without liborbx.so (no -DWITH_ORBX=ON): the four C-ABI entry points tools/validate_opencv.cpp names, as refusals.
// The primitive legs and the profile table need no GPU; --orbx says so and stops.
#include "orbx.h"
extern "C" {
int orbx_create(orbx_ctx** out, int, float, int, int, int, int) { if (out) *out = nullptr; return ORBX_E_DEVICE; }
void orbx_destroy(orbx_ctx*) {}
const char* orbx_last_error(const orbx_ctx*) { return "opencv_pin was built without liborbx.so (cmake -DWITH_ORBX=ON)"; }
int orbx_keypoint_capacity(const orbx_ctx*) { return 0; }
int orbx_set_option(orbx_ctx*, const char*, int) { return ORBX_E_DEVICE; }
int orbx_extract(orbx_ctx*, const uint8_t*, int, int, size_t, int, int, orbx_keypoint*, uint8_t*, int*, int*) { return ORBX_E_DEVICE; }
}
