// orbx — the DIAGNOSTIC ABI (include/orbx_debug.h): stage dumps of an extraction and numeric test hooks (device cos / sin / fastAtan2 /
// pattern-rotation digests, the exact sort on caller data, the counter-calibration copy).  Test and tooling infrastructure: built into
// liborbx_debug.so, NEVER into liborbx.so — the product library carries no debug entry point and no debug kernel.
//
// The hooks read the buffers a product context (an orbx_ctx made by liborbx.so) left behind and run small kernels of their own that share the
// product's device functions (block_gnu_sort, fast_atan2_deg, the glibc-exact sincosf, the rotated test pattern), so this translation unit is
// the extractor's with ORBX_DEBUG_ABI defined; it is compiled with -fvisibility=hidden and exports the orbx_debug_* symbols alone.
#define ORBX_DEBUG_ABI 1
#include "../../include/orbx_debug.h"
#include "orbx_extractor.hip"
