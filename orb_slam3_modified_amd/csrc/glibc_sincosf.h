// Bit-exact re-statement of glibc 2.35's cosf/sinf (x86-64 FMA ifunc variant) for arguments in [0, 120).
//
// Why this exists: the reference's steered-BRIEF rotation is `a = cos(angle), b = sin(angle)` on a float
// (src/ORBextractor.cc:111-112, std::cos(float) -> glibc cosf).  glibc's cosf/sinf are NOT correctly rounded
// (<= 0.56 ULP), so a device libm or a correctly-rounded evaluation would differ in the last bit for ~1% of
// angles and could flip a rounded tap coordinate.  The descriptor bits must be bit-identical to the CPU path,
// so the device evaluates the very same double-precision polynomial, with the same fused/un-fused operation
// sequence the host's libm executes.  The coefficient table and the operation order were read out of
// /lib/x86_64-linux-gnu/libm.so.6 (glibc 2.35-0ubuntu3.11: `__sincosf_table`, `__sinf_fma`, `__cosf_fma`);
// tests/support/check_sincosf.cpp compares this header against the host's cosf/sinf over EVERY float in
// [0, 2*pi] (1.09e9 arguments) and tests/test_sincosf.py runs a strided subset on each CI pass.
//
// Usable from host code (needs a hardware-or-correct fma()) and from HIP device code (v_fma_f64 is IEEE).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD static inline
#endif

namespace orbx_glibc {

// coefficient layout: c0 c1 s1 c2 s2 c3 s3 c4 (as stored by glibc); set 1 is used when quadrant & 2.
struct Poly { double c0, c1, s1, c2, s2, c3, s3, c4; };

ORBX_HD double hpi_inv_2p24() { return 0x1.45f306dc9c883p+23; }
ORBX_HD double hpi() { return 0x1.921fb54442d18p+0; }

ORBX_HD Poly poly(int neg) {
  Poly p;
  p.s1 = -0x1.555545995a603p-3;
  p.s2 = 0x1.1107605230bc4p-7;
  p.s3 = -0x1.994eb3774cf24p-13;
  if (!neg) {
    p.c0 = 0x1p0; p.c1 = -0x1.ffffffd0c621cp-2; p.c2 = 0x1.55553e1068f19p-5;
    p.c3 = -0x1.6c087e89a359dp-10; p.c4 = 0x1.99343027bf8c3p-16;
  } else {
    p.c0 = -0x1p0; p.c1 = 0x1.ffffffd0c621cp-2; p.c2 = -0x1.55553e1068f19p-5;
    p.c3 = 0x1.6c087e89a359dp-10; p.c4 = -0x1.99343027bf8c3p-16;
  }
  return p;
}

// operation order of the FMA build: every a*b+c below is ONE fused op, every other * is a rounded product.
ORBX_HD float sin_poly(double xs, double x2, const Poly& p) {
  double t = fma(x2, p.s3, p.s2);
  double x3 = x2 * xs;
  double x5 = x2 * x3;
  double s = fma(x3, p.s1, xs);
  return (float)fma(t, x5, s);
}
ORBX_HD float cos_poly(double x2, const Poly& p) {
  double x4 = x2 * x2;
  double c1 = fma(x2, p.c1, p.c0);
  double c2 = fma(x2, p.c4, p.c3);
  double x6 = x2 * x4;
  double c = fma(x4, p.c2, c1);
  return (float)fma(c2, x6, c);
}

ORBX_HD uint32_t abstop12(float y) {
  uint32_t u;
  memcpy(&u, &y, 4);
  return (u >> 20) & 0x7ff;
}

// reduce_fast: n = round(x * 2/pi) via the 2^24-prescaled multiply; r = fma(-n, pi/2, x)
ORBX_HD double reduce_fast(double x, int* np) {
  double r = x * hpi_inv_2p24();
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fma(-(double)n, hpi(), x);
}

ORBX_HD double quadrant_sign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }

// valid for 0 <= y < 120 (the path only produces [0, 2*pi])
ORBX_HD float sinf_exact(float y) {
  double x = (double)y;
  uint32_t top = abstop12(y);
  if (top <= 0x3f3) {  // |y| < pi/4
    if (top <= 0x397) return y;  // |y| < 2^-12
    return sin_poly(x, x * x, poly(0));
  }
  int n;
  double xr = reduce_fast(x, &n);
  double x2 = xr * xr;
  Poly p = poly((n & 2) != 0);
  if (n & 1) return cos_poly(x2, p);
  return sin_poly(xr * quadrant_sign(n), x2, p);
}

ORBX_HD float cosf_exact(float y) {
  double x = (double)y;
  uint32_t top = abstop12(y);
  if (top <= 0x3f3) {
    if (top <= 0x397) return 1.0f;
    return cos_poly(x * x, poly(0));
  }
  int n;
  double xr = reduce_fast(x, &n);
  double x2 = xr * xr;
  Poly p = poly((n & 2) != 0);
  if (n & 1) return sin_poly(xr * quadrant_sign(n), x2, p);
  return cos_poly(x2, p);
}

}  // namespace orbx_glibc
