// TEST INFRASTRUCTURE.  What does THIS compiler make of an expression of the shape the reference rotates its test pattern with
// (src/ORBextractor.cc:118-120: an int coordinate times a float, plus / minus another such product, rounded half to even) when it is
// built the way the reference's CMakeLists.txt:10-13 builds (-O3 and the FMA instructions -march=native brings, default -ffp-contract)?
// tests/test_opencv_variants.py compiles this file twice (with -mfma and with -ffp-contract=off) and compares the digests over 60 million
// operand sets with the oracle's two settings of "brief_fma" (orbo_rot_probe_hash) — the settings disagree about once in ten million.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct P { int x, y; };

__attribute__((noinline)) void rot(const P* p, float a, float b, int* ry, int* rx) {
  *ry = (int)lrintf(p->x * b + p->y * a);
  *rx = (int)lrintf(p->x * a - p->y * b);
}

// operands derived from a fixed integer mix (the oracle's orbo_rot_probe_hash walks the same sequence); prints one digest
static inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 1000000u;
  uint64_t h = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t r0 = mix32(2 * i + 1), r1 = mix32(2 * i + 2);
    P p{(int)(r0 % 27u) - 13, (int)((r0 >> 8) % 27u) - 13};
    const float ang = (float)(r1 % 36000001u) * 1e-5f * (float)(M_PI / 180.f);
    const float a = cosf(ang), b = sinf(ang);
    int ry, rx;
    rot(&p, a, b, &ry, &rx);
    h += (uint64_t)(uint32_t)(ry * 64 + rx + 4096) * (0x9E3779B97F4A7C15ull + 2ull * (uint64_t)(i & 1023u));
  }
  printf("%016llx\n", (unsigned long long)h);
  return 0;
}
