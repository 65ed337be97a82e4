// Exhaustive / strided comparison of csrc/glibc_sincosf.h against the host glibc cosf/sinf.
// usage: check_sincosf [stride]   (stride 1 = every float in [0, 2*pi]; exit code 0 iff no mismatch)
// Build: g++ -O2 -std=c++17 -mfma -ffp-contract=off check_sincosf.cpp -o check_sincosf -lpthread
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
#include "../../orb_slam3_modified_amd/csrc/glibc_sincosf.h"

static float (*volatile p_cosf)(float) = cosf;
static float (*volatile p_sinf)(float) = sinf;

int main(int argc, char** argv) {
  uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
  float hi = 6.2831855f * 1.0001f;  // a little beyond 360 deg * factorPI
  uint32_t hib;
  memcpy(&hib, &hi, 4);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  std::atomic<uint64_t> bad{0}, total{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      uint64_t b = 0, n = 0;
      for (uint64_t u = (uint64_t)t * stride; u <= hib; u += (uint64_t)nt * stride) {
        uint32_t ub = (uint32_t)u;
        float y;
        memcpy(&y, &ub, 4);
        float s0 = p_sinf(y), c0 = p_cosf(y);
        float s1 = orbx_glibc::sinf_exact(y), c1 = orbx_glibc::cosf_exact(y);
        if (memcmp(&s0, &s1, 4) || memcmp(&c0, &c1, 4)) {
          if (b < 5) fprintf(stderr, "mismatch y=%a sin %a vs %a cos %a vs %a\n", y, s0, s1, c0, c1);
          b++;
        }
        n++;
      }
      bad += b;
      total += n;
    });
  for (auto& x : th) x.join();
  printf("checked %llu arguments, %llu mismatches\n", (unsigned long long)total.load(), (unsigned long long)bad.load());
  return bad.load() ? 1 : 0;
}
