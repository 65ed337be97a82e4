// TEST INFRASTRUCTURE: include/orbx_cv_calibrate.h's std::sort probe.  Prints the digest of (a) the toolchain's std::sort, (b) the repository's
// sequential restatement of libstdc++'s introsort (csrc/gnu_sort.h — the statement the device code is checked against), (c) std::stable_sort
// (a different tie order: the probe must tell it apart), and the constant the header compares with.
#include <cstdio>
#include <vector>

extern "C" { struct orbx_ctx; int orbx_set_option(orbx_ctx*, const char*, int) { return 0; } }
#define ORBX_NO_CV_CALIBRATION 1
#include "orbx_cv_calibrate.h"
#include "gnu_sort.h"

int main() {
  typedef std::vector<std::pair<int, int> > V;
  const unsigned long long a = orbx_cv::sort_probe_digest([](V& v) {
    std::sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
  });
  const unsigned long long b = orbx_cv::sort_probe_digest([](V& v) {
    const int n = (int)v.size();
    std::vector<orbx_sort::elem_t> e((size_t)n), tmp((size_t)n);
    for (int i = 0; i < n; i++) e[(size_t)i] = (orbx_sort::elem_t)(unsigned)v[(size_t)i].first << 32 | (unsigned)v[(size_t)i].second;
    std::vector<int> ia((size_t)n), ir((size_t)n), lo((size_t)n), hi((size_t)n), work((size_t)(6 * (n / 8 + 2)));
    orbx_sort::gnu_sort_model(e.data(), n, ia.data(), ir.data(), lo.data(), hi.data(), tmp.data(), work.data());
    for (int i = 0; i < n; i++) v[(size_t)i] = std::make_pair((int)(e[(size_t)i] >> 32), (int)(unsigned)e[(size_t)i]);
  });
  const unsigned long long c = orbx_cv::sort_probe_digest([](V& v) {
    std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
  });
  std::printf("std_sort %016llx gnu_sort_h %016llx stable_sort %016llx constant %016llx is_libstdcxx %d\n", a, b, c,
              (unsigned long long)orbx_cv::kLibstdcxxSortDigest, (int)orbx_cv::std_sort_is_libstdcxx());
  return 0;
}
