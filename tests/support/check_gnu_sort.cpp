// Compares csrc/gnu_sort.h with the host libstdc++ std::sort: identical permutation required (ties included).
// Build: g++ -O2 -std=c++17 check_gnu_sort.cpp -o check_gnu_sort ; exit code 0 iff all cases identical.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
static long g_heap_calls = 0;
#define ORBX_SORT_COUNT_HEAP g_heap_calls
#include "../../orb_slam3_modified_amd/csrc/gnu_sort.h"

struct Node { int ulx; };
typedef std::pair<int, Node*> Item;
// same shape as the reference comparator (src/ORBextractor.cc:538-553): non-const refs, (count, UL.x)
static bool cmp(Item& a, Item& b) {
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->ulx < b.second->ulx;
}


static bool run_case(const std::vector<int>& cnt, const std::vector<int>& ulx) {
  size_t n = cnt.size();
  std::vector<Node> nodes(n);
  std::vector<Item> ref(n);
  std::vector<orbx_sort::elem_t> mine(n);
  for (size_t i = 0; i < n; i++) {
    nodes[i].ulx = ulx[i];
    ref[i] = Item(cnt[i], &nodes[i]);
    uint32_t key = ((uint32_t)cnt[i] << 13) | (uint32_t)ulx[i];
    mine[i] = ((uint64_t)key << 32) | (uint32_t)i;
  }
  std::sort(ref.begin(), ref.end(), cmp);
  std::vector<orbx_sort::elem_t> model(mine), tmp(n + 1);
  std::vector<int> ia(n + 1), ir(n + 1), slo(n + 1), shi(n + 1), work(6 * (n / 8 + 2));
  orbx_sort::gnu_sort(mine.data(), (int)n);
  orbx_sort::gnu_sort_model(model.data(), (int)n, ia.data(), ir.data(), slo.data(), shi.data(), tmp.data(), work.data());
  for (size_t i = 0; i < n; i++)
    if (model[i] != mine[i]) {
      fprintf(stderr, "MODEL MISMATCH n=%zu at %zu: serial %u parallel-model %u\n", n, i, (uint32_t)mine[i], (uint32_t)model[i]);
      return false;
    }
  for (size_t i = 0; i < n; i++) {
    size_t ri = (size_t)(ref[i].second - nodes.data());
    if (ri != (uint32_t)mine[i]) {
      fprintf(stderr, "MISMATCH n=%zu at %zu: ref %zu mine %u\n", n, i, ri, (uint32_t)mine[i]);
      return false;
    }
  }
  return true;
}

// median-of-3 killer (Musser): forces quadratic partitioning -> depth limit -> heapsort fallback
static std::vector<int> killer(int n) {
  std::vector<int> v(n);
  int k = n / 2;
  for (int i = 1; i <= k; i++) {
    if (i % 2) { v[i - 1] = i; v[i] = k + i; }
    v[k + i - 1] = 2 * i;
  }
  return v;
}

int main() {
  std::mt19937 rng(12345);
  long cases = 0;
  for (int rep = 0; rep < 4000; rep++) {
    int n = rng() % (rep < 3000 ? 400 : 3000);
    int kc = 1 + rng() % 6, kx = 1 + rng() % 8;
    std::vector<int> c(n), x(n);
    int mode = rng() % 5;
    for (int i = 0; i < n; i++) {
      c[i] = 2 + rng() % kc;
      x[i] = (rng() % kx) * 37;
      if (mode == 1) c[i] = 2 + i * kc / std::max(n, 1);            // ascending runs
      if (mode == 2) c[i] = 2 + (n - i) * kc / std::max(n, 1);      // descending runs
      if (mode == 3) c[i] = 2 + std::min(i, n - i) % (kc + 1);      // organ pipe
    }
    if (!run_case(c, x)) return 1;
    cases++;
  }
  for (int n : {17, 33, 64, 100, 257, 1000, 4096, 20000}) {
    std::vector<int> c = killer(n), x(n, 0);
    for (int& v : c) v = v % 2000 + 2;
    if (!run_case(c, x)) return 1;
    std::vector<int> c2 = killer(n);
    for (int& v : c2) v += 2;
    std::vector<int> x2(n);
    for (int i = 0; i < n; i++) { x2[i] = c2[i] % 8192; c2[i] = 2 + (c2[i] >> 13); }
    if (!run_case(c2, x2)) return 1;
    cases += 2;
  }
  printf("gnu_sort == std::sort on %ld cases (%ld heapsort fallbacks exercised)\n", cases, g_heap_calls);
  if (g_heap_calls == 0) { fprintf(stderr, "heapsort fallback never exercised\n"); return 2; }
  return 0;
}
