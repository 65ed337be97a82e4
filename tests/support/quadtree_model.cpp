// TEST INFRASTRUCTURE: sequential "array formulation" of the reference quadtree distribution
// (src/ORBextractor.cc:555-779), i.e. the data-parallel restatement the HIP kernel csrc/orbx_quadtree.hip
// implements: nodes live in a flat list array, every node owns a contiguous segment of a point array,
// a pass = 4-way stable partition of every expandable segment + closed-form positions of the children
// in the next list (std::list push_front order) computed with prefix sums.  tests/test_quadtree_model.py
// checks this file against the std::list/std::sort oracle on random inputs, so a disagreement between
// kernel and oracle can be attributed to the kernel's parallel primitives rather than to these formulas.
//
// Build: g++ -O2 -std=c++17 -shared -fPIC quadtree_model.cpp -o libquadtree_model.so
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../orb_slam3_modified_amd/csrc/gnu_sort.h"

namespace {

struct QNode { int x0, y0, x1, y1, start, count; };
struct Pt { int x, y, r; };

struct Kids { int c[4]; QNode n[4]; };

// children rectangles + counts of `nd` (DivideNode, src/ORBextractor.cc:480-536); optionally scatter
Kids split(const QNode& nd, const std::vector<Pt>& cur, std::vector<Pt>* nxt) {
  Kids k;
  const int hx = (int)std::ceil(static_cast<float>(nd.x1 - nd.x0) / 2);
  const int hy = (int)std::ceil(static_cast<float>(nd.y1 - nd.y0) / 2);
  const int sx = nd.x0 + hx, sy = nd.y0 + hy;
  k.n[0] = {nd.x0, nd.y0, sx, sy, 0, 0};
  k.n[1] = {sx, nd.y0, nd.x1, sy, 0, 0};
  k.n[2] = {nd.x0, sy, sx, nd.y1, 0, 0};
  k.n[3] = {sx, sy, nd.x1, nd.y1, 0, 0};
  for (int c = 0; c < 4; c++) k.c[c] = 0;
  auto child_of = [&](const Pt& p) { return (p.x < sx ? 0 : 1) + (p.y < sy ? 0 : 2); };
  for (int i = 0; i < nd.count; i++) k.c[child_of(cur[nd.start + i])]++;
  int off = nd.start;
  for (int c = 0; c < 4; c++) { k.n[c].start = off; k.n[c].count = k.c[c]; off += k.c[c]; }
  if (nxt) {
    int w[4] = {k.n[0].start, k.n[1].start, k.n[2].start, k.n[3].start};
    for (int i = 0; i < nd.count; i++) {
      const Pt& p = cur[nd.start + i];
      (*nxt)[w[child_of(p)]++] = p;
    }
  }
  return k;
}

}  // namespace

extern "C" int qtm_distribute(const int* xs, const int* ys, const int* rs, int n, int minX, int maxX, int minY, int maxY,
                              int N, int* out_x, int* out_y, int* out_r, int cap) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::vector<Pt> cur(n), nxt(n);
  std::vector<QNode> L;
  {  // roots: stable bucket by (int)(x / hX)
    std::vector<int> cnt(nIni, 0), off(nIni + 1, 0);
    for (int i = 0; i < n; i++) cnt[(int)((float)xs[i] / hX)]++;
    for (int b = 0; b < nIni; b++) off[b + 1] = off[b] + cnt[b];
    std::vector<int> w(off.begin(), off.end() - 1);
    for (int i = 0; i < n; i++) cur[w[(int)((float)xs[i] / hX)]++] = {xs[i], ys[i], rs[i]};
    for (int b = 0; b < nIni; b++)
      if (cnt[b] > 0) L.push_back({(int)(hX * (float)b), 0, (int)(hX * (float)(b + 1)), maxY - minY, off[b], cnt[b]});
  }
  std::vector<uint64_t> E;  // expand list: key=(count<<13|x0) high, list index low
  bool finish = false;
  while (!finish) {
    const int prevSize = (int)L.size();
    // ---- full pass: split every node with count > 1
    std::vector<Kids> kids(L.size());
    std::vector<int> kpre(L.size() + 1, 0), qpre(L.size() + 1, 0), spre(L.size() + 1, 0);
    for (size_t i = 0; i < L.size(); i++) {
      int k = 0, q = 0, s = 0;
      if (L[i].count > 1) {
        kids[i] = split(L[i], cur, &nxt);
        for (int c = 0; c < 4; c++) { k += kids[i].c[c] > 0; q += kids[i].c[c] > 1; }
      } else {
        nxt[L[i].start] = cur[L[i].start];
        s = 1;
      }
      kpre[i + 1] = kpre[i] + k; qpre[i + 1] = qpre[i] + q; spre[i + 1] = spre[i] + s;
    }
    const int totalKids = kpre[L.size()], nToExpand = qpre[L.size()];
    std::vector<QNode> Ln(totalKids + spre[L.size()]);
    E.assign(nToExpand, 0);
    for (size_t i = 0; i < L.size(); i++) {
      if (L[i].count > 1) {
        int k = kpre[i + 1] - kpre[i], ci = 0, qi = 0;
        for (int c = 0; c < 4; c++) {
          if (kids[i].c[c] == 0) continue;
          int pos = totalKids - kpre[i + 1] + (k - 1 - ci);
          Ln[pos] = kids[i].n[c];
          if (kids[i].c[c] > 1) {
            uint32_t key = ((uint32_t)kids[i].c[c] << 13) | (uint32_t)kids[i].n[c].x0;
            E[qpre[i] + qi] = ((uint64_t)key << 32) | (uint32_t)pos;
            qi++;
          }
          ci++;
        }
      } else {
        Ln[totalKids + spre[i]] = L[i];
      }
    }
    L.swap(Ln);
    cur.swap(nxt);
    if ((int)L.size() >= N || (int)L.size() == prevSize) {
      finish = true;
    } else if ((int)L.size() + nToExpand * 3 > N) {
      // ---- sorted expansion
      while (!finish) {
        const int prev2 = (int)L.size();
        const int m = (int)E.size();
        orbx_sort::gnu_sort(E.data(), m);
        std::vector<Kids> kd(m);
        std::vector<int> kk(m), qq(m);
        for (int j = 0; j < m; j++) {
          kd[j] = split(L[(uint32_t)E[j]], cur, nullptr);
          kk[j] = qq[j] = 0;
          for (int c = 0; c < 4; c++) { kk[j] += kd[j].c[c] > 0; qq[j] += kd[j].c[c] > 1; }
        }
        int jstar = 0, running = (int)L.size();
        for (int j = m - 1; j >= 0; j--) {
          running += kk[j] - 1;
          if (running >= N) { jstar = j; break; }
        }
        std::vector<char> erased(L.size(), 0);
        for (int j = jstar; j < m; j++) erased[(uint32_t)E[j]] = 1;
        int front = 0;
        for (int j = jstar; j < m; j++) front += kk[j];
        std::vector<QNode> Ln2;
        Ln2.resize(front);
        std::vector<uint64_t> En;
        // point movement: processed nodes scatter, everything else copies
        for (size_t i = 0; i < L.size(); i++)
          if (!erased[i]) for (int t = 0; t < L[i].count; t++) nxt[L[i].start + t] = cur[L[i].start + t];
        for (int j = jstar; j < m; j++) kd[j] = split(L[(uint32_t)E[j]], cur, &nxt);
        // list front: children of jstar first ... children of m-1 last, each group reversed (n4..n1)
        int base = 0;
        std::vector<int> basej(m, 0);
        for (int j = jstar; j < m; j++) { basej[j] = base; base += kk[j]; }
        // expand list in creation order: j = m-1 down to jstar, children n1..n4
        for (int j = m - 1; j >= jstar; j--) {
          int ci = 0;
          for (int c = 0; c < 4; c++) {
            if (kd[j].c[c] == 0) continue;
            int pos = basej[j] + (kk[j] - 1 - ci);
            Ln2[pos] = kd[j].n[c];
            if (kd[j].c[c] > 1) {
              uint32_t key = ((uint32_t)kd[j].c[c] << 13) | (uint32_t)kd[j].n[c].x0;
              En.push_back(((uint64_t)key << 32) | (uint32_t)pos);
            }
            ci++;
          }
        }
        for (size_t i = 0; i < L.size(); i++)
          if (!erased[i]) Ln2.push_back(L[i]);
        // surviving old nodes moved by `front - (#erased before them)`: expand-list payloads only refer to
        // freshly created children (positions < front), so no remapping is needed.
        L.swap(Ln2);
        cur.swap(nxt);
        E.swap(En);
        if ((int)L.size() >= N || (int)L.size() == prev2) finish = true;
      }
    }
  }
  int nout = 0;
  for (const QNode& nd : L) {
    int best = nd.start;
    for (int t = 1; t < nd.count; t++)
      if (cur[nd.start + t].r > cur[best].r) best = nd.start + t;
    if (nout < cap) { out_x[nout] = cur[best].x; out_y[nout] = cur[best].y; out_r[nout] = cur[best].r; }
    nout++;
  }
  return nout;
}

extern "C" void qtm_sort(uint64_t* v, int n) { orbx_sort::gnu_sort(v, n); }
