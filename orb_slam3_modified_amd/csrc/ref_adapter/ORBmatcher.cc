// orbx adapter — replaces src/ORBmatcher.cc of the reference (lturing/ORB_SLAM3_modified) behind the unchanged class
// interface of include/ORBmatcher.h.  Every routine has the same three phases:
//
//   1. a host pre-pass over the reference's own objects (MapPoint / KeyFrame / Frame, Sophus / Eigen, GeometricCamera) that
//      evaluates the per-query gates and geometry exactly as the reference's loop head does and flattens the survivors into
//      query arrays (centre, radius, level range, descriptor);
//   2. ONE device pass per camera (include/orbx.h): the grid window of every query (Frame / KeyFrame::GetFeaturesInArea), the
//      Hamming distance of every candidate and — for the routines whose queries are independent (Fuse x2, SearchBySim3,
//      SURVEY.md §3.3) — the arg-min itself; vocabulary-node routines send their node groups to orbx_nn_groups (near candidates only);
//   3. a host replay in the reference's query order of whatever depends on earlier queries (occupancy of keypoints, stolen
//      matches, map mutation) and the rotation-consistency filter, written with the reference's own comparison operators
//      (each routine has its own accept test: `<= TH_HIGH`, `<= TH_LOW`, `< TH_LOW`, `<= TH_LOW*ratioHamming`, ...).
//
// Citations `:n` are lines of the reference's src/ORBmatcher.cc.  Two-camera rigs (Frame::Nleft != -1, KeyFrame::NLeft != -1)
// are handled like the reference handles them: the left and the right camera are two grids over two keypoint arrays that share
// one descriptor matrix and one map-point vector.
#include "ORBmatcher.h"

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

using namespace std;

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte POD orbx_keypoint mirrors");

namespace {

constexpr int kCols = 64, kRows = 48;   // FRAME_GRID_COLS x FRAME_GRID_ROWS (include/Frame.h:44-45); the device grid is built for these

// ORBX_TRACE_MATCHER=1: phase times of the two per-frame routines on stderr (diagnostics)
struct PhaseTrace {
  static bool on() { static const bool v = std::getenv("ORBX_TRACE_MATCHER") != nullptr; return v; }
  const char* what;
  std::chrono::steady_clock::time_point t0, last;
  char buf[256];
  int len = 0;
  explicit PhaseTrace(const char* w) : what(w) { if (on()) t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char* phase) {
    if (!on()) return;
    const auto now = std::chrono::steady_clock::now();
    len += std::snprintf(buf + len, sizeof(buf) - len, " %s %.1f", phase, std::chrono::duration<double, std::micro>(now - last).count());
    last = now;
  }
  ~PhaseTrace() {
    if (on()) std::fprintf(stderr, "[orbx matcher] %s:%s us (total %.1f)\n", what, buf, std::chrono::duration<double, std::micro>(last - t0).count());
  }
};

// the rigid part of a similarity: rotation, and the translation brought to unit scale (what every Sim3 routine of the reference starts from)
inline Sophus::SE3f rigid_part(const Sophus::Sim3f& S) { return Sophus::SE3f(S.rotationMatrix(), S.translation() / S.scale()); }

[[noreturn]] void fail(const char* routine, orbx_ctx* ctx) {
  throw std::runtime_error(std::string("ORBmatcher::") + routine + ": " + (ctx ? orbx_last_error(ctx) : "no orbx context"));
}

// mGrid[ix][iy] flattened cell by cell — the lists exactly as AssignFeaturesToGrid left them
struct FlatGrid {
  std::vector<int32_t> start, idx;
  orbx_grid g;
  template <class CellFn>
  void build(float minX, float minY, float invW, float invH, int expected, CellFn cell) {
    start.clear(); idx.clear();
    start.reserve(kCols * kRows + 1); idx.reserve(expected);
    start.push_back(0);
    for (int ix = 0; ix < kCols; ix++)
      for (int iy = 0; iy < kRows; iy++) {
        const std::vector<size_t>& c = cell(ix, iy);
        for (size_t v : c) idx.push_back((int32_t)v);
        start.push_back((int32_t)idx.size());
      }
    g.min_x = minX; g.min_y = minY; g.inv_w = invW; g.inv_h = invH;
    g.cell_start = start.data(); g.cell_idx = idx.data();
  }
};

void frame_grid(const Frame& F, bool bRight, FlatGrid& fg) {
  fg.build(Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, F.N,
           [&](int ix, int iy) -> const std::vector<size_t>& { return bRight ? F.mGridRight[ix][iy] : F.mGrid[ix][iy]; });
}
void keyframe_grid(KeyFrame* pKF, bool bRight, FlatGrid& fg) {
  if (pKF->mnGridCols != kCols || pKF->mnGridRows != kRows) throw std::runtime_error("ORBmatcher: keyframe grid is not 64 x 48");
  fg.build((float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, pKF->N,
           [&](int ix, int iy) -> const std::vector<size_t>& { return bRight ? pKF->mGridRight[ix][iy] : pKF->mGrid[ix][iy]; });
}

// The keypoints a grid is searched over.  GetFeaturesInArea tests the window on mvKeysUn (single camera) or on mvKeys /
// mvKeysRight (rig); the level a candidate loop reads may come from another array (`levels`), so for rigs a merged copy is
// made: position from `pos`, octave from `levels`.
struct KeyView {
  const orbx_keypoint* p = nullptr;
  int n = 0;
  std::vector<cv::KeyPoint> merged;
  void direct(const std::vector<cv::KeyPoint>& v) { p = (const orbx_keypoint*)v.data(); n = (int)v.size(); }
  void merge(const std::vector<cv::KeyPoint>& pos, const std::vector<cv::KeyPoint>& levels) {
    merged = pos;
    for (size_t i = 0; i < merged.size() && i < levels.size(); i++) merged[i].octave = levels[i].octave;
    direct(merged);
  }
};

// rows of a descriptor matrix as one contiguous block
struct DescView {
  const unsigned char* p = nullptr;
  std::vector<unsigned char> copy;
  explicit DescView(const cv::Mat& m) {
    if (m.rows == 0) return;
    if (m.isContinuous()) { p = m.ptr<unsigned char>(0); return; }
    copy.resize((size_t)m.rows * 32);
    for (int r = 0; r < m.rows; r++) std::memcpy(&copy[(size_t)r * 32], m.ptr<unsigned char>(r), 32);
    p = copy.data();
  }
};

struct Queries {
  std::vector<float> x, y, r, aux;
  std::vector<int32_t> lo, hi;
  std::vector<unsigned char> desc;
  void reserve(size_t n) { x.reserve(n); y.reserve(n); r.reserve(n); aux.reserve(n); lo.reserve(n); hi.reserve(n); desc.reserve(n * 32); }
  int size() const { return (int)x.size(); }
  int add(float qx, float qy, float qr, int qlo, int qhi, const cv::Mat& d, float qaux = 0.f) {
    x.push_back(qx); y.push_back(qy); r.push_back(qr); aux.push_back(qaux); lo.push_back(qlo); hi.push_back(qhi);
    const unsigned char* s = d.ptr<unsigned char>(0);
    desc.insert(desc.end(), s, s + 32);
    return size() - 1;
  }
};

// Candidate lists of a query batch, in GetFeaturesInArea's order, with the distances: positions [begin(q), end(q)) of `cand` / `dist`.
// Either copies (row_ptr / own vectors) or a VIEW of the device pass's own output in the context's pinned blob (orbx_target_search_view:
// no copy-out; valid until the thread's next device call — the two per-frame Tracking routines read their lists before any other call).
struct Lists {
  std::vector<int32_t> row_ptr, cand_v, dist_v;
  const orbx_list_span* spans = nullptr;
  struct Column {
    const int32_t* own = nullptr; const orbx_candidate* view = nullptr; bool second = false;
    int32_t operator[](int c) const { return view ? (second ? view[c].dist : view[c].idx) : own[c]; }
  } cand, dist;
  // view mode only: the two smallest (distance, list position) of the query's whole segment, from the device pass (orbx_list_span)
  bool has_best() const { return spans != nullptr; }
  const orbx_list_span& span(int q) const { return spans[q]; }
  int begin(int q) const { return spans ? spans[q].start : row_ptr[q]; }
  int end(int q) const { return spans ? spans[q].start + spans[q].count : row_ptr[q + 1]; }
  void own() { spans = nullptr; cand = Column{cand_v.data(), nullptr, false}; dist = Column{dist_v.data(), nullptr, true}; }
  void view(const orbx_list_span* s, const orbx_candidate* p) { spans = s; cand = Column{nullptr, p, false}; dist = Column{nullptr, p, true}; }
};

// ---- resident search targets -----------------------------------------------------------------------------------------------------
// A Frame / KeyFrame is searched many times and never changes after construction: what a routine searches — keypoints, descriptor
// rows, the flattened grid and, for Fuse, mvuRight + mvInvLevelSigma2 — is uploaded once per thread context (orbx_target_*) and kept
// in a small LRU.  The key is (kind, mnId, variant, count, descriptor address) PLUS a 64-bit digest of the contents (every keypoint,
// every descriptor byte, mvuRight): ids are NOT unique per process — Tracking::Reset() sets KeyFrame::nNextId = Frame::nNextId = 0
// (src/Tracking.cc:3819-3820, reached from System.cc:327,402,477) and an Atlas load restores the counters — so a later Frame can
// carry a recycled id, the same count and a recycled malloc address.  A hit therefore has to match the digest; an entry whose key
// matches but whose digest does not proves that ids were recycled, and everything cached before that moment is dropped (refilled,
// not freed: the device blocks are reused).  A hit costs the digest (≈ 60 KB of host reads, 2-3 µs), no grid flattening, no upload.
struct TargetSpec {
  const orbx_keypoint* kps = nullptr;
  const unsigned char* desc = nullptr;
  int n = 0;
  FlatGrid grid;
  const float* ur = nullptr;
  const float* sig = nullptr;
  int nlevels = 0;
  KeyView keys;      // owns a merged copy when the routine needs one
  DescView* dview = nullptr;
};

// 64-bit digest of a byte block: four independent multiply-xorshift lanes over 8-byte words (order-sensitive)
inline uint64_t digest_bytes(const void* p, size_t bytes, uint64_t seed) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t a0 = seed ^ 0x9e3779b97f4a7c15ull, a1 = seed + 0xc2b2ae3d27d4eb4full, a2 = ~seed * 0x165667b19e3779f9ull, a3 = seed ^ (uint64_t)bytes;
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    std::memcpy(w, b + i, 32);
    a0 = (a0 ^ w[0]) * 0xff51afd7ed558ccdull; a0 ^= a0 >> 29;
    a1 = (a1 ^ w[1]) * 0xc4ceb9fe1a85ec53ull; a1 ^= a1 >> 31;
    a2 = (a2 ^ w[2]) * 0x9fb21c651e98df25ull; a2 ^= a2 >> 30;
    a3 = (a3 ^ w[3]) * 0xd6e8feb86659fd93ull; a3 ^= a3 >> 32;
  }
  for (; i < bytes; i++) { a0 = (a0 ^ b[i]) * 0xff51afd7ed558ccdull; a0 ^= a0 >> 29; }
  uint64_t h = a0 ^ (a1 * 0x9e3779b97f4a7c15ull) ^ (a2 << 1 | a2 >> 63) ^ (a3 * 0xc2b2ae3d27d4eb4full);
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
  return h;
}

// everything a target is made of, as the caller's objects hold it (the grid and the merged rig keypoints derive from these)
struct TargetContent {
  const std::vector<cv::KeyPoint>* keys[2] = {nullptr, nullptr};
  const cv::Mat* desc = nullptr;
  const std::vector<float>* ur = nullptr;
  const std::vector<float>* sig = nullptr;
  float bounds[4] = {0, 0, 0, 0};   // grid origin and inverse cell size: a grid is a function of (keypoints, bounds)
  uint64_t digest() const {
    uint64_t h = digest_bytes(bounds, sizeof(bounds), 0x0b5e55edull);
    for (const std::vector<cv::KeyPoint>* k : keys)
      if (k && !k->empty()) h = digest_bytes(k->data(), k->size() * sizeof(cv::KeyPoint), h);
    if (desc && desc->rows > 0) {
      if (desc->isContinuous()) h = digest_bytes(desc->ptr<unsigned char>(0), (size_t)desc->rows * 32, h);
      else for (int r = 0; r < desc->rows; r++) h = digest_bytes(desc->ptr<unsigned char>(r), 32, h);
    }
    if (ur && !ur->empty()) h = digest_bytes(ur->data(), ur->size() * sizeof(float), h);
    if (sig && !sig->empty()) h = digest_bytes(sig->data(), sig->size() * sizeof(float), h);
    return h;
  }
};

struct TargetCache {
  struct Entry { int kind; unsigned long id; int variant; int n; const void* dptr; uint64_t tag; orbx_target* t; unsigned long stamp; };
  static constexpr size_t kMax = 64;
  std::vector<Entry> e;
  unsigned long clock = 0;
  unsigned long recycled_ids = 0;   // how often a recycled id was detected (diagnostics, tests)
  void clear() { for (Entry& x : e) orbx_target_destroy(x.t); e.clear(); }
  // nothing cached so far may be trusted any more; the device blocks stay and are refilled by later misses
  void invalidate() { for (Entry& x : e) { x.n = -1; x.stamp = 0; } }
  template <class Build>
  orbx_target* get(orbx_ctx* ctx, const char* routine, int kind, unsigned long id, int variant, int n, const void* dptr, const TargetContent& content,
                   Build build) {
    const uint64_t tag = content.digest();
    for (Entry& x : e)
      if (x.kind == kind && x.id == id && x.variant == variant && x.n == n && x.dptr == dptr) {
        if (x.tag == tag) { x.stamp = ++clock; return x.t; }
        recycled_ids++;
        invalidate();
        break;
      }
    TargetSpec sp;
    build(sp);
    Entry* slot = nullptr;
    for (Entry& x : e) if (x.n < 0) { slot = &x; break; }   // an invalidated block first
    if (!slot && e.size() < kMax) {
      orbx_target* t = nullptr;
      if (orbx_target_create(ctx, sp.kps, sp.desc, sp.n, &sp.grid.g, sp.ur, sp.sig, sp.nlevels, &t) != ORBX_OK) fail(routine, ctx);
      e.push_back(Entry{kind, id, variant, n, dptr, tag, t, 0});
      slot = &e.back();
    } else {
      if (!slot) {
        slot = &e[0];
        for (Entry& x : e) if (x.stamp < slot->stamp) slot = &x;
      }
      slot->n = -1;   // not a valid key while it is refilled
      if (orbx_target_assign(ctx, slot->t, sp.kps, sp.desc, sp.n, &sp.grid.g, sp.ur, sp.sig, sp.nlevels) != ORBX_OK) fail(routine, ctx);
      slot->kind = kind; slot->id = id; slot->variant = variant; slot->n = n; slot->dptr = dptr; slot->tag = tag;
    }
    slot->stamp = ++clock;
    return slot->t;
  }
};

struct ContextHolder {
  orbx_ctx* c = nullptr;
  TargetCache cache;
  ~ContextHolder() { cache.clear(); if (c) orbx_destroy(c); }   // targets before their context
};
ContextHolder& holder() {
  static thread_local ContextHolder h;
  return h;
}

enum { kFrameLeft = 0, kFrameRight = 1, kKeyFrameUn = 2, kFuseLeft = 3, kFuseRight = 4 };

// a Frame's left camera (mvKeysUn, or mvKeys of a rig: Frame::GetFeaturesInArea tests the window on those) / right camera of a rig
orbx_target* frame_target(const char* routine, const Frame& F, bool bRight) {
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  const std::vector<cv::KeyPoint>& keys = bRight ? F.mvKeysRight : (F.Nleft == -1 ? F.mvKeysUn : F.mvKeys);
  TargetContent tc;
  tc.keys[0] = &keys; tc.desc = &F.mDescriptors;
  tc.bounds[0] = Frame::mnMinX; tc.bounds[1] = Frame::mnMinY; tc.bounds[2] = Frame::mfGridElementWidthInv; tc.bounds[3] = Frame::mfGridElementHeightInv;
  return holder().cache.get(ctx, routine, 'F', F.mnId, bRight ? kFrameRight : kFrameLeft, (int)keys.size(), F.mDescriptors.data, tc, [&](TargetSpec& sp) {
    static thread_local std::vector<unsigned char> rows;
    DescView D(F.mDescriptors);
    const unsigned char* d = D.p ? D.p + (bRight ? (size_t)F.Nleft * 32 : 0) : nullptr;
    if (!D.copy.empty()) { rows = D.copy; d = rows.data() + (bRight ? (size_t)F.Nleft * 32 : 0); }
    sp.kps = (const orbx_keypoint*)keys.data(); sp.desc = d; sp.n = (int)keys.size();
    // A Frame's grid is a pure function of these keypoints and the static bounds (Frame::AssignFeaturesToGrid, src/Frame.cc:385-416:
    // every keypoint in index order into the cell PosInGrid names) — so it is rebuilt on the device from the uploaded keypoints
    // (k_window_grid, the same cell arithmetic, lists in the same order) instead of being flattened on the host, checked and
    // uploaded: the assign is asynchronous and hides behind the host pre-pass of the search that needs it.  Up to 32 768 keypoints
    // (the device sort's capacity); beyond that the host lists are sent.
    if ((int)keys.size() <= 32768) {
      sp.grid.g.min_x = Frame::mnMinX; sp.grid.g.min_y = Frame::mnMinY; sp.grid.g.inv_w = Frame::mfGridElementWidthInv; sp.grid.g.inv_h = Frame::mfGridElementHeightInv;
      sp.grid.g.cell_start = nullptr; sp.grid.g.cell_idx = nullptr;
    } else frame_grid(F, bRight, sp.grid);
  });
}

// what the KeyFrame-side routines search: mvKeysUn-levels over the grid the keyframe holds (a rig: window on mvKeys,
// src/KeyFrame.cc:735-737, level from mvKeysUn, e.g. :508)
orbx_target* keyframe_target(const char* routine, KeyFrame* pKF) {
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  TargetContent tc;
  tc.keys[0] = &pKF->mvKeysUn; if (pKF->NLeft != -1) tc.keys[1] = &pKF->mvKeys;
  tc.desc = &pKF->mDescriptors;
  tc.bounds[0] = (float)pKF->mnMinX; tc.bounds[1] = (float)pKF->mnMinY; tc.bounds[2] = pKF->mfGridElementWidthInv; tc.bounds[3] = pKF->mfGridElementHeightInv;
  return holder().cache.get(ctx, routine, 'K', pKF->mnId, kKeyFrameUn, (int)pKF->mvKeysUn.size(), pKF->mDescriptors.data, tc, [&](TargetSpec& sp) {
    static thread_local std::vector<unsigned char> rows;
    if (pKF->NLeft == -1) sp.keys.direct(pKF->mvKeysUn);
    else sp.keys.merge(pKF->mvKeys, pKF->mvKeysUn);
    DescView D(pKF->mDescriptors);
    const unsigned char* d = D.p;
    if (!D.copy.empty()) { rows = D.copy; d = rows.data(); }
    sp.kps = sp.keys.p; sp.desc = d; sp.n = sp.keys.n;
    keyframe_grid(pKF, false, sp.grid);
  });
}

// Fuse(KeyFrame*, points): KeyFrame::GetFeaturesInArea(..., bRight) with the level and reprojection gates (:1262-1296)
orbx_target* fuse_target(const char* routine, KeyFrame* pKF, bool bRight) {
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  const std::vector<cv::KeyPoint>& keys = pKF->NLeft == -1 ? pKF->mvKeysUn : (bRight ? pKF->mvKeysRight : pKF->mvKeys);
  TargetContent tc;
  tc.keys[0] = &keys; tc.desc = &pKF->mDescriptors; tc.ur = &pKF->mvuRight; tc.sig = &pKF->mvInvLevelSigma2;
  tc.bounds[0] = (float)pKF->mnMinX; tc.bounds[1] = (float)pKF->mnMinY; tc.bounds[2] = pKF->mfGridElementWidthInv; tc.bounds[3] = pKF->mfGridElementHeightInv;
  return holder().cache.get(ctx, routine, 'K', pKF->mnId, bRight ? kFuseRight : kFuseLeft, (int)keys.size(), pKF->mDescriptors.data, tc, [&](TargetSpec& sp) {
    static thread_local std::vector<unsigned char> rows;
    DescView D(pKF->mDescriptors);
    const unsigned char* d = D.p;
    if (!D.copy.empty()) { rows = D.copy; d = rows.data(); }
    sp.kps = (const orbx_keypoint*)keys.data(); sp.n = (int)keys.size();
    sp.desc = d ? d + (bRight ? (size_t)pKF->NLeft * 32 : 0) : nullptr;
    keyframe_grid(pKF, bRight, sp.grid);
    sp.ur = pKF->mvuRight.data(); sp.sig = pKF->mvInvLevelSigma2.data(); sp.nlevels = (int)pKF->mvInvLevelSigma2.size();
  });
}

void window_lists(const char* routine, orbx_target* T, const Queries& Q, Lists& L, bool view_ok = false) {
  const int nq = Q.size();
  L.row_ptr.assign(nq + 1, 0);
  L.own();
  if (nq == 0 || orbx_target_size(T) == 0) return;
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  if (view_ok) {
    const orbx_list_span* spans = nullptr;
    const orbx_candidate* pool = nullptr;
    if (orbx_target_search_view(ctx, T, nullptr, Q.x.data(), Q.y.data(), Q.r.data(), Q.lo.data(), Q.hi.data(), Q.desc.data(), nullptr, nq, &spans, &pool) < 0)
      fail(routine, ctx);
    L.view(spans, pool);
    return;
  }
  if (L.cand_v.size() < 4096) { L.cand_v.resize(4096); L.dist_v.resize(4096); }
  for (int attempt = 0; attempt < 2; attempt++) {
    const int rc = orbx_target_search(ctx, T, nullptr, Q.x.data(), Q.y.data(), Q.r.data(), Q.lo.data(), Q.hi.data(), Q.desc.data(), nullptr, nq,
                                      L.row_ptr.data(), L.cand_v.data(), L.dist_v.data(), (int)L.cand_v.size(), nullptr, nullptr, nullptr, nullptr);
    if (rc >= 0) { L.own(); return; }
    if (rc != ORBX_E_CAPACITY || attempt) fail(routine, ctx);
    const size_t need = (size_t)L.row_ptr[nq] + 64;   // row_ptr is complete on ORBX_E_CAPACITY
    L.cand_v.resize(need); L.dist_v.resize(need);
  }
}

// The two per-frame routines of Tracking split their points into two halves and overlap their own host passes with the device
// (orbx_target_search_view_begin / _end): the pre-pass of the second half runs while the device works on the first, the replay of the first
// half while it works on the second.  The replay stays in the reference's order (first half first), the device pass does not depend on
// what the replay binds, so the results are the same calls' results.  ORBX_SEARCH_PIPELINE=0 switches it off (A/B, tests).
bool search_pipeline_enabled() {
  static const bool on = [] { const char* e = std::getenv("ORBX_SEARCH_PIPELINE"); return !(e && std::atoi(e) == 0); }();
  return on;
}
// queries [q0, q0 + nq) of Q against T, issued only; -1 = nothing was issued (no queries or an empty target).  Q's vectors must not reallocate
// before window_lists_end (the callers reserve them for the whole call up front).
int window_lists_begin(const char* routine, orbx_target* T, const Queries& Q, int q0, int nq) {
  if (nq == 0 || !T || orbx_target_size(T) == 0) return -1;
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  const int slot = orbx_target_search_view_begin(ctx, T, nullptr, Q.x.data() + q0, Q.y.data() + q0, Q.r.data() + q0, Q.lo.data() + q0, Q.hi.data() + q0,
                                                 Q.desc.data() + (size_t)q0 * 32, nullptr, nq);
  if (slot < 0) fail(routine, ctx);
  return slot;
}
void window_lists_end(const char* routine, int& slot, int nq, Lists& L) {
  L.row_ptr.assign(nq + 1, 0);
  L.own();
  if (slot < 0) return;
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  const orbx_list_span* spans = nullptr;
  const orbx_candidate* pool = nullptr;
  const int s = slot;
  slot = -1;   // whatever happens now, the call has been collected: _end cleared the slot's pending flag before anything could fail
  if (orbx_target_search_view_end(ctx, s, &spans, &pool) < 0) fail(routine, ctx);
  if (spans) L.view(spans, pool);
}
// The two issued halves of a pipelined search: whatever leaves the routine between a _begin and its _end (fail() throws) must not leave a
// slot of the thread's context pending for ever — every later pipelined search of this thread would find "both view blobs hold a pending
// call".  Slots still >= 0 at scope exit are cancelled.
struct PendingHalves {
  int a = -1, b = -1;
  ~PendingHalves() {
    for (int s : {a, b})
      if (s >= 0) (void)orbx_target_search_view_cancel(ORBmatcher::DefaultContext(), s);
  }
};

void window_best(const char* routine, orbx_target* T, bool reprojection_gate, const Queries& Q, std::vector<int32_t>& bestIdx,
                 std::vector<int32_t>& bestDist) {
  const int nq = Q.size();
  bestIdx.assign(nq, -1); bestDist.assign(nq, 256);
  if (nq == 0 || orbx_target_size(T) == 0) return;
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  const int rc = orbx_target_nearest(ctx, T, reprojection_gate ? 1 : 0, Q.x.data(), Q.y.data(), Q.r.data(), Q.lo.data(), Q.hi.data(),
                                     reprojection_gate ? Q.aux.data() : nullptr, Q.desc.data(), nq, bestIdx.data(), bestDist.data());
  if (rc != ORBX_OK) fail(routine, ctx);
}

// feature vectors of two frames walked node by node (the while-loop shared by the three vocabulary routines, e.g. :244-262,
// :396-408): fn(indices1, indices2) for every common node, ascending
template <class Fn>
void common_nodes(const DBoW2::FeatureVector& a, const DBoW2::FeatureVector& b, Fn fn) {
  DBoW2::FeatureVector::const_iterator ia = a.begin(), ib = b.begin();
  while (ia != a.end() && ib != b.end()) {
    if (ia->first == ib->first) { fn(ia->second, ib->second); ++ia; ++ib; }
    else if (ia->first < ib->first) ia = a.lower_bound(ib->first);
    else ib = b.lower_bound(ia->first);
  }
}

// The node-by-node searches (SearchByBoW x 2, SearchForTriangulation) compare every feature of a vocabulary node on one side with every
// feature of the same node on the other; their replays only ever USE candidates within a distance bound.  orbx_nn_groups takes the groups
// as they are (no per-query repetition of the candidate lists) and returns, per query, the candidates within the bound in list order.
struct NodeGroups {
  std::vector<int32_t> q_group, group_ptr, group_cand;   // group_ptr starts with 0
  std::vector<unsigned char> qd;
  NodeGroups() { group_ptr.push_back(0); }
  int open_group() const { return (int)group_ptr.size() - 1; }           // the group whose candidates are being appended
  void close_group() { group_ptr.push_back((int32_t)group_cand.size()); }
  void add_query(const unsigned char* d) { q_group.push_back(open_group()); qd.insert(qd.end(), d, d + 32); }
  int nq() const { return (int)q_group.size(); }
};
struct NearLists {
  std::vector<int32_t> off, cnt;
  std::vector<orbx_candidate> ent;
  int begin(int q) const { return off[q]; }
  int end(int q) const { return off[q] + cnt[q]; }
};
// the distance beyond which `(float)best < ratio * (float)second` holds for every best <= th: a second-best candidate further away than
// this cannot fail the ratio test, so the replay may treat it as absent (second = 256, as when the list has no second entry)
int ratio_bound(float ratio, int th) {
  int t = th;
  while (t < 255 && !(static_cast<float>(th) < ratio * static_cast<float>(t + 1))) t++;
  return t;
}
void near_lists(const char* routine, const NodeGroups& G, const cv::Mat& trainDescriptors, int max_dist, NearLists& out) {
  const int nq = G.nq();
  out.off.assign(nq + 1, 0); out.cnt.assign(nq + 1, 0); out.ent.clear();
  if (nq == 0) return;
  DescView D(trainDescriptors);
  orbx_ctx* ctx = ORBmatcher::DefaultContext();
  int cap = 8 * nq + 4096;
  for (int attempt = 0; attempt < 2; attempt++) {
    out.ent.resize(cap);
    int n = 0;
    const int rc = orbx_nn_groups(ctx, G.qd.data(), G.q_group.data(), nq, D.p, trainDescriptors.rows, G.group_ptr.data(), G.group_cand.data(),
                                  G.open_group(), max_dist, out.off.data(), out.cnt.data(), out.ent.data(), cap, &n);
    if (rc == ORBX_OK) { out.ent.resize(n); return; }
    if (rc != ORBX_E_CAPACITY || attempt == 1) fail(routine, ctx);
    cap = n + 64;   // the pass reported how many entries it needs
  }
}

// rotation-consistency histogram (e.g. :345-352): bin = round(rot / 30) over rot in [0, 360)
struct RotHist {
  std::vector<int>* bins;   // [30], per-thread storage reused from call to call (the reference reserves 30 x 500 ints per call)
  RotHist() {
    static thread_local std::vector<int> store[30];
    bins = store;
    for (int i = 0; i < 30; i++) { store[i].clear(); if (store[i].capacity() < 500) store[i].reserve(500); }
  }
  void add(float angle1, float angle2, int what) {
    const float factor = 1.0f / ORBmatcher::HISTO_LENGTH;
    float rot = angle1 - angle2;
    if (rot < 0.0) rot += 360.0f;
    int bin = round(rot * factor);
    if (bin == ORBmatcher::HISTO_LENGTH) bin = 0;
    if (bin >= 0 && bin < ORBmatcher::HISTO_LENGTH) bins[bin].push_back(what);
  }
};

}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

orbx_ctx* ORBmatcher::DefaultContext() {
  ContextHolder& h = holder();
  if (!h.c && orbx_create(&h.c, 1, 1.2f, 1, 20, 7, -1) != ORBX_OK) {
    h.c = nullptr;
    throw std::runtime_error("ORBmatcher: no MI355X / HIP device");
  }
  return h.c;
}

void ORBmatcher::InvalidateTargets() { holder().cache.invalidate(); }
unsigned long ORBmatcher::RecycledIdsSeen() { return holder().cache.recycled_ids; }

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbx_hamming(a.ptr<unsigned char>(), b.ptr<unsigned char>()); }

void ORBmatcher::KnnMatch2(const cv::Mat& queryDesc, const cv::Mat& trainDesc, std::vector<int>& idx, std::vector<int>& dist) {
  idx.assign((size_t)queryDesc.rows * 2, -1);
  dist.assign((size_t)queryDesc.rows * 2, 256);
  if (queryDesc.rows == 0) return;
  DescView q(queryDesc), t(trainDesc);
  orbx_ctx* ctx = DefaultContext();
  if (orbx_knn2_allpairs(ctx, q.p, queryDesc.rows, t.p, trainDesc.rows, idx.data(), dist.data()) != ORBX_OK) fail("KnnMatch2", ctx);
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) {   // :212-218
  if (viewCos > 0.998) return 2.5;
  else return 4.0;
}

void ORBmatcher::ComputeThreeMaxima(vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {   // :2012-2053
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByProjection(Frame&, map points)  :43-210
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints) {
  int nmatches = 0;
  PhaseTrace tr("SearchByProjection(F, points)");
  const bool bFactor = th != 1.0;
  const bool rig = F.Nleft != -1;
  const int nMP = (int)vpMapPoints.size();
  // phase 1 (:49-74, :143-149): one window per map point and camera it is predicted in
  Queries QL, QR;
  QL.reserve(nMP);
  std::vector<int> qLeft(nMP, -1), qRight(nMP, -1);
  auto prepass = [&](int iMP0, int iMP1) {
  for (int iMP = iMP0; iMP < iMP1; iMP++) {
    MapPoint* pMP = vpMapPoints[iMP];
    if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
    if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
    if (pMP->isBad()) continue;
    if (pMP->mbTrackInView) {
      const int& nPredictedLevel = pMP->mnTrackScaleLevel;
      float r = RadiusByViewingCos(pMP->mTrackViewCos);
      if (bFactor) r *= th;
      qLeft[iMP] = QL.add(pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel,
                          pMP->GetDescriptor(), pMP->mTrackProjXR);
    }
    if (rig && pMP->mbTrackInViewR) {
      const int& nPredictedLevel = pMP->mnTrackScaleLevelR;
      if (nPredictedLevel != -1) {
        float r = RadiusByViewingCos(pMP->mTrackViewCosR);
        qRight[iMP] = QR.add(pMP->mTrackProjXR, pMP->mTrackProjYR, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel,
                             pMP->GetDescriptor());
      }
    }
  }
  };
  const std::vector<cv::KeyPoint>& keysL = rig ? F.mvKeys : F.mvKeysUn;
  Lists LR;
  // phase 3 (:76-140, :151-207): a keypoint bound to an observed map point — before the call or by an earlier map point of
  // this call — is no candidate.  LL holds the lists of the left queries [qbase, ...): position ql = q - qbase
  auto replay = [&](int iMP0, int iMP1, const Lists& LL, int qbase) {
  for (int iMP = iMP0; iMP < iMP1; iMP++) {
    if (qLeft[iMP] < 0 && qRight[iMP] < 0) continue;
    MapPoint* pMP = vpMapPoints[iMP];
    if (qLeft[iMP] >= 0) {
      const int q = qLeft[iMP], ql = q - qbase;
      if (LL.begin(ql) != LL.end(ql)) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        auto eligible = [&](size_t idx) {   // the gates of :89-101 on one candidate
          if (F.mvpMapPoints[idx])
            if (F.mvpMapPoints[idx]->Observations() > 0) return false;
          if (F.Nleft == -1 && F.mvuRight[idx] > 0) {
            const float er = fabs(pMP->mTrackProjXR - F.mvuRight[idx]);
            if (er > QL.r[q]) return false;
          }
          return true;
        };
        // the device pass found the two smallest (distance, list position) of the whole list: when both pass the gates here they are the
        // two smallest of the gated list as well (a list of one: the best alone), and the loop is not needed
        bool have = false;
        if (LL.has_best()) {
          const orbx_list_span& sp = LL.span(ql);
          if (sp.best_idx >= 0 && eligible((size_t)sp.best_idx) && (sp.count == 1 || (sp.second_idx >= 0 && eligible((size_t)sp.second_idx)))) {
            bestDist = sp.best_dist; bestIdx = sp.best_idx; bestLevel = keysL[sp.best_idx].octave;
            if (sp.count > 1) { bestDist2 = sp.second_dist; bestLevel2 = keysL[sp.second_idx].octave; }
            have = true;
          }
        }
        if (!have)
        for (int c = LL.begin(ql); c < LL.end(ql); c++) {
          const size_t idx = LL.cand[c];
          if (!eligible(idx)) continue;
          const int dist = LL.dist[c];
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = keysL[idx].octave; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = keysL[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;   // (also skips the right camera, like :125-126)
          if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
            F.mvpMapPoints[bestIdx] = pMP;
            if (F.Nleft != -1 && F.mvLeftToRightMatch[bestIdx] != -1) { F.mvpMapPoints[F.mvLeftToRightMatch[bestIdx] + F.Nleft] = pMP; nmatches++; }
            nmatches++;
          }
        }
      }
    }
    if (qRight[iMP] >= 0) {
      const int q = qRight[iMP];
      if (LR.begin(q) == LR.end(q)) continue;
      int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
      auto eligibleR = [&](size_t idx) {
        if (F.mvpMapPoints[idx + F.Nleft])
          if (F.mvpMapPoints[idx + F.Nleft]->Observations() > 0) return false;
        return true;
      };
      bool haveR = false;
      if (LR.has_best()) {   // as on the left: the two smallest of the whole list, when both pass the gate
        const orbx_list_span& sp = LR.span(q);
        if (sp.best_idx >= 0 && eligibleR((size_t)sp.best_idx) && (sp.count == 1 || (sp.second_idx >= 0 && eligibleR((size_t)sp.second_idx)))) {
          bestDist = sp.best_dist; bestIdx = sp.best_idx; bestLevel = F.mvKeysRight[sp.best_idx].octave;
          if (sp.count > 1) { bestDist2 = sp.second_dist; bestLevel2 = F.mvKeysRight[sp.second_idx].octave; }
          haveR = true;
        }
      }
      if (!haveR)
      for (int c = LR.begin(q); c < LR.end(q); c++) {
        const size_t idx = LR.cand[c];
        if (!eligibleR(idx)) continue;
        const int dist = LR.dist[c];
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.mvKeysRight[idx].octave; bestIdx = idx; }
        else if (dist < bestDist2) { bestLevel2 = F.mvKeysRight[idx].octave; bestDist2 = dist; }
      }
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
        if (F.Nleft != -1 && F.mvRightToLeftMatch[bestIdx] != -1) { F.mvpMapPoints[F.mvRightToLeftMatch[bestIdx]] = pMP; nmatches++; }
        F.mvpMapPoints[bestIdx + F.Nleft] = pMP;
        nmatches++;
      }
    }
  }
  };
  if (!rig && search_pipeline_enabled() && nMP >= 128) {
    // two halves, the host passes of one under the device pass of the other (the frame is resident already: SearchLocalPoints follows
    // TrackWithMotionModel / TrackReferenceKeyFrame on the same frame)
    orbx_target* TL = frame_target("SearchByProjection", F, false);
    tr.mark("target");
    const int mid = nMP / 2;
    prepass(0, mid);
    const int nA = QL.size();
    PendingHalves ph;
    ph.a = window_lists_begin("SearchByProjection", TL, QL, 0, nA);
    prepass(mid, nMP);
    const int nB = QL.size() - nA;
    if (nA + nB == 0) return 0;
    ph.b = window_lists_begin("SearchByProjection", TL, QL, nA, nB);
    tr.mark("prepass");
    Lists LA, LB;
    window_lists_end("SearchByProjection", ph.a, nA, LA);
    replay(0, mid, LA, 0);
    window_lists_end("SearchByProjection", ph.b, nB, LB);
    tr.mark("half");
    replay(mid, nMP, LB, nA);
    tr.mark("replay");
    return nmatches;
  }
  prepass(0, nMP);
  if (QL.size() == 0 && QR.size() == 0) return 0;
  tr.mark("prepass");
  // phase 2
  Lists LL;
  orbx_target* TL = QL.size() ? frame_target("SearchByProjection", F, false) : nullptr;
  orbx_target* TR = QR.size() ? frame_target("SearchByProjection", F, true) : nullptr;
  tr.mark("target");
  if (TL) window_lists("SearchByProjection", TL, QL, LL, /*view_ok=*/true);   // a rig keeps two lists alive: the two view blobs alternate
  if (TR) window_lists("SearchByProjection", TR, QR, LR, /*view_ok=*/true);
  tr.mark("device");
  replay(0, nMP, LL, 0);
  tr.mark("replay");
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByBoW(KeyFrame*, Frame&)  :223-425
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) {
  const vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches = vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
  int nmatches = 0;
  // phase 1: keyframe features with a good map point, node by node; candidates = the frame's features of the node (one list per node)
  std::vector<unsigned int> qKF;
  NodeGroups G;
  common_nodes(pKF->mFeatVec, F.mFeatVec, [&](const vector<unsigned int>& vIndicesKF, const vector<unsigned int>& vIndicesF) {
    bool any = false;
    for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
      const unsigned int realIdxKF = vIndicesKF[iKF];
      MapPoint* pMP = vpMapPointsKF[realIdxKF];
      if (!pMP) continue;
      if (pMP->isBad()) continue;
      qKF.push_back(realIdxKF);
      G.add_query(pKF->mDescriptors.ptr<unsigned char>((int)realIdxKF));
      any = true;
    }
    if (any) {
      for (size_t iF = 0; iF < vIndicesF.size(); iF++) G.group_cand.push_back((int32_t)vIndicesF[iF]);
      G.close_group();
    }
  });
  const int nq = (int)qKF.size();
  if (nq == 0) return 0;
  // phase 2: per query the frame features of its node within the distance the replay can tell apart — a match needs best <= TH_LOW, and a
  // second-best beyond ratio_bound passes the ratio test whatever its value
  NearLists NL;
  near_lists("SearchByBoW", G, F.mDescriptors, ratio_bound(mfNNratio, TH_LOW), NL);
  // phase 3 (:264-391): a frame feature taken by an earlier keyframe feature is no candidate
  RotHist rot;
  auto kfKey = [&](unsigned int i) -> const cv::KeyPoint& {
    return (!pKF->mpCamera2) ? pKF->mvKeysUn[i] : (i >= (unsigned int)pKF->NLeft) ? pKF->mvKeysRight[i - pKF->NLeft] : pKF->mvKeys[i];
  };
  for (int q = 0; q < nq; q++) {
    const unsigned int realIdxKF = qKF[q];
    MapPoint* pMP = vpMapPointsKF[realIdxKF];
    int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
    int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
    for (int c = NL.begin(q); c < NL.end(q); c++) {
      const unsigned int realIdxF = NL.ent[c].idx;
      if (vpMapPointMatches[realIdxF]) continue;
      const int d = NL.ent[c].dist;
      if (F.Nleft == -1) {
        if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = realIdxF; }
        else if (d < bestDist2) { bestDist2 = d; }
      } else {
        if (realIdxF < (unsigned int)F.Nleft && d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = realIdxF; }
        else if (realIdxF < (unsigned int)F.Nleft && d < bestDist2) { bestDist2 = d; }
        if (realIdxF >= (unsigned int)F.Nleft && d < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = d; bestIdxFR = realIdxF; }
        else if (realIdxF >= (unsigned int)F.Nleft && d < bestDist2R) { bestDist2R = d; }
      }
    }
    if (bestDist1 <= TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMapPointMatches[bestIdxF] = pMP;
        if (mbCheckOrientation) {
          const cv::KeyPoint& Fkp = (!pKF->mpCamera2 || F.Nleft == -1) ? F.mvKeys[bestIdxF]
                                    : (bestIdxF >= F.Nleft) ? F.mvKeysRight[bestIdxF - F.Nleft] : F.mvKeys[bestIdxF];
          rot.add(kfKey(realIdxKF).angle, Fkp.angle, bestIdxF);
        }
        nmatches++;
      }
      if (bestDist1R <= TH_LOW) {   // the right camera's best is accepted without the ratio test (`|| true`, :359)
        vpMapPointMatches[bestIdxFR] = pMP;
        if (mbCheckOrientation) {
          const cv::KeyPoint& Fkp = (!F.mpCamera2) ? F.mvKeys[bestIdxFR] : (bestIdxFR >= F.Nleft) ? F.mvKeysRight[bestIdxFR - F.Nleft] : F.mvKeys[bestIdxFR];
          rot.add(kfKey(realIdxKF).angle, Fkp.angle, bestIdxFR);
        }
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) { vpMapPointMatches[rot.bins[i][j]] = static_cast<MapPoint*>(NULL); nmatches--; }
    }
  }
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByProjection(KeyFrame*, Sim3, points[, their keyframes])  :427-532, :534-646 — one body, two projections
// ---------------------------------------------------------------------------------------------------------------------

static int SearchByProjectionSim3(const char* routine, KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints,
                                  const vector<KeyFrame*>* vpPointsKFs, vector<MapPoint*>& vpMatched, vector<KeyFrame*>* vpMatchedKF, int th,
                                  float ratioHamming) {
  const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;   // the keyframe's pinhole intrinsics
  const Sophus::SE3f Tcw = rigid_part(Scw);
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
  spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
  // phase 1 (:445-491 / :553-610)
  Queries Q;
  std::vector<int> owner;
  for (int iMP = 0, iendMP = vpPoints.size(); iMP < iendMP; iMP++) {
    MapPoint* pMP = vpPoints[iMP];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    Eigen::Vector3f p3Dc = Tcw * p3Dw;
    if (p3Dc(2) < 0.0) continue;
    float u, v;
    if (!vpPointsKFs) {
      const Eigen::Vector2f uv = pKF->mpCamera->project(p3Dc);   // :461
      u = uv(0); v = uv(1);
    } else {
      const float invz = 1 / p3Dc(2);                            // :571-576
      const float x = p3Dc(0) * invz;
      const float y = p3Dc(1) * invz;
      u = fx * x + cx;
      v = fy * y + cy;
    }
    if (!pKF->IsInImage(u, v)) continue;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    Eigen::Vector3f PO = p3Dw - Ow;
    const float dist = PO.norm();
    if (dist < minDistance || dist > maxDistance) continue;
    Eigen::Vector3f Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist) continue;
    int nPredictedLevel = pMP->PredictScale(dist, pKF);
    const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
    Q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor());
    owner.push_back(iMP);
  }
  if (Q.size() == 0) return 0;
  // phase 2
  Lists L;
  window_lists(routine, keyframe_target(routine, pKF), Q, L);
  // phase 3 (:497-528): a keypoint matched before, or by an earlier point of this call, is taken
  int nmatches = 0;
  for (int q = 0; q < Q.size(); q++) {
    int bestDist = 256, bestIdx = -1;
    for (int c = L.begin(q); c < L.end(q); c++) {
      const size_t idx = L.cand[c];
      if (vpMatched[idx]) continue;
      const int dist = L.dist[c];
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= ORBmatcher::TH_LOW * ratioHamming) {
      vpMatched[bestIdx] = vpPoints[owner[q]];
      if (vpMatchedKF) (*vpMatchedKF)[bestIdx] = (*vpPointsKFs)[owner[q]];
      nmatches++;
    }
  }
  return nmatches;
}

int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th,
                                   float ratioHamming) {
  return SearchByProjectionSim3("SearchByProjection", pKF, Scw, vpPoints, nullptr, vpMatched, nullptr, th, ratioHamming);
}

int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                                   const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                                   std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming) {
  return SearchByProjectionSim3("SearchByProjection", pKF, Scw, vpPoints, &vpPointsKFs, vpMatched, &vpMatchedKF, th, ratioHamming);
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchForInitialization  :648-763
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize) {
  int nmatches = 0;
  vnMatches12 = vector<int>(F1.mvKeysUn.size(), -1);
  // phase 1: the level-0 keypoints of F1, each around its previous match
  Queries Q;
  std::vector<int> owner;
  for (size_t i1 = 0, iend1 = F1.mvKeysUn.size(); i1 < iend1; i1++) {
    const int level1 = F1.mvKeysUn[i1].octave;
    if (level1 > 0) continue;
    Q.add(vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, level1, level1, F1.mDescriptors.row(i1));
    owner.push_back((int)i1);
  }
  if (Q.size() == 0) return 0;
  // phase 2
  Lists L;
  window_lists("SearchForInitialization", frame_target("SearchForInitialization", F2, false), Q, L);
  // phase 3 (:676-763): a later keypoint may take a match from an earlier one
  RotHist rot;
  vector<int> vMatchedDistance(F2.mvKeysUn.size(), INT_MAX);
  vector<int> vnMatches21(F2.mvKeysUn.size(), -1);
  for (int q = 0; q < Q.size(); q++) {
    const int i1 = owner[q];
    if (L.begin(q) == L.end(q)) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int c = L.begin(q); c < L.end(q); c++) {
      const size_t i2 = L.cand[c];
      const int dist = L.dist[c];
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) rot.add(F1.mvKeysUn[i1].angle, F2.mvKeysUn[bestIdx2].angle, i1);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) {
        const int idx1 = rot.bins[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)
    if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByBoW(KeyFrame*, KeyFrame*)  :765-905
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) {
  const vector<cv::KeyPoint>& vKeysUn1 = pKF1->mvKeysUn;
  const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const vector<cv::KeyPoint>& vKeysUn2 = pKF2->mvKeysUn;
  const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
  vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
  vector<bool> vbMatched2(vpMapPoints2.size(), false);
  int nmatches = 0;
  // phase 1 (:797-833): features with a good map point on both sides (right-camera features of a rig are skipped)
  std::vector<size_t> q1;
  NodeGroups G;
  common_nodes(pKF1->mFeatVec, pKF2->mFeatVec, [&](const vector<unsigned int>& v1, const vector<unsigned int>& v2) {
    bool any = false;
    for (size_t i1 = 0, iend1 = v1.size(); i1 < iend1; i1++) {
      const size_t idx1 = v1[i1];
      if (pKF1->NLeft != -1 && idx1 >= pKF1->mvKeysUn.size()) continue;
      MapPoint* pMP1 = vpMapPoints1[idx1];
      if (!pMP1) continue;
      if (pMP1->isBad()) continue;
      q1.push_back(idx1);
      G.add_query(pKF1->mDescriptors.ptr<unsigned char>((int)idx1));
      any = true;
    }
    if (!any) return;
    for (size_t i2 = 0, iend2 = v2.size(); i2 < iend2; i2++) {   // the candidate filter does not depend on the query
      const size_t idx2 = v2[i2];
      if (pKF2->NLeft != -1 && idx2 >= pKF2->mvKeysUn.size()) continue;
      MapPoint* pMP2 = vpMapPoints2[idx2];
      if (!pMP2) continue;
      if (pMP2->isBad()) continue;
      G.group_cand.push_back((int32_t)idx2);
    }
    G.close_group();
  });
  const int nq = (int)q1.size();
  if (nq == 0) return 0;
  // phase 2
  NearLists NL;
  near_lists("SearchByBoW", G, pKF2->mDescriptors, ratio_bound(mfNNratio, TH_LOW), NL);
  // phase 3 (:821-867): strict `< TH_LOW` here
  RotHist rot;
  for (int q = 0; q < nq; q++) {
    const size_t idx1 = q1[q];
    int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
    for (int c = NL.begin(q); c < NL.end(q); c++) {
      const size_t idx2 = NL.ent[c].idx;
      if (vbMatched2[idx2]) continue;
      const int d = NL.ent[c].dist;
      if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = idx2; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    if (bestDist1 < TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMatches12[idx1] = vpMapPoints2[bestIdx2];
        vbMatched2[bestIdx2] = true;
        if (mbCheckOrientation) rot.add(vKeysUn1[idx1].angle, vKeysUn2[bestIdx2].angle, idx1);
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) { vpMatches12[rot.bins[i][j]] = static_cast<MapPoint*>(NULL); nmatches--; }
    }
  }
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchForTriangulation  :907-1146
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// The cameras of two keyframes relative to each other: (R, t)[a][b] takes camera b of the second keyframe into camera a of the first — a / b = 0
// for the left (or only) camera, 1 for the right camera of a two-camera rig — and the epipole: the first keyframe's camera centre seen by the
// second keyframe's left camera.  One table instead of the reference's five named transforms and the per-pair selection among them.
struct PairGeometry {
  Eigen::Matrix3f R[2][2];
  Eigen::Vector3f t[2][2];
  GeometricCamera* cam1[2];
  GeometricCamera* cam2[2];
  Eigen::Vector2f epipole;
  bool rig;
  PairGeometry(KeyFrame* k1, KeyFrame* k2) : rig(k1->mpCamera2 && k2->mpCamera2) {
    cam1[0] = k1->mpCamera; cam1[1] = k1->mpCamera2; cam2[0] = k2->mpCamera; cam2[1] = k2->mpCamera2;
    epipole = k2->mpCamera->project(k2->GetPose() * k1->GetCameraCenter());
    const int nc = rig ? 2 : 1;
    for (int a = 0; a < nc; a++)
      for (int b = 0; b < nc; b++) {
        const Sophus::SE3f T = (a ? k1->GetRightPose() : k1->GetPose()) * (b ? k2->GetRightPoseInverse() : k2->GetPoseInverse());
        R[a][b] = T.rotationMatrix();
        t[a][b] = T.translation();
      }
  }
};
}  // namespace

int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo,
                                       const bool bCoarse) {
  // relative geometry of the two keyframes' cameras (:913-943) as a table [camera of KF1][camera of KF2] (0 = left / only, 1 = right of a rig)
  const PairGeometry geo(pKF1, pKF2);
  const Eigen::Vector2f& ep = geo.epipole;
  // phase 1 (:963-996): features of KF1 without a map point against the ones of KF2 in the same node that have none either
  auto key1 = [&](size_t i) -> const cv::KeyPoint& {
    return (pKF1->NLeft == -1) ? pKF1->mvKeysUn[i] : (i < (size_t)pKF1->NLeft) ? pKF1->mvKeys[i] : pKF1->mvKeysRight[i - pKF1->NLeft];
  };
  auto key2 = [&](size_t i) -> const cv::KeyPoint& {
    return (pKF2->NLeft == -1) ? pKF2->mvKeysUn[i] : (i < (size_t)pKF2->NLeft) ? pKF2->mvKeys[i] : pKF2->mvKeysRight[i - pKF2->NLeft];
  };
  std::vector<size_t> q1;
  NodeGroups G;
  common_nodes(pKF1->mFeatVec, pKF2->mFeatVec, [&](const vector<unsigned int>& v1, const vector<unsigned int>& v2) {
    bool any = false;
    for (size_t i1 = 0, iend1 = v1.size(); i1 < iend1; i1++) {
      const size_t idx1 = v1[i1];
      if (pKF1->GetMapPoint(idx1)) continue;
      const bool bStereo1 = (!pKF1->mpCamera2 && pKF1->mvuRight[idx1] >= 0);
      if (bOnlyStereo)
        if (!bStereo1) continue;
      q1.push_back(idx1);
      G.add_query(pKF1->mDescriptors.ptr<unsigned char>((int)idx1));
      any = true;
    }
    if (!any) return;
    for (size_t i2 = 0, iend2 = v2.size(); i2 < iend2; i2++) {   // the candidate filter does not depend on the query
      const size_t idx2 = v2[i2];
      if (pKF2->GetMapPoint(idx2)) continue;   // (vbMatched2 is never set in this routine)
      const bool bStereo2 = (!pKF2->mpCamera2 && pKF2->mvuRight[idx2] >= 0);
      if (bOnlyStereo)
        if (!bStereo2) continue;
      G.group_cand.push_back((int32_t)idx2);
    }
    G.close_group();
  });
  const int nq = (int)q1.size();
  // phase 2: the replay skips every candidate beyond TH_LOW (:1013)
  NearLists NL;
  near_lists("SearchForTriangulation", G, pKF2->mDescriptors, TH_LOW, NL);
  // phase 3 (:1010-1100): among the candidates that pass the distance tests, the LAST one that also passes the epipole and
  // epipolar tests wins (`dist>bestDist -> continue`, then `<=` replaces)
  int nmatches = 0;
  vector<int> vMatches12(pKF1->N, -1);
  RotHist rot;
  for (int q = 0; q < nq; q++) {
    const size_t idx1 = q1[q];
    const bool bStereo1 = (!pKF1->mpCamera2 && pKF1->mvuRight[idx1] >= 0);
    const cv::KeyPoint& kp1 = key1(idx1);
    const bool bRight1 = (pKF1->NLeft == -1 || idx1 < (size_t)pKF1->NLeft) ? false : true;
    int bestDist = TH_LOW;
    int bestIdx2 = -1;
    for (int c = NL.begin(q); c < NL.end(q); c++) {
      const size_t idx2 = NL.ent[c].idx;
      const int d = NL.ent[c].dist;
      if (d > TH_LOW || d > bestDist) continue;
      const bool bStereo2 = (!pKF2->mpCamera2 && pKF2->mvuRight[idx2] >= 0);
      const cv::KeyPoint& kp2 = key2(idx2);
      const bool bRight2 = (pKF2->NLeft == -1 || idx2 < (size_t)pKF2->NLeft) ? false : true;
      if (!bStereo1 && !bStereo2 && !pKF1->mpCamera2) {
        const float distex = ep(0) - kp2.pt.x;
        const float distey = ep(1) - kp2.pt.y;
        if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) continue;
      }
      const int c1 = geo.rig && bRight1, c2 = geo.rig && bRight2;   // which camera of each keyframe holds the two keypoints (:1066-1095)
      if (bCoarse || geo.cam1[c1]->epipolarConstrain(geo.cam2[c2], kp1, kp2, geo.R[c1][c2], geo.t[c1][c2], pKF1->mvLevelSigma2[kp1.octave],
                                                     pKF2->mvLevelSigma2[kp2.octave])) {
        bestIdx2 = idx2;
        bestDist = d;
      }
    }
    if (bestIdx2 >= 0) {
      const cv::KeyPoint& kp2 = key2(bestIdx2);
      vMatches12[idx1] = bestIdx2;
      nmatches++;
      if (mbCheckOrientation) rot.add(kp1.angle, kp2.angle, idx1);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) { vMatches12[rot.bins[i][j]] = -1; nmatches--; }
    }
  }
  vMatchedPairs.clear();
  vMatchedPairs.reserve(nmatches);
  for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
    if (vMatches12[i] < 0) continue;
    vMatchedPairs.push_back(make_pair(i, vMatches12[i]));
  }
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fuse(KeyFrame*, map points, th, bRight)  :1148-1338 — the arg-min comes from the device, nothing is replayed
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th, const bool bRight) {
  GeometricCamera* pCamera;
  Sophus::SE3f Tcw;
  Eigen::Vector3f Ow;
  if (bRight) { Tcw = pKF->GetRightPose(); Ow = pKF->GetRightCameraCenter(); pCamera = pKF->mpCamera2; }
  else { Tcw = pKF->GetPose(); Ow = pKF->GetCameraCenter(); pCamera = pKF->mpCamera; }
  const float& bf = pKF->mbf;
  const int nMPs = vpMapPoints.size();
  // the geometry of one point (:1193-1243); false when a test of the loop head rejects it
  struct Proj { float u, v, ur, radius; int level; };
  auto project = [&](MapPoint* pMP, Proj& o) -> bool {
    Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    Eigen::Vector3f p3Dc = Tcw * p3Dw;
    if (p3Dc(2) < 0.0f) return false;
    const float invz = 1 / p3Dc(2);
    const Eigen::Vector2f uv = pCamera->project(p3Dc);
    if (!pKF->IsInImage(uv(0), uv(1))) return false;
    const float ur = uv(0) - bf * invz;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    Eigen::Vector3f PO = p3Dw - Ow;
    const float dist3D = PO.norm();
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    Eigen::Vector3f Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist3D) return false;
    int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
    o.u = uv(0); o.v = uv(1); o.ur = ur; o.level = nPredictedLevel;
    o.radius = th * pKF->mvScaleFactors[nPredictedLevel];
    return true;
  };
  // phase 1: every point that passes the loop head NOW.  A point that fails isBad() / IsInKeyFrame() now fails them later too
  // (both only ever become true); the tests are repeated at its turn in phase 3 because the fusions before it may make them true.
  Queries Q;
  std::vector<int> qOf(nMPs, -1);
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP) continue;
    if (pMP->isBad()) continue;
    else if (pMP->IsInKeyFrame(pKF)) continue;
    Proj p;
    if (!project(pMP, p)) continue;
    qOf[i] = Q.add(p.u, p.v, p.radius, p.level - 1, p.level, pMP->GetDescriptor(), p.ur);
  }
  if (Q.size() == 0) return 0;
  // phase 2: window (KeyFrame::GetFeaturesInArea(..., bRight)), level and reprojection gates (:1262-1296), arg-min
  orbx_target* target = fuse_target("Fuse", pKF, bRight);
  std::vector<int32_t> bestIdx, bestDist;
  window_best("Fuse", target, true, Q, bestIdx, bestDist);
  // phase 3 (:1176-1192, :1311-1335): the map is changed point by point, in order
  std::set<MapPoint*> touched;   // points whose descriptor a Replace() of this call recomputed
  int nFused = 0;
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP) continue;
    if (pMP->isBad()) continue;
    else if (pMP->IsInKeyFrame(pKF)) continue;
    if (qOf[i] < 0) continue;
    int best = bestIdx[qOf[i]], bestD = bestDist[qOf[i]];
    if (touched.count(pMP)) {   // its descriptor changed since phase 1: redo this one query
      Proj p;
      if (!project(pMP, p)) continue;
      Queries one;
      one.add(p.u, p.v, p.radius, p.level - 1, p.level, pMP->GetDescriptor(), p.ur);
      std::vector<int32_t> bi, bd;
      window_best("Fuse", target, true, one, bi, bd);
      best = bi[0]; bestD = bd[0];
    }
    if (bestD <= TH_LOW) {
      const int bestIdxKF = bRight ? best + pKF->NLeft : best;   // (`if(bRight) idx += pKF->NLeft`, :1298)
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdxKF);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) { pMP->Replace(pMPinKF); touched.insert(pMPinKF); }
          else { pMPinKF->Replace(pMP); touched.insert(pMP); }
        }
      } else {
        pMP->AddObservation(pKF, bestIdxKF);
        pKF->AddMapPoint(pMP, bestIdxKF);
      }
      nFused++;
    }
  }
  return nFused;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fuse(KeyFrame*, Sim3, points, th, vpReplacePoint)  :1340-1455
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint) {
  const Sophus::SE3f Tcw = rigid_part(Scw);
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  const set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
  const int nPoints = vpPoints.size();
  // phase 1 (:1358-1405)
  Queries Q;
  std::vector<int> owner;
  for (int iMP = 0; iMP < nPoints; iMP++) {
    MapPoint* pMP = vpPoints[iMP];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    Eigen::Vector3f p3Dc = Tcw * p3Dw;
    if (p3Dc(2) < 0.0f) continue;
    const Eigen::Vector2f uv = pKF->mpCamera->project(p3Dc);
    if (!pKF->IsInImage(uv(0), uv(1))) continue;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    Eigen::Vector3f PO = p3Dw - Ow;
    const float dist3D = PO.norm();
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    Eigen::Vector3f Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist3D) continue;
    const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
    const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
    Q.add(uv(0), uv(1), radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor());
    owner.push_back(iMP);
  }
  if (Q.size() == 0) return 0;
  // phase 2 (:1411-1433)
  std::vector<int32_t> bestIdx, bestDist;
  window_best("Fuse", keyframe_target("Fuse", pKF), false, Q, bestIdx, bestDist);
  // phase 3 (:1436-1451)
  int nFused = 0;
  for (int q = 0; q < Q.size(); q++) {
    if (bestDist[q] <= TH_LOW) {
      const int iMP = owner[q];
      MapPoint* pMP = vpPoints[iMP];
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[q]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, bestIdx[q]);
        pKF->AddMapPoint(pMP, bestIdx[q]);
      }
      nFused++;
    }
  }
  return nFused;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchBySim3  :1457-1674
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th) {
  const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
  Sophus::SE3f T1w = pKF1->GetPose();
  Sophus::SE3f T2w = pKF2->GetPose();
  Sophus::Sim3f S21 = S12.inverse();
  const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const int N1 = vpMapPoints1.size();
  const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N2 = vpMapPoints2.size();
  vector<bool> vbAlreadyMatched1(N1, false);
  vector<bool> vbAlreadyMatched2(N2, false);
  for (int i = 0; i < N1; i++) {
    MapPoint* pMP = vpMatches12[i];
    if (pMP) {
      vbAlreadyMatched1[i] = true;
      int idx2 = get<0>(pMP->GetIndexInKeyFrame(pKF2));
      if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
    }
  }
  // one direction: the points of `from` (seen from camera Tfw) moved by S into the camera of `to` (:1493-1573 / :1576-1653)
  auto direction = [&](const vector<MapPoint*>& points, const vector<bool>& already, const Sophus::SE3f& Tfw, const Sophus::Sim3f& S,
                       KeyFrame* pTo, vector<int>& vnMatch) {
    Queries Q;
    std::vector<int> owner;
    for (int i = 0, n = points.size(); i < n; i++) {
      MapPoint* pMP = points[i];
      if (!pMP || already[i]) continue;
      if (pMP->isBad()) continue;
      Eigen::Vector3f p3Dw = pMP->GetWorldPos();
      Eigen::Vector3f p3Dcf = Tfw * p3Dw;
      Eigen::Vector3f p3Dct = S * p3Dcf;
      if (p3Dct(2) < 0.0) continue;
      const float invz = 1.0 / p3Dct(2);
      const float x = p3Dct(0) * invz;
      const float y = p3Dct(1) * invz;
      const float u = fx * x + cx;
      const float v = fy * y + cy;
      if (!pTo->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance();
      const float minDistance = pMP->GetMinDistanceInvariance();
      const float dist3D = p3Dct.norm();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, pTo);
      const float radius = th * pTo->mvScaleFactors[nPredictedLevel];
      Q.add(u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor());
      owner.push_back(i);
    }
    if (Q.size() == 0) return;
    std::vector<int32_t> bestIdx, bestDist;
    window_best("SearchBySim3", keyframe_target("SearchBySim3", pTo), false, Q, bestIdx, bestDist);
    for (int q = 0; q < Q.size(); q++)
      if (bestDist[q] <= TH_HIGH) vnMatch[owner[q]] = bestIdx[q];
  };
  vector<int> vnMatch1(N1, -1);
  vector<int> vnMatch2(N2, -1);
  direction(vpMapPoints1, vbAlreadyMatched1, T1w, S21, pKF2, vnMatch1);
  direction(vpMapPoints2, vbAlreadyMatched2, T2w, S12, pKF1, vnMatch2);
  // agreement of the two directions (:1655-1673)
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      int idx1 = vnMatch2[idx2];
      if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
    }
  }
  return nFound;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByProjection(CurrentFrame, LastFrame, th, bMono)  :1676-1885
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
  int nmatches = 0;
  PhaseTrace tr("SearchByProjection(Cur, Last)");
  const Sophus::SE3f Tcw = CurrentFrame.GetPose();
  const Eigen::Vector3f twc = Tcw.inverse().translation();
  const Sophus::SE3f Tlw = LastFrame.GetPose();
  const Eigen::Vector3f tlc = Tlw * twc;
  const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;
  const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
  const bool rig = CurrentFrame.Nleft != -1;
  // the current frame becomes resident FIRST: its upload and the device-side grid build are asynchronous and run under the host
  // pre-pass below (TrackWithMotionModel is the first search of a new frame)
  orbx_target* TL = frame_target("SearchByProjection", CurrentFrame, false);
  orbx_target* TR = rig ? frame_target("SearchByProjection", CurrentFrame, true) : nullptr;
  tr.mark("target");
  // phase 1 (:1694-1733, :1792-1809): project the last frame's map points into the current camera(s)
  Queries QL, QR;
  QL.reserve(LastFrame.N);
  std::vector<int> qLeft(LastFrame.N, -1), qRight(LastFrame.N, -1);
  auto prepass = [&](int i0, int i1) {
  for (int i = i0; i < i1; i++) {
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP) continue;
    if (LastFrame.mvbOutlier[i]) continue;
    Eigen::Vector3f x3Dw = pMP->GetWorldPos();
    Eigen::Vector3f x3Dc = Tcw * x3Dw;
    const float invzc = 1.0 / x3Dc(2);
    if (invzc < 0) continue;
    Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
    if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
    if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
    int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
    float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
    const int lo = bForward ? nLastOctave : bBackward ? 0 : nLastOctave - 1;
    const int hi = bForward ? -1 : bBackward ? nLastOctave : nLastOctave + 1;
    const float ur = uv(0) - CurrentFrame.mbf * invzc;
    qLeft[i] = QL.add(uv(0), uv(1), radius, lo, hi, pMP->GetDescriptor(), ur);
    if (rig) {
      Eigen::Vector3f x3Dr = CurrentFrame.GetRelativePoseTrl() * x3Dc;
      Eigen::Vector2f uvr = CurrentFrame.mpCamera->project(x3Dr);
      qRight[i] = QR.add(uvr(0), uvr(1), radius, lo, hi, pMP->GetDescriptor());
    }
  }
  };
  Lists LR;
  // phase 3 (:1735-1858).  LL holds the lists of the left queries [qbase, ...): position ql = q - qbase
  RotHist rot;
  auto lastKey = [&](int i) -> const cv::KeyPoint& {
    return (LastFrame.Nleft == -1) ? LastFrame.mvKeysUn[i] : (i < LastFrame.Nleft) ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
  };
  auto replay = [&](int i0, int i1, const Lists& LL, int qbase) {
  for (int i = i0; i < i1; i++) {
    if (qLeft[i] < 0) continue;
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    const int q = qLeft[i], ql = q - qbase;
    if (LL.begin(ql) == LL.end(ql)) continue;   // (`if(vIndices2.empty()) continue;` skips the right camera too, :1735-1736)
    int bestDist = 256, bestIdx2 = -1;
    // the gates of :1741-1760 on one candidate
    auto eligible = [&](size_t i2) {
      if (CurrentFrame.mvpMapPoints[i2])
        if (CurrentFrame.mvpMapPoints[i2]->Observations() > 0) return false;
      if (CurrentFrame.Nleft == -1 && CurrentFrame.mvuRight[i2] > 0) {
        const float er = fabs(QL.aux[q] - CurrentFrame.mvuRight[i2]);
        if (er > QL.r[q]) return false;
      }
      return true;
    };
    // The device pass already found the first minimum of the whole list; if that candidate passes the gates here — which depend on
    // what earlier points of this loop took — it is also the first minimum of the gated list and the loop is not needed.
    if (LL.has_best() && LL.span(ql).best_idx >= 0 && eligible((size_t)LL.span(ql).best_idx)) {
      bestDist = LL.span(ql).best_dist; bestIdx2 = LL.span(ql).best_idx;
    } else
    for (int c = LL.begin(ql); c < LL.end(ql); c++) {
      const size_t i2 = LL.cand[c];
      if (!eligible(i2)) continue;
      const int dist = LL.dist[c];
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      CurrentFrame.mvpMapPoints[bestIdx2] = pMP;
      nmatches++;
      if (mbCheckOrientation) {
        const cv::KeyPoint& kpCF = (CurrentFrame.Nleft == -1) ? CurrentFrame.mvKeysUn[bestIdx2]
                                   : (bestIdx2 < CurrentFrame.Nleft) ? CurrentFrame.mvKeys[bestIdx2] : CurrentFrame.mvKeysRight[bestIdx2 - CurrentFrame.Nleft];
        rot.add(lastKey(i).angle, kpCF.angle, bestIdx2);
      }
    }
    if (rig) {
      const int qr = qRight[i];
      int bestDistR = 256, bestIdxR = -1;
      auto eligibleR = [&](size_t i2) {
        if (CurrentFrame.mvpMapPoints[i2 + CurrentFrame.Nleft])
          if (CurrentFrame.mvpMapPoints[i2 + CurrentFrame.Nleft]->Observations() > 0) return false;
        return true;
      };
      // as on the left: the first minimum of the whole list, when it passes the gate, is the first minimum of the gated list
      if (LR.has_best() && LR.span(qr).best_idx >= 0 && eligibleR((size_t)LR.span(qr).best_idx)) {
        bestDistR = LR.span(qr).best_dist; bestIdxR = LR.span(qr).best_idx;
      } else
      for (int c = LR.begin(qr); c < LR.end(qr); c++) {
        const size_t i2 = LR.cand[c];
        if (!eligibleR(i2)) continue;
        const int dist = LR.dist[c];
        if (dist < bestDistR) { bestDistR = dist; bestIdxR = i2; }
      }
      if (bestDistR <= TH_HIGH) {
        CurrentFrame.mvpMapPoints[bestIdxR + CurrentFrame.Nleft] = pMP;
        nmatches++;
        if (mbCheckOrientation) rot.add(lastKey(i).angle, CurrentFrame.mvKeysRight[bestIdxR].angle, bestIdxR + CurrentFrame.Nleft);
      }
    }
  }
  };
  const int N = LastFrame.N;
  if (!rig && search_pipeline_enabled() && N >= 128) {
    // two halves, the host passes of one under the device pass of the other (ORBmatcher.cc's window_lists_begin / _end)
    const int mid = N / 2;
    prepass(0, mid);
    const int nA = QL.size();
    PendingHalves ph;
    ph.a = window_lists_begin("SearchByProjection", TL, QL, 0, nA);
    prepass(mid, N);
    const int nB = QL.size() - nA;
    if (nA + nB == 0) return 0;
    ph.b = window_lists_begin("SearchByProjection", TL, QL, nA, nB);
    tr.mark("prepass");
    Lists LA, LB;
    window_lists_end("SearchByProjection", ph.a, nA, LA);
    replay(0, mid, LA, 0);
    window_lists_end("SearchByProjection", ph.b, nB, LB);
    tr.mark("half");
    replay(mid, N, LB, nA);
  } else {
    prepass(0, N);
    if (QL.size() == 0) return 0;
    tr.mark("prepass");
    // phase 2
    Lists LL;
    window_lists("SearchByProjection", TL, QL, LL, /*view_ok=*/true);   // two view blobs alternate: the left lists stay readable
    if (rig) window_lists("SearchByProjection", TR, QR, LR, /*view_ok=*/true);
    tr.mark("device");
    replay(0, N, LL, 0);
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) { CurrentFrame.mvpMapPoints[rot.bins[i][j]] = static_cast<MapPoint*>(NULL); nmatches--; }
      }
    }
  }
  tr.mark("replay");
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------------
// SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist)  :1887-2010
// ---------------------------------------------------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {
  int nmatches = 0;
  const Sophus::SE3f Tcw = CurrentFrame.GetPose();
  Eigen::Vector3f Ow = Tcw.inverse().translation();
  const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
  // phase 1 (:1904-1934)
  Queries Q;
  std::vector<int> owner;
  for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
    MapPoint* pMP = vpMPs[i];
    if (!pMP) continue;
    if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
    Eigen::Vector3f x3Dw = pMP->GetWorldPos();
    Eigen::Vector3f x3Dc = Tcw * x3Dw;
    const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
    if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
    if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
    Eigen::Vector3f PO = x3Dw - Ow;
    float dist3D = PO.norm();
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
    const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
    Q.add(uv(0), uv(1), radius, nPredictedLevel - 1, nPredictedLevel + 1, pMP->GetDescriptor());
    owner.push_back((int)i);
  }
  if (Q.size() == 0) return 0;
  // phase 2
  Lists L;
  window_lists("SearchByProjection", frame_target("SearchByProjection", CurrentFrame, false), Q, L);
  // phase 3 (:1941-1982): any keypoint that already has a map point is taken
  RotHist rot;
  for (int q = 0; q < Q.size(); q++) {
    const int i = owner[q];
    int bestDist = 256, bestIdx2 = -1;
    for (int c = L.begin(q); c < L.end(q); c++) {
      const size_t i2 = L.cand[c];
      if (CurrentFrame.mvpMapPoints[i2]) continue;
      const int dist = L.dist[c];
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= ORBdist) {
      CurrentFrame.mvpMapPoints[bestIdx2] = vpMPs[i];
      nmatches++;
      if (mbCheckOrientation) rot.add(pKF->mvKeysUn[i].angle, CurrentFrame.mvKeysUn[bestIdx2].angle, bestIdx2);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rot.bins, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rot.bins[i].size(); j < jend; j++) { CurrentFrame.mvpMapPoints[rot.bins[i][j]] = NULL; nmatches--; }
      }
    }
  }
  return nmatches;
}

}  // namespace ORB_SLAM3
