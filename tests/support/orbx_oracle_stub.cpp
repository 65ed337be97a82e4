// TEST INFRASTRUCTURE: the C-ABI entry points the drop-in ORBmatcher (csrc/ref_adapter/ORBmatcher.cc) — and, for
// tools/streamed_frontend.cpp, the extractor / vocabulary adapters of include/ — call, served by the
// CPU oracle (oracle/liborb_oracle.so) instead of liborbx.so.  It exists so that the HOST logic of the drop-in — the
// geometry pre-passes, the replays, the bookkeeping — can be compared with the reference's src/ORBmatcher.cc in the CPU
// suite (tests/test_matcher_world.py::test_dropin_host_logic_equals_reference).  Never linked into the product; the GPU test
// of the same scenarios links the real liborbx.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "orbx.h"

extern "C" {
int mo_hamming(const uint8_t* a, const uint8_t* b);
void mo_nn_csr(const uint8_t* q, int nq, const uint8_t* t, const int32_t* row_ptr, const int32_t* cand, int last_wins, int32_t* best_idx,
               int32_t* best_dist, int32_t* second_idx, int32_t* second_dist, int32_t* dist_out);
void mo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
int mo_window_search_grid(const void* kps, const uint8_t* desc, int n, const void* grid, const uint8_t* kp_skip, const float* kp_uright,
                          const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, const uint8_t* q_desc,
                          const float* q_xr, int nq, int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx,
                          int32_t* best_dist, int32_t* second_idx, int32_t* second_dist);
void mo_window_nearest(const void* kps, const uint8_t* desc, int n, const void* grid, const float* kp_uright, const float* inv_level_sigma2,
                       const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, const float* q_ur,
                       const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist);
}

extern "C" {
void* orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);
void orbo_destroy(void* h);
int orbo_level_size(void* h, int level, int* w, int* hgt);
int orbo_level_copy(void* h, int level, int blurred, uint8_t* dst, int dst_stride);
int orbo_extract(void* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, void* kps, uint8_t* desc, int cap, int* n_out,
                 int* mono_out);
void orbo_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* quota, int* umax16);
void* mo_voc_load(const char* path);
void mo_voc_free(void* h);
void mo_voc_descend(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node);
int mo_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* ids, double* vals, uint32_t* fv_node, uint32_t* fv_feat, int* n_fv);
double mo_score_l1(const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb);
void* mo_kfdb_new();
void mo_kfdb_free(void* h);
void mo_kfdb_add(void* h, long kf_id, const uint32_t* ids, const double* vals, int n);
void mo_kfdb_erase(void* h, long kf_id);
int mo_kfdb_query(void* h, const uint32_t* q_ids, const double* q_vals, int nq, const long* exclude, int n_exclude, int nMinWords, long* out_kf,
                  int32_t* out_words, double* out_score, int* maxCommon, int* minCommon);
}

#include <map>
#include <utility>

struct orbx_ctx { void* ora = nullptr; int nfeatures = 0, nlevels = 0; std::vector<std::vector<uint8_t> > pyr; std::vector<int> pw, ph; };
extern "C" int mo_stereo_matches(const void* kpsL, const uint8_t* descL, int N, const void* kpsR, const uint8_t* descR, int Nr, const uint8_t* const* pyrL,
                                 const uint8_t* const* pyrR, const int32_t* w, const int32_t* h, const int32_t* pitch, const float* scale,
                                 const float* inv_scale, float mb, float mbf, float* mvuRight, float* mvDepth);
struct orbx_kfdb { void* h = nullptr; std::map<int64_t, std::pair<std::vector<uint32_t>, std::vector<double> > > rows; };
struct orbx_voc { void* h = nullptr; std::vector<uint8_t> last_desc; int last_levelsup = 0; };

extern "C" {

int orbx_create(orbx_ctx** out, int nfeatures, float sf, int nlevels, int iniTh, int minTh, int) {
  orbx_ctx* c = new orbx_ctx();
  c->ora = orbo_create(nfeatures, sf, nlevels, iniTh, minTh); c->nfeatures = nfeatures; c->nlevels = nlevels;
  *out = c;
  return ORBX_OK;
}
void orbx_destroy(orbx_ctx* c) { if (c) { orbo_destroy(c->ora); delete c; } }
// ---- extractor / vocabulary adapters (include/ORBextractor.h, ORBVocabulary.h) over the oracle
int orbx_keypoint_capacity(const orbx_ctx* c) { return c->nfeatures + 35 * c->nlevels + 64; }
int orbx_scale_tables(const orbx_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* quota) {
  orbo_tables(c->ora, scale, inv_scale, sigma2, inv_sigma2, quota, nullptr);
  return ORBX_OK;
}
int orbx_set_host_pyramid(orbx_ctx*, int) { return ORBX_OK; }
// the adapter's OpenCV calibration (include/orbx_cv_calibrate.h) sets the variant it detected in the shim's cv:: functions — which ARE the
// oracle's primitives under the oracle's current switches, i.e. what this stub computes with anyway
int orbx_set_option(orbx_ctx*, const char*, int) { return ORBX_OK; }
int orbx_bow_transform_published(orbx_voc*, const void*, int, int, uint32_t*, double*, uint32_t*) { return 1; }   // nothing is ever precomputed here
// orbx_nn_groups by plain loops over the oracle's Hamming distance: per query the candidates of its group within max_dist, in list order
int orbx_nn_groups(orbx_ctx*, const uint8_t* q_desc, const int32_t* q_group, int nq, const uint8_t* t_desc, int, const int32_t* group_ptr,
                   const int32_t* group_cand, int, int max_dist, int32_t* q_off, int32_t* q_cnt, orbx_candidate* entries, int pool_cap, int* n_entries) {
  int n = 0;
  for (int q = 0; q < nq; q++) {
    q_off[q] = n; q_cnt[q] = 0;
    for (int c = group_ptr[q_group[q]]; c < group_ptr[q_group[q] + 1]; c++) {
      const int d = mo_hamming(q_desc + (size_t)q * 32, t_desc + (size_t)group_cand[c] * 32);
      if (d > max_dist) continue;
      if (n < pool_cap) { entries[n].idx = group_cand[c]; entries[n].dist = d; }
      n++; q_cnt[q]++;
    }
  }
  *n_entries = n;
  return n > pool_cap ? ORBX_E_CAPACITY : ORBX_OK;
}
int orbx_publish_descriptors(orbx_ctx*, const void*, int) { return ORBX_OK; }
// the host mirror of mvImagePyramid (levels >= 1) that include/ORBextractor.h hands to Frame::ComputeStereoMatches: the oracle's levels
int orbx_host_pyramid_level(orbx_ctx* c, int level, const uint8_t** data, size_t* stride, int* w, int* h) {
  if (level < 1 || level >= c->nlevels || (size_t)level >= c->pyr.size() || c->pyr[level].empty()) return ORBX_E_INVALID;
  *data = c->pyr[level].data(); *stride = (size_t)c->pw[level]; *w = c->pw[level]; *h = c->ph[level];
  return ORBX_OK;
}
int orbx_extract(orbx_ctx* c, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int* n_out,
                 int* mono_out) {
  if (!img || rows <= 0 || cols <= 0) return ORBX_E_EMPTY;
  if (orbo_extract(c->ora, img, rows, cols, (int)stride, lap0, lap1, kps, desc, orbx_keypoint_capacity(c), n_out, mono_out) != 0) return ORBX_E_INVALID;
  c->pyr.assign(c->nlevels, std::vector<uint8_t>()); c->pw.assign(c->nlevels, 0); c->ph.assign(c->nlevels, 0);
  for (int l = 0; l < c->nlevels; l++) {   // level 0 is kept for orbx_stereo_matches only (orbx_host_pyramid_level serves levels >= 1)
    if (orbo_level_size(c->ora, l, &c->pw[l], &c->ph[l]) != 0) continue;
    c->pyr[l].resize((size_t)c->pw[l] * c->ph[l]);
    orbo_level_copy(c->ora, l, 0, c->pyr[l].data(), c->pw[l]);
  }
  return ORBX_OK;
}
// Frame::ComputeStereoMatches on the pyramids of the two contexts' last extractions (the patched src/Frame.cc, integration/Frame_stereo.patch)
int orbx_stereo_matches(orbx_ctx* L, orbx_ctx* R, const orbx_keypoint* kpsL, const uint8_t* descL, int nL, const orbx_keypoint* kpsR, const uint8_t* descR,
                        int nR, float mb, float mbf, float* u_right, float* depth, int* nmatches) {
  const int nl = L->nlevels;
  if (R->nlevels != nl || (int)L->pyr.size() != nl || (int)R->pyr.size() != nl) return ORBX_E_INVALID;
  std::vector<const uint8_t*> pl(nl), pr(nl);
  std::vector<int32_t> w(nl), h(nl);
  std::vector<float> scale(nl), inv_scale(nl), s2(nl), is2(nl);
  std::vector<int32_t> quota(nl);
  orbo_tables(L->ora, scale.data(), inv_scale.data(), s2.data(), is2.data(), quota.data(), nullptr);
  for (int l = 0; l < nl; l++) {
    if (L->pw[l] != R->pw[l] || L->ph[l] != R->ph[l] || L->pyr[l].empty() || R->pyr[l].empty()) return ORBX_E_INVALID;
    pl[l] = L->pyr[l].data(); pr[l] = R->pyr[l].data(); w[l] = L->pw[l]; h[l] = L->ph[l];
  }
  const int kept = mo_stereo_matches(kpsL, descL, nL, kpsR, descR, nR, pl.data(), pr.data(), w.data(), h.data(), w.data(), scale.data(), inv_scale.data(), mb,
                                     mbf, u_right, depth);
  if (nmatches) *nmatches = kept;
  return ORBX_OK;
}
int orbx_voc_load_text(orbx_ctx*, const char* path, orbx_voc** out) {
  void* h = mo_voc_load(path);
  if (!h) return ORBX_E_INVALID;
  *out = new orbx_voc();
  (*out)->h = h;
  return ORBX_OK;
}
void orbx_voc_destroy(orbx_voc* v) { if (v) { mo_voc_free(v->h); delete v; } }
int orbx_voc_info(const orbx_voc*, int*, int*, int*, int* nwords) { if (nwords) *nwords = 1; return ORBX_OK; }
int orbx_voc_save_text(const orbx_voc*, const char*) { return ORBX_E_INVALID; }
int orbx_voc_save_binary(const orbx_voc*, const char*) { return ORBX_E_INVALID; }
int orbx_voc_load_binary(orbx_ctx*, const char*, orbx_voc**) { return ORBX_E_INVALID; }
int orbx_bow_transform(orbx_voc* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node) {
  mo_voc_descend(v->h, desc, n, levelsup, word, weight, node);
  v->last_desc.assign(desc, desc + (size_t)n * 32); v->last_levelsup = levelsup;   // orbx_bow_finalize below re-derives the vector from these
  return ORBX_OK;
}
int orbx_bow_finalize(const orbx_voc* v, const uint32_t*, const double*, int n, uint32_t* ids, double* vals, int* n_out) {
  std::vector<uint32_t> fn(n + 1), ff(n + 1);
  int nfv = 0;
  *n_out = mo_voc_transform(v->h, v->last_desc.data(), n, v->last_levelsup, ids, vals, fn.data(), ff.data(), &nfv);
  return ORBX_OK;
}
// ---- keyframe database (include/KeyFrameDatabase.h drop-in) over the oracle's inverted-file restatement
int orbx_kfdb_create(orbx_ctx*, orbx_kfdb** out) { *out = new orbx_kfdb(); (*out)->h = mo_kfdb_new(); return ORBX_OK; }
void orbx_kfdb_destroy(orbx_kfdb* db) { if (db) { mo_kfdb_free(db->h); delete db; } }
int orbx_kfdb_size(const orbx_kfdb* db) { return (int)db->rows.size(); }
int orbx_kfdb_add(orbx_kfdb* db, int64_t id, const uint32_t* ids, const double* vals, int n) {
  if (db->rows.count(id)) return ORBX_E_INVALID;
  db->rows[id] = std::make_pair(std::vector<uint32_t>(ids, ids + n), std::vector<double>(vals, vals + n));
  mo_kfdb_add(db->h, (long)id, ids, vals, n);
  return ORBX_OK;
}
int orbx_kfdb_erase(orbx_kfdb* db, int64_t id) { if (db->rows.erase(id)) mo_kfdb_erase(db->h, (long)id); return ORBX_OK; }
int orbx_kfdb_clear(orbx_kfdb* db) { mo_kfdb_free(db->h); db->h = mo_kfdb_new(); db->rows.clear(); return ORBX_OK; }
int orbx_kfdb_sharing(orbx_kfdb* db, const uint32_t* q_ids, int nq, int64_t* kf_ids, int32_t* common_words, int cap, int* n_sharing) {
  std::vector<double> qv(nq + 1, 1.0), sc(db->rows.size() + 1);
  std::vector<long> kf(db->rows.size() + 1);
  std::vector<int32_t> w(db->rows.size() + 1);
  int mx = 0, mn = 0;
  const int n = mo_kfdb_query(db->h, q_ids, qv.data(), nq, nullptr, 0, 0, kf.data(), w.data(), sc.data(), &mx, &mn);
  *n_sharing = n;
  if (n > cap) return ORBX_E_CAPACITY;
  for (int i = 0; i < n; i++) { kf_ids[i] = kf[i]; common_words[i] = w[i]; }
  return ORBX_OK;
}
int orbx_kfdb_score(orbx_kfdb* db, const uint32_t* q_ids, const double* q_vals, int nq, const int64_t* kf_ids, int n, double* scores) {
  for (int i = 0; i < n; i++) {
    auto it = db->rows.find(kf_ids[i]);
    if (it == db->rows.end()) return ORBX_E_INVALID;
    scores[i] = mo_score_l1(q_ids, q_vals, nq, it->second.first.data(), it->second.second.data(), (int)it->second.first.size());
  }
  return ORBX_OK;
}
double orbx_bow_score_l1(const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb) { return mo_score_l1(ida, va, na, idb, vb, nb); }
const char* orbx_last_error(const orbx_ctx*) { return "oracle stub"; }
int orbx_hamming(const uint8_t a[32], const uint8_t b[32]) { return mo_hamming(a, b); }

int orbx_nn_csr(orbx_ctx*, const uint8_t* q, int nq, const uint8_t* t, int, const int32_t* row_ptr, const int32_t* cand, int last_wins,
                int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist, int32_t* dist_out) {
  int32_t* tmp = new int32_t[4 * (size_t)(nq > 0 ? nq : 1)];
  mo_nn_csr(q, nq, t, row_ptr, cand, last_wins, best_idx ? best_idx : tmp, best_dist ? best_dist : tmp + nq, second_idx ? second_idx : tmp + 2 * nq,
            second_dist ? second_dist : tmp + 3 * nq, dist_out);
  delete[] tmp;
  return ORBX_OK;
}

int orbx_knn2_allpairs(orbx_ctx*, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
  mo_knn2(q, nq, t, nt, idx, dist);
  return ORBX_OK;
}

int orbx_window_search_grid(orbx_ctx*, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid, const uint8_t* kp_skip,
                            const float* kp_uright, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                            const int32_t* qmax, const uint8_t* q_desc, const float* q_xr, int nq, int32_t* row_ptr, int32_t* cand,
                            int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  const int r = mo_window_search_grid(kps, desc, n, grid, kp_skip, kp_uright, qx, qy, qr, qmin, qmax, q_desc, q_xr, nq, row_ptr, cand, dist,
                                      cand_cap, best_idx, best_dist, second_idx, second_dist);
  return r < 0 ? ORBX_E_CAPACITY : r;
}

int orbx_window_nearest(orbx_ctx*, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid, const float* kp_uright,
                        const float* inv_level_sigma2, int, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                        const int32_t* qmax, const float* q_ur, const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist) {
  mo_window_nearest(kps, desc, n, grid, kp_uright, inv_level_sigma2, qx, qy, qr, qmin, qmax, q_ur, q_desc, nq, best_idx, best_dist);
  return ORBX_OK;
}

// resident targets: host copies here (the stub has no device); searches are the oracle's window passes on those copies
}  // extern "C"

struct orbx_target {
  std::vector<unsigned char> kps, desc;
  std::vector<int32_t> cs, ci;
  std::vector<float> ur, sig;
  int n = 0;
  bool has_grid = false;
  orbx_grid g;
  void fill(const orbx_keypoint* k, const uint8_t* d, int n_, const orbx_grid* grid, const float* u, const float* s, int nlevels) {
    n = n_;
    kps.assign((const unsigned char*)k, (const unsigned char*)k + sizeof(orbx_keypoint) * (size_t)n);
    desc.assign(d, d + (size_t)n * 32);
    has_grid = grid->cell_start != nullptr;
    g = *grid;
    if (has_grid) {
      cs.assign(grid->cell_start, grid->cell_start + 64 * 48 + 1);
      ci.assign(grid->cell_idx, grid->cell_idx + cs.back());
      g.cell_start = cs.data(); g.cell_idx = ci.data();
    }
    ur.clear(); sig.clear();
    if (u) ur.assign(u, u + n);
    if (s) sig.assign(s, s + nlevels);
  }
};

extern "C" {

int orbx_target_create(orbx_ctx*, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid, const float* kp_uright,
                       const float* inv_level_sigma2, int nlevels, orbx_target** target) {
  *target = new orbx_target();
  (*target)->fill(kps, desc, n, grid, kp_uright, inv_level_sigma2, nlevels);
  return ORBX_OK;
}
int orbx_target_assign(orbx_ctx*, orbx_target* target, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                       const float* kp_uright, const float* inv_level_sigma2, int nlevels) {
  target->fill(kps, desc, n, grid, kp_uright, inv_level_sigma2, nlevels);
  return ORBX_OK;
}
void orbx_target_destroy(orbx_target* target) { delete target; }
int orbx_target_size(const orbx_target* target) { return target->n; }

int orbx_target_search(orbx_ctx*, const orbx_target* T, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                       const int32_t* qmax, const uint8_t* q_desc, const float* q_xr, int nq, int32_t* row_ptr, int32_t* cand, int32_t* dist,
                       int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  const int r = mo_window_search_grid(T->kps.data(), T->desc.data(), T->n, &T->g, kp_skip, T->ur.empty() ? nullptr : T->ur.data(), qx, qy, qr, qmin, qmax,
                                      q_desc, q_xr, nq, row_ptr, cand, dist, cand_cap, best_idx, best_dist, second_idx, second_dist);
  return r < 0 ? ORBX_E_CAPACITY : r;
}

int orbx_target_search_view(orbx_ctx* ctx, const orbx_target* T, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                            const int32_t* qmax, const uint8_t* q_desc, const float* q_xr, int nq, const orbx_list_span** spans, const orbx_candidate** pool) {
  static thread_local std::vector<int32_t> rp, cand, dist;
  // like the library: two result sets used alternately — a view stays valid while the next view call runs (a rig's left and right lists)
  static thread_local std::vector<orbx_list_span> sp2[2];
  static thread_local std::vector<orbx_candidate> pl2[2];
  static thread_local int par = 0;
  par ^= 1;
  std::vector<orbx_list_span>& sp = sp2[par];
  std::vector<orbx_candidate>& pl = pl2[par];
  rp.assign(nq + 1, 0);
  cand.resize(1 << 16); dist.resize(1 << 16);
  int rc = orbx_target_search(ctx, T, kp_skip, qx, qy, qr, qmin, qmax, q_desc, q_xr, nq, rp.data(), cand.data(), dist.data(), (int)cand.size(), nullptr, nullptr,
                              nullptr, nullptr);
  if (rc == ORBX_E_CAPACITY) {
    cand.resize(rp[nq] + 64); dist.resize(rp[nq] + 64);
    rc = orbx_target_search(ctx, T, kp_skip, qx, qy, qr, qmin, qmax, q_desc, q_xr, nq, rp.data(), cand.data(), dist.data(), (int)cand.size(), nullptr, nullptr,
                            nullptr, nullptr);
  }
  if (rc < 0) return rc;
  sp.resize(nq); pl.resize(rp[nq] + 1);
  // the device pool is not in query order (segments are reserved as the waves finish): mimic that with a reversed segment layout, so that
  // a reader that assumed CSR order would fail
  int at = 0;
  for (int q = nq - 1; q >= 0; q--) {
    sp[q].start = at; sp[q].count = rp[q + 1] - rp[q];
    sp[q].best_idx = sp[q].second_idx = -1; sp[q].best_dist = sp[q].second_dist = 256; sp[q].reserved0 = sp[q].reserved1 = 0;
    for (int c = rp[q]; c < rp[q + 1]; c++) {
      pl[at].idx = cand[c]; pl[at].dist = dist[c]; at++;
      // the two smallest (distance, list position)
      if (dist[c] < sp[q].best_dist) { sp[q].second_idx = sp[q].best_idx; sp[q].second_dist = sp[q].best_dist; sp[q].best_idx = cand[c]; sp[q].best_dist = dist[c]; }
      else if (dist[c] < sp[q].second_dist) { sp[q].second_idx = cand[c]; sp[q].second_dist = dist[c]; }
    }
  }
  *spans = sp.data(); *pool = pl.data();
  return rp[nq];
}
// issue + wait: the stub has nothing to overlap — _begin remembers the arguments, _end runs the call (same results, same lifetimes)
namespace {
struct StubViewCall { bool pending = false; const orbx_target* T; const uint8_t* skip; const float *qx, *qy, *qr, *qxr; const int32_t *qmin, *qmax; const uint8_t* qd; int nq; };
thread_local StubViewCall g_view_calls[2];
thread_local int g_view_next = 0;
}  // namespace
int orbx_target_search_view_begin(orbx_ctx*, const orbx_target* T, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                                  const int32_t* qmax, const uint8_t* q_desc, const float* q_xr, int nq) {
  const int slot = g_view_next; g_view_next ^= 1;
  if (g_view_calls[slot].pending) return ORBX_E_INVALID;
  g_view_calls[slot] = StubViewCall{true, T, kp_skip, qx, qy, qr, q_xr, qmin, qmax, q_desc, nq};
  return slot;
}
int orbx_target_search_view_end(orbx_ctx* ctx, int slot, const orbx_list_span** spans, const orbx_candidate** pool) {
  if (slot < 0 || slot > 1 || !g_view_calls[slot].pending) return ORBX_E_INVALID;
  const StubViewCall c = g_view_calls[slot];
  g_view_calls[slot].pending = false;
  return orbx_target_search_view(ctx, c.T, c.skip, c.qx, c.qy, c.qr, c.qmin, c.qmax, c.qd, c.qxr, c.nq, spans, pool);
}
int orbx_target_search_view_cancel(orbx_ctx*, int slot) {
  if (slot < 0 || slot > 1) return ORBX_E_INVALID;
  g_view_calls[slot].pending = false;
  return ORBX_OK;
}
int orbx_target_nearest(orbx_ctx*, const orbx_target* T, int reprojection_gate, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                        const int32_t* qmax, const float* q_ur, const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist) {
  mo_window_nearest(T->kps.data(), T->desc.data(), T->n, &T->g, reprojection_gate ? T->ur.data() : nullptr, reprojection_gate ? T->sig.data() : nullptr,
                    qx, qy, qr, qmin, qmax, reprojection_gate ? q_ur : nullptr, q_desc, nq, best_idx, best_dist);
  return ORBX_OK;
}

}  // extern "C"
