// TEST INFRASTRUCTURE: the C-ABI entry points the drop-in ORBmatcher (csrc/ref_adapter/ORBmatcher.cc) calls, served by the
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

struct orbx_ctx { int dummy; };

extern "C" {

int orbx_create(orbx_ctx** out, int, float, int, int, int, int) { *out = new orbx_ctx(); return ORBX_OK; }
void orbx_destroy(orbx_ctx* c) { delete c; }
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

int orbx_target_nearest(orbx_ctx*, const orbx_target* T, int reprojection_gate, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                        const int32_t* qmax, const float* q_ur, const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist) {
  mo_window_nearest(T->kps.data(), T->desc.data(), T->n, &T->g, reprojection_gate ? T->ur.data() : nullptr, reprojection_gate ? T->sig.data() : nullptr,
                    qx, qy, qr, qmin, qmax, reprojection_gate ? q_ur : nullptr, q_desc, nq, best_idx, best_dist);
  return ORBX_OK;
}

}  // extern "C"
