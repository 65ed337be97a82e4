// orbx — KeyFrameDatabase with the keyframes' BowVectors resident in HBM.
//
// The reference keeps an inverted file word -> list<KeyFrame*> (src/KeyFrameDatabase.cc:39-45) because a CPU can only
// afford to touch the keyframes that share a word with the query.  On the GPU the whole database is scanned instead:
// one wave per keyframe intersects the query's sorted word ids with the keyframe's (a few thousand keyframes x ~1000
// words = a few MB, one pass).  The results are the ones the reference computes in the first two phases of its five
// Detect* routines (:100-165, :228-310, :468-535, :604-665, :733-790): the keyframes sharing a word with the query in the
// reference's list order, their common-word counts, maxCommonWords / minCommonWords, and L1Scoring::score
// (Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) for those above the threshold — bit-identical doubles, because the
// matched terms are accumulated in ascending word id order like the reference's merge loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

struct KfRow { int32_t start, len; };

constexpr int kKfdbMaxQuery = 8192;   // query words staged in LDS (4 B ids + 8 B values each); longer queries are read from HBM / L2

// position of `id` in the ascending array a[0, n), or -1
__device__ __forceinline__ int find_sorted(const uint32_t* a, int n, uint32_t id) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < id) lo = mid + 1; else hi = mid;
  }
  return (lo < n && a[lo] == id) ? lo : -1;
}

// Phase 1: per keyframe row the number of words shared with the query and the smallest shared word id (which decides
// the row's place in the reference's lKFsSharingWords list); the maximum count over the active rows.
template <bool INLDS>
__global__ __launch_bounds__(256) void k_kfdb_common(const uint32_t* __restrict__ qid, int nq, const KfRow* __restrict__ rows,
                                                     const uint8_t* __restrict__ active, int nrows, const uint32_t* __restrict__ ids,
                                                     int32_t* __restrict__ common, uint32_t* __restrict__ first_word,
                                                     int32_t* __restrict__ max_common) {
  extern __shared__ __align__(16) uint32_t s_qbuf[];
  __shared__ int wg_max;   // one global atomicMax per workgroup: thousands of them on one address serialise in the L2
  const uint32_t* s_q = qid;
  if (threadIdx.x == 0) wg_max = 0;
  if (INLDS) {
    for (int i = threadIdx.x; i < nq; i += blockDim.x) s_qbuf[i] = qid[i];
    s_q = s_qbuf;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int cnt = 0;
  uint32_t first = 0xffffffffu;
  if (r < nrows && active[r]) {
    const KfRow row = rows[r];
    for (int b0 = 0; b0 < row.len; b0 += 64) {
      const int j = b0 + lane;
      bool hit = false;
      uint32_t id = 0;
      if (j < row.len) {
        id = ids[row.start + j];
        hit = find_sorted(s_q, nq, id) >= 0;
      }
      const unsigned long long bal = __ballot(hit);
      if (bal) {
        if (cnt == 0) first = __shfl(id, __ffsll((long long)bal) - 1);   // ids ascend along the row
        cnt += __popcll(bal);
      }
    }
  }
  if (lane == 0 && r < nrows) {
    common[r] = cnt;
    first_word[r] = first;
    if (cnt) atomicMax(&wg_max, cnt);
  }
  __syncthreads();
  if (threadIdx.x == 0 && wg_max) atomicMax(max_common, wg_max);
}

// Phase 2: L1 score of the rows with more than minCommonWords = (int)(maxCommonWords * 0.8f) common words (at least
// `min_words_floor`, DetectBestCandidates' nMinWords, :514-517).  The matched terms are added in ascending word order by a
// wave-uniform serial loop over the ballot bits, which reproduces the double rounding of the reference's merge loop.
template <bool INLDS>
__global__ __launch_bounds__(256) void k_kfdb_score(const uint32_t* __restrict__ qid, const double* __restrict__ qv, int nq,
                                                    const KfRow* __restrict__ rows, int nrows, const uint32_t* __restrict__ ids,
                                                    const double* __restrict__ vals, const int32_t* __restrict__ common,
                                                    const int32_t* __restrict__ max_common, int min_words_floor,
                                                    double* __restrict__ scores) {
  extern __shared__ __align__(16) uint32_t s_qbuf[];
  const uint32_t* s_q = qid;
  const double* s_v = qv;
  if (INLDS) {
    double* vbuf = (double*)(s_qbuf + ((nq + 1) & ~1));
    for (int i = threadIdx.x; i < nq; i += blockDim.x) { s_qbuf[i] = qid[i]; vbuf[i] = qv[i]; }
    __syncthreads();
    s_q = s_qbuf; s_v = vbuf;
  }
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  int min_common = (int)__fmul_rn((float)*max_common, 0.8f);
  if (min_common < min_words_floor) min_common = min_words_floor;
  if (common[r] <= min_common) {   // wave-uniform
    if (lane == 0) scores[r] = -1.0;
    return;
  }
  const KfRow row = rows[r];
  double score = 0.0;
  for (int b0 = 0; b0 < row.len; b0 += 64) {
    const int j = b0 + lane;
    double t = 0.0;
    bool hit = false;
    if (j < row.len) {
      const int p = find_sorted(s_q, nq, ids[row.start + j]);
      if (p >= 0) {
        hit = true;
        const double vi = s_v[p], wi = vals[row.start + j];
        t = __dsub_rn(__dsub_rn(fabs(__dsub_rn(vi, wi)), fabs(vi)), fabs(wi));
      }
    }
    unsigned long long bal = __ballot(hit);
    while (bal) {
      const int b = __ffsll((long long)bal) - 1;
      bal &= bal - 1;
      score = __dadd_rn(score, __shfl(t, b));
    }
  }
  if (lane == 0) scores[r] = -score / 2.0;
}

}  // namespace orbx

using namespace orbx;

struct orbx_kfdb {
  orbx_ctx* ctx = nullptr;
  struct HostRow { int64_t kf_id; int32_t start, len; uint64_t seq; bool alive; };
  std::vector<HostRow> rows;
  std::unordered_map<int64_t, int> row_of;   // alive keyframes only
  std::vector<uint32_t> h_ids;               // host mirror of the CSR payload (for compaction)
  std::vector<double> h_vals;
  uint64_t next_seq = 0;
  size_t dead_nnz = 0;
  // device
  uint32_t* d_ids = nullptr; double* d_vals = nullptr; size_t d_cap = 0, d_nnz = 0;   // uploaded prefix of h_ids / h_vals
  KfRow* d_rows = nullptr; uint8_t* d_active = nullptr; int32_t* d_common = nullptr; uint32_t* d_first = nullptr;
  double* d_scores = nullptr; int32_t* d_max = nullptr; size_t d_rows_cap = 0, d_rows_n = 0;
  uint32_t* d_qid = nullptr; double* d_qv = nullptr; size_t d_q_cap = 0;
};

namespace {

int kfdb_free_device(orbx_kfdb* db) {
  for (void* p : {(void*)db->d_ids, (void*)db->d_vals, (void*)db->d_rows, (void*)db->d_active, (void*)db->d_common, (void*)db->d_first,
                  (void*)db->d_scores, (void*)db->d_max, (void*)db->d_qid, (void*)db->d_qv})
    if (p) (void)hipFree(p);
  db->d_ids = nullptr; db->d_vals = nullptr; db->d_rows = nullptr; db->d_active = nullptr; db->d_common = nullptr; db->d_first = nullptr;
  db->d_scores = nullptr; db->d_max = nullptr; db->d_qid = nullptr; db->d_qv = nullptr;
  db->d_cap = db->d_nnz = db->d_rows_cap = db->d_rows_n = db->d_q_cap = 0;
  return ORBX_OK;
}

// drop the erased rows from the host mirror; the device copy is rebuilt by the next sync
void kfdb_compact(orbx_kfdb* db) {
  std::vector<orbx_kfdb::HostRow> rows;
  std::vector<uint32_t> ids;
  std::vector<double> vals;
  ids.reserve(db->h_ids.size() - db->dead_nnz); vals.reserve(db->h_ids.size() - db->dead_nnz);
  db->row_of.clear();
  for (const auto& r : db->rows) {
    if (!r.alive) continue;
    orbx_kfdb::HostRow n = r;
    n.start = (int32_t)ids.size();
    ids.insert(ids.end(), db->h_ids.begin() + r.start, db->h_ids.begin() + r.start + r.len);
    vals.insert(vals.end(), db->h_vals.begin() + r.start, db->h_vals.begin() + r.start + r.len);
    db->row_of[n.kf_id] = (int)rows.size();
    rows.push_back(n);
  }
  db->rows.swap(rows); db->h_ids.swap(ids); db->h_vals.swap(vals);
  db->dead_nnz = 0;
  db->d_nnz = 0; db->d_rows_n = 0;   // everything is re-uploaded
}

// bring the device copy up to date: append the new payload, (re)upload the row table
int kfdb_sync(orbx_kfdb* db) {
  orbx_ctx* ctx = db->ctx;
  hipStream_t st = ctx->stream;
  const size_t nnz = db->h_ids.size(), nrows = db->rows.size();
  if (nnz > db->d_cap) {
    const size_t cap = std::max<size_t>(std::max<size_t>(nnz, 2 * db->d_cap), 1 << 16);
    uint32_t* ni = nullptr; double* nv = nullptr;
    ORBX_HIP(ctx, hipMalloc((void**)&ni, cap * sizeof(uint32_t)));
    ORBX_HIP(ctx, hipMalloc((void**)&nv, cap * sizeof(double)));
    if (db->d_nnz) {
      ORBX_HIP(ctx, hipMemcpyAsync(ni, db->d_ids, db->d_nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      ORBX_HIP(ctx, hipMemcpyAsync(nv, db->d_vals, db->d_nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
      ORBX_HIP(ctx, hipStreamSynchronize(st));
    }
    if (db->d_ids) (void)hipFree(db->d_ids);
    if (db->d_vals) (void)hipFree(db->d_vals);
    db->d_ids = ni; db->d_vals = nv; db->d_cap = cap;
  }
  if (nnz > db->d_nnz) {
    ORBX_HIP(ctx, hipMemcpyAsync(db->d_ids + db->d_nnz, db->h_ids.data() + db->d_nnz, (nnz - db->d_nnz) * sizeof(uint32_t),
                                 hipMemcpyHostToDevice, st));
    ORBX_HIP(ctx, hipMemcpyAsync(db->d_vals + db->d_nnz, db->h_vals.data() + db->d_nnz, (nnz - db->d_nnz) * sizeof(double),
                                 hipMemcpyHostToDevice, st));
    db->d_nnz = nnz;
  }
  if (nrows > db->d_rows_cap) {
    const size_t cap = std::max<size_t>(std::max<size_t>(nrows, 2 * db->d_rows_cap), 1024);
    ORBX_HIP(ctx, hipStreamSynchronize(st));
    for (void* p : {(void*)db->d_rows, (void*)db->d_active, (void*)db->d_common, (void*)db->d_first, (void*)db->d_scores})
      if (p) (void)hipFree(p);
    db->d_rows = nullptr; db->d_active = nullptr; db->d_common = nullptr; db->d_first = nullptr; db->d_scores = nullptr;
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_rows, cap * sizeof(KfRow)));
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_active, cap));
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_common, cap * sizeof(int32_t)));
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_first, cap * sizeof(uint32_t)));
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_scores, cap * sizeof(double)));
    db->d_rows_cap = cap; db->d_rows_n = 0;
  }
  if (!db->d_max) ORBX_HIP(ctx, hipMalloc((void**)&db->d_max, sizeof(int32_t)));
  if (nrows > db->d_rows_n) {
    std::vector<KfRow> tmp(nrows - db->d_rows_n);
    for (size_t i = db->d_rows_n; i < nrows; i++) tmp[i - db->d_rows_n] = KfRow{db->rows[i].start, db->rows[i].len};
    ORBX_HIP(ctx, hipMemcpyAsync(db->d_rows + db->d_rows_n, tmp.data(), tmp.size() * sizeof(KfRow), hipMemcpyHostToDevice, st));
    ORBX_HIP(ctx, hipStreamSynchronize(st));   // tmp is pageable and dies here
    db->d_rows_n = nrows;
  }
  return ORBX_OK;
}

}  // namespace

extern "C" {

int orbx_kfdb_create(orbx_ctx* ctx, orbx_kfdb** out) {
  if (!ctx || !out) return ORBX_E_INVALID;
  orbx_kfdb* db = new orbx_kfdb();
  db->ctx = ctx;
  *out = db;
  return ORBX_OK;
}

void orbx_kfdb_destroy(orbx_kfdb* db) {
  if (!db) return;
  (void)hipSetDevice(db->ctx->device);
  (void)hipStreamSynchronize(db->ctx->stream);
  kfdb_free_device(db);
  delete db;
}

int orbx_kfdb_size(const orbx_kfdb* db) { return db ? (int)db->row_of.size() : ORBX_E_INVALID; }

int orbx_kfdb_add(orbx_kfdb* db, int64_t kf_id, const uint32_t* ids, const double* vals, int n) {
  if (!db || n < 0 || (n > 0 && (!ids || !vals))) return ORBX_E_INVALID;
  if (db->row_of.count(kf_id)) return set_err(db->ctx, ORBX_E_INVALID, "orbx_kfdb_add: keyframe id already in the database");
  for (int i = 1; i < n; i++)
    if (ids[i] <= ids[i - 1]) return set_err(db->ctx, ORBX_E_INVALID, "orbx_kfdb_add: word ids must ascend strictly (a BowVector is an ordered map)");
  if (db->h_ids.size() + (size_t)n >= (1ull << 31)) return set_err(db->ctx, ORBX_E_CAPACITY, "orbx_kfdb_add: more than 2^31 database entries");
  orbx_kfdb::HostRow r{kf_id, (int32_t)db->h_ids.size(), n, db->next_seq++, true};
  db->h_ids.insert(db->h_ids.end(), ids, ids + n);
  db->h_vals.insert(db->h_vals.end(), vals, vals + n);
  db->row_of[kf_id] = (int)db->rows.size();
  db->rows.push_back(r);
  return ORBX_OK;
}

int orbx_kfdb_erase(orbx_kfdb* db, int64_t kf_id) {
  if (!db) return ORBX_E_INVALID;
  auto it = db->row_of.find(kf_id);
  if (it == db->row_of.end()) return ORBX_OK;   // the reference's erase of an absent keyframe is a no-op too (:47-66)
  db->rows[it->second].alive = false;
  db->dead_nnz += (size_t)db->rows[it->second].len;
  db->row_of.erase(it);
  if (db->dead_nnz > (1u << 16) && 2 * db->dead_nnz > db->h_ids.size()) kfdb_compact(db);
  return ORBX_OK;
}

int orbx_kfdb_clear(orbx_kfdb* db) {
  if (!db) return ORBX_E_INVALID;
  db->rows.clear(); db->row_of.clear(); db->h_ids.clear(); db->h_vals.clear();
  db->dead_nnz = 0; db->d_nnz = 0; db->d_rows_n = 0;
  return ORBX_OK;
}

}  // extern "C"

namespace {

// shared front of the query entry points: arguments, device copy up to date, query words (and values) uploaded
int kfdb_begin(orbx_kfdb* db, const char* who, const uint32_t* q_ids, const double* q_vals, int nq) {
  orbx_ctx* ctx = db->ctx;
  for (int i = 1; i < nq; i++)
    if (q_ids[i] <= q_ids[i - 1]) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": word ids must ascend strictly");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int rc = kfdb_sync(db);
  if (rc != ORBX_OK) return rc;
  hipStream_t st = ctx->stream;
  if ((size_t)nq > db->d_q_cap) {
    ORBX_HIP(ctx, hipStreamSynchronize(st));
    if (db->d_qid) (void)hipFree(db->d_qid);
    if (db->d_qv) (void)hipFree(db->d_qv);
    db->d_qid = nullptr; db->d_qv = nullptr;
    const size_t qc = std::max<size_t>(nq, 2048);
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_qid, qc * sizeof(uint32_t)));
    ORBX_HIP(ctx, hipMalloc((void**)&db->d_qv, qc * sizeof(double)));
    db->d_q_cap = qc;
  }
  ORBX_HIP(ctx, hipMemcpyAsync(db->d_qid, q_ids, (size_t)nq * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  if (q_vals) ORBX_HIP(ctx, hipMemcpyAsync(db->d_qv, q_vals, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, st));
  return ORBX_OK;
}

// a query of up to 8192 words (every configuration of the reference's yaml files) is staged in LDS; a longer one (the
// 20 000-feature frames of Examples/Monocular/mi.yaml) is searched where it lies, in HBM / L2
int kfdb_launch_common(orbx_kfdb* db, int nq, int nrows) {
  hipStream_t st = db->ctx->stream;
  const dim3 grid((nrows + 3) / 4), block(256);
  if (nq <= kKfdbMaxQuery)
    hipLaunchKernelGGL(k_kfdb_common<true>, grid, block, (size_t)nq * sizeof(uint32_t), st, db->d_qid, nq, db->d_rows, db->d_active, nrows, db->d_ids,
                       db->d_common, db->d_first, db->d_max);
  else
    hipLaunchKernelGGL(k_kfdb_common<false>, grid, block, 0, st, db->d_qid, nq, db->d_rows, db->d_active, nrows, db->d_ids, db->d_common, db->d_first,
                       db->d_max);
  ORBX_HIP(db->ctx, hipGetLastError());
  return ORBX_OK;
}

int kfdb_launch_score(orbx_kfdb* db, const char* who, int nq, int nrows, int min_words_floor) {
  orbx_ctx* ctx = db->ctx;
  hipStream_t st = ctx->stream;
  const dim3 grid((nrows + 3) / 4), block(256);
  const bool inlds = nq <= kKfdbMaxQuery;
  const size_t lds2 = inlds ? (size_t)((nq + 1) & ~1) * sizeof(uint32_t) + (size_t)nq * sizeof(double) : 0;
  if (lds2 > 64 * 1024 && ensure_dynamic_lds((const void*)k_kfdb_score<true>, (int)lds2) != hipSuccess)
    return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": query does not fit the LDS");
  if (inlds)
    hipLaunchKernelGGL(k_kfdb_score<true>, grid, block, lds2, st, db->d_qid, db->d_qv, nq, db->d_rows, nrows, db->d_ids, db->d_vals, db->d_common,
                       db->d_max, min_words_floor, db->d_scores);
  else
    hipLaunchKernelGGL(k_kfdb_score<false>, grid, block, 0, st, db->d_qid, db->d_qv, nq, db->d_rows, nrows, db->d_ids, db->d_vals, db->d_common,
                       db->d_max, min_words_floor, db->d_scores);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

// the reference's list order: query words ascending, inside one word's inverted list the order of add() (:39-45)
void kfdb_list_order(const orbx_kfdb* db, const std::vector<int32_t>& common, const std::vector<uint32_t>& first, std::vector<int>& order) {
  order.clear();
  for (int r = 0; r < (int)common.size(); r++)
    if (common[r] > 0) order.push_back(r);
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    if (first[a] != first[b]) return first[a] < first[b];
    return db->rows[a].seq < db->rows[b].seq;
  });
}

}  // namespace

extern "C" {

int orbx_kfdb_query(orbx_kfdb* db, const uint32_t* q_ids, const double* q_vals, int nq, const int64_t* exclude, int n_exclude,
                    int min_words_floor, int64_t* kf_ids, int32_t* common_words, double* scores, int cap, int* n_sharing,
                    int* max_common_words, int* min_common_words) {
  if (!db || nq < 0 || n_exclude < 0 || cap < 0 || !n_sharing || (nq > 0 && (!q_ids || !q_vals)) || (n_exclude > 0 && !exclude) ||
      (cap > 0 && (!kf_ids || !common_words || !scores)))
    return db ? set_err(db->ctx, ORBX_E_INVALID, "orbx_kfdb_query: bad arguments") : ORBX_E_INVALID;
  orbx_ctx* ctx = db->ctx;
  *n_sharing = 0;
  if (max_common_words) *max_common_words = 0;
  if (min_common_words) *min_common_words = 0;
  const int nrows = (int)db->rows.size();
  if (nrows == 0 || nq == 0 || db->row_of.empty()) {
    for (int i = 1; i < nq; i++)
      if (q_ids[i] <= q_ids[i - 1]) return set_err(ctx, ORBX_E_INVALID, "orbx_kfdb_query: word ids must ascend strictly");
    return ORBX_OK;
  }
  int rc = kfdb_begin(db, "orbx_kfdb_query", q_ids, q_vals, nq);
  if (rc != ORBX_OK) return rc;
  hipStream_t st = ctx->stream;
  // rows taking part: alive and not excluded by the caller (connected keyframes, keyframes of another map, ...)
  std::vector<uint8_t> active(nrows);
  for (int r = 0; r < nrows; r++) active[r] = db->rows[r].alive;
  for (int i = 0; i < n_exclude; i++) {
    auto it = db->row_of.find(exclude[i]);
    if (it != db->row_of.end()) active[it->second] = 0;
  }
  ORBX_HIP(ctx, hipMemcpyAsync(db->d_active, active.data(), (size_t)nrows, hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemsetAsync(db->d_max, 0, sizeof(int32_t), st));
  if ((rc = kfdb_launch_common(db, nq, nrows)) != ORBX_OK) return rc;
  if ((rc = kfdb_launch_score(db, "orbx_kfdb_query", nq, nrows, min_words_floor)) != ORBX_OK) return rc;
  std::vector<int32_t> common(nrows);
  std::vector<uint32_t> first(nrows);
  std::vector<double> sc(nrows);
  int32_t maxc = 0;
  ORBX_HIP(ctx, hipMemcpyAsync(common.data(), db->d_common, (size_t)nrows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(first.data(), db->d_first, (size_t)nrows * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(sc.data(), db->d_scores, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(&maxc, db->d_max, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  std::vector<int> order;
  kfdb_list_order(db, common, first, order);
  int minc = (int)((float)maxc * 0.8f);
  if (minc < min_words_floor) minc = min_words_floor;
  if ((int)order.size() > cap) return set_err(ctx, ORBX_E_CAPACITY, "orbx_kfdb_query: output capacity below the number of keyframes sharing a word");
  for (size_t i = 0; i < order.size(); i++) {
    const int r = order[i];
    kf_ids[i] = db->rows[r].kf_id;
    common_words[i] = common[r];
    scores[i] = sc[r];
  }
  *n_sharing = (int)order.size();
  if (max_common_words) *max_common_words = maxc;
  if (min_common_words) *min_common_words = minc;
  return ORBX_OK;
}

// ---- the two phases on their own: what a drop-in KeyFrameDatabase needs (csrc/ref_adapter/KeyFrameDatabase.cc).  The five Detect*
// routines differ in WHICH sharing keyframes enter their list (same map / other map / not connected) and in side effects on the ones
// that do not; with the full list in hand that is host logic over the caller's own objects, and the thresholds follow from the
// filtered list, so the scores are asked for afterwards, for exactly the keyframes the routine selected.
int orbx_kfdb_sharing(orbx_kfdb* db, const uint32_t* q_ids, int nq, int64_t* kf_ids, int32_t* common_words, int cap, int* n_sharing) {
  if (!db || nq < 0 || cap < 0 || !n_sharing || (nq > 0 && !q_ids) || (cap > 0 && (!kf_ids || !common_words)))
    return db ? set_err(db->ctx, ORBX_E_INVALID, "orbx_kfdb_sharing: bad arguments") : ORBX_E_INVALID;
  orbx_ctx* ctx = db->ctx;
  *n_sharing = 0;
  const int nrows = (int)db->rows.size();
  if (nrows == 0 || nq == 0 || db->row_of.empty()) return ORBX_OK;
  int rc = kfdb_begin(db, "orbx_kfdb_sharing", q_ids, nullptr, nq);
  if (rc != ORBX_OK) return rc;
  hipStream_t st = ctx->stream;
  std::vector<uint8_t> active(nrows);
  for (int r = 0; r < nrows; r++) active[r] = db->rows[r].alive;
  ORBX_HIP(ctx, hipMemcpyAsync(db->d_active, active.data(), (size_t)nrows, hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemsetAsync(db->d_max, 0, sizeof(int32_t), st));
  if ((rc = kfdb_launch_common(db, nq, nrows)) != ORBX_OK) return rc;
  std::vector<int32_t> common(nrows);
  std::vector<uint32_t> first(nrows);
  ORBX_HIP(ctx, hipMemcpyAsync(common.data(), db->d_common, (size_t)nrows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(first.data(), db->d_first, (size_t)nrows * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  std::vector<int> order;
  kfdb_list_order(db, common, first, order);
  *n_sharing = (int)order.size();
  if ((int)order.size() > cap) return set_err(ctx, ORBX_E_CAPACITY, "orbx_kfdb_sharing: output capacity below the number of keyframes sharing a word");
  for (size_t i = 0; i < order.size(); i++) { kf_ids[i] = db->rows[order[i]].kf_id; common_words[i] = common[order[i]]; }
  return ORBX_OK;
}

int orbx_kfdb_score(orbx_kfdb* db, const uint32_t* q_ids, const double* q_vals, int nq, const int64_t* kf_ids, int n, double* scores) {
  if (!db || nq < 0 || n < 0 || (nq > 0 && (!q_ids || !q_vals)) || (n > 0 && (!kf_ids || !scores)))
    return db ? set_err(db->ctx, ORBX_E_INVALID, "orbx_kfdb_score: bad arguments") : ORBX_E_INVALID;
  orbx_ctx* ctx = db->ctx;
  if (n == 0) return ORBX_OK;
  const int nrows = (int)db->rows.size();
  std::vector<int32_t> sel(nrows, 0);
  std::vector<int> row(n);
  for (int i = 0; i < n; i++) {
    auto it = db->row_of.find(kf_ids[i]);
    if (it == db->row_of.end()) return set_err(ctx, ORBX_E_INVALID, "orbx_kfdb_score: keyframe id not in the database");
    row[i] = it->second;
    sel[it->second] = 1;
  }
  if (nq == 0) {   // L1Scoring::score of an empty vector against anything: no common term, -0.0 / 2 -> the reference returns -0 = 0
    for (int i = 0; i < n; i++) scores[i] = 0.0;
    return ORBX_OK;
  }
  int rc = kfdb_begin(db, "orbx_kfdb_score", q_ids, q_vals, nq);
  if (rc != ORBX_OK) return rc;
  hipStream_t st = ctx->stream;
  // k_kfdb_score scores the rows with common > (int)(max * 0.8f): selected rows carry 1, the others 0, max = 0
  ORBX_HIP(ctx, hipMemcpyAsync(db->d_common, sel.data(), (size_t)nrows * sizeof(int32_t), hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemsetAsync(db->d_max, 0, sizeof(int32_t), st));
  if ((rc = kfdb_launch_score(db, "orbx_kfdb_score", nq, nrows, 0)) != ORBX_OK) return rc;
  std::vector<double> sc(nrows);
  ORBX_HIP(ctx, hipMemcpyAsync(sc.data(), db->d_scores, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  for (int i = 0; i < n; i++) scores[i] = sc[row[i]];
  return ORBX_OK;
}

}  // extern "C"
