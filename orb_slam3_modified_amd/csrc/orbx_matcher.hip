// orbx matcher + bag-of-words: Hamming nearest-neighbour kernels (wave per query, 64-bit popcount) and the
// DBoW2 vocabulary descent / L1 scoring, behind the C ABI of include/orbx.h.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

// ---------------------------------------------------------------------------------------------------
// Hamming kernels.  A descriptor is 32 bytes = 4 x u64; distance = sum popcount(a ^ b)
// (ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2058-2074 — the SWAR bit hack there IS popcount).
// ---------------------------------------------------------------------------------------------------
struct Desc { unsigned long long w[4]; };

__device__ __forceinline__ Desc load_desc(const uint8_t* p) {
  const uint4* q = (const uint4*)p;  // 32-byte rows of a contiguous [n][32] array are 16-byte aligned
  const uint4 a = q[0], b = q[1];
  Desc d;
  d.w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  d.w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  d.w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  d.w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
  return d;
}
__device__ __forceinline__ int hamming(const Desc& a, const Desc& b) {
  return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]);
}

constexpr unsigned long long kNoKey = ~0ull;

// keep the two smallest keys.  By value: with reference parameters the pair ended up in scratch memory (a private segment of
// 24 bytes on k_nn_csr / k_knn2, and a kernel with a private segment pays a scratch set-up on every queue that first runs it).
struct Top2 { unsigned long long k1, k2; };
__device__ __forceinline__ Top2 top2_push(Top2 t, unsigned long long k) {
  if (k < t.k1) { t.k2 = t.k1; t.k1 = k; }
  else if (k < t.k2) t.k2 = k;
  return t;
}
__device__ __forceinline__ Top2 top2_wave_reduce(Top2 t) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long p1 = __shfl_xor(t.k1, o), p2 = __shfl_xor(t.k2, o);
    const unsigned long long lo = t.k1 < p1 ? t.k1 : p1, hi = t.k1 < p1 ? p1 : t.k1;
    const unsigned long long s2 = t.k2 < p2 ? t.k2 : p2;
    t.k1 = lo;
    t.k2 = hi < s2 ? hi : s2;
  }
  return t;
}

// Guided NN over CSR candidate lists: one wave per query, lanes stride the candidates.
// key = dist << 32 | position   (first minimum wins, strict `<`)  or
//       dist << 32 | ~position  (last minimum wins, SearchForTriangulation's `<=`).
__global__ __launch_bounds__(256) void k_nn_csr(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ tr,
                                                const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ cand,
                                                int last_wins, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                int32_t* __restrict__ second_idx, int32_t* __restrict__ second_dist,
                                                int32_t* __restrict__ dist_out, unsigned* __restrict__ ctr,
                                                unsigned long long* __restrict__ done_host) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ int wg_done;
  if (done_host) {   // block-uniform
    if (threadIdx.x == 0) wg_done = 0;
    __syncthreads();
  }
  if (qi >= nq) return;
  const Desc dq = load_desc(q + (size_t)qi * 32);
  const int b = row_ptr[qi], e = row_ptr[qi + 1];
  Top2 t2{kNoKey, kNoKey};
  for (int c = b + lane; c < e; c += 64) {
    const int ti = cand[c];
    const int d = hamming(dq, load_desc(tr + (size_t)ti * 32));
    if (dist_out) dist_out[c] = d;
    const uint32_t pos = (uint32_t)(c - b);
    t2 = top2_push(t2, ((unsigned long long)d << 32) | (last_wins ? (0xffffffffu - pos) : pos));
  }
  t2 = top2_wave_reduce(t2);
  const unsigned long long k1 = t2.k1, k2 = t2.k2;
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const unsigned long long k = r ? k2 : k1;
      int32_t* oi = r ? second_idx : best_idx;
      int32_t* od = r ? second_dist : best_dist;
      if (k == kNoKey) { if (oi) oi[qi] = -1; if (od) od[qi] = 256; continue; }
      uint32_t pos = (uint32_t)k;
      if (last_wins) pos = 0xffffffffu - pos;
      if (oi) oi[qi] = cand[b + (int)pos];
      if (od) od[qi] = (int32_t)(k >> 32);
    }
  }
  // host-buffer calls write their results into mapped pinned memory: the last wave of every workgroup makes the workgroup's stores visible to the host
  // before it counts the workgroup, the last workgroup raises the done word the host polls (the counter wraps back to zero)
  // (a system-scope release writes the L2 back: once per workgroup, by the last of its waves, not once per query — orbx_window.hip)
  if (done_host) {
    const int nactive = min(4, nq - (int)blockIdx.x * 4);
    // a workgroup-scope release emits no vmcnt wait on gfx9 outside tgsplit mode: drain this wave's own stores (records, pool entries)
    // before it counts itself — the last wave's wide fence below only waits for ITS outstanding stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0 && atomicAdd(&wg_done, 1) == nactive - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __threadfence_system();
      if (atomicInc(ctr, gridDim.x - 1) == gridDim.x - 1) {
        __threadfence();
        __atomic_store_n(done_host, 1ull, __ATOMIC_RELEASE);
        __threadfence_system();
      }
    }
  }
}

// Guided matching inside vocabulary nodes (SearchByBoW, SearchForTriangulation): every query of a group (= a node both sides share) is
// compared with the group's whole candidate list, but the host replays only ever use candidates within a distance bound (TH_LOW, or the
// bound beyond which the ratio test passes anyway): a wave per query counts its candidates within `max_dist`, reserves that many entries
// of a pool with one atomic, and writes (train index, distance) in list order.  The O(pairs) arrays of the CSR form (candidate lists
// repeated per query, every distance shipped back) never exist: the host builds O(queries + features) and reads O(near candidates).
// pool_ctr: [0] = entries reserved so far (reset by the last workgroup), [1] = workgroup counter of the publication protocol.
__global__ __launch_bounds__(256) void k_nn_groups(const uint8_t* __restrict__ q, const int32_t* __restrict__ q_group, int nq,
                                                   const uint8_t* __restrict__ tr, const int32_t* __restrict__ group_ptr,
                                                   const int32_t* __restrict__ group_cand, int max_dist, int pool_cap,
                                                   int32_t* __restrict__ q_off, int32_t* __restrict__ q_cnt, int2* __restrict__ pool,
                                                   unsigned* __restrict__ pool_ctr, unsigned long long* __restrict__ done_host) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ int wg_done;
  if (threadIdx.x == 0) wg_done = 0;
  __syncthreads();
  if (qi >= nq) return;
  const Desc dq = load_desc(q + (size_t)qi * 32);
  const int g = q_group[qi];
  const int b = group_ptr[g], e = group_ptr[g + 1];
  int cnt = 0;
  for (int c0 = b; c0 < e; c0 += 64) {
    const int c = c0 + lane;
    const bool near = c < e && hamming(dq, load_desc(tr + (size_t)group_cand[c] * 32)) <= max_dist;
    cnt += __popcll(__ballot(near));
  }
  int base = 0;
  if (lane == 0 && cnt > 0) base = (int)atomicAdd(&pool_ctr[0], (unsigned)cnt);
  base = __builtin_amdgcn_readfirstlane(base);
  const bool fits = cnt > 0 && base + cnt <= pool_cap;
  if (lane == 0) { q_off[qi] = base; q_cnt[qi] = cnt; }   // the host sees the total and refuses an overflowing pass as a whole
  if (fits) {
    int run = base;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int c0 = b; c0 < e; c0 += 64) {
      const int c = c0 + lane;
      int ti = 0, d = 0;
      if (c < e) { ti = group_cand[c]; d = hamming(dq, load_desc(tr + (size_t)ti * 32)); }
      const bool near = c < e && d <= max_dist;
      const unsigned long long m = __ballot(near);
      if (near) pool[run + __popcll(m & lt)] = make_int2(ti, d);
      run += __popcll(m);
    }
  }
  const int nactive = min(4, nq - (int)blockIdx.x * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0 && atomicAdd(&wg_done, 1) == nactive - 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __threadfence_system();
    if (atomicInc(&pool_ctr[1], gridDim.x - 1) == gridDim.x - 1) {
      __threadfence();
      const unsigned total = atomicExch(&pool_ctr[0], 0u);   // every workgroup has reserved: the final count, and the counter is zero for the next call
      __atomic_store_n(done_host, ((unsigned long long)total << 1) | 1ull, __ATOMIC_RELEASE);
      __threadfence_system();
    }
  }
}

// All-pairs 2-NN, small problems (the SLAM sizes, ~1000 x 1000, are launch-bound): wave per query, the train set is
// streamed through LDS in tiles shared by the block's 4 queries; one launch.
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ tr, int nt,
                                              int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
  __shared__ uint4 tile[256 * 2];  // 256 train descriptors
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + w;
  Desc dq;
  if (qi < nq) dq = load_desc(q + (size_t)qi * 32);
  else dq.w[0] = dq.w[1] = dq.w[2] = dq.w[3] = 0;
  Top2 t2{kNoKey, kNoKey};
  for (int t0 = 0; t0 < nt; t0 += 256) {
    const int nload = min(256, nt - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nload * 2; i += 256) tile[i] = ((const uint4*)(tr + (size_t)t0 * 32))[i];
    __syncthreads();
    for (int c = lane; c < nload; c += 64) {
      const uint4 a = tile[c * 2], b = tile[c * 2 + 1];
      Desc dt;
      dt.w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
      dt.w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
      dt.w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
      dt.w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
      t2 = top2_push(t2, ((unsigned long long)hamming(dq, dt) << 32) | (uint32_t)(t0 + c));
    }
  }
  t2 = top2_wave_reduce(t2);
  const unsigned long long k1 = t2.k1, k2 = t2.k2;
  if (lane == 0 && qi < nq) {
    idx[qi * 2] = k1 == kNoKey ? -1 : (int32_t)(uint32_t)k1;
    dist[qi * 2] = k1 == kNoKey ? 256 : (int32_t)(k1 >> 32);
    idx[qi * 2 + 1] = k2 == kNoKey ? -1 : (int32_t)(uint32_t)k2;
    dist[qi * 2 + 1] = k2 == kNoKey ? 256 : (int32_t)(k2 >> 32);
  }
}

// All-pairs 2-NN (cv::BFMatcher(NORM_HAMMING).knnMatch(k=2), src/Frame.cc:1144).  Large problems: VALU-bound formulation: every lane
// owns ONE query (its 32 bytes live in 8 VGPRs) and the whole wave walks the same train descriptor, whose address is
// wave-uniform, so it arrives through the scalar cache and is used as SGPR operands: per pair 8 xor + 8 accumulating
// popcounts + the top-2 update, no LDS and no per-lane memory traffic.  The train set is cut into segments
// (blockIdx.y) for parallelism; k_knn2_merge takes, per query, the two smallest (distance, train index) keys of the
// per-segment partial results (ties -> lower train index, like a sequential scan).
constexpr int kKnnSeg = 64;  // train descriptors per segment

__device__ __forceinline__ void knn2_merge_query(const unsigned long long* part, int nq, int nseg, int qi, int32_t* idx, int32_t* dist) {
  unsigned long long k1 = kNoKey, k2 = kNoKey;
  for (int s = 0; s < nseg; s++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const unsigned long long key = part[((size_t)s * nq + qi) * 2 + j];
      const bool lt = key < k1;
      const unsigned long long hi = lt ? k1 : key;
      k1 = lt ? key : k1;
      k2 = hi < k2 ? hi : k2;
    }
  }
  idx[qi * 2] = k1 == kNoKey ? -1 : (int32_t)(uint32_t)k1;
  dist[qi * 2] = k1 == kNoKey ? 256 : (int32_t)(k1 >> 32);
  idx[qi * 2 + 1] = k2 == kNoKey ? -1 : (int32_t)(uint32_t)k2;
  dist[qi * 2 + 1] = k2 == kNoKey ? 256 : (int32_t)(k2 >> 32);
}

__global__ __launch_bounds__(256) void k_knn2_partial(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ tr,
                                                      int nt, unsigned long long* __restrict__ part) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const int seg = blockIdx.y, t0 = seg * kKnnSeg, t1 = min(t0 + kKnnSeg, nt);
  uint32_t a[8];
  {
    const uint4* qp = (const uint4*)(q + (size_t)min(qi, nq - 1) * 32);
    const uint4 lo = qp[0], hi = qp[1];
    a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w; a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
  }
  // 32-bit keys: distance (0..256) << 22 | train index (< 2^22): the top-2 update is three min/max instructions
  uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
#pragma unroll 4
  for (int t = t0; t < t1; t++) {
    const uint32_t* tp = (const uint32_t*)(tr + (size_t)t * 32);  // wave-uniform address -> scalar loads
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) d += __popc(a[w] ^ tp[w]);
    const uint32_t key = (d << 22) | (uint32_t)t;
    const uint32_t hi = max(k1, key);
    k1 = min(k1, key);
    k2 = min(k2, hi);
  }
  if (qi < nq) {
    auto widen = [](uint32_t k) { return k == 0xffffffffu ? kNoKey : (((unsigned long long)(k >> 22)) << 32) | (k & 0x3fffffu); };
    part[((size_t)seg * nq + qi) * 2] = widen(k1);
    part[((size_t)seg * nq + qi) * 2 + 1] = widen(k2);
  }
}

__global__ __launch_bounds__(256) void k_knn2_merge(const unsigned long long* __restrict__ part, int nq, int nseg,
                                                    int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi < nq) knn2_merge_query(part, nq, nseg, qi, idx, dist);
}

// ---------------------------------------------------------------------------------------------------
// Vocabulary tree (DBoW2 TemplatedVocabulary<cv::Mat, FORB>)
// ---------------------------------------------------------------------------------------------------
struct DevNode { int32_t child_begin, nchild, word_id, pad; double weight; };

// wave per feature: lanes < nchild evaluate one child each; first minimum wins (TemplatedVocabulary.h:1238-1249)
struct BowLeaf { int word_id; uint32_t nid; double weight; };
__device__ __forceinline__ BowLeaf bow_descend_wave(const DevNode* __restrict__ nodes, const uint8_t* __restrict__ slot_desc,
                                                    const int32_t* __restrict__ slot_node, const Desc df, int L, int levelsup, int lane) {
  const int nid_level = L - levelsup;
  uint32_t nid = 0;
  int final_id = 0, level = 0;
  DevNode nd = nodes[0];
  while (nd.nchild > 0) {
    ++level;
    unsigned long long key = kNoKey;
    for (int c = lane; c < nd.nchild; c += 64) {
      const int d = hamming(df, load_desc(slot_desc + (size_t)(nd.child_begin + c) * 32));
      const unsigned long long k = ((unsigned long long)d << 32) | (uint32_t)c;
      key = k < key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long p = __shfl_xor(key, o); key = p < key ? p : key; }
    final_id = slot_node[nd.child_begin + (int)(uint32_t)key];
    if (level == nid_level) nid = (uint32_t)final_id;
    nd = nodes[final_id];
  }
  return BowLeaf{nd.word_id, nid, nd.weight};
}

__global__ __launch_bounds__(256) void k_bow_descend(const DevNode* __restrict__ nodes, const uint8_t* __restrict__ slot_desc,
                                                     const int32_t* __restrict__ slot_node, const uint8_t* __restrict__ desc,
                                                     int n, int L, int levelsup, uint32_t* __restrict__ word,
                                                     double* __restrict__ weight, uint32_t* __restrict__ node_out) {
  const int lane = threadIdx.x & 63;
  const int fi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fi >= n) return;
  const BowLeaf lf = bow_descend_wave(nodes, slot_desc, slot_node, load_desc(desc + (size_t)fi * 32), L, levelsup, lane);
  if (lane == 0) {
    word[fi] = (uint32_t)lf.word_id;
    weight[fi] = lf.weight;
    node_out[fi] = lf.nid;
  }
}

// The host-buffer form (orbx_bow_transform): descriptors read from, and {word, node, weight} records written to, the call's mapped
// pinned blob; the last workgroup raises the done word the host polls (publication protocol of k_nn_csr above).  One launch, no copy
// operation, no stream synchronisation: what is left of a call is the launch and the PCIe round trip.
struct BowRecord { uint32_t word, node; double weight; };
__global__ __launch_bounds__(256) void k_bow_descend_direct(const DevNode* __restrict__ nodes, const uint8_t* __restrict__ slot_desc,
                                                            const int32_t* __restrict__ slot_node, const uint8_t* __restrict__ desc_host, int n,
                                                            int L, int levelsup, BowRecord* __restrict__ rec_host, unsigned* __restrict__ ctr,
                                                            unsigned long long* __restrict__ done_host) {
  const int lane = threadIdx.x & 63;
  const int fi = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ int wg_done;
  if (threadIdx.x == 0) wg_done = 0;
  __syncthreads();
  if (fi >= n) return;
  const BowLeaf lf = bow_descend_wave(nodes, slot_desc, slot_node, load_desc(desc_host + (size_t)fi * 32), L, levelsup, lane);
  if (lane == 0) rec_host[fi] = BowRecord{(uint32_t)lf.word_id, lf.nid, lf.weight};
  const int nactive = min(4, n - (int)blockIdx.x * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's record has left before the wave counts itself
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0 && atomicAdd(&wg_done, 1) == nactive - 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __threadfence_system();
    if (atomicInc(ctr, gridDim.x - 1) == gridDim.x - 1) {
      __threadfence();
      __atomic_store_n(done_host, 1ull, __ATOMIC_RELEASE);
      __threadfence_system();
    }
  }
}

// The single-frame extraction graph's tail (orbx_voc attached to the extractor context): the same descent over the descriptors the graph has
// just produced; the frame's keypoint count is only known on the device.  Records go straight into the caller-visible pinned block.
__global__ __launch_bounds__(256) void k_bow_descend_graph(const DevNode* __restrict__ nodes, const uint8_t* __restrict__ slot_desc,
                                                           const int32_t* __restrict__ slot_node, const uint8_t* __restrict__ desc,
                                                           const int32_t* __restrict__ counts, int L, int levelsup, BowRecord* __restrict__ rec) {
  const int lane = threadIdx.x & 63;
  const int fi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fi >= counts[0]) return;
  const BowLeaf lf = bow_descend_wave(nodes, slot_desc, slot_node, load_desc(desc + (size_t)fi * 32), L, levelsup, lane);
  if (lane == 0) rec[fi] = BowRecord{(uint32_t)lf.word_id, lf.nid, lf.weight};
}

// L1Scoring::score (ScoringObject.cpp:23-68): thread per database vector, sequential merge in ascending id
// order so the double accumulation order equals std::map iteration.
__global__ __launch_bounds__(256) void k_bow_score_l1(const uint32_t* __restrict__ qid, const double* __restrict__ qv, int nq,
                                                      const int32_t* __restrict__ db_ptr, const uint32_t* __restrict__ did,
                                                      const double* __restrict__ dv, int ndb, double* __restrict__ scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ndb) return;
  int a = 0, b = db_ptr[i];
  const int be = db_ptr[i + 1];
  double score = 0;
  while (a < nq && b < be) {
    const uint32_t ia = qid[a], ib = did[b];
    if (ia == ib) {
      const double vi = qv[a], wi = dv[b];
      const double t = __dsub_rn(__dsub_rn(fabs(__dsub_rn(vi, wi)), fabs(vi)), fabs(wi));
      score = __dadd_rn(score, t);
      ++a; ++b;
    } else if (ia < ib) ++a;
    else ++b;
  }
  scores[i] = -score / 2.0;
}

}  // namespace orbx

struct orbx_voc {
  orbx_ctx* ctx = nullptr;
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<int32_t> parent;            // per node (0 = root)
  std::vector<std::vector<int32_t>> children;
  std::vector<uint8_t> is_leaf;
  std::vector<uint8_t> desc;              // [nnodes][32]
  std::vector<double> weight;
  std::vector<int32_t> word_id;           // per node, -1 for inner nodes
  int nwords = 0;
  // device copy
  orbx::DevNode* d_nodes = nullptr;
  uint8_t* d_slot_desc = nullptr;
  int32_t* d_slot_node = nullptr;
};

hipError_t orbx::launch_bow_records(const orbx_voc* v, const uint8_t* d_desc, const int32_t* d_counts, int cap, int levelsup, void* rec_out,
                                    hipStream_t st) {
  if (!v || v->parent.size() <= 1 || !v->d_nodes || cap <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_bow_descend_graph, dim3((cap + 3) / 4), dim3(256), 0, st, v->d_nodes, v->d_slot_desc, v->d_slot_node, d_desc, d_counts, v->L, levelsup,
                     (BowRecord*)rec_out);
  return hipGetLastError();
}
int orbx::voc_device(const orbx_voc* v) { return v && v->ctx ? v->ctx->device : -1; }

using namespace orbx;

namespace {

int upload_voc(orbx_voc* v) {
  orbx_ctx* ctx = v->ctx;
  const int nn = (int)v->parent.size();
  std::vector<DevNode> nodes(nn);
  std::vector<uint8_t> sdesc;
  std::vector<int32_t> snode;
  for (int i = 0; i < nn; i++) {
    DevNode& d = nodes[i];
    d.child_begin = (int32_t)snode.size();
    d.nchild = (int32_t)v->children[i].size();
    d.word_id = v->word_id[i];
    d.pad = 0;
    d.weight = v->weight[i];
    for (int32_t c : v->children[i]) {
      snode.push_back(c);
      sdesc.insert(sdesc.end(), v->desc.begin() + (size_t)c * 32, v->desc.begin() + (size_t)c * 32 + 32);
    }
  }
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ORBX_HIP(ctx, hipMalloc((void**)&v->d_nodes, sizeof(DevNode) * nn));
  ORBX_HIP(ctx, copy_sync(ctx, v->d_nodes, nodes.data(), sizeof(DevNode) * nn, hipMemcpyHostToDevice));
  ORBX_HIP(ctx, hipMalloc((void**)&v->d_slot_desc, std::max<size_t>(sdesc.size(), 32)));
  ORBX_HIP(ctx, hipMalloc((void**)&v->d_slot_node, std::max<size_t>(snode.size(), 1) * sizeof(int32_t)));
  if (!snode.empty()) {
    ORBX_HIP(ctx, copy_sync(ctx, v->d_slot_desc, sdesc.data(), sdesc.size(), hipMemcpyHostToDevice));
    ORBX_HIP(ctx, copy_sync(ctx, v->d_slot_node, snode.data(), snode.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return ORBX_OK;
}

int finish_voc(orbx_voc* v) {
  const int nn = (int)v->parent.size();
  v->children.assign(nn, {});
  v->word_id.assign(nn, -1);
  v->nwords = 0;
  for (int i = 1; i < nn; i++) {
    const int p = v->parent[i];
    if (p < 0 || p >= nn) return set_err(v->ctx, ORBX_E_FORMAT, "vocabulary: parent id out of range");
    v->children[p].push_back(i);
    if (v->is_leaf[i]) v->word_id[i] = v->nwords++;
  }
  for (int i = 0; i < nn; i++)
    if (!v->is_leaf[i] && v->children[i].empty() && nn > 1 && i != 0)
      return set_err(v->ctx, ORBX_E_FORMAT, "vocabulary: inner node without children");
  return upload_voc(v);
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

}  // namespace

extern "C" {

int orbx_hamming(const uint8_t a[32], const uint8_t b[32]) {
  int d = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t x, y;
    std::memcpy(&x, a + 8 * i, 8);
    std::memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

int orbx_nn_csr_device(orbx_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const int32_t* d_row_ptr,
                       const int32_t* d_cand, int last_wins, int32_t* d_best_idx, int32_t* d_best_dist,
                       int32_t* d_second_idx, int32_t* d_second_dist, int32_t* d_dist_out, void* stream) {
  if (!ctx || nq < 0 || nt < 0) return ORBX_E_INVALID;
  if (nq == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  hipLaunchKernelGGL(k_nn_csr, dim3((nq + 3) / 4), dim3(256), 0, st, d_q, nq, d_t, d_row_ptr, d_cand, last_wins, d_best_idx,
                     d_best_dist, d_second_idx, d_second_dist, d_dist_out, (unsigned*)nullptr, (unsigned long long*)nullptr);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

// input blob from mapped pinned memory into HBM (the kernel gathers train descriptors at random); cheaper than a hipMemcpyAsync
__global__ __launch_bounds__(256) void k_nn_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

int orbx_nn_csr(orbx_ctx* ctx, const uint8_t* q_desc, int nq, const uint8_t* t_desc, int nt, const int32_t* row_ptr,
                const int32_t* cand, int last_wins, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                int32_t* second_dist, int32_t* dist_out) {
  if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!q_desc || !row_ptr))) return ORBX_E_INVALID;
  if (nq == 0) return ORBX_OK;
  const int nnz = row_ptr[nq];
  if (nnz < 0 || (nnz > 0 && (!cand || !t_desc))) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_csr: bad candidate lists");
  for (int i = 0; i < nnz; i++)
    if (cand[i] < 0 || cand[i] >= nt) return set_err(ctx, ORBX_E_INVALID, "candidate index out of range");
  static const bool trace = getenv("ORBX_TRACE_WINDOW") != nullptr;   // phase times of every call on stderr (diagnostics)
  const auto tr0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  // one packed blob in (one H2D copy), one launch, one blob out (one D2H copy): no per-call allocation
  const bool want_best = best_idx || best_dist || second_idx || second_dist;
  BlobLayout in, out;
  const size_t o_q = in.add((size_t)nq * 32), o_t = in.add((size_t)nt * 32), o_rp = in.add(4 * (size_t)(nq + 1)), o_c = in.add(4 * (size_t)nnz);
  const size_t p_b = out.add(want_best ? 16 * (size_t)nq : 0), p_d = out.add(dist_out ? 4 * (size_t)nnz : 0), p_done = out.add(16);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, in.size + out.size, &h));
  uint8_t* hin = h;
  uint8_t* hout = h + in.size;
  std::memcpy(hin + o_q, q_desc, (size_t)nq * 32);
  if (nt) std::memcpy(hin + o_t, t_desc, (size_t)nt * 32);
  std::memcpy(hin + o_rp, row_ptr, 4 * (size_t)(nq + 1));
  if (nnz) std::memcpy(hin + o_c, cand, 4 * (size_t)nnz);
  const double us_pack = since();
  ctx->arena.rewind();
  hipError_t aerr = hipSuccess;
  uint8_t* din = (uint8_t*)ctx->arena.alloc(in.size, &aerr);
  ORBX_HIP(ctx, aerr);
  uint8_t* dout = (uint8_t*)ctx->arena.alloc(std::max<size_t>(out.size, 256), &aerr);
  ORBX_HIP(ctx, aerr);
  hipStream_t st = ctx->stream;
  // direct mode (as the window pass, orbx_window.hip): inputs pulled by a copy kernel from the mapped pinned blob, results written
  // into it, the host polls a done word; "window_direct" = 0 selects copies + stream synchronisation
  uint8_t* hdev = nullptr;
  const bool direct = ctx->window_direct && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  double us_issue;
  if (direct) {
    if (!ctx->d_win_ctr) { ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_win_ctr, 64)); ctx->win_ctr_dirty = true; }
    if (ctx->win_ctr_dirty) { ORBX_HIP(ctx, hipMemsetAsync(ctx->d_win_ctr, 0, 64, st)); ctx->win_ctr_dirty = false; }
    volatile unsigned long long* done = (volatile unsigned long long*)(hout + p_done);
    __atomic_store_n(done, 0ull, __ATOMIC_RELEASE);
    const int n16 = (int)((in.size + 15) / 16);
    hipLaunchKernelGGL(k_nn_stage_in, dim3(std::min((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4*)hdev, (uint4*)din, n16);
    int32_t* rb = (int32_t*)(hdev + in.size + p_b);
    ctx->win_ctr_dirty = true;
    hipLaunchKernelGGL(k_nn_csr, dim3((nq + 3) / 4), dim3(256), 0, st, din + o_q, nq, din + o_t, (const int32_t*)(din + o_rp),
                       (const int32_t*)(din + o_c), last_wins, want_best ? rb : nullptr, want_best ? rb + nq : nullptr,
                       want_best ? rb + 2 * (size_t)nq : nullptr, want_best ? rb + 3 * (size_t)nq : nullptr,
                       dist_out ? (int32_t*)(hdev + in.size + p_d) : nullptr, (unsigned*)ctx->d_win_ctr + 8, (unsigned long long*)(hdev + in.size + p_done));
    ORBX_HIP(ctx, hipGetLastError());
    us_issue = since();
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {
          if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
          return set_err(ctx, ORBX_E_DEVICE, "orbx_nn_csr: the pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    ctx->win_ctr_dirty = false;
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(din, hin, in.size, hipMemcpyHostToDevice, st));
    int32_t* db = (int32_t*)(dout + p_b);
    int rc = orbx_nn_csr_device(ctx, din + o_q, nq, din + o_t, nt, (const int32_t*)(din + o_rp), (const int32_t*)(din + o_c), last_wins,
                                want_best ? db : nullptr, want_best ? db + nq : nullptr, want_best ? db + 2 * (size_t)nq : nullptr,
                                want_best ? db + 3 * (size_t)nq : nullptr, dist_out ? (int32_t*)(dout + p_d) : nullptr, st);
    if (rc != ORBX_OK) return rc;
    if (out.size) ORBX_HIP(ctx, hipMemcpyAsync(hout, dout, out.size, hipMemcpyDeviceToHost, st));
    us_issue = since();
    ORBX_HIP(ctx, hipStreamSynchronize(st));
  }
  const double us_sync = since();
  const int32_t* hb = (const int32_t*)(hout + p_b);
  if (best_idx) std::memcpy(best_idx, hb, 4 * (size_t)nq);
  if (best_dist) std::memcpy(best_dist, hb + nq, 4 * (size_t)nq);
  if (second_idx) std::memcpy(second_idx, hb + 2 * (size_t)nq, 4 * (size_t)nq);
  if (second_dist) std::memcpy(second_dist, hb + 3 * (size_t)nq, 4 * (size_t)nq);
  if (dist_out && nnz) std::memcpy(dist_out, hout + p_d, 4 * (size_t)nnz);
  if (trace)
    std::fprintf(stderr, "[orbx window] orbx_nn_csr nq=%d nt=%d pairs=%d in=%zu B out=%zu B: pack %.1f us, issue %.1f, wait %.1f, scatter %.1f\n", nq, nt, nnz,
                 in.size, out.size, us_pack, us_issue - us_pack, us_sync - us_issue, since() - us_sync);
  return ORBX_OK;
}

int orbx_nn_groups(orbx_ctx* ctx, const uint8_t* q_desc, const int32_t* q_group, int nq, const uint8_t* t_desc, int nt, const int32_t* group_ptr,
                   const int32_t* group_cand, int ngroups, int max_dist, int32_t* q_off, int32_t* q_cnt, orbx_candidate* entries, int pool_cap,
                   int* n_entries) {
  if (n_entries) *n_entries = 0;
  if (!ctx || nq < 0 || nt < 0 || ngroups < 0 || pool_cap < 0 || !n_entries || (nq > 0 && (!q_desc || !q_group || !group_ptr || !q_off || !q_cnt)) ||
      (pool_cap > 0 && !entries))
    return ORBX_E_INVALID;
  if (nq == 0) return ORBX_OK;
  if (ngroups == 0 || group_ptr[0] != 0) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_groups: bad group lists");
  for (int g = 0; g < ngroups; g++)
    if (group_ptr[g + 1] < group_ptr[g]) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_groups: group_ptr not ascending");
  const int ncand = group_ptr[ngroups];
  if (ncand > 0 && (!group_cand || !t_desc)) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_groups: bad candidate lists");
  for (int i = 0; i < ncand; i++)
    if (group_cand[i] < 0 || group_cand[i] >= nt) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_groups: candidate index out of range");
  for (int i = 0; i < nq; i++)
    if (q_group[i] < 0 || q_group[i] >= ngroups) return set_err(ctx, ORBX_E_INVALID, "orbx_nn_groups: query group out of range");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  BlobLayout in, out;
  const size_t o_q = in.add((size_t)nq * 32), o_g = in.add(4 * (size_t)nq), o_t = in.add((size_t)nt * 32), o_gp = in.add(4 * (size_t)(ngroups + 1)),
               o_gc = in.add(4 * (size_t)std::max(ncand, 1));
  const size_t p_off = out.add(4 * (size_t)nq), p_cnt = out.add(4 * (size_t)nq), p_pool = out.add(8 * (size_t)std::max(pool_cap, 1)), p_done = out.add(16);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, in.size + out.size, &h));
  uint8_t* hout = h + in.size;
  std::memcpy(h + o_q, q_desc, (size_t)nq * 32);
  std::memcpy(h + o_g, q_group, 4 * (size_t)nq);
  if (nt) std::memcpy(h + o_t, t_desc, (size_t)nt * 32);
  std::memcpy(h + o_gp, group_ptr, 4 * (size_t)(ngroups + 1));
  if (ncand) std::memcpy(h + o_gc, group_cand, 4 * (size_t)ncand);
  ctx->arena.rewind();
  hipError_t aerr = hipSuccess;
  uint8_t* din = (uint8_t*)ctx->arena.alloc(in.size, &aerr);
  ORBX_HIP(ctx, aerr);
  uint8_t* dout = (uint8_t*)ctx->arena.alloc(out.size, &aerr);
  ORBX_HIP(ctx, aerr);
  hipStream_t st = ctx->stream;
  if (!ctx->d_win_ctr) { ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_win_ctr, 64)); ctx->win_ctr_dirty = true; }
  if (ctx->win_ctr_dirty) { ORBX_HIP(ctx, hipMemsetAsync(ctx->d_win_ctr, 0, 64, st)); ctx->win_ctr_dirty = false; }
  uint8_t* hdev = nullptr;
  const bool direct = ctx->window_direct && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  volatile unsigned long long* done = (volatile unsigned long long*)(hout + p_done);
  __atomic_store_n(done, 0ull, __ATOMIC_RELEASE);
  ctx->win_ctr_dirty = true;
  const int n16 = (int)((in.size + 15) / 16);
  uint8_t* res = direct ? hdev + in.size : dout;   // where the kernel writes: the mapped blob itself, or HBM + one copy back
  if (direct) hipLaunchKernelGGL(k_nn_stage_in, dim3(std::min((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4*)hdev, (uint4*)din, n16);
  else ORBX_HIP(ctx, hipMemcpyAsync(din, h, in.size, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_nn_groups, dim3((nq + 3) / 4), dim3(256), 0, st, din + o_q, (const int32_t*)(din + o_g), nq, din + o_t, (const int32_t*)(din + o_gp),
                     (const int32_t*)(din + o_gc), max_dist, pool_cap, (int32_t*)(res + p_off), (int32_t*)(res + p_cnt), (int2*)(res + p_pool),
                     (unsigned*)ctx->d_win_ctr + 10, (unsigned long long*)(res + p_done));
  ORBX_HIP(ctx, hipGetLastError());
  if (direct) {
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {
          if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
          return set_err(ctx, ORBX_E_DEVICE, "orbx_nn_groups: the pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(hout, dout, out.size, hipMemcpyDeviceToHost, st));
    ORBX_HIP(ctx, hipStreamSynchronize(st));
  }
  ctx->win_ctr_dirty = false;
  const unsigned long long dw = *done;
  const int total = (int)(dw >> 1);
  *n_entries = total;
  if (total > pool_cap) return set_err(ctx, ORBX_E_CAPACITY, "orbx_nn_groups: more near candidates than the pool holds");
  std::memcpy(q_off, hout + p_off, 4 * (size_t)nq);
  std::memcpy(q_cnt, hout + p_cnt, 4 * (size_t)nq);
  if (total) std::memcpy(entries, hout + p_pool, 8 * (size_t)total);
  return ORBX_OK;
}

int orbx_knn2_allpairs_device(orbx_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                              int32_t* d_dist, void* stream) {
  if (!ctx || nq < 0 || nt < 0) return ORBX_E_INVALID;
  if (nq == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  if (nt <= 2048) {  // launch-bound sizes: one launch, wave per query
    hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, st, d_q, nq, d_t, nt, d_idx, d_dist);
    ORBX_HIP(ctx, hipGetLastError());
    return ORBX_OK;
  }
  const int nseg = std::max(1, (nt + kKnnSeg - 1) / kKnnSeg), nqb = (nq + 255) / 256;
  if (nseg > 65535 || nt >= (1 << 22)) return set_err(ctx, ORBX_E_CAPACITY, "knn2: more than 4 M train descriptors");
  const size_t need = (size_t)nseg * nq * 2 * sizeof(unsigned long long);
  if (need > ctx->knn_ws_bytes) {  // grow-only workspace for the per-segment partial top-2
    ORBX_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->d_knn_ws) (void)hipFree(ctx->d_knn_ws);
    ctx->d_knn_ws = nullptr; ctx->knn_ws_bytes = 0;
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_knn_ws, need));
    ctx->knn_ws_bytes = need;
  }
  hipLaunchKernelGGL(k_knn2_partial, dim3(nqb, nseg), dim3(256), 0, st, d_q, nq, d_t, nt, ctx->d_knn_ws);
  hipLaunchKernelGGL(k_knn2_merge, dim3(nqb), dim3(256), 0, st, ctx->d_knn_ws, nq, nseg, d_idx, d_dist);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

int orbx_knn2_allpairs(orbx_ctx* ctx, const uint8_t* q_desc, int nq, const uint8_t* t_desc, int nt, int32_t* idx,
                       int32_t* dist) {
  if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!q_desc || !idx || !dist))) return ORBX_E_INVALID;
  if (nq == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf<uint8_t> dq, dt;
  DevBuf<int32_t> di, dd;
  ORBX_HIP(ctx, dq.alloc((size_t)nq * 32)); ORBX_HIP(ctx, dt.alloc((size_t)nt * 32));
  ORBX_HIP(ctx, di.alloc((size_t)nq * 2)); ORBX_HIP(ctx, dd.alloc((size_t)nq * 2));
  ORBX_HIP(ctx, copy_sync(ctx, dq.p, q_desc, (size_t)nq * 32, hipMemcpyHostToDevice));
  if (nt) ORBX_HIP(ctx, copy_sync(ctx, dt.p, t_desc, (size_t)nt * 32, hipMemcpyHostToDevice));
  int rc = orbx_knn2_allpairs_device(ctx, dq.p, nq, dt.p, nt, di.p, dd.p, ctx->stream);
  if (rc != ORBX_OK) return rc;
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ORBX_HIP(ctx, copy_sync(ctx, idx, di.p, sizeof(int32_t) * nq * 2, hipMemcpyDeviceToHost));
  ORBX_HIP(ctx, copy_sync(ctx, dist, dd.p, sizeof(int32_t) * nq * 2, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

// ---- vocabulary ------------------------------------------------------------------------------------

int orbx_voc_create(orbx_ctx* ctx, int k, int L, int scoring, int weighting, int n, const int32_t* parent,
                    const uint8_t* is_leaf, const uint8_t* desc, const double* weight, orbx_voc** out) {
  if (!ctx || !out || n < 0 || (n > 0 && (!parent || !is_leaf || !desc || !weight))) return ORBX_E_INVALID;
  *out = nullptr;
  if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
    return set_err(ctx, ORBX_E_FORMAT, "vocabulary header out of range (TemplatedVocabulary.h:1359)");
  orbx_voc* v = new (std::nothrow) orbx_voc();
  if (!v) return ORBX_E_CAPACITY;
  v->ctx = ctx; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->parent.assign(n + 1, 0); v->is_leaf.assign(n + 1, 0); v->desc.assign((size_t)(n + 1) * 32, 0); v->weight.assign(n + 1, 0.0);
  for (int i = 0; i < n; i++) {
    v->parent[i + 1] = parent[i];
    v->is_leaf[i + 1] = is_leaf[i] ? 1 : 0;
    std::memcpy(&v->desc[(size_t)(i + 1) * 32], desc + (size_t)i * 32, 32);
    v->weight[i + 1] = weight[i];
  }
  int rc = finish_voc(v);
  if (rc != ORBX_OK) { orbx_voc_destroy(v); return rc; }
  *out = v;
  return ORBX_OK;
}

// TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1424): "k L scoring weighting" then one
// line per node "parent isLeaf b0..b31 weight"; node ids are the 1-based line order.
int orbx_voc_load_text(orbx_ctx* ctx, const char* path, orbx_voc** out) {
  if (!ctx || !path || !out) return ORBX_E_INVALID;
  *out = nullptr;
  std::ifstream f(path);
  if (!f.is_open()) return set_err(ctx, ORBX_E_FORMAT, std::string("cannot open ") + path);
  std::string line;
  if (!std::getline(f, line)) return set_err(ctx, ORBX_E_FORMAT, "empty vocabulary file");
  int k = -1, L = -1, n1 = -1, n2 = -1;
  { std::stringstream ss(line); ss >> k >> L >> n1 >> n2; }
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)
    return set_err(ctx, ORBX_E_FORMAT, "Vocabulary loading failure: This is not a correct text file!");
  std::vector<int32_t> parent;
  std::vector<uint8_t> leaf, desc;
  std::vector<double> weight;
  while (std::getline(f, line)) {
    if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // no phantom node
    std::stringstream ss(line);
    int pid = 0, isleaf = 0;
    ss >> pid >> isleaf;
    uint8_t d[32];
    for (int i = 0; i < 32; i++) { int b = 0; ss >> b; d[i] = (uint8_t)b; }
    double w = 0;
    ss >> w;
    if (ss.fail()) return set_err(ctx, ORBX_E_FORMAT, "vocabulary: malformed node line");
    parent.push_back(pid); leaf.push_back(isleaf > 0); desc.insert(desc.end(), d, d + 32); weight.push_back(w);
  }
  return orbx_voc_create(ctx, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

// TemplatedVocabulary::saveToTextFile (TemplatedVocabulary.h:1428-1449): same bytes — the header with its double blank,
// "parent isLeaf b0 .. b31  weight" per node with the stream's default 6-significant-digit doubles (the format is lossy
// in the weights; that is the reference's format, not a choice made here).
int orbx_voc_save_text(const orbx_voc* v, const char* path) {
  if (!v || !path) return ORBX_E_INVALID;
  std::ofstream f(path);
  if (!f.is_open()) return set_err(v->ctx, ORBX_E_FORMAT, std::string("cannot open ") + path);
  f << v->k << " " << v->L << " " << " " << v->scoring << " " << v->weighting << std::endl;
  for (size_t i = 1; i < v->parent.size(); i++) {
    f << (unsigned int)v->parent[i] << " " << (v->is_leaf[i] ? 1 : 0) << " ";
    for (int b = 0; b < 32; b++) f << (int)v->desc[i * 32 + b] << " ";
    f << " " << v->weight[i] << std::endl;
  }
  f.close();
  return f.fail() ? set_err(v->ctx, ORBX_E_FORMAT, std::string("write failed: ") + path) : ORBX_OK;
}

// Binary cache of a loaded vocabulary: the text format costs seconds for ORBvoc's 1.08 M nodes and
// rounds the weights to 6 digits; the cache is exact and loads at file-read speed.  Layout (little endian):
// "ORBXVOC1", int32 k, L, scoring, weighting, int64 n, then parent[n] int32, is_leaf[n] u8, desc[n][32] u8, weight[n] f64.
static const char kVocMagic[8] = {'O', 'R', 'B', 'X', 'V', 'O', 'C', '1'};

int orbx_voc_save_binary(const orbx_voc* v, const char* path) {
  if (!v || !path) return ORBX_E_INVALID;
  std::ofstream f(path, std::ios::binary);
  if (!f.is_open()) return set_err(v->ctx, ORBX_E_FORMAT, std::string("cannot open ") + path);
  const int32_t hdr[4] = {v->k, v->L, v->scoring, v->weighting};
  const int64_t n = (int64_t)v->parent.size() - 1;
  f.write(kVocMagic, 8); f.write((const char*)hdr, sizeof(hdr)); f.write((const char*)&n, 8);
  if (n > 0) {
    f.write((const char*)(v->parent.data() + 1), n * sizeof(int32_t));
    f.write((const char*)(v->is_leaf.data() + 1), n);
    f.write((const char*)(v->desc.data() + 32), n * 32);
    f.write((const char*)(v->weight.data() + 1), n * sizeof(double));
  }
  f.close();
  return f.fail() ? set_err(v->ctx, ORBX_E_FORMAT, std::string("write failed: ") + path) : ORBX_OK;
}

int orbx_voc_load_binary(orbx_ctx* ctx, const char* path, orbx_voc** out) {
  if (!ctx || !path || !out) return ORBX_E_INVALID;
  *out = nullptr;
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return set_err(ctx, ORBX_E_FORMAT, std::string("cannot open ") + path);
  char magic[8]; int32_t hdr[4]; int64_t n = -1;
  f.read(magic, 8); f.read((char*)hdr, sizeof(hdr)); f.read((char*)&n, 8);
  if (f.fail() || std::memcmp(magic, kVocMagic, 8) != 0 || n < 0 || n > (1ll << 28))
    return set_err(ctx, ORBX_E_FORMAT, "not an orbx vocabulary cache");
  std::vector<int32_t> parent((size_t)n);
  std::vector<uint8_t> leaf((size_t)n), desc((size_t)n * 32);
  std::vector<double> weight((size_t)n);
  if (n > 0) {
    f.read((char*)parent.data(), n * sizeof(int32_t)); f.read((char*)leaf.data(), n); f.read((char*)desc.data(), n * 32);
    f.read((char*)weight.data(), n * sizeof(double));
  }
  if (f.fail()) return set_err(ctx, ORBX_E_FORMAT, "vocabulary cache truncated");
  f.peek();
  if (!f.eof()) return set_err(ctx, ORBX_E_FORMAT, "vocabulary cache has trailing bytes");
  return orbx_voc_create(ctx, hdr[0], hdr[1], hdr[2], hdr[3], (int)n, parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

void orbx_voc_destroy(orbx_voc* v) {
  if (!v) return;
  orbx::voc_detach_all(v);   // extractor contexts that descend this tree inside their single-frame graph
  if (v->d_nodes) (void)hipFree(v->d_nodes);
  if (v->d_slot_desc) (void)hipFree(v->d_slot_desc);
  if (v->d_slot_node) (void)hipFree(v->d_slot_node);
  delete v;
}

int orbx_voc_info(const orbx_voc* v, int* k, int* L, int* nnodes, int* nwords) {
  if (!v) return ORBX_E_INVALID;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (nnodes) *nnodes = (int)v->parent.size();
  if (nwords) *nwords = v->nwords;
  return ORBX_OK;
}

int orbx_bow_transform_device(orbx_voc* v, const uint8_t* d_desc, int n, int levelsup, uint32_t* d_word, double* d_weight,
                              uint32_t* d_node, void* stream) {
  if (!v || n < 0) return ORBX_E_INVALID;
  orbx_ctx* ctx = v->ctx;
  if (v->parent.size() <= 1) return set_err(ctx, ORBX_E_INVALID, "empty vocabulary");
  if (n == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  hipLaunchKernelGGL(k_bow_descend, dim3((n + 3) / 4), dim3(256), 0, st, v->d_nodes, v->d_slot_desc, v->d_slot_node, d_desc, n,
                     v->L, levelsup, d_word, d_weight, d_node);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

int orbx_bow_transform(orbx_voc* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node) {
  if (!v || n < 0 || (n > 0 && (!desc || !word || !weight || !node))) return ORBX_E_INVALID;
  if (n == 0) return ORBX_OK;
  orbx_ctx* ctx = v->ctx;
  if (v->parent.size() <= 1) return set_err(ctx, ORBX_E_INVALID, "empty vocabulary");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // one pinned, mapped blob per call: [descriptors | records | done word]
  BlobLayout lay;
  const size_t o_d = lay.add((size_t)n * 32), o_r = lay.add(sizeof(BowRecord) * (size_t)n), o_done = lay.add(16);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, lay.size, &h));
  std::memcpy(h + o_d, desc, (size_t)n * 32);
  uint8_t* hdev = nullptr;
  const bool direct = ctx->window_direct && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  const BowRecord* rec = (const BowRecord*)(h + o_r);
  if (direct) {
    if (!ctx->d_win_ctr) { ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_win_ctr, 64)); ctx->win_ctr_dirty = true; }
    if (ctx->win_ctr_dirty) { ORBX_HIP(ctx, hipMemsetAsync(ctx->d_win_ctr, 0, 64, st)); ctx->win_ctr_dirty = false; }
    volatile unsigned long long* done = (volatile unsigned long long*)(h + o_done);
    __atomic_store_n(done, 0ull, __ATOMIC_RELEASE);
    ctx->win_ctr_dirty = true;
    hipLaunchKernelGGL(k_bow_descend_direct, dim3((n + 3) / 4), dim3(256), 0, st, v->d_nodes, v->d_slot_desc, v->d_slot_node, hdev + o_d, n, v->L,
                       levelsup, (BowRecord*)(hdev + o_r), (unsigned*)ctx->d_win_ctr + 9, (unsigned long long*)(hdev + o_done));
    ORBX_HIP(ctx, hipGetLastError());
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {
          if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
          return set_err(ctx, ORBX_E_DEVICE, "orbx_bow_transform: the pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    ctx->win_ctr_dirty = false;
    for (int i = 0; i < n; i++) { word[i] = rec[i].word; node[i] = rec[i].node; weight[i] = rec[i].weight; }
    return ORBX_OK;
  }
  // copies + stream synchronisation ("window_direct" = 0, or no mapped host memory): same results
  ctx->arena.rewind();
  hipError_t aerr = hipSuccess;
  uint8_t* dd = (uint8_t*)ctx->arena.alloc((size_t)n * 32, &aerr);
  ORBX_HIP(ctx, aerr);
  uint32_t* dw = (uint32_t*)ctx->arena.alloc(4 * (size_t)n, &aerr);
  ORBX_HIP(ctx, aerr);
  uint32_t* dn = (uint32_t*)ctx->arena.alloc(4 * (size_t)n, &aerr);
  ORBX_HIP(ctx, aerr);
  double* dwt = (double*)ctx->arena.alloc(8 * (size_t)n, &aerr);
  ORBX_HIP(ctx, aerr);
  ORBX_HIP(ctx, hipMemcpyAsync(dd, h + o_d, (size_t)n * 32, hipMemcpyHostToDevice, st));
  const int rc = orbx_bow_transform_device(v, dd, n, levelsup, dw, dwt, dn, st);
  if (rc != ORBX_OK) return rc;
  ORBX_HIP(ctx, hipMemcpyAsync(word, dw, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(weight, dwt, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipMemcpyAsync(node, dn, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  return ORBX_OK;
}

// BowVector::addWeight / addIfNotExist + normalize (BowVector.cpp:34-85) and the weighting switch of
// TemplatedVocabulary::transform (TemplatedVocabulary.h:1145-1192): host, ordered map, doubles.
int orbx_bow_finalize(const orbx_voc* v, const uint32_t* word, const double* weight, int n, uint32_t* ids, double* vals,
                      int* n_out) {
  if (!v || n < 0 || !n_out || (n > 0 && (!word || !weight || !ids || !vals))) return ORBX_E_INVALID;
  // the reference accumulates into an ordered map feature by feature (addWeight: `vit->second += v` in feature order); sorting
  // (word, feature index) keys gives the same words in ascending order and, inside a word, the same order of additions
  const bool tf = v->weighting == 0 /*TF_IDF*/ || v->weighting == 1 /*TF*/;
  static thread_local std::vector<uint64_t> keys, tmp;
  keys.clear();
  uint32_t maxw = 0;
  for (int i = 0; i < n; i++)
    if (weight[i] > 0) { keys.push_back(((uint64_t)word[i] << 32) | (uint32_t)i); maxw = std::max(maxw, word[i]); }
  // ascending (word, feature): the keys start out in feature order, so a STABLE sort by word alone does it — LSD radix, 11 bits a pass
  // (two passes for a 10^6-word vocabulary) instead of a comparison sort
  {
    tmp.resize(keys.size());
    uint32_t cnt[2048];
    for (int shift = 0; shift < 32 && (maxw >> shift) != 0; shift += 11) {
      std::memset(cnt, 0, sizeof(cnt));
      for (uint64_t k : keys) cnt[(uint32_t)(k >> (32 + shift)) & 2047u]++;
      uint32_t run = 0;
      for (int b = 0; b < 2048; b++) { const uint32_t c = cnt[b]; cnt[b] = run; run += c; }
      for (uint64_t k : keys) tmp[cnt[(uint32_t)(k >> (32 + shift)) & 2047u]++] = k;
      keys.swap(tmp);
    }
  }
  int k = 0;
  for (size_t a = 0; a < keys.size();) {
    const uint32_t w = (uint32_t)(keys[a] >> 32);
    double acc = weight[(uint32_t)keys[a]];
    size_t b = a + 1;
    for (; b < keys.size() && (uint32_t)(keys[b] >> 32) == w; b++)
      if (tf) acc += weight[(uint32_t)keys[b]];      // IDF / BINARY: addIfNotExist keeps the first value
    ids[k] = w; vals[k] = acc; k++;
    a = b;
  }
  // mustNormalize: L1_NORM->L1, L2_NORM->L2, CHI_SQUARE/KL/BHATTACHARYYA->L1, DOT_PRODUCT->none (ScoringObject.h)
  const bool must = v->scoring != 5;
  const bool l2 = v->scoring == 1;
  if (tf && k > 0 && !must) {
    const double nd = (double)k;
    for (int j = 0; j < k; j++) vals[j] /= nd;
  }
  if (must) {
    double norm = 0.0;
    if (!l2) { for (int j = 0; j < k; j++) norm += std::fabs(vals[j]); }
    else { for (int j = 0; j < k; j++) norm += vals[j] * vals[j]; norm = std::sqrt(norm); }
    if (norm > 0.0) for (int j = 0; j < k; j++) vals[j] /= norm;
  }
  *n_out = k;
  return ORBX_OK;
}

double orbx_bow_score_l1(const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb) {
  int a = 0, b = 0;
  double score = 0;
  while (a < na && b < nb) {
    if (ida[a] == idb[b]) {
      const double vi = va[a], wi = vb[b];
      score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
      ++a; ++b;
    } else if (ida[a] < idb[b]) ++a;
    else ++b;
  }
  return -score / 2.0;
}

int orbx_bow_score_l1_batch(orbx_ctx* ctx, const uint32_t* q_ids, const double* q_vals, int nq, const int32_t* db_ptr,
                            const uint32_t* db_ids, const double* db_vals, int ndb, double* scores) {
  if (!ctx || nq < 0 || ndb < 0 || (ndb > 0 && (!db_ptr || !scores))) return ORBX_E_INVALID;
  if (ndb == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int nnz = db_ptr[ndb];
  DevBuf<uint32_t> dqi, ddi; DevBuf<double> dqv, ddv, dsc; DevBuf<int32_t> dp;
  ORBX_HIP(ctx, dqi.alloc(nq)); ORBX_HIP(ctx, dqv.alloc(nq)); ORBX_HIP(ctx, ddi.alloc(nnz)); ORBX_HIP(ctx, ddv.alloc(nnz));
  ORBX_HIP(ctx, dsc.alloc(ndb)); ORBX_HIP(ctx, dp.alloc(ndb + 1));
  if (nq) { ORBX_HIP(ctx, copy_sync(ctx, dqi.p, q_ids, sizeof(uint32_t) * nq, hipMemcpyHostToDevice));
            ORBX_HIP(ctx, copy_sync(ctx, dqv.p, q_vals, sizeof(double) * nq, hipMemcpyHostToDevice)); }
  if (nnz) { ORBX_HIP(ctx, copy_sync(ctx, ddi.p, db_ids, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice));
             ORBX_HIP(ctx, copy_sync(ctx, ddv.p, db_vals, sizeof(double) * nnz, hipMemcpyHostToDevice)); }
  ORBX_HIP(ctx, copy_sync(ctx, dp.p, db_ptr, sizeof(int32_t) * (ndb + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_bow_score_l1, dim3((ndb + 255) / 256), dim3(256), 0, ctx->stream, dqi.p, dqv.p, nq, dp.p, ddi.p, ddv.p,
                     ndb, dsc.p);
  ORBX_HIP(ctx, hipGetLastError());
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ORBX_HIP(ctx, copy_sync(ctx, scores, dsc.p, sizeof(double) * ndb, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

}  // extern "C"
