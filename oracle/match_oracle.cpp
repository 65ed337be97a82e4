// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header).  **parity unpinned**: the reference holds no
// tests or golden vectors for the matcher / DBoW2 path.
//
// CPU restatement of
//   * ORBmatcher::DescriptorDistance                     src/ORBmatcher.cc:2058-2074 (== FORB::distance, FORB.cpp:77-96)
//   * the candidate-loop pattern of the Search*/Fuse routines (best / second best, strict `<`, and the
//     `<=` variant of SearchForTriangulation)            src/ORBmatcher.cc:96-118, :1010-1080
//   * cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)            as used at src/Frame.cc:1144
//   * DBoW2 vocabulary: loadFromTextFile, transform, BowVector::addWeight/normalize, FeatureVector::addFeature,
//     L1Scoring::score    Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259,1338-1424, BowVector.cpp:34-85,
//                         FeatureVector.cpp:30-45, ScoringObject.cpp:23-68
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

// src/ORBmatcher.cc:2058-2074 — the SWAR population count, 8 x 32 bit
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4);
    std::memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

struct VNode {
  int parent = 0;
  std::vector<int> children;
  uint8_t desc[32] = {0};
  double weight = 0;
  int word_id = 0;
  bool isLeaf() const { return children.empty(); }
};

struct Voc {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<VNode> nodes;
  int nwords = 0;
};

}  // namespace

extern "C" {

int mo_hamming(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// candidate loop with first-minimum-wins (strict <) or last-minimum-wins (<=) update, plus runner-up
void mo_nn_csr(const uint8_t* q, int nq, const uint8_t* t, const int32_t* row_ptr, const int32_t* cand, int last_wins,
               int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist, int32_t* dist_out) {
  for (int i = 0; i < nq; i++) {
    int bestDist = 256, bestDist2 = 256, bestIdx = -1, bestIdx2 = -1;
    for (int c = row_ptr[i]; c < row_ptr[i + 1]; c++) {
      const int idx = cand[c];
      const int dist = descriptor_distance(q + (size_t)i * 32, t + (size_t)idx * 32);
      if (dist_out) dist_out[c] = dist;
      if (!last_wins) {
        // src/ORBmatcher.cc:103-118
        if (dist < bestDist) { bestDist2 = bestDist; bestIdx2 = bestIdx; bestDist = dist; bestIdx = idx; }
        else if (dist < bestDist2) { bestDist2 = dist; bestIdx2 = idx; }
      } else {
        // src/ORBmatcher.cc:1017: `if(dist>bestDist) continue;` then replace -> ties go to the latest candidate
        if (dist <= bestDist) { bestDist2 = bestDist; bestIdx2 = bestIdx; bestDist = dist; bestIdx = idx; }
        else if (dist <= bestDist2) { bestDist2 = dist; bestIdx2 = idx; }
      }
    }
    best_idx[i] = bestIdx; best_dist[i] = bestDist; second_idx[i] = bestIdx2; second_dist[i] = bestDist2;
  }
}

// brute-force 2-NN, ties to the lower train index
void mo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
  for (int i = 0; i < nq; i++) {
    int d1 = 256, d2 = 256, i1 = -1, i2 = -1;
    for (int j = 0; j < nt; j++) {
      const int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; }
      else if (d < d2) { d2 = d; i2 = j; }
    }
    idx[2 * i] = i1; dist[2 * i] = d1; idx[2 * i + 1] = i2; dist[2 * i + 1] = d2;
  }
}

// TemplatedVocabulary::loadFromTextFile, TemplatedVocabulary.h:1338-1424 (blank lines skipped: SURVEY F14)
void* mo_voc_load(const char* path) {
  std::ifstream f(path);
  if (!f.is_open()) return nullptr;
  Voc* v = new Voc();
  std::string s;
  std::getline(f, s);
  {
    std::stringstream ss;
    ss << s;
    ss >> v->k; ss >> v->L; ss >> v->scoring; ss >> v->weighting;
  }
  if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || v->scoring < 0 || v->scoring > 5 || v->weighting < 0 || v->weighting > 3) {
    delete v;
    return nullptr;
  }
  v->nodes.resize(1);
  while (std::getline(f, s)) {
    if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::stringstream ss;
    ss << s;
    int nid = (int)v->nodes.size();
    v->nodes.resize(nid + 1);
    int pid;
    ss >> pid;
    v->nodes[nid].parent = pid;
    v->nodes[pid].children.push_back(nid);
    int nIsLeaf;
    ss >> nIsLeaf;
    for (int i = 0; i < 32; i++) { int b; ss >> b; v->nodes[nid].desc[i] = (uint8_t)b; }
    ss >> v->nodes[nid].weight;
    if (nIsLeaf > 0) v->nodes[nid].word_id = v->nwords++;
  }
  return v;
}
void mo_voc_free(void* h) { delete (Voc*)h; }

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1218-1259
void mo_voc_descend(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node) {
  const Voc& v = *(Voc*)h;
  for (int f = 0; f < n; f++) {
    const uint8_t* feature = desc + (size_t)f * 32;
    const int nid_level = v.L - levelsup;
    unsigned nid = 0;
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const std::vector<int>& nodes = v.nodes[final_id].children;
      final_id = nodes[0];
      double best_d = descriptor_distance(feature, v.nodes[final_id].desc);
      for (size_t c = 1; c < nodes.size(); c++) {
        const int id = nodes[c];
        const double d = descriptor_distance(feature, v.nodes[id].desc);
        if (d < best_d) { best_d = d; final_id = id; }
      }
      if (current_level == nid_level) nid = (unsigned)final_id;
    } while (!v.nodes[final_id].isLeaf());
    word[f] = (uint32_t)v.nodes[final_id].word_id;
    weight[f] = v.nodes[final_id].weight;
    node[f] = nid;
  }
}

// transform(features, BowVector&, FeatureVector&, levelsup) for TF_IDF + L1 (the ORBvoc configuration),
// TemplatedVocabulary.h:1127-1194 + BowVector.cpp:34-85.  Output: ascending ids + values; feature-vector as
// (node id, feature index) pairs in std::map order.
int mo_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* ids, double* vals, uint32_t* fv_node,
                     uint32_t* fv_feat, int* n_fv) {
  const Voc& v = *(Voc*)h;
  std::vector<uint32_t> word(n), node(n);
  std::vector<double> w(n);
  mo_voc_descend(h, desc, n, levelsup, word.data(), w.data(), node.data());
  std::map<unsigned, double> bow;
  std::map<unsigned, std::vector<unsigned>> fv;
  const bool tf = v.weighting == 0 || v.weighting == 1;
  for (int i = 0; i < n; i++) {
    if (w[i] > 0) {
      auto it = bow.lower_bound(word[i]);
      if (it != bow.end() && !(bow.key_comp()(word[i], it->first))) { if (tf) it->second += w[i]; }
      else bow.insert(it, std::make_pair(word[i], w[i]));
      fv[node[i]].push_back((unsigned)i);
    }
  }
  const bool must = v.scoring != 5;
  if (tf && !bow.empty() && !must) {
    const double nd = (double)bow.size();
    for (auto& kv : bow) kv.second /= nd;
  }
  if (must) {
    double norm = 0.0;
    if (v.scoring != 1) { for (auto& kv : bow) norm += std::fabs(kv.second); }
    else { for (auto& kv : bow) norm += kv.second * kv.second; norm = std::sqrt(norm); }
    if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
  }
  int k = 0;
  for (auto& kv : bow) { ids[k] = kv.first; vals[k] = kv.second; k++; }
  int m = 0;
  for (auto& kv : fv) for (unsigned fi : kv.second) { fv_node[m] = kv.first; fv_feat[m] = fi; m++; }
  *n_fv = m;
  return k;
}

// L1Scoring::score, ScoringObject.cpp:23-68 (lower_bound jumps == plain merge advance)
double mo_score_l1(const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb) {
  std::map<unsigned, double> v1, v2;
  for (int i = 0; i < na; i++) v1[ida[i]] = va[i];
  for (int i = 0; i < nb; i++) v2[idb[i]] = vb[i];
  auto v1_it = v1.begin(), v2_it = v2.begin();
  double score = 0;
  while (v1_it != v1.end() && v2_it != v2.end()) {
    const double& vi = v1_it->second;
    const double& wi = v2_it->second;
    if (v1_it->first == v2_it->first) {
      score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
      ++v1_it; ++v2_it;
    } else if (v1_it->first < v2_it->first) {
      v1_it = v1.lower_bound(v2_it->first);
    } else {
      v2_it = v2.lower_bound(v1_it->first);
    }
  }
  score = -score / 2.0;
  return score;
}

}  // extern "C"
