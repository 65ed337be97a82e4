// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header).  PARITY STATUS: pinned.  The reference holds no tests or
// golden vectors for the matcher / DBoW2 path, but its own code is compiled here (oracle/ref_fragments.mk):
// the vendored DBoW2 (oracle/_ref/libref_dbow2.so; tests/test_ref_fragments.py compares the vocabulary / scoring
// functions below with it) and src/ORBmatcher.cc itself (oracle/_ref/ref_matcher_world, against the object model of
// tests/support/ref_world; tests/test_matcher_world.py compares the drop-in matcher — running on the window / nn
// primitives at the end of this file — with it on all 12 routines).  Round 3: src/Frame.cc is compiled too (oracle/_ref/ref_frame_world,
// tests/support/frame_world*): tests/test_frame_pins.py holds the Frame.cc restatements below — ComputeStereoMatches, AssignFeaturesToGrid /
// PosInGrid, GetFeaturesInArea, the UndistortKeyPoints wrapper — to what that compiled file leaves in a Frame, bit for bit; and
// src/KeyFrameDatabase.cc (oracle/_ref/ref_kfdb_world, tests/test_kfdb_world.py).  Still recalled, not pinned: cv::undistortPoints itself.
//
// CPU restatement of
//   * ORBmatcher::DescriptorDistance                     src/ORBmatcher.cc:2058-2074 (== FORB::distance, FORB.cpp:77-96)
//   * the candidate-loop pattern of the Search*/Fuse routines (best / second best, strict `<`, and the
//     `<=` variant of SearchForTriangulation)            src/ORBmatcher.cc:96-118, :1010-1080
//   * cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)            as used at src/Frame.cc:1144
//   * Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea          src/Frame.cc:385-416, :725-735, :657-723
//   * ORBmatcher::SearchForInitialization + ComputeThreeMaxima             src/ORBmatcher.cc:648-763, :2012-2053
//   * Frame::ComputeStereoMatches (row-band Hamming search + 11x11 SAD refinement on the pyramids)   src/Frame.cc:811-981
//   * DBoW2 vocabulary: loadFromTextFile, transform, BowVector::addWeight/normalize, FeatureVector::addFeature,
//     L1Scoring::score    Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259,1338-1424, BowVector.cpp:34-85,
//                         FeatureVector.cpp:30-45, ScoringObject.cpp:23-68
#include <algorithm>
#include <climits>
#include <cassert>
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

// ---------------------------------------------------------------------------------------------------------
// Frame grid (include/Frame.h:44-45: FRAME_GRID_ROWS 48, FRAME_GRID_COLS 64)
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kGridCols = 64, kGridRows = 48;
struct MKeyPt { float x, y, size, angle, response; int32_t octave, class_id; };

struct FrameGrid {
  float minX, minY, invW, invH;
  const MKeyPt* kps;
  int n;
  std::vector<size_t> cell[kGridCols][kGridRows];
  // src/Frame.cc:153-160 (grid element inverses), :385-416 (AssignFeaturesToGrid), :725-735 (PosInGrid)
  FrameGrid(const MKeyPt* k, int n_, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY) : kps(k), n(n_) {
    minX = mnMinX; minY = mnMinY;
    invW = static_cast<float>(kGridCols) / static_cast<float>(mnMaxX - mnMinX);
    invH = static_cast<float>(kGridRows) / static_cast<float>(mnMaxY - mnMinY);
    for (int i = 0; i < n; i++) {
      const int posX = (int)std::round((kps[i].x - minX) * invW);
      const int posY = (int)std::round((kps[i].y - minY) * invH);
      if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) continue;
      cell[posX][posY].push_back((size_t)i);
    }
  }
  // src/Frame.cc:657-723
  std::vector<size_t> area(const float& x, const float& y, const float& r, const int minLevel, const int maxLevel) const {
    std::vector<size_t> out;
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - factorX) * invW));
    if (nMinCellX >= kGridCols) return out;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - minX + factorX) * invW));
    if (nMaxCellX < 0) return out;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - factorY) * invH));
    if (nMinCellY >= kGridRows) return out;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - minY + factorY) * invH));
    if (nMaxCellY < 0) return out;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (size_t idx : cell[ix][iy]) {
          const MKeyPt& kp = kps[idx];
          if (bCheckLevels) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          const float distx = kp.x - x, disty = kp.y - y;
          if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) out.push_back(idx);
        }
    return out;
  }
};

// src/ORBmatcher.cc:2012-2053
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
}  // namespace

extern "C" {

// Frame::GetFeaturesInArea for a list of queries -> CSR (row_ptr [nq+1], cand [cap]); returns nnz (or -1 on overflow)
int mo_features_in_area(const void* kps, int n, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, const float* qx,
                        const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, int nq, int32_t* row_ptr,
                        int32_t* cand, int cap) {
  FrameGrid g((const MKeyPt*)kps, n, mnMinX, mnMinY, mnMaxX, mnMaxY);
  int nnz = 0;
  row_ptr[0] = 0;
  for (int q = 0; q < nq; q++) {
    for (size_t i : g.area(qx[q], qy[q], qr[q], qmin[q], qmax[q])) {
      if (nnz >= cap) return -1;
      cand[nnz++] = (int32_t)i;
    }
    row_ptr[q + 1] = nnz;
  }
  return nnz;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:648-763.
// prev_xy: [n1][2] in/out (vbPrevMatched).  Returns nmatches.
int mo_search_for_initialization(const void* kps1_, const uint8_t* desc1, int n1, const void* kps2_, const uint8_t* desc2, int n2,
                                 float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, float* prev_xy, int windowSize,
                                 float mfNNratio, int mbCheckOrientation, int32_t* vnMatches12) {
  const MKeyPt* kps1 = (const MKeyPt*)kps1_;
  const MKeyPt* kps2 = (const MKeyPt*)kps2_;
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  FrameGrid F2(kps2, n2, mnMinX, mnMinY, mnMaxX, mnMaxY);
  int nmatches = 0;
  for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
  std::vector<int> rotHist[30];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
  for (int i1 = 0; i1 < n1; i1++) {
    const MKeyPt kp1 = kps1[i1];
    const int level1 = kp1.octave;
    if (level1 > 0) continue;
    const std::vector<size_t> vIndices2 = F2.area(prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      const int dist = descriptor_distance(d1, desc2 + i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = kps1[i1].angle - kps2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          assert(bin >= 0 && bin < HISTO_LENGTH);
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (vnMatches12[i1] >= 0) { prev_xy[2 * i1] = kps2[vnMatches12[i1]].x; prev_xy[2 * i1 + 1] = kps2[vnMatches12[i1]].y; }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints),
// src/ORBmatcher.cc:43-141, for F.Nleft == -1 (the second block, :143-210, only runs for two-camera rigs).
// The objects are flattened: kp_obs[i] = F.mvpMapPoints[i] ? Observations() : -1 (in/out), mp_* = the MapPoint fields the
// routine reads; mp_skip[i] folds `!mbTrackInView`, `bFarPoints && mTrackDepth > thFarPoints` and `isBad()` (:53-60).
// kp_match[i] = index of the map point written to F.mvpMapPoints[i] by this call (-1: untouched).  Returns nmatches.
int mo_search_by_projection(const void* kps_, const uint8_t* desc, const float* mvuRight, int32_t* kp_obs, int n, float mnMinX,
                            float mnMinY, float mnMaxX, float mnMaxY, const float* mvScaleFactors, const uint8_t* mp_skip,
                            const float* mTrackProjX, const float* mTrackProjY, const float* mTrackProjXR, const float* mTrackViewCos,
                            const int32_t* mnTrackScaleLevel, const uint8_t* mp_desc, const int32_t* mp_obs, int nmp, float th,
                            float mfNNratio, int32_t* kp_match) {
  const MKeyPt* kps = (const MKeyPt*)kps_;
  const int TH_HIGH = 100;
  FrameGrid F(kps, n, mnMinX, mnMinY, mnMaxX, mnMaxY);
  for (int i = 0; i < n; i++) kp_match[i] = -1;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nmp; iMP++) {
    if (mp_skip[iMP]) continue;
    const int& nPredictedLevel = mnTrackScaleLevel[iMP];
    float r = (mTrackViewCos[iMP] > 0.998) ? 2.5 : 4.0;   // RadiusByViewingCos, :214-220
    if (bFactor) r *= th;
    const std::vector<size_t> vIndices =
        F.area(mTrackProjX[iMP], mTrackProjY[iMP], r * mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = mp_desc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t idx : vIndices) {
      if (kp_obs[idx] > 0) continue;   // F.mvpMapPoints[idx] && Observations() > 0
      if (mvuRight && mvuRight[idx] > 0) {
        const float er = fabs(mTrackProjXR[iMP] - mvuRight[idx]);
        if (er > r * mvScaleFactors[nPredictedLevel]) continue;
      }
      const int dist = descriptor_distance(MPdescriptor, desc + idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = (int)idx;
      } else if (dist < bestDist2) {
        bestLevel2 = kps[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
        kp_match[bestIdx] = iMP;           // F.mvpMapPoints[bestIdx] = pMP
        kp_obs[bestIdx] = mp_obs[iMP];
        nmatches++;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), src/ORBmatcher.cc:1676-1885,
// for CurrentFrame.Nleft == -1, from the point where the last frame's map point i has been projected (:1705-1718 are the
// caller's Sophus / camera-model arithmetic): lp_valid[i] folds `pMP && !mvbOutlier[i] && invzc >= 0 && uv inside the
// image bounds`; lp_u / lp_v = uv, lp_invz = invzc, lp_octave = nLastOctave, lp_angle = the last frame's keypoint angle.
// direction: 0 = neither, 1 = bForward, 2 = bBackward (:1691-1692).  kp_obs as in mo_search_by_projection.
// kp_match[i2]: -1 untouched, >= 0 index i of the last-frame point now in CurrentFrame.mvpMapPoints[i2], -2 set to NULL by
// the rotation filter.  Returns nmatches.
int mo_search_by_projection_last(const void* kps_, const uint8_t* desc, const float* mvuRight, int32_t* kp_obs, int n, float mnMinX,
                                 float mnMinY, float mnMaxX, float mnMaxY, const float* mvScaleFactors, float mbf,
                                 const uint8_t* lp_valid, const float* lp_u, const float* lp_v, const float* lp_invz,
                                 const int32_t* lp_octave, const float* lp_angle, const uint8_t* lp_desc, const int32_t* lp_obs, int nlast,
                                 float th, int direction, int mbCheckOrientation, int32_t* kp_match) {
  const MKeyPt* kps = (const MKeyPt*)kps_;
  const int TH_HIGH = 100, HISTO_LENGTH = 30;
  FrameGrid Cur(kps, n, mnMinX, mnMinY, mnMaxX, mnMaxY);
  for (int i = 0; i < n; i++) kp_match[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[30];
  const float factor = 1.0f / HISTO_LENGTH;
  const bool bForward = direction == 1, bBackward = direction == 2;
  for (int i = 0; i < nlast; i++) {
    if (!lp_valid[i]) continue;
    const float u = lp_u[i], v = lp_v[i], invzc = lp_invz[i];
    const int nLastOctave = lp_octave[i];
    float radius = th * mvScaleFactors[nLastOctave];
    std::vector<size_t> vIndices2;
    if (bForward) vIndices2 = Cur.area(u, v, radius, nLastOctave, -1);
    else if (bBackward) vIndices2 = Cur.area(u, v, radius, 0, nLastOctave);
    else vIndices2 = Cur.area(u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = lp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t i2 : vIndices2) {
      if (kp_obs[i2] > 0) continue;
      if (mvuRight && mvuRight[i2] > 0) {
        const float ur = u - mbf * invzc;
        const float er = fabs(ur - mvuRight[i2]);
        if (er > radius) continue;
      }
      const int dist = descriptor_distance(dMP, desc + i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= TH_HIGH) {
      kp_match[bestIdx2] = i;
      kp_obs[bestIdx2] = lp_obs[i];
      nmatches++;
      if (mbCheckOrientation) {
        float rot = lp_angle[i] - kps[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        assert(bin >= 0 && bin < HISTO_LENGTH);
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          kp_match[rotHist[i][j]] = -2;   // CurrentFrame.mvpMapPoints[...] = NULL
          kp_obs[rotHist[i][j]] = -1;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

}  // extern "C"

extern "C" {

// Frame::ComputeStereoMatches, src/Frame.cc:811-981.  Pyramid level l of the left / right extractor:
// pyrL[l] / pyrR[l] point to w[l] x h[l] images with `pitch[l]` bytes per row (mvImagePyramid[l]).
// Outputs mvuRight / mvDepth (n entries, -1 where unmatched).  Returns the number of matches kept.
int mo_stereo_matches(const void* kpsL_, const uint8_t* descL, int N, const void* kpsR_, const uint8_t* descR, int Nr,
                      const uint8_t* const* pyrL, const uint8_t* const* pyrR, const int32_t* w, const int32_t* h,
                      const int32_t* pitch, const float* mvScaleFactors, const float* mvInvScaleFactors, float mb, float mbf,
                      float* mvuRight, float* mvDepth) {
  const MKeyPt* mvKeys = (const MKeyPt*)kpsL_;
  const MKeyPt* mvKeysRight = (const MKeyPt*)kpsR_;
  const int TH_HIGH = 100, TH_LOW = 50;
  for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = h[0];
  std::vector<std::vector<size_t>> vRowIndices(nRows, std::vector<size_t>());
  for (int iR = 0; iR < Nr; iR++) {
    const MKeyPt& kp = mvKeysRight[iR];
    const float& kpY = kp.y;
    const float r = 2.0f * mvScaleFactors[mvKeysRight[iR].octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // the reference indexes unchecked (keypoints sit >= 16 px inside)
  }
  const float minZ = mb;
  const float minD = 0;
  const float maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < N; iL++) {
    const MKeyPt& kpL = mvKeys[iL];
    const int& levelL = kpL.octave;
    const float& vL = kpL.y;
    const float& uL = kpL.x;
    const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t* dL = descL + (size_t)iL * 32;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const MKeyPt& kpR = mvKeysRight[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float& uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = descriptor_distance(dL, descR + iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = mvKeysRight[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int wnd = 5;
      const int lvl = kpL.octave;
      const uint8_t* IL = pyrL[lvl];
      const uint8_t* IRi = pyrR[lvl];
      int bestDistS = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      std::vector<float> vDists(2 * L + 1);
      const float iniu = scaleduR0 + L - wnd;
      const float endu = scaleduR0 + L + wnd + 1;
      if (iniu < 0 || endu >= w[lvl]) continue;
      const int r0 = (int)(scaledvL - wnd), cL0 = (int)(scaleduL - wnd);
      for (int incR = -L; incR <= +L; incR++) {
        const int cR0 = (int)(scaleduR0 + incR - wnd);
        double acc = 0;  // cv::norm(IL, IR, NORM_L1) on CV_8U: sum of absolute differences
        for (int rr = 0; rr < 2 * wnd + 1; rr++)
          for (int cc = 0; cc < 2 * wnd + 1; cc++)
            acc += std::abs((int)IL[(size_t)(r0 + rr) * pitch[lvl] + cL0 + cc] - (int)IRi[(size_t)(r0 + rr) * pitch[lvl] + cR0 + cc]);
        float dist = (float)acc;
        if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return 0;   // the reference reads vDistIdx[0] of an empty vector here (undefined); nothing to filter
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    mvuRight[vDistIdx[i].second] = -1;
    mvDepth[vDistIdx[i].second] = -1;
    kept--;
  }
  return kept;
}

}  // extern "C"

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:223-425, for
// F.Nleft == -1 and a single-camera keyframe (pKF->mpCamera2 == nullptr).  Flattened state: kf_valid[i] = vpMapPointsKF[i] &&
// !isBad(); feature vectors as (node, index) pair lists in map iteration order (nodes ascending, indices in insertion order).
// match_kf[iF] = index of the keyframe feature whose map point ends up in vpMapPointMatches[iF], -1 for NULL.  Returns nmatches.
#include <map>
extern "C" int mo_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, int nkf,
                                const uint32_t* kf_fv_node, const uint32_t* kf_fv_idx, int n_kf_fv, const uint8_t* f_desc,
                                const float* f_angle, int nf, const uint32_t* f_fv_node, const uint32_t* f_fv_idx, int n_f_fv,
                                float mfNNratio, int mbCheckOrientation, int32_t* match_kf) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  std::map<unsigned, std::vector<unsigned>> vFeatVecKF, FFeatVec;   // DBoW2::FeatureVector
  for (int i = 0; i < n_kf_fv; i++) vFeatVecKF[kf_fv_node[i]].push_back(kf_fv_idx[i]);
  for (int i = 0; i < n_f_fv; i++) FFeatVec[f_fv_node[i]].push_back(f_fv_idx[i]);
  for (int i = 0; i < nf; i++) match_kf[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[30];
  const float factor = 1.0f / HISTO_LENGTH;
  auto KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
  auto Fit = FFeatVec.begin(), Fend = FFeatVec.end();
  while (KFit != KFend && Fit != Fend) {
    if (KFit->first == Fit->first) {
      const std::vector<unsigned> vIndicesKF = KFit->second;
      const std::vector<unsigned> vIndicesF = Fit->second;
      for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
        const unsigned realIdxKF = vIndicesKF[iKF];
        if (!kf_valid[realIdxKF]) continue;   // !pMP || pMP->isBad()
        const uint8_t* dKF = kf_desc + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
          const unsigned realIdxF = vIndicesF[iF];
          if (match_kf[realIdxF] >= 0) continue;   // vpMapPointMatches[realIdxF]
          const int dist = descriptor_distance(dKF, f_desc + (size_t)realIdxF * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            match_kf[bestIdxF] = (int)realIdxKF;
            if (mbCheckOrientation) {
              float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              assert(bin >= 0 && bin < HISTO_LENGTH);
              rotHist[bin].push_back(bestIdxF);
            }
            nmatches++;
          }
        }
      }
      KFit++;
      Fit++;
    } else if (KFit->first < Fit->first) {
      KFit = vFeatVecKF.lower_bound(Fit->first);
    } else {
      Fit = FFeatVec.lower_bound(KFit->first);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { match_kf[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------
// KeyFrameDatabase (src/KeyFrameDatabase.cc), the part every Detect* routine starts with.
// ---------------------------------------------------------------------------------------------------------
#include <list>
#include <map>
#include <set>
namespace {
struct OKeyFrame {
  long mnId;
  std::vector<uint32_t> ids;      // mBowVec (ordered map: ascending word id)
  std::vector<double> vals;
  long mnQuery = -1;              // mnRelocQuery / mnLoopQuery / mnPlaceRecognitionQuery
  int mnWords = 0;                // mnRelocWords / ...
};
struct OKfDb {
  std::map<uint32_t, std::list<OKeyFrame*>> mvInvertedFile;   // word -> keyframes in insertion order
  std::map<long, OKeyFrame*> kfs;
  long next_query = 1;
};
}  // namespace

extern "C" {

void* mo_kfdb_new() { return new OKfDb(); }
void mo_kfdb_free(void* h) {
  OKfDb* db = (OKfDb*)h;
  for (auto& kv : db->kfs) delete kv.second;
  delete db;
}
// KeyFrameDatabase::add, :39-45
void mo_kfdb_add(void* h, long kf_id, const uint32_t* ids, const double* vals, int n) {
  OKfDb* db = (OKfDb*)h;
  OKeyFrame* pKF = new OKeyFrame();
  pKF->mnId = kf_id; pKF->ids.assign(ids, ids + n); pKF->vals.assign(vals, vals + n);
  db->kfs[kf_id] = pKF;
  for (int i = 0; i < n; i++) db->mvInvertedFile[ids[i]].push_back(pKF);
}
// KeyFrameDatabase::erase, :47-66
void mo_kfdb_erase(void* h, long kf_id) {
  OKfDb* db = (OKfDb*)h;
  auto it = db->kfs.find(kf_id);
  if (it == db->kfs.end()) return;
  OKeyFrame* pKF = it->second;
  for (uint32_t w : pKF->ids) {
    std::list<OKeyFrame*>& lKFs = db->mvInvertedFile[w];
    for (auto lit = lKFs.begin(), lend = lKFs.end(); lit != lend; lit++)
      if (pKF == *lit) { lKFs.erase(lit); break; }
  }
  db->kfs.erase(it);
  delete pKF;
}
// The common opening of DetectLoopCandidates (:100-165), DetectBestCandidates (:468-535), DetectNBestCandidates (:604-665)
// and DetectRelocalizationCandidates (:733-790): share-a-word list, maxCommonWords, minCommonWords, scores.
// exclude = spConnectedKeyFrames (and, for the loop / merge split, the keyframes of the other map).  Returns the list length.
int mo_kfdb_query(void* h, const uint32_t* q_ids, const double* q_vals, int nq, const long* exclude, int n_exclude, int nMinWords,
                  long* out_kf, int32_t* out_words, double* out_score, int* maxCommon, int* minCommon) {
  OKfDb* db = (OKfDb*)h;
  const long query_id = db->next_query++;
  std::set<long> spConnected(exclude, exclude + n_exclude);
  std::list<OKeyFrame*> lKFsSharingWords;
  for (int i = 0; i < nq; i++) {
    auto f = db->mvInvertedFile.find(q_ids[i]);
    if (f == db->mvInvertedFile.end()) continue;
    std::list<OKeyFrame*>& lKFs = f->second;
    for (auto lit = lKFs.begin(), lend = lKFs.end(); lit != lend; lit++) {
      OKeyFrame* pKFi = *lit;
      if (pKFi->mnQuery != query_id) {
        pKFi->mnWords = 0;
        if (!spConnected.count(pKFi->mnId)) {
          pKFi->mnQuery = query_id;
          lKFsSharingWords.push_back(pKFi);
        }
      }
      pKFi->mnWords++;
    }
  }
  *maxCommon = 0; *minCommon = 0;
  if (lKFsSharingWords.empty()) return 0;
  int maxCommonWords = 0;
  for (OKeyFrame* p : lKFsSharingWords)
    if (p->mnWords > maxCommonWords) maxCommonWords = p->mnWords;
  int minCommonWords = maxCommonWords * 0.8f;
  if (minCommonWords < nMinWords) minCommonWords = nMinWords;   // :514-517 (nMinWords = 0 in the other routines)
  int k = 0;
  for (OKeyFrame* pKFi : lKFsSharingWords) {
    out_kf[k] = pKFi->mnId;
    out_words[k] = pKFi->mnWords;
    out_score[k] = -1.0;
    if (pKFi->mnWords > minCommonWords)
      out_score[k] = mo_score_l1(q_ids, q_vals, nq, pKFi->ids.data(), pKFi->vals.data(), (int)pKFi->ids.size());
    k++;
  }
  *maxCommon = maxCommonWords; *minCommon = minCommonWords;
  return k;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Generic window primitives over a caller-held grid (the device entry points orbx_window_search_grid /
// orbx_window_nearest compute the same): Frame::GetFeaturesInArea (src/Frame.cc:657-723) == KeyFrame::GetFeaturesInArea
// (src/KeyFrame.cc:704-748, no level filter there: pass -1 / -1) for a list of queries, every candidate's
// DescriptorDistance, and the running best / second of the candidate loops (src/ORBmatcher.cc:96-118).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct MoGrid {   // == orbx_grid (include/orbx.h)
  float min_x, min_y, inv_w, inv_h;
  const int32_t* cell_start;   // [64*48 + 1] over cells in mGrid[ix][iy] order (ix major), or NULL: assign here
  const int32_t* cell_idx;
};

struct GridCells {
  std::vector<int32_t> start, idx;
  float minX, minY, invW, invH;
  const MKeyPt* kps;
  GridCells(const MKeyPt* k, int n, const MoGrid& g) : minX(g.min_x), minY(g.min_y), invW(g.inv_w), invH(g.inv_h), kps(k) {
    const int ncell = kGridCols * kGridRows;
    if (g.cell_start) {
      start.assign(g.cell_start, g.cell_start + ncell + 1);
      idx.assign(g.cell_idx, g.cell_idx + start[ncell]);
      return;
    }
    // Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:385-416, :725-735)
    std::vector<std::vector<int32_t> > cell(ncell);
    for (int i = 0; i < n; i++) {
      const int posX = (int)std::round((kps[i].x - minX) * invW);
      const int posY = (int)std::round((kps[i].y - minY) * invH);
      if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) continue;
      cell[posX * kGridRows + posY].push_back(i);
    }
    start.assign(1, 0);
    for (int c = 0; c < ncell; c++) { idx.insert(idx.end(), cell[c].begin(), cell[c].end()); start.push_back((int32_t)idx.size()); }
  }
  template <class Fn>
  void area(float x, float y, float r, int minLevel, int maxLevel, Fn&& fn) const {
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - factorX) * invW));
    if (nMinCellX >= kGridCols) return;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - minX + factorX) * invW));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - factorY) * invH));
    if (nMinCellY >= kGridRows) return;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - minY + factorY) * invH));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (int j = start[ix * kGridRows + iy]; j < start[ix * kGridRows + iy + 1]; j++) {
          const MKeyPt& kp = kps[idx[j]];
          if (bCheckLevels) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          const float distx = kp.x - x, disty = kp.y - y;
          if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) fn(idx[j]);
        }
  }
};
}  // namespace

extern "C" {

// CSR of window candidates + distances + best / second (first minimum wins).  kp_skip / kp_uright + q_xr: the optional gates
// of orbx_window_search (src/ORBmatcher.cc:81-93).  Returns nnz, or -1 when cand_cap is too small (row_ptr is complete then).
int mo_window_search_grid(const void* kps_, const uint8_t* desc, int n, const void* grid, const uint8_t* kp_skip, const float* kp_uright,
                          const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax,
                          const uint8_t* q_desc, const float* q_xr, int nq, int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap,
                          int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  GridCells G((const MKeyPt*)kps_, n, *(const MoGrid*)grid);
  int nnz = 0;
  bool overflow = false;
  row_ptr[0] = 0;
  for (int q = 0; q < nq; q++) {
    int bd = 256, bd2 = 256, bi = -1, bi2 = -1;
    G.area(qx[q], qy[q], qr[q], qmin[q], qmax[q], [&](int idx) {
      if (kp_skip && kp_skip[idx]) return;
      if (kp_uright && q_xr && kp_uright[idx] > 0.f && std::fabs(q_xr[q] - kp_uright[idx]) > qr[q]) return;
      const int d = descriptor_distance(q_desc + (size_t)q * 32, desc + (size_t)idx * 32);
      if (nnz < cand_cap) { if (cand) cand[nnz] = idx; if (dist) dist[nnz] = d; } else overflow = true;
      nnz++;
      if (d < bd) { bd2 = bd; bi2 = bi; bd = d; bi = idx; }
      else if (d < bd2) { bd2 = d; bi2 = idx; }
    });
    row_ptr[q + 1] = nnz;
    if (best_idx) best_idx[q] = bi;
    if (best_dist) best_dist[q] = bd;
    if (second_idx) second_idx[q] = bi2;
    if (second_dist) second_dist[q] = bd2;
  }
  return (overflow && (cand || dist)) ? -1 : nnz;
}

// arg-min only, with the optional reprojection gate of ORBmatcher::Fuse (src/ORBmatcher.cc:1269-1296): a candidate with a
// right coordinate (kp_uright >= 0) must satisfy (ex^2 + ey^2 + er^2) * invSigma2[level] <= 7.8, a monocular one
// (ex^2 + ey^2) * invSigma2[level] <= 5.99; float products, comparison in double as in the reference.
void mo_window_nearest(const void* kps_, const uint8_t* desc, int n, const void* grid, const float* kp_uright, const float* inv_level_sigma2,
                       const float* qx, const float* qy, const float* qr, const int32_t* qmin, const int32_t* qmax, const float* q_ur,
                       const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist) {
  const MKeyPt* kps = (const MKeyPt*)kps_;
  GridCells G(kps, n, *(const MoGrid*)grid);
  for (int q = 0; q < nq; q++) {
    int bd = 256, bi = -1;
    G.area(qx[q], qy[q], qr[q], qmin[q], qmax[q], [&](int idx) {
      if (inv_level_sigma2) {
        const MKeyPt& kp = kps[idx];
        const int kpLevel = kp.octave;
        if (kp_uright[idx] >= 0) {
          const float ex = qx[q] - kp.x, ey = qy[q] - kp.y, er = q_ur[q] - kp_uright[idx];
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * inv_level_sigma2[kpLevel] > 7.8) return;
        } else {
          const float ex = qx[q] - kp.x, ey = qy[q] - kp.y;
          const float e2 = ex * ex + ey * ey;
          if (e2 * inv_level_sigma2[kpLevel] > 5.99) return;
        }
      }
      const int d = descriptor_distance(q_desc + (size_t)q * 32, desc + (size_t)idx * 32);
      if (d < bd) { bd = d; bi = idx; }
    });
    best_idx[q] = bi; best_dist[q] = bd;
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints (src/Frame.cc:747-780): mvKeysUn = mvKeys when mDistCoef[0] == 0, else
// cv::undistortPoints(points, points, K, distCoef, cv::Mat(), K) on the keypoint coordinates.
// **parity unpinned**: the loop below restates OpenCV 4.x's cvUndistortPointsInternal from recollection (calib3d /
// imgproc undistort.dispatch.cpp) — K and the coefficients widened from CV_32F to double, the default termination
// criteria of the 6-argument overload (MAX_ITER, 5 iterations), no tilt model (k[12] = k[13] = 0), R = I, P = K:
//   x = (u - cx) / fx via ifx = 1./fx ... 5 fixed-point iterations of the radial (k1 k2 k3) + tangential (p1 p2) model
//   ... (xx, yy, ww) = K * (x, y, 1), result narrowed to float.
// ---------------------------------------------------------------------------------------------------------
extern "C" void mo_undistort_points(const float* xy_in, int n, float fx_, float fy_, float cx_, float cy_, const float* dist, int ndist,
                                    float* xy_out) {
  double k[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < ndist && i < 14; i++) k[i] = (double)dist[i];   // k1 k2 p1 p2 [k3 ...]
  const double fx = fx_, fy = fy_, cx = cx_, cy = cy_;
  const double ifx = 1. / fx, ify = 1. / fy;
  const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};      // P * R with R = I, P = K
  for (int i = 0; i < n; i++) {
    double x = xy_in[2 * i], y = xy_in[2 * i + 1];
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    xy_out[2 * i] = (float)(xx * ww);
    xy_out[2 * i + 1] = (float)(yy * ww);
  }
}
