/* orbx adapter — drop-in for the reference's include/ORBVocabulary.h (:27-29,
 * `typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary`) and for the parts of
 * DBoW2's BowVector.h / FeatureVector.h that Frame / KeyFrame / KeyFrameDatabase / ORBmatcher see:
 *   DBoW2::BowVector      = std::map<WordId, WordValue>           (Thirdparty/DBoW2/DBoW2/BowVector.h)
 *   DBoW2::FeatureVector  = std::map<NodeId, std::vector<unsigned>> (Thirdparty/DBoW2/DBoW2/FeatureVector.h)
 *   ORBVocabulary::loadFromTextFile / transform(features, bow, fv, levelsup) / score(a, b) / size / empty
 *     (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424, :1127-1194, :162)
 * The tree descent of every descriptor runs on the GPU (k_bow_descend); the ordered-map accumulation and the L1
 * normalisation stay on the host in ascending-id order so the doubles are bit-identical to the reference's.
 * Known, documented divergence: blank lines in the vocabulary file are skipped (the reference manufactures a phantom
 * node with uninitialised bytes from a trailing newline).
 */
#ifndef ORBVOCABULARY_H
#define ORBVOCABULARY_H

#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"
#include "orbx_cv_compat.h"

// Same include guards as the reference's Thirdparty/DBoW2/DBoW2/BowVector.h / FeatureVector.h: where those (or a stand-in
// for them) were included first, their classes are the ones used.
#ifndef __D_T_BOW_VECTOR__
#define __D_T_BOW_VECTOR__
namespace DBoW2 {

typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;

class BowVector : public std::map<WordId, WordValue> {
 public:
  // BowVector.cpp:34-46
  void addWeight(WordId id, WordValue v) {
    iterator vit = this->lower_bound(id);
    if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
    else this->insert(vit, value_type(id, v));
  }
};

}  // namespace DBoW2
#endif
#ifndef __D_T_FEATURE_VECTOR__
#define __D_T_FEATURE_VECTOR__
namespace DBoW2 {

class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {
 public:
  // FeatureVector.cpp:30-45
  void addFeature(NodeId id, unsigned int i_feature) {
    iterator vit = this->lower_bound(id);
    if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
    else {
      vit = this->insert(vit, value_type(id, std::vector<unsigned int>()));
      vit->second.push_back(i_feature);
    }
  }
};

}  // namespace DBoW2
#endif

namespace ORB_SLAM3 {

class ORBVocabulary {
 public:
  explicit ORBVocabulary(int device_id = -1) {
    if (orbx_create(&ctx_, 1, 1.2f, 1, 20, 7, device_id) != ORBX_OK)
      throw std::runtime_error("ORBVocabulary: no MI355X / HIP device (there is no CPU fallback)");
  }
  ~ORBVocabulary() { orbx_voc_destroy(voc_); orbx_destroy(ctx_); }
  ORBVocabulary(const ORBVocabulary&) = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;

  bool loadFromTextFile(const std::string& filename) {
    orbx_voc_destroy(voc_);
    voc_ = nullptr;
    return orbx_voc_load_text(ctx_, filename.c_str(), &voc_) == ORBX_OK;
  }
  // TemplatedVocabulary.h:1428-1449
  void saveToTextFile(const std::string& filename) const {
    if (!voc_ || orbx_voc_save_text(voc_, filename.c_str()) != ORBX_OK) throw std::runtime_error("ORBVocabulary::saveToTextFile failed");
  }
  // exact binary cache (orbx format; loads at file-read speed instead of parsing 1.08 M text lines)
  bool loadFromBinaryFile(const std::string& filename) {
    orbx_voc_destroy(voc_);
    voc_ = nullptr;
    return orbx_voc_load_binary(ctx_, filename.c_str(), &voc_) == ORBX_OK;
  }
  void saveToBinaryFile(const std::string& filename) const {
    if (!voc_ || orbx_voc_save_binary(voc_, filename.c_str()) != ORBX_OK) throw std::runtime_error("ORBVocabulary::saveToBinaryFile failed");
  }
  unsigned int size() const { int w = 0; if (voc_) orbx_voc_info(voc_, nullptr, nullptr, nullptr, &w); return (unsigned)w; }
  bool empty() const { return size() == 0; }

  // transform(features, BowVector&, FeatureVector&, levelsup): TemplatedVocabulary.h:1127-1194
  void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    const int n = (int)features.size();
    if (!voc_ || n == 0) return;
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t)i * 32], features[i].ptr<unsigned char>(), 32);
    std::vector<uint32_t> word(n), node(n), ids(n);
    std::vector<double> weight(n), vals(n);
    {  // Tracking, LocalMapping and LoopClosing all call transform on the one vocabulary: one caller at a time on its context
      std::lock_guard<std::mutex> lock(mu_);
      if (orbx_bow_transform(voc_, desc.data(), n, levelsup, word.data(), weight.data(), node.data()) != ORBX_OK)
        throw std::runtime_error(std::string("ORBVocabulary::transform: ") + orbx_last_error(ctx_));
    }
    int nnz = 0;
    orbx_bow_finalize(voc_, word.data(), weight.data(), n, ids.data(), vals.data(), &nnz);
    for (int k = 0; k < nnz; k++) v.insert(v.end(), DBoW2::BowVector::value_type(ids[k], vals[k]));
    for (int i = 0; i < n; i++)
      if (weight[i] > 0) fv.addFeature(node[i], (unsigned)i);
  }

  // score(a, b): L1Scoring::score, ScoringObject.cpp:23-68 (ORBvoc.txt is an L1-norm vocabulary)
  double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
    std::vector<uint32_t> ia, ib;
    std::vector<double> va, vb;
    for (const auto& kv : a) { ia.push_back(kv.first); va.push_back(kv.second); }
    for (const auto& kv : b) { ib.push_back(kv.first); vb.push_back(kv.second); }
    return orbx_bow_score_l1(ia.data(), va.data(), (int)ia.size(), ib.data(), vb.data(), (int)ib.size());
  }

  orbx_voc* Handle() { return voc_; }
  orbx_ctx* Context() { return ctx_; }

 private:
  orbx_ctx* ctx_ = nullptr;
  mutable std::mutex mu_;
  orbx_voc* voc_ = nullptr;
};

}  // namespace ORB_SLAM3

#endif  // ORBVOCABULARY_H
