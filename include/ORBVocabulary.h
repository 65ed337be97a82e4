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

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"
#include "orbx_cv_compat.h"

// ---- DBoW2::BowVector / DBoW2::FeatureVector ---------------------------------------------------------------------------------
// Inside the reference's tree (its root on the include path, as its CMakeLists.txt:83 has it) the reference's OWN two headers
// are used, so every translation unit — whichever of KeyFrame.h:24-25 / Frame.h:25-26 / this header it reaches first — sees one
// definition, Boost serialize() members included (BowVector.h:62-67, FeatureVector.h:27-32; archived by KeyFrame.h:130-131,
// instantiated from System.cc:1464-1468).  Their out-of-line members come from Thirdparty/DBoW2/lib/libDBoW2.so, which the
// reference links anyway (CMakeLists.txt:126).  Define ORBX_OWN_DBOW2_TYPES to force the self-contained classes below.
#if !defined(ORBX_OWN_DBOW2_TYPES) && defined(__has_include)
#if __has_include("Thirdparty/DBoW2/DBoW2/BowVector.h") && __has_include("Thirdparty/DBoW2/DBoW2/FeatureVector.h") && \
    __has_include(<boost/serialization/serialization.hpp>)
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#endif
#endif

// Stand-alone builds (no reference tree): the same two classes, header-only, member for member — with the same include guards,
// so that a later include of the reference's headers is a no-op, and with serialize() whenever Boost.Serialization exists.
#if defined(__has_include)
#if __has_include(<boost/serialization/serialization.hpp>) && __has_include(<boost/serialization/map.hpp>)
#include <boost/serialization/serialization.hpp>
#include <boost/serialization/map.hpp>
#define ORBX_HAVE_BOOST_SERIALIZATION 1
#endif
#endif

#ifndef __D_T_BOW_VECTOR__
#define __D_T_BOW_VECTOR__
#include <cmath>
#include <fstream>
#include <iostream>
namespace DBoW2 {

typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;

enum LNorm { L1, L2 };                                                                  // BowVector.h:33-37
enum WeightingType { TF_IDF, TF, IDF, BINARY };                                        // :40-46
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };     // :49-57

class BowVector : public std::map<WordId, WordValue> {
#ifdef ORBX_HAVE_BOOST_SERIALIZATION
  friend class boost::serialization::access;
  template <class Archive>
  void serialize(Archive& ar, const int /*version*/) {   // BowVector.h:62-67: the map base, nothing else
    ar& boost::serialization::base_object<std::map<WordId, WordValue> >(*this);
  }
#endif

 public:
  BowVector() {}
  ~BowVector() {}
  // BowVector.cpp:34-46
  void addWeight(WordId id, WordValue v) {
    iterator vit = this->lower_bound(id);
    if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
    else this->insert(vit, value_type(id, v));
  }
  // BowVector.cpp:50-58
  void addIfNotExist(WordId id, WordValue v) {
    iterator vit = this->lower_bound(id);
    if (vit == this->end() || this->key_comp()(id, vit->first)) this->insert(vit, value_type(id, v));
  }
  // BowVector.cpp:62-84: ascending-id accumulation of |v| (L1) or v*v then sqrt (L2); divides when the norm is positive
  void normalize(LNorm norm_type) {
    double norm = 0.0;
    if (norm_type == DBoW2::L1) {
      for (iterator it = begin(); it != end(); ++it) norm += std::fabs(it->second);
    } else {
      for (iterator it = begin(); it != end(); ++it) norm += it->second * it->second;
      norm = std::sqrt(norm);
    }
    if (norm > 0.0)
      for (iterator it = begin(); it != end(); ++it) it->second /= norm;
  }
  // BowVector.cpp:88-102: "<id, value>, <id, value>"
  friend std::ostream& operator<<(std::ostream& out, const BowVector& v) {
    const char* sep = "";
    for (const_iterator it = v.begin(); it != v.end(); ++it, sep = ", ") out << sep << "<" << it->first << ", " << it->second << ">";
    return out;
  }
  // BowVector.cpp:106-126: the dense row of W values, zeros spelled "0 "
  void saveM(const std::string& filename, size_t W) const {
    std::fstream f(filename.c_str(), std::ios::out);
    WordId next = 0;
    for (const_iterator it = begin(); it != end(); ++it) {
      for (; next < it->first; ++next) f << "0 ";
      f << it->second << " ";
      next = it->first + 1;
    }
    for (; next < (WordId)W; ++next) f << "0 ";
  }
};

}  // namespace DBoW2
#endif
#ifndef __D_T_FEATURE_VECTOR__
#define __D_T_FEATURE_VECTOR__
#include <iostream>
namespace DBoW2 {

class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {
#ifdef ORBX_HAVE_BOOST_SERIALIZATION
  friend class boost::serialization::access;
  template <class Archive>
  void serialize(Archive& ar, const int /*version*/) {   // FeatureVector.h:27-32
    ar& boost::serialization::base_object<std::map<NodeId, std::vector<unsigned int> > >(*this);
  }
#endif

 public:
  FeatureVector() {}
  ~FeatureVector() {}
  // FeatureVector.cpp:30-45
  void addFeature(NodeId id, unsigned int i_feature) {
    iterator vit = this->lower_bound(id);
    if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
    else {
      vit = this->insert(vit, value_type(id, std::vector<unsigned int>()));
      vit->second.push_back(i_feature);
    }
  }
  // FeatureVector.cpp:49-82: "<node: [f, f, ...]>, <node: [...]>"
  friend std::ostream& operator<<(std::ostream& out, const FeatureVector& v) {
    const char* sep = "";
    for (const_iterator it = v.begin(); it != v.end(); ++it, sep = ", ") {
      out << sep << "<" << it->first << ": [";
      for (size_t i = 0; i < it->second.size(); i++) out << (i ? ", " : "") << it->second[i];
      out << "]>";
    }
    return out;
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
    static const bool trace = std::getenv("ORBX_TRACE_BOW") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    Scratch& s = scratch();
    s.word.resize(n); s.node.resize(n); s.ids.resize(n); s.weight.resize(n); s.vals.resize(n);
    // Frame::ComputeBoW right after Frame::ExtractORB: `features` are the rows of Frame::mDescriptors, the very buffer the extractor adapter
    // filled and published.  Once this vocabulary is attached to that extractor's context (the first call below does it) the extraction's
    // own graph has already run the descent and the records wait in its pinned result block: no device round trip here.
    bool served = false;
    {
      const unsigned char* base = features[0].ptr<unsigned char>();
      bool contiguous = true;
      for (int i = 1; i < n && contiguous; i++) contiguous = features[i].ptr<unsigned char>() == base + (size_t)i * 32;
      if (contiguous) {
        const int rc = orbx_bow_transform_published(voc_, base, n, levelsup, s.word.data(), s.weight.data(), s.node.data());
        if (rc < 0) throw std::runtime_error(std::string("ORBVocabulary::transform: ") + orbx_last_error(ctx_));
        served = rc == ORBX_OK;
      }
    }
    double t_pack = us();
    if (!served) {  // Tracking, LocalMapping and LoopClosing all call transform on the one vocabulary: one caller at a time on its context
      s.desc.resize((size_t)n * 32);
      for (int i = 0; i < n; i++) std::memcpy(&s.desc[(size_t)i * 32], features[i].ptr<unsigned char>(), 32);
      t_pack = us();
      std::lock_guard<std::mutex> lock(mu_);
      if (orbx_bow_transform(voc_, s.desc.data(), n, levelsup, s.word.data(), s.weight.data(), s.node.data()) != ORBX_OK)
        throw std::runtime_error(std::string("ORBVocabulary::transform: ") + orbx_last_error(ctx_));
    }
    const double t_dev = us();
    int nnz = 0;
    orbx_bow_finalize(voc_, s.word.data(), s.weight.data(), n, s.ids.data(), s.vals.data(), &nnz);
    const double t_fin = us();
    for (int k = 0; k < nnz; k++) v.insert(v.end(), DBoW2::BowVector::value_type(s.ids[k], s.vals[k]));   // ascending ids: O(1) hinted inserts
    // FeatureVector: the reference appends feature i to node[i]'s list in feature order (TemplatedVocabulary.h:1166-1170); grouping the
    // (node, i) pairs by a stable sort first gives every list in the same order with ONE hinted map insert per node instead of a
    // tree lookup per feature
    s.order.clear();
    for (int i = 0; i < n; i++)
      if (s.weight[i] > 0) s.order.push_back(((uint64_t)s.node[i] << 32) | (uint32_t)i);
    std::sort(s.order.begin(), s.order.end());   // keys are unique (i is part of the key): feature order inside a node is ascending i
    for (size_t a = 0; a < s.order.size();) {
      const uint32_t nid = (uint32_t)(s.order[a] >> 32);
      size_t b = a;
      while (b < s.order.size() && (uint32_t)(s.order[b] >> 32) == nid) b++;
      DBoW2::FeatureVector::iterator it = fv.insert(fv.end(), DBoW2::FeatureVector::value_type(nid, std::vector<unsigned int>()));
      it->second.reserve(b - a);
      for (size_t k = a; k < b; k++) it->second.push_back((unsigned int)(uint32_t)s.order[k]);
      a = b;
    }
    if (trace)
      std::fprintf(stderr, "[orbx bow] n=%d nnz=%d: pack %.1f us, device call %.1f, finalize %.1f, maps %.1f\n", n, nnz, t_pack, t_dev - t_pack, t_fin - t_dev,
                   us() - t_fin);
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
  struct Scratch {   // per calling thread: transform runs once per frame / keyframe, its buffers are reused
    std::vector<uint8_t> desc;
    std::vector<uint32_t> word, node, ids;
    std::vector<double> weight, vals;
    std::vector<uint64_t> order;
  };
  static Scratch& scratch() { static thread_local Scratch s; return s; }
  orbx_ctx* ctx_ = nullptr;
  mutable std::mutex mu_;
  orbx_voc* voc_ = nullptr;
};

}  // namespace ORB_SLAM3

#endif  // ORBVOCABULARY_H
