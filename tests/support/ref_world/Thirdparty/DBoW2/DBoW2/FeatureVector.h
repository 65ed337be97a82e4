// TEST INFRASTRUCTURE: DBoW2::FeatureVector / BowVector as the containers the matcher path sees
// (Thirdparty/DBoW2/DBoW2/FeatureVector.h, BowVector.h: std::map subclasses).  Same include guards as the reference's
// headers and as include/ORBVocabulary.h, so whichever comes first wins.
#pragma once
#include <map>
#include <vector>
#ifndef __D_T_BOW_VECTOR__
#define __D_T_BOW_VECTOR__
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {
 public:
  void addWeight(WordId id, WordValue v);   // Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-46
};
inline void BowVector::addWeight(WordId id, WordValue v) {
  iterator vit = this->lower_bound(id);
  if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
  else this->insert(vit, value_type(id, v));
}
}  // namespace DBoW2
#endif
#ifndef __D_T_FEATURE_VECTOR__
#define __D_T_FEATURE_VECTOR__
namespace DBoW2 {
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {
 public:
  void addFeature(NodeId id, unsigned int i_feature);   // Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:30-45
};
inline void FeatureVector::addFeature(NodeId id, unsigned int i_feature) {
  iterator vit = this->lower_bound(id);
  if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
  else {
    vit = this->insert(vit, value_type(id, std::vector<unsigned int>()));
    vit->second.push_back(i_feature);
  }
}
}  // namespace DBoW2
#endif
