// TEST INFRASTRUCTURE: `ar & mBowVec; ar & mFeatVec;` (include/KeyFrame.h:130-131 of the reference, instantiated from
// src/System.cc:1464-1468) must compile and round-trip with the DBoW2::BowVector / FeatureVector that include/ORBVocabulary.h
// puts in front of the reference's translation units.  A miniature archive pair stands in for boost::archive::binary_[io]archive:
// it dispatches class types to boost::serialization::access::serialize — the entry point Boost itself uses — and stores
// std::map / std::vector / arithmetic values as bytes.
//   build A (-DORBX_OWN_DBOW2_TYPES, or no reference tree): the header's own classes
//   build B (-I <reference root>):                          the header defers to Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.h
#include <cstdio>
#include <cstring>
#include <map>
#include <sstream>
#include <type_traits>
#include <vector>

// The product header wants a HIP-free translation unit here: only the DBoW2 types are under test.
#include "ORBVocabulary.h"

namespace {

struct Bytes { std::vector<unsigned char> b; size_t rd = 0; };

template <bool Saving>
class MiniArchive {
 public:
  explicit MiniArchive(Bytes& s) : s_(s) {}
  template <class T>
  typename std::enable_if<std::is_arithmetic<T>::value, MiniArchive&>::type operator&(T& v) {
    if (Saving) { const unsigned char* p = (const unsigned char*)&v; s_.b.insert(s_.b.end(), p, p + sizeof(T)); }
    else { std::memcpy(&v, &s_.b[s_.rd], sizeof(T)); s_.rd += sizeof(T); }
    return *this;
  }
  template <class T>
  MiniArchive& operator&(std::vector<T>& v) {
    unsigned long long n = v.size();
    *this & n;
    if (!Saving) v.resize((size_t)n);
    for (T& x : v) *this & x;
    return *this;
  }
  template <class K, class V>
  MiniArchive& operator&(std::map<K, V>& m) {
    unsigned long long n = m.size();
    *this & n;
    if (Saving) {
      for (auto& kv : m) { K k = kv.first; *this & k; *this & kv.second; }
    } else {
      m.clear();
      for (unsigned long long i = 0; i < n; i++) { K k; V v; *this & k; *this & v; m.insert(m.end(), std::make_pair(k, v)); }
    }
    return *this;
  }
  // class types with a (private) serialize member: through boost::serialization::access, as Boost does
  template <class T>
  typename std::enable_if<std::is_class<T>::value, MiniArchive&>::type operator&(T& t) {
    boost::serialization::access::serialize(*this, t, 0u);
    return *this;
  }

 private:
  Bytes& s_;
};

// the members of include/KeyFrame.h that matter here
struct KeyFrameSlice {
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  template <class Archive>
  void serialize(Archive& ar, const int) {
    ar& mBowVec;     // KeyFrame.h:130
    ar& mFeatVec;    // KeyFrame.h:131
  }
};

}  // namespace

int main() {
  KeyFrameSlice a, b;
  for (unsigned i = 0; i < 300; i++) {
    const unsigned w = (i * 2654435761u) % 5000u;
    a.mBowVec.addWeight(w, 0.25 + i * 1e-3);
    a.mBowVec.addIfNotExist(w, 99.0);          // exists: no change
    a.mFeatVec.addFeature(w % 37, i);
  }
  a.mBowVec.addIfNotExist(5001, 0.5);
  a.mBowVec.normalize(DBoW2::L1);
  double l1 = 0;
  for (auto& kv : a.mBowVec) l1 += kv.second;
  Bytes bytes;
  MiniArchive<true> oa(bytes);
  oa& a;
  MiniArchive<false> ia(bytes);
  ia& b;
  const bool same = static_cast<std::map<DBoW2::WordId, DBoW2::WordValue>&>(a.mBowVec) == static_cast<std::map<DBoW2::WordId, DBoW2::WordValue>&>(b.mBowVec) &&
                    static_cast<std::map<DBoW2::NodeId, std::vector<unsigned int> >&>(a.mFeatVec) == static_cast<std::map<DBoW2::NodeId, std::vector<unsigned int> >&>(b.mFeatVec);
  DBoW2::BowVector small;
  small.addWeight(3, 0.5); small.addWeight(7, 0.25);
  DBoW2::FeatureVector fsmall;
  fsmall.addFeature(2, 5); fsmall.addFeature(2, 6); fsmall.addFeature(9, 1);
  std::ostringstream os;
  os << small << " | " << fsmall;
#if defined(ORBX_OWN_DBOW2_TYPES)
  const char* which = "own";
#else
  const char* which = "tree";
#endif
  std::printf("%s roundtrip=%d words=%zu nodes=%zu bytes=%zu l1=%.17g print=%s\n", which, (int)same, b.mBowVec.size(), b.mFeatVec.size(), bytes.b.size(), l1,
              os.str().c_str());
  return same && b.mBowVec.size() > 200 ? 0 : 1;
}
