// TEST INFRASTRUCTURE: C wrapper around the REFERENCE's own DBoW2 code (compiled from /root/reference by
// oracle/ref_fragments.mk into oracle/_ref/libref_dbow2.so).  Used only to validate the oracle's restatement
// (tests/test_ref_fragments.py); never shipped, never measured as product.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "DBoW2/FORB.h"
#include "DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> RefVocabulary;  // include/ORBVocabulary.h:27-29

static cv::Mat to_mat(const uint8_t* d) {
  cv::Mat m(1, 32, CV_8U);
  std::memcpy(m.ptr<unsigned char>(), d, 32);
  return m;
}

extern "C" {

int ref_forb_distance(const uint8_t* a, const uint8_t* b) { return DBoW2::FORB::distance(to_mat(a), to_mat(b)); }

void* ref_voc_load(const char* path) {
  RefVocabulary* v = new RefVocabulary();
  if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
  return v;
}
void ref_voc_save(void* h, const char* path) { ((RefVocabulary*)h)->saveToTextFile(path); }   // TemplatedVocabulary.h:1428-1449
void ref_voc_free(void* h) { delete (RefVocabulary*)h; }
int ref_voc_size(void* h) { return (int)((RefVocabulary*)h)->size(); }

// transform(features, BowVector&, FeatureVector&, levelsup)
int ref_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* ids, double* vals, uint32_t* fv_node,
                      uint32_t* fv_feat, int* n_fv) {
  std::vector<cv::Mat> feats;
  for (int i = 0; i < n; i++) feats.push_back(to_mat(desc + (size_t)i * 32));
  DBoW2::BowVector bv;
  DBoW2::FeatureVector fv;
  ((RefVocabulary*)h)->transform(feats, bv, fv, levelsup);
  int k = 0;
  for (auto& kv : bv) { ids[k] = kv.first; vals[k] = kv.second; k++; }
  int m = 0;
  for (auto& kv : fv) for (unsigned f : kv.second) { fv_node[m] = kv.first; fv_feat[m] = f; m++; }
  *n_fv = m;
  return k;
}

double ref_voc_score(void* h, const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb) {
  DBoW2::BowVector a, b;
  for (int i = 0; i < na; i++) a[ida[i]] = va[i];
  for (int i = 0; i < nb; i++) b[idb[i]] = vb[i];
  return ((RefVocabulary*)h)->score(a, b);
}

}  // extern "C"
