/* orbx adapter — drop-in replacement for the reference's include/KeyFrameDatabase.h (:42-97): the same class, constructors and public
 * routines with the same signatures (add / erase / clear / clearMap, DetectLoopCandidates, DetectCandidates, DetectBestCandidates,
 * DetectNBestCandidates, DetectRelocalizationCandidates, SetORBVocabulary; PreSave / PostLoad are declared and never defined in the
 * reference either), so that Tracking.cc, LoopClosing.cc, KeyFrame.cc, Map.cc, Atlas.cc and System.cc compile and link unchanged.
 * The definitions live in orb_slam3_modified_amd/csrc/ref_adapter/KeyFrameDatabase.cc (it replaces src/KeyFrameDatabase.cc in the
 * reference's build): there is no inverted file — the keyframes' BowVectors are resident in HBM as CSR
 * (orbx_kfdb_*), one device pass lists every keyframe sharing a word with the query in the reference's list order with its
 * common-word count, a second one scores the keyframes a routine selected; which keyframes enter a list (same map / another map /
 * not connected), the side effects on KeyFrame::mn*Query / mn*Words / m*Score and the covisibility accumulation run on the host over
 * the reference's own objects, in the reference's order.
 *
 * Like the reference's header this one includes the reference's KeyFrame.h / Frame.h / Map.h and ORBVocabulary.h — it is meant to
 * sit in the reference's include/ directory next to this repository's ORBVocabulary.h.  In this repository's tests the same names
 * resolve to the small object model of tests/support/ref_world/.
 */
#ifndef KEYFRAMEDATABASE_H
#define KEYFRAMEDATABASE_H

#include <list>
#include <map>
#include <mutex>
#include <set>
#include <unordered_map>
#include <vector>

#include "KeyFrame.h"
#include "Frame.h"
#include "ORBVocabulary.h"
#include "Map.h"

#if defined(__has_include)
#if __has_include(<boost/serialization/base_object.hpp>)
#include <boost/serialization/base_object.hpp>
#include <boost/serialization/vector.hpp>
#include <boost/serialization/list.hpp>
#define ORBX_KFDB_HAVE_BOOST 1
#endif
#endif

#include "orbx.h"

namespace ORB_SLAM3 {

class KeyFrame;
class Frame;
class Map;

class KeyFrameDatabase {
#ifdef ORBX_KFDB_HAVE_BOOST
  friend class boost::serialization::access;
  template <class Archive>
  void serialize(Archive& ar, const unsigned int /*version*/) {   // include/KeyFrameDatabase.h:51-55
    ar& mvBackupInvertedFileId;
  }
#endif

 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  KeyFrameDatabase() {}
  KeyFrameDatabase(const ORBVocabulary& voc);
  ~KeyFrameDatabase();
  KeyFrameDatabase(const KeyFrameDatabase&) = delete;
  KeyFrameDatabase& operator=(const KeyFrameDatabase&) = delete;

  void add(KeyFrame* pKF);
  void erase(KeyFrame* pKF);
  void clear();
  void clearMap(Map* pMap);

  // Loop Detection (DEPRECATED in the reference, still compiled)
  std::vector<KeyFrame*> DetectLoopCandidates(KeyFrame* pKF, float minScore);

  // Loop and Merge Detection
  void DetectCandidates(KeyFrame* pKF, float minScore, std::vector<KeyFrame*>& vpLoopCand, std::vector<KeyFrame*>& vpMergeCand);
  void DetectBestCandidates(KeyFrame* pKF, std::vector<KeyFrame*>& vpLoopCand, std::vector<KeyFrame*>& vpMergeCand, int nMinWords);
  void DetectNBestCandidates(KeyFrame* pKF, std::vector<KeyFrame*>& vpLoopCand, std::vector<KeyFrame*>& vpMergeCand, int nNumCandidates);

  // Relocalization
  std::vector<KeyFrame*> DetectRelocalizationCandidates(Frame* F, Map* pMap);

  void PreSave();                                                   // declared, never defined (as in the reference)
  void PostLoad(std::map<long unsigned int, KeyFrame*> mpKFid);     // declared, never defined (as in the reference)
  void SetORBVocabulary(ORBVocabulary* pORBVoc);

 protected:
  // Associated vocabulary
  const ORBVocabulary* mpVoc = nullptr;

  // For save relation without pointer (include/KeyFrameDatabase.h:89-90: the only member the archive holds)
  std::vector<std::list<long unsigned int> > mvBackupInvertedFileId;

  // Mutex
  std::mutex mMutex;

 private:
  struct Sharing;                       // one query's sharing list: keyframes in the reference's list order + common-word counts
  void EnsureDb();
  bool Share(const DBoW2::BowVector& q, Sharing& out, std::unique_lock<std::mutex>& lock);
  void Score(const DBoW2::BowVector& q, const std::vector<KeyFrame*>& sel, std::vector<float>& si);

  orbx_ctx* ctx_ = nullptr;             // own context: queries come from Tracking (relocalisation) and LoopClosing under mMutex
  orbx_kfdb* db_ = nullptr;
  std::unordered_map<long unsigned int, KeyFrame*> kfs_;   // KeyFrame of every row resident in HBM
};

}  // namespace ORB_SLAM3

#endif  // KEYFRAMEDATABASE_H
