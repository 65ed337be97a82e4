// orbx adapter — replaces src/KeyFrameDatabase.cc of the reference (lturing/ORB_SLAM3_modified) behind the unchanged class interface
// of include/KeyFrameDatabase.h.  Citations `:n` are lines of the reference's src/KeyFrameDatabase.cc.
//
// The reference walks an inverted file word -> list<KeyFrame*> once per query word and does three things at every visit of a
// keyframe: decide whether it enters the routine's list (first visit only), count the visit, and — implicitly, by the walk order —
// fix the list order.  Here the keyframes' BowVectors are rows of a CSR block in HBM and ONE device pass (orbx_kfdb_sharing) returns
// what the whole walk leaves behind: every keyframe that shares a word, in the walk's order of first visits (smallest shared word,
// then add() order), with the number of visits.  `Visit` below then applies the per-routine rule once per keyframe with that count,
// which is the closed form of the reference's loop body executed `count` times.  The thresholds (maxCommonWords, 0.8f, nMinWords)
// follow from the routine's own list, mpVoc->score runs on the device for exactly the keyframes above them (orbx_kfdb_score, doubles
// bit-identical to L1Scoring::score), and the covisibility accumulation / candidate selection run on the host over the reference's
// own KeyFrame objects with the reference's float arithmetic, in list order.
//
// Divergences, stated: (i) DetectNBestCandidates skips a bad keyframe where the reference's `continue` (:707-708) never advances and
// spins forever; (ii) add() of a keyframe that is already in the database replaces its row (the reference would list it twice);
// (iii) the reference's add() indexes mvInvertedFile[word] without a bounds check — a word id beyond the vocabulary is undefined
// behaviour there and simply a word here.
#include "KeyFrameDatabase.h"

#include <algorithm>
#include <stdexcept>
#include <string>

using namespace std;

namespace ORB_SLAM3 {

struct KeyFrameDatabase::Sharing {
  std::vector<KeyFrame*> kf;
  std::vector<int> words;
};

namespace {

// the three fields a routine uses on every keyframe (include/KeyFrame.h:335-346)
struct Slot {
  long unsigned int KeyFrame::*query;
  int KeyFrame::*words;
  float KeyFrame::*score;
};
const Slot kLoop = {&KeyFrame::mnLoopQuery, &KeyFrame::mnLoopWords, &KeyFrame::mLoopScore};
const Slot kMerge = {&KeyFrame::mnMergeQuery, &KeyFrame::mnMergeWords, &KeyFrame::mMergeScore};
const Slot kPlace = {&KeyFrame::mnPlaceRecognitionQuery, &KeyFrame::mnPlaceRecognitionWords, &KeyFrame::mPlaceRecognitionScore};
const Slot kReloc = {&KeyFrame::mnRelocQuery, &KeyFrame::mnRelocWords, &KeyFrame::mRelocScore};

// What `count` visits of keyframe k by the inverted-file walk leave behind (e.g. :119-130).  Every visit does
//     if (k->query != id) { k->words = 0; if (listable) { k->query = id; list.push_back(k); } }   k->words++;
// so a listable keyframe is pushed at its first visit and ends with `count` words; one that is not listable keeps a query != id,
// is reset at EVERY visit and ends with 1; one whose query already equals id (the same id asked twice in a row) is neither reset
// nor pushed and gains `count`.
inline void Visit(KeyFrame* k, const Slot& s, long unsigned int id, int count, bool listable, std::vector<KeyFrame*>& list) {
  if (k->*(s.query) != id) {
    if (listable) { k->*(s.query) = id; list.push_back(k); k->*(s.words) = count; }
    else k->*(s.words) = 1;
  } else k->*(s.words) += count;
}

// maxCommonWords over a list and the threshold derived from it (e.g. :143-150): `int minCommonWords = maxCommonWords*0.8f;`
inline int MinCommonWords(const std::vector<KeyFrame*>& list, const Slot& s) {
  int maxCommonWords = 0;
  for (KeyFrame* k : list) maxCommonWords = std::max(maxCommonWords, k->*(s.words));
  return (int)(maxCommonWords * 0.8f);
}

typedef std::vector<std::pair<float, KeyFrame*> > ScoreList;

// Covisibility accumulation (e.g. :176-205): every scored keyframe collects the scores of its ten best covisible keyframes that took
// part in this query (`counts(pKF2)`), and hands its place to the best-scoring member of that group.  Float sums in list order.
template <class Counts>
float Accumulate(const ScoreList& scored, const Slot& s, Counts counts, float bestAccScore, ScoreList& acc) {
  acc.clear();
  acc.reserve(scored.size());
  for (const std::pair<float, KeyFrame*>& e : scored) {
    const std::vector<KeyFrame*> vpNeighs = e.second->GetBestCovisibilityKeyFrames(10);
    float bestScore = e.first, accScore = e.first;
    KeyFrame* pBestKF = e.second;
    for (KeyFrame* pKF2 : vpNeighs) {
      if (!counts(pKF2)) continue;
      const float s2 = pKF2->*(s.score);
      accScore += s2;
      if (s2 > bestScore) { pBestKF = pKF2; bestScore = s2; }
    }
    acc.push_back(std::make_pair(accScore, pBestKF));
    if (accScore > bestAccScore) bestAccScore = accScore;
  }
  return bestAccScore;
}

// every group representative whose accumulated score exceeds 0.75 * best, once each, in list order (e.g. :207-226)
template <class Emit>
void Retain(const ScoreList& acc, float bestAccScore, Emit emit) {
  const float minScoreToRetain = 0.75f * bestAccScore;
  std::set<KeyFrame*> spAlreadyAddedKF;
  for (const std::pair<float, KeyFrame*>& e : acc)
    if (e.first > minScoreToRetain && spAlreadyAddedKF.insert(e.second).second) emit(e.second);
}

void Flatten(const DBoW2::BowVector& v, std::vector<uint32_t>& ids, std::vector<double>& vals) {
  ids.clear(); vals.clear();
  ids.reserve(v.size()); vals.reserve(v.size());
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) { ids.push_back(it->first); vals.push_back(it->second); }
}

[[noreturn]] void Fail(const char* what, orbx_ctx* ctx) {
  throw std::runtime_error(std::string("KeyFrameDatabase::") + what + ": " + (ctx ? orbx_last_error(ctx) : "no orbx context"));
}

}  // namespace

// the device context is created here, so that a missing GPU fails at start-up (System's constructor) and not as an uncaught exception in
// the LocalMapping thread's first add()
KeyFrameDatabase::KeyFrameDatabase(const ORBVocabulary& voc) : mpVoc(&voc) { EnsureDb(); }

KeyFrameDatabase::~KeyFrameDatabase() {
  if (db_) orbx_kfdb_destroy(db_);
  if (ctx_) orbx_destroy(ctx_);
}

void KeyFrameDatabase::EnsureDb() {
  if (db_) return;
  if (orbx_create(&ctx_, 1, 1.2f, 1, 20, 7, -1) != ORBX_OK) { ctx_ = nullptr; throw std::runtime_error("KeyFrameDatabase: no MI355X / HIP device (there is no CPU fallback)"); }
  if (orbx_kfdb_create(ctx_, &db_) != ORBX_OK) Fail("KeyFrameDatabase", ctx_);
}

void KeyFrameDatabase::add(KeyFrame* pKF) {   // :39-45
  unique_lock<mutex> lock(mMutex);
  EnsureDb();
  std::vector<uint32_t> ids;
  std::vector<double> vals;
  Flatten(pKF->mBowVec, ids, vals);
  if (kfs_.count(pKF->mnId)) orbx_kfdb_erase(db_, (int64_t)pKF->mnId);
  if (orbx_kfdb_add(db_, (int64_t)pKF->mnId, ids.data(), vals.data(), (int)ids.size()) != ORBX_OK) Fail("add", ctx_);
  kfs_[pKF->mnId] = pKF;
}

void KeyFrameDatabase::erase(KeyFrame* pKF) {   // :47-66
  unique_lock<mutex> lock(mMutex);
  if (!db_) return;
  std::unordered_map<long unsigned int, KeyFrame*>::iterator it = kfs_.find(pKF->mnId);
  if (it == kfs_.end() || it->second != pKF) return;
  orbx_kfdb_erase(db_, (int64_t)pKF->mnId);
  kfs_.erase(it);
}

void KeyFrameDatabase::clear() {   // :68-72
  unique_lock<mutex> lock(mMutex);
  if (db_) orbx_kfdb_clear(db_);
  kfs_.clear();
}

void KeyFrameDatabase::clearMap(Map* pMap) {   // :74-98
  unique_lock<mutex> lock(mMutex);
  if (!db_) return;
  for (std::unordered_map<long unsigned int, KeyFrame*>::iterator it = kfs_.begin(); it != kfs_.end();) {
    if (pMap == it->second->GetMap()) { orbx_kfdb_erase(db_, (int64_t)it->first); it = kfs_.erase(it); }
    else ++it;
  }
}

void KeyFrameDatabase::SetORBVocabulary(ORBVocabulary* pORBVoc) {   // :850-858
  mpVoc = pORBVoc;
  clear();
}

// device pass 1: every keyframe sharing a word with q, in list order.  Takes mMutex through the CALLER's lock object, which stays locked
// until the caller has finished its Visit loop — the reference does the whole walk, marks included, under the mutex (:106-136 etc.)
bool KeyFrameDatabase::Share(const DBoW2::BowVector& q, Sharing& out, std::unique_lock<std::mutex>& lock) {
  out.kf.clear(); out.words.clear();
  lock = std::unique_lock<std::mutex>(mMutex);
  if (!db_ || kfs_.empty() || q.empty()) return false;
  std::vector<uint32_t> ids;
  std::vector<double> vals;
  Flatten(q, ids, vals);
  std::vector<int64_t> kf(kfs_.size());
  std::vector<int32_t> words(kfs_.size());
  int n = 0;
  if (orbx_kfdb_sharing(db_, ids.data(), (int)ids.size(), kf.data(), words.data(), (int)kf.size(), &n) != ORBX_OK) Fail("Detect*", ctx_);
  out.kf.reserve(n); out.words.reserve(n);
  for (int i = 0; i < n; i++) {
    out.kf.push_back(kfs_.at((long unsigned int)kf[i]));
    out.words.push_back(words[i]);
  }
  return n > 0;
}

// device pass 2: float si = mpVoc->score(q, pKFi->mBowVec) for the selected keyframes.  A keyframe another thread erased between the
// two passes (the reference scores through its pointer, outside the mutex) is scored on the host from its own mBowVec.
void KeyFrameDatabase::Score(const DBoW2::BowVector& q, const std::vector<KeyFrame*>& sel, std::vector<float>& si) {
  si.assign(sel.size(), 0.f);
  if (sel.empty()) return;
  std::vector<uint32_t> ids, kids;
  std::vector<double> vals, kvals;
  Flatten(q, ids, vals);
  unique_lock<mutex> lock(mMutex);
  std::vector<int64_t> resident;
  std::vector<size_t> where;
  for (size_t i = 0; i < sel.size(); i++) {
    std::unordered_map<long unsigned int, KeyFrame*>::iterator it = kfs_.find(sel[i]->mnId);
    if (it != kfs_.end() && it->second == sel[i]) { resident.push_back((int64_t)sel[i]->mnId); where.push_back(i); }
    else {
      Flatten(sel[i]->mBowVec, kids, kvals);
      si[i] = (float)orbx_bow_score_l1(ids.data(), vals.data(), (int)ids.size(), kids.data(), kvals.data(), (int)kids.size());
    }
  }
  std::vector<double> sc(resident.size());
  if (!resident.empty() && orbx_kfdb_score(db_, ids.data(), vals.data(), (int)ids.size(), resident.data(), (int)resident.size(), sc.data()) != ORBX_OK)
    Fail("Detect*", ctx_);
  for (size_t j = 0; j < resident.size(); j++) si[where[j]] = (float)sc[j];
}

namespace {

// the scored part of a list: keyframes with more than minCommonWords common words get their score; `keep` decides which of them go on
template <class Keep>
void ScoreList_(KeyFrameDatabase* self, void (KeyFrameDatabase::*score)(const DBoW2::BowVector&, const std::vector<KeyFrame*>&, std::vector<float>&),
                const DBoW2::BowVector& q, const std::vector<KeyFrame*>& list, const Slot& s, int minCommonWords, Keep keep, ScoreList& out) {
  std::vector<KeyFrame*> sel;
  for (KeyFrame* k : list)
    if (k->*(s.words) > minCommonWords) sel.push_back(k);
  std::vector<float> si;
  (self->*score)(q, sel, si);
  out.clear();
  for (size_t i = 0; i < sel.size(); i++) {
    sel[i]->*(s.score) = si[i];
    if (keep(si[i])) out.push_back(std::make_pair(si[i], sel[i]));
  }
}

}  // namespace

// :100-226
vector<KeyFrame*> KeyFrameDatabase::DetectLoopCandidates(KeyFrame* pKF, float minScore) {
  set<KeyFrame*> spConnectedKeyFrames = pKF->GetConnectedKeyFrames();
  Sharing sh;
  std::unique_lock<std::mutex> walk;
  Share(pKF->mBowVec, sh, walk);
  const long unsigned int id = pKF->mnId;
  std::vector<KeyFrame*> list;
  for (size_t i = 0; i < sh.kf.size(); i++) {
    KeyFrame* pKFi = sh.kf[i];
    if (pKFi->GetMap() == pKF->GetMap()) Visit(pKFi, kLoop, id, sh.words[i], !spConnectedKeyFrames.count(pKFi), list);   // a loop candidate must be in the same map
  }
  if (walk.owns_lock()) walk.unlock();
  if (list.empty()) return vector<KeyFrame*>();
  const int minCommonWords = MinCommonWords(list, kLoop);
  ScoreList scored, acc;
  ScoreList_(this, &KeyFrameDatabase::Score, pKF->mBowVec, list, kLoop, minCommonWords, [&](float si) { return si >= minScore; }, scored);
  if (scored.empty()) return vector<KeyFrame*>();
  const float best = Accumulate(scored, kLoop, [&](KeyFrame* k) { return k->mnLoopQuery == id && k->mnLoopWords > minCommonWords; }, minScore, acc);
  vector<KeyFrame*> vpLoopCandidates;
  vpLoopCandidates.reserve(acc.size());
  Retain(acc, best, [&](KeyFrame* k) { vpLoopCandidates.push_back(k); });
  return vpLoopCandidates;
}

// :228-466
void KeyFrameDatabase::DetectCandidates(KeyFrame* pKF, float minScore, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand) {
  set<KeyFrame*> spConnectedKeyFrames = pKF->GetConnectedKeyFrames();
  Sharing sh;
  std::unique_lock<std::mutex> walk;
  Share(pKF->mBowVec, sh, walk);
  const long unsigned int id = pKF->mnId;
  std::vector<KeyFrame*> loop, merge;
  for (size_t i = 0; i < sh.kf.size(); i++) {
    KeyFrame* pKFi = sh.kf[i];
    const bool listable = !spConnectedKeyFrames.count(pKFi);
    if (pKFi->GetMap() == pKF->GetMap()) Visit(pKFi, kLoop, id, sh.words[i], listable, loop);
    else if (!pKFi->GetMap()->IsBad()) Visit(pKFi, kMerge, id, sh.words[i], listable, merge);
  }
  if (walk.owns_lock()) walk.unlock();
  if (loop.empty() && merge.empty()) return;
  struct Group { std::vector<KeyFrame*>* list; const Slot* slot; vector<KeyFrame*>* out; };
  const Group groups[2] = {{&loop, &kLoop, &vpLoopCand}, {&merge, &kMerge, &vpMergeCand}};
  for (const Group& g : groups) {
    if (g.list->empty()) continue;
    const Slot& s = *g.slot;
    const int minCommonWords = MinCommonWords(*g.list, s);
    ScoreList scored, acc;
    ScoreList_(this, &KeyFrameDatabase::Score, pKF->mBowVec, *g.list, s, minCommonWords, [&](float si) { return si >= minScore; }, scored);
    if (scored.empty()) continue;
    const float best = Accumulate(scored, s, [&](KeyFrame* k) { return k->*(s.query) == id && k->*(s.words) > minCommonWords; }, minScore, acc);
    g.out->reserve(acc.size());
    Retain(acc, best, [&](KeyFrame* k) { g.out->push_back(k); });
  }
  // :453-464 — every keyframe the walk touched, of any map, loses both marks
  for (KeyFrame* pKFi : sh.kf) { pKFi->mnLoopQuery = -1; pKFi->mnMergeQuery = -1; }
}

// :468-591
void KeyFrameDatabase::DetectBestCandidates(KeyFrame* pKF, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand, int nMinWords) {
  set<KeyFrame*> spConnectedKF = pKF->GetConnectedKeyFrames();
  Sharing sh;
  std::unique_lock<std::mutex> walk;
  Share(pKF->mBowVec, sh, walk);
  const long unsigned int id = pKF->mnId;
  std::vector<KeyFrame*> list;
  for (size_t i = 0; i < sh.kf.size(); i++)
    if (spConnectedKF.find(sh.kf[i]) == spConnectedKF.end()) Visit(sh.kf[i], kPlace, id, sh.words[i], true, list);   // connected keyframes are skipped untouched (:485-488)
  if (walk.owns_lock()) walk.unlock();
  if (list.empty()) return;
  int minCommonWords = MinCommonWords(list, kPlace);
  if (minCommonWords < nMinWords) minCommonWords = nMinWords;
  ScoreList scored, acc;
  ScoreList_(this, &KeyFrameDatabase::Score, pKF->mBowVec, list, kPlace, minCommonWords, [](float) { return true; }, scored);
  if (scored.empty()) return;
  const float best = Accumulate(scored, kPlace, [&](KeyFrame* k) { return k->mnPlaceRecognitionQuery == id; }, 0.f, acc);
  vpLoopCand.reserve(acc.size());
  vpMergeCand.reserve(acc.size());
  Retain(acc, best, [&](KeyFrame* k) { (pKF->GetMap() == k->GetMap() ? vpLoopCand : vpMergeCand).push_back(k); });
}

// :599-731
void KeyFrameDatabase::DetectNBestCandidates(KeyFrame* pKF, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand, int nNumCandidates) {
  set<KeyFrame*> spConnectedKF = pKF->GetConnectedKeyFrames();
  Sharing sh;
  std::unique_lock<std::mutex> walk;
  Share(pKF->mBowVec, sh, walk);
  const long unsigned int id = pKF->mnId;
  std::vector<KeyFrame*> list;
  for (size_t i = 0; i < sh.kf.size(); i++) Visit(sh.kf[i], kPlace, id, sh.words[i], !spConnectedKF.count(sh.kf[i]), list);
  if (walk.owns_lock()) walk.unlock();
  if (list.empty()) return;
  const int minCommonWords = MinCommonWords(list, kPlace);
  ScoreList scored, acc;
  ScoreList_(this, &KeyFrameDatabase::Score, pKF->mBowVec, list, kPlace, minCommonWords, [](float) { return true; }, scored);
  if (scored.empty()) return;
  Accumulate(scored, kPlace, [&](KeyFrame* k) { return k->mnPlaceRecognitionQuery == id; }, 0.f, acc);
  // lAccScoreAndMatch.sort(compFirst) (:699): std::list::sort is a stable merge sort
  std::stable_sort(acc.begin(), acc.end(), [](const std::pair<float, KeyFrame*>& a, const std::pair<float, KeyFrame*>& b) { return a.first > b.first; });
  vpLoopCand.reserve(nNumCandidates);
  vpMergeCand.reserve(nNumCandidates);
  set<KeyFrame*> spAlreadyAddedKF;
  for (size_t i = 0; i < acc.size() && ((int)vpLoopCand.size() < nNumCandidates || (int)vpMergeCand.size() < nNumCandidates); i++) {
    KeyFrame* pKFi = acc[i].second;
    if (pKFi->isBad()) continue;   // the reference never advances here (:707-708)
    if (!spAlreadyAddedKF.count(pKFi)) {
      if (pKF->GetMap() == pKFi->GetMap() && (int)vpLoopCand.size() < nNumCandidates) vpLoopCand.push_back(pKFi);
      else if (pKF->GetMap() != pKFi->GetMap() && (int)vpMergeCand.size() < nNumCandidates && !pKFi->GetMap()->IsBad()) vpMergeCand.push_back(pKFi);
      spAlreadyAddedKF.insert(pKFi);
    }
  }
}

// :733-848
vector<KeyFrame*> KeyFrameDatabase::DetectRelocalizationCandidates(Frame* F, Map* pMap) {
  Sharing sh;
  std::unique_lock<std::mutex> walk;
  Share(F->mBowVec, sh, walk);
  const long unsigned int id = F->mnId;
  std::vector<KeyFrame*> list;
  for (size_t i = 0; i < sh.kf.size(); i++) Visit(sh.kf[i], kReloc, id, sh.words[i], true, list);
  if (walk.owns_lock()) walk.unlock();
  if (list.empty()) return vector<KeyFrame*>();
  const int minCommonWords = MinCommonWords(list, kReloc);
  ScoreList scored, acc;
  ScoreList_(this, &KeyFrameDatabase::Score, F->mBowVec, list, kReloc, minCommonWords, [](float) { return true; }, scored);
  if (scored.empty()) return vector<KeyFrame*>();
  const float best = Accumulate(scored, kReloc, [&](KeyFrame* k) { return k->mnRelocQuery == id; }, 0.f, acc);
  // :826-845 — a representative of another map is dropped before the already-added test (it may come back from a later entry)
  vector<KeyFrame*> vpRelocCandidates;
  vpRelocCandidates.reserve(acc.size());
  const float minScoreToRetain = 0.75f * best;
  set<KeyFrame*> spAlreadyAddedKF;
  for (const std::pair<float, KeyFrame*>& e : acc) {
    if (!(e.first > minScoreToRetain)) continue;
    if (e.second->GetMap() != pMap) continue;
    if (spAlreadyAddedKF.insert(e.second).second) vpRelocCandidates.push_back(e.second);
  }
  return vpRelocCandidates;
}

}  // namespace ORB_SLAM3
