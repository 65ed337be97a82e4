#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Every statement of the reference that CALLS the drop-in boundary from outside the files already compiled here
(src/Frame.cc, src/ORBmatcher.cc, src/KeyFrameDatabase.cc are compiled whole by oracle/ref_fragments.mk): the ORBmatcher constructions and
Search* / Fuse calls of src/Tracking.cc, src/LocalMapping.cc and src/LoopClosing.cc, the ORBextractor constructions of src/Tracking.cc, the
KeyFrameDatabase and vocabulary calls — lifted VERBATIM, at build time, from the reference's files where they lie (nothing of them is stored in
this repository) into one translation unit whose scaffolding declares the identifiers each statement uses with the types the reference's own
headers give them.  Compiling that unit (-fsyntax-only) over the drop-in headers (include/ORBmatcher.h, ORBextractor.h, KeyFrameDatabase.h,
ORBVocabulary.h) and tests/support/ref_world checks what the scenario drivers cannot: the reference's OWN phrasing of every call — argument types,
temporaries, defaulted parameters, overload resolution (`matcher.Fuse(pKFi, vpMapPointMatches, true)` binds `true` to `const float th` in the
reference, and must here) — against the replacement's declarations.  Tracking.cc / LocalMapping.cc / LoopClosing.cc as WHOLE files need Eigen,
g2o, Pangolin and the rest of the control plane, which this container does not have.

    python tools/gen_callsites.py /root/reference out.cpp       (tests/test_callsites.py compiles the result)
"""
import os
import re
import sys

# (context name, file, [(line, token that must be on that line)], scaffolding declared before the lines: identifier types as in the reference's headers)
CONTEXTS = [
    ("Tracking_newParameterLoader", "src/Tracking.cc", [(597, "new ORBextractor"), (600, "new ORBextractor"), (603, "new ORBextractor(5*nFeatures")],
     # include/Tracking.h:303-304 (ORBextractor* mpORBextractorLeft, *mpORBextractorRight; ORBextractor* mpIniORBextractor); src/Tracking.cc:590-595 locals
     "ORBextractor *mpORBextractorLeft, *mpORBextractorRight, *mpIniORBextractor; int nFeatures = 1000; int nLevels = 8; int fIniThFAST = 20; int fMinThFAST = 7; float fScaleFactor = 1.2f;"),
    ("Tracking_ParseORBParamFile", "src/Tracking.cc", [(1285, "new ORBextractor"), (1288, "new ORBextractor"), (1291, "new ORBextractor(5*nFeatures")],
     "ORBextractor *mpORBextractorLeft, *mpORBextractorRight, *mpIniORBextractor; int nFeatures = 1000; int nLevels = 8; int fIniThFAST = 20; int fMinThFAST = 7; float fScaleFactor = 1.2f;"),
    ("Tracking_MonocularInitialization", "src/Tracking.cc", [(2494, "ORBmatcher matcher(0.9,true)"), (2495, "SearchForInitialization")],
     # include/Tracking.h:176-181: Frame mCurrentFrame, mInitialFrame; std::vector<int> mvIniMatches; std::vector<cv::Point2f> mvbPrevMatched
     "Frame& mInitialFrame = W.f0; Frame& mCurrentFrame = W.f1; std::vector<cv::Point2f> mvbPrevMatched; std::vector<int> mvIniMatches;"),
    ("Tracking_TrackReferenceKeyFrame", "src/Tracking.cc", [(2730, "ORBmatcher matcher(0.7,true)"), (2733, "SearchByBoW")],
     "KeyFrame* mpReferenceKF = W.kf; Frame& mCurrentFrame = W.f1; vector<MapPoint*> vpMapPointMatches;"),
    ("Tracking_TrackWithMotionModel", "src/Tracking.cc", [(2859, "ORBmatcher matcher(0.9,true)"), (2889, "SearchByProjection(mCurrentFrame,mLastFrame,th"), (2897, "2*th")],
     "Frame& mCurrentFrame = W.f1; Frame& mLastFrame = W.f0; int th = 15; int mSensor = System::MONOCULAR;"),
    ("Tracking_SearchLocalPoints", "src/Tracking.cc", [(3393, "ORBmatcher matcher(0.8)"), (3416, "mvpLocalMapPoints")],
     # include/LocalMapping.h:102-103: bool mbFarPoints; float mThFarPoints
     "Frame& mCurrentFrame = W.f1; std::vector<MapPoint*> mvpLocalMapPoints; int th = 1; LocalMapping* mpLocalMapper = &W.lm;"),
    ("Tracking_Relocalization", "src/Tracking.cc", [(3620, "DetectRelocalizationCandidates"), (3631, "ORBmatcher matcher(0.75,true)"), (3651, "SearchByBoW(pKF,mCurrentFrame"),
                                                     (3670, "ORBmatcher matcher2(0.9,true)"), (3729, "sFound,10,100"), (3743, "sFound,3,64")],
     "Frame& mCurrentFrame = W.f1; KeyFrameDatabase* mpKeyFrameDB = W.db; Atlas* mpAtlas = &W.atlas; KeyFrame* pKF = W.kf; int i = 0; "
     "vector<vector<MapPoint*> > vvpMapPointMatches(1); set<MapPoint*> sFound;"),
    ("Tracking_Reset", "src/Tracking.cc", [(3809, "mpKeyFrameDB->clear()"), (3869, "clearMap(pMap)")], "KeyFrameDatabase* mpKeyFrameDB = W.db; Map* pMap = W.map;"),
    ("LocalMapping_CreateNewMapPoints", "src/LocalMapping.cc", [(412, "ORBmatcher matcher(th,false)"), (463, "vMatchedIndices"), (466, "SearchForTriangulation")],
     "float th = 0.6f; KeyFrame* mpCurrentKeyFrame = W.kf; KeyFrame* pKF2 = W.kf2; bool bCoarse = false;"),
    ("LocalMapping_SearchInNeighbors", "src/LocalMapping.cc", [(766, "ORBmatcher matcher;"), (767, "GetMapPointMatches"), (772, "matcher.Fuse(pKFi,vpMapPointMatches)"),
                                                               (773, "vpMapPointMatches,true"), (781, "vpFuseCandidates"), (802, "Fuse(mpCurrentKeyFrame,vpFuseCandidates)"), (803, "vpFuseCandidates,true")],
     "KeyFrame* mpCurrentKeyFrame = W.kf; KeyFrame* pKFi = W.kf2;"),
    ("LoopClosing_DetectCommonRegionsFromBoW", "src/LoopClosing.cc", [(591, "ORBmatcher matcherBoW(0.9, true)"), (592, "ORBmatcher matcher(0.75, true)"), (662, "matcherBoW.SearchByBoW"),
                                                                      (755, "vpKeyFrames, vpMatchedMP, vpMatchedKF, 8, 1.5"), (777, "vpMatchedMP, 5, 1.0")],
     # :749 Sophus::Sim3f mScw = Converter::toSophus(gScw); :751-754, :775-776 the matched vectors (the second vpMatchedMP of :775 lives in an inner scope there)
     "KeyFrame* mpCurrentKF = W.kf; vector<KeyFrame*> vpCovKFi(1, W.kf2); vector<vector<MapPoint*> > vvpMatchedMPs(1); int j = 0; Sophus::Sim3f mScw; "
     "vector<MapPoint*> vpMapPoints; vector<KeyFrame*> vpKeyFrames; vector<MapPoint*> vpMatchedMP; vector<KeyFrame*> vpMatchedKF;"),
    ("LoopClosing_FindMatchesByProjection", "src/LoopClosing.cc", [(961, "ORBmatcher matcher(0.9, true)"), (963, "vpMatchedMapPoints.resize"), (964, "vpMatchedMapPoints, 3, 1.5")],
     "KeyFrame* pCurrentKF = W.kf; Sophus::Sim3f mScw; vector<MapPoint*> vpMapPoints; vector<MapPoint*> vpMatchedMapPoints;"),
    ("LoopClosing_SearchAndFuse_corrected", "src/LoopClosing.cc", [(2117, "ORBmatcher matcher(0.8)"), (2132, "vpReplacePoints(vpMapPoints.size()"), (2133, "Fuse(pKFi,Scw,vpMapPoints,4,vpReplacePoints)")],
     "KeyFrame* pKFi = W.kf; Sophus::Sim3f Scw; vector<MapPoint*> vpMapPoints;"),
    ("LoopClosing_SearchAndFuse_keyframes", "src/LoopClosing.cc", [(2159, "ORBmatcher matcher(0.8)"), (2177, "vpReplacePoints(vpMapPoints.size()"), (2178, "Fuse(pKF,Scw,vpMapPoints,4,vpReplacePoints)")],
     "KeyFrame* pKF = W.kf; Sophus::Sim3f Scw; vector<MapPoint*> vpMapPoints;"),
    ("LoopClosing_NewDetectCommonRegions", "src/LoopClosing.cc", [(343, "mpKeyFrameDB->add(mpCurrentKF)"), (491, "DetectNBestCandidates(mpCurrentKF, vpLoopBowCand, vpMergeBowCand,3)")],
     "KeyFrameDatabase* mpKeyFrameDB = W.db; KeyFrame* mpCurrentKF = W.kf; vector<KeyFrame*> vpLoopBowCand, vpMergeBowCand;"),
    ("KeyFrame_ComputeBoW", "src/KeyFrame.cc", [(105, "mpORBvocabulary->transform(vCurrentDesc,mBowVec,mFeatVec,4)")],
     "ORBVocabulary* mpORBvocabulary = W.voc; vector<cv::Mat> vCurrentDesc; DBoW2::BowVector mBowVec; DBoW2::FeatureVector mFeatVec;"),
    ("System_ctor", "src/System.cc", [(118, "mpVocabulary->loadFromTextFile(strVocFile)")], "ORBVocabulary* mpVocabulary = W.voc; string strVocFile;"),
]

PRELUDE = r'''// GENERATED by tools/gen_callsites.py from the reference's sources where they lie: do not commit.  TEST INFRASTRUCTURE.
#include <set>
#include <string>
#include <utility>
#include <vector>
#include "ref_world.h"          // tests/support/ref_world: Frame, KeyFrame, MapPoint, Map over the shim types (what the reference's own .cc files compile against here)
#include "ORBextractor.h"       // the drop-in boundary: include/
#include "ORBVocabulary.h"
#include "ORBmatcher.h"
#include "KeyFrameDatabase.h"
using namespace std;
using namespace ORB_SLAM3;
namespace {
// the few names of the control plane the lifted statements mention: include/System.h:86-93 (eSensor), include/LocalMapping.h:102-103, include/Atlas.h:86
struct System { enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2, IMU_MONOCULAR = 3, IMU_STEREO = 4, IMU_RGBD = 5 }; };
struct LocalMapping { bool mbFarPoints = false; float mThFarPoints = 0.f; };
struct Atlas { Map* map = nullptr; Map* GetCurrentMap() { return map; } };
struct World { Frame &f0, &f1; KeyFrame *kf, *kf2; Map* map; KeyFrameDatabase* db; ORBVocabulary* voc; LocalMapping lm; Atlas atlas; };
}  // namespace
'''


# ---- second unit: the Frame constructions of src/Tracking.cc (GrabImageStereo / GrabImageRGBD / GrabImageMonocular), compiled like
# oracle/ref_fragments.mk compiles dropin_frame_world: the reference's UNMODIFIED include/Frame.h over include/ORBextractor.h + ORBVocabulary.h and
# the stand-ins of tests/support/frame_world — the extractor and vocabulary pointers Tracking owns are handed to Frame's constructors as written there
FRAME_LINES = [(1496, "mpORBextractorLeft,mpORBextractorRight,mpORBVocabulary,mK,mDistCoef,mbf,mThDepth,mpCamera);"),
               (1498, "mpCamera,mpCamera2,mTlr);"), (1500, "mpCamera,&mLastFrame,*mpImuCalib);"), (1502, "mpCamera2,mTlr,&mLastFrame,*mpImuCalib);"),
               (1546, "imDepth,timestamp,mpORBextractorLeft,mpORBVocabulary"), (1548, "imDepth,timestamp,mpORBextractorLeft,mpORBVocabulary"),
               (1590, "mpIniORBextractor,mpORBVocabulary,mpCamera,mDistCoef,mbf,mThDepth);"), (1592, "mpORBextractorLeft,mpORBVocabulary,mpCamera,mDistCoef,mbf,mThDepth);"),
               (1598, "mpIniORBextractor,mpORBVocabulary,mpCamera,mDistCoef,mbf,mThDepth,&mLastFrame,*mpImuCalib);"),
               (1601, "mpORBextractorLeft,mpORBVocabulary,mpCamera,mDistCoef,mbf,mThDepth,&mLastFrame,*mpImuCalib);")]
FRAME_PRELUDE = r'''// GENERATED by tools/gen_callsites.py --frames from the reference's src/Tracking.cc where it lies: do not commit.  TEST INFRASTRUCTURE.
// compile with the flags of oracle/ref_fragments.mk's dropin_frame_world (-include tests/support/frame_world/prelude.h -DFRAME_WORLD_DROPIN ...)
#include "Frame.h"              // the reference's own include/Frame.h; "ORBextractor.h" / "ORBVocabulary.h" resolve to the drop-in headers
using namespace std;
using namespace ORB_SLAM3;
// include/Tracking.h:176-182, 303-310, 326-340: the members the statements mention
void callsites_Tracking_GrabImage(cv::Mat& mImGray, cv::Mat& imGrayRight, cv::Mat& imDepth, double timestamp, ORBextractor* mpORBextractorLeft,
                                  ORBextractor* mpORBextractorRight, ORBextractor* mpIniORBextractor, ORBVocabulary* mpORBVocabulary, cv::Mat& mK,
                                  cv::Mat& mDistCoef, float mbf, float mThDepth, GeometricCamera* mpCamera, GeometricCamera* mpCamera2, Sophus::SE3f& mTlr,
                                  Frame& mLastFrame, Frame& mCurrentFrame, IMU::Calib* mpImuCalib) {
'''


def frames_unit(ref, out):
    path = os.path.join(ref, "src/Tracking.cc")
    src = open(path, encoding="utf-8", errors="replace").read().splitlines()
    body = []
    for ln, token in FRAME_LINES:
        text = src[ln - 1]
        if token not in text or "mCurrentFrame = Frame(" not in text:
            raise SystemExit(f"src/Tracking.cc:{ln} does not hold the expected Frame construction any more (found: {text.strip()[:100]})")
        body.append(f'#line {ln} "{path}"\n{text}')
    open(out, "w").write(FRAME_PRELUDE + "\n".join(body) + "\n}\n")
    print(f"{len(body)} Frame constructions of src/Tracking.cc -> {out}")


def main():
    if "--frames" in sys.argv:
        a = [x for x in sys.argv[1:] if x != "--frames"]
        return frames_unit(a[0], a[1])
    ref, out = sys.argv[1], sys.argv[2]
    parts = [PRELUDE]
    cache = {}
    n = 0
    for name, rel, lines, scaffold in CONTEXTS:
        path = os.path.join(ref, rel)
        if path not in cache:
            cache[path] = open(path, encoding="utf-8", errors="replace").read().splitlines()
        src = cache[path]
        body = []
        for ln, token in lines:
            text = src[ln - 1]
            if token not in text:
                raise SystemExit(f"{rel}:{ln} does not hold `{token}` any more (found: {text.strip()[:100]}): the reference moved, update tools/gen_callsites.py")
            body.append(f'#line {ln} "{path}"\n{text}')
            n += 1
        parts.append(f"// ---- {rel}: {name}\nvoid callsites_{name}(World& W) {{\n  {scaffold}\n" + "\n".join(body) + "\n}\n")
    open(out, "w").write("\n".join(parts))
    print(f"{n} statements of the reference in {len(CONTEXTS)} contexts -> {out}")


if __name__ == "__main__":
    main()
