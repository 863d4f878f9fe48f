// frame_shim.h — the members of ORB_SLAM2::Frame / MapPoint that the hot matchers read or
// write (reference: include/Frame.h:100-188, include/MapPoint.h), so that ORBmatcher.cc
// compiles and is testable here without the rest of ORB-SLAM2.  In the reference tree, define
// ORBX_HAVE_ORBSLAM2 and the real "Frame.h" / "MapPoint.h" are used instead: the sources only
// touch members that exist there under the same names.
#pragma once
#ifdef ORBX_HAVE_ORBSLAM2
#include "Frame.h"
#include "MapPoint.h"
#else
#include <cmath>
#include <set>
#include <vector>
#include "cv_shim.h"
#include "ORBextractor.h"

namespace ORB_SLAM2 {

class Frame;

class MapPoint {
public:
    MapPoint() : mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(1.f), mTrackProjX(0), mTrackProjY(0),
                 mTrackProjXR(0), mbBad(false), nObs(0) { mWorldPos = cv::Mat::zeros(3, 1, CV_32F); }
    // tracking variables set by Frame::isInFrustum (src/Frame.cc:284-340)
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos, mTrackProjX, mTrackProjY, mTrackProjXR;
    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }  // src/MapPoint.cc:319-323
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    // scale-invariance distances (src/MapPoint.cc:385-395) and level prediction (:414-429)
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    inline int PredictScale(const float &currentDist, Frame *pF);
    bool mbBad;
    int nObs;
    float mfMinDistance = 0.f, mfMaxDistance = 0.f;
    cv::Mat mDescriptor, mWorldPos;
};

class Frame {
public:
    Frame() : mpORBextractorLeft(nullptr), mpORBextractorRight(nullptr), mbf(0), mb(0), N(0) {}
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    static float fx, fy, cx, cy;
    float mbf, mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    cv::Mat mTcw;
    std::vector<float> mvScaleFactors, mvInvScaleFactors;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    int mnScaleLevels = 0;          // src/Frame.cc:69
    float mfScaleFactor = 0.f, mfLogScaleFactor = 0.f;  // :70-71
};

inline int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    using namespace std;
    float ratio = mfMaxDistance / currentDist;
    int nScale = ceil(log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}

class KeyFrame {
public:
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }  // src/KeyFrame.cc
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<MapPoint *> mvpMapPoints;
};

}  // namespace ORB_SLAM2
#endif
