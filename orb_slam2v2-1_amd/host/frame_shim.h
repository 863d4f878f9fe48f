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
#include <vector>
#include "cv_shim.h"
#include "ORBextractor.h"

namespace ORB_SLAM2 {

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
    bool mbBad;
    int nObs;
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
};

}  // namespace ORB_SLAM2
#endif
