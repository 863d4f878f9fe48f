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
#include <map>
#include <set>
#include <vector>
#include "cv_shim.h"
#include "ORBextractor.h"
#include "ORBVocabulary.h"

namespace ORB_SLAM2 {

class Frame;
class KeyFrame;

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
    inline int PredictScale(const float &currentDist, KeyFrame *pKF);   // src/MapPoint.cc:397-412
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    // observation graph, reduced to what ORBmatcher::Fuse / SearchBySim3 touch
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }            // src/MapPoint.cc
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    inline void AddObservation(KeyFrame *pKF, size_t idx);                                // src/MapPoint.cc:108-119
    inline void Replace(MapPoint *pMP);                                                   // src/MapPoint.cc:187-225
    std::map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    void SetDescriptor(const cv::Mat &d) { mDescriptor = d.clone(); }   // the store at src/MapPoint.cc:313-316 (test shim only)
    bool mbBad;
    int nObs;
    long unsigned int mnLastFrameSeen = 0;   // include/MapPoint.h
    int mnVisible = 1;
    void IncreaseVisible(int n = 1) { mnVisible += n; }
    float mfMinDistance = 0.f, mfMaxDistance = 0.f;
    cv::Mat mDescriptor, mWorldPos;
    cv::Mat mNormalVector = cv::Mat::zeros(3, 1, CV_32F);
    std::map<KeyFrame *, size_t> mObservations;
};

class Frame {
public:
    Frame() : mpORBextractorLeft(nullptr), mpORBextractorRight(nullptr), mbf(0), mb(0), N(0) {}
    long unsigned int mnId = 0;
    // BoW (include/Frame.h:133-138, src/Frame.cc:410-417)
    ORBVocabulary *mpORBvocabulary = nullptr;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    void ComputeBoW() {
        if (mBowVec.empty()) {
            std::vector<cv::Mat> vCurrentDesc;   // Converter::toDescriptorVector
            for (int j = 0; j < mDescriptors.rows; j++) vCurrentDesc.push_back(mDescriptors.row(j));
            mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
        }
    }
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

class KeyFrame {   // include/KeyFrame.h: the members the projection matchers read or write
public:
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }  // src/KeyFrame.cc
    std::set<MapPoint *> GetMapPoints() {
        std::set<MapPoint *> s;
        for (size_t i = 0; i < mvpMapPoints.size(); i++)
            if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
        return s;
    }
    bool isBad() { return mbBad; }
    bool mbBad = false;
    ORBVocabulary *mpORBvocabulary = nullptr;   // src/KeyFrame.cc:59-68
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    void ComputeBoW() {
        if (mBowVec.empty() || mFeatVec.empty()) {
            std::vector<cv::Mat> vCurrentDesc;
            for (int j = 0; j < mDescriptors.rows; j++) vCurrentDesc.push_back(mDescriptors.row(j));
            mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
        }
    }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = static_cast<MapPoint *>(NULL); }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetRotation() {
        cv::Mat R(3, 3, CV_32F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c);
        return R;
    }
    cv::Mat GetTranslation() {
        cv::Mat t(3, 1, CV_32F);
        for (int r = 0; r < 3; r++) t.at<float>(r) = Tcw.at<float>(r, 3);
        return t;
    }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    void SetPose(const cv::Mat &Tcw_) {   // src/KeyFrame.cc:70-84: Ow = -Rwc*tcw (gemm, alpha = -1)
        Tcw = Tcw_.clone();
        Ow = cv::Mat(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)Tcw.at<float>(k, i) * (double)Tcw.at<float>(k, 3);
            Ow.at<float>(i) = (float)(s * -1.0);
        }
    }
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0.f;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2, mvLevelSigma2;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;   // ints: include/KeyFrame.h:199-202
    float mfGridElementWidthInv = 0.f, mfGridElementHeightInv = 0.f;
    std::vector<MapPoint *> mvpMapPoints;
    cv::Mat Tcw = cv::Mat::eye(4, 4, CV_32F), Ow = cv::Mat::zeros(3, 1, CV_32F);
};

inline int MapPoint::PredictScale(const float &currentDist, KeyFrame *pKF) {
    using namespace std;
    float ratio = mfMaxDistance / currentDist;
    int nScale = ceil(log(ratio) / pKF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
    return nScale;
}
inline void MapPoint::AddObservation(KeyFrame *pKF, size_t idx) {
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    if (pKF->mvuRight[idx] >= 0) nObs += 2; else nObs++;
}
inline void MapPoint::Replace(MapPoint *pMP) {   // without the found/visible counters, descriptor refresh and Map erase
    if (pMP == this) return;
    std::map<KeyFrame *, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    for (std::map<KeyFrame *, size_t>::iterator mit = obs.begin(), mend = obs.end(); mit != mend; mit++) {
        KeyFrame *pKF = mit->first;
        if (!pMP->IsInKeyFrame(pKF)) { pKF->ReplaceMapPointMatch(mit->second, pMP); pMP->AddObservation(pKF, mit->second); }
        else pKF->EraseMapPointMatch(mit->second);
    }
}

}  // namespace ORB_SLAM2
#endif
