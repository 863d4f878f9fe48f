// ORBmatcher.h — the hot routines of the reference's include/ORBmatcher.h:37-102 with the
// same names, signatures and constants, executed on an MI355X through include/orbx.h.
// Every public matcher of the reference's ORBmatcher is covered.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>
#include "cv_shim.h"
#include "frame_shim.h"
#include "orbx.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1649-1665)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches.
    // Used to track the local map (Tracking)                         (src/ORBmatcher.cc:45-129)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking)                   (src/ORBmatcher.cc:1330-1472)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking)                              (src/ORBmatcher.cc:1474-1601)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                           const int ORBdist);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing)                          (src/ORBmatcher.cc:290-403)
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints,
                           std::vector<MapPoint *> &vpMatched, int th);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    // In the stereo and RGB-D case, s12=1                            (src/ORBmatcher.cc:1104-1328)
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12,
                     const cv::Mat &R12, const cv::Mat &t12, const float th);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints. (src/ORBmatcher.cc:827-977)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.
    //                                                                (src/ORBmatcher.cc:979-1102)
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th,
             std::vector<MapPoint *> &vpReplacePoint);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame.
    // Brute force constrained to ORB that belong to the same vocabulary node (at a certain level)
    // Used in Relocalisation and Loop Detection                      (src/ORBmatcher.cc:159-288, 522-655)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.   (src/ORBmatcher.cc:657-825)
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);

    // Matching for the Map Initialization (only used in the monocular case) (src/ORBmatcher.cc:405-520)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                std::vector<int> &vnMatches12, int windowSize = 10);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    // GPU used by the matcher entry points (default: ORBX_DEVICE or 0)
    static int device;

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

// Frame::ComputeStereoMatches (src/Frame.cc:481-655): fills F.mvuRight / F.mvDepth from
// F.mvKeys / mvKeysRight / descriptors and the two extractors' device-resident pyramids.
// Call it from Frame::ComputeStereoMatches() in place of the CPU body. Returns #matches (<0: error).
int ComputeStereoMatchesHIP(Frame &F);

// The stereo Frame constructor's feature part in ONE GPU call (src/Frame.cc:78-84 + 481-655): ExtractORB(0, imLeft),
// ExtractORB(1, imRight) and ComputeStereoMatches.  Fills F.mvKeys / mDescriptors / mvKeysRight / mDescriptorsRight / N / mvuRight /
// mvDepth exactly as the three reference calls do, from F.mpORBextractorLeft alone (both images go through that extractor as one
// batch of two; the right extractor, its thread and the re-upload of the keypoints are not needed).  Replace
//     thread threadLeft(&Frame::ExtractORB,this,0,imLeft); thread threadRight(&Frame::ExtractORB,this,1,imRight); ...join...
//     N = mvKeys.size(); ... ComputeStereoMatches();
// by   ExtractStereoFrameHIP(*this, imLeft, imRight);   (mbf / mb must be set: they are constructor arguments / :114).
// Returns the number of stereo matches (< 0: error, outputs empty).
int ExtractStereoFrameHIP(Frame &F, const cv::Mat &imLeft, const cv::Mat &imRight);

// Tracking::SearchLocalPoints (src/Tracking.cc:1305-1339) from "Project points in frame" on: Frame::isInFrustum
// (src/Frame.cc:284-340) of every local map point that was not seen in this frame and is not bad, then
// ORBmatcher(nnratio).SearchByProjection(F, vpLocalMapPoints, th) — both in ONE GPU call.  Leaves on each point
// what isInFrustum leaves (mbTrackInView, mTrackProjX/XR/Y, mnTrackScaleLevel, mTrackViewCos), calls
// IncreaseVisible() for the points in view, fills F.mvpMapPoints; returns the matches, nToMatch as :1320-1323.
int SearchLocalPointsHIP(Frame &F, const std::vector<MapPoint *> &vpLocalMapPoints, float th, float nnratio, int &nToMatch);

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317) for MANY map points in one GPU call
// (the reference calls it point by point inside loops: src/LocalMapping.cc:170-180,520-535,
// src/Tracking.cc CreateInitialMap*, src/LoopClosing.cc SearchAndFuse).  Gathers the descriptors of every
// point's non-bad observing keyframes, picks the row with the least median Hamming distance on the GPU and
// returns it per point (empty Mat: the reference returns early and leaves mDescriptor untouched).
// In the reference tree MapPoint::ComputeDistinctiveDescriptors() itself keeps the final locked store.
int ComputeDistinctiveDescriptorsHIP(const std::vector<MapPoint *> &vpMapPoints, std::vector<cv::Mat> &vBest);

}  // namespace ORB_SLAM2
#endif
