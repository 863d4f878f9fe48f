// ORBextractor.h — drop-in replacement of the reference's include/ORBextractor.h:45-111.
// Same class name, namespace, constructor, operator() and getters, same public
// mvImagePyramid; the work happens on an MI355X through the C ABI in include/orbx.h.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>
#include "cv_shim.h"
#include "orbx.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // reference: src/ORBextractor.cc:410-470.  `device` selects the GPU (default 0 or the
    // ORBX_DEVICE environment variable).  Throws nothing: on failure the object is inert and
    // operator() leaves its outputs empty, ok() / lastError() tell why.
    // `flavour`: which OpenCV build this extractor stands in for (orbx_flavour_t, include/orbx.h: the rounding of
    // cv::GaussianBlur's column pass).  The reference's callers (src/Tracking.cc:119-125) pass five arguments; a deployment
    // selects the flavour of the build it replaces with the environment variable ORBX_GAUSS_ROUNDING = half_up | sse2 | taps:k0,k1,k2,k3, read
    // once per constructor (NULL argument); the default is half_up.
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = -1,
                 const orbx_flavour_t *flavour = nullptr);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image (mask ignored, as in the
    // reference: src/ORBextractor.cc:1043).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host copies of the pyramid levels of the last image, as in the reference
    // (Frame::ComputeStereoMatches reads them on the CPU: src/Frame.cc:488,578,595).
    // Each Mat is the inner ROI of a bordered buffer, like the reference's (:1113-1115).
    std::vector<cv::Mat> mvImagePyramid;
    // When the GPU stereo matcher is used nobody reads the host copies: skip the download.
    void SetMaterializePyramid(bool on) { mbMaterializePyramid = on; }

    bool ok() const { return mpHandle != nullptr; }
    const char *lastError() const { return orbx_last_error(); }
    orbx_extractor_t *handle() { return mpHandle; }  // for ORBmatcher / ComputeStereoMatches

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    orbx_extractor_t *mpHandle;
    bool mbMaterializePyramid;
    std::vector<cv::Mat> mvPadded;  // bordered buffers backing mvImagePyramid
    std::vector<orbx_keypoint_t> mvKpBuf;
};

}  // namespace ORB_SLAM2
#endif
