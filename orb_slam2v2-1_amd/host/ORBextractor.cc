// ORBextractor.cc — host side of the drop-in ORBextractor: a thin C++ class over the C ABI
// (include/orbx.h).  Mirrors the reference's interface and error behaviour
// (src/ORBextractor.cc:410-470, 1043-1105); no pixel is processed on the CPU.
#include "ORBextractor.h"
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ORB_SLAM2 {

static_assert(sizeof(orbx_keypoint_t) == 28, "cv::KeyPoint wire layout");

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST,
                           int device, const orbx_flavour_t *flavour)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), mpHandle(nullptr), mbMaterializePyramid(true) {
    if (device < 0) {
        const char *e = std::getenv("ORBX_DEVICE");
        device = e ? std::atoi(e) : 0;
    }
    orbx_flavour_t fl = {};
    if (flavour) fl = *flavour;
    else if (const char *g = std::getenv("ORBX_GAUSS_ROUNDING")) {
        int k[4];
        if (!std::strcmp(g, "sse2")) fl.gauss_rounding = ORBX_GAUSS_ROUND_SSE2;
        else if (std::sscanf(g, "taps:%d,%d,%d,%d", &k[0], &k[1], &k[2], &k[3]) == 4) {   // OpenCV >= 3.4.1: the build's Q8 taps, centre first
            fl.gauss_rounding = ORBX_GAUSS_FIXED_TAPS;
            for (int i = 0; i < 4; i++) fl.gauss_taps[i] = k[i];
        } else if (std::strcmp(g, "half_up")) std::fprintf(stderr, "ORBextractor: ORBX_GAUSS_ROUNDING=%s ignored (half_up | sse2 | taps:k0,k1,k2,k3)\n", g);
    }
    if (orbx_create_flavoured(nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST, device, &fl, &mpHandle) != ORBX_OK) {
        std::fprintf(stderr, "ORBextractor: %s\n", orbx_last_error());  // the reference logs with cerr too
        mpHandle = nullptr;
        return;
    }
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    orbx_get_tables(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                    mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data());
    mvImagePyramid.resize(nlevels);
    mvPadded.resize(nlevels);
}

ORBextractor::~ORBextractor() { orbx_destroy(mpHandle); }

void ORBextractor::operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors) {
    if (image.empty()) return;  // :1046-1047
    // cv::_InputArray / cv::_OutputArray (real OpenCV) expose neither ptr() nor cols / rows / step: go through getMat(),
    // as the reference does (`Mat image = _image.getMat();`, :1049); the shim's Mat has a trivial getMat()
    const cv::Mat im = image.getMat();
    assert(im.type() == CV_8UC1);  // :1050
    if (!mpHandle) { _keypoints.clear(); _descriptors.release(); return; }

    int cap = orbx_max_keypoints(mpHandle) + 256;
    mvKpBuf.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    int rc = orbx_extract(mpHandle, im.ptr(0), im.cols, im.rows, (int)im.step, mvKpBuf.data(),
                          desc.ptr(0), cap, &n);
    if (rc != ORBX_OK) {
        std::fprintf(stderr, "ORBextractor: %s\n", orbx_last_error());
        _keypoints.clear(); _descriptors.release();
        return;
    }
    _keypoints.clear();
    _keypoints.reserve(n);
    if (n == 0) _descriptors.release();  // :1064-1065
    else {
        _descriptors.create(n, 32, CV_8U);  // :1068
        cv::Mat out = _descriptors.getMat();   // shares the buffer create() allocated (:1069)
        for (int i = 0; i < n; i++) std::memcpy(out.ptr(i), desc.ptr(i), 32);
    }
    for (int i = 0; i < n; i++) {
        const orbx_keypoint_t &s = mvKpBuf[i];
        cv::KeyPoint kp;
        kp.pt.x = s.x; kp.pt.y = s.y; kp.size = s.size; kp.angle = s.angle; kp.response = s.response;
        kp.octave = s.octave; kp.class_id = s.class_id;
        _keypoints.push_back(kp);
    }
    if (mbMaterializePyramid) {
        for (int l = 0; l < nlevels; l++) {
            int w = 0, h = 0;
            if (orbx_pyramid_host(mpHandle, 0, l, 1, nullptr, 0, &w, &h) != ORBX_OK) break;
            mvPadded[l].create(h, w, CV_8U);
            orbx_pyramid_host(mpHandle, 0, l, 1, mvPadded[l].ptr(0), (int)mvPadded[l].step, &w, &h);
            // inner ROI of the bordered buffer: temp(Rect(EDGE, EDGE, sz.width, sz.height))  (:1115)
            mvImagePyramid[l] = cv::Mat(h - 38, w - 38, CV_8U, mvPadded[l].ptr(19) + 19, mvPadded[l].step);
        }
    }
}

}  // namespace ORB_SLAM2
