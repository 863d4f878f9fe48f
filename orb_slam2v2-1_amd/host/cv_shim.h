// cv_shim.h — the subset of OpenCV's core types the ORB hot path touches, for building the
// host-side classes where OpenCV is absent (this image).  Define ORBX_HAVE_OPENCV to compile
// the very same ORBextractor / ORBmatcher sources against the real <opencv2/core/core.hpp>.
#pragma once
#ifdef ORBX_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#include <cmath>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F

namespace cv {
typedef unsigned char uchar;
template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
};
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;

struct KeyPoint {  // field order of cv::KeyPoint (reference: include/BoostArchiver.h:47-57)
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};

// 2-D, single-channel, reference-counted matrix: enough for gray images, N x 32 descriptor
// tables and 4x4 / 3x1 float poses.
class Mat {
public:
    int rows, cols;
    size_t step;
    uchar *data;
    Mat() : rows(0), cols(0), step(0), data(nullptr), type_(CV_8U) {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t stp = 0)
        : rows(r), cols(c), step(stp ? stp : (size_t)c * esz(type)), data((uchar *)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && buf_) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        buf_.reset(new uchar[(size_t)r * step + 64], std::default_delete<uchar[]>());
        data = buf_.get();
    }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * esz(type_));
        return m;
    }
    uchar *ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    template <typename T> T &at(int r, int c = 0) { return ((T *)(data + (size_t)r * step))[c]; }
    template <typename T> const T &at(int r, int c = 0) const { return ((const T *)(data + (size_t)r * step))[c]; }
    Mat row(int r) const { Mat m(1, cols, type_, (void *)ptr(r), step); m.buf_ = buf_; return m; }
    // cv::_InputArray / cv::_OutputArray hand out the matrix with getMat(); here InputArray / OutputArray ARE Mat
    // references, so that the same source line (`cv::Mat im = image.getMat();`) compiles against both
    Mat getMat() const { return *this; }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, (size_t)r * m.step); return m; }
    static Mat eye(int r, int c, int type) {
        Mat m = zeros(r, c, type);
        for (int i = 0; i < (r < c ? r : c); i++) m.at<float>(i, i) = 1.f;
        return m;
    }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_;
    std::shared_ptr<uchar> buf_;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
#endif
