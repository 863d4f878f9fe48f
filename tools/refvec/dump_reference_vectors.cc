// dump_reference_vectors.cc — runs the REFERENCE implementation (kimwin2/ORB_SLAM2v2-1) on the committed synthetic inputs and
// writes what it computes, stage by stage, into one "ORBVEC01" file per case.  Dropping those files into tests/golden/ pins
// this repository's CPU oracle (and through it the HIP path) against a real ORB-SLAM2 + OpenCV build:
// tests/test_reference_vectors.py consumes them with no further code.
//
// This program is NOT built or run by this repository (OpenCV and the reference's dependencies are absent from its image).
// It is compiled by a maintainer who has OpenCV (2.4.x or 3.x, the versions the reference's README names) and a checkout of
// the reference; it #includes the reference's own sources from that checkout - nothing of the reference is copied here.
//
//   A. extractor only (needs OpenCV + the reference's src/ORBextractor.cc, include/ORBextractor.h):
//        g++ -O3 -march=native -std=c++11 -I$REF/include -I. dump_reference_vectors.cc $REF/src/ORBextractor.cc \
//            `pkg-config --cflags --libs opencv` -lpthread -o dump_reference_vectors
//   B. + Frame::ComputeStereoMatches (needs the reference's full build tree: libORB_SLAM2.so, Eigen, DBoW2, g2o ...):
//        g++ -O3 -march=native -std=c++11 -DWITH_FRAME -I$REF -I$REF/include -I/usr/include/eigen3 -I. \
//            dump_reference_vectors.cc -L$REF/lib -lORB_SLAM2 `pkg-config --cflags --libs opencv` -lpthread -o dump_reference_vectors
//      (-O3 -march=native mirror the reference's CMakeLists.txt:26-27; see tests/golden/README.md for the flags that matter.)
//   run:  python tools/refvec/write_refvec_inputs.py in/ ;  ./dump_reference_vectors in/ out/ ;  cp out/ref_*.orbvec tests/golden/
//
// What is dumped per case (keys under "<case>/", see tests/golden/README.md):
//   the constructor tables (a1), per level the pyramid / its 19-px border / the 7x7 Gaussian blur as CRC-32 (pixel arrays
//   for cases marked full) (a2, a8), cv::FAST on the level's FAST region at both thresholds (a4), DistributeOctTree called
//   directly on those candidates (a5, a6), the keypoints ComputeKeyPointsOctTree keeps (a3, a7), operator()'s final
//   keypoints and descriptors (a9, a10, a18) and - build B - mvuRight / mvDepth of Frame::ComputeStereoMatches (a17).
//
// The quad-tree tie-break.  DistributeOctTree sorts (size, ExtractorNode*) pairs (src/ORBextractor.cc:684,721): among nodes of
// equal size the ORDER OF HEAP ADDRESSES decides, so the reference's output depends on the allocator's state.  This program
// replaces the global operator new with a bump allocator whose addresses grow monotonically and are never reused; the sort
// then visits later-created nodes first, deterministically - the rule this repository's oracle and kernels fix (DESIGN.md,
// section 3, decision 1).  Build with -DSYSTEM_MALLOC to see the platform allocator's behaviour instead.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#ifndef SYSTEM_MALLOC
#include <sys/mman.h>
namespace {
const size_t kArenaBytes = (size_t)96 << 30;   // virtual reservation, committed lazily (MAP_NORESERVE)
char *g_arena = NULL;
std::atomic<size_t> g_arena_off(0);
void *bump_alloc(size_t n, size_t align) {
    if (!g_arena) {   // first call happens during static initialisation, single-threaded
        void *p = mmap(NULL, kArenaBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { fprintf(stderr, "bump allocator: mmap failed\n"); abort(); }
        g_arena = (char *)p;
    }
    if (align < 16) align = 16;
    const size_t need = (n + align - 1) / align * align + align;
    const size_t off = g_arena_off.fetch_add(need);
    if (off + need > kArenaBytes) { fprintf(stderr, "bump allocator: arena exhausted\n"); abort(); }
    return (void *)(((uintptr_t)(g_arena + off) + align - 1) / align * align);
}
}  // namespace
void *operator new(size_t n) { return bump_alloc(n, 16); }
void *operator new[](size_t n) { return bump_alloc(n, 16); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { return bump_alloc(n, 16); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return bump_alloc(n, 16); }
void operator delete(void *) noexcept {}
void operator delete[](void *) noexcept {}
void operator delete(void *, size_t) noexcept {}
void operator delete[](void *, size_t) noexcept {}
#endif

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "ORBextractor.h"   // the reference's include/ORBextractor.h
#ifdef WITH_FRAME
#include "Frame.h"          // the reference's include/Frame.h (pulls in Eigen, DBoW2, ...)
#endif

#include "refvec_io.h"

namespace {

const int kLevels = 8, kIniTh = 20, kMinTh = 7, kEdge = 19;
const float kScale = 1.2f, kFx = 718.856f, kFy = 718.856f, kCx = 607.1928f, kCy = 185.2157f, kBf = 386.1448f;

// protected members of the reference class, reached the way C++ allows: from a derived class
struct Probe : public ORB_SLAM2::ORBextractor {
    Probe(int nf) : ORB_SLAM2::ORBextractor(nf, kScale, kLevels, kIniTh, kMinTh) {}
    void pyramid(const cv::Mat &img) { ComputePyramid(img); }
    void level_keypoints(std::vector<std::vector<cv::KeyPoint> > &all) { ComputeKeyPointsOctTree(all); }
    std::vector<cv::KeyPoint> distribute(const std::vector<cv::KeyPoint> &keys, int minX, int maxX, int minY, int maxY, int N, int level) {
        return DistributeOctTree(keys, minX, maxX, minY, maxY, N, level);
    }
    const std::vector<int> &features_per_level() const { return mnFeaturesPerLevel; }
    const std::vector<int> &umax_table() const { return umax; }
};

void put_xyr(refvec::Writer &w, const std::string &name, const std::vector<cv::KeyPoint> &k, int addx, int addy) {
    std::vector<int32_t> v(3 * k.size());
    for (size_t i = 0; i < k.size(); i++) {
        v[3 * i] = (int32_t)k[i].pt.x + addx;      // integral at this point of the pipeline
        v[3 * i + 1] = (int32_t)k[i].pt.y + addy;
        v[3 * i + 2] = (int32_t)k[i].response;
    }
    w.put2d(name, refvec::I32, k.size(), 3, v.empty() ? NULL : &v[0]);
}

void put_keypoints28(refvec::Writer &w, const std::string &name, const std::vector<cv::KeyPoint> &k) {
    // the 28-byte layout the reference serialises (include/BoostArchiver.h:47-57): x y size angle response octave class_id
    std::vector<uint8_t> v(28 * k.size());
    for (size_t i = 0; i < k.size(); i++) {
        float f[5] = {k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response};
        int32_t n[2] = {k[i].octave, k[i].class_id};
        memcpy(&v[28 * i], f, 20);
        memcpy(&v[28 * i + 20], n, 8);
    }
    w.put2d(name, refvec::U8, k.size(), 28, v.empty() ? NULL : &v[0]);
}

void put_image(refvec::Writer &w, const std::string &name, const cv::Mat &m) {
    cv::Mat c = m.isContinuous() ? m : m.clone();
    w.put2d(name, refvec::U8, c.rows, c.cols, c.data);
}

double crc_of(const cv::Mat &m) { return (double)refvec::crc32_rows(m.data, m.rows, m.cols, m.step); }

bool dump_case(const std::string &in_dir, const std::string &out_dir, const std::string &name, int w, int h, int nf, int stereo, int full) {
    std::vector<uint8_t> pl, pr;
    int iw = 0, ih = 0;
    if (!refvec::read_pgm((in_dir + "/" + name + ".pgm").c_str(), pl, iw, ih) || iw != w || ih != h) {
        fprintf(stderr, "%s: cannot read %dx%d left image\n", name.c_str(), w, h);
        return false;
    }
    cv::Mat left(h, w, CV_8UC1, &pl[0]), right;
    if (stereo) {
        if (!refvec::read_pgm((in_dir + "/" + name + "_right.pgm").c_str(), pr, iw, ih) || iw != w || ih != h) {
            fprintf(stderr, "%s: cannot read right image\n", name.c_str());
            return false;
        }
        right = cv::Mat(h, w, CV_8UC1, &pr[0]);
    }
    refvec::Writer out((out_dir + "/ref_" + name + ".orbvec").c_str());
    if (!out.ok()) { fprintf(stderr, "cannot write into %s\n", out_dir.c_str()); return false; }
    const std::string p = name + "/";
    const int32_t meta[8] = {w, h, nf, kLevels, kIniTh, kMinTh, stereo, full};
    const float meta_f[3] = {kScale, kFx, kBf};
    out.put1d(p + "meta", refvec::I32, 8, meta);
    out.put1d(p + "meta_f", refvec::F32, 3, meta_f);
    {
        std::ostringstream info;
        info << "producer=reference opencv=" << CV_VERSION
#if CV_MAJOR_VERSION >= 3
             << " ipp=" << (cv::ipp::useIPP() ? 1 : 0) << " optimized=" << (cv::useOptimized() ? 1 : 0)
#endif
#ifdef SYSTEM_MALLOC
             << " allocator=system"
#else
             << " allocator=bump"
#endif
#ifdef WITH_FRAME
             << " frame=1"
#else
             << " frame=0"
#endif
             << " compiler=" << __VERSION__;
        out.put_text(p + "info", info.str());
    }
    {
        std::vector<double> c(1, crc_of(left));
        if (stereo) c.push_back(crc_of(right));
        out.put1d(p + "image_crc", refvec::F64, c.size(), &c[0]);
    }
    Probe ex(nf);
    {
        std::vector<float> sf = ex.GetScaleFactors();
        out.put1d(p + "scale_factors", refvec::F32, sf.size(), &sf[0]);
        std::vector<int32_t> fl(ex.features_per_level().begin(), ex.features_per_level().end()), um(ex.umax_table().begin(), ex.umax_table().end());
        out.put1d(p + "features_per_level", refvec::I32, fl.size(), &fl[0]);
        out.put1d(p + "umax", refvec::I32, um.size(), &um[0]);
    }
    // ---- a2 / a8 / a4 / a5-a6: per level, on the pyramid ComputePyramid builds (src/ORBextractor.cc:1107-1132)
    ex.pyramid(left);
    for (int l = 0; l < kLevels; l++) {
        std::ostringstream qs;
        qs << p << "L" << l << "/";
        const std::string q = qs.str();
        const cv::Mat &lvl = ex.mvImagePyramid[l];
        cv::Mat padded = lvl.clone();   // placeholder type; replaced by the ROI's parent below
        {   // the level is the inner ROI of a buffer with a 19-px BORDER_REFLECT_101 frame (:1113-1115): walk out to it
            cv::Mat roi = lvl;
            roi.adjustROI(kEdge, kEdge, kEdge, kEdge);
            padded = roi;
        }
        cv::Mat blur = lvl.clone();
        cv::GaussianBlur(blur, blur, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);   // :1085-1086
        const int32_t size[2] = {lvl.cols, lvl.rows};
        out.put1d(q + "size", refvec::I32, 2, size);
        const double crc[3] = {crc_of(lvl), crc_of(padded), crc_of(blur)};
        out.put1d(q + "crc", refvec::F64, 3, crc);
        if (full) {
            put_image(out, q + "pyramid", lvl);
            put_image(out, q + "padded", padded);
            put_image(out, q + "blur", blur);
        }
        // cv::FAST on the level's whole FAST region [minBorder, maxBorder) (:773-776) at both thresholds: pins the primitive
        // (segment test, cornerScore, 3x3 NMS, 3-px frame, row-major order) without the cell loop
        const int minB = kEdge - 3, maxX = lvl.cols - kEdge + 3, maxY = lvl.rows - kEdge + 3;
        const cv::Mat sub = lvl.rowRange(minB, maxY).colRange(minB, maxX);
        std::vector<cv::KeyPoint> f20, f7;
        cv::FAST(sub, f20, kIniTh, true);
        cv::FAST(sub, f7, kMinTh, true);
        put_xyr(out, q + "fast20", f20, 0, 0);
        put_xyr(out, q + "fast7", f7, 0, 0);
        // DistributeOctTree called directly on those candidates (coordinates relative to minBorder, as in :822-833)
        std::vector<cv::KeyPoint> kept = ex.distribute(f7, minB, maxX, minB, maxY, ex.features_per_level()[l], l);
        put_xyr(out, q + "octree_direct", kept, 0, 0);
    }
    // ---- a3 / a7: what ComputeKeyPointsOctTree keeps per level (cell loop, threshold fallback, quad-tree), level coordinates
    {
        std::vector<std::vector<cv::KeyPoint> > all;
        ex.level_keypoints(all);
        for (int l = 0; l < kLevels; l++) {
            std::ostringstream qs;
            qs << p << "L" << l << "/";
            put_xyr(out, qs.str() + "keypoints", all[l], 0, 0);
            std::vector<float> ang(all[l].size());
            for (size_t i = 0; i < ang.size(); i++) ang[i] = all[l][i].angle;
            out.put1d(qs.str() + "angles", refvec::F32, ang.size(), ang.empty() ? NULL : &ang[0]);
        }
    }
    // ---- a9 / a10 / a18: operator() (:1043-1105)
    std::vector<cv::KeyPoint> kl, kr;
    cv::Mat dl, dr;
    {
        ORB_SLAM2::ORBextractor exl(nf, kScale, kLevels, kIniTh, kMinTh);
        exl(left, cv::Mat(), kl, dl);
        put_keypoints28(out, p + "keypoints", kl);
        put_image(out, p + "descriptors", dl);
    }
    if (stereo) {
        ORB_SLAM2::ORBextractor exr(nf, kScale, kLevels, kIniTh, kMinTh);
        exr(right, cv::Mat(), kr, dr);
        put_keypoints28(out, p + "keypoints_right", kr);
        put_image(out, p + "descriptors_right", dr);
#ifdef WITH_FRAME
        // ---- a17: Frame::ComputeStereoMatches through the stereo Frame constructor (src/Frame.cc:61-120, 481-655)
        ORB_SLAM2::ORBextractor fl(nf, kScale, kLevels, kIniTh, kMinTh), fr(nf, kScale, kLevels, kIniTh, kMinTh);
        cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
        K.at<float>(0, 0) = kFx; K.at<float>(1, 1) = kFy; K.at<float>(0, 2) = kCx; K.at<float>(1, 2) = kCy;
        cv::Mat dist = cv::Mat::zeros(4, 1, CV_32F);
        ORB_SLAM2::Frame::mbInitialComputations = true;    // image bounds / fx for THIS image size (:100-117)
        ORB_SLAM2::Frame F(left, right, 0.0, &fl, &fr, static_cast<ORB_SLAM2::ORBVocabulary *>(NULL), K, dist, kBf, 35.f * kBf / kFx);
        out.put1d(p + "mvuRight", refvec::F32, F.mvuRight.size(), F.mvuRight.empty() ? NULL : &F.mvuRight[0]);
        out.put1d(p + "mvDepth", refvec::F32, F.mvDepth.size(), F.mvDepth.empty() ? NULL : &F.mvDepth[0]);
        put_keypoints28(out, p + "frame_keypoints", F.mvKeys);   // must equal "keypoints" (same extractor, same image)
#endif
    }
    printf("%s: %zu keypoints%s\n", name.c_str(), kl.size(), stereo ? " (stereo)" : "");
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s INPUT_DIR OUTPUT_DIR      (INPUT_DIR from tools/refvec/write_refvec_inputs.py)\n", argv[0]);
        return 2;
    }
    const std::string in_dir = argv[1], out_dir = argv[2];
    FILE *mf = fopen((in_dir + "/manifest.txt").c_str(), "r");
    if (!mf) { fprintf(stderr, "no manifest.txt in %s\n", in_dir.c_str()); return 2; }
    char name[256];
    int w, h, nf, stereo, full, bad = 0;
    while (fscanf(mf, "%255s %d %d %d %d %d", name, &w, &h, &nf, &stereo, &full) == 6)
        if (!dump_case(in_dir, out_dir, name, w, h, nf, stereo, full)) bad++;
    fclose(mf);
    return bad ? 1 : 0;
}
