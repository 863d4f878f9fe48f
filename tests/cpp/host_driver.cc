// host_driver.cc — exercises the C++ host classes (ORB_SLAM2::ORBextractor / ORBmatcher /
// ComputeStereoMatchesHIP) the way Frame.cc / Tracking.cc call them; pytest feeds it raw
// files and compares its outputs with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;

static std::vector<unsigned char> slurp(const std::string &p) {
    std::vector<unsigned char> v;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (n && fread(v.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}
static void dump(const std::string &p, const void *d, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) exit(2);
    if (n) fwrite(d, 1, n, f);
    fclose(f);
}
static void dump_kps(const std::string &p, const std::vector<cv::KeyPoint> &k) {
    std::vector<orbx_keypoint_t> o(k.size());
    for (size_t i = 0; i < k.size(); i++) {
        o[i].x = k[i].pt.x; o[i].y = k[i].pt.y; o[i].size = k[i].size; o[i].angle = k[i].angle;
        o[i].response = k[i].response; o[i].octave = k[i].octave; o[i].class_id = k[i].class_id;
    }
    dump(p, o.data(), o.size() * sizeof(orbx_keypoint_t));
}
static void fill_frame(Frame &F, ORBextractor *ex, const cv::Mat &im, int w, int h) {
    (*ex)(im, cv::Mat(), F.mvKeys, F.mDescriptors);  // Frame::ExtractORB (src/Frame.cc:247-253)
    F.N = (int)F.mvKeys.size();
    F.mvKeysUn = F.mvKeys;  // no distortion (src/Frame.cc:421-425)
    F.mvuRight.assign(F.N, -1.f); F.mvDepth.assign(F.N, -1.f);
    F.mvpMapPoints.assign(F.N, (MapPoint *)NULL);
    F.mvbOutlier.assign(F.N, false);
    F.mvScaleFactors = ex->GetScaleFactors(); F.mvInvScaleFactors = ex->GetInverseScaleFactors();
    Frame::mnMinX = 0.f; Frame::mnMaxX = (float)w; Frame::mnMinY = 0.f; Frame::mnMaxY = (float)h;
    Frame::mfGridElementWidthInv = 64.f / (Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = 48.f / (Frame::mnMaxY - Frame::mnMinY);
    F.mTcw = cv::Mat::eye(4, 4, CV_32F);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "extract" && argc == 7) {
        const int w = atoi(argv[3]), h = atoi(argv[4]), nf = atoi(argv[5]);
        const std::string out = argv[6];
        std::vector<unsigned char> img = slurp(argv[2]);
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        if (!ex.ok()) { fprintf(stderr, "%s\n", ex.lastError()); return 3; }
        cv::Mat im(h, w, CV_8UC1, img.data());
        std::vector<cv::KeyPoint> k; cv::Mat d;
        ex(im, cv::Mat(), k, d);
        dump_kps(out + ".kps", k);
        dump(out + ".desc", d.empty() ? NULL : d.ptr(0), (size_t)d.rows * 32);
        // mvImagePyramid as a host cv::Mat (inner ROI of a bordered buffer): level 3, compact
        const cv::Mat &p = ex.mvImagePyramid[3];
        std::vector<unsigned char> lv((size_t)p.rows * p.cols);
        for (int r = 0; r < p.rows; r++) memcpy(&lv[(size_t)r * p.cols], p.ptr(r), p.cols);
        dump(out + ".pyr3", lv.data(), lv.size());
        printf("%d %d %d\n", (int)k.size(), p.cols, p.rows);
        // empty image: outputs untouched (:1046-1047)
        std::vector<cv::KeyPoint> k2(3); cv::Mat d2;
        ex(cv::Mat(), cv::Mat(), k2, d2);
        if (k2.size() != 3) return 4;
        return 0;
    }
    if (mode == "stereo" && argc == 10) {
        const int w = atoi(argv[4]), h = atoi(argv[5]), nf = atoi(argv[6]);
        const float mbf = (float)atof(argv[7]), fx = (float)atof(argv[8]);
        const std::string out = argv[9];
        std::vector<unsigned char> l = slurp(argv[2]), r = slurp(argv[3]);
        ORBextractor exl(nf, 1.2f, 8, 20, 7), exr(nf, 1.2f, 8, 20, 7);
        if (!exl.ok() || !exr.ok()) return 3;
        exl.SetMaterializePyramid(false); exr.SetMaterializePyramid(false);
        Frame F;
        F.mpORBextractorLeft = &exl; F.mpORBextractorRight = &exr;
        fill_frame(F, &exl, cv::Mat(h, w, CV_8UC1, l.data()), w, h);
        exr(cv::Mat(h, w, CV_8UC1, r.data()), cv::Mat(), F.mvKeysRight, F.mDescriptorsRight);
        Frame::fx = fx; F.mbf = mbf; F.mb = F.mbf / Frame::fx;  // src/Frame.cc:114
        const int nm = ComputeStereoMatchesHIP(F);
        dump(out + ".uright", F.mvuRight.data(), F.mvuRight.size() * 4);
        dump(out + ".depth", F.mvDepth.data(), F.mvDepth.size() * 4);
        printf("%d %d\n", F.N, nm);
        return nm < 0 ? 5 : 0;
    }
    if (mode == "init" && argc == 8) {
        const int w = atoi(argv[4]), h = atoi(argv[5]), nf = atoi(argv[6]);
        const std::string out = argv[7];
        std::vector<unsigned char> a = slurp(argv[2]), b = slurp(argv[3]);
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        if (!ex.ok()) return 3;
        Frame F1, F2;
        fill_frame(F1, &ex, cv::Mat(h, w, CV_8UC1, a.data()), w, h);
        fill_frame(F2, &ex, cv::Mat(h, w, CV_8UC1, b.data()), w, h);
        std::vector<cv::Point2f> prev(F1.mvKeysUn.size());
        for (size_t i = 0; i < prev.size(); i++) prev[i] = F1.mvKeysUn[i].pt;  // src/Tracking.cc:716-718
        std::vector<int> m12;
        ORBmatcher matcher(0.9f, true);  // src/Tracking.cc:742
        const int n = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
        dump(out + ".m12", m12.data(), m12.size() * 4);
        dump(out + ".prev", prev.data(), prev.size() * 8);
        printf("%d %d %d\n", F1.N, F2.N, n);
        return 0;
    }
    if (mode == "projmp" && argc == 9) {
        // argv: img w h nf mps.bin(orbm_mappoint_t[m]) mpdesc.bin out
        const int w = atoi(argv[3]), h = atoi(argv[4]), nf = atoi(argv[5]);
        const std::string out = argv[8];
        std::vector<unsigned char> a = slurp(argv[2]), mraw = slurp(argv[6]), draw = slurp(argv[7]);
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        if (!ex.ok()) return 3;
        Frame F;
        fill_frame(F, &ex, cv::Mat(h, w, CV_8UC1, a.data()), w, h);
        const int m = (int)(mraw.size() / sizeof(orbm_mappoint_t));
        const orbm_mappoint_t *mp = (const orbm_mappoint_t *)mraw.data();
        std::vector<MapPoint> pts(m);
        std::vector<MapPoint *> vp(m);
        for (int i = 0; i < m; i++) {
            pts[i].mbTrackInView = mp[i].in_view != 0;
            pts[i].mTrackProjX = mp[i].proj_x; pts[i].mTrackProjY = mp[i].proj_y; pts[i].mTrackProjXR = mp[i].proj_xr;
            pts[i].mnTrackScaleLevel = mp[i].level; pts[i].mTrackViewCos = mp[i].view_cos; pts[i].nObs = mp[i].observations;
            pts[i].mDescriptor = cv::Mat(1, 32, CV_8U);
            memcpy(pts[i].mDescriptor.ptr(0), &draw[(size_t)32 * i], 32);
            vp[i] = &pts[i];
        }
        ORBmatcher matcher(0.8f);  // src/Tracking.cc:1329
        const int n = matcher.SearchByProjection(F, vp, 3.0f);
        std::vector<int> held(F.N, -1);
        for (int i = 0; i < F.N; i++)
            if (F.mvpMapPoints[i]) held[i] = (int)(F.mvpMapPoints[i] - &pts[0]);
        dump(out + ".held", held.data(), held.size() * 4);
        printf("%d %d\n", F.N, n);
        return 0;
    }
    if (mode == "projkf" && argc == 13) {
        // argv: img w h nf kf.bin(oracle_kfpoint_t[m]) kfdesc.bin th orbdist fx,fy,cx,cy tx tz out
        const int w = atoi(argv[3]), h = atoi(argv[4]), nf = atoi(argv[5]);
        const float th = (float)atof(argv[8]);
        const int orbdist = atoi(argv[9]);
        float cam[4];
        sscanf(argv[10], "%f,%f,%f,%f", &cam[0], &cam[1], &cam[2], &cam[3]);
        const std::string out = argv[12];
        std::vector<unsigned char> a = slurp(argv[2]), kraw = slurp(argv[6]), draw = slurp(argv[7]);
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        if (!ex.ok()) return 3;
        Frame F;
        fill_frame(F, &ex, cv::Mat(h, w, CV_8UC1, a.data()), w, h);
        Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
        F.mnScaleLevels = ex.GetLevels(); F.mfScaleFactor = ex.GetScaleFactor(); F.mfLogScaleFactor = log(F.mfScaleFactor);
        float tx = 0, tz = 0;
        sscanf(argv[11], "%f,%f", &tx, &tz);
        F.mTcw.at<float>(0, 3) = tx; F.mTcw.at<float>(2, 3) = tz;
        struct KfPt { int valid; float wx, wy, wz, maxd, mind, angle; };
        const int m = (int)(kraw.size() / sizeof(KfPt));
        const KfPt *kp = (const KfPt *)kraw.data();
        std::vector<MapPoint> pts(m);
        KeyFrame KF;
        KF.mvKeysUn.resize(m); KF.mvpMapPoints.resize(m);
        std::set<MapPoint *> already;
        for (int i = 0; i < m; i++) {
            pts[i].mWorldPos.at<float>(0) = kp[i].wx; pts[i].mWorldPos.at<float>(1) = kp[i].wy; pts[i].mWorldPos.at<float>(2) = kp[i].wz;
            pts[i].mfMaxDistance = kp[i].maxd; pts[i].mfMinDistance = kp[i].mind;
            pts[i].mDescriptor = cv::Mat(1, 32, CV_8U);
            memcpy(pts[i].mDescriptor.ptr(0), &draw[(size_t)32 * i], 32);
            KF.mvKeysUn[i].angle = kp[i].angle;
            KF.mvpMapPoints[i] = &pts[i];
            // invalid points alternate between the three ways the reference skips a point
            if (!kp[i].valid) { if (i % 3 == 0) KF.mvpMapPoints[i] = NULL; else if (i % 3 == 1) pts[i].mbBad = true; else already.insert(&pts[i]); }
        }
        ORBmatcher matcher(0.9f, true);
        const int n = matcher.SearchByProjection(F, &KF, already, th, orbdist);
        std::vector<int> held(F.N, -1);
        for (int i = 0; i < F.N; i++)
            if (F.mvpMapPoints[i]) held[i] = (int)(F.mvpMapPoints[i] - &pts[0]);
        dump(out + ".held", held.data(), held.size() * 4);
        printf("%d %d\n", F.N, n);
        return 0;
    }
    fprintf(stderr, "usage: host_driver extract|stereo|init|projmp|projkf ...\n");
    return 2;
}
