// host_driver.cc — exercises the C++ host classes (ORB_SLAM2::ORBextractor / ORBmatcher /
// ComputeStereoMatchesHIP) the way Frame.cc / Tracking.cc call them; pytest feeds it raw
// files and compares its outputs with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
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

// ---- KeyFrame-side matchers (SearchByProjection(KF,Scw) / Fuse x2 / SearchBySim3) ----
struct Mp3d { int valid; float wx, wy, wz, nx, ny, nz, maxd, mind; };
static void fill_keyframe(KeyFrame &KF, ORBextractor *ex, const cv::Mat &im, const float *bounds, const float *cam) {
    std::vector<cv::KeyPoint> keys;
    (*ex)(im, cv::Mat(), keys, KF.mDescriptors);
    KF.mvKeysUn = keys; KF.N = (int)keys.size();
    KF.mvuRight.assign(KF.N, -1.f);
    KF.mvpMapPoints.assign(KF.N, (MapPoint *)NULL);
    KF.mvScaleFactors = ex->GetScaleFactors(); KF.mvInvLevelSigma2 = ex->GetInverseScaleSigmaSquares();
    KF.mnScaleLevels = ex->GetLevels(); KF.mfLogScaleFactor = log(ex->GetScaleFactor());   // src/Frame.cc:69-71
    Frame::mnMinX = bounds[0]; Frame::mnMinY = bounds[1]; Frame::mnMaxX = bounds[2]; Frame::mnMaxY = bounds[3];
    Frame::mfGridElementWidthInv = 64.f / (Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = 48.f / (Frame::mnMaxY - Frame::mnMinY);
    KF.mnMinX = Frame::mnMinX; KF.mnMinY = Frame::mnMinY; KF.mnMaxX = Frame::mnMaxX; KF.mnMaxY = Frame::mnMaxY;  // float -> int (src/KeyFrame.cc:41)
    KF.mfGridElementWidthInv = Frame::mfGridElementWidthInv; KF.mfGridElementHeightInv = Frame::mfGridElementHeightInv;
    KF.fx = cam[0]; KF.fy = cam[1]; KF.cx = cam[2]; KF.cy = cam[3]; KF.mbf = cam[4];
}
static void fill_points(std::vector<MapPoint> &pts, const Mp3d *mp, const unsigned char *desc, int m) {
    pts.resize(m);
    for (int i = 0; i < m; i++) {
        pts[i].mWorldPos.at<float>(0) = mp[i].wx; pts[i].mWorldPos.at<float>(1) = mp[i].wy; pts[i].mWorldPos.at<float>(2) = mp[i].wz;
        pts[i].mNormalVector.at<float>(0) = mp[i].nx; pts[i].mNormalVector.at<float>(1) = mp[i].ny; pts[i].mNormalVector.at<float>(2) = mp[i].nz;
        pts[i].mfMaxDistance = mp[i].maxd; pts[i].mfMinDistance = mp[i].mind;
        pts[i].mDescriptor = cv::Mat(1, 32, CV_8U);
        memcpy(pts[i].mDescriptor.ptr(0), desc + (size_t)32 * i, 32);
    }
}
static cv::Mat mat4(const float *v) {
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.at<float>(r, c) = v[r * 4 + c];
    return T;
}

static int kf_modes(int argc, char **argv) {
    // argv: kf <sub> img w h nf minx,miny,maxx,maxy fx,fy,cx,cy,mbf pts.bin pdesc.bin aux.bin th out
    if (argc != 14) return 2;
    const std::string sub = argv[2];
    const int w = atoi(argv[4]), h = atoi(argv[5]), nf = atoi(argv[6]);
    float bounds[4], cam[5];
    sscanf(argv[7], "%f,%f,%f,%f", &bounds[0], &bounds[1], &bounds[2], &bounds[3]);
    sscanf(argv[8], "%f,%f,%f,%f,%f", &cam[0], &cam[1], &cam[2], &cam[3], &cam[4]);
    std::vector<unsigned char> img = slurp(argv[3]), praw = slurp(argv[9]), draw = slurp(argv[10]), aux = slurp(argv[11]);
    const float th = (float)atof(argv[12]);
    const std::string out = argv[13];
    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    if (!ex.ok()) return 3;
    KeyFrame KF;
    fill_keyframe(KF, &ex, cv::Mat(h, w, CV_8UC1, img.data()), bounds, cam);
    const int n = KF.N, m = (int)(praw.size() / sizeof(Mp3d));
    const Mp3d *mp = (const Mp3d *)praw.data();
    std::vector<MapPoint> pts;
    fill_points(pts, mp, draw.data(), m);
    std::vector<MapPoint> ext(n);                        // holders that are not in the list
    std::vector<MapPoint *> vp(m);
    for (int i = 0; i < m; i++) vp[i] = &pts[i];
    const float *fa = (const float *)aux.data();
    ORBmatcher matcher(0.8f, true);
    std::vector<int> res;
    int ret = 0;
    if (sub == "projsim3") {
        // aux: Scw[16] f32, matched[n] i32 (-1 / -2)
        if (aux.size() != 64 + 4 * (size_t)n) { fprintf(stderr, "aux size: n=%d\n", n); return 6; }
        const int *matched0 = (const int *)(fa + 16);
        std::vector<MapPoint *> vpMatched(n, (MapPoint *)NULL);
        for (int j = 0; j < n; j++) if (matched0[j] == -2) vpMatched[j] = &ext[j];
        for (int i = 0; i < m; i++) if (!mp[i].valid) pts[i].mbBad = true;
        ret = matcher.SearchByProjection(&KF, mat4(fa), vp, vpMatched, (int)th);
        res.assign(n, -1);
        for (int j = 0; j < n; j++)
            if (vpMatched[j]) res[j] = (vpMatched[j] >= &pts[0] && vpMatched[j] < &pts[0] + m) ? (int)(vpMatched[j] - &pts[0]) : -2;
    } else if (sub == "fuse" || sub == "fusesim3") {
        // aux: T[16] f32, uright[n] f32, slot[n], ext_obs[n], ext_bad[n], bad[m], obs[m], null[m] i32
        if (aux.size() != 64 + 4 * (size_t)(4 * n + 3 * m)) { fprintf(stderr, "aux size: n=%d\n", n); return 6; }
        const float *ur = fa + 16;
        const int *slot = (const int *)(ur + n), *eobs = slot + n, *ebad = eobs + n, *bad = ebad + n, *obs = bad + m, *isnull = obs + m;
        for (int j = 0; j < n; j++) KF.mvuRight[j] = ur[j];
        for (int i = 0; i < m; i++) { pts[i].mbBad = bad[i] != 0; pts[i].nObs = obs[i]; }
        for (int j = 0; j < n; j++) {
            if (slot[j] >= 0) { KF.mvpMapPoints[j] = &pts[slot[j]]; pts[slot[j]].mObservations[&KF] = j; }
            else if (slot[j] == -2) { KF.mvpMapPoints[j] = &ext[j]; ext[j].nObs = eobs[j]; ext[j].mbBad = ebad[j] != 0; ext[j].mObservations[&KF] = j; }
        }
        std::vector<MapPoint *> vpReplace(m, (MapPoint *)NULL);
        if (sub == "fuse") {
            for (int i = 0; i < m; i++) if (isnull[i]) vp[i] = NULL;
            KF.SetPose(mat4(fa));
            ret = matcher.Fuse(&KF, vp, th);
        } else {
            ret = matcher.Fuse(&KF, mat4(fa), vp, th, vpReplace);
        }
        // out: slot[n], ext_bad[n], bad[m], in_kf[m], obs[m], replace[m]
        for (int j = 0; j < n; j++) {
            MapPoint *p = KF.mvpMapPoints[j];
            res.push_back(!p ? -1 : (p >= &pts[0] && p < &pts[0] + m) ? (int)(p - &pts[0]) : -2);
        }
        for (int j = 0; j < n; j++) res.push_back(ext[j].mbBad ? 1 : 0);
        for (int i = 0; i < m; i++) res.push_back(pts[i].mbBad ? 1 : 0);
        for (int i = 0; i < m; i++) res.push_back(pts[i].IsInKeyFrame(&KF) ? 1 : 0);
        for (int i = 0; i < m; i++) res.push_back(pts[i].nObs);
        for (int i = 0; i < m; i++) {
            MapPoint *p = vpReplace[i];
            res.push_back(!p ? -1 : (p >= &pts[0] && p < &pts[0] + m) ? (int)(p - &pts[0]) : -2);
        }
    } else return 2;
    dump(out + ".i32", res.data(), res.size() * 4);
    printf("%d %d\n", n, ret);
    return 0;
}

static int sim3_mode(int argc, char **argv) {
    // argv: sim3 img1 img2 w h nf bounds cam pts1.bin pd1.bin pts2.bin pd2.bin aux.bin th out
    // aux: T1w[16], T2w[16], s12, R12[9], t12[3] f32; pre12[n1] i32 (-1 or index of a KF2 slot)
    if (argc != 16) return 2;
    const int w = atoi(argv[4]), h = atoi(argv[5]), nf = atoi(argv[6]);
    float bounds[4], cam[5];
    sscanf(argv[7], "%f,%f,%f,%f", &bounds[0], &bounds[1], &bounds[2], &bounds[3]);
    sscanf(argv[8], "%f,%f,%f,%f,%f", &cam[0], &cam[1], &cam[2], &cam[3], &cam[4]);
    std::vector<unsigned char> i1 = slurp(argv[2]), i2 = slurp(argv[3]), p1 = slurp(argv[9]), d1 = slurp(argv[10]),
                               p2 = slurp(argv[11]), d2 = slurp(argv[12]), aux = slurp(argv[13]);
    const float th = (float)atof(argv[14]);
    const std::string out = argv[15];
    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    if (!ex.ok()) return 3;
    KeyFrame KF1, KF2;
    fill_keyframe(KF1, &ex, cv::Mat(h, w, CV_8UC1, i1.data()), bounds, cam);
    fill_keyframe(KF2, &ex, cv::Mat(h, w, CV_8UC1, i2.data()), bounds, cam);
    const int n1 = KF1.N, n2 = KF2.N;
    if ((int)(p1.size() / sizeof(Mp3d)) != n1 || (int)(p2.size() / sizeof(Mp3d)) != n2) { fprintf(stderr, "n1=%d n2=%d\n", n1, n2); return 6; }
    std::vector<MapPoint> pts1, pts2;
    fill_points(pts1, (const Mp3d *)p1.data(), d1.data(), n1);
    fill_points(pts2, (const Mp3d *)p2.data(), d2.data(), n2);
    const Mp3d *m1 = (const Mp3d *)p1.data(), *m2 = (const Mp3d *)p2.data();
    // valid: 1 = a good point, 0 = no point in the slot, 2 = a bad point
    for (int i = 0; i < n1; i++) if (m1[i].valid) { KF1.mvpMapPoints[i] = &pts1[i]; pts1[i].mObservations[&KF1] = i; pts1[i].mbBad = m1[i].valid == 2; }
    for (int i = 0; i < n2; i++) if (m2[i].valid) { KF2.mvpMapPoints[i] = &pts2[i]; pts2[i].mObservations[&KF2] = i; pts2[i].mbBad = m2[i].valid == 2; }
    const float *fa = (const float *)aux.data();
    KF1.SetPose(mat4(fa)); KF2.SetPose(mat4(fa + 16));
    const float s12 = fa[32];
    cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R12.at<float>(r, c) = fa[33 + r * 3 + c]; t12.at<float>(r) = fa[42 + r]; }
    const int *pre = (const int *)(fa + 45);
    std::vector<MapPoint *> vpMatches12(n1, (MapPoint *)NULL);
    for (int i = 0; i < n1; i++) if (pre[i] >= 0) vpMatches12[i] = &pts2[pre[i]];
    ORBmatcher matcher(0.75f, true);  // src/LoopClosing.cc
    const int ret = matcher.SearchBySim3(&KF1, &KF2, vpMatches12, s12, R12, t12, th);
    std::vector<int> res(n1, -1);
    for (int i = 0; i < n1; i++) if (vpMatches12[i]) res[i] = (int)(vpMatches12[i] - &pts2[0]);
    dump(out + ".i32", res.data(), res.size() * 4);
    printf("%d %d %d\n", n1, n2, ret);
    return 0;
}

static int distinct_mode(int argc, char **argv) {
    // argv: distinct offsets.bin(int32[m+1]) desc.bin(u8[total][32]) badkf out
    // point p is observed by keyframes 0..N_p-1 at index p; keyframe `badkf` is bad (its rows are skipped, :275)
    if (argc != 6) return 2;
    std::vector<unsigned char> oraw = slurp(argv[2]), draw = slurp(argv[3]);
    const int badkf = atoi(argv[4]);
    const std::string out = argv[5];
    const int m = (int)(oraw.size() / 4) - 1;
    const int *off = (const int *)oraw.data();
    int K = 0;
    for (int p = 0; p < m; p++) K = std::max(K, off[p + 1] - off[p]);
    std::vector<KeyFrame> kfs(K);            // contiguous: std::map<KeyFrame*,...> iterates them in index order
    for (int k = 0; k < K; k++) { kfs[k].mDescriptors = cv::Mat::zeros(m, 32, CV_8U); kfs[k].mvuRight.assign(m, -1.f); kfs[k].mbBad = k == badkf; }
    std::vector<MapPoint> pts(m);
    std::vector<MapPoint *> vp(m);
    for (int p = 0; p < m; p++) {
        for (int k = 0; k < off[p + 1] - off[p]; k++) {
            memcpy(kfs[k].mDescriptors.ptr(p), &draw[(size_t)32 * (off[p] + k)], 32);
            pts[p].AddObservation(&kfs[k], p);
        }
        vp[p] = &pts[p];
        if (p % 11 == 10) pts[p].mbBad = true;   // bad points return early (:260-262)
    }
    std::vector<cv::Mat> best;
    const int n = ComputeDistinctiveDescriptorsHIP(vp, best);
    std::vector<unsigned char> res((size_t)33 * m, 0);
    for (int p = 0; p < m; p++)
        if (!best[p].empty()) { res[(size_t)33 * p] = 1; memcpy(&res[(size_t)33 * p + 1], best[p].ptr(0), 32); }
    dump(out + ".best", res.data(), res.size());
    printf("%d %d\n", m, n);
    return n < 0 ? 5 : 0;
}

static int local_mode(int argc, char **argv) {
    // argv: local img w h nf fx,fy,cx,cy,mbf pts.bin(Mp3d[m]) pdesc.bin aux.bin th out
    // aux: Tcw[16] f32; uright[n] f32; holder[n] i32 (-1 / -2 / list index); ext_obs[n] i32; obs[m] i32; seen[m] i32
    if (argc != 12) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]), nf = atoi(argv[5]);
    float cam[5];
    sscanf(argv[6], "%f,%f,%f,%f,%f", &cam[0], &cam[1], &cam[2], &cam[3], &cam[4]);
    std::vector<unsigned char> img = slurp(argv[2]), praw = slurp(argv[7]), draw = slurp(argv[8]), aux = slurp(argv[9]);
    const float th = (float)atof(argv[10]);
    const std::string out = argv[11];
    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    if (!ex.ok()) return 3;
    Frame F;
    fill_frame(F, &ex, cv::Mat(h, w, CV_8UC1, img.data()), w, h);
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
    F.mbf = cam[4]; F.mb = F.mbf / Frame::fx;
    F.mnScaleLevels = ex.GetLevels(); F.mfScaleFactor = ex.GetScaleFactor(); F.mfLogScaleFactor = log(F.mfScaleFactor);
    F.mnId = 7;
    const int n = F.N, m = (int)(praw.size() / sizeof(Mp3d));
    if (aux.size() != 64 + 4 * (size_t)(3 * n + 2 * m)) { fprintf(stderr, "aux size: n=%d m=%d\n", n, m); return 6; }
    const float *fa = (const float *)aux.data();
    F.mTcw = mat4(fa);
    const float *ur = fa + 16;
    const int *holder = (const int *)(ur + n), *eobs = holder + n, *obs = eobs + n, *seen = obs + m;
    for (int j = 0; j < n; j++) F.mvuRight[j] = ur[j];
    const Mp3d *mp = (const Mp3d *)praw.data();
    std::vector<MapPoint> pts, ext(n);
    fill_points(pts, mp, draw.data(), m);
    std::vector<MapPoint *> vp(m);
    for (int i = 0; i < m; i++) {
        vp[i] = &pts[i];
        pts[i].nObs = obs[i];
        pts[i].mbBad = mp[i].valid == 0 && !seen[i];      // invalid = bad, or already seen in this frame
        pts[i].mnLastFrameSeen = seen[i] ? F.mnId : 3;
        pts[i].mbTrackInView = (i % 5) == 0;              // stale flags from an earlier frame must not matter
    }
    for (int j = 0; j < n; j++) {
        if (holder[j] >= 0) F.mvpMapPoints[j] = &pts[holder[j]];
        else if (holder[j] == -2) { F.mvpMapPoints[j] = &ext[j]; ext[j].nObs = eobs[j]; }
    }
    int nToMatch = 0;
    const int ret = SearchLocalPointsHIP(F, vp, th, 0.8f, nToMatch);
    std::vector<int> res;
    for (int j = 0; j < n; j++) {
        MapPoint *p = F.mvpMapPoints[j];
        res.push_back(!p ? -1 : (p >= &pts[0] && p < &pts[0] + m) ? (int)(p - &pts[0]) : -2);
    }
    for (int i = 0; i < m; i++) res.push_back(pts[i].mnVisible);
    dump(out + ".i32", res.data(), res.size() * 4);
    printf("%d %d %d\n", n, ret, nToMatch);
    return 0;
}

static int bow_mode(int argc, char **argv) {
    // argv: bow voc.txt d1.bin(u8[n1][32]) a1.bin(f32[n1]) v1.bin(u8[n1]: 0 no point, 1 good, 2 bad) d2.bin a2.bin v2.bin ratio out
    // KeyFrame 1 = (d1, a1, v1); Frame / KeyFrame 2 = (d2, a2, v2).  Writes BowVec / FeatVec of both and the two SearchByBoW results.
    if (argc != 11) return 2;
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[2])) return 3;
    std::vector<unsigned char> d1 = slurp(argv[3]), a1 = slurp(argv[4]), v1 = slurp(argv[5]), d2 = slurp(argv[6]), a2 = slurp(argv[7]), v2 = slurp(argv[8]);
    const float ratio = (float)atof(argv[9]);
    const std::string out = argv[10];
    const int n1 = (int)v1.size(), n2 = (int)v2.size();
    KeyFrame K1, K2;
    Frame F;
    std::vector<MapPoint> p1(n1), p2(n2);
    K1.mpORBvocabulary = K2.mpORBvocabulary = F.mpORBvocabulary = &voc;
    K1.mDescriptors = cv::Mat(n1, 32, CV_8U, d1.data()).clone(); K2.mDescriptors = cv::Mat(n2, 32, CV_8U, d2.data()).clone();
    F.mDescriptors = K2.mDescriptors.clone(); F.N = n2;
    K1.mvKeysUn.resize(n1); K1.mvpMapPoints.assign(n1, (MapPoint *)NULL);
    K2.mvKeysUn.resize(n2); K2.mvpMapPoints.assign(n2, (MapPoint *)NULL);
    F.mvKeys.resize(n2);
    for (int i = 0; i < n1; i++) { K1.mvKeysUn[i].angle = ((const float *)a1.data())[i]; if (v1[i]) { K1.mvpMapPoints[i] = &p1[i]; p1[i].mbBad = v1[i] == 2; } }
    for (int i = 0; i < n2; i++) { K2.mvKeysUn[i].angle = F.mvKeys[i].angle = ((const float *)a2.data())[i]; if (v2[i]) { K2.mvpMapPoints[i] = &p2[i]; p2[i].mbBad = v2[i] == 2; } }
    K1.ComputeBoW(); K2.ComputeBoW(); F.ComputeBoW();
    {   // BowVec (word, value) pairs and FeatVec (node, count, items...) of keyframe 1
        std::vector<double> bow;
        for (DBoW2::BowVector::const_iterator it = K1.mBowVec.begin(); it != K1.mBowVec.end(); ++it) { bow.push_back((double)it->first); bow.push_back(it->second); }
        dump(out + ".bow", bow.data(), bow.size() * 8);
        std::vector<int> fv;
        for (DBoW2::FeatureVector::const_iterator it = K1.mFeatVec.begin(); it != K1.mFeatVec.end(); ++it) {
            fv.push_back((int)it->first); fv.push_back((int)it->second.size());
            for (size_t j = 0; j < it->second.size(); j++) fv.push_back((int)it->second[j]);
        }
        dump(out + ".fv", fv.data(), fv.size() * 4);
    }
    ORBmatcher matcher(ratio, true);
    std::vector<MapPoint *> mF, m12;
    const int nA = matcher.SearchByBoW(&K1, F, mF);
    const int nB = matcher.SearchByBoW(&K1, &K2, m12);
    std::vector<int> res;
    for (int i = 0; i < n2; i++) res.push_back(mF[i] ? (int)(mF[i] - &p1[0]) : -1);       // F feature -> KF1 point
    for (int i = 0; i < n1; i++) res.push_back(m12[i] ? (int)(m12[i] - &p2[0]) : -1);     // KF1 feature -> KF2 point
    dump(out + ".i32", res.data(), res.size() * 4);
    printf("%d %d %d %d\n", n1, n2, nA, nB);
    return 0;
}

static int tri_mode(int argc, char **argv) {
    // argv: tri voc.txt d1.bin k1.bin(orbx_keypoint_t[n1]) m1.bin(u8 has map point) ur1.bin(f32) d2.bin k2.bin m2.bin ur2.bin aux.bin onlyStereo out
    // aux: F12[9], T1w[16], T2w[16], fx, fy, cx, cy (f32)
    if (argc != 14) return 2;
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[2])) return 3;
    std::vector<unsigned char> d1 = slurp(argv[3]), k1 = slurp(argv[4]), m1 = slurp(argv[5]), u1 = slurp(argv[6]), d2 = slurp(argv[7]),
                               k2 = slurp(argv[8]), m2 = slurp(argv[9]), u2 = slurp(argv[10]), aux = slurp(argv[11]);
    const bool onlyStereo = atoi(argv[12]) != 0;
    const std::string out = argv[13];
    const int n1 = (int)m1.size(), n2 = (int)m2.size();
    const float *fa = (const float *)aux.data();
    KeyFrame K[2];
    std::vector<MapPoint> pts(n1 + n2);
    for (int s = 0; s < 2; s++) {
        const int n = s ? n2 : n1;
        const orbx_keypoint_t *kp = (const orbx_keypoint_t *)(s ? k2.data() : k1.data());
        K[s].mpORBvocabulary = &voc;
        K[s].N = n;
        K[s].mDescriptors = cv::Mat(n, 32, CV_8U, (s ? d2 : d1).data()).clone();
        K[s].mvKeysUn.resize(n); K[s].mvpMapPoints.assign(n, (MapPoint *)NULL);
        K[s].mvuRight.assign((const float *)(s ? u2 : u1).data(), (const float *)(s ? u2 : u1).data() + n);
        for (int i = 0; i < n; i++) {
            K[s].mvKeysUn[i].pt.x = kp[i].x; K[s].mvKeysUn[i].pt.y = kp[i].y; K[s].mvKeysUn[i].octave = kp[i].octave; K[s].mvKeysUn[i].angle = kp[i].angle;
            if ((s ? m2 : m1)[i]) K[s].mvpMapPoints[i] = &pts[s * n1 + i];
        }
        K[s].mvScaleFactors.resize(8); K[s].mvLevelSigma2.resize(8);
        K[s].mvScaleFactors[0] = 1.0f; K[s].mvLevelSigma2[0] = 1.0f;
        for (int l = 1; l < 8; l++) { K[s].mvScaleFactors[l] = K[s].mvScaleFactors[l - 1] * 1.2f; K[s].mvLevelSigma2[l] = K[s].mvScaleFactors[l] * K[s].mvScaleFactors[l]; }   // src/ORBextractor.cc:417-424
        K[s].fx = fa[41]; K[s].fy = fa[42]; K[s].cx = fa[43]; K[s].cy = fa[44];
        K[s].SetPose(mat4(fa + 9 + 16 * s));
        K[s].ComputeBoW();
    }
    cv::Mat F12(3, 3, CV_32F);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F12.at<float>(r, c) = fa[r * 3 + c];
    ORBmatcher matcher(0.6f, false);   // src/LocalMapping.cc:223
    std::vector<std::pair<size_t, size_t> > pairs;
    const int n = matcher.SearchForTriangulation(&K[0], &K[1], F12, pairs, onlyStereo);
    std::vector<int> res;
    for (size_t i = 0; i < pairs.size(); i++) { res.push_back((int)pairs[i].first); res.push_back((int)pairs[i].second); }
    dump(out + ".i32", res.data(), res.size() * 4);
    printf("%d %d %d\n", n1, n2, n);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "kf") return kf_modes(argc, argv);
    if (mode == "sim3") return sim3_mode(argc, argv);
    if (mode == "distinct") return distinct_mode(argc, argv);
    if (mode == "local") return local_mode(argc, argv);
    if (mode == "bow") return bow_mode(argc, argv);
    if (mode == "tri") return tri_mode(argc, argv);
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
    if (mode == "stereoframe" && argc == 10) {   // the stereo Frame constructor's feature part in one GPU call
        const int w = atoi(argv[4]), h = atoi(argv[5]), nf = atoi(argv[6]);
        const float mbf = (float)atof(argv[7]), fx = (float)atof(argv[8]);
        const std::string out = argv[9];
        std::vector<unsigned char> l = slurp(argv[2]), r = slurp(argv[3]);
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        if (!ex.ok()) return 3;
        Frame F;
        F.mpORBextractorLeft = &ex; F.mpORBextractorRight = &ex;
        Frame::fx = fx; F.mbf = mbf; F.mb = F.mbf / Frame::fx;  // src/Frame.cc:114
        int nm = 0;
        for (int rep = 0; rep < 2; rep++) nm = ExtractStereoFrameHIP(F, cv::Mat(h, w, CV_8UC1, l.data()), cv::Mat(h, w, CV_8UC1, r.data()));
        dump_kps(out + ".kps", F.mvKeys);
        dump(out + ".desc", F.mDescriptors.empty() ? NULL : F.mDescriptors.ptr(0), (size_t)F.mDescriptors.rows * 32);
        dump_kps(out + ".kpsr", F.mvKeysRight);
        dump(out + ".descr", F.mDescriptorsRight.empty() ? NULL : F.mDescriptorsRight.ptr(0), (size_t)F.mDescriptorsRight.rows * 32);
        dump(out + ".uright", F.mvuRight.data(), F.mvuRight.size() * 4);
        dump(out + ".depth", F.mvDepth.data(), F.mvDepth.size() * 4);
        printf("%d %d %d\n", F.N, (int)F.mvKeysRight.size(), nm);
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
    fprintf(stderr, "usage: host_driver extract|stereo|init|projmp|projkf|kf|sim3 ...\n");
    return 2;
}
