// ORBmatcher.cc — host side of the hot matchers: gather the pointer graph (Frame, MapPoint*)
// into the flat arrays of the C ABI, call the HIP path, scatter the results back exactly where
// the reference writes them.  No descriptor is compared on the CPU.
#include "ORBmatcher.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;  // src/ORBmatcher.cc:37-39
int ORBmatcher::device = std::getenv("ORBX_DEVICE") ? std::atoi(std::getenv("ORBX_DEVICE")) : 0;

#ifndef ORBX_HAVE_ORBSLAM2
float Frame::fx, Frame::fy, Frame::cx, Frame::cy;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
#endif

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbm_hamming(a.ptr(0), b.ptr(0)); }

static void gather_keypoints(const std::vector<cv::KeyPoint> &v, std::vector<orbx_keypoint_t> &out) {
    out.resize(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        out[i].x = v[i].pt.x; out[i].y = v[i].pt.y; out[i].size = v[i].size; out[i].angle = v[i].angle;
        out[i].response = v[i].response; out[i].octave = v[i].octave; out[i].class_id = v[i].class_id;
    }
}
static void gather_descriptors(const cv::Mat &m, int n, std::vector<uint8_t> &out) {
    out.resize((size_t)32 * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) std::memcpy(&out[(size_t)32 * i], m.ptr(i), 32);
}
static orbm_grid_geom_t grid_of(const Frame &) {
    orbm_grid_geom_t g;
    g.min_x = Frame::mnMinX; g.min_y = Frame::mnMinY; g.max_x = Frame::mnMaxX; g.max_y = Frame::mnMaxY;
    g.inv_w = Frame::mfGridElementWidthInv; g.inv_h = Frame::mfGridElementHeightInv;
    return g;
}
static int fail(const char *what) { std::fprintf(stderr, "ORBmatcher::%s: %s\n", what, orbx_last_error()); return 0; }

int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th) {
    const int n = F.N, m = (int)vpMapPoints.size();
    if (n == 0 || m == 0) return 0;
    std::vector<orbx_keypoint_t> kun; gather_keypoints(F.mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(F.mDescriptors, n, desc);
    std::vector<orbm_mappoint_t> mps(m);
    std::vector<uint8_t> mpd((size_t)32 * m);
    std::map<MapPoint *, int> index;
    for (int i = 0; i < m; i++) {
        MapPoint *p = vpMapPoints[i];
        index[p] = i;
        mps[i].in_view = p->mbTrackInView && !p->isBad();
        mps[i].proj_x = p->mTrackProjX; mps[i].proj_y = p->mTrackProjY; mps[i].proj_xr = p->mTrackProjXR;
        mps[i].level = p->mnTrackScaleLevel; mps[i].view_cos = p->mTrackViewCos;
        mps[i].observations = p->Observations();
        if (mps[i].in_view) { cv::Mat d = p->GetDescriptor(); std::memcpy(&mpd[(size_t)32 * i], d.ptr(0), 32); }
    }
    std::vector<int32_t> holder(n, -1), ext(n, 0);
    for (int i = 0; i < n; i++)
        if (F.mvpMapPoints[i]) {
            std::map<MapPoint *, int>::iterator it = index.find(F.mvpMapPoints[i]);
            if (it != index.end()) holder[i] = it->second;
            else { holder[i] = -2; ext[i] = F.mvpMapPoints[i]->Observations(); }
        }
    const std::vector<int32_t> before = holder;
    const orbm_grid_geom_t g = grid_of(F);
    int nmatches = 0;
    if (orbm_search_by_projection_mp(kun.data(), desc.data(), F.mvuRight.data(), n, &g, F.mvScaleFactors.data(),
                                     (int)F.mvScaleFactors.size(), mps.data(), mpd.data(), m, holder.data(), ext.data(),
                                     th, mfNNratio, device, &nmatches) != ORBX_OK)
        return fail("SearchByProjection");
    for (int i = 0; i < n; i++)
        if (holder[i] != before[i] && holder[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[holder[i]];  // :122
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
    const int n = CurrentFrame.N, nl = LastFrame.N;
    if (n == 0 || nl == 0) return 0;
    std::vector<orbx_keypoint_t> kun; gather_keypoints(CurrentFrame.mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(CurrentFrame.mDescriptors, n, desc);
    std::vector<orbm_lastpoint_t> last(nl);
    std::vector<uint8_t> ld((size_t)32 * nl);
    std::map<MapPoint *, int> index;
    for (int i = 0; i < nl; i++) {
        MapPoint *p = LastFrame.mvpMapPoints[i];
        std::memset(&last[i], 0, sizeof(last[i]));
        last[i].has_mp = p && !LastFrame.mvbOutlier[i];
        last[i].octave = LastFrame.mvKeys[i].octave;
        last[i].angle = LastFrame.mvKeysUn[i].angle;
        if (p) {
            if (!index.count(p)) index[p] = i;
            last[i].observations = p->Observations();
        }
        if (last[i].has_mp) {
            cv::Mat x = p->GetWorldPos();
            last[i].wx = x.at<float>(0); last[i].wy = x.at<float>(1); last[i].wz = x.at<float>(2);
            cv::Mat d = p->GetDescriptor();
            std::memcpy(&ld[(size_t)32 * i], d.ptr(0), 32);
        }
    }
    std::vector<int32_t> holder(n, -1), ext(n, 0);
    for (int i = 0; i < n; i++)
        if (CurrentFrame.mvpMapPoints[i]) {
            std::map<MapPoint *, int>::iterator it = index.find(CurrentFrame.mvpMapPoints[i]);
            if (it != index.end()) holder[i] = it->second;
            else { holder[i] = -2; ext[i] = CurrentFrame.mvpMapPoints[i]->Observations(); }
        }
    const std::vector<int32_t> before = holder;
    float Tc[16], Tl[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { Tc[4 * r + c] = CurrentFrame.mTcw.at<float>(r, c); Tl[4 * r + c] = LastFrame.mTcw.at<float>(r, c); }
    orbm_camera_t cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, CurrentFrame.mbf, CurrentFrame.mb};
    const orbm_grid_geom_t g = grid_of(CurrentFrame);
    int nmatches = 0;
    if (orbm_search_by_projection_frame(kun.data(), desc.data(), CurrentFrame.mvuRight.data(), n, &g,
                                        CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(), &cam,
                                        Tc, Tl, last.data(), ld.data(), nl, holder.data(), ext.data(), th, bMono ? 1 : 0,
                                        mbCheckOrientation ? 1 : 0, device, &nmatches) != ORBX_OK)
        return fail("SearchByProjection");
    for (int i = 0; i < n; i++)
        if (holder[i] != before[i]) {
            if (holder[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[holder[i]];  // :1430
            else if (holder[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);  // :1463
        }
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound,
                                   const float th, const int ORBdist) {
    const int n = CurrentFrame.N;
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    const int m = (int)vpMPs.size();
    if (n == 0 || m == 0) return 0;
    // projection + scale prediction stay on the host (they are O(m) scalar work and
    // MapPoint::PredictScale goes through the platform's logf, :1491-1527); the window search,
    // Hamming argmin and rotation histogram run on the GPU
    const float *T = CurrentFrame.mTcw.ptr<float>(0);
    const size_t ts = CurrentFrame.mTcw.step / sizeof(float);
    float Ow[3];
    for (int i = 0; i < 3; i++) {  // Ow = -Rcw.t()*tcw: cv::gemm accumulates CV_32F products in double
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)T[k * ts + i] * (double)T[k * ts + 3];
        Ow[i] = (float)(s * -1.0);
    }
    std::vector<orbm_window_query_t> q(m);
    std::vector<uint8_t> qd((size_t)32 * m);
    for (int i = 0; i < m; i++) {
        orbm_window_query_t &o = q[i];
        std::memset(&o, 0, sizeof(o));
        o.min_level = o.max_level = -1; o.ur_tol = -1.0f; o.blocks = 1;
        MapPoint *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        o.angle = pKF->mvKeysUn[i].angle;
        cv::Mat x3Dw = pMP->GetWorldPos();
        float xw[3] = {x3Dw.at<float>(0), x3Dw.at<float>(1), x3Dw.at<float>(2)}, x3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)T[r * ts + k] * (double)xw[k];
            x3[r] = (float)(s + (double)T[r * ts + 3]);
        }
        const float xc = x3[0], yc = x3[1];
        const float invzc = 1.0 / x3[2];
        const float u = Frame::fx * xc * invzc + Frame::cx;
        const float v = Frame::fy * yc * invzc + Frame::cy;
        if (u < Frame::mnMinX || u > Frame::mnMaxX) continue;
        if (v < Frame::mnMinY || v > Frame::mnMaxY) continue;
        const float p0 = xw[0] - Ow[0], p1 = xw[1] - Ow[1], p2 = xw[2] - Ow[2];
        float dist3D = (float)std::sqrt((double)p0 * p0 + (double)p1 * p1 + (double)p2 * p2);  // cv::norm(PO)
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        o.valid = 1; o.u = u; o.v = v;
        o.radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        o.min_level = nPredictedLevel - 1; o.max_level = nPredictedLevel + 1;
        cv::Mat dMP = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)32 * i], dMP.ptr(0), 32);
    }
    std::vector<orbx_keypoint_t> kun; gather_keypoints(CurrentFrame.mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(CurrentFrame.mDescriptors, n, desc);
    std::vector<int32_t> holder(n, -1);
    for (int i = 0; i < n; i++)
        if (CurrentFrame.mvpMapPoints[i]) holder[i] = -2;  // any holder blocks (:1543)
    const std::vector<int32_t> before = holder;
    const orbm_grid_geom_t g = grid_of(CurrentFrame);
    int nmatches = 0;
    if (orbm_match_windows(kun.data(), desc.data(), NULL, n, &g, q.data(), qd.data(), m, holder.data(), NULL, ORBdist,
                           mbCheckOrientation ? 1 : 0, device, &nmatches) != ORBX_OK)
        return fail("SearchByProjection");
    for (int i = 0; i < n; i++)
        if (holder[i] != before[i]) {
            if (holder[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[holder[i]];                   // :1563
            else if (holder[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);  // :1592
        }
    return nmatches;
}

int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                        std::vector<int> &vnMatches12, int windowSize) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);  // :408
    if (n1 == 0 || n2 == 0) return 0;
    std::vector<orbx_keypoint_t> k1, k2; gather_keypoints(F1.mvKeysUn, k1); gather_keypoints(F2.mvKeysUn, k2);
    std::vector<uint8_t> d1, d2; gather_descriptors(F1.mDescriptors, n1, d1); gather_descriptors(F2.mDescriptors, n2, d2);
    std::vector<float> prev(2 * (size_t)n1);
    for (int i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    std::vector<int32_t> m12(n1, -1);
    const orbm_grid_geom_t g = grid_of(F2);
    int nmatches = 0;
    if (orbm_search_for_initialization(k1.data(), d1.data(), n1, k2.data(), d2.data(), n2, &g, prev.data(), m12.data(),
                                       windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, device, &nmatches) != ORBX_OK)
        return fail("SearchForInitialization");
    for (int i = 0; i < n1; i++) {
        vnMatches12[i] = m12[i];
        vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1];  // :514-517
    }
    return nmatches;
}

int ComputeStereoMatchesHIP(Frame &F) {
    const int N = F.N, Nr = (int)F.mvKeysRight.size();
    F.mvuRight = std::vector<float>(N, -1.0f);  // :483-484
    F.mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0 || !F.mpORBextractorLeft || !F.mpORBextractorRight) return 0;
    std::vector<orbx_keypoint_t> kl, kr; gather_keypoints(F.mvKeys, kl); gather_keypoints(F.mvKeysRight, kr);
    std::vector<uint8_t> dl, dr; gather_descriptors(F.mDescriptors, N, dl); gather_descriptors(F.mDescriptorsRight, Nr, dr);
    int nm = 0;
    if (orbm_stereo(F.mpORBextractorLeft->handle(), F.mpORBextractorRight->handle(), kl.data(), dl.data(), N, kr.data(),
                    dr.data(), Nr, F.mbf, F.mb, F.mvuRight.data(), F.mvDepth.data(), &nm) != ORBX_OK) {
        std::fprintf(stderr, "ComputeStereoMatches: %s\n", orbx_last_error());
        return -1;
    }
    return nm;
}

}  // namespace ORB_SLAM2
