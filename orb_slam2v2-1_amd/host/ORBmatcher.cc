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
#define ORBX_HOST_MAX_LEVELS 16
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
    if (orbm_match_windows(kun.data(), desc.data(), NULL, n, &g, NULL, q.data(), qd.data(), m, holder.data(), NULL, ORBdist,
                           mbCheckOrientation ? 1 : 0, device, &nmatches) != ORBX_OK)
        return fail("SearchByProjection");
    for (int i = 0; i < n; i++)
        if (holder[i] != before[i]) {
            if (holder[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[holder[i]];                   // :1563
            else if (holder[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);  // :1592
        }
    return nmatches;
}

// ---------------------------------------------------------------------------------------------
// KeyFrame-side projection matchers.  The projection arithmetic below is the
// reference's cv::Mat expressions written out with OpenCV's evaluation rules: Mat*Mat(+Mat) is a
// gemm that accumulates in double and rounds once, Mat/scalar and scalar*Mat scale by a FLOAT,
// Mat::dot and cv::norm accumulate in double.
struct Pose3 { float R[9], t[3], Ow[3]; };

// :298-303 / :988-992
static void decompose_sim3(const cv::Mat &Scw, Pose3 &P) {
    double d = 0;
    for (int k = 0; k < 3; k++) d += (double)Scw.at<float>(0, k) * (double)Scw.at<float>(0, k);
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[r * 3 + c] = Scw.at<float>(r, c) * inv + 0.0f;
        P.t[r] = Scw.at<float>(r, 3) * inv + 0.0f;
    }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)P.R[k * 3 + i] * (double)P.t[k];
        P.Ow[i] = (float)(s * -1.0);
    }
}
// :829-838
static void pose_of(KeyFrame *pKF, Pose3 &P) {
    cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[r * 3 + c] = Rcw.at<float>(r, c);
        P.t[r] = tcw.at<float>(r); P.Ow[r] = Ow.at<float>(r);
    }
}
static inline float gemm_row(const float *a, const float *b, float c) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)a[k] * (double)b[k];
    return (float)(s + (double)c);
}
static void empty_query(orbm_window_query_t &o) {
    std::memset(&o, 0, sizeof(o));
    o.min_level = o.max_level = -1; o.ur_tol = -1.0f; o.blocks = 1;
}
// :320-360 / :854-892 / :1010-1051: one map point -> one search window in pKF
static void project_to_window(MapPoint *pMP, const Pose3 &P, KeyFrame *pKF, float th, orbm_window_query_t &o, uint8_t *qd) {
    cv::Mat p3Dw = pMP->GetWorldPos();
    const float pw[3] = {p3Dw.at<float>(0), p3Dw.at<float>(1), p3Dw.at<float>(2)};
    float pc[3];
    for (int r = 0; r < 3; r++) pc[r] = gemm_row(P.R + 3 * r, pw, P.t[r]);
    if (pc[2] < 0.0) return;                                     // depth must be positive
    const float invz = 1 / pc[2];
    const float x = pc[0] * invz, y = pc[1] * invz;
    const float u = pKF->fx * x + pKF->cx, v = pKF->fy * y + pKF->cy;
    if (!pKF->IsInImage(u, v)) return;
    const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
    const float PO[3] = {pw[0] - P.Ow[0], pw[1] - P.Ow[1], pw[2] - P.Ow[2]};
    const float dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) return;
    cv::Mat Pn = pMP->GetNormal();                               // viewing angle must be less than 60 deg
    const double dot = (double)PO[0] * Pn.at<float>(0) + (double)PO[1] * Pn.at<float>(1) + (double)PO[2] * Pn.at<float>(2);
    if (dot < 0.5 * dist) return;
    const int nPredictedLevel = pMP->PredictScale(dist, pKF);
    o.valid = 1; o.u = u; o.v = v;
    o.radius = th * pKF->mvScaleFactors[nPredictedLevel];
    o.min_level = nPredictedLevel - 1; o.max_level = nPredictedLevel;
    o.ur_c = u - pKF->mbf * invz;                                // ur of Fuse (:872)
    cv::Mat dMP = pMP->GetDescriptor();
    std::memcpy(qd, dMP.ptr(0), 32);
}
// the KeyFrame queries with its own int bounds the cell lists its Frame built with float bounds
static orbm_grid_geom_t grid_query_of(const KeyFrame *pKF) {
    orbm_grid_geom_t g;
    g.min_x = (float)pKF->mnMinX; g.min_y = (float)pKF->mnMinY; g.max_x = (float)pKF->mnMaxX; g.max_y = (float)pKF->mnMaxY;
    g.inv_w = pKF->mfGridElementWidthInv; g.inv_h = pKF->mfGridElementHeightInv;
    return g;
}
static orbm_grid_geom_t grid_assign_of(const KeyFrame *pKF) {
    orbm_grid_geom_t g;
    g.min_x = Frame::mnMinX; g.min_y = Frame::mnMinY; g.max_x = Frame::mnMaxX; g.max_y = Frame::mnMaxY;
    g.inv_w = pKF->mfGridElementWidthInv; g.inv_h = pKF->mfGridElementHeightInv;
    return g;
}

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints,
                                   std::vector<MapPoint *> &vpMatched, int th) {
    const int n = (int)pKF->mvKeysUn.size(), m = (int)vpPoints.size();
    if (n == 0 || m == 0) return 0;
    Pose3 P;
    decompose_sim3(Scw, P);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());   // :306-307
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    std::vector<orbm_window_query_t> q(m);
    std::vector<uint8_t> qd((size_t)32 * m);
    for (int i = 0; i < m; i++) {
        empty_query(q[i]);
        MapPoint *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        project_to_window(pMP, P, pKF, (float)th, q[i], &qd[(size_t)32 * i]);
        q[i].ur_tol = -1.0f;
    }
    std::vector<orbx_keypoint_t> kun; gather_keypoints(pKF->mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(pKF->mDescriptors, n, desc);
    std::vector<int32_t> holder(n, -1);
    for (int i = 0; i < n; i++) if (vpMatched[i]) holder[i] = -2;              // if(vpMatched[idx]) continue; (:375)
    const orbm_grid_geom_t g = grid_query_of(pKF), ga = grid_assign_of(pKF);
    int nmatches = 0;
    if (orbm_match_windows(kun.data(), desc.data(), NULL, n, &g, &ga, q.data(), qd.data(), m, holder.data(), NULL, TH_LOW, 0,
                           device, &nmatches) != ORBX_OK)
        return fail("SearchByProjection");
    for (int i = 0; i < n; i++) if (holder[i] >= 0) vpMatched[i] = vpPoints[holder[i]];   // :396
    return nmatches;
}

int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th) {
    const int n = (int)pKF->mvKeysUn.size(), nMPs = (int)vpMapPoints.size();
    if (n == 0 || nMPs == 0) return 0;
    Pose3 P;
    pose_of(pKF, P);
    // The window search of a point never reads what an earlier point changed (descriptors and positions
    // only change for points that are already done or already in pKF), so all searches run first, in
    // parallel on the GPU; the map-point bookkeeping then runs in the reference's order.
    std::vector<orbm_window_query_t> q(nMPs);
    std::vector<uint8_t> qd((size_t)32 * nMPs);
    for (int i = 0; i < nMPs; i++) {
        empty_query(q[i]);
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        project_to_window(pMP, P, pKF, th, q[i], &qd[(size_t)32 * i]);
    }
    std::vector<orbx_keypoint_t> kun; gather_keypoints(pKF->mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(pKF->mDescriptors, n, desc);
    std::vector<int32_t> bestIdx(nMPs), bestDist(nMPs);
    const orbm_grid_geom_t g = grid_query_of(pKF), ga = grid_assign_of(pKF);
    if (orbm_best_in_windows(kun.data(), desc.data(), pKF->mvuRight.data(), n, &g, &ga, q.data(), qd.data(), nMPs,
                             pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), bestIdx.data(),
                             bestDist.data(), device) != ORBX_OK)
        return fail("Fuse");
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;    // :851, with the state of THIS moment
        if (!q[i].valid) continue;
        if (bestDist[i] <= TH_LOW) {                             // :954-973
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx[i]);
                pKF->AddMapPoint(pMP, bestIdx[i]);
            }
            nFused++;
        }
    }
    return nFused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th,
                     std::vector<MapPoint *> &vpReplacePoint) {
    const int n = (int)pKF->mvKeysUn.size(), nPoints = (int)vpPoints.size();
    if (n == 0 || nPoints == 0) return 0;
    Pose3 P;
    decompose_sim3(Scw, P);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();           // :995
    std::vector<orbm_window_query_t> q(nPoints);
    std::vector<uint8_t> qd((size_t)32 * nPoints);
    for (int i = 0; i < nPoints; i++) {
        empty_query(q[i]);
        MapPoint *pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        project_to_window(pMP, P, pKF, th, q[i], &qd[(size_t)32 * i]);
    }
    std::vector<orbx_keypoint_t> kun; gather_keypoints(pKF->mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(pKF->mDescriptors, n, desc);
    std::vector<int32_t> bestIdx(nPoints), bestDist(nPoints);
    const orbm_grid_geom_t g = grid_query_of(pKF), ga = grid_assign_of(pKF);
    if (orbm_best_in_windows(kun.data(), desc.data(), NULL, n, &g, &ga, q.data(), qd.data(), nPoints, NULL, 0, bestIdx.data(),
                             bestDist.data(), device) != ORBX_OK)
        return fail("Fuse");
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        if (!q[iMP].valid || bestDist[iMP] > TH_LOW) continue;
        MapPoint *pMP = vpPoints[iMP];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[iMP]);                    // :1086-1097
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx[iMP]);
            pKF->AddMapPoint(pMP, bestIdx[iMP]);
        }
        nFused++;
    }
    return nFused;
}

int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12,
                             const cv::Mat &R12, const cv::Mat &t12, const float th) {
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;    // :1107-1110 (KF1's for both)
    Pose3 P1, P2;
    pose_of(pKF1, P1); pose_of(pKF2, P2);
    float sR12[9], sR21[9], t12f[3], t21[3];
    const float a12 = s12, a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            sR12[r * 3 + c] = R12.at<float>(r, c) * a12 + 0.0f;                // :1121
            sR21[r * 3 + c] = R12.at<float>(c, r) * a21 + 0.0f;                // :1122
        }
        t12f[r] = t12.at<float>(r);
    }
    for (int i = 0; i < 3; i++) {                                              // :1123
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)sR21[i * 3 + k] * (double)t12f[k];
        t21[i] = (float)(s * -1.0);
    }
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    if (N1 == 0 || N2 == 0) return 0;
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                             // :1134-1144
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    std::vector<int32_t> vnMatch[2], bestDist[2];
    for (int dir = 0; dir < 2; dir++) {   // 0: KF1's points into KF2 (:1150-1227), 1: KF2's into KF1 (:1230-1307)
        const std::vector<MapPoint *> &src = dir ? vpMapPoints2 : vpMapPoints1;
        const std::vector<bool> &already = dir ? vbAlreadyMatched2 : vbAlreadyMatched1;
        KeyFrame *pKFdst = dir ? pKF1 : pKF2;
        const Pose3 &Pw = dir ? P2 : P1;
        const float *sR = dir ? sR12 : sR21, *tt = dir ? t12f : t21;
        const int ns = (int)src.size(), nd = (int)pKFdst->mvKeysUn.size();
        std::vector<orbm_window_query_t> q(ns);
        std::vector<uint8_t> qd((size_t)32 * ns);
        for (int i = 0; i < ns; i++) {
            empty_query(q[i]);
            MapPoint *pMP = src[i];
            if (!pMP || already[i]) continue;
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            const float pw[3] = {p3Dw.at<float>(0), p3Dw.at<float>(1), p3Dw.at<float>(2)};
            float pa[3], pb[3];
            for (int r = 0; r < 3; r++) pa[r] = gemm_row(Pw.R + 3 * r, pw, Pw.t[r]);
            for (int r = 0; r < 3; r++) pb[r] = gemm_row(sR + 3 * r, pa, tt[r]);
            if (pb[2] < 0.0) continue;
            const float invz = 1.0 / pb[2];
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!pKFdst->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = (float)std::sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKFdst);
            q[i].valid = 1; q[i].u = u; q[i].v = v;
            q[i].radius = th * pKFdst->mvScaleFactors[nPredictedLevel];
            q[i].min_level = nPredictedLevel - 1; q[i].max_level = nPredictedLevel;
            cv::Mat dMP = pMP->GetDescriptor();
            std::memcpy(&qd[(size_t)32 * i], dMP.ptr(0), 32);
        }
        std::vector<orbx_keypoint_t> kun; gather_keypoints(pKFdst->mvKeysUn, kun);
        std::vector<uint8_t> desc; gather_descriptors(pKFdst->mDescriptors, nd, desc);
        vnMatch[dir].assign(ns, -1); bestDist[dir].assign(ns, 256);
        const orbm_grid_geom_t g = grid_query_of(pKFdst), ga = grid_assign_of(pKFdst);
        if (orbm_best_in_windows(kun.data(), desc.data(), NULL, nd, &g, &ga, q.data(), qd.data(), ns, NULL, 0,
                                 vnMatch[dir].data(), bestDist[dir].data(), device) != ORBX_OK)
            return fail("SearchBySim3");
        for (int i = 0; i < ns; i++)
            if (bestDist[dir][i] > TH_HIGH) vnMatch[dir][i] = -1;              // :1223 / :1303
    }
    int nFound = 0;                                                            // :1309-1325
    for (int i1 = 0; i1 < N1; i1++) {
        const int idx2 = vnMatch[0][i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch[1][idx2];
            if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
    }
    return nFound;
}

// ---------------------------------------------------------------------------------------------
// BoW-guided matchers.  The merge of the two FeatureVectors (src/ORBmatcher.cc:176-248,
// :542-625) runs here; the brute-force search inside each shared vocabulary node runs on the GPU, one
// wavefront per node (a feature belongs to one node, so nodes never interact).
static void intersect_feature_vectors(const DBoW2::FeatureVector &v1, const DBoW2::FeatureVector &v2,
                                      std::vector<int32_t> &s1, std::vector<int32_t> &i1, std::vector<int32_t> &s2,
                                      std::vector<int32_t> &i2) {
    s1.assign(1, 0); s2.assign(1, 0); i1.clear(); i2.clear();
    DBoW2::FeatureVector::const_iterator f1it = v1.begin(), f2it = v2.begin(), f1end = v1.end(), f2end = v2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            i1.insert(i1.end(), f1it->second.begin(), f1it->second.end());
            i2.insert(i2.end(), f2it->second.begin(), f2it->second.end());
            s1.push_back((int32_t)i1.size()); s2.push_back((int32_t)i2.size());
            f1it++; f2it++;
        } else if (f1it->first < f2it->first) f1it = v1.lower_bound(f2it->first);
        else f2it = v2.lower_bound(f1it->first);
    }
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));   // :163
    const int nq = (int)vpMapPointsKF.size(), nc = F.N;
    if (nq == 0 || nc == 0) return 0;
    std::vector<int32_t> sq, iq, sc, ic;
    intersect_feature_vectors(pKF->mFeatVec, F.mFeatVec, sq, iq, sc, ic);
    std::vector<uint8_t> qd, cd, qv(nq);
    gather_descriptors(pKF->mDescriptors, nq, qd); gather_descriptors(F.mDescriptors, nc, cd);
    std::vector<float> qa(nq), ca(nc);
    for (int i = 0; i < nq; i++) { qa[i] = pKF->mvKeysUn[i].angle; MapPoint *p = vpMapPointsKF[i]; qv[i] = p && !p->isBad(); }   // :195-199, :237
    for (int i = 0; i < nc; i++) ca[i] = F.mvKeys[i].angle;                                                                        // :241
    std::vector<int32_t> match(nq, -1);
    int nmatches = 0;
    if (orbm_search_by_bow(qd.data(), qa.data(), qv.data(), nq, cd.data(), ca.data(), NULL, nc, sq.data(), iq.data(), sc.data(),
                           ic.data(), (int)sq.size() - 1, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches,
                           device) != ORBX_OK)
        return fail("SearchByBoW");
    for (int i = 0; i < nq; i++) if (match[i] >= 0) vpMapPointMatches[match[i]] = vpMapPointsKF[i];   // :235
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) {
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));      // :534
    const int nq = (int)vpMapPoints1.size(), nc = (int)vpMapPoints2.size();
    if (nq == 0 || nc == 0) return 0;
    std::vector<int32_t> sq, iq, sc, ic;
    intersect_feature_vectors(pKF1->mFeatVec, pKF2->mFeatVec, sq, iq, sc, ic);
    std::vector<uint8_t> qd, cd, qv(nq), cv(nc);
    gather_descriptors(pKF1->mDescriptors, nq, qd); gather_descriptors(pKF2->mDescriptors, nc, cd);
    std::vector<float> qa(nq), ca(nc);
    for (int i = 0; i < nq; i++) { qa[i] = pKF1->mvKeysUn[i].angle; MapPoint *p = vpMapPoints1[i]; qv[i] = p && !p->isBad(); }   // :560-564
    for (int i = 0; i < nc; i++) { ca[i] = pKF2->mvKeysUn[i].angle; MapPoint *p = vpMapPoints2[i]; cv[i] = p && !p->isBad(); }   // :576-580
    std::vector<int32_t> match(nq, -1);
    int nmatches = 0;
    if (orbm_search_by_bow(qd.data(), qa.data(), qv.data(), nq, cd.data(), ca.data(), cv.data(), nc, sq.data(), iq.data(), sc.data(),
                           ic.data(), (int)sq.size() - 1, TH_LOW - 1 /* bestDist1 < TH_LOW, :599 */, mfNNratio,
                           mbCheckOrientation ? 1 : 0, match.data(), &nmatches, device) != ORBX_OK)
        return fail("SearchByBoW");
    for (int i = 0; i < nq; i++) if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];       // :603
    return nmatches;
}

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
                                       std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo) {
    vMatchedPairs.clear();
    const int nq = pKF1->N, nc = pKF2->N;
    if (nq == 0 || nc == 0) return 0;
    // epipole in the second image (:663-670): C2 = R2w*Cw + t2w is a gemm (double accumulation, one rounding)
    cv::Mat Cw = pKF1->GetCameraCenter(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    float C2[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)R2w.at<float>(r, k) * (double)Cw.at<float>(k);
        C2[r] = (float)(s + (double)t2w.at<float>(r));
    }
    const float invz = 1.0f / C2[2];
    const float ex = pKF2->fx * C2[0] * invz + pKF2->cx;
    const float ey = pKF2->fy * C2[1] * invz + pKF2->cy;
    std::vector<int32_t> sq, iq, sc, ic;
    intersect_feature_vectors(pKF1->mFeatVec, pKF2->mFeatVec, sq, iq, sc, ic);
    std::vector<orbx_keypoint_t> k1, k2; gather_keypoints(pKF1->mvKeysUn, k1); gather_keypoints(pKF2->mvKeysUn, k2);
    std::vector<uint8_t> qd, cd, qf(nq), cf(nc);
    gather_descriptors(pKF1->mDescriptors, nq, qd); gather_descriptors(pKF2->mDescriptors, nc, cd);
    for (int i = 0; i < nq; i++) {
        const bool stereo = pKF1->mvuRight[i] >= 0;
        qf[i] = (uint8_t)((!pKF1->GetMapPoint(i) && (!bOnlyStereo || stereo) ? 1 : 0) | (stereo ? 2 : 0));   // :700-710
    }
    for (int i = 0; i < nc; i++) {
        const bool stereo = pKF2->mvuRight[i] >= 0;
        cf[i] = (uint8_t)((!pKF2->GetMapPoint(i) && (!bOnlyStereo || stereo) ? 1 : 0) | (stereo ? 2 : 0));   // :721-731
    }
    float F[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[r * 3 + c] = F12.at<float>(r, c);
    std::vector<int32_t> match(nq, -1);
    int nmatches = 0;
    if (orbm_search_for_triangulation(k1.data(), qd.data(), qf.data(), nq, k2.data(), cd.data(), cf.data(), nc, sq.data(), iq.data(),
                                      sc.data(), ic.data(), (int)sq.size() - 1, F, ex, ey, pKF2->mvScaleFactors.data(),
                                      pKF2->mvLevelSigma2.data(), (int)pKF2->mvScaleFactors.size(), TH_LOW,
                                      mbCheckOrientation ? 1 : 0, match.data(), &nmatches, device) != ORBX_OK)
        return fail("SearchForTriangulation");
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < nq; i++)
        if (match[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[i]));   // :812-822
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

int SearchLocalPointsHIP(Frame &F, const std::vector<MapPoint *> &vpLocalMapPoints, float th, float nnratio, int &nToMatch) {
    nToMatch = 0;
    const int n = F.N, m = (int)vpLocalMapPoints.size();
    if (m == 0) return 0;
    std::vector<orbm_worldpoint_t> pts(m);
    std::vector<uint8_t> mpd((size_t)32 * m);
    std::map<MapPoint *, int> index;
    for (int i = 0; i < m; i++) {
        MapPoint *pMP = vpLocalMapPoints[i];
        index[pMP] = i;
        std::memset(&pts[i], 0, sizeof(pts[i]));
        pts[i].observations = pMP->Observations();
        if (pMP->mnLastFrameSeen == F.mnId) continue;          // src/Tracking.cc:1312-1313
        if (pMP->isBad()) continue;                            // :1314-1315
        pts[i].valid = 1;
        cv::Mat P = pMP->GetWorldPos(), Pn = pMP->GetNormal();
        pts[i].wx = P.at<float>(0); pts[i].wy = P.at<float>(1); pts[i].wz = P.at<float>(2);
        pts[i].nx = Pn.at<float>(0); pts[i].ny = Pn.at<float>(1); pts[i].nz = Pn.at<float>(2);
        // mfMaxDistance / mfMinDistance are protected in include/MapPoint.h:156-157 (the getters return them
        // scaled): in the reference tree declare this function a friend of MapPoint
        pts[i].max_distance = pMP->mfMaxDistance;
        pts[i].min_distance = pMP->mfMinDistance;
        cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&mpd[(size_t)32 * i], d.ptr(0), 32);
    }
    std::vector<orbx_keypoint_t> kun; gather_keypoints(F.mvKeysUn, kun);
    std::vector<uint8_t> desc; gather_descriptors(F.mDescriptors, n, desc);
    std::vector<int32_t> holder(n, -1), ext(n, 0);
    for (int i = 0; i < n; i++)
        if (F.mvpMapPoints[i]) {
            std::map<MapPoint *, int>::iterator it = index.find(F.mvpMapPoints[i]);
            if (it != index.end()) holder[i] = it->second;
            else { holder[i] = -2; ext[i] = F.mvpMapPoints[i]->Observations(); }
        }
    const std::vector<int32_t> before = holder;
    const orbm_grid_geom_t g = grid_of(F);
    orbm_camera_t cam;
    cam.fx = Frame::fx; cam.fy = Frame::fy; cam.cx = Frame::cx; cam.cy = Frame::cy; cam.mbf = F.mbf; cam.mb = F.mb;
    float T[16], thr[ORBX_HOST_MAX_LEVELS];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[r * 4 + c] = F.mTcw.at<float>(r, c);
    const int nlevels = (int)F.mvScaleFactors.size();
    if (nlevels > ORBX_HOST_MAX_LEVELS || orbm_predict_scale_thresholds(F.mfLogScaleFactor, nlevels, thr) != ORBX_OK) return fail("SearchLocalPoints");
    std::vector<orbm_mappoint_t> proj(m);
    int nmatches = 0;
    if (orbm_search_local_points(kun.data(), desc.data(), F.mvuRight.data(), n, &g, F.mvScaleFactors.data(), nlevels, pts.data(),
                                 mpd.data(), m, T, &cam, 0.5f, thr, holder.data(), ext.data(), th, nnratio, ORBmatcher::device,
                                 &nmatches, proj.data()) != ORBX_OK)
        return fail("SearchLocalPoints");
    for (int i = 0; i < m; i++) {
        MapPoint *pMP = vpLocalMapPoints[i];
        if (!pts[i].valid) continue;
        pMP->mbTrackInView = proj[i].in_view != 0;             // src/Frame.cc:286,329
        if (proj[i].in_view) {
            pMP->mTrackProjX = proj[i].proj_x; pMP->mTrackProjXR = proj[i].proj_xr; pMP->mTrackProjY = proj[i].proj_y;
            pMP->mnTrackScaleLevel = proj[i].level; pMP->mTrackViewCos = proj[i].view_cos;
            pMP->IncreaseVisible();                            // src/Tracking.cc:1320-1322
            nToMatch++;
        }
    }
    for (int i = 0; i < n; i++)
        if (holder[i] != before[i] && holder[i] >= 0) F.mvpMapPoints[i] = vpLocalMapPoints[holder[i]];
    return nmatches;
}

int ComputeDistinctiveDescriptorsHIP(const std::vector<MapPoint *> &vpMapPoints, std::vector<cv::Mat> &vBest) {
    const int m = (int)vpMapPoints.size();
    vBest.assign(m, cv::Mat());
    std::vector<int32_t> offsets(m + 1, 0);
    std::vector<uint8_t> rows;
    std::vector<cv::Mat> src;   // the candidate rows, in the order the reference visits them
    for (int p = 0; p < m; p++) {
        MapPoint *pMP = vpMapPoints[p];
        if (pMP && !pMP->isBad()) {                                            // :260-262
            std::map<KeyFrame *, size_t> observations = pMP->GetObservations();
            for (std::map<KeyFrame *, size_t>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
                KeyFrame *pKF = mit->first;
                if (!pKF->isBad()) {                                           // :275-276
                    cv::Mat row = pKF->mDescriptors.row((int)mit->second);
                    rows.insert(rows.end(), row.ptr(0), row.ptr(0) + 32);
                    src.push_back(row);
                }
            }
        }
        offsets[p + 1] = (int32_t)src.size();
    }
    if (m == 0) return 0;
    std::vector<int32_t> best(m, -1);
    if (orbm_distinctive_descriptors(rows.empty() ? NULL : rows.data(), offsets.data(), m, best.data(), NULL,
                                     ORBmatcher::device) != ORBX_OK) {
        std::fprintf(stderr, "ComputeDistinctiveDescriptors: %s\n", orbx_last_error());
        return -1;
    }
    int n = 0;
    for (int p = 0; p < m; p++)
        if (best[p] >= 0) { vBest[p] = src[offsets[p] + best[p]].clone(); n++; }   // :313-316
    return n;
}

int ExtractStereoFrameHIP(Frame &F, const cv::Mat &imLeft, const cv::Mat &imRight) {
    F.mvKeys.clear(); F.mvKeysRight.clear(); F.mDescriptors.release(); F.mDescriptorsRight.release();
    F.mvuRight.clear(); F.mvDepth.clear(); F.N = 0;
    if (!F.mpORBextractorLeft || !F.mpORBextractorLeft->ok() || imLeft.empty() || imRight.empty()) return 0;
    if (imLeft.cols != imRight.cols || imLeft.rows != imRight.rows || imLeft.step != imRight.step) {
        std::fprintf(stderr, "ExtractStereoFrame: left and right image differ in size or stride\n");
        return -1;
    }
    orbx_extractor_t *h = F.mpORBextractorLeft->handle();
    // the latency form (include/orbx.h: orbx_stereo_frame_view): no copy command in either direction - the first kernel reads the images
    // where they lie (pinned capture buffers from orbx_host_alloc are read over the bus; a pageable cv::Mat is staged by the call) and the
    // last kernel writes the frame's record into pinned memory of the handle, which `v` points into until the call after the next
    orbx_stereo_view_t v;
    if (orbx_stereo_frame_view(h, imLeft.ptr(0), imRight.ptr(0), imLeft.cols, imLeft.rows, (int)imLeft.step, F.mbf, F.mb, &v) != ORBX_OK) {
        std::fprintf(stderr, "ExtractStereoFrame: %s\n", orbx_last_error());
        return -1;
    }
    const int nl = v.nl, nr = v.nr, nm = v.nmatch;
    auto fill = [](const orbx_keypoint_t *src, const uint8_t *dsrc, int n, std::vector<cv::KeyPoint> &keys, cv::Mat &desc) {
        keys.reserve(n);
        for (int i = 0; i < n; i++) {
            const orbx_keypoint_t &s = src[i];
            cv::KeyPoint kp;
            kp.pt.x = s.x; kp.pt.y = s.y; kp.size = s.size; kp.angle = s.angle; kp.response = s.response;
            kp.octave = s.octave; kp.class_id = s.class_id;
            keys.push_back(kp);
        }
        if (n > 0) {
            desc.create(n, 32, CV_8U);
            for (int i = 0; i < n; i++) std::memcpy(desc.ptr(i), dsrc + (size_t)32 * i, 32);
        }
    };
    fill(v.kl, v.dl, nl, F.mvKeys, F.mDescriptors);
    fill(v.kr, v.dr, nr, F.mvKeysRight, F.mDescriptorsRight);
    F.N = nl;                                         // :86
    F.mvuRight.assign(v.uright, v.uright + nl);       // -1 where unmatched, as :483-484 initialise them
    F.mvDepth.assign(v.depth, v.depth + nl);
    return nm;
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
