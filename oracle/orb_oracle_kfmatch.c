/*
 * orb_oracle_kfmatch.c — CPU ORACLE (test infrastructure, never on the product path):
 * restatement on flat arrays of the KeyFrame-side projection matchers:
 *   ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)   src/ORBmatcher.cc:290-403
 *   ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th)                              src/ORBmatcher.cc:827-977
 *   ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)            src/ORBmatcher.cc:979-1102
 *   ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)      src/ORBmatcher.cc:1104-1328
 *   KeyFrame::GetFeaturesInArea / IsInImage                                   src/KeyFrame.cc:572-616
 *   MapPoint::PredictScale(dist, KeyFrame*)                                   src/MapPoint.cc:397-412
 * cv::Mat arithmetic follows OpenCV 2.4.11/3.2 [external]: Mat*Mat and Mat*Mat+Mat are gemm with
 * double accumulation, result = (float)(alpha*sum + beta*c); Mat/scalar and scalar*Mat are
 * convertTo with a FLOAT scale; Mat::dot and cv::norm(L2) accumulate in double.
 * PARITY UNPINNED (see orb_oracle.h).  Compile with -ffp-contract=off.
 *
 * Map-point bookkeeping model used by the two Fuse restatements (the reference mutates a pointer
 * graph; here it is flat state): the listed map points have no observations in other keyframes,
 * so MapPoint::Replace (src/MapPoint.cc:187-225) reduces to "the replaced point becomes bad, and
 * if it occupied slot j of this keyframe the replacing point takes slot j and gains one
 * observation (two if mvuRight[j] >= 0, src/MapPoint.cc:108-119)".
 */
#include "orb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100
#define TH_LOW 50

/* gemm row: (float)(sum_k (double)a[k]*(double)b[k] + (double)c) */
static float row_madd(const float *a, int astride, const float *b, float c) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)a[k * astride] * (double)b[k];
    return (float)(s + (double)c);
}

/* Rcw(3x3 row-major), tcw, Ow from a Sim3 matrix Scw (4x4 row-major), :298-303 */
void oracle_decompose_sim3(const float *S, float *Rcw, float *tcw, float *Ow) {
    double d = 0;
    for (int k = 0; k < 3; k++) d += (double)S[k] * (double)S[k];        /* sRcw.row(0).dot(sRcw.row(0)) */
    const float scw = (float)sqrt(d);
    const float inv = (float)(1.0 / (double)scw);                         /* Mat / s == convertTo(alpha = 1./s) */
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = S[r * 4 + c] * inv + 0.0f;
        tcw[r] = S[r * 4 + 3] * inv + 0.0f;
    }
    for (int i = 0; i < 3; i++) {                                         /* Ow = -Rcw.t()*tcw : gemm alpha = -1 */
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        Ow[i] = (float)(s * -1.0);
    }
}

int oracle_predict_scale_ratio(float ratio, float log_sf, int nlevels) {   /* MapPoint::PredictScale from ratio on */
    int nScale = (int)ceilf(logf(ratio) / log_sf);
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}
static int predict_scale(float max_distance, float dist, float log_sf, int nlevels) {
    const float ratio = max_distance / dist;
    int nScale = (int)ceilf(logf(ratio) / log_sf);
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}

/* projection shared by :312-360, :844-892, :1002-1051: Rcw/tcw/Ow given.  check_normal: the
 * viewing-angle test (absent from SearchBySim3).  Writes one window query; ur (u - bf*invz) in ur_c. */
static void project_point(const oracle_mappoint3d_t *p, const float *Rcw, const float *tcw, const float *Ow,
                          const oracle_grid_geom_t *g, const oracle_cam_t *cam, const float *sf, int nlevels,
                          float log_sf, float th, int check_normal, oracle_window_query_t *o) {
    memset(o, 0, sizeof(*o));
    o->min_level = o->max_level = -1; o->ur_tol = -1.0f; o->blocks = 1;
    if (!p->valid) return;
    const float pw[3] = {p->wx, p->wy, p->wz};
    float pc[3];
    for (int r = 0; r < 3; r++) pc[r] = row_madd(Rcw + 3 * r, 1, pw, tcw[r]);
    if (pc[2] < 0.0) return;
    const float invz = 1 / pc[2];
    const float x = pc[0] * invz, y = pc[1] * invz;
    const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
    if (!(u >= g->min_x && u < g->max_x && v >= g->min_y && v < g->max_y)) return;   /* KeyFrame::IsInImage */
    const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
    const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    const float maxDistance = 1.2f * p->max_distance, minDistance = 0.8f * p->min_distance;
    if (dist < minDistance || dist > maxDistance) return;
    if (check_normal) {
        const double dot = (double)PO[0] * p->nx + (double)PO[1] * p->ny + (double)PO[2] * p->nz;
        if (dot < 0.5 * dist) return;
    }
    const int lvl = predict_scale(p->max_distance, dist, log_sf, nlevels);
    o->valid = 1; o->u = u; o->v = v; o->radius = th * sf[lvl];
    o->min_level = lvl - 1; o->max_level = lvl;
    o->ur_c = u - cam->mbf * invz;
}

void oracle_sim3_window_queries(const oracle_mappoint3d_t *pts, int m, const oracle_grid_geom_t *g, const float *sf,
                                int nlevels, float log_sf, const oracle_cam_t *cam, const float *Scw, float th,
                                oracle_window_query_t *q) {
    float R[9], t[3], Ow[3];
    oracle_decompose_sim3(Scw, R, t, Ow);
    for (int i = 0; i < m; i++) project_point(&pts[i], R, t, Ow, g, cam, sf, nlevels, log_sf, th, 1, &q[i]);
}

/* pose of a KeyFrame (Tcw 4x4 row-major): Rcw, tcw, Ow = -Rcw^T tcw (src/KeyFrame.cc SetPose) */
static void pose_parts(const float *T, float *R, float *t, float *Ow) {
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c]; t[r] = T[r * 4 + 3]; }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)R[k * 3 + i] * (double)t[k];
        Ow[i] = (float)(s * -1.0);
    }
}

void oracle_pose_window_queries(const oracle_mappoint3d_t *pts, int m, const oracle_grid_geom_t *g, const float *sf,
                                int nlevels, float log_sf, const oracle_cam_t *cam, const float *Tcw, float th,
                                oracle_window_query_t *q) {
    float R[9], t[3], Ow[3];
    pose_parts(Tcw, R, t, Ow);
    for (int i = 0; i < m; i++) project_point(&pts[i], R, t, Ow, g, cam, sf, nlevels, log_sf, th, 1, &q[i]);
}

/* inner loops :372-392 / :905-951 / :1064-1081 / :1203-1221: first minimum in GetFeaturesInArea order */
void oracle_best_in_windows(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                            const oracle_grid_geom_t *g, const oracle_grid_geom_t *ga, const oracle_window_query_t *q, const uint8_t *qdesc, int m,
                            const float *inv_sigma2, int nlevels, int32_t *best_idx, int32_t *best_dist) {
    oracle_grid_t *grid = oracle_grid_build(kun, n, ga ? ga : g);
    oracle_grid_set_query_geom(grid, g);
    int32_t *idxs = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < m; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (!q[i].valid) continue;
        const int nc = oracle_grid_query(grid, q[i].u, q[i].v, q[i].radius, -1, -1, idxs, n);
        for (int c = 0; c < nc; c++) {
            const int idx = idxs[c];
            const int kpLevel = kun[idx].octave;
            if (kpLevel < q[i].min_level || kpLevel > q[i].max_level) continue;
            if (inv_sigma2) {
                const int lv = kpLevel < 0 ? 0 : (kpLevel >= nlevels ? nlevels - 1 : kpLevel);
                const float ex = q[i].u - kun[idx].x, ey = q[i].v - kun[idx].y;
                if (uright && uright[idx] >= 0) {
                    const float er = q[i].ur_c - uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[lv] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[lv] > 5.99) continue;
                }
            }
            const int dist = oracle_hamming(qdesc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (dist < best_dist[i]) { best_dist[i] = dist; best_idx[i] = idx; }
        }
    }
    free(idxs); oracle_grid_free(grid);
}

/* SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) :290-403.  pts[i].valid =
 * !isBad && !spAlreadyFound.count(pMP).  matched[n] in/out: -1 empty, -2 held before the call,
 * >= 0 index into pts. */
int oracle_search_by_projection_sim3(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                                     const oracle_grid_geom_t *ga,
                                     const float *sf, int nlevels, float log_sf, const oracle_cam_t *cam,
                                     const float *Scw, const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m,
                                     int32_t *matched, int th) {
    int nmatches = 0;
    oracle_window_query_t *q = (oracle_window_query_t *)malloc(sizeof(*q) * (m > 0 ? m : 1));
    oracle_sim3_window_queries(pts, m, g, sf, nlevels, log_sf, cam, Scw, (float)th, q);
    oracle_grid_t *grid = oracle_grid_build(kun, n, ga ? ga : g);
    oracle_grid_set_query_geom(grid, g);
    int32_t *idxs = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < m; i++) {
        if (!q[i].valid) continue;
        const int nc = oracle_grid_query(grid, q[i].u, q[i].v, q[i].radius, -1, -1, idxs, n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = idxs[c];
            if (matched[idx] != -1) continue;                                   /* if(vpMatched[idx]) :375 */
            const int kpLevel = kun[idx].octave;
            if (kpLevel < q[i].min_level || kpLevel > q[i].max_level) continue; /* :380 */
            const int dist = oracle_hamming(pdesc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { matched[bestIdx] = i; nmatches++; }
    }
    free(idxs); free(q); oracle_grid_free(grid);
    return nmatches;
}

/* Fuse(pKF, vpMapPoints, th) :827-977.  State: bad[m], in_kf[m], obs[m] (in/out); slot[n] in/out
 * (-1 empty, -2 external holder with ext_obs[j] / ext_bad[j], >= 0 index into pts).
 * action[m] out: 0 nothing, 1 added to an empty slot, 2 pMP replaced by the slot's point (pMP bad),
 * 3 the slot's point replaced by pMP, 4 slot's point was bad (counted, nothing done). */
int oracle_fuse(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n, const oracle_grid_geom_t *g,
                const oracle_grid_geom_t *ga,
                const float *sf, const float *inv_sigma2, int nlevels, float log_sf, const oracle_cam_t *cam,
                const float *Tcw, const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m, int32_t *bad,
                int32_t *in_kf, int32_t *obs, int32_t *slot, const int32_t *ext_obs, int32_t *ext_bad, float th,
                int32_t *best_idx, int32_t *action) {
    int nFused = 0;
    oracle_window_query_t *q = (oracle_window_query_t *)malloc(sizeof(*q) * (m > 0 ? m : 1));
    int32_t *bd = (int32_t *)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    /* the candidate search never reads the state below, so it is evaluated for every point first */
    oracle_pose_window_queries(pts, m, g, sf, nlevels, log_sf, cam, Tcw, th, q);
    oracle_best_in_windows(kun, desc, uright, n, g, ga, q, pdesc, m, inv_sigma2, nlevels, best_idx, bd);
    for (int i = 0; i < m; i++) {
        action[i] = 0;
        if (!pts[i].valid) { best_idx[i] = -1; continue; }                      /* if(!pMP) :848 */
        if (bad[i] || in_kf[i]) { best_idx[i] = -1; continue; }                 /* :851 */
        if (!q[i].valid || bd[i] > TH_LOW) { best_idx[i] = -1; continue; }
        const int j = best_idx[i], h = slot[j];
        if (h != -1) {
            const int hbad = h == -2 ? ext_bad[j] : bad[h];
            const int hobs = h == -2 ? ext_obs[j] : obs[h];
            if (!hbad) {
                if (hobs > obs[i]) { bad[i] = 1; action[i] = 2; }               /* pMP->Replace(pMPinKF) */
                else {                                                          /* pMPinKF->Replace(pMP) */
                    if (h == -2) ext_bad[j] = 1; else { bad[h] = 1; in_kf[h] = 0; }
                    slot[j] = i; in_kf[i] = 1; obs[i] += (uright && uright[j] >= 0) ? 2 : 1;
                    action[i] = 3;
                }
            } else action[i] = 4;
        } else {                                                                /* :969-970 */
            slot[j] = i; in_kf[i] = 1; obs[i] += (uright && uright[j] >= 0) ? 2 : 1;
            action[i] = 1;
        }
        nFused++;
    }
    free(q); free(bd);
    return nFused;
}

/* Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) :979-1102.  pts[i].valid = !isBad && !spAlreadyFound.count(pMP).
 * slot[n] in/out as above; replace[m] out: -1 none, -2 external holder, >= 0 list index (vpReplacePoint[iMP]). */
int oracle_fuse_sim3(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                     const oracle_grid_geom_t *ga, const float *sf,
                     int nlevels, float log_sf, const oracle_cam_t *cam, const float *Scw,
                     const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m, const int32_t *bad, int32_t *slot,
                     const int32_t *ext_bad, float th, int32_t *best_idx, int32_t *replace) {
    int nFused = 0;
    oracle_window_query_t *q = (oracle_window_query_t *)malloc(sizeof(*q) * (m > 0 ? m : 1));
    int32_t *bd = (int32_t *)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    oracle_sim3_window_queries(pts, m, g, sf, nlevels, log_sf, cam, Scw, th, q);
    oracle_best_in_windows(kun, desc, NULL, n, g, ga, q, pdesc, m, NULL, nlevels, best_idx, bd);
    for (int i = 0; i < m; i++) {
        replace[i] = -1;
        if (!q[i].valid || bd[i] > TH_LOW) { best_idx[i] = -1; continue; }
        const int j = best_idx[i], h = slot[j];
        if (h != -1) {
            const int hbad = h == -2 ? ext_bad[j] : bad[h];
            if (!hbad) replace[i] = h;                                          /* :1089-1090 */
        } else slot[j] = i;                                                     /* :1094-1095 */
        nFused++;
    }
    free(q); free(bd);
    return nFused;
}

/* SearchBySim3 :1104-1328.  pts1[i].valid = pMP && !vbAlreadyMatched1[i] && !isBad (same for 2).
 * T1w / T2w: keyframe poses (4x4); R12 3x3 row-major, t12[3].  match12[n1] out: index in KF2 or -1. */
int oracle_search_by_sim3(const oracle_kp_t *k1, const uint8_t *d1, int n1, const oracle_kp_t *k2, const uint8_t *d2,
                          int n2, const oracle_grid_geom_t *g, const oracle_grid_geom_t *ga, const float *sf, int nlevels,
                          float log_sf,
                          const oracle_cam_t *cam, const float *T1w, const float *T2w, float s12, const float *R12,
                          const float *t12, const oracle_mappoint3d_t *pts1, const uint8_t *pd1,
                          const oracle_mappoint3d_t *pts2, const uint8_t *pd2, float th, int32_t *match12) {
    float sR12[9], sR21[9], t21[3];
    const float a12 = s12, a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            sR12[r * 3 + c] = R12[r * 3 + c] * a12 + 0.0f;                      /* s12*R12 :1121 */
            sR21[r * 3 + c] = R12[c * 3 + r] * a21 + 0.0f;                      /* (1.0/s12)*R12.t() :1122 */
        }
    for (int i = 0; i < 3; i++) {                                               /* t21 = -sR21*t12 :1123 */
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)sR21[i * 3 + k] * (double)t12[k];
        t21[i] = (float)(s * -1.0);
    }
    oracle_window_query_t *q1 = (oracle_window_query_t *)malloc(sizeof(*q1) * (n1 > 0 ? n1 : 1));
    oracle_window_query_t *q2 = (oracle_window_query_t *)malloc(sizeof(*q2) * (n2 > 0 ? n2 : 1));
    for (int dir = 0; dir < 2; dir++) {
        const int np = dir ? n2 : n1;
        const oracle_mappoint3d_t *pts = dir ? pts2 : pts1;
        const float *Tw = dir ? T2w : T1w, *sR = dir ? sR12 : sR21, *tt = dir ? t12 : t21;
        oracle_window_query_t *q = dir ? q2 : q1;
        for (int i = 0; i < np; i++) {
            oracle_mappoint3d_t p = pts[i];
            memset(&q[i], 0, sizeof(q[i]));
            q[i].min_level = q[i].max_level = -1; q[i].ur_tol = -1.0f; q[i].blocks = 1;
            if (!p.valid) continue;
            const float pw[3] = {p.wx, p.wy, p.wz};
            float pa[3], pb[3];
            for (int r = 0; r < 3; r++) pa[r] = row_madd(Tw + 4 * r, 1, pw, Tw[4 * r + 3]);   /* :1161 / :1241 */
            for (int r = 0; r < 3; r++) pb[r] = row_madd(sR + 3 * r, 1, pa, tt[r]);           /* :1162 / :1242 */
            /* from here the shared projection with the camera-frame point as "world" and identity pose;
             * dist3D = cv::norm(p3Dc) is then |p - 0| (:1181 / :1261), no normal test */
            if (pb[2] < 0.0) continue;
            const float invz = (float)(1.0 / (double)pb[2]);
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
            if (!(u >= g->min_x && u < g->max_x && v >= g->min_y && v < g->max_y)) continue;
            const float dist = (float)sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
            if (dist < 0.8f * p.min_distance || dist > 1.2f * p.max_distance) continue;
            const int lvl = predict_scale(p.max_distance, dist, log_sf, nlevels);
            q[i].valid = 1; q[i].u = u; q[i].v = v; q[i].radius = th * sf[lvl];
            q[i].min_level = lvl - 1; q[i].max_level = lvl;
        }
    }
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1)), *m2 = (int32_t *)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    int32_t *b1 = (int32_t *)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1)), *b2 = (int32_t *)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    oracle_best_in_windows(k2, d2, NULL, n2, g, ga, q1, pd1, n1, NULL, nlevels, m1, b1);   /* KF1 points into KF2 */
    oracle_best_in_windows(k1, d1, NULL, n1, g, ga, q2, pd2, n2, NULL, nlevels, m2, b2);   /* KF2 points into KF1 */
    for (int i = 0; i < n1; i++) if (b1[i] > TH_HIGH) m1[i] = -1;
    for (int i = 0; i < n2; i++) if (b2[i] > TH_HIGH) m2[i] = -1;
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {                                           /* :1312-1325 */
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(q1); free(q2); free(m1); free(m2); free(b1); free(b2);
    return nFound;
}

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317) for one map point with n observed
 * descriptors: returns BestIdx, *median_out = BestMedian.  Literal: full distance matrix, sort per row. */
static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int oracle_distinctive_descriptor(const uint8_t *desc, int n, int *median_out) {
    if (n <= 0) { if (median_out) *median_out = 0; return -1; }
    int *dist = (int *)malloc(sizeof(int) * (size_t)n * n), *row = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) {
        dist[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; j++) {
            const int d = oracle_hamming(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
            dist[(size_t)i * n + j] = d; dist[(size_t)j * n + i] = d;
        }
    }
    int BestMedian = 0x7FFFFFFF, BestIdx = 0;
    for (int i = 0; i < n; i++) {
        memcpy(row, dist + (size_t)i * n, sizeof(int) * n);
        qsort(row, n, sizeof(int), int_cmp);
        const int median = row[(size_t)(0.5 * (n - 1))];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    free(dist); free(row);
    if (median_out) *median_out = BestMedian;
    return BestIdx;
}

/* Frame::isInFrustum (src/Frame.cc:284-340) for m map points; PredictScale with the C
 * library's logf exactly as src/MapPoint.cc:414-429.  pts[i].valid = the caller's skip conditions
 * (src/Tracking.cc:1312-1316); obs[i] is copied to the output. */
void oracle_is_in_frustum(const oracle_mappoint3d_t *pts, const int32_t *obs, int m, const float *Tcw,
                          const oracle_cam_t *cam, const oracle_grid_geom_t *g, float viewingCosLimit, float log_sf,
                          int nlevels, oracle_mp_t *out) {
    float R[9], t[3], Ow[3];
    pose_parts(Tcw, R, t, Ow);                              /* mRcw, mtcw, mOw (src/Frame.cc:272-279) */
    for (int i = 0; i < m; i++) {
        oracle_mp_t *o = &out[i];
        memset(o, 0, sizeof(*o));
        o->observations = obs ? obs[i] : 0;
        if (!pts[i].valid) continue;
        const float P[3] = {pts[i].wx, pts[i].wy, pts[i].wz};
        float Pc[3];
        for (int r = 0; r < 3; r++) Pc[r] = row_madd(R + 3 * r, 1, P, t[r]);
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = cam->fx * Pc[0] * invz + cam->cx;
        const float v = cam->fy * Pc[1] * invz + cam->cy;
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        const float maxDistance = 1.2f * pts[i].max_distance, minDistance = 0.8f * pts[i].min_distance;
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = (float)(((double)PO[0] * pts[i].nx + (double)PO[1] * pts[i].ny + (double)PO[2] * pts[i].nz) / dist);
        if (viewCos < viewingCosLimit) continue;
        o->in_view = 1; o->proj_x = u; o->proj_xr = u - cam->mbf * invz; o->proj_y = v;
        o->level = predict_scale(pts[i].max_distance, dist, log_sf, nlevels);
        o->view_cos = viewCos;
    }
}
