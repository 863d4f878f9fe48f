/*
 * orb_oracle.h — CPU ORACLE for the ORB front-end + Hamming matchers.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (kimwin2/ORB_SLAM2v2-1: src/ORBextractor.cc, src/ORBmatcher.cc,
 * src/Frame.cc) plus the OpenCV 2.4.11/3.2 primitives it calls (cv::FAST,
 * cv::resize INTER_LINEAR, cv::GaussianBlur, cv::fastAtan2, cvRound), which are
 * NOT in the reference tree.  Nothing under orb_slam2v2-1_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, as the checker.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be
 * compiled here (OpenCV absent), so this oracle is pinned only by analytically
 * derived known-answer tables and hand-built images.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint field order (reference: include/BoostArchiver.h:47-57); 28 bytes */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} oracle_kp_t;

/* FAST candidate before the quad-tree: coordinates relative to minBorder (16,16) */
typedef struct { int32_t x, y, score; } oracle_cand_t;

typedef struct orb_oracle orb_oracle_t;

orb_oracle_t *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void oracle_destroy(orb_oracle_t *o);

/* ctor tables (src/ORBextractor.cc:410-470) */
const float *oracle_scale_factors(const orb_oracle_t *o);
const float *oracle_inv_scale_factors(const orb_oracle_t *o);
const float *oracle_level_sigma2(const orb_oracle_t *o);
const float *oracle_inv_level_sigma2(const orb_oracle_t *o);
const int32_t *oracle_features_per_level(const orb_oracle_t *o);
const int32_t *oracle_umax(const orb_oracle_t *o);

/* ORBextractor::operator() (src/ORBextractor.cc:1043-1105). Returns keypoint count
 * (<= cap, or -needed if cap is too small), -1000 on bad arguments. */
int oracle_extract(orb_oracle_t *o, const uint8_t *img, int w, int h, int stride,
                   oracle_kp_t *kps, uint8_t *desc, int cap);

/* stage introspection after oracle_extract (for staged parity tests) */
int oracle_pyramid_level(const orb_oracle_t *o, int level, const uint8_t **padded, int *w, int *h,
                         int *padded_stride); /* padded points at the (w+38)x(h+38) buffer */
int oracle_level_candidates(const orb_oracle_t *o, int level, const oracle_cand_t **c);
int oracle_level_keypoints(const orb_oracle_t *o, int level, const oracle_cand_t **c);
const uint8_t *oracle_blurred_level(const orb_oracle_t *o, int level); /* w*h, stride w */
/* per-stage wall time of the last oracle_extract, seconds:
 * [0] pyramid [1] FAST cells [2] quad-tree [3] orientation [4] blur [5] descriptors */
const double *oracle_stage_seconds(const orb_oracle_t *o);

/* primitives exposed for unit tests */
float oracle_fast_atan2(float y, float x);
int oracle_cv_round(double v);
/* cos / sin of the descriptor angle (src/ORBextractor.cc:113): mode 0 restated glibc cosf/sinf (default), 1 host libm
 * cosf/sinf, 2 (float)cos((double)) - see orb_oracle_extract.c.  The mode is process-global: tests only. */
void oracle_set_sincos_mode(int mode);
void oracle_sincosf(float angle, float *s, float *c);
void oracle_sincosf_array(const float *a, int n, float *s, float *c);
int oracle_fast_score(const uint8_t *p, int stride, int threshold); /* cornerScore<16>; p = centre */
int oracle_fast_is_corner(const uint8_t *p, int stride, int threshold);
/* cv::FAST(img, thr, nms=true) on a w x h sub-image; writes up to cap (x,y,score), returns n */
int oracle_fast_detect(const uint8_t *img, int stride, int w, int h, int threshold,
                       oracle_cand_t *out, int cap);
void oracle_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride);
void oracle_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
/* Flavours of the column-pass rounding of cv::GaussianBlur on 8U (OpenCV <= 3.3 filter.cpp; see orb_oracle_extract.c):
 * HALF_UP = the scalar FixedPtCastEx everywhere (the default, what rounds 1-3 encoded), SSE2 = SymmColumnVec_32s8u's
 * round-half-to-even for the columns x < (w & ~3), scalar for the tail.  Both are hypotheses until reference vectors arrive. */
#define ORACLE_GAUSS_HALF_UP 0
#define ORACLE_GAUSS_SSE2 1
/* FIXED_TAPS = OpenCV >= 3.4.1's bit-exact fixed-point Gaussian: HALF_UP's arithmetic on the Q8 taps of the build (centre first; 55 49 34 18
 * reproduces HALF_UP; releases that diffuse the rounding error use taps adding up to 256) - oracle_set_gauss_taps / oracle_gaussian_blur7_taps */
#define ORACLE_GAUSS_FIXED_TAPS 2
void oracle_gaussian_blur7_taps(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, const int taps[4]);
int oracle_set_gauss_taps(orb_oracle_t *o, int k0, int k1, int k2, int k3);
void oracle_gaussian_blur7_flavour(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int flavour);
int oracle_set_gauss_flavour(orb_oracle_t *o, int flavour);
int oracle_get_gauss_flavour(const orb_oracle_t *o);
int oracle_gauss_round_half_even(int sum);
int oracle_gauss_round_sse2_literal(int s0, int s1, int s2, int s3, int s4, int s5, int s6);
/* DistributeOctTree on a candidate list (coords relative to minBorder); region size
 * width x height = (maxX-minX) x (maxY-minY). Returns number of selected, in list order. */
int oracle_distribute_octtree(const oracle_cand_t *cands, int n, int width, int height, int N,
                              oracle_cand_t *out, int cap);

/* ---- matchers ---- */
/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1649-1665) */
int oracle_hamming(const uint8_t *a, const uint8_t *b);

/* Frame::ComputeStereoMatches (src/Frame.cc:481-655). pyramids given as arrays of
 * inner-ROI pointers/strides per level. Outputs uright[N], depth[N] (-1 = none).
 * Returns number of surviving matches. */
typedef struct {
    const uint8_t *ptr; int32_t w, h, stride;
} oracle_img_t;
int oracle_stereo_match(const oracle_kp_t *kl, const uint8_t *dl, int nl,
                        const oracle_kp_t *kr, const uint8_t *dr, int nr,
                        const oracle_img_t *pyr_l, const oracle_img_t *pyr_r, int nlevels,
                        const float *scale_factors, const float *inv_scale_factors,
                        float mbf, float mb, float *uright, float *depth);

/* Frame lookup grid (src/Frame.cc:230-245, 342-407; include/Frame.h:37-38) */
typedef struct {
    float min_x, min_y, max_x, max_y;          /* mnMinX.. */
    float inv_w, inv_h;                         /* mfGridElementWidthInv/HeightInv */
} oracle_grid_geom_t;
typedef struct oracle_grid oracle_grid_t;
oracle_grid_t *oracle_grid_build(const oracle_kp_t *kps, int n, const oracle_grid_geom_t *g);
void oracle_grid_free(oracle_grid_t *g);
void oracle_grid_set_query_geom(oracle_grid_t *g, const oracle_grid_geom_t *query_geom);
int oracle_grid_query(const oracle_grid_t *g, float x, float y, float r, int min_level, int max_level,
                      int32_t *out, int cap);

/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1603-1644) on bin sizes */
void oracle_three_maxima(const int32_t *sizes, int L, int *i1, int *i2, int *i3);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520).
 * prev_matched[2*n1] is updated in place; matches12[n1] out. Returns nmatches. */
int oracle_search_for_initialization(const oracle_kp_t *k1, const uint8_t *d1, int n1,
                                     const oracle_kp_t *k2, const uint8_t *d2, int n2,
                                     const oracle_grid_geom_t *g2, float *prev_matched,
                                     int32_t *matches12, int window, float nnratio, int check_ori);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:45-129) on gathered map-point arrays.
 * mp_*: per map point: in_view flag (mbTrackInView && !isBad), proj x/y/xr, level, viewcos,
 * 32-byte descriptor, observations (>0 means a frame slot holding it blocks).
 * frame_mp[n]: index of the map point currently held by keypoint i (-1 none) — in/out.
 * frame_mp_obs[n]: Observations() of a pre-existing holder not in the mp list (only >0 matters)
 *                  for entries where frame_mp[i] == -2 (external holder). */
typedef struct {
    int32_t in_view; float proj_x, proj_y, proj_xr; int32_t level; float view_cos; int32_t observations;
} oracle_mp_t;
int oracle_search_by_projection_mp(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                                   const oracle_grid_geom_t *g, const float *scale_factors,
                                   const oracle_mp_t *mps, const uint8_t *mp_desc, int m,
                                   int32_t *frame_mp, const int32_t *frame_ext_obs,
                                   float th, float nnratio);

/* ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono)
 * (src/ORBmatcher.cc:1330-1472). last_*: per last-frame keypoint: has_mp (non-null &&
 * !outlier), world pos, descriptor of the map point, observations, octave + angle of
 * last keypoint. Tcw_cur / Tcw_last are row-major 4x4 float. cur_mp[n] in/out holds the
 * index i of the last-frame keypoint whose map point is assigned (-1 none, -2 external holder). */
typedef struct {
    int32_t has_mp; float wx, wy, wz; int32_t observations; int32_t octave; float angle;
} oracle_lastpt_t;
typedef struct { float fx, fy, cx, cy, mbf, mb; } oracle_cam_t;
int oracle_search_by_projection_frame(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                                      const oracle_grid_geom_t *g, const float *scale_factors,
                                      const oracle_cam_t *cam, const float *Tcw_cur, const float *Tcw_last,
                                      const oracle_lastpt_t *last, const uint8_t *last_desc, int nlast,
                                      int32_t *cur_mp, const int32_t *cur_ext_obs,
                                      float th, int mono, int check_ori);

/* ---- ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1474-1601), used by Tracking::Relocalization.
 * kf[i]: map point i of pKF->GetMapPointMatches(): valid = pMP && !isBad() && !sAlreadyFound.count(pMP). */
typedef struct {
    int32_t valid; float wx, wy, wz; float max_distance, min_distance; /* mfMaxDistance, mfMinDistance */
    float angle;                                                        /* pKF->mvKeysUn[i].angle */
} oracle_kfpoint_t;
/* window query the projection stage produces for the matcher (same layout as orbm_window_query_t) */
typedef struct {
    int32_t valid; float u, v, radius; int32_t min_level, max_level; float angle; int32_t blocks;
    float ur_c, ur_tol;
} oracle_window_query_t;
/* projection + PredictScale (src/MapPoint.cc:414-429) -> one window query per keyframe point */
void oracle_kf_window_queries(const oracle_kfpoint_t *kf, int m, const oracle_grid_geom_t *g, const float *scale_factors,
                              int nlevels, float log_scale_factor, const oracle_cam_t *cam, const float *Tcw_cur,
                              float th, oracle_window_query_t *q);
/* cur_mp[n] in/out: -1 empty, -2 held by a map point outside the list (blocks), >= 0 index into kf */
int oracle_search_by_projection_kf(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                                   const float *scale_factors, int nlevels, float log_scale_factor,
                                   const oracle_cam_t *cam, const float *Tcw_cur, const oracle_kfpoint_t *kf,
                                   const uint8_t *kf_desc, int m, int32_t *cur_mp, float th, int orb_dist, int check_ori);

/* ---- SURVEY §8(f) rank 1, KeyFrame side (orb_oracle_kfmatch.c): SearchByProjection(KeyFrame*, Scw, ...)
 * (src/ORBmatcher.cc:290-403), Fuse x2 (:827-1102), SearchBySim3 (:1104-1328). */
typedef struct {
    int32_t valid;                       /* the caller's static skip conditions (null / bad / already found) */
    float wx, wy, wz;                    /* GetWorldPos() */
    float nx, ny, nz;                    /* GetNormal() */
    float max_distance, min_distance;    /* mfMaxDistance, mfMinDistance */
} oracle_mappoint3d_t;
void oracle_decompose_sim3(const float *Scw16, float *Rcw9, float *tcw3, float *Ow3);
void oracle_sim3_window_queries(const oracle_mappoint3d_t *pts, int m, const oracle_grid_geom_t *g, const float *sf,
                                int nlevels, float log_sf, const oracle_cam_t *cam, const float *Scw, float th,
                                oracle_window_query_t *q);
void oracle_pose_window_queries(const oracle_mappoint3d_t *pts, int m, const oracle_grid_geom_t *g, const float *sf,
                                int nlevels, float log_sf, const oracle_cam_t *cam, const float *Tcw, float th,
                                oracle_window_query_t *q);
/* g: the KeyFrame's query bounds; ga: the bounds the keypoints were binned with (NULL = g) */
void oracle_best_in_windows(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                            const oracle_grid_geom_t *g, const oracle_grid_geom_t *ga, const oracle_window_query_t *q, const uint8_t *qdesc, int m,
                            const float *inv_sigma2, int nlevels, int32_t *best_idx, int32_t *best_dist);
int oracle_search_by_projection_sim3(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                                     const oracle_grid_geom_t *ga,
                                     const float *sf, int nlevels, float log_sf, const oracle_cam_t *cam,
                                     const float *Scw, const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m,
                                     int32_t *matched, int th);
int oracle_fuse(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n, const oracle_grid_geom_t *g,
                const oracle_grid_geom_t *ga,
                const float *sf, const float *inv_sigma2, int nlevels, float log_sf, const oracle_cam_t *cam,
                const float *Tcw, const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m, int32_t *bad,
                int32_t *in_kf, int32_t *obs, int32_t *slot, const int32_t *ext_obs, int32_t *ext_bad, float th,
                int32_t *best_idx, int32_t *action);
int oracle_fuse_sim3(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                     const oracle_grid_geom_t *ga, const float *sf,
                     int nlevels, float log_sf, const oracle_cam_t *cam, const float *Scw,
                     const oracle_mappoint3d_t *pts, const uint8_t *pdesc, int m, const int32_t *bad, int32_t *slot,
                     const int32_t *ext_bad, float th, int32_t *best_idx, int32_t *replace);
int oracle_search_by_sim3(const oracle_kp_t *k1, const uint8_t *d1, int n1, const oracle_kp_t *k2, const uint8_t *d2,
                          int n2, const oracle_grid_geom_t *g, const oracle_grid_geom_t *ga, const float *sf, int nlevels,
                          float log_sf,
                          const oracle_cam_t *cam, const float *T1w, const float *T2w, float s12, const float *R12,
                          const float *t12, const oracle_mappoint3d_t *pts1, const uint8_t *pd1,
                          const oracle_mappoint3d_t *pts2, const uint8_t *pd2, float th, int32_t *match12);

/* MapPoint::PredictScale (src/MapPoint.cc:414-429) given ratio = mfMaxDistance/currentDist */
int oracle_predict_scale_ratio(float ratio, float log_sf, int nlevels);
/* Frame::isInFrustum (src/Frame.cc:284-340) for m map points */
void oracle_is_in_frustum(const oracle_mappoint3d_t *pts, const int32_t *obs, int m, const float *Tcw,
                          const oracle_cam_t *cam, const oracle_grid_geom_t *g, float viewingCosLimit, float log_sf,
                          int nlevels, oracle_mp_t *out);
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317), one map point; returns BestIdx (-1: no rows) */
int oracle_distinctive_descriptor(const uint8_t *desc, int n, int *median_out);

/* ---- SURVEY §8(f) rank 3 (orb_oracle_bow.c): DBoW2 vocabulary transform and ORBmatcher::SearchByBoW x2 */
typedef struct oracle_voc oracle_voc_t;
oracle_voc_t *oracle_voc_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent,
                                const uint8_t *is_leaf, const uint8_t *desc, const double *weight);
void oracle_voc_free(oracle_voc_t *v);
int oracle_voc_words(const oracle_voc_t *v);
void oracle_voc_transform_one(const oracle_voc_t *v, const uint8_t *feature, int levelsup, int32_t *word_id, double *weight,
                              int32_t *nid);
int oracle_voc_transform(const oracle_voc_t *v, const uint8_t *features, int n, int levelsup, int32_t *bow_word,
                         double *bow_value, int32_t *fv_node, int32_t *fv_start, int32_t *fv_items, int *nfv_out);
int oracle_search_for_triangulation(const oracle_kp_t *k1, const uint8_t *qd, const uint8_t *qf, int nq, const oracle_kp_t *k2,
                                    const uint8_t *cd, const uint8_t *cf, int nc, const int32_t *nqs, const int32_t *qit,
                                    const int32_t *ncs, const int32_t *cit, int nnodes, const float *F12, float ex, float ey,
                                    const float *sf, const float *sigma2, int th_low, int check_ori, int32_t *match_q);
int oracle_search_by_bow(const uint8_t *qd, const float *qa, const uint8_t *qv, int nq, const uint8_t *cd, const float *ca,
                         const uint8_t *cv, int nc, const int32_t *nqs, const int32_t *qit, const int32_t *ncs,
                         const int32_t *cit, int nnodes, int th_low, int strict_lt, float nnratio, int check_ori,
                         int32_t *match_q);

#ifdef __cplusplus
}
#endif
#endif
