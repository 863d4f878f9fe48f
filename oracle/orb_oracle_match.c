/*
 * orb_oracle_match.c — CPU ORACLE (test infrastructure, never on the product path):
 * restatement of the Hamming matchers of the reference on flat arrays.
 *   ORBmatcher::DescriptorDistance            src/ORBmatcher.cc:1649-1665
 *   ORBmatcher::ComputeThreeMaxima            src/ORBmatcher.cc:1603-1644
 *   ORBmatcher::SearchForInitialization       src/ORBmatcher.cc:405-520
 *   ORBmatcher::SearchByProjection(F, MPs)    src/ORBmatcher.cc:45-129
 *   ORBmatcher::SearchByProjection(F, F)      src/ORBmatcher.cc:1330-1472
 *   Frame::ComputeStereoMatches               src/Frame.cc:481-655
 *   Frame grid build / query                  src/Frame.cc:230-245, 342-407
 * PARITY UNPINNED (see orb_oracle.h).  Compile with -ffp-contract=off.
 */
#include "orb_oracle.h"
#include <math.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30 /* src/ORBmatcher.cc:37-39 */
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64 /* include/Frame.h:37-38 */

int oracle_hamming(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void oracle_three_maxima(const int32_t *sizes, int L, int *ind1, int *ind2, int *ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1; /* callers initialise them to -1 (:491-493, :1452-1454) */
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ------------------------------------------------------------------ grid */
struct oracle_grid {
    oracle_grid_geom_t g;
    const oracle_kp_t *kps; int n;
    int32_t *cell_start; /* [COLS*ROWS+1], cell index = ix*ROWS+iy */
    int32_t *items;
};
static int pos_in_grid(const oracle_grid_geom_t *g, const oracle_kp_t *kp, int *px, int *py) {
    *px = (int)roundf((kp->x - g->min_x) * g->inv_w);
    *py = (int)roundf((kp->y - g->min_y) * g->inv_h);
    if (*px < 0 || *px >= FRAME_GRID_COLS || *py < 0 || *py >= FRAME_GRID_ROWS) return 0;
    return 1;
}
oracle_grid_t *oracle_grid_build(const oracle_kp_t *kps, int n, const oracle_grid_geom_t *geom) {
    oracle_grid_t *g = (oracle_grid_t *)calloc(1, sizeof(*g));
    g->g = *geom; g->kps = kps; g->n = n;
    const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    g->cell_start = (int32_t *)calloc(NC + 1, sizeof(int32_t));
    g->items = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t *cnt = (int32_t *)calloc(NC, sizeof(int32_t));
    int px, py;
    for (int i = 0; i < n; i++) if (pos_in_grid(geom, &kps[i], &px, &py)) cnt[px * FRAME_GRID_ROWS + py]++;
    for (int c = 0; c < NC; c++) g->cell_start[c + 1] = g->cell_start[c] + cnt[c];
    memset(cnt, 0, sizeof(int32_t) * NC);
    for (int i = 0; i < n; i++)
        if (pos_in_grid(geom, &kps[i], &px, &py)) {
            int c = px * FRAME_GRID_ROWS + py;
            g->items[g->cell_start[c] + cnt[c]++] = i; /* insertion order = index order */
        }
    free(cnt);
    return g;
}
/* KeyFrame keeps the Frame's cell lists but queries them with its own int-truncated bounds
 * (include/KeyFrame.h:199-202 `const int mnMinX..`, src/KeyFrame.cc:41,53,577-589) */
void oracle_grid_set_query_geom(oracle_grid_t *g, const oracle_grid_geom_t *q) { if (g && q) g->g = *q; }
void oracle_grid_free(oracle_grid_t *g) { if (g) { free(g->cell_start); free(g->items); free(g); } }

int oracle_grid_query(const oracle_grid_t *gr, float x, float y, float r, int minLevel, int maxLevel,
                      int32_t *out, int cap) {
    const oracle_grid_geom_t *g = &gr->g;
    int n = 0;
    int nMinCellX = (int)floorf((x - g->min_x - r) * g->inv_w); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - g->min_x + r) * g->inv_w); if (nMaxCellX > FRAME_GRID_COLS - 1) nMaxCellX = FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - g->min_y - r) * g->inv_h); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - g->min_y + r) * g->inv_h); if (nMaxCellY > FRAME_GRID_ROWS - 1) nMaxCellY = FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            int c = ix * FRAME_GRID_ROWS + iy;
            for (int j = gr->cell_start[c]; j < gr->cell_start[c + 1]; j++) {
                const oracle_kp_t *kp = &gr->kps[gr->items[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = gr->items[j]; n++; }
            }
        }
    return n;
}

/* --------------------------------------------------------------- stereo */
typedef struct { int dist; int idx; } distidx_t;
static int distidx_cmp(const void *a, const void *b) {
    const distidx_t *x = (const distidx_t *)a, *y = (const distidx_t *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;
}
int oracle_stereo_match(const oracle_kp_t *kl, const uint8_t *dl, int N,
                        const oracle_kp_t *kr, const uint8_t *dr, int Nr,
                        const oracle_img_t *pyr_l, const oracle_img_t *pyr_r, int nlevels,
                        const float *sf, const float *isf,
                        float mbf, float mb, float *uright, float *depth) {
    (void)nlevels;
    for (int i = 0; i < N; i++) { uright[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyr_l[0].h;
    /* row table; vectors restated as count + fill */
    int32_t *rstart = (int32_t *)calloc(nRows + 1, sizeof(int32_t));
    for (int pass = 0; pass < 1; pass++)
        for (int iR = 0; iR < Nr; iR++) {
            const float kpY = kr[iR].y, r = 2.0f * sf[kr[iR].octave];
            const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
            for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rstart[yi + 1]++;
        }
    for (int y = 0; y < nRows; y++) rstart[y + 1] += rstart[y];
    int32_t *ritems = (int32_t *)malloc(sizeof(int32_t) * (rstart[nRows] > 0 ? rstart[nRows] : 1));
    int32_t *rfill = (int32_t *)calloc(nRows, sizeof(int32_t));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kr[iR].y, r = 2.0f * sf[kr[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) ritems[rstart[yi] + rfill[yi]++] = iR;
    }
    free(rfill);
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    distidx_t *vDistIdx = (distidx_t *)malloc(sizeof(distidx_t) * (N > 0 ? N : 1));
    int nd = 0;
    for (int iL = 0; iL < N; iL++) {
        const oracle_kp_t *kpL = &kl[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue; /* reference: OOB */
        const int c0 = rstart[row], c1 = rstart[row + 1];
        if (c0 == c1) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        const uint8_t *dL = dl + 32 * (size_t)iL;
        for (int iC = c0; iC < c1; iC++) {
            const int iR = ritems[iC];
            const oracle_kp_t *kpR = &kr[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = oracle_hamming(dL, dr + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = isf[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const oracle_img_t *imL = &pyr_l[kpL->octave], *imR = &pyr_r[kpL->octave];
            const int cy = (int)scaledvL, cxl = (int)scaleduL;
            float vDists[11];
            int bestDistS = INT_MAX, bestincR = 0;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR->w) continue;
            const int cL = imL->ptr[(size_t)cy * imL->stride + cxl];
            for (int incR = -L; incR <= +L; incR++) {
                const int cxr = (int)(scaleduR0 + incR);
                const int cR = imR->ptr[(size_t)cy * imR->stride + cxr];
                double s = 0; /* cv::norm(NORM_L1) on CV_32F accumulates in double */
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        float a = (float)imL->ptr[(size_t)(cy + dy) * imL->stride + cxl + dx] - (float)cL;
                        float b = (float)imR->ptr[(size_t)(cy + dy) * imR->stride + cxr + dx] - (float)cR;
                        s += fabsf(a - b);
                    }
                float dist = (float)s;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = sf[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)(uL - 0.01); }
                depth[iL] = mbf / disparity;
                uright[iL] = bestuR;
                vDistIdx[nd].dist = bestDistS; vDistIdx[nd].idx = iL; nd++;
            }
        }
    }
    int kept = nd;
    if (nd > 0) { /* reference: UB on an empty vector (:642) */
        qsort(vDistIdx, nd, sizeof(distidx_t), distidx_cmp);
        const float median = (float)vDistIdx[nd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            uright[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1; kept--;
        }
    }
    free(vDistIdx); free(rstart); free(ritems);
    return kept;
}

/* ------------------------------------------------ SearchForInitialization */
int oracle_search_for_initialization(const oracle_kp_t *k1, const uint8_t *d1, int n1,
                                     const oracle_kp_t *k2, const uint8_t *d2, int n2,
                                     const oracle_grid_geom_t *g2, float *prev, int32_t *m12,
                                     int window, float nnratio, int check_ori) {
    int nmatches = 0;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * HISTO_LENGTH * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hn[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    int32_t *vMatchedDistance = (int32_t *)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    int32_t *m21 = (int32_t *)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) { vMatchedDistance[i] = INT_MAX; m21[i] = -1; }
    oracle_grid_t *grid = oracle_grid_build(k2, n2, g2);
    for (int i1 = 0; i1 < n1; i1++) {
        int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        int nc = oracle_grid_query(grid, prev[2 * i1], prev[2 * i1 + 1], (float)window, level1, level1, idx, n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            int i2 = idx[c];
            int dist = oracle_hamming(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { m12[m21[bestIdx2]] = -1; nmatches--; }
                m12[i1] = bestIdx2; m21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist; nmatches++;
                if (check_ori) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    hist[(size_t)bin * n1 + hn[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) {
                int idx1 = hist[(size_t)i * n1 + j];
                if (m12[idx1] >= 0) { m12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (m12[i1] >= 0) { prev[2 * i1] = k2[m12[i1]].x; prev[2 * i1 + 1] = k2[m12[i1]].y; }
    oracle_grid_free(grid);
    free(hist); free(vMatchedDistance); free(m21); free(idx);
    return nmatches;
}

/* ------------------------------------- SearchByProjection(Frame, MapPoints) */
static float radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }

/* slot holder test of :87-89 / :1405-1407: frame_mp[idx] >= 0 -> map point of the list,
 * -2 -> an external holder whose Observations() is ext_obs[idx], -1 -> empty */
static int slot_blocked(const int32_t *holder, const int32_t *ext_obs, int idx,
                        const int32_t *obs_of, int obs_stride_bytes) {
    int hm = holder[idx];
    if (hm == -1) return 0;
    if (hm == -2) return ext_obs && ext_obs[idx] > 0;
    return *(const int32_t *)((const char *)obs_of + (size_t)hm * obs_stride_bytes) > 0;
}

int oracle_search_by_projection_mp(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                                   const oracle_grid_geom_t *g, const float *sf,
                                   const oracle_mp_t *mps, const uint8_t *mp_desc, int m,
                                   int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio) {
    int nmatches = 0;
    const int bFactor = th != 1.0;
    oracle_grid_t *grid = oracle_grid_build(kun, n, g);
    int32_t *idxs = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int iMP = 0; iMP < m; iMP++) {
        const oracle_mp_t *p = &mps[iMP];
        if (!p->in_view) continue;
        const int lvl = p->level;
        float r = radius_by_viewing_cos(p->view_cos);
        if (bFactor) r *= th;
        int nc = oracle_grid_query(grid, p->proj_x, p->proj_y, r * sf[lvl], lvl - 1, lvl, idxs, n);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = idxs[c];
            if (slot_blocked(frame_mp, ext_obs, idx, &mps[0].observations, sizeof(oracle_mp_t))) continue;
            if (uright[idx] > 0) {
                const float er = fabsf(p->proj_xr - uright[idx]);
                if (er > r * sf[lvl]) continue;
            }
            const int dist = oracle_hamming(mp_desc + 32 * (size_t)iMP, desc + 32 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
                bestLevel = kun[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) { bestLevel2 = kun[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            frame_mp[bestIdx] = iMP;
            nmatches++;
        }
    }
    free(idxs); oracle_grid_free(grid);
    return nmatches;
}

/* ---------------------------------------- SearchByProjection(Frame, Frame) */
int oracle_search_by_projection_frame(const oracle_kp_t *kun, const uint8_t *desc, const float *uright, int n,
                                      const oracle_grid_geom_t *g, const float *sf,
                                      const oracle_cam_t *cam, const float *Tc, const float *Tl,
                                      const oracle_lastpt_t *last, const uint8_t *last_desc, int nlast,
                                      int32_t *cur_mp, const int32_t *ext_obs,
                                      float th, int mono, int check_ori) {
    int nmatches = 0;
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * HISTO_LENGTH * (size_t)(nlast > 0 ? nlast : 1));
    int32_t hn[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    /* twc = -Rcw^T tcw ; tlc = Rlw twc + tlw.  cv::Mat products of CV_32F go through
     * cv::gemm -> GEMMSingleMul<float,double>: products and sums in double, one rounding
     * to float of (sum*alpha + c*beta)  [OpenCV matmul.cpp, external] */
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tc[k * 4 + i] * (double)Tc[k * 4 + 3];
        twc[i] = (float)(s * -1.0);
    }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tl[i * 4 + k] * (double)twc[k];
        tlc[i] = (float)(s + (double)Tl[i * 4 + 3]);
    }
    const int bForward = tlc[2] > cam->mb && !mono;
    const int bBackward = -tlc[2] > cam->mb && !mono;
    oracle_grid_t *grid = oracle_grid_build(kun, n, g);
    int32_t *idxs = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < nlast; i++) {
        const oracle_lastpt_t *p = &last[i];
        if (!p->has_mp) continue;
        float x3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            s += (double)Tc[r * 4 + 0] * (double)p->wx; s += (double)Tc[r * 4 + 1] * (double)p->wy;
            s += (double)Tc[r * 4 + 2] * (double)p->wz;
            x3[r] = (float)(s + (double)Tc[r * 4 + 3]);
        }
        const float xc = x3[0], yc = x3[1];
        const float invzc = (float)(1.0 / x3[2]);
        if (invzc < 0) continue;
        float u = cam->fx * xc * invzc + cam->cx;
        float v = cam->fy * yc * invzc + cam->cy;
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        int nLastOctave = p->octave;
        float radius = th * sf[nLastOctave];
        int nc;
        if (bForward) nc = oracle_grid_query(grid, u, v, radius, nLastOctave, -1, idxs, n);
        else if (bBackward) nc = oracle_grid_query(grid, u, v, radius, 0, nLastOctave, idxs, n);
        else nc = oracle_grid_query(grid, u, v, radius, nLastOctave - 1, nLastOctave + 1, idxs, n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = idxs[c];
            if (slot_blocked(cur_mp, ext_obs, i2, &last[0].observations, sizeof(oracle_lastpt_t))) continue;
            if (uright[i2] > 0) {
                const float ur = u - cam->mbf * invzc;
                const float er = fabsf(ur - uright[i2]);
                if (er > radius) continue;
            }
            const int dist = oracle_hamming(last_desc + 32 * (size_t)i, desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_mp[bestIdx2] = i;
            nmatches++;
            if (check_ori) {
                float rot = p->angle - kun[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[(size_t)bin * nlast + hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hn[i]; j++) { cur_mp[hist[(size_t)i * nlast + j]] = -1; nmatches--; }
    }
    free(hist); free(idxs); oracle_grid_free(grid);
    return nmatches;
}

/* ---------------------- SearchByProjection(Frame, KeyFrame, sAlreadyFound, th, ORBdist) */
void oracle_kf_window_queries(const oracle_kfpoint_t *kf, int m, const oracle_grid_geom_t *g, const float *sf,
                              int nlevels, float mfLogScaleFactor, const oracle_cam_t *cam, const float *Tc, float th,
                              oracle_window_query_t *q) {
    /* Ow = -Rcw^T tcw (:1478-1480), cv::gemm double accumulation */
    float Ow[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tc[k * 4 + i] * (double)Tc[k * 4 + 3];
        Ow[i] = (float)(s * -1.0);
    }
    for (int i = 0; i < m; i++) {
        oracle_window_query_t *o = &q[i];
        memset(o, 0, sizeof(*o));
        o->min_level = o->max_level = -1; o->ur_tol = -1.0f; o->blocks = 1; o->angle = kf[i].angle;
        if (!kf[i].valid) continue;
        float x3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            s += (double)Tc[r * 4 + 0] * (double)kf[i].wx; s += (double)Tc[r * 4 + 1] * (double)kf[i].wy;
            s += (double)Tc[r * 4 + 2] * (double)kf[i].wz;
            x3[r] = (float)(s + (double)Tc[r * 4 + 3]);
        }
        const float xc = x3[0], yc = x3[1];
        const float invzc = (float)(1.0 / x3[2]);
        const float u = cam->fx * xc * invzc + cam->cx;
        const float v = cam->fy * yc * invzc + cam->cy;
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        /* PO = x3Dw - Ow; dist3D = cv::norm(PO): L2 of a CV_32F Mat accumulates in double */
        const float p0 = kf[i].wx - Ow[0], p1 = kf[i].wy - Ow[1], p2 = kf[i].wz - Ow[2];
        const float dist3D = (float)sqrt((double)p0 * p0 + (double)p1 * p1 + (double)p2 * p2);
        const float maxDistance = 1.2f * kf[i].max_distance, minDistance = 0.8f * kf[i].min_distance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        /* MapPoint::PredictScale(dist3D, &CurrentFrame) (src/MapPoint.cc:414-429) */
        const float ratio = kf[i].max_distance / dist3D;
        int nScale = (int)ceilf(logf(ratio) / mfLogScaleFactor);
        if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
        o->valid = 1; o->u = u; o->v = v; o->radius = th * sf[nScale];
        o->min_level = nScale - 1; o->max_level = nScale + 1;
    }
}

int oracle_search_by_projection_kf(const oracle_kp_t *kun, const uint8_t *desc, int n, const oracle_grid_geom_t *g,
                                   const float *sf, int nlevels, float mfLogScaleFactor, const oracle_cam_t *cam,
                                   const float *Tc, const oracle_kfpoint_t *kf, const uint8_t *kf_desc, int m,
                                   int32_t *cur_mp, float th, int ORBdist, int check_ori) {
    int nmatches = 0;
    oracle_window_query_t *q = (oracle_window_query_t *)malloc(sizeof(*q) * (m > 0 ? m : 1));
    oracle_kf_window_queries(kf, m, g, sf, nlevels, mfLogScaleFactor, cam, Tc, th, q);
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * HISTO_LENGTH * (size_t)(m > 0 ? m : 1));
    int32_t hn[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    oracle_grid_t *grid = oracle_grid_build(kun, n, g);
    int32_t *idxs = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < m; i++) {
        if (!q[i].valid) continue;
        const int nc = oracle_grid_query(grid, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, idxs, n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = idxs[c];
            if (cur_mp[i2] != -1) continue;                 /* if(CurrentFrame.mvpMapPoints[i2]) continue;  (:1543) */
            const int dist = oracle_hamming(kf_desc + 32 * (size_t)i, desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            cur_mp[bestIdx2] = i;
            nmatches++;
            if (check_ori) {
                float rot = kf[i].angle - kun[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[(size_t)bin * m + hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hn[i]; j++) { cur_mp[hist[(size_t)i * m + j]] = -1; nmatches--; }
    }
    free(hist); free(idxs); free(q); oracle_grid_free(grid);
    return nmatches;
}
