/*
 * orb_oracle_bow.c — CPU ORACLE (test infrastructure, never on the product path): restatement of
 *   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1230-1271
 *   TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup)  :1147-1214 (TF_IDF / TF / IDF / BINARY, L1 / L2)
 *   BowVector::addWeight / addIfNotExist / normalize                          Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-85
 *   FeatureVector::addFeature                                                 Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:29-43
 *   FORB::distance                                                            Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101
 *   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)                           src/ORBmatcher.cc:159-288
 *   ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...)                        src/ORBmatcher.cc:522-655
 * on flat arrays.  The vocabulary is given as parent[] / is_leaf[] / desc[] / weight[] per node in file order
 * (TemplatedVocabulary::loadFromTextFile :1351-1436: children in id order, words numbered in id order).
 * PARITY UNPINNED (see orb_oracle.h).
 */
#include "orb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HISTO_LENGTH 30

struct oracle_voc {
    int k, L, scoring, weighting, nnodes, nwords;
    int32_t *child_start, *children, *word_of;
    uint8_t *desc;
    double *weight;
};

oracle_voc_t *oracle_voc_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent,
                                const uint8_t *is_leaf, const uint8_t *desc, const double *weight) {
    oracle_voc_t *v = (oracle_voc_t *)calloc(1, sizeof(*v));
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nnodes = nnodes;
    v->child_start = (int32_t *)calloc(nnodes + 1, sizeof(int32_t));
    v->children = (int32_t *)malloc(sizeof(int32_t) * nnodes);
    v->word_of = (int32_t *)malloc(sizeof(int32_t) * nnodes);
    v->desc = (uint8_t *)malloc((size_t)32 * nnodes);
    v->weight = (double *)malloc(sizeof(double) * nnodes);
    memcpy(v->desc, desc, (size_t)32 * nnodes);
    memcpy(v->weight, weight, sizeof(double) * nnodes);
    int32_t *fill = (int32_t *)calloc(nnodes, sizeof(int32_t));
    for (int i = 1; i < nnodes; i++) v->child_start[parent[i] + 1]++;
    for (int i = 0; i < nnodes; i++) v->child_start[i + 1] += v->child_start[i];
    for (int i = 1; i < nnodes; i++) v->children[v->child_start[parent[i]] + fill[parent[i]]++] = i;
    free(fill);
    int nw = 0;
    for (int i = 0; i < nnodes; i++) v->word_of[i] = (i > 0 && is_leaf[i]) ? nw++ : -1;
    v->nwords = nw;
    return v;
}
void oracle_voc_free(oracle_voc_t *v) {
    if (!v) return;
    free(v->child_start); free(v->children); free(v->word_of); free(v->desc); free(v->weight); free(v);
}
int oracle_voc_words(const oracle_voc_t *v) { return v->nwords; }

/* :1230-1271 */
void oracle_voc_transform_one(const oracle_voc_t *v, const uint8_t *feature, int levelsup, int32_t *word_id, double *weight,
                              int32_t *nid) {
    const int nid_level = v->L - levelsup;
    if (nid_level <= 0 && nid) *nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int s = v->child_start[final_id], e = v->child_start[final_id + 1];
        final_id = v->children[s];
        double best_d = oracle_hamming(feature, v->desc + 32 * (size_t)final_id);
        for (int j = s + 1; j < e; j++) {
            const int id = v->children[j];
            const double d = oracle_hamming(feature, v->desc + 32 * (size_t)id);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid && current_level == nid_level) *nid = final_id;
    } while (v->child_start[final_id] != v->child_start[final_id + 1]);
    *word_id = v->word_of[final_id] < 0 ? 0 : v->word_of[final_id];
    *weight = v->word_of[final_id] < 0 ? 0.0 : v->weight[final_id];
}

/* :1147-1214 + BowVector / FeatureVector.  Outputs as sorted arrays (std::map order):
 * bow_word[nbow], bow_value[nbow]; fv_node[nfv], fv_start[nfv+1], fv_items[] (feature indices in insertion order).
 * Capacities: n entries each (nfv+1 for fv_start).  Returns nbow; *nfv_out = number of FeatureVector nodes. */
int oracle_voc_transform(const oracle_voc_t *v, const uint8_t *features, int n, int levelsup, int32_t *bow_word,
                         double *bow_value, int32_t *fv_node, int32_t *fv_start, int32_t *fv_items, int *nfv_out) {
    int nbow = 0, nfv = 0;
    int32_t *fnode = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    uint8_t *added = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    const int must = 1;                                   /* L1 / L2 / chi-square scoring normalise; KL etc. not restated */
    const int l2 = v->scoring == 1;
    const int tf = v->weighting == 0 || v->weighting == 1; /* TF_IDF, TF */
    for (int i = 0; i < n; i++) {
        int32_t id, nid = 0;
        double w;
        oracle_voc_transform_one(v, features + 32 * (size_t)i, levelsup, &id, &w, &nid);
        fnode[i] = nid;
        if (w > 0) {
            added[i] = 1;
            int pos = 0;                                  /* lower_bound in the sorted map */
            while (pos < nbow && bow_word[pos] < id) pos++;
            if (pos < nbow && bow_word[pos] == id) { if (tf) bow_value[pos] += w; }
            else {
                memmove(bow_word + pos + 1, bow_word + pos, sizeof(int32_t) * (nbow - pos));
                memmove(bow_value + pos + 1, bow_value + pos, sizeof(double) * (nbow - pos));
                bow_word[pos] = id; bow_value[pos] = w; nbow++;
            }
        }
    }
    (void)must;
    {   /* v.normalize(norm) */
        double norm = 0.0;
        if (!l2) for (int i = 0; i < nbow; i++) norm += fabs(bow_value[i]);
        else { for (int i = 0; i < nbow; i++) norm += bow_value[i] * bow_value[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < nbow; i++) bow_value[i] /= norm;
    }
    /* FeatureVector: nodes in increasing id, features in insertion (= index) order */
    for (int i = 0; i < n; i++) {
        if (!added[i]) continue;
        int pos = 0;
        while (pos < nfv && fv_node[pos] < fnode[i]) pos++;
        if (!(pos < nfv && fv_node[pos] == fnode[i])) {
            memmove(fv_node + pos + 1, fv_node + pos, sizeof(int32_t) * (nfv - pos));
            fv_node[pos] = fnode[i]; nfv++;
        }
    }
    int k = 0;
    for (int p = 0; p < nfv; p++) {
        fv_start[p] = k;
        for (int i = 0; i < n; i++) if (added[i] && fnode[i] == fv_node[p]) fv_items[k++] = i;
    }
    fv_start[nfv] = k;
    free(fnode); free(added);
    *nfv_out = nfv;
    return nbow;
}

/* SearchByBoW on the intersected node lists (both overloads, see orbm_search_by_bow in include/orbx.h):
 * strict_lt = 0: bestDist1 <= th_low (:233), 1: bestDist1 < th_low (:599). */
int oracle_search_by_bow(const uint8_t *qd, const float *qa, const uint8_t *qv, int nq, const uint8_t *cd, const float *ca,
                         const uint8_t *cv, int nc, const int32_t *nqs, const int32_t *qit, const int32_t *ncs,
                         const int32_t *cit, int nnodes, int th_low, int strict_lt, float nnratio, int check_ori,
                         int32_t *match_q) {
    int nmatches = 0;
    uint8_t *taken = (uint8_t *)calloc(nc > 0 ? nc : 1, 1);
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * HISTO_LENGTH * (size_t)(nq > 0 ? nq : 1));
    int32_t hn[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nq; i++) match_q[i] = -1;
    for (int j = 0; j < nnodes; j++)
        for (int q = nqs[j]; q < nqs[j + 1]; q++) {
            const int iq = qit[q];
            if (!qv[iq]) continue;
            int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
            for (int p = ncs[j]; p < ncs[j + 1]; p++) {
                const int ic = cit[p];
                if (taken[ic]) continue;
                if (cv && !cv[ic]) continue;
                const int dist = oracle_hamming(qd + 32 * (size_t)iq, cd + 32 * (size_t)ic);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = ic; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (strict_lt ? bestDist1 < th_low : bestDist1 <= th_low) {
                if ((float)bestDist1 < nnratio * (float)bestDist2) {
                    match_q[iq] = bestIdx;
                    taken[bestIdx] = 1;
                    if (check_ori) {
                        float rot = qa[iq] - ca[bestIdx];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        hist[(size_t)bin * nq + hn[bin]++] = iq;
                    }
                    nmatches++;
                }
            }
        }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int t = 0; t < hn[i]; t++) { match_q[hist[(size_t)i * nq + t]] = -1; nmatches--; }
    }
    free(taken); free(hist);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:140-157) */
static int check_dist_epipolar_line(const oracle_kp_t *kp1, const oracle_kp_t *kp2, const float *F12, const float *sigma2) {
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2[kp2->octave];
}

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-825) on the intersected node lists.
 * qf / cf: bit 0 usable (no map point, stereo rule of bOnlyStereo applied by the caller), bit 1 mvuRight >= 0. */
int oracle_search_for_triangulation(const oracle_kp_t *k1, const uint8_t *qd, const uint8_t *qf, int nq, const oracle_kp_t *k2,
                                    const uint8_t *cd, const uint8_t *cf, int nc, const int32_t *nqs, const int32_t *qit,
                                    const int32_t *ncs, const int32_t *cit, int nnodes, const float *F12, float ex, float ey,
                                    const float *sf, const float *sigma2, int th_low, int check_ori, int32_t *match_q) {
    int nmatches = 0;
    uint8_t *matched2 = (uint8_t *)calloc(nc > 0 ? nc : 1, 1);
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * HISTO_LENGTH * (size_t)(nq > 0 ? nq : 1));
    int32_t hn[HISTO_LENGTH] = {0};
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nq; i++) match_q[i] = -1;
    for (int j = 0; j < nnodes; j++)
        for (int q = nqs[j]; q < nqs[j + 1]; q++) {
            const int idx1 = qit[q];
            if (!(qf[idx1] & 1)) continue;
            const int bStereo1 = (qf[idx1] & 2) != 0;
            int bestDist = th_low, bestIdx2 = -1;
            for (int p = ncs[j]; p < ncs[j + 1]; p++) {
                const int idx2 = cit[p];
                if (matched2[idx2] || !(cf[idx2] & 1)) continue;
                const int bStereo2 = (cf[idx2] & 2) != 0;
                const int dist = oracle_hamming(qd + 32 * (size_t)idx1, cd + 32 * (size_t)idx2);
                if (dist > th_low || dist > bestDist) continue;
                if (!bStereo1 && !bStereo2) {
                    const float distex = ex - k2[idx2].x, distey = ey - k2[idx2].y;
                    if (distex * distex + distey * distey < 100 * sf[k2[idx2].octave]) continue;
                }
                if (check_dist_epipolar_line(&k1[idx1], &k2[idx2], F12, sigma2)) { bestIdx2 = idx2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                match_q[idx1] = bestIdx2;
                matched2[bestIdx2] = 1;
                nmatches++;
                if (check_ori) {
                    float rot = k1[idx1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    hist[(size_t)bin * nq + hn[bin]++] = idx1;
                }
            }
        }
    if (check_ori) {
        int ind1, ind2, ind3;
        oracle_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int t = 0; t < hn[i]; t++) { match_q[hist[(size_t)i * nq + t]] = -1; nmatches--; }
    }
    free(matched2); free(hist);
    return nmatches;
}
