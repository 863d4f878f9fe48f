// orbx_bow.hip — SURVEY §8(f) rank 3: the DBoW2 vocabulary descent (TemplatedVocabulary::transform,
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1230-1271 with FORB::distance, FORB.cpp:81-101) and the
// BoW-guided matchers ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288, 522-655) on gfx950.
//   k_voc_descend   16 lanes per feature: lane c scores child c of the current node (4 x popcll), a 4-step
//                   lane-group minimum on (distance, child position) picks the first minimum, until a leaf
//   k_bow_match     one wave per vocabulary node shared by the two feature vectors: the node's queries in
//                   order, candidates across the lanes, top-2 by wave minimum; the "already matched" flags of
//                   the candidates live in one 64-bit register per lane (candidate p -> lane p%64, bit p/64)
//   k_bow_orient    rotation histogram + three maxima, one workgroup
#include "orbx_match_dev.h"
#include <math.h>
#include <stdlib.h>
#include <string>
#include <vector>

struct orbv_vocabulary {
    int k, L, scoring, weighting, nnodes, nwords, device;
    std::vector<int32_t> word_of;     // node -> word id (-1: not a word)
    std::vector<double> weight;       // node weight
    std::vector<int32_t> child_start; // CSR of children (host copy, for orbv_info / validation)
    uint8_t *d_desc = nullptr;        // [nnodes][32]
    int32_t *d_child_start = nullptr; // [nnodes + 1]
    int32_t *d_children = nullptr;    // [nnodes - 1]
    int32_t *d_level = nullptr;       // [nnodes] depth of the node (root 0)
};

#define VOC_GROUP 16
__global__ __launch_bounds__(256) void k_voc_descend(const uint8_t *__restrict__ ndesc, const int32_t *__restrict__ cstart,
                                                     const int32_t *__restrict__ children, const uint8_t *__restrict__ feat,
                                                     int n, int nid_level, int32_t *__restrict__ leaf,
                                                     int32_t *__restrict__ nid) {
    const int f = (blockIdx.x * 256 + threadIdx.x) / VOC_GROUP, c = threadIdx.x & (VOC_GROUP - 1);
    const bool live = f < n;
    const Desc256 d = load_desc(feat + (size_t)(live ? f : 0) * 32);
    int node = 0, level = 0, at = 0;   // at: node on the path at nid_level (0 = root when nid_level <= 0)
    // every lane group walks its own path; groups of a wave leave the loop together (the shuffles need whole waves)
    bool done = !live;
    while (__ballot(!done)) {
        const int s = cstart[node], e = cstart[node + 1];
        u64 best = ~0ull;
        if (!done)
            for (int j = s + c; j < e; j += VOC_GROUP) {
                const int ch = children[j];
                const u64 key = ((u64)ham(d, load_desc(ndesc + (size_t)ch * 32)) << 40) | ((u64)(j - s) << 20) | (u64)0;
                best = key < best ? key : best;
            }
#pragma unroll
        for (int o = 1; o < VOC_GROUP; o <<= 1) {
            const u64 t = shfl_xor_u64(best, o);
            best = t < best ? t : best;
        }
        if (!done) {
            node = children[s + (int)((best >> 20) & 0xFFFFF)];
            level++;
            if (level == nid_level) at = node;
            done = cstart[node] == cstart[node + 1];   // Node::isLeaf(): no children
        }
    }
    if (live && c == 0) { leaf[f] = node; nid[f] = at; }
}

extern "C" void orbv_destroy(orbv_vocabulary_t *v) {
    if (!v) return;
    if (v->d_desc || v->d_child_start || v->d_children || v->d_level) {
        hipSetDevice(v->device);
        if (v->d_desc) hipFree(v->d_desc);
        if (v->d_child_start) hipFree(v->d_child_start);
        if (v->d_children) hipFree(v->d_children);
        if (v->d_level) hipFree(v->d_level);
    }
    delete v;
}

extern "C" int orbv_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent, const uint8_t *is_leaf,
                           const uint8_t *desc, const double *weight, int device, orbv_vocabulary_t **out) {
    if (!out) return ORBX_ERR_ARG;
    *out = nullptr;
    if (nnodes < 2 || !parent || !is_leaf || !desc || !weight || k < 1 || L < 1) { orbx_set_error("orbv_create: bad arguments"); return ORBX_ERR_ARG; }
    for (int i = 1; i < nnodes; i++)
        if (parent[i] < 0 || parent[i] >= i) { orbx_set_error("orbv_create: node %d has parent %d (must precede it)", i, parent[i]); return ORBX_ERR_ARG; }
    orbv_vocabulary *v = new orbv_vocabulary();
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nnodes = nnodes; v->device = device;
    v->word_of.assign(nnodes, -1);
    v->weight.assign(weight, weight + nnodes);
    int nw = 0;
    for (int i = 1; i < nnodes; i++) if (is_leaf[i]) v->word_of[i] = nw++;   // m_words order (:1421-1428)
    v->nwords = nw;
    std::vector<int32_t> cnt(nnodes + 1, 0), children(nnodes - 1), fill(nnodes, 0);
    for (int i = 1; i < nnodes; i++) cnt[parent[i] + 1]++;
    for (int i = 0; i < nnodes; i++) cnt[i + 1] += cnt[i];
    for (int i = 1; i < nnodes; i++) children[cnt[parent[i]] + fill[parent[i]]++] = i;   // children in id order
    v->child_start = cnt;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(&v->d_desc, (size_t)nnodes * 32);
    if (e == hipSuccess) e = hipMalloc(&v->d_child_start, sizeof(int32_t) * (nnodes + 1));
    if (e == hipSuccess) e = hipMalloc(&v->d_children, sizeof(int32_t) * (nnodes - 1));
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, desc, (size_t)nnodes * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_start, cnt.data(), sizeof(int32_t) * (nnodes + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_children, children.data(), sizeof(int32_t) * (nnodes - 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        orbx_set_error("orbv_create: %s", hipGetErrorString(e));
        orbv_destroy(v);
        return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu || e == hipErrorInsufficientDriver) ? ORBX_ERR_NO_DEVICE : ORBX_ERR_HIP;
    }
    *out = v;
    return ORBX_OK;
}

extern "C" int orbv_load_text(const char *path, int device, orbv_vocabulary_t **out) {
    if (!path || !out) return ORBX_ERR_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "r");
    if (!f) { orbx_set_error("orbv_load_text: cannot open %s", path); return ORBX_ERR_ARG; }
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        fclose(f);
        orbx_set_error("orbv_load_text: %s is not a vocabulary text file", path);   // :1372-1376
        return ORBX_ERR_ARG;
    }
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    for (;;) {
        int pid, isleaf;
        if (fscanf(f, "%d %d", &pid, &isleaf) != 2) break;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32; i++) { int b; if (fscanf(f, "%d", &b) != 1) { ok = false; break; } d[i] = (uint8_t)b; }
        double w;
        if (!ok || fscanf(f, "%lf", &w) != 1) { fclose(f); orbx_set_error("orbv_load_text: truncated node %zu", parent.size()); return ORBX_ERR_ARG; }
        parent.push_back(pid); leaf.push_back(isleaf > 0 ? 1 : 0); weight.push_back(w);
        desc.insert(desc.end(), d, d + 32);
    }
    fclose(f);
    return orbv_create(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), device, out);
}

extern "C" int orbv_info(const orbv_vocabulary_t *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords) {
    if (!v) return ORBX_ERR_ARG;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return ORBX_OK;
}

struct BowScratch { uint8_t *d = nullptr; size_t cap = 0; int device = -1; };
static thread_local BowScratch g_bs;
static int scratch(int device, size_t need, uint8_t **p) {
    ORBX_HIP(hipSetDevice(device));
    if (g_bs.device != device || g_bs.cap < need) {
        if (g_bs.d) { hipSetDevice(g_bs.device); hipFree(g_bs.d); hipSetDevice(device); g_bs.d = nullptr; g_bs.cap = 0; }
        const size_t cap = need * 2 > ((size_t)1 << 20) ? need * 2 : ((size_t)1 << 20);
        ORBX_HIP(hipMalloc(&g_bs.d, cap));
        g_bs.cap = cap; g_bs.device = device;
    }
    *p = g_bs.d;
    return ORBX_OK;
}
void orbx_internal_release_bow_scratch() {
    if (g_bs.d) { hipSetDevice(g_bs.device); hipFree(g_bs.d); }
    g_bs.d = nullptr; g_bs.cap = 0; g_bs.device = -1;
}
#define ALN(x) (((x) + 255) & ~(size_t)255)

extern "C" int orbv_transform(const orbv_vocabulary_t *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                              int32_t *node_id, double *weight) {
    if (!v || n < 0 || (n > 0 && (!desc || !word_id))) { orbx_set_error("orbv_transform: bad arguments"); return ORBX_ERR_ARG; }
    if (n == 0) return ORBX_OK;
    uint8_t *base;
    int rc = scratch(v->device, ALN((size_t)n * 32) + 2 * ALN((size_t)n * 4), &base);
    if (rc) return rc;
    uint8_t *dfeat = base;
    int32_t *dleaf = (int32_t *)(base + ALN((size_t)n * 32)), *dnid = (int32_t *)((uint8_t *)dleaf + ALN((size_t)n * 4));
    ORBX_HIP(hipMemcpy(dfeat, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    (void)hipGetLastError();
    const int groups_per_block = 256 / VOC_GROUP;
    hipLaunchKernelGGL(k_voc_descend, dim3((n + groups_per_block - 1) / groups_per_block), dim3(256), 0, 0, v->d_desc,
                       v->d_child_start, v->d_children, dfeat, n, v->L - levelsup, dleaf, dnid);
    ORBX_HIP(hipGetLastError());
    std::vector<int32_t> leaf(n);
    ORBX_HIP(hipMemcpy(leaf.data(), dleaf, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (node_id) ORBX_HIP(hipMemcpy(node_id, dnid, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        // a childless node that the file did not flag as a word keeps Node()'s defaults: word 0, weight 0 (:313-318)
        const int w = v->word_of[leaf[i]];
        word_id[i] = w < 0 ? 0 : w;
        if (weight) weight[i] = w < 0 ? 0.0 : v->weight[leaf[i]];
    }
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_match(const uint8_t *__restrict__ qd, const uint8_t *__restrict__ qv,
                                                   const uint8_t *__restrict__ cd, const uint8_t *__restrict__ cv,
                                                   const int32_t *__restrict__ nqs, const int32_t *__restrict__ qit,
                                                   const int32_t *__restrict__ ncs, const int32_t *__restrict__ cit,
                                                   int nnodes, int max_dist, float nnratio, int32_t *__restrict__ match_q,
                                                   int32_t *__restrict__ overflow) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= nnodes) return;
    const int q0 = nqs[j], q1 = nqs[j + 1], c0 = ncs[j], nc = ncs[j + 1] - c0;
    if (nc > 64 * 64) { if (lane == 0) *overflow = 1; return; }   // one taken-bit per candidate in a 64-bit lane register
    u64 taken = 0;
    for (int q = q0; q < q1; q++) {
        const int iq = qit[q];
        if (!qv[iq]) continue;                                   // !pMP || pMP->isBad()   (:195-199)
        const Desc256 dq = load_desc(qd + (size_t)iq * 32);
        u64 k1 = ~0ull, k2 = ~0ull;                              // two smallest (dist << 32 | position) of this lane
        for (int p = lane; p < nc; p += 64) {
            if ((taken >> (p >> 6)) & 1ull) continue;            // vpMapPointMatches[realIdxF] / vbMatched2   (:213, :575)
            const int ic = cit[c0 + p];
            if (cv && !cv[ic]) continue;                         // !pMP2 || pMP2->isBad()   (:573-579)
            const u64 key = ((u64)ham(dq, load_desc(cd + (size_t)ic * 32)) << 32) | (unsigned)p;
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
        const u64 b1 = wave_min_u64(k1);
        if (b1 == ~0ull) continue;
        const u64 b2 = wave_min_u64(k1 == b1 ? k2 : k1);         // best of the others: bestDist2 (256 if none)
        const int bestDist1 = (int)(b1 >> 32), bestDist2 = b2 == ~0ull ? 256 : (int)(b2 >> 32);
        if (bestDist1 <= max_dist && (float)bestDist1 < nnratio * (float)bestDist2) {
            const int p = (int)(b1 & 0xFFFFFFFFu);
            if (lane == (p & 63)) taken |= 1ull << (p >> 6);
            if (lane == 0) match_q[iq] = cit[c0 + p];
        }
    }
}

__global__ __launch_bounds__(256) void k_bow_orient(const float *__restrict__ qa, const float *__restrict__ ca, int nq,
                                                    int32_t *__restrict__ match_q, int check_ori,
                                                    int32_t *__restrict__ nmatches) {
    __shared__ int hn[HISTO_LENGTH], ind[3], cnt;
    const int tid = threadIdx.x;
    if (tid < HISTO_LENGTH) hn[tid] = 0;
    if (tid == 0) cnt = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = tid; i < nq; i += 256) {
        const int m = match_q[i];
        if (m < 0) continue;
        atomicAdd(&cnt, 1);
        if (check_ori) {
            float rot = qa[i] - ca[m];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            atomicAdd(&hn[bin], 1);
        }
    }
    __syncthreads();
    if (check_ori) {
        if (tid == 0) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
        __syncthreads();
        for (int i = tid; i < nq; i += 256) {
            const int m = match_q[i];
            if (m < 0) continue;
            float rot = qa[i] - ca[m];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin != ind[0] && bin != ind[1] && bin != ind[2]) { match_q[i] = -1; atomicSub(&cnt, 1); }
        }
        __syncthreads();
    }
    if (tid == 0) *nmatches = cnt;
}

extern "C" int orbm_search_by_bow(const uint8_t *q_desc, const float *q_angle, const uint8_t *q_valid, int nq,
                                  const uint8_t *c_desc, const float *c_angle, const uint8_t *c_valid, int nc,
                                  const int32_t *node_qstart, const int32_t *q_items, const int32_t *node_cstart,
                                  const int32_t *c_items, int nnodes, int max_dist, float nnratio, int check_orientation,
                                  int32_t *match_q, int *nmatches, int device) {
    if (nq < 0 || nc < 0 || nnodes < 0 || !nmatches || (nq > 0 && (!q_desc || !q_angle || !q_valid || !match_q)) ||
        (nc > 0 && (!c_desc || !c_angle)) || (nnodes > 0 && (!node_qstart || !node_cstart))) {
        orbx_set_error("orbm_search_by_bow: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    for (int i = 0; i < nq; i++) match_q[i] = -1;
    if (nq == 0 || nc == 0 || nnodes == 0) return ORBX_OK;
    const int tq = node_qstart[nnodes], tc = node_cstart[nnodes];
    if (node_qstart[0] != 0 || node_cstart[0] != 0 || tq < 0 || tc < 0 || (tq > 0 && !q_items) || (tc > 0 && !c_items)) { orbx_set_error("orbm_search_by_bow: bad node lists"); return ORBX_ERR_ARG; }
    for (int j = 0; j < nnodes; j++)
        if (node_qstart[j + 1] < node_qstart[j] || node_cstart[j + 1] < node_cstart[j]) { orbx_set_error("orbm_search_by_bow: node lists not monotonic at %d", j); return ORBX_ERR_ARG; }
    for (int i = 0; i < tq; i++) if (q_items[i] < 0 || q_items[i] >= nq) { orbx_set_error("q_items[%d] out of range", i); return ORBX_ERR_ARG; }
    for (int i = 0; i < tc; i++) if (c_items[i] < 0 || c_items[i] >= nc) { orbx_set_error("c_items[%d] out of range", i); return ORBX_ERR_ARG; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += ALN(bytes); return o; };
    const size_t o_qd = take((size_t)nq * 32), o_qa = take((size_t)nq * 4), o_qv = take(nq), o_cd = take((size_t)nc * 32),
                 o_ca = take((size_t)nc * 4), o_cv = take(nc), o_nqs = take((size_t)(nnodes + 1) * 4), o_ncs = take((size_t)(nnodes + 1) * 4),
                 o_qi = take((size_t)(tq > 0 ? tq : 1) * 4), o_ci = take((size_t)(tc > 0 ? tc : 1) * 4), o_m = take((size_t)nq * 4), o_out = take(16);
    uint8_t *base;
    int rc = scratch(device, off, &base);
    if (rc) return rc;
#define H2DB(o, src, bytes) ORBX_HIP(hipMemcpy(base + (o), (src), (bytes), hipMemcpyHostToDevice))
    H2DB(o_qd, q_desc, (size_t)nq * 32); H2DB(o_qa, q_angle, (size_t)nq * 4); H2DB(o_qv, q_valid, nq);
    H2DB(o_cd, c_desc, (size_t)nc * 32); H2DB(o_ca, c_angle, (size_t)nc * 4);
    if (c_valid) H2DB(o_cv, c_valid, nc);
    H2DB(o_nqs, node_qstart, (size_t)(nnodes + 1) * 4); H2DB(o_ncs, node_cstart, (size_t)(nnodes + 1) * 4);
    if (tq > 0) H2DB(o_qi, q_items, (size_t)tq * 4);
    if (tc > 0) H2DB(o_ci, c_items, (size_t)tc * 4);
    ORBX_HIP(hipMemset(base + o_m, 0xFF, (size_t)nq * 4));
    ORBX_HIP(hipMemset(base + o_out, 0, 16));
    (void)hipGetLastError();
    int32_t *dout = (int32_t *)(base + o_out);
    hipLaunchKernelGGL(k_bow_match, dim3((nnodes + 3) / 4), dim3(256), 0, 0, base + o_qd, base + o_qv, base + o_cd,
                       c_valid ? base + o_cv : (const uint8_t *)nullptr, (const int32_t *)(base + o_nqs), (const int32_t *)(base + o_qi),
                       (const int32_t *)(base + o_ncs), (const int32_t *)(base + o_ci), nnodes, max_dist, nnratio,
                       (int32_t *)(base + o_m), dout + 1);
    hipLaunchKernelGGL(k_bow_orient, dim3(1), dim3(256), 0, 0, (const float *)(base + o_qa), (const float *)(base + o_ca), nq,
                       (int32_t *)(base + o_m), check_orientation, dout);
    ORBX_HIP(hipGetLastError());
    int32_t out[2] = {0, 0};
    ORBX_HIP(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost));
    if (out[1]) { orbx_set_error("orbm_search_by_bow: a vocabulary node holds more than 4096 candidate features"); return ORBX_ERR_UNSUPPORTED; }
    ORBX_HIP(hipMemcpy(match_q, base + o_m, (size_t)nq * 4, hipMemcpyDeviceToHost));
    *nmatches = out[0];
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-825): same node structure as SearchByBoW, but the best
// candidate is the LAST one (in list order) of minimum distance among those that pass the two geometric tests,
// which depend on the pair only: key = dist << 32 | ~position, one wave minimum.
struct TriGeom { float F[9]; float ex, ey; float sf[ORBX_MAX_LEVELS], sig2[ORBX_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void k_tri_match(const orbx_keypoint_t *__restrict__ k1, const uint8_t *__restrict__ qd,
                                                   const uint8_t *__restrict__ qf, const orbx_keypoint_t *__restrict__ k2,
                                                   const uint8_t *__restrict__ cd, const uint8_t *__restrict__ cf,
                                                   const int32_t *__restrict__ nqs, const int32_t *__restrict__ qit,
                                                   const int32_t *__restrict__ ncs, const int32_t *__restrict__ cit,
                                                   int nnodes, TriGeom G, int max_dist, int32_t *__restrict__ match_q,
                                                   int32_t *__restrict__ overflow) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= nnodes) return;
    const int q0 = nqs[j], q1 = nqs[j + 1], c0 = ncs[j], nc = ncs[j + 1] - c0;
    if (nc > 64 * 64) { if (lane == 0) *overflow = 1; return; }
    u64 taken = 0;
    for (int q = q0; q < q1; q++) {
        const int iq = qit[q];
        const int fq = qf[iq];
        if (!(fq & 1)) continue;
        const bool stereo1 = (fq & 2) != 0;
        const orbx_keypoint_t kp1 = k1[iq];
        const Desc256 dq = load_desc(qd + (size_t)iq * 32);
        // epipolar line in the second image l = x1'F12 = [a b c]   (:142-145)
        const float a = kp1.x * G.F[0] + kp1.y * G.F[3] + G.F[6];
        const float b = kp1.x * G.F[1] + kp1.y * G.F[4] + G.F[7];
        const float c = kp1.x * G.F[2] + kp1.y * G.F[5] + G.F[8];
        const float den = a * a + b * b;
        u64 best = ~0ull;
        for (int p = lane; p < nc; p += 64) {
            if ((taken >> (p >> 6)) & 1ull) continue;                    // vbMatched2   (:724)
            const int ic = cit[c0 + p];
            const int fc = cf[ic];
            if (!(fc & 1)) continue;
            const int dist = ham(dq, load_desc(cd + (size_t)ic * 32));
            if (dist > max_dist) continue;                               // :737 (dist > bestDist is implied by the minimum)
            const orbx_keypoint_t kp2 = k2[ic];
            const int oct = min(max(kp2.octave, 0), ORBX_MAX_LEVELS - 1);
            if (!stereo1 && !(fc & 2)) {                                 // :742-748
                const float distex = G.ex - kp2.x, distey = G.ey - kp2.y;
                if (distex * distex + distey * distey < 100 * G.sf[oct]) continue;
            }
            const float num = a * kp2.x + b * kp2.y + c;                 // :147-156
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (!((double)dsqr < 3.84 * (double)G.sig2[oct])) continue;
            const u64 key = ((u64)dist << 32) | (u64)(0xFFFFFFFFu - (unsigned)p);
            best = key < best ? key : best;
        }
        best = wave_min_u64(best);
        if (best == ~0ull) continue;
        const int p = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFu));
        if (lane == (p & 63)) taken |= 1ull << (p >> 6);
        if (lane == 0) match_q[iq] = cit[c0 + p];
    }
}
__global__ __launch_bounds__(256) void k_kp_angles(const orbx_keypoint_t *__restrict__ k, int n, float *__restrict__ a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = k[i].angle;
}

extern "C" int orbm_search_for_triangulation(const orbx_keypoint_t *kp1, const uint8_t *q_desc, const uint8_t *q_flags, int nq,
                                             const orbx_keypoint_t *kp2, const uint8_t *c_desc, const uint8_t *c_flags, int nc,
                                             const int32_t *node_qstart, const int32_t *q_items, const int32_t *node_cstart,
                                             const int32_t *c_items, int nnodes, const float *F12_9, float ex, float ey,
                                             const float *scale_factors, const float *level_sigma2, int nlevels, int max_dist,
                                             int check_orientation, int32_t *match_q, int *nmatches, int device) {
    if (nq < 0 || nc < 0 || nnodes < 0 || !nmatches || !F12_9 || !scale_factors || !level_sigma2 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS ||
        (nq > 0 && (!kp1 || !q_desc || !q_flags || !match_q)) || (nc > 0 && (!kp2 || !c_desc || !c_flags)) ||
        (nnodes > 0 && (!node_qstart || !node_cstart))) {
        orbx_set_error("orbm_search_for_triangulation: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    for (int i = 0; i < nq; i++) match_q[i] = -1;
    if (nq == 0 || nc == 0 || nnodes == 0) return ORBX_OK;
    const int tq = node_qstart[nnodes], tc = node_cstart[nnodes];
    if (node_qstart[0] != 0 || node_cstart[0] != 0 || tq < 0 || tc < 0 || (tq > 0 && !q_items) || (tc > 0 && !c_items)) { orbx_set_error("orbm_search_for_triangulation: bad node lists"); return ORBX_ERR_ARG; }
    for (int j = 0; j < nnodes; j++)
        if (node_qstart[j + 1] < node_qstart[j] || node_cstart[j + 1] < node_cstart[j]) { orbx_set_error("orbm_search_for_triangulation: node lists not monotonic at %d", j); return ORBX_ERR_ARG; }
    for (int i = 0; i < tq; i++) if (q_items[i] < 0 || q_items[i] >= nq) { orbx_set_error("q_items[%d] out of range", i); return ORBX_ERR_ARG; }
    for (int i = 0; i < tc; i++) if (c_items[i] < 0 || c_items[i] >= nc) { orbx_set_error("c_items[%d] out of range", i); return ORBX_ERR_ARG; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += ALN(bytes); return o; };
    const size_t o_k1 = take((size_t)nq * sizeof(orbx_keypoint_t)), o_qd = take((size_t)nq * 32), o_qf = take(nq), o_qa = take((size_t)nq * 4),
                 o_k2 = take((size_t)nc * sizeof(orbx_keypoint_t)), o_cd = take((size_t)nc * 32), o_cf = take(nc), o_ca = take((size_t)nc * 4),
                 o_nqs = take((size_t)(nnodes + 1) * 4), o_ncs = take((size_t)(nnodes + 1) * 4), o_qi = take((size_t)(tq > 0 ? tq : 1) * 4),
                 o_ci = take((size_t)(tc > 0 ? tc : 1) * 4), o_m = take((size_t)nq * 4), o_out = take(16);
    uint8_t *base;
    int rc = scratch(device, off, &base);
    if (rc) return rc;
    H2DB(o_k1, kp1, (size_t)nq * sizeof(orbx_keypoint_t)); H2DB(o_qd, q_desc, (size_t)nq * 32); H2DB(o_qf, q_flags, nq);
    H2DB(o_k2, kp2, (size_t)nc * sizeof(orbx_keypoint_t)); H2DB(o_cd, c_desc, (size_t)nc * 32); H2DB(o_cf, c_flags, nc);
    H2DB(o_nqs, node_qstart, (size_t)(nnodes + 1) * 4); H2DB(o_ncs, node_cstart, (size_t)(nnodes + 1) * 4);
    if (tq > 0) H2DB(o_qi, q_items, (size_t)tq * 4);
    if (tc > 0) H2DB(o_ci, c_items, (size_t)tc * 4);
    ORBX_HIP(hipMemset(base + o_m, 0xFF, (size_t)nq * 4));
    ORBX_HIP(hipMemset(base + o_out, 0, 16));
    TriGeom G;
    memcpy(G.F, F12_9, sizeof(G.F));
    G.ex = ex; G.ey = ey;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) { G.sf[l] = scale_factors[l < nlevels ? l : nlevels - 1]; G.sig2[l] = level_sigma2[l < nlevels ? l : nlevels - 1]; }
    (void)hipGetLastError();
    int32_t *dout = (int32_t *)(base + o_out);
    hipLaunchKernelGGL(k_kp_angles, dim3((nq + 255) / 256), dim3(256), 0, 0, (const orbx_keypoint_t *)(base + o_k1), nq, (float *)(base + o_qa));
    hipLaunchKernelGGL(k_kp_angles, dim3((nc + 255) / 256), dim3(256), 0, 0, (const orbx_keypoint_t *)(base + o_k2), nc, (float *)(base + o_ca));
    hipLaunchKernelGGL(k_tri_match, dim3((nnodes + 3) / 4), dim3(256), 0, 0, (const orbx_keypoint_t *)(base + o_k1), base + o_qd, base + o_qf,
                       (const orbx_keypoint_t *)(base + o_k2), base + o_cd, base + o_cf, (const int32_t *)(base + o_nqs),
                       (const int32_t *)(base + o_qi), (const int32_t *)(base + o_ncs), (const int32_t *)(base + o_ci), nnodes, G, max_dist,
                       (int32_t *)(base + o_m), dout + 1);
    hipLaunchKernelGGL(k_bow_orient, dim3(1), dim3(256), 0, 0, (const float *)(base + o_qa), (const float *)(base + o_ca), nq,
                       (int32_t *)(base + o_m), check_orientation, dout);
    ORBX_HIP(hipGetLastError());
    int32_t out[2] = {0, 0};
    ORBX_HIP(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost));
    if (out[1]) { orbx_set_error("orbm_search_for_triangulation: a vocabulary node holds more than 4096 candidate features"); return ORBX_ERR_UNSUPPORTED; }
    ORBX_HIP(hipMemcpy(match_q, base + o_m, (size_t)nq * 4, hipMemcpyDeviceToHost));
    *nmatches = out[0];
    return ORBX_OK;
}
