// orbx_extract.hip — MI355X (gfx950) ORB front-end: hand-written HIP kernels + C ABI.
//
// Replaces ORBextractor::operator() of kimwin2/ORB_SLAM2v2-1 (reference:
// src/ORBextractor.cc:1043-1105) for batches of equally sized frames:
//   K1 k_pyramid_fused               ComputePyramid, one launch      (:1107-1132)
//   K2 k_fast_cells                  per-cell cv::FAST + fallback   (:789-829)
//   K3 k_octree                      DistributeOctTree              (:539-763)
//   K4 k_describe                    IC_Angle + GaussianBlur + rBRIEF (:77-147, :1085-1090)
// Integer / bitwise work: no MFMA.  Build with -ffp-contract=off (the float expressions of
// the reference are evaluated operation by operation).
#include "orbx_extract_dev.h"
#ifdef ORBX_DEVELOPER
#include "orbx_dev.h"
#endif
#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <algorithm>

// ------------------------------------------------------------------------------------
// error string
static thread_local char g_err[512] = "";
void orbx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *orbx_last_error(void) { return g_err; }
extern "C" const char *orbx_version(void) { return "orbx 0.1 (gfx950)"; }
extern "C" int orbx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The matcher entry points keep grow-only device scratch + pinned mirrors + one stream per HOST THREAD (so that they
// are re-entrant without locks).  A thread that is about to exit - or wants its memory back - calls this.
extern "C" int orbx_thread_release_scratch(void) {
    orbx_internal_release_match_scratch();
    orbx_internal_release_arena();
    orbx_internal_release_bow_scratch();
    return ORBX_OK;
}

// Per-handle options (include/orbx.h): which of several kernels / launch arrangements with identical results the handle uses.
// There is no process-global switch.  Keys 0, 1, 7 stop a kernel after phase n (ablation timing; outputs incomplete) and exist only
// in a developer build (-DORBX_DEVELOPER).
extern "C" int orbx_set_option(orbx_extractor_t *h, int key, int value) {
    static const signed char maxv[ORBX_NUM_OPTIONS] = {/*0*/ -1, -1, -2, 64, 3, 4, 3, -1, ORBX_MAX_CHUNKS, 1, 3, 2, 1, 2, 127, ORBX_MAX_LEVELS,
                                                       /*16*/ 2, -2, 1, ORBX_MAX_LEVELS, 2, 40, 2, 1, 2, 3, 1, 1, -2, -2, -2, -2};
    if (!h || key < 0 || key >= ORBX_NUM_OPTIONS || maxv[key] == -2) { orbx_set_error("orbx_set_option: unknown key %d", key); return ORBX_ERR_ARG; }
#ifdef ORBX_DEVELOPER
    if (maxv[key] == -1) { if (value < 0) return ORBX_ERR_ARG; h->opt[key] = value; return ORBX_OK; }
#else
    if (maxv[key] == -1) { orbx_set_error("orbx_set_option: key %d needs a library built with -DORBX_DEVELOPER", key); return ORBX_ERR_ARG; }
#endif
    if (value < 0 || (key != ORBX_OPT_BLUR_THRESHOLD && value > maxv[key])) { orbx_set_error("orbx_set_option: key %d does not take %d", key, value); return ORBX_ERR_ARG; }
    h->opt[key] = value;
    return ORBX_OK;
}
extern "C" int orbx_get_option(const orbx_extractor_t *h, int key, int *value) {
    if (!h || !value || key < 0 || key >= ORBX_NUM_OPTIONS) return ORBX_ERR_ARG;
    *value = h->opt[key];
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
// host side
static int cv_round(double v) { return (int)lrint(v); }
static int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static int cv_ceil(double v) { int i = (int)v; return i + (v > i); }
static short sat_short_round(float v) {
    int i = cv_round(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

extern "C" int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                           int device, orbx_extractor_t **out) {
    return orbx_create_flavoured(nfeatures, scale_factor, nlevels, ini_th, min_th, device, nullptr, out);
}
extern "C" int orbx_get_flavour(const orbx_extractor_t *h, orbx_flavour_t *out) {
    if (!h || !out) return ORBX_ERR_ARG;
    *out = h->flavour;
    return ORBX_OK;
}
extern "C" int orbx_create_flavoured(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                                     int device, const orbx_flavour_t *flavour, orbx_extractor_t **out) {
    if (!out) { orbx_set_error("orbx_create: out is NULL"); return ORBX_ERR_ARG; }
    *out = nullptr;
    if (flavour) {
        bool ok = flavour->gauss_rounding == ORBX_GAUSS_ROUND_HALF_UP || flavour->gauss_rounding == ORBX_GAUSS_ROUND_SSE2 ||
                  flavour->gauss_rounding == ORBX_GAUSS_FIXED_TAPS;
        for (int i = 0; i < 3; i++) ok = ok && flavour->reserved[i] == 0;
        const int32_t *k = flavour->gauss_taps;
        if (flavour->gauss_rounding == ORBX_GAUSS_FIXED_TAPS) {   // Q8 taps, centre first; a row sum (<= 255 * their sum) must fit 16 bits
            for (int i = 0; i < 4; i++) ok = ok && k[i] >= 0 && k[i] <= 255;
            ok = ok && k[0] >= 1 && k[0] + 2 * (k[1] + k[2] + k[3]) <= 257;
        } else
            for (int i = 0; i < 4; i++) ok = ok && k[i] == 0;
        if (!ok) {
            orbx_set_error("orbx_create_flavoured: unknown flavour (gauss_rounding=%d, gauss_taps=%d %d %d %d)", flavour->gauss_rounding, k[0], k[1], k[2], k[3]);
            return ORBX_ERR_ARG;
        }
    }
    if (nfeatures < 1 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !(scale_factor > 1.0f) || ini_th < 0 ||
        min_th < 0 || ini_th > 255 || min_th > 255) {
        orbx_set_error("orbx_create: bad arguments (nfeatures=%d scale=%f nlevels=%d ini=%d min=%d)", nfeatures,
                       scale_factor, nlevels, ini_th, min_th);
        return ORBX_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0 || device < 0 || device >= ndev) {
        orbx_set_error("orbx_create: no usable HIP device (count=%d, requested %d): %s", ndev, device,
                       e == hipSuccess ? "ok" : hipGetErrorString(e));
        return ORBX_ERR_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_extractor *h = new orbx_extractor();
    memset(h, 0, sizeof(*h));
    h->nfeatures = nfeatures; h->nlevels = nlevels; h->ini_th = ini_th; h->min_th = min_th;
    h->device = device; h->scale_factor = scale_factor;
    if (flavour) h->flavour = *flavour;
    // scale tables (:415-431)
    h->sf[0] = 1.0f; h->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        h->sf[i] = (float)(h->sf[i - 1] * h->scale_factor);
        h->sig2[i] = h->sf[i] * h->sf[i];
    }
    for (int i = 0; i < nlevels; i++) { h->isf[i] = 1.0f / h->sf[i]; h->isig2[i] = 1.0f / h->sig2[i]; }
    // features per level (:435-446)
    float factor = (float)(1.0f / h->scale_factor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        h->nfeat[level] = cv_round(nDesired);
        sum += h->nfeat[level];
        nDesired *= factor;
    }
    h->nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax (:454-469)
    {
        int v, v0, vmax = cv_floor(ORBX_HALF_PATCH * sqrtf(2.f) / 2 + 1);
        int vmin = cv_ceil(ORBX_HALF_PATCH * sqrtf(2.f) / 2);
        const double hp2 = ORBX_HALF_PATCH * ORBX_HALF_PATCH;
        for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(sqrt(hp2 - v * v));
        for (v = ORBX_HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
            h->umax[v] = v0;
            ++v0;
        }
    }
    h->max_kp = 0;
    ORBX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // host-mapped word the quad-tree kernels store a call's sequence number into when they flag a corner-sparse level (launch_chunk)
    ORBX_HIP(hipHostMalloc((void **)&h->h_sparseSeen, sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
    *h->h_sparseSeen = (int32_t)0xC0000000u;   // far behind call 0 in unsigned distance
    ORBX_HIP(hipHostGetDevicePointer((void **)&h->d_sparseSeen, h->h_sparseSeen, 0));
    for (int i = 0; i < ORBX_SIDE_STREAMS; i++) {
        ORBX_HIP(hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
        ORBX_HIP(hipEventCreateWithFlags(&h->evJoin[i], hipEventDisableTiming));
    }
    for (int i = 0; i < ORBX_MAX_CHUNKS; i++) ORBX_HIP(hipEventCreateWithFlags(&h->evPyr[i], hipEventDisableTiming));
    // evFastDone orders the side stream's kernels behind FAST for SPEED only (they touch nothing FAST reads or writes): no memory fence
    // is needed when it fires.
    ORBX_HIP(hipEventCreateWithFlags(&h->evFastDone, hipEventDisableTiming | hipEventDisableSystemFence));
    ORBX_HIP(hipEventCreateWithFlags(&h->evGather, hipEventDisableTiming));
    ORBX_HIP(hipEventCreateWithFlags(&h->evOctA, hipEventDisableTiming));
    ORBX_HIP(hipEventCreateWithFlags(&h->evPrefetch, hipEventDisableTiming | hipEventDisableSystemFence));   // consumer: a kernel of the same device
    for (int r = 0; r < ORBX_EV_RING; r++)
        // timing-only events: no system-scope fence (cache write-back + invalidate) when they complete — with the default
        // flags every stage boundary of a profiled batch cost ~5 us of idle GPU, which the step time then contained
        for (int i = 0; i < ORBX_NUM_STAGES; i++) ORBX_HIP(hipEventCreateWithFlags(&h->ev[r][i], hipEventDisableSystemFence));
    *out = h;
    return ORBX_OK;
}

static void free_plan(orbx_extractor *h) {
    hipFree(h->d_cellRaw); h->d_cellRaw = nullptr;
    hipFree(h->d_octFallback); h->d_octFallback = nullptr;
    hipFree(h->d_pyrAlt); h->d_pyrAlt = nullptr; h->pyrAltBytes = 0; h->pfValid = 0; h->prevPyrValid = 0;
    hipFree(h->d_pyrNext); h->d_pyrNext = nullptr; h->pyrNextBytes = 0;
    hipFree(h->d_blur); hipFree(h->d_blurAlt); h->d_blur = h->d_blurAlt = nullptr; h->blurBytes = h->blurAltBytes = 0; h->blurMaskLast = h->blurMaskAlt = 0;
    hipFree(h->d_octPart); hipFree(h->d_octLeaf); hipFree(h->d_octBest); hipFree(h->d_octState);
    h->d_octPart = nullptr; h->d_octLeaf = nullptr; h->d_octBest = nullptr; h->d_octState = nullptr;
    hipFree(h->d_geom); hipFree(h->d_tab); hipFree(h->d_pyr); hipFree(h->d_cellCnt); hipFree(h->d_slots);
    hipFree(h->d_cand); hipFree(h->d_lvlKp); hipFree(h->d_nodeOf); hipFree(h->d_candCnt); hipFree(h->d_lvlCnt);
    hipFree(h->d_sparse); h->d_sparse = nullptr;
    hipFree(h->d_histCnt); hipFree(h->d_histBest); h->d_histCnt = h->d_histBest = nullptr;
    hipFree(h->d_octPartBest); hipFree(h->d_octPartCnt); hipFree(h->d_octSliceState); h->d_octPartBest = h->d_octPartCnt = nullptr; h->d_octSliceState = nullptr;
    h->d_geom = nullptr; h->d_tab = nullptr; h->d_pyr = nullptr; h->d_cellCnt = nullptr; h->d_slots = nullptr;
    h->d_cand = nullptr; h->d_lvlKp = nullptr; h->d_nodeOf = nullptr; h->d_candCnt = nullptr; h->d_lvlCnt = nullptr;
    h->pw = h->ph = h->pB = 0;
}

extern "C" int orbx_destroy(orbx_extractor_t *h) {
    if (!h) return ORBX_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->last_valid) hipStreamSynchronize(h->last_stream);
    if (h->st_stream) hipStreamSynchronize(h->st_stream);
    orbx_internal_free_stereo_scratch(h);
    free_plan(h);
    hipFree(h->d_in); hipFree(h->d_kps); hipFree(h->d_desc); hipFree(h->d_counts); hipFree(h->d_dbgBlur);
    hipFree(h->d_sfr); if (h->h_sfr) hipHostFree(h->h_sfr);
    for (int i = 0; i < 2; i++) { hipFree(h->fv_d[i]); if (h->fv_h[i]) hipHostFree(h->fv_h[i]); }
    if (h->fv_stage) hipHostFree(h->fv_stage);
    if (h->fv_flag) hipHostFree(h->fv_flag);
    if (h->h_sparseSeen) hipHostFree(h->h_sparseSeen);
    if (h->h_kps) { hipHostFree(h->h_kps); hipHostFree(h->h_desc); hipHostFree(h->h_counts); }
    for (int r = 0; r < ORBX_EV_RING; r++)
        for (int i = 0; i < ORBX_NUM_STAGES; i++) hipEventDestroy(h->ev[r][i]);
    for (int i = 0; i < ORBX_SIDE_STREAMS; i++) { hipStreamSynchronize(h->side[i]); hipStreamDestroy(h->side[i]); hipEventDestroy(h->evJoin[i]); }
    for (int i = 0; i < ORBX_MAX_CHUNKS; i++) hipEventDestroy(h->evPyr[i]);
    hipEventDestroy(h->evFastDone); hipEventDestroy(h->evPrefetch); hipEventDestroy(h->evGather); hipEventDestroy(h->evOctA);
    hipFree(h->d_kpsB); hipFree(h->d_descB);
    hipStreamDestroy(h->stream);
    delete h;
    return ORBX_OK;
}

extern "C" int orbx_get_levels(const orbx_extractor_t *h) { return h ? h->nlevels : ORBX_ERR_ARG; }
extern "C" float orbx_get_scale_factor(const orbx_extractor_t *h) { return h ? (float)h->scale_factor : 0.f; }
extern "C" int orbx_get_tables(const orbx_extractor_t *h, float *sf, float *isf, float *s2, float *is2,
                               int32_t *nf, int32_t *umax16) {
    if (!h) return ORBX_ERR_ARG;
    for (int i = 0; i < h->nlevels; i++) {
        if (sf) sf[i] = h->sf[i];
        if (isf) isf[i] = h->isf[i];
        if (s2) s2[i] = h->sig2[i];
        if (is2) is2[i] = h->isig2[i];
        if (nf) nf[i] = h->nfeat[i];
    }
    if (umax16) memcpy(umax16, h->umax, sizeof(int32_t) * 16);
    return ORBX_OK;
}
extern "C" int orbx_max_keypoints(const orbx_extractor_t *h) {
    if (!h) return ORBX_ERR_ARG;
    // every level returns at most max(N+2, 4*nIni) nodes; nIni is image dependent (<= 64)
    return h->max_kp > 0 ? h->max_kp : h->nfeatures + 3 * h->nlevels;
}

// Build the size-dependent plan: level geometry (:773-787, :1111-1112), resize coefficient
// tables (cv::resize), quad-tree root tables (:543-569), buffer sizes.
static int ensure_plan(orbx_extractor *h, int w, int hgt, int B) {
    if (h->pw == w && h->ph == hgt && h->pB >= B) return ORBX_OK;
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < ORBX_SIDE_STREAMS; i++) ORBX_HIP(hipStreamSynchronize(h->side[i]));
    const int keepB = (h->pw == w && h->ph == hgt) ? h->pB : 0;
    free_plan(h);
    B = std::max(B, keepB);
    std::vector<int32_t> tab;
    size_t poff = 0, slotOff = 0, keyOff = 0;
    int cellBase = 0, lvlKpOff = 0, maxNodeCap = 0, maxCells = 0, maxPyrWords = 0, maxDeepWords = 0, maxBestWords = 0;
    unsigned bigMask = 0;
    int maxTw = 0, maxTh = 0;
    int kpBound = 0;
    // Depth cap of the quad-tree count / best-key pyramids: the largest that fits the LDS budget of k_octree_pyr (a level that
    // needs more depth than its pyramid has is redone by the exact form inside the same workgroup, so the cap costs speed only)
    int pyrCellCap = 16384;
    for (;; pyrCellCap >>= 2) {
        int mNodeCap = 0, mPyr = 0, mBest = 0, mPath = 0, mCells = 0;
        for (int l = 0; l < h->nlevels; l++) {
            const int lw = cv_round((float)w * h->isf[l]), lh = cv_round((float)hgt * h->isf[l]);
            const int rW = lw - 2 * ORBX_MINB, rH = lh - 2 * ORBX_MINB;
            if (rW < 30 || rH < 30) break;   // reported below
            const int nIni = std::min(std::max((int)roundf((float)rW / rH), 1), ORBX_MAX_ROOTS), N = h->nfeat[l];
            int d = 1;
            while ((nIni << (2 * d)) < 4 * std::max(N, 1) && d < 7) d++;
            while ((nIni << (2 * d)) > pyrCellCap && d > 1) d--;
            mNodeCap = std::max(mNodeCap, std::max(N + 3, 4 * nIni) + 4 * nIni + 1);
            mPyr = std::max(mPyr, nIni * (((1 << (2 * d)) - 1) / 3) + ((nIni << (2 * d)) + 1) / 2 + 1);
            mBest = std::max(mBest, nIni * (((1 << (2 * (d + 1))) - 1) / 3));
            mPath = std::max(mPath, rW + rH);
            mCells = std::max(mCells, (int)((float)rW / 30) * (int)((float)rH / 30));
        }
        int p2 = 1;
        while (p2 < mNodeCap) p2 <<= 1;
        const size_t need = sizeof(unsigned long long) * p2 + (size_t)mNodeCap * (8 + 8 + 16 + 4 + 2 + 1) + 4 * (size_t)mPyr + 2 * (size_t)mPath +
                            4 * (size_t)mBest + 4 * (size_t)(mCells + 1) + 64 + 8;
        if (need <= 150 * 1024 || pyrCellCap <= 64) break;
    }
    for (int l = 0; l < h->nlevels; l++) {
        LevelGeom &g = h->geom[l];
        memset(&g, 0, sizeof(g));
        const float scale = h->isf[l];
        g.w = cv_round((float)w * scale);
        g.h = cv_round((float)hgt * scale);
        g.regW = g.w - 2 * ORBX_MINB;
        g.regH = g.h - 2 * ORBX_MINB;
        if (g.regW < 30 || g.regH < 30) {
            orbx_set_error("level %d is %dx%d: the reference needs (w-32)>=30 and (h-32)>=30 at every level "
                           "(nCols/nRows would be 0, src/ORBextractor.cc:784-787)", l, g.w, g.h);
            return ORBX_ERR_ARG;
        }
        if (g.w > 4095 || g.h > 4095) {
            orbx_set_error("image %dx%d too large: candidate coordinates are packed in 12 bits", w, hgt);
            return ORBX_ERR_UNSUPPORTED;
        }
        g.pstride = (g.w + 2 * ORBX_EDGE + 63) & ~63;
        g.prows = g.h + 2 * ORBX_EDGE;
        g.poff = poff;
        poff += (size_t)g.pstride * g.prows;
        const float W = 30;
        const float width = (float)g.regW, height = (float)g.regH;
        g.nCols = (int)(width / W);
        g.nRows = (int)(height / W);
        g.wCell = (int)ceilf(width / g.nCols);
        g.hCell = (int)ceilf(height / g.nRows);
        g.ncells = g.nCols * g.nRows;
        g.cellBase = cellBase;
        cellBase += g.ncells;
        g.capc = ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2);
        g.slotOff = slotOff;
        slotOff += (size_t)g.ncells * g.capc;
        g.keyOff = keyOff;
        g.keyCap = g.ncells * g.capc;
        if (g.keyCap > 0xFFFFFF) {   // best-key election packs (response << 24 | ~index) in one LDS word
            orbx_set_error("level %d can hold %d FAST candidates (> 2^24): unsupported", l, g.keyCap);
            return ORBX_ERR_UNSUPPORTED;
        }
        keyOff += ((size_t)g.keyCap + 7) & ~(size_t)7;  // 16-B aligned key blocks (uint4 / ushort4 sweeps)
        g.N = h->nfeat[l];
        g.nIni = (int)roundf((float)g.regW / g.regH);
        if (g.nIni < 1 || g.nIni > ORBX_MAX_ROOTS) {
            orbx_set_error("aspect ratio gives %d quad-tree roots at level %d (reference divides by zero "
                           "below 1; supported up to %d)", g.nIni, l, ORBX_MAX_ROOTS);
            return g.nIni < 1 ? ORBX_ERR_ARG : ORBX_ERR_UNSUPPORTED;
        }
        g.nodeCap = std::max(g.N + 3, 4 * g.nIni) + 4 * g.nIni + 1;
        {   // depth of the count pyramid: enough cells for the passes of a dense level, bounded by LDS
            int d = 1;
            while ((g.nIni << (2 * d)) < 4 * std::max(g.N, 1) && d < 7) d++;
            while ((g.nIni << (2 * d)) > pyrCellCap && d > 1) d--;   // (cell codes must fit 16 bits: <= 16384; less when LDS is short)
            g.pyrDepth = d;
            const int words = g.nIni * (((1 << (2 * d)) - 1) / 3) + ((g.nIni << (2 * d)) + 1) / 2 + 1;
            maxPyrWords = std::max(maxPyrWords, words);
            maxBestWords = std::max(maxBestWords, g.nIni * (((1 << (2 * (d + 1))) - 1) / 3));
            maxDeepWords = std::max(maxDeepWords, ((g.nIni << (2 * d)) + 1) / 2 + 1);
            // a level with this many FAST cells carries tens of thousands of keys: in a small batch its quad-tree is shared by several
            // workgroups (single image, orbx_extract host to host: 1920x1080 / 4000 features 418 -> 348 us).  Smaller levels gain
            // nothing reliable: with every level >= 100 cells shared, 1241x376 / 1000 features went 184 -> 176 us, / 2000 features 225 -> 233 us
            if (g.ncells >= 600) bigMask |= 1u << l;
        }
        kpBound += std::max(g.N + 2, 4 * g.nIni);
        maxNodeCap = std::max(maxNodeCap, g.nodeCap);
        maxCells = std::max(maxCells, g.ncells);
        g.lvlKpOff = lvlKpOff;
        lvlKpOff += g.nodeCap;
        g.scale = h->sf[l];
        g.size = (float)(int)(31 * h->sf[l]);
        maxTw = std::max(maxTw, g.wCell + 6);
        maxTh = std::max(maxTh, g.hCell + 6);
        // resize tables of cv::resize(INTER_LINEAR, 8UC1) from level l-1
        if (l > 0) {
            const int sw = h->geom[l - 1].w, shh = h->geom[l - 1].h;
            const double inv_scale_x = (double)g.w / sw, inv_scale_y = (double)g.h / shh;
            const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
            g.xofsOff = (int)tab.size();
            tab.resize(tab.size() + g.w);
            g.xalphaOff = (int)tab.size();
            tab.resize(tab.size() + g.w);
            for (int dx = 0; dx < g.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx + 1 >= sw) {
                    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
                }
                const short a0 = sat_short_round((1.f - fx) * 2048), a1 = sat_short_round(fx * 2048);
                tab[g.xofsOff + dx] = sx;
                tab[g.xalphaOff + dx] = (int32_t)((uint32_t)(uint16_t)a0 | ((uint32_t)(uint16_t)a1 << 16));
            }
            g.yofsOff = (int)tab.size();
            tab.resize(tab.size() + g.h);
            g.ybetaOff = (int)tab.size();
            tab.resize(tab.size() + g.h);
            for (int dy = 0; dy < g.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor(fy);
                fy -= sy;
                const short b0 = sat_short_round((1.f - fy) * 2048), b1 = sat_short_round(fy * 2048);
                tab[g.yofsOff + dy] = sy;
                tab[g.ybetaOff + dy] = (int32_t)((uint32_t)(uint16_t)b0 | ((uint32_t)(uint16_t)b1 << 16));
            }
        }
        // quad-tree roots (:543-569): boxes and the key -> root map, tabulated in host float
        {
            const float hX = (float)g.regW / g.nIni;
            g.rootBoxOff = (int)tab.size();
            for (int i = 0; i <= g.nIni; i++) tab.push_back((int)(hX * (float)i));
            const int words = (g.regW + 3) / 4;
            g.rootTabOff = (int)(tab.size() * 4);
            size_t base = tab.size();
            tab.resize(tab.size() + words, 0);
            uint8_t *rt = (uint8_t *)&tab[base];
            for (int x = 0; x < g.regW; x++) {
                size_t r = (size_t)((float)x / hX);
                if (r >= (size_t)g.nIni) r = g.nIni - 1;
                rt[x] = (uint8_t)r;
            }
            // DivideNode splits x and y independently (:483-484, :513-525), so the path of a key down to depth
            // pyrDepth is (bits decided by x alone) interleaved with (bits decided by y alone): two lookups
            // replace pyrDepth rounds of box arithmetic per key in k_octree_pyr.  Cell code at depth d =
            // (xPath[x] | yPath[y]) >> 2*(pyrDepth - d), child index = xbit | ybit << 1 like child_of().
            const int D = g.pyrDepth;
            g.xPathOff = (int)tab.size();
            for (int x = 0; x < g.regW; x++) {
                const int r = ((const uint8_t *)&tab[base])[x];
                int lo = tab[g.rootBoxOff + r], hi = tab[g.rootBoxOff + r + 1];
                uint32_t p = 0;
                for (int d = 0; d < D; d++) {
                    const int mx = lo + ((hi - lo + 1) >> 1);
                    const uint32_t bit = x < mx ? 0u : 1u;
                    if (bit) lo = mx; else hi = mx;
                    p = (p << 2) | bit;
                }
                tab.push_back((int32_t)(((uint32_t)r << (2 * D)) | p));
            }
            g.yPathOff = (int)tab.size();
            for (int y = 0; y < g.regH; y++) {
                int lo = 0, hi = g.regH;
                uint32_t p = 0;
                for (int d = 0; d < D; d++) {
                    const int my = lo + ((hi - lo + 1) >> 1);
                    const uint32_t bit = y < my ? 0u : 1u;
                    if (bit) lo = my; else hi = my;
                    p = (p << 2) | (bit << 1);
                }
                tab.push_back((int32_t)p);
            }
        }
    }
    {   // fused-pyramid tile spans, per axis: own_l partitions level l, comp_l = own_l + needs of comp_{l+1}
        const int Lc = h->nlevels - 1;
        int T = (int)lrintf((h->opt[3] > 0 ? (float)h->opt[3] : 64.0f) / h->sf[Lc]);
        T = std::max(4, std::min(64, (T + 2) & ~3));
        int maxDim = 0, maxPar = 0;
        for (int axis = 0; axis < 2; axis++) {
            auto dim = [&](int l) { return axis == 0 ? h->geom[l].w : h->geom[l].h; };
            const int nt = (dim(Lc) + T - 1) / T;
            if (axis == 0) h->pyrTilesX = nt; else h->pyrTilesY = nt;
            const size_t off = tab.size();
            tab.resize(off + (size_t)2 * h->nlevels * nt);  // PyrSpan = 4 shorts = 2 int32
            short *sp = (short *)&tab[off];
            if (axis == 0) h->pyrXSpanOff = (int)off; else h->pyrYSpanOff = (int)off;
            for (int t = 0; t < nt; t++) {
                int c0 = 0, c1 = 0, parSum = 0;
                for (int l = Lc; l >= 0; l--) {
                    const int d = dim(l), dc = dim(Lc);
                    const int b0 = std::min(t * T, dc), b1 = std::min((t + 1) * T, dc);
                    const int o0 = (int)((long long)b0 * d / dc), o1 = (t == nt - 1) ? d : (int)((long long)b1 * d / dc);
                    if (l == Lc) { c0 = o0; c1 = o1; }
                    else {
                        // rows/cols of level l read by comp_{l+1} = [c0, c1)
                        const LevelGeom &gn = h->geom[l + 1];
                        const int ofsOff = axis == 0 ? gn.xofsOff : gn.yofsOff;
                        int n0 = tab[ofsOff + c0], n1 = tab[ofsOff + c1 - 1] + 1;
                        n0 = std::min(std::max(n0, 0), d - 1);
                        n1 = std::min(std::max(n1, 0), d - 1);
                        c0 = std::min(n0, o0);
                        c1 = std::max(n1 + 1, o1);
                    }
                    short *e = sp + 4 * ((size_t)l * nt + t);
                    e[0] = (short)o0; e[1] = (short)o1; e[2] = (short)c0; e[3] = (short)c1;
                    maxDim = std::max(maxDim, c1 - c0);
                    parSum += c1 - c0;
                }
                maxPar = std::max(maxPar, parSum);
            }
        }
        h->pyrMaxDim = (maxDim + 3) & ~3;
        h->pyrBufBytes = (h->pyrMaxDim * h->pyrMaxDim + 15) & ~15;
        h->pyrMaxPar = (maxPar + 3) & ~3;
        h->pyrLdsBytes = 2 * (size_t)h->pyrBufBytes + (size_t)h->pyrMaxPar * 16 + 16;
        if (h->pyrLdsBytes > 150 * 1024) { orbx_set_error("pyramid tile needs %zu B of LDS", h->pyrLdsBytes); return ORBX_ERR_UNSUPPORTED; }
    }
    for (int variant = 0; variant < 2; variant++) {   // level chains of small batches (ChainPlan, orbx_extract_dev.h): levels 1 .. nlevels-1 in groups of up to
        // CL levels, the first group the short one - planned twice: chains of up to 4 levels (variant 0) and of up to PC_MAXL = 7 (variant 1)
        const int CL = variant == 0 ? 4 : PC_MAXL;
        h->nChains[variant] = 0;
        const int nbTotal = h->nlevels - 1;
        if (nbTotal >= 1 && h->scale_factor <= 2.0) {
            const int nch = (nbTotal + CL - 1) / CL;
            int la = 0;
            bool ok = true;
            for (int c = 0; c < nch && ok; c++) {
                const int nb = c == 0 ? nbTotal - CL * (nch - 1) : CL, lb = la + nb;
                ChainPlan cp;
                memset(&cp, 0, sizeof(cp));
                cp.la = la; cp.lb = lb;
                const int TY = nb <= 4 ? 8 : 16;   // (long chains: taller tiles, less of the accumulated row halo per owned row)
                bool fits = false;
                for (int TX = 96; TX >= 16 && !fits; TX -= 4) {
                    const size_t mark = tab.size();
                    int maxW = 0, maxRows = 0;
                    std::vector<int> cwv((size_t)(nb + 1), 0), chv((size_t)(nb + 1), 0);
                    for (int axis = 0; axis < 2; axis++) {
                        auto dim = [&](int l) { return axis == 0 ? h->geom[l].w : h->geom[l].h; };
                        const int T = axis == 0 ? TX : TY, dc = dim(lb), nt = (dc + T - 1) / T;
                        if (axis == 0) cp.tilesX = nt; else cp.tilesY = nt;
                        const size_t off = tab.size();
                        tab.resize(off + (size_t)2 * (nb + 1) * nt);   // PyrSpan = 4 shorts = 2 int32
                        if (axis == 0) cp.xSpanOff = (int)off; else cp.ySpanOff = (int)off;
                        for (int t = 0; t < nt; t++) {
                            int c0 = 0, c1 = 0;
                            for (int l = lb; l >= la; l--) {
                                const int d = dim(l);
                                const int b0 = std::min(t * T, dc), b1 = std::min((t + 1) * T, dc);
                                const int o0 = (int)((long long)b0 * d / dc), o1 = (t == nt - 1) ? d : (int)((long long)b1 * d / dc);
                                if (l == lb) { c0 = o0; c1 = o1; }
                                else {   // rows / cols of level l read by comp_{l+1} = [c0, c1)
                                    const LevelGeom &gn = h->geom[l + 1];
                                    const int ofsOff = axis == 0 ? gn.xofsOff : gn.yofsOff;
                                    int n0 = tab[ofsOff + c0], n1 = tab[ofsOff + c1 - 1] + 1;
                                    n0 = std::min(std::max(n0, 0), d - 1);
                                    n1 = std::min(std::max(n1, 0), d - 1);
                                    c0 = std::min(n0, o0);
                                    c1 = std::max(n1 + 1, o1);
                                }
                                short *e = (short *)&tab[off] + 4 * ((size_t)(l - la) * nt + t);
                                e[0] = (short)o0; e[1] = (short)o1; e[2] = (short)c0; e[3] = (short)c1;
                                int &mx = axis == 0 ? cwv[(size_t)(l - la)] : chv[(size_t)(l - la)];
                                mx = std::max(mx, c1 - c0);
                            }
                        }
                    }
                    int buf = 0;
                    for (int k = 0; k <= nb; k++) {
                        if (k > 0) { maxW = std::max(maxW, cwv[(size_t)k]); maxRows = std::max(maxRows, chv[(size_t)k]); }
                        buf = std::max(buf, (((cwv[(size_t)k] + 8 + 3) & ~3)) * chv[(size_t)k]);
                    }
                    cp.bufBytes = (buf + 15) & ~15;
                    cp.maxRows = maxRows;
                    fits = maxW <= 128 && 2 * (size_t)cp.bufBytes + (size_t)nb * maxRows * 8 <= 60 * 1024;
                    if (!fits) tab.resize(mark);
                }
                ok = fits;
                if (ok) h->chains[variant][h->nChains[variant]++] = cp;
                la = lb;
            }
            if (!ok) h->nChains[variant] = 0;
        }
    }
    if (maxTw > 65 || maxTh > 65) { orbx_set_error("cell window %dx%d exceeds 65", maxTw, maxTh); return ORBX_ERR_UNSUPPORTED; }
    {   // k_fast_strips takes the levels whose cells are at most 32 px wide (16 pixel pairs = one DPP row), k_fast_cells the rest
        int nstrips = 0;
        h->stripLevels = 0;
        for (int l = 0; l <= ORBX_MAX_LEVELS; l++) {
            h->stripBase[l] = nstrips;
            if (l < h->nlevels && h->geom[l].wCell <= 32) {
                nstrips += h->geom[l].nRows * ((h->geom[l].nCols + 3) / 4);
                h->stripLevels |= 1u << l;
            }
        }
        h->totalStrips = nstrips;
    }
    h->max_kp = kpBound;
    h->totalCells = cellBase;
    h->maxNodeCap = maxNodeCap;
    h->lvlKpCap = lvlKpOff;
    h->pyrImgBytes = (poff + 255) & ~(size_t)255;
    h->slotsPerImg = slotOff;
    h->keysPerImg = (keyOff + 7) & ~(size_t)7;
    h->fastTileStride = (maxTw + 6 + 3) & ~3;        // dwords per pair-tile row (column = byte offset in the aligned row)
    h->fastScoreStride = (maxTw - 6 + 4 + 3) & ~3;   // bytes per score row: 2-px left halo + >= 2 right
    h->fastTileRows = maxTh;
    h->fastLdsPerWave = (4 * h->fastTileStride * maxTh + h->fastScoreStride * (maxTh - 4) + 15) & ~15;
    {   // octree LDS
        int pow2 = 1;
        while (pow2 < maxNodeCap) pow2 <<= 1;
        const int scratch = std::max(4 * maxNodeCap, maxCells + 1);
        size_t bytes = sizeof(unsigned long long) * pow2 + (size_t)maxNodeCap * (8 * 2 + 4 * 2 + 4 * 2 + 2 * 4 + 2 + 2 + 1) +
                       4 * (size_t)scratch + 64;
        h->octLdsBytes = bytes;
        h->octPyrWords = maxPyrWords;
        int maxPath = 0;   // the level's path tables ride in LDS (k_octree_pyr / k_octree_big)
        for (int l = 0; l < h->nlevels; l++) maxPath = std::max(maxPath, h->geom[l].regW + h->geom[l].regH);
        h->octPyrLdsBytes = sizeof(unsigned long long) * pow2 + (size_t)maxNodeCap * (8 + 8 + 16 + 4 + 2 + 1) + 4 * (size_t)maxPyrWords + 2 * (size_t)maxPath + 4 * (size_t)maxBestWords + 4 * (size_t)(maxCells + 1) + 64 + 8;
        if (h->octPyrLdsBytes > 150 * 1024) { orbx_set_error("quad-tree pyramid needs %zu B of LDS", h->octPyrLdsBytes); return ORBX_ERR_UNSUPPORTED; }
        if (bytes > 150 * 1024) {
            orbx_set_error("quad-tree needs %zu B of LDS (features per level %d): unsupported", bytes, maxNodeCap);
            return ORBX_ERR_UNSUPPORTED;
        }
    }
    const size_t Bz = (size_t)B;
    ORBX_HIP(hipMalloc(&h->d_geom, sizeof(LevelGeom) * ORBX_MAX_LEVELS));
    ORBX_HIP(hipMalloc(&h->d_tab, sizeof(int32_t) * std::max<size_t>(tab.size(), 1)));
    ORBX_HIP(hipMalloc(&h->d_pyr, h->pyrImgBytes * Bz));
    ORBX_HIP(hipMalloc(&h->d_cellCnt, sizeof(uint32_t) * h->totalCells * Bz));
    ORBX_HIP(hipMalloc(&h->d_cellRaw, sizeof(uint32_t) * h->totalCells * Bz));
    ORBX_HIP(hipMalloc(&h->d_slots, sizeof(uint32_t) * (h->slotsPerImg * Bz + 64)));   // + 64: the 16-byte list reads of k_octree_pyr may run past the last cell's block
    ORBX_HIP(hipMalloc(&h->d_cand, sizeof(uint32_t) * h->keysPerImg * Bz));
    ORBX_HIP(hipMalloc(&h->d_nodeOf, sizeof(uint16_t) * h->keysPerImg * Bz));
    ORBX_HIP(hipMalloc(&h->d_candCnt, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMalloc(&h->d_lvlCnt, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMalloc(&h->d_sparse, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMemset(h->d_sparse, 0, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));   // first call: every level scored row by row
    ORBX_HIP(hipMalloc(&h->d_lvlKp, sizeof(uint32_t) * (size_t)h->lvlKpCap * Bz));
    ORBX_HIP(hipMalloc(&h->d_octFallback, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMemset(h->d_octFallback, 0, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    {   // small batches: the quad-tree's deepest-depth histogram, filled by the FAST stage (FastHist) - 4 image slots
        int maxDeep = 0;
        for (int l = 0; l < h->nlevels; l++) maxDeep = std::max(maxDeep, h->geom[l].nIni << (2 * h->geom[l].pyrDepth));
        h->histStride = (maxDeep + 3) & ~3;
        const size_t words = (size_t)ORBX_HIST_IMAGES * h->nlevels * h->histStride;
        ORBX_HIP(hipMalloc(&h->d_histCnt, sizeof(uint32_t) * words));
        ORBX_HIP(hipMalloc(&h->d_histBest, sizeof(uint32_t) * words));
        ORBX_HIP(hipMemset(h->d_histCnt, 0, sizeof(uint32_t) * words));
        ORBX_HIP(hipMemset(h->d_histBest, 0, sizeof(uint32_t) * words));
    }
    h->octBigMask = bigMask; h->octDeepMax = maxDeepWords;
    {
        const size_t slots = Bz * h->nlevels;
        ORBX_HIP(hipMalloc(&h->d_octPart, sizeof(uint32_t) * slots * OCT_BIG_K * (size_t)maxDeepWords));
        ORBX_HIP(hipMalloc(&h->d_octLeaf, sizeof(uint32_t) * slots * (size_t)maxPyrWords));
        ORBX_HIP(hipMalloc(&h->d_octBest, sizeof(uint32_t) * slots * (size_t)maxNodeCap));
        ORBX_HIP(hipMalloc(&h->d_octState, sizeof(int32_t) * slots * 4));
        ORBX_HIP(hipMemset(h->d_octState, 0, sizeof(int32_t) * slots * 4));
        // shared sweeps of large levels in a batch (OctSrc::nslice): best-key partials beside the count partials of d_octPart, arrival counters
        h->octSliceStride = 0;
        for (int l = 0; l < h->nlevels; l++) if (h->geom[l].ncells >= 600) h->octSliceStride = std::max(h->octSliceStride, (h->geom[l].nIni << (2 * h->geom[l].pyrDepth)));
        if (h->octSliceStride > 0) {
            ORBX_HIP(hipMalloc(&h->d_octPartBest, sizeof(uint32_t) * slots * OCT_MAX_SLICES * (size_t)h->octSliceStride));
            ORBX_HIP(hipMalloc(&h->d_octPartCnt, sizeof(uint32_t) * slots * OCT_MAX_SLICES * (size_t)h->octSliceStride));
            ORBX_HIP(hipMalloc(&h->d_octSliceState, sizeof(int32_t) * slots));
            ORBX_HIP(hipMemset(h->d_octSliceState, 0, sizeof(int32_t) * slots));
        }
    }
    ORBX_HIP(hipMemcpy(h->d_geom, h->geom, sizeof(LevelGeom) * ORBX_MAX_LEVELS, hipMemcpyHostToDevice));
    if (!tab.empty()) ORBX_HIP(hipMemcpy(h->d_tab, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice));
    h->pw = w; h->ph = hgt; h->pB = B;
    return ORBX_OK;
}

// add the finished event set `slot` to the per-stage accumulators
static int harvest_events(orbx_extractor *h, int slot) {
    hipEvent_t *ev = h->ev[slot];
    if (h->ev_pending[slot] == 2) {   // mode 2: only the FAST kernel was bracketed
        float ms = 0;
        ORBX_HIP(hipEventSynchronize(ev[2]));
        ORBX_HIP(hipEventElapsedTime(&ms, ev[1], ev[2]));
        h->acc_ms[1] += ms;
        h->acc_n++;
        h->ev_pending[slot] = 0;
        return ORBX_OK;
    }
    ORBX_HIP(hipEventSynchronize(ev[4]));
    for (int i = 0; i < 4; i++) {
        float ms = 0;
        ORBX_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        h->acc_ms[i] += ms;
    }
    float tot = 0;
    ORBX_HIP(hipEventElapsedTime(&tot, ev[0], ev[4]));
    h->acc_ms[4] += tot;
    h->acc_n++;
    h->ev_pending[slot] = 0;
    return ORBX_OK;
}

__global__ void k_nop() {}

// Which levels are blurred as a whole (k_blur_levels) instead of per keypoint inside k_describe, and their tiles.  A level of P_l
// pixels with a budget of N_l keypoints costs N_l * 43 * 40 blurred pixels per keypoint block in the fused form; the level-wide
// form costs P_l once plus a second staging load per keypoint.  Rule: N_l * 37^2 * 100 >= thr * P_l (developer knob 14 = thr in
// percent; knob 13: 1 = never, 2 = every level).  Results are identical either way.
// MEASURED (round 3, MI355X): the level-wide form LOSES at every size tried, so the default threshold is out of reach and the
// form stays a tested alternative.  64 stereo frames 1241x376: 1000 features 0.650 ms per step with no level blurred as a whole,
// 0.694 with the rule at 120 %, 0.901 with every level; 2000 features 0.858 / 1.047 / 1.047; 32 images 1920x1080 / 4000 features
// 0.736 / 0.724 / 0.925.  Alone on the GPU at 2000 features: k_describe 0.325 -> 0.19 ms (its staging grows from 9 to 12 loads per
// lane, which is what it is bound by once the blur passes are gone), k_blur_levels 0.26 ms (its row loop serialises on the
// reflected-border branches; a branch-free version would still move 2.4 bytes per pixel: ~0.08 ms), so even the best case wins
// ~0.05 of 0.89 ms at 2000 features and nothing at 1000.
#define ORBX_BLUR_THR 100000000
// a level whose last call kept fewer FAST candidates per 30-px cell than this is "corner-sparse" (the benchmark's dense frames: ~60;
// smooth natural scenes: 2-4)
#define ORBX_SPARSE_PER_CELL 16
static unsigned blur_plan(const orbx_extractor *h, BlurPlan &bp, int &totalTiles) {
    unsigned mask = 0;
    totalTiles = 0;
    const long thr = h->opt[14] > 0 ? h->opt[14] : ORBX_BLUR_THR;
    for (int l = 0; l <= ORBX_MAX_LEVELS; l++) {
        bp.tileBase[l] = totalTiles;
        if (l >= h->nlevels) continue;
        const LevelGeom &g = h->geom[l];
        const bool on = h->opt[13] == 2 || (h->opt[13] == 0 && (long)g.N * 1369 * 100 >= thr * (long)g.w * g.h);
        bp.tilesX[l] = 0;
        if (!on) continue;
        mask |= 1u << l;
        const int ndw = ((g.w + ORBX_EDGE - 1) >> 2) - 4 + 1;
        bp.tilesX[l] = (ndw + 63) / 64;
        totalTiles += bp.tilesX[l] * ((g.h + BLUR_R - 1) / BLUR_R);
    }
    return mask;
}
// the Gaussian's taps as the kernels take them: k3 | k2 << 8 | k1 << 16 | k0 << 24 (orbx_flavour_t lists the centre first)
static uint32_t gauss_taps_packed(const orbx_extractor *h) {
    if (h->flavour.gauss_rounding != ORBX_GAUSS_FIXED_TAPS) return ORBX_GAUSS_TAPS_DEFAULT;
    const int32_t *k = h->flavour.gauss_taps;
    return (uint32_t)k[3] | ((uint32_t)k[2] << 8) | ((uint32_t)k[1] << 16) | ((uint32_t)k[0] << 24);
}
static int ensure_blur(orbx_extractor *h, uint8_t **buf, size_t *bytes) {
    const size_t need = h->pyrImgBytes * (size_t)h->pB;
    if (*bytes >= need) return ORBX_OK;
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < ORBX_SIDE_STREAMS; i++) ORBX_HIP(hipStreamSynchronize(h->side[i]));
    hipFree(*buf); *buf = nullptr; *bytes = 0;
    ORBX_HIP(hipMalloc(buf, need));
    *bytes = need;
    return ORBX_OK;
}
// the level-wide blur behind a pyramid (same stream); returns the mask of blurred levels through *maskOut
static int launch_blur(orbx_extractor *h, const uint8_t *pyr, uint8_t **blurBuf, size_t *blurBytes, int b0, int B, hipStream_t st, unsigned *maskOut) {
    BlurPlan bp;
    int tiles = 0;
    const unsigned mask = blur_plan(h, bp, tiles);
    *maskOut = mask;
    if (!mask) return ORBX_OK;
    int rc = ensure_blur(h, blurBuf, blurBytes);
    if (rc) return rc;
    const size_t off = (size_t)b0 * h->pyrImgBytes;   // images [b0, b0 + B) of both buffers
    hipLaunchKernelGGL(k_blur_levels, dim3((tiles + 3) / 4, B), dim3(256), 0, st, pyr + off, *blurBuf + off, h->pyrImgBytes, h->d_geom, h->nlevels, tiles, bp,
                       h->flavour.gauss_rounding, gauss_taps_packed(h));
    return ORBX_OK;
}

// K1: ComputePyramid of B images into pyr.  Developer knob 5: 0 / 2 = one launch per level (the default), 1 = every level in the
// fused launch, 3 = hybrid.
static bool pyramid_fused_all(const orbx_extractor *h) { return h->opt[5] == 1 || h->scale_factor > 3.0; }
static void launch_pyramid(orbx_extractor *h, const uint8_t *d_imgs, uint8_t *pyr, int B, int stride, size_t img_stride, hipStream_t st) {
    const int nl = h->nlevels;
    if (pyramid_fused_all(h)) {   // writes every frame itself (a lane's two source byte pairs fit 8 bytes only up to scale 3)
        hipLaunchKernelGGL(k_pyramid_fused, dim3(h->pyrTilesX * h->pyrTilesY, B), dim3(256), h->pyrLdsBytes, st, d_imgs,
                           stride, img_stride, pyr, h->pyrImgBytes, h->d_geom, nl, h->d_tab, h->pyrXSpanOff,
                           h->pyrYSpanOff, h->pyrTilesX, h->pyrTilesY, h->pyrBufBytes, h->pyrMaxPar, 0);
        return;
    }
    // Hybrid form (developer knob 5 = 3, parity-tested, NOT used by default): levels 1, 2 one launch each, levels 3.. from ONE launch
    // that chains them through LDS - VERDICT r01's proposal for the launch-latency-bound small levels.  Measured: one 1241x376
    // image 193 us per orbx_extract call against 184 us with seven launches (the chain's five barriers and its generic per-pixel
    // indexing cost more than five ~4-us launches), and for a batch it spends five times the instructions per pixel while the
    // pyramid runs beside VALU-bound kernels.
    const bool hybrid = h->opt[5] == 3;
    const int lastSingle = hybrid ? std::min(2, nl - 1) : nl - 1;
    const LevelGeom &g0 = h->geom[0];
    // level 0 of a small batch: source rows staged in LDS (ORBX_OPT_PAD_FORM: 0 = by batch size, 1 = never, 2 = always) - the image of a latency
    // call is read in pinned host memory, where each byte should cross the bus once, in aligned transactions
    const size_t padLds = (size_t)PAD_ROWS_PER_BLOCK * stride + 32;
    if ((h->opt[24] == 2 || (h->opt[24] == 0 && B <= ORBX_HIST_IMAGES)) && padLds <= 60 * 1024)
        hipLaunchKernelGGL(k_pyr_pad_rows, dim3((g0.h + PAD_ROWS_PER_BLOCK - 1) / PAD_ROWS_PER_BLOCK, B), dim3(256), padLds, st, d_imgs, stride, img_stride, pyr,
                           h->pyrImgBytes, h->d_geom);
    else
        hipLaunchKernelGGL(k_pyr_pad<true>, dim3(((g0.pstride >> 4) * g0.prows + 255) / 256, 1, B), dim3(256), 0, st, d_imgs, stride,
                           img_stride, pyr, h->pyrImgBytes, h->d_geom, 0);
    // ORBX_OPT_PYRAMID_FORM = 4 (and small batches by default, ORBX_OPT_PYR_CHAINS): the levels in chains, one launch per chain
    const int cv = h->opt[25] == 3 ? 1 : 0;   // ORBX_OPT_PYR_CHAINS: 2 / 3 = chains of up to 4 / 7 levels (0: the default, 1: never)
    if (h->nChains[cv] > 0 && !hybrid && (h->opt[5] == 4 || (h->opt[5] == 0 && h->opt[25] != 1 && B <= ORBX_HIST_IMAGES))) {
        for (int c = 0; c < h->nChains[cv]; c++) {
            const ChainPlan &cp = h->chains[cv][c];
            const size_t lds = 2 * (size_t)cp.bufBytes + (size_t)(cp.lb - cp.la) * cp.maxRows * 8;
            hipLaunchKernelGGL(k_pyr_chain, dim3(cp.tilesX * cp.tilesY, B), dim3(256), lds, st, pyr, h->pyrImgBytes, h->d_geom, h->d_tab, cp);
        }
        return;
    }
    for (int l = 1; l <= lastSingle; l++) {
        // 16 output rows per wave while that still leaves every SIMD several waves (8192 = 8 per SIMD), else 8; same pixels either way
        const int nxc = (h->geom[l].w + 1 + 127) / 128, nb16 = (h->geom[l].h + 15) / 16, nb8 = (h->geom[l].h + 7) / 8;
        const bool tall = h->scale_factor <= 1.25f && (h->opt[22] == 0 ? (size_t)nxc * nb16 * B >= 8192 : h->opt[22] == 2);   // (beyond 1.25 the 16-row form cannot cover a band from 22 source rows)
        if (tall)
            hipLaunchKernelGGL((k_pyr_level<16, 22>), dim3((nxc * nb16 + 3) / 4, B), dim3(256), 0, st, pyr, h->pyrImgBytes, h->d_geom, l, h->d_tab, nxc, nb16);
        else
            hipLaunchKernelGGL((k_pyr_level<8, 12>), dim3((nxc * nb8 + 3) / 4, B), dim3(256), 0, st, pyr, h->pyrImgBytes, h->d_geom, l, h->d_tab, nxc, nb8);
    }
    if (lastSingle < nl - 1)
        hipLaunchKernelGGL(k_pyramid_fused, dim3(h->pyrTilesX * h->pyrTilesY, B), dim3(256), h->pyrLdsBytes, st, d_imgs,
                           stride, img_stride, pyr, h->pyrImgBytes, h->d_geom, nl, h->d_tab, h->pyrXSpanOff,
                           h->pyrYSpanOff, h->pyrTilesX, h->pyrTilesY, h->pyrBufBytes, h->pyrMaxPar, lastSingle);
}

// The per-image buffers of a handle as seen by ONE chunk of a batch: every base pointer already points at the chunk's first image.
struct ChunkView {
    const uint8_t *imgs; uint8_t *pyr; uint32_t *cellCnt, *cellRaw, *slots, *cand, *lvlKp; uint16_t *nodeOf;
    int32_t *candCnt, *lvlCnt, *octFallback, *sparse; orbx_keypoint_t *kps; uint8_t *desc; int32_t *counts;
    int b0;            // first image of the chunk
    size_t octSlot0;   // first (image, level) slot of the chunk in the multi-workgroup quad-tree scratch (d_octPart / Leaf / Best / State)
};
static ChunkView chunk_view(const orbx_extractor *h, const uint8_t *d_imgs, size_t img_stride, orbx_keypoint_t *d_kps, uint8_t *d_desc,
                            int32_t *d_counts, int cap, int b0) {
    ChunkView v;
    const size_t z = (size_t)b0;
    v.imgs = d_imgs + z * img_stride; v.pyr = h->d_pyr + z * h->pyrImgBytes;
    v.cellCnt = h->d_cellCnt + z * h->totalCells; v.cellRaw = h->d_cellRaw + z * h->totalCells;
    v.slots = h->d_slots + z * h->slotsPerImg; v.cand = h->d_cand + z * h->keysPerImg; v.nodeOf = h->d_nodeOf + z * h->keysPerImg;
    v.lvlKp = h->d_lvlKp + z * h->lvlKpCap;
    v.sparse = h->d_sparse + z * h->nlevels;
    v.candCnt = h->d_candCnt + z * h->nlevels; v.lvlCnt = h->d_lvlCnt + z * h->nlevels; v.octFallback = h->d_octFallback + z * h->nlevels;
    v.kps = d_kps + z * cap; v.desc = d_desc + z * cap * 32; v.counts = d_counts + b0;
    v.octSlot0 = z * h->nlevels; v.b0 = b0;
    return v;
}

// One chunk (B images from v) through the whole pipeline on stream st.  ev / prof / profFast: this chunk carries the stage events;
// evPyrDone (may be NULL) is recorded behind the chunk's pyramid.
static void launch_pyramid(orbx_extractor *h, const uint8_t *d_imgs, uint8_t *pyr, int B, int stride, size_t img_stride, hipStream_t st);
static int launch_chunk(orbx_extractor *h, const ChunkView &v, int B, int stride, size_t img_stride, int cap, hipStream_t st,
                        hipEvent_t *ev, bool prof, bool profFast, hipEvent_t evPyrDone, bool skipPyr) {
    const int nl = h->nlevels;
    CellBases cb;
    for (int l = 0; l <= ORBX_MAX_LEVELS; l++) cb.v[l] = l < nl ? h->geom[l].cellBase : h->totalCells;
    const uint8_t *d_imgs = v.imgs;
    orbx_keypoint_t *d_kps = v.kps; uint8_t *d_desc = v.desc; int32_t *d_counts = v.counts;
    int aSplit = 0;   // > 0: split call, the levels [0, aSplit) and [aSplit, nl) take different streams behind k_gather
    if (prof) ORBX_HIP(hipEventRecord(ev[0], st));
    if (!skipPyr) {   // K1 (skipped when the pyramid was built ahead: then its level-wide blur was, too)
        launch_pyramid(h, d_imgs, v.pyr, B, stride, img_stride, st);
        unsigned m = 0;
        int rc = launch_blur(h, h->d_pyr, &h->d_blur, &h->blurBytes, v.b0, B, st, &m);
        if (rc) return rc;
        h->blurMaskLast = m;
    }
    if (evPyrDone) ORBX_HIP(hipEventRecord(evPyrDone, st));
    // With the pyramid built ahead nothing but a stream wait (for that pyramid) sits in front of the FAST launch, and a timing
    // event recorded right behind a pending wait can be stamped before the wait is over: the bracket then reads wait + FAST
    // (seen as 0.30 instead of 0.27 ms in one run out of four).  An empty kernel orders the stamp behind the wait.
    if (profFast && skipPyr && h->opt[12] == 0) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st);
    if (profFast) ORBX_HIP(hipEventRecord(ev[1], st));
    // (ORBX_OPT_PREFETCH_GATE = 3: a pyramid built ahead may start as soon as THIS call's FAST stage may - it then runs beside it)
    if (h->pfUsed && evPyrDone == nullptr && h->opt[10] == 3) ORBX_HIP(hipEventRecord(h->evFastDone, st));
    // decisions of the quad-tree stage that the FAST stage needs to know
    // developer knob 4: 0 default, 1 = the exact form alone, 2 = EVERY level by the multi-workgroup form, 3 = none
    const bool usePyr = h->opt[4] != 1;
    // (round 5: a batch whose FAST stage histograms its emissions for the quad-tree - at most ORBX_HIST_IMAGES images, every level by
    // k_fast_cells - needs no shared sweep at all: one 1920x1080 image went through gather 6 + k_octree_big 53 + 15 us)
    const bool stripsWanted = h->totalStrips > 0 && (h->opt[6] == 0 ? (size_t)h->totalStrips * B >= 4096 : h->opt[6] == 3);
    const bool histWanted = h->opt[23] == 0 && B <= ORBX_HIST_IMAGES && h->lastChunks == 1 && !stripsWanted && h->opt[0] == 0 && h->opt[18] != 1 &&
                            h->opt[7] == 0 && h->opt[1] == 0;
    const bool multiWg = h->opt[4] == 2 || (h->opt[4] != 3 && B <= 4 && h->octBigMask != 0 && !histWanted);
    // Fused: k_octree_pyr reads the FAST stage's cell lists in place (no k_gather launch, no compacted key array: -35 us per
    // 128 images 1241x376, -200 us per 64 images 1920x1080 in the pipelined step).  Not for the multi-workgroup form, the exact
    // form alone and the phase-stop knobs, which sweep the compacted array (developer knob 18 = 1: never fused).
    const bool fused = usePyr && !multiWg && (h->opt[7] == 0 || h->opt[7] == 8 || h->opt[7] == 9) && h->opt[1] == 0 && h->opt[18] != 1;   // (7 = 8 / 9, developer build: time stamps, no stop)
    const int sparsePerCell = h->opt[16] == 2 ? 1 << 20 : ORBX_SPARSE_PER_CELL;
    OctSrc osrc = {};
    if (fused) {
        osrc.cellCnt = v.cellCnt; osrc.cellRaw = v.cellRaw; osrc.slots = v.slots; osrc.slotsPerImg = h->slotsPerImg;
        osrc.totalCells = h->totalCells; osrc.iniTh = h->ini_th; osrc.minTh = h->min_th; osrc.candCntOut = v.candCnt;
        osrc.sparseFlag = v.sparse; osrc.sparsePerCell = sparsePerCell; osrc.candOut = v.cand;
        osrc.sparseSeen = h->opt[20] != 0 ? h->d_sparseSeen : nullptr; osrc.callSeq = h->callSeq;   // (the hint only serves the compaction forms)
        h->candStale = std::max(h->candStale, v.b0 + B);
    }
    // ORBX_OPT_OCT_HIST (0 = by batch size, 1 = never): with at most ORBX_HIST_IMAGES images and every level done by k_fast_cells, the FAST
    // stage histograms its emissions for the quad-tree (FastHist) and k_octree_pyr loads the histogram instead of sweeping the keys
    const bool histOct = fused && histWanted;
    FastHist fhist = {};
    if (histOct) {
        fhist.cnt = h->d_histCnt; fhist.best = h->d_histBest; fhist.stride = h->histStride; fhist.tab = h->d_tab;
        osrc.histCnt = h->d_histCnt; osrc.histBest = h->d_histBest; osrc.histStride = h->histStride;
    }
    int pow2 = 1;
    while (pow2 < h->maxNodeCap) pow2 <<= 1;
    int maxCellsL = 0;
    for (int l = 0; l < nl; l++) maxCellsL = std::max(maxCellsL, h->geom[l].ncells);
    const int scratch = std::max(4 * h->maxNodeCap, maxCellsL + 1);
    const size_t ldsOct = std::max(h->octPyrLdsBytes, h->octLdsBytes);
    const bool wideOct = h->opt[11] == 0 ? h->octBigMask != 0 : h->opt[11] == 2;
    bool compact = false;   // the compaction kernel takes the corner-sparse (image, level)s of the strip levels
    int earlyLv = 0;   // > 0: the strips of the levels [0, earlyLv) are launched first and their quad-tree starts beside the FAST of the rest
    {   // K2
        // developer knob 6: 1 = every level by k_fast_cells (compile-time tile strides), 2 = ... with run-time strides
        // a strip is a longer job than a cell (a wave walks ~33 rows): with few images the one-wave-per-cell kernel finishes
        // sooner (13 vs 29 us for one 1241x376 image); once the strips fill the GPU they win (2.5 vs 3.1 us per image).  Same results.
        const bool strips = h->totalStrips > 0 && (h->opt[6] == 0 ? (size_t)h->totalStrips * B >= 4096 : h->opt[6] == 3);
        const unsigned stripLevels = strips ? h->stripLevels : 0u;
        if (strips) {
            StripBases sb;
            for (int l = 0; l <= ORBX_MAX_LEVELS; l++) sb.v[l] = h->stripBase[l];
            const int32_t *spf = h->opt[16] == 1 ? (const int32_t *)nullptr : v.sparse;   // ORBX_OPT_ROW_PRETEST: 1 = never the sparse path, 2 = always
            // ORBX_OPT_SPARSE_FORM 1 / 2 (alternatives, measured no faster than the default row skip inside the strip kernel - DESIGN.md
            // section 5): the corner-sparse (image, level)s - flagged by the previous call's quad-tree - leave the strip kernel and are done
            // by a compaction kernel (below).  That costs a launch whose waves all return at once when nothing is flagged, so it is added
            // only while the handle has recently met a sparse level: the quad-tree of image slot 0 stores the call's sequence number
            // into a host-mapped word when it flags one (a hint that lags by the calls in flight; a wrong hint costs speed only,
            // because both kernels take the SAME device flags).
            compact = spf != nullptr && h->opt[20] != 0 &&
                      (h->opt[16] == 2 || (h->h_sparseSeen && (uint32_t)h->callSeq - (uint32_t)*(volatile int32_t *)h->h_sparseSeen <= 16u));   // (sequence numbers wrap: unsigned distance)
            // Early quad-tree (developer knob 19: a >= 2 = levels [0, a); default 0 = off): the quad-tree of the large levels is ONE
            // workgroup per level walking a serial chain - the critical path behind FAST.  Their strips go first, in a launch of their
            // own, and their quad-tree starts on a second stream as soon as that launch is done, beside the FAST of the remaining
            // levels.  Parity-tested and MEASURED SLOWER at every size (64 stereo frames 1241x376: 0.635 -> 0.649 ms per step with
            // a = 2, 0.657 with a = 3; 2000 features 0.838 -> 0.874; 1920x1080 x 64 1.254 -> 1.274; 752x480 0.599 -> 0.607): FAST loses to
            // the quad-tree workgroups what the shorter chain behind it gains, plus two cross-stream events.  Off by default.
            const int ea = h->opt[19];
            if (fused && !prof && !compact && h->opt[19] >= 2 && h->opt[15] < 2 && h->lastChunks == 1 && B >= 8 && nl > ea &&
                (h->stripLevels & ((1u << ea) - 1u)) == (1u << ea) - 1u && h->d_dbgBlur == nullptr)
                earlyLv = ea;
            const int sA = earlyLv ? h->stripBase[earlyLv] : 0;
            if (earlyLv) {
                hipLaunchKernelGGL(k_fast_strips, dim3((sA + FAST_WAVES - 1) / FAST_WAVES, B), dim3(64 * FAST_WAVES), 0, st,
                                   v.pyr, h->pyrImgBytes, h->d_geom, nl, sA, h->totalCells, v.cellCnt, v.cellRaw,
                                   v.slots, h->slotsPerImg, h->ini_th, h->min_th, sb, spf, 0, compact ? 1 : 0);
                hipStream_t s2 = h->side[1];
                ORBX_HIP(hipEventRecord(h->evGather, st));
                ORBX_HIP(hipStreamWaitEvent(s2, h->evGather, 0));
                if (wideOct) {
                    ORBX_HIP(hipFuncSetAttribute((const void *)k_octree_pyr_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsOct));
                    hipLaunchKernelGGL(k_octree_pyr_wide, dim3(B, earlyLv), dim3(OCT_T_WIDE), ldsOct, s2, h->d_geom, nl, v.cand, h->keysPerImg, v.candCnt,
                                       v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap, pow2, h->octPyrWords, v.octFallback, 0, v.nodeOf,
                                       scratch, 0, 0u, 0, osrc);
                } else {
                    ORBX_HIP(hipFuncSetAttribute((const void *)k_octree_pyr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsOct));
                    hipLaunchKernelGGL(k_octree_pyr, dim3(B, earlyLv), dim3(OCT_T), ldsOct, s2, h->d_geom, nl, v.cand, h->keysPerImg, v.candCnt,
                                       v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap, pow2, h->octPyrWords, v.octFallback, 0, v.nodeOf,
                                       scratch, 0, 0u, 0, osrc);
                }
                ORBX_HIP(hipEventRecord(h->evOctA, s2));
            }
            if (h->totalStrips > sA)
                hipLaunchKernelGGL(k_fast_strips, dim3((h->totalStrips - sA + FAST_WAVES - 1) / FAST_WAVES, B), dim3(64 * FAST_WAVES), 0, st,
                                   v.pyr, h->pyrImgBytes, h->d_geom, nl, h->totalStrips, h->totalCells, v.cellCnt, v.cellRaw,
                                   v.slots, h->slotsPerImg, h->ini_th, h->min_th, sb, spf, sA, compact ? 1 : 0);
        }
        dim3 grid((h->totalCells + FAST_WAVES - 1) / FAST_WAVES, B);
        const int es = (h->fastScoreStride == h->fastTileStride - 8 && h->opt[6] != 2) ? h->fastTileStride : 0;
#define ORBX_LAUNCH_FAST(EST, SP, LDSW)                                                                                  \
    hipLaunchKernelGGL((k_fast_cells<EST, SP>), grid, dim3(64 * FAST_WAVES), (size_t)(LDSW) * FAST_WAVES, st,              \
                       v.pyr, h->pyrImgBytes, h->d_geom, nl, h->totalCells, v.cellCnt, v.cellRaw, v.slots, \
                       h->slotsPerImg, h->ini_th, h->min_th, h->fastTileStride, h->fastScoreStride, h->fastTileRows, \
                       (LDSW), h->opt[0], cb, stripLevels, v.sparse, (SP) ? FastHist{} : fhist)
        if (compact && h->opt[20] == 1) {   // the flagged (image, level)s: the strip kernel's compaction twin
            StripBases sb;
            for (int l = 0; l <= ORBX_MAX_LEVELS; l++) sb.v[l] = h->stripBase[l];
            hipLaunchKernelGGL(k_fast_strips_sparse, dim3((h->totalStrips + FAST_WAVES - 1) / FAST_WAVES, B), dim3(64 * FAST_WAVES), 0, st,
                               v.pyr, h->pyrImgBytes, h->d_geom, nl, h->totalStrips, h->totalCells, v.cellCnt, v.cellRaw,
                               v.slots, h->slotsPerImg, h->ini_th, h->min_th, sb, v.sparse);
        } else if (compact) {   // ORBX_OPT_SPARSE_FORM = 2: the cell kernel's compaction form (+ the queue of 16-bit entries behind a wave's tiles)
            const int ldsw = h->fastLdsPerWave + ((32 * h->fastTileRows + 15) & ~15);
            switch (es) {
            case 44: ORBX_LAUNCH_FAST(44, true, ldsw); break;
            case 48: ORBX_LAUNCH_FAST(48, true, ldsw); break;
            case 52: ORBX_LAUNCH_FAST(52, true, ldsw); break;
            default: ORBX_LAUNCH_FAST(0, true, ldsw); break;
            }
        }
        if (stripLevels != (1u << nl) - 1u) {   // levels with wider cells (the coarsest ones of small images)
            switch (es) {   // the strides of the usual 30-px cell grids; anything else takes the run-time-stride instance
            case 44: ORBX_LAUNCH_FAST(44, false, h->fastLdsPerWave); break;
            case 48: ORBX_LAUNCH_FAST(48, false, h->fastLdsPerWave); break;
            case 52: ORBX_LAUNCH_FAST(52, false, h->fastLdsPerWave); break;
            default: ORBX_LAUNCH_FAST(0, false, h->fastLdsPerWave); break;
            }
        }
#undef ORBX_LAUNCH_FAST
    }
    if (profFast) ORBX_HIP(hipEventRecord(ev[2], st));
    const bool gate = h->pfUsed && evPyrDone == nullptr;   // a pyramid built ahead starts behind this FAST stage (knob 10: 1 behind the quad-tree, 2 behind the descriptors)
    if (gate && h->opt[10] == 0) ORBX_HIP(hipEventRecord(h->evFastDone, st));
    {   // K3
        if (!fused)
            hipLaunchKernelGGL(k_gather, dim3((h->totalCells + GATHER_CELLS_PER_BLOCK - 1) / GATHER_CELLS_PER_BLOCK, B),
                               dim3(256), 0, st, h->d_geom, nl, h->totalCells, v.cellCnt, v.cellRaw, v.slots,
                               h->slotsPerImg, v.cand, h->keysPerImg, v.candCnt, h->ini_th, h->min_th, cb, v.sparse, sparsePerCell,
                               h->opt[20] != 0 ? h->d_sparseSeen : (int32_t *)nullptr, h->callSeq);
        // developer knob 15: a >= 2 = split call at level a (default 0: one launch sequence)
        aSplit = (usePyr && !prof && h->lastChunks == 1 && B >= 8 && h->opt[7] == 0 && h->opt[1] == 0 && h->opt[15] >= 2 &&
                  !multiWg && h->d_dbgBlur == nullptr)
                     ? std::min(h->opt[15], nl - 1) : 0;   // (default: no split - measured slower)
        if (nl < 3) aSplit = 0;
        if (aSplit > 0) {   // scratch records of the levels [a, nl)
            const size_t need = (size_t)h->pB * cap * 60;
            if (h->splitBytes < need) {
                ORBX_HIP(hipStreamSynchronize(st));
                ORBX_HIP(hipStreamSynchronize(h->side[1]));
                hipFree(h->d_kpsB); hipFree(h->d_descB); h->d_kpsB = nullptr; h->d_descB = nullptr; h->splitBytes = 0;
                ORBX_HIP(hipMalloc(&h->d_kpsB, (size_t)h->pB * cap * sizeof(orbx_keypoint_t)));
                ORBX_HIP(hipMalloc(&h->d_descB, (size_t)h->pB * cap * 32));
                h->splitBytes = need;
            }
        }
        if (usePyr) {   // a level whose tree outgrows the count pyramid is redone by the same block with the exact form: one launch
            const size_t lds = std::max(h->octPyrLdsBytes, h->octLdsBytes);
            // The multi-workgroup form shortens ONE image's critical path (a 1920x1080 level 0: 195 us alone in its workgroup); a
            // batch already fills the GPU with one workgroup per (image, level), and the extra hand-offs then cost more than they save
            // (batch 32 of 1920x1080: 274 us against 215), so it is taken for small batches only.  Same results either way.
            // (only the multi-workgroup form shares levels: it sweeps the COMPACTED keys, which exist only when k_gather ran - i.e. when the call is not fused)
            const unsigned bigMask = !multiWg ? 0u : h->opt[4] == 2 ? (1u << nl) - 1u : h->octBigMask;
            OctBig big = {};
            // the kernels index this scratch by the chunk-local image: chunks that run side by side on two streams get disjoint slots
            big.part = h->d_octPart + v.octSlot0 * OCT_BIG_K * (size_t)h->octDeepMax; big.leaf = h->d_octLeaf + v.octSlot0 * (size_t)h->octPyrWords;
            big.best = h->d_octBest + v.octSlot0 * (size_t)h->maxNodeCap; big.state = h->d_octState + v.octSlot0 * 4;
            big.K = OCT_BIG_K; big.deepMax = h->octDeepMax; big.pyrMax = h->octPyrWords;
            for (int l = 0; l < nl; l++) if ((bigMask >> l) & 1u) big.levelOf[big.nBig++] = l;
            // 1024-thread instances for images with a large level (>= 600 FAST cells; developer knob 11: 1 = never, 2 = always)
            const bool wide = h->opt[11] == 0 ? h->octBigMask != 0 : h->opt[11] == 2;
#define ORBX_OCT_LAUNCH_ON(STREAM, KERN, KERNW, GRID, LDS, ...)                                                            \
    do {                                                                                                                \
        if (wide) {                                                                                                     \
            ORBX_HIP(hipFuncSetAttribute((const void *)KERNW, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
            hipLaunchKernelGGL(KERNW, GRID, dim3(OCT_T_WIDE), LDS, STREAM, __VA_ARGS__);                                \
        } else {                                                                                                        \
            ORBX_HIP(hipFuncSetAttribute((const void *)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));  \
            hipLaunchKernelGGL(KERN, GRID, dim3(OCT_T), LDS, STREAM, __VA_ARGS__);                                      \
        }                                                                                                               \
    } while (0)
#define ORBX_OCT_LAUNCH(KERN, KERNW, GRID, LDS, ...) ORBX_OCT_LAUNCH_ON(st, KERN, KERNW, GRID, LDS, __VA_ARGS__)
            if (big.nBig > 0 && h->opt[7] == 0 && h->opt[1] == 0) {
                // large levels: K workgroups histogram, the last one to arrive runs the passes; the same launch carries the other
                // levels (one workgroup each, listed behind the large ones) ...
                int nall = big.nBig;
                for (int l = 0; l < nl; l++) if (!((bigMask >> l) & 1u)) big.levelOf[nall++] = l;
                ORBX_OCT_LAUNCH(k_octree_big<1>, k_octree_big_wide<1>, dim3(OCT_BIG_K, nl, B), lds, h->d_geom, nl, v.cand, h->keysPerImg,
                                v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap, pow2, h->octPyrWords,
                                v.octFallback, v.nodeOf, scratch, big);
                // ... then K workgroups elect the best key per node, the last one writes the level's keypoints
                ORBX_OCT_LAUNCH(k_octree_big<2>, k_octree_big_wide<2>, dim3(OCT_BIG_K, big.nBig, B), lds, h->d_geom, nl, v.cand, h->keysPerImg,
                                v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap, pow2, h->octPyrWords,
                                v.octFallback, v.nodeOf, scratch, big);
            } else if (earlyLv > 0) {
                // the quad-tree of the levels [0, earlyLv) is already running on the second stream (started behind their strips)
                ORBX_OCT_LAUNCH(k_octree_pyr, k_octree_pyr_wide, dim3(B, nl - earlyLv), lds, h->d_geom, nl, v.cand,
                                h->keysPerImg, v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap,
                                pow2, h->octPyrWords, v.octFallback, 0, v.nodeOf, scratch, 0, 0u, earlyLv, osrc);
                ORBX_HIP(hipStreamWaitEvent(st, h->evOctA, 0));
            } else if (aSplit > 0) {
                // Split call: the quad-tree of the large levels [0, a) - one workgroup per level walking a serial chain, the critical
                // path of this stage - moves to a second stream, and the small levels [a, nl) go ahead on the caller's stream:
                // their quad-tree, then their descriptors (a VALU-bound kernel that fills the GPU) into scratch arrays, beside that
                // chain.  The descriptors of [0, a) follow when both are done and move the scratch records behind their own.
                hipStream_t s2 = h->side[1];
                ORBX_HIP(hipEventRecord(h->evGather, st));
                ORBX_HIP(hipStreamWaitEvent(s2, h->evGather, 0));
                ORBX_OCT_LAUNCH_ON(s2, k_octree_pyr, k_octree_pyr_wide, dim3(B, aSplit), lds, h->d_geom, nl, v.cand,
                                   h->keysPerImg, v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap,
                                   pow2, h->octPyrWords, v.octFallback, 0, v.nodeOf, scratch, 0, 0u, 0, osrc);
                ORBX_HIP(hipEventRecord(h->evOctA, s2));
                ORBX_OCT_LAUNCH(k_octree_pyr, k_octree_pyr_wide, dim3(B, nl - aSplit), lds, h->d_geom, nl, v.cand,
                                h->keysPerImg, v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap,
                                pow2, h->octPyrWords, v.octFallback, 0, v.nodeOf, scratch, 0, 0u, aSplit, osrc);
            } else {   // no large level (or a phase-stop knob is set): one workgroup per level, one launch
                // ... except that in a BATCH the sweep of a large level (>= 600 FAST cells) MAY be shared by two or four workgroups
                // (ORBX_OPT_OCT_SLICES = 1; off by default): the level-0 workgroup of a 1920x1080 image is the critical path of the stage
                // (113 us, 64 of them its sweep), and sharing the sweeps takes the stage ALONE from 120 to 90 us at batch 32 - but the pipelined
                // step gets slower (0.6105 -> 0.6277 ms at batch 32, 1.186 -> 1.256 ms at batch 64): beside the next pyramid and the previous
                // matcher the extra 1024-thread workgroups cost more than the shorter critical path gives back
                OctSrc os = osrc;
                int kmax = 1;
                if (fused && !histOct && h->opt[26] == 1 && h->d_octPartBest && h->opt[7] == 0 && h->opt[1] == 0) {
                    for (int l = 0; l < nl; l++) {
                        const int k = h->geom[l].ncells >= 1600 ? 4 : h->geom[l].ncells >= 600 ? 2 : 1;
                        os.nslice[l] = (unsigned char)k;
                        kmax = std::max(kmax, k);
                    }
                    os.partCnt = h->d_octPartCnt + v.octSlot0 * OCT_MAX_SLICES * (size_t)h->octSliceStride;
                    os.partBest = h->d_octPartBest + v.octSlot0 * OCT_MAX_SLICES * (size_t)h->octSliceStride;
                    os.sliceState = h->d_octSliceState + v.octSlot0;
                    os.maxSlices = OCT_MAX_SLICES; os.partStride = h->octSliceStride;
                }
                dim3 ogrid(B, nl);
                if (kmax > 1) {   // linear grid: every slice of the large levels in front of the small levels
                    os.linear = 1; os.nImages = B;
                    int tot = 0;
                    for (int l = 0; l <= ORBX_MAX_LEVELS; l++) { os.blkPrefix[l] = tot; if (l < nl) tot += B * std::max(1, (int)os.nslice[l]); }
                    ogrid = dim3(tot);
                }
                ORBX_OCT_LAUNCH(k_octree_pyr, k_octree_pyr_wide, ogrid, lds, h->d_geom, nl, v.cand,
                                h->keysPerImg, v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap,
                                pow2, h->octPyrWords, v.octFallback, h->opt[7], v.nodeOf, scratch, h->opt[1], 0u, 0, os);
            }
        } else {        // developer knob 4 = 1: the exact form alone
            const bool wide = h->opt[11] == 0 ? h->octBigMask != 0 : h->opt[11] == 2;
            ORBX_OCT_LAUNCH(k_octree, k_octree_wide, dim3(B, nl), h->octLdsBytes, h->d_geom, nl, v.cand, v.nodeOf,
                            h->keysPerImg, v.candCnt, v.lvlKp, h->lvlKpCap, v.lvlCnt, h->d_tab, h->maxNodeCap, pow2,
                            scratch, h->opt[1]);
#undef ORBX_OCT_LAUNCH
#undef ORBX_OCT_LAUNCH_ON
        }
    }
    if (prof) ORBX_HIP(hipEventRecord(ev[3], st));
    if (gate && h->opt[10] == 1) ORBX_HIP(hipEventRecord(h->evFastDone, st));
    {   // K4 (one instance per flavour of the Gaussian's column rounding: the default pays nothing for the other)
        const bool sse2 = h->flavour.gauss_rounding == ORBX_GAUSS_ROUND_SSE2;
        // ORBX_OPT_DESC_LDS_PAD (KB): unused dynamic LDS per workgroup = fewer resident k_describe workgroups per CU, i.e. wave slots
        // left for the pyramid kernels that run beside it in a pipelined step (tuning only)
        const size_t descPad = (size_t)h->opt[21] * 1024;
        const bool ftaps = h->flavour.gauss_rounding == ORBX_GAUSS_FIXED_TAPS;
        const uint32_t taps = gauss_taps_packed(h);
        const auto kDesc = aSplit > 0 ? (ftaps ? k_describe<ORBX_GAUSS_FIXED_TAPS, true> : sse2 ? k_describe<ORBX_GAUSS_ROUND_SSE2, true> : k_describe<ORBX_GAUSS_ROUND_HALF_UP, true>)
                                      : (ftaps ? k_describe<ORBX_GAUSS_FIXED_TAPS, false> : sse2 ? k_describe<ORBX_GAUSS_ROUND_SSE2, false> : k_describe<ORBX_GAUSS_ROUND_HALF_UP, false>);
        const uint8_t *blurp = h->blurMaskLast ? h->d_blur + (size_t)v.b0 * h->pyrImgBytes : nullptr;
        if (aSplit > 0) {
            int boundA = 0, boundB = 0;
            for (int l = 0; l < nl; l++) (l < aSplit ? boundA : boundB) += std::max(h->geom[l].N + 2, 4 * h->geom[l].nIni);
            const int maxoA = std::min(cap, boundA), maxoB = std::min(cap, boundB);
            orbx_keypoint_t *kB = h->d_kpsB + (size_t)v.b0 * cap;
            uint8_t *dB = h->d_descB + (size_t)v.b0 * cap * 32;
            const int nbB = (maxoB + DESC_WAVES - 1) / DESC_WAVES, nbA = (maxoA + DESC_WAVES - 1) / DESC_WAVES;
            DescGroup gB = {aSplit, nl, 0, nbB, nullptr, nullptr};
            gB.taps = taps;
            hipLaunchKernelGGL(kDesc, dim3(nbB, B), dim3(64 * DESC_WAVES), descPad, st, v.pyr, h->pyrImgBytes, h->d_geom, nl,
                               v.lvlKp, h->lvlKpCap, v.lvlCnt, kB, dB, d_counts, cap, (uint8_t *)nullptr, blurp, h->blurMaskLast, gB);
            ORBX_HIP(hipStreamWaitEvent(st, h->evOctA, 0));
            DescGroup gA = {0, aSplit, 1, nbA, kB, dB};
            gA.taps = taps;
            hipLaunchKernelGGL(kDesc, dim3(nbA + (maxoB + DESC_COPY_PER_BLOCK - 1) / DESC_COPY_PER_BLOCK, B), dim3(64 * DESC_WAVES), descPad, st,
                               v.pyr, h->pyrImgBytes, h->d_geom, nl, v.lvlKp, h->lvlKpCap, v.lvlCnt, d_kps, d_desc, d_counts, cap,
                               (uint8_t *)nullptr, blurp, h->blurMaskLast, gA);
        } else {
            const int maxo = std::min(cap, h->max_kp);
            dim3 grid((maxo + DESC_WAVES - 1) / DESC_WAVES, B);
            DescGroup gAll = {0, nl, 1, (int)grid.x, nullptr, nullptr};
            gAll.taps = taps;
            gAll.hostDelta = h->descHostDelta;   // latency form: records also stored into their pinned host twin
            hipLaunchKernelGGL(kDesc, grid, dim3(64 * DESC_WAVES), descPad, st, v.pyr, h->pyrImgBytes, h->d_geom, nl,
                               v.lvlKp, h->lvlKpCap, v.lvlCnt, d_kps, d_desc, d_counts, cap, h->d_dbgBlur, blurp, h->blurMaskLast, gAll);
        }
    }
    if (prof) ORBX_HIP(hipEventRecord(ev[4], st));
    if (gate && h->opt[10] == 2) ORBX_HIP(hipEventRecord(h->evFastDone, st));
    return ORBX_OK;
}

// Chunks a batch of B images is cut into (developer knob 8; default ONE).  The one rule for launch_pipeline and orbx_fast_kernels.
static int chunk_count(const orbx_extractor *h, int B, bool prof, bool skipPyr) {
    int nch = h->opt[8] <= 1 ? 1 : std::min(h->opt[8], ORBX_MAX_CHUNKS);
    nch = std::min(nch, B);
    if (prof || skipPyr || h->opt[0] || h->opt[1] || h->opt[7]) nch = 1;
    return nch;
}

// A batch runs as up to ORBX_MAX_CHUNKS chunks of images.  Chunk 0 goes to the caller's stream, the others to the handle's side
// streams, and chunk c's pyramid waits for chunk c-1's: the memory-bound pyramid and the latency-bound gather / quad-tree of one
// chunk then overlap the issue-bound FAST and descriptor kernels of another (the kernels are the same, per-image results do not
// depend on the chunking), and the caller's stream waits for the side streams at the end, so the call keeps its stream semantics.
static int launch_pipeline(orbx_extractor *h, const uint8_t *d_imgs, int B, int w, int hgt, int stride,
                           size_t img_stride, orbx_keypoint_t *d_kps, uint8_t *d_desc, int32_t *d_counts,
                           int cap, hipStream_t st, bool skipPyr = false) {
    // profiling 1: events at every stage boundary; 2: only around the FAST stage (an event costs ~4.5 us of idle GPU, so
    // a throughput measurement brackets just the kernel it reports); mode 3 = mode 2 on every 4th call only
    const bool prof = h->profiling == 1, profFast = h->profiling != 0 && (h->profiling != 3 || (h->prof_calls++ & 3) == 0);
    (void)hipGetLastError();  // drop stale errors of other HIP users in this process
    hipEvent_t *ev = nullptr;
    if (profFast) {
        const int slot = h->ev_head % ORBX_EV_RING;
        if (h->ev_pending[slot]) { int rc = harvest_events(h, slot); if (rc) return rc; }
        ev = h->ev[slot];
    }
    // developer knob 8: n >= 2 = n chunks (default: one - measured on 64 stereo frames, two chunks: 773 us against 742, the
    // latency-bound gather / quad-tree do not shrink with the chunk and the pyramid slows the FAST it overlaps by as much as it gains).
    const int nch = chunk_count(h, B, prof, skipPyr);
    h->lastChunks = nch;
    h->candStale = 0;
    h->callSeq = (int)((unsigned)h->callSeq + 1u);   // wraps (compared by unsigned distance)
    h->prevPyrValid = skipPyr ? 1 : 0;   // d_pyr is overwritten unless this call took a pyramid built ahead (then d_pyrAlt keeps the previous one)
    h->framesStale = (!pyramid_fused_all(h) && h->nlevels > 1) ? B : 0;   // frames of levels >= 1: written on demand (ensure_frames)
    int b0 = 0;
    for (int c = 0; c < nch; c++) {
        const int Bc = (B - b0) / (nch - c);
        const ChunkView v = chunk_view(h, d_imgs, img_stride, d_kps, d_desc, d_counts, cap, b0);
        hipStream_t sc = c == 0 ? st : h->side[(c - 1) % ORBX_SIDE_STREAMS];
        if (c > 0) ORBX_HIP(hipStreamWaitEvent(sc, h->evPyr[c - 1], 0));
        int rc = launch_chunk(h, v, Bc, stride, img_stride, cap, sc, ev, prof && c == 0, profFast && c == 0, c + 1 < nch ? h->evPyr[c] : nullptr, skipPyr);
        if (rc) return rc;
        b0 += Bc;
    }
    for (int s = 0; s < std::min(nch - 1, ORBX_SIDE_STREAMS); s++) {   // the caller's stream carries on when every side stream is done
        ORBX_HIP(hipEventRecord(h->evJoin[s], h->side[s]));
        ORBX_HIP(hipStreamWaitEvent(st, h->evJoin[s], 0));
    }
    if (profFast) { h->ev_pending[h->ev_head % ORBX_EV_RING] = (unsigned char)(h->profiling == 3 ? 2 : h->profiling); h->ev_head++; }
    ORBX_HIP(hipGetLastError());
    h->last_stream = st;
    h->last_valid = 1;
    h->lastB = B;
    (void)w; (void)hgt;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_extractor_t *h, const uint8_t *d_imgs, int B, int w, int hgt,
                                         int stride, size_t image_stride_bytes, orbx_keypoint_t *d_kps,
                                         uint8_t *d_desc, int32_t *d_counts, int cap, void *stream) {
    if (!h || !d_imgs || !d_kps || !d_desc || !d_counts || B < 1 || w < 1 || hgt < 1 || stride < w || cap < 1) {
        orbx_set_error("orbx_extract_batch_device: bad arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, w, hgt, B);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default (null) stream, as in every HIP API
    // the pyramid of exactly this batch was built ahead (orbx_extract_batch_device_prefetch): take that buffer, skip K1
    const bool three = h->pyrBuffers == 3;
    const bool ahead = h->pfValid && (three ? h->d_pyrNext != nullptr : h->d_pyrAlt != nullptr) && h->pfImgs == d_imgs && h->pfB == B && h->pfW == w && h->pfH == hgt && h->pfStride == stride &&
                       h->pfImgStride == image_stride_bytes && h->opt[9] == 0;
    h->pfValid = 0;
    h->prevPyrValid = ahead ? 1 : 0;   // the buffers swap: the other one keeps the previous call's pyramid until the next one is built into it
    if (ahead) {
        if (three) {   // previous <- current <- next <- (old previous: its matcher was issued on the side stream before the next prefetch is)
            if (h->pyrAltBytes < h->pyrImgBytes * (size_t)h->pB) {   // first rotation: the "previous" buffer does not exist yet
                ORBX_HIP(hipStreamSynchronize(h->side[0]));
                hipFree(h->d_pyrAlt); h->d_pyrAlt = nullptr;
                ORBX_HIP(hipMalloc(&h->d_pyrAlt, h->pyrImgBytes * (size_t)h->pB));
                h->pyrAltBytes = h->pyrImgBytes * (size_t)h->pB;
            }
            uint8_t *oldPrev = h->d_pyrAlt;
            h->d_pyrAlt = h->d_pyr; h->d_pyr = h->d_pyrNext; h->d_pyrNext = oldPrev;
        } else
            std::swap(h->d_pyr, h->d_pyrAlt);
        std::swap(h->d_blur, h->d_blurAlt); std::swap(h->blurBytes, h->blurAltBytes); std::swap(h->blurMaskLast, h->blurMaskAlt);
        ORBX_HIP(hipStreamWaitEvent(st, h->evPrefetch, 0));
    }
    return launch_pipeline(h, d_imgs, B, w, hgt, stride, image_stride_bytes, d_kps, d_desc, d_counts, cap, st, ahead);
}

// Software pipelining across batches: start ComputePyramid of the NEXT batch now, into the handle's second pyramid buffer, on a
// stream of the handle's own.  It is ordered behind the FAST stage of the extraction call issued last, i.e. it runs beside that
// call's gather / quad-tree (latency-bound), descriptors and the stereo matcher instead of in front of the next call's FAST.
extern "C" int orbx_extract_batch_device_prefetch(orbx_extractor_t *h, const uint8_t *d_imgs, int B, int w, int hgt, int stride,
                                                  size_t image_stride_bytes, void *side_stream) {
    if (!h || !d_imgs || B < 1 || w < 1 || hgt < 1 || stride < w) { orbx_set_error("orbx_extract_batch_device_prefetch: bad arguments"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, w, hgt, B);
    if (rc) return rc;
    const size_t need = h->pyrImgBytes * (size_t)h->pB;
    const bool three = h->pyrBuffers == 3;
    uint8_t **tgt = three ? &h->d_pyrNext : &h->d_pyrAlt;
    size_t *tgtBytes = three ? &h->pyrNextBytes : &h->pyrAltBytes;
    if (*tgtBytes < need) {
        ORBX_HIP(hipStreamSynchronize(h->side[0]));
        hipFree(*tgt); *tgt = nullptr; *tgtBytes = 0;
        ORBX_HIP(hipMalloc(tgt, need));
        *tgtBytes = need;
    }
    hipStream_t sd = side_stream ? (hipStream_t)side_stream : h->side[0];   // the caller's side stream (work it queued there comes first) or the handle's own
    if (!three) h->prevPyrValid = 0;   // two buffers: the pyramid built ahead overwrites the previous one (three: it has a buffer of its own)
    (void)hipGetLastError();
    if (h->pfUsed && h->last_valid) ORBX_HIP(hipStreamWaitEvent(sd, h->evFastDone, 0));   // (the first time there is no such event yet: the
    else if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));             //  second buffer is new, wait for the handle to be idle)
    h->pfUsed = 1;
    launch_pyramid(h, d_imgs, *tgt, B, stride, image_stride_bytes, sd);
    rc = launch_blur(h, *tgt, &h->d_blurAlt, &h->blurAltBytes, 0, B, sd, &h->blurMaskAlt);
    if (rc) return rc;
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipEventRecord(h->evPrefetch, sd));
    h->pfValid = 1; h->pfImgs = d_imgs; h->pfB = B; h->pfW = w; h->pfH = hgt; h->pfStride = stride; h->pfImgStride = image_stride_bytes;
    return ORBX_OK;
}

// Two (default) or three pyramid buffers.  With three, orbx_extract_batch_device_prefetch builds the next pyramid into a buffer of its
// own and the previous call's pyramid stays valid, so a pipeline may issue the pyramid of batch i+1 BEFORE the matcher of batch i-1 on
// its side stream (pipeline.FrontEnd(stereo_late=True): the matcher then runs beside the descriptor kernel instead of beside the
// quad-tree).  The buffer the next prefetch overwrites is the one the matcher issued before it (same stream) has read.
extern "C" int orbx_set_pyramid_buffers(orbx_extractor_t *h, int n) {
    if (!h || (n != 2 && n != 3)) { orbx_set_error("orbx_set_pyramid_buffers: 2 or 3"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < ORBX_SIDE_STREAMS; i++) ORBX_HIP(hipStreamSynchronize(h->side[i]));
    h->pyrBuffers = n;
    h->pfValid = 0; h->prevPyrValid = 0;
    return ORBX_OK;
}

// Orders `stream` behind the FAST stage of the extraction call issued last on h (the event the pyramid built ahead waits for):
// what a caller queues on `stream` afterwards runs beside that call's gather / quad-tree / descriptor kernels, not beside its FAST.
extern "C" int orbx_stream_wait_fast_stage(orbx_extractor_t *h, void *stream) {
    if (!h) { orbx_set_error("orbx_stream_wait_fast_stage: bad arguments"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    if (h->pfUsed && h->last_valid) ORBX_HIP(hipStreamWaitEvent((hipStream_t)stream, h->evFastDone, 0));
    else if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));   // no event yet (nothing was built ahead so far): wait on the host
    return ORBX_OK;
}

// The handle's own side stream (the one pyramids are built ahead on when the caller names none).  A process has few hardware
// queues and the runtime deals streams to them round-robin: a stream the caller creates may share its queue with the caller's main
// stream and then overlaps nothing; this one was created next to the handle and is known to sit on a queue of its own.
extern "C" void *orbx_side_stream(orbx_extractor_t *h) { return h ? (void *)h->side[0] : nullptr; }

// ~2 ms of nothing on one wave: the "busy" side of the queue probe below (constant 100 MHz counter)
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < 100000 && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(32);   // bounded whatever the counter does
}
// true iff a command on `cand` starts while `busy` is still occupied, i.e. the two streams do not share a hardware queue
static bool streams_independent(hipStream_t busy, hipStream_t cand, hipEvent_t eb, hipEvent_t ec) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, busy, 200000ll);
    if (hipEventRecord(eb, busy) != hipSuccess) return false;
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, cand);
    if (hipEventRecord(ec, cand) != hipSuccess) return false;
    if (hipEventSynchronize(ec) != hipSuccess) return false;
    const bool indep = hipEventQuery(eb) == hipErrorNotReady;   // the spin is still running: the candidate did not wait behind it
    (void)hipEventSynchronize(eb);
    (void)hipGetLastError();
    return indep;
}
// The side stream of the handle, made sure NOT to share a hardware queue with `main_stream` (the stream the caller issues its
// extraction calls on).  The runtime deals streams to a handful of hardware queues round-robin and has no API that tells which;
// two streams on one queue run strictly in order, and the software pipeline (matcher of the previous batch, pyramid of the next on
// the side stream beside the caller's kernels) then overlaps nothing - seen on every handle after the first of a process (the second
// FrontEnd of bench.py: 1.53 instead of 1.13 ms per end-to-end step).  So the side stream is probed: a 2-ms one-wave spin goes to
// main_stream, an empty kernel to the candidate; the candidate is kept if its kernel ran while the spin was still going.  One-off,
// a few ms; the caller's stream is occupied for that long.  Returns the (possibly replaced) side stream.
extern "C" void *orbx_side_stream_for(orbx_extractor_t *h, void *main_stream) {
    if (!h) return nullptr;
    if (hipSetDevice(h->device) != hipSuccess) return (void *)h->side[0];
    hipStream_t ms = (hipStream_t)main_stream;
    hipEvent_t eb = nullptr, ec = nullptr;
    if (hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ec, hipEventDisableTiming) != hipSuccess) {
        if (eb) hipEventDestroy(eb);
        return (void *)h->side[0];
    }
    (void)hipStreamSynchronize(ms);
    (void)hipStreamSynchronize(h->side[0]);
    if (!streams_independent(ms, h->side[0], eb, ec)) {
        std::vector<hipStream_t> tried;
        hipStream_t found = nullptr;
        for (int i = 0; i < 12 && !found; i++) {
            hipStream_t c = nullptr;
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
            if (streams_independent(ms, c, eb, ec) && streams_independent(c, h->side[1], eb, ec)) found = c;
            else tried.push_back(c);
        }
        for (hipStream_t c : tried) hipStreamDestroy(c);
        if (found) {
            if (h->st_stream == h->side[0]) h->st_stream = nullptr;       // (nothing may keep the handle of the stream that goes away)
            if (h->last_valid && h->last_stream == h->side[0]) h->last_valid = 0;
            hipStreamDestroy(h->side[0]);
            h->side[0] = found;
            h->pfValid = 0; h->prevPyrValid = 0;
        }
    }
    hipEventDestroy(eb); hipEventDestroy(ec);
    (void)hipGetLastError();
    return (void *)h->side[0];
}

static int ensure_staging(orbx_extractor *h, size_t in_bytes, int B, int cap) {
    if (h->d_in_bytes < in_bytes) {
        hipFree(h->d_in); h->d_in = nullptr; h->d_in_bytes = 0;
        ORBX_HIP(hipMalloc(&h->d_in, in_bytes));
        h->d_in_bytes = in_bytes;
    }
    if (h->out_cap < cap || h->out_B < B) {
        hipFree(h->d_kps); hipFree(h->d_desc); hipFree(h->d_counts);
        h->d_kps = nullptr; h->d_desc = nullptr; h->d_counts = nullptr;
        const int nb = std::max(B, h->out_B), nc = std::max(cap, h->out_cap);
        ORBX_HIP(hipMalloc(&h->d_kps, sizeof(orbx_keypoint_t) * (size_t)nb * nc));
        ORBX_HIP(hipMalloc(&h->d_desc, (size_t)32 * nb * nc));
        ORBX_HIP(hipMalloc(&h->d_counts, sizeof(int32_t) * nb));
        // pinned mirrors: the results of a host-API call come down in three copies and ONE synchronisation
        if (h->h_kps) { hipHostFree(h->h_kps); hipHostFree(h->h_desc); hipHostFree(h->h_counts); h->h_kps = nullptr; h->h_desc = nullptr; h->h_counts = nullptr; }
        ORBX_HIP(hipHostMalloc((void **)&h->h_kps, sizeof(orbx_keypoint_t) * (size_t)nb * nc, hipHostMallocDefault));
        ORBX_HIP(hipHostMalloc((void **)&h->h_desc, (size_t)32 * nb * nc, hipHostMallocDefault));
        ORBX_HIP(hipHostMalloc((void **)&h->h_counts, sizeof(int32_t) * nb, hipHostMallocDefault));
        h->out_cap = nc; h->out_B = nb;
    }
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_extractor_t *h, const uint8_t *const *imgs, int B, int w, int hgt,
                                  int stride, orbx_keypoint_t *kps, uint8_t *desc, int cap, int *n_out) {
    if (!h || !imgs || !kps || !desc || !n_out || B < 1 || cap < 1) {
        orbx_set_error("orbx_extract_batch: bad arguments");
        return ORBX_ERR_ARG;
    }
    if (w <= 0 || hgt <= 0) {  // empty image: silent return (:1046-1047)
        for (int b = 0; b < B; b++) n_out[b] = 0;
        return ORBX_OK;
    }
    if (stride < w) { orbx_set_error("stride < width"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, w, hgt, B);
    if (rc) return rc;
    // One linear copy per image with the caller's row stride kept on the device (the kernels take a stride):
    // a 2-D copy of rows whose width is not a multiple of 4 bytes (1241!) runs ~100x slower than a linear one.
    const size_t span = (size_t)stride * (hgt - 1) + w;
    const size_t img_bytes = (span + 255) & ~(size_t)255;
    rc = ensure_staging(h, img_bytes * B, B, cap);
    if (rc) return rc;
    const int dcap = h->out_cap;
    for (int b = 0; b < B; b++) {
        if (!imgs[b]) { orbx_set_error("imgs[%d] is NULL", b); return ORBX_ERR_ARG; }
        ORBX_HIP(hipMemcpyAsync(h->d_in + img_bytes * b, imgs[b], span, hipMemcpyHostToDevice, h->stream));
    }
    rc = launch_pipeline(h, h->d_in, B, w, hgt, stride, img_bytes, h->d_kps, h->d_desc, h->d_counts, dcap, h->stream);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(h->h_counts, h->d_counts, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP(hipMemcpyAsync(h->h_kps, h->d_kps, sizeof(orbx_keypoint_t) * (size_t)B * dcap, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)32 * B * dcap, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP(hipStreamSynchronize(h->stream));
    int status = ORBX_OK;
    for (int b = 0; b < B; b++) {
        int n = h->h_counts[b];
        if (n > cap) { n = cap; status = ORBX_ERR_CAPACITY; orbx_set_error("frame %d produced %d keypoints, cap %d", b, h->h_counts[b], cap); }
        n_out[b] = n;
        if (n > 0) {
            memcpy(kps + (size_t)b * cap, h->h_kps + (size_t)b * dcap, sizeof(orbx_keypoint_t) * n);
            memcpy(desc + (size_t)b * cap * 32, h->h_desc + (size_t)b * dcap * 32, (size_t)32 * n);
        }
    }
    return status;
}

// One stereo frame host to host in ONE call: what the reference's stereo Frame constructor does with two extractor threads and a CPU
// matcher (src/Frame.cc:78-84: ExtractORB(0, imLeft) || ExtractORB(1, imRight), then ComputeStereoMatches, :481-655).  Both images
// go up, are extracted as one batch of two on this handle (slots 0 / 1), matched on the device, and everything comes down behind ONE
// synchronisation: no second extractor, no re-upload of the keypoints the extractor just produced.
extern "C" int orbx_stereo_frame(orbx_extractor_t *h, const uint8_t *left, const uint8_t *right, int w, int hgt, int stride,
                                 float mbf, float mb, int cap, orbx_keypoint_t *kl, uint8_t *dl, int *nl, orbx_keypoint_t *kr,
                                 uint8_t *dr, int *nr, float *uright, float *depth, int *nmatch) {
    if (!h || !kl || !dl || !nl || !kr || !dr || !nr || !uright || !depth || cap < 1) {
        orbx_set_error("orbx_stereo_frame: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nl = 0; *nr = 0;
    if (nmatch) *nmatch = 0;
    if (!left || !right || w <= 0 || hgt <= 0) return ORBX_OK;   // empty image (:1046-1047)
    if (stride < w) { orbx_set_error("stride < width"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, w, hgt, 2);
    if (rc) return rc;
    const size_t span = (size_t)stride * (hgt - 1) + w, img_bytes = (span + 255) & ~(size_t)255;
    rc = ensure_staging(h, img_bytes * 2, 2, cap);
    if (rc) return rc;
    const int dcap = h->out_cap;
    if (h->sfr_cap < dcap) {
        hipFree(h->d_sfr); h->d_sfr = nullptr;
        if (h->h_sfr) { hipHostFree(h->h_sfr); h->h_sfr = nullptr; }
        h->sfr_cap = 0;
        ORBX_HIP(hipMalloc(&h->d_sfr, sizeof(float) * (2 * (size_t)dcap + 4)));
        ORBX_HIP(hipHostMalloc((void **)&h->h_sfr, sizeof(float) * (2 * (size_t)dcap + 4), hipHostMallocDefault));
        h->sfr_cap = dcap;
    }
    hipStream_t st = h->stream;
    ORBX_HIP(hipMemcpyAsync(h->d_in, left, span, hipMemcpyHostToDevice, st));
    ORBX_HIP(hipMemcpyAsync(h->d_in + img_bytes, right, span, hipMemcpyHostToDevice, st));
    rc = launch_pipeline(h, h->d_in, 2, w, hgt, stride, img_bytes, h->d_kps, h->d_desc, h->d_counts, dcap, st);
    if (rc) return rc;
    float *d_ur = h->d_sfr, *d_dp = h->d_sfr + dcap;
    int32_t *d_nm = (int32_t *)(h->d_sfr + 2 * (size_t)dcap);
    rc = orbm_stereo_batch_device(h, h, 1, 0, 1, h->d_kps, h->d_desc, h->d_counts, h->d_kps + dcap, h->d_desc + (size_t)dcap * 32,
                                  h->d_counts + 1, dcap, mbf, mb, d_ur, d_dp, d_nm, st);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(h->h_counts, h->d_counts, sizeof(int32_t) * 2, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(h->h_kps, h->d_kps, sizeof(orbx_keypoint_t) * 2 * (size_t)dcap, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)32 * 2 * dcap, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(h->h_sfr, h->d_sfr, sizeof(float) * (2 * (size_t)dcap + 4), hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    int status = ORBX_OK;
    int n0 = h->h_counts[0], n1 = h->h_counts[1];
    if (n0 > cap || n1 > cap) {
        orbx_set_error("stereo frame produced %d / %d keypoints, cap %d", n0, n1, cap);
        status = ORBX_ERR_CAPACITY;
        n0 = std::min(n0, cap); n1 = std::min(n1, cap);
    }
    *nl = n0; *nr = n1;
    if (n0 > 0) {
        memcpy(kl, h->h_kps, sizeof(orbx_keypoint_t) * n0);
        memcpy(dl, h->h_desc, (size_t)32 * n0);
        memcpy(uright, h->h_sfr, sizeof(float) * n0);
        memcpy(depth, h->h_sfr + dcap, sizeof(float) * n0);
    }
    if (n1 > 0) {
        memcpy(kr, h->h_kps + dcap, sizeof(orbx_keypoint_t) * n1);
        memcpy(dr, h->h_desc + (size_t)dcap * 32, (size_t)32 * n1);
    }
    if (nmatch) *nmatch = ((const int32_t *)(h->h_sfr + 2 * (size_t)dcap))[0];
    return status;
}

// ---- the latency form: one stereo frame, no copy commands (include/orbx.h: orbx_stereo_frame_view)
extern "C" void *orbx_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void orbx_host_free(void *p) { if (p) (void)hipHostFree(p); }

// the address a kernel reads the image at, or NULL when the memory is ordinary pageable host memory
static const uint8_t *device_visible(const uint8_t *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (older runtimes: an error for unregistered memory)
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged) return (const uint8_t *)a.devicePointer;
    return nullptr;
}

extern "C" int orbx_stereo_frame_view(orbx_extractor_t *h, const uint8_t *left, const uint8_t *right, int w, int hgt, int stride,
                                      float mbf, float mb, orbx_stereo_view_t *view) {
    if (!h || !view) { orbx_set_error("orbx_stereo_frame_view: bad arguments"); return ORBX_ERR_ARG; }
    memset(view, 0, sizeof(*view));
    if (!left || !right || w <= 0 || hgt <= 0) return ORBX_OK;   // empty image (:1046-1047)
    if (stride < w) { orbx_set_error("stride < width"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    int rc = ensure_plan(h, w, hgt, 2);
    if (rc) return rc;
    const int cap = (h->max_kp + 3) & ~3;   // the record's blocks are 16-byte aligned for any multiple of 4
    const size_t recBytes = (size_t)128 * cap + 16;
    if (h->fv_cap < cap) {
        ORBX_HIP(hipStreamSynchronize(h->stream));
        for (int i = 0; i < 2; i++) {
            hipFree(h->fv_d[i]); h->fv_d[i] = nullptr;
            if (h->fv_h[i]) { hipHostFree(h->fv_h[i]); h->fv_h[i] = nullptr; }
        }
        h->fv_cap = 0;
        for (int i = 0; i < 2; i++) {
            ORBX_HIP(hipMalloc(&h->fv_d[i], recBytes));
            ORBX_HIP(hipMemset(h->fv_d[i], 0, recBytes));
            ORBX_HIP(hipHostMalloc((void **)&h->fv_h[i], recBytes, hipHostMallocDefault));
            ORBX_HIP(hipHostGetDevicePointer((void **)&h->fv_hdev[i], h->fv_h[i], 0));
        }
        h->fv_cap = cap;
    }
    const size_t span = (size_t)stride * (hgt - 1) + w, img_bytes = (span + 255) & ~(size_t)255;
    const uint8_t *dl = device_visible(left), *dr = device_visible(right);
    if (!dl || !dr) {   // pageable memory: one memcpy per image into the handle's pinned staging buffer, read from there
        if (h->fv_stage_bytes < 2 * img_bytes) {
            ORBX_HIP(hipStreamSynchronize(h->stream));
            if (h->fv_stage) { hipHostFree(h->fv_stage); h->fv_stage = nullptr; h->fv_stage_bytes = 0; }
            ORBX_HIP(hipHostMalloc((void **)&h->fv_stage, 2 * img_bytes, hipHostMallocDefault));
            ORBX_HIP(hipHostGetDevicePointer((void **)&h->fv_stage_dev, h->fv_stage, 0));
            h->fv_stage_bytes = 2 * img_bytes;
        }
        if (!dl) { memcpy(h->fv_stage, left, span); dl = h->fv_stage_dev; }
        if (!dr) { memcpy(h->fv_stage + img_bytes, right, span); dr = h->fv_stage_dev + img_bytes; }
    }
    const int i = h->fv_next;
    h->fv_next ^= 1;
    uint8_t *rec = h->fv_d[i];
    const size_t c = (size_t)cap;
    hipStream_t st = h->stream;
    // image b of the batch of two sits at dl + b * (dr - dl): the kernels add the (unsigned, possibly wrapped) difference once
    // the descriptor kernel stores every record twice - HBM and the pinned twin - unless an option took the split-call path (its scratch records move later)
    h->descHostDelta = h->opt[15] >= 2 ? 0 : (long long)((intptr_t)h->fv_hdev[i] - (intptr_t)rec);
    const long long twin = h->descHostDelta;
    rc = launch_pipeline(h, dl, 2, w, hgt, stride, (size_t)((uintptr_t)dr - (uintptr_t)dl), (orbx_keypoint_t *)rec, rec + 56 * c, (int32_t *)(rec + 128 * c), cap, st);
    h->descHostDelta = 0;
    if (rc) return rc;
    if (!h->fv_flag) {
        ORBX_HIP(hipHostMalloc((void **)&h->fv_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *h->fv_flag = 0;
        ORBX_HIP(hipHostGetDevicePointer((void **)&h->fv_flag_dev, h->fv_flag, 0));
    }
    int armed = 0;
    const int seq = ++h->fv_seq;
    rc = orbx_internal_stereo_frame_record(h, rec, h->fv_hdev[i], cap, mbf, mb, st, twin != 0, h->opt[27] == 0 ? h->fv_flag_dev : nullptr, seq, &armed);
    if (rc) return rc;
    // The last kernel stores the call's number into the completion word behind the record; the host polls it (a stream synchronisation
    // adds 10-15 us of wake-up latency to a 120-us call) and falls back to the stream after a few milliseconds.  ORBX_OPT_STREAM_SYNC = 1: the stream wait.
    bool done = false;
    if (armed)
        for (int spin = 0; spin < 200000 && !done; spin++) {
            done = __atomic_load_n(h->fv_flag, __ATOMIC_ACQUIRE) == seq;
            if (!done) __builtin_ia32_pause();
        }
    if (!done) ORBX_HIP(hipStreamSynchronize(st));
    const uint8_t *hr = h->fv_h[i];
    const int32_t *tail = (const int32_t *)(hr + 128 * c);
    view->nl = std::min(tail[0], cap); view->nr = std::min(tail[1], cap); view->nmatch = tail[2]; view->cap = cap;
    view->kl = (const orbx_keypoint_t *)hr; view->kr = (const orbx_keypoint_t *)(hr + 28 * c);
    view->dl = hr + 56 * c; view->dr = hr + 88 * c;
    view->uright = (const float *)(hr + 120 * c); view->depth = (const float *)(hr + 124 * c);
    view->d_kl = (const orbx_keypoint_t *)rec; view->d_kr = (const orbx_keypoint_t *)(rec + 28 * c);
    view->d_dl = rec + 56 * c; view->d_dr = rec + 88 * c;
    view->d_uright = (const float *)(rec + 120 * c); view->d_depth = (const float *)(rec + 124 * c);
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_extractor_t *h, const uint8_t *img, int w, int hgt, int stride,
                            orbx_keypoint_t *kps, uint8_t *desc, int cap, int *n_out) {
    if (!h || !n_out) { orbx_set_error("orbx_extract: bad arguments"); return ORBX_ERR_ARG; }
    if (!img || w <= 0 || hgt <= 0) { *n_out = 0; return ORBX_OK; }  // empty image (:1046-1047)
    const uint8_t *imgs[1] = {img};
    return orbx_extract_batch(h, imgs, 1, w, hgt, stride, kps, desc, cap, n_out);
}

int orbx_internal_level(const orbx_extractor *h, int level, int *w, int *hgt, int *pstride,
                        unsigned long long *poff) {
    if (!h || level < 0 || level >= h->nlevels || h->pw == 0) return ORBX_ERR_ARG;
    const LevelGeom &g = h->geom[level];
    *w = g.w; *hgt = g.h; *pstride = g.pstride; *poff = g.poff;
    return ORBX_OK;
}

// copyMakeBorder of the levels >= 1 of the last batch, for callers that look at the padded buffers
static int ensure_frames(orbx_extractor *h) {
    if (h->framesStale <= 0) return ORBX_OK;
    const int nl = h->nlevels, B = h->framesStale;
    hipStream_t st = h->last_stream;
    const LevelGeom &g1 = h->geom[1];
    const int p1 = (g1.w + 2 * ORBX_EDGE + 3) >> 2, tot1 = 2 * ORBX_EDGE * p1 + g1.h * 12;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_pyr_pad<false>, dim3((tot1 + 255) / 256, nl - 1, B), dim3(256), 0, st, (const uint8_t *)nullptr, 0, (size_t)0,
                       h->d_pyr, h->pyrImgBytes, h->d_geom, 1);
    ORBX_HIP(hipGetLastError());
    h->framesStale = 0;
    return ORBX_OK;
}

extern "C" int orbx_pyramid_device(orbx_extractor_t *h, int b, int level, const uint8_t **d_ptr, int *w, int *hgt,
                                   int *stride) {
    if (!h || !d_ptr || level < 0 || level >= h->nlevels || h->pw == 0 || b < 0 || b >= h->pB) {
        orbx_set_error("orbx_pyramid_device: bad arguments or no frame extracted yet");
        return ORBX_ERR_ARG;
    }
    if (level > 0) {   // the caller may walk into the 19-px frame around the ROI, as with the reference's padded cv::Mat
        ORBX_HIP(hipSetDevice(h->device));
        const int rc = ensure_frames(h);
        if (rc) return rc;
    }
    const LevelGeom &g = h->geom[level];
    *d_ptr = h->d_pyr + (size_t)b * h->pyrImgBytes + g.poff + (size_t)ORBX_EDGE * g.pstride + ORBX_EDGE;
    if (w) *w = g.w;
    if (hgt) *hgt = g.h;
    if (stride) *stride = g.pstride;
    return ORBX_OK;
}

extern "C" int orbx_pyramid_host(orbx_extractor_t *h, int b, int level, int padded, uint8_t *dst, int dst_stride,
                                 int *w, int *hgt) {
    if (!h || level < 0 || level >= h->nlevels || h->pw == 0 || b < 0 || b >= h->pB) {
        orbx_set_error("orbx_pyramid_host: bad arguments or no frame extracted yet");
        return ORBX_ERR_ARG;
    }
    const LevelGeom &g = h->geom[level];
    const int ow = padded ? g.w + 2 * ORBX_EDGE : g.w, oh = padded ? g.h + 2 * ORBX_EDGE : g.h;
    if (w) *w = ow;
    if (hgt) *hgt = oh;
    if (!dst) return ORBX_OK;
    if (dst_stride < ow) { orbx_set_error("dst_stride < width"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    if (padded && level > 0) { const int rc = ensure_frames(h); if (rc) return rc; }
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    const uint8_t *src = h->d_pyr + (size_t)b * h->pyrImgBytes + g.poff +
                         (padded ? 0 : (size_t)ORBX_EDGE * g.pstride + ORBX_EDGE);
    // linear device-to-host copy of the row span, rows unpacked on the host (2-D copies of odd widths are very slow)
    const size_t span = (size_t)g.pstride * (oh - 1) + ow;
    static thread_local std::vector<uint8_t> tmp;
    if (tmp.size() < span) tmp.resize(span);
    ORBX_HIP(hipMemcpy(tmp.data(), src, span, hipMemcpyDeviceToHost));
    for (int r = 0; r < oh; r++) memcpy(dst + (size_t)r * dst_stride, tmp.data() + (size_t)r * g.pstride, ow);
    return ORBX_OK;
}

extern "C" int orbx_level_counts(orbx_extractor_t *h, int b, int32_t *candidates, int32_t *keypoints) {
    if (!h || h->pw == 0 || b < 0 || b >= h->pB || !h->last_valid) { orbx_set_error("orbx_level_counts: bad arguments or no frame extracted yet"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    ORBX_HIP(hipStreamSynchronize(h->last_stream));
    if (candidates) ORBX_HIP(hipMemcpy(candidates, h->d_candCnt + (size_t)b * h->nlevels, sizeof(int32_t) * h->nlevels, hipMemcpyDeviceToHost));
    if (keypoints) ORBX_HIP(hipMemcpy(keypoints, h->d_lvlCnt + (size_t)b * h->nlevels, sizeof(int32_t) * h->nlevels, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

#ifdef ORBX_DEVELOPER   // ---- read-only stage hooks: developer build only (include/orbx_dev.h)
// the compacted key arrays of the last call, when k_octree_pyr read the cell lists in place: gathered now, for the test hooks
static int ensure_cand(orbx_extractor *h) {
    if (h->candStale <= 0) return ORBX_OK;
    const int nl = h->nlevels, B = h->candStale;
    CellBases cb;
    for (int l = 0; l <= ORBX_MAX_LEVELS; l++) cb.v[l] = l < nl ? h->geom[l].cellBase : h->totalCells;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gather, dim3((h->totalCells + GATHER_CELLS_PER_BLOCK - 1) / GATHER_CELLS_PER_BLOCK, B), dim3(256), 0, h->last_stream,
                       h->d_geom, nl, h->totalCells, h->d_cellCnt, h->d_cellRaw, h->d_slots, h->slotsPerImg, h->d_cand, h->keysPerImg,
                       h->d_candCnt, h->ini_th, h->min_th, cb, (int32_t *)nullptr, 0, (int32_t *)nullptr, 0);
    ORBX_HIP(hipGetLastError());
    h->candStale = 0;
    return ORBX_OK;
}

// test / probe hook: which (image, level)s of the last call took the exact form of the quad-tree because the count pyramid was too shallow
extern "C" int orbx_debug_octree_fallbacks(orbx_extractor_t *h, int32_t *out, int n) {
    if (!h || !out || n < 1 || h->pw == 0 || n > h->pB * h->nlevels) { orbx_set_error("orbx_debug_octree_fallbacks: bad arguments"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    ORBX_HIP(hipMemcpy(out, h->d_octFallback, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_debug_level_points(orbx_extractor_t *h, int b, int level, int stage, int32_t *out, int cap,
                                       int *n_out) {
    if (!h || !n_out || level < 0 || level >= h->nlevels || h->pw == 0 || b < 0 || b >= h->pB || stage < 0 || stage > 1) {
        orbx_set_error("orbx_debug_level_points: bad arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP(hipSetDevice(h->device));
    if (stage == 0) { const int rc = ensure_cand(h); if (rc) return rc; }
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    const LevelGeom &g = h->geom[level];
    int32_t n = 0;
    const int32_t *cntp = (stage == 0 ? h->d_candCnt : h->d_lvlCnt) + b * h->nlevels + level;
    ORBX_HIP(hipMemcpy(&n, cntp, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n_out = n;
    if (!out || n == 0) return ORBX_OK;
    const int m = std::min(n, cap);
    std::vector<uint32_t> tmp(m);
    const uint32_t *src = stage == 0 ? h->d_cand + (size_t)b * h->keysPerImg + g.keyOff
                                     : h->d_lvlKp + (size_t)b * h->lvlKpCap + g.lvlKpOff;
    ORBX_HIP(hipMemcpy(tmp.data(), src, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; i++) {
        out[3 * i] = (int32_t)(tmp[i] & 0xFFF);
        out[3 * i + 1] = (int32_t)((tmp[i] >> 12) & 0xFFF);
        out[3 * i + 2] = (int32_t)(tmp[i] >> 24);
    }
    return n > cap ? ORBX_ERR_CAPACITY : ORBX_OK;
}

#endif   // ORBX_DEVELOPER

// Which FAST kernel(s) a batch of B images of the planned size runs (the rule of launch_pipeline): for benchmarks that name the
// kernel they time.  *strips = 1 if k_fast_strips takes part, *cells = 1 if k_fast_cells does.
extern "C" int orbx_fast_kernels(const orbx_extractor_t *h, int B, int *strips, int *cells, int *images_per_launch) {
    if (!h || h->pw == 0 || B < 1) { orbx_set_error("orbx_fast_kernels: no plan yet"); return ORBX_ERR_ARG; }
    // the last call's own chunk count when it was a batch of this size (it knows whether its pyramid was built ahead), else the rule
    const int nch = (h->last_valid && h->lastB == B) ? h->lastChunks : chunk_count(h, B, h->profiling == 1, false);
    B = B / nch;                                   // the first chunk is the one whose FAST stage carries the events
    if (images_per_launch) *images_per_launch = B;
    const bool st = h->totalStrips > 0 && (h->opt[6] == 0 ? (size_t)h->totalStrips * B >= 4096 : h->opt[6] == 3);
    const unsigned lv = st ? h->stripLevels : 0u;
    if (strips) *strips = st ? 1 : 0;
    if (cells) *cells = lv != (1u << h->nlevels) - 1u ? 1 : 0;
    return ORBX_OK;
}

#ifdef ORBX_DEVELOPER
// Test hook for SURVEY section 8 row a8 (cv::GaussianBlur 7x7, sigma 2, fused into k_describe and never stored): the next
// single-image orbx_extract calls also write, for keypoint i, the 37x37 blurred block centred on it (the only blurred
// pixels the descriptor can read) to a device buffer, fetched here.  enable = 0 frees the buffer.
extern "C" int orbx_debug_blur_patches(orbx_extractor_t *h, int enable, uint8_t *out, int n) {
    if (!h) return ORBX_ERR_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    if (out && n > 0) {
        if (!h->d_dbgBlur || n > h->dbgBlurCap) { orbx_set_error("orbx_debug_blur_patches: not enabled or n too large"); return ORBX_ERR_ARG; }
        ORBX_HIP(hipMemcpy(out, h->d_dbgBlur, (size_t)n * 37 * 37, hipMemcpyDeviceToHost));
        return ORBX_OK;
    }
    hipFree(h->d_dbgBlur); h->d_dbgBlur = nullptr; h->dbgBlurCap = 0;
    if (enable) {
        h->dbgBlurCap = h->max_kp > 0 ? h->max_kp + 300 : h->nfeatures + 3 * h->nlevels + 300;
        ORBX_HIP(hipMalloc(&h->d_dbgBlur, (size_t)h->dbgBlurCap * 37 * 37));
    }
    return ORBX_OK;
}

// Test hook: level `level` of image b as k_blur_levels left it in the last call (inner ROI), and the mask of levels that were
// blurred as a whole (bit l).  A level outside the mask was blurred per keypoint inside k_describe: ORBX_ERR_ARG for it.
extern "C" int orbx_debug_blurred_level(orbx_extractor_t *h, int b, int level, uint8_t *dst, int dst_stride, unsigned *mask_out) {
    if (!h || h->pw == 0 || b < 0 || b >= h->pB) { orbx_set_error("orbx_debug_blurred_level: bad arguments"); return ORBX_ERR_ARG; }
    if (mask_out) *mask_out = h->blurMaskLast;
    if (!dst) return ORBX_OK;
    if (level < 0 || level >= h->nlevels || !((h->blurMaskLast >> level) & 1u) || !h->d_blur) {
        orbx_set_error("orbx_debug_blurred_level: level %d was not blurred as a whole (mask 0x%x)", level, h->blurMaskLast);
        return ORBX_ERR_ARG;
    }
    const LevelGeom &g = h->geom[level];
    if (dst_stride < g.w) { orbx_set_error("dst_stride < width"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_valid) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    const uint8_t *src = h->d_blur + (size_t)b * h->pyrImgBytes + g.poff + (size_t)ORBX_EDGE * g.pstride + ORBX_EDGE;
    const size_t span = (size_t)g.pstride * (g.h - 1) + g.w;
    std::vector<uint8_t> tmp(span);
    ORBX_HIP(hipMemcpy(tmp.data(), src, span, hipMemcpyDeviceToHost));
    for (int r = 0; r < g.h; r++) memcpy(dst + (size_t)r * dst_stride, tmp.data() + (size_t)r * g.pstride, g.w);
    return ORBX_OK;
}

#endif   // ORBX_DEVELOPER

extern "C" int orbx_set_profiling(orbx_extractor_t *h, int enabled) {
    if (!h) return ORBX_ERR_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    for (int r = 0; r < ORBX_EV_RING; r++)
        if (h->ev_pending[r]) { ORBX_HIP(hipEventSynchronize(h->ev[r][h->ev_pending[r] == 2 ? 2 : 4])); h->ev_pending[r] = 0; }
    if (enabled < 0 || enabled > 3) { orbx_set_error("orbx_set_profiling: mode %d", enabled); return ORBX_ERR_ARG; }
    h->profiling = enabled;
    if (enabled) {   // the ordering kernel in front of the FAST bracket: its first launch loads code, keep that out of the timed calls
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, h->stream);
        ORBX_HIP(hipStreamSynchronize(h->stream));
    }
    h->prof_calls = 0;
    h->ev_head = 0;
    h->acc_n = 0;
    for (int i = 0; i < ORBX_NUM_STAGES; i++) h->acc_ms[i] = 0;
    return ORBX_OK;
}
extern "C" int orbx_get_stage_ms(orbx_extractor_t *h, float *ms, int *ncalls) {
    if (!h || !ms) return ORBX_ERR_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    for (int r = 0; r < ORBX_EV_RING; r++)
        if (h->ev_pending[r]) { int rc = harvest_events(h, r); if (rc) return rc; }
    if (h->acc_n == 0) { orbx_set_error("no profiled batch recorded"); return ORBX_ERR_ARG; }
    for (int i = 0; i < ORBX_NUM_STAGES; i++) ms[i] = (float)(h->acc_ms[i] / (double)h->acc_n);
    if (ncalls) *ncalls = (int)h->acc_n;
    return ORBX_OK;
}
