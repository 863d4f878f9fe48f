/*
 * orb_oracle_extract.c — CPU ORACLE (test infrastructure, never shipped / never on the
 * product path): restatement of ORBextractor (reference: src/ORBextractor.cc) and of the
 * OpenCV 2.4.11/3.2 primitives it calls.  PARITY UNPINNED (see orb_oracle.h).
 *
 * Compile with -ffp-contract=off: the reference's float expressions are evaluated
 * operation by operation (no FMA), and the HIP path is built the same way.
 */
#include "orb_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "orb_oracle_sincosf.h"

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define EDGE_THRESHOLD 19 /* src/ORBextractor.cc:72-74 */
#define MAX_LEVELS 32

static const int8_t k_pattern[1024] = {
#include "../include/orb_pattern_31.inc"
};

struct orb_oracle {
    int nfeatures, nlevels, ini_th, min_th;
    int gauss_flavour;   /* ORACLE_GAUSS_*: column rounding of cv::GaussianBlur (oracle_set_gauss_flavour) */
    int gauss_taps[4];   /* ORACLE_GAUSS_FIXED_TAPS: the build's Q8 taps, centre first (oracle_set_gauss_taps) */
    double scale_factor; /* member is double: include/ORBextractor.h:98 */
    float sf[MAX_LEVELS], isf[MAX_LEVELS], sig2[MAX_LEVELS], isig2[MAX_LEVELS];
    int32_t nfeat[MAX_LEVELS];
    int32_t umax[HALF_PATCH_SIZE + 1];
    /* per-call state */
    uint8_t *pyr[MAX_LEVELS];  /* padded buffers */
    uint8_t *blur[MAX_LEVELS]; /* blurred inner images, stride w */
    int pw[MAX_LEVELS], ph[MAX_LEVELS];
    oracle_cand_t *cand[MAX_LEVELS]; int ncand[MAX_LEVELS];
    oracle_cand_t *sel[MAX_LEVELS];  int nsel[MAX_LEVELS];
    double stage_s[6];
};

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* cvRound: lrint semantics = round half to even (SSE2 cvtsd2si) */
int oracle_cv_round(double v) { return (int)lrint(v); }
static int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static int cv_ceil(double v) { int i = (int)v; return i + (v > i); }

/* ------------------------------------------------------------------ ctor */
orb_oracle_t *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    if (nlevels < 1 || nlevels > MAX_LEVELS || nfeatures < 1) return NULL;
    orb_oracle_t *o = (orb_oracle_t *)calloc(1, sizeof(*o));
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->ini_th = ini_th; o->min_th = min_th;
    o->scale_factor = scale_factor;
    /* src/ORBextractor.cc:415-431 */
    o->sf[0] = 1.0f; o->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        o->sf[i] = (float)(o->sf[i - 1] * o->scale_factor);
        o->sig2[i] = o->sf[i] * o->sf[i];
    }
    for (int i = 0; i < nlevels; i++) {
        o->isf[i] = 1.0f / o->sf[i];
        o->isig2[i] = 1.0f / o->sig2[i];
    }
    /* :435-446 */
    float factor = (float)(1.0f / o->scale_factor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        o->nfeat[level] = oracle_cv_round(nDesired);
        sum += o->nfeat[level];
        nDesired *= factor;
    }
    o->nfeat[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* :454-469 */
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) o->umax[v] = oracle_cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    return o;
}

static void free_frame_state(orb_oracle_t *o) {
    for (int l = 0; l < MAX_LEVELS; l++) {
        free(o->pyr[l]); o->pyr[l] = NULL;
        free(o->blur[l]); o->blur[l] = NULL;
        free(o->cand[l]); o->cand[l] = NULL; o->ncand[l] = 0;
        free(o->sel[l]); o->sel[l] = NULL; o->nsel[l] = 0;
    }
}
int oracle_set_gauss_flavour(orb_oracle_t *o, int flavour) {
    if (!o || (flavour != ORACLE_GAUSS_HALF_UP && flavour != ORACLE_GAUSS_SSE2)) return -1;
    o->gauss_flavour = flavour;
    return 0;
}
int oracle_get_gauss_flavour(const orb_oracle_t *o) { return o->gauss_flavour; }
/* ORACLE_GAUSS_FIXED_TAPS: the bit-exact 8-bit Gaussian of OpenCV >= 3.4.1 on the taps of the build (see below) */
static int gauss_taps_ok(const int k[4]) {
    for (int i = 0; i < 4; i++) if (k[i] < 0 || k[i] > 255) return 0;
    return k[0] >= 1 && k[0] + 2 * (k[1] + k[2] + k[3]) <= 257;
}
int oracle_set_gauss_taps(orb_oracle_t *o, int k0, int k1, int k2, int k3) {
    const int k[4] = {k0, k1, k2, k3};
    if (!o || !gauss_taps_ok(k)) return -1;
    o->gauss_flavour = ORACLE_GAUSS_FIXED_TAPS;
    memcpy(o->gauss_taps, k, sizeof(k));
    return 0;
}
void oracle_destroy(orb_oracle_t *o) { if (o) { free_frame_state(o); free(o); } }

const float *oracle_scale_factors(const orb_oracle_t *o) { return o->sf; }
const float *oracle_inv_scale_factors(const orb_oracle_t *o) { return o->isf; }
const float *oracle_level_sigma2(const orb_oracle_t *o) { return o->sig2; }
const float *oracle_inv_level_sigma2(const orb_oracle_t *o) { return o->isig2; }
const int32_t *oracle_features_per_level(const orb_oracle_t *o) { return o->nfeat; }
const int32_t *oracle_umax(const orb_oracle_t *o) { return o->umax; }
const double *oracle_stage_seconds(const orb_oracle_t *o) { return o->stage_s; }

/* -------------------------------------------------------- cv::fastAtan2 */
/* OpenCV 2.4.11 modules/core/src/mathfuncs.cpp / 3.2 mathfuncs_core.cpp (scalar path) */
float oracle_fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------ cv::FAST */
/* TYPE_9_16 ring, OpenCV fast.cpp makeOffsets order */
static const int ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16> (OpenCV fast_score.cpp), literal k+=2 form */
int oracle_fast_score(const uint8_t *p, int stride, int threshold) {
    enum { K = 8, N = K * 3 + 1 };
    int k, v = p[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - p[ring_dy[k & 15] * stride + ring_dx[k & 15]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

/* segment test of FAST_t<16>: more than K=8 contiguous ring pixels (of 16+9 unrolled)
 * all < v-t or all > v+t */
int oracle_fast_is_corner(const uint8_t *p, int stride, int threshold) {
    int v = p[0], cd = 0, cb = 0;
    /* high-speed rejection of FAST_t<16> (a necessary condition for a 9-arc) */
    {
#define TAB(k) (p[ring_dy[k] * stride + ring_dx[k]] < v - threshold ? 1 : p[ring_dy[k] * stride + ring_dx[k]] > v + threshold ? 2 : 0)
        int d = TAB(0) | TAB(8);
        if (d == 0) return 0;
        d &= TAB(2) | TAB(10); d &= TAB(4) | TAB(12); d &= TAB(6) | TAB(14);
        if (d == 0) return 0;
        d &= TAB(1) | TAB(9); d &= TAB(3) | TAB(11); d &= TAB(5) | TAB(13); d &= TAB(7) | TAB(15);
        if (d == 0) return 0;
#undef TAB
    }
    for (int k = 0; k < 25; k++) {
        int x = p[ring_dy[k & 15] * stride + ring_dx[k & 15]];
        if (x < v - threshold) { if (++cd > 8) return 1; } else cd = 0;
        if (x > v + threshold) { if (++cb > 8) return 1; } else cb = 0;
    }
    return 0;
}

int oracle_fast_detect(const uint8_t *img, int stride, int w, int h, int threshold,
                       oracle_cand_t *out, int cap) {
    /* FAST_t<16>(img, kps, threshold, nonmax=true): scores of rows 3..h-4, cols 3..w-4,
     * everything else 0; keep strict 8-neighbour maxima; emit row-major. */
    if (w < 7 || h < 7) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    int *score = (int *)calloc((size_t)w * h, sizeof(int));
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const uint8_t *p = img + (size_t)i * stride + j;
            if (oracle_fast_is_corner(p, stride, threshold))
                score[i * w + j] = oracle_fast_score(p, stride, threshold);
        }
    int n = 0;
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            int s = score[i * w + j];
            /* only detected corners are NMS candidates (cornerpos list); a non-corner has
             * score 0 and cannot pass the strict comparisons below */
            if (s == 0) continue;
            const int *r0 = score + (i - 1) * w + j, *r1 = score + i * w + j, *r2 = score + (i + 1) * w + j;
            if (s > r1[-1] && s > r1[1] && s > r0[-1] && s > r0[0] && s > r0[1] &&
                s > r2[-1] && s > r2[0] && s > r2[1]) {
                if (n < cap) { out[n].x = j; out[n].y = i; out[n].score = s; }
                n++;
            }
        }
    free(score);
    return n;
}

/* ---------------------------------------------- cv::resize INTER_LINEAR 8UC1 */
/* OpenCV imgproc/src/imgwarp.cpp: resizeGeneric_ with HResizeLinear<uchar,int,short,2048>
 * and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> */
static short sat_short_round(float v) {
    int i = oracle_cv_round(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
void oracle_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride) {
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (xmax > dx) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short_round((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_short_round(fx * 2048);
    }
    int *rows[2];
    rows[0] = (int *)malloc(sizeof(int) * dw);
    rows[1] = (int *)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy0 = cv_floor(fy);
        fy -= sy0;
        short b0 = sat_short_round((1.f - fy) * 2048), b1 = sat_short_round(fy * 2048);
        for (int k = 0; k < 2; k++) {
            int sy = sy0 + k; /* clip(sy0 - ksize2 + 1 + k, 0, ssize.height) */
            sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
            const uint8_t *S = src + (size_t)sy * sstride;
            int *D = rows[k];
            int dx = 0;
            for (; dx < xmax; dx++) {
                int sx = xofs[dx];
                D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
            }
            for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
        }
        uint8_t *d = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            d[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows[0]); free(rows[1]);
}

static int reflect101(int i, int n) {
    /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* copyMakeBorder(..., EDGE_THRESHOLD x4, BORDER_REFLECT_101) into the padded buffer
 * whose inner ROI already holds the level */
static void make_border(uint8_t *padded, int w, int h) {
    const int E = EDGE_THRESHOLD, ps = w + 2 * E;
    for (int y = -E; y < h + E; y++) {
        int sy = reflect101(y, h);
        uint8_t *drow = padded + (size_t)(y + E) * ps;
        const uint8_t *srow = padded + (size_t)(sy + E) * ps + E;
        for (int x = -E; x < w + E; x++) {
            if (y >= 0 && y < h && x >= 0 && x < w) continue;
            drow[x + E] = srow[reflect101(x, w)];
        }
    }
}

/* -------------------------------------- cv::GaussianBlur 7x7 sigma=2, 8U (<=3.3) */
/* createSeparableLinearFilter: 8U smooth symmetric kernels -> 8-bit fixed point
 * (cvRound(k*256)), int32 rows, columns (sum + 2^15) >> 16, saturate. */
static void gauss7_fixed(int k[7]) {
    /* getGaussianKernel(7, 2, CV_32F) */
    float cf[7]; double sum = 0;
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = oracle_cv_round((double)cf[i] * 256.0);
    }
}
/* Column-pass rounding, the one step of this primitive that OpenCV <= 3.3 implements twice (filter.cpp):
 *   ORACLE_GAUSS_HALF_UP (0)  FixedPtCastEx<int, uchar>: (sum + 2^15) >> 16 - the scalar code, used for EVERY column by a build
 *                             without SSE2 (or with NEON, whose SymmColumnVec_32s8u is not covered here);
 *   ORACLE_GAUSS_SSE2    (1)  SymmColumnVec_32s8u on x86 (SSE2 is baseline on x86-64): the columns x < (w & ~3) go through float -
 *                             int32 row sums -> float, times the float kernel k/65536, summed in float (exact below 256: every
 *                             partial sum is a multiple of 2^-16 under 2^8), _mm_cvtps_epi32 = ROUND-HALF-TO-EVEN, signed then
 *                             unsigned saturating packs; the last w % 4 columns fall through to the scalar FixedPtCastEx.
 * The two differ exactly where sum mod 65536 == 32768 and sum >> 16 is even (~1 pixel in 131 072).  Written from memory of the
 * 2.4.11 / 3.2 sources [external, unverified here: parity unpinned]; which one a given reference build follows is what
 * oracle.refvec.compare reports from the per-level blur checksums. */
static inline int gauss_col_sum(const int *tmp, const int k[7], int x, int y, int w, int h) {
    int acc = 0;
    for (int i = -3; i <= 3; i++) acc += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
    return acc;
}
#if defined(__SSE2__)
#include <emmintrin.h>
/* the literal vector path for four columns x .. x+3 of row y: products and sums in float, cvtps (MXCSR default: nearest even) */
static void gauss_col_sse2_4(const int *tmp, const int k[7], int x, int y, int w, int h, uint8_t out[4]) {
    float kf[4];
    for (int i = 0; i <= 3; i++) kf[i] = (float)k[3 + i] * (1.f / 65536.f);   /* _kernel.convertTo(kernel, CV_32F, 1./(1 << bits)) */
    const int *S0 = tmp + (size_t)y * w + x;
    __m128 s = _mm_mul_ps(_mm_cvtepi32_ps(_mm_loadu_si128((const __m128i *)S0)), _mm_set1_ps(kf[0]));   /* delta = 0 */
    for (int i = 1; i <= 3; i++) {
        const int *Sp = tmp + (size_t)reflect101(y + i, h) * w + x, *Sm = tmp + (size_t)reflect101(y - i, h) * w + x;
        const __m128i xs = _mm_add_epi32(_mm_loadu_si128((const __m128i *)Sp), _mm_loadu_si128((const __m128i *)Sm));
        s = _mm_add_ps(s, _mm_mul_ps(_mm_cvtepi32_ps(xs), _mm_set1_ps(kf[i])));
    }
    __m128i r = _mm_cvtps_epi32(s);
    r = _mm_packs_epi32(r, r);
    r = _mm_packus_epi16(r, r);
    const int v = _mm_cvtsi128_si32(r);
    memcpy(out, &v, 4);
}
#endif
/* ORACLE_GAUSS_FIXED_TAPS (2): OpenCV >= 3.4.1 took 8-bit GaussianBlur out of the filter engine: fixed-point rows (ufixedpoint16: pixel x Q8
 * tap, exact - 255 * 257 = 65535 still fits), fixed-point columns (ufixedpoint32: Q8.8 row sum x Q8 tap, exact) and ONE rounding
 * (sum + 2^15) >> 16, the same in its scalar and SIMD code - i.e. the arithmetic of HALF_UP, on the taps that release computes:
 * cvRound(256 g_i) = 18 34 49 55 .. (sum 257, what <= 3.3 used: then FIXED_TAPS == HALF_UP bit for bit), or, in later releases,
 * taps with the rounding error diffused so that they add up to 256.  Which taps a build uses is data (oracle.refvec.fit_gauss_taps
 * recovers them from a blurred level), so they are a parameter.  [external, from memory of smooth.dispatch.cpp: parity unpinned] */
static void gauss_blur_core(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int flavour, const int k[7]);
void oracle_gaussian_blur7_taps(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, const int taps[4]) {
    const int k[7] = {taps[3], taps[2], taps[1], taps[0], taps[1], taps[2], taps[3]};
    gauss_blur_core(src, w, h, sstride, dst, dstride, ORACLE_GAUSS_HALF_UP, k);
}
void oracle_gaussian_blur7_flavour(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int flavour) {
    int k[7];
    gauss7_fixed(k);
    gauss_blur_core(src, w, h, sstride, dst, dstride, flavour, k);
}
static void gauss_blur_core(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int flavour, const int k[7]) {
    int *tmp = (int *)malloc(sizeof(int) * ((size_t)w * h + 4));
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = -3; i <= 3; i++) acc += k[i + 3] * s[reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    const int wv = flavour == ORACLE_GAUSS_SSE2 ? (w & ~3) : 0;   /* columns the vector path takes */
    for (int y = 0; y < h; y++) {
        int x = 0;
        for (; x < wv; x += 4) {
#if defined(__SSE2__)
            gauss_col_sse2_4(tmp, k, x, y, w, h, dst + (size_t)y * dstride + x);
#else
            for (int j = 0; j < 4; j++) {   /* the closed form of the float path: round half to even, saturate */
                const int acc = gauss_col_sum(tmp, k, x + j, y, w, h);
                const int r = (acc + 32767 + ((acc >> 16) & 1)) >> 16;
                dst[(size_t)y * dstride + x + j] = (uint8_t)(r > 255 ? 255 : r);
            }
#endif
        }
        for (; x < w; x++) {
            int acc = gauss_col_sum(tmp, k, x, y, w, h);
            acc = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(acc > 255 ? 255 : acc < 0 ? 0 : acc);
        }
    }
    free(tmp);
}
void oracle_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    oracle_gaussian_blur7_flavour(src, w, h, sstride, dst, dstride, ORACLE_GAUSS_HALF_UP);
}
/* the integer closed form of the SSE2 column rounding (what the HIP kernels compute); tests compare it with the literal path */
int oracle_gauss_round_half_even(int sum) { const int r = (sum + 32767 + ((sum >> 16) & 1)) >> 16; return r > 255 ? 255 : r; }
int oracle_gauss_round_sse2_literal(int s0, int s1, int s2, int s3, int s4, int s5, int s6) {
    /* seven int32 row sums of one column -> the byte the vector path stores (same arithmetic as gauss_col_sse2_4) */
#if defined(__SSE2__)
    int k[7];
    gauss7_fixed(k);
    int col[7 * 4] = {0};
    const int sv[7] = {s0, s1, s2, s3, s4, s5, s6};
    for (int i = 0; i < 7; i++) col[i * 4] = sv[i];
    uint8_t out[4];
    gauss_col_sse2_4(col, k, 0, 3, 4, 7, out);
    return out[0];
#else
    (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5; (void)s6;
    return -1;
#endif
}

/* ------------------------------------------------------------ quad-tree */
/* DistributeOctTree / ExtractorNode::DivideNode (src/ORBextractor.cc:481-763).
 * The std::list is restated as a doubly linked list; keys are indices into cands.
 * Tie-break of the (size, pointer) sort (:684): the reference's order of equal-sized
 * nodes depends on heap addresses; this build FIXES it to "later-created node sorts
 * higher" (i.e. as if node addresses grew monotonically with creation). */
typedef struct qnode {
    int x0, x1, y0, y1; /* UL.x, UR.x, UL.y, BL.y */
    int *keys; int nkeys;
    int no_more;
    long seq; /* creation order */
    struct qnode *prev, *next;
} qnode_t;
typedef struct { qnode_t *head, *tail; int size; long seq; } qlist_t;

static qnode_t *qnode_new(qlist_t *L, int cap) {
    qnode_t *n = (qnode_t *)calloc(1, sizeof(qnode_t));
    n->keys = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    n->seq = L->seq++;
    return n;
}
static void qlist_push_front(qlist_t *L, qnode_t *n) {
    n->prev = NULL; n->next = L->head;
    if (L->head) L->head->prev = n; else L->tail = n;
    L->head = n; L->size++;
}
static void qlist_push_back(qlist_t *L, qnode_t *n) {
    n->next = NULL; n->prev = L->tail;
    if (L->tail) L->tail->next = n; else L->head = n;
    L->tail = n; L->size++;
}
static qnode_t *qlist_erase(qlist_t *L, qnode_t *n) {
    qnode_t *nx = n->next;
    if (n->prev) n->prev->next = n->next; else L->head = n->next;
    if (n->next) n->next->prev = n->prev; else L->tail = n->prev;
    L->size--;
    free(n->keys); free(n);
    return nx;
}
static void divide_node(qlist_t *L, const qnode_t *p, const oracle_cand_t *c, qnode_t *ch[4]) {
    const int halfX = (int)ceilf((float)(p->x1 - p->x0) / 2);
    const int halfY = (int)ceilf((float)(p->y1 - p->y0) / 2);
    const int mx = p->x0 + halfX, my = p->y0 + halfY;
    for (int i = 0; i < 4; i++) ch[i] = qnode_new(L, p->nkeys);
    ch[0]->x0 = p->x0; ch[0]->x1 = mx;    ch[0]->y0 = p->y0; ch[0]->y1 = my;
    ch[1]->x0 = mx;    ch[1]->x1 = p->x1; ch[1]->y0 = p->y0; ch[1]->y1 = my;
    ch[2]->x0 = p->x0; ch[2]->x1 = mx;    ch[2]->y0 = my;    ch[2]->y1 = p->y1;
    ch[3]->x0 = mx;    ch[3]->x1 = p->x1; ch[3]->y0 = my;    ch[3]->y1 = p->y1;
    for (int i = 0; i < p->nkeys; i++) {
        const oracle_cand_t *kp = &c[p->keys[i]];
        int q;
        if ((float)kp->x < (float)mx) q = ((float)kp->y < (float)my) ? 0 : 2;
        else q = ((float)kp->y < (float)my) ? 1 : 3;
        ch[q]->keys[ch[q]->nkeys++] = p->keys[i];
    }
    for (int i = 0; i < 4; i++) if (ch[i]->nkeys == 1) ch[i]->no_more = 1;
}
typedef struct { int size; long seq; qnode_t *node; } qsize_t;
static int qsize_cmp(const void *a, const void *b) {
    const qsize_t *x = (const qsize_t *)a, *y = (const qsize_t *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : x->seq > y->seq ? 1 : 0;
}
/* push the non-empty children to the front (n1..n4 order), record expandable ones */
static void add_children(qlist_t *L, qnode_t *ch[4], qsize_t *vs, int *nvs, int *n_to_expand) {
    for (int i = 0; i < 4; i++) {
        if (ch[i]->nkeys > 0) {
            qlist_push_front(L, ch[i]);
            if (ch[i]->nkeys > 1) {
                if (n_to_expand) (*n_to_expand)++;
                vs[*nvs].size = ch[i]->nkeys; vs[*nvs].seq = ch[i]->seq; vs[*nvs].node = ch[i];
                (*nvs)++;
            }
        } else { free(ch[i]->keys); free(ch[i]); }
    }
}

int oracle_distribute_octtree(const oracle_cand_t *c, int n, int width, int height, int N,
                              oracle_cand_t *out, int cap) {
    const int nIni = (int)roundf((float)width / height); /* :543 */
    if (nIni < 1) return -1; /* reference: division by zero / OOB (UB) */
    const float hX = (float)width / nIni;
    qlist_t L = {0};
    qnode_t **ini = (qnode_t **)malloc(sizeof(qnode_t *) * nIni);
    for (int i = 0; i < nIni; i++) {
        qnode_t *ni = qnode_new(&L, n);
        ni->x0 = (int)(hX * (float)i);
        ni->x1 = (int)(hX * (float)(i + 1));
        ni->y0 = 0; ni->y1 = height;
        qlist_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) {
        size_t r = (size_t)((float)c[i].x / hX); /* :569 */
        if (r >= (size_t)nIni) r = nIni - 1;     /* cannot happen for x < width */
        ini[r]->keys[ini[r]->nkeys++] = i;
    }
    free(ini);
    for (qnode_t *it = L.head; it;) {
        if (it->nkeys == 1) { it->no_more = 1; it = it->next; }
        else if (it->nkeys == 0) it = qlist_erase(&L, it);
        else it = it->next;
    }
    int finish = 0;
    qsize_t *vs = (qsize_t *)malloc(sizeof(qsize_t) * (size_t)(4 * (n + nIni) + 16));
    qsize_t *vprev = (qsize_t *)malloc(sizeof(qsize_t) * (size_t)(4 * (n + nIni) + 16));
    int nvs = 0;
    while (!finish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvs = 0;
        for (qnode_t *it = L.head; it;) {
            if (it->no_more) { it = it->next; continue; }
            qnode_t *ch[4];
            divide_node(&L, it, c, ch);
            add_children(&L, ch, vs, &nvs, &nToExpand);
            it = qlist_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                int nprev = nvs;
                memcpy(vprev, vs, sizeof(qsize_t) * nvs);
                nvs = 0;
                qsort(vprev, nprev, sizeof(qsize_t), qsize_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    qnode_t *ch[4];
                    divide_node(&L, vprev[j].node, c, ch);
                    add_children(&L, ch, vs, &nvs, NULL);
                    qlist_erase(&L, vprev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
    free(vs); free(vprev);
    /* retain the best point in each node (:744-760), first max wins */
    int m = 0;
    for (qnode_t *it = L.head; it; it = it->next) {
        int best = it->keys[0];
        for (int k = 1; k < it->nkeys; k++)
            if ((float)c[it->keys[k]].score > (float)c[best].score) best = it->keys[k];
        if (m < cap) out[m] = c[best];
        m++;
    }
    while (L.head) qlist_erase(&L, L.head);
    return m;
}

/* -------------------------------------------------------- IC_Angle (:77-104) */
static float ic_angle(const uint8_t *center, int step, const int32_t *umax) {
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return oracle_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------- computeOrbDescriptor (:108-147) */
/* `cos(angle)` / `sin(angle)` with float angle under `using namespace std;` (:67) are the float overloads = libm's
 * cosf / sinf.  Mode 0 (default): the restatement of glibc's algorithm (orb_oracle_sincosf.h, platform independent);
 * mode 1: this host's libm cosf / sinf; mode 2: the double functions rounded to float (what round 1 assumed) - the
 * last two exist so that tests can count what the choice changes. */
static int g_sincos_mode = 0;
void oracle_set_sincos_mode(int mode) { g_sincos_mode = mode; }
void oracle_sincosf(float angle, float *s, float *c) {
    if (g_sincos_mode == 1) { *c = cosf(angle); *s = sinf(angle); }
    else if (g_sincos_mode == 2) { *c = (float)cos((double)angle); *s = (float)sin((double)angle); }
    else { *c = rs_cosf(angle); *s = rs_sinf(angle); }
}
void oracle_sincosf_array(const float *a, int n, float *s, float *c) { for (int i = 0; i < n; i++) oracle_sincosf(a[i], &s[i], &c[i]); }
static void orb_descriptor(float kp_angle, const uint8_t *center, int step, uint8_t *desc) {
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = kp_angle * factorPI;
    float a, b;
    oracle_sincosf(angle, &b, &a);
    const int8_t *pat = k_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t *q = pat + 4 * k;
            int t0 = center[oracle_cv_round(q[0] * b + q[1] * a) * step + oracle_cv_round(q[0] * a - q[1] * b)];
            int t1 = center[oracle_cv_round(q[2] * b + q[3] * a) * step + oracle_cv_round(q[2] * a - q[3] * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* -------------------------------------------------------------- extract */
int oracle_extract(orb_oracle_t *o, const uint8_t *img, int w, int h, int stride,
                   oracle_kp_t *kps, uint8_t *desc, int cap) {
    if (!o || !img || w <= 0 || h <= 0) return -1000;
    const int E = EDGE_THRESHOLD;
    free_frame_state(o);
    memset(o->stage_s, 0, sizeof(o->stage_s));
    double t0 = now_s();
    /* ComputePyramid (:1107-1132) */
    for (int l = 0; l < o->nlevels; l++) {
        float scale = o->isf[l];
        int lw = oracle_cv_round((float)w * scale), lh = oracle_cv_round((float)h * scale);
        /* the reference divides by nCols = (w-32)/30 and nRows = (h-32)/30 (:784-787): a level
         * with either below 1 is undefined behaviour there, an argument error here */
        if (lw - 2 * (E - 3) < 30 || lh - 2 * (E - 3) < 30) return -1000;
        o->pw[l] = lw; o->ph[l] = lh;
        int ps = lw + 2 * E;
        o->pyr[l] = (uint8_t *)malloc((size_t)ps * (lh + 2 * E));
        uint8_t *inner = o->pyr[l] + (size_t)E * ps + E;
        if (l == 0) {
            for (int y = 0; y < lh; y++) memcpy(inner + (size_t)y * ps, img + (size_t)y * stride, lw);
        } else {
            int pps = o->pw[l - 1] + 2 * E;
            oracle_resize_linear(o->pyr[l - 1] + (size_t)E * pps + E, o->pw[l - 1], o->ph[l - 1], pps,
                                 inner, lw, lh, ps);
        }
        make_border(o->pyr[l], lw, lh);
    }
    double t1 = now_s(); o->stage_s[0] = t1 - t0;

    /* ComputeKeyPointsOctTree (:765-853) */
    const float W = 30;
    for (int l = 0; l < o->nlevels; l++) {
        double ta = now_s();
        const int lw = o->pw[l], lh = o->ph[l], ps = lw + 2 * E;
        const uint8_t *inner = o->pyr[l] + (size_t)E * ps + E;
        const int minBorderX = E - 3, minBorderY = minBorderX;
        const int maxBorderX = lw - E + 3, maxBorderY = lh - E + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
        int capc = 1024, nc = 0;
        oracle_cand_t *cd = (oracle_cand_t *)malloc(sizeof(oracle_cand_t) * capc);
        oracle_cand_t cell[4096];
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const uint8_t *sub = inner + (size_t)(int)iniY * ps + (int)iniX;
                int sw = (int)maxX - (int)iniX, sh = (int)maxY - (int)iniY;
                int nk = oracle_fast_detect(sub, ps, sw, sh, o->ini_th, cell, 4096);
                if (nk == 0) nk = oracle_fast_detect(sub, ps, sw, sh, o->min_th, cell, 4096);
                if (nk > 4096) nk = 4096;
                for (int k = 0; k < nk; k++) {
                    if (nc == capc) { capc *= 2; cd = (oracle_cand_t *)realloc(cd, sizeof(oracle_cand_t) * capc); }
                    cd[nc].x = cell[k].x + j * wCell;
                    cd[nc].y = cell[k].y + i * hCell;
                    cd[nc].score = cell[k].score;
                    nc++;
                }
            }
        }
        o->cand[l] = cd; o->ncand[l] = nc;
        double tb = now_s(); o->stage_s[1] += tb - ta;
        /* the list holds at most max(N + 3, 4 * nIni) nodes: the first pass splits EVERY root before N is looked at (:606-672) */
        const int nIniL = (int)roundf((float)(maxBorderX - minBorderX) / (float)(maxBorderY - minBorderY));
        int capk = o->nfeat[l] + 8;
        if (capk < 4 * nIniL + 8) capk = 4 * nIniL + 8;
        o->sel[l] = (oracle_cand_t *)malloc(sizeof(oracle_cand_t) * capk);
        int ns = oracle_distribute_octtree(cd, nc, maxBorderX - minBorderX, maxBorderY - minBorderY,
                                           o->nfeat[l], o->sel[l], capk);
        if (ns < 0) return -1000;
        if (ns > capk) return -1000; /* cannot happen: see capk */
        o->nsel[l] = ns;
        o->stage_s[2] += now_s() - tb;
    }
    int total = 0;
    for (int l = 0; l < o->nlevels; l++) total += o->nsel[l];
    if (total > cap) return -total;

    /* orientation (:851-852), blur + descriptors (:1076-1104) */
    int off = 0;
    for (int l = 0; l < o->nlevels; l++) {
        const int lw = o->pw[l], lh = o->ph[l], ps = lw + 2 * E;
        const uint8_t *inner = o->pyr[l] + (size_t)E * ps + E;
        const int scaledPatchSize = (int)(PATCH_SIZE * o->sf[l]);
        double ta = now_s();
        for (int k = 0; k < o->nsel[l]; k++) {
            oracle_kp_t *kp = &kps[off + k];
            kp->x = (float)(o->sel[l][k].x + (E - 3));
            kp->y = (float)(o->sel[l][k].y + (E - 3));
            kp->response = (float)o->sel[l][k].score;
            kp->octave = l; kp->class_id = -1;
            kp->size = (float)scaledPatchSize;
            kp->angle = ic_angle(inner + (size_t)oracle_cv_round(kp->y) * ps + oracle_cv_round(kp->x), ps, o->umax);
        }
        double tb = now_s(); o->stage_s[3] += tb - ta;
        if (o->nsel[l] == 0) continue;
        o->blur[l] = (uint8_t *)malloc((size_t)lw * lh);
        if (o->gauss_flavour == ORACLE_GAUSS_FIXED_TAPS) oracle_gaussian_blur7_taps(inner, lw, lh, ps, o->blur[l], lw, o->gauss_taps);
        else oracle_gaussian_blur7_flavour(inner, lw, lh, ps, o->blur[l], lw, o->gauss_flavour);
        double tc = now_s(); o->stage_s[4] += tc - tb;
        for (int k = 0; k < o->nsel[l]; k++) {
            oracle_kp_t *kp = &kps[off + k];
            orb_descriptor(kp->angle, o->blur[l] + (size_t)oracle_cv_round(kp->y) * lw + oracle_cv_round(kp->x),
                           lw, desc + (size_t)(off + k) * 32);
        }
        if (l != 0) {
            float scale = o->sf[l];
            for (int k = 0; k < o->nsel[l]; k++) { kps[off + k].x *= scale; kps[off + k].y *= scale; }
        }
        o->stage_s[5] += now_s() - tc;
        off += o->nsel[l];
    }
    return total;
}

int oracle_pyramid_level(const orb_oracle_t *o, int level, const uint8_t **padded, int *w, int *h,
                         int *padded_stride) {
    if (level < 0 || level >= o->nlevels || !o->pyr[level]) return -1;
    *padded = o->pyr[level]; *w = o->pw[level]; *h = o->ph[level];
    *padded_stride = o->pw[level] + 2 * EDGE_THRESHOLD;
    return 0;
}
int oracle_level_candidates(const orb_oracle_t *o, int level, const oracle_cand_t **c) {
    *c = o->cand[level]; return o->ncand[level];
}
int oracle_level_keypoints(const orb_oracle_t *o, int level, const oracle_cand_t **c) {
    *c = o->sel[level]; return o->nsel[level];
}
const uint8_t *oracle_blurred_level(const orb_oracle_t *o, int level) { return o->blur[level]; }
