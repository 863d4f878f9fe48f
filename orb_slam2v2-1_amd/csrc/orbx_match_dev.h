// orbx_match_dev.h — device helpers shared by the matcher kernels (orbx_match.hip: exact
// single-workgroup kernels; orbx_match_fast.hip: parallel candidate search + speculative
// sequential resolution).
#pragma once
#include "orbx_internal.h"
#include <limits.h>

#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30
#define GRID_COLS 64
#define GRID_ROWS 48

typedef unsigned long long u64;

// LDS hand-off inside ONE wave (lanes of a wave run in lock step; the fence orders the LDS accesses)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct Desc256 { u64 w[4]; };
__device__ __forceinline__ Desc256 load_desc(const uint8_t *p) {
    Desc256 d;
    const uint4 a = ((const uint4 *)p)[0], b = ((const uint4 *)p)[1];
    d.w[0] = (u64)a.x | ((u64)a.y << 32); d.w[1] = (u64)a.z | ((u64)a.w << 32);
    d.w[2] = (u64)b.x | ((u64)b.y << 32); d.w[3] = (u64)b.z | ((u64)b.w << 32);
    return d;
}
__device__ __forceinline__ int ham(const Desc256 &a, const Desc256 &b) {
    return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) +
           __popcll(a.w[3] ^ b.w[3]);
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return (u64)lo | ((u64)hi << 32);
}
// Wave64 reductions on the DPP data path (row shifts + row broadcasts, ~7 VALU steps) instead of six
// ds_bpermute round trips through the LDS crossbar: these sit on the critical path of one-wave-per-item kernels.
// After the steps lane 63 holds the result; it is returned to every lane through an SGPR.
#define ORBX_DPP_ROW_SHR(n) (0x110 + (n))
#define ORBX_DPP_ROW_BCAST15 0x142
#define ORBX_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(1), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(2), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(4), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(8), 0xf, 0xf, true);     // lane 15 of every row: row sum
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_BCAST15, 0xa, 0xf, true);   // rows 1, 3 += lane 15 of the row before
    v += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_BCAST31, 0xc, 0xf, true);   // rows 2, 3 += lane 31
    return __builtin_amdgcn_readlane(v, 63);
}
// Wave64 inclusive prefix sum on the DPP data path (row shifts, then the two row broadcasts)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2, 3
    return v;
}
// Rank select in a 256-bin histogram (LDS), run by ONE full wave: the first bin h with sum(hist[0..h]) > target and the
// rank left inside it.  Lane i owns bins 4i..4i+3.  Requires target < sum(hist).
__device__ __forceinline__ void hist256_select(const int *hist, int target, int lane, int *bin, int *rank) {
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const int s = h0 + h1 + h2 + h3, inc = wave_incl_scan_dpp(s);
    const unsigned long long hit = __ballot(inc > target);
    if (hit && lane == __builtin_ctzll(hit)) {
        int acc = inc - s, k = 0;
        if (acc + h0 <= target) { acc += h0; k = 1;
            if (acc + h1 <= target) { acc += h1; k = 2;
                if (acc + h2 <= target) { acc += h2; k = 3; } } }
        *bin = 4 * lane + k;
        *rank = target - acc;
    }
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ u64 dpp_min_step(u64 v) {
    // lanes that the masks disable or that read out of range see ~0 (the identity of min)
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(unsigned)v, CTRL, ROWMASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(unsigned)(v >> 32), CTRL, ROWMASK, 0xf, false);
    const u64 t = (u64)lo | ((u64)hi << 32);
    return t < v ? t : v;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    v = dpp_min_step<ORBX_DPP_ROW_SHR(1), 0xf>(v);
    v = dpp_min_step<ORBX_DPP_ROW_SHR(2), 0xf>(v);
    v = dpp_min_step<ORBX_DPP_ROW_SHR(4), 0xf>(v);
    v = dpp_min_step<ORBX_DPP_ROW_SHR(8), 0xf>(v);
    v = dpp_min_step<ORBX_DPP_ROW_BCAST15, 0xa>(v);
    v = dpp_min_step<ORBX_DPP_ROW_BCAST31, 0xc>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return (u64)lo | ((u64)hi << 32);
}


// Frame::GetFeaturesInArea (src/Frame.cc:342-395) as a predicate + scan-order key, see
// orbx_match.hip "guided searches".
struct AreaQuery { int x0, x1, y0, y1; bool empty; bool checkLevels; int minLevel, maxLevel; float x, y, r; };
__device__ __forceinline__ AreaQuery make_query(const orbm_grid_geom_t &g, float x, float y, float r, int minLevel,
                                                int maxLevel) {
    AreaQuery q;
    q.x = x; q.y = y; q.r = r; q.minLevel = minLevel; q.maxLevel = maxLevel;
    q.empty = false;
    q.x0 = max(0, (int)floorf((x - g.min_x - r) * g.inv_w));
    if (q.x0 >= GRID_COLS) q.empty = true;
    q.x1 = min(GRID_COLS - 1, (int)ceilf((x - g.min_x + r) * g.inv_w));
    if (q.x1 < 0) q.empty = true;
    q.y0 = max(0, (int)floorf((y - g.min_y - r) * g.inv_h));
    if (q.y0 >= GRID_ROWS) q.empty = true;
    q.y1 = min(GRID_ROWS - 1, (int)ceilf((y - g.min_y + r) * g.inv_h));
    if (q.y1 < 0) q.empty = true;
    q.checkLevels = (minLevel > 0) || (maxLevel >= 0);
    return q;
}
// cell code of a keypoint: cellx<<8 | celly, 0xFFFF when PosInGrid fails
__device__ __forceinline__ unsigned cell_code(const orbm_grid_geom_t &g, const orbx_keypoint_t &kp) {
    const int px = (int)roundf((kp.x - g.min_x) * g.inv_w), py = (int)roundf((kp.y - g.min_y) * g.inv_h);
    if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) return 0xFFFFu;
    return (unsigned)(px << 8 | py);
}
__device__ __forceinline__ bool in_area(const AreaQuery &q, unsigned code, const orbx_keypoint_t &kp) {
    if (code == 0xFFFFu) return false;
    const int cx = (int)(code >> 8), cy = (int)(code & 0xFF);
    if (cx < q.x0 || cx > q.x1 || cy < q.y0 || cy > q.y1) return false;
    if (q.checkLevels) {
        if (kp.octave < q.minLevel) return false;
        if (q.maxLevel >= 0 && kp.octave > q.maxLevel) return false;
    }
    const float distx = kp.x - q.x, disty = kp.y - q.y;
    return fabsf(distx) < q.r && fabsf(disty) < q.r;
}
__device__ __forceinline__ bool in_area_xy(const AreaQuery &q, unsigned code, float kx, float ky, int octave) {   // in_area on unpacked fields
    if (code == 0xFFFFu) return false;
    const int cx = (int)(code >> 8), cy = (int)(code & 0xFF);
    if (cx < q.x0 || cx > q.x1 || cy < q.y0 || cy > q.y1) return false;
    if (q.checkLevels) {
        if (octave < q.minLevel) return false;
        if (q.maxLevel >= 0 && octave > q.maxLevel) return false;
    }
    const float distx = kx - q.x, disty = ky - q.y;
    return fabsf(distx) < q.r && fabsf(disty) < q.r;
}
__device__ __forceinline__ u64 scan_key(int dist, unsigned code, int j) {
    return ((u64)dist << 28) | ((u64)(code >> 8) << 22) | ((u64)(code & 0xFF) << 16) | (u64)j;
}

// ORBmatcher::ComputeThreeMaxima (:1603-1644)
__device__ inline void three_maxima(const int *histo, int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

