// orbx_describe.hip — IC_Angle + GaussianBlur + steered BRIEF (src/ORBextractor.cc:77-147, :1085-1090): k_describe
// (part of the ORB extractor, see orbx_extract.hip for the pipeline and the C ABI)
#include "orbx_extract_dev.h"
#include <math.h>
#include <float.h>
// ------------------------------------------------------------------------------------
// constant tables
// The other two tables of the path live in the code that uses them: umax (IC_Angle disc, :454-469) inside make_ic_tab,
// the 7-tap sigma=2 Gaussian in 8-bit fixed point {18, 34, 49, 55, 49, 34, 18} (cvRound(k*256), sum 257) as the packed
// weights of k_describe's two blur passes.

// ------------------------------------------------------------------------------------
// K4: one wave per kept keypoint: IC_Angle (:77-104) on the un-blurred level, 7x7 sigma=2
// Gaussian (8-bit fixed point [18 34 49 55 49 34 18], (sum+2^15)>>16) of the 37x37
// neighbourhood the 256 rotated test pairs can touch (|tap| <= 18), then the steered BRIEF
// bits (:108-147) packed with one ballot per 64 pairs.  The blurred level is never written
// to memory: blur is a pure function of the 43x43 source patch, which is staged in LDS from
// the padded (BORDER_REFLECT_101) level, so border handling is identical to cv::GaussianBlur.
// LDS traffic is kept to wide accesses (sub-dword LDS reads were the bottleneck of the first
// version): the horizontal pass reads one b128 per 4 outputs and uses v_dot4_u32_u8 on
// byte-aligned windows; the vertical pass slides a 7-row register window down a column pair.
#define PR 21                    // source patch radius = 18 + 3
#define PROWS (2 * PR + 1)       // 43
#define PSTRIDE 48               // 12 dwords per patch row (16-byte aligned rows)
#define PPAD 16                  // slack behind the patch: the last row's b128 read may run over
#define TROWS PROWS
#define TCOLS (2 * ORBX_DESC_R + 1)  // 37
#define TGROUPS 10               // horizontal pass: 10 groups of 4 outputs per row (cols 0..39)
#define BSTRIDE 40
// The horizontal pass leaves its u16 sums TRANSPOSED: column c of the block is a run of TCS bytes, row r at byte 2r, so that a
// dword holds the rows (2p, 2p+1) of one column - the pair one v_dot2_u32_u16 of the vertical pass multiplies with two taps (the
// first version kept rows of 40 columns and re-paired consecutive rows with 46 v_perm per lane).  The four columns of a group are
// contiguous (22 dwords each); the groups start at c_reach.gbase - a few dwords apart, chosen with the lanes' order in the vertical
// pass by a bank model (tools/desc_lds_layout.py) so that the 16-bit stores of the horizontal pass and the dword reads of the
// vertical pass spread over the 32 banks: modelled 300 -> 216 LDS cycles per keypoint for the two passes.
#define TCS 88
#define TT_BYTES 3680
// patch | pad | transposed horizontal-pass intermediate.  The blurred 37x40 block OVERLAYS the patch, which nobody reads after the
// horizontal pass.  (The gather form of a level blurred as a whole keeps its 37 rows of 40 bytes where the intermediate would be.)
#define DESC_LDS_PER_WAVE (PROWS * PSTRIDE + PPAD + TT_BYTES)
static_assert(DESC_WAVES * DESC_LDS_PER_WAVE <= 160 * 1024 / 7, "seven workgroups per CU");
static_assert(TT_BYTES >= TCOLS * BSTRIDE + 4, "gather form: the blurred block sits where the intermediate would be");

// Which pixels of the 37x37 blurred block a descriptor can read at all.  A tap is (cvRound(x*b + y*a), cvRound(x*a - y*b)) of a
// pattern point with x^2 + y^2 <= 338 (src/ORBextractor.cc:119-120, 150-408): the rotated point lies on a circle of radius
// <= 18.39, so the rounded tap (i, j) satisfies (|i| - 1/2)^2 + (|j| - 1/2)^2 <= 338 (+ 2 of slack for the float arithmetic):
// a disc of 1133 of the 1369 pixels.  Both blur passes skip what lies outside it: c_reach.hh[c] = largest |row offset| column
// c - 18 needs, the item list of the horizontal pass (370 of 430 (row, column group) items: six rounds of the wave instead of
// seven) and the (column pair, first row) of every lane of the vertical pass (runs of 12 rows instead of 3 x 13 for every pair).
// Everything outside the disc holds whatever the skipped work would have overwritten: no tap reads it.
#define DESC_H_ITERS 6
#define DESC_V_ROWS 12
// hitem = byte offset of the item's 16 source bytes in the patch | byte offset of its first output in the transposed intermediate << 16
// vlane = column pair | first row << 8 | byte offset of the pair's first column at that row << 16
struct DescReach { uint8_t hh[40]; uint32_t hitem[DESC_H_ITERS * 64]; uint32_t vlane[64]; int nh, nv, cover; };
constexpr int kDescGbase[TGROUPS] = {0, 89, 181, 275, 369, 461, 552, 643, 734, 828};                 // dword offset of every group of four columns (tools/desc_lds_layout.py)
constexpr signed char kDescVlane[64][2] = {{7, 26}, {12, 12}, {12, 26}, {6, 12}, {5, 0}, {7, 12}, {2, 26}, {11, 26}, {7, 0}, {3, 14}, {7, 24}, {8, 0}, {9, 12}, {10, 24}, {8, 12}, {16, 4}, {10, 12}, {13, 12}, {18, 24}, {8, 26}, {5, 24}, {8, 24}, {13, 0}, {5, 12}, {15, 4}, {18, 24}, {9, 26}, {9, 0}, {11, 12}, {9, 24}, {6, 26}, {12, 24}, {3, 26}, {18, 12}, {18, 24}, {0, 10}, {15, 16}, {4, 26}, {10, 0}, {16, 16}, {4, 2}, {12, 0}, {2, 16}, {6, 24}, {4, 14}, {1, 18}, {14, 14}, {6, 0}, {14, 2}, {14, 26}, {0, 22}, {13, 24}, {16, 26}, {11, 24}, {3, 2}, {10, 26}, {11, 0}, {17, 20}, {18, 24}, {15, 26}, {18, 24}, {1, 6}, {2, 4}, {17, 8}};   // (column pair, first row) of every lane of the vertical pass (same tool)
static_assert(4 * (kDescGbase[TGROUPS - 1] + 4 * (TCS / 4)) <= TT_BYTES, "transposed intermediate");
constexpr DescReach make_desc_reach() {
    DescReach t{};
    for (int c = 0; c < 40; c++) {
        int best = -1;
        const int ax = c < TCOLS ? (c > ORBX_DESC_R ? c - ORBX_DESC_R : ORBX_DESC_R - c) : 99;
        for (int d = 0; d <= ORBX_DESC_R; d++) {
            const int tx = ax > 0 ? 2 * ax - 1 : 0, ty = d > 0 ? 2 * d - 1 : 0;
            if (tx * tx + ty * ty <= 4 * (338 + 2)) best = d;
        }
        t.hh[c] = (uint8_t)(best < 0 ? 0 : best);
        if (c >= TCOLS) t.hh[c] = 0;
    }
    int n = 0;
    for (int r = 0; r < PROWS; r++) {   // intermediate row r (row offset r - 21) is needed by column c iff |r - 21| <= hh[c] + 3
        int g0 = 99, g1 = -1;
        for (int c = 0; c < TCOLS; c++) {
            const int dy = r > PR ? r - PR : PR - r;
            if (dy <= t.hh[c] + 3) { if ((c >> 2) < g0) g0 = c >> 2; if ((c >> 2) > g1) g1 = c >> 2; }
        }
        for (int g = g0; g <= g1; g++) t.hitem[n++] = (uint32_t)(r * PSTRIDE + 4 * g) | ((uint32_t)(4 * kDescGbase[g] + 2 * r) << 16);
    }
    t.nh = n;
    for (int i = n; i < DESC_H_ITERS * 64; i++) t.hitem[i] = t.hitem[n - 1];   // spare slots redo the last item (same values)
    // the vertical pass: lane -> (column pair, run of DESC_V_ROWS rows from an even row), in the order the bank model chose; `cover`
    // counts the pixels of the disc that some run computes (checked below: all 1133 - a column pair needs rows 18 - h .. 18 + h, h = the
    // taller of its two columns)
    int cover = 0;
    for (int cp = 0; cp < 19; cp++) {
        const int h = t.hh[2 * cp] > t.hh[2 * cp + 1] ? t.hh[2 * cp] : t.hh[2 * cp + 1];
        for (int r = ORBX_DESC_R - h; r <= ORBX_DESC_R + h; r++) {
            bool hit = false;
            for (int i = 0; i < 64; i++) hit = hit || (kDescVlane[i][0] == cp && r >= kDescVlane[i][1] && r < kDescVlane[i][1] + DESC_V_ROWS);
            cover += hit ? 1 : 0;
        }
        cover -= 2 * h + 1;   // (0 when every row is covered)
    }
    t.cover = cover;
    int m = 0;
    for (int i = 0; i < 64; i++) {
        const int cp = kDescVlane[i][0], r0 = kDescVlane[i][1];
        if (cp < 0 || cp > 18 || r0 < 0 || (r0 & 1) || r0 + DESC_V_ROWS > TCOLS + 1) { m = 999; break; }
        const int c0 = 2 * cp;
        t.vlane[i] = (uint32_t)cp | ((uint32_t)r0 << 8) | ((uint32_t)(4 * (kDescGbase[c0 >> 2] + (c0 & 3) * (TCS / 4)) + 2 * r0) << 16);
        m++;
    }
    t.nv = m;
    return t;
}
__constant__ const DescReach c_reach = make_desc_reach();
static_assert(make_desc_reach().nh <= DESC_H_ITERS * 64 && make_desc_reach().nh > (DESC_H_ITERS - 1) * 64, "horizontal pass: rounds of the wave");
static_assert(make_desc_reach().nv == 64 && make_desc_reach().cover == 0, "vertical pass: the runs of the 64 lanes cover the disc");
// the rBRIEF pattern as floats (the rotation runs in float: src/ORBextractor.cc:119-120): no integer -> float conversions per tap
__constant__ float c_patternf[1024] = {
#include "../../include/orb_pattern_31.inc"
};
// tools/phase_count.py: with -DORBX_PHASE_MARKERS the kernel carries assembler comments at its phase boundaries, and the per-phase
// instruction mix is counted from the disassembly (the hot path is straight-line: every loop is unrolled).  Never in the product
// build: a volatile asm statement is a scheduling barrier.
#ifdef ORBX_PHASE_MARKERS
#define PHASE(name) asm volatile("; ORBX_PHASE " name)
#else
#define PHASE(name)
#endif

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2 of OpenCV 2.4.11 / 3.2 (scalar path)
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
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

// cos(angle), sin(angle) of src/ORBextractor.cc:113.  `angle` is a float and the file says `using namespace std;` (:67), so
// these are std::cos(float) / std::sin(float) = the C library's cosf / sinf, not the double functions rounded to float (the
// two differ for ~1 % of angles).  Restated: glibc >= 2.28 sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c,
// sincosf.h, sincosf_data.c) for |y| < 120: the argument in double, n = round(y * 2/pi) from a scaled double -> int32
// truncation, x = y - n * pi/2, a degree-7 sine / degree-8 cosine polynomial in double evaluated in glibc's operation
// order (no FMA contraction: the build uses -ffp-contract=off), one rounding to float.  The second coefficient table of
// glibc (quadrants 2, 3) is the first with the cosine polynomial negated, i.e. the negated result.  The oracle's copy of
// this algorithm equals the host libm for every float in [0, 6.2832] (oracle/orb_oracle_sincosf.h).
__device__ __forceinline__ void sincosf_glibc(float y, float &sn, float &cs) {
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;          // abstop12
    if (top < ((0x39800000u >> 20) & 0x7ffu)) { sn = y; cs = 1.0f; return; }   // |y| < 2^-12
    double x = (double)y;
    int n = 0;
    if (top >= ((0x3f490fdbu >> 20) & 0x7ffu)) {                       // |y| >= pi/4 (abstop12 of 0x1.921FB6p-1f): reduce_fast
        const double r = x * 0x1.45F306DC9C883p+23;                    // 2/pi * 2^24
        n = ((int)r + 0x800000) >> 24;
        x = x - (double)n * 0x1.921FB54442D18p0;
    }
    const double sgn = ((n + 1) & 2) ? -1.0 : 1.0;                     // sign[n & 3] = {1, -1, -1, 1}
    const double xs = x * sgn, x2 = x * x;
    // sine polynomial (n even in sinf_poly)
    const double x3 = xs * x2, s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13, x7 = x3 * x2,
                 s = xs + x3 * -0x1.555545995a603p-3;
    const float SP = (float)(s + x7 * s1);
    // cosine polynomial (n odd), first table
    const double x4 = x2 * x2, c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16, c1 = 0x1p0 + x2 * -0x1.ffffffd0c621cp-2,
                 x6 = x4 * x2, c = c1 + x4 * 0x1.55553e1068f19p-5;
    float CP = (float)(c + x6 * c2);
    if (n & 2) CP = -CP;
    sn = (n & 1) ? CP : SP;
    cs = (n & 1) ? SP : CP;
}

// IC_Angle weights per lane (lane = 2 * (v + 15) + half: row v of the radius-15 disc, u = -15..0 or u = 1..16), as bytes for
// v_dot4_u32_u8: m = 1 inside the disc (|u| <= umax[|v|], u <= 15), w = |u| inside.  Built at compile time.
struct IcTab { uint32_t m[64][4], w[64][4]; };
constexpr IcTab make_ic_tab() {
    IcTab t{};
    constexpr int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int lane = 0; lane < 62; lane++) {
        const int v = (lane >> 1) - 15, half = lane & 1, d = um[v < 0 ? -v : v];
        for (int k = 0; k < 16; k++) {
            const int u = half ? k + 1 : k - 15, au = u < 0 ? -u : u;
            const bool in = au <= d && u <= 15;
            t.m[lane][k >> 2] |= (in ? 1u : 0u) << (8 * (k & 3));
            t.w[lane][k >> 2] |= (in ? (uint32_t)au : 0u) << (8 * (k & 3));
        }
    }
    return t;
}
__constant__ const IcTab c_ic = make_ic_tab();


// ------------------------------------------------------------------------------------
// K1b: GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of whole levels (src/ORBextractor.cc:1085-1086), for the levels whose
// keypoints' 37x37 blocks would add up to more pixels than the level has (k_describe then only gathers).  Same fixed-point
// arithmetic as the fused form: rows -> sum of taps [18 34 49 55 49 34 18] (<= 65535; the flavour's own taps for ORBX_GAUSS_FIXED_TAPS),
// columns -> (sum + 2^15) >> 16, saturated.
// One wave = a tile of 64 dwords (256 columns) x BLUR_R rows of a level's padded buffer: a lane owns ONE dword column; its left
// / right neighbours come from the adjacent lanes (DPP wave shifts; lanes 0 and 63 load theirs), the horizontal pass is 10
// v_dot4_u32_u8 with the taps shifted to each output's byte offset, the vertical pass 4 v_dot2_u32_u16 per output on row pairs
// built once per source row.  The frame of a level is never read: dwords that hold columns outside the level (and rows outside
// it) are gathered byte by byte through reflect101, so the result is the reflected-border blur for every inner pixel whether or
// not the level's 19-px frame has been written.  The blurred buffer has the geometry of the pyramid (same offsets and strides).
__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

__global__ __launch_bounds__(256) void k_blur_levels(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, size_t pyrImgBytes,
                                                     const LevelGeom *__restrict__ geom, int nlevels, int totalTiles, BlurPlan bp,
                                                     int gaussRounding, uint32_t taps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);
    const int t = bx * 4 + wave;
    if (t >= totalTiles) return;
    int l = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && t >= bp.tileBase[i]) ? 1 : 0;
    const LevelGeom g = geom[l];
    const int tt = t - bp.tileBase[l], ntx = bp.tilesX[l];
    const int ty = tt / ntx, tx = tt - ty * ntx;
    const int y0 = ty * BLUR_R, pstr4 = g.pstride >> 2;
    const int D0 = 4 + tx * 64, DL = (g.w + ORBX_EDGE - 1) >> 2;   // first dword of the tile; dword of the level's last column
    const int D = D0 + lane;
    const uint8_t *src = pyr + (size_t)b * pyrImgBytes + g.poff;
    // dword q of a padded row holds columns x = 4q - 19 .. 4q - 16 of the level; reflected where that is outside [0, w)
    const bool tileEdge = tx == 0 || D0 + 64 >= DL;   // wave-uniform: some dword of the tile (or its right neighbour) needs reflection
    const int Dx = lane == 0 ? D - 1 : D + 1;         // the neighbour dword lanes 0 / 63 fetch themselves
    const int Dc = min(D, pstr4 - 1), Dxc = min(max(Dx, 0), pstr4 - 1);
    int oc[4] = {0, 0, 0, 0}, ox[4] = {0, 0, 0, 0};
    bool eC = false, eX = false;
    if (tileEdge) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int xc = 4 * D + p - ORBX_EDGE, xx = 4 * Dx + p - ORBX_EDGE;
            const int rc = reflect101c(xc, g.w), rx = reflect101c(xx, g.w);
            eC |= rc != xc; eX |= rx != xx;
            oc[p] = rc + ORBX_EDGE; ox[p] = rx + ORBX_EDGE;
        }
    }
    const bool ends = lane == 0 || lane == 63;
    uint32_t v[BLUR_SRC], ex[BLUR_SRC];
#pragma unroll
    for (int k = 0; k < BLUR_SRC; k++) {
        const int ry = reflect101c(y0 - 3 + k, g.h);                              // wave-uniform
        const uint8_t *rowp = src + (size_t)(ry + ORBX_EDGE) * g.pstride;
        if (tileEdge && eC) v[k] = (uint32_t)rowp[oc[0]] | ((uint32_t)rowp[oc[1]] << 8) | ((uint32_t)rowp[oc[2]] << 16) | ((uint32_t)rowp[oc[3]] << 24);
        else v[k] = ((const uint32_t *)rowp)[Dc];
        ex[k] = 0;
        if (ends) {
            if (tileEdge && eX) ex[k] = (uint32_t)rowp[ox[0]] | ((uint32_t)rowp[ox[1]] << 8) | ((uint32_t)rowp[ox[2]] << 16) | ((uint32_t)rowp[ox[3]] << 24);
            else ex[k] = ((const uint32_t *)rowp)[Dxc];
        }
    }
    // ---- horizontal pass: output j of the dword = taps over bytes 1+j .. 7+j of (prev | cur | next)
    // (taps: k3 | k2 << 8 | k1 << 16 | k0 << 24, a kernel argument - the level-wide form is not the default path; ORBX_GAUSS_TAPS_DEFAULT unless
    // the handle's flavour is ORBX_GAUSS_FIXED_TAPS)
    const uint32_t T0 = taps & 0xFFu, T1 = (taps >> 8) & 0xFFu, T2 = (taps >> 16) & 0xFFu, T3 = taps >> 24;
    const uint32_t WP0 = (T0 << 8) | (T1 << 16) | (T2 << 24), WC0 = T3 | (T2 << 8) | (T1 << 16) | (T0 << 24);
    const uint32_t WP1 = (T0 << 16) | (T1 << 24), WC1 = T2 | (T3 << 8) | (T2 << 16) | (T1 << 24), WN1 = T0;
    const uint32_t WP2 = (T0 << 24), WC2 = T1 | (T2 << 8) | (T3 << 16) | (T2 << 24), WN2 = T1 | (T0 << 8);
    const uint32_t WC3 = T0 | (T1 << 8) | (T2 << 16) | (T3 << 24), WN3 = T2 | (T1 << 8) | (T0 << 16);
    uint32_t h01[BLUR_SRC], h23[BLUR_SRC];
#pragma unroll
    for (int k = 0; k < BLUR_SRC; k++) {
        const uint32_t c = v[k];
        uint32_t pv = dpp_wave_shr1(c), nx = dpp_wave_shl1(c);
        pv = lane == 0 ? ex[k] : pv;
        nx = lane == 63 ? ex[k] : nx;
        uint32_t o0 = __builtin_amdgcn_udot4(pv, WP0, 0u, false);
        o0 = __builtin_amdgcn_udot4(c, WC0, o0, false);
        uint32_t o1 = __builtin_amdgcn_udot4(pv, WP1, 0u, false);
        o1 = __builtin_amdgcn_udot4(c, WC1, o1, false);
        o1 = __builtin_amdgcn_udot4(nx, WN1, o1, false);
        uint32_t o2 = __builtin_amdgcn_udot4(pv, WP2, 0u, false);
        o2 = __builtin_amdgcn_udot4(c, WC2, o2, false);
        o2 = __builtin_amdgcn_udot4(nx, WN2, o2, false);
        uint32_t o3 = __builtin_amdgcn_udot4(c, WC3, 0u, false);
        o3 = __builtin_amdgcn_udot4(nx, WN3, o3, false);
        h01[k] = o0 | (o1 << 16);
        h23[k] = o2 | (o3 << 16);
    }
    // ---- vertical pass: out[r] = (18,34).pair[r] + (49,55).pair[r+2] + (49,34).pair[r+4] + (0,18).pair[r+5] + 2^15, pair[k] = rows (k, k+1)
    const uint32_t W01 = T0 | (T1 << 16), W23 = T2 | (T3 << 16), W45 = T2 | (T1 << 16), W6 = T0 << 16;
    uint32_t pr[BLUR_SRC - 1][4];
#pragma unroll
    for (int k = 0; k < BLUR_SRC - 1; k++) {
        pr[k][0] = __builtin_amdgcn_perm(h01[k + 1], h01[k], 0x05040100u);
        pr[k][1] = __builtin_amdgcn_perm(h01[k + 1], h01[k], 0x07060302u);
        pr[k][2] = __builtin_amdgcn_perm(h23[k + 1], h23[k], 0x05040100u);
        pr[k][3] = __builtin_amdgcn_perm(h23[k + 1], h23[k], 0x07060302u);
    }
    uint8_t *dst = blur + (size_t)b * pyrImgBytes + g.poff;
    const bool col_ok = D <= DL;
    // column rounding (orbx_flavour_t): half up = + 2^15; SSE2 = half to even for the level's columns x < (w & ~3):
    // + 32767 + bit 16 of the sum (a v_bfe whose WIDTH is 0 for the other columns)
    uint32_t rc[4], bw[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = 4 * D + q - ORBX_EDGE;
        const bool even = gaussRounding == ORBX_GAUSS_ROUND_SSE2 && x < (g.w & ~3);
        rc[q] = even ? 32767u : 32768u;
        bw[q] = even ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < BLUR_R; r++) {
        uint32_t a[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t s = udot2_u16_acc(pr[r][q], W01, 0u);
            s = udot2_u16_acc(pr[r + 2][q], W23, s);
            s = udot2_u16_acc(pr[r + 4][q], W45, s);
            s = udot2_u16_acc(pr[r + 5][q], W6, s);
            s = s + __builtin_amdgcn_ubfe(s, 16u, bw[q]) + rc[q];
            a[q] = min(s, 0xFFFFFFu);   // the taps sum to 257: a saturated neighbourhood reaches 257 * 65535 + 2^15 > 2^24 (-> 255)
        }
        const uint32_t out = __builtin_amdgcn_perm(a[1], a[0], 0x0c0c0602u) | __builtin_amdgcn_perm(a[3], a[2], 0x06020c0cu);
        const int y = y0 + r;
        if (col_ok && y < g.h) ((uint32_t *)(dst + (size_t)(y + ORBX_EDGE) * g.pstride))[D] = out;
    }
}

template <int GAUSS, bool SPLIT>
__global__ __launch_bounds__(64 * DESC_WAVES) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_describe(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels,
    const uint32_t *__restrict__ lvlKp, int lvlKpCap, const int32_t *__restrict__ lvlCnt,
    orbx_keypoint_t *__restrict__ kps, uint8_t *__restrict__ desc, int32_t *__restrict__ counts, int cap, uint8_t *__restrict__ dbgBlur,
    const uint8_t *__restrict__ blur, unsigned blurMask, DescGroup grp) {
    __shared__ __align__(16) uint8_t smem[DESC_WAVES * DESC_LDS_PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);   // all patches of an image are read through ONE L2
    int o = bx * DESC_WAVES + wave;
    PHASE("locate");
    // locate (level, k) of output ordinal o: level-major concatenation (:1076-1104)
    int l = 0, base = 0, total = 0, obefore = 0;
    {   // lane i < nlevels holds the count of level i: ONE load, a prefix sum inside the first DPP row (16 lanes >= ORBX_MAX_LEVELS:
        // four row_shr additions, no LDS round trips), a ballot
        static_assert(ORBX_MAX_LEVELS <= 16, "the level counts fit one DPP row");
        const int c = lane < nlevels ? lvlCnt[b * nlevels + lane] : 0;
        int inc = c;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);   // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);   // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);   // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);   // row_shr:8
        total = __builtin_amdgcn_readlane(inc, ORBX_MAX_LEVELS - 1);
        if (SPLIT) {
            if (grp.writeCounts && bx == 0 && threadIdx.x == 0) counts[b] = min(total, cap);
            // keypoints in front of this launch's first level (0 for a launch that starts at level 0; the scratch arrays of the later
            // levels are indexed from their own first keypoint)
            const int before = grp.lvBegin > 0 ? __builtin_amdgcn_readlane(inc, max(grp.lvBegin - 1, 0)) : 0;
            if (bx >= grp.copyBlock0) {
                // move the records of the levels [lvEnd, nlevels) from the scratch arrays to their place behind this launch's levels
                const int cA = __builtin_amdgcn_readlane(inc, grp.lvEnd - 1), cB = total - cA;
                const int t = threadIdx.x, i = (bx - grp.copyBlock0) * DESC_COPY_PER_BLOCK + (t >> 4), part = t & 15;
                if (i < cB && cA + i < cap && part < 15) {
                    const size_t si = (size_t)b * cap + i, di = (size_t)b * cap + cA + i;
                    if (part < 7) ((uint32_t *)(kps + di))[part] = ((const uint32_t *)(grp.kpsScratch + si))[part];
                    else ((uint32_t *)(desc + di * 32))[part - 7] = ((const uint32_t *)(grp.descScratch + si * 32))[part - 7];
                }
                return;
            }
            o += before;   // ordinal over all levels
            obefore = before;
            const unsigned long long hit = __ballot(lane >= grp.lvBegin && lane < grp.lvEnd && o < inc);
            if (!hit || o - before >= cap) return;  // wave-uniform
            l = __builtin_ctzll(hit);
        } else {   // the usual launch: every level, into the caller's arrays
            if (bx == 0 && threadIdx.x == 0) counts[b] = min(total, cap);
            const unsigned long long hit = __ballot(lane < nlevels && o < inc);
            if (!hit || o >= cap) return;  // wave-uniform
            l = __builtin_ctzll(hit);
        }
        base = __builtin_amdgcn_readlane(inc - c, l);
    }
    l = __builtin_amdgcn_readfirstlane(l);
    const LevelGeom g = geom[l];
    const uint32_t key = lvlKp[(size_t)b * lvlKpCap + g.lvlKpOff + (o - base)];
    const int cx = (int)(key & 0xFFF) + ORBX_MINB, cy = (int)((key >> 12) & 0xFFF) + ORBX_MINB;
    const int score = (int)(key >> 24);

    PHASE("stage");
    // table entries of this lane, requested before the patch so that their latency is hidden behind the staging
    // (the 4 x 64 pattern pairs of the four ballot rounds, IC_Angle byte weights)
    typedef float float4v __attribute__((ext_vector_type(4)));
    float4v pat[4];    // (x0, y0, x1, y1) of this lane's pair in each of the four ballot rounds
#pragma unroll
    for (int r = 0; r < 4; r++) pat[r] = ((const float4v *)c_patternf)[r * 64 + lane];
    uint32_t hitem[DESC_H_ITERS];   // this lane's item (source offset | destination offset << 16) in each round of the horizontal pass
#pragma unroll
    for (int it = 0; it < DESC_H_ITERS; it++) hitem[it] = c_reach.hitem[it * 64 + lane];
    const uint32_t vlane = c_reach.vlane[lane];   // (column pair | first row << 8) of the vertical pass
    uint32_t icm[4], icw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { icm[j] = c_ic.m[lane][j]; icw[j] = c_ic.w[lane][j]; }
    uint8_t *P = smem + wave * DESC_LDS_PER_WAVE;                     // source patch [43][48] (+pad)
    uint32_t *Tm = (uint32_t *)(P + PROWS * PSTRIDE + PPAD);          // horizontal pass, u16 [43][40]
    uint8_t *Bl = P;                                                  // blurred [37][40], in place of the patch (TCOLS * BSTRIDE <= PROWS * PSTRIDE)

    // ---- stage the 43x43 patch with aligned dword loads (pstride is a multiple of 64)
    const uint8_t *lvl = pyr + (size_t)b * pyrImgBytes + g.poff;
    // byte offset of the patch origin inside the level: the same in every lane (one keypoint per wave) - made scalar so that
    // the staging loads are "scalar base + 32-bit lane offset" (a padded level is far smaller than 4 GiB)
    const size_t a = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((cy + ORBX_EDGE - PR) * g.pstride + (cx + ORBX_EDGE - PR));
    int sh = (int)(a & 3);
    const uint32_t *src = (const uint32_t *)(lvl + (a - sh));
    const int pstr4 = g.pstride >> 2;
    // The 19-px REFLECT_101 frame of the levels >= 1 is only ever read HERE, by the few keypoints closer than PR to a
    // level's edge (the frame of level 0 comes with the copy of the input).  Those keypoints mirror the coordinates
    // themselves, so the pipeline never writes the frames of levels >= 1 (orbx_pyramid_host writes them on demand).
#ifdef ORBX_PHASE_MARKERS   // the counted build keeps the hot path only: no edge keypoint, no level-wide blur, no test hook
    const bool edge = false;
    blurMask = 0; dbgBlur = nullptr;
#else
    const bool edge = l > 0 && (cx < PR || cy < PR || cx + PR >= g.w || cy + PR >= g.h);   // wave-uniform
#endif
    // Level blurred as a whole by k_blur_levels (levels whose keypoints' 37x37 blocks add up to more pixels than the level has):
    // stage the 31 rows of the IC_Angle disc from the level and the 37x37 block from the blurred level; no blur passes here.
    const bool pre = (blurMask >> l) & 1u;                                                  // wave-uniform
    int shB = 0;
    if (pre) {
        constexpr int NA = (31 * 12 + 63) / 64, NB = (TCOLS * 10 + 63) / 64;   // 6 + 6 loads per lane, all in flight before the first LDS write
        uint32_t va[NA], vb[NB];
        const uint8_t *sbase = (const uint8_t *)src + (size_t)(PR - 15) * g.pstride;   // row cy - 15 of the patch window
#pragma unroll
        for (int k = 0; k < NA; k++) {
            const int i = min(lane + 64 * k, 31 * 12 - 1), r = i / 12, cc = i - r * 12;
            va[k] = *(const uint32_t *)(sbase + (uint32_t)(r * pstr4 + cc) * 4u);
        }
        // blurred pixel (cx - 18 + c, cy - 18 + r): same padded geometry as the level
        const size_t a2 = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((cy + ORBX_EDGE - ORBX_DESC_R) * g.pstride + (cx + ORBX_EDGE - ORBX_DESC_R));
        shB = (int)(a2 & 3);
        const uint8_t *bbase = blur + (size_t)b * pyrImgBytes + g.poff + (a2 - shB);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = min(lane + 64 * k, TCOLS * 10 - 1), r = i / 10, cc = i - r * 10;
            vb[k] = *(const uint32_t *)(bbase + (uint32_t)(r * pstr4 + cc) * 4u);
        }
#pragma unroll
        for (int k = 0; k < NA; k++) ((uint32_t *)(P + (PR - 15) * PSTRIDE))[min(lane + 64 * k, 31 * 12 - 1)] = va[k];
#pragma unroll
        for (int k = 0; k < NB; k++) Tm[min(lane + 64 * k, TCOLS * 10 - 1)] = vb[k];   // rows of BSTRIDE = 40 bytes = 10 dwords
    } else if (edge) {
        sh = 0;
        const uint8_t *inner = lvl + (size_t)ORBX_EDGE * g.pstride + ORBX_EDGE;
        for (int base0 = 0; base0 < PROWS * PSTRIDE; base0 += 64 * 8) {
            uint8_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = min(base0 + lane + 64 * k, PROWS * PSTRIDE - 1);
                const int r = i / PSTRIDE, c = i - r * PSTRIDE;
                v[k] = inner[(size_t)reflect101c(cy - PR + r, g.h) * g.pstride + reflect101c(cx - PR + c, g.w)];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) P[min(base0 + lane + 64 * k, PROWS * PSTRIDE - 1)] = v[k];   // unconditional, clamped (see below)
        }
    } else
    {   // all (PROWS*12 + 63) / 64 loads of a lane are in flight before the first LDS write: one memory latency per keypoint
        constexpr int NI = (PROWS * 12 + 63) / 64;
        uint32_t v[NI];
        // item i = lane + 64 * k is dword c = i % 12 of row r = i / 12: +64 items = 5 rows + 4 dwords with a carry into the
        // row.  The patch origin is wave-uniform (scalar base), so a lane only keeps a 32-bit byte offset.
        const uint8_t *sbase = (const uint8_t *)src;
        int c = lane % 12;
        uint32_t off = (uint32_t)((lane / 12) * pstr4 + c) * 4u;
        const uint32_t step = (uint32_t)(5 * pstr4 + 4) * 4u, stepCarry = (uint32_t)(6 * pstr4 + 4 - 12) * 4u;
#pragma unroll
        for (int k = 0; k < NI - 1; k++) {           // items < 512 <= PROWS * 12: no clamping
            v[k] = *(const uint32_t *)(sbase + off);
            c += 4;
            const bool carry = c >= 12;
            c -= carry ? 12 : 0;
            off += carry ? stepCarry : step;
        }
        {   // last step: items 512 .. 575, clamped to the last dword of the patch
            static_assert(NI == 9 && PROWS * 12 == 516, "tail of the patch staging");
            const int i = min(lane + 64 * (NI - 1), PROWS * 12 - 1), r = i / 12, cc = i - r * 12;
            v[NI - 1] = *(const uint32_t *)(sbase + (uint32_t)(r * pstr4 + cc) * 4u);
        }
        // unconditional writes to the clamped slot (lanes past the end rewrite the last dword with its own value): a store
        // under a lane condition lets the compiler sink the last load behind a divergent branch = one more memory latency
#pragma unroll
        for (int k = 0; k < NI; k++) ((uint32_t *)P)[min(lane + 64 * k, PROWS * 12 - 1)] = v[k];
    }
    wave_sync();
    // pixel (cx-21+c, cy-21+r) is byte P[r*48 + sh + c], r,c in [0,43)

    PHASE("ic_angle");
    // ---- IC_Angle: two lanes per row v of the radius-15 disc (u = -15..0 | 1..15)
    int m10 = 0, m01 = 0;
    {   // all 64 lanes: the table weights of lanes 62, 63 are zero (their row, v = 16, lies inside the patch)
        const int v = (lane >> 1) - 15, half = lane & 1;
        const int o0 = sh + PR - 15 + 16 * half;  // byte offset of u = -15 (half 0) / u = 1 (half 1)
        const uint32_t *row = (const uint32_t *)(P + (PR + v) * PSTRIDE) + (o0 >> 2);
        const int sa8 = o0 & 3;
        uint32_t w[5], wa[4];
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = row[k];
#pragma unroll
        for (int k = 0; k < 4; k++) wa[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sa8);  // 16 bytes from o0
        // sum of the pixels and of |u| * pixel over this lane's 16 columns: 8 byte dot products with the table weights
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s0 = __builtin_amdgcn_udot4(wa[j], icm[j], s0, false);
            s1 = __builtin_amdgcn_udot4(wa[j], icw[j], s1, false);
        }
        m10 = half ? (int)s1 : -(int)s1;   // u <= 0 in half 0
        m01 = v * (int)s0;
    }
    m10 = wave_total_i32(m10);
    m01 = wave_total_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- horizontal 7-tap pass: 4 outputs per lane-iteration from one aligned b128 read.  Output j of a group is the
    // 7-tap sum over bytes o .. o+6 of those 16 bytes, o = sh + j.  Instead of realigning the DATA (9 v_alignbyte per
    // group) the WEIGHTS are shifted: W[j] = taps << 8*o as a 128-bit constant, wave-uniform (scalar registers), and the
    // sum is a v_dot4_u32_u8 per dword the window can touch - 13 dot products per group, no alignment instructions.
    PHASE("horizontal");
    const uint8_t *Bl0 = Bl;           // blurred pixel (cx - 18 + c, cy - 18 + r) = Bl0[r * BSTRIDE + c]
    if (pre) Bl0 = (const uint8_t *)Tm + shB;
    else {
    uint32_t Wt[4][4];
    {
        // ORBX_GAUSS_FIXED_TAPS: the taps are the handle's (grp.taps, wave-uniform); the other flavours keep their literals
        const uint32_t tp = GAUSS == ORBX_GAUSS_FIXED_TAPS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.taps) : ORBX_GAUSS_TAPS_DEFAULT;
        const unsigned long long T0 = tp & 0xFFu, T1 = (tp >> 8) & 0xFFu, T2 = (tp >> 16) & 0xFFu, T3 = tp >> 24;
        const unsigned __int128 K7 = (unsigned __int128)(T0 | (T1 << 8) | (T2 << 16) | (T3 << 24) | (T2 << 32) | (T1 << 40) | (T0 << 48));
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const unsigned __int128 v = K7 << (8 * (sh + jj));
#pragma unroll
            for (int m = 0; m < 4; m++) Wt[jj][m] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> (32 * m)));
        }
    }
    // round `it`: this lane's item (row r, column group cg) of the list c_reach.hitem - only what lies inside the disc
    uint8_t *Tt = (uint8_t *)Tm;
#pragma unroll
    for (int it = 0; it < DESC_H_ITERS; it++) {
        const uint32_t *pr = (const uint32_t *)(P + (hitem[it] & 0xFFFFu));   // row r, dwords cg .. cg+3 = bytes 4cg .. 4cg+15
        const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = pr[3];
        uint32_t oo[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {   // window bytes sh+jj .. sh+jj+6 <= 12: dword 3 only for jj = 3 (<= 65535 in total)
            uint32_t acc = __builtin_amdgcn_udot4(d0, Wt[jj][0], 0u, false);
            acc = __builtin_amdgcn_udot4(d1, Wt[jj][1], acc, false);
            acc = __builtin_amdgcn_udot4(d2, Wt[jj][2], acc, false);
            if (jj == 3) acc = __builtin_amdgcn_udot4(d3, Wt[jj][3], acc, false);
            oo[jj] = acc;
        }
        uint16_t *dst = (uint16_t *)(Tt + (hitem[it] >> 16));   // column 4cg + jj, row r: four 16-bit stores off one address
#pragma unroll
        for (int jj = 0; jj < 4; jj++) dst[jj * (TCS / 2)] = (uint16_t)oo[jj];
    }
    wave_sync();
    PHASE("vertical");

    // ---- vertical pass: lane = (column pair, run of DESC_V_ROWS rows starting at an even row).  A dword of the transposed
    // intermediate holds rows (2p, 2p+1) of one column, so every output is four v_dot2_u32_u16 on dwords as they come from LDS:
    //   even row r: (18,34).pair[r/2] + (49,55).pair[r/2+1] + (49,34).pair[r/2+2] + (18, 0).pair[r/2+3]
    //   odd  row r: ( 0,18).pair[(r-1)/2] + (34,49).pair[..+1] + (55,49).pair[..+2] + (34,18).pair[..+3]
    // and the rounded bytes of both columns leave through one v_perm (byte 2 of the sums clamped to 2^24 - 1).
    {
        const int cp = (int)(vlane & 0xFFu), r0 = (int)((vlane >> 8) & 0xFFu);
        const uint32_t *col = (const uint32_t *)(Tt + (vlane >> 16));   // rows (r0, r0 + 1) of column 2cp; column 2cp + 1 follows TCS bytes behind
        constexpr int NP = DESC_V_ROWS / 2 + 3;   // row pairs a run touches
        uint32_t A[NP], B[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) { A[k] = col[k]; B[k] = col[k + TCS / 4]; }
        // column rounding of the flavour (include/orbx.h): half up = the 2^15 the sums start from; SSE2 = round half to EVEN for the
        // level's columns x < (w & ~3) - sum + 32767 + bit 16 of the sum, the bit taken by a v_bfe whose WIDTH is 0 for the
        // columns of the scalar tail (which keep + 2^15).  Blurred column c of the block is level column cx - 18 + c.
        uint32_t rc0 = 0, rc1 = 0, bw0 = 0, bw1 = 0;
        if (GAUSS == ORBX_GAUSS_ROUND_SSE2) {
            const int x0 = cx - ORBX_DESC_R + 2 * cp, wv = g.w & ~3;
            bw0 = x0 < wv ? 1u : 0u; bw1 = x0 + 1 < wv ? 1u : 0u;
            rc0 = (32768u - bw0) << 8; rc1 = (32768u - bw1) << 8;
        }
        // The taps carry a factor 256, so a sum S is held as 256 S and the CLAMP bit of v_dot2_u32_u16 saturates it at 2^32 - exactly
        // where the rounded byte would pass 255 (the taps add up to 257: a saturated patch reaches 257 * 65535 + 2^15 > 2^24); every
        // term is non-negative, so saturating the partial sums saturates the total.  The result byte is bits 31:24: no v_min.
        constexpr uint32_t A_INIT = GAUSS == ORBX_GAUSS_ROUND_SSE2 ? 0u : 1u << 23;
        const uint32_t tv = GAUSS == ORBX_GAUSS_FIXED_TAPS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)grp.taps) : ORBX_GAUSS_TAPS_DEFAULT;
        const uint32_t V0 = tv & 0xFFu, V1 = (tv >> 8) & 0xFFu, V2 = (tv >> 16) & 0xFFu, V3 = tv >> 24;   // (constants unless FIXED_TAPS)
        const uint32_t WE[4] = {(V0 | (V1 << 16)) << 8, (V2 | (V3 << 16)) << 8, (V2 | (V1 << 16)) << 8, V0 << 8};
        const uint32_t WO[4] = {V0 << 24, (V1 | (V2 << 16)) << 8, (V3 | (V2 << 16)) << 8, (V1 | (V0 << 16)) << 8};
        uint8_t *out = Bl + r0 * BSTRIDE + 2 * cp;
#pragma unroll
        for (int i = 0; i < DESC_V_ROWS; i++) {
            uint32_t a0 = A_INIT, a1 = A_INIT;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = (i & 1) ? WO[k] : WE[k];
                a0 = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, A[(i >> 1) + k]), __builtin_bit_cast(ushort2v, w), a0, true);
                a1 = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, B[(i >> 1) + k]), __builtin_bit_cast(ushort2v, w), a1, true);
            }
            if (GAUSS == ORBX_GAUSS_ROUND_SSE2) {   // + (32767 + bit 16 of S) << 8, saturating
                a0 = __builtin_elementwise_add_sat(a0, (__builtin_amdgcn_ubfe(a0, 24u, bw0) << 8) + rc0);
                a1 = __builtin_elementwise_add_sat(a1, (__builtin_amdgcn_ubfe(a1, 24u, bw1) << 8) + rc1);
            }
            *(uint16_t *)(out + i * BSTRIDE) = (uint16_t)__builtin_amdgcn_perm(a1, a0, 0x0c0c0703u);   // rows r0 .. r0+11 <= 37: inside the block's 38 x 40 bytes
        }
    }
    wave_sync();
    }   // !pre

    PHASE("brief");
    if (dbgBlur) {   // test hook (wave-uniform, NULL in production): the 37x37 blurred block around the keypoint, rows of 37 bytes;
        // pixels no tap can reach (outside the disc of c_reach) are reported as 0 - the fused form never computes them
        uint8_t *o37 = dbgBlur + ((size_t)b * cap + o) * (TCOLS * TCOLS);
        for (int i = lane; i < TCOLS * TCOLS; i += 64) {
            const int rr = i / TCOLS, cc = i % TCOLS, dy = rr > ORBX_DESC_R ? rr - ORBX_DESC_R : ORBX_DESC_R - rr;
            o37[i] = dy <= c_reach.hh[cc] ? Bl0[rr * BSTRIDE + cc] : (uint8_t)0;
        }
    }
    // ---- steered BRIEF: 4 rounds x 64 pairs, one ballot = 8 descriptor bytes
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = angle * factorPI;
    float ca, sa;
    sincosf_glibc(ang, sa, ca);
    // LDS byte address of the block's centre as a float (exact: far below 2^24); a tap's address is then ONE float multiply-add of
    // its rounded coordinates - cvRound(row) * 40 + (cvRound(col) + centre), every term a small integer - and one conversion
    const float centre = (float)(int)(uint32_t)(uintptr_t)(Bl0 - smem + ORBX_DESC_R * BSTRIDE + ORBX_DESC_R);
    unsigned long long bits[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float x0 = pat[r].x, y0 = pat[r].y, x1 = pat[r].z, y1 = pat[r].w;
        // (x*sa + y*ca, x*ca - y*sa) on the packed-f32 pipe: two multiplies and one add per point, each rounded on its own
        // exactly like the scalar form (no contraction)
        typedef float float2v __attribute__((ext_vector_type(2)));
        const float2v sc = {sa, ca}, cs = {ca, -sa};
        const float2v p0 = (float2v){x0, x0} * sc + (float2v){y0, y0} * cs;
        const float2v p1 = (float2v){x1, x1} * sc + (float2v){y1, y1} * cs;
        const int i0 = (int)__builtin_fmaf(__builtin_rintf(p0.x), (float)BSTRIDE, __builtin_rintf(p0.y) + centre);
        const int i1 = (int)__builtin_fmaf(__builtin_rintf(p1.x), (float)BSTRIDE, __builtin_rintf(p1.y) + centre);
        const int t0 = smem[i0], t1 = smem[i1];
        bits[r] = __ballot(t0 < t1);
    }
    PHASE("store");
    const size_t oi = (size_t)b * cap + (o - obefore);   // a launch that starts behind level 0 fills the scratch arrays from their start
    if (lane < 4) ((unsigned long long *)(desc + oi * 32))[lane] = bits[lane];
    if (grp.hostDelta && lane < 4) ((unsigned long long *)(desc + oi * 32 + grp.hostDelta))[lane] = bits[lane];   // (wave-uniform condition)
    if (lane == 0) {
        orbx_keypoint_t kp;
        kp.x = (float)cx;
        kp.y = (float)cy;
        if (l != 0) { kp.x *= g.scale; kp.y *= g.scale; }  // pt *= mvScaleFactor[level]  (:1095-1101)
        kp.size = g.size;
        kp.angle = angle;
        kp.response = (float)score;
        kp.octave = l;
        kp.class_id = -1;
        kps[oi] = kp;
        if (grp.hostDelta) *(orbx_keypoint_t *)((uint8_t *)(kps + oi) + grp.hostDelta) = kp;
    }
}

#define ORBX_DESC_INSTANCE(G, S)                                                                                                             \
    template __global__ void k_describe<G, S>(const uint8_t *, size_t, const LevelGeom *, int, const uint32_t *, int, const int32_t *, \
                                              orbx_keypoint_t *, uint8_t *, int32_t *, int, uint8_t *, const uint8_t *, unsigned, DescGroup);
ORBX_DESC_INSTANCE(ORBX_GAUSS_ROUND_HALF_UP, false)
ORBX_DESC_INSTANCE(ORBX_GAUSS_ROUND_HALF_UP, true)
ORBX_DESC_INSTANCE(ORBX_GAUSS_ROUND_SSE2, false)
ORBX_DESC_INSTANCE(ORBX_GAUSS_ROUND_SSE2, true)
ORBX_DESC_INSTANCE(ORBX_GAUSS_FIXED_TAPS, false)
ORBX_DESC_INSTANCE(ORBX_GAUSS_FIXED_TAPS, true)

#ifdef ORBX_DEVELOPER
#include "orbx_dev.h"
// ---- test hook: the device's cosf / sinf restatement on an array of angles (tests compare it with the oracle's and with
// the host libm over the whole angle domain; a descriptor only ever exercises the angles its keypoints happen to have)
__global__ __launch_bounds__(256) void k_debug_sincosf(const float *__restrict__ a, int n, float *__restrict__ s, float *__restrict__ c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincosf_glibc(a[i], sn, cs);
    s[i] = sn; c[i] = cs;
}
extern "C" int orbx_debug_sincosf(const float *angles, int n, float *sin_out, float *cos_out, int device) {
    if (!angles || !sin_out || !cos_out || n < 1) { orbx_set_error("orbx_debug_sincosf: bad arguments"); return ORBX_ERR_ARG; }
    ORBX_HIP(hipSetDevice(device));
    float *d = nullptr;
    ORBX_HIP(hipMalloc(&d, sizeof(float) * 3 * (size_t)n));
    hipError_t e = hipMemcpy(d, angles, sizeof(float) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_debug_sincosf, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, d + n, d + 2 * (size_t)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(sin_out, d + n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cos_out, d + 2 * (size_t)n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
    hipFree(d);
    ORBX_HIP(e);
    return ORBX_OK;
}
#endif   // ORBX_DEVELOPER
