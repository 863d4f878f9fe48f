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
#include "orbx_internal.h"
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

int g_debug[8] = {0};
extern "C" int orbx_debug_set(int key, int value) { if (key < 0 || key >= 8) return ORBX_ERR_ARG; g_debug[key] = value; return ORBX_OK; }

// ------------------------------------------------------------------------------------
// constant tables
__constant__ int8_t c_pattern[1024] = {
#include "../../include/orb_pattern_31.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// 7x7 sigma=2 Gaussian in 8-bit fixed point (cvRound(k*256)), sum 257
__constant__ int c_gauss[7] = {18, 34, 49, 55, 49, 34, 18};

// wave-synchronous LDS hand-off: all 64 lanes of a wave run in lock-step; drain the LDS
// queue and forbid the compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Wave64 inclusive prefix sum / total on the DPP data path (row shifts, then the two row broadcasts): ~6 VALU steps
// instead of six ds_bpermute round trips — these scans sit on the critical path of single-wave code (quad-tree passes).
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_total_i32(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63); }

__device__ __forceinline__ int reflect101(int i, int n) {  // valid for -n < i < 2n-1
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------
// K1: ComputePyramid (:1107-1132) in ONE launch.  A workgroup owns a tile of the coarsest
// level and, through the resize source offsets, the corresponding rectangles of every finer
// level.  It loads its level-0 rectangle from the input once, then computes level after level
// from the previous one held in LDS (ping-pong), so a level is never read back from memory to
// build the next.  Rectangles: own_l partitions level l across the tiles; comp_l = own_l plus
// whatever comp_{l+1} needs (1-2 px of halo per level, recomputed by neighbouring tiles, never
// written twice).  Arithmetic per pixel is exactly K1b's: 8UC1 fixed-point bilinear of OpenCV
// <= 3.3 with the host-built coefficient tables; the 19-px BORDER_REFLECT_101 frame
// (copyMakeBorder, :1122-1128) is written by the owner of the mirrored inner pixel.
struct PyrSpan { short o0, o1, c0, c1; };  // owned [o0,o1) and computed [c0,c1) range along one axis

__global__ __launch_bounds__(256) void k_pyramid_fused(
    const uint8_t *__restrict__ src, int sstride, size_t simg, uint8_t *__restrict__ pyr, size_t pyrImgBytes,
    const LevelGeom *__restrict__ geom, int nlevels, const int32_t *__restrict__ tab, int xSpanOff, int ySpanOff,
    int tilesX, int tilesY, int bufBytes, int maxPar) {
    extern __shared__ __align__(16) uint8_t smem[];
    // per-column {i0 | i1<<16, a0 | a1<<16} and per-row {r0 | r1<<16, b0 | b1<<16} of every level
    uint2 *xpar = (uint2 *)(smem + 2 * bufBytes), *ypar = xpar + maxPar;
    __shared__ PyrSpan sX[ORBX_MAX_LEVELS], sY[ORBX_MAX_LEVELS];
    __shared__ int xo[ORBX_MAX_LEVELS + 1], yo[ORBX_MAX_LEVELS + 1];
    const int tid = threadIdx.x, tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX, b = blockIdx.y;
    if (tid < nlevels) sX[tid] = ((const PyrSpan *)(tab + xSpanOff))[tid * tilesX + tx];
    else if (tid >= 32 && tid < 32 + nlevels) sY[tid - 32] = ((const PyrSpan *)(tab + ySpanOff))[(tid - 32) * tilesY + ty];
    __syncthreads();
    if (tid == 0) {
        int ax = 0, ay = 0;
        for (int l = 0; l < nlevels; l++) {
            xo[l] = ax; yo[l] = ay;
            ax += sX[l].c1 - sX[l].c0; ay += sY[l].c1 - sY[l].c0;
        }
        xo[nlevels] = ax; yo[nlevels] = ay;
    }
    __syncthreads();
    {   // ONE round of global loads: the resize parameters of every level + the level-0 rectangle
        const int nx = xo[nlevels], ny = yo[nlevels];
        for (int i = tid; i < nx + ny; i += 256) {
            const bool isx = i < nx;
            const int j = isx ? i : i - nx;
            const int *off = isx ? xo : yo;
            int l = 1;
            while (l + 1 <= nlevels && j >= off[l + 1]) l++;   // level of entry j (level 0 has no parameters)
            if (j < off[1]) continue;
            const LevelGeom *g = geom + l;
            const PyrSpan cs = isx ? sX[l] : sY[l], ps = isx ? sX[l - 1] : sY[l - 1];
            const int k = cs.c0 + (j - off[l]);
            uint2 q;
            if (isx) {
                const int sx = tab[g->xofsOff + k], sw = g[-1].w;
                q.x = (uint32_t)(sx - ps.c0) | ((uint32_t)(min(sx + 1, sw - 1) - ps.c0) << 16);  // clamp acts only where a1 == 0
                q.y = (uint32_t)tab[g->xalphaOff + k];
                xpar[j] = q;
            } else {
                const int sy = tab[g->yofsOff + k], shh = g[-1].h;
                q.x = (uint32_t)(min(max(sy, 0), shh - 1) - ps.c0) | ((uint32_t)(min(max(sy + 1, 0), shh - 1) - ps.c0) << 16);
                q.y = (uint32_t)tab[g->ybetaOff + k];
                ypar[j] = q;
            }
        }
        const PyrSpan X = sX[0], Y = sY[0];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0;
        const uint8_t *s = src + (size_t)b * simg + (size_t)Y.c0 * sstride + X.c0;
        const unsigned M = ((1u << 20) + cw - 1) / cw;
        for (int i = tid; i < cw * ch; i += 256) {
            const int y = (int)(((unsigned)i * M) >> 20), x = i - y * cw;
            smem[i] = s[(size_t)y * sstride + x];
        }
    }
    __syncthreads();
    uint8_t *base = pyr + (size_t)b * pyrImgBytes;
    for (int l = 0; l < nlevels; l++) {
        const PyrSpan X = sX[l], Y = sY[l];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0;
        uint8_t *cur = smem + (l & 1) * bufBytes;  // ping-pong; plain offsets keep the LDS address space
        const LevelGeom *g = geom + l;
        const int lw = g->w, lh = g->h, pstride = g->pstride;
        if (l > 0) {
            const uint8_t *prev = smem + ((l & 1) ^ 1) * bufBytes;
            const int pw = sX[l - 1].c1 - sX[l - 1].c0;
            const uint2 *xp = xpar + xo[l], *yp = ypar + yo[l];
            const unsigned M = ((1u << 20) + cw - 1) / cw;
            for (int i = tid; i < cw * ch; i += 256) {
                const int y = (int)(((unsigned)i * M) >> 20), x = i - y * cw;
                const uint2 yq = yp[y], xq = xp[x];
                const uint8_t *S0 = prev + (yq.x & 0xFFFF) * pw, *S1 = prev + (yq.x >> 16) * pw;
                const int b0 = (int16_t)(yq.y & 0xFFFF), b1 = (int16_t)(yq.y >> 16);
                const int a0 = (int16_t)(xq.y & 0xFFFF), a1 = (int16_t)(xq.y >> 16);
                const int i0 = xq.x & 0xFFFF, i1 = xq.x >> 16;
                const int h0 = S0[i0] * a0 + S0[i1] * a1;
                const int h1 = S1[i0] * a0 + S1[i1] * a1;
                cur[i] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
            __syncthreads();
        }
        // write the owned rectangle and its mirror images in the 19-px REFLECT_101 frame
        uint8_t *dst = base + g->poff;
        const int ow = X.o1 - X.o0, oh = Y.o1 - Y.o0;
        const unsigned Mo = ((1u << 20) + ow - 1) / max(ow, 1);
        // workgroup-uniform: does the owned rectangle touch a band that is mirrored into the frame?
        const bool edgeX = X.o0 <= ORBX_EDGE || X.o1 >= lw - ORBX_EDGE, edgeY = Y.o0 <= ORBX_EDGE || Y.o1 >= lh - ORBX_EDGE;
        if (!edgeX && !edgeY) {   // interior tile (the common case): plain copy
            for (int i = tid; i < ow * oh; i += 256) {
                const int iy = (int)(((unsigned)i * Mo) >> 20), ix = i - iy * ow;
                dst[(size_t)(Y.o0 + iy + ORBX_EDGE) * pstride + X.o0 + ix + ORBX_EDGE] = cur[(Y.o0 + iy - Y.c0) * cw + (X.o0 + ix - X.c0)];
            }
        } else {
            for (int i = tid; i < ow * oh; i += 256) {
                const int iy = (int)(((unsigned)i * Mo) >> 20), ix = i - iy * ow;
                const int x = X.o0 + ix, y = Y.o0 + iy;
                const uint8_t v = cur[(y - Y.c0) * cw + (x - X.c0)];
                const int px = x + ORBX_EDGE, py = y + ORBX_EDGE;
                const int mx = (x >= 1 && x <= ORBX_EDGE) ? ORBX_EDGE - x
                               : (x >= lw - 1 - ORBX_EDGE && x <= lw - 2) ? 2 * (lw - 1) - x + ORBX_EDGE : -1;
                const int my = (y >= 1 && y <= ORBX_EDGE) ? ORBX_EDGE - y
                               : (y >= lh - 1 - ORBX_EDGE && y <= lh - 2) ? 2 * (lh - 1) - y + ORBX_EDGE : -1;
                dst[(size_t)py * pstride + px] = v;
                if (mx >= 0) dst[(size_t)py * pstride + mx] = v;
                if (my >= 0) {
                    dst[(size_t)my * pstride + px] = v;
                    if (mx >= 0) dst[(size_t)my * pstride + mx] = v;
                }
            }
        }
        // level l+1 writes the other buffer; the barrier after its compute orders this level's
        // reads of `cur` before `cur` is overwritten by level l+2
    }
}

// ------------------------------------------------------------------------------------
// K1 (level-per-launch form, the default): ComputePyramid as  pad(level 0) -> resize 1..L-1 -> pad(1..L-1).
//  * k_pyr_level: a wave owns 128 output columns x PYR_RW output rows; a lane owns TWO fixed
//    columns (2j-1, 2j: the pair is 2-byte aligned in the padded row), so everything that depends
//    on the column — source offset, v_perm selector that lifts the two source bytes into a u16
//    pair, the packed (a0,a1) — is set up once.  Per source row and lane: ONE aligned 8-byte load,
//    two v_perm + two v_dot2_u32_u16 (the horizontal pass of both columns); consecutive output
//    rows share a source row (sy advances by 1 or 2), which is kept in registers, so a row costs
//    ~1.2 loads.  Vertical pass and rounding exactly as cv::resize's VResizeLinear (8UC1, <= 3.3).
//  * k_pyr_pad: copyMakeBorder(REFLECT_101) (:1122-1128) as a gather, one aligned dword per
//    thread; for level 0 it also is the copy of the input into the padded buffer.
// No LDS, no barriers, no dependent chain inside a workgroup (the fused kernel above waits ~45 %
// of its time on its 8-level chain).
#define PYR_RW 8   // output rows per wave
#define PYR_SR 12  // source rows fetched up front: covers PYR_RW rows at scale factors up to ~1.4
__device__ __forceinline__ int reflect101c(int p, int n) {   // reflect101 + clamp (dword tails past the frame)
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return min(max(p, 0), n - 1);
}
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b), 0u, false);
}
__device__ __forceinline__ uint32_t udot2_u16_acc(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b), c, false);
}

__global__ __launch_bounds__(256) void k_pyr_level(uint8_t *__restrict__ pyr, size_t pyrImgBytes,
                                                   const LevelGeom *__restrict__ geom, int l,
                                                   const int32_t *__restrict__ tab, int nxc, int nbands) {
    int bx, b;
    xcd_block_map(bx, b);
    const int wave = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (wave >= nxc * nbands) return;
    const int band = wave / nxc, xc = wave - band * nxc;
    const LevelGeom G = geom[l];
    const int sw = geom[l - 1].w, sh = geom[l - 1].h, sps = geom[l - 1].pstride;
    uint8_t *base = pyr + (size_t)b * pyrImgBytes;
    const uint8_t *srow0 = base + geom[l - 1].poff + (size_t)ORBX_EDGE * sps;   // padded row of source row 0
    uint8_t *drow0 = base + G.poff + (size_t)ORBX_EDGE * G.pstride + ORBX_EDGE;
    const int x0 = xc * 128 + 2 * lane - 1, x1 = x0 + 1;
    // The pair store is 2-byte aligned (ORBX_EDGE + x0 is even).  Columns -1 and w fall on frame bytes
    // next to the inner row, which k_pyr_pad rewrites afterwards; lanes further right store nothing.
    const bool vst = x0 < G.w;
    const int xa = min(max(x0, 0), G.w - 1), xb = min(x1, G.w - 1);
    const int ca = ORBX_EDGE + tab[G.xofsOff + xa], cb = ORBX_EDGE + tab[G.xofsOff + xb];   // byte column in the padded source row
    const uint32_t aa = (uint32_t)tab[G.xalphaOff + xa], ab = (uint32_t)tab[G.xalphaOff + xb];
    const int A = ca & ~3;                         // cb - ca <= 2: both byte pairs lie inside [A, A+8)
    const uint32_t oa = (uint32_t)(ca - A), ob = (uint32_t)(cb - A);
    const uint32_t selA = oa | ((oa + 1) << 16) | 0x0C000C00u, selB = ob | ((ob + 1) << 16) | 0x0C000C00u;
    (void)sw;
    const int y0 = band * PYR_RW, nrow = min(PYR_RW, G.h - y0);
    // fast path: the band's source rows rf .. rf+PYR_SR-1 are fetched up front (one memory latency per
    // wave), then consumed in order.  Lane i < 8 holds the row table of output row y0+i; bit k of
    // `mask` says "the output row whose second source row is rf+k is due after source row k".
    const int yl = min(y0 + (lane & 7), G.h - 1);
    const int vsy = tab[G.yofsOff + yl];
    const uint32_t vbt = (uint32_t)tab[G.ybetaOff + yl];
    const int rf = __builtin_amdgcn_readfirstlane(vsy);
    const int kk = vsy + 1 - rf;
    const int prevsy = __shfl_up(vsy, 1);
    const bool okl = (lane & 7) >= nrow || (rf >= 0 && vsy + 1 <= sh - 1 && kk < PYR_SR && ((lane & 7) == 0 || vsy > prevsy));
    const bool regular = (__ballot(okl) & 0xFFull) == 0xFFull;
    if (regular) {
        uint32_t m = (lane & 7) < nrow ? 1u << (kk & 31) : 0u;
        m |= __shfl_xor(m, 1); m |= __shfl_xor(m, 2); m |= __shfl_xor(m, 4);
        const uint32_t mask = __builtin_amdgcn_readfirstlane(m);
        uint2 q[PYR_SR];
#pragma unroll
        for (int k = 0; k < PYR_SR; k++) q[k] = *(const uint2 *)(srow0 + (size_t)min(rf + k, sh - 1) * sps + A);
        uint32_t tpa = 0, tpb = 0;
        int cnt = 0;
        uint8_t *d = drow0 + (size_t)y0 * G.pstride + x0;
#pragma unroll
        for (int k = 0; k < PYR_SR; k++) {
            const uint32_t tca = udot2_u16(__builtin_amdgcn_perm(q[k].y, q[k].x, selA), aa) >> 4;
            const uint32_t tcb = udot2_u16(__builtin_amdgcn_perm(q[k].y, q[k].x, selB), ab) >> 4;
            if (k > 0 && ((mask >> k) & 1u)) {   // wave-uniform
                const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)vbt, cnt);
                const uint32_t b0 = bb & 0xFFFFu, b1 = bb >> 16;
                const uint32_t pa = (((b0 * tpa) >> 16) + ((b1 * tca) >> 16) + 2) >> 2;
                const uint32_t pb = (((b0 * tpb) >> 16) + ((b1 * tcb) >> 16) + 2) >> 2;
                if (vst) *(uint16_t *)d = (uint16_t)(pa | (pb << 8));
                d += G.pstride;
                cnt++;
            }
            tpa = tca; tpb = tcb;
        }
        return;
    }
    // general path (clamped source rows): one output row at a time
    int cr0 = -1, cr1 = -1;
    uint32_t t0a = 0, t0b = 0, t1a = 0, t1b = 0;
    for (int y = y0; y < y0 + nrow; y++) {
        const int syy = tab[G.yofsOff + y];
        const uint32_t bb = (uint32_t)tab[G.ybetaOff + y];
        const int r0 = min(max(syy, 0), sh - 1), r1 = min(max(syy + 1, 0), sh - 1);
        if (r0 == cr1) { t0a = t1a; t0b = t1b; cr0 = cr1; }
        else if (r0 != cr0) {
            const uint2 q = *(const uint2 *)(srow0 + (size_t)r0 * sps + A);
            t0a = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selA), aa) >> 4;
            t0b = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selB), ab) >> 4;
            cr0 = r0;
        }
        if (r1 != cr1) {
            if (r1 == cr0) { t1a = t0a; t1b = t0b; }
            else {
                const uint2 q = *(const uint2 *)(srow0 + (size_t)r1 * sps + A);
                t1a = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selA), aa) >> 4;
                t1b = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selB), ab) >> 4;
            }
            cr1 = r1;
        }
        const uint32_t b0 = bb & 0xFFFFu, b1 = bb >> 16;
        const uint32_t pa = (((b0 * t0a) >> 16) + ((b1 * t1a) >> 16) + 2) >> 2;
        const uint32_t pb = (((b0 * t0b) >> 16) + ((b1 * t1b) >> 16) + 2) >> 2;
        if (vst) *(uint16_t *)(drow0 + (size_t)y * G.pstride + x0) = (uint16_t)(pa | (pb << 8));
    }
}

// FULL: every dword of the padded level from the input image (level 0).  !FULL: only the dwords
// that contain frame bytes, gathered from the level's own inner pixels (levels >= 1, blockIdx.y).
template <bool FULL>
__global__ __launch_bounds__(256) void k_pyr_pad(const uint8_t *__restrict__ img, int sstride, size_t simg,
                                                 uint8_t *__restrict__ pyr, size_t pyrImgBytes,
                                                 const LevelGeom *__restrict__ geom, int l0) {
    const int l = l0 + blockIdx.y;
    const LevelGeom G = geom[l];
    uint8_t *lvl = pyr + (size_t)blockIdx.z * pyrImgBytes + G.poff;
    const uint8_t *src = FULL ? img + (size_t)blockIdx.z * simg : lvl + (size_t)ORBX_EDGE * G.pstride + ORBX_EDGE;
    const int ss = FULL ? sstride : G.pstride;
    const int pw4 = (G.w + 2 * ORBX_EDGE + 3) >> 2, rows = G.h + 2 * ORBX_EDGE;
    const int LW = (ORBX_EDGE >> 2) + 1, R0 = (ORBX_EDGE + G.w) >> 2, side = LW + (pw4 - R0);
    if (FULL) {   // 16-byte chunks of the padded rows; interior chunks are one (unaligned) 16-byte load
        const int pc = G.pstride >> 4, totalc = pc * rows;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < totalc; i += gridDim.x * 256) {
            const int py = i / pc, c = i - py * pc;
            const uint8_t *srow = src + (size_t)reflect101c(py - ORBX_EDGE, G.h) * ss;
            const int px = c * 16 - ORBX_EDGE;
            uint4 v;
            if (px >= 0 && px + 15 < G.w) __builtin_memcpy(&v, srow + px, 16);
            else {
                uint32_t t[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    t[u] = (uint32_t)srow[reflect101c(px + 4 * u, G.w)] | ((uint32_t)srow[reflect101c(px + 4 * u + 1, G.w)] << 8) |
                           ((uint32_t)srow[reflect101c(px + 4 * u + 2, G.w)] << 16) | ((uint32_t)srow[reflect101c(px + 4 * u + 3, G.w)] << 24);
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            *(uint4 *)(lvl + (size_t)py * G.pstride + c * 16) = v;
        }
        return;
    }
    const int total = 2 * ORBX_EDGE * pw4 + G.h * side;
    // floor(i / d) == (i * M) >> 24 with M = 2^24 / d + 1 for every i < 2^24 / d (frame items: < 2^14): no integer division
    const unsigned Mp = (1u << 24) / (unsigned)pw4 + 1u, Ms = (1u << 24) / (unsigned)side + 1u;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int py, p4;
        if (i < 2 * ORBX_EDGE * pw4) {
            const int r = (int)(((unsigned long long)(unsigned)i * Mp) >> 24);
            p4 = i - r * pw4;
            py = r < ORBX_EDGE ? r : G.h + r;           // top frame rows, then bottom frame rows
        } else {
            const int j = i - 2 * ORBX_EDGE * pw4, r = (int)(((unsigned long long)(unsigned)j * Ms) >> 24), k = j - r * side;
            py = ORBX_EDGE + r;
            p4 = k < LW ? k : R0 + (k - LW);
        }
        const uint8_t *srow = src + (size_t)reflect101c(py - ORBX_EDGE, G.h) * ss;
        const int px = p4 * 4 - ORBX_EDGE;
        const uint32_t v = (uint32_t)srow[reflect101c(px, G.w)] | ((uint32_t)srow[reflect101c(px + 1, G.w)] << 8) |
                           ((uint32_t)srow[reflect101c(px + 2, G.w)] << 16) | ((uint32_t)srow[reflect101c(px + 3, G.w)] << 24);
        *(uint32_t *)(lvl + (size_t)py * G.pstride + p4 * 4) = v;
    }
}

// ------------------------------------------------------------------------------------
// K2: one wave per 30-px cell (:789-829).  The cell window (cell + 6 px) is staged in LDS
// as one dword per pixel holding the pixel PAIR (p, p+1) in two 16-bit halves, so that the
// FAST-9/16 score of two horizontally adjacent pixels is computed at once with packed 16-bit
// VALU ops from 17 ds_read_b32 (the halves are used as f16 denormals, see below):
//     d[k]   = centre - ring[k]                                (signed, both pixels)
//     dark   = max over the 16 nine-arcs of min d   (3-input minima: windows of 3, then of 9)
//     bright = -min over the arcs of max d
//     S      = max(dark, bright) - 1  if > t_lo = min(iniTh, minTh), else 0
// which is cornerScore<16> of cv::FAST (threshold independent) and its segment test.  Scores
// of the evaluated area (window minus its 3-px frame, exactly cv::FAST's loop bounds) go to
// an LDS tile with a zero halo: the 3x3 strict-max NMS sees zeros outside the evaluated area,
// as cv::FAST never scores them.  Per-cell threshold fallback: {S >= iniTh} if non-empty
// else {S >= minTh}; this equals running cv::FAST(iniTh) and, if empty, cv::FAST(minTh).
// Output: row-major ordered candidates (x | y<<12 | score<<24, relative to minBorder) in
// the cell's slot block + count.
// first global cell number of every level, passed BY VALUE (kernel arguments sit in SGPRs): finding a cell's level
// must not start a chain of dependent loads at the head of every wave
struct CellBases { int v[ORBX_MAX_LEVELS + 1]; };
__device__ __forceinline__ int level_of_cell(const CellBases &cb, int nlevels, int gc) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && gc >= cb.v[i]) ? 1 : 0;
    return l;
}
#define FAST_WAVES 4
#define FAST_STG 8     // window dword pairs per lane fetched in one go (8 x 64 >= a 36x38 window's 456 items)
typedef short short2v __attribute__((ext_vector_type(2)));

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v pk_min3(half2v a, half2v b, half2v c) {
    return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c);
}
__device__ __forceinline__ half2v pk_max3(half2v a, half2v b, half2v c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

// FAST-9/16 score of the pixel pair (px, px+1) of row py of the evaluated area -> score tile
__device__ __forceinline__ void fast_score_pair(const uint32_t *E, int ES, int sh, uint8_t *Sc, int SS, int tlo, int cw,
                                                int py, int px) {
    const uint32_t *q = E + (py + 3) * ES + px + 3 + sh;
    const uint32_t *qm3 = q - 3 * ES, *qm2 = q - 2 * ES, *qm1 = q - ES, *qp1 = q + ES, *qp2 = q + 2 * ES,
                   *qp3 = q + 3 * ES;
    const uint32_t vv = q[0];
    uint32_t rr[16];
    rr[0] = qp3[0];   rr[1] = qp3[1];   rr[2] = qp2[2];   rr[3] = qp1[3];
    rr[4] = q[3];     rr[5] = qm1[3];   rr[6] = qm2[2];   rr[7] = qm3[1];
    rr[8] = qm3[0];   rr[9] = qm3[-1];  rr[10] = qm2[-2]; rr[11] = qm1[-3];
    rr[12] = q[-3];   rr[13] = qp1[-3]; rr[14] = qp2[-2]; rr[15] = qp3[-1];
    // A 16-bit half holding the integer n in [0,255] IS the f16 denormal n*2^-24, so the
    // pixel pairs can be fed to the packed f16 pipe unchanged: differences, 3-input
    // minima/maxima (v_pk_minimum3_f16 / v_pk_maximum3_f16, gfx950) and negation are exact
    // on these values, and a positive result's bit pattern is again the integer.
    const half2v v = __builtin_bit_cast(half2v, vv);
    half2v d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - __builtin_bit_cast(half2v, rr[k]);
    // score + 1 = max(max_k min(arc_k), max_k min(-arc_k)) over the 16 nine-arcs arc_k = d[k..k+8].  Two neighbouring arcs
    // share eight elements: max(min arc_2j, min arc_2j+1) = min(C_j, max(d[2j], d[2j+9])) with C_j = min d[2j+1..2j+8],
    // and C_j is two of the eight 4-windows q[t] = min d[2t+1..2t+4]: 36 packed ops per polarity instead of 40.
    half2v pmn[8], pmx[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        pmn[t] = __builtin_elementwise_minimum(d[2 * t + 1], d[(2 * t + 2) & 15]);
        pmx[t] = __builtin_elementwise_maximum(d[2 * t + 1], d[(2 * t + 2) & 15]);
    }
    half2v qmn[8], qmx[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        qmn[t] = __builtin_elementwise_minimum(pmn[t], pmn[(t + 1) & 7]);
        qmx[t] = __builtin_elementwise_maximum(pmx[t], pmx[(t + 1) & 7]);
    }
    half2v dk[8], bt[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const half2v e0 = d[2 * t], e1 = d[(2 * t + 9) & 15];
        dk[t] = pk_min3(qmn[t], qmn[(t + 2) & 7], __builtin_elementwise_maximum(e0, e1));
        bt[t] = pk_max3(qmx[t], qmx[(t + 2) & 7], __builtin_elementwise_minimum(e0, e1));
    }
    const half2v dark = pk_max3(pk_max3(dk[0], dk[1], dk[2]), pk_max3(dk[3], dk[4], dk[5]), __builtin_elementwise_maximum(dk[6], dk[7]));
    const half2v brt = pk_min3(pk_min3(bt[0], bt[1], bt[2]), pk_min3(bt[3], bt[4], bt[5]), __builtin_elementwise_minimum(bt[6], bt[7]));
    const short2v best = __builtin_bit_cast(short2v, __builtin_elementwise_maximum(dark, -brt));
    const int s0 = best.x, s1 = best.y;
    const uint32_t o0 = s0 > tlo ? (uint32_t)(s0 - 1) : 0u;
    const uint32_t o1 = (s1 > tlo && px + 1 < cw) ? (uint32_t)(s1 - 1) : 0u;
    *(uint16_t *)(Sc + (py + 1) * SS + px + 2) = (uint16_t)(o0 | (o1 << 8));
}

// strict 3x3 maximum test of the pair (px, px+1): scores v0/v1 and keep flags
__device__ __forceinline__ void fast_nms_pair(const uint8_t *Sc, int SS, int cw, int py, int px, bool &k0, bool &k1,
                                              int &v0, int &v1) {
    // pixels px-1 .. px+2 of a row are bytes o .. o+3 of the two aligned dwords at (row + px) & ~3 (SS % 4 == 0, px even:
    // o = 1 or 3): ONE 8-byte LDS read per row and three v_perm with lane-constant selectors
    const int px4 = px & ~3;
    const uint32_t o = (px & 2) ? 3u : 1u, osel = o * 0x00010001u;
    const uint8_t *sc = Sc + (py + 1) * SS + px4;  // 4-byte aligned: pixel px4-2+k is byte k
    half2v l3[3], m3[3], r3[3];
#pragma unroll
    for (int rw = 0; rw < 3; rw++) {
        const uint32_t *w32 = (const uint32_t *)(sc + (rw - 1) * SS);   // two dwords (ds_read2_b32: 4-byte alignment is enough)
        uint2 w;
        w.x = w32[0]; w.y = w32[1];
        l3[rw] = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(w.y, w.x, 0x0c010c00u + osel));  // (px-1, px)
        m3[rw] = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(w.y, w.x, 0x0c020c01u + osel));  // (px, px+1)
        r3[rw] = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(w.y, w.x, 0x0c030c02u + osel));  // (px+1, px+2)
    }
    const half2v nb = pk_max3(pk_max3(l3[0], m3[0], r3[0]), pk_max3(l3[2], m3[2], r3[2]),
                              __builtin_elementwise_maximum(l3[1], r3[1]));
    const short2v gt = __builtin_bit_cast(short2v, m3[1] - nb);  // > 0 iff strictly greater
    const short2v cv = __builtin_bit_cast(short2v, m3[1]);
    v0 = cv.x; v1 = cv.y;
    k0 = gt.x > 0;
    k1 = gt.y > 0 && px + 1 < cw;
}

// ES_T != 0: the tile strides are compile-time constants (pair tile ES_T dwords, score tile ES_T - 8 bytes), so every LDS
// address of the ring / NMS reads is ONE base register + an immediate offset; with run-time strides the score loop spent
// 21 of its 150 VALU instructions per pixel pair on address arithmetic.  ES_T == 0: run-time strides (any configuration).
template <int ES_T>
__global__ __launch_bounds__(64 * FAST_WAVES) void k_fast_cells(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels,
    int totalCells, uint32_t *__restrict__ cellCnt, uint32_t *__restrict__ slots, size_t slotsPerImg,
    int iniTh, int minTh, int ESrt, int SSrt, int tileRows, int ldsPerWave, int phaseLimit, CellBases cb) {
    const int ES = ES_T ? ES_T : ESrt, SS = ES_T ? ES_T - 8 : SSrt;
    extern __shared__ __align__(16) uint8_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);
    const int gc = bx * FAST_WAVES + wave;
    if (gc >= totalCells) return;  // wave-uniform; the kernel uses no block barrier
    const int l = level_of_cell(cb, nlevels, gc);
    const LevelGeom g = geom[l];
    const int c = gc - g.cellBase;
    const int ci = c / g.nCols, cj = c - ci * g.nCols;
    const int maxBX = g.w - ORBX_MINB, maxBY = g.h - ORBX_MINB;
    const int iniY = ORBX_MINB + ci * g.hCell, iniX = ORBX_MINB + cj * g.wCell;
    int maxY = iniY + g.hCell + 6, maxX = iniX + g.wCell + 6;
    uint32_t *cnt = cellCnt + (size_t)b * totalCells + gc;
    if (iniY >= maxBY - 3 || iniX >= maxBX - 6) {  // skipped rows / columns (:794-795,803-804)
        if (lane == 0) *cnt = 0;
        return;
    }
    if (maxY > maxBY) maxY = maxBY;
    if (maxX > maxBX) maxX = maxBX;
    const int tw = maxX - iniX, th = maxY - iniY;  // FAST sub-image
    const int cw = tw - 6, ch = th - 6;            // evaluated area (rows/cols 3 .. dim-4)
    if (cw <= 0 || ch <= 0) {
        if (lane == 0) *cnt = 0;
        return;
    }
    uint32_t *E = (uint32_t *)(smem + (size_t)wave * ldsPerWave);  // pair tile [th][ES] dwords
    uint8_t *Sc = (uint8_t *)(E + (size_t)ES * tileRows);          // score tile [ch+2][SS], pixel (0,0) at +SS+2

    // stage the window: aligned dword loads (pstride % 4 == 0, so every row has the same misalignment)
    const size_t a = (size_t)(ORBX_EDGE + iniY) * g.pstride + ORBX_EDGE + iniX;
    const int sh = (int)(a & 3);
    {
        const uint32_t *src = (const uint32_t *)(pyr + (size_t)b * pyrImgBytes + g.poff + (a - sh));
        const int nd = (sh + tw + 3) >> 2, pstr4 = g.pstride >> 2, items = nd * th;
        // E column index = byte offset inside the aligned row (window column + sh): every item
        // is one aligned 16-byte LDS write, no bounds checks
        // ALL global loads of the window are issued before the first use (FAST_STG x 2 dwords per lane in flight):
        // one memory latency per cell instead of one per 64 items — this phase was a third of the kernel.
        const unsigned M = (1u << 20) / (unsigned)nd + 1u;   // floor(i / nd) == (i * M) >> 20 for i < 2^10, nd <= 2^6
        for (int base = 0; base < items; base += 64 * FAST_STG) {
            uint32_t d0[FAST_STG], d1[FAST_STG];
            int rr[FAST_STG], qq[FAST_STG];
#pragma unroll
            for (int k = 0; k < FAST_STG; k++) {
                const int i = min(base + lane + 64 * k, items - 1);
                rr[k] = (int)(((unsigned)i * M) >> 20);
                qq[k] = i - rr[k] * nd;
                const uint32_t *p = src + (size_t)rr[k] * pstr4 + qq[k];
                d0[k] = p[0]; d1[k] = p[1];
            }
#pragma unroll
            for (int k = 0; k < FAST_STG; k++) {
                if (base + lane + 64 * k < items) {
                    uint4 e;  // bytes b0..b3 of d0 and b4 = first byte of d1 -> pairs (b0,b1) (b1,b2) (b2,b3) (b3,b4)
                    e.x = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c010c00u);
                    e.y = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c020c01u);
                    e.z = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c030c02u);
                    e.w = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c040c03u);
                    *(uint4 *)(E + rr[k] * ES + 4 * qq[k]) = e;
                }
            }
        }
        const int nz = ((ch + 2) * SS) >> 2;  // zero the score tile (halo + odd tail columns)
        for (int i = lane; i < nz; i += 64) ((uint32_t *)Sc)[i] = 0;
    }
    wave_sync();
    if (phaseLimit == 1) return;

    const int tlo = max(min(iniTh, minTh), 0);
    const int pw2 = (cw + 1) >> 1;
    // scores, two pixels per lane.  Cells up to 32 px wide (the rule) use a fixed lane -> (row mod 4,
    // pair) map: no per-iteration index arithmetic; wider cells walk a flat pair index.
    if (pw2 <= 16) {
        const int j = lane & 15, r4 = lane >> 4;
        if (j < pw2)
            for (int py = r4; py < ch; py += 4) fast_score_pair(E, ES, sh, Sc, SS, tlo, cw, py, 2 * j);
    } else {
        const int npairs = pw2 * ch;
        int py = 0, j = lane;
        while (j >= pw2) { j -= pw2; py++; }
        for (int p = lane; p < npairs; p += 64) {
            fast_score_pair(E, ES, sh, Sc, SS, tlo, cw, py, 2 * j);
            j += 64;
            while (j >= pw2) { j -= pw2; py++; }
        }
    }
    wave_sync();
    if (phaseLimit == 2) return;

    // NMS on pixel pairs (same packed-f16 trick: scores are integers 0..254), appending the
    // survivors in row-major order to an LDS list (px | py<<8 | score<<16); E is free again.
    uint32_t *Lst = E;
    bool anyIni = false;
    int nL = 0;
    if (pw2 <= 16) {
        // fixed lane -> (row mod 4, pair) map as in the score phase: column, byte selectors and the keep-mask of the odd
        // pixel are loop invariants, a row step is one address add, and the iniTh test is ONE ballot after the loop
        const int j = lane & 15, px = 2 * j, r4 = lane >> 4;
        const bool colOk = j < pw2, k1ok = px + 1 < cw;
        const uint32_t osel = ((px & 2) ? 3u : 1u) * 0x00010001u;
        const uint32_t selL = 0x0c010c00u + osel, selM = 0x0c020c01u + osel, selR = 0x0c030c02u + osel;
        const uint8_t *col = Sc + (px & ~3);   // pixels px-1 .. px+2 of a row = bytes of the two aligned dwords here
        bool ini = false;
        for (int r0 = 0; r0 < ch; r0 += 4) {
            const int py = r0 + r4;
            const bool act = colOk && py < ch;
            const uint32_t *top = (const uint32_t *)(col + min(py, ch - 1) * SS);   // tile rows py, py+1, py+2
            const uint32_t a0 = top[0], a1 = top[1], b0 = top[SS / 4], b1 = top[SS / 4 + 1], c0 = top[SS / 2],
                           c1 = top[SS / 2 + 1];
            const half2v lt = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(a1, a0, selL)),
                         mt = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(a1, a0, selM)),
                         rt = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(a1, a0, selR)),
                         lm = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(b1, b0, selL)),
                         mm = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(b1, b0, selM)),
                         rm = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(b1, b0, selR)),
                         lb = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(c1, c0, selL)),
                         mb = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(c1, c0, selM)),
                         rb = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(c1, c0, selR));
            const half2v nb = pk_max3(pk_max3(lt, mt, rt), pk_max3(lb, mb, rb), __builtin_elementwise_maximum(lm, rm));
            const short2v gt = __builtin_bit_cast(short2v, mm - nb);   // > 0 iff strictly greater than all 8 neighbours
            const uint32_t cv = __builtin_bit_cast(uint32_t, mm);
            const int v0 = (int)(cv & 0xFFFFu), v1 = (int)(cv >> 16);
            const bool k0 = act && gt.x > 0, k1 = act && k1ok && gt.y > 0;
            ini |= (k0 && v0 >= iniTh) || (k1 && v1 >= iniTh);
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
            const int pos = nL + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u)) +
                            (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
            const uint32_t w0 = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)v0 << 16);
            if (k0) Lst[pos] = w0;
            if (k1) Lst[pos + (k0 ? 1 : 0)] = (uint32_t)(px + 1) | ((uint32_t)py << 8) | ((uint32_t)v1 << 16);
            nL += __popcll(m0) + __popcll(m1);
        }
        anyIni = __ballot(ini) != 0ull;
    } else {
        const int npairs = pw2 * ch;
        int py = 0, j = lane;
        while (j >= pw2) { j -= pw2; py++; }
        for (int base = 0; base < npairs; base += 64) {
            bool k0 = false, k1 = false;
            int v0 = 0, v1 = 0;
            const int px = 2 * j;
            if (base + lane < npairs) fast_nms_pair(Sc, SS, cw, py, px, k0, k1, v0, v1);
            anyIni |= (__ballot((k0 && v0 >= iniTh) || (k1 && v1 >= iniTh)) != 0ull);
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1), lt = (1ull << lane) - 1ull;
            int pos = nL + __popcll(m0 & lt) + __popcll(m1 & lt);
            if (k0) Lst[pos++] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)v0 << 16);
            if (k1) Lst[pos] = (uint32_t)(px + 1) | ((uint32_t)py << 8) | ((uint32_t)v1 << 16);
            nL += __popcll(m0) + __popcll(m1);
            j += 64;
            while (j >= pw2) { j -= pw2; py++; }
        }
    }
    wave_sync();
    if (phaseLimit == 3) return;

    // per-cell threshold fallback (:809-816) + ordered emission
    const int thr = anyIni ? iniTh : minTh;
    uint32_t *out = slots + (size_t)b * slotsPerImg + g.slotOff + (size_t)c * g.capc;
    int total = 0;
    for (int base = 0; base < nL; base += 64) {
        const int i = base + lane;
        uint32_t e = 0;
        bool emit = false;
        if (i < nL) {
            e = Lst[i];
            emit = (int)(e >> 16) >= thr;
        }
        const unsigned long long m = __ballot(emit);
        if (emit) {
            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < g.capc)
                out[pos] = (uint32_t)((e & 0xFF) + 3 + cj * g.wCell) | ((uint32_t)(((e >> 8) & 0xFF) + 3 + ci * g.hCell) << 12) |
                           ((e >> 16) << 24);
        }
        total += __popcll(m);
    }
    if (lane == 0) *cnt = (uint32_t)min(total, g.capc);
}

// ------------------------------------------------------------------------------------
// K2b: per (level, image): exclusive scan of the cell counts = offsets of the cell lists in
// the ordered concatenation (vToDistributeKeys order, :789-828).
__global__ __launch_bounds__(256) void k_cell_scan(const LevelGeom *__restrict__ geom, int nlevels, int totalCells,
                                                   const uint32_t *__restrict__ cellCnt, uint32_t *__restrict__ cellOff,
                                                   int32_t *__restrict__ candCnt) {
    __shared__ int wsum[4];
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LevelGeom g = geom[l];
    const uint32_t *cc = cellCnt + (size_t)b * totalCells + g.cellBase;
    uint32_t *co = cellOff + (size_t)b * totalCells + g.cellBase;
    const int chunk = (g.ncells + 255) / 256;
    const int beg = min(tid * chunk, g.ncells), end = min(beg + chunk, g.ncells);
    int s = 0;
    for (int c = beg; c < end; c++) s += (int)cc[c];
    const int inc = wave_incl_scan_i32(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = inc - s, tot = 0;
    for (int w = 0; w < 4; w++) {
        if (w < wave) off += wsum[w];
        tot += wsum[w];
    }
    for (int c = beg; c < end; c++) {
        co[c] = (uint32_t)off;
        off += (int)cc[c];
    }
    if (tid == 0) candCnt[b * nlevels + l] = tot;
}

// K2c: sixteen lanes per cell copy its candidate list to its place in the level's key array.
#define GATHER_CELLS_PER_BLOCK 16
__global__ __launch_bounds__(256) void k_gather(const LevelGeom *__restrict__ geom, int nlevels, int totalCells,
                                                const uint32_t *__restrict__ cellCnt, const uint32_t *__restrict__ cellOff,
                                                const uint32_t *__restrict__ slots, size_t slotsPerImg,
                                                uint32_t *__restrict__ cand, size_t keysPerImg, CellBases cb) {
    const int sub = threadIdx.x & 15;
    int bx, b;
    xcd_block_map(bx, b);
    const int gc = bx * GATHER_CELLS_PER_BLOCK + (threadIdx.x >> 4);
    if (gc >= totalCells) return;
    const int l = level_of_cell(cb, nlevels, gc);
    const int c = gc - geom[l].cellBase, capc = geom[l].capc;
    const int cn = (int)cellCnt[(size_t)b * totalCells + gc], off = (int)cellOff[(size_t)b * totalCells + gc];
    const uint32_t *src = slots + (size_t)b * slotsPerImg + geom[l].slotOff + (size_t)c * capc;
    uint32_t *dst = cand + (size_t)b * keysPerImg + geom[l].keyOff + off;
    for (int j0 = 0; j0 < cn; j0 += 64) {   // four loads in flight per lane before the stores
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = src[min(j0 + sub + 16 * k, cn - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) if (j0 + sub + 16 * k < cn) dst[j0 + sub + 16 * k] = v[k];
    }
}

// K3: DistributeOctTree (:539-763), one workgroup per (level, image).
//
// Parallel restatement of the reference's std::list surgery (validated against the literal
// CPU oracle).  Facts it relies on:
//  * every insertion is push_front, so the list is always ordered by DESCENDING creation
//    time; the node array here IS the list (index 0 = front);
//  * a pass visits expandable nodes (created in the previous pass, >1 key) in an order O,
//    creates the non-empty children n1..n4 of each and erases the parent:
//       new list = reverse(created sequence) ++ (old list minus the split parents);
//  * phase 1 (:594-665): O = list order, all expandable nodes are split;
//    phase 2 (:673-737): O = sort by (size desc, tie), split until size >= N.
//    Tie-break of equal sizes: the reference compares heap pointers (:684); this build
//    fixes "later-created first" == smaller list index first.
//  * a key's child is a pure function of (x, y, parent box): keys never move in memory,
//    only their 16-bit node index is rewritten.
#define OCT_T 512   // 1024-thread workgroups are resident one per CU only; 512 packs 2x better at batch 128 and costs 6 us on a single frame

struct OctLds {
    short4 *box[2];
    uint32_t *cnt[2];   // bit31 = fresh (created in the previous pass)
    uint32_t *hist;     // [4*cap] children key counts, also scratch
    uint16_t *childIdx; // [4*cap]
    uint16_t *survIdx;  // [cap]
    uint16_t *xlist;    // [cap] expandable nodes in visiting order
    int *pn, *pg;       // [cap] inclusive prefix of created children / gain by rank
    uint8_t *split;     // [cap]
    unsigned long long *skey;  // [pow2(cap)]
};

__device__ __forceinline__ int child_of(int x, int y, short4 bx) {
    const int mx = bx.x + ((bx.y - bx.x + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)   (:483)
    const int my = bx.z + ((bx.w - bx.z + 1) >> 1);  // UL.y + ceil((BR.y-UL.y)/2)   (:484)
    return (x < mx ? 0 : 1) | (y < my ? 0 : 2);      // n1,n2,n3,n4                   (:513-525)
}
__device__ __forceinline__ short4 child_box(short4 bx, int q) {
    const short mx = (short)(bx.x + ((bx.y - bx.x + 1) >> 1));
    const short my = (short)(bx.z + ((bx.w - bx.z + 1) >> 1));
    short4 r;
    r.x = (q & 1) ? mx : bx.x;
    r.y = (q & 1) ? bx.y : mx;
    r.z = (q & 2) ? my : bx.z;
    r.w = (q & 2) ? bx.w : my;
    return r;
}

// four consecutive keys / node indices of one thread (16-B / 8-B accesses; the level's key block is 16-B aligned)
__device__ __forceinline__ void load_keys4(const uint32_t *keys, int i0, int n, uint32_t key[4]) {
    if (i0 + 3 < n) {
        const uint4 v = *(const uint4 *)(keys + i0);
        key[0] = v.x; key[1] = v.y; key[2] = v.z; key[3] = v.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = i0 + u < n ? keys[i0 + u] : 0u;
    }
}
__device__ __forceinline__ void load_nof4(const uint16_t *nof, int i0, int n, int kk[4]) {
    if (i0 + 3 < n) {
        const ushort4 v = *(const ushort4 *)(nof + i0);
        kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) kk[u] = i0 + u < n ? (int)nof[i0 + u] : 0;
    }
}
__device__ __forceinline__ void store_nof4(uint16_t *nof, int i0, int n, const int kk[4]) {
    if (i0 + 3 < n) {
        ushort4 v;
        v.x = (unsigned short)kk[0]; v.y = (unsigned short)kk[1]; v.z = (unsigned short)kk[2]; v.w = (unsigned short)kk[3];
        *(ushort4 *)(nof + i0) = v;
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) if (i0 + u < n) nof[i0 + u] = (uint16_t)kk[u];
    }
}

// exclusive scan of one int per thread across the block; returns the exclusive prefix and
// writes the block total to *total (all threads).  wsum: LDS int[OCT_T/64 + 1].
__device__ __forceinline__ int block_scan_excl(int v, int *wsum, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // protect wsum reuse
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < OCT_T / 64; w++) {
        const int s = wsum[w];
        if (w < wave) woff += s;
        tot += s;
    }
    *total = tot;
    return woff + inc - v;
}

// in-place exclusive scan of an LDS int array a[0..m) (m arbitrary); returns total
__device__ int array_scan_excl(int *a, int m, int *wsum) {
    const int chunk = (m + OCT_T - 1) / OCT_T;
    const int beg = min((int)threadIdx.x * chunk, m), end = min(beg + chunk, m);
    int s = 0;
    for (int i = beg; i < end; i++) s += a[i];
    int total;
    int off = block_scan_excl(s, wsum, &total);
    for (int i = beg; i < end; i++) {
        const int t = a[i];
        a[i] = off;
        off += t;
    }
    __syncthreads();
    return total;
}

// K3 (main path): DistributeOctTree from a COUNT PYRAMID.  A key's path through the quad-tree
// is a pure function of its coordinates (root by tabulated x/hX, then ceil-halved boxes), so
// ONE sweep over the keys histograms them at a fixed depth Dm and the key count of every node
// of every shallower depth follows by summing children.  All the list surgery of the passes
// (which nodes are split, in which order, where the break falls) then runs on node counts only
// — no further key sweep — and one final sweep walks every key down to its leaf to elect the
// best response per node.  Two sweeps over the keys instead of one per pass.  If a pass would
// need counts deeper than Dm (sparse, clustered candidates) the level is flagged and redone by
// the sweep-per-pass kernel k_octree below: results never depend on the path taken.
__device__ __forceinline__ uint32_t pyr_count(const uint32_t *pyr, int nIni, int Dm, int d, uint32_t c) {
    const uint32_t off = (uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u);
    if (d == Dm) return (pyr[off + (c >> 1)] >> (16 * (c & 1))) & 0xFFFFu;
    return pyr[off + c];
}

__global__ __launch_bounds__(OCT_T) void k_octree_pyr(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand, size_t keysPerImg,
    const int32_t *__restrict__ candCnt, uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int pyrWords, int32_t *__restrict__ fallback) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int l = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;  // level-major: large levels start first
    const LevelGeom g = geom[l];
    const int Dm = g.pyrDepth, nIni = g.nIni, N = g.N;
    uint8_t *sp = smem;
    unsigned long long *skey = (unsigned long long *)sp; sp += sizeof(unsigned long long) * pow2cap;
    uint32_t *cntA = (uint32_t *)sp; sp += 4 * 2 * capMax;   // [2][cap] key count, bit31 = fresh
    uint32_t *nidA = (uint32_t *)sp; sp += 4 * 2 * capMax;   // [2][cap] depth << 28 | cell
    uint32_t *hist = (uint32_t *)sp; sp += 4 * 4 * capMax;   // children counts of list node k; later best[]
    int *pn = (int *)sp; sp += 4 * capMax;
    uint32_t *pyr = (uint32_t *)sp; sp += 4 * (size_t)pyrWords;
    uint16_t *xlist = (uint16_t *)sp; sp += 2 * capMax;
    uint8_t *split = sp; sp += capMax;
    __shared__ int sh_L, sh_Lnew, sh_finish, sh_phase, sh_abort;

    const uint32_t *keys = cand + (size_t)b * keysPerImg + g.keyOff;
    const int n = candCnt[b * nlevels + l];
    const int32_t *xPath = tab + g.xPathOff, *yPath = tab + g.yPathOff;
    const uint32_t offDeep = (uint32_t)nIni * (((1u << (2 * Dm)) - 1u) / 3u);

    // ---- 1. histogram of the keys at depth Dm (two 16-bit counters per word)
    for (int i = tid; i < pyrWords; i += OCT_T) pyr[i] = 0;
    if (tid == 0) sh_abort = 0;
    __syncthreads();
    for (int i0 = 4 * tid; i0 < n; i0 += 4 * OCT_T) {
        uint32_t key[4], c[4];
        load_keys4(keys, i0, n, key);
#pragma unroll
        for (int u = 0; u < 4; u++)   // unconditional (a missing key is 0): all 8 lookups in flight at once
            c[u] = (uint32_t)xPath[key[u] & 0xFFF] | (uint32_t)yPath[(key[u] >> 12) & 0xFFF];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + u < n) atomicAdd(&pyr[offDeep + (c[u] >> 1)], 1u << (16 * (c[u] & 1)));
    }
    __syncthreads();
    // ---- 2. counts of the shallower depths
    for (int d = Dm - 1; d >= 0; d--) {
        const uint32_t off = (uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u);
        const int ne = nIni << (2 * d);
        for (int e = tid; e < ne; e += OCT_T) {
            uint32_t s = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) s += pyr_count(pyr, nIni, Dm, d + 1, 4u * e + q);
            pyr[off + e] = s;
        }
        __syncthreads();
    }
    // ---- 3. root nodes (:543-592)
    if (tid == 0) {
        int L0 = 0;
        for (int r = 0; r < nIni; r++) {
            const uint32_t c = pyr[r];
            if (c > 0) { cntA[L0] = c | 0x80000000u; nidA[L0] = (uint32_t)r; L0++; }
        }
        sh_L = L0;
    }
    __syncthreads();
    int L = sh_L, cur = 0, phase = 1;

    // ---- 4. passes: list bookkeeping on node counts only, by wave 0
    while (true) {
        uint32_t *cnt = cntA + cur * capMax, *ncnt = cntA + (cur ^ 1) * capMax;
        uint32_t *nid = nidA + cur * capMax, *nnid = nidA + (cur ^ 1) * capMax;
        if (tid < 64) {
            const int lane = tid;
            // children counts of every expandable node from the pyramid
            bool deep = false;
            for (int k = lane; k < L; k += 64) {
                const uint32_t cv = cnt[k];
                if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) {
                    const int d = (int)(nid[k] >> 28);
                    const uint32_t c = nid[k] & 0x0FFFFFFFu;
                    if (d + 1 > Dm) deep = true;
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) hist[4 * k + q] = pyr_count(pyr, nIni, Dm, d + 1, 4u * c + q);
                    }
                }
            }
            if (__ballot(deep)) { if (lane == 0) sh_abort = 1; }
            wave_sync();
            if (!sh_abort) {
                // visiting order of the expandable (fresh, >1 key) nodes
                int E;
                if (phase == 1) {   // list order
                    const int chunk = (L + 63) >> 6;
                    const int beg = min(lane * chunk, L), end = min(beg + chunk, L);
                    int s = 0;
                    for (int k = beg; k < end; k++) {
                        const uint32_t cv = cnt[k];
                        s += ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) ? 1 : 0;
                    }
                    const int inc = wave_incl_scan_i32(s);
                    E = __builtin_amdgcn_readlane(inc, 63);
                    int off = inc - s;
                    for (int k = beg; k < end; k++) {
                        const uint32_t cv = cnt[k];
                        if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) xlist[off++] = (uint16_t)k;
                    }
                } else {            // (size desc, later-created first): bitonic sort of (~size, list index)
                    int P = 1;
                    while (P < L) P <<= 1;
                    int e = 0;
                    for (int k = lane; k < P; k += 64) {
                        unsigned long long key = ~0ull;
                        if (k < L) {
                            const uint32_t cv = cnt[k];
                            if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) {
                                key = ((unsigned long long)(0x7FFFFFFFu - (cv & 0x7FFFFFFFu)) << 32) | (unsigned)k;
                                e++;
                            }
                        }
                        skey[k] = key;
                    }
                    wave_sync();
                    for (int kk2 = 2; kk2 <= P; kk2 <<= 1)
                        for (int j = kk2 >> 1; j > 0; j >>= 1) {
                            for (int i = lane; i < P; i += 64) {
                                const int ixj = i ^ j;
                                if (ixj > i) {
                                    const unsigned long long a = skey[i], c2 = skey[ixj];
                                    if ((a > c2) == ((i & kk2) == 0)) { skey[i] = c2; skey[ixj] = a; }
                                }
                            }
                            wave_sync();
                        }
                    e = wave_total_i32(e);
                    E = e;
                    for (int k = lane; k < E; k += 64) xlist[k] = (uint16_t)(skey[k] & 0xFFFFu);
                }
                wave_sync();
                // children created per rank -> exclusive prefix by rank; number of parents split
                int Sp = E, C = 0;
                {
                    const int chunk = (E + 63) >> 6;
                    const int beg = min(lane * chunk, E), end = min(beg + chunk, E);
                    int s = 0;
                    for (int r = beg; r < end; r++) {
                        const int k = xlist[r];
                        s += (hist[4 * k] > 0) + (hist[4 * k + 1] > 0) + (hist[4 * k + 2] > 0) + (hist[4 * k + 3] > 0);
                    }
                    const int inc = wave_incl_scan_i32(s);
                    int off = inc - s;
                    int hit = 0x7FFFFFFF;
                    for (int r = beg; r < end; r++) {
                        const int k = xlist[r];
                        const int nz = (hist[4 * k] > 0) + (hist[4 * k + 1] > 0) + (hist[4 * k + 2] > 0) + (hist[4 * k + 3] > 0);
                        pn[r] = off;
                        if (phase == 2) {
                            const int after = L + off + nz - (r + 1), before = L + off - r;
                            if (after >= N && before < N) hit = r + 1;   // the break at :730-731
                        }
                        off += nz;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) hit = min(hit, __shfl_xor(hit, o));
                    if (phase == 2 && hit != 0x7FFFFFFF) Sp = hit;
                    wave_sync();
                    if (Sp > 0) {
                        const int k = xlist[Sp - 1];
                        C = pn[Sp - 1] + (hist[4 * k] > 0) + (hist[4 * k + 1] > 0) + (hist[4 * k + 2] > 0) + (hist[4 * k + 3] > 0);
                    }
                }
                const int Lnew = L - Sp + C;
                for (int k = lane; k < L; k += 64) split[k] = 0;
                wave_sync();
                // create children: creation sequence s -> list index C-1-s (every insertion is push_front)
                int nexp = 0;
                for (int r = lane; r < Sp; r += 64) {
                    const int k = xlist[r];
                    split[k] = 1;
                    int s = pn[r];
                    const uint32_t pd = nid[k] >> 28, pc = nid[k] & 0x0FFFFFFFu;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t hc = hist[4 * k + q];
                        if (hc > 0) {
                            const int ni = C - 1 - s;
                            ncnt[ni] = hc | 0x80000000u;
                            nnid[ni] = ((pd + 1) << 28) | (4u * pc + q);
                            if (hc > 1) nexp++;
                            s++;
                        }
                    }
                }
                nexp = wave_total_i32(nexp);
                wave_sync();
                {   // survivors keep their relative order behind the new nodes
                    const int chunk = (L + 63) >> 6;
                    const int beg = min(lane * chunk, L), end = min(beg + chunk, L);
                    int s = 0;
                    for (int k = beg; k < end; k++) s += split[k] ? 0 : 1;
                    const int inc = wave_incl_scan_i32(s);
                    int off = inc - s;
                    for (int k = beg; k < end; k++)
                        if (!split[k]) {
                            const int ni = C + off++;
                            ncnt[ni] = cnt[k] & 0x7FFFFFFFu;  // no longer fresh
                            nnid[ni] = nid[k];
                        }
                }
                const bool fin = (Lnew >= N || Lnew == L);   // :669-672, :733-734
                if (lane == 0) {
                    sh_Lnew = Lnew;
                    sh_finish = fin ? 1 : 0;
                    sh_phase = (!fin && phase == 1 && Lnew + 3 * nexp > N) ? 2 : phase;
                }
            }
        }
        __syncthreads();
        if (sh_abort) {   // counts deeper than the pyramid are needed: hand the level to k_octree
            if (tid == 0) fallback[b * nlevels + l] = 1;
            return;
        }
        L = sh_Lnew;
        phase = sh_phase;
        cur ^= 1;
        if (sh_finish) break;
        __syncthreads();
    }

    // ---- 5. leaf map (depth, cell) -> list index, in place of the counts
    {
        const uint32_t *nid = nidA + cur * capMax;
        __syncthreads();
        for (int i = tid; i < pyrWords; i += OCT_T) pyr[i] = 0xFFFFFFFFu;
        for (int i = tid; i < L; i += OCT_T) hist[i] = 0;  // best[]
        __syncthreads();
        for (int k = tid; k < L; k += OCT_T) {
            const int d = (int)(nid[k] >> 28);
            const uint32_t c = nid[k] & 0x0FFFFFFFu;
            const uint32_t off = (uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u);
            if (d == Dm) ((uint16_t *)(pyr + off))[c] = (uint16_t)k;
            else pyr[off + c] = (uint32_t)k;
        }
        __syncthreads();
    }
    // ---- 6. every key walks down to its leaf; best key of the node, first maximum wins (:744-760)
    for (int i0 = 4 * tid; i0 < n; i0 += 4 * OCT_T) {
        uint32_t key[4], cd[4], node[4];
        load_keys4(keys, i0, n, key);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            cd[u] = (uint32_t)xPath[key[u] & 0xFFF] | (uint32_t)yPath[(key[u] >> 12) & 0xFFF];
            node[u] = 0xFFFFFFFFu;
        }
        // the leaves partition the region: exactly one cell on a key's path is in the map, so the depths
        // are probed independently (4 keys x 1 depth in flight) instead of as a dependent descent
        for (int d = 0; d < Dm; d++) {
            const uint32_t off = (uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u);
            const int sh = 2 * (Dm - d);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t v = pyr[off + (cd[u] >> sh)];
                node[u] = v != 0xFFFFFFFFu ? v : node[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t v = ((const uint16_t *)(pyr + offDeep))[cd[u]];
            node[u] = v != 0xFFFFu ? v : node[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + u < n && node[u] != 0xFFFFFFFFu)
                atomicMax(&hist[node[u]], ((key[u] >> 24) << 20) | (0xFFFFFu - (uint32_t)(i0 + u)));
    }
    __syncthreads();
    // ---- 7. output in list order
    uint32_t *okp = lvlKp + (size_t)b * lvlKpCap + g.lvlKpOff;
    const int Lout = min(L, g.nodeCap);
    for (int k = tid; k < Lout; k += OCT_T) okp[k] = keys[0xFFFFFu - (hist[k] & 0xFFFFFu)];
    if (tid == 0) { lvlCnt[b * nlevels + l] = Lout; fallback[b * nlevels + l] = 0; }
}

// K3 (fallback): one sweep over the keys per pass; runs only for levels k_octree_pyr flagged
__global__ __launch_bounds__(OCT_T) void k_octree(
    const LevelGeom *__restrict__ geom, int nlevels, int totalCells, const uint32_t *__restrict__ cellCnt,
    const uint32_t *__restrict__ slots, size_t slotsPerImg, uint32_t *__restrict__ cand,
    uint16_t *__restrict__ nodeOf, size_t keysPerImg, const int32_t *__restrict__ candCnt,
    uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int scratchInts, int dbgStop,
    const int32_t *__restrict__ fallback) {
    extern __shared__ __align__(16) uint8_t smem[];
    // level-major block order: the large levels start first and the small ones fill the gaps
    const int l = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    if (fallback && !fallback[b * nlevels + l]) return;  // done by k_octree_pyr
    const LevelGeom g = geom[l];
    // ---- carve LDS
    uint8_t *sp = smem;
    OctLds S;
    S.skey = (unsigned long long *)sp; sp += sizeof(unsigned long long) * pow2cap;
    S.box[0] = (short4 *)sp; sp += sizeof(short4) * capMax;
    S.box[1] = (short4 *)sp; sp += sizeof(short4) * capMax;
    S.cnt[0] = (uint32_t *)sp; sp += 4 * capMax;
    S.cnt[1] = (uint32_t *)sp; sp += 4 * capMax;
    S.hist = (uint32_t *)sp; sp += 4 * (size_t)scratchInts;  // >= max(4*cap, ncells+1)
    S.pn = (int *)sp; sp += 4 * capMax;
    S.pg = (int *)sp; sp += 4 * capMax;
    S.childIdx = (uint16_t *)sp; sp += 2 * 4 * capMax;
    S.survIdx = (uint16_t *)sp; sp += 2 * capMax;
    S.xlist = (uint16_t *)sp; sp += 2 * capMax;
    S.split = sp; sp += capMax;
    __shared__ int rootCnt[ORBX_MAX_ROOTS], rootMap[ORBX_MAX_ROOTS];

    uint32_t *keys = cand + (size_t)b * keysPerImg + g.keyOff;
    uint16_t *nof = nodeOf + (size_t)b * keysPerImg + g.keyOff;

    // ---- A/B. keys were gathered in vToDistributeKeys order by k_cell_scan + k_gather
    const int n = candCnt[b * nlevels + l];
    const uint8_t *rootOf = (const uint8_t *)tab + g.rootTabOff;
    if (tid < ORBX_MAX_ROOTS) rootCnt[tid] = 0;
    __syncthreads();
    {   // keys per root (:569): per-wave ballot counts, one LDS atomic per wave and root
        const int lane = tid & 63;
        for (int b0 = 0; b0 < n; b0 += 4 * OCT_T) {
            const int i0 = b0 + 4 * tid;
            uint32_t key[4];
            load_keys4(keys, i0, n, key);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = i0 + u < n ? (int)rootOf[key[u] & 0xFFF] : -1;
                for (int q = 0; q < g.nIni; q++) {
                    const unsigned long long m = __ballot(r == q);
                    if (lane == 0 && m) atomicAdd(&rootCnt[q], (int)__popcll(m));
                }
            }
        }
    }
    __syncthreads();
    if (dbgStop == 2) return;
    __shared__ int sh_L;
    if (tid == 0) {
        int L0 = 0;
        for (int r = 0; r < g.nIni; r++) {
            if (rootCnt[r] > 0) {
                short4 bx;
                bx.x = (short)tab[g.rootBoxOff + r];
                bx.y = (short)tab[g.rootBoxOff + r + 1];
                bx.z = 0;
                bx.w = (short)g.regH;
                S.box[0][L0] = bx;
                S.cnt[0][L0] = (uint32_t)rootCnt[r] | 0x80000000u;
                rootMap[r] = L0++;
            }
        }
        sh_L = L0;
    }
    __syncthreads();
    int L = sh_L;
    int cur = 0, phase = 1;
    const int N = g.N;
    for (int i = tid; i < 4 * L; i += OCT_T) S.hist[i] = 0;  // coff is dead from here on
    __syncthreads();
    // first sweep: list index of the root + children histogram of the expandable roots
    for (int b0 = 0; b0 < n; b0 += 4 * OCT_T) {
        const int i0 = b0 + 4 * tid;
        uint32_t key[4];
        int kk[4];
        load_keys4(keys, i0, n, key);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = rootMap[rootOf[key[u] & 0xFFF]];
            kk[u] = k;
            int bin = -1;
            if (i0 + u < n) {
                const uint32_t cv = S.cnt[0][k];
                if ((cv & 0x7FFFFFFFu) > 1) bin = 4 * k + child_of(key[u] & 0xFFF, (key[u] >> 12) & 0xFFF, S.box[0][k]);
            }
            if (bin >= 0) atomicAdd(&S.hist[bin], 1u);
        }
        store_nof4(nof, i0, n, kk);
    }
    __syncthreads();

    if (dbgStop == 3) return;
    int npass = 0;
    // ---- C. passes.  On entry S.hist holds the children key counts of every expandable node.
    // The list bookkeeping of a pass touches only O(list size) entries: it is done by wave 0
    // alone with wave-synchronous LDS hand-offs (no workgroup barriers); the other waves wait.
    __shared__ int sh_Lnew, sh_finish, sh_phase;
    while (true) {
        // plain offsets (no runtime-indexed pointer arrays): keeps the accesses in the LDS address space
        short4 *box = S.box[0] + cur * capMax, *nbox = S.box[0] + (cur ^ 1) * capMax;
        uint32_t *cnt = S.cnt[0] + cur * capMax, *ncnt = S.cnt[0] + (cur ^ 1) * capMax;
        if (tid < 64) {
            const int lane = tid;
            // 1. visiting order of the expandable (fresh, >1 key) nodes
            int E;
            if (phase == 1) {   // list order
                const int chunk = (L + 63) >> 6;
                const int beg = min(lane * chunk, L), end = min(beg + chunk, L);
                int s = 0;
                for (int k = beg; k < end; k++) {
                    const uint32_t cv = cnt[k];
                    s += ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) ? 1 : 0;
                }
                const int inc = wave_incl_scan_i32(s);
                E = __builtin_amdgcn_readlane(inc, 63);
                int off = inc - s;
                for (int k = beg; k < end; k++) {
                    const uint32_t cv = cnt[k];
                    if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) S.xlist[off++] = (uint16_t)k;
                }
            } else {            // (size desc, later-created first): bitonic sort of (~size, list index)
                int P = 1;
                while (P < L) P <<= 1;
                int e = 0;
                for (int k = lane; k < P; k += 64) {
                    unsigned long long key = ~0ull;
                    if (k < L) {
                        const uint32_t cv = cnt[k];
                        if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) {
                            key = ((unsigned long long)(0x7FFFFFFFu - (cv & 0x7FFFFFFFu)) << 32) | (unsigned)k;
                            e++;
                        }
                    }
                    S.skey[k] = key;
                }
                wave_sync();
                for (int kk2 = 2; kk2 <= P; kk2 <<= 1)
                    for (int j = kk2 >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < P; i += 64) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const unsigned long long a = S.skey[i], c2 = S.skey[ixj];
                                if ((a > c2) == ((i & kk2) == 0)) { S.skey[i] = c2; S.skey[ixj] = a; }
                            }
                        }
                        wave_sync();
                    }
                e = wave_total_i32(e);
                E = e;
                for (int k = lane; k < E; k += 64) S.xlist[k] = (uint16_t)(S.skey[k] & 0xFFFFu);
            }
            wave_sync();
            // 2. children created per rank -> exclusive prefix by rank (in S.pn)
            int Sp = E, C = 0;
            {
                const int chunk = (E + 63) >> 6;
                const int beg = min(lane * chunk, E), end = min(beg + chunk, E);
                int s = 0;
                for (int r = beg; r < end; r++) {
                    const int k = S.xlist[r];
                    s += (S.hist[4 * k] > 0) + (S.hist[4 * k + 1] > 0) + (S.hist[4 * k + 2] > 0) + (S.hist[4 * k + 3] > 0);
                }
                const int inc = wave_incl_scan_i32(s);
                int off = inc - s;
                // 3. number of parents split: phase 2 stops at the first rank that reaches N (:730-731)
                int hit = 0x7FFFFFFF;
                for (int r = beg; r < end; r++) {
                    const int k = S.xlist[r];
                    const int nz = (S.hist[4 * k] > 0) + (S.hist[4 * k + 1] > 0) + (S.hist[4 * k + 2] > 0) + (S.hist[4 * k + 3] > 0);
                    S.pn[r] = off;
                    if (phase == 2) {
                        const int after = L + off + nz - (r + 1), before = L + off - r;
                        if (after >= N && before < N) hit = r + 1;
                    }
                    off += nz;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) hit = min(hit, __shfl_xor(hit, o));
                if (phase == 2 && hit != 0x7FFFFFFF) Sp = hit;
                wave_sync();
                if (Sp > 0) {
                    const int k = S.xlist[Sp - 1];
                    C = S.pn[Sp - 1] + (S.hist[4 * k] > 0) + (S.hist[4 * k + 1] > 0) + (S.hist[4 * k + 2] > 0) + (S.hist[4 * k + 3] > 0);
                }
            }
            const int Lnew = L - Sp + C;
            for (int k = lane; k < L; k += 64) S.split[k] = 0;
            wave_sync();
            // 4. create children: creation sequence s -> list index C-1-s (every insertion is push_front)
            int nexp = 0;
            for (int r = lane; r < Sp; r += 64) {
                const int k = S.xlist[r];
                S.split[k] = 1;
                int s = S.pn[r];
                const short4 pb = box[k];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t hc = S.hist[4 * k + q];
                    if (hc > 0) {
                        const int ni = C - 1 - s;
                        nbox[ni] = child_box(pb, q);
                        ncnt[ni] = hc | 0x80000000u;
                        S.childIdx[4 * k + q] = (uint16_t)ni;
                        if (hc > 1) nexp++;
                        s++;
                    }
                }
            }
            nexp = wave_total_i32(nexp);
            wave_sync();
            {   // survivors keep their relative order behind the new nodes
                const int chunk = (L + 63) >> 6;
                const int beg = min(lane * chunk, L), end = min(beg + chunk, L);
                int s = 0;
                for (int k = beg; k < end; k++) s += S.split[k] ? 0 : 1;
                const int inc = wave_incl_scan_i32(s);
                int off = inc - s;
                for (int k = beg; k < end; k++)
                    if (!S.split[k]) {
                        const int ni = C + off++;
                        nbox[ni] = box[k];
                        ncnt[ni] = cnt[k] & 0x7FFFFFFFu;  // no longer fresh
                        S.survIdx[k] = (uint16_t)ni;
                    }
            }
            // 5. termination (:669-672, :733-734)
            const bool fin = (Lnew >= N || Lnew == L);
            if (lane == 0) {
                sh_Lnew = Lnew;
                sh_finish = fin ? 1 : 0;
                sh_phase = (!fin && phase == 1 && Lnew + 3 * nexp > N) ? 2 : phase;
            }
        }
        __syncthreads();
        const int Lnew = sh_Lnew;
        const bool finish = sh_finish != 0;
        phase = sh_phase;
        // 6. one sweep over the keys: new node index + (children histogram of the next pass |
        //    best key of every node, first maximum wins (:744-760))
        const int nz = finish ? Lnew : 4 * Lnew;
        for (int i = tid; i < nz; i += OCT_T) S.hist[i] = 0;
        __syncthreads();
        for (int b0 = 0; b0 < n; b0 += 4 * OCT_T) {
            const int i0 = b0 + 4 * tid;
            uint32_t key[4];
            int kk[4];
            load_keys4(keys, i0, n, key);
            load_nof4(nof, i0, n, kk);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u;
                int bin = -1;
                if (i < n) {
                    const int ko = kk[u];
                    const int x = key[u] & 0xFFF, y = (key[u] >> 12) & 0xFFF;
                    const int k = S.split[ko] ? (int)S.childIdx[4 * ko + child_of(x, y, box[ko])] : (int)S.survIdx[ko];
                    kk[u] = k;
                    if (finish) {
                        atomicMax(&S.hist[k], ((key[u] >> 24) << 20) | (0xFFFFFu - (uint32_t)i));
                    } else {
                        const uint32_t cv = ncnt[k];
                        if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) bin = 4 * k + child_of(x, y, nbox[k]);
                    }
                }
                if (bin >= 0) atomicAdd(&S.hist[bin], 1u);
            }
            if (!finish) store_nof4(nof, i0, n, kk);
        }
        __syncthreads();
        L = Lnew;
        cur ^= 1;
        if (finish) break;
        if (dbgStop >= 4 && ++npass >= dbgStop - 3) return;
    }

    // ---- D. output in list order
    const uint32_t *best = S.hist;
    uint32_t *okp = lvlKp + (size_t)b * lvlKpCap + g.lvlKpOff;
    const int Lout = min(L, g.nodeCap);
    for (int k = tid; k < Lout; k += OCT_T) okp[k] = keys[0xFFFFFu - (best[k] & 0xFFFFFu)];
    if (tid == 0) lvlCnt[b * nlevels + l] = Lout;
}

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
#define DESC_WAVES 4
#define PR 21                    // source patch radius = 18 + 3
#define PROWS (2 * PR + 1)       // 43
#define PSTRIDE 48               // 12 dwords per patch row (16-byte aligned rows)
#define PPAD 16                  // slack behind the patch: the last row's b128 read may run over
#define TROWS PROWS
#define TCOLS (2 * ORBX_DESC_R + 1)  // 37
#define TGROUPS 10               // horizontal pass: 10 groups of 4 outputs per row (cols 0..39)
#define TSTRIDE4 20              // dwords per row of the u16 intermediate (40 columns)
#define BSTRIDE 40
#define DESC_LDS_PER_WAVE (PROWS * PSTRIDE + PPAD + TROWS * TSTRIDE4 * 4 + TCOLS * BSTRIDE + 8)

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

// (float)cos((double)a), (float)sin((double)a) for a in [0, 2*pi] (src/ORBextractor.cc:112-113).
// The generic double-precision library routines cost ~220 fp64 instructions per wave (two range
// reductions with a Payne-Hanek path); here: one Cody-Waite reduction by k*pi/2 (k <= 4, exact
// product with the 33-bit head of pi/2) and the fdlibm kernel polynomials on |r| <= pi/4, < 1 ulp
// in double, so the value rounded to float is the library's (differences need a double result
// within 1e-16 of a float rounding boundary).
__device__ __forceinline__ void sincos_0_2pi(float af, float &sn, float &cs) {
    const double a = (double)af;
    const double k = rint(a * 6.36619772367581382433e-01);                       // 2/pi
    const double r = (a - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int q = (int)k & 3;
    const double sq = (q & 1) ? c : s, cq = (q & 1) ? s : c;
    sn = (float)((q & 2) ? -sq : sq);
    cs = (float)(((q + 1) & 2) ? -cq : cq);
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

__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels,
    const uint32_t *__restrict__ lvlKp, int lvlKpCap, const int32_t *__restrict__ lvlCnt,
    orbx_keypoint_t *__restrict__ kps, uint8_t *__restrict__ desc, int32_t *__restrict__ counts, int cap) {
    __shared__ __align__(16) uint8_t smem[DESC_WAVES * DESC_LDS_PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);   // all patches of an image are read through ONE L2
    const int o = bx * DESC_WAVES + wave;
    // locate (level, k) of output ordinal o: level-major concatenation (:1076-1104)
    int l = 0, base = 0, total = 0;
    {   // lane i < nlevels holds the count of level i: ONE load, a 4-step prefix sum, a ballot (not nlevels dependent loads)
        const int c = lane < nlevels ? lvlCnt[b * nlevels + lane] : 0;
        int inc = c;
#pragma unroll
        for (int d = 1; d < ORBX_MAX_LEVELS; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        total = __builtin_amdgcn_readlane(inc, ORBX_MAX_LEVELS - 1);
        if (bx == 0 && threadIdx.x == 0) counts[b] = min(total, cap);
        const unsigned long long hit = __ballot(lane < nlevels && o < inc);
        if (!hit || o >= cap) return;  // wave-uniform
        l = __builtin_ctzll(hit);
        base = __builtin_amdgcn_readlane(inc - c, l);
    }
    l = __builtin_amdgcn_readfirstlane(l);
    const LevelGeom g = geom[l];
    const uint32_t key = lvlKp[(size_t)b * lvlKpCap + g.lvlKpOff + (o - base)];
    const int cx = (int)(key & 0xFFF) + ORBX_MINB, cy = (int)((key >> 12) & 0xFFF) + ORBX_MINB;
    const int score = (int)(key >> 24);

    uint8_t *P = smem + wave * DESC_LDS_PER_WAVE;                     // source patch [43][48] (+pad)
    uint32_t *Tm = (uint32_t *)(P + PROWS * PSTRIDE + PPAD);          // horizontal pass, u16 [43][40]
    uint8_t *Bl = (uint8_t *)(Tm + TROWS * TSTRIDE4);                 // blurred [37][40]

    // ---- stage the 43x43 patch with aligned dword loads (pstride is a multiple of 64)
    const uint8_t *lvl = pyr + (size_t)b * pyrImgBytes + g.poff;
    const size_t a = (size_t)(cy + ORBX_EDGE - PR) * g.pstride + (size_t)(cx + ORBX_EDGE - PR);
    int sh = __builtin_amdgcn_readfirstlane((int)(a & 3));
    const uint32_t *src = (const uint32_t *)(lvl + (a - sh));
    const int pstr4 = g.pstride >> 2;
    // The 19-px REFLECT_101 frame of the levels >= 1 is only ever read HERE, by the few keypoints closer than PR to a
    // level's edge (the frame of level 0 comes with the copy of the input).  Those keypoints mirror the coordinates
    // themselves, so the pipeline never writes the frames of levels >= 1 (orbx_pyramid_host writes them on demand).
    const bool edge = l > 0 && (cx < PR || cy < PR || cx + PR >= g.w || cy + PR >= g.h);   // wave-uniform
    if (edge) {
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
            for (int k = 0; k < 8; k++)
                if (base0 + lane + 64 * k < PROWS * PSTRIDE) P[base0 + lane + 64 * k] = v[k];
        }
    } else
    {   // all (PROWS*12 + 63) / 64 loads of a lane are in flight before the first LDS write: one memory latency per keypoint
        constexpr int NI = (PROWS * 12 + 63) / 64;
        uint32_t v[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = min(lane + 64 * k, PROWS * 12 - 1);
            const int r = i / 12, c = i - r * 12;
            v[k] = src[(size_t)r * pstr4 + c];
        }
#pragma unroll
        for (int k = 0; k < NI; k++)
            if (lane + 64 * k < PROWS * 12) ((uint32_t *)P)[lane + 64 * k] = v[k];
    }
    wave_sync();
    // pixel (cx-21+c, cy-21+r) is byte P[r*48 + sh + c], r,c in [0,43)

    // ---- IC_Angle: two lanes per row v of the radius-15 disc (u = -15..0 | 1..15)
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        const int v = (lane >> 1) - 15, half = lane & 1;
        const int d = c_umax[v < 0 ? -v : v];
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
            s0 = __builtin_amdgcn_udot4(wa[j], c_ic.m[lane][j], s0, false);
            s1 = __builtin_amdgcn_udot4(wa[j], c_ic.w[lane][j], s1, false);
        }
        m10 = half ? (int)s1 : -(int)s1;   // u <= 0 in half 0
        m01 = v * (int)s0;
        (void)d;
    }
    m10 = wave_total_i32(m10);
    m01 = wave_total_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- horizontal 7-tap pass: 4 outputs per lane-iteration from one aligned b128 read
    const uint32_t K0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), K1 = 49u | (34u << 8) | (18u << 16);
    for (int i = lane; i < TROWS * TGROUPS; i += 64) {
        const int r = i / TGROUPS, cg = i - r * TGROUPS;
        const uint32_t *pr = (const uint32_t *)(P + r * PSTRIDE) + cg;  // dwords cg .. cg+3 hold bytes 4cg .. 4cg+15
        const uint32_t d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = pr[3];
        // w0..w2 = bytes (sh+4cg) .. +11 : source columns 4cg .. 4cg+11
        const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                       w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
        uint32_t oo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t lo = j ? __builtin_amdgcn_alignbyte(w1, w0, j) : w0;   // bytes j .. j+3
            const uint32_t hi = j ? __builtin_amdgcn_alignbyte(w2, w1, j) : w1;   // bytes j+4 .. j+7
            oo[j] = __builtin_amdgcn_udot4(hi, K1, __builtin_amdgcn_udot4(lo, K0, 0u, false), false);  // <= 65535
        }
        uint2 st;
        st.x = oo[0] | (oo[1] << 16);
        st.y = oo[2] | (oo[3] << 16);
        *(uint2 *)(Tm + r * TSTRIDE4 + cg * 2) = st;
    }
    wave_sync();

    // ---- vertical pass: lane = (column pair, row segment of 13 output rows).  A dword of the intermediate holds the u16
    // values of two columns; v_perm re-pairs two consecutive ROWS of one column, so that one v_dot2_u32_u16 applies two
    // taps: out[r] = (18,34).pair[r] + (49,55).pair[r+2] + (49,34).pair[r+4] + (0,18).pair[r+5] + 2^15, and the rounded bytes of
    // both columns leave through one more v_perm (byte 2 of the sums clamped to 2^24 - 1).
    if (lane < 60) {
        const int cp = lane % 20, seg = lane / 20;
        const int r0 = seg * 13, nr = seg == 2 ? 11 : 13;
        const uint32_t *col = Tm + r0 * TSTRIDE4 + cp;
        uint32_t T[19];   // rows r0 .. r0+18; for the last segment rows 43, 44 lie in Bl: read, never used by a stored output
#pragma unroll
        for (int k = 0; k < 19; k++) T[k] = col[k * TSTRIDE4];
        const uint32_t W01 = 18u | (34u << 16), W23 = 49u | (55u << 16), W45 = 49u | (34u << 16);
#pragma unroll
        for (int rr = 0; rr < 13; rr++) {
            uint32_t a0 = 1u << 15, a1 = 1u << 15;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t w = k == 0 ? W01 : k == 1 ? W23 : W45;
                a0 = udot2_u16_acc(__builtin_amdgcn_perm(T[rr + 2 * k + 1], T[rr + 2 * k], 0x05040100u), w, a0);
                a1 = udot2_u16_acc(__builtin_amdgcn_perm(T[rr + 2 * k + 1], T[rr + 2 * k], 0x07060302u), w, a1);
            }
            // 7th tap = high half of pair[rr+5] = (row rr+5, row rr+6), a pair the next output row needs anyway
            a0 = udot2_u16_acc(__builtin_amdgcn_perm(T[rr + 6], T[rr + 5], 0x05040100u), 18u << 16, a0);
            a1 = udot2_u16_acc(__builtin_amdgcn_perm(T[rr + 6], T[rr + 5], 0x07060302u), 18u << 16, a1);
            a0 = min(a0, 0xFFFFFFu);   // the taps sum to 257: a saturated patch reaches 257 * 65535 + 2^15 > 2^24 (-> 255)
            a1 = min(a1, 0xFFFFFFu);
            if (rr < nr) *(uint16_t *)(Bl + (r0 + rr) * BSTRIDE + 2 * cp) = (uint16_t)__builtin_amdgcn_perm(a1, a0, 0x0c0c0602u);
        }
    }
    wave_sync();

    // ---- steered BRIEF: 4 rounds x 64 pairs, one ballot = 8 descriptor bytes
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = angle * factorPI;
    float ca, sa;
    sincos_0_2pi(ang, sa, ca);
    const uint8_t *Bc = Bl + ORBX_DESC_R * BSTRIDE + ORBX_DESC_R;
    unsigned long long bits[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pair = r * 64 + lane;
        const int8_t *q = c_pattern + 4 * pair;
        const float x0 = (float)q[0], y0 = (float)q[1], x1 = (float)q[2], y1 = (float)q[3];
        const int t0 = Bc[__float2int_rn(x0 * sa + y0 * ca) * BSTRIDE + __float2int_rn(x0 * ca - y0 * sa)];
        const int t1 = Bc[__float2int_rn(x1 * sa + y1 * ca) * BSTRIDE + __float2int_rn(x1 * ca - y1 * sa)];
        bits[r] = __ballot(t0 < t1);
    }
    const size_t oi = (size_t)b * cap + o;
    if (lane < 4) ((unsigned long long *)(desc + oi * 32))[lane] = bits[lane];
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
    }
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
    if (!out) { orbx_set_error("orbx_create: out is NULL"); return ORBX_ERR_ARG; }
    *out = nullptr;
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
    for (int r = 0; r < ORBX_EV_RING; r++)
        // timing-only events: no system-scope fence (cache write-back + invalidate) when they complete — with the default
        // flags every stage boundary of a profiled batch cost ~5 us of idle GPU, which the step time then contained
        for (int i = 0; i < ORBX_NUM_STAGES; i++) ORBX_HIP(hipEventCreateWithFlags(&h->ev[r][i], hipEventDisableSystemFence));
    *out = h;
    return ORBX_OK;
}

static void free_plan(orbx_extractor *h) {
    hipFree(h->d_cellOff); h->d_cellOff = nullptr;
    hipFree(h->d_octFallback); h->d_octFallback = nullptr;
    hipFree(h->d_geom); hipFree(h->d_tab); hipFree(h->d_pyr); hipFree(h->d_cellCnt); hipFree(h->d_slots);
    hipFree(h->d_cand); hipFree(h->d_lvlKp); hipFree(h->d_nodeOf); hipFree(h->d_candCnt); hipFree(h->d_lvlCnt);
    h->d_geom = nullptr; h->d_tab = nullptr; h->d_pyr = nullptr; h->d_cellCnt = nullptr; h->d_slots = nullptr;
    h->d_cand = nullptr; h->d_lvlKp = nullptr; h->d_nodeOf = nullptr; h->d_candCnt = nullptr; h->d_lvlCnt = nullptr;
    h->pw = h->ph = h->pB = 0;
}

extern "C" int orbx_destroy(orbx_extractor_t *h) {
    if (!h) return ORBX_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->last_stream) hipStreamSynchronize(h->last_stream);
    free_plan(h);
    hipFree(h->d_in); hipFree(h->d_kps); hipFree(h->d_desc); hipFree(h->d_counts);
    if (h->h_kps) { hipHostFree(h->h_kps); hipHostFree(h->h_desc); hipHostFree(h->h_counts); }
    for (int r = 0; r < ORBX_EV_RING; r++)
        for (int i = 0; i < ORBX_NUM_STAGES; i++) hipEventDestroy(h->ev[r][i]);
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
    if (h->last_stream) ORBX_HIP(hipStreamSynchronize(h->last_stream));
    const int keepB = (h->pw == w && h->ph == hgt) ? h->pB : 0;
    free_plan(h);
    B = std::max(B, keepB);
    std::vector<int32_t> tab;
    size_t poff = 0, slotOff = 0, keyOff = 0;
    int cellBase = 0, lvlKpOff = 0, maxNodeCap = 0, maxCells = 0, maxPyrWords = 0;
    int maxTw = 0, maxTh = 0;
    int kpBound = 0;
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
        if (g.keyCap > 0xFFFFF) {
            orbx_set_error("level %d can hold %d FAST candidates (> 2^20): unsupported", l, g.keyCap);
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
            while ((g.nIni << (2 * d)) > 16384 && d > 1) d--;
            g.pyrDepth = d;
            const int words = g.nIni * (((1 << (2 * d)) - 1) / 3) + ((g.nIni << (2 * d)) + 1) / 2 + 1;
            maxPyrWords = std::max(maxPyrWords, words);
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
        int T = (int)lrintf((g_debug[3] > 0 ? (float)g_debug[3] : 64.0f) / h->sf[Lc]);
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
    if (maxTw > 65 || maxTh > 65) { orbx_set_error("cell window %dx%d exceeds 65", maxTw, maxTh); return ORBX_ERR_UNSUPPORTED; }
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
        h->octPyrLdsBytes = sizeof(unsigned long long) * pow2 + (size_t)maxNodeCap * (8 + 8 + 16 + 4 + 2 + 1) + 4 * (size_t)maxPyrWords + 64;
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
    ORBX_HIP(hipMalloc(&h->d_cellOff, sizeof(uint32_t) * h->totalCells * Bz));
    ORBX_HIP(hipMalloc(&h->d_slots, sizeof(uint32_t) * h->slotsPerImg * Bz));
    ORBX_HIP(hipMalloc(&h->d_cand, sizeof(uint32_t) * h->keysPerImg * Bz));
    ORBX_HIP(hipMalloc(&h->d_nodeOf, sizeof(uint16_t) * h->keysPerImg * Bz));
    ORBX_HIP(hipMalloc(&h->d_candCnt, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMalloc(&h->d_lvlCnt, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMalloc(&h->d_lvlKp, sizeof(uint32_t) * (size_t)h->lvlKpCap * Bz));
    ORBX_HIP(hipMalloc(&h->d_octFallback, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
    ORBX_HIP(hipMemset(h->d_octFallback, 0, sizeof(int32_t) * ORBX_MAX_LEVELS * Bz));
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

static int launch_pipeline(orbx_extractor *h, const uint8_t *d_imgs, int B, int w, int hgt, int stride,
                           size_t img_stride, orbx_keypoint_t *d_kps, uint8_t *d_desc, int32_t *d_counts,
                           int cap, hipStream_t st) {
    const int nl = h->nlevels;
    CellBases cb;
    for (int l = 0; l <= ORBX_MAX_LEVELS; l++) cb.v[l] = l < nl ? h->geom[l].cellBase : h->totalCells;
    // profiling 1: events at every stage boundary; 2: only around k_fast_cells (an event costs ~4.5 us of idle GPU, so
    // a throughput measurement brackets just the kernel it reports)
    const bool prof = h->profiling == 1, profFast = h->profiling != 0;
    (void)hipGetLastError();  // drop stale errors of other HIP users in this process
    hipEvent_t *ev = nullptr;
    if (profFast) {
        const int slot = h->ev_head % ORBX_EV_RING;
        if (h->ev_pending[slot]) { int rc = harvest_events(h, slot); if (rc) return rc; }
        ev = h->ev[slot];
        if (prof) ORBX_HIP(hipEventRecord(ev[0], st));
    }
    if (g_debug[5] == 0 && h->scale_factor <= 3.0) {   // K1, one launch per level (a lane's two source byte pairs fit 8 bytes)
        const LevelGeom &g0 = h->geom[0];
        hipLaunchKernelGGL(k_pyr_pad<true>, dim3(((g0.pstride >> 4) * g0.prows + 255) / 256, 1, B), dim3(256), 0, st, d_imgs, stride,
                           img_stride, h->d_pyr, h->pyrImgBytes, h->d_geom, 0);
        for (int l = 1; l < nl; l++) {
            const int nxc = (h->geom[l].w + 1 + 127) / 128, nbands = (h->geom[l].h + PYR_RW - 1) / PYR_RW;
            hipLaunchKernelGGL(k_pyr_level, dim3((nxc * nbands + 3) / 4, B), dim3(256), 0, st, h->d_pyr, h->pyrImgBytes,
                               h->d_geom, l, h->d_tab, nxc, nbands);
        }
        h->framesStale = nl > 1 ? B : 0;   // frames of levels >= 1: written on demand (ensure_frames)
    } else {   // K1, fused form (orbx_debug_set(5, 1)): writes every frame itself
        h->framesStale = 0;
        hipLaunchKernelGGL(k_pyramid_fused, dim3(h->pyrTilesX * h->pyrTilesY, B), dim3(256), h->pyrLdsBytes, st, d_imgs,
                           stride, img_stride, h->d_pyr, h->pyrImgBytes, h->d_geom, nl, h->d_tab, h->pyrXSpanOff,
                           h->pyrYSpanOff, h->pyrTilesX, h->pyrTilesY, h->pyrBufBytes, h->pyrMaxPar);
    }
    if (profFast) ORBX_HIP(hipEventRecord(ev[1], st));
    {   // K2
        dim3 grid((h->totalCells + FAST_WAVES - 1) / FAST_WAVES, B);
#define ORBX_LAUNCH_FAST(EST)                                                                                         \
    hipLaunchKernelGGL(k_fast_cells<EST>, grid, dim3(64 * FAST_WAVES), (size_t)h->fastLdsPerWave * FAST_WAVES, st,  \
                       h->d_pyr, h->pyrImgBytes, h->d_geom, nl, h->totalCells, h->d_cellCnt, h->d_slots,            \
                       h->slotsPerImg, h->ini_th, h->min_th, h->fastTileStride, h->fastScoreStride, h->fastTileRows, \
                       h->fastLdsPerWave, g_debug[0], cb)
        const int es = (h->fastScoreStride == h->fastTileStride - 8 && g_debug[6] == 0) ? h->fastTileStride : 0;
        switch (es) {   // the strides of the usual 30-px cell grids; anything else takes the run-time-stride instance
        case 44: ORBX_LAUNCH_FAST(44); break;
        case 48: ORBX_LAUNCH_FAST(48); break;
        case 52: ORBX_LAUNCH_FAST(52); break;
        default: ORBX_LAUNCH_FAST(0); break;
        }
#undef ORBX_LAUNCH_FAST
    }
    if (profFast) ORBX_HIP(hipEventRecord(ev[2], st));
    {   // K3
        int pow2 = 1;
        while (pow2 < h->maxNodeCap) pow2 <<= 1;
        int maxCells = 0;
        for (int l = 0; l < nl; l++) maxCells = std::max(maxCells, h->geom[l].ncells);
        const int scratch = std::max(4 * h->maxNodeCap, maxCells + 1);
        ORBX_HIP(hipFuncSetAttribute((const void *)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->octLdsBytes));
        hipLaunchKernelGGL(k_cell_scan, dim3(nl, B), dim3(256), 0, st, h->d_geom, nl, h->totalCells, h->d_cellCnt,
                           h->d_cellOff, h->d_candCnt);
        hipLaunchKernelGGL(k_gather, dim3((h->totalCells + GATHER_CELLS_PER_BLOCK - 1) / GATHER_CELLS_PER_BLOCK, B),
                           dim3(256), 0, st, h->d_geom, nl, h->totalCells, h->d_cellCnt, h->d_cellOff, h->d_slots,
                           h->slotsPerImg, h->d_cand, h->keysPerImg, cb);
        const bool usePyr = g_debug[4] == 0;
        if (usePyr) {
            ORBX_HIP(hipFuncSetAttribute((const void *)k_octree_pyr, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)h->octPyrLdsBytes));
            hipLaunchKernelGGL(k_octree_pyr, dim3(B, nl), dim3(OCT_T), h->octPyrLdsBytes, st, h->d_geom, nl, h->d_cand,
                               h->keysPerImg, h->d_candCnt, h->d_lvlKp, h->lvlKpCap, h->d_lvlCnt, h->d_tab, h->maxNodeCap,
                               pow2, h->octPyrWords, h->d_octFallback);
        }
        hipLaunchKernelGGL(k_octree, dim3(B, nl), dim3(OCT_T), h->octLdsBytes, st, h->d_geom, nl, h->totalCells,
                           h->d_cellCnt, h->d_slots, h->slotsPerImg, h->d_cand, h->d_nodeOf, h->keysPerImg,
                           h->d_candCnt, h->d_lvlKp, h->lvlKpCap, h->d_lvlCnt, h->d_tab, h->maxNodeCap, pow2, scratch,
                           g_debug[1], usePyr ? h->d_octFallback : (const int32_t *)nullptr);
    }
    if (prof) ORBX_HIP(hipEventRecord(ev[3], st));
    {   // K4
        const int maxo = std::min(cap, h->max_kp);
        dim3 grid((maxo + DESC_WAVES - 1) / DESC_WAVES, B);
        hipLaunchKernelGGL(k_describe, grid, dim3(64 * DESC_WAVES), 0, st, h->d_pyr, h->pyrImgBytes, h->d_geom, nl,
                           h->d_lvlKp, h->lvlKpCap, h->d_lvlCnt, d_kps, d_desc, d_counts, cap);
    }
    if (prof) ORBX_HIP(hipEventRecord(ev[4], st));
    if (profFast) { h->ev_pending[h->ev_head % ORBX_EV_RING] = (unsigned char)h->profiling; h->ev_head++; }
    ORBX_HIP(hipGetLastError());
    h->last_stream = st;
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
    return launch_pipeline(h, d_imgs, B, w, hgt, stride, image_stride_bytes, d_kps, d_desc, d_counts, cap, st);
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
    if (h->last_stream) ORBX_HIP(hipStreamSynchronize(h->last_stream));
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

extern "C" int orbx_debug_level_points(orbx_extractor_t *h, int b, int level, int stage, int32_t *out, int cap,
                                       int *n_out) {
    if (!h || !n_out || level < 0 || level >= h->nlevels || h->pw == 0 || b < 0 || b >= h->pB || stage < 0 || stage > 1) {
        orbx_set_error("orbx_debug_level_points: bad arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP(hipSetDevice(h->device));
    if (h->last_stream) ORBX_HIP(hipStreamSynchronize(h->last_stream));
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

extern "C" int orbx_set_profiling(orbx_extractor_t *h, int enabled) {
    if (!h) return ORBX_ERR_ARG;
    ORBX_HIP(hipSetDevice(h->device));
    for (int r = 0; r < ORBX_EV_RING; r++)
        if (h->ev_pending[r]) { ORBX_HIP(hipEventSynchronize(h->ev[r][h->ev_pending[r] == 2 ? 2 : 4])); h->ev_pending[r] = 0; }
    if (enabled < 0 || enabled > 2) { orbx_set_error("orbx_set_profiling: mode %d", enabled); return ORBX_ERR_ARG; }
    h->profiling = enabled;
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
