// orbx_pyramid.hip — ORBextractor::ComputePyramid (src/ORBextractor.cc:1107-1132): k_pyr_pad, k_pyr_level, k_pyramid_fused
// (part of the ORB extractor, see orbx_extract.hip for the pipeline and the C ABI)
#include "orbx_extract_dev.h"
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

__global__ __launch_bounds__(256) void k_pyramid_fused(
    const uint8_t *__restrict__ src, int sstride, size_t simg, uint8_t *__restrict__ pyr, size_t pyrImgBytes,
    const LevelGeom *__restrict__ geom, int nlevels, const int32_t *__restrict__ tab, int xSpanOff, int ySpanOff,
    int tilesX, int tilesY, int bufBytes, int maxPar, int l0) {
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
        for (int l = 0; l < nlevels; l++) {   // parameter space of the levels this launch builds (l > l0)
            xo[l] = ax; yo[l] = ay;
            if (l > l0) { ax += sX[l].c1 - sX[l].c0; ay += sY[l].c1 - sY[l].c0; }
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
            int l = l0 + 1;
            while (l + 1 <= nlevels && j >= off[l + 1]) l++;   // level of entry j (the source level l0 has no parameters)
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
        const PyrSpan X = sX[l0], Y = sY[l0];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0;
        // l0 == 0: the input image; l0 > 0 (levels 1..l0 were built by k_pyr_level): the inner rows of level l0 in the pyramid
        const int ss = l0 == 0 ? sstride : geom[l0].pstride;
        const uint8_t *s = (l0 == 0 ? src + (size_t)b * simg
                                    : pyr + (size_t)b * pyrImgBytes + geom[l0].poff + (size_t)ORBX_EDGE * ss + ORBX_EDGE) + (size_t)Y.c0 * ss + X.c0;
        const unsigned M = ((1u << 20) + cw - 1) / cw;
        uint8_t *first = smem + (l0 & 1) * bufBytes;
        for (int i = tid; i < cw * ch; i += 256) {
            const int y = (int)(((unsigned)i * M) >> 20), x = i - y * cw;
            first[i] = s[(size_t)y * ss + x];
        }
    }
    __syncthreads();
    uint8_t *base = pyr + (size_t)b * pyrImgBytes;
    for (int l = l0; l < nlevels; l++) {
        const PyrSpan X = sX[l], Y = sY[l];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0;
        uint8_t *cur = smem + (l & 1) * bufBytes;  // ping-pong; plain offsets keep the LDS address space
        const LevelGeom *g = geom + l;
        const int lw = g->w, lh = g->h, pstride = g->pstride;
        if (l > l0) {
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
        if (l == l0 && l0 > 0) continue;   // the source level is in memory already
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
//  * k_pyr_level: a wave owns 128 output columns x RW (8 or 16) output rows; a lane owns TWO fixed
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

// RW output rows per wave from SR source rows fetched up front (8 from 12: the form for small levels and small batches; 16 from 22 when
// the level still gives every SIMD several waves: the per-wave set-up - column tables, selectors, row tables - and the source rows two
// neighbouring bands both fetch are paid half as often, 17 -> 14 VALU instructions per output pixel).
// VResizeLinear's (beta * (S >> 4)) >> 16 as ONE 24-bit multiply: the horizontal sum keeps its scale with the low four bits cleared
// (S & ~15 = (S >> 4) << 4, below 2^19), beta (<= 2048) is shifted up by 12 (below 2^24), and the high half of the 48-bit product
// (b << 12) * ((S >> 4) << 4) is (b * (S >> 4)) >> 16 exactly.  v_mul_hi_u32_u24 is a full-rate instruction; the multiply + shift it
// replaces were two.
__device__ __forceinline__ uint32_t vmulhi24(uint32_t a, uint32_t b) { return __umulhi(a & 0xFFFFFFu, b & 0xFFFFFFu); }   // (the masks are known no-ops: they let the compiler pick the 24-bit form)
template <int RW, int SR>
__global__ __launch_bounds__(256) void k_pyr_level(uint8_t *__restrict__ pyr, size_t pyrImgBytes,
                                                   const LevelGeom *__restrict__ geom, int l,
                                                   const int32_t *__restrict__ tab, int nxc, int nbands) {
    static_assert((RW == 8 || RW == 16) && SR <= 32, "row table: one lane per output row, a 32-bit mask of source rows");
    constexpr int RWM = RW - 1;
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
    const int y0 = band * RW, nrow = min(RW, G.h - y0);
    // fast path: the band's source rows rf .. rf+SR-1 are fetched up front (one memory latency per
    // wave), then consumed in order.  Lane i < RW holds the row table of output row y0+i; bit k of
    // `mask` says "the output row whose second source row is rf+k is due after source row k".
    const int yl = min(y0 + (lane & RWM), G.h - 1);
    const int vsy = tab[G.yofsOff + yl];
    const uint32_t vbt = (uint32_t)tab[G.ybetaOff + yl];
    const int rf = __builtin_amdgcn_readfirstlane(vsy);
    const int kk = vsy + 1 - rf;
    const int prevsy = __shfl_up(vsy, 1);
    const bool okl = (lane & RWM) >= nrow || (rf >= 0 && vsy + 1 <= sh - 1 && kk < SR && ((lane & RWM) == 0 || vsy > prevsy));
    constexpr unsigned long long rowLanes = (1ull << RW) - 1ull;
    const bool regular = (__ballot(okl) & rowLanes) == rowLanes;
    if (regular) {
        uint32_t m = (lane & RWM) < nrow ? 1u << (kk & 31) : 0u;
        m |= __shfl_xor(m, 1); m |= __shfl_xor(m, 2); m |= __shfl_xor(m, 4);
        if (RW == 16) m |= __shfl_xor(m, 8);
        const uint32_t mask = __builtin_amdgcn_readfirstlane(m);
        const int lastk = 31 - __builtin_clz(mask | 1u);   // source rows past the last output row's second one are neither fetched nor filtered
        uint2 q[SR];
#pragma unroll
        for (int k = 0; k < SR; k++) {   // scalar row pointer + the lane's 32-bit column offset: no per-lane address arithmetic
            const uint8_t *rowp = srow0 + (uint32_t)(min(rf + k, sh - 1) * sps);   // 32-bit scalar product (a padded level is far smaller than 4 GiB)
            q[k] = make_uint2(0u, 0u);
            if (k <= lastk) q[k] = *(const uint2 *)(rowp + (uint32_t)A);   // wave-uniform
        }
        uint32_t tpa = 0, tpb = 0;
        int cnt = 0;
        uint8_t *drow = drow0 + (size_t)y0 * G.pstride - 1;   // scalar; the lane's column x0 = xoff - 1 with xoff >= 0
        const uint32_t xoff = (uint32_t)(x0 + 1);
#pragma unroll
        for (int k = 0; k < SR; k++) {
            if (k > lastk) continue;   // wave-uniform
            const uint32_t tca = udot2_u16(__builtin_amdgcn_perm(q[k].y, q[k].x, selA), aa) & 0x7FFF0u;
            const uint32_t tcb = udot2_u16(__builtin_amdgcn_perm(q[k].y, q[k].x, selB), ab) & 0x7FFF0u;
            if (k > 0 && ((mask >> k) & 1u)) {   // wave-uniform
                const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)vbt, cnt);
                const uint32_t b0 = (bb & 0xFFFu) << 12, b1 = ((bb >> 16) & 0xFFFu) << 12;   // scalar; beta <= 2048
                const uint32_t pa = (vmulhi24(b0, tpa) + vmulhi24(b1, tca) + 2) >> 2;
                const uint32_t pb = (vmulhi24(b0, tpb) + vmulhi24(b1, tcb) + 2) >> 2;
                if (vst) *(uint16_t *)(drow + xoff) = (uint16_t)(pa | (pb << 8));
                drow += G.pstride;
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
            t0a = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selA), aa) & 0x7FFF0u;
            t0b = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selB), ab) & 0x7FFF0u;
            cr0 = r0;
        }
        if (r1 != cr1) {
            if (r1 == cr0) { t1a = t0a; t1b = t0b; }
            else {
                const uint2 q = *(const uint2 *)(srow0 + (size_t)r1 * sps + A);
                t1a = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selA), aa) & 0x7FFF0u;
                t1b = udot2_u16(__builtin_amdgcn_perm(q.y, q.x, selB), ab) & 0x7FFF0u;
            }
            cr1 = r1;
        }
        const uint32_t b0 = (bb & 0xFFFu) << 12, b1 = ((bb >> 16) & 0xFFFu) << 12;
        const uint32_t pa = (vmulhi24(b0, t0a) + vmulhi24(b1, t1a) + 2) >> 2;
        const uint32_t pb = (vmulhi24(b0, t0b) + vmulhi24(b1, t1b) + 2) >> 2;
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
        // Chunk order: first every chunk that lies inside the image row (one 16-byte load), then the few per row that touch
        // the frame (byte gathers) - in row-major order each of those sat in a different wave and made nearly EVERY wave run
        // both paths (213 VALU instructions per wave for a copy).
        const int pc = G.pstride >> 4, totalc = pc * rows;
        const int cI0 = (ORBX_EDGE + 15) >> 4, cI1 = max((G.w + ORBX_EDGE) >> 4, cI0);   // interior chunks: cI0 <= c < cI1
        const int nI = cI1 - cI0, nE = pc - nI, totalI = nI * rows;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < totalc; i += gridDim.x * 256) {
            int py, c;
            if (i < totalI) {
                py = i / nI;
                c = cI0 + (i - py * nI);
            } else {
                const int j = i - totalI;
                py = j / nE;
                const int k = j - py * nE;
                c = k < cI0 ? k : cI1 + (k - cI0);
            }
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

// Level 0 of a SMALL batch (round 5): the same padded level as k_pyr_pad<true>, built from source rows staged in LDS.  The image of a
// latency call sits in pinned HOST memory and is read over the bus by this kernel (no copy command): there every load is a bus
// transaction, so each source byte is fetched exactly once, by ALIGNED 16-byte loads over one contiguous span per workgroup (PAD_ROWS
// whole source rows; an aligned 16-byte load that holds one valid byte never leaves that byte's page).  k_pyr_pad<true> reads
// unaligned 16-byte chunks (split into several transactions) and gathers the frame bytes one by one: 28.8 us for two 1241x376
// images against ~19 at the bus rate.  The padded rows - the rows themselves and their BORDER_REFLECT_101 mirror images above /
// below the image - are then assembled from LDS.  Dynamic LDS: PAD_ROWS * sstride + 32 bytes.
#define PAD_ROWS PAD_ROWS_PER_BLOCK   // (orbx_extract_dev.h: the launch sizes its LDS with it)
__global__ __launch_bounds__(256) void k_pyr_pad_rows(const uint8_t *__restrict__ img, int sstride, size_t simg, uint8_t *__restrict__ pyr,
                                                      size_t pyrImgBytes, const LevelGeom *__restrict__ geom) {
    extern __shared__ __align__(16) uint8_t rows_lds[];
    const LevelGeom G = geom[0];
    const int tid = threadIdx.x, r0 = blockIdx.x * PAD_ROWS, nr = min(PAD_ROWS, G.h - r0);
    const uint8_t *src = img + (size_t)blockIdx.y * simg;
    uint8_t *lvl = pyr + (size_t)blockIdx.y * pyrImgBytes + G.poff;
    // ---- 1. the span of the nr source rows, aligned outwards to 16 bytes
    const uintptr_t p0 = (uintptr_t)(src + (size_t)r0 * sstride), p1 = p0 + (size_t)(nr - 1) * sstride + G.w;
    const uintptr_t a0 = p0 & ~(uintptr_t)15;
    const int nch = (int)(((p1 + 15) & ~(uintptr_t)15) - a0) >> 4, lead = (int)(p0 - a0);   // LDS byte of source byte (r, x): lead + (r - r0) * sstride + x
    for (int i0 = 0; i0 < nch; i0 += 4 * 256) {   // a thread's (up to) four chunks are all requested before the first is stored: one bus round trip, not four
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + tid + 256 * k; v[k] = ((const uint4 *)a0)[min(i, nch - 1)]; }
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = i0 + tid + 256 * k; if (i < nch) ((uint4 *)rows_lds)[i] = v[k]; }
    }
    __syncthreads();
    // ---- 2. padded rows: 16-byte chunks of the row itself (target 0) and of its mirror images (targets 1, 2)
    const int pc = G.pstride >> 4;
    for (int t = 0; t < 3; t++) {
        for (int i = tid; i < nr * pc; i += 256) {
            const int sr = i / pc, c = i - sr * pc, r = r0 + sr;
            int py;
            if (t == 0) py = r + ORBX_EDGE;
            else if (t == 1) { if (r < 1 || r > ORBX_EDGE) continue; py = ORBX_EDGE - r; }                                  // rows above the image
            else { if (r < G.h - 1 - ORBX_EDGE || r > G.h - 2) continue; py = 2 * (G.h - 1) - r + ORBX_EDGE; }            // rows below
            const uint8_t *row = rows_lds + lead + sr * sstride;
            const int px = c * 16 - ORBX_EDGE;
            uint4 v;
            if (px >= 0 && px + 15 < G.w) {   // inside the row: 16 contiguous bytes at any LDS alignment = five aligned dwords, byte-aligned
                const uint32_t off = (uint32_t)(row - rows_lds) + (uint32_t)px, sh = off & 3u;
                const uint32_t *d = (const uint32_t *)(rows_lds + (off & ~3u));
                const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
                v = make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                               __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
            } else {
                uint32_t w4[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    w4[u] = (uint32_t)row[reflect101c(px + 4 * u, G.w)] | ((uint32_t)row[reflect101c(px + 4 * u + 1, G.w)] << 8) |
                            ((uint32_t)row[reflect101c(px + 4 * u + 2, G.w)] << 16) | ((uint32_t)row[reflect101c(px + 4 * u + 3, G.w)] << 24);
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            *(uint4 *)(lvl + (size_t)py * G.pstride + c * 16) = v;
        }
    }
}

// Level chain (see ChainPlan, orbx_extract_dev.h).  256 threads; a lane owns TWO adjacent columns of the tile's rectangle at every level
// (a rectangle is at most 128 columns wide: the host sizes the tiles accordingly), the four waves take the rows round-robin.
// LDS: two level buffers (ping-pong, row pitch a multiple of 4 with >= 8 bytes of slack, so every source access is two ALIGNED dwords)
// and the row parameters of every level (source rows relative to the buffer, clamped, + the packed betas).
__global__ __launch_bounds__(256) void k_pyr_chain(uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom,
                                                   const int32_t *__restrict__ tab, ChainPlan cp) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ PyrSpan sX[PC_MAXL + 1], sY[PC_MAXL + 1];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int tx = blockIdx.x % cp.tilesX, ty = blockIdx.x / cp.tilesX, b = blockIdx.y;
    const int nb = cp.lb - cp.la;     // levels built
    const int ntX = cp.tilesX, ntY = cp.tilesY;
    if (tid <= nb) sX[tid] = ((const PyrSpan *)(tab + cp.xSpanOff))[tid * ntX + tx];
    else if (tid >= 32 && tid <= 32 + nb) sY[tid - 32] = ((const PyrSpan *)(tab + cp.ySpanOff))[(tid - 32) * ntY + ty];
    __syncthreads();
    uint8_t *buf[2] = {smem, smem + cp.bufBytes};
    uint2 *ypar = (uint2 *)(smem + 2 * cp.bufBytes);      // [nb][maxRows]
    uint8_t *base = pyr + (size_t)b * pyrImgBytes;
    // ---- column parameters of my two columns at every built level (registers) and the row parameters (LDS): one round of loads,
    // issued together with the source rectangle
    int ia[PC_MAXL], ob[PC_MAXL]; uint32_t aa[PC_MAXL], ab[PC_MAXL];
#pragma unroll
    for (int k = 0; k < PC_MAXL; k++) {
        ia[k] = 0; ob[k] = 0; aa[k] = 0; ab[k] = 0;
        if (k < nb) {
            const LevelGeom *g = geom + cp.la + 1 + k;
            const PyrSpan X = sX[k + 1], Xp = sX[k];
            const int xa = min(X.c0 + 2 * lane, g->w - 1), xb = min(xa + 1, g->w - 1);
            const int sa = tab[g->xofsOff + xa], sb2 = tab[g->xofsOff + xb];
            aa[k] = (uint32_t)tab[g->xalphaOff + xa]; ab[k] = (uint32_t)tab[g->xalphaOff + xb];
            ia[k] = sa - Xp.c0;                 // column of the first source byte in the previous level's buffer (>= 0 inside my rectangle)
            ob[k] = sb2 - sa;                   // 0, 1 or 2 (scale factor <= 3)
            const PyrSpan Y = sY[k + 1], Yp = sY[k];
            const int sh = g[-1].h, ch = Y.c1 - Y.c0;
            for (int y = tid; y < ch; y += 256) {
                const int sy = tab[g->yofsOff + Y.c0 + y];
                uint2 q;
                q.x = (uint32_t)(min(max(sy, 0), sh - 1) - Yp.c0) | ((uint32_t)(min(max(sy + 1, 0), sh - 1) - Yp.c0) << 16);
                q.y = (uint32_t)tab[g->ybetaOff + Y.c0 + y];
                ypar[k * cp.maxRows + y] = q;
            }
        }
    }
    {   // source rectangle: rows of level la (inner pixels of the padded buffer), aligned dword pairs -> LDS dwords
        const LevelGeom *g = geom + cp.la;
        const PyrSpan X = sX[0], Y = sY[0];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0, pitch = (cw + 8 + 3) & ~3, nd = pitch >> 2;
        const uint8_t *s0 = base + g->poff + (size_t)(ORBX_EDGE + Y.c0) * g->pstride + ORBX_EDGE + X.c0;
        const uint32_t shf = (uint32_t)((uintptr_t)s0 & 3);   // pstride % 4 == 0: the same misalignment in every row
        const uint8_t *s0a = s0 - shf;
        const unsigned M = ((1u << 20) + nd - 1) / nd;
        for (int i = tid; i < nd * ch; i += 256) {
            const int r = (int)(((unsigned)i * M) >> 20), k = i - r * nd;
            const uint32_t *p = (const uint32_t *)(s0a + (size_t)r * g->pstride) + k;     // (reads up to 7 bytes past the rectangle: still inside the padded row)
            ((uint32_t *)(buf[0] + r * pitch))[k] = __builtin_amdgcn_alignbyte(p[1], p[0], shf);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PC_MAXL; k++) {
        if (k >= nb) break;
        const LevelGeom *g = geom + cp.la + 1 + k;
        const PyrSpan X = sX[k + 1], Y = sY[k + 1];
        const int cw = X.c1 - X.c0, ch = Y.c1 - Y.c0;
        const int ppitch = (sX[k].c1 - sX[k].c0 + 8 + 3) & ~3, pitch = (cw + 8 + 3) & ~3;
        const uint8_t *prev = buf[k & 1];
        uint8_t *cur = buf[(k & 1) ^ 1];
        const int A = ia[k] & ~3;
        const uint32_t oa = (uint32_t)(ia[k] - A), obb = oa + (uint32_t)ob[k];
        const uint32_t selA = oa | ((oa + 1) << 16) | 0x0C000C00u, selB = obb | ((obb + 1) << 16) | 0x0C000C00u;
        const bool act = 2 * lane < cw;
        const int xa = X.c0 + 2 * lane;
        const bool ownA = xa >= X.o0 && xa < X.o1, ownB = xa + 1 >= X.o0 && xa + 1 < X.o1 && 2 * lane + 1 < cw;
        uint8_t *drow0 = base + g->poff + (size_t)ORBX_EDGE * g->pstride + ORBX_EDGE + xa;
        const uint2 *yp = ypar + k * cp.maxRows;
        for (int y = wave; y < ch; y += 4) {       // wave-uniform row
            const uint2 q = yp[y];
            const uint32_t r0 = q.x & 0xFFFFu, r1 = q.x >> 16;
            const uint32_t b0 = (q.y & 0xFFFu) << 12, b1 = ((q.y >> 16) & 0xFFFu) << 12;
            if (act) {
                const uint32_t *p0 = (const uint32_t *)(prev + r0 * ppitch + A), *p1 = (const uint32_t *)(prev + r1 * ppitch + A);
                const uint32_t d00 = p0[0], d01 = p0[1], d10 = p1[0], d11 = p1[1];
                const uint32_t t0a = udot2_u16(__builtin_amdgcn_perm(d01, d00, selA), aa[k]) & 0x7FFF0u;
                const uint32_t t0b = udot2_u16(__builtin_amdgcn_perm(d01, d00, selB), ab[k]) & 0x7FFF0u;
                const uint32_t t1a = udot2_u16(__builtin_amdgcn_perm(d11, d10, selA), aa[k]) & 0x7FFF0u;
                const uint32_t t1b = udot2_u16(__builtin_amdgcn_perm(d11, d10, selB), ab[k]) & 0x7FFF0u;
                const uint32_t pa = (vmulhi24(b0, t0a) + vmulhi24(b1, t1a) + 2) >> 2;
                const uint32_t pb = (vmulhi24(b0, t0b) + vmulhi24(b1, t1b) + 2) >> 2;
                *(uint16_t *)(cur + y * pitch + 2 * lane) = (uint16_t)(pa | (pb << 8));
                const int yy = Y.c0 + y;
                if (yy >= Y.o0 && yy < Y.o1) {     // wave-uniform: the owner writes the level
                    uint8_t *d = drow0 + (size_t)yy * g->pstride;
                    if (ownA) d[0] = (uint8_t)pa;
                    if (ownB) d[1] = (uint8_t)pb;
                }
            }
        }
        __syncthreads();
    }
}

template __global__ void k_pyr_level<8, 12>(uint8_t *, size_t, const LevelGeom *, int, const int32_t *, int, int);
template __global__ void k_pyr_level<16, 22>(uint8_t *, size_t, const LevelGeom *, int, const int32_t *, int, int);
template __global__ void k_pyr_pad<true>(const uint8_t *, int, size_t, uint8_t *, size_t, const LevelGeom *, int);
template __global__ void k_pyr_pad<false>(const uint8_t *, int, size_t, uint8_t *, size_t, const LevelGeom *, int);
