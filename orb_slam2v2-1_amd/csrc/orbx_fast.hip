// orbx_fast.hip — per-cell cv::FAST with threshold fallback (src/ORBextractor.cc:789-829): k_fast_cells, k_cell_scan, k_gather
// (part of the ORB extractor, see orbx_extract.hip for the pipeline and the C ABI)
#include "orbx_extract_dev.h"
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
                // scalar window origin + 32-bit lane offset (a padded level is far smaller than 4 GiB)
                const uint32_t *p = (const uint32_t *)((const uint8_t *)src + (uint32_t)(rr[k] * pstr4 + qq[k]) * 4u);
                d0[k] = p[0]; d1[k] = p[1];
            }
#pragma unroll
            for (int k = 0; k < FAST_STG; k++) {
                // unconditional: lanes past the end hold the clamped LAST item and rewrite it with the same value (a store
                // under a lane condition lets the compiler sink that slot's load behind a divergent branch: one more latency)
                uint4 e;  // bytes b0..b3 of d0 and b4 = first byte of d1 -> pairs (b0,b1) (b1,b2) (b2,b3) (b3,b4)
                e.x = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c010c00u);
                e.y = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c020c01u);
                e.z = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c030c02u);
                e.w = __builtin_amdgcn_perm(d1[k], d0[k], 0x0c040c03u);
                *(uint4 *)(E + rr[k] * ES + 4 * qq[k]) = e;
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
        // unconditional stores to the clamped slot (lanes past the end rewrite the last key with itself): under a lane
        // condition the compiler sinks the load next to its store, behind a divergent branch
#pragma unroll
        for (int k = 0; k < 4; k++) dst[min(j0 + sub + 16 * k, cn - 1)] = v[k];
    }
}

// the tile strides of the usual 30-px cell grids + the run-time-stride instance
#define ORBX_FAST_INSTANCE(EST)                                                                                              \
    template __global__ void k_fast_cells<EST>(const uint8_t *, size_t, const LevelGeom *, int, int, uint32_t *, uint32_t *, \
                                               size_t, int, int, int, int, int, int, int, CellBases)
ORBX_FAST_INSTANCE(0);
ORBX_FAST_INSTANCE(44);
ORBX_FAST_INSTANCE(48);
ORBX_FAST_INSTANCE(52);
