// orbx_fast.hip — per-cell cv::FAST with threshold fallback (src/ORBextractor.cc:789-829): k_fast_strips, k_fast_cells, k_gather
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

// cornerScore<16> + 1 of a pixel PAIR from its centre dword vv and its 16 ring dwords (each holds the two pixels in 16-bit halves).
// A 16-bit half holding the integer n in [0,255] IS the f16 denormal n*2^-24, so the pixel pairs can be fed to the packed
// f16 pipe unchanged: 3-input minima/maxima (v_pk_minimum3_f16 / v_pk_maximum3_f16, gfx950), differences and negation are
// exact on these values, and a positive result's bit pattern is again the integer.
//
// cornerScore + 1 = max(dark, bright), dark = max over the 16 nine-arcs of min_k (v - r[k]), bright = max over the arcs of
// min_k (r[k] - v).  The centre is the same in every term, so it leaves the minima: dark = v - A, bright = B - v with
//     A = min over arcs of (max of the arc's ring pixels),   B = max over arcs of (min of the arc's ring pixels)
// computed on the ring values themselves: no per-position difference (16 packed subtractions per pair less).
// Two neighbouring arcs share eight elements: min(max arc_2j, max arc_2j+1) = max(C_j, min(r[2j], r[2j+9])) with
// C_j = max r[2j+1..2j+8], and C_j is two of the eight 4-windows q[t] = max r[2t+1..2t+4]: 36 packed ops per polarity.
// Returns the two values as signed 16-bit halves (negative = far from a corner).
__device__ __forceinline__ half2v fast_ring_score(uint32_t vv, const uint32_t (&rr)[16]) {
    const half2v v = __builtin_bit_cast(half2v, vv);
    half2v d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = __builtin_bit_cast(half2v, rr[k]);
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
    half2v lo[8], hi[8];   // per arc pair: the larger of the two arc minima / the smaller of the two arc maxima
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const half2v e0 = d[2 * t], e1 = d[(2 * t + 9) & 15];
        lo[t] = pk_min3(qmn[t], qmn[(t + 2) & 7], __builtin_elementwise_maximum(e0, e1));
        hi[t] = pk_max3(qmx[t], qmx[(t + 2) & 7], __builtin_elementwise_minimum(e0, e1));
    }
    const half2v Bv = pk_max3(pk_max3(lo[0], lo[1], lo[2]), pk_max3(lo[3], lo[4], lo[5]), __builtin_elementwise_maximum(lo[6], lo[7]));
    const half2v Av = pk_min3(pk_min3(hi[0], hi[1], hi[2]), pk_min3(hi[3], hi[4], hi[5]), __builtin_elementwise_minimum(hi[6], hi[7]));
    return __builtin_elementwise_maximum(v - Av, Bv - v);
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
    const short2v best = __builtin_bit_cast(short2v, fast_ring_score(vv, rr));
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
//
// SPARSE (round 4): the same kernel for the corner-sparse levels of the strip kernel's range (cells up to 32 px wide whose
// (image slot, level) the previous call flagged: a few candidates per cell, the regime of real footage).  There the 76-operation
// score is wasted on most pixels, and skipping it lane by lane saves nothing (an instruction issues for the whole wave), so the
// work is COMPACTED: (1) every pixel pair goes through the exact five-pixel bound on score + 1 (every nine-arc of the ring holds
// r[0] or r[8] and r[4] or r[12]; the necessary condition cv::FAST itself tests first), 9 packed operations instead of 76; (2) the
// pairs that can reach min(iniTh, minTh) are appended to an LDS queue in row-major order (ballot prefix); (3) the queue is scored
// 64 pairs at a time with per-lane ring addressing, the scores scattered into the zeroed score tile; (4) the 3x3 suppression and
// the ordered emission run over the queue entries only.  A pair that fails the bound has a score below both thresholds: the dense
// form writes 0 for it as well, so the score tile - and everything behind it - is identical.  sparseFlag selects the cells:
// SPARSE instances take the flagged (image, level)s of stripLevels, k_fast_strips (skipSparse) leaves exactly those alone.
template <int ES_T, bool SPARSE>
__global__ __launch_bounds__(64 * FAST_WAVES) void k_fast_cells(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels,
    int totalCells, uint32_t *__restrict__ cellCnt, uint32_t *__restrict__ cellRaw, uint32_t *__restrict__ slots, size_t slotsPerImg,
    int iniTh, int minTh, int ESrt, int SSrt, int tileRows, int ldsPerWave, int phaseLimit, CellBases cb, unsigned stripLevels,
    const int32_t *__restrict__ sparseFlag, FastHist fh) {
    const int ES = ES_T ? ES_T : ESrt, SS = ES_T ? ES_T - 8 : SSrt;
    extern __shared__ __align__(16) uint8_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);
    const int gc = bx * FAST_WAVES + wave;
    if (gc >= totalCells) return;  // wave-uniform; the kernel uses no block barrier
    const int l = level_of_cell(cb, nlevels, gc);
    if (SPARSE) {   // the flagged (image, level)s of the strip kernel's levels, nothing else
        if (!((stripLevels >> l) & 1u) || __builtin_amdgcn_readfirstlane(sparseFlag[b * nlevels + l]) == 0) return;
    } else if ((stripLevels >> l) & 1u) return;   // this level's cells are k_fast_strips' (cells up to 32 px wide)
    const LevelGeom g = geom[l];
    const int c = gc - g.cellBase;
    const int ci = c / g.nCols, cj = c - ci * g.nCols;
    const int maxBX = g.w - ORBX_MINB, maxBY = g.h - ORBX_MINB;
    const int iniY = ORBX_MINB + ci * g.hCell, iniX = ORBX_MINB + cj * g.wCell;
    int maxY = iniY + g.hCell + 6, maxX = iniX + g.wCell + 6;
    uint32_t *cnt = cellCnt + (size_t)b * totalCells + gc;
    uint32_t *raw = cellRaw + (size_t)b * totalCells + gc;   // list length | threshold choice << 31 (see k_gather)
    if (iniY >= maxBY - 3 || iniX >= maxBX - 6) {  // skipped rows / columns (:794-795,803-804)
        if (lane == 0) { *cnt = 0; *raw = 0; }
        return;
    }
    if (maxY > maxBY) maxY = maxBY;
    if (maxX > maxBX) maxX = maxBX;
    const int tw = maxX - iniX, th = maxY - iniY;  // FAST sub-image
    const int cw = tw - 6, ch = th - 6;            // evaluated area (rows/cols 3 .. dim-4)
    if (cw <= 0 || ch <= 0) {
        if (lane == 0) { *cnt = 0; *raw = 0; }
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
    // SPARSE: queue of the pairs that pass the bound, (row << 4 | pair) as 16-bit entries behind the score tile (capacity: every pair)
    uint16_t *Q = (uint16_t *)(smem + (size_t)wave * ldsPerWave + (size_t)ES * tileRows * 4 + (((size_t)SS * (tileRows - 4) + 15) & ~(size_t)15));
    int nQ = 0;
    if (SPARSE) {   // (cells of the strip levels: pw2 <= 16)
        const int j = lane & 15, r4 = lane >> 4;
        const half2v tl2 = __builtin_bit_cast(half2v, (uint32_t)tlo * 0x00010001u);
        const uint32_t vmask = (2 * j < cw ? 0xFFFFu : 0u) | (2 * j + 1 < cw ? 0xFFFF0000u : 0u);
        const uint32_t *q = E + (r4 + 3) * ES + 2 * j + 3 + sh;
        for (int r0 = 0; r0 < ch; r0 += 4, q += 4 * ES) {
            const int py = r0 + r4;
            // (rows past the cell: the window rows exist in the tile's allocation, the verdict is masked)
            const half2v v = __builtin_bit_cast(half2v, q[0]), a0 = __builtin_bit_cast(half2v, q[3 * ES]), a4 = __builtin_bit_cast(half2v, q[3]),
                         a8 = __builtin_bit_cast(half2v, q[-3 * ES]), a12 = __builtin_bit_cast(half2v, q[-3]);
            const half2v ub = __builtin_elementwise_minimum(__builtin_elementwise_maximum(a0, a8), __builtin_elementwise_maximum(a4, a12)) - v;
            const half2v ud = v - __builtin_elementwise_maximum(__builtin_elementwise_minimum(a0, a8), __builtin_elementwise_minimum(a4, a12));
            const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(ub, ud), tl2) - tl2) & vmask;
            const bool pass = u != 0 && py < ch;   // U > min(iniTh, minTh) for one of the pair's pixels inside the evaluated area
            const unsigned long long m = __ballot(pass);
            if (pass) Q[nQ + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)((py << 4) | j);
            nQ += __popcll(m);
        }
        wave_sync();
        for (int base = 0; base < nQ; base += 64) {
            const uint32_t e = Q[min(base + lane, nQ - 1)];   // lanes past the end redo the last entry (same values)
            fast_score_pair(E, ES, sh, Sc, SS, tlo, cw, (int)(e >> 4), 2 * (int)(e & 15u));
        }
    } else
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
    if (SPARSE) {   // only a queued pair can hold a survivor; the queue is in row-major order, so the list is too
        for (int base = 0; base < nQ; base += 64) {
            bool k0 = false, k1 = false;
            int v0 = 0, v1 = 0;
            const uint32_t e = Q[min(base + lane, nQ - 1)];
            const int py = (int)(e >> 4), px = 2 * (int)(e & 15u);
            if (base + lane < nQ) fast_nms_pair(Sc, SS, cw, py, px, k0, k1, v0, v1);
            k0 = k0 && v0 > 0;   // (a pair queued for its other pixel: a zero score is no strict maximum of anything it would be emitted for)
            k1 = k1 && v1 > 0;
            anyIni |= (__ballot((k0 && v0 >= iniTh) || (k1 && v1 >= iniTh)) != 0ull);
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1), lt = (1ull << lane) - 1ull;
            int pos = nL + __popcll(m0 & lt) + __popcll(m1 & lt);
            if (k0) Lst[pos++] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)v0 << 16);
            if (k1) Lst[pos] = (uint32_t)(px + 1) | ((uint32_t)py << 8) | ((uint32_t)v1 << 16);
            nL += __popcll(m0) + __popcll(m1);
        }
    } else if (pw2 <= 16) {
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
            if (pos < g.capc) {
                const uint32_t kx = (e & 0xFF) + 3 + cj * g.wCell, ky = ((e >> 8) & 0xFF) + 3 + ci * g.hCell;
                out[pos] = kx | (ky << 12) | ((e >> 16) << 24);
                if (fh.cnt) {   // (wave-uniform) quad-tree histogram at the L2: deepest cell of the key's path, count + best key (response, then
                                // the EARLIEST position in vToDistributeKeys order = smallest slot index: :744-760 keeps the first maximum)
                    const uint32_t cc = (uint32_t)fh.tab[g.xPathOff + kx] | (uint32_t)fh.tab[g.yPathOff + ky];
                    const size_t ho = ((size_t)b * nlevels + l) * fh.stride + cc;
                    atomicAdd(fh.cnt + ho, 1u);
                    atomicMax(fh.best + ho, ((e >> 16) << 24) | (0xFFFFFFu - (uint32_t)(c * g.capc + pos)));
                }
            }
        }
        total += __popcll(m);
    }
    // this kernel's list is already filtered by the cell's threshold: k_gather's filter passes every entry
    if (lane == 0) { *cnt = (uint32_t)min(total, g.capc); *raw = (uint32_t)min(total, g.capc) | (anyIni ? 0x80000000u : 0u); }
}

// ------------------------------------------------------------------------------------
// K2 for levels whose cells are at most 32 px wide (every level of the usual image sizes except the coarsest ones): ONE WAVE PER
// STRIP OF FOUR horizontally adjacent cells, the window rows streamed through LDS.
//   * lane = 16 * cell + pixel pair: a cell is one 16-lane DPP row, so the horizontal neighbours of the 3x3 non-maximum
//     suppression come from row_shr:1 / row_shl:1 with zero fill at the row ends - exactly cv::FAST's "scores outside the
//     evaluated area are 0" at a cell seam - and the vertical neighbours are the lane's own previous / next iteration: the
//     scores never leave the registers (k_fast_cells: a byte tile in LDS, zeroed, written, read back with 9 v_perm per pair);
//   * the iteration is one evaluated ROW of the four cells (30 rows x 16 lanes per cell instead of 8 x 64: no half-empty last
//     iteration), reading the 7 window rows it needs from a ring of 8 rows in LDS; the next window row is in flight from global
//     memory while a row is scored (k_fast_cells: the whole window first, a fifth of its time spent waiting for it).  Rows 0..5 of
//     the ring are mirrored behind it, so that any 7 consecutive rows are contiguous: one base register + immediates;
//   * the threshold is not applied before the suppression: a pixel that passes the threshold t beats every neighbour below t
//     anyway, and one that does not is never emitted, so NMS on the raw scores keeps the same pixels;
//   * survivors with score >= min(iniTh, minTh) go straight to the cell's slot list in row-major order (ballot prefix inside the
//     cell's 16 lanes); the per-cell threshold fallback (:809-816) is decided at the end of the strip from per-lane counters
//     (nA: survivors >= iniTh, nB: >= minTh) and applied by k_gather while it compacts: count = nA ? nA : nB.
// tools/phase_count.py: assembler comments at the phase boundaries of k_fast_strips (-DORBX_PHASE_MARKERS only: a volatile asm is a scheduling barrier)
#ifdef ORBX_PHASE_MARKERS
#define FPHASE(name) asm volatile("; ORBX_PHASE " name)
#else
#define FPHASE(name)
#endif
#define STRIP_ES 144      // dwords per tile row: 4 x 32 px + 6 px of window + alignment shift (<= 3) + the last pair's partner, 16-B rows
#define STRIP_SLOTS 8     // ring of 8 window rows (the loop is unrolled by 8 rows, so every ring slot is a compile-time offset)
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }

__global__ __launch_bounds__(64 * FAST_WAVES) void k_fast_strips(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels, int totalStrips,
    int totalCells, uint32_t *__restrict__ cellCnt, uint32_t *__restrict__ cellRaw, uint32_t *__restrict__ slots, size_t slotsPerImg,
    int iniTh, int minTh, StripBases sb, const int32_t *__restrict__ sparseFlag, int strip0, int skipSparse) {
    __shared__ __align__(16) uint32_t smem[FAST_WAVES * STRIP_SLOTS * STRIP_ES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);
    const int strip = strip0 + bx * FAST_WAVES + wave;   // strip0: first strip of this launch (a call may launch the large levels' strips first)
    if (strip >= totalStrips) return;   // wave-uniform; no block barrier in this kernel (totalStrips: end of this launch's range)
    int l = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && strip >= sb.v[i]) ? 1 : 0;
    const LevelGeom g = geom[l];
    // Corner-sparse level (the previous call found few candidates in this image slot's level, k_gather leaves the verdict): every
    // row is first put through a cheap bound on the score and scored only if some pixel of the wave's 128 can reach the threshold.
    const bool pretest = sparseFlag != nullptr && __builtin_amdgcn_readfirstlane(sparseFlag[b * nlevels + l]) != 0;   // wave-uniform
    if (pretest && skipSparse) return;   // this (image, level) is the compaction kernel's (k_fast_cells<.., true> of the same call)
    const int s = strip - sb.v[l];
    const int ng = (g.nCols + 3) >> 2;
    const int ci = s / ng, cj0 = 4 * (s - ci * ng);
    const int c = lane >> 4, j = lane & 15, cj = cj0 + c;
    const bool exists = cj < g.nCols;
    const int maxBX = g.w - ORBX_MINB, maxBY = g.h - ORBX_MINB;
    const int iniY = ORBX_MINB + ci * g.hCell, Xs = ORBX_MINB + cj0 * g.wCell;
    const int th = min(iniY + g.hCell + 6, maxBY) - iniY, ch = th - 6;          // FAST sub-image rows, evaluated rows
    const int cw = exists ? min(g.wCell, maxBX - (ORBX_MINB + cj * g.wCell) - 6) : 0;   // evaluated columns of MY cell (<= 0: skipped, :803-804)
    const int cw0 = min(g.wCell, maxBX - Xs - 6);                               // ... of the strip's first cell (wave-uniform)
    const size_t cellIdx = (size_t)b * totalCells + g.cellBase + ci * g.nCols + cj;
    if (iniY >= maxBY - 3 || ch <= 0 || cw0 <= 0) {   // skipped rows (:794-795) / nothing to evaluate
        if (j == 0 && exists) { cellCnt[cellIdx] = 0; cellRaw[cellIdx] = 0; }
        return;
    }
    const int cjL = min(cj0 + 3, g.nCols - 1);
    const int tws = min(ORBX_MINB + cjL * g.wCell + g.wCell + 6, maxBX) - Xs;   // strip window width: its cells' windows overlap by 6 px

    // ---- window rows: aligned dword pairs -> four pixel-PAIR dwords -> ring (pstride % 4 == 0: every row has the same misalignment)
    uint32_t *E = smem + wave * (STRIP_SLOTS * STRIP_ES);
    const size_t a = (size_t)(ORBX_EDGE + iniY) * g.pstride + ORBX_EDGE + Xs;
    const int sh = (int)(a & 3);
    const uint8_t *src = pyr + (size_t)b * pyrImgBytes + g.poff + (a - sh);
    const int nd = (sh + tws + 3) >> 2;                       // aligned source dwords per window row (<= 36)
    const int qi = min(lane, nd - 1);                         // lanes past the row repeat its last item (same value, same address)
    const uint32_t loff = (uint32_t)qi * 4u;
    const int chS = __builtin_amdgcn_readfirstlane(ch), thS = __builtin_amdgcn_readfirstlane(th), pstrideS = __builtin_amdgcn_readfirstlane(g.pstride);
    // everything that depends on the row number alone is kept on the scalar unit (readfirstlane): the loop counter, the ring
    // slot, the row's byte offset (a padded level is far smaller than 4 GiB) - the vector ALU is the kernel's bottleneck
    auto load_row = [&](int r) -> uint2 {
        const uint32_t ro = (uint32_t)(min(r, thS - 1) * pstrideS);
        const uint32_t *p = (const uint32_t *)(src + ro + loff);   // scalar row base + 32-bit lane offset
        uint2 d;
        d.x = p[0]; d.y = p[1];
        return d;
    };
    uint32_t *Ew = E + 4 * qi;
    auto write_row = [&](int slot, uint2 d) {   // slot = row & 7, a compile-time constant at every call site
        uint4 e;  // bytes b0..b3 of d.x and b4 = first byte of d.y -> pairs (b0,b1) (b1,b2) (b2,b3) (b3,b4)
        e.x = __builtin_amdgcn_perm(d.y, d.x, 0x0c010c00u);
        e.y = __builtin_amdgcn_perm(d.y, d.x, 0x0c020c01u);
        e.z = __builtin_amdgcn_perm(d.y, d.x, 0x0c030c02u);
        e.w = __builtin_amdgcn_perm(d.y, d.x, 0x0c040c03u);
        *(uint4 *)(Ew + slot * STRIP_ES) = e;
    };
    {   // rows 0..6 (the first evaluated row's ring) + row 7 in flight: all eight loads issued before the first LDS write
        uint2 d[7];
#pragma unroll
        for (int r = 0; r < 7; r++) d[r] = load_row(r);
#pragma unroll
        for (int r = 0; r < 7; r++) write_row(r, d[r]);   // rows 0..6 -> slots 0..6
    }
    uint2 pre = load_row(7);
    wave_sync();

    const int tlo = max(min(iniTh, minTh), 0), thi = max(iniTh, minTh);
    const int i0 = c * g.wCell + 3 + 2 * j;                   // strip column of my pair's first pixel
    const uint32_t *q0 = E + sh + i0;
    const uint32_t vmask = (2 * j < cw ? 0xFFFFu : 0u) | (2 * j + 1 < cw ? 0xFFFF0000u : 0u);
    // the lanes of my cell below me / up to me
    const uint32_t ltc = (1u << j) - 1u;
    const int cs = 16 * c;   // my cell's 16 lanes inside a ballot
    const half2v tl2 = __builtin_bit_cast(half2v, (uint32_t)tlo * 0x00010001u);
    const uint32_t x0 = (uint32_t)(cj * g.wCell + 3 + 2 * j), ybase = (uint32_t)(ci * g.hCell + 3);
    // slot list of my cell: scalar base of the level's slots + 32-bit lane offset
    uint32_t *lvlSlots = slots + (size_t)b * slotsPerImg + g.slotOff;
    const uint32_t cellOff = (uint32_t)((ci * g.nCols + cj) * g.capc);
    // NMS state: S1 = scores (+1) of row y-1, H1 / H2 = 3-wide horizontal maxima of rows y-1 / y-2, LR1 = max(left, right) of row y-1
    uint32_t S1 = 0, H1 = 0, H2 = 0, LR1 = 0;
    uint32_t nRaw = 0, nHi = 0;   // survivors >= min(iniTh, minTh) so far in my cell (same in its 16 lanes) / MY survivors >= max(iniTh, minTh)
    // one evaluated row: scores of row y, suppression verdict for row y-1 (sign flags of its two pixels, its scores), ring update
    // R = y & 7 is a compile-time constant at every call site (the loop below advances by 8 rows): window row y + k sits in ring
    // slot (R + k) & 7, an immediate offset from ONE base register - no mirror of the ring's first rows, no per-row address
    auto rowstep = [&](const int R, int y, uint32_t &kpOut, uint32_t &sOut) {
        uint32_t S = 0;
        FPHASE("ring_reads");
        if (y < chS) {   // wave-uniform
            const uint32_t *q = q0;
#define SLOT(k) (((R + (k)) & 7) * STRIP_ES)
            const uint32_t vv = q[SLOT(3)];
            uint32_t rr[16];
            rr[0] = q[SLOT(6)];  rr[4] = q[SLOT(3) + 3];  rr[8] = q[SLOT(0)];  rr[12] = q[SLOT(3) - 3];
            bool go = true;
            FPHASE("row_pretest");
            if (pretest) {   // wave-uniform
                // Every nine-arc of the ring holds r[0] or r[8] and r[4] or r[12].  So bright = (max over arcs of the arc's minimum) - v
                // <= min(max(r0, r8), max(r4, r12)) - v and dark = v - (min over arcs of the arc's maximum) <= v - max(min(r0, r8),
                // min(r4, r12)): an upper bound U of score + 1 from five pixels (the necessary condition cv::FAST itself tests first).
                // A pixel with U <= min(iniTh, minTh) has a score below both thresholds: it is never emitted and never beats a
                // neighbour that is, so its score may be taken as 0; when that holds for all 128 pixels of the row the 76-operation
                // score is skipped.  Same keypoints by construction.
                const half2v v = __builtin_bit_cast(half2v, vv), a0 = __builtin_bit_cast(half2v, rr[0]), a4 = __builtin_bit_cast(half2v, rr[4]),
                             a8 = __builtin_bit_cast(half2v, rr[8]), a12 = __builtin_bit_cast(half2v, rr[12]);
                const half2v ub = __builtin_elementwise_minimum(__builtin_elementwise_maximum(a0, a8), __builtin_elementwise_maximum(a4, a12)) - v;
                const half2v ud = v - __builtin_elementwise_maximum(__builtin_elementwise_minimum(a0, a8), __builtin_elementwise_minimum(a4, a12));
                const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(ub, ud), tl2) - tl2) & vmask;
                go = __ballot(u != 0) != 0;   // U - tlo > 0 for some pixel of the evaluated area (exact on these values: max(U, tlo) - tlo >= 0)
            }
            FPHASE("ring_reads");
            if (go) {
                rr[1] = q[SLOT(6) + 1];   rr[2] = q[SLOT(5) + 2];   rr[3] = q[SLOT(4) + 3];
                rr[5] = q[SLOT(2) + 3];   rr[6] = q[SLOT(1) + 2];   rr[7] = q[SLOT(0) + 1];
                rr[9] = q[SLOT(0) - 1];   rr[10] = q[SLOT(1) - 2];  rr[11] = q[SLOT(2) - 3];
                rr[13] = q[SLOT(4) - 3];  rr[14] = q[SLOT(5) - 2];  rr[15] = q[SLOT(6) - 1];
                FPHASE("score");
                const half2v best = fast_ring_score(vv, rr);
                // score + 1, clamped at 0, pixels outside the cell's evaluated area = 0
                S = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(best, (half2v){(_Float16)0, (_Float16)0})) & vmask;
            }
#undef SLOT
        }
        // row y: (left neighbour's second pixel, my first) and (my second, right neighbour's first); zero beyond the cell
        FPHASE("suppression");
        const uint32_t Lp = __builtin_amdgcn_perm(S, dpp_row_shr1(S), 0x05040302u);   // bytes: shr.hi | S.lo << 16
        const uint32_t Rp = __builtin_amdgcn_perm(dpp_row_shl1(S), S, 0x05040302u);   // bytes: S.hi | shl.lo << 16
        const half2v Lh = __builtin_bit_cast(half2v, Lp), Rh = __builtin_bit_cast(half2v, Rp), Sh = __builtin_bit_cast(half2v, S);
        const uint32_t H0 = __builtin_bit_cast(uint32_t, pk_max3(Lh, Sh, Rh));
        // max(left, right, tlo): the threshold min(iniTh, minTh) rides along as a third "neighbour" (score + 1 > tlo <=> score >= tlo)
        const uint32_t LR0 = __builtin_bit_cast(uint32_t, pk_max3(Lh, Rh, tl2));
        // strict 3x3 maximum of row y-1 (rows y-2 and y through their horizontal maxima) and score >= tlo: ONE sign test per
        // pixel (exact on these values; a negative f16 is a negative int16)
        const half2v nb = pk_max3(__builtin_bit_cast(half2v, H2), __builtin_bit_cast(half2v, H0), __builtin_bit_cast(half2v, LR1));
        kpOut = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, S1) - nb);
        sOut = S1;
        S1 = S; H2 = H1; H1 = H0; LR1 = LR0;
        // stream: window row y+7 (loaded during this iteration) replaces row y-1 in the ring; row y+8 goes in flight
        FPHASE("row_stream");
        if (y + 7 < thS) write_row((R + 7) & 7, pre);   // wave-uniform
        pre = load_row(y + 8);
        wave_sync();
    };
    // Two rows per emission: the two pixels of a pair and the pair of the next row are all neighbours of each other, so among the
    // four at most one is a strict 3x3 maximum - a lane emits at most one key per two rows.  Eight rows per loop iteration, so that
    // a row's ring slots are compile-time constants.
    auto rowpair = [&](const int R, int y) {     // rows y, y + 1 (R = y & 7): verdicts for rows y - 1 and y
        uint32_t kpA, sA, kpB, sB;
        rowstep(R, y, kpA, sA);          // verdict for row y-1
        rowstep(R + 1, y + 1, kpB, sB);  // verdict for row y   (y + 1 > chS: an all-zero row, nothing survives)
        FPHASE("emission");
        const short2v fa = __builtin_bit_cast(short2v, kpA), fb = __builtin_bit_cast(short2v, kpB);
        const bool inA = fa.x > 0 || fa.y > 0, inB = fb.x > 0 || fb.y > 0;
        const unsigned long long mA = __ballot(inA), mB = __ballot(inB);
        if (mA | mB) {   // wave-uniform
            const uint32_t aA = (uint32_t)(mA >> cs) & 0xFFFFu, aB = (uint32_t)(mB >> cs) & 0xFFFFu;   // my cell's 16 lanes of the ballots
            const bool second = inB ? fb.y > 0 : fa.y > 0;                       // my survivor is the pair's second pixel
            const uint32_t sv = inB ? sB : sA;
            const uint32_t sc = second ? sv >> 16 : sv & 0xFFFFu;                // its score + 1
            // row-major order: row y-1 before row y, inside a row the survivors of my cell's lower lanes first
            const uint32_t pos = cellOff + nRaw + (inB ? __popc(aA) + __popc(aB & ltc) : __popc(aA & ltc));
            // (a cell cannot hold more than capc = ceil(w/2) * ceil(h/2) strict 3x3 maxima: no two of them are neighbours)
            if (inA || inB)
                lvlSlots[pos] = (sc << 24) + (second ? x0 + 1u : x0) + ((ybase + (uint32_t)(y - 1) + (inB ? 1u : 0u)) << 12) - (1u << 24);
            nHi += (inA || inB) && sc > (uint32_t)thi ? 1u : 0u;
            nRaw += __popc(aA) + __popc(aB);
        }
    };
    FPHASE("loop");
    for (int yv = 0; yv <= chS; yv += 8) {
        const int y = __builtin_amdgcn_readfirstlane(yv);   // everything derived from the row number stays on the scalar unit
        FPHASE("loop");
        rowpair(0, y);
        if (y + 2 <= chS) rowpair(2, y + 2);   // wave-uniform
        if (y + 4 <= chS) rowpair(4, y + 4);
        if (y + 6 <= chS) rowpair(6, y + 6);
    }
    // per-cell totals: the 16 lanes of a cell
    FPHASE("epilogue");
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nHi += __shfl_xor(nHi, o);
    if (j == 0 && exists) {
        // {>= iniTh} if non-empty else {>= minTh} (:809-816): one of the two sets is the whole list, the other the nHi entries
        const uint32_t nA = iniTh >= minTh ? nHi : nRaw, nB = iniTh >= minTh ? nRaw : nHi;
        cellCnt[cellIdx] = min(nA ? nA : nB, (uint32_t)g.capc);
        cellRaw[cellIdx] = min(nRaw, (uint32_t)g.capc) | (nA ? 0x80000000u : 0u);
    }
}

// ------------------------------------------------------------------------------------
// K2s: the strip kernel for CORNER-SPARSE levels (a few FAST candidates per cell: real footage; the (image slot, level)s the
// previous call's quad-tree flagged).  Same wave = strip of four cells, same lane = pixel pair, same suppression in registers and
// same ordered emission as k_fast_strips - but the 76-operation score is computed only where it can matter, and COMPACTED:
//   * every pair of a row goes through the five-pixel bound on score + 1 (every nine-arc of the ring holds r[0] or r[8] and r[4]
//     or r[12]: the necessary condition cv::FAST itself tests first; exact - a pair that fails it has a score below both thresholds,
//     is never emitted and never beats a pixel that is, so its score may be taken as 0);
//   * the pairs that pass are appended to an LDS queue (ballot prefix), eight evaluated rows at a time; the queue is then scored
//     64 pairs per round with per-lane ring addresses - skipping the score lane by lane would save nothing, an instruction issues for
//     the whole wave - and the scores land in a sparse tile (one dword per lane and row, zeroed by the pre-test);
//   * the suppression / emission of the block's rows then runs exactly as in k_fast_strips with the row's scores read from the tile.
// The window rows are kept as BYTES (a ring of 16 rows + a mirror of its first six, so that any seven consecutive rows are one
// base address + immediates): 6.2 KB of LDS per wave.  A packed pixel pair (two 16-bit halves, the f16-denormal trick of
// fast_ring_score) costs one 16-bit LDS read and one v_perm (lds_pair).
#define SP_RS 144          // ring row stride in bytes (>= 3 + 4 * 32 + 6 + 4)
#define SP_SLOTS 16
#define SP_MIRROR 6
#define SP_K 8             // evaluated rows per block
#define SP_LDS_PER_WAVE ((SP_SLOTS + SP_MIRROR) * SP_RS + SP_K * 64 * 2 + SP_K * 64 * 2 + 16)
// packed pixel pair (p[0] | p[1] << 16) from two neighbouring ring bytes: two ds_read_u8 and one v_lshl_or.  Measured alternatives:
// ONE unaligned ds_read_u16 + v_perm made the kernel 4x slower (0.95 instead of 0.21 ms per 128 images: misaligned LDS accesses
// are replayed), and ds_read_u8_d16 + ds_read_u8_d16_hi - which would build the pair in the LDS unit with no VALU work - ZERO the
// other half of the register on gfx950 (d16 loads keep it only without SRAM ECC): the first version read 0 in every low half.
__device__ __forceinline__ uint32_t lds_pair(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 16); }

__global__ __launch_bounds__(64 * FAST_WAVES) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_fast_strips_sparse(
    const uint8_t *__restrict__ pyr, size_t pyrImgBytes, const LevelGeom *__restrict__ geom, int nlevels, int totalStrips,
    int totalCells, uint32_t *__restrict__ cellCnt, uint32_t *__restrict__ cellRaw, uint32_t *__restrict__ slots, size_t slotsPerImg,
    int iniTh, int minTh, StripBases sb, const int32_t *__restrict__ sparseFlag) {
    __shared__ __align__(16) uint8_t smem[FAST_WAVES * SP_LDS_PER_WAVE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);
    const int strip = bx * FAST_WAVES + wave;
    if (strip >= totalStrips) return;   // wave-uniform; no block barrier in this kernel
    int l = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && strip >= sb.v[i]) ? 1 : 0;
    if (__builtin_amdgcn_readfirstlane(sparseFlag[b * nlevels + l]) == 0) return;   // k_fast_strips' (wave-uniform)
    const LevelGeom g = geom[l];
    const int s = strip - sb.v[l];
    const int ng = (g.nCols + 3) >> 2;
    const int ci = s / ng, cj0 = 4 * (s - ci * ng);
    const int c = lane >> 4, j = lane & 15, cj = cj0 + c;
    const bool exists = cj < g.nCols;
    const int maxBX = g.w - ORBX_MINB, maxBY = g.h - ORBX_MINB;
    const int iniY = ORBX_MINB + ci * g.hCell, Xs = ORBX_MINB + cj0 * g.wCell;
    const int th = min(iniY + g.hCell + 6, maxBY) - iniY, ch = th - 6;          // FAST sub-image rows, evaluated rows
    const int cw = exists ? min(g.wCell, maxBX - (ORBX_MINB + cj * g.wCell) - 6) : 0;   // evaluated columns of MY cell (<= 0: skipped, :803-804)
    const int cw0 = min(g.wCell, maxBX - Xs - 6);                               // ... of the strip's first cell (wave-uniform)
    const size_t cellIdx = (size_t)b * totalCells + g.cellBase + ci * g.nCols + cj;
    if (iniY >= maxBY - 3 || ch <= 0 || cw0 <= 0) {   // skipped rows (:794-795) / nothing to evaluate
        if (j == 0 && exists) { cellCnt[cellIdx] = 0; cellRaw[cellIdx] = 0; }
        return;
    }
    const int cjL = min(cj0 + 3, g.nCols - 1);
    const int tws = min(ORBX_MINB + cjL * g.wCell + g.wCell + 6, maxBX) - Xs;   // strip window width: its cells' windows overlap by 6 px

    uint8_t *Rg = smem + wave * SP_LDS_PER_WAVE;                              // ring [16 + 6][SP_RS] bytes
    uint16_t *Tile = (uint16_t *)(Rg + (SP_SLOTS + SP_MIRROR) * SP_RS);        // [SP_K][64] scores (+1, <= 255) of a block's rows, two bytes per lane
    uint16_t *Q = (uint16_t *)(Tile + SP_K * 64);                             // [SP_K * 64] entries: row-in-block << 8 | lane << 2 | valid pixels
    uint32_t *RowHit = (uint32_t *)(Q + SP_K * 64);                           // bit r: row r of the block holds a score >= min(iniTh, minTh)
    const size_t a = (size_t)(ORBX_EDGE + iniY) * g.pstride + ORBX_EDGE + Xs;
    const int sh = (int)(a & 3);
    const uint8_t *src = pyr + (size_t)b * pyrImgBytes + g.poff + (a - sh);
    const int nd = (sh + tws + 3) >> 2;                       // aligned source dwords per window row (<= 36)
    const int qi = min(lane, nd - 1);                         // lanes past the row repeat its last item (same value, same address)
    const uint32_t loff = (uint32_t)qi * 4u;
    const int chS = __builtin_amdgcn_readfirstlane(ch), thS = __builtin_amdgcn_readfirstlane(th), pstrideS = __builtin_amdgcn_readfirstlane(g.pstride);
    auto load_row = [&](int r) -> uint32_t {
        const uint32_t ro = (uint32_t)(min(r, thS - 1) * pstrideS);
        return *(const uint32_t *)(src + ro + loff);          // scalar row base + 32-bit lane offset
    };
    auto write_row = [&](int r, uint32_t d) {                 // window row r -> ring slot r & 15 (+ its mirror behind the ring)
        const int slot = r & (SP_SLOTS - 1);
        *(uint32_t *)(Rg + slot * SP_RS + loff) = d;
        if (slot < SP_MIRROR) *(uint32_t *)(Rg + (slot + SP_SLOTS) * SP_RS + loff) = d;   // wave-uniform
    };
    {   // window rows 0..13 = the first block's
        uint32_t d[SP_K + 6];
#pragma unroll
        for (int r = 0; r < SP_K + 6; r++) d[r] = load_row(r);
#pragma unroll
        for (int r = 0; r < SP_K + 6; r++) write_row(r, d[r]);
    }
    wave_sync();

    const int tlo = max(min(iniTh, minTh), 0), thi = max(iniTh, minTh);
    const int i0 = c * g.wCell + 3 + 2 * j;                   // strip column of my pair's first pixel
    const uint8_t *myCol = Rg + sh + i0;                      // my pair in ring slot 0
    const uint32_t vbits = (2 * j < cw ? 1u : 0u) | (2 * j + 1 < cw ? 2u : 0u);
    const uint32_t vmask = (vbits & 1u ? 0xFFFFu : 0u) | (vbits & 2u ? 0xFFFF0000u : 0u);
    const uint32_t ltc = (1u << j) - 1u;
    const int cs = 16 * c;   // my cell's 16 lanes inside a ballot
    const half2v tl2 = __builtin_bit_cast(half2v, (uint32_t)tlo * 0x00010001u);
    const uint32_t x0 = (uint32_t)(cj * g.wCell + 3 + 2 * j), ybase = (uint32_t)(ci * g.hCell + 3);
    uint32_t *lvlSlots = slots + (size_t)b * slotsPerImg + g.slotOff;
    const uint32_t cellOff = (uint32_t)((ci * g.nCols + cj) * g.capc);
    const uint32_t wCellS = (uint32_t)__builtin_amdgcn_readfirstlane(g.wCell);
    uint32_t S1 = 0, H1 = 0, H2 = 0, LR1 = 0;
    uint32_t nRaw = 0, nHi = 0;
    uint32_t hitPrev = 0;   // the last row of the previous block held a score above the threshold
    // suppression verdict for row y-1 from the scores S of row y (identical to k_fast_strips' rowstep behind its score)
    auto nms_row = [&](uint32_t S, uint32_t &kpOut, uint32_t &sOut) {
        const uint32_t Lp = __builtin_amdgcn_perm(S, dpp_row_shr1(S), 0x05040302u);
        const uint32_t Rp = __builtin_amdgcn_perm(dpp_row_shl1(S), S, 0x05040302u);
        const half2v Lh = __builtin_bit_cast(half2v, Lp), Rh = __builtin_bit_cast(half2v, Rp), Sh = __builtin_bit_cast(half2v, S);
        const uint32_t H0 = __builtin_bit_cast(uint32_t, pk_max3(Lh, Sh, Rh));
        const uint32_t LR0 = __builtin_bit_cast(uint32_t, pk_max3(Lh, Rh, tl2));
        const half2v nb = pk_max3(__builtin_bit_cast(half2v, H2), __builtin_bit_cast(half2v, H0), __builtin_bit_cast(half2v, LR1));
        kpOut = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, S1) - nb);
        sOut = S1;
        S1 = S; H2 = H1; H1 = H0; LR1 = LR0;
    };
    auto emit_pair = [&](int y, uint32_t kpA, uint32_t sA, uint32_t kpB, uint32_t sB) {   // verdicts for rows y - 1 (A) and y (B)
        const short2v fa = __builtin_bit_cast(short2v, kpA), fb = __builtin_bit_cast(short2v, kpB);
        const bool inA = fa.x > 0 || fa.y > 0, inB = fb.x > 0 || fb.y > 0;
        const unsigned long long mA = __ballot(inA), mB = __ballot(inB);
        if (mA | mB) {   // wave-uniform
            const uint32_t aA = (uint32_t)(mA >> cs) & 0xFFFFu, aB = (uint32_t)(mB >> cs) & 0xFFFFu;
            const bool second = inB ? fb.y > 0 : fa.y > 0;
            const uint32_t sv = inB ? sB : sA;
            const uint32_t sc = second ? sv >> 16 : sv & 0xFFFFu;
            const uint32_t pos = cellOff + nRaw + (inB ? __popc(aA) + __popc(aB & ltc) : __popc(aA & ltc));
            if (inA || inB)
                lvlSlots[pos] = (sc << 24) + (second ? x0 + 1u : x0) + ((ybase + (uint32_t)(y - 1) + (inB ? 1u : 0u)) << 12) - (1u << 24);
            nHi += (inA || inB) && sc > (uint32_t)thi ? 1u : 0u;
            nRaw += __popc(aA) + __popc(aB);
        }
    };

    for (int yv = 0; yv <= chS; yv += SP_K) {
        const int y0 = __builtin_amdgcn_readfirstlane(yv);
        // the next block's new window rows (y0 + 14 .. y0 + 21) go in flight now and into the ring behind this block's scoring
        uint32_t pre[SP_K];
#pragma unroll
        for (int r = 0; r < SP_K; r++) pre[r] = load_row(y0 + SP_K + 6 + r);
        // ---- pre-test of the block's rows, queue of the pairs that pass, zeroed tile
        int nQ = 0;
        if (lane == 0) *RowHit = 0;
#pragma unroll
        for (int r = 0; r < SP_K; r++) {
            Tile[r * 64 + lane] = 0;
            if (y0 + r < chS) {   // wave-uniform
                const uint8_t *ar = myCol + ((y0 + r) & (SP_SLOTS - 1)) * SP_RS;   // window row y0 + r (= ring row of dy = -3)
                const uint32_t v = lds_pair(ar + 3 * SP_RS), a0 = lds_pair(ar + 6 * SP_RS), a8 = lds_pair(ar),
                               a4 = lds_pair(ar + 3 * SP_RS + 3), a12 = lds_pair(ar + 3 * SP_RS - 3);
                const half2v vh = __builtin_bit_cast(half2v, v), h0 = __builtin_bit_cast(half2v, a0), h4 = __builtin_bit_cast(half2v, a4),
                             h8 = __builtin_bit_cast(half2v, a8), h12 = __builtin_bit_cast(half2v, a12);
                const half2v ub = __builtin_elementwise_minimum(__builtin_elementwise_maximum(h0, h8), __builtin_elementwise_maximum(h4, h12)) - vh;
                const half2v ud = vh - __builtin_elementwise_maximum(__builtin_elementwise_minimum(h0, h8), __builtin_elementwise_minimum(h4, h12));
                const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(ub, ud), tl2) - tl2) & vmask;
                const bool pass = u != 0;   // U > min(iniTh, minTh) for one of my two pixels inside the evaluated area
                const unsigned long long m = __ballot(pass);
                if (pass) Q[nQ + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                              (uint16_t)((r << 8) | (lane << 2) | (int)vbits);
                nQ += __popcll(m);
            }
        }
        wave_sync();
        // ---- the queue, 64 pairs per round: the full score with per-lane ring addresses
        for (int base = 0; base < nQ; base += 64) {
            const uint32_t e = Q[min(base + lane, nQ - 1)];   // lanes past the end redo the last entry (same value to the same place)
            const uint32_t er = e >> 8, el = (e >> 2) & 63u, ev = e & 3u;
            // ring address of the entry's pair at window row y0 + er: its lane's column + its row's slot (the mirror makes the seven
            // rows contiguous: one address + immediates)
            const uint8_t *ea = Rg + sh + 3 + (el >> 4) * wCellS + 2u * (el & 15u) + (((uint32_t)y0 + er) & (SP_SLOTS - 1)) * SP_RS;
            const uint32_t vv = lds_pair(ea + 3 * SP_RS);
            uint32_t rr[16];
            rr[0] = lds_pair(ea + 6 * SP_RS);       rr[1] = lds_pair(ea + 6 * SP_RS + 1);  rr[2] = lds_pair(ea + 5 * SP_RS + 2);
            rr[3] = lds_pair(ea + 4 * SP_RS + 3);   rr[4] = lds_pair(ea + 3 * SP_RS + 3);  rr[5] = lds_pair(ea + 2 * SP_RS + 3);
            rr[6] = lds_pair(ea + 1 * SP_RS + 2);   rr[7] = lds_pair(ea + 1);              rr[8] = lds_pair(ea);
            rr[9] = lds_pair(ea - 1);               rr[10] = lds_pair(ea + 1 * SP_RS - 2); rr[11] = lds_pair(ea + 2 * SP_RS - 3);
            rr[12] = lds_pair(ea + 3 * SP_RS - 3);  rr[13] = lds_pair(ea + 4 * SP_RS - 3); rr[14] = lds_pair(ea + 5 * SP_RS - 2);
            rr[15] = lds_pair(ea + 6 * SP_RS - 1);
            const half2v best = fast_ring_score(vv, rr);
            const uint32_t em = (ev & 1u ? 0xFFFFu : 0u) | (ev & 2u ? 0xFFFF0000u : 0u);
            const uint32_t S = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(best, (half2v){(_Float16)0, (_Float16)0})) & em;
            Tile[er * 64 + el] = (uint16_t)__builtin_amdgcn_perm(0u, S, 0x0c0c0200u);   // (lo, hi) -> two bytes
            // rows that hold a score + 1 above min(iniTh, minTh): only they (and their neighbours) need the suppression below
            const uint32_t over = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(half2v, S), tl2) - tl2);
            if (over != 0) __hip_atomic_fetch_or(RowHit, 1u << er, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_sync();
        // ---- the new window rows take the slots of rows this block is done with
#pragma unroll
        for (int r = 0; r < SP_K; r++)
            if (y0 + SP_K + 6 + r < thS) write_row(y0 + SP_K + 6 + r, pre[r]);   // wave-uniform
        // ---- suppression + ordered emission, rows y0 .. y0 + 7 (rows past the cell are all-zero: the last verdicts).  A row without
        // a score above the threshold can neither hold a survivor nor beat one (the threshold rides along as a neighbour), i.e. it
        // may be taken as all-zero: a pair of rows is processed only if one of them or the row above holds such a score - in
        // corner-sparse scenes most do not - and otherwise just leaves the state of two zero rows behind.
        const uint32_t hits = ((uint32_t)__builtin_amdgcn_readfirstlane((int)*RowHit) << 1) | hitPrev;   // bit r + 1: row y0 + r, bit 0: row y0 - 1
        hitPrev = (hits >> SP_K) & 1u;
#pragma unroll
        for (int r = 0; r < SP_K; r += 2) {
            if (y0 + r <= chS) {   // wave-uniform
                if ((hits >> r) & 7u) {   // rows y0 + r - 1, y0 + r, y0 + r + 1 (wave-uniform)
                    uint32_t kpA, sA, kpB, sB;
                    nms_row(__builtin_amdgcn_perm(0u, (uint32_t)Tile[r * 64 + lane], 0x0c010c00u), kpA, sA);         // scores of row y0 + r: verdict for row y0 + r - 1
                    nms_row(__builtin_amdgcn_perm(0u, (uint32_t)Tile[(r + 1) * 64 + lane], 0x0c010c00u), kpB, sB);   // verdict for row y0 + r
                    emit_pair(y0 + r, kpA, sA, kpB, sB);
                } else { S1 = 0; H1 = 0; H2 = 0; LR1 = __builtin_bit_cast(uint32_t, tl2); }
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nHi += __shfl_xor(nHi, o);
    if (j == 0 && exists) {
        const uint32_t nA = iniTh >= minTh ? nHi : nRaw, nB = iniTh >= minTh ? nRaw : nHi;
        cellCnt[cellIdx] = min(nA ? nA : nB, (uint32_t)g.capc);
        cellRaw[cellIdx] = min(nRaw, (uint32_t)g.capc) | (nA ? 0x80000000u : 0u);
    }
}

// ------------------------------------------------------------------------------------
// K2c: the cell lists -> the level's key array in vToDistributeKeys order (:789-828), applying the per-cell threshold
// (cellRaw bit 31: some survivor reached iniTh -> keep score >= iniTh, else score >= minTh).  A block = 16 consecutive cells,
// 16 lanes per cell.  A cell's place in its level's array = number of kept keys of the level's earlier cells: the cells before
// the block are summed by the whole block (<= a few hundred counts), the ones inside it by a 16-entry scan - no separate scan
// kernel.  The cell that closes a level writes the level's total (candCnt).
__global__ __launch_bounds__(256) void k_gather(const LevelGeom *__restrict__ geom, int nlevels, int totalCells,
                                                const uint32_t *__restrict__ cellCnt, const uint32_t *__restrict__ cellRaw,
                                                const uint32_t *__restrict__ slots, size_t slotsPerImg,
                                                uint32_t *__restrict__ cand, size_t keysPerImg, int32_t *__restrict__ candCnt,
                                                int iniTh, int minTh, CellBases cb, int32_t *__restrict__ sparseFlag, int sparsePerCell,
                                                int32_t *__restrict__ sparseSeen, int callSeq) {
    __shared__ int wsum[4], ccnt[GATHER_CELLS_PER_BLOCK], clvl[GATHER_CELLS_PER_BLOCK];
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    int bx, b;
    xcd_block_map(bx, b);
    const int gc0 = bx * GATHER_CELLS_PER_BLOCK, gc = gc0 + grp;
    const uint32_t *cc = cellCnt + (size_t)b * totalCells;
    const bool live = gc < totalCells;
    const int gcl = live ? gc : totalCells - 1;
    // everything that does not depend on the cell's place in the key array is requested first - its own list included - so that
    // the dependent loads of this kernel (counts -> place -> stores) overlap instead of queueing up: it is latency-bound
    const int l = level_of_cell(cb, nlevels, gcl);
    const int cn = live ? (int)cc[gcl] : 0;
    const uint32_t rw = live ? cellRaw[(size_t)b * totalCells + gcl] : 0u;
    const LevelGeom *gp = geom + l;
    const int capc = gp->capc, cellBase = gp->cellBase, ncells = gp->ncells;
    const unsigned long long slotOff = gp->slotOff, keyOff = gp->keyOff;
    const int nraw = (int)(rw & 0x7FFFFFFFu);
    const uint32_t thr = (rw >> 31) ? (uint32_t)iniTh : (uint32_t)minTh;
    const uint32_t *src = slots + (size_t)b * slotsPerImg + slotOff + (size_t)(gcl - cellBase) * capc;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = src[min(sub + 16 * k, max(nraw - 1, 0))];
    // kept keys of the cells [base0, gc0) of the block's first level
    const int l0 = level_of_cell(cb, nlevels, gc0), base0 = cb.v[l0];
    // (four independent loads per round: a 1920x1080 level 0 has 2108 cells, and one load per round made the later blocks of such
    //  a level wait for eight dependent memory latencies)
    int part = 0;
    for (int i = base0 + tid; i < gc0; i += 1024) {
        int q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = (int)cc[min(i + 256 * k, gc0 - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++) part += i + 256 * k < gc0 ? q[k] : 0;
    }
    part = wave_total_i32(part);
    if ((tid & 63) == 0) wsum[tid >> 6] = part;
    if (sub == 0) { ccnt[grp] = cn; clvl[grp] = live ? l : -1; }
    __syncthreads();
    if (!live) return;
    int off = l == l0 ? wsum[0] + wsum[1] + wsum[2] + wsum[3] : 0;
    for (int k = 0; k < grp; k++) off += clvl[k] == l ? ccnt[k] : 0;
    if (sub == 0 && gc == cellBase + ncells - 1) {
        candCnt[b * nlevels + l] = off + cn;
        // verdict for the NEXT call's FAST stage on this image slot: a level with few candidates per cell is put through the
        // row pre-test of k_fast_strips (speed only, never results)
        if (sparseFlag) {
            const int sparse = (off + cn) < sparsePerCell * ncells ? 1 : 0;
            sparseFlag[b * nlevels + l] = sparse;
            if (sparse && b == 0 && sparseSeen) __hip_atomic_store(sparseSeen, callSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t *dst = cand + (size_t)b * keysPerImg + keyOff + off;
    const int gshift = (tid & 48);   // my group's 16 lanes inside the wave's ballot
    int kept = 0;
    for (int j0 = 0; j0 < nraw; j0 += 64) {   // four loads in flight per lane before the stores (the first four: above)
        if (j0 > 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = src[min(j0 + sub + 16 * k, nraw - 1)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool keep = j0 + sub + 16 * k < nraw && (v[k] >> 24) >= thr;
            const uint32_t m = (uint32_t)(__ballot(keep) >> gshift) & 0xFFFFu;
            if (keep) dst[kept + __popc(m & ((1u << sub) - 1u))] = v[k];
            kept += __popc(m);
        }
    }
}

// the tile strides of the usual 30-px cell grids + the run-time-stride instance, each in the dense and the compaction form
#define ORBX_FAST_INSTANCE(EST, SP)                                                                                              \
    template __global__ void k_fast_cells<EST, SP>(const uint8_t *, size_t, const LevelGeom *, int, int, uint32_t *, uint32_t *, \
                                                   uint32_t *, size_t, int, int, int, int, int, int, int, CellBases, unsigned, const int32_t *, FastHist)
ORBX_FAST_INSTANCE(0, false);
ORBX_FAST_INSTANCE(44, false);
ORBX_FAST_INSTANCE(48, false);
ORBX_FAST_INSTANCE(52, false);
ORBX_FAST_INSTANCE(0, true);
ORBX_FAST_INSTANCE(44, true);
ORBX_FAST_INSTANCE(48, true);
ORBX_FAST_INSTANCE(52, true);
