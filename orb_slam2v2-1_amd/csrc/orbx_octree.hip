// orbx_octree.hip — ORBextractor::DistributeOctTree (src/ORBextractor.cc:539-763): k_octree_pyr and its exact fallback k_octree
// (part of the ORB extractor, see orbx_extract.hip for the pipeline and the C ABI)
#include "orbx_extract_dev.h"

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

__device__ __forceinline__ void octree_exact_level(const LevelGeom *geom, int nlevels, const uint32_t *cand, uint16_t *nodeOf, size_t keysPerImg,
                                                const int32_t *candCnt, uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab,
                                                int capMax, int pow2cap, int scratchInts, int dbgStop, int l, int b);
// MODE 0: the whole level in this workgroup (k_octree_pyr).  Large levels (a 1920x1080 level 0 carries ~70 k keys; its two key
// sweeps are bound by the LDS atomics of ONE CU: 47 + 86 us of a 195-us block) are shared by K workgroups instead:
// MODE 1 (k_octree_big<1>): slice s histograms its K-th of the keys in LDS and writes the partial histogram to global memory;
//         the workgroup that finishes LAST (a counter, no spinning) sums the K partials, builds the count pyramid, runs the
//         passes and leaves leaf map + list length in global memory;
// MODE 2 (k_octree_big<2>, the next launch): slice s loads the leaf map, elects the best key per node over its K-th of the keys
//         (LDS atomicMax, then one global atomicMax per node it touched); the last workgroup writes the level's keypoints.
// (struct OctBig: orbx_extract_dev.h - part [B][nBig][K][deepMax] partial histograms, leaf [B][nBig][pyrMax] leaf map,
// best [B][nBig][capMax] best key per node, state [B][nBig][4]: arrival counters A and B, list length (-1: done by the exact form))
// Phase-2 visiting order (size descending, later-created first: :689-731 sorts (size, node) pairs and walks them from the back) for
// lists of up to 64 * R <= 512 nodes, IN REGISTERS: lane holds the keys of list indices lane + 64 r, a key is
// (2^22 - 1 - size) << 10 | list index (a level whose key capacity reaches 2^22 keeps the 64-bit LDS form of the caller).  A bitonic
// network whose exchanges with a partner index >= 64 away are register moves inside the lane and the others one ds_bpermute: no LDS
// storage, no hand-off between stages.  Round 4: the LDS form of the same network (one read-compare-write round trip and a
// wave_sync per stage) took ~20 us at 256 nodes - most of a four-pass level's pass time (tools/octree_pass_time_probe.py).
template <int R>
__device__ __forceinline__ int phase2_order_regs(const uint32_t *cnt, int L, uint16_t *xlist, int lane) {
    uint32_t key[R];
    int e = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int k = lane + 64 * r;
        key[r] = ~0u;
        if (k < L) {
            const uint32_t cv = cnt[k];
            if ((cv & 0x80000000u) && (cv & 0x7FFFFFFFu) > 1) {
                key[r] = ((0x3FFFFFu - (cv & 0x3FFFFFu)) << 10) | (uint32_t)k;
                e++;
            }
        }
    }
    auto lane_stage = [&](int j, bool up, uint32_t &a) {   // exchange with lane ^ j; up: the lower index keeps the smaller key
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ j) << 2, (int)a);
        a = (((lane & j) == 0) == up) ? min(a, b) : max(a, b);
    };
    for (int kk2 = 2; kk2 <= 32; kk2 <<= 1)
        for (int j = kk2 >> 1; j > 0; j >>= 1) {
            const bool up = (lane & kk2) == 0;
#pragma unroll
            for (int r = 0; r < R; r++) lane_stage(j, up, key[r]);
        }
#pragma unroll
    for (int K = 64; K <= 64 * R; K <<= 1) {
#pragma unroll
        for (int j = K >> 1; j >= 64; j >>= 1) {
            const int jj = j >> 6;
#pragma unroll
            for (int r = 0; r < R; r++)
                if (!(r & jj)) {
                    const bool up = ((r << 6) & K) == 0;
                    const uint32_t lo = min(key[r], key[r ^ jj]), hi = max(key[r], key[r ^ jj]);
                    key[r] = up ? lo : hi;
                    key[r ^ jj] = up ? hi : lo;
                }
        }
        for (int j = 32; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) lane_stage(j, ((r << 6) & K) == 0, key[r]);
        }
    }
    const int E = wave_total_i32(e);
#pragma unroll
    for (int r = 0; r < R; r++)
        if (lane + 64 * r < E) xlist[lane + 64 * r] = (uint16_t)(key[r] & 0x3FFu);
    return E;
}
// dispatch on the list length (wave-uniform); false: the caller's LDS network takes the list (longer than 512 nodes - a register budget -, or sizes that need more than 22 bits)
__device__ __forceinline__ bool phase2_order_in_regs(const uint32_t *cnt, int L, int keyCap, uint16_t *xlist, int lane, int &E) {
    // (Round 5 also took lists of up to 1024 nodes in registers in the 1024-thread build, 16 keys per lane: the passes of a 1920x1080 level
    // did not get shorter - their lists pass 512 nodes only in the last pass - and the kernel went from 101 to 114 VGPRs, i.e. from 416 to
    // 480 of a SIMD's 512 registers for as long as a workgroup lives: beside it no wave of the descriptor kernel or of the pyramid built
    // ahead fits any more, and the pipelined step of 64 1920x1080 images got 25 us LONGER (1.159 -> 1.184 ms).  Removed.)
    constexpr int LMAX = 512;
    if (keyCap >= (1 << 22) || L > LMAX) return false;
    if (L <= 64) E = phase2_order_regs<1>(cnt, L, xlist, lane);
    else if (L <= 128) E = phase2_order_regs<2>(cnt, L, xlist, lane);
    else if (L <= 256) E = phase2_order_regs<4>(cnt, L, xlist, lane);
    else E = phase2_order_regs<8>(cnt, L, xlist, lane);
    return true;
}

template <int MODE>
__device__ __forceinline__ void octree_pyr_body(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand, size_t keysPerImg,
    const int32_t *__restrict__ candCnt, uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int pyrWords, int32_t *__restrict__ fallback, int dbgStop,
    uint16_t *__restrict__ nodeOf, int scratchInts, int dbgStopExact, int l, int b, int slice, int bigIdx, const OctBig &big, const OctSrc &src) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
    const LevelGeom g = geom[l];
    const int Dm = g.pyrDepth, nIni = g.nIni, N = g.N;
#ifdef ORBX_DEVELOPER
    const unsigned long long dvT0 = wall_clock64();
    unsigned long long dvTA = dvT0, dvTB = dvT0;
#endif
    uint8_t *sp = smem;
    unsigned long long *skey = (unsigned long long *)sp; sp += sizeof(unsigned long long) * pow2cap;
    uint32_t *cntA = (uint32_t *)sp; sp += 4 * 2 * capMax;   // [2][cap] key count, bit31 = fresh
    uint32_t *nidA = (uint32_t *)sp; sp += 4 * 2 * capMax;   // [2][cap] depth << 28 | cell
    uint32_t *hist = (uint32_t *)sp; sp += 4 * 4 * capMax;   // children counts of list node k; later best[]
    int *pn = (int *)sp; sp += 4 * capMax;
    uint32_t *pyr = (uint32_t *)sp; sp += 4 * (size_t)pyrWords;
    uint16_t *xlist = (uint16_t *)sp; sp += 2 * capMax;
    uint8_t *split = sp; sp += capMax;
    __shared__ int sh_L, sh_Lnew, sh_finish, sh_phase, sh_abort, sh_last;

    // FUSED (MODE 0 with src.slots): the keys come straight from the FAST stage's per-cell lists - no k_gather launch, no compacted
    // key array (k_gather materialises it on demand for the test hooks; a level that falls back to the exact form gathers itself)
    const bool fused = MODE == 0 && src.slots != nullptr;   // uniform
    const uint32_t *keys = cand + (size_t)b * keysPerImg + g.keyOff;
    const int n = fused ? 0 : candCnt[b * nlevels + l];
    // the level's path tables (regW + regH words, x then y) are staged into LDS: a sweep iteration then waits for ONE global
    // latency (its keys, requested an iteration ahead) instead of two dependent ones (keys, then tables)
    // (16-bit entries: a cell code at depth Dm is below nIni << 2 Dm <= 16384, ensure_plan caps Dm accordingly)
    uint16_t *pathL = (uint16_t *)(smem + (((size_t)(sp - smem) + 3) & ~(size_t)3));
    const int nPath = g.regW + g.regH;
    {
        const int32_t *tp = tab + g.xPathOff;
        for (int i = tid; i < nPath; i += OCT_T) pathL[i] = (uint16_t)tp[i];
    }
    const uint16_t *xPath = pathL, *yPath = pathL + g.regW;
    const uint32_t offDeep = (uint32_t)nIni * (((1u << (2 * Dm)) - 1u) / 3u);
    // MODE 0: best key (response << 24 | ~index: the first maximum wins, :744-760) of every cell of every depth, beside the counts:
    // the histogram sweep elects it per deep cell, the shallower depths follow by maxima of four children, and a node of the final
    // list - a cell (depth, code) - reads its keypoint straight from here.  No leaf map, no second sweep over the keys.
    constexpr bool CB = MODE == 0;
    uint32_t *bestP = (uint32_t *)(pathL + ((nPath + 1) & ~1));   // [nIni * (4^(Dm+1) - 1) / 3], depth d at nIni * (4^d - 1) / 3 like the counts
    int *cellOff = (int *)(bestP + (uint32_t)nIni * (((1u << (2 * (Dm + 1))) - 1u) / 3u));   // fused: [ncells + 1] first raw index of every FAST cell
    __shared__ int wsumF[OCT_T / 64 + 1];
    const uint32_t *fCnt = fused ? src.cellCnt + (size_t)b * src.totalCells + g.cellBase : nullptr;
    const uint32_t *fRaw = fused ? src.cellRaw + (size_t)b * src.totalCells + g.cellBase : nullptr;
    const uint32_t *fSlots = fused ? src.slots + (size_t)b * src.slotsPerImg + g.slotOff : nullptr;

    // this workgroup's share of the keys (MODE 0: all of them), in whole groups of four
    const int iLo = MODE == 0 ? 0 : (int)(((long long)n * slice / big.K) & ~3ll);
    const int iHi = MODE == 0 || slice == big.K - 1 ? n : (int)(((long long)n * (slice + 1) / big.K) & ~3ll);
    const size_t bigSlot = MODE == 0 ? 0 : (size_t)b * big.nBig + bigIdx;
    int32_t *bstate = MODE == 0 ? nullptr : big.state + 4 * bigSlot;
    uint32_t *gleaf = MODE == 0 ? nullptr : big.leaf + bigSlot * big.pyrMax, *gbest = MODE == 0 ? nullptr : big.best + bigSlot * capMax;
    int L = 0, cur = 0;
  if (MODE != 2) {
    // ---- 1. histogram of the keys at depth Dm (two 16-bit counters per word)
    for (int i = tid; i < pyrWords; i += OCT_T) pyr[i] = 0;
    if (CB) for (int i = tid; i < (nIni << (2 * Dm)); i += OCT_T) bestP[offDeep + i] = 0;
    if (tid == 0) sh_abort = 0;
    __syncthreads();
    // HIST (small batches): the FAST stage histogrammed its emissions at the L2 (FastHist, orbx_extract_dev.h): deepest-depth counts
    // and best keys (the tie-break index is the key's SLOT index cell * capc + position: the same order as vToDistributeKeys) are
    // loaded - two 16-bit counters per LDS word - and the global arrays zeroed for the next call.  No sweep, no raw-index scan.
    const bool fromHist = fused && src.histCnt != nullptr;   // uniform
    if (fromHist) {
        uint32_t *hc = src.histCnt + ((size_t)b * nlevels + l) * src.histStride, *hb = src.histBest + ((size_t)b * nlevels + l) * src.histStride;
        const int nDeep4 = (nIni << (2 * Dm)) >> 2;    // (a multiple of four cells: Dm >= 1)
        for (int i = tid; i < nDeep4; i += OCT_T) {
            const uint4 cv = ((const uint4 *)hc)[i], bv = ((const uint4 *)hb)[i];
            pyr[offDeep + 2 * i] = cv.x | (cv.y << 16);
            pyr[offDeep + 2 * i + 1] = cv.z | (cv.w << 16);
            bestP[offDeep + 4 * i] = bv.x; bestP[offDeep + 4 * i + 1] = bv.y; bestP[offDeep + 4 * i + 2] = bv.z; bestP[offDeep + 4 * i + 3] = bv.w;
            ((uint4 *)hc)[i] = make_uint4(0u, 0u, 0u, 0u);
            ((uint4 *)hb)[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else if (fused) {
        // One sweep over the FAST stage's cell lists, read in place.  The tie-break index of a key is its SLOT index cell * capc + position
        // in the cell's raw list: the same order as vToDistributeKeys (which only drops entries), and the winner's slot is the index
        // itself (rounds 3-4 used the position in the concatenated raw lists: a scan over the cells in front of the sweep and a binary
        // search behind it).  Round 5: a LARGE level's sweep is shared by K workgroups (src.nslice, slice = blockIdx.z): each sweeps its
        // K-th of the cells into its own LDS histogram, leaves it in global memory, and the one that arrives last adds the others' to
        // its own and carries on - in a batch of 1920x1080 images the level-0 workgroup swept ~70 k keys through the LDS atomics of
        // one CU for 64 of its 109 us while the CUs of the small levels had long finished.
        const int K = MODE == 0 && src.nslice[l] > 1 ? (int)src.nslice[l] : 1;
        const int cBeg = (int)((long long)g.ncells * slice / K), cEnd = (int)((long long)g.ncells * (slice + 1) / K);
#ifdef ORBX_DEVELOPER
        dvTA = wall_clock64();
#endif
        constexpr int CPR = OCT_T / 16;
        const int sub = tid & 15, cgrp = tid >> 4, capc = g.capc;
        auto fetch = [&](int cell, uint4 &e, uint32_t &rw) {
            const int cc = min(cell, g.ncells - 1);
            rw = fRaw[cc];
            e = *(const uint4 *)(fSlots + (size_t)cc * capc + 4 * sub);
        };
        // (Round 5 also kept FOUR rounds of cells in flight instead of one: the sweep of a 1920x1080 level 0 stayed at 63.8 us - it is bound by
        // the LDS work of its CU, two table reads and up to two atomics per key, not by the memory round trip.)
        uint4 en; uint32_t rwn;
        fetch(cBeg + cgrp, en, rwn);
        for (int c0 = cBeg; c0 < cEnd; c0 += CPR) {
            const int cell = c0 + cgrp;
            uint4 e = en; const uint32_t rw = rwn;
            fetch(c0 + CPR + cgrp, en, rwn);
            const int nraw = cell < cEnd ? (int)(rw & 0x7FFFFFFFu) : 0;
            const uint32_t thr = (rw >> 31) ? (uint32_t)src.iniTh : (uint32_t)src.minTh;
            const int base = cell * capc;
            for (int j0 = 0; j0 < nraw; j0 += 64) {   // (wave-divergent trip count only for cells with more than 64 raw entries)
                if (j0 > 0) e = *(const uint4 *)(fSlots + (size_t)cell * capc + j0 + 4 * sub);
                const uint32_t key[4] = {e.x, e.y, e.z, e.w};
                uint32_t c[4], inc[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + 4 * sub + u;
                    const bool ok = j < nraw && (key[u] >> 24) >= thr;
                    const uint32_t kk = j < nraw ? key[u] : 0u;
                    c[u] = (uint32_t)xPath[kk & 0xFFF] | (uint32_t)yPath[(kk >> 12) & 0xFFF];
                    inc[u] = ok ? 1u : 0u;
                    bv[u] = ((kk >> 24) << 24) | (0xFFFFFFu - (uint32_t)(base + j));
                }
#pragma unroll
                for (int u = 0; u < 3; u++)
                    if (c[u] == c[u + 1]) { inc[u + 1] += inc[u]; bv[u + 1] = inc[u] ? (inc[u + 1] > inc[u] ? max(bv[u], bv[u + 1]) : bv[u]) : bv[u + 1]; inc[u] = 0; }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (inc[u]) {
                        atomicAdd(&pyr[offDeep + (c[u] >> 1)], inc[u] << (16 * (c[u] & 1)));
                        atomicMax(&bestP[offDeep + c[u]], bv[u]);
                    }
            }
        }
        if (K > 1) {   // my partial histogram -> global; the last of the K to arrive adds the others' to its own (hand-off as in MODE 1 below)
            __syncthreads();
            const int nDeep = nIni << (2 * Dm), deepWords = (nDeep + 1) >> 1;
            const size_t slot = (size_t)b * nlevels + l;
            uint32_t *pcnt = src.partCnt + (slot * src.maxSlices + slice) * (size_t)src.partStride;
            uint32_t *pbst = src.partBest + (slot * src.maxSlices + slice) * (size_t)src.partStride;
            for (int i = tid; i < deepWords; i += OCT_T) pcnt[i] = pyr[offDeep + i];
            for (int i = tid; i < nDeep; i += OCT_T) pbst[i] = bestP[offDeep + i];
            __syncthreads();
            if (tid == 0) {
                sh_last = __hip_atomic_fetch_add(&src.sliceState[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == K - 1;
                if (sh_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // only the last arrival pays the invalidation
            }
            __syncthreads();
            if (!sh_last) return;
            for (int k = 0; k < K; k++) {
                if (k == slice) continue;
                const uint32_t *qc = src.partCnt + (slot * src.maxSlices + k) * (size_t)src.partStride;
                const uint32_t *qb = src.partBest + (slot * src.maxSlices + k) * (size_t)src.partStride;
                for (int i = tid; i < deepWords; i += OCT_T) pyr[offDeep + i] += qc[i];     // two 16-bit counters per word, no carry between them
                for (int i = tid; i < nDeep; i += OCT_T) bestP[offDeep + i] = max(bestP[offDeep + i], qb[i]);
            }
            if (tid == 0) src.sliceState[slot] = 0;   // ready for the next call
        }
    } else {
    uint32_t nkey[4];
    if (iLo + 4 * tid < iHi) load_keys4(keys, iLo + 4 * tid, n, nkey);
    for (int i0 = iLo + 4 * tid; i0 < iHi; i0 += 4 * OCT_T) {
        uint32_t key[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = nkey[u];
        load_keys4(keys, i0 + 4 * OCT_T < iHi ? i0 + 4 * OCT_T : i0, n, nkey);   // the next iteration's keys (unconditional: no load behind a branch)
#pragma unroll
        for (int u = 0; u < 4; u++)   // a missing key is 0
            c[u] = (uint32_t)xPath[key[u] & 0xFFF] | (uint32_t)yPath[(key[u] >> 12) & 0xFFF];
        // consecutive keys (row-major inside a FAST cell) mostly share the deep cell: count runs, one LDS atomic per run
        uint32_t inc[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            inc[u] = i0 + u < iHi ? 1u : 0u;
            bv[u] = ((key[u] >> 24) << 24) | (0xFFFFFFu - (uint32_t)(i0 + u));
        }
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (c[u] == c[u + 1]) { inc[u + 1] += inc[u]; bv[u + 1] = inc[u] ? max(bv[u], bv[u + 1]) : bv[u + 1]; inc[u] = 0; }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (inc[u]) {
                atomicAdd(&pyr[offDeep + (c[u] >> 1)], inc[u] << (16 * (c[u] & 1)));
                if (CB) atomicMax(&bestP[offDeep + c[u]], bv[u]);
            }
    }
    }   // !fused
    __syncthreads();
#ifdef ORBX_DEVELOPER
    dvTB = wall_clock64();
#endif
    if (MODE == 1) {   // partial histogram -> global; the last workgroup to arrive carries on with the sum of all K
        const int deepWords = ((nIni << (2 * Dm)) + 1) >> 1;
        uint32_t *gp = big.part + (bigSlot * big.K + slice) * big.deepMax;
        for (int i = tid; i < deepWords; i += OCT_T) gp[i] = pyr[offDeep + i];
        // hand-off across workgroups (they may sit on different XCDs, each with its own L2): every wave's stores are done at the
        // barrier; then ONE lane releases at agent scope (L2 write-back), counts the workgroup, and - if it is the last to arrive -
        // an acquire fence invalidates this CU's L1 and the stale L2 lines, so that the whole workgroup may read the partials with
        // plain loads.  (A fence in every thread makes every one of them flush / invalidate the caches: 5-15x slower at batch 32.)
        __syncthreads();
        if (tid == 0) {
            sh_last = __hip_atomic_fetch_add(&bstate[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == big.K - 1;
            if (sh_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // only the last arrival pays the invalidation
        }
        __syncthreads();
        if (!sh_last) return;
        const uint32_t *ga = big.part + bigSlot * big.K * big.deepMax;
        for (int i = tid; i < deepWords; i += OCT_T) {
            uint32_t sum = 0;
            for (int k = 0; k < big.K; k++) sum += ga[(size_t)k * big.deepMax + i];   // two 16-bit counters per word, no carry between them
            pyr[offDeep + i] = sum;
        }
        if (tid == 0) bstate[0] = 0;   // ready for the next call
        __syncthreads();
    }
    if (dbgStop == 1) return;
    // ---- 2. counts of the shallower depths
    for (int d = Dm - 1; d >= 0; d--) {
        const uint32_t off = (uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u);
        const int ne = nIni << (2 * d);
        for (int e = tid; e < ne; e += OCT_T) {
            uint32_t s = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) s += pyr_count(pyr, nIni, Dm, d + 1, 4u * e + q);
            pyr[off + e] = s;
            if (CB) {
                const uint32_t *ch = bestP + (uint32_t)nIni * (((1u << (2 * (d + 1))) - 1u) / 3u) + 4u * e;
                bestP[off + e] = max(max(ch[0], ch[1]), max(ch[2], ch[3]));
            }
        }
        __syncthreads();
    }
    if (dbgStop == 2) return;
    // ---- 3. root nodes (:543-592)
    if (tid == 0) {
        int L0 = 0, kept = 0;
        for (int r = 0; r < nIni; r++) {
            const uint32_t c = pyr[r];
            kept += (int)c;
            if (c > 0) { cntA[L0] = c | 0x80000000u; nidA[L0] = (uint32_t)r; L0++; }
        }
        sh_L = L0;
        if (fused) {   // kept keys of the level (= the sum of the root counts), verdict for the next call's FAST
            src.candCntOut[b * nlevels + l] = kept;
            if (src.sparseFlag) {
                const int sparse = kept < src.sparsePerCell * g.ncells ? 1 : 0;
                src.sparseFlag[b * nlevels + l] = sparse;
                if (sparse && b == 0 && src.sparseSeen) __hip_atomic_store(src.sparseSeen, src.callSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    L = sh_L;
    int phase = 1;
#ifdef ORBX_DEVELOPER
    const unsigned long long dvT1 = wall_clock64(); int dvPasses = 0;
#endif

    // ---- 4. passes: list bookkeeping on node counts only, by wave 0.  (Round 3 tried every step of a pass on the whole workgroup -
    // scans by the block, the phase-2 order by counting instead of the one-wave bitonic sort: identical results, and in a batch the
    // kernel got SLOWER, 85 -> 119 us at 1241x376 x 128 images, 159 -> 206 us at 1920x1080 x 64: with four workgroups per CU the
    // other waves' barriers and loops take issue slots from the workgroups that are in their key sweep.  One wave it stays.)
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
                } else if (phase2_order_in_regs(cnt, L, g.keyCap, xlist, lane, E)) {   // (size desc, later-created first)
                } else {            // ... longer lists: the same bitonic network of (~size, list index) through LDS
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
        if (sh_abort) {   // counts deeper than the pyramid are needed: this block redoes its level with the exact form (same LDS, carved anew)
            if (tid == 0) { fallback[b * nlevels + l] = 1; if (MODE == 1) bstate[2] = -1; }   // (a record: tests look at it; MODE 2 skips the level)
            __syncthreads();
            if (fused) {   // the exact form sweeps the compacted keys: this block gathers its level (what k_gather does for a whole batch)
                uint32_t *dstBase = src.candOut + (size_t)b * keysPerImg + g.keyOff;
                for (int i = tid; i < g.ncells; i += OCT_T) cellOff[i] = (int)fCnt[i];
                __syncthreads();
                array_scan_excl(cellOff, g.ncells, wsumF);
                const int sub = tid & 15, gshift = tid & 48;
                for (int c0 = 0; c0 < g.ncells; c0 += OCT_T / 16) {
                    const int cell = c0 + (tid >> 4);
                    const bool live = cell < g.ncells;
                    const uint32_t rw = live ? fRaw[cell] : 0u;
                    const int nraw = (int)(rw & 0x7FFFFFFFu);
                    const uint32_t thr = (rw >> 31) ? (uint32_t)src.iniTh : (uint32_t)src.minTh;
                    const uint32_t *sp2 = fSlots + (size_t)min(cell, g.ncells - 1) * g.capc;
                    uint32_t *dst = dstBase + (live ? cellOff[cell] : 0);
                    int keptc = 0;
                    const int rounds = (g.capc + 63) / 64;   // uniform bound (the ballots need every lane; entries past a list are masked)
                    for (int r = 0; r < rounds; r++) {
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const int j = 64 * r + sub + 16 * k4;
                            const uint32_t v4 = j < nraw ? sp2[j] : 0u;
                            const bool keep = j < nraw && (v4 >> 24) >= thr;
                            const uint32_t m = (uint32_t)(__ballot(keep) >> gshift) & 0xFFFFu;
                            if (keep) dst[keptc + __popc(m & ((1u << sub) - 1u))] = v4;
                            keptc += __popc(m);
                        }
                    }
                }
                __syncthreads();
            }
            octree_exact_level(geom, nlevels, cand, nodeOf, keysPerImg, candCnt, lvlKp, lvlKpCap, lvlCnt, tab, capMax, pow2cap, scratchInts,
                               dbgStopExact, l, b);
            return;
        }
        L = sh_Lnew;
        phase = sh_phase;
        cur ^= 1;
#ifdef ORBX_DEVELOPER
        dvPasses++;
#endif
        if (sh_finish) break;
        __syncthreads();
    }

#ifdef ORBX_DEVELOPER
    const unsigned long long dvT2 = wall_clock64();
#endif
    if (dbgStop == 3) return;
    if (CB) {   // ---- 5'. a node of the list is a cell of the pyramid: its best key is already there
        const uint32_t *nid = nidA + cur * capMax;
        uint32_t *okp = lvlKp + (size_t)b * lvlKpCap + g.lvlKpOff;
        const int Lout = min(L, g.nodeCap);
        for (int k = tid; k < Lout; k += OCT_T) {
            const int d = (int)(nid[k] >> 28);
            const uint32_t v = bestP[(uint32_t)nIni * (((1u << (2 * d)) - 1u) / 3u) + (nid[k] & 0x0FFFFFFFu)];
            const int idx = (int)(0xFFFFFFu - (v & 0xFFFFFFu));
            if (fused) okp[k] = fSlots[idx];   // the index IS the slot (cell * capc + position)
            else
                okp[k] = keys[idx];
        }
#ifdef ORBX_DEVELOPER
        // developer build, option 7 = 9: the record holds (passes << 24 | pass loop in 0.1 us << 12 | time since kernel entry in 0.1 us) instead
        // of the 0 / 1 flag; 7 = 8: the finer split - four 8-bit fields in 0.25-us units: entry -> sweep start -> sweep end -> first pass ->
        // last pass.  With the option at 0 the record is the product build's (orbx_debug_octree_fallbacks reads 0 / 1).
        if (dbgStop == 8 || dbgStop == 9) {
            if (tid == 0) { lvlCnt[b * nlevels + l] = Lout; const unsigned long long t = wall_clock64();
                if (dbgStop == 8)
                    fallback[b * nlevels + l] = ((int)min((dvTA - dvT0) / 25ull, 255ull) << 24) | ((int)min((dvTB - dvTA) / 25ull, 255ull) << 16) |
                                                ((int)min((dvT1 - dvTB) / 25ull, 255ull) << 8) | (int)min((dvT2 - dvT1) / 25ull, 255ull);
                else
                    fallback[b * nlevels + l] = (dvPasses << 24) | ((int)min((dvT2 - dvT1) / 10ull, 4095ull) << 12) | (int)min((t - dvT0) / 10ull, 4095ull); }
            return;
        }
#endif
        if (tid == 0) { lvlCnt[b * nlevels + l] = Lout; fallback[b * nlevels + l] = 0; }
        return;
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
    if (MODE == 1) {   // hand the leaf map to the K workgroups of the next launch
        for (int i = tid; i < pyrWords; i += OCT_T) gleaf[i] = pyr[i];
        for (int i = tid; i < L; i += OCT_T) gbest[i] = 0;
        if (tid == 0) bstate[2] = L;
        return;
    }
  } else {   // MODE 2: the leaf map of the previous launch
    L = bstate[2];
    if (L < 0) return;   // the level was finished by the exact form
    for (int i = tid; i < pyrWords; i += OCT_T) pyr[i] = gleaf[i];
    for (int i = tid; i < L; i += OCT_T) hist[i] = 0;
    __syncthreads();
  }
    // ---- 6. every key walks down to its leaf; best key of the node, first maximum wins (:744-760)
    uint32_t nkey2[4];
    if (iLo + 4 * tid < iHi) load_keys4(keys, iLo + 4 * tid, n, nkey2);
    for (int i0 = iLo + 4 * tid; i0 < iHi; i0 += 4 * OCT_T) {
        uint32_t key[4], cd[4], node[4];
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = nkey2[u];
        load_keys4(keys, i0 + 4 * OCT_T < iHi ? i0 + 4 * OCT_T : i0, n, nkey2);
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
        // a thread's four keys are consecutive in cell-major order and mostly fall into the same node: merge equal
        // neighbours first (the maximum of two packed values is what two atomicMax would leave) - LDS atomics on one
        // address are serialised, and a level-0 node receives ~70 keys
        uint32_t val[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            val[u] = ((key[u] >> 24) << 24) | (0xFFFFFFu - (uint32_t)(i0 + u));
            if (i0 + u >= iHi) node[u] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (node[u] == node[u + 1]) { val[u + 1] = max(val[u], val[u + 1]); node[u] = 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (node[u] != 0xFFFFFFFFu) atomicMax(&hist[node[u]], val[u]);
    }
    __syncthreads();
    if (dbgStop == 4) return;
    if (MODE == 2) {   // merge my nodes into the level's, count myself; the last workgroup to arrive writes the output
        for (int k = tid; k < L; k += OCT_T) { const uint32_t v = hist[k]; if (v) atomicMax(&gbest[k], v); }
        __syncthreads();
        if (tid == 0) sh_last = __hip_atomic_fetch_add(&bstate[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == big.K - 1;   // one lane: the acquire orders the loads below behind the other workgroups' atomicMax
        __syncthreads();
        if (!sh_last) return;
        // the atomics above executed at the memory side; read their result the same way (no cached copy can be stale)
        for (int k = tid; k < L; k += OCT_T) hist[k] = __hip_atomic_load(&gbest[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) bstate[1] = 0;
        __syncthreads();
    }
    // ---- 7. output in list order
    uint32_t *okp = lvlKp + (size_t)b * lvlKpCap + g.lvlKpOff;
    const int Lout = min(L, g.nodeCap);
    for (int k = tid; k < Lout; k += OCT_T) okp[k] = keys[0xFFFFFFu - (hist[k] & 0xFFFFFFu)];
    if (tid == 0) { lvlCnt[b * nlevels + l] = Lout; fallback[b * nlevels + l] = 0; }
}

// Registers: three 512-thread workgroups per CU is what the LDS allows (~42 KB each at 1241x376 / 1000 features), i.e. 6 waves per SIMD = 80 VGPRs.
// Left to itself the compiler takes 84 (round 3; with the register sort of the phase-2 order, 115) and only TWO workgroups fit: the 1024
// workgroups of a 128-image batch then run in two rounds, 58 us instead of the ~43 us the longest workgroup lives.
#ifndef OCT_NARROW_WAVES
#define OCT_NARROW_WAVES 6
#endif
// The 1024-thread build (one workgroup per CU: 4 waves per SIMD) would take 102 VGPRs = 416 of a SIMD's 512 for as long as a workgroup lives -
// ~110 us for level 0 of a 1920x1080 image - and beside it at most one wave of the descriptor kernel or of the pyramid built ahead fits.  Held to
// 80 (36 B of scratch per lane, in wave 0's pass code) the pipelined step of 64 such images takes 1.145 instead of 1.172 ms; at 96: 1.162,
// at 64 (96 B of scratch): 1.161; with round 5's short-lived 114-VGPR form (16-key register sort): 1.184.  A kernel that runs BESIDE others is
// sized by what it leaves them, not by what it could use.
#ifndef OCT_WIDE_WAVES
#define OCT_WIDE_WAVES 6
#endif
#if OCT_T == 512
#define OCT_PYR_WAVES __attribute__((amdgpu_waves_per_eu(OCT_NARROW_WAVES, OCT_NARROW_WAVES)))
#else
#define OCT_PYR_WAVES __attribute__((amdgpu_waves_per_eu(OCT_WIDE_WAVES, OCT_WIDE_WAVES)))
#endif
__global__ __launch_bounds__(OCT_T) OCT_PYR_WAVES void k_octree_pyr(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand, size_t keysPerImg,
    const int32_t *__restrict__ candCnt, uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int pyrWords, int32_t *__restrict__ fallback, int dbgStop,
    uint16_t *__restrict__ nodeOf, int scratchInts, int dbgStopExact, unsigned bigMask, int l0, OctSrc src) {
    int l = l0 + (int)blockIdx.y, b = blockIdx.x, slice = 0;  // level-major: large levels start first (l0: first level of a group launch)
    if (src.linear) {   // shared sweeps (OctSrc): linear grid, level-major, then slice, then image
        const int lin = blockIdx.x;
        l = 0;
#pragma unroll
        for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && lin >= src.blkPrefix[i]) ? 1 : 0;
        const int r = lin - src.blkPrefix[l];
        slice = r / src.nImages;
        b = r - slice * src.nImages;
    }
    if ((bigMask >> l) & 1u) return;             // shared by several workgroups: k_octree_big
    // A level is ONE workgroup walking a serial chain: when other kernels share its CU (the pyramid built ahead, the stereo
    // matcher of the previous batch), its waves take the issue slots first - the chain is the critical path, the others are not.
    __builtin_amdgcn_s_setprio(3);
    OctBig none = {};
    octree_pyr_body<0>(geom, nlevels, cand, keysPerImg, candCnt, lvlKp, lvlKpCap, lvlCnt, tab, capMax, pow2cap, pyrWords, fallback, dbgStop,
                       nodeOf, scratchInts, dbgStopExact, l, b, slice, 0, none, src);
}
template <int MODE>
__global__ __launch_bounds__(OCT_T) void k_octree_big(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand, size_t keysPerImg,
    const int32_t *__restrict__ candCnt, uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int pyrWords, int32_t *__restrict__ fallback,
    uint16_t *__restrict__ nodeOf, int scratchInts, OctBig big) {
    const int slice = blockIdx.x, b = blockIdx.z;
    if (MODE == 1 && (int)blockIdx.y >= big.nBig) {
        // the first launch also carries the levels that one workgroup handles alone (grid y = nBig .. nlevels-1, slice 0 only):
        // they run beside the large levels' histograms and passes instead of in a launch of their own
        if (slice != 0) return;
        OctSrc nosrc = {};
        octree_pyr_body<0>(geom, nlevels, cand, keysPerImg, candCnt, lvlKp, lvlKpCap, lvlCnt, tab, capMax, pow2cap, pyrWords, fallback, 0,
                           nodeOf, scratchInts, 0, big.levelOf[blockIdx.y], b, 0, 0, big, nosrc);
        return;
    }
    const int bigIdx = blockIdx.y;
    OctSrc nosrc2 = {};
    octree_pyr_body<MODE>(geom, nlevels, cand, keysPerImg, candCnt, lvlKp, lvlKpCap, lvlCnt, tab, capMax, pow2cap, pyrWords, fallback, 0,
                          nodeOf, scratchInts, 0, big.levelOf[bigIdx], b, slice, bigIdx, big, nosrc2);
}
template __global__ void k_octree_big<1>(const LevelGeom *, int, const uint32_t *, size_t, const int32_t *, uint32_t *, int, int32_t *,
                                         const int32_t *, int, int, int, int32_t *, uint16_t *, int, OctBig);
template __global__ void k_octree_big<2>(const LevelGeom *, int, const uint32_t *, size_t, const int32_t *, uint32_t *, int, int32_t *,
                                         const int32_t *, int, int, int, int32_t *, uint16_t *, int, OctBig);

// K3 (exact form): one sweep over the keys per pass.  Called by k_octree_pyr for the levels whose tree outgrows the count
// pyramid (a block that finds out simply carries on here: no second launch), and launched on its own with developer knob 4.
__device__ __forceinline__ void octree_exact_level(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand,
    uint16_t *__restrict__ nodeOf, size_t keysPerImg, const int32_t *__restrict__ candCnt,
    uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int scratchInts, int dbgStop, int l, int b) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x;
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

    const uint32_t *keys = cand + (size_t)b * keysPerImg + g.keyOff;
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
                        atomicMax(&S.hist[k], ((key[u] >> 24) << 24) | (0xFFFFFFu - (uint32_t)i));
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
    for (int k = tid; k < Lout; k += OCT_T) okp[k] = keys[0xFFFFFFu - (best[k] & 0xFFFFFFu)];
    if (tid == 0) lvlCnt[b * nlevels + l] = Lout;
}

__global__ __launch_bounds__(OCT_T) void k_octree(
    const LevelGeom *__restrict__ geom, int nlevels, const uint32_t *__restrict__ cand,
    uint16_t *__restrict__ nodeOf, size_t keysPerImg, const int32_t *__restrict__ candCnt,
    uint32_t *__restrict__ lvlKp, int lvlKpCap, int32_t *__restrict__ lvlCnt,
    const int32_t *__restrict__ tab, int capMax, int pow2cap, int scratchInts, int dbgStop) {
    // level-major block order: the large levels start first and the small ones fill the gaps
    octree_exact_level(geom, nlevels, cand, nodeOf, keysPerImg, candCnt, lvlKp, lvlKpCap, lvlCnt, tab, capMax, pow2cap, scratchInts,
                       dbgStop, blockIdx.y, blockIdx.x);
}
