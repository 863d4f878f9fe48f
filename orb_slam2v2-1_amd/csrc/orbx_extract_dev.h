// Device-side helpers and kernel declarations shared by the extractor's translation units
// (orbx_pyramid.hip, orbx_fast.hip, orbx_octree.hip, orbx_describe.hip; launches in orbx_extract.hip).
#pragma once
#include "orbx_internal.h"

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

// ---- launch geometry the host side needs
struct PyrSpan { short o0, o1, c0, c1; };  // owned [o0,o1) and computed [c0,c1) range along one axis
#define FAST_WAVES 4
#define GATHER_CELLS_PER_BLOCK 16
#ifndef OCT_T
#define OCT_T 512   // 1024-thread workgroups are resident one per CU only; 512 packs 2x better at batch 128 and costs 6 us on a single frame
#endif
#define OCT_T_WIDE 1024   // the same kernels built a second time (orbx_octree_wide.hip) for images whose level 0 has >= 600 FAST cells: twice the threads on the LDS-bound key sweeps of ONE level (1920x1080 level 0: 174 -> 150 us)
#define DESC_WAVES 4
// One k_describe launch covers the levels [lvBegin, lvEnd).  The default is every level into the caller's arrays.  A call may be
// split (launch_chunk): the levels [a, nlevels) are described first, into scratch arrays, beside the quad-tree of the large
// levels [0, a) on another stream; the launch for [0, a) then writes the caller's arrays directly and its blocks
// bx >= copyBlock0 move the scratch records of [a, nlevels) behind them (their place depends on the counts of [0, a)).
struct DescGroup {
    int lvBegin, lvEnd;
    int writeCounts;                     // this launch writes counts[b] (the total over ALL levels)
    int copyBlock0;                      // first copy block (grid.x when there is nothing to move)
    const orbx_keypoint_t *kpsScratch;   // records of the levels [lvEnd, nlevels) to move
    const uint8_t *descScratch;
    uint32_t taps;                       // k_describe<ORBX_GAUSS_FIXED_TAPS, .>: the Gaussian's Q8 taps k3 | k2 << 8 | k1 << 16 | k0 << 24 (k0 = centre)
    long long hostDelta;                 // != 0 (latency form): every keypoint / descriptor / count store is repeated at address + hostDelta - the
                                         // frame record's twin in pinned host memory - so that the last kernel only has mvuRight / mvDepth left to move
};
#define DESC_COPY_PER_BLOCK 16
#define BLUR_R 16                // output rows per wave of k_blur_levels
#define BLUR_SRC (BLUR_R + 6)    // source rows a tile reads
struct BlurPlan { int tileBase[ORBX_MAX_LEVELS + 1]; int tilesX[ORBX_MAX_LEVELS]; };   // tiles of the levels blurred as a whole (others own none)
__global__ void k_blur_levels(const uint8_t *pyr, uint8_t *blur, size_t pyrImgBytes, const LevelGeom *geom, int nlevels, int totalTiles,
                              BlurPlan bp, int gaussRounding, uint32_t taps);                                                                                        // orbx_describe.hip
struct CellBases { int v[ORBX_MAX_LEVELS + 1]; };
__device__ __forceinline__ int level_of_cell(const CellBases &cb, int nlevels, int gc) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) l += (i < nlevels && gc >= cb.v[i]) ? 1 : 0;
    return l;
}

// ---- kernels (definitions: see the file named on the right)
__global__ void k_pyramid_fused(const uint8_t *src, int sstride, size_t simg, uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom,
                                int nlevels, const int32_t *tab, int xSpanOff, int ySpanOff, int tilesX, int tilesY, int bufBytes,
                                int maxPar, int l0);                                                             // orbx_pyramid.hip
template <int RW, int SR>
__global__ void k_pyr_level(uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, int l, const int32_t *tab, int nxc,
                            int nbands);                                                                         // orbx_pyramid.hip
template <bool FULL>
__global__ void k_pyr_pad(const uint8_t *img, int sstride, size_t simg, uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom,
                          int l0);                                                                               // orbx_pyramid.hip
// ---- level CHAINS (small batches, round 5): up to PC_MAXL consecutive levels built by ONE launch from the level in front of them.
// Per-level launches of a batch of one or two images are latency-bound (a level kernel lasts ~4.8 us however little it resizes: seven
// of them are 34 us of a stereo frame); a chain keeps the intermediate levels of a tile in LDS, so the only trips to memory are the
// source rectangle at the start and each level's own rectangle on the way out.  A workgroup owns a tile of the chain's LAST level and,
// through the resize source offsets, the matching rectangles of the levels before it (own_l partitions level l; comp_l = own_l +
// what comp_{l+1} reads: 1-2 px of halo per level, recomputed by the neighbours, written by the owner only) - the span tables of
// k_pyramid_fused, built per chain.  Arithmetic per pixel pair is k_pyr_level's.
// (struct ChainPlan, PC_MAXL: orbx_internal.h)
__global__ void k_pyr_chain(uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, const int32_t *tab, ChainPlan cp);   // orbx_pyramid.hip
__global__ void k_pyr_pad_rows(const uint8_t *img, int sstride, size_t simg, uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom);   // orbx_pyramid.hip
#define PAD_ROWS_PER_BLOCK 4
struct StripBases { int v[ORBX_MAX_LEVELS + 1]; };   // first strip of every level (levels with cells wider than 32 px own none)
__global__ void k_fast_strips(const uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, int nlevels, int totalStrips, int totalCells,
                              uint32_t *cellCnt, uint32_t *cellRaw, uint32_t *slots, size_t slotsPerImg, int iniTh, int minTh,
                              StripBases sb, const int32_t *sparseFlag, int strip0, int skipSparse);                                                                    // orbx_fast.hip
__global__ void k_fast_strips_sparse(const uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, int nlevels, int totalStrips, int totalCells,
                                     uint32_t *cellCnt, uint32_t *cellRaw, uint32_t *slots, size_t slotsPerImg, int iniTh, int minTh,
                                     StripBases sb, const int32_t *sparseFlag);                                                               // orbx_fast.hip
// Small batches (B <= 4, round 5): the FAST stage also histograms what it emits - per (image, level), count and best key of every
// cell of the quad-tree's count pyramid at its deepest depth - with atomics at the L2, spread over every CU that runs FAST cells;
// k_octree_pyr then LOADS that histogram instead of sweeping the level's keys through the LDS atomics of its one CU (a 1241x376
// level 0 at 2000 features: 15.5 us of a 39-us workgroup), and zeroes it again for the next call.  cnt == NULL: off.
struct FastHist { uint32_t *cnt, *best; int stride; const int32_t *tab; };
template <int ES_T, bool SPARSE>   // SPARSE: the compaction form for the flagged (image, level)s of the strip levels
__global__ void k_fast_cells(const uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, int nlevels, int totalCells,
                             uint32_t *cellCnt, uint32_t *cellRaw, uint32_t *slots, size_t slotsPerImg, int iniTh, int minTh, int ESrt,
                             int SSrt, int tileRows, int ldsPerWave, int phaseLimit, CellBases cb, unsigned stripLevels,
                             const int32_t *sparseFlag, FastHist fh);  // orbx_fast.hip
__global__ void k_gather(const LevelGeom *geom, int nlevels, int totalCells, const uint32_t *cellCnt, const uint32_t *cellRaw,
                         const uint32_t *slots, size_t slotsPerImg, uint32_t *cand, size_t keysPerImg, int32_t *candCnt, int iniTh,
                         int minTh, CellBases cb, int32_t *sparseFlag, int sparsePerCell, int32_t *sparseSeen, int callSeq);                                                               // orbx_fast.hip
struct OctBig {   // multi-workgroup quad-tree of large levels (orbx_octree.hip)
    uint32_t *part, *leaf, *best;
    int32_t *state;
    int K, nBig, deepMax, pyrMax;
    int levelOf[ORBX_MAX_LEVELS];
};
#define OCT_BIG_K 8   // workgroups sharing one large level
// Key source of k_octree_pyr when no k_gather ran (slots != NULL): the FAST stage's per-cell lists, read in place
struct OctSrc {
    const uint32_t *cellCnt, *cellRaw, *slots;
    size_t slotsPerImg;
    int totalCells, iniTh, minTh;
    int32_t *candCntOut;     // kept keys per (image, level): what k_gather would have written
    int32_t *sparseFlag;     // verdict for the next call's FAST stage (may be NULL)
    int sparsePerCell;
    int32_t *sparseSeen;     // host-mapped word: the sequence number of the last call that found a corner-sparse level (may be NULL)
    int callSeq;
    uint32_t *candOut;       // compacted keys, written only by a level that falls back to the exact form
    uint32_t *histCnt, *histBest; int histStride;   // != NULL: the deepest-depth histogram of every (image, level) as the FAST stage left it (FastHist)
    // large levels of a batch: the sweep of level l is shared by nslice[l] workgroups (blockIdx.z); their partial histograms (counts: two 16-bit
    // counters per word, best keys: one word per cell) and the arrival counter of every (image, level)
    unsigned char nslice[ORBX_MAX_LEVELS];
    uint32_t *partCnt, *partBest; int32_t *sliceState; int maxSlices, partStride;
    // with shared sweeps the grid is LINEAR: workgroups [blkPrefix[l], blkPrefix[l + 1]) are level l's, slice-major inside (slice * B + image), so that every
    // slice of the large levels is dispatched before the small levels (in a (B, levels, slices) grid the extra slices queued behind ALL first slices and
    // started when the small levels were done: nothing gained)
    int linear, nImages, blkPrefix[ORBX_MAX_LEVELS + 1];
};
#define OCT_MAX_SLICES 4
__global__ void k_octree_pyr(const LevelGeom *geom, int nlevels, const uint32_t *cand, size_t keysPerImg, const int32_t *candCnt,
                             uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax, int pow2cap,
                             int pyrWords, int32_t *fallback, int dbgStop, uint16_t *nodeOf, int scratchInts, int dbgStopExact,
                             unsigned bigMask, int l0, OctSrc src);                                              // orbx_octree.hip
template <int MODE>
__global__ void k_octree_big(const LevelGeom *geom, int nlevels, const uint32_t *cand, size_t keysPerImg, const int32_t *candCnt,
                             uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax, int pow2cap,
                             int pyrWords, int32_t *fallback, uint16_t *nodeOf, int scratchInts, OctBig big);    // orbx_octree.hip
__global__ void k_octree(const LevelGeom *geom, int nlevels, const uint32_t *cand, uint16_t *nodeOf, size_t keysPerImg,
                         const int32_t *candCnt, uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax,
                         int pow2cap, int scratchInts, int dbgStop);                                             // orbx_octree.hip
__global__ void k_octree_pyr_wide(const LevelGeom *geom, int nlevels, const uint32_t *cand, size_t keysPerImg, const int32_t *candCnt,
                                  uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax, int pow2cap,
                                  int pyrWords, int32_t *fallback, int dbgStop, uint16_t *nodeOf, int scratchInts, int dbgStopExact,
                                  unsigned bigMask, int l0, OctSrc src);                                         // orbx_octree_wide.hip
template <int MODE>
__global__ void k_octree_big_wide(const LevelGeom *geom, int nlevels, const uint32_t *cand, size_t keysPerImg, const int32_t *candCnt,
                                  uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax, int pow2cap,
                                  int pyrWords, int32_t *fallback, uint16_t *nodeOf, int scratchInts, OctBig big);
__global__ void k_octree_wide(const LevelGeom *geom, int nlevels, const uint32_t *cand, uint16_t *nodeOf, size_t keysPerImg,
                              const int32_t *candCnt, uint32_t *lvlKp, int lvlKpCap, int32_t *lvlCnt, const int32_t *tab, int capMax,
                              int pow2cap, int scratchInts, int dbgStop);
template <int GAUSS, bool SPLIT>   // ORBX_GAUSS_ROUND_* / ORBX_GAUSS_FIXED_TAPS: column rounding / taps of the fused Gaussian (orbx_flavour_t); SPLIT: a launch of a split call (DescGroup)
__global__ void k_describe(const uint8_t *pyr, size_t pyrImgBytes, const LevelGeom *geom, int nlevels, const uint32_t *lvlKp,
                           int lvlKpCap, const int32_t *lvlCnt, orbx_keypoint_t *kps, uint8_t *desc, int32_t *counts,
                           int cap, uint8_t *dbgBlur, const uint8_t *blur, unsigned blurMask, DescGroup grp);                                                                             // orbx_describe.hip
