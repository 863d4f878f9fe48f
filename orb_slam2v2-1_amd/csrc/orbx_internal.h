// orbx_internal.h — shared declarations of the HIP implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "orbx.h"

#define ORBX_MAX_LEVELS 16
// the Gaussian's Q8 taps k3 | k2 << 8 | k1 << 16 | k0 << 24 of every flavour but ORBX_GAUSS_FIXED_TAPS: cvRound(256 g_i) = 18 34 49 55 (sum 257)
#define ORBX_GAUSS_TAPS_DEFAULT (18u | (34u << 8) | (49u << 16) | (55u << 24))
#define ORBX_EDGE 19        // EDGE_THRESHOLD            (reference: src/ORBextractor.cc:74)
#define ORBX_MINB 16        // minBorderX = EDGE-3       (reference: src/ORBextractor.cc:773)
#define ORBX_HALF_PATCH 15  // HALF_PATCH_SIZE           (reference: src/ORBextractor.cc:73)
#define ORBX_DESC_R 18      // max |rotated tap| : pattern radius^2 = 338 -> cvRound <= 18
#define ORBX_MAX_ROOTS 64
#define ORBX_EV_RING 32
#define ORBX_MAX_CHUNKS 4      // chunks a batch may be cut into (launch_pipeline)
#define ORBX_SIDE_STREAMS 2    // handle-owned streams for the chunks behind the first
#define ORBX_HIST_IMAGES 4     // batches up to this size: the FAST stage histograms its emissions for the quad-tree (FastHist)

void orbx_set_error(const char *fmt, ...);

#define ORBX_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            orbx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                       \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ||                \
                    e_ == hipErrorNoBinaryForGpu || e_ == hipErrorInsufficientDriver)       \
                       ? ORBX_ERR_NO_DEVICE                                                 \
                       : ORBX_ERR_HIP;                                                      \
        }                                                                                   \
    } while (0)

// Per-level geometry, built on the host (exactly the reference's arithmetic) and read by
// every kernel from device memory.
struct LevelGeom {
    int w, h;            // inner level size: cvRound(cols*inv), cvRound(rows*inv)   (:1111-1112)
    int pstride, prows;  // padded buffer: row stride in bytes (multiple of 64), rows = h+38
    unsigned long long poff;  // byte offset of the padded level inside one image's pyramid block
    int regW, regH;      // maxBorderX-minBorderX, maxBorderY-minBorderY                 (:773-781)
    int nCols, nRows, wCell, hCell;  //                                                  (:784-787)
    int cellBase, ncells;            // global cell numbering across levels
    int capc;                        // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2)
    unsigned long long slotOff;      // uint32 offset of this level's slots inside one image's slot block
    int N;                           // mnFeaturesPerLevel[level]
    int nIni;                        // round(regW/regH)                                  (:543)
    int nodeCap;                     // quad-tree list capacity N + 3 + 4*nIni
    int pyrDepth;                    // depth Dm of the quad-tree count pyramid (k_octree_pyr)
    unsigned long long keyOff;       // element offset of this level's keys inside one image's key block
    int keyCap;                      // ncells*capc
    int lvlKpOff;                    // offset of this level's kept keypoints in the per-image list
    int xofsOff, xalphaOff, yofsOff, ybetaOff;  // resize tables (int32 units into d_tab), levels >= 1
    int rootTabOff, rootBoxOff;      // byte offset (uint8 rootOf[x]) / int32 offset (root x bounds)
    int xPathOff, yPathOff;          // int32 offsets: quad-tree path of a key at depth pyrDepth, per x (root<<2D | even bits) / per y (odd bits)
    float scale;                     // mvScaleFactor[level]
    float size;                      // (float)(int)(31*scale)                            (:837,846)
};

// One launch of k_pyr_chain (orbx_extract_dev.h): up to PC_MAXL consecutive pyramid levels built from the level in front of them
#define PC_MAXL 7
struct ChainPlan {
    int la, lb;                  // source level, last level built (lb - la <= PC_MAXL)
    int tilesX, tilesY;
    int xSpanOff, ySpanOff;      // int32 offsets into tab: PyrSpan [lb - la + 1][tiles] per axis (level la first)
    int bufBytes, maxRows;       // one LDS level buffer, rows of the ypar table per level
};

struct orbx_extractor {
    int nfeatures, nlevels, ini_th, min_th, device;
    double scale_factor;
    float sf[ORBX_MAX_LEVELS], isf[ORBX_MAX_LEVELS], sig2[ORBX_MAX_LEVELS], isig2[ORBX_MAX_LEVELS];
    int32_t nfeat[ORBX_MAX_LEVELS];
    int32_t umax[16];
    int max_kp;  // nfeatures + 3*nlevels (+ roots) upper bound per image
    orbx_flavour_t flavour;          // which OpenCV build the handle stands in for (orbx_create_flavoured); fixed for the handle's life
    int opt[ORBX_NUM_OPTIONS];       // per-handle options (orbx_set_option): alternative kernels / arrangements with identical results

    // plan (depends on image size / batch capacity)
    int pw, ph, pB;  // planned image size and batch capacity (0 = none)
    LevelGeom geom[ORBX_MAX_LEVELS];
    int totalCells, maxNodeCap, lvlKpCap;
    size_t pyrImgBytes, slotsPerImg, keysPerImg;
    int fastTileStride, fastScoreStride, fastTileRows, fastLdsPerWave;
    int stripBase[ORBX_MAX_LEVELS + 1], totalStrips;   // k_fast_strips: first strip of every level, strips per image
    unsigned stripLevels;                              // bit l: level l is k_fast_strips' (cells <= 32 px wide), else k_fast_cells'
    size_t octLdsBytes, octPyrLdsBytes;
    int octPyrWords;
    int32_t *d_octFallback;
    // multi-workgroup quad-tree of large levels: which levels, scratch (sized for every level so that a developer knob can force it)
    unsigned octBigMask; int octDeepMax;
    uint32_t *d_octPart, *d_octLeaf, *d_octBest; int32_t *d_octState;
    int pyrTilesX, pyrTilesY, pyrXSpanOff, pyrYSpanOff, pyrBufBytes, pyrMaxDim, pyrMaxPar;
    size_t pyrLdsBytes;
    ChainPlan chains[2][8]; int nChains[2];   // level chains of small batches (k_pyr_chain): [0] up to 4 levels per chain, [1] up to PC_MAXL
    // device buffers
    LevelGeom *d_geom;
    int32_t *d_tab;
    uint8_t *d_pyr;
    uint32_t *d_cellCnt, *d_cellRaw, *d_slots, *d_cand, *d_lvlKp;
    uint16_t *d_nodeOf;
    int32_t *d_candCnt, *d_lvlCnt;
    int candStale;       // > 0: the compacted key arrays (d_cand / d_candCnt of every level) of that many images were not written by the last call
                         // (k_octree_pyr read the cell lists in place); k_gather materialises them on demand (test hooks)
    int32_t *h_sparseSeen, *d_sparseSeen; int callSeq;   // host-mapped word: sequence number of the last call that flagged a corner-sparse level
    uint32_t *d_octPartBest, *d_octPartCnt; int32_t *d_octSliceState; int octSliceStride;   // shared sweeps of large levels in a batch (OctSrc::nslice)
    uint32_t *d_histCnt, *d_histBest; int histStride;   // [ORBX_HIST_IMAGES][nlevels][histStride] deepest-depth histogram of small batches (FastHist)
    int32_t *d_sparse;   // [B][nlevels] verdict of the last call: level with few FAST candidates (k_gather writes, k_fast_strips of the next call reads)
    // staging for the host API
    uint8_t *d_in; size_t d_in_bytes;
    orbx_keypoint_t *d_kps; uint8_t *d_desc; int32_t *d_counts; int out_cap, out_B;
    orbx_keypoint_t *h_kps; uint8_t *h_desc; int32_t *h_counts;   // pinned mirrors of the three above
    float *d_sfr; float *h_sfr; int sfr_cap;   // orbx_stereo_frame: mvuRight | mvDepth | nmatch of one frame (device + pinned mirror)
    // orbx_stereo_frame_view: two alternating frame records (HBM + pinned host twin, and the twin as kernels address it), pinned staging for pageable images
    uint8_t *fv_d[2], *fv_h[2], *fv_hdev[2]; int fv_cap, fv_next;
    uint8_t *fv_stage, *fv_stage_dev; size_t fv_stage_bytes;
    int32_t *fv_flag, *fv_flag_dev; int fv_seq;   // completion word of a latency call (coherent pinned memory): its last kernel stores the call's number, the host polls it
    long long descHostDelta;   // != 0 while a latency call is being issued: k_describe repeats its stores at address + delta (the record's pinned twin)
    uint8_t *d_dbgBlur; int dbgBlurCap;   // test hook: blurred 37x37 blocks of a single-image call (orbx_debug_blur_patches)
    hipStream_t stream;      // own stream
    hipStream_t side[ORBX_SIDE_STREAMS]; hipEvent_t evPyr[ORBX_MAX_CHUNKS], evJoin[ORBX_SIDE_STREAMS]; int lastChunks;   // chunk overlap (launch_pipeline)
    // pyramid of the NEXT batch, built ahead on side[0] into a second buffer (orbx_extract_batch_device_prefetch)
    uint8_t *d_pyrAlt; size_t pyrAltBytes; int pfValid, pfUsed, pfB, pfW, pfH, pfStride; const uint8_t *pfImgs; size_t pfImgStride;
    // levels blurred as a whole (k_blur_levels) for k_describe's gather form: buffers with the pyramid's geometry, swapped with it
    uint8_t *d_blur, *d_blurAlt; size_t blurBytes, blurAltBytes; unsigned blurMaskLast, blurMaskAlt;
    // split call (launch_chunk): records of the small levels described ahead, and the events between the two streams
    orbx_keypoint_t *d_kpsB; uint8_t *d_descB; size_t splitBytes; hipEvent_t evGather, evOctA;
    uint8_t *d_pyrNext; size_t pyrNextBytes; int pyrBuffers;   // three-buffer mode (orbx_set_pyramid_buffers): the pyramid built ahead gets a buffer of its own, the previous one survives it
    int prevPyrValid;   // d_pyrAlt still holds the pyramid of the call before the last one (orbm_stereo_batch_device_prev)
    hipEvent_t evFastDone, evPrefetch;
    hipStream_t last_stream; // stream of the last batch call (NULL is a stream too: the HIP default stream)
    int last_valid;          // ... once there has been one
    int lastB;
    int framesStale;         // > 0: the frames of levels >= 1 of that many images have not been written (see ensure_frames)
    // scratch of Frame::ComputeStereoMatches when this handle is the LEFT extractor (orbx_match.hip: stereo_scratch_reserve)
    int32_t *st_sad; uint2 *st_rc; int32_t *st_binStart; uint4 *st_items; size_t st_n; int st_nB;
    hipStream_t st_stream;   // stream of the last stereo call on this handle
    // per-stage HIP-event timing: a ring of event sets so that timing never forces a sync
    int profiling; unsigned prof_calls;
    hipEvent_t ev[ORBX_EV_RING][ORBX_NUM_STAGES];
    unsigned char ev_pending[ORBX_EV_RING];
    int ev_head;
    double acc_ms[ORBX_NUM_STAGES];
    long acc_n;
};

// extractor internals used by the matcher side
void orbx_internal_free_stereo_scratch(orbx_extractor *h);   // orbx_match.hip
// ComputeStereoMatches of the frame in image slots 0 / 1 of h with the record layout of orbx_stereo_frame_view (orbx_match.hip)
int orbx_internal_stereo_frame_record(orbx_extractor *h, uint8_t *d_rec, uint8_t *rec_hostdev, int cap, float mbf, float mb, hipStream_t st, bool recordsOnHost,
                                      int32_t *doneFlag, int doneSeq, int *flagArmed);
void orbx_internal_release_match_scratch();                  // orbx_match.hip      (thread-local staging pair)
void orbx_internal_release_arena();                          // orbx_match_fast.hip (thread-local arena)
void orbx_internal_release_bow_scratch();                    // orbx_bow.hip        (thread-local scratch)
int orbx_internal_level(const orbx_extractor *h, int level, int *w, int *hgt, int *pstride,
                        unsigned long long *poff);

// orbx_match_fast.hip: parallel candidate search + speculative resolution.  Return ORBX_OK
// (results written), 1 (= fall back to the exact one-workgroup kernel) or a negative error.
int fast_search_for_initialization(const orbx_keypoint_t *k1, const uint8_t *d1, int n1, const orbx_keypoint_t *k2,
                                   const uint8_t *d2, int n2, const orbm_grid_geom_t *g2, float *prev, int32_t *m12,
                                   int window, float nnratio, int check_ori, int device, int *nmatches);
// frame arrays that are already device-resident (the *_device matcher entry points): the kernels then run on `stream`
struct DevFrame { hipStream_t stream; };
struct FrustumArgs { const orbm_worldpoint_t *pts; const float *Tcw16; const orbm_camera_t *cam; float viewCosLimit; const float *thr; };
int fast_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                 const orbm_grid_geom_t *g, const float *sf, int nlevels, const orbm_mappoint_t *mps,
                                 const uint8_t *mp_desc, int m, int32_t *frame_mp, const int32_t *ext_obs, float th,
                                 float nnratio, int device, int *nmatches, const FrustumArgs *world = nullptr,
                                 orbm_mappoint_t *proj_out = nullptr, const DevFrame *dev = nullptr);
int fast_is_in_frustum(const orbm_worldpoint_t *pts, int m, const float *Tcw16, const orbm_camera_t *cam,
                       const orbm_grid_geom_t *g, float viewCosLimit, const float *thr, int nlevels, orbm_mappoint_t *out,
                       int device);
int fast_search_by_projection_frame(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                    const orbm_grid_geom_t *g, const float *sf, int nlevels, const orbm_camera_t *cam,
                                    const float *Tc16, const float *Tl16, const orbm_lastpoint_t *last,
                                    const uint8_t *last_desc, int nlast, int32_t *cur_mp, const int32_t *ext_obs,
                                    float th, int mono, int check_ori, int device, int *nmatches, const DevFrame *dev = nullptr);
int fast_match_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                       const orbm_grid_geom_t *g, const orbm_grid_geom_t *ga, const orbm_window_query_t *q,
                       const uint8_t *qdesc, int m,
                       int32_t *holder, const int32_t *ext_blocks, int max_dist, int check_ori, int device, int *nmatches);
int fast_best_in_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                         const orbm_grid_geom_t *g, const orbm_grid_geom_t *ga, const orbm_window_query_t *q,
                         const uint8_t *qdesc, int m, const float *inv_sigma2, int nlevels, int32_t *best_idx,
                         int32_t *best_dist, int device);
int fast_distinctive_descriptors(const uint8_t *desc, const int32_t *offsets, int npoints, int32_t *best_row,
                                 int32_t *best_median, int device);
extern thread_local int t_matchExact;   // orbm_set_thread_option(ORBM_OPT_EXACT_KERNELS): this thread's guided searches take the exact one-workgroup kernels

// XCD-aware block -> (image, block-in-image) map for grids of (blocks per image, images).  Workgroups are dealt round-robin
// to the 8 XCDs in linear-id order and every XCD has its own L2, so with the identity map neighbouring blocks - which
// share 128-byte lines of the same pyramid rows - land in different L2s and each line crosses the fabric several times
// (measured on k_fast_cells: FETCH_SIZE 2.0x the algorithmic bytes).  With this map XCD x works through a contiguous
// eighth of the (image, block) list, i.e. whole images.  A bijection for any grid size.
__device__ __forceinline__ void xcd_block_map(int &bx, int &b) {
    const unsigned nbx = gridDim.x, total = nbx * gridDim.y, lin = blockIdx.y * nbx + blockIdx.x;
    const unsigned xcd = lin & 7u, idx = lin >> 3, q = total >> 3, r = total & 7u;
    const unsigned pos = xcd * q + min(xcd, r) + idx;   // XCD x owns q (+1 if x < r) consecutive positions
    // the division runs on the vector ALU (no scalar float unit): hand the (uniform) results back to scalar registers so that
    // everything derived from them - image base pointers above all - stays scalar
    b = __builtin_amdgcn_readfirstlane((int)(pos / nbx));
    bx = __builtin_amdgcn_readfirstlane((int)(pos - (unsigned)b * nbx));
}
