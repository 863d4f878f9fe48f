/*
 * orbx.h — C ABI of the MI355X-native ORB front-end + Hamming matchers.
 *
 * This is the drop-in boundary for ONE hot path of kimwin2/ORB_SLAM2v2-1:
 *   ORBextractor::operator()            reference: src/ORBextractor.cc:1043-1105
 *   Frame::ComputeStereoMatches         reference: src/Frame.cc:481-655
 *   ORBmatcher::SearchForInitialization reference: src/ORBmatcher.cc:405-520
 *   ORBmatcher::SearchByProjection x2   reference: src/ORBmatcher.cc:45-129, 1330-1472
 *   ORBmatcher::DescriptorDistance      reference: src/ORBmatcher.cc:1649-1665
 * Plain pointers and sizes only; POD structs; the caller allocates every output.
 * All functions return 0 on success or a negative orbx_status; nothing throws.
 * The implementation is HIP for gfx950 only — there is NO CPU fallback: every entry
 * point that needs the GPU fails with ORBX_ERR_NO_DEVICE when none is usable.
 *
 * Pointers named d_* are DEVICE pointers (e.g. torch tensor .data_ptr()); `stream`
 * is a hipStream_t passed as void* (NULL = the HIP default stream, as in every HIP API; this
 * is also what torch.cuda.current_stream().cuda_stream is for torch's default stream).
 */
#ifndef ORBX_H
#define ORBX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ORBX_OK = 0,
    ORBX_ERR_ARG = -1,         /* bad argument (NULL, non-positive size, level too small, ...) */
    ORBX_ERR_NO_DEVICE = -2,   /* no usable HIP device / kernel image for this GPU */
    ORBX_ERR_HIP = -3,         /* a HIP runtime call failed (see orbx_last_error) */
    ORBX_ERR_CAPACITY = -4,    /* caller-provided capacity too small */
    ORBX_ERR_UNSUPPORTED = -5  /* configuration outside the implemented range */
} orbx_status;

/* cv::KeyPoint layout the reference serialises (include/BoostArchiver.h:47-57): 28 bytes */
typedef struct {
    float x, y;       /* pt, in level-0 pixel units (pt *= mvScaleFactor[octave], :1095-1101) */
    float size;       /* (int)(31 * mvScaleFactor[octave]) */
    float angle;      /* IC_Angle, degrees [0,360) */
    float response;   /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} orbx_keypoint_t;

typedef struct orbx_extractor orbx_extractor_t;

/* ---- extractor: replaces class ORBextractor (include/ORBextractor.h:45-111) ---------- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (src/ORBextractor.cc:410-470).  device = HIP device ordinal. */
int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                int device, orbx_extractor_t **out);
int orbx_destroy(orbx_extractor_t *h);

/* Flavour of the OpenCV primitives behind the reference's calls.  The reference does not contain them (OpenCV, version unpinned:
 * CMakeLists.txt:52-58, README "2.4.11 and 3.2") and ships no vectors, so where OpenCV itself computes a primitive in two ways the
 * choice is a property of the reference BUILD this library stands in for.  It is fixed when the handle is created and never changes
 * afterwards (orbx_get_flavour reads it back); orbx_create = every field 0.
 *   gauss_rounding - cv::GaussianBlur(7x7, sigma 2) on 8U (src/ORBextractor.cc:1085-1086), rounding of the column pass, OpenCV <= 3.3:
 *     ORBX_GAUSS_ROUND_HALF_UP  (sum + 2^15) >> 16 on every column: the scalar FixedPtCastEx (builds without SSE2 / NEON column code)
 *     ORBX_GAUSS_ROUND_SSE2     x86 builds (SSE2 is baseline on x86-64): SymmColumnVec_32s8u rounds the columns x < (w & ~3) of a
 *                               level half to EVEN (_mm_cvtps_epi32); the last w % 4 columns take the scalar half-up code.
 *   The two differ at ~1 pixel in 131 072 (sum mod 65536 == 32768 with an even quotient).
 *     ORBX_GAUSS_FIXED_TAPS     OpenCV >= 3.4.1: the bit-exact 8-bit Gaussian (rows in ufixedpoint16, columns in ufixedpoint32, one
 *                               rounding (sum + 2^15) >> 16, scalar and SIMD code agree) - the arithmetic of HALF_UP on the seven Q8 taps
 *                               gauss_taps[] = { k0 (centre), k1, k2, k3 } the BUILD uses.  { 55, 49, 34, 18 } = cvRound(256 g_i), sum 257:
 *                               identical to HALF_UP (3.4.1 up to the error-diffused kernel); later releases spread the rounding error so
 *                               that the taps add up to 256 - { 56, 48, 34, 18 } by the left-to-right diffusion restated in
 *                               oracle.refvec.diffused_taps.  When in doubt the taps are FITTED from a reference-vector file's blurred
 *                               level (oracle.refvec.fit_gauss_taps; tests/golden/README.md).  k0 + 2 (k1 + k2 + k3) <= 257, k0 >= 1.
 *   Which one a given reference build follows is reported by the reference-vector consumer (tests/golden/README.md); all are
 *   hypotheses until vectors exist. */
#define ORBX_GAUSS_ROUND_HALF_UP 0
#define ORBX_GAUSS_ROUND_SSE2 1
#define ORBX_GAUSS_FIXED_TAPS 2
typedef struct {
    int32_t gauss_rounding;   /* ORBX_GAUSS_ROUND_* / ORBX_GAUSS_FIXED_TAPS */
    int32_t gauss_taps[4];    /* ORBX_GAUSS_FIXED_TAPS: Q8 taps, centre first; every other flavour: must be 0 */
    int32_t reserved[3];      /* must be 0 */
} orbx_flavour_t;
int orbx_create_flavoured(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                          int device, const orbx_flavour_t *flavour, orbx_extractor_t **out);
int orbx_get_flavour(const orbx_extractor_t *h, orbx_flavour_t *out);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:62-83); arrays of nlevels floats,
 * any may be NULL.  features_per_level / umax expose mnFeaturesPerLevel and umax[16]. */
int orbx_get_levels(const orbx_extractor_t *h);
float orbx_get_scale_factor(const orbx_extractor_t *h);
int orbx_get_tables(const orbx_extractor_t *h, float *scale_factors, float *inv_scale_factors,
                    float *level_sigma2, float *inv_level_sigma2, int32_t *features_per_level,
                    int32_t *umax16);
/* upper bound of keypoints one image can produce (nfeatures + 3 per level) */
int orbx_max_keypoints(const orbx_extractor_t *h);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) for one host image
 * (src/ORBextractor.cc:1043-1105).  img: 8-bit gray, `stride` bytes per row.  Writes up
 * to `cap` keypoints (28 B each) and descriptors (32 B each, row-major N x 32) into host
 * buffers and *n_out.  Empty image (w==0||h==0||img==NULL): returns ORBX_OK with
 * *n_out = 0 and outputs untouched (reference :1046-1047).  Synchronous. */
int orbx_extract(orbx_extractor_t *h, const uint8_t *img, int w, int hgt, int stride,
                 orbx_keypoint_t *kps, uint8_t *desc, int cap, int *n_out);

/* Batched many-frame mode on host buffers: B images of identical size.  imgs[b] points
 * at image b.  kps: B*cap entries, desc: B*cap*32 bytes, n_out: B counts.  Synchronous. */
int orbx_extract_batch(orbx_extractor_t *h, const uint8_t *const *imgs, int B, int w, int hgt,
                       int stride, orbx_keypoint_t *kps, uint8_t *desc, int cap, int *n_out);

/* Batched mode on DEVICE buffers (inputs already resident in HBM), asynchronous on
 * `stream`.  d_imgs: B images, image b at d_imgs + b*image_stride_bytes, rows `stride`
 * bytes apart.  d_kps [B][cap], d_desc [B][cap][32], d_counts [B] int32 are device
 * outputs; per-image layout is level-major exactly as the reference concatenates
 * (:1076-1104).  If a frame produces more than cap keypoints its count is clamped to cap
 * (cap >= orbx_max_keypoints() never clamps). */
int orbx_extract_batch_device(orbx_extractor_t *h, const uint8_t *d_imgs, int B, int w, int hgt,
                              int stride, size_t image_stride_bytes, orbx_keypoint_t *d_kps,
                              uint8_t *d_desc, int32_t *d_counts, int cap, void *stream);

/* Software pipelining across batches (optional).  Starts ComputePyramid (src/ORBextractor.cc:1107-1132) of the NEXT batch on a
 * side stream - `side_stream`, or a stream owned by the handle when that is NULL - into a second pyramid buffer, ordered behind
 * the FAST stage of the orbx_extract_batch_device call issued last on this handle: the memory-bound pyramid then runs beside that
 * call's latency-bound gather / quad-tree and its descriptor kernel instead of in front of the next call's FAST.  The next
 * orbx_extract_batch_device call with the SAME (d_imgs, B, w, hgt, stride, image_stride_bytes) takes that pyramid and skips its
 * own; any other call ignores it.
 * Contract: the images must be complete in HBM when this is called (it does not wait for the caller's main stream) and must stay
 * unchanged until that next call has been issued; extraction calls on one handle come from one stream.  Results are identical with
 * or without it. */
int orbx_extract_batch_device_prefetch(orbx_extractor_t *h, const uint8_t *d_imgs, int B, int w, int hgt, int stride,
                                       size_t image_stride_bytes, void *side_stream);

/* Two (default) or three pyramid buffers per handle.  With three, the pyramid built ahead gets a buffer of its own and the previous
 * call's pyramid stays valid across orbx_extract_batch_device_prefetch, so the pattern below may issue the prefetch of batch i+1
 * BEFORE orbm_stereo_batch_device_prev of batch i-1 on the side stream (the matcher then runs beside the descriptor kernel of batch i
 * instead of beside its quad-tree: better when there are many keypoints per image).  Costs one more pyramid buffer of HBM. */
int orbx_set_pyramid_buffers(orbx_extractor_t *h, int n);

/* Orders `stream` behind the FAST stage of the extraction call issued last on h (the point the pyramid built ahead waits for).
 * With orbm_stereo_batch_device_prev this lets a throughput pipeline run the stereo matcher of batch i-1 beside the gather /
 * quad-tree / descriptor kernels of batch i:
 *     orbx_extract_batch_device(h, batch i, main);  orbx_stream_wait_fast_stage(h, side);
 *     orbm_stereo_batch_device_prev(h, h, ... arrays of batch i-1 ..., side);
 *     orbx_extract_batch_device_prefetch(h, batch i+1, ..., side);      (same side stream: the matcher still reads the buffer) */
int orbx_stream_wait_fast_stage(orbx_extractor_t *h, void *stream);

/* The handle's own side stream (hipStream_t), for the pattern above.  HIP deals streams to a handful of hardware queues
 * round-robin, so a stream the caller creates may share its queue with the main stream and overlap nothing. */
void *orbx_side_stream(orbx_extractor_t *h);
/* The same, made sure not to share a hardware queue with main_stream (two streams on one queue run strictly in order and overlap
 * nothing): probed once with a 2-ms one-wave spin on main_stream and an empty kernel on the candidate, replaced by a fresh stream
 * if it queued behind the spin.  Call it once per handle with the stream the extraction calls will be issued on (every handle
 * after the first of a process is likely to need the replacement).  Returns the side stream to use. */
void *orbx_side_stream_for(orbx_extractor_t *h, void *main_stream);

/* mvImagePyramid[level] of image `b` of the last call (include/ORBextractor.h:85): copies
 * the inner level (padded=0) or the whole bordered buffer (padded=1, 19 px border,
 * BORDER_REFLECT_101) into dst (dst_stride bytes per row) and reports its size. dst may
 * be NULL to query the size only.  Synchronises the handle's last stream. */
int orbx_pyramid_host(orbx_extractor_t *h, int b, int level, int padded, uint8_t *dst, int dst_stride,
                      int *w, int *hgt);
/* Device view of the same level: pointer to the inner ROI's first pixel and its row stride. */
int orbx_pyramid_device(orbx_extractor_t *h, int b, int level, const uint8_t **d_ptr, int *w, int *hgt,
                        int *stride);

/* FAST candidates kept per level (what vToDistributeKeys holds, src/ORBextractor.cc:789-829) and keypoints DistributeOctTree returned
 * per level for image slot b of the last call: candidates[nlevels], keypoints[nlevels] (either may be NULL).  Read-only statistics. */
int orbx_level_counts(orbx_extractor_t *h, int b, int32_t *candidates, int32_t *keypoints);

/* Per-stage GPU time in ms, measured with HIP events recorded on the launch stream around
 * each stage of every batch call while profiling is enabled (a ring of event sets, so no
 * call ever waits for the GPU): [0] pyramid (k_pyr_*) [1] FAST cells (k_fast_cells)
 * [2] quad-tree (k_octree) [3] orientation+blur+descriptor (k_describe) [4] whole call.
 * orbx_get_stage_ms returns the AVERAGE per call since orbx_set_profiling(h,mode) and the number
 * of calls averaged (ncalls may be NULL).  mode 0: off; 1: all stage boundaries; 2: only the two events
 * around k_fast_cells ([1] is filled, the rest stays 0) - every recorded event idles the GPU for ~4.5 us,
 * so a throughput run brackets just the kernel whose duration it reports; 3: as 2, on every 4th call only (the average is over the
 * bracketed calls). */
#define ORBX_NUM_STAGES 5
/* Stage [1] is k_fast_strips (one wave per strip of four cells; levels whose cells are at most 32 px wide, batches that fill
 * the GPU) and / or k_fast_cells (one wave per cell; the other levels, small batches): which of them a batch of B images of
 * the planned size runs.  Same results either way.  A call is ONE chunk by default (ORBX_OPT_CHUNKS cuts a batch into up to
 * four chunks whose kernels overlap on the handle's side streams; the call keeps its stream semantics); the events of stage
 * [1] bracket the first chunk's launch, which covers *images_per_launch images (= B by default). */
int orbx_fast_kernels(const orbx_extractor_t *h, int B, int *strips, int *cells, int *images_per_launch);
int orbx_set_profiling(orbx_extractor_t *h, int enabled);
int orbx_get_stage_ms(orbx_extractor_t *h, float *ms5, int *ncalls);

/* ---- matchers: replace the hot ORBmatcher / Frame routines --------------------------- */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1649-1665): host, pure. */
int orbm_hamming(const uint8_t *a32, const uint8_t *b32);

/* All-pairs 256-bit Hamming distances on device: d_out[i*nb + j] (uint16) =
 * distance(d_a[i], d_b[j]).  Building block + bandwidth probe of the matchers. */
int orbm_hamming_matrix_device(const uint8_t *d_a, int na, const uint8_t *d_b, int nb,
                               uint16_t *d_out, void *stream);

/* Multi-GPU result exchange: pack the device outputs of B frames into ONE fixed-capacity record per
 * frame so that a step needs a single all-gather:
 *   [ cap x 28 B keypoints | cap x 32 B descriptors | cap x 4 B mvuRight | cap x 4 B mvDepth | int32 count | 12 B zero ]
 * record size = orbx_record_bytes(cap) = 68*cap + 16.  d_uright / d_depth may be NULL (monocular: zeros).  One kernel,
 * asynchronous on stream.  orb_slam2v2-1_amd/batching.py holds the matching unpack. */
int orbx_record_bytes(int cap);
int orbx_pack_records_device(const orbx_keypoint_t *d_kps, const uint8_t *d_desc, const float *d_uright,
                             const float *d_depth, const int32_t *d_counts, int B, int cap, uint8_t *d_records,
                             void *stream);

/* Frame::ComputeStereoMatches (src/Frame.cc:481-655) for B stereo frames.  Left/right
 * keypoints + descriptors + counts are the DEVICE outputs of two extractors (hl, hr) whose
 * pyramids of the same batch are still resident.  d_uright / d_depth: [B][cap] float
 * (mvuRight / mvDepth, -1 = no match).  mbf = baseline*fx, mb = mbf/fx (Frame.cc:114).
 * d_nmatch [B] (may be NULL) = number of surviving matches. Asynchronous on stream.
 * Frame b reads the pyramid of image slot left_slot0+b of hl and right_slot0+b of hr; hl and
 * hr may be the SAME handle when one extractor processed left and right images in one batch
 * (e.g. slots [0,B) left, [B,2B) right). */
int orbm_stereo_batch_device(orbx_extractor_t *hl, orbx_extractor_t *hr, int B,
                             int left_slot0, int right_slot0,
                             const orbx_keypoint_t *d_kl, const uint8_t *d_dl, const int32_t *d_nl,
                             const orbx_keypoint_t *d_kr, const uint8_t *d_dr, const int32_t *d_nr,
                             int cap, float mbf, float mb, float *d_uright, float *d_depth,
                             int32_t *d_nmatch, void *stream);

/* orbm_stereo_batch_device on the pyramids of the extraction call BEFORE the last one on hl / hr.  Valid only while that
 * pyramid still exists: the last extraction call took a pyramid built ahead (the two buffers swapped) and the next
 * orbx_extract_batch_device_prefetch has not been issued yet - otherwise ORBX_ERR_ARG.  Issue it on the stream that builds the
 * next pyramid afterwards (stream order protects the buffer). */
int orbm_stereo_batch_device_prev(orbx_extractor_t *hl, orbx_extractor_t *hr, int B,
                             int left_slot0, int right_slot0,
                             const orbx_keypoint_t *d_kl, const uint8_t *d_dl, const int32_t *d_nl,
                             const orbx_keypoint_t *d_kr, const uint8_t *d_dr, const int32_t *d_nr,
                             int cap, float mbf, float mb, float *d_uright, float *d_depth,
                             int32_t *d_nmatch, void *stream);
/* One stereo frame, host to host, in ONE call: the stereo Frame constructor's ExtractORB(left) || ExtractORB(right) followed by
 * ComputeStereoMatches (src/Frame.cc:78-84, 481-655).  Both images are extracted as one batch of two on h (a second extractor handle
 * is not needed), matched on the device and everything comes down behind one synchronisation.  kl / dl / uright / depth: left
 * keypoints, descriptors, mvuRight, mvDepth (cap entries each); kr / dr: right keypoints and descriptors; *nl, *nr, *nmatch counts.
 * More than cap keypoints: clamped, ORBX_ERR_CAPACITY.  Empty image: ORBX_OK with zero counts.  Synchronous. */
int orbx_stereo_frame(orbx_extractor_t *h, const uint8_t *left, const uint8_t *right, int w, int hgt, int stride,
                      float mbf, float mb, int cap, orbx_keypoint_t *kl, uint8_t *dl, int *nl, orbx_keypoint_t *kr,
                      uint8_t *dr, int *nr, float *uright, float *depth, int *nmatch);

/* The LATENCY form of the same call (round 5) - the shape the reference runs this path in: ONE stereo frame at a time, the next
 * frame needing this one's pose (src/Tracking.cc:275 -> Frame::Frame, src/Frame.cc:61-120).  Nothing is copied by a copy command:
 *  - left / right may be PINNED host memory (orbx_host_alloc, hipHostMalloc, hipHostRegister: the first kernel reads the pixels
 *    over the bus itself), device memory, or ordinary pageable memory (then the call stages them through a pinned buffer of the
 *    handle with one memcpy each - the price of a pageable cv::Mat; allocate capture buffers with orbx_host_alloc to avoid it);
 *  - the results are one fixed-layout record per frame that the LAST kernel writes to pinned host memory of the handle; *view
 *    points into it (host) and into the same record in HBM (d_*: what the device-resident matchers orbm_*_device take).
 * The handle keeps TWO records and alternates: the view of a call stays valid until the call after the next one on the same
 * handle (the previous frame's descriptors are what SearchByProjection(cur, last) reads, src/ORBmatcher.cc:1330-1472).
 * Results are those of orbx_stereo_frame bit for bit.  Empty image: ORBX_OK, zero counts, pointers NULL.  Synchronous. */
typedef struct {
    int32_t nl, nr, nmatch, cap;              /* keypoints left / right, stereo matches, rows allocated per array */
    const orbx_keypoint_t *kl, *kr;           /* host (pinned, owned by the handle) */
    const uint8_t *dl, *dr;                   /* 32 B per keypoint */
    const float *uright, *depth;              /* mvuRight, mvDepth of the left keypoints */
    const orbx_keypoint_t *d_kl, *d_kr;       /* the same arrays in HBM */
    const uint8_t *d_dl, *d_dr;
    const float *d_uright, *d_depth;
} orbx_stereo_view_t;
int orbx_stereo_frame_view(orbx_extractor_t *h, const uint8_t *left, const uint8_t *right, int w, int hgt, int stride,
                           float mbf, float mb, orbx_stereo_view_t *view);
/* Pinned host memory for image / capture buffers (cv::Mat can wrap it: cv::Mat(rows, cols, CV_8UC1, ptr)); NULL on failure. */
void *orbx_host_alloc(size_t bytes);
void orbx_host_free(void *p);

/* Host-buffer convenience for one frame (synchronous); pyramids come from hl / hr, which
 * must have just extracted the left / right image (image slot 0). */
int orbm_stereo(orbx_extractor_t *hl, orbx_extractor_t *hr,
                const orbx_keypoint_t *kl, const uint8_t *dl, int nl,
                const orbx_keypoint_t *kr, const uint8_t *dr, int nr,
                float mbf, float mb, float *uright, float *depth, int *nmatch);

/* Frame lookup grid geometry (src/Frame.cc:90-105, 397-407; include/Frame.h:37-38) */
typedef struct {
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float inv_w, inv_h;               /* mfGridElementWidthInv, mfGridElementHeightInv */
} orbm_grid_geom_t;

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (src/ORBmatcher.cc:405-520) on host arrays (undistorted keypoints).  prev_matched
 * [2*n1] (x,y) is updated in place; matches12[n1] out; *nmatches out. */
int orbm_search_for_initialization(const orbx_keypoint_t *k1, const uint8_t *d1, int n1,
                                   const orbx_keypoint_t *k2, const uint8_t *d2, int n2,
                                   const orbm_grid_geom_t *g2, float *prev_matched, int32_t *matches12,
                                   int window, float nnratio, int check_orientation, int device,
                                   int *nmatches);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:45-129), map points gathered into flat arrays by the C++ wrapper. */
typedef struct {
    int32_t in_view;          /* mbTrackInView && !isBad() */
    float proj_x, proj_y, proj_xr; /* mTrackProjX/Y/XR */
    int32_t level;            /* mnTrackScaleLevel */
    float view_cos;           /* mTrackViewCos */
    int32_t observations;     /* Observations() */
} orbm_mappoint_t;
/* frame_mp[n] in/out: index of the list's map point held by keypoint i, -1 none,
 * -2 = held by a map point outside the list whose Observations() is ext_obs[i]. */
int orbm_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                 const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                 const orbm_mappoint_t *mps, const uint8_t *mp_desc, int m,
                                 int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio,
                                 int device, int *nmatches);

/* ORBmatcher::SearchByProjection(Frame &cur, const Frame &last, th, bMono)
 * (src/ORBmatcher.cc:1330-1472). */
typedef struct {
    int32_t has_mp;        /* mvpMapPoints[i] && !mvbOutlier[i] */
    float wx, wy, wz;      /* GetWorldPos() */
    int32_t observations;  /* Observations() */
    int32_t octave;        /* LastFrame.mvKeys[i].octave */
    float angle;           /* LastFrame.mvKeysUn[i].angle */
} orbm_lastpoint_t;
typedef struct { float fx, fy, cx, cy, mbf, mb; } orbm_camera_t;
/* cur_mp[n] in/out: index i of the last-frame keypoint whose map point keypoint j holds. */
int orbm_search_by_projection_frame(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright,
                                    int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                    int nlevels, const orbm_camera_t *cam, const float *Tcw_cur16,
                                    const float *Tcw_last16, const orbm_lastpoint_t *last,
                                    const uint8_t *last_desc, int nlast, int32_t *cur_mp,
                                    const int32_t *ext_obs, float th, int mono, int check_orientation,
                                    int device, int *nmatches);

/* Device-resident form for a tracking front-end (Tracking::TrackWithMotionModel, src/Tracking.cc:1010-1041): the current
 * frame's keypoints, descriptors and mvuRight are the DEVICE outputs of orbx_extract_batch_device / orbm_stereo_batch_device
 * (first n entries of the frame's rows) and d_last_desc the previous frame's descriptor rows, still in HBM; only the
 * per-keypoint map-point records (`last`, 28 B each), the poses and the holders cross the bus.  The kernels run on
 * `stream`, i.e. behind the kernels that produce the arrays; the call returns when the results are on the host.
 * Same results as orbm_search_by_projection_frame. */
int orbm_search_by_projection_frame_device(const orbx_keypoint_t *d_kun, const uint8_t *d_desc, const float *d_uright,
                                           int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                           int nlevels, const orbm_camera_t *cam, const float *Tcw_cur16,
                                           const float *Tcw_last16, const orbm_lastpoint_t *last,
                                           const uint8_t *d_last_desc, int nlast, int32_t *cur_mp,
                                           const int32_t *ext_obs, float th, int mono, int check_orientation,
                                           int device, int *nmatches, void *stream);

/* ---- the step before SearchByProjection(F, MPs) — Frame::isInFrustum
 * (src/Frame.cc:284-340) for all local map points at once (Tracking::SearchLocalPoints,
 * src/Tracking.cc:1290-1340).  A point's tracking variables come back as orbm_mappoint_t. */
typedef struct {
    int32_t valid;                      /* candidate: not already matched in this frame, !isBad() (:1306-1317) */
    float wx, wy, wz;                   /* GetWorldPos() */
    float nx, ny, nz;                   /* GetNormal() */
    float max_distance, min_distance;   /* mfMaxDistance, mfMinDistance (invariance range = 1.2x / 0.8x) */
    int32_t observations;               /* Observations(), copied to the output record */
} orbm_worldpoint_t;
/* MapPoint::PredictScale (src/MapPoint.cc:414-429) is ceil(log(ratio)/mfLogScaleFactor) with the C
 * library's float log, which a GPU cannot reproduce bit for bit at level boundaries.  The device
 * therefore compares ratio = mfMaxDistance/dist with nlevels-1 float thresholds that this HOST function
 * finds by bisection with the host's own logf: thresholds[k] = the smallest float r with
 * ceil(logf(r)/log_scale_factor) >= k+1.  Level = number of thresholds <= ratio (exact wherever logf is
 * monotonic).  No GPU needed. */
int orbm_predict_scale_thresholds(float log_scale_factor, int nlevels, float *thresholds);
/* Tcw16: row-major 4x4 camera pose; g: image bounds mnMinX..mnMaxY; out[m] as Frame::isInFrustum leaves
 * mbTrackInView / mTrackProjX / mTrackProjXR / mTrackProjY / mnTrackScaleLevel / mTrackViewCos. */
int orbm_is_in_frustum(const orbm_worldpoint_t *pts, int m, const float *Tcw16, const orbm_camera_t *cam,
                       const orbm_grid_geom_t *g, float viewing_cos_limit, const float *thresholds, int nlevels,
                       orbm_mappoint_t *out, int device);
/* isInFrustum + SearchByProjection(F, MPs) without the host round trip of the projections
 * (Tracking::SearchLocalPoints).  proj_out[m] (may be NULL) receives the isInFrustum records. */
int orbm_search_local_points(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                             const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                             const orbm_worldpoint_t *pts, const uint8_t *mp_desc, int m, const float *Tcw16,
                             const orbm_camera_t *cam, float viewing_cos_limit, const float *thresholds,
                             int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio, int device,
                             int *nmatches, orbm_mappoint_t *proj_out);

/* orbm_search_local_points with the frame's keypoints / descriptors / mvuRight already in HBM (see
 * orbm_search_by_projection_frame_device); the map points' records and descriptors come from the host map. */
int orbm_search_local_points_device(const orbx_keypoint_t *d_kun, const uint8_t *d_desc, const float *d_uright, int n,
                                    const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                    const orbm_worldpoint_t *pts, const uint8_t *mp_desc, int m, const float *Tcw16,
                                    const orbm_camera_t *cam, float viewing_cos_limit, const float *thresholds,
                                    int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio, int device,
                                    int *nmatches, orbm_mappoint_t *proj_out, void *stream);

/* Generic projected-window matcher: the common core of the reference's SearchByProjection
 * family once the caller has projected its map points.  For every valid
 * query, in order: candidates = Frame::GetFeaturesInArea(u, v, radius, min_level, max_level)
 * (src/Frame.cc:342-395); a candidate is skipped when its slot already has a blocking holder
 * (and, if ur_tol >= 0, when uright[j] > 0 && |ur_c - uright[j]| > ur_tol); the first minimum
 * of the Hamming distance wins; if it is <= max_dist, holder[best] = query index, nmatches++,
 * and with check_orientation the match enters the 30-bin rotation histogram
 * (query angle - keypoint angle) whose bins outside the three maxima are undone at the end
 * (holder = -1, nmatches--), exactly as src/ORBmatcher.cc:1549-1597 does.
 * holder[n] in/out: -1 empty, >= 0 index of the query holding the slot, -2 held from outside
 * (ext_blocks[j] != 0 says whether that holder blocks; NULL = it blocks).  A query's own
 * claim blocks later queries iff queries[q].blocks != 0.
 * ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1474-1601) = projection + MapPoint::PredictScale on the host
 * (orb_slam2v2-1_amd/host/ORBmatcher.cc) + this call with blocks = 1, max_dist = ORBdist. */
typedef struct {
    int32_t valid;
    float u, v, radius;
    int32_t min_level, max_level;
    float angle;      /* orientation of the query's own keypoint (pKF->mvKeysUn[i].angle) */
    int32_t blocks;
    float ur_c, ur_tol;
} orbm_window_query_t;
/* g_assign (both calls below): a KeyFrame keeps the cell lists its Frame built with float bounds
 * (src/KeyFrame.cc:41,53) but queries them with int-truncated bounds (include/KeyFrame.h:199-202,
 * src/KeyFrame.cc:577-589): pass the Frame's geometry here and the KeyFrame's in g.  NULL = g. */
int orbm_match_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                       const orbm_grid_geom_t *g, const orbm_grid_geom_t *g_assign,
                       const orbm_window_query_t *queries, const uint8_t *query_desc,
                       int m, int32_t *holder, const int32_t *ext_blocks, int max_dist, int check_orientation,
                       int device, int *nmatches);

/* Stateless projected-window search: the matching core of ORBmatcher::Fuse (both overloads,
 * src/ORBmatcher.cc:827-977, 979-1102) and ORBmatcher::SearchBySim3 (:1104-1328), whose queries
 * never read what an earlier query wrote.  For every valid query: candidates =
 * KeyFrame::GetFeaturesInArea(u, v, radius) (src/KeyFrame.cc:572-611, same cells and order as
 * Frame's) with octave in [min_level, max_level] (the callers pass [pred-1, pred]); with
 * inv_level_sigma2 != NULL each candidate must also pass Fuse's reprojection gate (:916-940):
 * uright[j] >= 0 ? (ex^2+ey^2+er^2)*inv_level_sigma2[octave] <= 7.8, er = ur_c - uright[j]
 *               : (ex^2+ey^2)*inv_level_sigma2[octave] <= 5.99      (uright NULL = all mono).
 * Out: best_idx[m] = first minimum of the Hamming distance in scan order (-1: no candidate),
 * best_dist[m] (256 when none).  The caller applies its own threshold (TH_LOW / TH_HIGH) and
 * map-point bookkeeping (orb_slam2v2-1_amd/host/ORBmatcher.cc). angle, blocks, ur_tol unused. */
int orbm_best_in_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                         const orbm_grid_geom_t *g, const orbm_grid_geom_t *g_assign,
                         const orbm_window_query_t *queries, const uint8_t *query_desc,
                         int m, const float *inv_level_sigma2, int nlevels, int32_t *best_idx, int32_t *best_dist,
                         int device);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317), batched over map points
 * (SURVEY §8(f) rank 4).  Map point p owns the descriptors desc[offsets[p] .. offsets[p+1]) (the rows
 * of its non-bad observing keyframes, in std::map iteration order).  For each row i: all Hamming
 * distances to the point's rows (itself included, 0), median = sorted[(size_t)(0.5*(N-1))]; the FIRST
 * row with the least median wins.  Out: best_row[p] (index inside the point's block, -1 if it has no
 * rows), best_median[p] (may be NULL).  The caller copies row best_row[p] into mDescriptor. */
int orbm_distinctive_descriptors(const uint8_t *desc, const int32_t *offsets, int npoints, int32_t *best_row,
                                 int32_t *best_median, int device);

/* Per-handle options: which of several kernels / launch arrangements with IDENTICAL results a handle uses.  State of the handle
 * (no process-global switch exists: two handles on two threads may run with different options); set between calls, from the thread
 * that issues the handle's calls.  Defaults (every value 0) are the measured-fastest choices; the tests use the others to cover
 * every kernel against the oracle.  Unknown keys / values: ORBX_ERR_ARG.
 *   ORBX_OPT_PYR_TILE      3   tile size of the fused pyramid kernel at the coarsest level (0 = 64; takes effect at the next plan)
 *   ORBX_OPT_OCTREE_FORM   4   1 = quad-tree by the exact form alone, 2 = every level by the multi-workgroup form, 3 = no level by it
 *                              (default: levels with >= 600 FAST cells when the batch is at most 4 images)
 *   ORBX_OPT_PYRAMID_FORM  5   1 = pyramid by the one-launch fused kernel, 3 = levels 3.. by it (default: one launch per level)
 *   ORBX_OPT_FAST_FORM     6   1 = every level's FAST by k_fast_cells, 2 = ... with run-time tile strides, 3 = k_fast_strips even for
 *                              a small batch (default: by batch size)
 *   ORBX_OPT_CHUNKS        8   n >= 2: a batch is cut into n chunks (<= 4) on the handle's side streams (default one)
 *   ORBX_OPT_IGNORE_AHEAD  9   1 = pyramids built ahead are ignored
 *   ORBX_OPT_PREFETCH_GATE 10  where a pyramid built ahead may start: 0 behind FAST, 1 behind the quad-tree, 2 behind the descriptors,
 *                              3 together with FAST (pays when the quad-tree launch fills the GPU more than once and nothing else runs
 *                              beside FAST: pipeline.FrontEnd picks it for such mono batches)
 *   ORBX_OPT_OCTREE_WIDTH  11  quad-tree kernels in the 1024-thread build never (1) / always (2) (default: by image size)
 *   ORBX_OPT_NO_ORDER_KERNEL 12  1 = no ordering kernel in front of the FAST stage's start event
 *   ORBX_OPT_BLUR_FORM     13  1 = no level blurred as a whole, 2 = every level (orbx_debug_blurred_level); 14 = the rule's threshold
 *   ORBX_OPT_SPLIT_CALL    15  a >= 2: quad-tree of levels [0, a) on a second stream beside quad-tree + descriptors of the others
 *   ORBX_OPT_ROW_PRETEST   16  corner-sparse FAST path of k_fast_strips never (1) / on every level (2) (default: by the previous
 *                              call's candidate density)
 *   ORBX_OPT_GATHER        18  1 = k_gather + compacted key arrays instead of the quad-tree reading the FAST cell lists in place
 *   ORBX_OPT_EARLY_OCTREE  19  a >= 2: strips of levels [0, a) launched first, their quad-tree beside the FAST of the others
 *   ORBX_OPT_SPARSE_FORM   20  corner-sparse levels: 0 = rows of 128 pixels that cannot hold a corner are skipped inside k_fast_strips
 *                              (default), 1 = k_fast_strips_sparse (the pixel pairs that pass the five-pixel bound are queued and
 *                              scored 64 at a time), 2 = the same compaction inside the cell kernel; 1 and 2 measured no faster
 *   ORBX_OPT_DESC_LDS_PAD  21  KB of unused LDS per k_describe workgroup: fewer of them resident per CU, more wave slots for the
 *                              kernels of a neighbouring stream (tuning of the pipelined step; default 0)
 *   ORBX_OPT_PYR_ROWS      22  output rows per wave of k_pyr_level: 1 = always 8, 2 = always 16 (default: 16 while the level gives
 *                              every SIMD several waves, else 8).  2 only pays up to scale factor 1.25: beyond it 16 output rows
 *                              need more than the 22 source rows the form fetches and every band takes the row-by-row path
 *                              (correct, much slower than the 8-row form)
 *   ORBX_OPT_OCT_HIST      23  quad-tree input of batches of up to four images: 0 = the FAST stage histograms its emissions at the L2
 *                              and k_octree_pyr loads the histogram (no key sweep on one CU: the latency form, default), 1 = never
 *   ORBX_OPT_PAD_FORM      24  level 0 (copy of the input + copyMakeBorder): 0 = source rows staged in LDS for batches of up to four
 *                              images (each input byte read once, aligned: the input of a latency call sits in pinned host memory),
 *                              1 = never (k_pyr_pad: unaligned 16-byte chunks), 2 = always
 *   ORBX_OPT_PYR_CHAINS    25  levels >= 1 of batches of up to four images: 0 = in chains of up to four levels per launch (k_pyr_chain:
 *                              a tile's intermediate levels stay in LDS - per-level launches of one or two images are latency-bound),
 *                              1 = one launch per level as in a large batch, 2 / 3 = chains of up to four / seven levels.  ORBX_OPT_PYRAMID_FORM = 4
 *                              forces the chains at any batch size
 *   ORBX_OPT_OCT_SLICES    26  quad-tree of a batch: 1 = the key sweep of a level with >= 600 FAST cells is shared by two workgroups, >= 1600
 *                              by four (partial histograms summed by the last to arrive); 0 = one workgroup per level (default: the
 *                              shared form is faster alone and slower inside the pipelined step)
 *   ORBX_OPT_STREAM_SYNC   27  orbx_stereo_frame_view: 0 = the host polls the completion word the last kernel stores behind the record (the
 *                              stream wait as the fall-back), 1 = it waits for the stream
 * Keys 0, 1 and 7 (stop a kernel after phase n: outputs incomplete) exist only in a library built with -DORBX_DEVELOPER
 * (tools/octree_phase_probe.py); the default build refuses them. */
#define ORBX_OPT_PYR_TILE 3
#define ORBX_OPT_OCTREE_FORM 4
#define ORBX_OPT_PYRAMID_FORM 5
#define ORBX_OPT_FAST_FORM 6
#define ORBX_OPT_CHUNKS 8
#define ORBX_OPT_IGNORE_AHEAD 9
#define ORBX_OPT_PREFETCH_GATE 10
#define ORBX_OPT_OCTREE_WIDTH 11
#define ORBX_OPT_NO_ORDER_KERNEL 12
#define ORBX_OPT_BLUR_FORM 13
#define ORBX_OPT_BLUR_THRESHOLD 14
#define ORBX_OPT_SPLIT_CALL 15
#define ORBX_OPT_ROW_PRETEST 16
#define ORBX_OPT_GATHER 18
#define ORBX_OPT_EARLY_OCTREE 19
#define ORBX_OPT_SPARSE_FORM 20
#define ORBX_OPT_DESC_LDS_PAD 21
#define ORBX_OPT_PYR_ROWS 22
#define ORBX_OPT_OCT_HIST 23
#define ORBX_OPT_PAD_FORM 24
#define ORBX_OPT_PYR_CHAINS 25
#define ORBX_OPT_OCT_SLICES 26
#define ORBX_OPT_STREAM_SYNC 27
#define ORBX_NUM_OPTIONS 32
int orbx_set_option(orbx_extractor_t *h, int key, int value);
int orbx_get_option(const orbx_extractor_t *h, int key, int *value);
/* The guided-search matchers keep their scratch per host thread, and so their options: ORBM_OPT_EXACT_KERNELS = 1 makes the
 * calling thread's matcher calls take the exact one-workgroup kernels instead of candidate search + resolution; ORBM_OPT_RESOLVER
 * = 1 makes the resolution the single-wave speculative walk of rounds 1-4 instead of the whole-workgroup fixed-point iteration
 * (k_resolve_par, the default since round 5).  Identical results whatever the setting; the tests run all three.  Thread-local. */
#define ORBM_OPT_EXACT_KERNELS 2
#define ORBM_OPT_RESOLVER 3
#define ORBM_OPT_STREAM_SYNC 4   /* 1 = a guided search waits for its stream (hipStreamSynchronize); 0 (default) = it polls the completion word its
                                   * last kernel stores behind the results in pinned memory, with the stream wait as the fall-back after a few ms */
int orbm_set_thread_option(int key, int value);
/* The read-only stage hooks the staged parity tests look through (orbx_debug_* / orbm_debug_*) are NOT part of this library: they
 * exist in the developer build only (liborbx_hip_dev.so, -DORBX_DEVELOPER) and are declared in include/orbx_dev.h. */

/* ---- misc ---------------------------------------------------------------------------- */
/* ---- DBoW2 vocabulary descent and the BoW-guided matchers.
 * Vocabulary = the tree of Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:310-340 (nodes: parent, children in id
 * order, 32-byte FORB descriptor, weight; leaves are words, numbered in node-id order) held on the GPU. */
typedef struct orbv_vocabulary orbv_vocabulary_t;
/* nnodes includes the root (node 0, no descriptor).  parent[i] < i for i >= 1.  is_leaf[i]: the loader's nIsLeaf
 * flag (word ids follow it); a node is treated as a leaf by the descent iff it has no children (Node::isLeaf). */
int orbv_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent, const uint8_t *is_leaf,
                const uint8_t *desc, const double *weight, int device, orbv_vocabulary_t **out);
/* TemplatedVocabulary::loadFromTextFile (:1351-1436, the ORBvoc.txt format).  Difference: the reference's
 * `while(!f.eof())` loop turns the file's trailing empty line into one more child of the root with an
 * UNINITIALISED descriptor (undefined behaviour); empty lines are skipped here. */
int orbv_load_text(const char *path, int device, orbv_vocabulary_t **out);
void orbv_destroy(orbv_vocabulary_t *v);
int orbv_info(const orbv_vocabulary_t *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords);
/* transform(feature, word_id, weight, nid, levelsup) of TemplatedVocabulary.h:1230-1271 for n descriptors:
 * at every level the child with the smallest FORB::distance (first minimum in child order) until a leaf.
 * word_id[n], node_id[n] (the node on the path at level L - levelsup; 0 when that level is <= 0), weight[n].
 * The BowVector / FeatureVector maps are filled from these on the host, in feature order
 * (orb_slam2v2-1_amd/host/ORBVocabulary.h), because their double sums depend on that order. */
int orbv_transform(const orbv_vocabulary_t *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                   int32_t *node_id, double *weight);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-288) and SearchByBoW(KeyFrame*, KeyFrame*,
 * ...) (:522-655) after the caller has intersected the two FeatureVectors: node j pairs the query features
 * q_items[node_qstart[j] .. node_qstart[j+1]) with the candidate features c_items[node_cstart[j] .. ).  Queries are
 * visited in that order; candidates in list order; a query needs q_valid (map point present and not bad), a
 * candidate needs c_valid (NULL = all valid) and must not have been taken by an earlier query; accept iff
 * best <= max_dist && (float)best < nnratio * (float)second (second = smallest distance among the other
 * candidates, 256 if none); with check_orientation the matches outside the three largest 30-degree bins of
 * (q_angle - c_angle) are dropped at the end.  A feature belongs to one node only, so nodes are independent:
 * one wavefront per node.  match_q[nq]: candidate feature index or -1. */
int orbm_search_by_bow(const uint8_t *q_desc, const float *q_angle, const uint8_t *q_valid, int nq,
                       const uint8_t *c_desc, const float *c_angle, const uint8_t *c_valid, int nc,
                       const int32_t *node_qstart, const int32_t *q_items, const int32_t *node_cstart,
                       const int32_t *c_items, int nnodes, int max_dist, float nnratio, int check_orientation,
                       int32_t *match_q, int *nmatches, int device);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-825) on the intersected node lists (same layout as
 * orbm_search_by_bow).  flags: bit 0 = usable (query: no map point yet :703-704 and, with bOnlyStereo, stereo
 * :708-710; candidate: no map point :724-725 and the stereo rule :729-731), bit 1 = mvuRight >= 0.  For each
 * usable query, over the usable candidates of its node not taken by an earlier query, in list order:
 * dist <= TH_LOW and dist <= best so far (:737), then - both monocular - the candidate must be at least
 * sqrt(100*scale_factors[octave2]) away from the epipole (ex, ey) (:742-748), then CheckDistEpipolarLine
 * (:140-157: squared distance to the line x1'F12 < 3.84*level_sigma2[octave2]); a passing candidate becomes the
 * best — so among equal distances the LAST passing one wins.  F12_9: row-major 3x3.  Rotation consistency as in
 * orbm_search_by_bow.  match_q[nq]: candidate feature index or -1. */
int orbm_search_for_triangulation(const orbx_keypoint_t *kp1, const uint8_t *q_desc, const uint8_t *q_flags, int nq,
                                  const orbx_keypoint_t *kp2, const uint8_t *c_desc, const uint8_t *c_flags, int nc,
                                  const int32_t *node_qstart, const int32_t *q_items, const int32_t *node_cstart,
                                  const int32_t *c_items, int nnodes, const float *F12_9, float ex, float ey,
                                  const float *scale_factors, const float *level_sigma2, int nlevels, int max_dist,
                                  int check_orientation, int32_t *match_q, int *nmatches, int device);

/* The host-array matcher entry points keep grow-only device scratch, a pinned mirror and one non-blocking stream
 * PER HOST THREAD (re-entrant without locks: the reference calls matchers from Tracking, LocalMapping and LoopClosing
 * threads at once, src/LocalMapping.cc:223, src/LoopClosing.cc:249).  Nothing is freed implicitly; a thread calls
 * this before it exits or to hand the memory back.  Safe to call at any time, any number of times. */
int orbx_thread_release_scratch(void);

const char *orbx_last_error(void);      /* thread-local description of the last failure */
const char *orbx_version(void);
int orbx_device_count(void);            /* number of HIP devices visible (0 if none) */

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
