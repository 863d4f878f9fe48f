// orbx_match.hip — MI355X (gfx950) Hamming matchers: hand-written HIP kernels + C ABI.
//
//   orbm_hamming                     ORBmatcher::DescriptorDistance         (src/ORBmatcher.cc:1649-1665)
//   k_hamming_matrix                 all-pairs 256-bit distances (building block / bandwidth probe)
//   k_stereo_match + k_stereo_median Frame::ComputeStereoMatches            (src/Frame.cc:481-655)
//   k_search_init                    ORBmatcher::SearchForInitialization    (src/ORBmatcher.cc:405-520)
//   k_search_proj_mp                 ORBmatcher::SearchByProjection(F,MPs)  (src/ORBmatcher.cc:45-129)
//   k_search_proj_frame              ORBmatcher::SearchByProjection(F,F)    (src/ORBmatcher.cc:1330-1472)
//
// Descriptor distance = 4 x popcount64 of the XOR; argmin / top-2 reductions run on packed
// 64-bit keys (distance | scan-order | index) with wavefront shuffles, so the result is the
// same element the reference's sequential "first strictly smaller wins" scan selects.
#include "orbx_match_dev.h"
#define ORBX_FAST_FALLBACK_RC 1   // what the fast_* entry points return when the exact kernels must run
#include <math.h>
#include <algorithm>
#include <vector>

// orbm_set_thread_option (include/orbx.h): like the matchers' scratch, their one option is per host thread - no process-global state
thread_local int t_matchExact = 0;
thread_local int t_matchResolver = 0;
thread_local int t_matchStreamSync = 0;
extern "C" int orbm_set_thread_option(int key, int value) {
    if ((key != ORBM_OPT_EXACT_KERNELS && key != ORBM_OPT_RESOLVER && key != ORBM_OPT_STREAM_SYNC) || (value != 0 && value != 1)) { orbx_set_error("orbm_set_thread_option: key %d / value %d", key, value); return ORBX_ERR_ARG; }
    (key == ORBM_OPT_EXACT_KERNELS ? t_matchExact : key == ORBM_OPT_RESOLVER ? t_matchResolver : t_matchStreamSync) = value;
    return ORBX_OK;
}

extern "C" int orbm_hamming(const uint8_t *a, const uint8_t *b) {
    if (!a || !b) return ORBX_ERR_ARG;
    int d = 0;
    for (int i = 0; i < 4; i++) {
        u64 x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

// ------------------------------------------------------------------------------------
// all-pairs Hamming: thread j keeps descriptor b_j in registers; 64 a-rows per block are
// staged in LDS and broadcast; uint16 outputs are written row-coalesced.
#define HM_ROWS 64
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t *__restrict__ A, int na,
                                                        const uint8_t *__restrict__ Bm, int nb,
                                                        uint16_t *__restrict__ out) {
    __shared__ u64 sa[HM_ROWS * 4];
    const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * HM_ROWS;
    const int rows = min(HM_ROWS, na - i0);
    if (threadIdx.x < rows * 4) sa[threadIdx.x] = ((const u64 *)(A + (size_t)i0 * 32))[threadIdx.x];
    __syncthreads();
    if (j >= nb) return;
    const Desc256 b = load_desc(Bm + (size_t)j * 32);
    for (int i = 0; i < rows; i++) {
        const int d = __popcll(sa[4 * i] ^ b.w[0]) + __popcll(sa[4 * i + 1] ^ b.w[1]) +
                      __popcll(sa[4 * i + 2] ^ b.w[2]) + __popcll(sa[4 * i + 3] ^ b.w[3]);
        out[(size_t)(i0 + i) * nb + j] = (uint16_t)d;
    }
}

extern "C" int orbm_hamming_matrix_device(const uint8_t *d_a, int na, const uint8_t *d_b, int nb,
                                          uint16_t *d_out, void *stream) {
    if (!d_a || !d_b || !d_out || na < 1 || nb < 1) { orbx_set_error("orbm_hamming_matrix_device: bad arguments"); return ORBX_ERR_ARG; }
    dim3 grid((nb + 255) / 256, (na + HM_ROWS - 1) / HM_ROWS);
    (void)hipGetLastError();  // drop stale errors of other HIP users in this process
    hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(256), 0, (hipStream_t)stream, d_a, na, d_b, nb, d_out);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
// one fixed-capacity record per frame for the result all-gather (layout: orbx.h); a thread moves one dword
__global__ __launch_bounds__(256) void k_pack_records(const uint32_t *__restrict__ kps, const uint32_t *__restrict__ desc,
                                                      const uint32_t *__restrict__ ur, const uint32_t *__restrict__ dp,
                                                      const int32_t *__restrict__ counts, int cap, int recDwords,
                                                      uint32_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= recDwords) return;
    const size_t fb = (size_t)b * cap;
    uint32_t v = 0;
    if (i < 7 * cap) v = kps[fb * 7 + i];
    else if (i < 15 * cap) v = desc[fb * 8 + (i - 7 * cap)];
    else if (i < 16 * cap) v = ur ? ur[fb + (i - 15 * cap)] : 0u;
    else if (i < 17 * cap) v = dp ? dp[fb + (i - 16 * cap)] : 0u;
    else if (i == 17 * cap) v = (uint32_t)counts[b];
    out[(size_t)b * recDwords + i] = v;
}
extern "C" int orbx_record_bytes(int cap) { return cap < 1 ? ORBX_ERR_ARG : 68 * cap + 16; }
extern "C" int orbx_pack_records_device(const orbx_keypoint_t *d_kps, const uint8_t *d_desc, const float *d_uright,
                                        const float *d_depth, const int32_t *d_counts, int B, int cap, uint8_t *d_records,
                                        void *stream) {
    if (!d_kps || !d_desc || !d_counts || !d_records || B < 1 || cap < 1) { orbx_set_error("orbx_pack_records_device: bad arguments"); return ORBX_ERR_ARG; }
    const int recDwords = 17 * cap + 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_pack_records, dim3((recDwords + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_kps,
                       (const uint32_t *)d_desc, (const uint32_t *)d_uright, (const uint32_t *)d_depth, d_counts, cap, recDwords,
                       (uint32_t *)d_records);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:481-655)
struct StereoLevels {
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], pstrideL[ORBX_MAX_LEVELS], pstrideR[ORBX_MAX_LEVELS];
    unsigned long long poffL[ORBX_MAX_LEVELS], poffR[ORBX_MAX_LEVELS];
    float sf[ORBX_MAX_LEVELS], isf[ORBX_MAX_LEVELS];
    int nlevels, nRows;
};

#define ST_WAVES 8
#define ST_CAND 256   // candidates of one left keypoint kept in LDS before the Hamming step
// right keypoints in the compact form the candidate loop needs (8 B, coalesced):
// x = minr | maxr << 12 | octave << 24 (row band floor(y-r)..ceil(y+r), r = 2*scale, :498-508), y = bits of pt.x
__device__ __forceinline__ uint2 stereo_right_record(const StereoLevels &lv, const orbx_keypoint_t &kp) {
    const float r = 2.0f * lv.sf[kp.octave];
    int maxr = (int)ceilf(kp.y + r), minr = (int)floorf(kp.y - r);
    minr = max(minr, 0);
    maxr = min(maxr, 4095);
    uint2 o;
    o.x = maxr < minr ? 0xFFFu : ((uint32_t)minr | ((uint32_t)maxr << 12) | ((uint32_t)kp.octave << 24));  // empty band: min > max
    o.y = __float_as_uint(kp.x);
    return o;
}
// Row bins for the candidate search (the reference's vRowIndices, :495-508, at a coarser grain): bins of BH rows, BH a power of
// two chosen so that the tallest row band touches at most ST_MAX_SPAN bins (BH = 4 rows for the usual 8 levels x 1.2: a band is
// <= 18 rows); a right keypoint is entered in every bin its band touches and a left keypoint only looks at the right keypoints of
// ITS bin.  Round 4: rounds 1-3 used BH >= the tallest band (32 rows: at most two bins per keypoint, ~125 candidates per left
// keypoint at 1000 features, ~250 at 2000 - two to four rounds of the 64-lane candidate test); with 4-row bins a left keypoint tests
// ~35 / ~65 entries: one round (k_stereo_match 44 -> see DESIGN.md section 6).
#define ST_MAX_BINS 512
#define ST_MAX_SPAN 6     // bin entries per right keypoint at most (the item array holds ST_MAX_SPAN * cap entries per frame)
#define SB_T 1024   // (round 5: 256 threads walked a frame's ~2000 right keypoints in eight dependent rounds of global loads - 11 us for one frame)
__global__ __launch_bounds__(SB_T) void k_stereo_bins(StereoLevels lv, const orbx_keypoint_t *__restrict__ kr,
                                                     uint2 *__restrict__ rc, const int32_t *__restrict__ nr, int cap,
                                                     int bhShift, int nbins, int32_t *__restrict__ binStart,
                                                     uint4 *__restrict__ items) {
    __shared__ int cnt[ST_MAX_BINS + 1], fill[ST_MAX_BINS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Nr = min(nr[b], cap);
    uint2 *rcb = rc + (size_t)b * cap;
    for (int i = tid; i <= nbins; i += SB_T) cnt[i] = 0;
    for (int i = tid; i < nbins; i += SB_T) fill[i] = 0;
    __syncthreads();
    for (int i = tid; i < Nr; i += SB_T) {   // the compact records are made here (each thread re-reads only its own below)
        const uint2 rec = stereo_right_record(lv, kr[(size_t)b * cap + i]);
        rcb[i] = rec;
        const uint32_t x = rec.x;
        const int minr = (int)(x & 0xFFF), maxr = (int)((x >> 12) & 0xFFF);
        if (maxr < minr) continue;
        const int b0 = min(minr >> bhShift, nbins - 1), b1 = min(maxr >> bhShift, nbins - 1);
        for (int bb = b0; bb <= b1; bb++) atomicAdd(&cnt[bb + 1], 1);   // <= ST_MAX_SPAN bins (the host chooses bhShift accordingly)
    }
    __syncthreads();
    // inclusive scan of the bin counts by ONE wave, eight bins per lane (round 5: thread 0 walking the <= 512 bins alone was a chain
    // of dependent LDS round trips - ~10 of this kernel's 12 us for one frame)
    if (tid < 64) {
        int v[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = 8 * tid + k; v[k] = i < nbins ? cnt[i + 1] : 0; s += v[k]; }
        int run = wave_incl_scan_dpp(s) - s;
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = 8 * tid + k; run += v[k]; if (i < nbins) cnt[i + 1] = run; }
    }
    __syncthreads();
    int32_t *bs = binStart + (size_t)b * (ST_MAX_BINS + 1);
    uint4 *it = items + (size_t)b * ST_MAX_SPAN * cap;   // a bin entry is the whole record + the keypoint index: ONE load per candidate
    for (int i = tid; i <= nbins; i += SB_T) bs[i] = cnt[i];
    for (int i = tid; i < Nr; i += SB_T) {
        const uint2 rec = rcb[i];
        const uint32_t x = rec.x;
        const int minr = (int)(x & 0xFFF), maxr = (int)((x >> 12) & 0xFFF);
        if (maxr < minr) continue;
        const int b0 = min(minr >> bhShift, nbins - 1), b1 = min(maxr >> bhShift, nbins - 1);
        const uint4 e = make_uint4(rec.x, rec.y, (uint32_t)i, 0u);
        for (int bb = b0; bb <= b1; bb++) it[cnt[bb] + atomicAdd(&fill[bb], 1)] = e;
    }
}
// one wave per left keypoint: row-band candidate test (:498-508, :535), level and
// disparity-range tests (:548-553), Hamming argmin (first minimum in iR order, :558-562),
// then the 11x11 SAD over 11 shifts (:577-607), parabola (:613-620), disparity (:623-636).
__global__ __launch_bounds__(64 * ST_WAVES) void k_stereo_match(
    StereoLevels lv, const uint8_t *__restrict__ pyrL, size_t pyrImgL, const uint8_t *__restrict__ pyrR,
    size_t pyrImgR, const orbx_keypoint_t *__restrict__ kl, const uint8_t *__restrict__ dl,
    const int32_t *__restrict__ nl, const orbx_keypoint_t *__restrict__ kr, const uint8_t *__restrict__ dr,
    const int32_t *__restrict__ nr, int cap, float mbf, float mb, float *__restrict__ uright,
    float *__restrict__ depth, int32_t *__restrict__ sad, const uint2 *__restrict__ rc, const int32_t *__restrict__ binStart,
    const uint4 *__restrict__ binItems, int bhShift, int nbins) {
    // Candidates = the right keypoints of this keypoint's row bin; the survivors of the exact tests are compacted
    // (ballot prefix) so that their descriptors are fetched by all lanes at once.
    __shared__ uint16_t s_cand[ST_WAVES][ST_CAND];
    __shared__ float s_candU[ST_WAVES][ST_CAND];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int bx, b;
    xcd_block_map(bx, b);   // both pyramids of a frame are read through ONE L2
    const int iL = bx * ST_WAVES + wave;
    if (iL >= cap) return;
    const size_t o = (size_t)b * cap + iL;
    const orbx_keypoint_t kpL = kl[o];      // requested together with the count (slot o exists even when iL >= N)
    const int N = min(nl[b], cap);
    if (iL >= N) {
        if (lane == 0) { uright[o] = -1.0f; depth[o] = -1.0f; sad[o] = -1; }
        return;
    }
    float out_u = -1.0f, out_d = -1.0f;
    int out_s = -1;
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    const float minU = uL - maxD, maxU = uL - minD;
    bool ok = row >= 0 && row < lv.nRows && !(maxU < 0);
    if (ok) {
        const Desc256 dL = load_desc(dl + o * 32);
        u64 best = ~0ull;
        const uint8_t *drb = dr + (size_t)b * cap * 32;
        uint16_t *cl = s_cand[wave];
        float *cu = s_candU[wave];
        int cnt = 0;
        float bestU = 0.f;   // pt.x of this lane's best candidate (the winner's is broadcast after the argmin)
        const int32_t *bs = binStart + (size_t)b * (ST_MAX_BINS + 1);
        const uint4 *bit = binItems + (size_t)b * ST_MAX_SPAN * cap;
        const int bin = min(row >> bhShift, nbins - 1);
        const int p0 = bs[bin], p1 = bs[bin + 1];
        // two bin chunks per round, their entries requested together and unconditionally from clamped slots (a load under
        // `if (ip < p1)` sits behind a divergent branch: one memory latency per 64 entries)
        for (int i0 = p0; i0 < p1; i0 += 128) {
          uint4 q2[2];
#pragma unroll
          for (int hh = 0; hh < 2; hh++) q2[hh] = bit[min(i0 + 64 * hh + lane, p1 - 1)];
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const int ip = i0 + 64 * hh + lane;
            const uint4 q = q2[hh];
            const int iR = (int)q.z;
            const int minr = (int)(q.x & 0xFFF), maxr = (int)((q.x >> 12) & 0xFFF), octR = (int)(q.x >> 24);
            const float uR = __uint_as_float(q.y);
            const bool pass = ip < p1 && row >= minr && row <= maxr && octR >= levelL - 1 && octR <= levelL + 1 && uR >= minU && uR <= maxU;
            const u64 m = __ballot(pass);
            if (m) {
                if (cnt + __popcll(m) > ST_CAND) {   // list full (wave-uniform): score what is staged, start over
                    wave_sync();
                    for (int c = lane; c < cnt; c += 64) {
                        const int jR = cl[c];
                        const int dist = ham(dL, load_desc(drb + (size_t)jR * 32));
                        if (dist < TH_HIGH) { const u64 key = ((u64)dist << 32) | (unsigned)jR; if (key < best) { best = key; bestU = cu[c]; } }
                    }
                    wave_sync();
                    cnt = 0;
                }
                if (pass) {
                    const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    cl[pos] = (uint16_t)iR;
                    cu[pos] = uR;
                }
                cnt += __popcll(m);
            }
          }
        }
        wave_sync();
        for (int c = lane; c < cnt; c += 64) {
            const int jR = cl[c];
            const int dist = ham(dL, load_desc(drb + (size_t)jR * 32));
            if (dist < TH_HIGH) { const u64 key = ((u64)dist << 32) | (unsigned)jR; if (key < best) { best = key; bestU = cu[c]; } }
        }
        const u64 mine = best;
        best = wave_min_u64(best);
        const int bestDist = best == ~0ull ? TH_HIGH : (int)(best >> 32);
        const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
        if (bestDist < thOrbDist) {
            // pt.x of the winner from the lane that holds it (keys are unique: the index is part of the key)
            const float uR0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(bestU), (int)__builtin_ctzll(__ballot(mine == best))));
            const float scaleFactor = lv.isf[levelL];
            const float scaleduL = roundf(kpL.x * scaleFactor);
            const float scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (!(iniu < 0 || endu >= lv.w[levelL])) {
                const uint8_t *IL = pyrL + (size_t)b * pyrImgL + lv.poffL[levelL] +
                                    (size_t)ORBX_EDGE * lv.pstrideL[levelL] + ORBX_EDGE;
                const uint8_t *IR = pyrR + (size_t)b * pyrImgR + lv.poffR[levelL] +
                                    (size_t)ORBX_EDGE * lv.pstrideR[levelL] + ORBX_EDGE;
                const int sL = lv.pstrideL[levelL], sR = lv.pstrideR[levelL];
                const int cy = (int)scaledvL, cxl = (int)scaleduL, cxr0 = (int)scaleduR0;
                // The 11 x 11 window against 11 shifts (:577-607).  Round 4: lane = (DPP row g, window row r): the lane loads its row of the
                // left window once (11 bytes: three dwords from the row's own byte address) and 19 bytes of the right row that cover the
                // windows of the shifts g, g + 4, g + 8 - a shift of four pixels is the next dword, so the three windows are the same byte
                // selectors on neighbouring registers.  The reference subtracts the window centres and takes the L1 norm of the difference:
                // |(a - cL) - (b - cR)|; with + 256 on both sides the terms are u16, a pixel pair is one packed value and
                // v_sad_u16 sums |x0 - y0| + |x1 - y1| + acc in ONE instruction: 18 instructions per (row, shift) for 11 pixels, the rows
                // of a shift summed by four DPP additions.  (Rounds 1-3: two window pixels per lane and a wave-wide sum per shift: 242
                // VALU instructions and 24 byte loads per keypoint; now ~100 and 8 dword loads.)
                typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
                const int g = lane >> 4, rw = min(lane & 15, 10);
                const bool rowLane = (lane & 15) < 11;
                const u32_unaligned *pl = (const u32_unaligned *)(IL + (size_t)(cy + rw - w) * sL + cxl - w);
                const u32_unaligned *pr = (const u32_unaligned *)(IR + (size_t)(cy + rw - w) * sR + cxr0 - L - w + g);
                const uint32_t l0 = pl[0], l1 = pl[1], l2 = pl[2];
                uint32_t q[5];
#pragma unroll
                for (int k = 0; k < 5; k++) q[k] = pr[k];
                // window centres: the left one is byte 5 of row 5, the right one of shift g + 4k byte 4k + 5 of row 5 - held by lane 16g + 5
                const uint32_t cL = ((uint32_t)__builtin_amdgcn_readlane((int)l1, 5) >> 8) & 0xFFu;
                const int src5 = ((lane & 48) | 5) << 2;
                uint32_t biasR[3];
#pragma unroll
                for (int k = 0; k < 3; k++)
                    biasR[k] = (256u - (((uint32_t)__builtin_amdgcn_ds_bpermute(src5, (int)q[k + 1]) >> 8) & 0xFFu)) * 0x00010001u;
                const uint32_t biasL = (256u - cL) * 0x00010001u;
                constexpr uint32_t S01 = 0x0c010c00u, S23 = 0x0c030c02u, S2_ = 0x0c0c0c02u;   // bytes (0,1) / (2,3) / (2, none) of a dword as two u16
                uint32_t Lp[6];
                Lp[0] = __builtin_amdgcn_perm(0u, l0, S01) + biasL; Lp[1] = __builtin_amdgcn_perm(0u, l0, S23) + biasL;
                Lp[2] = __builtin_amdgcn_perm(0u, l1, S01) + biasL; Lp[3] = __builtin_amdgcn_perm(0u, l1, S23) + biasL;
                Lp[4] = __builtin_amdgcn_perm(0u, l2, S01) + biasL; Lp[5] = __builtin_amdgcn_perm(0u, l2, S2_) + (biasL & 0xFFFFu);
                uint32_t acc[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t bR = biasR[k];
                    uint32_t a_ = __builtin_amdgcn_sad_u16(Lp[0], __builtin_amdgcn_perm(0u, q[k], S01) + bR, 0u);
                    a_ = __builtin_amdgcn_sad_u16(Lp[1], __builtin_amdgcn_perm(0u, q[k], S23) + bR, a_);
                    a_ = __builtin_amdgcn_sad_u16(Lp[2], __builtin_amdgcn_perm(0u, q[k + 1], S01) + bR, a_);
                    a_ = __builtin_amdgcn_sad_u16(Lp[3], __builtin_amdgcn_perm(0u, q[k + 1], S23) + bR, a_);
                    a_ = __builtin_amdgcn_sad_u16(Lp[4], __builtin_amdgcn_perm(0u, q[k + 2], S01) + bR, a_);
                    a_ = __builtin_amdgcn_sad_u16(Lp[5], __builtin_amdgcn_perm(0u, q[k + 2], S2_) + (bR & 0xFFFFu), a_);
                    a_ = rowLane ? a_ : 0u;
                    // sum over the window rows = the 16 lanes of the DPP row (lanes 11..15 hold 0): the total ends up in lane 16g + 15
                    a_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_, 0x111, 0xf, 0xf, true);
                    a_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_, 0x112, 0xf, 0xf, true);
                    a_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_, 0x114, 0xf, 0xf, true);
                    a_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_, 0x118, 0xf, 0xf, true);
                    acc[k] = a_;
                }
                float vDists[11];
                int bestDistS = INT_MAX, bestincR = 0;
#pragma unroll
                for (int incR = -L; incR <= L; incR++) {   // shift incR + 5 = g + 4k: DPP row g, accumulator k
                    const int sft = incR + L;
                    const int sv = __builtin_amdgcn_readlane((int)acc[sft >> 2], 16 * (sft & 3) + 15);
                    const float dist = (float)sv;
                    if (dist < (float)bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                    vDists[L + incR] = dist;
                }
                if (!(bestincR == -L || bestincR == L)) {
                    float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                    for (int k = 1; k < 10; k++)
                        if (k == L + bestincR) { dist1 = vDists[k - 1]; dist2 = vDists[k]; dist3 = vDists[k + 1]; }
                    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = lv.sf[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
                        float disparity = (uL - bestuR);
                        if (disparity >= minD && disparity < maxD) {
                            if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
                            out_d = mbf / disparity;
                            out_u = bestuR;
                            out_s = bestDistS;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) { uright[o] = out_u; depth[o] = out_d; sad[o] = out_s; }
}

// median of the SAD distances of the accepted matches (sort + vDistIdx[size/2], :641-642) by a two-level radix select -> the
// threshold 1.5 * 1.4 * median of :643.  Block-wide (SM_T threads, every one calls it); returns the number of accepted matches
// (0: no threshold).  sd_lds: [N] when useLds (filled here), else the global array is re-read.
#define SM_T 1024   // k_stereo_median (round 5: 256 -> 1024 threads, a frame's SAD values in two rounds of loads instead of eight)
#define SF_T 1024   // k_stereo_finish (256-thread workgroups measured slower: 13.9 against 11.4 us)
template <int T>
__device__ __forceinline__ int stereo_sad_threshold(const int32_t *__restrict__ sadg, int N, int32_t *sd_lds, int useLds, float *thDist) {
    __shared__ int sh_nd, hist[256], sh_hi, sh_rank, sh_lo, sh_rest;
    const int tid = threadIdx.x;
    if (tid == 0) sh_nd = 0;
    __syncthreads();
    const int32_t *sd = useLds ? sd_lds : sadg;
    int c = 0;
    for (int i = tid; i < N; i += T) {
        const int s = sadg[i];
        if (useLds) sd_lds[i] = s;
        c += s >= 0;
    }
    if (c) atomicAdd(&sh_nd, c);
    __syncthreads();
    const int nd = sh_nd;
    if (nd == 0) return 0;   // reference: UB on the empty vector (:642); defined here as "no matches"
    // element of rank nd/2 of the sorted values (< 2^16: 121*510 max): (high byte, low byte) histograms
    const int target = nd / 2;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += T) { const int s = sd[i]; if (s >= 0) atomicAdd(&hist[(s >> 8) & 0xFF], 1); }
    __syncthreads();
    if (tid < 64) hist256_select(hist, target, tid, &sh_hi, &sh_rank);   // (a serial 256-step scan by one thread cost 5 us)
    __syncthreads();
    const int hi8 = sh_hi;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += T) { const int s = sd[i]; if (s >= 0 && ((s >> 8) & 0xFF) == hi8) atomicAdd(&hist[s & 0xFF], 1); }
    __syncthreads();
    if (tid < 64) hist256_select(hist, sh_rank, tid, &sh_lo, &sh_rest);
    __syncthreads();
    const float median = (float)((hi8 << 8) | sh_lo);
    *thDist = 1.5f * 1.4f * median;
    return nd;
}
// one workgroup per frame: drop every match with dist >= 1.5*1.4*median (:643-654)
__global__ __launch_bounds__(SM_T) void k_stereo_median(const int32_t *__restrict__ nl, int cap,
                                                        float *__restrict__ uright, float *__restrict__ depth,
                                                        const int32_t *__restrict__ sad,
                                                        int32_t *__restrict__ nmatch, int useLds) {
    extern __shared__ int32_t sd_lds[];  // [cap] sad (or -1) when useLds; larger frames re-read the global array
    __shared__ int sh_keep;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = min(nl[b], cap);
    const size_t o = (size_t)b * cap;
    if (tid == 0) sh_keep = 0;
    float thDist = 0.0f;
    const int nd = stereo_sad_threshold<SM_T>(sad + o, N, sd_lds, useLds, &thDist);
    if (nd == 0) {
        if (nmatch && tid == 0) nmatch[b] = 0;
        return;
    }
    const int32_t *sd = useLds ? sd_lds : sad + o;
    int keep = 0;
    for (int i = tid; i < N; i += SM_T) {
        const int s = sd[i];
        if (s < 0) continue;
        if ((float)s < thDist) keep++;
        else { uright[o + i] = -1; depth[o + i] = -1; }
    }
    if (keep) atomicAdd(&sh_keep, keep);
    __syncthreads();
    if (nmatch && tid == 0) nmatch[b] = sh_keep;
}

// ONE stereo frame, last launch of the latency path (orbx_stereo_frame_view): the median step above AND the frame's whole result
// record - keypoints and descriptors of both images, mvuRight, mvDepth, counts - written to PINNED HOST memory by the kernel
// itself.  Every workgroup finds the threshold for itself (2000 values: a few hundred nanoseconds; no hand-off between
// workgroups), then moves its slice of the record, 16 bytes per lane, applying the :643-654 rule to the mvuRight / mvDepth words on
// the way; workgroup 0 also applies it to the device arrays (the guided searches that follow read mvuRight in HBM) and writes the
// match count.  Replaces k_stereo_median + k_pack_records + a device-to-host copy command and the ~10-us hand-over between the
// compute queue and the copy engine (one frame of the tracking chain: 10.5 + 4.2 + 10 + 7 us).
// Record (bytes; cap a multiple of 4): kps L @0 | kps R @28 cap | desc L @56 cap | desc R @88 cap | mvuRight @120 cap | mvDepth @124 cap
// | @128 cap: int32 nl, nr, nmatch, 0.
__global__ __launch_bounds__(SF_T) void k_stereo_finish(uint4 *rec, uint4 *__restrict__ rec_host, int cap,
                                                        const int32_t *__restrict__ sad, int useLds, int uBegin, int32_t *doneFlag, int doneSeq) {
    extern __shared__ int32_t sd_lds[];
    __shared__ int sh_keep;
    const int tid = threadIdx.x;
    int32_t *tail = (int32_t *)((uint8_t *)rec + (size_t)128 * cap);
    float *uright = (float *)((uint8_t *)rec + (size_t)120 * cap), *depth = uright + cap;
    // my unit of the record is requested first: nothing below depends on it until the copy-out (k_stereo_match wrote sad = -1 for
    // every slot behind the last keypoint, so the median step walks all cap slots and needs no count either: no dependent load chain)
    const int total = 8 * cap + 1, ur0 = (120 * cap) >> 4, ur1 = (128 * cap) >> 4;
    const int u0 = uBegin + blockIdx.x * SF_T + tid;   // uBegin = first unit this launch moves (k_describe may have stored the keypoint / descriptor blocks already)
    uint4 v0 = make_uint4(0u, 0u, 0u, 0u);
    if (u0 < total) v0 = rec[u0];
    if (tid == 0) sh_keep = 0;
    float thDist = 0.0f;
    const int N = cap;
    const int nd = stereo_sad_threshold<SF_T>(sad, N, sd_lds, useLds, &thDist);
    const int32_t *sd = useLds ? sd_lds : sad;
    {   // the match count (every workgroup: the one that copies the record's tail needs it) and, by workgroup 0, the rule on the device arrays
        int keep = 0;
        if (nd > 0)
            for (int i = tid; i < N; i += SF_T) {
                const int s = sd[i];
                if (s < 0) continue;
                if ((float)s < thDist) keep++;
                else if (blockIdx.x == 0) { uright[i] = -1; depth[i] = -1; }
            }
        if (keep) atomicAdd(&sh_keep, keep);
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0) tail[2] = sh_keep;
    }
    // copy-out: 16-byte units [0, 8 cap + 1); units of the mvuRight / mvDepth block get the rule applied (the device words may or may
    // not have been rewritten by workgroup 0 yet: the rule gives the same word either way)
    for (int u = u0; u < total; u += gridDim.x * SF_T) {
        uint4 v = u == u0 ? v0 : rec[u];
        if (u >= ur0 && u < ur1) {
            const int i0 = ((u - ur0) * 4) % cap;     // keypoint of the unit's first word (mvuRight and mvDepth blocks: cap words each)
            uint32_t *w = (uint32_t *)&v;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k;
                if (nd > 0 && i < N) { const int s = sd[i]; if (s >= 0 && !((float)s < thDist)) w[k] = 0xBF800000u; }   // -1.0f
            }
        } else if (u == total - 1)
            v.z = (uint32_t)sh_keep;
        rec_host[u] = v;
    }
    // one workgroup moves the whole tail (the usual latency call): the call's sequence number goes into the handle's completion word behind
    // everything, and the host polls that word instead of waiting for the stream (see arena_wait, orbx_match_fast.hip)
    if (doneFlag && gridDim.x == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores to the (uncached) host record have been acknowledged; ONE wave fences and signals
        __syncthreads();
        if (tid == 0) { __threadfence_system(); __hip_atomic_store(doneFlag, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

static int fill_stereo_levels(orbx_extractor *hl, orbx_extractor *hr, StereoLevels *lv) {
    if (hl->nlevels != hr->nlevels || hl->pw != hr->pw || hl->ph != hr->ph || hl->pw == 0) {
        orbx_set_error("stereo: left/right extractors must have extracted images of the same size and levels");
        return ORBX_ERR_ARG;
    }
    memset(lv, 0, sizeof(*lv));
    lv->nlevels = hl->nlevels;
    for (int l = 0; l < hl->nlevels; l++) {
        lv->w[l] = hl->geom[l].w; lv->h[l] = hl->geom[l].h;
        lv->pstrideL[l] = hl->geom[l].pstride; lv->pstrideR[l] = hr->geom[l].pstride;
        lv->poffL[l] = hl->geom[l].poff; lv->poffR[l] = hr->geom[l].poff;
        lv->sf[l] = hl->sf[l]; lv->isf[l] = hl->isf[l];
    }
    lv->nRows = hl->geom[0].h;
    return ORBX_OK;
}

// Scratch of one ComputeStereoMatches call (SAD distances, compact right-keypoint records, row bins): it belongs to the
// LEFT extractor handle (orbx_extractor::st_*), like that handle's pyramid and candidate buffers, so two handles driven
// on two streams - from one host thread or several - never share it.  Calls on ONE handle must be stream-ordered, the
// same rule as for its extraction buffers.
static int stereo_scratch_reserve(orbx_extractor *hl, int B, int cap) {
    const size_t need = (size_t)B * cap;
    if (hl->st_n >= need && hl->st_nB >= B) return ORBX_OK;
    if (hl->last_valid) ORBX_HIP(hipStreamSynchronize(hl->last_stream));
    if (hl->st_stream) ORBX_HIP(hipStreamSynchronize(hl->st_stream));
    hipFree(hl->st_sad); hipFree(hl->st_rc); hipFree(hl->st_binStart); hipFree(hl->st_items);
    hl->st_sad = nullptr; hl->st_rc = nullptr; hl->st_binStart = nullptr; hl->st_items = nullptr; hl->st_n = 0; hl->st_nB = 0;
    ORBX_HIP(hipMalloc(&hl->st_sad, sizeof(int32_t) * need));
    ORBX_HIP(hipMalloc(&hl->st_rc, sizeof(uint2) * need));
    ORBX_HIP(hipMalloc(&hl->st_items, sizeof(uint4) * ST_MAX_SPAN * need));
    ORBX_HIP(hipMalloc(&hl->st_binStart, sizeof(int32_t) * (ST_MAX_BINS + 1) * (size_t)B));
    hl->st_n = need; hl->st_nB = B;
    return ORBX_OK;
}
void orbx_internal_free_stereo_scratch(orbx_extractor *h) {
    hipFree(h->st_sad); hipFree(h->st_rc); hipFree(h->st_binStart); hipFree(h->st_items);
    h->st_sad = nullptr; h->st_rc = nullptr; h->st_binStart = nullptr; h->st_items = nullptr; h->st_n = 0; h->st_nB = 0;
}

#define SM_LDS_CAP 12288   // k_stereo_median keeps a frame's SAD values in LDS up to this many keypoints (48 KB), beyond: global memory
static int stereo_batch_impl(orbx_extractor_t *hl, orbx_extractor_t *hr, int B, int left_slot0,
                             int right_slot0, const orbx_keypoint_t *d_kl, const uint8_t *d_dl, const int32_t *d_nl,
                             const orbx_keypoint_t *d_kr, const uint8_t *d_dr, const int32_t *d_nr,
                             int cap, float mbf, float mb, float *d_uright, float *d_depth,
                             int32_t *d_nmatch, void *stream, bool prev, uint8_t *finish_rec = nullptr, uint8_t *finish_host = nullptr, bool finish_tail_only = false,
                             int32_t *doneFlag = nullptr, int doneSeq = 0, int *flagArmed = nullptr) {
    if (!hl || !hr || !d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_uright || !d_depth || B < 1 ||
        cap < 1 || left_slot0 < 0 || right_slot0 < 0 || left_slot0 + B > hl->pB || right_slot0 + B > hr->pB) {
        orbx_set_error("orbm_stereo_batch_device: bad arguments");
        return ORBX_ERR_ARG;
    }
    if (cap > 65535) { orbx_set_error("orbm_stereo: %d keypoints per image (limit 65535)", cap); return ORBX_ERR_UNSUPPORTED; }
    if (prev && !(hl->prevPyrValid && hr->prevPyrValid)) {
        orbx_set_error("orbm_stereo_batch_device_prev: the pyramid of the call before the last one is gone (the last extraction call did "
                       "not take a pyramid built ahead, or the next one has been started already)");
        return ORBX_ERR_ARG;
    }
    const uint8_t *pyrL = prev ? hl->d_pyrAlt : hl->d_pyr, *pyrR = prev ? hr->d_pyrAlt : hr->d_pyr;
    StereoLevels lv;
    int rc = fill_stereo_levels(hl, hr, &lv);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(hl->device));
    rc = stereo_scratch_reserve(hl, B, cap);
    if (rc) return rc;
    // bin height: the smallest power of two with which the tallest row band (maxr - minr + 1 <= 2r + 3 rows, r = 2 * scale of the
    // coarsest level; a band of n rows touches at most ((n - 1) >> shift) + 2 bins) stays within ST_MAX_SPAN bins and the image within
    // ST_MAX_BINS bins
    const int maxBand = (int)(4.0f * lv.sf[lv.nlevels - 1]) + 4;
    int bhShift = 0;
    while (((maxBand - 1) >> bhShift) + 2 > ST_MAX_SPAN) bhShift++;
    while (((lv.nRows + (1 << bhShift) - 1) >> bhShift) > ST_MAX_BINS) bhShift++;
    const int nbins = std::max(1, (lv.nRows + (1 << bhShift) - 1) >> bhShift);
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP default (null) stream
    dim3 grid((cap + ST_WAVES - 1) / ST_WAVES, B);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_stereo_bins, dim3(B), dim3(SB_T), 0, st, lv, d_kr, hl->st_rc, d_nr, cap, bhShift, nbins, hl->st_binStart, hl->st_items);
    hipLaunchKernelGGL(k_stereo_match, grid, dim3(64 * ST_WAVES), 0, st, lv, pyrL + (size_t)left_slot0 * hl->pyrImgBytes, hl->pyrImgBytes,
                       pyrR + (size_t)right_slot0 * hr->pyrImgBytes, hr->pyrImgBytes, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, mbf, mb, d_uright, d_depth,
                       hl->st_sad, hl->st_rc, hl->st_binStart, hl->st_items, bhShift, nbins);
    const int useLds = cap <= SM_LDS_CAP;
    if (finish_rec)   // one frame, latency path: median step + the whole record to pinned host memory, one launch
    {
        const int uBegin = finish_tail_only ? (120 * cap) >> 4 : 0;
        const int fgrid = (8 * cap + 1 - uBegin + SF_T - 1) / SF_T;
        if (flagArmed) *flagArmed = doneFlag != nullptr && fgrid == 1;
        hipLaunchKernelGGL(k_stereo_finish, dim3(fgrid), dim3(SF_T), useLds ? sizeof(int32_t) * cap : 0, st,
                           (uint4 *)finish_rec, (uint4 *)finish_host, cap, hl->st_sad, useLds, uBegin, fgrid == 1 ? doneFlag : (int32_t *)nullptr, doneSeq);
    }
    else
        hipLaunchKernelGGL(k_stereo_median, dim3(B), dim3(SM_T), useLds ? sizeof(int32_t) * cap : 0, st, d_nl, cap, d_uright,
                           d_depth, hl->st_sad, d_nmatch, useLds);
    ORBX_HIP(hipGetLastError());
    hl->st_stream = st;
    return ORBX_OK;
}

extern "C" int orbm_stereo_batch_device(orbx_extractor_t *hl, orbx_extractor_t *hr, int B, int left_slot0,
                                        int right_slot0, const orbx_keypoint_t *d_kl, const uint8_t *d_dl, const int32_t *d_nl,
                                        const orbx_keypoint_t *d_kr, const uint8_t *d_dr, const int32_t *d_nr,
                                        int cap, float mbf, float mb, float *d_uright, float *d_depth,
                                        int32_t *d_nmatch, void *stream) {
    return stereo_batch_impl(hl, hr, B, left_slot0, right_slot0, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, mbf, mb, d_uright, d_depth,
                             d_nmatch, stream, false);
}

int orbx_internal_stereo_frame_record(orbx_extractor *h, uint8_t *d_rec, uint8_t *rec_hostdev, int cap, float mbf, float mb, hipStream_t st, bool recordsOnHost,
                                      int32_t *doneFlag, int doneSeq, int *flagArmed) {
    const size_t c = (size_t)cap;
    orbx_keypoint_t *kl = (orbx_keypoint_t *)d_rec, *kr = (orbx_keypoint_t *)(d_rec + 28 * c);
    uint8_t *dl = d_rec + 56 * c, *dr = d_rec + 88 * c;
    float *ur = (float *)(d_rec + 120 * c), *dp = (float *)(d_rec + 124 * c);
    int32_t *tail = (int32_t *)(d_rec + 128 * c);
    return stereo_batch_impl(h, h, 1, 0, 1, kl, dl, tail, kr, dr, tail + 1, cap, mbf, mb, ur, dp, tail + 2, (void *)st, false, d_rec, rec_hostdev, recordsOnHost, doneFlag, doneSeq, flagArmed);
}

// The same on the pyramids of the extraction call BEFORE the last one (software pipelining: the matcher of batch i-1 issued
// after the extraction of batch i, see include/orbx.h).
extern "C" int orbm_stereo_batch_device_prev(orbx_extractor_t *hl, orbx_extractor_t *hr, int B, int left_slot0,
                                             int right_slot0, const orbx_keypoint_t *d_kl, const uint8_t *d_dl, const int32_t *d_nl,
                                             const orbx_keypoint_t *d_kr, const uint8_t *d_dr, const int32_t *d_nr,
                                             int cap, float mbf, float mb, float *d_uright, float *d_depth,
                                             int32_t *d_nmatch, void *stream) {
    return stereo_batch_impl(hl, hr, B, left_slot0, right_slot0, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, mbf, mb, d_uright, d_depth,
                             d_nmatch, stream, true);
}

// thread-local device scratch with a pinned host mirror of the same layout and a non-blocking stream of its own: the
// host-array entry points below move their inputs with ONE upload and their results with ONE download + ONE
// synchronisation of THAT stream (no hipMalloc per call, no device-wide synchronisation: other threads' extractors and
// matchers keep running, as the reference's Tracking / LocalMapping / LoopClosing threads call matchers concurrently,
// src/LocalMapping.cc:223, src/LoopClosing.cc:249)
struct StagePair { uint8_t *d = nullptr, *h = nullptr; size_t cap = 0; int device = -1; hipStream_t st = nullptr; };
static thread_local StagePair g_sp;
static void stage_release() {
    if (g_sp.device < 0) return;
    hipSetDevice(g_sp.device);
    if (g_sp.st) { hipStreamSynchronize(g_sp.st); hipStreamDestroy(g_sp.st); g_sp.st = nullptr; }
    if (g_sp.d) hipFree(g_sp.d);
    if (g_sp.h) hipHostFree(g_sp.h);
    g_sp.d = nullptr; g_sp.h = nullptr; g_sp.cap = 0; g_sp.device = -1;
}
static int stage_reserve(int device, size_t need) {
    if (g_sp.device == device && g_sp.cap >= need) return ORBX_OK;
    stage_release();
    ORBX_HIP(hipSetDevice(device));
    const size_t cap = std::max(need * 2, (size_t)1 << 20);
    ORBX_HIP(hipMalloc(&g_sp.d, cap));
    ORBX_HIP(hipHostMalloc((void **)&g_sp.h, cap, hipHostMallocDefault));
    ORBX_HIP(hipStreamCreateWithFlags(&g_sp.st, hipStreamNonBlocking));
    g_sp.cap = cap; g_sp.device = device;
    return ORBX_OK;
}
// layout of one call inside the StagePair: take() hands out 256-B aligned offsets; everything taken before mark_inputs()
// is uploaded in one copy
struct StagePlan {
    size_t off = 0, in_end = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
    void mark_inputs() { in_end = off; }
};
template <typename T> static inline T *stage_dev(size_t o) { return (T *)(g_sp.d + o); }
static inline void stage_put(size_t o, const void *src, size_t bytes) { if (bytes) memcpy(g_sp.h + o, src, bytes); }
void orbx_internal_release_match_scratch() { stage_release(); }

extern "C" int orbm_stereo(orbx_extractor_t *hl, orbx_extractor_t *hr, const orbx_keypoint_t *kl,
                           const uint8_t *dl, int nl, const orbx_keypoint_t *kr, const uint8_t *dr, int nr,
                           float mbf, float mb, float *uright, float *depth, int *nmatch) {
    if (!hl || !hr || nl < 0 || nr < 0 || (nl > 0 && (!kl || !dl || !uright || !depth)) || (nr > 0 && (!kr || !dr))) {
        orbx_set_error("orbm_stereo: bad arguments");
        return ORBX_ERR_ARG;
    }
    if (nmatch) *nmatch = 0;
    if (nl == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(hl->device));
    const int cap = std::max(std::max(nl, nr), 1);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_kl = 0, o_dl = o_kl + al(sizeof(orbx_keypoint_t) * cap), o_kr = o_dl + al((size_t)32 * cap),
                 o_dr = o_kr + al(sizeof(orbx_keypoint_t) * cap), o_n = o_dr + al((size_t)32 * cap), o_in_end = o_n + 256,
                 o_u = o_in_end, o_d = o_u + al(sizeof(float) * cap), o_end = o_d + al(sizeof(float) * cap);
    int rc = stage_reserve(hl->device, o_end);
    if (rc) return rc;
    memcpy(g_sp.h + o_kl, kl, sizeof(orbx_keypoint_t) * nl); memcpy(g_sp.h + o_dl, dl, (size_t)32 * nl);
    if (nr) { memcpy(g_sp.h + o_kr, kr, sizeof(orbx_keypoint_t) * nr); memcpy(g_sp.h + o_dr, dr, (size_t)32 * nr); }
    int32_t cnt[3] = {nl, nr, 0};
    memcpy(g_sp.h + o_n, cnt, sizeof(cnt));
    hipStream_t st = hl->stream;
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, o_in_end, hipMemcpyHostToDevice, st));
    int32_t *dn = (int32_t *)(g_sp.d + o_n);
    rc = orbm_stereo_batch_device(hl, hr, 1, 0, 0, (orbx_keypoint_t *)(g_sp.d + o_kl), g_sp.d + o_dl, dn,
                                  (orbx_keypoint_t *)(g_sp.d + o_kr), g_sp.d + o_dr, dn + 1, cap, mbf, mb,
                                  (float *)(g_sp.d + o_u), (float *)(g_sp.d + o_d), dn + 2, st);
    if (rc) return rc;
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_n, g_sp.d + o_n, o_end - o_n, hipMemcpyDeviceToHost, st));   // counts | uright | depth
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(uright, g_sp.h + o_u, sizeof(float) * nl);
    memcpy(depth, g_sp.h + o_d, sizeof(float) * nl);
    if (nmatch) *nmatch = ((const int32_t *)(g_sp.h + o_n))[2];
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------
// guided searches.  Frame::GetFeaturesInArea (src/Frame.cc:342-395) is restated as a
// predicate: keypoint j is returned iff its grid cell (PosInGrid, :397-407) lies inside the
// query's cell rectangle, its octave passes the level test and |dx|<r && |dy|<r; the
// returned ORDER is column-major over cells, insertion (= index) order inside a cell, so the
// scan position of j is the tuple (cellx, celly, j).  Sequential "first strictly smaller
// wins" scans (:102-114, :441-457, :1414-1426) therefore select the minimum of the packed key
//     dist << 28 | cellx << 22 | celly << 16 | j
// and the runner-up is the second smallest key (stable top-2).
#define SQ_T 256
struct Top2 { u64 k1, k2; };
__device__ __forceinline__ void top2_insert(Top2 &t, u64 k) {
    if (k < t.k1) { t.k2 = t.k1; t.k1 = k; }
    else if (k < t.k2) t.k2 = k;
}
__device__ __forceinline__ Top2 top2_merge(const Top2 &a, const Top2 &b) {
    Top2 r;
    r.k1 = a.k1 < b.k1 ? a.k1 : b.k1;
    const u64 hi = a.k1 < b.k1 ? b.k1 : a.k1, lo2 = a.k2 < b.k2 ? a.k2 : b.k2;
    r.k2 = hi < lo2 ? hi : lo2;
    return r;
}
// block-wide top-2 (all threads get the result); sh: u64[2 * SQ_T/64]
__device__ Top2 block_top2(Top2 t, u64 *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Top2 u;
        u.k1 = shfl_xor_u64(t.k1, o);
        u.k2 = shfl_xor_u64(t.k2, o);
        t = top2_merge(t, u);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sh[2 * wave] = t.k1; sh[2 * wave + 1] = t.k2; }
    __syncthreads();
    Top2 r;
    r.k1 = sh[0]; r.k2 = sh[1];
#pragma unroll
    for (int w = 1; w < SQ_T / 64; w++) {
        Top2 u;
        u.k1 = sh[2 * w]; u.k2 = sh[2 * w + 1];
        r = top2_merge(r, u);
    }
    return r;
}

#ifdef ORBX_DEVELOPER
#include "orbx_dev.h"
// ---- test hook: Frame::GetFeaturesInArea (src/Frame.cc:342-395) as the matchers see it - the predicate in_area() and the
// scan-order key (column-major over cells, index order inside a cell).  Every matcher takes "the first minimum in this
// order"; here the order itself comes out: key (cellx, celly, j) of every keypoint the query returns, ~0 for the others,
// sorted by the caller.
__global__ __launch_bounds__(256) void k_debug_area(const orbx_keypoint_t *__restrict__ k, int n, orbm_grid_geom_t g, float x, float y, float r,
                                                    int minLevel, int maxLevel, unsigned long long *__restrict__ keys) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const AreaQuery q = make_query(g, x, y, r, minLevel, maxLevel);
    const unsigned code = cell_code(g, k[j]);
    keys[j] = (!q.empty && in_area(q, code, k[j])) ? scan_key(0, code, j) : ~0ull;
}
extern "C" int orbm_debug_features_in_area(const orbx_keypoint_t *kun, int n, const orbm_grid_geom_t *g, float x, float y, float r,
                                           int min_level, int max_level, int32_t *out_idx, int *n_out, int device) {
    if (!kun || n < 1 || n > 65535 || !g || !out_idx || !n_out) { orbx_set_error("orbm_debug_features_in_area: bad arguments"); return ORBX_ERR_ARG; }
    StagePlan pl;
    const size_t o_k = pl.take(sizeof(orbx_keypoint_t) * n);
    pl.mark_inputs();
    const size_t o_keys = pl.take(8 * (size_t)n);
    int rc = stage_reserve(device, pl.off);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(device));
    hipStream_t st = g_sp.st;
    stage_put(o_k, kun, sizeof(orbx_keypoint_t) * n);
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, pl.in_end, hipMemcpyHostToDevice, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_debug_area, dim3((n + 255) / 256), dim3(256), 0, st, stage_dev<orbx_keypoint_t>(o_k), n, *g, x, y, r, min_level,
                       max_level, stage_dev<unsigned long long>(o_keys));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_keys, g_sp.d + o_keys, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    const unsigned long long *keys = (const unsigned long long *)(g_sp.h + o_keys);
    std::vector<unsigned long long> v;
    for (int j = 0; j < n; j++) if (keys[j] != ~0ull) v.push_back(keys[j]);
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size(); i++) out_idx[i] = (int32_t)(v[i] & 0xFFFF);
    *n_out = (int)v.size();
    return ORBX_OK;
}
#endif   // ORBX_DEVELOPER

// ---- SearchForInitialization: one workgroup, F1 keypoints in order (the steal / gate on
// vMatchedDistance makes iteration i1 depend on all earlier ones)
__global__ __launch_bounds__(SQ_T) void k_search_init(
    const orbx_keypoint_t *__restrict__ k1, const uint8_t *__restrict__ d1, int n1,
    const orbx_keypoint_t *__restrict__ k2, const uint8_t *__restrict__ d2, int n2, orbm_grid_geom_t g,
    float *__restrict__ prev, int32_t *__restrict__ m12, int32_t *__restrict__ m21, int32_t *__restrict__ vmd,
    uint16_t *__restrict__ code2, int32_t *__restrict__ bin1, int window, float nnratio, int check_ori,
    int32_t *__restrict__ nmatches_out) {
    __shared__ u64 sh[2 * SQ_T / 64];
    __shared__ int hn[HISTO_LENGTH];
    __shared__ int sh_nm;
    const int tid = threadIdx.x;
    for (int j = tid; j < n2; j += SQ_T) { code2[j] = (uint16_t)cell_code(g, k2[j]); vmd[j] = INT_MAX; m21[j] = -1; }
    for (int i = tid; i < n1; i += SQ_T) { m12[i] = -1; bin1[i] = -1; }
    if (tid < HISTO_LENGTH) hn[tid] = 0;
    if (tid == 0) sh_nm = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i1 = 0; i1 < n1; i1++) {
        const orbx_keypoint_t kp1 = k1[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        const AreaQuery q = make_query(g, prev[2 * i1], prev[2 * i1 + 1], (float)window, level1, level1);
        if (q.empty) continue;
        const Desc256 da = load_desc(d1 + (size_t)i1 * 32);
        Top2 t;
        t.k1 = t.k2 = ~0ull;
        for (int j = tid; j < n2; j += SQ_T) {
            const unsigned code = code2[j];
            if (!in_area(q, code, k2[j])) continue;
            const int dist = ham(da, load_desc(d2 + (size_t)j * 32));
            if (vmd[j] <= dist) continue;
            top2_insert(t, scan_key(dist, code, j));
        }
        t = block_top2(t, sh);
        if (tid == 0 && t.k1 != ~0ull) {
            const int bestDist = (int)(t.k1 >> 28), bestIdx2 = (int)(t.k1 & 0xFFFF);
            const int bestDist2 = t.k2 == ~0ull ? INT_MAX : (int)(t.k2 >> 28);
            if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { m12[m21[bestIdx2]] = -1; sh_nm--; }
                m12[i1] = bestIdx2; m21[bestIdx2] = i1; vmd[bestIdx2] = bestDist; sh_nm++;
                if (check_ori) {
                    float rot = kp1.angle - k2[bestIdx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    bin1[i1] = bin;
                    hn[bin]++;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    __shared__ int ind[3];
    if (tid == 0 && check_ori) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
    __syncthreads();
    if (check_ori) {
        int dec = 0;
        for (int i = tid; i < n1; i += SQ_T) {
            const int bn = bin1[i];
            if (bn >= 0 && bn != ind[0] && bn != ind[1] && bn != ind[2] && m12[i] >= 0) { m12[i] = -1; dec++; }
        }
        if (dec) atomicSub(&sh_nm, dec);
    }
    __syncthreads();
    for (int i = tid; i < n1; i += SQ_T)
        if (m12[i] >= 0) { prev[2 * i] = k2[m12[i]].x; prev[2 * i + 1] = k2[m12[i]].y; }
    if (tid == 0) *nmatches_out = sh_nm;
}

extern "C" int orbm_search_for_initialization(const orbx_keypoint_t *k1, const uint8_t *d1, int n1,
                                              const orbx_keypoint_t *k2, const uint8_t *d2, int n2,
                                              const orbm_grid_geom_t *g2, float *prev_matched, int32_t *matches12,
                                              int window, float nnratio, int check_orientation, int device,
                                              int *nmatches) {
    if (n1 < 0 || n2 < 0 || !g2 || !nmatches || (n1 > 0 && (!k1 || !d1 || !prev_matched || !matches12)) ||
        (n2 > 0 && (!k2 || !d2)) || n2 > 65535) {
        orbx_set_error("orbm_search_for_initialization: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    if (n1 == 0) return ORBX_OK;
    if (!t_matchExact && n2 > 0) {
        const int frc = fast_search_for_initialization(k1, d1, n1, k2, d2, n2, g2, prev_matched, matches12, window, nnratio,
                                                       check_orientation, device, nmatches);
        if (frc <= 0) return frc;  // done or error; > 0: exact fallback below
    }
    StagePlan pl;
    const size_t o_k1 = pl.take(sizeof(orbx_keypoint_t) * n1), o_d1 = pl.take((size_t)32 * n1), o_k2 = pl.take(sizeof(orbx_keypoint_t) * n2),
                 o_d2 = pl.take((size_t)32 * n2), o_prev = pl.take(sizeof(float) * 2 * n1);
    pl.mark_inputs();
    const size_t o_m12 = pl.take(4 * (size_t)n1), o_nm = pl.take(4), o_m21 = pl.take(4 * (size_t)n2), o_vmd = pl.take(4 * (size_t)n2),
                 o_code = pl.take(2 * (size_t)n2), o_bin = pl.take(4 * (size_t)n1);
    int rc = stage_reserve(device, pl.off);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(device));
    hipStream_t st = g_sp.st;
    stage_put(o_k1, k1, sizeof(orbx_keypoint_t) * n1); stage_put(o_d1, d1, (size_t)32 * n1);
    stage_put(o_k2, k2, sizeof(orbx_keypoint_t) * n2); stage_put(o_d2, d2, (size_t)32 * n2);
    stage_put(o_prev, prev_matched, sizeof(float) * 2 * n1);
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, pl.in_end, hipMemcpyHostToDevice, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_search_init, dim3(1), dim3(SQ_T), 0, st, stage_dev<orbx_keypoint_t>(o_k1), stage_dev<uint8_t>(o_d1), n1,
                       stage_dev<orbx_keypoint_t>(o_k2), stage_dev<uint8_t>(o_d2), n2, *g2, stage_dev<float>(o_prev),
                       stage_dev<int32_t>(o_m12), stage_dev<int32_t>(o_m21), stage_dev<int32_t>(o_vmd), stage_dev<uint16_t>(o_code),
                       stage_dev<int32_t>(o_bin), window, nnratio, check_orientation, stage_dev<int32_t>(o_nm));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_prev, g_sp.d + o_prev, o_m21 - o_prev, hipMemcpyDeviceToHost, st));   // prev | m12 | nm
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(prev_matched, g_sp.h + o_prev, sizeof(float) * 2 * n1);
    memcpy(matches12, g_sp.h + o_m12, 4 * (size_t)n1);
    *nmatches = *(const int32_t *)(g_sp.h + o_nm);
    return ORBX_OK;
}

// ---- SearchByProjection(Frame, MapPoints)
__device__ __forceinline__ bool slot_blocked_mp(const int32_t *holder, const int32_t *ext_obs, int idx,
                                                const orbm_mappoint_t *mps) {
    const int hm = holder[idx];
    if (hm == -1) return false;
    if (hm == -2) return ext_obs && ext_obs[idx] > 0;
    return mps[hm].observations > 0;
}
__global__ __launch_bounds__(SQ_T) void k_search_proj_mp(
    const orbx_keypoint_t *__restrict__ kun, const uint8_t *__restrict__ desc, const float *__restrict__ uright,
    int n, orbm_grid_geom_t g, const float *__restrict__ sf, const orbm_mappoint_t *__restrict__ mps,
    const uint8_t *__restrict__ mp_desc, int m, int32_t *__restrict__ frame_mp, const int32_t *__restrict__ ext_obs,
    uint16_t *__restrict__ code, float th, float nnratio, int32_t *__restrict__ nmatches_out) {
    __shared__ u64 sh[2 * SQ_T / 64];
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += SQ_T) code[j] = (uint16_t)cell_code(g, kun[j]);
    __syncthreads();
    const bool bFactor = th != 1.0;
    int nm = 0;
    for (int iMP = 0; iMP < m; iMP++) {
        const orbm_mappoint_t p = mps[iMP];
        if (!p.in_view) continue;
        const int lvl = p.level;
        float r = p.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:131-137)
        if (bFactor) r *= th;
        const float rs = r * sf[lvl];
        const AreaQuery q = make_query(g, p.proj_x, p.proj_y, rs, lvl - 1, lvl);
        if (q.empty) continue;
        const Desc256 da = load_desc(mp_desc + (size_t)iMP * 32);
        Top2 t;
        t.k1 = t.k2 = ~0ull;
        for (int j = tid; j < n; j += SQ_T) {
            const unsigned c = code[j];
            if (!in_area(q, c, kun[j])) continue;
            if (slot_blocked_mp(frame_mp, ext_obs, j, mps)) continue;
            if (uright[j] > 0) {
                const float er = fabsf(p.proj_xr - uright[j]);
                if (er > rs) continue;
            }
            top2_insert(t, scan_key(ham(da, load_desc(desc + (size_t)j * 32)), c, j));
        }
        t = block_top2(t, sh);
        if (t.k1 != ~0ull) {
            const int bestDist = (int)(t.k1 >> 28), bestIdx = (int)(t.k1 & 0xFFFF);
            const int bestLevel = kun[bestIdx].octave;
            int bestDist2 = 256, bestLevel2 = -1;
            if (t.k2 != ~0ull) { bestDist2 = (int)(t.k2 >> 28); bestLevel2 = kun[(int)(t.k2 & 0xFFFF)].octave; }
            if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)) {
                if (tid == 0) frame_mp[bestIdx] = iMP;
                nm++;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) *nmatches_out = nm;
}

static int exact_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                         const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                         const orbm_mappoint_t *mps, const uint8_t *mp_desc, int m, int32_t *frame_mp,
                                         const int32_t *ext_obs, float th, float nnratio, int device, int *nmatches,
                                         const DevFrame *dev = nullptr);

// MapPoint::PredictScale as a threshold table (see include/orbx.h): host code, uses the host's logf on purpose
static int predict_scale_host(float ratio, float log_sf, int nlevels) {
    int nScale = (int)ceilf(logf(ratio) / log_sf);
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}
extern "C" int orbm_predict_scale_thresholds(float log_scale_factor, int nlevels, float *thresholds) {
    if (!(log_scale_factor > 0.0f) || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || (nlevels > 1 && !thresholds)) {
        orbx_set_error("orbm_predict_scale_thresholds: bad arguments");
        return ORBX_ERR_ARG;
    }
    for (int k = 0; k + 1 < nlevels; k++) {
        // smallest positive float r with predict_scale(r) >= k+1: bisection on the bit pattern (floats > 0 are ordered like ints)
        uint32_t lo = 0x3F800000u /* 1.0f -> level 0 */, hi = 0x7F7FFFFFu /* FLT_MAX -> top level */;
        while (lo + 1 < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float r;
            memcpy(&r, &mid, 4);
            if (predict_scale_host(r, log_scale_factor, nlevels) >= k + 1) hi = mid; else lo = mid;
        }
        memcpy(&thresholds[k], &hi, 4);
    }
    return ORBX_OK;
}

extern "C" int orbm_is_in_frustum(const orbm_worldpoint_t *pts, int m, const float *Tcw16, const orbm_camera_t *cam,
                                  const orbm_grid_geom_t *g, float viewing_cos_limit, const float *thresholds, int nlevels,
                                  orbm_mappoint_t *out, int device) {
    if (m < 0 || !Tcw16 || !cam || !g || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || (nlevels > 1 && !thresholds) || (m > 0 && (!pts || !out))) {
        orbx_set_error("orbm_is_in_frustum: bad arguments");
        return ORBX_ERR_ARG;
    }
    if (m == 0) return ORBX_OK;
    return fast_is_in_frustum(pts, m, Tcw16, cam, g, viewing_cos_limit, thresholds, nlevels, out, device);
}

static int search_local_points_impl(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                    const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                    const orbm_worldpoint_t *pts, const uint8_t *mp_desc, int m, const float *Tcw16,
                                    const orbm_camera_t *cam, float viewing_cos_limit, const float *thresholds,
                                    int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio, int device,
                                    int *nmatches, orbm_mappoint_t *proj_out, const DevFrame *dev) {
    if (n < 0 || m < 0 || !g || !scale_factors || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !nmatches || !Tcw16 || !cam ||
        (nlevels > 1 && !thresholds) || (n > 0 && (!kun || !desc || !uright || !frame_mp)) || (m > 0 && (!pts || !mp_desc)) || n > 65535) {
        orbx_set_error("orbm_search_local_points: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    if (m == 0) return ORBX_OK;
    std::vector<orbm_mappoint_t> tmp;
    if (!proj_out) { tmp.resize(m); proj_out = tmp.data(); }
    if (n == 0) return fast_is_in_frustum(pts, m, Tcw16, cam, g, viewing_cos_limit, thresholds, nlevels, proj_out, device);
    FrustumArgs w = {pts, Tcw16, cam, viewing_cos_limit, thresholds};
    int frc = ORBX_FAST_FALLBACK_RC;
    if (!t_matchExact)
        frc = fast_search_by_projection_mp(kun, desc, uright, n, g, scale_factors, nlevels, nullptr, mp_desc, m, frame_mp, ext_obs,
                                           th, nnratio, device, nmatches, &w, proj_out, dev);
    else {
        const int rc = fast_is_in_frustum(pts, m, Tcw16, cam, g, viewing_cos_limit, thresholds, nlevels, proj_out, device);
        if (rc) return rc;
    }
    if (frc <= 0) return frc;
    return exact_search_by_projection_mp(kun, desc, uright, n, g, scale_factors, nlevels, proj_out, mp_desc, m, frame_mp, ext_obs,
                                         th, nnratio, device, nmatches, dev);
}
extern "C" int orbm_search_local_points(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                        const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                        const orbm_worldpoint_t *pts, const uint8_t *mp_desc, int m, const float *Tcw16,
                                        const orbm_camera_t *cam, float viewing_cos_limit, const float *thresholds,
                                        int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio, int device,
                                        int *nmatches, orbm_mappoint_t *proj_out) {
    return search_local_points_impl(kun, desc, uright, n, g, scale_factors, nlevels, pts, mp_desc, m, Tcw16, cam, viewing_cos_limit,
                                    thresholds, frame_mp, ext_obs, th, nnratio, device, nmatches, proj_out, nullptr);
}
extern "C" int orbm_search_local_points_device(const orbx_keypoint_t *d_kun, const uint8_t *d_desc, const float *d_uright, int n,
                                               const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                               const orbm_worldpoint_t *pts, const uint8_t *mp_desc, int m, const float *Tcw16,
                                               const orbm_camera_t *cam, float viewing_cos_limit, const float *thresholds,
                                               int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio, int device,
                                               int *nmatches, orbm_mappoint_t *proj_out, void *stream) {
    const DevFrame dev = {(hipStream_t)stream};
    return search_local_points_impl(d_kun, d_desc, d_uright, n, g, scale_factors, nlevels, pts, mp_desc, m, Tcw16, cam, viewing_cos_limit,
                                    thresholds, frame_mp, ext_obs, th, nnratio, device, nmatches, proj_out, &dev);
}

extern "C" int orbm_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright,
                                            int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                            int nlevels, const orbm_mappoint_t *mps, const uint8_t *mp_desc, int m,
                                            int32_t *frame_mp, const int32_t *ext_obs, float th, float nnratio,
                                            int device, int *nmatches) {
    if (n < 0 || m < 0 || !g || !scale_factors || nlevels < 1 || !nmatches ||
        (n > 0 && (!kun || !desc || !uright || !frame_mp)) || (m > 0 && (!mps || !mp_desc)) || n > 65535) {
        orbx_set_error("orbm_search_by_projection_mp: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    if (n == 0 || m == 0) return ORBX_OK;
    for (int i = 0; i < m; i++)
        if (mps[i].in_view && (mps[i].level < 0 || mps[i].level >= nlevels)) {
            orbx_set_error("map point %d: level %d out of range", i, mps[i].level);
            return ORBX_ERR_ARG;
        }
    if (!t_matchExact) {
        const int frc = fast_search_by_projection_mp(kun, desc, uright, n, g, scale_factors, nlevels, mps, mp_desc, m,
                                                     frame_mp, ext_obs, th, nnratio, device, nmatches);
        if (frc <= 0) return frc;
    }
    return exact_search_by_projection_mp(kun, desc, uright, n, g, scale_factors, nlevels, mps, mp_desc, m, frame_mp, ext_obs,
                                         th, nnratio, device, nmatches);
}

static int exact_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                         const orbm_grid_geom_t *g, const float *scale_factors, int nlevels,
                                         const orbm_mappoint_t *mps, const uint8_t *mp_desc, int m, int32_t *frame_mp,
                                         const int32_t *ext_obs, float th, float nnratio, int device, int *nmatches,
                                         const DevFrame *dev) {
    StagePlan pl;   // dev: the frame's keypoints / descriptors / mvuRight are device arrays, used in place
    const size_t o_k = pl.take(dev ? 0 : sizeof(orbx_keypoint_t) * n), o_d = pl.take(dev ? 0 : (size_t)32 * n), o_u = pl.take(dev ? 0 : 4 * (size_t)n),
                 o_sf = pl.take(4 * (size_t)nlevels), o_mp = pl.take(sizeof(orbm_mappoint_t) * m), o_md = pl.take((size_t)32 * m),
                 o_eo = pl.take(ext_obs ? 4 * (size_t)n : 0), o_fm = pl.take(4 * (size_t)n);
    pl.mark_inputs();
    const size_t o_nm = pl.take(4), o_code = pl.take(2 * (size_t)n);
    int rc = stage_reserve(device, pl.off);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(device));
    hipStream_t st = dev ? dev->stream : g_sp.st;
    if (!dev) { stage_put(o_k, kun, sizeof(orbx_keypoint_t) * n); stage_put(o_d, desc, (size_t)32 * n); stage_put(o_u, uright, 4 * (size_t)n); }
    stage_put(o_sf, scale_factors, 4 * (size_t)nlevels); stage_put(o_mp, mps, sizeof(orbm_mappoint_t) * m);
    stage_put(o_md, mp_desc, (size_t)32 * m); stage_put(o_fm, frame_mp, 4 * (size_t)n);
    if (ext_obs) stage_put(o_eo, ext_obs, 4 * (size_t)n);
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, pl.in_end, hipMemcpyHostToDevice, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_search_proj_mp, dim3(1), dim3(SQ_T), 0, st, dev ? kun : stage_dev<orbx_keypoint_t>(o_k),
                       dev ? desc : stage_dev<uint8_t>(o_d), dev ? uright : stage_dev<float>(o_u), n, *g, stage_dev<float>(o_sf),
                       stage_dev<orbm_mappoint_t>(o_mp), stage_dev<uint8_t>(o_md), m,
                       stage_dev<int32_t>(o_fm), ext_obs ? stage_dev<int32_t>(o_eo) : (int32_t *)nullptr, stage_dev<uint16_t>(o_code), th,
                       nnratio, stage_dev<int32_t>(o_nm));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_fm, g_sp.d + o_fm, o_code - o_fm, hipMemcpyDeviceToHost, st));   // frame_mp | nm
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(frame_mp, g_sp.h + o_fm, 4 * (size_t)n);
    *nmatches = *(const int32_t *)(g_sp.h + o_nm);
    return ORBX_OK;
}

// ---- SearchByProjection(cur Frame, last Frame)
__global__ __launch_bounds__(SQ_T) void k_search_proj_frame(
    const orbx_keypoint_t *__restrict__ kun, const uint8_t *__restrict__ desc, const float *__restrict__ uright,
    int n, orbm_grid_geom_t g, const float *__restrict__ sf, orbm_camera_t cam, const float *__restrict__ Tc,
    const float *__restrict__ Tl, const orbm_lastpoint_t *__restrict__ last, const uint8_t *__restrict__ last_desc,
    int nlast, int32_t *__restrict__ cur_mp, const int32_t *__restrict__ ext_obs, uint16_t *__restrict__ code,
    int32_t *__restrict__ hist_idx, int32_t *__restrict__ hist_bin, float th, int mono, int check_ori,
    int32_t *__restrict__ nmatches_out) {
    __shared__ u64 sh[2 * SQ_T / 64];
    __shared__ int hn[HISTO_LENGTH];
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += SQ_T) code[j] = (uint16_t)cell_code(g, kun[j]);
    if (tid < HISTO_LENGTH) hn[tid] = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    // twc = -Rcw^T tcw; tlc = Rlw twc + tlw (:1343-1351): cv::gemm on CV_32F accumulates in double
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tc[k * 4 + i] * (double)Tc[k * 4 + 3];
        twc[i] = (float)(s * -1.0);
    }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tl[i * 4 + k] * (double)twc[k];
        tlc[i] = (float)(s + (double)Tl[i * 4 + 3]);
    }
    const bool bForward = tlc[2] > cam.mb && !mono;
    const bool bBackward = -tlc[2] > cam.mb && !mono;
    int nm = 0, nh = 0;
    for (int i = 0; i < nlast; i++) {
        const orbm_lastpoint_t p = last[i];
        if (!p.has_mp) continue;
        float x3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            s += (double)Tc[r * 4 + 0] * (double)p.wx;
            s += (double)Tc[r * 4 + 1] * (double)p.wy;
            s += (double)Tc[r * 4 + 2] * (double)p.wz;
            x3[r] = (float)(s + (double)Tc[r * 4 + 3]);
        }
        const float xc = x3[0], yc = x3[1];
        const float invzc = (float)(1.0 / (double)x3[2]);
        if (invzc < 0) continue;
        const float u = cam.fx * xc * invzc + cam.cx;
        const float v = cam.fy * yc * invzc + cam.cy;
        if (u < g.min_x || u > g.max_x) continue;
        if (v < g.min_y || v > g.max_y) continue;
        const int nLastOctave = p.octave;
        const float radius = th * sf[nLastOctave];
        AreaQuery q;
        if (bForward) q = make_query(g, u, v, radius, nLastOctave, -1);
        else if (bBackward) q = make_query(g, u, v, radius, 0, nLastOctave);
        else q = make_query(g, u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (q.empty) continue;
        const Desc256 da = load_desc(last_desc + (size_t)i * 32);
        Top2 t;
        t.k1 = t.k2 = ~0ull;
        for (int j = tid; j < n; j += SQ_T) {
            const unsigned c = code[j];
            if (!in_area(q, c, kun[j])) continue;
            {
                const int hm = cur_mp[j];
                if (hm == -2) { if (ext_obs && ext_obs[j] > 0) continue; }
                else if (hm >= 0 && last[hm].observations > 0) continue;
            }
            if (uright[j] > 0) {
                const float ur = u - cam.mbf * invzc;
                const float er = fabsf(ur - uright[j]);
                if (er > radius) continue;
            }
            top2_insert(t, scan_key(ham(da, load_desc(desc + (size_t)j * 32)), c, j));
        }
        t = block_top2(t, sh);
        if (t.k1 != ~0ull) {
            const int bestDist = (int)(t.k1 >> 28), bestIdx2 = (int)(t.k1 & 0xFFFF);
            if (bestDist <= TH_HIGH) {
                nm++;
                if (tid == 0) {
                    cur_mp[bestIdx2] = i;
                    if (check_ori) {
                        float rot = p.angle - kun[bestIdx2].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        hist_idx[nh] = bestIdx2;
                        hist_bin[nh] = bin;
                        hn[bin]++;
                    }
                }
                nh++;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (check_ori) {
        __shared__ int ind[3];
        __shared__ int sh_dec;
        if (tid == 0) { three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]); sh_dec = 0; }
        __syncthreads();
        int dec = 0;
        for (int k = tid; k < nh; k += SQ_T) {
            const int bn = hist_bin[k];
            if (bn != ind[0] && bn != ind[1] && bn != ind[2]) { cur_mp[hist_idx[k]] = -1; dec++; }  // :1463-1464
        }
        if (dec) atomicAdd(&sh_dec, dec);
        __syncthreads();
        nm -= sh_dec;
    }
    if (tid == 0) *nmatches_out = nm;
}

static int search_by_projection_frame_impl(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright,
                                           int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                           int nlevels, const orbm_camera_t *cam, const float *Tcw_cur16,
                                           const float *Tcw_last16, const orbm_lastpoint_t *last,
                                           const uint8_t *last_desc, int nlast, int32_t *cur_mp,
                                           const int32_t *ext_obs, float th, int mono, int check_orientation,
                                           int device, int *nmatches, const DevFrame *dev) {
    if (n < 0 || nlast < 0 || !g || !scale_factors || nlevels < 1 || !cam || !Tcw_cur16 || !Tcw_last16 || !nmatches ||
        (n > 0 && (!kun || !desc || !uright || !cur_mp)) || (nlast > 0 && (!last || !last_desc)) || n > 65535) {
        orbx_set_error("orbm_search_by_projection_frame: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    if (n == 0 || nlast == 0) return ORBX_OK;
    for (int i = 0; i < nlast; i++)
        if (last[i].has_mp && (last[i].octave < 0 || last[i].octave >= nlevels)) {
            orbx_set_error("last point %d: octave %d out of range", i, last[i].octave);
            return ORBX_ERR_ARG;
        }
    if (!t_matchExact) {
        const int frc = fast_search_by_projection_frame(kun, desc, uright, n, g, scale_factors, nlevels, cam, Tcw_cur16,
                                                        Tcw_last16, last, last_desc, nlast, cur_mp, ext_obs, th, mono,
                                                        check_orientation, device, nmatches, dev);
        if (frc <= 0) return frc;
    }
    StagePlan pl;   // dev: kun / desc / uright / last_desc are device arrays, used in place
    const size_t o_k = pl.take(dev ? 0 : sizeof(orbx_keypoint_t) * n), o_d = pl.take(dev ? 0 : (size_t)32 * n), o_u = pl.take(dev ? 0 : 4 * (size_t)n),
                 o_sf = pl.take(4 * (size_t)nlevels), o_T = pl.take(4 * 32), o_l = pl.take(sizeof(orbm_lastpoint_t) * nlast),
                 o_ld = pl.take(dev ? 0 : (size_t)32 * nlast), o_eo = pl.take(ext_obs ? 4 * (size_t)n : 0), o_cm = pl.take(4 * (size_t)n);
    pl.mark_inputs();
    const size_t o_nm = pl.take(4), o_code = pl.take(2 * (size_t)n), o_hi = pl.take(4 * (size_t)nlast), o_hb = pl.take(4 * (size_t)nlast);
    int rc = stage_reserve(device, pl.off);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(device));
    hipStream_t st = dev ? dev->stream : g_sp.st;
    if (!dev) {
        stage_put(o_k, kun, sizeof(orbx_keypoint_t) * n); stage_put(o_d, desc, (size_t)32 * n); stage_put(o_u, uright, 4 * (size_t)n);
        stage_put(o_ld, last_desc, (size_t)32 * nlast);
    }
    stage_put(o_sf, scale_factors, 4 * (size_t)nlevels);
    stage_put(o_T, Tcw_cur16, 64); stage_put(o_T + 64, Tcw_last16, 64);
    stage_put(o_l, last, sizeof(orbm_lastpoint_t) * nlast);
    stage_put(o_cm, cur_mp, 4 * (size_t)n);
    if (ext_obs) stage_put(o_eo, ext_obs, 4 * (size_t)n);
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, pl.in_end, hipMemcpyHostToDevice, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_search_proj_frame, dim3(1), dim3(SQ_T), 0, st, dev ? kun : stage_dev<orbx_keypoint_t>(o_k),
                       dev ? desc : stage_dev<uint8_t>(o_d), dev ? uright : stage_dev<float>(o_u), n, *g, stage_dev<float>(o_sf), *cam,
                       stage_dev<float>(o_T), stage_dev<float>(o_T) + 16, stage_dev<orbm_lastpoint_t>(o_l),
                       dev ? last_desc : stage_dev<uint8_t>(o_ld), nlast, stage_dev<int32_t>(o_cm),
                       ext_obs ? stage_dev<int32_t>(o_eo) : (int32_t *)nullptr, stage_dev<uint16_t>(o_code), stage_dev<int32_t>(o_hi),
                       stage_dev<int32_t>(o_hb), th, mono, check_orientation, stage_dev<int32_t>(o_nm));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_cm, g_sp.d + o_cm, o_code - o_cm, hipMemcpyDeviceToHost, st));   // cur_mp | nm
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(cur_mp, g_sp.h + o_cm, 4 * (size_t)n);
    *nmatches = *(const int32_t *)(g_sp.h + o_nm);
    return ORBX_OK;
}

extern "C" int orbm_search_by_projection_frame(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright,
                                               int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                               int nlevels, const orbm_camera_t *cam, const float *Tcw_cur16,
                                               const float *Tcw_last16, const orbm_lastpoint_t *last,
                                               const uint8_t *last_desc, int nlast, int32_t *cur_mp,
                                               const int32_t *ext_obs, float th, int mono, int check_orientation,
                                               int device, int *nmatches) {
    return search_by_projection_frame_impl(kun, desc, uright, n, g, scale_factors, nlevels, cam, Tcw_cur16, Tcw_last16, last, last_desc,
                                           nlast, cur_mp, ext_obs, th, mono, check_orientation, device, nmatches, nullptr);
}
extern "C" int orbm_search_by_projection_frame_device(const orbx_keypoint_t *d_kun, const uint8_t *d_desc, const float *d_uright,
                                                      int n, const orbm_grid_geom_t *g, const float *scale_factors,
                                                      int nlevels, const orbm_camera_t *cam, const float *Tcw_cur16,
                                                      const float *Tcw_last16, const orbm_lastpoint_t *last,
                                                      const uint8_t *d_last_desc, int nlast, int32_t *cur_mp,
                                                      const int32_t *ext_obs, float th, int mono, int check_orientation,
                                                      int device, int *nmatches, void *stream) {
    const DevFrame dev = {(hipStream_t)stream};
    return search_by_projection_frame_impl(d_kun, d_desc, d_uright, n, g, scale_factors, nlevels, cam, Tcw_cur16, Tcw_last16, last,
                                           d_last_desc, nlast, cur_mp, ext_obs, th, mono, check_orientation, device, nmatches, &dev);
}

// ---- generic projected-window matcher, exact one-workgroup form (fallback of fast_match_windows)
__global__ __launch_bounds__(SQ_T) void k_match_windows_exact(
    const orbx_keypoint_t *__restrict__ kun, const uint8_t *__restrict__ desc, const float *__restrict__ uright, int n,
    orbm_grid_geom_t g, orbm_grid_geom_t ga, const orbm_window_query_t *__restrict__ qs,
    const uint8_t *__restrict__ qdesc, int m,
    int32_t *__restrict__ holder, const int32_t *__restrict__ ext_blocks, uint16_t *__restrict__ code,
    int32_t *__restrict__ hist_idx, int32_t *__restrict__ hist_bin, int max_dist, int check_ori,
    int32_t *__restrict__ nmatches_out) {
    __shared__ u64 sh[2 * SQ_T / 64];
    __shared__ int hn[HISTO_LENGTH];
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += SQ_T) code[j] = (uint16_t)cell_code(ga, kun[j]);
    if (tid < HISTO_LENGTH) hn[tid] = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0, nh = 0;
    for (int i = 0; i < m; i++) {
        const orbm_window_query_t p = qs[i];
        if (!p.valid) continue;
        const AreaQuery q = make_query(g, p.u, p.v, p.radius, p.min_level, p.max_level);
        if (q.empty) continue;
        const Desc256 da = load_desc(qdesc + (size_t)i * 32);
        Top2 t;
        t.k1 = t.k2 = ~0ull;
        for (int j = tid; j < n; j += SQ_T) {
            const unsigned c = code[j];
            if (!in_area(q, c, kun[j])) continue;
            const int hm = holder[j];
            if (hm == -2) { if (!ext_blocks || ext_blocks[j] != 0) continue; }
            else if (hm >= 0 && qs[hm].blocks != 0) continue;
            if (p.ur_tol >= 0.0f && uright && uright[j] > 0 && fabsf(p.ur_c - uright[j]) > p.ur_tol) continue;
            top2_insert(t, scan_key(ham(da, load_desc(desc + (size_t)j * 32)), c, j));
        }
        t = block_top2(t, sh);
        if (t.k1 != ~0ull) {
            const int bestDist = (int)(t.k1 >> 28), best = (int)(t.k1 & 0xFFFF);
            if (bestDist <= max_dist) {
                nm++;
                if (tid == 0) {
                    holder[best] = i;
                    if (check_ori) {
                        float rot = p.angle - kun[best].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        hist_idx[nh] = best;
                        hist_bin[nh] = bin;
                        hn[bin]++;
                    }
                }
                nh++;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (check_ori) {
        __shared__ int ind[3];
        __shared__ int sh_dec;
        if (tid == 0) { three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]); sh_dec = 0; }
        __syncthreads();
        int dec = 0;
        for (int k = tid; k < nh; k += SQ_T) {
            const int bn = hist_bin[k];
            if (bn != ind[0] && bn != ind[1] && bn != ind[2]) { holder[hist_idx[k]] = -1; dec++; }
        }
        if (dec) atomicAdd(&sh_dec, dec);
        __syncthreads();
        nm -= sh_dec;
    }
    if (tid == 0) *nmatches_out = nm;
}

extern "C" int orbm_distinctive_descriptors(const uint8_t *desc, const int32_t *offsets, int npoints, int32_t *best_row,
                                            int32_t *best_median, int device) {
    if (npoints < 0 || (npoints > 0 && (!offsets || !best_row))) { orbx_set_error("orbm_distinctive_descriptors: bad arguments"); return ORBX_ERR_ARG; }
    if (npoints == 0) return ORBX_OK;
    if (offsets[0] < 0) { orbx_set_error("offsets[0] < 0"); return ORBX_ERR_ARG; }
    for (int p = 0; p < npoints; p++)
        if (offsets[p + 1] < offsets[p]) { orbx_set_error("offsets not monotonic at %d", p); return ORBX_ERR_ARG; }
    if (offsets[npoints] > 0 && !desc) { orbx_set_error("orbm_distinctive_descriptors: desc is NULL"); return ORBX_ERR_ARG; }
    return fast_distinctive_descriptors(desc, offsets, npoints, best_row, best_median, device);
}

extern "C" int orbm_best_in_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                    const orbm_grid_geom_t *g, const orbm_grid_geom_t *g_assign,
                                    const orbm_window_query_t *queries,
                                    const uint8_t *query_desc, int m, const float *inv_level_sigma2, int nlevels,
                                    int32_t *best_idx, int32_t *best_dist, int device) {
    if (n < 0 || m < 0 || !g || (n > 0 && (!kun || !desc)) || (m > 0 && (!queries || !query_desc || !best_idx || !best_dist)) ||
        n > 65535 || (inv_level_sigma2 && nlevels <= 0)) {
        orbx_set_error("orbm_best_in_windows: bad arguments");
        return ORBX_ERR_ARG;
    }
    for (int i = 0; i < m; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (n == 0 || m == 0) return ORBX_OK;
    return fast_best_in_windows(kun, desc, uright, n, g, g_assign ? g_assign : g, queries, query_desc, m, inv_level_sigma2, nlevels, best_idx,
                                best_dist, device);
}

extern "C" int orbm_match_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                  const orbm_grid_geom_t *g, const orbm_grid_geom_t *g_assign,
                                  const orbm_window_query_t *queries, const uint8_t *query_desc,
                                  int m, int32_t *holder, const int32_t *ext_blocks, int max_dist, int check_orientation,
                                  int device, int *nmatches) {
    if (n < 0 || m < 0 || !g || !nmatches || (n > 0 && (!kun || !desc || !holder)) || (m > 0 && (!queries || !query_desc)) ||
        n > 65535) {
        orbx_set_error("orbm_match_windows: bad arguments");
        return ORBX_ERR_ARG;
    }
    *nmatches = 0;
    if (n == 0 || m == 0) return ORBX_OK;
    for (int i = 0; i < n; i++)
        if (holder[i] < -2 || holder[i] >= m) { orbx_set_error("holder[%d] = %d out of range", i, holder[i]); return ORBX_ERR_ARG; }
    if (!t_matchExact) {
        const int frc = fast_match_windows(kun, desc, uright, n, g, g_assign ? g_assign : g, queries, query_desc, m, holder, ext_blocks, max_dist,
                                           check_orientation, device, nmatches);
        if (frc <= 0) return frc;
    }
    StagePlan pl;
    const size_t o_k = pl.take(sizeof(orbx_keypoint_t) * n), o_d = pl.take((size_t)32 * n), o_q = pl.take(sizeof(orbm_window_query_t) * m),
                 o_qd = pl.take((size_t)32 * m), o_u = pl.take(uright ? 4 * (size_t)n : 0), o_eb = pl.take(ext_blocks ? 4 * (size_t)n : 0),
                 o_h = pl.take(4 * (size_t)n);
    pl.mark_inputs();
    const size_t o_nm = pl.take(4), o_code = pl.take(2 * (size_t)n), o_hi = pl.take(4 * (size_t)m), o_hb = pl.take(4 * (size_t)m);
    int rc = stage_reserve(device, pl.off);
    if (rc) return rc;
    ORBX_HIP(hipSetDevice(device));
    hipStream_t st = g_sp.st;
    stage_put(o_k, kun, sizeof(orbx_keypoint_t) * n); stage_put(o_d, desc, (size_t)32 * n);
    stage_put(o_q, queries, sizeof(orbm_window_query_t) * m); stage_put(o_qd, query_desc, (size_t)32 * m);
    stage_put(o_h, holder, 4 * (size_t)n);
    if (uright) stage_put(o_u, uright, 4 * (size_t)n);
    if (ext_blocks) stage_put(o_eb, ext_blocks, 4 * (size_t)n);
    ORBX_HIP(hipMemcpyAsync(g_sp.d, g_sp.h, pl.in_end, hipMemcpyHostToDevice, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_match_windows_exact, dim3(1), dim3(SQ_T), 0, st, stage_dev<orbx_keypoint_t>(o_k), stage_dev<uint8_t>(o_d),
                       uright ? stage_dev<float>(o_u) : (float *)nullptr, n, *g, g_assign ? *g_assign : *g,
                       stage_dev<orbm_window_query_t>(o_q), stage_dev<uint8_t>(o_qd), m, stage_dev<int32_t>(o_h),
                       ext_blocks ? stage_dev<int32_t>(o_eb) : (int32_t *)nullptr, stage_dev<uint16_t>(o_code), stage_dev<int32_t>(o_hi),
                       stage_dev<int32_t>(o_hb), max_dist, check_orientation, stage_dev<int32_t>(o_nm));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(g_sp.h + o_h, g_sp.d + o_h, o_code - o_h, hipMemcpyDeviceToHost, st));   // holder | nm
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(holder, g_sp.h + o_h, 4 * (size_t)n);
    *nmatches = *(const int32_t *)(g_sp.h + o_nm);
    return ORBX_OK;
}
