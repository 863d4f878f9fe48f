// orbx_match_fast.hip — guided searches (SearchForInitialization, SearchByProjection x2) as
// a PARALLEL candidate search followed by a SPECULATIVE sequential resolution.
//
// The reference loops over its queries in order and every accepted match changes what later
// queries may take (mvpMapPoints[idx] already set, :87-89 / :1405-1407; vMatchedDistance gate
// and steal, :444-445,463-470).  Two observations make this parallel without changing a bit:
//  A. the expensive part of a query — window test, level test, stereo gate, Hamming distance
//     of every candidate — does not depend on earlier queries.  k_cand runs it with one wave
//     per query and keeps the QK best candidates sorted by the reference's scan-order key
//     (dist | cellx | celly | index), plus the number of candidates found.
//  B. the state only ever BLOCKS more candidates (a keypoint gets a holder; vMatchedDistance
//     only decreases), so a query's decision can only change through a keypoint that is its
//     current best or second-best.  k_resolve_* walks the queries 64 at a time: every lane
//     decides from the current state, a lane may commit if no EARLIER lane of the chunk claims
//     one of its two keypoints, the longest conflict-free prefix commits, the rest re-decides.
// If a query runs out of its QK candidates while more existed, or sizes exceed the LDS plan,
// the call falls back to the exact one-workgroup kernels of orbx_match.hip (same results).
#include "orbx_match_dev.h"
#include <math.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>

#define QK 8
#define CAND_CAP 512  // candidates staged per query in LDS before the top-QK selection

struct GQuery {
    float x, y, r;
    int32_t minLevel, maxLevel, valid;
    float ur_c, ur_tol;  // stereo gate: skip j if uright[j] > 0 && |ur_c - uright[j]| > ur_tol (ur_tol < 0: off)
};

// key = dist << 32 | cellx << 26 | celly << 20 | index << 4 | octave: ordered like the reference's scan
__device__ __forceinline__ u64 fast_key(int dist, unsigned code, int j, int octave) {
    return ((u64)dist << 32) | ((u64)(code >> 8) << 26) | ((u64)(code & 0xFF) << 20) | ((u64)j << 4) | (u64)(octave & 15);
}
#define KEY_DIST(k) ((int)((k) >> 32))
#define KEY_IDX(k) ((int)(((k) >> 4) & 0xFFFF))
#define KEY_OCT(k) ((int)((k) & 15))

__global__ __launch_bounds__(256) void k_cell_codes(const orbx_keypoint_t *__restrict__ kp, int n, orbm_grid_geom_t g,
                                                    uint16_t *__restrict__ code) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) code[j] = (uint16_t)cell_code(g, kp[j]);
}

// A keypoint as the candidate scan reads it: ONE coalesced 16-byte load per lane and round instead of a 2-byte code, three fields of a
// 28-byte record, a flag byte and (stereo gate) a float from four arrays (round 5: k_cand is bound by exactly these loads - 2000
// waves x 2000 keypoints).  x, y: float bits; z: cell code | octave << 16 | blocked-before-the-call << 24; w: mvuRight bits.
__device__ __forceinline__ uint4 compact_kp(const orbm_grid_geom_t &g, const orbx_keypoint_t &kp, bool blocked, float ur) {
    return make_uint4(__float_as_uint(kp.x), __float_as_uint(kp.y), cell_code(g, kp) | ((uint32_t)(kp.octave & 0xFF) << 16) | (blocked ? 1u << 24 : 0u),
                      __float_as_uint(ur));
}
__global__ __launch_bounds__(256) void k_compact_kps(const orbx_keypoint_t *__restrict__ kp, int n, orbm_grid_geom_t g, uint4 *__restrict__ ckp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) ckp[j] = compact_kp(g, kp[j], false, 0.0f);
}

// ---- A. one wave per query: all candidates -> LDS; out: the QK smallest keys (sorted), or with
// FULL the whole sorted list (stride CAND_CAP) for searches whose runner-up can be deep in the list
template <bool FULL>
__global__ __launch_bounds__(256) void k_cand(const GQuery *__restrict__ qs, const uint8_t *__restrict__ qdesc, int m,
                                              const uint4 *__restrict__ ckp, const uint8_t *__restrict__ desc, int n, orbm_grid_geom_t g,
                                              u64 *__restrict__ keys, int32_t *__restrict__ ncand) {
    // ONE QUERY PER WORKGROUP (round 5; rounds 1-4: one per wave, four per workgroup): the four waves scan a quarter of the frame's records
    // each and append their survivors to one LDS list; then all 256 lanes fetch the survivors' descriptors, and wave 0 selects.  A query
    // is a chain of ~1800 dependent-ish instructions; with one wave per query 2000 queries put two waves on a SIMD (nothing to hide the
    // latencies behind) and ran two rounds of workgroups: 13.7 us for 2012 x 2012.  Same work, four times the waves, a quarter of the chain.
    __shared__ u64 L[CAND_CAP];
    __shared__ int sh_cnt;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int qi = blockIdx.x;
    if (qi >= m) return;   // (workgroup-uniform: a launch for zero queries still starts one workgroup)
    const GQuery Q = qs[qi];
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    bool live = false;
    Desc256 da = {};
    if (Q.valid) {
        const AreaQuery aq = make_query(g, Q.x, Q.y, Q.r, Q.minLevel, Q.maxLevel);
        if (!aq.empty) {
            live = true;
            da = load_desc(qdesc + (size_t)qi * 32);
            for (int j0 = 64 * wave; j0 < n; j0 += 256) {
                const int j = j0 + lane;
                const uint4 r = ckp[min(j, n - 1)];
                bool ok = false;
                u64 key = ~0ull;
                if (j < n) {
                    const unsigned c = r.z & 0xFFFFu;
                    const int oct = (int)((r.z >> 16) & 0xFFu);
                    ok = in_area_xy(aq, c, __uint_as_float(r.x), __uint_as_float(r.y), oct) && !((r.z >> 24) & 1u);
                    if (ok && Q.ur_tol >= 0.0f) {
                        const float uu = __uint_as_float(r.w);
                        if (uu > 0 && fabsf(Q.ur_c - uu) > Q.ur_tol) ok = false;
                    }
                    if (ok) key = fast_key(0, c, j, oct);     // (the distance follows below, for all survivors at once)
                }
                const u64 mk = __ballot(ok);
                if (mk) {   // wave-uniform: one LDS atomic per wave and round reserves the survivors' places
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sh_cnt, __popcll(mk));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (ok) {
                        const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
                        if (pos < CAND_CAP) L[pos] = key;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int cnt = sh_cnt;
    const int nl = min(cnt, CAND_CAP);
    if (live)   // (workgroup-uniform) Hamming distances of the survivors, their descriptor loads all in flight together
        for (int i = tid; i < nl; i += 256) {
            const u64 k0 = L[i];
            L[i] = k0 | ((u64)ham(da, load_desc(desc + (size_t)KEY_IDX(k0) * 32)) << 32);
        }
    __syncthreads();
    if (wave != 0) return;     // the selection / sort is one wave's job
    if (FULL) {   // bitonic sort of the staged list (wave-synchronous), then copy out
        int P = 1;
        while (P < nl) P <<= 1;
        for (int i = nl + lane; i < P; i += 64) L[i] = ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int kk = 2; kk <= P; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < P; i += 64) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const u64 a = L[i], b = L[ixj];
                        if ((a > b) == ((i & kk) == 0)) { L[i] = b; L[ixj] = a; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        for (int i = lane; i < nl; i += 64) keys[(size_t)qi * CAND_CAP + i] = L[i];
        if (lane == 0) ncand[qi] = cnt;
        return;
    }
    u64 last = 0;
    bool first = true;
    u64 mine = ~0ull;
    for (int r = 0; r < QK; r++) {  // selection of the QK smallest keys (keys are unique)
        u64 best = ~0ull;
        for (int i = lane; i < nl; i += 64) {
            const u64 k = L[i];
            if ((first || k > last) && k < best) best = k;
        }
        best = wave_min_u64(best);
        if (lane == r) mine = best;
        last = best;
        first = false;
        if (best == ~0ull) break;
    }
    if (lane < QK) keys[(size_t)qi * QK + lane] = mine;
    if (lane == 0) ncand[qi] = cnt;  // cnt > CAND_CAP makes the resolver report an overflow if it matters
}

// lanes j > i of the chunk whose best / second keypoint equals lane i's claim are in conflict
__device__ __forceinline__ bool chunk_conflict(int claim, int myBest, int mySecond, u64 pending) {
    const int lane = threadIdx.x & 63;
    bool conflict = false;
    for (int i = 0; i < 63; i++) {
        if (!((pending >> i) & 1ull)) continue;  // wave-uniform
        const int ci = __builtin_amdgcn_readlane(claim, i);
        if (ci >= 0 && lane > i && (ci == myBest || ci == mySecond)) conflict = true;
    }
    return conflict;
}

// The same test through an LDS table (one byte per keypoint, 0 = unclaimed, else lane + 1): every claiming lane
// settles the table entry of its keypoint on the SMALLEST claiming lane (a write / read-back round per
// competitor, one or two in practice), every active lane looks its two keypoints up, claimants clear their
// entry.  O(1) per round instead of 63 readlane steps — the resolver is a single wave, so this is its critical path.
__device__ __forceinline__ bool chunk_conflict_lds(uint8_t *claimtab, int claim, int myBest, int mySecond, bool act) {
    const int lane = threadIdx.x & 63;
    const uint8_t me = (uint8_t)(lane + 1);
    for (;;) {
        bool wrote = false;
        if (claim >= 0) {
            const uint8_t cur = claimtab[claim];
            if (cur == 0 || cur > me) { claimtab[claim] = me; wrote = true; }
        }
        wave_sync();
        if (!__ballot(wrote)) break;
    }
    bool conflict = false;
    if (act) {
        const uint8_t c1 = myBest >= 0 ? claimtab[myBest] : (uint8_t)0, c2 = mySecond >= 0 ? claimtab[mySecond] : (uint8_t)0;
        conflict = (c1 != 0 && c1 < me) || (c2 != 0 && c2 < me);
    }
    wave_sync();
    if (claim >= 0) claimtab[claim] = 0;
    wave_sync();
    return conflict;
}

// ---- B1. SearchByProjection(Frame, MapPoints): resolution  (src/ORBmatcher.cc:98-125)
__global__ __launch_bounds__(64) void k_resolve_mp(const u64 *__restrict__ keys, const int32_t *__restrict__ ncand,
                                                   const orbm_mappoint_t *__restrict__ mps, int m, int n,
                                                   int32_t *__restrict__ frame_mp, float nnratio,
                                                   int32_t *__restrict__ out /* [0] nmatches [1] overflow */) {
    extern __shared__ uint8_t blocked[];  // [n] dynamic: keypoint got a holder with Observations() > 0; then the claim table [n]
    uint8_t *claimtab = blocked + ((n + 15) & ~15);
    const int lane = threadIdx.x;
    for (int j = lane; j < n; j += 64) { blocked[j] = 0; claimtab[j] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int nm = 0;
    bool overflow = false;
    u64 kn[QK];   // the next chunk's queries, requested a chunk ahead (see k_resolve_frame)
    int ncn = 0, obsn = 0;
    auto fetch = [&](int q) {
        const int qc = min(q, m - 1);
        ncn = ncand[qc];
        obsn = mps[qc].observations;
#pragma unroll
        for (int r = 0; r < QK; r++) kn[r] = keys[(size_t)qc * QK + r];
    };
    if (m > 0) fetch(lane);
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int qi = c0 + lane;
        u64 k[QK];
#pragma unroll
        for (int r = 0; r < QK; r++) k[r] = kn[r];
        const int nc = qi < m ? ncn : 0, obs = obsn;
        if (c0 + 64 < m) fetch(c0 + 64 + lane);   // wave-uniform
        if (nc > CAND_CAP) overflow = true;  // k_cand dropped candidates: its top-QK is not trustworthy
        u64 pending = __ballot(qi < m && nc > 0);
        while (pending) {
            const bool act = (pending >> lane) & 1ull;
            int best = -1, second = -1, bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1;
            bool ranout = false;
            if (act) {
                int found = 0;
#pragma unroll
                for (int r = 0; r < QK; r++) {
                    if (found < 2 && k[r] != ~0ull) {
                        const int idx = KEY_IDX(k[r]);
                        if (!blocked[idx]) {
                            if (found == 0) { best = idx; bestDist = KEY_DIST(k[r]); bestLevel = KEY_OCT(k[r]); }
                            else { second = idx; bestDist2 = KEY_DIST(k[r]); bestLevel2 = KEY_OCT(k[r]); }
                            found++;
                        }
                    }
                }
                ranout = found < 2 && nc > QK;  // more candidates existed than were kept
            }
            const bool accept = act && best >= 0 && bestDist <= TH_HIGH &&
                                !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2);
            const bool conflict = chunk_conflict_lds(claimtab, accept ? best : -1, best, second, act);
            const u64 cm = __ballot(conflict);
            const u64 commit = cm ? (pending & ((1ull << __builtin_ctzll(cm)) - 1ull)) : pending;
            const bool mineCommits = (commit >> lane) & 1ull;
            if (mineCommits && ranout) overflow = true;
            if (mineCommits && accept) {
                frame_mp[best] = qi;       // :122
                if (obs > 0) blocked[best] = 1;
                nm++;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            pending &= ~commit;
        }
    }
    nm = wave_sum_i32(nm);
    const u64 ov = __ballot(overflow);
    if (lane == 0) { out[0] = nm; out[1] = ov ? 1 : 0; }
}

// ---- B2. SearchForInitialization: resolution  (src/ORBmatcher.cc:433-512)
__global__ __launch_bounds__(64) void k_resolve_init(const u64 *__restrict__ keys, const int32_t *__restrict__ ncand,
                                                     const orbx_keypoint_t *__restrict__ k1, const orbx_keypoint_t *__restrict__ k2,
                                                     int n1, int n2, float *__restrict__ prev, int32_t *__restrict__ m12,
                                                     int32_t *__restrict__ bin1, float nnratio, int check_ori,
                                                     int32_t *__restrict__ out) {
    extern __shared__ int32_t sm[];  // vmd[n2], m21[n2], claim table [n2] bytes
    int32_t *vmd = sm, *m21 = sm + n2;
    uint8_t *claimtab = (uint8_t *)(sm + 2 * n2);
    __shared__ int hn[HISTO_LENGTH];
    __shared__ int ind[3];
    const int lane = threadIdx.x;
    for (int j = lane; j < n2; j += 64) { vmd[j] = INT_MAX; m21[j] = -1; claimtab[j] = 0; }
    for (int i = lane; i < n1; i += 64) { m12[i] = -1; bin1[i] = -1; }
    if (lane < HISTO_LENGTH) hn[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0;
    bool overflow = false;
    for (int c0 = 0; c0 < n1; c0 += 64) {
        const int qi = c0 + lane;
        int nc = 0;
        if (qi < n1) nc = ncand[qi];
        const u64 *kl = keys + (size_t)(qi < n1 ? qi : 0) * CAND_CAP;  // whole candidate list, sorted
        if (nc > CAND_CAP) overflow = true;
        nc = min(nc, CAND_CAP);
        int resume = 0;  // entries before the current best stay gated for good: re-decisions resume there
        u64 pending = __ballot(qi < n1 && nc > 0);
        while (pending) {
            const bool act = (pending >> lane) & 1ull;
            int best = -1, second = -1, bestDist = INT_MAX, bestDist2 = INT_MAX;
            if (act) {
                int found = 0;
                for (int p = resume; p < nc && found < 2; p++) {
                    const u64 key = kl[p];
                    const int idx = KEY_IDX(key), dist = KEY_DIST(key);
                    if (!(vmd[idx] <= dist)) {  // :444-445
                        if (found == 0) { best = idx; bestDist = dist; resume = p; }
                        else { second = idx; bestDist2 = dist; }
                        found++;
                    }
                }
            }
            const bool ranout = false;
            const bool accept = act && best >= 0 && bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio;
            const bool conflict = chunk_conflict_lds(claimtab, accept ? best : -1, best, second, act);
            const u64 cm = __ballot(conflict);
            const u64 commit = cm ? (pending & ((1ull << __builtin_ctzll(cm)) - 1ull)) : pending;
            const bool mineCommits = (commit >> lane) & 1ull;
            if (mineCommits && ranout) overflow = true;
            if (mineCommits && accept) {
                const int old = m21[best];
                if (old >= 0) { m12[old] = -1; nm--; }   // steal (:463-467)
                m12[qi] = best; m21[best] = qi; vmd[best] = bestDist; nm++;
                if (check_ori) {
                    float rot = k1[qi].angle - k2[best].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    bin1[qi] = bin;
                    atomicAdd(&hn[bin], 1);
                }
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            pending &= ~commit;
        }
    }
    if (check_ori) {
        if (lane == 0) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n1; i += 64) {
            const int bn = bin1[i];
            if (bn >= 0 && bn != ind[0] && bn != ind[1] && bn != ind[2] && m12[i] >= 0) { m12[i] = -1; nm--; }
        }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n1; i += 64)
        if (m12[i] >= 0) { prev[2 * i] = k2[m12[i]].x; prev[2 * i + 1] = k2[m12[i]].y; }
    nm = wave_sum_i32(nm);
    const u64 ov = __ballot(overflow);
    if (lane == 0) { out[0] = nm; out[1] = ov ? 1 : 0; }
}

// ---- B3. SearchByProjection(cur, last): resolution  (src/ORBmatcher.cc:1399-1469)
__global__ __launch_bounds__(64) void k_resolve_frame(const u64 *__restrict__ keys, const int32_t *__restrict__ ncand,
                                                      const orbm_lastpoint_t *__restrict__ last,
                                                      const orbx_keypoint_t *__restrict__ kun, int nlast, int n,
                                                      int32_t *__restrict__ cur_mp, int32_t *__restrict__ hist_idx,
                                                      int32_t *__restrict__ hist_bin, int check_ori,
                                                      int32_t *__restrict__ out) {
    extern __shared__ uint8_t blocked[];   // [n], then the claim table [n]
    uint8_t *claimtab = blocked + ((n + 15) & ~15);
    __shared__ int hn[HISTO_LENGTH];
    __shared__ int ind[3];
    const int lane = threadIdx.x;
    for (int j = lane; j < n; j += 64) { blocked[j] = 0; claimtab[j] = 0; }
    if (lane < HISTO_LENGTH) hn[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0, nh = 0;
    bool overflow = false;
    // a chunk's queries are requested one chunk ahead: the wave is alone in its workgroup, and with the loads at the head of the
    // chunk every 64 queries waited for a memory round trip of their own (2000 queries: 32 x 1.6 us of a 53-us kernel)
    u64 kn[QK];
    int ncn = 0, obsn = 0;
    float angn = 0;
    auto fetch = [&](int q) {
        const int qc = min(q, nlast - 1);
        ncn = ncand[qc];
        obsn = last[qc].observations;
        angn = last[qc].angle;
#pragma unroll
        for (int r = 0; r < QK; r++) kn[r] = keys[(size_t)qc * QK + r];
    };
    if (nlast > 0) fetch(lane);
    for (int c0 = 0; c0 < nlast; c0 += 64) {
        const int qi = c0 + lane;
        u64 k[QK];
#pragma unroll
        for (int r = 0; r < QK; r++) k[r] = kn[r];
        const int nc = qi < nlast ? ncn : 0, obs = obsn;
        const float ang = angn;
        if (c0 + 64 < nlast) fetch(c0 + 64 + lane);   // wave-uniform
        if (nc > CAND_CAP) overflow = true;
        u64 pending = __ballot(qi < nlast && nc > 0);
        while (pending) {
            const bool act = (pending >> lane) & 1ull;
            int best = -1, bestDist = 256;
            bool ranout = false;
            if (act) {
#pragma unroll
                for (int r = 0; r < QK; r++)
                    if (best < 0 && k[r] != ~0ull && !blocked[KEY_IDX(k[r])]) { best = KEY_IDX(k[r]); bestDist = KEY_DIST(k[r]); }
                ranout = best < 0 && nc > QK;
            }
            const bool accept = act && best >= 0 && bestDist <= TH_HIGH;
            const bool conflict = chunk_conflict_lds(claimtab, accept ? best : -1, best, -1, act);
            const u64 cm = __ballot(conflict);
            const u64 commit = cm ? (pending & ((1ull << __builtin_ctzll(cm)) - 1ull)) : pending;
            const bool mineCommits = (commit >> lane) & 1ull;
            if (mineCommits && ranout) overflow = true;
            const bool doit = mineCommits && accept;
            const u64 dm = __ballot(doit);
            if (doit) {
                cur_mp[best] = qi;  // :1430
                if (obs > 0) blocked[best] = 1;
                nm++;
                if (check_ori) {
                    float rot = ang - kun[best].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    const int pos = nh + __popcll(dm & ((1ull << lane) - 1ull));
                    hist_idx[pos] = best;
                    hist_bin[pos] = bin;
                    atomicAdd(&hn[bin], 1);
                }
            }
            nh += __popcll(dm);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            pending &= ~commit;
        }
    }
    if (check_ori) {
        if (lane == 0) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < nh; t += 64) {
            const int bn = hist_bin[t];
            if (bn != ind[0] && bn != ind[1] && bn != ind[2]) { cur_mp[hist_idx[t]] = -1; nm--; }  // :1463-1464
        }
    }
    nm = wave_sum_i32(nm);
    const u64 ov = __ballot(overflow);
    if (lane == 0) { out[0] = nm; out[1] = ov ? 1 : 0; }
}

// ---- B4. generic projected-window matcher: resolution (top-1, any blocking holder; :1539-1570)
__global__ __launch_bounds__(64) void k_resolve_windows(const u64 *__restrict__ keys, const int32_t *__restrict__ ncand,
                                                        const orbm_window_query_t *__restrict__ qs,
                                                        const orbx_keypoint_t *__restrict__ kun, int m, int n,
                                                        int32_t *__restrict__ holder, int32_t *__restrict__ hist_idx,
                                                        int32_t *__restrict__ hist_bin, int max_dist, int check_ori,
                                                        int32_t *__restrict__ out) {
    extern __shared__ uint8_t blocked[];   // [n], then the claim table [n]
    uint8_t *claimtab = blocked + ((n + 15) & ~15);
    __shared__ int hn[HISTO_LENGTH];
    __shared__ int ind[3];
    const int lane = threadIdx.x;
    for (int j = lane; j < n; j += 64) { blocked[j] = 0; claimtab[j] = 0; }
    if (lane < HISTO_LENGTH) hn[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0, nh = 0;
    bool overflow = false;
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int qi = c0 + lane;
        u64 k[QK];
        int nc = 0, blocks = 0;
        float ang = 0;
        if (qi < m) {
            nc = ncand[qi];
            blocks = qs[qi].blocks;
            ang = qs[qi].angle;
#pragma unroll
            for (int r = 0; r < QK; r++) k[r] = keys[(size_t)qi * QK + r];
        }
        if (nc > CAND_CAP) overflow = true;
        u64 pending = __ballot(qi < m && nc > 0);
        while (pending) {
            const bool act = (pending >> lane) & 1ull;
            int best = -1, bestDist = 256;
            bool ranout = false;
            if (act) {
#pragma unroll
                for (int r = 0; r < QK; r++)
                    if (best < 0 && k[r] != ~0ull && !blocked[KEY_IDX(k[r])]) { best = KEY_IDX(k[r]); bestDist = KEY_DIST(k[r]); }
                ranout = best < 0 && nc > QK;
            }
            const bool accept = act && best >= 0 && bestDist <= max_dist;
            const bool conflict = chunk_conflict_lds(claimtab, accept ? best : -1, best, -1, act);
            const u64 cm = __ballot(conflict);
            const u64 commit = cm ? (pending & ((1ull << __builtin_ctzll(cm)) - 1ull)) : pending;
            const bool mineCommits = (commit >> lane) & 1ull;
            if (mineCommits && ranout) overflow = true;
            const bool doit = mineCommits && accept;
            const u64 dm = __ballot(doit);
            if (doit) {
                holder[best] = qi;
                if (blocks) blocked[best] = 1;
                nm++;
                if (check_ori) {
                    float rot = ang - kun[best].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    const int pos = nh + __popcll(dm & ((1ull << lane) - 1ull));
                    hist_idx[pos] = best;
                    hist_bin[pos] = bin;
                    atomicAdd(&hn[bin], 1);
                }
            }
            nh += __popcll(dm);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            pending &= ~commit;
        }
    }
    if (check_ori) {
        if (lane == 0) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < nh; t += 64) {
            const int bn = hist_bin[t];
            if (bn != ind[0] && bn != ind[1] && bn != ind[2]) { holder[hist_idx[t]] = -1; nm--; }
        }
    }
    nm = wave_sum_i32(nm);
    const u64 ov = __ballot(overflow);
    if (lane == 0) { out[0] = nm; out[1] = ov ? 1 : 0; }
}

// ---- B'. the resolution as a FIXED POINT, by one workgroup of RP_T threads instead of one wave (round 5).
// The reference's loop is sequential because a committed match may BLOCK its keypoint for the queries behind it (mvpMapPoints[idx]
// set to a point with Observations() > 0, :87-89 / :1405-1407).  Write claim(q) for the keypoint query q commits to (none: -1) and
// T[j] = min { q : claim(q) = j and q blocks } for the first query that blocks keypoint j.  Query q sees exactly the keypoints
// with T[j] < q as blocked, so the reference's result is the unique assignment with claim(q) = decide(q, { j : T[j] < q }) for
// every q.  Iteration: start from decide(q, {}) for all q at once, rebuild T from the claims, re-decide every q, until no claim
// changes.  After round r the claims of the queries 0 .. r are final (query q's decision only depends on claims of queries
// below q), so it ends after at most m + 1 rounds - in practice two to four, each a few hundred nanoseconds of LDS traffic for
// the whole workgroup - and a fixed point IS the sequential result (induction over q).  The single-wave speculative resolver
// above took 47 us for 2000 queries (k_resolve_frame) and 19 us (k_resolve_mp) of a 100-us call; this one ~5.
// Afterwards: holder[j] = the LARGEST committed q with claim j (the loop overwrites), the orientation histogram over all
// commits (:1435-1469: an entry in a losing bin clears its keypoint even when a later query overwrote the holder), counts.
// MODE 0: SearchByProjection(F, MPs) (best + second, ratio test); 1: SearchByProjection(cur, last); 2: projected windows.
#define RP_T 1024       // (measured: 512 threads x 8 queries 26.8 us, 1024 x 4 18.2 us for 2012 queries - a thread's decisions are chains of dependent LDS reads)
#define RP_QMAX 4       // queries per thread: 2 (m <= 2048) or 4 (m <= 4096) - template parameter RP_Q; larger calls keep the single-wave resolver
template <int MODE, int RP_Q>
__global__ __launch_bounds__(RP_T) void k_resolve_par(const u64 *__restrict__ keys, const int32_t *__restrict__ ncand,
                                                      const void *__restrict__ meta, const orbx_keypoint_t *__restrict__ kun,
                                                      int m, int n, const int32_t *holder_in, int32_t *holder_out,
                                                      int32_t *holder_host, float nnratio, int max_dist, int check_ori,
                                                      int32_t *__restrict__ out, int32_t *__restrict__ out_host, int32_t *doneFlag, int doneSeq) {
#ifdef ORBX_DEVELOPER
    const unsigned long long dvT0 = wall_clock64();
#endif
    extern __shared__ __align__(16) uint32_t rp_lds[];
    uint32_t *cq = rp_lds;                                   // [m][QK] candidates of every query, best first: dist << 20 | index << 4 | octave (~0: none)
    uint32_t *bt = rp_lds + (size_t)QK * RP_T * RP_Q;        // [n] T[j] during the rounds, then the largest committed query + 1 (cq has a row for every (thread, r))
    uint32_t *dead = bt + n;                                 // [(n + 31) / 32] bit j: an orientation loser claimed keypoint j
    uint32_t *bt2 = dead + (n + 31) / 32;                    // [n] the blocking times of the NEXT round (double buffer: one barrier less per round)
    __shared__ int sh_flag[3];
    __shared__ int hn[HISTO_LENGTH];
    __shared__ int ind[3];
    __shared__ int sh_nm, sh_ov;
    const int tid = threadIdx.x;
    // (the candidates live in LDS, not in registers: 4 queries x 8 candidates per thread beside the decision's own state spilled)
    int nc[RP_Q], blk[RP_Q], claim[RP_Q];
    float ang[RP_Q];
    bool overflow = false;
#pragma unroll
    for (int r = 0; r < RP_Q; r++) {
        const int qi = tid + r * RP_T;
        nc[r] = 0; blk[r] = 0; ang[r] = 0.0f; claim[r] = -1;
        if (qi >= m) { ((uint4 *)(cq + (size_t)QK * qi))[0] = make_uint4(~0u, ~0u, ~0u, ~0u); ((uint4 *)(cq + (size_t)QK * qi))[1] = make_uint4(~0u, ~0u, ~0u, ~0u); }
        if (qi < m) {
            nc[r] = ncand[qi];
            if (MODE == 0) blk[r] = ((const orbm_mappoint_t *)meta)[qi].observations > 0;
            else if (MODE == 1) { const float2 qm = ((const float2 *)meta)[qi]; blk[r] = qm.x != 0.0f; ang[r] = qm.y; }   // (blocks, angle): k_queries_frame's compact copy
            else { blk[r] = ((const orbm_window_query_t *)meta)[qi].blocks != 0; ang[r] = ((const orbm_window_query_t *)meta)[qi].angle; }
            uint32_t e[QK];
#pragma unroll
            for (int k = 0; k < QK; k++) {
                e[k] = ~0u;
                if (nc[r] > 0) {
                    const u64 key = keys[(size_t)qi * QK + k];
                    if (key != ~0ull) e[k] = ((uint32_t)KEY_DIST(key) << 20) | ((uint32_t)KEY_IDX(key) << 4) | (uint32_t)KEY_OCT(key);
                }
            }
            ((uint4 *)(cq + (size_t)QK * qi))[0] = make_uint4(e[0], e[1], e[2], e[3]);
            ((uint4 *)(cq + (size_t)QK * qi))[1] = make_uint4(e[4], e[5], e[6], e[7]);
            if (nc[r] > CAND_CAP) overflow = true;   // k_cand dropped candidates: its top-QK is not trustworthy
        }
    }
    __shared__ int hnw[RP_T / 64][32];
    if (tid < HISTO_LENGTH) hn[tid] = 0;
    for (int i = tid; i < (RP_T / 64) * 32; i += RP_T) (&hnw[0][0])[i] = 0;
    if (tid == 0) { sh_nm = 0; sh_ov = 0; }
    // the holders before the call (they may sit in the pinned mirror: a bus round trip) are requested now, for the write-back at the end
    int32_t hin[RP_Q];
#pragma unroll
    for (int r = 0; r < RP_Q; r++) { const int j = tid + r * RP_T; hin[r] = j < n ? holder_in[j] : -1; }
    bool ranout[RP_Q];
    // the decision of query r of this thread given the blocking times in bt (useBt = false: nothing is blocked)
    // a decision in two steps, so that a thread's RP_Q decisions of a round overlap their LDS reads: fetch = the query's QK candidates
    // and their blocking times (btp == NULL: nothing is blocked), pick = the reference's accept rule on them
    auto fetch = [&](int r, const uint32_t *btp, uint32_t *e8, uint32_t *tb) {
        const uint32_t qi = (uint32_t)(tid + r * RP_T);
        const uint4 c0 = ((const uint4 *)(cq + (size_t)QK * qi))[0], c1 = ((const uint4 *)(cq + (size_t)QK * qi))[1];
        e8[0] = c0.x; e8[1] = c0.y; e8[2] = c0.z; e8[3] = c0.w; e8[4] = c1.x; e8[5] = c1.y; e8[6] = c1.z; e8[7] = c1.w;
#pragma unroll
        for (int k = 0; k < QK; k++) tb[k] = btp ? btp[e8[k] == ~0u ? 0u : (e8[k] >> 4) & 0xFFFFu] : 0xFFFFFFFFu;   // (an empty slot looks at entry 0 and ignores it)
    };
    auto pick = [&](int r, const uint32_t *e8, const uint32_t *tb) -> int {
        const uint32_t qi = (uint32_t)(tid + r * RP_T);
        int best = -1, bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1, found = 0;
#pragma unroll
        for (int k = 0; k < QK; k++) {
            const uint32_t e = e8[k];
            if (found < (MODE == 0 ? 2 : 1) && e != ~0u && !(tb[k] < qi)) {
                if (found == 0) { best = (int)((e >> 4) & 0xFFFFu); bestDist = (int)(e >> 20); bestLevel = (int)(e & 15u); }
                else { bestDist2 = (int)(e >> 20); bestLevel2 = (int)(e & 15u); }
                found++;
            }
        }
        ranout[r] = (MODE == 0 ? found < 2 : best < 0) && nc[r] > QK;   // more candidates existed than were kept
        bool accept;
        if (MODE == 0) accept = best >= 0 && bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2);
        else accept = best >= 0 && bestDist <= (MODE == 1 ? TH_HIGH : max_dist);
        return accept ? best : -1;
    };
    auto decide_all = [&](const uint32_t *btp) -> bool {   // every query of this thread; true if a claim changed
        uint32_t E8[RP_Q][QK], TB[RP_Q][QK];
#pragma unroll
        for (int r = 0; r < RP_Q; r++) fetch(r, btp, E8[r], TB[r]);     // (queries past m read slot rows inside the allocation: cq holds RP_T * RP_Q rows)
        bool changed = false;
#pragma unroll
        for (int r = 0; r < RP_Q; r++)
            if (nc[r] > 0) {
                const int nw = pick(r, E8[r], TB[r]);
                changed |= nw != claim[r];
                claim[r] = nw;
            }
        return changed;
    };
#pragma unroll
    for (int r = 0; r < RP_Q; r++) ranout[r] = false;
#ifdef ORBX_DEVELOPER
    const unsigned long long dvT1 = wall_clock64();
#endif
    __syncthreads();      // (fetch reads candidate rows of queries >= m too: every row of cq is written - or left alone - before anybody looks)
    decide_all(nullptr);
    for (int j = tid; j < n; j += RP_T) { bt[j] = 0xFFFFFFFFu; bt2[j] = 0xFFFFFFFFu; }
    if (tid < 3) sh_flag[tid] = 0;
    __syncthreads();
    int rounds = 0;
    for (int round = 0; round <= m + 1; round++) {   // two barriers per round: T of this round in one array while the other is reset for the next
        rounds++;
        uint32_t *A = (round & 1) ? bt2 : bt, *Bn = (round & 1) ? bt : bt2;
        if (tid == 0) sh_flag[(round + 1) % 3] = 0;
#pragma unroll
        for (int r = 0; r < RP_Q; r++)
            if (claim[r] >= 0 && blk[r]) atomicMin(&A[claim[r]], (uint32_t)(tid + r * RP_T));
        __syncthreads();
        const bool changed = decide_all(A);
        for (int j = tid; j < n; j += RP_T) Bn[j] = 0xFFFFFFFFu;     // (last read in the round before this one, behind two barriers)
        if (__ballot(changed) && (tid & 63) == 0) sh_flag[round % 3] = 1;
        __syncthreads();
        if (!sh_flag[round % 3]) break;
    }
#ifdef ORBX_DEVELOPER
    const unsigned long long dvT2 = wall_clock64();
#endif
    // holders, orientation histogram, counts
    for (int j = tid; j < n; j += RP_T) bt[j] = 0;
    for (int j = tid; j < (n + 31) / 32; j += RP_T) dead[j] = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    int nm = 0, bin[RP_Q];
#pragma unroll
    for (int r = 0; r < RP_Q; r++) {
        bin[r] = -1;
        if (ranout[r]) overflow = true;
        if (claim[r] >= 0) {
            atomicMax(&bt[claim[r]], (uint32_t)(tid + r * RP_T) + 1u);
            nm++;
            if (MODE != 0 && check_ori) {
                float rot = ang[r] - kun[claim[r]].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bn = (int)roundf(rot * factor);
                if (bn == HISTO_LENGTH) bn = 0;
                bin[r] = bn;
                atomicAdd(&hnw[tid >> 6][bn], 1);   // a histogram per wave (a few hundred commits on 30 bins of ONE histogram serialised the sixteen waves), summed below
            }
        }
    }
    __syncthreads();
    if (MODE != 0 && check_ori) {
        if (tid < HISTO_LENGTH) {
            int sum = 0;
#pragma unroll
            for (int wv = 0; wv < RP_T / 64; wv++) sum += hnw[wv][tid];
            hn[tid] = sum;
        }
        __syncthreads();
        if (tid == 0) three_maxima(hn, HISTO_LENGTH, ind[0], ind[1], ind[2]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RP_Q; r++)
            if (bin[r] >= 0 && bin[r] != ind[0] && bin[r] != ind[1] && bin[r] != ind[2]) {   // :1463-1464
                atomicOr(&dead[claim[r] >> 5], 1u << (claim[r] & 31));
                nm--;
            }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RP_Q; r++) {
        const int j = tid + r * RP_T;
        if (j < n) {
            const uint32_t wq = bt[j];
            const int32_t v = ((dead[j >> 5] >> (j & 31)) & 1u) ? -1 : wq ? (int32_t)wq - 1 : hin[r];
            holder_out[j] = v;
            if (holder_host) holder_host[j] = v;
        }
    }
    for (int j = tid + RP_Q * RP_T; j < n; j += RP_T) {
        const uint32_t wq = bt[j];
        const int32_t v = ((dead[j >> 5] >> (j & 31)) & 1u) ? -1 : wq ? (int32_t)wq - 1 : holder_in[j];
        holder_out[j] = v;
        if (holder_host) holder_host[j] = v;
    }
    nm = wave_sum_i32(nm);
    if ((tid & 63) == 0 && nm) atomicAdd(&sh_nm, nm);
    if (__ballot(overflow) && (tid & 63) == 0) sh_ov = 1;
    // (a system-scope fence in every wave costs ~0.3 us each and they serialise: +4 us on this kernel.  The results go to UNCACHED host
    // memory: once a wave's stores are acknowledged - s_waitcnt vmcnt(0) - they are on the host; one wave then fences and signals.)
    if (doneFlag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores into the pinned mirror have arrived before the barrier ...
    __syncthreads();
    if (tid == 0) {
        out[0] = sh_nm; out[1] = sh_ov; out[2] = rounds;     // [2]: rounds the iteration took (diagnostics)
        if (out_host) { out_host[0] = sh_nm; out_host[1] = sh_ov; out_host[2] = rounds; }
        // ... and the call's sequence number behind everything: the host polls this word instead of waiting for the stream (a stream
        // synchronisation costs 10-15 us of wake-up latency on top of the kernel - a third of a 45-us matcher call)
        if (doneFlag) { __threadfence_system(); __hip_atomic_store(doneFlag, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
#ifdef ORBX_DEVELOPER
        // developer build: [3] = time stamps in 0.1-us units (100-MHz counter): loads << 20 | rounds << 10 | epilogue
        const unsigned long long dvT3 = wall_clock64();
        const int stamp = ((int)min((dvT1 - dvT0) / 10ull, 1023ull) << 20) | ((int)min((dvT2 - dvT1) / 10ull, 1023ull) << 10) | (int)min((dvT3 - dvT2) / 10ull, 1023ull);
        out[3] = stamp;
        if (out_host) out_host[3] = stamp;
#endif
    }
}

__global__ __launch_bounds__(256) void k_queries_windows(const orbm_window_query_t *__restrict__ w, int m,
                                                         GQuery *__restrict__ q, const int32_t *__restrict__ holder,
                                                         const int32_t *__restrict__ ext_blocks, int n,
                                                         const orbx_keypoint_t *__restrict__ kp, const float *__restrict__ uright,
                                                         orbm_grid_geom_t gcode, uint4 *__restrict__ ckp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {   // the frame's keypoints as the candidate scan reads them (cell codes were a launch of their own, k_cell_codes)
        const int hm = holder[i];
        ckp[i] = compact_kp(gcode, kp[i], hm == -1 ? false : hm == -2 ? (ext_blocks ? (ext_blocks[i] != 0) : true) : (w[hm].blocks != 0), uright[i]);
    }
    if (i >= m) return;
    GQuery Q;
    Q.valid = w[i].valid ? 1 : 0;
    Q.x = w[i].u; Q.y = w[i].v; Q.r = w[i].radius;
    Q.minLevel = w[i].min_level; Q.maxLevel = w[i].max_level;
    Q.ur_c = w[i].ur_c; Q.ur_tol = w[i].ur_tol;
    q[i] = Q;
}

// ---- query builders
__global__ __launch_bounds__(256) void k_queries_init(const orbx_keypoint_t *__restrict__ k1, const float *__restrict__ prev,
                                                      int n1, int window, GQuery *__restrict__ q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    GQuery Q;
    const int level1 = k1[i].octave;
    Q.valid = level1 > 0 ? 0 : 1;                      // :421-423
    Q.x = prev[2 * i]; Q.y = prev[2 * i + 1]; Q.r = (float)window;
    Q.minLevel = level1; Q.maxLevel = level1;          // :425
    Q.ur_c = 0; Q.ur_tol = -1.0f;
    q[i] = Q;
}
__global__ __launch_bounds__(256) void k_queries_mp(const orbm_mappoint_t *__restrict__ mps, int m,
                                                    const float *__restrict__ sf, float th, GQuery *__restrict__ q,
                                                    const int32_t *__restrict__ frame_mp, const int32_t *__restrict__ ext_obs,
                                                    int n, const orbx_keypoint_t *__restrict__ kp, const float *__restrict__ uright,
                                                    orbm_grid_geom_t gcode, uint4 *__restrict__ ckp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {  // holders present before the call (:87-89)
        const int hm = frame_mp[i];
        ckp[i] = compact_kp(gcode, kp[i], hm == -1 ? false : hm == -2 ? (ext_obs && ext_obs[i] > 0) : (mps[hm].observations > 0), uright[i]);
    }
    if (i >= m) return;
    const orbm_mappoint_t p = mps[i];
    GQuery Q;
    Q.valid = p.in_view ? 1 : 0;
    float r = p.view_cos > 0.998 ? 2.5f : 4.0f;        // RadiusByViewingCos (:131-137)
    if (th != 1.0) r *= th;
    const int lvl = p.in_view ? p.level : 0;
    const float rs = r * sf[lvl];
    Q.x = p.proj_x; Q.y = p.proj_y; Q.r = rs;
    Q.minLevel = lvl - 1; Q.maxLevel = lvl;            // :66
    Q.ur_c = p.proj_xr; Q.ur_tol = rs;                 // :91-96
    q[i] = Q;
}
__global__ __launch_bounds__(256) void k_queries_frame(const orbm_lastpoint_t *__restrict__ last, int nlast,
                                                       const float *__restrict__ sf, orbm_camera_t cam,
                                                       orbm_grid_geom_t g, const float *__restrict__ Tc,
                                                       const float *__restrict__ Tl, float th, int mono,
                                                       GQuery *__restrict__ q, const int32_t *__restrict__ cur_mp,
                                                       const int32_t *__restrict__ ext_obs, int n,
                                                       const orbx_keypoint_t *__restrict__ kp, const float *__restrict__ uright,
                                                       uint4 *__restrict__ ckp, float2 *__restrict__ qmeta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int hm = cur_mp[i];
        ckp[i] = compact_kp(g, kp[i], hm == -1 ? false : hm == -2 ? (ext_obs && ext_obs[i] > 0) : (last[hm].observations > 0), uright[i]);
    }
    // my block's 256 records as coalesced dwords -> LDS: `last` may sit in pinned HOST memory, where a wave's seven strided field loads
    // fetched every 64-byte line seven times over the bus (the kernel ran 10 us for 56 KB)
    __shared__ uint32_t recs[256 * (sizeof(orbm_lastpoint_t) / 4)];
    {
        constexpr int W = sizeof(orbm_lastpoint_t) / 4;
        const int first = blockIdx.x * 256, cnt = max(0, min(256, nlast - first)) * W;
        const uint32_t *src = (const uint32_t *)(last + (cnt > 0 ? first : 0));   // (a block past the records stages nothing; its loads stay inside the array)
        uint32_t v[W];     // all W loads of a thread are requested before the first is stored: one bus round trip (a rolled loop made it W)
#pragma unroll
        for (int j = 0; j < W; j++) { const int k = threadIdx.x + 256 * j; v[j] = src[min(k, max(cnt - 1, 0))]; }
#pragma unroll
        for (int j = 0; j < W; j++) { const int k = threadIdx.x + 256 * j; if (k < cnt) recs[k] = v[j]; }
    }
    __syncthreads();
    if (i >= nlast) return;
    // twc = -Rcw^T tcw; tlc = Rlw twc + tlw (:1343-1351): cv::gemm on CV_32F accumulates in double
    float twc[3], tlc2 = 0;
    for (int a = 0; a < 3; a++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tc[k * 4 + a] * (double)Tc[k * 4 + 3];
        twc[a] = (float)(s * -1.0);
    }
    {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tl[2 * 4 + k] * (double)twc[k];
        tlc2 = (float)(s + (double)Tl[2 * 4 + 3]);
    }
    const bool bForward = tlc2 > cam.mb && !mono, bBackward = -tlc2 > cam.mb && !mono;
    const orbm_lastpoint_t p = ((const orbm_lastpoint_t *)recs)[threadIdx.x];
    qmeta[i] = make_float2(p.observations > 0 ? 1.0f : 0.0f, p.angle);   // what the resolver needs of the record (`last` may sit in host memory)
    GQuery Q;
    Q.valid = 0; Q.x = Q.y = Q.r = 0; Q.minLevel = Q.maxLevel = -1; Q.ur_c = 0; Q.ur_tol = -1.0f;
    if (p.has_mp) {
        float x3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            s += (double)Tc[r * 4 + 0] * (double)p.wx;
            s += (double)Tc[r * 4 + 1] * (double)p.wy;
            s += (double)Tc[r * 4 + 2] * (double)p.wz;
            x3[r] = (float)(s + (double)Tc[r * 4 + 3]);
        }
        const float invzc = (float)(1.0 / (double)x3[2]);
        if (!(invzc < 0)) {
            const float u = cam.fx * x3[0] * invzc + cam.cx, v = cam.fy * x3[1] * invzc + cam.cy;
            if (!(u < g.min_x || u > g.max_x) && !(v < g.min_y || v > g.max_y)) {
                const int oct = p.octave;
                const float radius = th * sf[oct];
                Q.valid = 1; Q.x = u; Q.y = v; Q.r = radius;
                if (bForward) { Q.minLevel = oct; Q.maxLevel = -1; }
                else if (bBackward) { Q.minLevel = 0; Q.maxLevel = oct; }
                else { Q.minLevel = oct - 1; Q.maxLevel = oct + 1; }
                Q.ur_c = u - cam.mbf * invzc; Q.ur_tol = radius;   // :1409-1414
            }
        }
    }
    q[i] = Q;
}

// ---- Frame::isInFrustum (src/Frame.cc:284-340) for a list of map points.  cv::Mat arithmetic as OpenCV
// evaluates it: Rcw*P+tcw is a gemm (double accumulation, one rounding), cv::norm and Mat::dot accumulate in
// double; PredictScale through the host-built threshold table (see orbm_predict_scale_thresholds).
struct FrustumPose { float R[9], t[3], Ow[3]; };
__global__ __launch_bounds__(256) void k_frustum(const orbm_worldpoint_t *__restrict__ pts, int m, FrustumPose P,
                                                 orbm_camera_t cam, orbm_grid_geom_t g, float viewCosLimit,
                                                 const float *__restrict__ thr, int nlevels,
                                                 orbm_mappoint_t *__restrict__ out, orbm_mappoint_t *__restrict__ out_host) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    __shared__ uint32_t recs[256 * (sizeof(orbm_worldpoint_t) / 4)];   // (the records may sit in pinned host memory: coalesced dwords, see k_queries_frame)
    {
        constexpr int W = sizeof(orbm_worldpoint_t) / 4;
        const int first = blockIdx.x * 256, cnt = max(0, min(256, m - first)) * W;
        const uint32_t *src = (const uint32_t *)(pts + (cnt > 0 ? first : 0));
        uint32_t v[W];
#pragma unroll
        for (int j = 0; j < W; j++) { const int k = threadIdx.x + 256 * j; v[j] = src[min(k, max(cnt - 1, 0))]; }
#pragma unroll
        for (int j = 0; j < W; j++) { const int k = threadIdx.x + 256 * j; if (k < cnt) recs[k] = v[j]; }
    }
    __syncthreads();
    if (i >= m) return;
    const orbm_worldpoint_t p = ((const orbm_worldpoint_t *)recs)[threadIdx.x];
    orbm_mappoint_t o;
    o.in_view = 0; o.proj_x = 0; o.proj_y = 0; o.proj_xr = 0; o.level = 0; o.view_cos = 0; o.observations = p.observations;
    if (p.valid) {
        const float pw[3] = {p.wx, p.wy, p.wz};
        float pc[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += (double)P.R[r * 3 + k] * (double)pw[k];
            pc[r] = (float)(s + (double)P.t[r]);
        }
        if (!(pc[2] < 0.0f)) {
            const float invz = 1.0f / pc[2];
            const float u = cam.fx * pc[0] * invz + cam.cx, v = cam.fy * pc[1] * invz + cam.cy;
            if (!(u < g.min_x || u > g.max_x) && !(v < g.min_y || v > g.max_y)) {
                const float maxDistance = 1.2f * p.max_distance, minDistance = 0.8f * p.min_distance;
                const float PO[3] = {pw[0] - P.Ow[0], pw[1] - P.Ow[1], pw[2] - P.Ow[2]};
                const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
                if (!(dist < minDistance || dist > maxDistance)) {
                    const double dot = (double)PO[0] * p.nx + (double)PO[1] * p.ny + (double)PO[2] * p.nz;
                    const float viewCos = (float)(dot / (double)dist);
                    if (!(viewCos < viewCosLimit)) {
                        const float ratio = p.max_distance / dist;
                        int lvl = 0;
                        for (int k = 0; k < nlevels - 1; k++) lvl += ratio >= thr[k] ? 1 : 0;
                        o.in_view = 1; o.proj_x = u; o.proj_xr = u - cam.mbf * invz; o.proj_y = v;
                        o.level = lvl; o.view_cos = viewCos;
                    }
                }
            }
        }
    }
    out[i] = o;
    if (out_host) out_host[i] = o;   // the caller's copy, written in place of a download
}
static void frustum_pose(const float *T, FrustumPose &P) {   // mRcw, mtcw, mOw = -mRcw.t()*mtcw (src/Frame.cc:272-279)
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P.R[r * 3 + c] = T[r * 4 + c]; P.t[r] = T[r * 4 + 3]; }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)P.R[k * 3 + i] * (double)P.t[k];
        P.Ow[i] = (float)(s * -1.0);
    }
}

// ---- host side: one grow-only device arena per host thread (no hipMalloc per call) with a PINNED host mirror of
// the same layout: every input is memcpy'd into the mirror and the whole used range goes up in ONE
// hipMemcpyAsync (UP ... FLUSH_UP); results come down into the mirror in one or two copies and ONE stream
// synchronisation (DOWN ... after the sync, copy out).  A call used to issue 8-10 pageable copies and 2 syncs.
struct Arena {
    uint8_t *base = nullptr, *hbase = nullptr, *hdev = nullptr; size_t cap = 0, off = 0, up_lo = 0, up_hi = 0; int device = -1; hipStream_t st = nullptr;
    int32_t *hflag = nullptr, *dflag = nullptr; int seq = 0;   // completion word of the last kernel of a call (coherent pinned memory) + the call counter
};
static thread_local Arena g_ar;
static int arena_begin(int device, size_t need) {
    ORBX_HIP(hipSetDevice(device));
    if (g_ar.device != device || g_ar.cap < need) {
        if (g_ar.base) { hipSetDevice(g_ar.device >= 0 ? g_ar.device : device); hipFree(g_ar.base); hipSetDevice(device); }
        if (g_ar.hbase) { hipHostFree(g_ar.hbase); g_ar.hbase = nullptr; }
        if (!g_ar.st || g_ar.device != device) {
            if (g_ar.st) hipStreamDestroy(g_ar.st);
            ORBX_HIP(hipStreamCreateWithFlags(&g_ar.st, hipStreamNonBlocking));
        }
        g_ar.base = nullptr; g_ar.cap = 0;
        const size_t cap = std::max(need * 2, (size_t)4 << 20);
        ORBX_HIP(hipMalloc(&g_ar.base, cap));
        ORBX_HIP(hipHostMalloc((void **)&g_ar.hbase, cap, hipHostMallocDefault));
        ORBX_HIP(hipHostGetDevicePointer((void **)&g_ar.hdev, g_ar.hbase, 0));   // the mirror as the kernels see it (small inputs are read, results written, in place)
        if (!g_ar.hflag) {
            ORBX_HIP(hipHostMalloc((void **)&g_ar.hflag, 64, hipHostMallocMapped | hipHostMallocCoherent));
            *g_ar.hflag = 0;
            ORBX_HIP(hipHostGetDevicePointer((void **)&g_ar.dflag, g_ar.hflag, 0));
        }
        g_ar.cap = cap; g_ar.device = device;
    }
    g_ar.off = 0;
    g_ar.up_lo = g_ar.cap; g_ar.up_hi = 0;
    return ORBX_OK;
}
void orbx_internal_release_arena() {
    if (g_ar.device < 0) return;
    hipSetDevice(g_ar.device);
    if (g_ar.st) { hipStreamSynchronize(g_ar.st); hipStreamDestroy(g_ar.st); g_ar.st = nullptr; }
    if (g_ar.base) hipFree(g_ar.base);
    if (g_ar.hbase) hipHostFree(g_ar.hbase);
    if (g_ar.hflag) hipHostFree(g_ar.hflag);
    g_ar.hflag = nullptr; g_ar.dflag = nullptr;
    g_ar.base = nullptr; g_ar.hbase = nullptr; g_ar.hdev = nullptr; g_ar.cap = 0; g_ar.device = -1;
}
template <typename T> static T *arena_get(size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T *p = (T *)(g_ar.base + g_ar.off);
    g_ar.off += bytes;
    return p;
}
static inline void arena_stage(void *dst, const void *src, size_t bytes) {
    const size_t o = (size_t)((uint8_t *)dst - g_ar.base);
    memcpy(g_ar.hbase + o, src, bytes);
    g_ar.up_lo = std::min(g_ar.up_lo, o);
    g_ar.up_hi = std::max(g_ar.up_hi, o + bytes);
}
#define UP(dst, src, count) arena_stage((dst), (src), sizeof(*(dst)) * (size_t)(count))
// A SMALL input that one kernel reads once: written into the pinned mirror and read there by the kernel (its device-visible address
// is returned) - no copy command in front of the first kernel (a call's upload was one ~10-us copy plus ~9 us until the kernel
// behind it started: the copy engine and the compute queue hand over through a signal)
template <typename T> static T *arena_zc(T *dev, const void *src, size_t count) {
    memcpy(g_ar.hbase + ((uint8_t *)dev - g_ar.base), src, sizeof(T) * count);
    return (T *)(g_ar.hdev + ((uint8_t *)dev - g_ar.base));
}
#define ZC(dst, src, count) arena_zc((dst), (src), (size_t)(count))
// everything staged since the last flush goes up in one copy (call before the first kernel that reads it)
#define FLUSH_UP()                                                                                                        \
    do {                                                                                                                  \
        if (g_ar.up_hi > g_ar.up_lo)                                                                                      \
            ORBX_HIP(hipMemcpyAsync(g_ar.base + g_ar.up_lo, g_ar.hbase + g_ar.up_lo, g_ar.up_hi - g_ar.up_lo,              \
                                    hipMemcpyHostToDevice, st));                                                          \
        g_ar.up_lo = g_ar.cap; g_ar.up_hi = 0;                                                                            \
        (void)hipGetLastError();                                                                                          \
    } while (0)
// device -> pinned mirror (same offset); valid after the stream is synchronised
template <typename T> static const T *arena_host(const T *dev) { return (const T *)(g_ar.hbase + ((const uint8_t *)dev - g_ar.base)); }
#define DOWN(dev, count) ORBX_HIP(hipMemcpyAsync((void *)arena_host(dev), (dev), sizeof(*(dev)) * (size_t)(count), hipMemcpyDeviceToHost, st))
// the pinned mirror of a device allocation as a kernel addresses it: results a kernel stores there are on the host when the stream
// is synchronised (no copy command, no copy kernel: a call's download used to be two blit kernels of ~5 us each behind the resolver)
template <typename T> static T *arena_hostdev(T *dev) { return (T *)(g_ar.hdev + ((uint8_t *)dev - g_ar.base)); }
#define ORBX_FAST_FALLBACK 1  // positive: not an error, the caller runs the exact legacy kernel
extern thread_local int t_matchResolver;   // ORBM_OPT_RESOLVER (orbx_match.hip)
extern thread_local int t_matchStreamSync; // ORBM_OPT_STREAM_SYNC (orbx_match.hip): 1 = always wait for the stream, never poll the completion word
// Wait for the call's last kernel: poll the completion word it stores behind its results (k_resolve_par), fall back to the stream
// synchronisation if it does not show up within a few milliseconds (or when the option says so, or the kernel has no such word).
static int arena_wait(hipStream_t st, int seq) {
    if (seq > 0 && !t_matchStreamSync) {
        for (int i = 0; i < 200000; i++) {
            if (__atomic_load_n(g_ar.hflag, __ATOMIC_ACQUIRE) == seq) return ORBX_OK;
            __builtin_ia32_pause();
        }
    }
    ORBX_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}
static inline int resolve_par_q(int m) { return m <= 2 * RP_T ? 2 : 4; }
static inline size_t resolve_par_lds(int m, int n) { return sizeof(uint32_t) * ((size_t)QK * RP_T * resolve_par_q(m) + 2 * (size_t)n + (size_t)(n + 31) / 32); }
static inline bool use_resolve_par(int m, int n) { return t_matchResolver == 0 && m <= RP_T * RP_QMAX && n <= 30000 && resolve_par_lds(m, n) <= 158 * 1024; }
#define RESOLVE_PAR_LAUNCH(MODE, M_, ...)                                                                                   \
    do {                                                                                                                  \
        const size_t lds_ = resolve_par_lds((M_), n);                                                                     \
        if (resolve_par_q(M_) == 2) {                                                                                     \
            if (lds_ > 48 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_resolve_par<MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
            hipLaunchKernelGGL((k_resolve_par<MODE, 2>), dim3(1), dim3(RP_T), lds_, st, __VA_ARGS__);                     \
        } else {                                                                                                          \
            if (lds_ > 48 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_resolve_par<MODE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
            hipLaunchKernelGGL((k_resolve_par<MODE, 4>), dim3(1), dim3(RP_T), lds_, st, __VA_ARGS__);                     \
        }                                                                                                                 \
    } while (0)

// Returns ORBX_OK (results written), ORBX_FAST_FALLBACK, or a negative error.
int fast_search_for_initialization(const orbx_keypoint_t *k1, const uint8_t *d1, int n1, const orbx_keypoint_t *k2,
                                   const uint8_t *d2, int n2, const orbm_grid_geom_t *g2, float *prev, int32_t *m12,
                                   int window, float nnratio, int check_ori, int device, int *nmatches) {
    if (n2 > 7000 || n1 > 65535) return ORBX_FAST_FALLBACK;  // LDS plan of k_resolve_init (9 B per F2 keypoint, 64 KB)
    const size_t need = (size_t)(n1 + n2) * (28 + 32 + 64 + 16) + (size_t)n1 * (CAND_CAP * 8 + 64) + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    hipStream_t st = g_ar.st;
    orbx_keypoint_t *dk1 = arena_get<orbx_keypoint_t>(n1), *dk2 = arena_get<orbx_keypoint_t>(n2);
    uint8_t *dd1 = arena_get<uint8_t>((size_t)32 * n1), *dd2 = arena_get<uint8_t>((size_t)32 * n2);
    float *dprev = arena_get<float>(2 * (size_t)n1);
    int32_t *dm12 = arena_get<int32_t>(n1), *dbin = arena_get<int32_t>(n1), *dnc = arena_get<int32_t>(n1), *dout = arena_get<int32_t>(4);
    uint4 *dckp = arena_get<uint4>(n2);
    GQuery *dq = arena_get<GQuery>(n1);
    u64 *dkeys = arena_get<u64>((size_t)n1 * CAND_CAP);
    UP(dk1, k1, n1); UP(dk2, k2, n2); UP(dd1, d1, (size_t)32 * n1); UP(dd2, d2, (size_t)32 * n2); UP(dprev, prev, 2 * (size_t)n1);
    FLUSH_UP();
    hipLaunchKernelGGL(k_compact_kps, dim3((n2 + 255) / 256), dim3(256), 0, st, dk2, n2, *g2, dckp);
    hipLaunchKernelGGL(k_queries_init, dim3((n1 + 255) / 256), dim3(256), 0, st, dk1, dprev, n1, window, dq);
    hipLaunchKernelGGL(k_cand<true>, dim3(std::max(n1, 1)), dim3(256), 0, st, dq, dd1, n1, dckp, dd2, n2, *g2, dkeys, dnc);
    hipLaunchKernelGGL(k_resolve_init, dim3(1), dim3(64), sizeof(int32_t) * 2 * (size_t)n2 + (size_t)((n2 + 15) & ~15), st, dkeys, dnc, dk1, dk2, n1, n2,
                       dprev, dm12, dbin, nnratio, check_ori, dout);
    ORBX_HIP(hipGetLastError());
    DOWN(dout, 2); DOWN(dprev, 2 * (size_t)n1); DOWN(dm12, n1);
    ORBX_HIP(hipStreamSynchronize(st));
    const int32_t *out = arena_host(dout);
    if (out[1]) return ORBX_FAST_FALLBACK;
    memcpy(prev, arena_host(dprev), sizeof(float) * 2 * (size_t)n1);
    memcpy(m12, arena_host(dm12), sizeof(int32_t) * (size_t)n1);
    *nmatches = out[0];
    return ORBX_OK;
}

int fast_is_in_frustum(const orbm_worldpoint_t *pts, int m, const float *Tcw16, const orbm_camera_t *cam,
                       const orbm_grid_geom_t *g, float viewCosLimit, const float *thr, int nlevels, orbm_mappoint_t *out,
                       int device) {
    int rc = arena_begin(device, (size_t)m * (sizeof(orbm_worldpoint_t) + sizeof(orbm_mappoint_t)) + 65536);
    if (rc) return rc;
    hipStream_t st = g_ar.st;
    orbm_worldpoint_t *dw = arena_get<orbm_worldpoint_t>(m);
    orbm_mappoint_t *dmp = arena_get<orbm_mappoint_t>(m);
    float *dthr = arena_get<float>(nlevels);
    UP(dw, pts, m);
    if (nlevels > 1) UP(dthr, thr, nlevels - 1);
    FrustumPose P;
    frustum_pose(Tcw16, P);
    FLUSH_UP();
    hipLaunchKernelGGL(k_frustum, dim3((m + 255) / 256), dim3(256), 0, st, dw, m, P, *cam, *g, viewCosLimit, dthr, nlevels, dmp, arena_hostdev(dmp));
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(out, arena_host(dmp), sizeof(orbm_mappoint_t) * (size_t)m);
    return ORBX_OK;
}

// world != NULL: the map-point records are produced on the device by k_frustum (no upload of projections);
// proj_out (host, may be NULL) then receives them — always before a fallback is reported.
int fast_search_by_projection_mp(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                 const orbm_grid_geom_t *g, const float *sf, int nlevels, const orbm_mappoint_t *mps,
                                 const uint8_t *mp_desc, int m, int32_t *frame_mp, const int32_t *ext_obs, float th,
                                 float nnratio, int device, int *nmatches, const FrustumArgs *world, orbm_mappoint_t *proj_out,
                                 const DevFrame *dev) {
    if (n > 30000 && !world) return ORBX_FAST_FALLBACK;
    const size_t need = (size_t)n * (28 + 32 + 32 + 16) + (size_t)m * (28 + 32 + QK * 8 + 64 + sizeof(orbm_worldpoint_t)) + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    int waitSeq = 0;   // > 0: the call's last kernel stores this number into the completion word (arena_wait)
    // dev: the frame's keypoints / descriptors / mvuRight are already in HBM (outputs of orbx_extract_batch_device and
    // orbm_stereo_batch_device) and the kernels run on the caller's stream, behind the kernels that produce them
    hipStream_t st = dev ? dev->stream : g_ar.st;
    orbx_keypoint_t *dk = dev ? const_cast<orbx_keypoint_t *>(kun) : arena_get<orbx_keypoint_t>(n);
    uint8_t *dd = dev ? const_cast<uint8_t *>(desc) : arena_get<uint8_t>((size_t)32 * n);
    uint8_t *dmd = arena_get<uint8_t>((size_t)32 * m);
    float *du = dev ? const_cast<float *>(uright) : arena_get<float>(n), *dsf = arena_get<float>(nlevels);
    orbm_mappoint_t *dmp = arena_get<orbm_mappoint_t>(m);
    int32_t *dfm = arena_get<int32_t>(n), *deo = arena_get<int32_t>(n), *dnc = arena_get<int32_t>(m), *dout = arena_get<int32_t>(4);
    GQuery *dq = arena_get<GQuery>(m);
    u64 *dkeys = arena_get<u64>((size_t)m * QK);
    if (!dev) { UP(dk, kun, n); UP(dd, desc, (size_t)32 * n); UP(du, uright, n); }
    // small inputs are read by the kernels in the pinned mirror (zero copy); with a device-resident frame nothing is copied up at all
    const float *zsf = ZC(dsf, sf, nlevels);
    const uint8_t *zmd = ZC(dmd, mp_desc, (size_t)32 * m);
    int32_t *zfm = ZC(dfm, frame_mp, n);
    const int32_t *zeo = ext_obs ? ZC(deo, ext_obs, n) : (const int32_t *)nullptr;
    if (world) {
        orbm_worldpoint_t *dw = arena_get<orbm_worldpoint_t>(m);
        float *dthr = arena_get<float>(nlevels);
        const orbm_worldpoint_t *zw = ZC(dw, world->pts, m);
        const float *zthr = nlevels > 1 ? ZC(dthr, world->thr, nlevels - 1) : dthr;
        FrustumPose P;
        frustum_pose(world->Tcw16, P);
        FLUSH_UP();
        hipLaunchKernelGGL(k_frustum, dim3((m + 255) / 256), dim3(256), 0, st, zw, m, P, *world->cam, *g, world->viewCosLimit, zthr,
                           nlevels, dmp, proj_out ? arena_hostdev(dmp) : (orbm_mappoint_t *)nullptr);
        if (n > 30000) {
            ORBX_HIP(hipStreamSynchronize(st));
            if (proj_out) memcpy(proj_out, arena_host(dmp), sizeof(orbm_mappoint_t) * (size_t)m);
            return ORBX_FAST_FALLBACK;
        }
    } else { UP(dmp, mps, m); FLUSH_UP(); }
    const int mx = std::max(n, m);
    uint4 *dckp = arena_get<uint4>(n);
    hipLaunchKernelGGL(k_queries_mp, dim3((mx + 255) / 256), dim3(256), 0, st, dmp, m, zsf, th, dq, zfm, zeo, n, dk, du, *g, dckp);
    hipLaunchKernelGGL(k_cand<false>, dim3(std::max(m, 1)), dim3(256), 0, st, dq, zmd, m, dckp, dd, n, *g, dkeys, dnc);
    if (use_resolve_par(m, n)) {   // results land in the pinned mirror straight from the kernel
        waitSeq = ++g_ar.seq;
        RESOLVE_PAR_LAUNCH(0, m, dkeys, dnc, (const void *)dmp, dk, m, n, zfm, zfm, (int32_t *)nullptr, nnratio, 0, 0, dout, arena_hostdev(dout), g_ar.dflag, waitSeq);
        ORBX_HIP(hipGetLastError());
    } else {
        ORBX_HIP(hipMemcpyAsync(dfm, arena_host(dfm), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));   // this resolver works on the device copy
        hipLaunchKernelGGL(k_resolve_mp, dim3(1), dim3(64), 2 * (size_t)((n + 15) & ~15), st, dkeys, dnc, dmp, m, n, dfm, nnratio, dout);
        ORBX_HIP(hipGetLastError());
        DOWN(dout, 2); DOWN(dfm, n);
    }
    rc = arena_wait(st, waitSeq);
    if (rc) return rc;
    if (world && proj_out) memcpy(proj_out, arena_host(dmp), sizeof(orbm_mappoint_t) * (size_t)m);
    const int32_t *out = arena_host(dout);
    if (getenv("ORBX_TRACE_RESOLVE")) fprintf(stderr, "k_resolve_par<mp>: %d queries, %d keypoints, %d rounds; developer build: loads %.1f us, rounds %.1f us, epilogue %.1f us\n", m, n, out[2], (out[3] >> 20) / 10.0, ((out[3] >> 10) & 1023) / 10.0, (out[3] & 1023) / 10.0);
    if (out[1]) return ORBX_FAST_FALLBACK;
    memcpy(frame_mp, arena_host(dfm), sizeof(int32_t) * (size_t)n);
    *nmatches = out[0];
    return ORBX_OK;
}

int fast_search_by_projection_frame(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                                    const orbm_grid_geom_t *g, const float *sf, int nlevels, const orbm_camera_t *cam,
                                    const float *Tc16, const float *Tl16, const orbm_lastpoint_t *last,
                                    const uint8_t *last_desc, int nlast, int32_t *cur_mp, const int32_t *ext_obs,
                                    float th, int mono, int check_ori, int device, int *nmatches, const DevFrame *dev) {
    if (n > 30000) return ORBX_FAST_FALLBACK;
    static const bool traceHost = getenv("ORBX_TRACE_HOST") != nullptr;   // developer aid: host-side time stamps of this call on stderr
    timespec ts0, ts1, ts2, ts3;
    if (traceHost) clock_gettime(CLOCK_MONOTONIC, &ts0);
    const size_t need = (size_t)n * (28 + 32 + 32 + 16) + (size_t)nlast * (28 + 32 + QK * 8 + 64 + 8) + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    int waitSeq = 0;
    hipStream_t st = dev ? dev->stream : g_ar.st;   // dev: kun / desc / uright / last_desc are device arrays (see fast_search_by_projection_mp)
    orbx_keypoint_t *dk = dev ? const_cast<orbx_keypoint_t *>(kun) : arena_get<orbx_keypoint_t>(n);
    uint8_t *dd = dev ? const_cast<uint8_t *>(desc) : arena_get<uint8_t>((size_t)32 * n);
    uint8_t *dld = dev ? const_cast<uint8_t *>(last_desc) : arena_get<uint8_t>((size_t)32 * nlast);
    float *du = dev ? const_cast<float *>(uright) : arena_get<float>(n), *dsf = arena_get<float>(nlevels), *dT = arena_get<float>(32);
    orbm_lastpoint_t *dl = arena_get<orbm_lastpoint_t>(nlast);
    int32_t *dcm = arena_get<int32_t>(n), *deo = arena_get<int32_t>(n), *dnc = arena_get<int32_t>(nlast), *dout = arena_get<int32_t>(4);
    int32_t *dhi = arena_get<int32_t>(nlast), *dhb = arena_get<int32_t>(nlast);
    GQuery *dq = arena_get<GQuery>(nlast);
    u64 *dkeys = arena_get<u64>((size_t)nlast * QK);
    float T2[32];
    memcpy(T2, Tc16, 64); memcpy(T2 + 16, Tl16, 64);
    if (!dev) { UP(dk, kun, n); UP(dd, desc, (size_t)32 * n); UP(du, uright, n); UP(dld, last_desc, (size_t)32 * nlast); }
    // small inputs: read by the kernels in the pinned mirror (see ZC)
    const float *zsf = ZC(dsf, sf, nlevels), *zT = ZC(dT, T2, 32);
    const orbm_lastpoint_t *zl = ZC(dl, last, nlast);
    int32_t *zcm = ZC(dcm, cur_mp, n);
    const int32_t *zeo = ext_obs ? ZC(deo, ext_obs, n) : (const int32_t *)nullptr;
    FLUSH_UP();
    const int mx = std::max(n, nlast);
    float2 *dqm = arena_get<float2>(nlast);
    uint4 *dckp = arena_get<uint4>(n);
    if (traceHost) clock_gettime(CLOCK_MONOTONIC, &ts1);
    hipLaunchKernelGGL(k_queries_frame, dim3((mx + 255) / 256), dim3(256), 0, st, zl, nlast, zsf, *cam, *g, zT, zT + 16, th,
                       mono, dq, zcm, zeo, n, dk, du, dckp, dqm);
    hipLaunchKernelGGL(k_cand<false>, dim3(std::max(nlast, 1)), dim3(256), 0, st, dq, dld, nlast, dckp, dd, n, *g, dkeys, dnc);
    if (use_resolve_par(nlast, n)) {
        waitSeq = ++g_ar.seq;
        RESOLVE_PAR_LAUNCH(1, nlast, dkeys, dnc, (const void *)dqm, dk, nlast, n, zcm, zcm, (int32_t *)nullptr, 0.0f, 0, check_ori, dout, arena_hostdev(dout), g_ar.dflag, waitSeq);
        ORBX_HIP(hipGetLastError());
    } else {
        ORBX_HIP(hipMemcpyAsync(dcm, arena_host(dcm), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        ORBX_HIP(hipMemcpyAsync(dl, arena_host(dl), sizeof(orbm_lastpoint_t) * (size_t)nlast, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_resolve_frame, dim3(1), dim3(64), 2 * (size_t)((n + 15) & ~15), st, dkeys, dnc, dl, dk, nlast, n, dcm,
                           dhi, dhb, check_ori, dout);
        ORBX_HIP(hipGetLastError());
        DOWN(dout, 2); DOWN(dcm, n);
    }
    if (traceHost) clock_gettime(CLOCK_MONOTONIC, &ts2);
    rc = arena_wait(st, waitSeq);
    if (rc) return rc;
    if (traceHost) {
        clock_gettime(CLOCK_MONOTONIC, &ts3);
        auto us = [](const timespec &a, const timespec &b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; };
        fprintf(stderr, "search_by_projection_frame host: stage inputs %.1f us, enqueue %.1f us, wait %.1f us\n", us(ts0, ts1), us(ts1, ts2), us(ts2, ts3));
    }
    const int32_t *out = arena_host(dout);
    if (getenv("ORBX_TRACE_RESOLVE")) fprintf(stderr, "k_resolve_par<frame>: %d queries, %d keypoints, %d rounds; developer build: loads %.1f us, rounds %.1f us, epilogue %.1f us\n", nlast, n, out[2], (out[3] >> 20) / 10.0, ((out[3] >> 10) & 1023) / 10.0, (out[3] & 1023) / 10.0);
    if (out[1]) return ORBX_FAST_FALLBACK;
    memcpy(cur_mp, arena_host(dcm), sizeof(int32_t) * (size_t)n);
    *nmatches = out[0];
    return ORBX_OK;
}

int fast_match_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                       const orbm_grid_geom_t *g, const orbm_grid_geom_t *ga, const orbm_window_query_t *q,
                       const uint8_t *qdesc, int m,
                       int32_t *holder, const int32_t *ext_blocks, int max_dist, int check_ori, int device, int *nmatches) {
    if (n > 30000) return ORBX_FAST_FALLBACK;
    const size_t need = (size_t)n * (28 + 32 + 32 + 16) + (size_t)m * (40 + 32 + QK * 8 + 96) + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    hipStream_t st = g_ar.st;
    orbx_keypoint_t *dk = arena_get<orbx_keypoint_t>(n);
    uint8_t *dd = arena_get<uint8_t>((size_t)32 * n), *dqd = arena_get<uint8_t>((size_t)32 * m);
    float *du = arena_get<float>(n);
    orbm_window_query_t *dw = arena_get<orbm_window_query_t>(m);
    int32_t *dh = arena_get<int32_t>(n), *deb = arena_get<int32_t>(n), *dnc = arena_get<int32_t>(m), *dout = arena_get<int32_t>(4);
    int32_t *dhi = arena_get<int32_t>(m), *dhb = arena_get<int32_t>(m);
    GQuery *dq = arena_get<GQuery>(m);
    u64 *dkeys = arena_get<u64>((size_t)m * QK);
    UP(dk, kun, n); UP(dd, desc, (size_t)32 * n); UP(dw, q, m); UP(dqd, qdesc, (size_t)32 * m); UP(dh, holder, n);
    if (uright) UP(du, uright, n);
    else {   // 0: no stereo coordinate, never gated (staged like the other inputs: the flush covers the whole range)
        const size_t o = (size_t)((uint8_t *)du - g_ar.base);
        memset(g_ar.hbase + o, 0, sizeof(float) * (size_t)n);
        g_ar.up_lo = std::min(g_ar.up_lo, o); g_ar.up_hi = std::max(g_ar.up_hi, o + sizeof(float) * (size_t)n);
    }
    if (ext_blocks) UP(deb, ext_blocks, n);
    FLUSH_UP();
    const int mx = std::max(n, m);
    uint4 *dckp = arena_get<uint4>(n);
    hipLaunchKernelGGL(k_queries_windows, dim3((mx + 255) / 256), dim3(256), 0, st, dw, m, dq, dh,
                       ext_blocks ? deb : (const int32_t *)nullptr, n, dk, du, *ga, dckp);
    hipLaunchKernelGGL(k_cand<false>, dim3(std::max(m, 1)), dim3(256), 0, st, dq, dqd, m, dckp, dd, n, *g, dkeys, dnc);
    if (use_resolve_par(m, n)) {
        RESOLVE_PAR_LAUNCH(2, m, dkeys, dnc, (const void *)dw, dk, m, n, dh, dh, arena_hostdev(dh), 0.0f, max_dist, check_ori, dout, arena_hostdev(dout), (int32_t *)nullptr, 0);
        ORBX_HIP(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_resolve_windows, dim3(1), dim3(64), 2 * (size_t)((n + 15) & ~15), st, dkeys, dnc, dw, dk, m, n, dh, dhi,
                           dhb, max_dist, check_ori, dout);
        ORBX_HIP(hipGetLastError());
        DOWN(dout, 2); DOWN(dh, n);
    }
    ORBX_HIP(hipStreamSynchronize(st));
    const int32_t *out = arena_host(dout);
    if (out[1]) return ORBX_FAST_FALLBACK;
    memcpy(holder, arena_host(dh), sizeof(int32_t) * (size_t)n);
    *nmatches = out[0];
    return ORBX_OK;
}

// ---- C. stateless window search (Fuse x2, SearchBySim3: src/ORBmatcher.cc:827-1328).  No query
// reads what an earlier one wrote, so this is k_cand reduced to its minimum: one wave per query,
// every lane keeps the smallest scan-order key of its keypoints, one wave-min at the end.
// GATE adds Fuse's per-candidate reprojection test (:916-940).
template <bool GATE>
__global__ __launch_bounds__(256) void k_best(const orbm_window_query_t *__restrict__ qs, const uint8_t *__restrict__ qdesc,
                                              int m, const orbx_keypoint_t *__restrict__ kps,
                                              const uint8_t *__restrict__ desc, const float *__restrict__ uright,
                                              const float *__restrict__ inv_sigma2, int nlevels,
                                              const uint16_t *__restrict__ code, int n, orbm_grid_geom_t g,
                                              int32_t *__restrict__ best_idx, int32_t *__restrict__ best_dist) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= m) return;
    const orbm_window_query_t Q = qs[qi];
    u64 best = ~0ull;
    if (Q.valid) {
        const AreaQuery aq = make_query(g, Q.u, Q.v, Q.radius, -1, -1);   // KeyFrame::GetFeaturesInArea: no levels
        if (!aq.empty) {
            const Desc256 da = load_desc(qdesc + (size_t)qi * 32);
            for (int j = lane; j < n; j += 64) {
                const unsigned c = code[j];
                const orbx_keypoint_t kp = kps[j];
                if (!in_area(aq, c, kp)) continue;
                if (kp.octave < Q.min_level || kp.octave > Q.max_level) continue;  // :913 / :1069 / :1209
                if (GATE) {
                    const int lvl = min(max(kp.octave, 0), nlevels - 1);
                    const float ex = Q.u - kp.x, ey = Q.v - kp.y;
                    const float kr = uright ? uright[j] : -1.0f;
                    if (kr >= 0) {                                      // :916-929
                        const float er = Q.ur_c - kr;
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * inv_sigma2[lvl]) > 7.8) continue;
                    } else {                                            // :930-940
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * inv_sigma2[lvl]) > 5.99) continue;
                    }
                }
                const u64 key = fast_key(ham(da, load_desc(desc + (size_t)j * 32)), c, j, kp.octave);
                best = key < best ? key : best;
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        best_idx[qi] = best == ~0ull ? -1 : KEY_IDX(best);
        best_dist[qi] = best == ~0ull ? 256 : KEY_DIST(best);
    }
}

int fast_best_in_windows(const orbx_keypoint_t *kun, const uint8_t *desc, const float *uright, int n,
                         const orbm_grid_geom_t *g, const orbm_grid_geom_t *ga, const orbm_window_query_t *q,
                         const uint8_t *qdesc, int m,
                         const float *inv_sigma2, int nlevels, int32_t *best_idx, int32_t *best_dist, int device) {
    const size_t need = (size_t)n * (28 + 32 + 16) + (size_t)m * (40 + 32 + 16) + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    hipStream_t st = g_ar.st;
    orbx_keypoint_t *dk = arena_get<orbx_keypoint_t>(n);
    uint8_t *dd = arena_get<uint8_t>((size_t)32 * n), *dqd = arena_get<uint8_t>((size_t)32 * m);
    float *du = arena_get<float>(n), *dis = arena_get<float>(nlevels > 0 ? nlevels : 1);
    orbm_window_query_t *dw = arena_get<orbm_window_query_t>(m);
    int32_t *dbi = arena_get<int32_t>(m), *dbd = arena_get<int32_t>(m);
    uint16_t *dcode = arena_get<uint16_t>(n);
    UP(dk, kun, n); UP(dd, desc, (size_t)32 * n); UP(dw, q, m); UP(dqd, qdesc, (size_t)32 * m);
    if (uright) UP(du, uright, n);
    if (inv_sigma2) UP(dis, inv_sigma2, nlevels);
    FLUSH_UP();
    hipLaunchKernelGGL(k_cell_codes, dim3((n + 255) / 256), dim3(256), 0, st, dk, n, *ga, dcode);
    if (inv_sigma2)
        hipLaunchKernelGGL(k_best<true>, dim3((m + 3) / 4), dim3(256), 0, st, dw, dqd, m, dk, dd,
                           uright ? du : (const float *)nullptr, dis, nlevels, dcode, n, *g, dbi, dbd);
    else
        hipLaunchKernelGGL(k_best<false>, dim3((m + 3) / 4), dim3(256), 0, st, dw, dqd, m, dk, dd, (const float *)nullptr,
                           (const float *)nullptr, nlevels, dcode, n, *g, dbi, dbd);
    ORBX_HIP(hipGetLastError());
    DOWN(dbi, m); DOWN(dbd, m);
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(best_idx, arena_host(dbi), sizeof(int32_t) * (size_t)m);
    memcpy(best_dist, arena_host(dbd), sizeof(int32_t) * (size_t)m);
    return ORBX_OK;
}

// ---- D. MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:252-317), one wave per map point.
// Lane j keeps the distances of the current row to rows j, j+64, ... in registers (DD_CH chunks);
// the median of a row is found by bisection on the VALUE (0..256): count(d <= mid) by ballot +
// popcount, 9 steps, no sort and no N x N matrix.  Points with more than 64*DD_CH rows recompute
// the distances in every bisection step instead of caching them.
#define DD_CH 4
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t *__restrict__ desc, const int32_t *__restrict__ offsets,
                                                     int npoints, int32_t *__restrict__ best_row,
                                                     int32_t *__restrict__ best_median) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= npoints) return;
    const int o0 = offsets[p], N = offsets[p + 1] - o0;
    if (N <= 0) { if (lane == 0) { best_row[p] = -1; if (best_median) best_median[p] = 0; } return; }
    const uint8_t *D = desc + (size_t)o0 * 32;
    const int k = (N - 1) >> 1;                       // (size_t)(0.5*(N-1))
    int bestMedian = INT_MAX, bestIdx = 0;
    const bool cached = N <= 64 * DD_CH;
    Desc256 dj[DD_CH];
    if (cached) {
#pragma unroll
        for (int c = 0; c < DD_CH; c++) dj[c] = load_desc(D + (size_t)min(c * 64 + lane, N - 1) * 32);
    }
    for (int i = 0; i < N; i++) {
        const Desc256 di = load_desc(D + (size_t)i * 32);
        int dist[DD_CH];
        if (cached) {
#pragma unroll
            for (int c = 0; c < DD_CH; c++) dist[c] = c * 64 + lane < N ? ham(di, dj[c]) : 1000;
        }
        int lo = 0, hi = 256;                         // smallest v with count(d <= v) >= k+1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            if (cached) {
#pragma unroll
                for (int c = 0; c < DD_CH; c++) cnt += __popcll(__ballot(dist[c] <= mid));
            } else {
                for (int j0 = 0; j0 < N; j0 += 64) {
                    const int j = j0 + lane;
                    const bool le = j < N && ham(di, load_desc(D + (size_t)j * 32)) <= mid;
                    cnt += __popcll(__ballot(le));
                }
            }
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        if (lo < bestMedian) { bestMedian = lo; bestIdx = i; }   // :304-308, first minimum
    }
    if (lane == 0) { best_row[p] = bestIdx; if (best_median) best_median[p] = bestMedian; }
}

int fast_distinctive_descriptors(const uint8_t *desc, const int32_t *offsets, int npoints, int32_t *best_row,
                                 int32_t *best_median, int device) {
    const int total = offsets[npoints];
    const size_t need = (size_t)total * 32 + (size_t)npoints * 16 + 65536;
    int rc = arena_begin(device, need);
    if (rc) return rc;
    hipStream_t st = g_ar.st;
    uint8_t *dd = arena_get<uint8_t>((size_t)32 * (total > 0 ? total : 1));
    int32_t *doff = arena_get<int32_t>(npoints + 1), *dbr = arena_get<int32_t>(npoints), *dbm = arena_get<int32_t>(npoints);
    if (total > 0) UP(dd, desc, (size_t)32 * total);
    UP(doff, offsets, npoints + 1);
    FLUSH_UP();
    hipLaunchKernelGGL(k_distinctive, dim3((npoints + 3) / 4), dim3(256), 0, st, dd, doff, npoints, dbr, dbm);
    ORBX_HIP(hipGetLastError());
    DOWN(dbr, npoints);
    if (best_median) DOWN(dbm, npoints);
    ORBX_HIP(hipStreamSynchronize(st));
    memcpy(best_row, arena_host(dbr), sizeof(int32_t) * (size_t)npoints);
    if (best_median) memcpy(best_median, arena_host(dbm), sizeof(int32_t) * (size_t)npoints);
    return ORBX_OK;
}
