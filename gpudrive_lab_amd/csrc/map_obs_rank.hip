// Road observations in the reference's row order, fast path: the heap history replayed on 16-bit RANKS.
//
// selectKNearestRoadEntities (reference src/knn.hpp:103-158) pushes every road that is closer than the running K-th
// distance through an SGI binary heap (src/binary_heap.hpp) and returns the heap ARRAY, so the output order depends
// on every insert that ever happened.  map_obs.hip replays that history on fp32 keys + u16 indices (1200 B of LDS per
// agent: 128 agents per CU, one lone wave per SIMD, two generations of workgroups at 1024 x 64).  The heap's
// behaviour depends only on the OUTCOME of key comparisons, so this file replays it on ranks:
//
//   k_knn_scan +  The agent's CANDIDATES (scan: one lane per agent over every road of the world; rank: one wave per agent
//   k_knn_rank    over its candidates): roads 0..K-1 plus every later
//                 road whose key is below a bound that provably is not smaller than the K-th distance the heap holds when
//                 the reference reaches that road.  The bound comes from the previous selection of the same agent: the
//                 K-th smallest distance over a fixed prefix of roads is 1-Lipschitz in the agent's position, and the
//                 previous replay recorded it at checkpoints (every 32 candidates).  A road that is not a candidate fails
//                 the reference's `cmp(current, heap[0])` test (src/knn.hpp:138-143) and never touches the heap, so
//                 replaying the candidates alone reproduces the heap exactly; candidates that are not inserts fail the
//                 same test in the replay.  The candidates' exact keys (gd_math.hpp ego_dist2, the reference's
//                 arithmetic) are ranked: e = (number of candidates with a smaller key + 1) << 5 | (index among the
//                 candidates with an EQUAL key).  key_a < key_b  <=>  (e_a >> 5) < (e_b >> 5), equal keys compare
//                 equal like in the reference, and e still names the candidate.  400 B of heap per agent instead of 1200.
//   k_knn_replay  one lane per agent, 64 agents per wave, the heap as u16 pairs in LDS: make_heap, then
//                 pop_heap / push_heap per insert, statement for statement the reference's algorithm.
//   k_knn_finish  one wave per agent: ranks back to road indices, radiusFilter (src/knn.hpp:83-97), the checkpoints'
//                 K-th distances for the next step, hand-over to k_map_rows.
//
// An agent without a usable bound (first selection after the worlds were built, a logged agent that reappears somewhere else)
// borrows a neighbour's checkpoints or is bounded afresh inside k_knn_scan; one with more candidates than the standard
// ranking holds (1272) is ranked by the long-list instantiation (2552).  Only an agent beyond that, with more equal keys than
// a rank can count (32; long lists 16), or in a world with fewer than K roads (or below rk_min_roads) raises the fallback flag
// of its group of 32 agents; k_map_obs (map_obs.hip) then selects for that group as before -- the same rows by construction
// -- and records checkpoints, so the group is back on this path at the next step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "engine.hpp"
#include "gd_math.hpp"

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr int CAP = GD_RANK_CAP;   // candidates per agent
constexpr int NCP = GD_RANK_NCP;   // checkpoints per agent
constexpr int SPL = GD_RANK_SPL;   // sorted slots handed to k_knn_finish
constexpr int KT = GD_RANK_KT;     // key table entries per agent
constexpr int TILE = 32;           // candidates between checkpoints (standard agents; long lists: 64)
// Two instantiations of the ranking.  The standard one takes up to CAP - 8 = 1272 candidates in 10 KB of LDS (sixteen waves
// per CU) and serves nearly every agent; an agent with more (unreduced Waymo polylines: 200 ln(R / 200) inserts and the
// superset on top, one agent in eight beyond 6,000 roads; fast agents late in an episode) is put on a list and ranked by
// the LONG instantiation afterwards: 2552 candidates, 20 KB, ranks of 12 + 4 bits instead of 11 + 5, a checkpoint every 64
// candidates so that their number still fits the 40 slots.  Round 3 sent such an agent's whole group of 32 to the history
// replay on keys (k_map_obs), at least a millisecond for the launch, and therefore kept worlds above 6,000 roads off the
// rank path altogether.
constexpr int CAP_LONG = GD_RANK_CAP_LONG;
template <int CAP_T>
struct RankGeo {
    static constexpr bool LONG = CAP_T > CAP;
    static constexpr int NG = CAP_T / 64;          // candidates per lane
    static constexpr int NMAX = CAP_T - 8;         // most candidates an agent is ranked with (room for the end markers)
    static constexpr int NB = CAP_T;               // ranking buckets
    static constexpr int NLIN = CAP_T * 3 / 5;     // of which linear in the key (up to 1.5 x the previous K-th key); the others take
    static constexpr int TSH = LONG ? 6 : 5;       // log2 (candidates between checkpoints)
    static constexpr int RSH = LONG ? 4 : 5;       // bits of a rank that count the equal keys before it
    static constexpr int KTN = LONG ? GD_RANK_KT_LONG : GD_RANK_KT;  // floats per key-table row
};
constexpr int NB = RankGeo<CAP>::NB;
                                   // the eight octaves above that, 64 each: about one candidate per bucket on either side
static_assert(K + (NCP - 1) * TILE >= CAP && K + (NCP - 1) * 64 >= CAP_LONG, "a checkpoint slot for every tile of candidates");
static_assert(CAP % 64 == 0 && NB % 128 == 0 && CAP < 2047 && CAP_LONG % 640 == 0 && CAP_LONG < 4095, "geometry; less + 1 fits 11 / 12 bits");
static_assert(SPL >= K + 31 && SPL <= CAP && KT >= CAP / 16 + 1, "every slot the finish looks up; an entry per 16 slots + the last");
constexpr int RK_FAR = 1 << 30;    // rk_n: no road of the world can be within the agent's radius
constexpr int RK_TIES = 1 << 30;   // rk_ticket: the agent took its place in the replay order, then fell back (equal keys)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive prefix sum over the 64 lanes (DPP row shifts and row broadcasts)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Wave-wide max / min / sum on the same DPP steps (lane 63 of the inclusive scan, broadcast): a `__shfl_xor` butterfly is six
// dependent LDS-crossbar round trips, this is six vector instructions and a readlane.  Lanes without a source keep their own
// value, which is neutral for max and min.
#define GD_DPP_SELF(v, ctrl, rows) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, GD_DPP_SELF(v, 0x111, 0xf));
    v = max(v, GD_DPP_SELF(v, 0x112, 0xf));
    v = max(v, GD_DPP_SELF(v, 0x114, 0xf));
    v = max(v, GD_DPP_SELF(v, 0x118, 0xf));
    v = max(v, GD_DPP_SELF(v, 0x142, 0xa));
    v = max(v, GD_DPP_SELF(v, 0x143, 0xc));
    return __builtin_amdgcn_readlane(v, 63);
}
// (non-negative floats order like their bit patterns)
__device__ __forceinline__ float wave_max_nonneg(float f) { return __int_as_float(wave_max(__float_as_int(f))); }
__device__ __forceinline__ float wave_min_nonneg(float f) { return __int_as_float(~wave_max(~__float_as_int(f))); }
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }

// Scratch rows that one kernel writes once and the next reads once (candidate words, ranks, heap arrays, tables): streaming
// (nt) accesses, so that they take no room from what the step reads again and again (the road arrays, the pose planes).
#ifndef GD_NT_SCAN_STORE
#define GD_NT_SCAN_STORE 0
#endif
#ifndef GD_NT_WORDS_LOAD
#define GD_NT_WORDS_LOAD 1
#endif
#ifndef GD_NT_FINISH_LOAD
#define GD_NT_FINISH_LOAD 1
#endif
#ifndef GD_NT_REPLAY_LOAD
#define GD_NT_REPLAY_LOAD 1
#endif
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void stream_store(T v, T *p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ uint4 stream_load(const uint4 *p) {
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store(uint4 v, uint4 *p) {
    u4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<u4v *>(p));
}
__device__ __forceinline__ uint2 stream_load(const uint2 *p) {
    const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void stream_store(uint2 v, uint2 *p) {
    u2v t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p));
}

// bounds audit (engine.hpp GD_RANK_AUDIT): `idx` must lie in [0, size); counts the violation and returns an index that does
__device__ __forceinline__ int audited(const DevSim &d, int idx, int size) {
    if ((unsigned int)idx < (unsigned int)size) return idx;
    atomicAdd(&d.rk_hist[GD_RANK_AUDIT], 1);
    return idx < 0 ? 0 : size - 1;
}

// key comparison on ranks: key(a) < key(b)  <=>  (a >> 5) < (b >> 5)  <=>  (a | 31) < b.  0 is "below everything".
// (tm: 31 for a standard agent's ranks, 15 for a long list's: RankGeo::RSH)
__device__ __forceinline__ bool rank_lt(unsigned int a, unsigned int b, unsigned int tm = 31u) { return (a | tm) < b; }

// ------------------------------------------------------------------------------------------------------------------
// k_knn_scan: one lane per agent, one workgroup (4 waves) per 64 agent slots of a world.  Every road of the world is tested
// against every agent's bound with the cheap form |p - e|^2 < bound * margin (the margin covers rounding and the squared
// norm of the stored quaternion, like k_map_obs's scan), 32 roads per candidate word; the roads come through LDS as
// broadcast reads, so a road costs a wave one LDS instruction and five vector instructions for 64 agents.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_TILE = 2048;  // roads staged in LDS at a time

template <int A_T>
__global__ __launch_bounds__(256) void k_knn_scan(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    constexpr int HALVES = A_T / 64;
    const int w = blockIdx.x / HALVES, a0 = (blockIdx.x % HALVES) * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = d.shape[w * 2 + 0];
    if (a0 >= n) return;
    const int a = a0 + lane;
    const bool live = a < n;
    const size_t i = (size_t)w * A_T + a;
    const size_t WA = (size_t)d.W * A_T;
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;

    __shared__ __attribute__((aligned(16))) float2 s_xy[SCAN_TILE];
    __shared__ unsigned short s_first[NCP][64];  // checkpoint q of lane l: first road it holds for ...
    __shared__ float s_thr[NCP][64];             // ... and the scan threshold it gives
    __shared__ int s_ncp[64];                    // checkpoints of lane l; 0: the lane takes no part in the scan
    __shared__ unsigned int s_needy[2];          // the agents to be bounded afresh (bit per lane), as the first wave saw them
    // (40 KB in all: four workgroups per CU, i.e. every workgroup of a 1024-world launch resident at once -- with 46 KB, three
    // per CU and a second generation, the kernel took 91 us instead of 61.  What only the set-up before the scan loop needs
    // therefore lives in the buffers of the loop: the transpose buffer and the road tile)
    constexpr int TB = 8;
    __shared__ __attribute__((aligned(16))) unsigned int s_tr[4][TB][65];
    static_assert(sizeof(float4) * 64 + sizeof(float) * 192 <= sizeof(unsigned int) * 4 * TB * 65 && 4 * 256 * 4 <= sizeof(float2) * SCAN_TILE, "aliases fit");
    float4 *const s_hdr = reinterpret_cast<float4 *>(&s_tr[0][0][0]);  // where and how many checkpoints lane l's previous selection recorded
    float *const s_bx = reinterpret_cast<float *>(s_hdr + 64), *const s_by = s_bx + 64, *const s_bm = s_by + 64;  // position and margin of lane l's agent (for the waves that bound it afresh)
    unsigned int (*const s_hist)[256] = reinterpret_cast<unsigned int (*)[256]>(s_xy);  // per wave: histogram of squared distances of the roads scanned so far

    float ex = 0.f, ey = 0.f;
    if (live) { ex = d.px[i]; ey = d.py[i]; }
    unsigned long long ranked = 0ull;  // wave 0: its lanes whose agents take the rank path ...
    int list_base = 0;                 // ... and where they go in rk_list
    if (wave == 0) {
        int ncp = 0;
        // (every lane's own checkpoint headers first, published: an agent without a usable bound of its own borrows one below)
        float4 h0 = make_float4(0.f, 0.f, __int_as_float(0), 0.f), h1 = h0;
        if (live) { h0 = d.cp_hdr[i]; h1 = d.cp_hdr[WA + i]; }
        s_hdr[lane] = h0;
        wave_sync();
        if (live) {
            // too far from every road of the world for any to be within the radius: no rows (a finished agent parked
            // at the padding position, src/sim.cpp:333-343)
            const float4 bb = d.road_bbox[w];
            const float ddx = fmaxf(fmaxf(bb.x - ex, ex - bb.z), 0.f), ddy = fmaxf(fmaxf(bb.y - ey, ey - bb.w), 0.f);
            const float far = d.p.observationRadius * 1.01f + 1.f;
            int state = 1, reason = -1;
            if (R >= K && ddx * ddx + ddy * ddy > far * far) {
                state = RK_FAR;
                // every road is beyond the radius: radiusFilter leaves nothing (src/knn.hpp:83-97, 156-157).  No replay, no
                // checkpoints; the (empty) hand-over to k_map_rows is written here
                d.sel_hdr[i * 2] = make_float4(ex, ey, d.qw[i], d.qz[i]);
                d.sel_hdr[i * 2 + 1] = make_float4(__int_as_float(0), __int_as_float(r0), 0.f, 0.f);
                d.cp_hdr[i] = make_float4(ex, ey, __int_as_float(0), 0.f);
            } else {
                // the previous selection's checkpoints, or those of the episode's first selection (a reset puts the agent
                // back where that one was made): whichever was recorded closer to where the agent is now
                const int n0 = __float_as_int(h0.z), n1 = __float_as_int(h1.z);
                const float m0 = n0 > 0 ? sqrtf((ex - h0.x) * (ex - h0.x) + (ey - h0.y) * (ey - h0.y)) : __builtin_inff();
                const float m1 = n1 > 0 ? sqrtf((ex - h1.x) * (ex - h1.x) + (ey - h1.y) * (ey - h1.y)) : __builtin_inff();
                int set = m1 < m0 ? 1 : 0;
                float move = set ? m1 : m0;
                int n_cp = set ? n1 : n0;
                size_t src = i;  // the agent whose checkpoint rows bound this selection
                // An agent far from where its own checkpoints were recorded (a logged agent back from the padding position,
                // src/sim.cpp:370-382: none at all, or the episode's first ones from somewhere else) borrows those of the
                // agent of its world whose previous selection was made nearest to where it is now: the K-th distance over a
                // prefix of the roads is a function of the position alone, 1-Lipschitz in it, whoever recorded it.  Without
                // that every road of the world was a candidate for such an agent: beyond the buffers in a world of 10,000
                // roads (the whole group of 32 to the fallback, a 3 ms step), the longest replay of the launch in smaller ones.
                if (move > 6.f) {
                    float best = move;
                    int donor = -1;
                    for (int j = 0; j < 64; j++) {
                        const float4 hj = s_hdr[j];
                        const float dj = __float_as_int(hj.z) > 0 ? sqrtf((ex - hj.x) * (ex - hj.x) + (ey - hj.y) * (ey - hj.y)) : __builtin_inff();
                        if (dj < best) { best = dj; donor = j; }
                    }
                    if (donor >= 0) {
                        move = best;
                        set = 0;
                        n_cp = __float_as_int(s_hdr[donor].z);
                        src = (size_t)w * A_T + a0 + donor;
                    }
                }
                const bool eligible = R >= K && R >= d.rk_min_roads && R <= d.rk_max_roads;  // small worlds: k_map_obs is as fast
                // a group that needed the fallback three selections in a row (agents that overflow even the long list)
                // stops paying for rank kernels whose work is thrown away; it tries again every 64th selection
                const int grp = (int)(i / 32);
                const bool bypass = eligible && d.rk_streak[grp] >= 3 && ((d.rk_hist[513] + grp) & 63) != 0;
                const float iw = d.qw[i], iz = d.qz[i];
                const float z2 = iz * iz;
                const float det = (1.f - 2.f * z2) * (1.f - 2.f * z2) + 4.f * z2 * (iw * iw);
                const float margin = 1.00001f / fminf(det, 1.f);
                if (!eligible || bypass) {
                    state = 0;
                    ncp = 0;
                    reason = bypass ? -5 : -2;  // -2: a world below rk_min_roads (or without K roads)
                    d.rk_fallback[i / 32] = 1;
                } else if (n_cp <= 0 || move > 6.f) {
                    // No usable bound -- the first selection after the worlds were built, a logged agent that reappears
                    // anywhere on the map (src/sim.cpp:370-382) with nobody near: the waves of this workgroup bound it afresh
                    // below (`bound_afresh`).  Round 3 sent such an agent's group to the history replay on keys.
                    ncp = -1;
                    s_bx[lane] = ex; s_by[lane] = ey; s_bm[lane] = margin;
                } else {
                    ncp = n_cp;
                    const unsigned short *cr = d.cp_road + ((size_t)set * WA + src) * NCP;
                    const float *ct = d.cp_T + ((size_t)set * WA + src) * NCP;
                    float t = 1.f;
                    for (int q = 0; q < ncp; q++) {
                        t = ct[q];
                        const float reach = sqrtf(t) * 1.0001f + move * 1.0001f + 1e-3f;
                        float thr = reach * reach * 1.0001f * margin;
                        if (!(thr >= 0.f)) thr = __builtin_inff();  // an unusable entry bounds nothing
                        s_first[q][lane] = cr[q];
                        s_thr[q][lane] = thr;
                    }
                    d.rk_tl[i] = t;  // the last K-th key: scales the ranking buckets
                }
            }
            d.rk_n[i] = state;
            d.rk_ticket[i] = reason;  // k_knn_rank takes a ticket (>= 0) once the agent's ranks exist; < -1: why it fell back
            if (state == RK_FAR) ncp = 0;
        }
        s_ncp[lane] = ncp;
        {
            // published once, before the barrier: the waves below must all see ONE mask (a wave that finishes an agent rewrites
            // its s_ncp entry, and a late reader of s_ncp would deal the turns differently and skip the barrier behind the loop)
            const unsigned long long nd = __ballot(ncp < 0);
            if (lane == 0) { s_needy[0] = (unsigned int)nd; s_needy[1] = (unsigned int)(nd >> 32); }
        }
        if (ncp < 0) ncp = 1;  // (bounded afresh below: on the rank path like the others)
        // the agents on the rank path, as a list: k_knn_rank's waves share THEM out, not the live agents (on the Waymo tiles
        // five agents in six are parked out of reach of every road, and a wave that drew three of the others set the pace).
        // One list per XCD (workgroup b runs on XCD b % 8; k_knn_rank's waves take the list of their own XCD, so an agent's
        // road points are gathered into the L2 that scanned them, and eight counters are asked instead of one: a thousand
        // workgroups adding to one address at once cost the kernel 10 us)
        ranked = __ballot(ncp > 0);
    }
    __syncthreads();
    // ---- agents without a usable bound: one wave each streams the world's roads once, in scan order, and keeps a histogram of
    // their squared distances (16 buckets per octave: the exponent and four mantissa bits).  At up to 39 road counts spaced
    // geometrically (the K-th distance falls like the logarithm of the roads seen) the upper edge of the bucket that holds the
    // K-th smallest distance so far is a bound of the K-th distance from there on -- a checkpoint like those a previous
    // selection leaves, at most 4.4 % loose, made without one.  About 6 us per such agent. ----
    {
        const unsigned long long needy = (unsigned long long)s_needy[0] | (unsigned long long)s_needy[1] << 32;
        if (needy != 0ull) {  // (workgroup-uniform)
            constexpr int BASE = (127 - 2) << 4;  // bucket 0: below 2^-2 m^2; bucket 255: 2^13.9 m^2 (118 m) and beyond = unbounded
            unsigned int *hist = s_hist[wave];
            int turn = 0;
            for (unsigned long long rest = needy; rest != 0ull; rest &= rest - 1ull, turn++) {
                if ((turn & 3) != wave) continue;  // wave-uniform
                const int al = __ffsll((long long)rest) - 1;
                const float bx = s_bx[al], by = s_by[al], mg = s_bm[al];
#pragma unroll
                for (int k = 0; k < 4; k++) hist[k * 64 + lane] = 0u;
                if (lane == 0) { s_first[0][al] = (unsigned short)((K / 32) * 32); s_thr[0][al] = __builtin_inff(); }
                const float ratio = exp2f(log2f(fmaxf((float)R / 256.f, 1.f)) / (float)(NCP - 2));
                float pf = 256.f, tlast = __builtin_inff();
                int pq = 256, q = 1;
                float2 nxt = lane < R ? d.road_xy[r0 + lane] : make_float2(0.f, 0.f);
                for (int rb = 0; rb < R; rb += 64) {
                    const float2 xy = nxt;
                    if (rb + 64 + lane < R) nxt = d.road_xy[r0 + rb + 64 + lane];
                    if (rb + lane < R) {
                        const float dx = xy.x - bx, dy = xy.y - by;
                        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                        atomicAdd(&hist[min(255, max(0, (int)(__float_as_uint(d2) >> 19) - BASE))], 1u);
                    }
                    if (rb + 64 == pq && q < NCP && pq < R) {  // wave-uniform: roads [0, pq) are in the histogram
                        wave_sync();
                        const uint4 h = reinterpret_cast<const uint4 *>(hist)[lane];  // lane l owns buckets 4 l .. 4 l + 3
                        const int own = (int)(h.x + h.y + h.z + h.w);
                        const int incl = wave_incl_scan(own), excl = incl - own;
                        int jb = 0, before = excl;
                        if (before + (int)h.x < K) { before += (int)h.x; jb = 1;
                            if (before + (int)h.y < K) { before += (int)h.y; jb = 2;
                                if (before + (int)h.z < K) { jb = 3; } } }
                        const unsigned long long cross = __ballot(excl < K && incl >= K);  // exactly one lane: pq >= 256 > K
                        const int Lc = __ffsll((long long)cross) - 1;
                        const int bucket = 4 * Lc + __builtin_amdgcn_readlane(jb, Lc);
                        const float T = bucket >= 255 ? __builtin_inff() : __uint_as_float((unsigned int)(bucket + 1 + BASE) << 19);
                        if (lane == 0) {
                            const float reach = sqrtf(T) * 1.0001f + 1e-3f;
                            float thr = reach * reach * 1.0001f * mg * 1.001f;  // (the squared distances here are the scan's own form)
                            if (!(thr >= 0.f)) thr = __builtin_inff();
                            s_first[q][al] = (unsigned short)min(pq, 65535);
                            s_thr[q][al] = thr;
                        }
                        tlast = T;
                        q++;
                        pf *= ratio;
                        pq = max(pq + 64, ((int)pf + 63) & ~63);
                    }
                }
                if (lane == 0) {
                    s_ncp[al] = q;
                    d.rk_tl[(size_t)w * A_T + a0 + al] = tlast < 1e30f ? tlast : 1.f;  // scales the ranking buckets
                }
            }
            __syncthreads();
        }
    }
    const int ncp = s_ncp[lane];
    if (__syncthreads_or(ncp > 0 ? 1 : 0) == 0) return;  // no agent of this workgroup is on the rank path (far, fallback, bypass)
    int q = -1;  // checkpoint in force for this lane
    const int nch = (R + 31) >> 5;
    // A wave takes 16 consecutive chunks of the tile, 8 at a time: the agents' words of a batch are handed over agent-major
    // (k_knn_rank reads an agent's words as one row) through an LDS transpose, 32 bytes per agent, wave and batch
    uint32_t *words = d.rk_words + ((size_t)w * A_T + a0) * GD_RANK_NCH;
    for (int tile = 0; tile < R; tile += SCAN_TILE) {
        __syncthreads();
        for (int r = tile + tid; r < min(R, tile + SCAN_TILE); r += 256) s_xy[r - tile] = d.road_xy[r0 + r];
        __syncthreads();
        // the place of this workgroup's ranked agents in their list: asked for once the wave's last loads are in (memory
        // operations complete in order: asked earlier it held up the road loads, asked at the very end nothing hides it)
        if (wave == 0 && ranked != 0ull && lane == 0 && tile + SCAN_TILE >= R)
            list_base = atomicAdd(&d.rk_hist[528 + (blockIdx.x & 7)], __popcll(ranked));
#pragma clang loop unroll(disable)
        for (int c_first = (tile >> 5) + wave * 16; c_first < (tile >> 5) + wave * 16 + 16 && c_first < nch; c_first += TB) {
#pragma clang loop unroll(disable)
            for (int k = 0; k < TB; k++) {
                const int c = c_first + k;
                unsigned int wd = 0;
                if (c < nch) {  // wave-uniform
                    const int base = c << 5;
                    while (q + 1 < ncp && (int)s_first[q + 1][lane] <= base) q++;
                    const float thr = q >= 0 ? s_thr[q][lane] : __builtin_inff();
                    const float2 *t = s_xy + (base - tile);
#pragma unroll
                    for (int j = 31; j >= 0; j--) {
                        const float2 xy = t[j];  // the same address in every lane: a broadcast read
                        const float dx = xy.x - ex, dy = xy.y - ey;
                        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                        wd = __builtin_amdgcn_alignbit(wd, __float_as_uint(d2 - thr), 31);  // (wd << 1) | (d2 < thr)
                    }
                    if (base < K) wd |= K - base >= 32 ? 0xffffffffu : (1u << (K - base)) - 1u;  // roads below K regardless
                    const int left = R - base;
                    if (left < 32) wd &= (1u << left) - 1u;
                }
                s_tr[wave][k][lane] = wd;
            }
            wave_sync();
#pragma unroll
            for (int it = 0; it < TB; it++) {
                const int ag = it * 8 + (lane >> 3), k = lane & 7;
#if GD_NT_SCAN_STORE
                if (c_first + k < nch && s_ncp[ag] > 0) stream_store(s_tr[wave][k][ag], words + (size_t)ag * GD_RANK_NCH + c_first + k);
#else
                if (c_first + k < nch && s_ncp[ag] > 0) words[(size_t)ag * GD_RANK_NCH + c_first + k] = s_tr[wave][k][ag];
#endif
            }
            wave_sync();
        }
    }
    if (wave == 0 && ranked != 0ull) {
        list_base = __builtin_amdgcn_readfirstlane(list_base);
        if (ncp > 0)
            d.rk_list[(size_t)(blockIdx.x & 7) * WA + audited(d, list_base + __popcll(ranked & ((1ull << lane) - 1ull)), (int)WA)] = (int)i;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_knn_rank: one wave per live agent.  Candidate words -> candidate list, exact keys, ranks.
// ------------------------------------------------------------------------------------------------------------------
// 10 KB of LDS per agent, so that sixteen waves fit a CU: the kernel is a chain of LDS and memory round trips, and with two
// waves per SIMD (20 KB) it waited 46 % of its cycles (profiles/r03: SQ_WAIT_ANY).  A lane keeps its up to 20 candidates
// (positions lane, lane + 64, ...) in registers through all passes; LDS holds only what lanes exchange.
template <int CAP_T>
struct RankLds {
    union {
        unsigned short cidx[CAP_T];       // candidate road indices in road order, while the word expansion hands them to their lanes
        struct {
            float skey[CAP_T];            // the keys in bucket order
            unsigned short spos[CAP_T];   // bucket order -> candidate position
        } s;
        unsigned short spc[CAP_T];        // at the end: sorted slot -> road index
    };
    unsigned int cnt2[RankGeo<CAP_T>::NB / 2];  // bucket counters, two u16 per word; then cursors: after the scatter the END of each bucket
};
static_assert(sizeof(RankLds<CAP>) <= 10240 && sizeof(RankLds<CAP_LONG>) <= 20480, "sixteen / eight waves per CU");

// What the ranking of one agent reads from global memory before it can start, fetched while the previous agent of the
// wave is being ranked (the fetches are a chain of dependent loads, several microseconds end to end).
constexpr int WPL = GD_RANK_NCH / 64;  // candidate words per lane
struct RankIn {
    int i, state, r0, R, li;  // li: the entry of the list this agent came from (a long list's scratch rows go by it)
    float ex, ey, iw, iz, t_last;
    unsigned int wd[WPL];
};
// A value at a wave-uniform address that this kernel does not change, read through the scalar cache.  As vector loads the
// chain list -> agent -> state / pose waited, at each link, for every store the wave had issued for the previous agent
// (vector memory operations are counted in order): a fifth of the kernel.
// INVARIANT (nothing enforces it but the call sites below): the scalar cache is not coherent with this kernel's own vector
// stores, so an address read through here must never be read AFTER any wave of the same launch has written it.  What is
// read this way: arrays written by EARLIER kernels only (rk_list, rk_hist[528..], rk_tl, road_off, the pose planes: the
// scalar cache is invalidated at every kernel boundary), and rk_n[i] -- which this kernel does rewrite, but only the wave
// that ranks agent i writes rk_n[i], and it reads it (rank_fetch) before it writes it (end of rank_agent); a cache line
// shared with another agent's already rewritten entry may be stale for THAT entry, which this wave never looks at.
template <typename T>
__device__ __forceinline__ T uniform_load(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}
template <int A_T>
__device__ __forceinline__ RankIn rank_fetch(const DevSim &d, const int *list, int li, int count, int lane) {
    RankIn in;
    in.li = li;
    in.i = li < count ? uniform_load(list + li) : 0;
    in.state = li < count ? uniform_load(d.rk_n + in.i) : 0;  // (k_knn_scan's; this kernel rewrites it once the agent is ranked)
    const int w = in.i / A_T;
    in.r0 = uniform_load(d.road_off + w);
    in.R = uniform_load(d.road_off + w + 1) - in.r0;
    in.ex = uniform_load(d.px + in.i); in.ey = uniform_load(d.py + in.i);
    in.iw = uniform_load(d.qw + in.i); in.iz = -uniform_load(d.qz + in.i);  // the INVERSE rotation
    in.t_last = uniform_load(d.rk_tl + in.i);
    const int nch = (in.R + 31) >> 5;
    const uint32_t *words = d.rk_words + (size_t)in.i * GD_RANK_NCH;  // agent-major: one coalesced read
#pragma unroll
    for (int k = 0; k < WPL; k++) {
        const int c = k * 64 + lane;
        in.wd[k] = (in.state == 1 && c < nch) ? (GD_NT_WORDS_LOAD ? stream_load(words + c) : words[c]) : 0u;
    }
    return in;
}

#ifdef GD_CLOCKS
// builds with -DGD_CLOCKS (tools/build_expt.sh clk -DGD_CLOCKS; nothing else differs from the product code): clock ticks per
// phase of the ranking, summed per wave in scalar registers, added up over the waves (gd_stat 10..17)
struct PhaseClock {
    unsigned int prev, sum[8];
    __device__ __forceinline__ void mark(int n) {
        const unsigned int t = (unsigned int)__builtin_amdgcn_s_memtime();
        sum[n] += t - prev;
        prev = t;
    }
};
#define GD_PHASE(n) clk.mark(n)
#else
struct PhaseClock {};
#define GD_PHASE(n)
#endif
template <int A_T, int CAP_T>
__device__ __forceinline__ void rank_agent(const DevSim &d, const RankIn &in, int lane, RankLds<CAP_T> &L, PhaseClock &clk, int next_entry,
                                           int count, const int *list, RankIn &nxt) {
    using G = RankGeo<CAP_T>;
    constexpr int NB = G::NB, NLIN = G::NLIN, NMAX = G::NMAX;
    constexpr unsigned int TM = (1u << G::RSH) - 1u;  // most equal keys before a candidate that its rank can count
    // The next agent's inputs are requested in the middle of this one, behind the last gather of the keys: vector memory
    // operations complete in order, so requested up front they (1.3 KB from HBM) were what every gather then waited for.
    if (__builtin_amdgcn_readfirstlane(in.state) != 1) {  // fallback or too far from every road (k_knn_scan)
        nxt = rank_fetch<A_T>(d, list, next_entry, count, lane);
        return;
    }
    // every lane fetched the same values: as scalars they index through scalar base addresses (a per-lane 64-bit pointer per
    // array costs two registers each, and reloading a spilled one made the wave wait for all its outstanding stores)
    auto uni = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
    auto unif = [](float v) -> float { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const int i = uni(in.i), r0 = uni(in.r0), R = uni(in.R);
    const int group = i / 32;  // 32 consecutive agent slots of a world: the fallback unit (a workgroup of k_map_obs)
    constexpr int NG = G::NG;  // candidates per lane

    // ---- how many candidates?  More than this instantiation ranks: on to the long-list instantiation, or to the fallback ----
    const int nch = (R + 31) >> 5;
    {
        int mine = 0;
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            const int c = k * 64 + lane;
            unsigned int wd = k * 64 < nch ? in.wd[k] : 0u;
            if (k == 0) wd = c < K / 32 ? 0u : (c == K / 32 ? wd & ~((1u << (K % 32)) - 1u) : wd);
            mine += __popc(wd);
        }
        const int total = K + wave_sum(mine);
        // (NMAX = CAP - 8: the sorted key array ends in eight +inf entries that reads past a bucket's end run into.  Round 3
        // took up to CAP candidates; with exactly CAP there was no room for an end marker and the clamp that stood in for it
        // re-read the LAST entry of the last bucket, which is not necessarily its largest: a candidate of that bucket could
        // count a smaller key twice)
        if (total > NMAX) {
            if (lane == 0) {
                bool passed_on = false;
                if (!G::LONG && total <= RankGeo<CAP_LONG>::NMAX) {
                    const int slot = atomicAdd(&d.rk_hist[GD_RH_LONG], 1);
                    if (slot < d.rk_nlong) {  // (rk_n stays 1, the ticket -1: the long-list launch finds the agent as k_knn_scan left it)
                        d.rk_longlist[slot] = i;
                        passed_on = true;
                    }
                }
                if (!passed_on) {
                    d.rk_n[i] = 0;
                    d.rk_ticket[i] = -3;  // more candidates than the buffers hold
                    d.rk_fallback[group] = 1;
                }
            }
            nxt = rank_fetch<A_T>(d, list, next_entry, count, lane);
            return;
        }
    }

    // ---- candidate words -> road indices in ascending order ----
    // Roads 0..K-1 are candidates regardless (k_knn_scan sets their bits) and their positions are their indices: written
    // directly, not bit by bit (the lanes that own those seven words would loop 32 times while the others wait)
    static_assert(K == 200 && CAP >= 256, "roads 0..K-1: four stores per lane");
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q * 64 + lane < K) L.cidx[q * 64 + lane] = (unsigned short)(q * 64 + lane);
    int nin = K;
#pragma unroll
    for (int k = 0; k < WPL; k++) {
        if (k * 64 >= nch) break;  // wave-uniform
        const int c = k * 64 + lane;
        unsigned int wd = in.wd[k];
        if (k == 0) wd = c < K / 32 ? 0u : (c == K / 32 ? wd & ~((1u << (K % 32)) - 1u) : wd);
        const int pc = __popc(wd);
#ifdef GD_CLOCKS
        clk.sum[7] += (unsigned int)wave_max(pc) << 8;  // the wave's trips through the loop below (gd_stat 17)
#endif
        const int incl = wave_incl_scan(pc);
        int pos = nin + incl - pc;
        nin += __builtin_amdgcn_readlane(incl, 63);
        while (wd) {
            L.cidx[pos++] = (unsigned short)((c << 5) + __ffs(wd) - 1);
            wd &= wd - 1u;
        }
    }
    wave_sync();
    // the agent's place in the replay order (k_knn_order: most candidates first).  Taken now: the counter's
    // answer is a trip to the L2 and back that nothing has to wait for until the agent is done
    const int bin = 255 - min(255, (nin - K) / 5);
    int ticket = 0;
    if (lane == 0) ticket = bin << 20 | atomicAdd(&d.rk_hist[bin], 1);
    GD_PHASE(1);
    if (GD_DIAG_IS(d.rk_dbg, 1)) {
        nxt = rank_fetch<A_T>(d, list, next_entry, count, lane);
        return;
    }
    // where this selection's checkpoints start to apply (their K-th distances are filled in by k_knn_finish): checkpoint 0
    // is the heap of the first K roads (road indices below K are candidates regardless), checkpoint q the heap after
    // candidate K + 32 q - 1, which holds for every road behind that candidate; rounded up to whole 32-road chunks of the scan
    if (lane <= (nin - K) >> G::TSH) {
        const int road = (int)L.cidx[K - 1 + (lane << G::TSH)];
        d.cp_road[(size_t)i * NCP + lane] = (unsigned short)(lane ? min(65535, (road + 1 + 31) & ~31) : (K / 32) * 32);
    }
    // A lane has 13 candidates on average, 20 at most.  The passes over them go in groups of GQ with one wave-uniform guard
    // per group and nothing conditional inside (idle lanes work on road 0): a guard per candidate is a branch per candidate,
    // and the compiler then waits for each load before it issues the next
    constexpr int GQ = 5;
    static_assert(NG % GQ == 0, "whole groups");
    int ci[NG];
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += GQ) {
#pragma unroll
        for (int u = 0; u < GQ; u++) ci[g0 + u] = 0;
        if (g0 * 64 < nin) {  // wave-uniform
            int v[GQ];
#pragma unroll
            for (int u = 0; u < GQ; u++) v[u] = (int)L.cidx[(g0 + u) * 64 + lane];  // (beyond nin: stale, dropped below)
#pragma unroll
            for (int u = 0; u < GQ; u++) ci[g0 + u] = (g0 + u) * 64 + lane < nin ? v[u] : 0;
        }
    }
    wave_sync();  // the index buffer becomes the sorted arrays
    // ---- exact keys (gd_math.hpp ego_dist2: the reference's arithmetic); the gathers of a group are in flight together ----
    const float ex = unif(in.ex), ey = unif(in.ey), iw = unif(in.iw), iz = unif(in.iz);
    const float kmax = d.radius_key_max;
    const float2 *rxy = d.road_xy + r0;
    float key[NG];
    const float t_last = unif(in.t_last);
    float split = (t_last > 0.f && t_last < 1e30f) ? t_last * 1.5f : 1.f;
    int nle = 0, nlow = 0;  // candidates inside the radius / below `split` (second count in the upper half)
    float kmax_seen = 0.f, kmin_seen = __builtin_inff();
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += GQ) {
#pragma unroll
        for (int u = 0; u < GQ; u++) key[g0 + u] = 0.f;
        if (g0 * 64 < nin) {  // wave-uniform
            float2 xy[GQ];
#pragma unroll
            for (int u = 0; u < GQ; u++) xy[u] = rxy[(unsigned int)ci[g0 + u]];
            // (all five before the first use: left alone, the scheduler pairs each load with its arithmetic to save registers)
            static_assert(GQ == 5, "operand list");
            asm volatile("" : "+v"(xy[0].x), "+v"(xy[0].y), "+v"(xy[1].x), "+v"(xy[1].y), "+v"(xy[2].x), "+v"(xy[2].y),
                              "+v"(xy[3].x), "+v"(xy[3].y), "+v"(xy[4].x), "+v"(xy[4].y));
#pragma unroll
            for (int u = 0; u < GQ; u++) {
                const int g = g0 + u;
                key[g] = ego_dist2(ex, ey, iw, iz, xy[u].x, xy[u].y);
                const bool on = g * 64 + lane < nin;
                nle += ((on & (key[g] <= kmax)) ? 1 : 0) + ((on & (key[g] < split)) ? 1 << 16 : 0);
                kmax_seen = fmaxf(kmax_seen, on ? key[g] : 0.f);
                kmin_seen = fminf(kmin_seen, on ? key[g] : __builtin_inff());
            }
        }
    }
    nxt = rank_fetch<A_T>(d, list, next_entry, count, lane);
    nle = wave_sum(nle);
    kmax_seen = wave_max_nonneg(kmax_seen);  // keys are sums of squares
    kmin_seen = wave_min_nonneg(kmin_seen);
    nlow = nle >> 16;
    nle &= 0xffff;
    GD_PHASE(2);
    if (GD_DIAG_IS(d.rk_dbg, 2)) return;

    // ---- ranks.  Counting sort into NB buckets (a monotone function of the key: linear up to 1.5 x the previous
    // K-th key, where most candidates lie, logarithmic beyond), then the exact order inside each bucket ----
    for (int b = lane; b < NB / 2; b += 64) L.cnt2[b] = 0u;
    wave_sync();
    // The previous K-th key says where the candidates are dense only while the agent is near where it was recorded.  After
    // a jump — fewer than a quarter of the candidates below `split` (all of them in a band far above it), or every one
    // below it (a logged agent back from the padding position, src/sim.cpp:333-343, with checkpoints recorded out there:
    // the whole world in bucket 0) — the buckets are linear between the smallest and the largest key instead: any
    // monotone function gives the same ranks, a poor one makes the order inside a bucket quadratic.
    const bool jumped = nlow * 4 < nin || nlow == nin;  // wave-uniform
    const float lin_lo = jumped ? kmin_seen : 0.f;
    const int nlin = jumped ? NB : NLIN;
    const float lin_scale = jumped ? (float)NB / fmaxf(kmax_seen - kmin_seen, 1e-30f) : (float)NLIN / split;
    if (jumped) split = __builtin_inff();
    const unsigned int split_bits = __float_as_uint(split);
    // above `split` the buckets are uniform in the key's bit pattern (i.e. logarithmic) up to the largest candidate key
    const float log_scale = (float)(NB - NLIN) / (float)(max(__float_as_uint(kmax_seen), split_bits + 1u) - split_bits + 1u);
    auto bucket_of = [&](float k) -> int {
        if (k < split) return min(nlin - 1, max(0, (int)((k - lin_lo) * lin_scale)));
        return NLIN + min(NB - NLIN - 1, (int)((float)(__float_as_uint(k) - split_bits) * log_scale));
    };
    // (the bucket rides in the upper half of the road-index register: registers decide how many waves a SIMD holds)
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g * 64 < nin) {  // wave-uniform
            const int b = bucket_of(key[g]);
            ci[g] |= b << 16;
            if (g * 64 + lane < nin) atomicAdd(&L.cnt2[b >> 1], 1u << ((b & 1) * 16));
        }
    }
    wave_sync();
    {
        // exclusive prefix over the buckets: lane l owns buckets l * (NB / 64) .. (an even number of them: whole words)
        constexpr int WPB = NB / 64 / 2;
        static_assert(WPB * 128 == NB, "whole counter words per lane");
        unsigned int own[WPB];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < WPB; k++) { own[k] = L.cnt2[lane * WPB + k]; sum += (int)(own[k] & 0xffffu) + (int)(own[k] >> 16); }
        int run = wave_incl_scan(sum) - sum;
#pragma unroll
        for (int k = 0; k < WPB; k++) {
            const int lo = run, hi = run + (int)(own[k] & 0xffffu);
            run = hi + (int)(own[k] >> 16);
            L.cnt2[lane * WPB + k] = (unsigned int)lo | (unsigned int)hi << 16;
        }
    }
    wave_sync();
    GD_PHASE(3);
    if (GD_DIAG_IS(d.rk_dbg, 3)) return;
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += GQ) {
        if (g0 * 64 < nin) {  // wave-uniform
            int sl[GQ];
#pragma unroll
            for (int u = 0; u < GQ; u++) {  // any order inside the bucket; idle lanes add nothing (to bucket 0)
                const int b = ci[g0 + u] >> 16, sh = (b & 1) * 16;
                const unsigned int one = (g0 + u) * 64 + lane < nin ? 1u << sh : 0u;
                sl[u] = (int)((atomicAdd(&L.cnt2[b >> 1], one) >> sh) & 0xffffu);
            }
#pragma unroll
            for (int u = 0; u < GQ; u++) {
                if ((g0 + u) * 64 + lane < nin) {
                    L.s.skey[sl[u]] = key[g0 + u];
                    L.s.spos[sl[u]] = (unsigned short)((g0 + u) * 64 + lane);
                }
            }
        }
    }
    wave_sync();
    GD_PHASE(4);
    if (GD_DIAG_IS(d.rk_dbg, 4)) return;
    const unsigned short *cur16 = reinterpret_cast<const unsigned short *>(L.cnt2);  // cursor of bucket b = its END
    // Candidates with a smaller key = those of the earlier buckets + the smaller ones of the own bucket.  The pass is a chain
    // of LDS round trips with little to issue in between, so it reads generously: the first six members of every bucket at
    // once (most hold one or two), then four more per trip while any lane's bucket has members left.  Only keys are read;
    // candidates that met their own key more than once (equal keys, rare) get their place among those afterwards.
    // (a candidate's rank e replaces its bucket in the upper half of the road-index register: registers decide how many
    // waves a SIMD holds)
    unsigned int eqmask = 0u;  // bit g: candidate g of this lane shares its key with another candidate
    float *const kt_row = G::LONG ? d.rk_kt_long + (size_t)uni(in.li) * G::KTN : d.rk_kt + (size_t)i * G::KTN;
    if (lane == 0) kt_row[(nin + 15) >> 4] = kmax_seen;  // behind the last multiple of 16: the largest key
    constexpr int U = 4, M = 6, STEP = 4;  // (M = 8: 399 us, 6: 391, 4: 405 -- two instructions per member read and key)
    // A read past the end of the own bucket meets keys of later buckets, which are larger (the bucket function is monotone)
    // and so count neither as smaller nor as equal: no bounds test per member.  Past the last candidate it meets +inf.
    // Past the last candidate: eight entries of +inf, so that the first M members are read at constant offsets from the
    // bucket's start with no clamp (three vector instructions per member less: k_knn_rank 407 -> 391 us)
    const int lim = nin;  // <= NMAX
    if (lane < 8) L.s.skey[nin + lane] = __builtin_inff();
    wave_sync();
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += U) {
        if (g0 * 64 >= nin) break;  // wave-uniform
        int s0[U], s1[U], less[U], eq[U], longest = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int b = ci[g0 + u] >> 16;
            s0[u] = b ? (int)cur16[b - 1] : 0;
            s1[u] = (g0 + u) * 64 + lane < nin ? (int)cur16[b] : s0[u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float mk[M];
            static_assert(M <= 8, "end markers");
#pragma unroll
            for (int k = 0; k < M; k++) mk[k] = L.s.skey[s0[u] + k];  // s0 < nin: at most nin + 6
            less[u] = s0[u];
            eq[u] = 0;
            longest = max(longest, s1[u] - s0[u]);
#pragma unroll
            for (int k = 0; k < M; k++) {
                less[u] += mk[k] < key[g0 + u] ? 1 : 0;
                eq[u] += mk[k] == key[g0 + u] ? 1 : 0;
            }
        }
        longest = wave_max(longest);
        if (GD_DIAG_IS(d.rk_dbg, 7)) longest = 0;
#ifdef GD_DIAG
        if (lane == 0 && d.rk_dbg == 9) atomicMax(&d.rk_hist[514], longest);  // diagnostic: the most crowded bucket of this selection
#endif
        for (int j = M; j < longest; j += STEP) {
            float mkk[U][STEP];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < STEP; k++) mkk[u][k] = L.s.skey[min(s0[u] + j + k, lim)];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < STEP; k++) {
                    less[u] += mkk[u][k] < key[g0 + u] ? 1 : 0;
                    eq[u] += mkk[u][k] == key[g0 + u] ? 1 : 0;
                }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool on = (g0 + u) * 64 + lane < nin;
            eqmask |= ((eq[u] > 1) & on) ? 1u << (g0 + u) : 0u;  // (a candidate meets itself once)
            ci[g0 + u] = (ci[g0 + u] & 0xffff) | ((less[u] + 1) << (16 + G::RSH));  // e = (less + 1) << RSH | tie, in bits 16..31
            // the key table for k_knn_finish: this key sits at the sorted slots [less, less + eq); whoever holds slot 16 j
            // writes entry j (candidates with equal keys write the same value)
            const int j = (less[u] + 15) >> 4;
            if (on && (j << 4) < less[u] + eq[u]) kt_row[j] = key[g0 + u];
        }
    }
    int too_many_ties = 0;
    const bool has_tie = __ballot(eqmask != 0u) != 0ull;  // the replay then compares ranks without their tie field
    if (has_tie) {
        // place among the candidates with the same key: those of them that come earlier in road order.  A rolled loop (the
        // register arrays are picked apart with selects on the wave-uniform g): twenty copies of it cost the common path
        // its registers
#pragma clang loop unroll(disable)
        for (int g = 0; g * 64 < nin; g++) {
            if (__ballot((eqmask >> g) & 1u) == 0ull) continue;  // wave-uniform
            int cg = 0;
#pragma unroll
            for (int k = 0; k < NG; k++) cg = k == g ? ci[k] : cg;
            int tie = 0;
            if ((eqmask >> g) & 1u) {
                // (the candidate's bucket made way for its rank: found again as the bucket whose range of sorted slots holds
                // `less`, eleven probes of the cursors -- this pass is rare)
                const int less_g = (int)((unsigned int)cg >> (16 + G::RSH)) - 1, p = g * 64 + lane;
                int lo = 0, hi = NB - 1;
#pragma clang loop unroll(disable)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)cur16[mid] > less_g) hi = mid; else lo = mid + 1;
                }
                const int b = lo;
                const int first = b ? (int)cur16[b - 1] : 0, end = (int)cur16[b];
                float kg = 0.f;  // the candidate's key, from its own entry of the bucket (the key registers are long gone)
                for (int m = first; m < end; m++) kg = (int)L.s.spos[m] == p ? L.s.skey[m] : kg;
                for (int m = first; m < end; m++) tie += (L.s.skey[m] == kg ? 1 : 0) & ((int)L.s.spos[m] < p ? 1 : 0);
            }
            too_many_ties |= tie > (int)TM ? 1 : 0;
#pragma unroll
            for (int k = 0; k < NG; k++) ci[k] |= k == g ? (tie & (int)TM) << 16 : 0;
        }
    }
    wave_sync();  // every read of the sorted arrays is done: their space becomes the slot -> road table
    GD_PHASE(5);
    if (GD_DIAG_IS(d.rk_dbg, 5)) return;
    if (__ballot(too_many_ties != 0) != 0ull) {
        if (lane == 0) {
            d.rk_n[i] = 0;
            d.rk_ticket[i] = ticket | RK_TIES;  // more than 32 candidates with one key (the place in the order stays taken)
            d.rk_fallback[group] = 1;
        }
        return;
    }
    // (rows through scalar base addresses and unsigned 32-bit lane offsets: see `uni` above)
    // The slot -> road table goes on for its first SPL slots only: the K elements the replayed heap ends with are the K
    // smallest keys (slot < K + 31 with equal keys), and that is all k_knn_finish looks up by slot -- the checkpoints'
    // K-th keys come from the key table above (round 3 wrote all CAP slots, 2.5 KB per agent, and k_knn_finish fetched the
    // whole row back for 240 scattered 2-byte reads)
    unsigned short *const E_row = G::LONG ? d.rk_E_long + (size_t)uni(in.li) * CAP_LONG : d.rk_E + (size_t)i * CAP;
    const unsigned int ulane = (unsigned int)lane;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g * 64 < nin) {  // wave-uniform
            if (g * 64 + lane < nin) {
                const unsigned int e = (unsigned int)ci[g] >> 16;
                stream_store((unsigned short)e, E_row + (g * 64u + ulane));
                const int slot = (int)(e >> G::RSH) - 1 + (int)(e & TM);
                if (slot < SPL) L.spc[slot] = (unsigned short)(ci[g] & 0xffff);
            }
        }
    }
    wave_sync();
    {
        static_assert(SPL == 256, "four slots per lane");
        const uint2 v = reinterpret_cast<const uint2 *>(L.spc)[lane];
        stream_store(v, reinterpret_cast<uint2 *>(d.rk_spc + (size_t)i * SPL) + ulane);
    }
    if (lane == 0) {
        d.rk_n[i] = nin | (nle << 16) | (has_tie ? 1 << 28 : 0) | (G::LONG ? 1 << 29 : 0);
        if (G::LONG) d.rk_longslot[i] = uni(in.li);
        d.rk_ticket[i] = ticket;
    }
    GD_PHASE(6);
}

// A wave ranks several agents in turn: tens of thousands of one-agent workgroups cost more in workgroup launches (each is
// handed its LDS first) than in work.
template <int A_T, int CAP_T>
__global__ __launch_bounds__(64, CAP_T > CAP ? 2 : 4) void k_knn_rank(DevSim d) {
    // (standard: at most 128 registers, four waves per SIMD like the LDS.  Measured: three waves per SIMD without spills
    // 437 us, four with seven spilled registers 391.  Long lists: 20 KB of LDS, two waves per SIMD, twice the registers)
    if (d.gate_any && *d.any_reset == 0) return;
    __shared__ RankLds<CAP_T> L;
    // Which agents a wave takes.  Standard: workgroup b runs on XCD b % 8 and shares out the list of the agents that
    // k_knn_scan's workgroups on that XCD put on the rank path (rk_list), one entry per wave and turn.  The road points an
    // agent gathers (32 KB per world on the bench scene) are then in that XCD's L2 already, and the waves resident at a time
    // work on a few dozen worlds instead of all of them (in agent-major order every generation of waves touched every world:
    // 32 MB against 4 MB of L2 per XCD; HBM bytes of the road observation 2.16 -> 1.36 GB per step).  Long lists: the one
    // list that the standard launch made of the agents it could not hold.
    constexpr bool LONG = RankGeo<CAP_T>::LONG;
    const int stride = LONG ? (int)gridDim.x : (int)(gridDim.x >> 3);  // (the standard grid is a multiple of 8 workgroups)
    const int xcd = blockIdx.x & 7;
    const int count = LONG ? min(uniform_load(d.rk_hist + GD_RH_LONG), d.rk_nlong) : uniform_load(d.rk_hist + 528 + xcd);
    const int *list = LONG ? d.rk_longlist : d.rk_list + (size_t)xcd * d.W * A_T;
    int t = LONG ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    RankIn cur = rank_fetch<A_T>(d, list, t, count, threadIdx.x);
    PhaseClock clk;
#ifdef GD_CLOCKS
    for (int k = 0; k < 8; k++) clk.sum[k] = 0u;
    clk.prev = (unsigned int)__builtin_amdgcn_s_memtime();
#endif
    for (; t < count; t += stride) {
        RankIn nxt;
        GD_PHASE(0);  // between agents: the buffers change hands
        rank_agent<A_T, CAP_T>(d, cur, threadIdx.x, L, clk, t + stride, count, list, nxt);
        wave_sync();  // the LDS buffers change hands
        cur = nxt;
    }
#ifdef GD_CLOCKS
    if (threadIdx.x == 0)
        for (int k = 0; k < 8; k++) atomicAdd(reinterpret_cast<unsigned int *>(&d.rk_hist[516 + k]), clk.sum[k] >> 8);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// k_knn_replay
// ------------------------------------------------------------------------------------------------------------------
// The heap of lane `l`: slots 1..K (1-based; the reference's array index is slot - 1) as u16 ranks, two per dword:
// pair j = (slot 2j, slot 2j + 1) at s_pair[j * 64 + l], so the children of slot g are pair g.  Slot K + 1 and, during
// the replay, slot K (whose element lives in a register) hold 0, which is below every rank.
#ifndef GD_RANK_AWR
#define GD_RANK_AWR 64
#endif
// agents per wave of the replay.  The round is one dependent chain (about 140 instructions and four LDS round trips), so a
// wave is as fast alone on its SIMD as beside a second one; measured at 1024 x 64 (us): 64 per wave 458, 32 per wave (two
// waves per SIMD) 492, 16 per wave 674; Waymo tiles 245 / 256 / 261 (tools/build_expt.sh awr32 -DGD_RANK_AWR=32)
constexpr int AWR = GD_RANK_AWR;
struct RankHeap {
    unsigned int *pr;  // this lane's column
    __device__ __forceinline__ unsigned int pair(int j) const { return pr[j * AWR]; }
    __device__ __forceinline__ unsigned int get(int g) const {
        return reinterpret_cast<const unsigned short *>(pr + (g >> 1) * AWR)[g & 1];
    }
    __device__ __forceinline__ void set(int g, unsigned int v) const {
        reinterpret_cast<unsigned short *>(pr + (g >> 1) * AWR)[g & 1] = (unsigned short)v;
    }
};

// Replay order: the agents on the rank path sorted by candidate count, longest first (counting sort over 256 bins: k_knn_rank
// takes a ticket in its bin, every workgroup of this kernel turns the bin counts into starts for itself -- 256 loads and a
// scan, cheaper than the kernel boundary that a launch of its own for them cost -- and places its agents).  A wave's rounds
// are its longest agent's, so agents of similar length share a wave; which wave an agent rides never changes its result.
// The bin counts stay as they are until k_knn_replay's first workgroup zeroes them for the next selection: every
// workgroup here reads them.
__global__ __launch_bounds__(256) void k_knn_order(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;
    __shared__ int s_part[4];
    __shared__ int s_start[256];
    const int tl = threadIdx.x;
    const int c = d.rk_hist[tl];
    const int incl = wave_incl_scan(c);
    if ((tl & 63) == 63) s_part[tl >> 6] = incl;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < (tl >> 6); k++) before += s_part[k];
    s_start[tl] = before + incl - c;  // start of bin tl
    if (blockIdx.x == 0 && tl == 255) {
        d.rk_hist[512] = before + incl;  // agents on the rank path
        d.rk_hist[513]++;                // selections so far (k_knn_scan staggers the retries of bypassing groups with it)
        for (int x = 0; x < 8; x++) d.rk_hist[528 + x] = 0;  // the next selection's lists of ranked agents start empty
        d.rk_hist[GD_RH_LONG] = 0;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + tl;
    if (t >= d.live_count) return;
    const int i = d.live_list[t];
    const int ticket = d.rk_ticket[i];
    if (ticket < 0) return;  // not on the rank path
    d.rk_order[audited(d, s_start[(ticket >> 20) & 255] + (ticket & 0xfffff), d.W * d.A)] = i;
}

__global__ __launch_bounds__(64) void k_knn_replay(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;
    const int lane = threadIdx.x;
    if (blockIdx.x == 0)  // the bin counts of this selection have been read by every workgroup of k_knn_order: ready for the next
        for (int k = 0; k < 4; k++) d.rk_hist[k * 64 + lane] = 0;
    const int li = blockIdx.x * AWR + lane;
    constexpr int NPAIR = 128;  // pairs 0..K/2 hold the heap; K/2 + 1 .. 127 stay 0: the "children" of slots beyond the heap
    __shared__ unsigned int s_pair[NPAIR * AWR];
    const RankHeap H{s_pair + (lane % AWR)};
    int i = 0, n = 0, has_tie = 0, is_long = 0;
    if (lane < AWR && li < d.rk_hist[512]) {
        i = d.rk_order[li];
        const int packed = d.rk_n[i];
        n = packed & 0xffff;
        has_tie = (packed >> 28) & 1;
        is_long = (packed >> 29) & 1;  // ranked by the long-list instantiation: its own rows, tile and rank format
        if (d.rk_fallback[i / 32] != 0) n = 0;  // the whole group is selected by k_map_obs
    }
    const bool on = n >= K;  // (with fewer than 64 agents per wave: never true for the upper lanes, which share LDS columns with the lower ones)
    if (__ballot(on) == 0ull) return;
    const unsigned short *E = (on && is_long) ? d.rk_E_long + (size_t)d.rk_longslot[i] * CAP_LONG : d.rk_E + (size_t)i * CAP;
    const unsigned int tm = is_long ? 15u : 31u;  // equal-key field of this agent's ranks (RankGeo::RSH)
    const int tsh = is_long ? 6 : 5;             // log2 (candidates between its checkpoints)

    // ---- the first K candidates are roads 0..K-1 in order (src/knn.hpp:112-120) ----
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(E);
        if (lane < AWR) {
            H.pr[0] = 0u;
            for (int j = K / 2; j < NPAIR; j++) H.pr[j * AWR] = 0u;
        }
#pragma clang loop unroll(disable)
        for (int k = 0; k < K / 8; k++) {
            const uint4 v = on ? stream_load(src + k) : make_uint4(0u, 0u, 0u, 0u);
            const unsigned int wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (lane < AWR) {
                    H.set(k * 8 + c * 2 + 1, wd[c] & 0xffffu);
                    H.set(k * 8 + c * 2 + 2, wd[c] >> 16);
                }
            }
        }
    }
    // ---- make_heap, src/binary_heap.hpp:170-185: parents K/2 .. 1, each __adjust_heap(hole, len = K, value) ----
    if (on) {
#pragma clang loop unroll(disable)
        for (int g = K / 2; g >= 1; g--) {
            const unsigned int x = H.get(g);
            int h = g;
            while (2 * h + 1 <= K) {  // both children exist: take the larger one (the right one unless it is smaller)
                const unsigned int p2 = H.pair(h);
                const unsigned int kl = p2 & 0xffffu, kr = p2 >> 16;
                const bool right = !rank_lt(kr, kl, tm);
                H.set(h, right ? kr : kl);
                h = 2 * h + (right ? 1 : 0);
            }
            if (2 * h == K) {  // a lone left child
                H.set(h, H.get(K));
                h = K;
            }
            while (h > g) {  // __push_heap towards the sift's own top
                const unsigned int pv = H.get(h >> 1);
                if (!rank_lt(pv, x, tm)) break;
                H.set(h, pv);
                h >>= 1;
            }
            H.set(h, x);
        }
    }
    unsigned int r[8];             // slots 1..7 (tree levels 0..2) live in registers during the replay; r[1] is heap[0]
#pragma unroll
    for (int j = 1; j < 8; j++) r[j] = H.get(j);
    unsigned int last = H.get(K);  // heap[K - 1], kept in a register as well
    if (lane < AWR) H.set(K, 0u);
    unsigned short *cpe = d.rk_cpe + (size_t)i * NCP;
    if (on) cpe[0] = (unsigned short)r[1];

    // ---- roads K.. : pop_heap + replace last + push_heap per insert (src/knn.hpp:128-151) ----
    int nmax = n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    const uint4 *blocks = reinterpret_cast<const uint4 *>(E + K);  // K * 2 bytes: 16-byte aligned
    uint4 cur = on ? (GD_NT_REPLAY_LOAD ? stream_load(blocks) : blocks[0]) : make_uint4(0u, 0u, 0u, 0u);
    uint4 nxt = on ? (GD_NT_REPLAY_LOAD ? stream_load(blocks + 1) : blocks[1]) : make_uint4(0u, 0u, 0u, 0u);
    // Two copies of the loop.  When none of the wave's agents has two candidates with one key (k_knn_rank reports it), every
    // rank's tie field is 0, a plain integer compare IS the key compare, "the larger child" is a max and the values along
    // the pop's path and the push's chain are medians of three (the moved children are non-increasing down the path, the
    // chain towards the leaf): about a third fewer instructions per insert.  The other copy compares ranks without their
    // tie field and selects.
#ifdef GD_CLOCKS
    int n_ins = 0;  // inserts of this lane's agent (gd_stat 20: total over the agents; 19: their candidates beyond K; 18: rounds of the first wave)
#endif
    auto replay = [&](auto ties_tag) {
        constexpr bool TIES = decltype(ties_tag)::value;
        auto lt = [&](unsigned int a, unsigned int b) -> bool { return TIES ? rank_lt(a, b, tm) : a < b; };
        auto larger = [&](unsigned int kl, unsigned int kr, bool &right) -> unsigned int {
            right = !lt(kr, kl);  // the right child unless it is smaller (src/binary_heap.hpp __adjust_heap)
            return TIES ? (right ? kr : kl) : max(kl, kr);
        };
        auto med3 = [](unsigned int a, unsigned int b, unsigned int c) -> unsigned int {
            return max(min(a, b), min(max(a, b), c));  // the backend folds this into v_med3_u32
        };
        // The ranks arrive eight at a time (16 bytes), two blocks ahead.  An outer loop per block: the request for block b + 2
        // is issued at the top and its registers are not touched for eight rounds (inside one flat loop the compiler copied the
        // freshly requested block into place at once, i.e. waited a memory round trip every eighth round: a quarter of the kernel)
        // One insert: candidate rank y goes through pop_heap / push_heap (called for the lanes whose candidate passes the
        // reference's test `cmp(current, heap[0])`, src/knn.hpp:138-143)
        auto insert = [&](const unsigned int y) {
            {
#ifdef GD_CLOCKS
                n_ins++;
#endif
                // pop_heap: the hole goes from the root to the bottom of the (K - 1)-element heap along the larger child.
                // Levels 0 and 1 are decided in registers (slots 1..7 live there during the replay); below that three, then two
                // levels per LDS round trip: a node's children pair, its grandchildren pairs (and great-grandchildren pairs)
                // are fetched together (pairs beyond the heap hold 0, which loses every comparison).  The ancestors of slot K that the push will meet are
                // requested now as well and patched where the pop's path went through them.
                int g[8];
                unsigned int ck[7];
                bool right0, right1;
                ck[0] = larger(r[2], r[3], right0);
                const unsigned int hl = right0 ? r[6] : r[4], hr = right0 ? r[7] : r[5];
                ck[1] = larger(hl, hr, right1);
                g[0] = 1;
                g[1] = 2 + (right0 ? 1 : 0);
                g[2] = 2 * g[1] + (right1 ? 1 : 0);
                const unsigned int q12 = H.get(12), q25 = H.get(25), q50 = H.get(50), q100 = H.get(100);
                // (two round trips: three levels below slot g[2] -- its children pair, both grandchildren pairs and all four
                // great-grandchildren pairs, seven dwords at constant offsets from one address -- then two levels below
                // g[5].  Round 3 and the first half of round 4 went two, two and one level: a round trip more per insert,
                // and the inserts of the agent with the most candidates are the kernel's duration: 452 -> 435 us)
                {
                    const int g2 = g[2];
                    const unsigned int pc = H.pair(g2), pl = H.pair(2 * g2), pr2 = H.pair(2 * g2 + 1);
                    const unsigned int p0 = H.pair(4 * g2), p1 = H.pair(4 * g2 + 1), p2 = H.pair(4 * g2 + 2), p3 = H.pair(4 * g2 + 3);
                    bool ra, rb, rc;
                    ck[2] = larger(pc & 0xffffu, pc >> 16, ra);
                    g[3] = 2 * g2 + (ra ? 1 : 0);
                    const unsigned int pg = ra ? pr2 : pl;
                    ck[3] = larger(pg & 0xffffu, pg >> 16, rb);
                    g[4] = 2 * g[3] + (rb ? 1 : 0);
                    const unsigned int pa = rb ? p1 : p0, pb = rb ? p3 : p2;
                    const unsigned int pgg = ra ? pb : pa;
                    ck[4] = larger(pgg & 0xffffu, pgg >> 16, rc);
                    g[5] = 2 * g[4] + (rc ? 1 : 0);
                }
                {
                    const unsigned int pc = H.pair(g[5]), pl = H.pair(2 * g[5]), pr2 = H.pair(2 * g[5] + 1);
                    bool ra, rb;
                    ck[5] = larger(pc & 0xffffu, pc >> 16, ra);
                    g[6] = 2 * g[5] + (ra ? 1 : 0);
                    const unsigned int pg = ra ? pr2 : pl;
                    ck[6] = larger(pg & 0xffffu, pg >> 16, rb);
                    g[7] = 2 * g[6] + (rb ? 1 : 0);
                }
                // the old last element climbs back from the leaf hole past every moved child that is smaller; the moved
                // children are non-increasing down the path, so "it passes level l" is monotone in l and the value that
                // ends up on level l is the median of (child moved from l, child moved from l + 1 ... ) -- see above
                unsigned int v[8];
                if (TIES) {
                    bool c[7];
#pragma unroll
                    for (int l = 0; l < 7; l++) c[l] = lt(ck[l], last);
#pragma unroll
                    for (int l = 0; l < 8; l++) {
                        if (l == 0) v[l] = c[0] ? last : ck[0];
                        else if (l == 7) v[l] = c[6] ? ck[6] : last;
                        else v[l] = c[l - 1] ? ck[l - 1] : (c[l] ? last : ck[l]);
                    }
                } else {
                    v[0] = max(ck[0], last);
#pragma unroll
                    for (int l = 1; l < 7; l++) v[l] = med3(ck[l - 1], ck[l], last);
                    v[7] = min(ck[6], last);
                }
                r[1] = v[0];
                r[2] = right0 ? r[2] : v[1];
                r[3] = right0 ? v[1] : r[3];
#pragma unroll
                for (int j = 4; j < 8; j++) r[j] = g[2] == j ? v[2] : r[j];
#pragma unroll
                for (int l = 3; l < 8; l++)
                    if (l < 6 || g[l] < K) H.set(g[l], v[l]);  // levels 3..5 are always inside the heap
                // push_heap: the new element climbs from slot K along 100, 50, 25, 12, 6, 3, 1
                static_assert(K == 200, "ancestor chain of slot K");
                const unsigned int qv[7] = {r[1], r[3], r[6], g[3] == 12 ? v[3] : q12, g[4] == 25 ? v[4] : q25,
                                            g[5] == 50 ? v[5] : q50, g[6] == 100 ? v[6] : q100};
                constexpr int chain[7] = {1, 3, 6, 12, 25, 50, 100};
                if (TIES) {
                    bool pp[7];
#pragma unroll
                    for (int u = 0; u < 7; u++) pp[u] = lt(qv[u], y);
                    r[1] = pp[0] ? y : r[1];
                    r[3] = pp[1] ? (pp[0] ? qv[0] : y) : r[3];
                    r[6] = pp[2] ? (pp[1] ? qv[1] : y) : r[6];
#pragma unroll
                    for (int u = 3; u < 7; u++)
                        if (pp[u]) H.set(chain[u], pp[u - 1] ? qv[u - 1] : y);
                    last = pp[6] ? qv[6] : y;
                } else {
                    // chain position u receives its parent's value if that is below y, y if only its own is, and keeps its
                    // own otherwise: the median of (parent, own, y), the chain being non-increasing towards the leaf
                    r[1] = max(qv[0], y);
                    r[3] = med3(qv[0], qv[1], y);
                    r[6] = med3(qv[1], qv[2], y);
#pragma unroll
                    for (int u = 3; u < 7; u++) H.set(chain[u], med3(qv[u - 1], qv[u], y));
                    last = min(qv[6], y);
                }
            }
        };
        // A tile of 32 candidates (what lies between two checkpoints) is four blocks of eight, each block's rounds unrolled:
        // a round takes its rank out of the block's registers with one instruction, and the checkpoint is written once per
        // tile (round 3 shifted the block by 16 bits and tested for the checkpoint in every round: a dozen instructions of ~130)
        static_assert(TILE == 32, "four blocks of eight candidates per checkpoint");
#pragma clang loop unroll(disable)
        for (int p0 = K; p0 < nmax; p0 += TILE) {
#pragma clang loop unroll(disable)
            for (int pb = p0; pb < min(p0 + TILE, nmax); pb += 8) {
                const uint4 w8 = cur;
                cur = nxt;
                // may run past this agent's candidates: the array ends in slack.  Every lane, idle ones too (they read row 0):
                // a merge with the old value would wait for the data
                nxt = GD_NT_REPLAY_LOAD ? stream_load(blocks + ((pb - K) >> 3) + 2) : blocks[((pb - K) >> 3) + 2];
                const unsigned int wd[4] = {w8.x, w8.y, w8.z, w8.w};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const unsigned int y = (k & 1) ? wd[k >> 1] >> 16 : wd[k >> 1] & 0xffffu;
                    if (pb + k < n && lt(y, r[1])) insert(y);
                }
            }
            // after candidate (p0 - K) + 32: a checkpoint of every standard agent, of a long list at every second tile
            const int after = p0 - K + TILE;
            if (p0 + TILE - 1 < n && (after & ((1 << tsh) - 1)) == 0) cpe[after >> tsh] = (unsigned short)r[1];
        }
    };
    if (__ballot(on && has_tie) != 0ull) replay(std::true_type{});
    else replay(std::false_type{});
#ifdef GD_CLOCKS
    {
        const int tot_ins = wave_sum(n_ins), tot_cand = wave_sum(on ? n - K : 0), longest = wave_max(on ? n - K : 0);
        if (lane == 0) {
            atomicAdd(&d.rk_hist[526], tot_ins);
            atomicAdd(&d.rk_hist[525], tot_cand);
            if (blockIdx.x == 0) atomicAdd(&d.rk_hist[524], longest);  // rounds of the longest wave
        }
    }
#endif
    // ---- the heap array, slot order, for k_knn_finish ----
    if (lane < AWR) {
        H.set(K, last);
#pragma unroll
        for (int j = 1; j < 8; j++) H.set(j, r[j]);
    }
    if (on) {
        unsigned int *out = d.rk_heap + (size_t)i * GD_RANK_HEAP_DW;
#pragma clang loop unroll(disable)
        for (int j = 0; j < GD_RANK_HEAP_DW; j += 4) {
            uint4 v;
            v.x = j + 0 <= K / 2 ? H.pair(j + 0) : 0u;
            v.y = j + 1 <= K / 2 ? H.pair(j + 1) : 0u;
            v.z = j + 2 <= K / 2 ? H.pair(j + 2) : 0u;
            v.w = j + 3 <= K / 2 ? H.pair(j + 3) : 0u;
            stream_store(v, reinterpret_cast<uint4 *>(out + j));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_knn_finish: one wave per agent.  Heap array (ranks) -> road indices, radiusFilter, the checkpoints' K-th keys, and the
// hand-over to k_map_rows: the selected roads in ASCENDING road index with the output row each belongs in.
// Everything it reads is a contiguous piece of the agent's scratch rows: the heap (416 B), the first SPL sorted slots
// (512 B), the key table (336 B), the checkpoint ranks (80 B).
// ------------------------------------------------------------------------------------------------------------------
template <int A_T>
__global__ __launch_bounds__(256) void k_knn_finish(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = blockIdx.x * 4 + wave;
    // the replay order: exactly the agents whose replay has run.  (The grid is sized for every live agent; the entry is
    // requested together with the count -- behind the count's end the array holds agent slots of earlier selections or zeros,
    // valid addresses all -- so that the wave's loads below are one round trip behind this one, not two)
    const int ranked_agents = d.rk_hist[512];
    const int i = d.rk_order[min(li, d.W * A_T - 1)];
    // Everything the wave reads from global memory depends on `i` alone and every address is valid for any agent slot (stale
    // at worst): requested together, before the first branch looks at any of it -- the kernel is a chain of memory round
    // trips with little arithmetic in between (round 3's version asked for flag, count, pose, heap and table one after the
    // other, each behind a branch on the one before)
    constexpr int NP = (K + 63) / 64;
    const int w = i / A_T;
    const int fell_back = d.rk_fallback[i / 32];
    const int packed = d.rk_n[i];
    const int r0 = d.road_off[w], r1 = d.road_off[w + 1];
    const float ex = d.px[i], ey = d.py[i];
    const float qw = d.qw[i], qz = d.qz[i];
    const uint32_t steps_left = d.steps[i];
    const int long_slot = d.rk_longslot[i];  // (meaningful only for an agent of the long-list instantiation: bit 29 of rk_n)
    const unsigned int *hp = d.rk_heap + (size_t)i * GD_RANK_HEAP_DW;
    unsigned int hpair[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) hpair[ps] = GD_NT_FINISH_LOAD ? stream_load(hp + min((ps * 64 + lane + 1) >> 1, GD_RANK_HEAP_DW - 1)) : hp[min((ps * 64 + lane + 1) >> 1, GD_RANK_HEAP_DW - 1)];
    static_assert(SPL == 256, "four slots per lane");
    const uint2 spc4 = GD_NT_FINISH_LOAD ? stream_load(reinterpret_cast<const uint2 *>(d.rk_spc + (size_t)i * SPL) + lane) : reinterpret_cast<const uint2 *>(d.rk_spc + (size_t)i * SPL)[lane];
    const unsigned int cp_e = d.rk_cpe[(size_t)i * NCP + min(lane, NCP - 1)];
    const unsigned short cp_r = d.cp_road[(size_t)i * NCP + min(lane, NCP - 1)];
    // (the compiler otherwise sinks each of these loads below the early returns that do not need it: flag, branch, count,
    // branch, the rest -- three round trips instead of one.  Naming them all as inputs here keeps them together above the
    // first branch)
    static_assert(NP == 4, "operand list");
    asm volatile("" ::"v"(fell_back), "v"(packed), "v"(ex), "v"(ey), "v"(qw), "v"(qz), "v"(steps_left), "v"(long_slot), "v"(hpair[0]),
                 "v"(hpair[1]), "v"(hpair[2]), "v"(hpair[3]), "v"(spc4.x), "v"(spc4.y), "v"(cp_e), "v"(cp_r), "s"(r0), "s"(r1));
    if (li >= ranked_agents || fell_back != 0) return;  // beyond the order / k_map_obs selects for this group
    // (agents out of reach of every road never get here: k_knn_scan wrote their empty hand-over)
    const int n = packed & 0xffff, nle = (packed >> 16) & 0xfff;
    if (packed <= 0 || packed == RK_FAR || n < K) return;  // took a place in the order, then fell back (equal keys)
    // the checkpoints' K-th keys (see below): the table entry is known as soon as the checkpoint's rank is
    const bool is_long = ((packed >> 29) & 1) != 0;
    const int rsh = is_long ? 4 : 5, tsh = is_long ? 6 : 5;  // rank format and checkpoint spacing (RankGeo)
    const unsigned int tm = (1u << rsh) - 1u;
    const int cp_slot = (int)(cp_e >> rsh) - 1 + (int)(cp_e & tm);
    const int ncp = 1 + ((n - K) >> tsh);
    const float *kt_row = is_long ? d.rk_kt_long + (size_t)long_slot * GD_RANK_KT_LONG : d.rk_kt + (size_t)i * KT;
    float cp_t = 0.f;
    if (lane < ncp) cp_t = kt_row[min(max((cp_slot + 15) >> 4, 0), (n + 15) >> 4)];
    __shared__ unsigned short s_spc[4][SPL];
    __shared__ unsigned short s_don[4][K];
    __shared__ unsigned int s_bm[4][GD_RANK_NCH];      // one bit per road of the world: selected
    __shared__ unsigned short s_wp[4][GD_RANK_NCH];    // selected roads below each bitmap word
    __shared__ unsigned int s_sorted[4][K];            // ascending road index -> road | output row << 16
    unsigned short *don = s_don[wave];
    unsigned int *bm = s_bm[wave];
    const unsigned long long lower = (1ull << lane) - 1ull;
    {
        reinterpret_cast<uint2 *>(s_spc[wave])[lane] = spc4;
#pragma unroll
        for (int k = 0; k < GD_RANK_NCH / 64; k++) bm[k * 64 + lane] = 0u;
    }
    wave_sync();

    // heap array -> road indices and in-radius flags (src/knn.hpp:88: length() <= radius; on ranks: fewer than `nle`
    // candidates have a smaller key)
    int road[NP];
    bool inr[NP];
    unsigned long long fl[NP];
    int m = 0;
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int t = ps * 64 + lane, g = t + 1;
        road[ps] = 0;
        inr[ps] = false;
        if (t < K) {
            const unsigned int p2 = hpair[ps];
            const unsigned int e = (g & 1) ? p2 >> 16 : p2 & 0xffffu;
            const int less = (int)(e >> rsh) - 1;
            // (a rank of 0 -- an element the replay never wrote -- would ask for slot -1; an element of the final heap has
            // fewer than K keys below it and at most 31 equal ones before it)
            road[ps] = s_spc[wave][audited(d, less + (int)(e & tm), min(n, SPL))];
            inr[ps] = less < nle;
        }
        fl[ps] = __ballot(inr[ps]);
        m += __popcll(fl[ps]);
    }
    // radiusFilter's swap-remove loop: with m in-radius elements, the out-of-radius slots below m (ascending) receive
    // the in-radius elements of [m, K) in DESCENDING slot order; everything else below m stays
    if (m < K) {
        int above = 0;  // in-radius slots above the current pass
#pragma unroll
        for (int ps = NP - 1; ps >= 0; ps--) {
            const int t = ps * 64 + lane;
            if (t < K && t >= m && inr[ps]) don[above + __popcll(fl[ps] & ~lower & ~(1ull << lane))] = (unsigned short)road[ps];
            above += __popcll(fl[ps]);
        }
        wave_sync();
        int before = 0;  // slots below the current pass
        int in_before = 0;
#pragma unroll
        for (int ps = 0; ps < NP; ps++) {
            const int t = ps * 64 + lane;
            if (t < m && !inr[ps]) {
                const int h = (t - before) - __popcll(fl[ps] & lower) + (before - in_before);  // out-of-radius slots below t
                road[ps] = don[h];
            }
            before += 64;
            in_before += __popcll(fl[ps]);
        }
    }
    // ---- hand-over: the m selected roads in ascending road index, each with its output row (k_map_rows gathers the
    // 32-byte records in that order and puts the rows where they belong).  A bitmap of the world's roads, prefix counts
    // per word, place = selected roads below. ----
    const int R = r1 - r0;
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int t = ps * 64 + lane;
        if (t < m) atomicOr(&bm[audited(d, road[ps], R) >> 5], 1u << (road[ps] & 31));
    }
    wave_sync();
    {
        constexpr int WL = GD_RANK_NCH / 64;  // bitmap words per lane: lane l owns words l * WL ..
        unsigned int own[WL];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < WL; k++) { own[k] = bm[lane * WL + k]; sum += __popc(own[k]); }
        int run = wave_incl_scan(sum) - sum;
#pragma unroll
        for (int k = 0; k < WL; k++) { s_wp[wave][lane * WL + k] = (unsigned short)run; run += __popc(own[k]); }
    }
    wave_sync();
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int t = ps * 64 + lane;
        if (t < m) {
            const int rd = min(max(road[ps], 0), R - 1);
            const int place = (int)s_wp[wave][rd >> 5] + __popc(bm[rd >> 5] & ((1u << (rd & 31)) - 1u));
            s_sorted[wave][audited(d, place, m)] = (unsigned int)rd | (unsigned int)t << 16;
        }
    }
    wave_sync();
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int q = ps * 64 + lane;
        if (q < K) {
            const unsigned int v = q < m ? s_sorted[wave][q] : (unsigned int)q << 16;
            d.sel_idx[(size_t)i * K + q] = (unsigned short)(v & 0xffffu);
            d.sel_slot[(size_t)i * K + q] = (unsigned char)(v >> 16);
        }
    }
    // the K-th distances at this selection's checkpoints: a key that is not below that of the element whose rank was on top
    // -- from the key table, the candidates' key at the next multiple of 16 sorted slots (any upper bound of the K-th
    // distance keeps the next selection's candidates a superset; this one is at most 15 ranks loose)
    // (a second copy is kept of the selection made at the start of an episode: a reset puts the agent back there)
    const bool at_start = steps_left == (uint32_t)GD_EPISODE_LEN;
    const size_t WA = (size_t)d.W * A_T;
    if (lane < ncp) {
        (void)audited(d, cp_slot, n);  // (a rank of 0 on top of the heap at a checkpoint: never written by the replay)
        d.cp_T[(size_t)i * NCP + lane] = cp_t;
        if (at_start) {
            d.cp_T[(WA + i) * NCP + lane] = cp_t;
            d.cp_road[(WA + i) * NCP + lane] = cp_r;
        }
    }
    if (lane == 0) {
        d.rk_streak[i / 32] = 0;  // the group got through without the fallback
        d.cp_hdr[i] = make_float4(ex, ey, __int_as_float(ncp), 0.f);
        if (at_start) d.cp_hdr[WA + i] = make_float4(ex, ey, __int_as_float(ncp), 0.f);
        d.sel_hdr[(size_t)i * 2] = make_float4(ex, ey, qw, qz);
        d.sel_hdr[(size_t)i * 2 + 1] = make_float4(__int_as_float(m), __int_as_float(r0), __int_as_float(1), 0.f);
    }
}

}  // namespace

#ifndef GD_RANK_LONG_GRID
#define GD_RANK_LONG_GRID 2048  // (an empty launch costs 5 us whatever its size; 256 waves ranked the long lists of the unreduced Waymo tiles in 614 us, 2048 in 157)
#endif
void launch_map_obs_rank(const DevSim &d, hipStream_t st) {
    if (d.live_count == 0) return;
    // the long-list ranking: persistent waves (eight per CU fit) over however many agents the standard launch passed on (none:
    // every wave reads the empty count and leaves)
    const dim3 glong(std::min(std::max(d.rk_nlong, 1), GD_RANK_LONG_GRID));
    const dim3 gr(std::min((d.live_count + 7) / 8 * 8, 256 * 8 * 4)), g4((d.live_count + 3) / 4), gw(d.W * (d.A / 64));  // rank: 8192 persistent waves, 16 per CU resident (LDS)
    if (d.A == 64) {
        hipLaunchKernelGGL((k_knn_scan<64>), gw, dim3(256), 0, st, d);
        hipLaunchKernelGGL((k_knn_rank<64, CAP>), gr, dim3(64), 0, st, d);
        if (d.rk_nlong > 0) hipLaunchKernelGGL((k_knn_rank<64, CAP_LONG>), glong, dim3(64), 0, st, d);
    } else {
        hipLaunchKernelGGL((k_knn_scan<128>), gw, dim3(256), 0, st, d);
        hipLaunchKernelGGL((k_knn_rank<128, CAP>), gr, dim3(64), 0, st, d);
        if (d.rk_nlong > 0) hipLaunchKernelGGL((k_knn_rank<128, CAP_LONG>), glong, dim3(64), 0, st, d);
    }
    hipLaunchKernelGGL(k_knn_order, dim3((d.live_count + 255) / 256), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_knn_replay, dim3((d.live_count + AWR - 1) / AWR), dim3(64), 0, st, d);
    if (d.A == 64) hipLaunchKernelGGL((k_knn_finish<64>), g4, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_knn_finish<128>), g4, dim3(256), 0, st, d);
}

}  // namespace gd
