// Road observations: collectMapObservationsSystem (reference src/sim.cpp:242-280) with
// selectKNearestRoadEntities (src/knn.hpp:103-158) on the SGI heap (src/binary_heap.hpp).
//
// The output row ORDER of the reference is an artefact of the heap's history: every candidate
// that was ever closer than the running K-th distance is pushed, including ones evicted later,
// and the final array layout depends on all of them.  It cannot be recovered from the final
// top-K set, so the kernel replays the exact insert sequence per agent:
//
//   * one 200-entry heap per agent in LDS, slot-major (keys[slot][agent]) so that an agent's
//     lane always hits its own bank whatever slot it touches;
//   * workgroups of NW waves, 16 agents per wave, 4 lanes per agent; a world with more than 16 NW
//     agent slots is several workgroups, placed on one XCD;
//   * SCAN (all 64 lanes): roads are processed in windows (2 x 32 over the first 512 roads, then
//     14 x 32); each 32-road chunk is staged in a per-wave LDS tile (one coalesced load issued two
//     chunks ahead; road bytes reach HBM once per world), every lane tests its 8 roads of the chunk
//     against the agent's K-th distance at the window start (a conservative superset of the true
//     inserts), and two cross-lane ORs fold the results into one candidate word per (agent, chunk);
//   * DRAIN (one lane per agent): each agent walks the set bits of its window words at its own
//     pace, re-tests the candidate against the live K-th distance and replays the reference's
//     pop_heap / push_heap as straight-line code (HeapCol::replace_top).
//
// What bounds it: LDS holds 128 agents per CU, every insert of an agent is a
// link of a serial chain, and one wave issues one vector instruction per 4 cycles whatever else is
// resident: time = 2 generations x lock-step rounds per wave x instructions per round x 4 cycles.
// The wave never synchronises with the other waves of the workgroup until the final write-out.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "engine.hpp"
#include "gd_math.hpp"

#ifndef GD_MAP_OBS_NW
#define GD_MAP_OBS_NW 4
#endif

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;

template <int S>
struct HeapCol {
    float *keys;          // column base: element s at keys[s * S]
    unsigned short *idx;
    __device__ __forceinline__ float key(int s) const { return keys[s * S]; }
    __device__ __forceinline__ unsigned short index(int s) const { return idx[s * S]; }
    __device__ __forceinline__ void move(int dst, int src) const {
        keys[dst * S] = keys[src * S];
        idx[dst * S] = idx[src * S];
    }
    __device__ __forceinline__ void set(int s, float k, unsigned short r) const {
        keys[s * S] = k;
        idx[s * S] = r;
    }
    // __push_heap, src/binary_heap.hpp:34-45.  Key and index of the parent are fetched together so
    // that each level costs one LDS round trip.
    __device__ __forceinline__ void push(int hole, int top, float xk, unsigned short xi) const {
        int parent = (hole - 1) / 2;
        while (hole > top) {
            const float pk = key(parent);
            const unsigned short pi = index(parent);
            if (!(pk < xk)) break;
            set(hole, pk, pi);
            hole = parent;
            parent = (hole - 1) / 2;
        }
        set(hole, xk, xi);
    }
    // __adjust_heap with comparator, src/binary_heap.hpp:112-130.  Both children (key + index) are
    // fetched in one batch per level; the stores need no wait, so the dependent chain is one LDS
    // round trip per level.
    __device__ __forceinline__ void adjust(int hole, int len, float xk, unsigned short xi) const {
        const int top = hole;
        int second = 2 * hole + 2;
        while (second < len) {
            const float kr = key(second), kl = key(second - 1);
            const unsigned short ir = index(second), il = index(second - 1);
            const bool left = kr < kl;
            set(hole, left ? kl : kr, left ? il : ir);
            second -= left ? 1 : 0;
            hole = second;
            second = 2 * (second + 1);
        }
        if (second == len) {
            move(hole, second - 1);
            hole = second - 1;
        }
        push(hole, top, xk, xi);
    }
    // make_heap, src/binary_heap.hpp:170-185, serial form
    __device__ __forceinline__ void make(int len) const {
        for (int parent = (len - 2) / 2; parent >= 0; parent--) adjust(parent, len, key(parent), index(parent));
    }
    // make_heap with `g` cooperating lanes (this lane is number `sub`).  The reference sifts parents
    // 99, 98, ..., 0 in turn; parents on one tree level own disjoint subtrees, so they commute and
    // can be sifted concurrently, level by level from the bottom, with the identical result.
    // Callers put a wave_sync() between levels (see k_map_obs).
    __device__ __forceinline__ void make_level(int len, int depth, int sub, int g) const {
        const int first = (1 << depth) - 1;
        const int last = min((2 << depth) - 2, (len - 2) / 2);
        for (int parent = first + sub; parent <= last; parent += g) adjust(parent, len, key(parent), index(parent));
    }
    // pop_heap + replace last + push_heap (src/knn.hpp:138-151) as straight-line code.
    //
    // pop_heap moves the root to slot K-1 (overwritten by the new element right after, so that store
    // is dropped), sifts the hole down from the root of the 199-element heap (always 6 levels, since
    // second = 2h+2 <= 126 < 199, and a 7th iff the level-6 node has children, h <= 98; `second == len`
    // cannot happen: second is even, len is odd), then pushes the old last element x up from the leaf
    // hole.  The values x meets on its way up are exactly the children just moved, which are still in
    // registers, so its climb needs no LDS read.  push_heap then lifts the new element y from slot K-1
    // along the fixed ancestor chain 99,49,24,11,5,2,0.  The caller keeps those seven slots in
    // registers across calls (`qk`/`qi`, level l = slot Q[l]; load_chain() fills them); they are
    // patched where the pop rewrote them and come back holding the chain's new contents, qk[0] being
    // the new root key.  There are no data-dependent loops and no serial predicate chains.
    static constexpr int Q[7] = {0, 2, 5, 11, 24, 49, 99};
    __device__ __forceinline__ void load_chain(float (&qk)[7], unsigned int (&qi)[7], float &lk, unsigned int &li) const {
#pragma unroll
        for (int l = 0; l < 7; l++) { qk[l] = key(Q[l]); qi[l] = index(Q[l]); }
        lk = key(K - 1);
        li = index(K - 1);
    }
    // `lk` / `li`: slot K-1, which only this function touches during a replay, also stays in registers.
    __device__ __forceinline__ void replace_top(float yk, unsigned int yi, float (&qk)[7], unsigned int (&qi)[7], float &lk,
                                                unsigned int &li) const {
        int ph[8];
        float ck[7];
        unsigned int ci[7];
        ph[0] = 0;
        // Decisions use keys only, two levels per LDS round trip: the hole's two children AND its four
        // grandchildren are fetched together, so the 6 unconditional levels cost 3 dependent round
        // trips; the indices of the chosen children are fetched afterwards in one batch.
#pragma unroll
        for (int l = 0; l < 6; l += 2) {
            const int c2 = 2 * ph[l] + 2;                    // children c2-1, c2
            const float kr = key(c2), kl = key(c2 - 1);
            const float g0 = key(2 * c2 - 1), g1 = key(2 * c2), g2 = key(2 * c2 + 1), g3 = key(2 * c2 + 2);
            const bool left = kr < kl;                       // children of (c2-1): 2c2-1, 2c2; of c2: 2c2+1, 2c2+2
            ck[l] = left ? kl : kr;
            ph[l + 1] = c2 - (left ? 1 : 0);
            const float hr = left ? g1 : g3, hl = left ? g0 : g2;
            const int d2 = 2 * ph[l + 1] + 2;
            const bool left2 = hr < hl;
            ck[l + 1] = left2 ? hl : hr;
            ph[l + 2] = d2 - (left2 ? 1 : 0);
        }
        // 7th level iff the level-6 node has children.  Without it the reads are clamped into the
        // array, ck[6] = -1 lets x rest at level 6 at the latest, and the level-7 store repeats level 6.
        const int second6 = 2 * ph[6] + 2;
        const bool has7 = second6 < K - 1;
        {
            const int s6 = has7 ? second6 : K - 2;
            const float kr = key(s6), kl = key(s6 - 1);
            const bool left = kr < kl;
            ck[6] = has7 ? (left ? kl : kr) : -1.f;
            ph[7] = s6 - (left ? 1 : 0);
            ci[6] = index(ph[7]);
        }
#pragma unroll
        for (int l = 0; l < 6; l++) ci[l] = index(ph[l + 1]);
        // x (the old last element) climbs from the leaf hole past every moved child that is smaller.
        // The moved children are non-increasing down the path (heap invariant), so "x passes level l"
        // is the monotone predicate c[l] = ck[l] < lk and needs no serial chain:
        //   slot ph[l] <- ck[l-1] if c[l-1]            (x went above: the moved child stays one lower)
        //              <- x       if c[l] && !c[l-1]
        //              <- ck[l]   otherwise            (x rests below)
        // For the keys this is the median of (ck[l-1], ck[l], lk) since ck[l-1] >= ck[l].
        bool c[8];
#pragma unroll
        for (int l = 0; l < 7; l++) c[l] = ck[l] < lk;
        c[7] = true;
        float nk[8];
        unsigned int ni[8];
        const float inf = __builtin_inff();
#pragma unroll
        for (int l = 0; l < 8; l++) {
            nk[l] = __builtin_amdgcn_fmed3f(l > 0 ? ck[l - 1] : inf, l < 7 ? ck[l] : -1.f, lk);
            ni[l] = c[l] ? li : (l < 7 ? ci[l] : 0u);
            if (l > 0) ni[l] = c[l - 1] ? ci[l - 1] : ni[l];
        }
#pragma unroll
        for (int l = 0; l < 7; l++) set(ph[l], nk[l], (unsigned short)ni[l]);
        set(has7 ? ph[7] : ph[6], has7 ? nk[7] : nk[6], (unsigned short)(has7 ? ni[7] : ni[6]));
        // current values of the ancestor chain of slot K-1 (patched where the pop rewrote a slot)
        qk[0] = nk[0]; qi[0] = ni[0];
#pragma unroll
        for (int l = 1; l < 7; l++) {
            const bool rewritten = ph[l] == Q[l];  // ph[l] lives on level l, like Q[l]
            qk[l] = rewritten ? nk[l] : qk[l];
            qi[l] = rewritten ? ni[l] : qi[l];
        }
        // y climbs from slot K-1 along the chain; the chain is non-increasing towards the leaf, so
        // p[u] = qk[u] < yk is monotone as well: chain position u (7 = slot K-1) receives
        // q[u-1] if p[u-1], y if p[u] && !p[u-1], and keeps its value otherwise (median again).
        bool p[8];
#pragma unroll
        for (int u = 0; u < 7; u++) p[u] = qk[u] < yk;
        p[7] = true;
        float ok[8];
        unsigned int oi[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            ok[u] = __builtin_amdgcn_fmed3f(u > 0 ? qk[u - 1] : inf, u < 7 ? qk[u] : -1.f, yk);
            oi[u] = p[u] ? yi : (u < 7 ? qi[u] : 0u);
            if (u > 0) oi[u] = p[u - 1] ? qi[u - 1] : oi[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) set(u == 7 ? K - 1 : Q[u], ok[u], (unsigned short)oi[u]);
#pragma unroll
        for (int u = 0; u < 7; u++) { qk[u] = ok[u]; qi[u] = oi[u]; }
        lk = ok[7]; li = oi[7];
    }
    // radiusFilter, src/knn.hpp:83-97 (swap-remove in heap-array order); returns newBeyond
    __device__ __forceinline__ int radius_filter(int len, float radius) const {
        int beyond = len, s = 0;
        while (s < beyond) {
            if (sqrtf(key(s)) <= radius) { ++s; continue; }
            --beyond;
            move(s, beyond);
        }
        return beyond;
    }
};

// Intra-wave ordering point for LDS traffic between lanes of one wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Row write-out shared by both implementations: one thread per (agent, slot); a wave writes 64
// consecutive 36-byte rows.
// Agents a0 .. a0+na-1 of the world are columns 0 .. na-1 of `s_count`; idx_of(col, s) is the road index in slot s.
template <int A_T, typename IdxOf>
__device__ __forceinline__ void write_rows(const DevSim &d, int w, int a0, int na, int r0, bool knn, IdxOf idx_of,
                                           const int *s_count, int tid, int nthreads) {
    const int rows = na * K;
    float *out = d.agent_map + ((size_t)w * A_T + a0) * K * 9;
    for (int p = tid; p < rows; p += nthreads) {
        const int col = p / K, s = p - col * K;
        float *o = out + (size_t)p * 9;
        if (s >= s_count[col]) {
            // k-NN pads with fillZeros (id 0, mapType 0: src/knn.hpp:19-28); the linear scan pads with
            // MapObservation::zero() (id -1, mapType -1: src/sim.cpp:277-279)
            const float pad = knn ? 0.f : -1.f;
            o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = (float)ET_None; o[7] = pad; o[8] = pad;
            continue;
        }
        const size_t ei = (size_t)w * A_T + a0 + col;
        const int r = r0 + idx_of(col, s);
        const float2 xy = d.road_xy[r];
        const float4 q0 = d.road_aux[(size_t)r * 2], q1 = d.road_aux[(size_t)r * 2 + 1];
        const Quat einv = quat_inv(quat_from_wz(d.qw[ei], d.qz[ei]));
        const V2 rel = ego_relative(d.px[ei], d.py[ei], einv, xy.x, xy.y);
        o[0] = rel.x; o[1] = rel.y;
        o[2] = q0.z; o[3] = q0.w; o[4] = q1.x;
        o[5] = quat_to_yaw_row(quat_mul(einv, quat_from_wz(q0.x, q0.y)));
        o[6] = q1.y; o[7] = q1.z; o[8] = q1.w;
    }
}

// ---- reference row order: workgroups of NW waves, 16 agents per wave, 4 lanes per agent ----
// A world is A/(16 NW) workgroups.  They sit 8 block ids apart, i.e. on the same XCD (block ids go round-robin
// over the 8 XCDs), so a world's road stream is served by one L2.
template <int A_T, int NW, int WW>
__global__ __launch_bounds__(NW * 64) void k_map_obs(DevSim d) {
    constexpr int C = 32;         // roads per chunk = one mask word
    constexpr int APW = 16;       // agents per wave
    constexpr int G = 64 / APW;   // lanes per agent
    constexpr int BA = NW * APW;  // agents (LDS heap columns) per workgroup
    constexpr int BPW = A_T / BA; // workgroups per world
    static_assert(BA * BPW == A_T && C % G == 0, "geometry");
    const int tid = threadIdx.x;
    const int w = ((int)blockIdx.x / (8 * BPW)) * 8 + ((int)blockIdx.x & 7);
    const int a0 = (((int)blockIdx.x >> 3) % BPW) * BA;
    if (w >= d.W) return;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int wave = tid >> 6, lane = tid & 63;
    const int al = lane % APW, sub = lane / APW;
    const int col = wave * APW + al;  // heap column of this agent
    const int a = a0 + col;
    const int n = d.shape[w * 2 + 0];
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const bool live = a < n;
    const size_t i = (size_t)w * A_T + a;

    __shared__ float s_keys[K * BA];
    __shared__ unsigned short s_idx[K * BA];
    __shared__ unsigned int s_mask[WW * BA];  // word c of heap column col at [c * BA + col]: candidate bits of chunk c
    __shared__ float2 s_tile[NW][C];
    __shared__ int s_count[BA];

    const float radius = d.p.observationRadius;
    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;
    const HeapCol<BA> heap{s_keys + col, s_idx + col};
    int count = 0;

    if (a0 + wave * APW < n) {  // wave-uniform: waves without live agents skip the scan
        float ex = 0.f, ey = 0.f;
        Quat inv{1.f, 0.f, 0.f, 0.f};
        if (live) {
            ex = d.px[i]; ey = d.py[i];
            inv = quat_inv(quat_from_wz(d.qw[i], d.qz[i]));
        }
        float2 *tile = s_tile[wave];
        // A lane owns PL = 32/G consecutive roads of every 32-road chunk (t = sub*PL + k); its PL pass
        // bits sit at bit sub*PL of the chunk word, and the G lanes of an agent OR their parts together
        // with log2(G) cross-lane steps: no per-road ballot.
        constexpr int PL = C / G;
        auto agent_or = [&](unsigned int part) -> unsigned int {
#pragma unroll
            for (int st = APW; st < 64; st <<= 1) part |= (unsigned int)__shfl_xor((int)part, st);
            return part;
        };

        // the (x, y) stream runs two chunks ahead of the scan in registers (lanes 0..31 carry a chunk)
        const float2 *rxy = d.road_xy + r0;
        auto load_chunk = [&](int base) -> float2 {
            return (lane < C && base + lane < R) ? rxy[base + lane] : make_float2(0.f, 0.f);
        };
        float2 pre1 = load_chunk(0), pre2 = load_chunk(C);
        // short windows (2 chunks) over the first 512 roads, while the K-th distance still falls quickly (right
        // after make_heap every road is a candidate, half of them false): the threshold is refreshed more
        // often where it pays (-2 % synthetic, -4 % Waymo tiles; 1 to 4 chunks over 384 to 1024 roads measured)
        constexpr int EARLY_ROADS = 512, EARLY_CHUNKS = 2;
        int wlen = 0;
        for (int win = 0; win < R; win += wlen) {
            wlen = (win < EARLY_ROADS ? EARLY_CHUNKS : WW) * C;
            const int win_end = min(R, win + wlen);
            float thr = (live && win >= K) ? heap.key(0) : -1.f;
            unsigned int nz = 0;  // bit c: chunk c of this window has candidates for this agent
            // ---- SCAN: all lanes; one mask word per 32-road chunk ----
            for (int base = win, c = 0; base < win_end; base += C, c++) {
                const int tn = min(C, R - base);
                const float2 cur = pre1;
                pre1 = pre2;
                pre2 = load_chunk(base + 2 * C);
                wave_sync();
                if (lane < C) tile[lane] = cur;
                wave_sync();
                if (knn) {
                    // roads with index < K go straight into the array (src/knn.hpp:112-120)
                    const int direct_end = min(tn, max(0, K - base));
                    if (direct_end > 0) {
                        if (live) {
#pragma unroll
                            for (int k = 0; k < PL; k++) {
                                const int t = sub * PL + k;
                                if (t < direct_end) {
                                    const float2 xy = tile[t];
                                    heap.set(base + t, ego_dist2(ex, ey, inv.w, inv.z, xy.x, xy.y), (unsigned short)(base + t));
                                }
                            }
                        }
                        if (base + direct_end == K) {
                            for (int depth = 6; depth >= 0; depth--) {  // parents 0..99 live on levels 0..6
                                wave_sync();
                                if (live) heap.make_level(K, depth, sub, G);
                            }
                            wave_sync();
                            thr = live ? heap.key(0) : -1.f;
                        }
                    }
                    // conservative candidates: closer than the K-th distance at the window start
                    unsigned int part = 0;
                    if (direct_end < tn) {
#pragma unroll
                        for (int k = 0; k < PL; k++) {
                            const int t = sub * PL + k;
                            const float2 xy = tile[t];
                            const bool pass = t >= direct_end && t < tn && ego_dist2(ex, ey, inv.w, inv.z, xy.x, xy.y) < thr;
                            part |= (pass ? 1u : 0u) << t;
                        }
                    }
                    const unsigned int word = agent_or(part);
                    if (sub == 0) s_mask[c * BA + col] = word;
                    nz |= (word != 0u ? 1u : 0u) << c;
                } else {
                    // AllEntitiesWithRadiusFiltering: first K in index order within the radius, sim.cpp:261-279
                    unsigned int part = 0;
#pragma unroll
                    for (int k = 0; k < PL; k++) {
                        const int t = sub * PL + k;
                        const float2 xy = tile[t];
                        const bool pass = live && t < tn && !(ego_dist2(ex, ey, inv.w, inv.z, xy.x, xy.y) > d.radius_key_max);
                        part |= (pass ? 1u : 0u) << t;
                    }
                    const unsigned int word = agent_or(part);
                    unsigned int mine = part;
                    while (mine) {
                        const int t = __ffs(mine) - 1;
                        mine &= mine - 1;
                        const int pos = count + __popc(word & ((1u << t) - 1u));
                        if (pos < K) s_idx[pos * BA + col] = (unsigned short)(base + t);
                    }
                    count += __popc(word);
                }
            }
            // ---- DRAIN: one lane per agent replays its candidates of this window in road order ----
            if (knn && win_end > K) {
                wave_sync();
                if (live && sub == 0 && !(d.debug_flags & 2)) {
                    // cursor over the candidate bits: `nz` names the non-empty words, so a lane never
                    // spins over empty ones, and the word after the current one is already in flight;
                    // the next candidate's (x, y) is fetched from L2 while the current one is replayed
                    unsigned int word = 0, wnx = 0;
                    int c = 0, cn = 0;
                    auto fetch = [&]() {
                        wnx = 0;
                        if (nz) {
                            cn = __ffs(nz) - 1;
                            nz &= nz - 1;
                            wnx = s_mask[cn * BA + col];
                        }
                    };
                    fetch();
                    auto next = [&](int &r) -> bool {
                        if (word == 0) {
                            word = wnx;
                            c = cn;
                            fetch();
                        }
                        if (word == 0) return false;
                        const int b = __ffs(word) - 1;
                        word &= word - 1;
                        r = win + c * C + b;
                        return true;
                    };
                    int r_cur = 0, r_nxt = 0;
                    bool has = next(r_cur);
                    float2 xy_cur = make_float2(0.f, 0.f);
                    if (has) xy_cur = rxy[r_cur];
                    float qk[7];  // ancestor chain of slot K-1 incl. the root: qk[0] is the K-th distance
                    unsigned int qi[7];
                    float lk;
                    unsigned int li;
                    heap.load_chain(qk, qi, lk, li);
                    while (has) {
                        const bool has_n = next(r_nxt);
                        float2 xy_nxt = xy_cur;
                        if (has_n) xy_nxt = rxy[r_nxt];
                        const float key = ego_dist2(ex, ey, inv.w, inv.z, xy_cur.x, xy_cur.y);
                        if (key < qk[0]) heap.replace_top(key, (unsigned int)r_cur, qk, qi, lk, li);
                        r_cur = r_nxt;
                        xy_cur = xy_nxt;
                        has = has_n;
                    }
                }
            }
        }
        wave_sync();
        if (live && sub == 0) {
            if (knn) count = heap.radius_filter(min(R, K), radius);
            s_count[col] = min(count, K);
        }
    }
    __syncthreads();
    if (d.debug_flags & 1) return;
    write_rows<A_T>(d, w, a0, max(0, min(BA, n - a0)), r0, knn,
                    [&](int c, int sl) -> int { return s_idx[sl * BA + c]; }, s_count, tid, NW * 64);
}

// ---- set-order mode (gd_config.knn_order = GD_KNN_SET_ORDER) ----
//
// Same row SET as the reference (the K nearest by (distance^2, road index), then the radius filter),
// rows in ascending road index instead of the reference's heap-history order.  Because the radius
// filter runs after the top-K, the result is simply "every in-radius road" whenever at most K roads
// are in radius, and the K smallest of the in-radius roads otherwise; no heap is needed.  Each wave
// takes its agents one at a time with all 64 lanes on the road stream: 64 roads per iteration, ballot
// compaction of the in-radius ones (key, road) into LDS in road order, and -- only when more than K are
// in radius -- an exact selection by bisection on the key bits (ties at the K-th distance go to the
// lowest road index).
template <int A_T, int NW>
__global__ __launch_bounds__(NW * 64) void k_map_obs_set(DevSim d) {
    const int w = blockIdx.x, tid = threadIdx.x;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int wave = tid >> 6, lane = tid & 63;
    const int n = d.shape[w * 2 + 0];
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;

    constexpr int CAP = 1024;  // in-radius candidates a wave can hold in LDS
    __shared__ unsigned short s_idx[K * A_T];
    __shared__ int s_count[A_T];
    __shared__ float s_ckey[NW][CAP];
    __shared__ unsigned short s_cidx[NW][CAP];

    // selection: each wave takes its agents one at a time, all 64 lanes cooperating
    constexpr int APW = A_T / NW;
    float *ckey = s_ckey[wave];
    unsigned short *cidx = s_cidx[wave];
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int al = 0; al < APW; al++) {
        const int a = wave * APW + al;
        if (a >= n) break;  // wave-uniform
        const size_t i = (size_t)w * A_T + a;
        const float ex = d.px[i], ey = d.py[i];
        const Quat inv = quat_inv(quat_from_wz(d.qw[i], d.qz[i]));
        // gather the in-radius candidates (key, road) in road order: lane = road, 64 consecutive roads per
        // iteration in one coalesced load issued an iteration ahead (the world's (x, y) stream is re-read
        // per agent from L1/L2; it reaches HBM once), ballot compaction into the wave's LDS buffer
        const float kmax = d.radius_key_max;
        auto in_radius = [&](float key) -> bool {
            // radiusFilter keeps length() <= radius (src/knn.hpp:88); the linear scan skips length() > radius.
            // sqrtf is monotone and correctly rounded, so both are comparisons of the squared key with
            // radius_key_max, the largest fp32 whose square root is <= radius (computed at gd_create)
            return knn ? (key <= kmax) : !(key > kmax);
        };
        const float2 *rxy = d.road_xy + r0;
        int nin = 0;
        constexpr int U = 4;  // 64-road groups per iteration: U loads in flight per lane, issued an iteration ahead
        float2 nxt[U];
#pragma unroll
        for (int u = 0; u < U; u++) nxt[u] = u * 64 + lane < R ? rxy[u * 64 + lane] : make_float2(0.f, 0.f);
        for (int rb = 0; rb < R; rb += 64 * U) {
            float2 cur[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                cur[u] = nxt[u];
                const int rn = rb + 64 * U + u * 64 + lane;
                if (rn < R) nxt[u] = rxy[rn];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = rb + u * 64 + lane;
                const float key = ego_dist2(ex, ey, inv.w, inv.z, cur[u].x, cur[u].y);
                const bool in = r < R && in_radius(key);
                const unsigned long long b = __ballot(in);
                const int pos = nin + __popcll(b & lower);
                if (in && pos < CAP) { ckey[pos] = key; cidx[pos] = (unsigned short)r; }
                nin += __popcll(b);
            }
        }
        wave_sync();
        int count = 0;
        if (nin <= K || !knn) {
            // every in-radius road (k-NN), or the first K of them (linear scan), in road order
            count = min(nin, K);
            for (int j = lane; j < count; j += 64) s_idx[j * A_T + a] = cidx[j];
        } else if (nin < CAP) {
            // K smallest by (key, road): bisection on the key bits for the K-th smallest key T
            // (largest T with count(key < T) < K), then everything below T plus the earliest ties
            // three probes per pass (counts packed 11 + 11 + 10 bits, nin < 1024: one cross-lane reduction
            // for all three): the search interval shrinks four-fold per pass, 16 passes for the 31 key bits
            unsigned int lo = 0u, hi = 0x7f800000u;
            while (lo < hi) {
                const unsigned int step = (hi - lo + 3u) / 4u;  // >= 1
                const unsigned int p1 = lo + step, p2 = min(hi, p1 + step), p3 = min(hi, p2 + step);
                unsigned int cnt = 0;
                for (int j = lane; j < nin; j += 64) {
                    const unsigned int kb = __float_as_uint(ckey[j]);
                    cnt += (kb < p1 ? 1u : 0u) + (kb < p2 ? 1u << 11 : 0u) + (kb < p3 ? 1u << 22 : 0u);
                }
                for (int off = 32; off > 0; off >>= 1) cnt += (unsigned int)__shfl_xor((int)cnt, off);
                const int c1 = (int)(cnt & 2047u), c2 = (int)((cnt >> 11) & 2047u), c3 = (int)(cnt >> 22);
                if (c3 < K) lo = p3;
                else if (c2 < K) { lo = p2; hi = p3 - 1u; }
                else if (c1 < K) { lo = p1; hi = p2 - 1u; }
                else hi = p1 - 1u;
            }
            int less = 0;
            for (int j = lane; j < nin; j += 64) less += __float_as_uint(ckey[j]) < lo ? 1 : 0;
            for (int off = 32; off > 0; off >>= 1) less += __shfl_xor(less, off);
            int need_ties = K - less;
            for (int jb = 0; jb < nin; jb += 64) {
                const int j = jb + lane;
                const unsigned int kb = j < nin ? __float_as_uint(ckey[j]) : 0xffffffffu;
                const bool tie = kb == lo;
                const unsigned long long tb = __ballot(tie);
                const bool take = kb < lo || (tie && (int)__popcll(tb & lower) < need_ties);
                need_ties -= min(need_ties, __popcll(tb));
                const unsigned long long kb2 = __ballot(take);
                if (take) s_idx[(count + __popcll(kb2 & lower)) * A_T + a] = cidx[j];
                count += __popcll(kb2);
            }
        } else {
            // more in-radius roads than the LDS candidate buffer holds: same selection, keys recomputed
            // from the road stream on every bisection step
            auto key_bits = [&](int r, bool &in) -> unsigned int {
                in = false;
                if (r >= R) return 0xffffffffu;
                const float2 xy = rxy[r];
                const float key = ego_dist2(ex, ey, inv.w, inv.z, xy.x, xy.y);
                in = in_radius(key);
                return in ? __float_as_uint(key) : 0xffffffffu;
            };
            unsigned int lo = 0u, hi = 0x7f800000u;
            while (lo < hi) {
                const unsigned int mid = lo + (hi - lo + 1) / 2;
                int cnt = 0;
                for (int rb = 0; rb < R; rb += 64) { bool in; cnt += __popcll(__ballot(key_bits(rb + lane, in) < mid)); }
                if (cnt < K) lo = mid; else hi = mid - 1;
            }
            int less = 0;
            for (int rb = 0; rb < R; rb += 64) { bool in; less += __popcll(__ballot(key_bits(rb + lane, in) < lo)); }
            int need_ties = K - less;
            for (int rb = 0; rb < R; rb += 64) {
                bool in;
                const unsigned int kb = key_bits(rb + lane, in);
                const bool tie = in && kb == lo;
                const unsigned long long tb = __ballot(tie);
                const bool take = (in && kb < lo) || (tie && (int)__popcll(tb & lower) < need_ties);
                need_ties -= min(need_ties, __popcll(tb));
                const unsigned long long kb2 = __ballot(take);
                if (take) s_idx[(count + __popcll(kb2 & lower)) * A_T + a] = (unsigned short)(rb + lane);
                count += __popcll(kb2);
            }
        }
        if (lane == 0) s_count[a] = min(count, K);
        wave_sync();
    }
    __syncthreads();
    write_rows<A_T>(d, w, 0, n, r0, knn, [&](int c, int sl) -> int { return s_idx[sl * A_T + c]; }, s_count, tid, NW * 64);
}

}  // namespace

void launch_map_obs(const DevSim &d, hipStream_t st) {
    const dim3 grid(d.W);
    if (d.knn_order == GD_KNN_SET_ORDER) {
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs_set<64, 4>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs_set<128, 8>), grid, dim3(512), 0, st, d);
        return;
    }
    constexpr int NW = GD_MAP_OBS_NW;
    const int blocks = ((d.W + 7) / 8) * 8 * (d.A / (16 * NW));
    if (d.A == 64) hipLaunchKernelGGL((k_map_obs<64, NW, 14>), dim3(blocks), dim3(NW * 64), 0, st, d);
    else hipLaunchKernelGGL((k_map_obs<128, NW, 14>), dim3(blocks), dim3(NW * 64), 0, st, d);
}

}  // namespace gd
