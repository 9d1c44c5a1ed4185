// Road observations: collectMapObservationsSystem (reference src/sim.cpp:242-280) with
// selectKNearestRoadEntities (src/knn.hpp:103-158) on the SGI heap (src/binary_heap.hpp).
//
// The output row ORDER of the reference is an artefact of the heap's history: every candidate
// that was ever closer than the running K-th distance is pushed, including ones evicted later,
// and the final array layout depends on all of them.  It cannot be recovered from the final
// top-K set, so the kernel replays the exact insert sequence of every agent.  An agent's inserts are
// a serial chain; what can be made wide is the number of chains a wave advances per instruction:
//
//   * ONE LANE PER AGENT for everything: a workgroup is one wave = 64 agent slots of one world.  The
//     200-entry heap of every agent lives in LDS, slot-major (keys[slot][lane] fp32, idx[slot][lane]
//     u16: 1200 B per agent, 2 waves per CU), so a lane always hits its own bank whatever slot it
//     touches; slots are 1-based (children of g are 2g and 2g+1); slot K (held in registers during
//     the replay), K+1 and K+2 are sentinels (-1) that stand in for the children a level-6 node may
//     not have;
//   * SCAN: roads are wave-uniform data, so their (x, y) pairs arrive through the scalar cache
//     (s_load) and cost no LDS and no vector memory traffic; every lane tests the 32 roads of a chunk
//     against its agent's live K-th distance with a cheap conservative bound (|p - e|^2 with a
//     margin instead of the reference's rotated form) and puts one candidate word per chunk into a
//     16-chunk ring in LDS.  A chunk is scanned when every lane has a free ring slot and a
//     quarter of the live lanes have run dry: lanes drain at their own pace up to 16 chunks apart,
//     which evens out who is busy when (the sum over windows of the busiest lane's count is what a
//     lock-step window scheme pays);
//   * DRAIN: one candidate per lane and round.  A lane pops the next set bit of its ring words,
//     fetches that road's (x, y) (one round ahead), recomputes the key exactly as the reference
//     does (gd_math.hpp ego_dist2), re-tests it against the live K-th distance and replays
//     pop_heap / push_heap as straight-line code (Heap::replace_top): the whole round is one basic
//     block of speculative LDS reads and selects, and only its stores are predicated.
//
// DESIGN.md section 5 has the cost model (rounds x cycles per round of a lone wave, two generations of
// 128 agents per CU) and the measurements.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "engine.hpp"
#include "gd_math.hpp"

#ifdef GD_STAMPS
// diagnostic build only (tools/stamps.sh): per-workgroup cycle counts of the phases of k_map_obs
__device__ unsigned long long g_stamps[8192][8];
extern "C" int gd_debug_read_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * n);
}
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var)
#endif

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr int LW = 64;        // lanes of a wave = agent slots per workgroup of the reference-order kernel
constexpr int SLOTS = K + 2;  // stored slots 1..K (the heap), K+1 and K+2 (sentinels); slot g is row g - 1

// Intra-wave ordering point for LDS traffic between lanes of one wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Column of lane `l` in the u16 index array: lanes l and l + 32 share a dword, so the 32 lanes of
// either half of the wave (the LDS services them separately) touch 32 different banks.
__device__ __forceinline__ int idx_col(int l) { return (l & 31) * 2 + (l >> 5); }

// One agent's heap: a column of the wave's LDS arrays.  Slot g (1-based; the reference's array index is
// g - 1) is at k[g * LW] / i[g * LW].
struct Heap {
    float *k;
    unsigned short *i;
    __device__ __forceinline__ float key(int g) const { return k[g * LW]; }
    __device__ __forceinline__ unsigned int index(int g) const { return i[g * LW]; }
    __device__ __forceinline__ void set(int g, float key, unsigned int idx) const {
        k[g * LW] = key;
        i[g * LW] = (unsigned short)idx;
    }
    __device__ __forceinline__ void move(int dst, int src) const { set(dst, key(src), index(src)); }
    // make_heap, src/binary_heap.hpp:170-185.  The reference sifts parents K/2, ..., 1 in turn; parents on one tree
    // level own disjoint subtrees, so they commute: the heap is built level by level from the bottom, N parents
    // of a level at a time in one instruction stream (their LDS round trips overlap).  Each sift is
    // __adjust_heap as straight-line code, like replace_top below: the hole goes down to the bottom of the
    // K-element heap along the larger child (slots K+1, K+2 hold -1 and stand in for missing children), then
    // the parent's own element x climbs back past every moved child that is smaller (monotone predicate).
    template <int L, int N>
    __device__ __forceinline__ void sift_level(const int (&start)[N]) const {
        int g[N][8];
        float ck[N][7], xk[N];
        unsigned int ci[N][7], xi[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            g[j][L] = start[j];
            xk[j] = key(start[j]);
            xi[j] = index(start[j]);
        }
#pragma unroll
        for (int l = L; l < 7; l++) {
#pragma unroll
            for (int j = 0; j < N; j++) {
                const int cb = l == 6 ? min(2 * g[j][l], K + 1) : 2 * g[j][l];
                const float kl = key(cb), kr = key(cb + 1);
                const bool right = !(kr < kl);
                ck[j][l] = right ? kr : kl;
                g[j][l + 1] = cb + (right ? 1 : 0);
                ci[j][l] = index(g[j][l + 1]);
            }
        }
        const float inf = __builtin_inff();
#pragma unroll
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int l = L; l < 8; l++) {
                const bool c_here = l < 7 ? ck[j][l] < xk[j] : true;
                const float nk = __builtin_amdgcn_fmed3f(l > L ? ck[j][l - 1] : inf, l < 7 ? ck[j][l] : -1.f, xk[j]);
                unsigned int ni = c_here ? xi[j] : ci[j][l < 7 ? l : 6];
                if (l > L) ni = (ck[j][l - 1] < xk[j]) ? ci[j][l - 1] : ni;
                set(g[j][l], nk, ni);
            }
        }
    }
    template <int L, int N>
    __device__ __forceinline__ void make_level(int first, int last) const {
        int p = first;
#pragma clang loop unroll(disable)
        for (; p + N - 1 <= last; p += N) {
            int start[N];
#pragma unroll
            for (int j = 0; j < N; j++) start[j] = p + j;
            sift_level<L, N>(start);
        }
#pragma clang loop unroll(disable)
        for (; p <= last; p++) {
            const int start[1] = {p};
            sift_level<L, 1>(start);
        }
    }
    // len == K (the only length the reference ever heapifies, src/knn.hpp:128)
    __device__ __forceinline__ void make() const {
        make_level<6, 4>(64, K / 2);
        make_level<5, 4>(32, 63);
        make_level<4, 4>(16, 31);
        make_level<3, 4>(8, 15);
        make_level<2, 4>(4, 7);
        make_level<1, 2>(2, 3);
        make_level<0, 1>(1, 1);
    }
    // pop_heap + replace last + push_heap (src/knn.hpp:138-151) as straight-line code.
    //
    // pop_heap moves the root to slot K (overwritten by the new element right after, so that store is
    // dropped), sifts the hole down from the root of the (K-1)-element heap -- always 6 levels, and a 7th
    // iff the level-6 node g has children (2g + 1 <= K - 1); otherwise sentinels are read, whose key -1
    // lets x rest at level 6 at the latest -- then pushes the old last element x up from the leaf hole.
    // The values x meets on its way up are exactly the children just moved, which are still in
    // registers, so its climb needs no LDS read.  push_heap then lifts the new element y from slot K
    // along the fixed ancestor chain 100, 50, 25, 12, 6, 3, 1.  The caller keeps those seven slots in
    // registers across calls (`qk`/`qi`, level l = slot Q[l]; load_chain() fills them); they are
    // patched where the pop rewrote them and come back holding the chain's new contents, qk[0] being
    // the new root key.  Slot K itself lives in `lk`/`li` for the whole replay (store_last() writes it
    // back).  There are no data-dependent loops and no serial predicate chains, every LDS read is
    // speculative (its address is in bounds whatever the lane's state) and only the stores and the
    // register updates depend on `valid`.
    static constexpr int Q[7] = {1, 3, 6, 12, 25, 50, 100};
    __device__ __forceinline__ void load_chain(float (&qk)[7], unsigned int (&qi)[7], float &lk, unsigned int &li) const {
#pragma unroll
        for (int l = 0; l < 7; l++) { qk[l] = key(Q[l]); qi[l] = index(Q[l]); }
        lk = key(K);
        li = index(K);
        k[K * LW] = -1.f;  // slot K lives in registers during the replay; its LDS copy is a third sentinel
    }
    __device__ __forceinline__ void store_last(float lk, unsigned int li) const { set(K, lk, li); }
    __device__ __forceinline__ void replace_top(bool valid, float yk, unsigned int yi, float (&qk)[7], unsigned int (&qi)[7],
                                                float &lk, unsigned int &li) const {
        int g[8];
        float ck[7];
        unsigned int ci[7];
        g[0] = 1;
        // Decisions use keys only, two levels per LDS round trip: the hole's two children AND its four
        // grandchildren are fetched together, so the 6 unconditional levels cost 3 dependent round
        // trips; the index of a chosen child is requested as soon as the child is known.
#pragma unroll
        for (int l = 0; l < 6; l += 2) {
            const int cb = 2 * g[l];  // children cb, cb + 1; grandchildren 2 cb .. 2 cb + 3
            const float kl = key(cb), kr = key(cb + 1);
            const float g0 = key(2 * cb), g1 = key(2 * cb + 1), g2 = key(2 * cb + 2), g3 = key(2 * cb + 3);
            const bool right = !(kr < kl);
            ck[l] = right ? kr : kl;
            g[l + 1] = cb + (right ? 1 : 0);
            ci[l] = index(g[l + 1]);
            const float hl = right ? g2 : g0, hr = right ? g3 : g1;
            const bool right2 = !(hr < hl);
            ck[l + 1] = right2 ? hr : hl;
            g[l + 2] = 2 * g[l + 1] + (right2 ? 1 : 0);
            ci[l + 1] = index(g[l + 2]);
        }
        {
            // children 2g, 2g + 1 <= K - 1, or sentinels: slot K (g = K/2; it holds -1 during the replay), K + 1, K + 2
            const int cb = min(2 * g[6], K + 1);
            const float kl = key(cb), kr = key(cb + 1);
            const bool right = !(kr < kl);
            ck[6] = right ? kr : kl;
            g[7] = cb + (right ? 1 : 0);
            ci[6] = index(g[7]);
        }
        // x (the old last element) climbs from the leaf hole past every moved child that is smaller.
        // The moved children are non-increasing down the path (heap invariant), so "x passes level l"
        // is the monotone predicate c[l] = ck[l] < lk and needs no serial chain:
        //   slot g[l] <- ck[l-1] if c[l-1]            (x went above: the moved child stays one lower)
        //             <- x       if c[l] && !c[l-1]
        //             <- ck[l]   otherwise            (x rests below)
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
        // current values of the ancestor chain of slot K (patched where the pop rewrote a slot)
        float pk[7];
        unsigned int pi[7];
        pk[0] = nk[0]; pi[0] = ni[0];
#pragma unroll
        for (int l = 1; l < 7; l++) {
            const bool rewritten = g[l] == Q[l];  // g[l] lives on level l, like Q[l]
            pk[l] = rewritten ? nk[l] : qk[l];
            pi[l] = rewritten ? ni[l] : qi[l];
        }
        // y climbs from slot K along the chain; the chain is non-increasing towards the leaf, so
        // p[u] = pk[u] < yk is monotone as well: chain position u (7 = slot K) receives
        // q[u-1] if p[u-1], y if p[u] && !p[u-1], and keeps its value otherwise (median again).
        bool p[8];
#pragma unroll
        for (int u = 0; u < 7; u++) p[u] = pk[u] < yk;
        p[7] = true;
        float ok[8];
        unsigned int oi[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            ok[u] = __builtin_amdgcn_fmed3f(u > 0 ? pk[u - 1] : inf, u < 7 ? pk[u] : -1.f, yk);
            oi[u] = p[u] ? yi : (u < 7 ? pi[u] : 0u);
            if (u > 0) oi[u] = p[u - 1] ? pi[u - 1] : oi[u];
        }
        // the register state is updated with selects, outside the predicated region: everything above stays
        // speculative (computed by every lane in one block) and only the stores are predicated
#pragma unroll
        for (int u = 0; u < 7; u++) {
            qk[u] = valid ? ok[u] : qk[u];
            qi[u] = valid ? oi[u] : qi[u];
        }
        lk = valid ? ok[7] : lk;
        li = valid ? oi[7] : li;
        if (valid) {
#pragma unroll
            for (int l = 0; l < 8; l++) set(g[l], nk[l], ni[l]);
#pragma unroll
            for (int u = 0; u < 7; u++) set(Q[u], ok[u], oi[u]);
        }
    }
    // radiusFilter, src/knn.hpp:83-97: swap-remove in heap-array order; returns newBeyond.  `kmax` is the largest
    // fp32 whose correctly rounded square root is <= radius: length() <= radius without the root.
    // The reference walks s up from 0 and, at every out-of-radius slot, pulls elements from the end of the array
    // until an in-radius one lands there.  With m in-radius elements the outcome is: the holes (out-of-radius
    // slots below m, ascending) receive the in-radius elements of [m, len) in DESCENDING slot order, everything
    // else below m stays.  So: one pass for the in-radius flags (`flags`: 7 words of this lane in LDS, word w at
    // flags[w * LW]), then one move per hole.
    __device__ __forceinline__ int radius_filter(int len, float kmax, unsigned int *flags) const {
        int m = 0;
#pragma unroll
        for (int w = 0; w < (K + 31) / 32; w++) {
            unsigned int f = 0;
#pragma unroll
            for (int b = 0; b < 32; b++) {
                const int s = w * 32 + b;
                if (s < K) f |= (s < len && key(s + 1) <= kmax) ? 1u << b : 0u;
            }
            flags[w * LW] = f;
            m += __popc(f);
        }
        if (m < len) {
            auto below = [](int limit, int w) -> unsigned int {  // bits of word w whose slot is < limit
                const int n = limit - w * 32;
                return n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : (1u << n) - 1u);
            };
            int hw = 0, dw = (len - 1) >> 5;
            unsigned int holes = ~flags[0] & below(m, 0);
            unsigned int donors = flags[dw * LW] & ~below(m, dw);
            for (;;) {
                while (holes == 0u && (hw + 1) * 32 < m) {
                    hw++;
                    holes = ~flags[hw * LW] & below(m, hw);
                }
                if (holes == 0u) break;
                while (donors == 0u) {  // as many donors as holes: never runs off the array
                    dw--;
                    donors = flags[dw * LW] & ~below(m, dw);
                }
                const int hole = hw * 32 + __ffs(holes) - 1;
                holes &= holes - 1u;
                const int db = 31 - __clz(donors);
                donors &= ~(1u << db);
                move(hole + 1, dw * 32 + db + 1);
            }
        }
        return m;
    }
};

// ---- row write-out (both selection kernels end by handing their selection to this one) ----
// One thread per (world, agent, slot) row of agent_roadmap_tensor; a wave writes 64 consecutive 36-byte rows.  The row is
// ReferenceFrame::observationOf (src/utils.hpp:36-49) of the selected road, or the padding row.  Splitting it from the
// selection lets the gather / atan2 / store work run at full occupancy instead of behind the LDS-bound selection waves.
template <int A_T>
__global__ __launch_bounds__(256) void k_map_rows(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;  // row index over [W][A][K]
    const size_t wa = p / K;
    const int s = (int)(p - wa * K);
    const int w = (int)(wa / A_T), a = (int)(wa - (size_t)w * A_T);
    if (w >= d.W || a >= d.shape[w * 2 + 0]) return;  // rows of padding agents are written at reset (k_init_padding_rows)
    // The rows are written once and not read again by the step: streaming (nt) stores keep them from pushing the
    // road and agent arrays, which every step re-reads, out of L2 / Infinity Cache.
    float *o = d.agent_map + p * 9;
    auto put = [&](float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8) {
        __builtin_nontemporal_store(v0, o + 0); __builtin_nontemporal_store(v1, o + 1); __builtin_nontemporal_store(v2, o + 2);
        __builtin_nontemporal_store(v3, o + 3); __builtin_nontemporal_store(v4, o + 4); __builtin_nontemporal_store(v5, o + 5);
        __builtin_nontemporal_store(v6, o + 6); __builtin_nontemporal_store(v7, o + 7); __builtin_nontemporal_store(v8, o + 8);
    };
    if (s >= d.sel_count[wa]) {
        // k-NN pads with fillZeros (id 0, mapType 0: src/knn.hpp:19-28); the linear scan pads with
        // MapObservation::zero() (id -1, mapType -1: src/sim.cpp:277-279)
        const float pad = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST ? 0.f : -1.f;
        put(0.f, 0.f, 0.f, 0.f, 0.f, 0.f, (float)ET_None, pad, pad);
        return;
    }
    const int r = d.road_off[w] + (int)d.sel_idx[p];
    const float2 xy = d.road_xy[r];
    const float4 q0 = d.road_aux[(size_t)r * 2], q1 = d.road_aux[(size_t)r * 2 + 1];
    const Quat einv = quat_inv(quat_from_wz(d.qw[wa], d.qz[wa]));
    const V2 rel = ego_relative(d.px[wa], d.py[wa], einv, xy.x, xy.y);
    put(rel.x, rel.y, q0.z, q0.w, q1.x, quat_to_yaw_row(quat_mul(einv, quat_from_wz(q0.x, q0.y))), q1.y, q1.z, q1.w);
}

// Selection hand-over from a workgroup that holds agents a0 .. a0+na-1 of world w as columns: idx_of(col, s) is the road
// index in slot s, count_of(col) the number of selected slots.  Consecutive threads store consecutive u16.
template <int A_T, typename IdxOf, typename CountOf>
__device__ __forceinline__ void store_selection(const DevSim &d, int w, int a0, int na, IdxOf idx_of, CountOf count_of, int tid,
                                                int nthreads) {
    const size_t wa0 = (size_t)w * A_T + a0;
    unsigned short *dst = d.sel_idx + wa0 * K;
    for (int q = tid; q < na * K; q += nthreads) {
        const int col = q / K, sl = q - col * K;
        dst[q] = (unsigned short)idx_of(col, sl);
    }
    for (int col = tid; col < na; col += nthreads) d.sel_count[wa0 + col] = count_of(col);
}

// ---- reference row order: one wave per 64 agent slots of a world, one lane per agent ----
template <int A_T>
__global__ __launch_bounds__(LW) void k_map_obs(DevSim d) {
    constexpr int C = 32;          // roads per chunk = one candidate word
    constexpr int RING = 16;       // chunks of candidate words a lane may lag behind the scan
    constexpr int BPW = A_T / LW;  // workgroups per world
    static_assert(BPW * LW == A_T, "geometry");
    const int lane = threadIdx.x;
    const int w = (int)blockIdx.x / BPW;
    const int a0 = ((int)blockIdx.x % BPW) * LW;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int n = d.shape[w * 2 + 0];
    if (a0 >= n) return;  // rows of padding agents are written at reset (k_init_padding_rows)
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const int a = a0 + lane;
    const bool live = a < n;
    const size_t i = (size_t)w * A_T + a;

    // one buffer, carved by hand: the ring comes first so that the 1-based heap columns (row g - 1 of the arrays,
    // i.e. base - one row + g rows) never form an address below the buffer
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[RING * LW * 4 + SLOTS * LW * 6 + LW * 4];
    unsigned int *s_ring = reinterpret_cast<unsigned int *>(s_buf);  // word of ring slot c of lane l at [c * LW + l]
    float *s_keys = reinterpret_cast<float *>(s_buf + RING * LW * 4) - LW;                        // [g * LW + lane], g >= 1
    unsigned short *s_idx = reinterpret_cast<unsigned short *>(s_buf + RING * LW * 4 + SLOTS * LW * 4) - LW;  // [g * LW + idx_col]
    float *s_stage = reinterpret_cast<float *>(s_buf + RING * LW * 4 + SLOTS * LW * 6);  // the chunk being scanned: (x, y) of 32 roads

    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;
    const Heap heap{s_keys + lane, s_idx + idx_col(lane)};
    const float2 *rxy = d.road_xy + r0;
    float ex = 0.f, ey = 0.f, iw = 1.f, iz = 0.f;  // pose; (iw, iz) is the INVERSE rotation
    if (live) {
        ex = d.px[i]; ey = d.py[i];
        iw = d.qw[i]; iz = -d.qz[i];
    }
    // A chunk of 32 roads is one coalesced 256-byte load: lane j holds float j of (x0, y0, x1, y1, ...).  It is
    // requested long before it is used, parked in LDS when its turn comes and read back with broadcast ds_read_b64
    // (every lane the same address: conflict-free; v_readlane would route every value through an SGPR, and a VALU
    // instruction that reads an SGPR a VALU instruction has just written waits for it: measured 3x the time).
    // Reads up to 96 roads past the world's last one stay inside the array (the next world's roads or the pad).
    const float *rf = reinterpret_cast<const float *>(rxy);
    auto load_chunk = [&](int first_road) -> float { return rf[first_road * 2 + lane]; };
    int count = 0;
#ifdef GD_STAMPS
    unsigned long long st_scan = 0, st_drain = 0, st_rounds = 0, st_scans = 0, st_init = 0, st_filter = 0, st_wscan = 0, st_wround = 0;
#endif
    STAMP(t_begin);

    if (knn) {
        // roads with index < K go straight into the array (src/knn.hpp:112-120)
        const int nfill = min(R, K);
        {
            float pre = load_chunk(0);
#pragma clang loop unroll(disable)
            for (int base = 0; base < nfill; base += C) {
                s_stage[lane] = pre;
                pre = load_chunk(base + C);
#pragma unroll
                for (int t = 0; t < C; t++) {
                    if (base + t < nfill) {
                        const float2 xy = reinterpret_cast<const float2 *>(s_stage)[t];
                        heap.set(base + t + 1, ego_dist2(ex, ey, iw, iz, xy.x, xy.y), (unsigned int)(base + t));
                    }
                }
            }
        }
        if (R >= K) {
            s_keys[(K + 1) * LW + lane] = -1.f;
            s_keys[(K + 2) * LW + lane] = -1.f;
            heap.make();
            float qk[7];  // ancestor chain of slot K incl. the root: qk[0] is the K-th distance
            unsigned int qi[7];
            float lk;
            unsigned int li;
            heap.load_chain(qk, qi, lk, li);
#ifdef GD_STAMPS
            st_init = __builtin_amdgcn_s_memtime() - t_begin;
#endif
            // The scan compares |p - e|^2 (one fma form) with thr * margin instead of the reference's rotated
            // form: the two differ by rounding (< 1e-6 relative) and by the squared norm of the stored
            // quaternion, det = (1 - 2 z^2)^2 + 4 z^2 w^2; the margin covers both, so the candidates are a
            // superset of the true inserts.  The drain re-tests every candidate exactly.
            const float z2 = iz * iz;
            const float det = (1.f - 2.f * z2) * (1.f - 2.f * z2) + 4.f * z2 * (iw * iw);
            const float margin = 1.00001f / fminf(det, 1.f);
            const int nch = (R - K + C - 1) / C;
            const int trig = max(1, (min(LW, n - a0) + 3) >> 2);  // scan when this many live lanes have run dry
            int head = 0;                     // chunks scanned so far (wave-uniform)
            // chunks head and head + 1 are in flight or landed: even chunks in pre_a, odd ones in pre_b (two copies of
            // the scan code, so that the wait for the chunk being scanned never covers a younger request)
            float pre_a = load_chunk(K), pre_b = load_chunk(K + C);
            unsigned int word = 0, nz = 0;    // this lane's current candidate word; ring slots with unread words
            int cw = 0;                       // chunk of `word`
            bool has = false;                 // (r_cur, xy_cur) is a candidate whose (x, y) has been requested
            int r_cur = 0;
            float2 xy_cur = make_float2(0.f, 0.f);
            auto scan_chunk = [&](float &pre) {
#ifdef GD_STAMPS
                {
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_wscan += __builtin_amdgcn_s_memtime() - t0;
                }
#endif
                const unsigned int slot = (unsigned int)head & (RING - 1);
                const int base = K + head * C;
                s_stage[lane] = pre;
                const float thr = live ? qk[0] * margin : -1.f;
                unsigned int wd = 0;
#pragma unroll
                for (int t = 0; t < C; t++) {
                    const float2 xy = reinterpret_cast<const float2 *>(s_stage)[t];
                    const float dx = xy.x - ex, dy = xy.y - ey;
                    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                    wd |= d2 < thr ? 1u << t : 0u;
                }
                pre = load_chunk(base + 2 * C);
                const int tn = R - base;
                wd &= tn >= C ? 0xffffffffu : (1u << tn) - 1u;
                s_ring[slot * LW + lane] = wd;
                nz |= (wd != 0u ? 1u : 0u) << slot;
                head++;
#ifdef GD_STAMPS
                st_scans++;
#endif
            };
            for (;;) {
                STAMP(t_s0);
                // ---- SCAN: chunks as long as every lane has the ring slot free and enough lanes are idle ----
                while (head < nch) {
                    const unsigned int slot = (unsigned int)head & (RING - 1);
                    const bool room = ((nz >> slot) & 1u) == 0u;
                    const bool idle = live && !has && word == 0u && nz == 0u;
                    if (!__all(room) || __popcll(__ballot(idle)) < trig) break;
                    if (head & 1) scan_chunk(pre_b);
                    else scan_chunk(pre_a);
                }
                STAMP(t_s1);
#ifdef GD_STAMPS
                st_scan += t_s1 - t_s0;
#endif
                if (!__any(has || word != 0u || nz != 0u)) break;  // nothing pending anywhere; then head == nch
                // ---- one DRAIN round ----
#ifdef GD_STAMPS
                {
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_wround += __builtin_amdgcn_s_memtime() - t0;
                }
#endif
                // next candidate: oldest unread ring word if the current one is used up, its lowest bit
                const bool refill = word == 0u && nz != 0u;
                const unsigned int rot = (nz | (nz << RING)) >> ((unsigned int)head & (RING - 1));  // bit j: chunk head - RING + j
                const int c_new = head - RING + (__ffs(rot) - 1);
                const unsigned int slot_new = (unsigned int)c_new & (RING - 1);
                const unsigned int fetched = s_ring[slot_new * LW + lane];
                word = refill ? fetched : word;
                cw = refill ? c_new : cw;
                nz = refill ? nz & ~(1u << slot_new) : nz;
                const bool has_n = word != 0u;
                const int r_nxt = K + cw * C + (__ffs(word) - 1);
                word &= word - 1u;
                const float2 xy_nxt = rxy[has_n ? r_nxt : 0];
                // current candidate: exact key, live test, replay
                const float key = ego_dist2(ex, ey, iw, iz, xy_cur.x, xy_cur.y);
                heap.replace_top(has && key < qk[0], key, (unsigned int)r_cur, qk, qi, lk, li);
                has = has_n;
                r_cur = r_nxt;
                xy_cur = xy_nxt;
#ifdef GD_STAMPS
                st_drain += __builtin_amdgcn_s_memtime() - t_s1;
                st_rounds++;
#endif
            }
            heap.store_last(lk, li);
        }
        STAMP(t_f0);
        if (live) count = heap.radius_filter(min(R, K), d.radius_key_max, s_ring + lane);
#ifdef GD_STAMPS
        st_filter = __builtin_amdgcn_s_memtime() - t_f0;
#endif
    } else {
        // AllEntitiesWithRadiusFiltering: first K in index order within the radius, sim.cpp:261-279
        const float kmax = d.radius_key_max;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int base = 0; base < R; base += C) {
            const float2 *p = rxy + base;
            const int tn = min(C, R - base);
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(8)
            for (int t = 0; t < tn; t++) {
                const float2 xy = p[t];
                const bool pass = live && !(ego_dist2(ex, ey, iw, iz, xy.x, xy.y) > kmax);
                if (pass && count < K) heap.i[(count + 1) * LW] = (unsigned short)(base + t);
                count += pass ? 1 : 0;
            }
            if (__all(count >= K || !live)) break;
        }
    }
    // ---- hand the selection to k_map_rows ----
    int *s_count = reinterpret_cast<int *>(s_ring);  // [LW]
    wave_sync();
    s_count[lane] = live ? min(count, K) : 0;
    wave_sync();
    STAMP(t_w0);
    store_selection<A_T>(d, w, a0, min(LW, n - a0), [&](int c, int sl) -> int { return s_idx[(sl + 1) * LW + idx_col(c)]; },
                         [&](int c) -> int { return s_count[c]; }, lane, LW);
#ifdef GD_STAMPS
    if (lane == 0 && blockIdx.x < 8192) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_stamps[blockIdx.x];
        o[0] = t_end - t_begin; o[1] = st_init; o[2] = st_scan; o[3] = st_drain; o[4] = st_rounds; o[5] = st_scans;
        o[6] = st_wscan; o[7] = st_wround;
    }
#endif
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
    store_selection<A_T>(d, w, 0, n, [&](int c, int sl) -> int { return s_idx[sl * A_T + c]; }, [&](int c) -> int { return s_count[c]; },
                         tid, NW * 64);
}

}  // namespace

void launch_map_obs(const DevSim &d, hipStream_t st) {
    if (d.knn_order == GD_KNN_SET_ORDER) {
        const dim3 grid(d.W);
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs_set<64, 4>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs_set<128, 8>), grid, dim3(512), 0, st, d);
    } else {
        const dim3 grid(d.W * (d.A / LW));
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs<64>), grid, dim3(LW), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs<128>), grid, dim3(LW), 0, st, d);
    }
    const size_t rows = (size_t)d.W * d.A * K;
    const dim3 rgrid((unsigned int)((rows + 255) / 256));
    if (d.A == 64) hipLaunchKernelGGL((k_map_rows<64>), rgrid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_map_rows<128>), rgrid, dim3(256), 0, st, d);
}

}  // namespace gd
