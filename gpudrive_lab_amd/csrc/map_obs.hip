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
#include <type_traits>

#include "engine.hpp"
#include "gd_math.hpp"
#include "map_rows.hpp"
#include "pack_cols.hpp"

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
#ifndef GD_TRIG_NUM
#define GD_TRIG_NUM 4  // eighths of the live lanes that must be idle before the next chunk is scanned
#endif
#ifndef GD_ROWS_PER_THREAD
#define GD_ROWS_PER_THREAD 4
#endif
constexpr int AW = GD_MAP_OBS_AW;  // agents (heap columns) per workgroup = per wave of the reference-order kernel
constexpr int G = 64 / AW;         // lanes per agent: lane = sub * AW + column
constexpr int SLOTS = K + 2;       // stored slots 1..K (the heap), K+1 and K+2 (sentinels); slot g is row g - 1
static_assert(AW == 16 || AW == 32 || AW == 64, "agents per wave");

// Position of column `c` in a row of the u16 index array.  With 64 columns, c and c + 32 share a dword, so the
// 32 lanes of either half of the wave (the LDS services them separately) touch 32 different banks.
__device__ __forceinline__ int idx_col(int c) { return AW == 64 ? (c & 31) * 2 + (c >> 5) : c; }

// One agent's heap: a column of the wave's LDS arrays.  Slot g (1-based; the reference's array index is
// g - 1) is at k[g * AW] / i[g * AW].  With G > 1 lanes per agent, every lane of an agent holds the same state and
// runs the same code; `owner` (sub-lane 0) is the one that stores.
struct Heap {
    float *k;
    unsigned short *i;
    __device__ __forceinline__ float key(int g) const { return k[g * AW]; }
    __device__ __forceinline__ unsigned int index(int g) const { return i[g * AW]; }
    __device__ __forceinline__ void set(int g, float key, unsigned int idx) const {
        k[g * AW] = key;
        i[g * AW] = (unsigned short)idx;
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
    // The parents of a level are dealt round-robin to the G lanes of the agent (lane `sub` takes first + sub,
    // first + sub + G, ...), N at a time per lane; the caller separates levels with wave_sync().
    template <int L, int N>
    __device__ __forceinline__ void make_level(int first, int last, int sub) const {
        int p = first + sub * N;
#pragma clang loop unroll(disable)
        for (; p + N - 1 <= last; p += N * G) {
            int start[N];
#pragma unroll
            for (int j = 0; j < N; j++) start[j] = p + j;
            sift_level<L, N>(start);
        }
        if (N > 1) {
#pragma clang loop unroll(disable)
            for (; p <= last; p++) {  // fewer than N left for this lane
                const int start[1] = {p};
                sift_level<L, 1>(start);
            }
        }
    }
    // len == K (the only length the reference ever heapifies, src/knn.hpp:128)
    __device__ __forceinline__ void make(int sub) const {
        constexpr int N = G == 1 ? 4 : 1;
        make_level<6, N>(64, K / 2, sub); wave_sync();
        make_level<5, N>(32, 63, sub); wave_sync();
        make_level<4, N>(16, 31, sub); wave_sync();
        make_level<3, N>(8, 15, sub); wave_sync();
        make_level<2, G == 1 ? 4 : 1>(4, 7, sub); wave_sync();
        make_level<1, G == 1 ? 2 : 1>(2, 3, sub); wave_sync();
        make_level<0, 1>(1, 1, sub); wave_sync();
    }
    // pop_heap + replace last + push_heap (src/knn.hpp:138-151) as straight-line code.
    //
    // pop_heap moves the root to slot K (overwritten by the new element right after, so that store is
    // dropped), sifts the hole down from the root of the (K-1)-element heap -- always 6 levels, and a 7th
    // iff the level-6 node g has children (2g + 1 <= K - 1); otherwise sentinels are read, whose key -1
    // lets x rest at level 6 at the latest -- then pushes the old last element x up from the leaf hole.
    // The values x meets on its way up are exactly the children just moved, which are still in
    // registers, so its climb needs no LDS read.  push_heap then lifts the new element y from slot K
    // along the fixed ancestor chain 100, 50, 25, 12, 6, 3, 1.
    //
    // State kept in registers for the whole replay (struct Top; load() / store() move it from / to LDS):
    //   * slots 1..7 (tree levels 0..2), keys and indices: the first two decisions of every pop and the top of
    //     the push chain need no LDS at all, and those slots are never written to LDS during the replay
    //     (LDS stores are the most expensive instructions of the round, and the next round's first read
    //     queues behind them);
    //   * the deeper chain slots 12, 25, 50, 100 (qk / qi; they are also kept current in LDS, where the pop
    //     reads them as ordinary children) and slot K (lk / li; its LDS copy holds the sentinel -1).
    // There are no data-dependent loops and no serial predicate chains: one basic block, executed by the lanes that
    // have an insert this round (a VALU instruction costs a lone wave the same 4 cycles whatever its EXEC mask, so
    // what counts is the instruction count: 16 selects per round to keep the idle lanes' state were more expensive
    // than the EXEC region).  With several lanes per agent only `owner` stores.
    struct Top {
        float tk[8];         // tk[g], g = 1..7; tk[1] is the K-th distance
        unsigned int ti[8];
        float qk[4];         // slots 12, 25, 50, 100
        unsigned int qi[4];
        float lk;            // slot K
        unsigned int li;
    };
    static constexpr int QD[4] = {12, 25, 50, 100};
    __device__ __forceinline__ void load(Top &t) const {
#pragma unroll
        for (int g = 1; g < 8; g++) { t.tk[g] = key(g); t.ti[g] = index(g); }
#pragma unroll
        for (int u = 0; u < 4; u++) { t.qk[u] = key(QD[u]); t.qi[u] = index(QD[u]); }
        t.lk = key(K);
        t.li = index(K);
        k[K * AW] = -1.f;  // slot K lives in registers during the replay; its LDS copy is a third sentinel
    }
    __device__ __forceinline__ void store(const Top &t) const {
#pragma unroll
        for (int g = 1; g < 8; g++) set(g, t.tk[g], t.ti[g]);
        set(K, t.lk, t.li);
    }
    __device__ __forceinline__ void replace_top(bool owner, float yk, unsigned int yi, Top &t) const {
        int g[8];
        float ck[7];
        unsigned int ci[7];
        g[0] = 1;
        // levels 0 and 1 out of registers
        const bool r0 = !(t.tk[3] < t.tk[2]);
        ck[0] = r0 ? t.tk[3] : t.tk[2];
        ci[0] = r0 ? t.ti[3] : t.ti[2];
        g[1] = 2 + (r0 ? 1 : 0);
        const float hl = r0 ? t.tk[6] : t.tk[4], hr = r0 ? t.tk[7] : t.tk[5];
        const unsigned int il = r0 ? t.ti[6] : t.ti[4], ir = r0 ? t.ti[7] : t.ti[5];
        const bool r1 = !(hr < hl);
        ck[1] = r1 ? hr : hl;
        ci[1] = r1 ? ir : il;
        g[2] = 2 * g[1] + (r1 ? 1 : 0);
        // levels 2..5: decisions use keys only, two levels per LDS round trip (the hole's two children AND its
        // four grandchildren are fetched together); the index of a chosen child is requested as soon as it is known
#pragma unroll
        for (int l = 2; l < 6; l += 2) {
            const int cb = 2 * g[l];  // children cb, cb + 1; grandchildren 2 cb .. 2 cb + 3
            const float kl = key(cb), kr = key(cb + 1);
            const float g0 = key(2 * cb), g1 = key(2 * cb + 1), g2 = key(2 * cb + 2), g3 = key(2 * cb + 3);
            const bool right = !(kr < kl);
            ck[l] = right ? kr : kl;
            g[l + 1] = cb + (right ? 1 : 0);
            ci[l] = index(g[l + 1]);
            const float gl = right ? g2 : g0, gr = right ? g3 : g1;
            const bool right2 = !(gr < gl);
            ck[l + 1] = right2 ? gr : gl;
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
        for (int l = 0; l < 7; l++) c[l] = ck[l] < t.lk;
        c[7] = true;
        float nk[8];
        unsigned int ni[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            // the ends of the path have only one neighbour: a select on the predicate already computed
            if (l == 0) nk[l] = c[0] ? t.lk : ck[0];
            else if (l == 7) nk[l] = c[6] ? ck[6] : t.lk;
            else nk[l] = __builtin_amdgcn_fmed3f(ck[l - 1], ck[l], t.lk);
            ni[l] = c[l] ? t.li : (l < 7 ? ci[l] : 0u);
            if (l > 0) ni[l] = c[l - 1] ? ci[l - 1] : ni[l];
        }
        // slots 1..7 after the pop
        float pk[8];
        unsigned int pi[8];
        pk[1] = nk[0]; pi[1] = ni[0];
        pk[2] = r0 ? t.tk[2] : nk[1]; pi[2] = r0 ? t.ti[2] : ni[1];
        pk[3] = r0 ? nk[1] : t.tk[3]; pi[3] = r0 ? ni[1] : t.ti[3];
#pragma unroll
        for (int j = 4; j < 8; j++) {
            const bool hit = (r0 == ((j & 2) != 0)) && (r1 == ((j & 1) != 0));  // g[2] == j
            pk[j] = hit ? nk[2] : t.tk[j];
            pi[j] = hit ? ni[2] : t.ti[j];
        }
        // the deeper chain slots after the pop (patched where the path went through them)
        float dk[4];
        unsigned int di[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool rewritten = g[u + 3] == QD[u];  // g[l] lives on level l, like the chain slot of that level
            dk[u] = rewritten ? nk[u + 3] : t.qk[u];
            di[u] = rewritten ? ni[u + 3] : t.qi[u];
        }
        // y climbs from slot K along the chain 100, 50, 25, 12, 6, 3, 1 (levels 6..0): with the chain as
        // q[0..6] = slots 1, 3, 6, 12, 25, 50, 100, non-increasing towards the leaf, p[u] = q[u] < yk is monotone as
        // well: chain position u (7 = slot K) receives q[u-1] if p[u-1], y if p[u] && !p[u-1], and keeps its value
        // otherwise (median again).
        const float q[7] = {pk[1], pk[3], pk[6], dk[0], dk[1], dk[2], dk[3]};
        const unsigned int qx[7] = {pi[1], pi[3], pi[6], di[0], di[1], di[2], di[3]};
        bool p[8];
#pragma unroll
        for (int u = 0; u < 7; u++) p[u] = q[u] < yk;
        p[7] = true;
        float ok[8];
        unsigned int oi[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u == 0) ok[u] = p[0] ? yk : q[0];
            else if (u == 7) ok[u] = p[6] ? q[6] : yk;
            else ok[u] = __builtin_amdgcn_fmed3f(q[u - 1], q[u], yk);
            oi[u] = p[u] ? yi : (u < 7 ? qx[u] : 0u);
            if (u > 0) oi[u] = p[u - 1] ? qx[u - 1] : oi[u];
        }
        pk[1] = ok[0]; pi[1] = oi[0];
        pk[3] = ok[1]; pi[3] = oi[1];
        pk[6] = ok[2]; pi[6] = oi[2];
#pragma unroll
        for (int j = 1; j < 8; j++) {
            t.tk[j] = pk[j];
            t.ti[j] = pi[j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            t.qk[u] = ok[u + 3];
            t.qi[u] = oi[u + 3];
        }
        t.lk = ok[7];
        t.li = oi[7];
        if (owner) {
#pragma unroll
            for (int l = 3; l < 8; l++) set(g[l], nk[l], ni[l]);
#pragma unroll
            for (int u = 0; u < 4; u++) set(QD[u], ok[u + 3], oi[u + 3]);
        }
    }
    // radiusFilter, src/knn.hpp:83-97: swap-remove in heap-array order; returns newBeyond.  `kmax` is the largest
    // fp32 whose correctly rounded square root is <= radius: length() <= radius without the root.
    // The reference walks s up from 0 and, at every out-of-radius slot, pulls elements from the end of the array
    // until an in-radius one lands there.  With m in-radius elements the outcome is: the holes (out-of-radius
    // slots below m, ascending) receive the in-radius elements of [m, len) in DESCENDING slot order, everything
    // else below m stays.  So: one pass for the in-radius flags (`flags`: 7 words of this lane in LDS, word w at
    // flags[w * AW]), then one move per hole.
    __device__ __forceinline__ int radius_filter(int len, float kmax, unsigned int *flags, bool owner) const {
        int m = 0;
#pragma unroll
        for (int w = 0; w < (K + 31) / 32; w++) {
            unsigned int f = 0;
#pragma unroll
            for (int b = 0; b < 32; b++) {
                const int s = w * 32 + b;
                if (s < K) f |= (s < len && key(s + 1) <= kmax) ? 1u << b : 0u;
            }
            if (owner) flags[w * AW] = f;
            m += __popc(f);
        }
        wave_sync();
        if (m < len) {
            auto below = [](int limit, int w) -> unsigned int {  // bits of word w whose slot is < limit
                const int n = limit - w * 32;
                return n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : (1u << n) - 1u);
            };
            int hw = 0, dw = (len - 1) >> 5;
            unsigned int holes = ~flags[0] & below(m, 0);
            unsigned int donors = flags[dw * AW] & ~below(m, dw);
            for (;;) {
                while (holes == 0u && (hw + 1) * 32 < m) {
                    hw++;
                    holes = ~flags[hw * AW] & below(m, hw);
                }
                if (holes == 0u) break;
                while (donors == 0u) {  // as many donors as holes: never runs off the array
                    dw--;
                    donors = flags[dw * AW] & ~below(m, dw);
                }
                const int hole = hw * 32 + __ffs(holes) - 1;
                holes &= holes - 1u;
                const int db = 31 - __clz(donors);
                donors &= ~(1u << db);
                if (owner) move(hole + 1, dw * 32 + db + 1);
            }
        }
        return m;
    }
};

// Per-agent cursor over the ring of candidate words, and one round of the drain.
#ifndef GD_RING
#define GD_RING 16
#endif
constexpr int RING = GD_RING;  // chunks of candidate words an agent may lag behind the scan
constexpr int C = 32;     // roads per chunk = one candidate word
struct Drain {
    unsigned int word = 0;  // unread candidate bits of chunk `cw`
    unsigned int nz = 0;    // ring slots that hold an unread, non-empty word of this agent
    int cw = 0;
    bool has = false;       // (r_cur, xy_cur) is a candidate whose (x, y) has been requested
    int done = 0;           // candidates replayed or rejected so far
    int r_cur = 0;
    float2 xy_cur = {0.f, 0.f};
    __device__ __forceinline__ bool pending() const { return has || word != 0u || nz != 0u; }
};
// One candidate per agent: the candidate fetched in the previous round is keyed exactly as the reference does
// (gd_math.hpp ego_dist2), re-tested against the live K-th distance and replayed; meanwhile the next one is popped
// (oldest unread ring word if the current one is used up, its lowest bit) and its (x, y) requested.  `ring` is
// this agent's column of the ring (slot c at ring[c * AW]), `head` the number of chunks scanned so far.
// `cpr` / `cpt`: this agent's checkpoint rows (engine.hpp cp_road / cp_T; null when the rank path is off): after every
// 64th candidate the K-th key in force is recorded together with the first road it holds for (rounded up to a whole
// 32-road chunk) -- what map_obs_rank.hip bounds the agent's next selection with.
__device__ __forceinline__ void drain_round(const Heap &heap, Heap::Top &top, Drain &s, const unsigned int *ring, const float2 *rxy,
                                            int head, float ex, float ey, float iw, float iz, bool owner,
                                            unsigned short *cpr, float *cpt) {
    const bool refill = s.word == 0u && s.nz != 0u;
    const unsigned int rot = (s.nz | (s.nz << RING)) >> ((unsigned int)head & (RING - 1));  // bit j: chunk head - RING + j
    const int c_new = head - RING + (__ffs(rot) - 1);
    const unsigned int slot_new = (unsigned int)c_new & (RING - 1);
    const unsigned int fetched = ring[slot_new * AW];
    const unsigned int word = refill ? fetched : s.word;
    const int cw = refill ? c_new : s.cw;
    const bool has_n = word != 0u;
    const int r_nxt = K + cw * C + (__ffs(word) - 1);
    const float2 xy_nxt = rxy[has_n ? r_nxt : 0];
    const float key = ego_dist2(ex, ey, iw, iz, s.xy_cur.x, s.xy_cur.y);
    if (s.has && key < top.tk[1]) heap.replace_top(owner, key, (unsigned int)s.r_cur, top);  // lanes without an insert sit the block out
    if (s.has) {
        s.done++;
        // (every 64th candidate: the 40 checkpoint slots then cover the longest list the rank path takes, 2560 candidates.
        // At every 32nd they ended after 1,248; an agent with a longer history -- unreduced Waymo polylines -- came back
        // from this fallback with nothing to bound its later roads, overflowed the rank path's buffers at once and was
        // sent here again, every other selection)
        if (cpr != nullptr && (s.done & 63) == 0 && (s.done >> 6) < GD_RANK_NCP && owner) {
            cpr[s.done >> 6] = (unsigned short)min(65535, (s.r_cur + 1 + 31) & ~31);
            cpt[s.done >> 6] = top.tk[1];
        }
    }
    s.nz = refill ? s.nz & ~(1u << slot_new) : s.nz;
    s.word = word & (word - 1u);
    s.cw = cw;
    s.has = has_n;
    s.r_cur = r_nxt;
    s.xy_cur = xy_nxt;
}

// ---- row write-out (both selection kernels end by handing their selection to this one) ----
// One thread per (world, agent, slot) row of agent_roadmap_tensor; a wave writes 64 consecutive 36-byte rows.  The row is
// ReferenceFrame::observationOf (src/utils.hpp:36-49) of the selected road, or the padding row.  Splitting it from the
// selection lets the gather / atan2 / store work run at full occupancy instead of behind the LDS-bound selection waves.
// (road_row: map_rows.hpp)

// Launch order of the next k_map_obs (see "launch order" further down): counting sort of its workgroups by the cycles they
// took, costliest first, by ONE workgroup of NT threads; `lds` needs 513 words.
template <int NT>
__device__ __forceinline__ void order_waves(const DevSim &d, int count, unsigned int *lds) {
    unsigned int *s_cnt = lds, *s_off = lds + 256, *s_max = lds + 512;
    const int tid = threadIdx.x;
    for (int b = tid; b < 256; b += NT) s_cnt[b] = 0u;
    if (tid == 0) *s_max = 0u;
    __syncthreads();
    unsigned int m = 0u;
    for (int i = tid; i < count; i += NT) m = max(m, d.wave_cost[i]);
    atomicMax(s_max, m);
    __syncthreads();
    const unsigned long long mx = max(*s_max, 1u);
    auto bucket = [&](unsigned int c) -> int { return 255 - (int)min(255ull, (unsigned long long)c * 255ull / mx); };  // costliest first
    for (int i = tid; i < count; i += NT) atomicAdd(&s_cnt[bucket(d.wave_cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned int run = 0u;
        for (int b = 0; b < 256; b++) { s_off[b] = run; run += s_cnt[b]; }
    }
    __syncthreads();
    for (int i = tid; i < count; i += NT) d.wave_order[atomicAdd(&s_off[bucket(d.wave_cost[i])], 1u)] = i;
    __syncthreads();
}

// A workgroup writes the K rows of ROWS_AB consecutive agent slots: one contiguous block of the tensor (ROWS_AB x 7200
// bytes).  Entry q of an agent's selection is gathered by thread q and its row is put where the selection kernel says it
// belongs (sel_slot; engine.hpp): the rank path hands its roads over in ASCENDING road index -- neighbouring threads then
// gather neighbouring 32-byte records (the nearest roads come in runs along their polylines) instead of the heap's order
// (136 -> us for the same 472 MB at 1024 x 64; set order, ascending by construction, always ran at 82) -- and the
// order only has to exist where the rows are stored.
#ifndef GD_SET_CAP
#define GD_SET_CAP 1024
#endif
// PACK (gd_attach_packed): the same rows also -- or only: DevSim::pack_only -- in the packed observation's 13 normalised columns
// (pack_cols.hpp), three agents per workgroup instead of five (the 13-column block fills the same LDS).
template <bool PACK> struct RowsGeo { static constexpr int AB = PACK ? 3 : 5; };
template <int A_T, bool PACK = false>
__global__ __launch_bounds__(256) void k_map_rows(DevSim d) {
    constexpr int ROWS_AB = RowsGeo<PACK>::AB;
    constexpr int U = GD_ROWS_PER_THREAD;  // entries per thread (256 apart): independent load chains in flight
    constexpr int RB = ROWS_AB * K;        // rows per workgroup
    static_assert(256 * U >= RB && (RB * 9) % 4 == 0 && (K * 9) % 4 == 0 && (K * 13) % 4 == 0, "every entry has a thread; whole 16-byte pieces per agent");
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    __shared__ __attribute__((aligned(16))) float s_rows[RB * (PACK ? 13 : 9)];
    static_assert(RB * 9 >= 513, "order_waves borrows the row buffer");
    static_assert(!PACK || 3 * K * 13 * 4 <= 5 * K * 9 * 4, "the packed block fits the LDS of the raw one");
    if (blockIdx.x == 0 && d.knn_order != GD_KNN_SET_ORDER)
        order_waves<256>(d, d.W * (A_T / AW), reinterpret_cast<unsigned int *>(s_rows));
    const size_t agents = (size_t)d.W * A_T;
    // Which agents a workgroup takes: workgroup b runs on XCD b % 8, and the XCDs take eighths of the tensor, so that a world's
    // road records (128 KB on the bench scene) are fetched into one L2 instead of all eight
    const unsigned int per_xcd = gridDim.x >> 3;  // (the grid is a multiple of 8 workgroups)
    const size_t a0 = (size_t)((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * ROWS_AB;
    if (a0 >= agents) return;
    // A thread takes four CONSECUTIVE entries of one agent (K is a multiple of four): the agent's header (pose, count, first
    // road, permuted: written by the selection kernel) is read once per thread, the four road indices come as one 8-byte
    // load and their four rows as one 4-byte load -- four loads where a thread per strided row issued sixteen; the roads'
    // 32-byte records follow in a second round trip.
    static_assert(U == 4 && K % 4 == 0, "four consecutive entries of one agent per thread");
    bool in[U];
    int r[U], dst[U];
    __shared__ unsigned char s_on[ROWS_AB];
    const int t4 = threadIdx.x * 4;
    const bool act = t4 < RB;
    const int al = min(t4 / K, ROWS_AB - 1);
    const int qf = act ? t4 - al * K : 0;
    const size_t wa = min(a0 + al, agents - 1);
    const float4 pose = d.sel_hdr[wa * 2];
    const float4 meta = d.sel_hdr[wa * 2 + 1];
    const int cnt = __float_as_int(meta.x);
    const bool permuted = __float_as_int(meta.z) != 0;
    const size_t e = wa * K + qf;  // a multiple of four: both loads are aligned
    const uint2 idx4 = *reinterpret_cast<const uint2 *>(d.sel_idx + e);
    const unsigned int slot4 = *reinterpret_cast<const unsigned int *>(d.sel_slot + e);
    // rows already written for exactly this pose (engine.hpp pose_stamp: it dies with the world's roads) are left in place: an
    // agent that did not move -- a parked car, a finished agent at the padding position -- selects the same roads and sees them
    // in the same place
    const uint4 st = d.pose_stamp[wa];
    const bool same = d.pose_skip != 0 && st.x != 0xffffffffu && st.x == __float_as_uint(pose.x) && st.y == __float_as_uint(pose.y) &&
                      st.z == __float_as_uint(pose.z) && st.w == __float_as_uint(pose.w);
    const bool on = act && a0 + al < agents && cnt >= 0 && !same;  // rows of padding agents are written at reset (k_init_padding_rows)
    if (act && qf == 0) {
        s_on[al] = on ? 1 : 0;
        if (a0 + al < agents && cnt >= 0 && same) atomicAdd(d.stat_skipped + (blockIdx.x & (GD_SKIP_SLOTS - 1)), 1ull);
    }
    // nothing to write for any agent of this workgroup (padding slots, agents that did not move): no gathers, no rows
    if (__syncthreads_or(on ? 1 : 0) == 0) return;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int q = qf + u;
        const int idx = (int)(((u & 2) ? idx4.y : idx4.x) >> ((u & 1) * 16)) & 0xffff;
        const int slot = (int)(slot4 >> (u * 8)) & 0xff;
        in[u] = q < cnt;
        r[u] = in[u] ? __float_as_int(meta.y) + idx : 0;
        dst[u] = al * K + ((in[u] && permuted) ? min(slot, K - 1) : q);  // (the permutation costs 2 us of the kernel's 100)
    }
    float4 q0[U], q1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        q0[u] = d.road_rec[(size_t)r[u] * 2];
        q1[u] = d.road_rec[(size_t)r[u] * 2 + 1];
    }
    // The block's rows are assembled in LDS (row stride 9 floats: conflict-free) and leave as whole 16-byte pieces in
    // row-major order, one piece per thread and pass; an agent's rows are 450 pieces, so the pieces of agents that must not
    // be written (padding agents, beyond the tensor) are skipped whole.  Streaming (nt) stores: the rows are written once
    // and not read again by the step, and must not push the road and agent arrays out of L2 / Infinity Cache.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;
    if (!PACK) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (act) road_row(s_rows + dst[u] * 9, in[u], knn, pose.x, pose.y, pose.z, pose.w, q0[u], q1[u]);
        }
        __syncthreads();
        float *out = d.agent_map + a0 * (size_t)(K * 9);  // a0 * 7200 bytes: 16-byte aligned
        constexpr int PPA = K * 9 / 4;  // pieces per agent
        for (int q = threadIdx.x; q < RB * 9 / 4; q += 256) {
            if (s_on[q / PPA] != 0)
                __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(s_rows + q * 4), reinterpret_cast<f4 *>(out + (size_t)q * 4));
        }
    } else {
        float raw[U][9];
#pragma unroll
        for (int u = 0; u < U; u++) road_row(raw[u], in[u], knn, pose.x, pose.y, pose.z, pose.w, q0[u], q1[u]);
        if (!d.pack_only) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!act) continue;
#pragma unroll
                for (int c = 0; c < 9; c++) s_rows[dst[u] * 9 + c] = raw[u][c];
            }
            __syncthreads();
            float *out = d.agent_map + a0 * (size_t)(K * 9);
            constexpr int PPA = K * 9 / 4;
            for (int q = threadIdx.x; q < RB * 9 / 4; q += 256) {
                if (s_on[q / PPA] != 0)
                    __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(s_rows + q * 4), reinterpret_cast<f4 *>(out + (size_t)q * 4));
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (act) pack_road_row(raw[u], s_rows + dst[u] * 13);
        }
        __syncthreads();
        constexpr int PACK_ROAD0 = 6 + (A_T - 1) * 6, PACK_D = PACK_ROAD0 + K * 13, PPA13 = K * 13 / 4;
        static_assert(PACK_ROAD0 % 4 == 0 && PACK_D % 4 == 0, "whole 16-byte pieces");
        for (int q = threadIdx.x; q < RB * 13 / 4; q += 256) {
            const int ag = q / PPA13;
            if (s_on[ag] != 0)
                __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(s_rows + q * 4),
                                            reinterpret_cast<f4 *>(d.pack + (a0 + ag) * (size_t)PACK_D + PACK_ROAD0 + (size_t)(q - ag * PPA13) * 4));
        }
    }
    if (on && qf == 0) d.pose_stamp[wa] = make_uint4(__float_as_uint(pose.x), __float_as_uint(pose.y), __float_as_uint(pose.z), __float_as_uint(pose.w));
}

// Selection hand-over from a workgroup that holds agents a0 .. a0+na-1 of world w as columns: idx_of(col, s) is the road
// index in slot s.  Consecutive threads store consecutive u16.  The agents' headers are written by their own lanes
// (write_header).
template <int A_T, typename IdxOf>
__device__ __forceinline__ void store_selection(const DevSim &d, int w, int a0, int na, IdxOf idx_of, int tid, int nthreads) {
    const size_t wa0 = (size_t)w * A_T + a0;
    unsigned short *dst = d.sel_idx + wa0 * K;
    for (int q = tid; q < na * K; q += nthreads) {
        const int col = q / K, sl = q - col * K;
        dst[q] = (unsigned short)idx_of(col, sl);
    }
}
// what k_map_rows needs to know about agent i besides its slots: pose (qz, not its inverse), row count, first road of the world
__device__ __forceinline__ void write_header(const DevSim &d, size_t i, float ex, float ey, float qw, float qz, int count, int road0) {
    d.sel_hdr[i * 2] = make_float4(ex, ey, qw, qz);
    d.sel_hdr[i * 2 + 1] = make_float4(__int_as_float(count), __int_as_float(road0), 0.f, 0.f);
}

// ---- reference row order: one wave per AW agent slots of a world, G = 64 / AW lanes per agent ----
template <int A_T>
__global__ __launch_bounds__(64) void k_map_obs(DevSim d) {
    constexpr int PL = C / G;      // roads of a chunk per lane
    constexpr int BPW = A_T / AW;  // workgroups per world
    static_assert(BPW * AW == A_T, "geometry");
    const int lane = threadIdx.x;
    const int col = lane % AW, sub = lane / AW;
    const bool owner = sub == 0;
    const int slot = d.wave_order[blockIdx.x];  // longest workgroups first (order_waves)
    const int w = slot / BPW;
    const int a0 = (slot % BPW) * AW;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    // after the rank replay (map_obs_rank.hip) only the groups that could not take it are selected here
    const bool rank_path = d.rk_on != 0 && d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;
    if (rank_path) {
        if (d.rk_fallback[slot] == 0) {
            // selected by the rank replay: nothing to do here, and nothing to schedule first next time (order_waves would
            // otherwise keep dealing this group the cycles of its last fallback)
            if (lane == 0) d.wave_cost[slot] = 0u;
            return;
        }
    }
    const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
    const int n = d.shape[w * 2 + 0];
    if (a0 >= n) return;  // rows of padding agents are written at reset (k_init_padding_rows)
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const int a = a0 + col;
    const bool live = a < n;
    const size_t i = (size_t)w * A_T + a;

    // one buffer, carved by hand: the ring comes first so that the 1-based heap columns (row g - 1 of the arrays,
    // i.e. base - one row + g rows) never form an address below the buffer
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[RING * AW * 4 + SLOTS * AW * 6];
    unsigned int *s_ring = reinterpret_cast<unsigned int *>(s_buf);  // word of ring slot c of column l at [c * AW + l]
    float *s_keys = reinterpret_cast<float *>(s_buf + RING * AW * 4) - AW;                                    // [g * AW + col], g >= 1
    unsigned short *s_idx = reinterpret_cast<unsigned short *>(s_buf + RING * AW * 4 + SLOTS * AW * 4) - AW;  // [g * AW + idx_col]

    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;
    const Heap heap{s_keys + col, s_idx + idx_col(col)};
    const float2 *rxy = d.road_xy + r0;
    float ex = 0.f, ey = 0.f, iw = 1.f, iz = 0.f;  // pose; (iw, iz) is the INVERSE rotation
    if (live) {
        ex = d.px[i]; ey = d.py[i];
        iw = d.qw[i]; iz = -d.qz[i];
    }
    // A chunk of 32 roads is one coalesced 256-byte load: lane j holds float j of (x0, y0, x1, y1, ...).  It is
    // requested long before it is used; a lane picks the (x, y) of its PL roads out of the register with
    // ds_bpermute (the LDS crossbar, no LDS memory).  Lane `sub` of an agent takes roads sub * PL .. sub * PL + PL - 1 of a chunk.
    // Reads up to 256 roads past the world's last one stay inside the array (the next world's roads or the pad).
    const float *rf = reinterpret_cast<const float *>(rxy);
    auto load_chunk = [&](int first_road) -> float { return rf[first_road * 2 + lane]; };
    auto road_of = [&](int chunk_reg, int k) -> float2 {  // road t = k * G + sub of the chunk held in `chunk_reg`
        const int t = k * G + sub;
        return make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(8 * t, chunk_reg)),
                           __int_as_float(__builtin_amdgcn_ds_bpermute(8 * t + 4, chunk_reg)));
    };
    auto agent_or = [&](unsigned int part) -> unsigned int {  // OR over the G lanes of an agent
#pragma unroll
        for (int st = AW; st < 64; st <<= 1) part |= (unsigned int)__shfl_xor((int)part, st);
        return part;
    };
    int count = 0;
    // the rank path's checkpoint rows of this agent (recorded here so that the group is back on that path next step)
    unsigned short *cpr = rank_path && live ? d.cp_road + i * GD_RANK_NCP : nullptr;
    float *cpt = rank_path && live ? d.cp_T + i * GD_RANK_NCP : nullptr;
    int cp_count = 0;
#ifdef GD_STAMPS
    unsigned long long st_scan = 0, st_drain = 0, st_rounds = 0, st_scans = 0, st_init = 0, st_filter = 0, st_wscan = 0, st_wround = 0;
    (void)st_wround;
#endif
    STAMP(t_begin);

    if (knn) {
        // roads with index < K go straight into the array (src/knn.hpp:112-120)
        const int nfill = min(R, K);
        {
            float pre = load_chunk(0);
#pragma clang loop unroll(disable)
            for (int base = 0; base < nfill; base += C) {
                const int cur = __float_as_int(pre);
                pre = load_chunk(base + C);
#pragma unroll
                for (int k = 0; k < PL; k++) {
                    const int t = base + k * G + sub;
                    const float2 xy = road_of(cur, k);
                    if (t < nfill) heap.set(t + 1, ego_dist2(ex, ey, iw, iz, xy.x, xy.y), (unsigned int)t);
                }
            }
        }
        if (R >= K) {
            if (owner) {
                s_keys[(K + 1) * AW + col] = -1.f;
                s_keys[(K + 2) * AW + col] = -1.f;
            }
            wave_sync();
            heap.make(sub);
            Heap::Top top;  // slots 1..7, the deeper chain slots and slot K; top.tk[1] is the K-th distance
            heap.load(top);
            wave_sync();  // every lane of an agent has read slot K before its sentinel value is visible
            if (cpr != nullptr && owner && live) {  // checkpoint 0: the heap of the first K roads
                cpr[0] = (unsigned short)((K / 32) * 32);
                cpt[0] = top.tk[1];
            }
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
            const int trig = max(1, (min(AW, n - a0) * GD_TRIG_NUM + 7) >> 3);  // scan when this many live agents have run dry
            int head = 0;                     // chunks scanned so far (wave-uniform)
            // chunks head and head + 1 are in flight or landed: even chunks in pre_a, odd ones in pre_b
            float pre_a = load_chunk(K), pre_b = load_chunk(K + C);
            Drain dr;
            const unsigned long long agents = __ballot(live && owner);  // one bit per live agent of this wave
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
                const int cur = __float_as_int(pre);
                const float thr = live ? top.tk[1] * margin : -1.f;
                // every (x, y) of the lane's roads is requested before the first one is used (one LDS-crossbar latency instead
                // of one per pair); the in-range bit of a road is the sign of d2 - thr (no compare -> mask -> select chain) and
                // is shifted in with one v_alignbit.  Lane `sub` takes roads sub * PL .. sub * PL + PL - 1 of the chunk.
                float2 xy[PL];
#pragma unroll
                for (int k = 0; k < PL; k++) {
                    const int t = sub * PL + k;
                    xy[k] = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute(8 * t, cur)),
                                        __int_as_float(__builtin_amdgcn_ds_bpermute(8 * t + 4, cur)));
                }
                __builtin_amdgcn_sched_barrier(0);
                unsigned int q = 0;
#pragma unroll
                for (int k = PL - 1; k >= 0; k--) {
                    const float dx = xy[k].x - ex, dy = xy[k].y - ey;
                    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                    q = __builtin_amdgcn_alignbit(q, __float_as_uint(d2 - thr), 31);  // (q << 1) | sign(d2 - thr): d2 < thr, both finite
                }
                const unsigned int part = q << (sub * PL);
                pre = load_chunk(base + 2 * C);
                unsigned int wd = agent_or(part);
                const int tn = R - base;
                wd &= tn >= C ? 0xffffffffu : (1u << tn) - 1u;
                if (owner) s_ring[slot * AW + col] = wd;
                dr.nz |= (wd != 0u ? 1u : 0u) << slot;
                head++;
#ifdef GD_STAMPS
                st_scans++;
#endif
            };
            unsigned long long pend = 0;  // agents with a candidate, unread bits or unread ring words
            for (;;) {
                STAMP(t_s0);
                // ---- SCAN: chunks as long as every agent has the ring slot free and enough agents are idle ----
                while (head < nch) {
                    const unsigned int slot = (unsigned int)head & (RING - 1);
                    if (__ballot((dr.nz >> slot) & 1u) != 0ull || __popcll(agents & ~pend) < trig) break;
                    if (head & 1) scan_chunk(pre_b);
                    else scan_chunk(pre_a);
                    pend = __ballot(dr.pending()) & agents;
                }
                STAMP(t_s1);
#ifdef GD_STAMPS
                st_scan += t_s1 - t_s0;
#endif
                if (pend == 0ull) break;  // nothing pending anywhere; then head == nch
                // ---- one DRAIN round ----
                drain_round(heap, top, dr, s_ring + col, rxy, head, ex, ey, iw, iz, owner, cpr, cpt);
                pend = __ballot(dr.pending()) & agents;
#ifdef GD_STAMPS
                st_drain += __builtin_amdgcn_s_memtime() - t_s1;
                st_rounds++;
#endif
            }
            if (owner) heap.store(top);
            cp_count = min(GD_RANK_NCP, 1 + (dr.done >> 6));
        }
        wave_sync();
        STAMP(t_f0);
        // radiusFilter (src/knn.hpp:156).  When the K-th distance itself is within the radius, every element of the heap
        // is, and the filter moves nothing: the common case wherever roads are dense.
        if (R >= K && __all(!live || heap.key(1) <= d.radius_key_max)) count = K;
        else count = heap.radius_filter(min(R, K), d.radius_key_max, s_ring + col, owner);
#ifdef GD_STAMPS
        st_filter = __builtin_amdgcn_s_memtime() - t_f0;
#endif
    } else {
        // AllEntitiesWithRadiusFiltering: first K in index order within the radius, sim.cpp:261-279.  Lane `sub` of an
        // agent tests road k * G + sub of every chunk; the G partial words are OR-ed and every lane derives the
        // positions of its own hits from the agent's word.
        const float kmax = d.radius_key_max;
        float pre = load_chunk(0);
#pragma clang loop unroll(disable)
        for (int base = 0; base < R; base += C) {
            const int cur = __float_as_int(pre);
            pre = load_chunk(base + C);
            unsigned int part = 0;
#pragma unroll
            for (int k = 0; k < PL; k++) {
                const int t = k * G + sub;
                const float2 xy = road_of(cur, k);
                const bool pass = live && base + t < R && !(ego_dist2(ex, ey, iw, iz, xy.x, xy.y) > kmax);
                part |= (pass ? 1u : 0u) << t;
            }
            const unsigned int wd = agent_or(part);
            unsigned int mine = part;
            while (mine) {
                const int t = __ffs(mine) - 1;
                mine &= mine - 1u;
                const int pos = count + __popc(wd & ((1u << t) - 1u));
                if (pos < K) heap.i[(pos + 1) * AW] = (unsigned short)(base + t);
            }
            count += __popc(wd);
            if (__all(count >= K || !live)) break;
        }
    }
    // ---- hand the selection to k_map_rows ----
    wave_sync();
    if (owner && live) write_header(d, i, ex, ey, iw, -iz, min(count, K), r0);
    if (rank_path) {
        if (owner && live) {
            d.cp_hdr[i] = make_float4(ex, ey, __int_as_float(cp_count), 0.f);
            d.rk_n[i] = -1;  // selected here (gd_debug_road_path)
            if (d.steps[i] == (uint32_t)GD_EPISODE_LEN) {  // the episode's first selection: kept for the next reset
                const size_t WA = (size_t)d.W * A_T;
                for (int q = 0; q < cp_count; q++) {
                    d.cp_road[(WA + i) * GD_RANK_NCP + q] = cpr[q];
                    d.cp_T[(WA + i) * GD_RANK_NCP + q] = cpt[q];
                }
                d.cp_hdr[WA + i] = make_float4(ex, ey, __int_as_float(cp_count), 0.f);
            }
        }
        if (lane == 0) {
            d.rk_fallback[slot] = 0;  // read at the top by every lane of this (only) wave of the group
            d.rk_streak[slot] = min(d.rk_streak[slot] + 1, 1 << 20);  // k_knn_finish zeroes it when the group gets through
        }
    }
    STAMP(t_w0);
    store_selection<A_T>(d, w, a0, min(AW, n - a0), [&](int c, int sl) -> int { return s_idx[(sl + 1) * AW + idx_col(c)]; }, lane, 64);
    if (lane == 0) d.wave_cost[slot] = (unsigned int)min(__builtin_amdgcn_s_memtime() - t_launch, 0xffffffffull);
#ifdef GD_STAMPS
    if (lane == 0 && blockIdx.x < 8192) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_stamps[blockIdx.x];
        o[0] = t_end - t_begin; o[1] = st_init; o[2] = st_scan; o[3] = st_drain; o[4] = st_rounds; o[5] = st_scans;
        o[6] = st_wscan; o[7] = st_wround;
    }
#endif
}

// ---- launch order of the next k_map_obs ----
// The LDS lets 1024 of its workgroups run at a time and a launch has 2048 or more, whose durations differ by +-20 % with
// the number of heap inserts their agents need; in index order the launch ends when an unlucky slot finishes its second
// long workgroup (measured: 18 % above the balanced time on the bench scene).  The agents move a fraction of a metre per
// step, so a workgroup's cycles in this launch predict the next: the workgroups are re-sorted longest first (counting
// sort over 256 cost buckets; the order inside a bucket is whatever the atomics give -- it only ever changes WHEN a
// workgroup runs, never what it computes).  The first workgroup of k_map_rows does it on its way in (order_waves above
// that kernel): the row kernel runs right after the selection and nothing reads the order before the next step.

// ---- set-order mode (gd_config.knn_order = GD_KNN_SET_ORDER) ----
//
// Same row SET as the reference (the K nearest by (distance^2, road index), then the radius filter), rows in the order
// of the world's cell-sorted road list (grid cell by grid cell, ascending road index inside a cell: what the gather yields;
// the full-stream path re-orders its selection into it, SetSel::to_cell_order) instead of the reference's heap-history order.  Because the radius filter runs after the
// top-K, the result is "every in-radius road" whenever fewer than K roads are in radius, and the K smallest of the
// in-radius roads otherwise; no heap is needed.  Each wave takes its agents one at a time, all 64 lanes cooperating:
//
//   * BOUND.  The K roads selected last step lie within sqrt(T) of where the agent then was (T = that step's K-th
//     key), hence within sqrt(T) + |movement| of where it is now: the new K-th distance cannot exceed that.  Only
//     roads within min(radius, that bound) are candidates -- a few hundred instead of every in-radius road (1,700 of
//     4,096 on the bench scene).  The bound is exact arithmetic on a conservative side (margin), holds for any
//     reference point as long as the world's roads are the same (set_maps resets it), and a teleported agent simply
//     gets the radius back.
//   * GATHER through the world's road grid (engine.cpp build_road_grid): the cells a row of the bound's box crosses
//     are one contiguous run of the CSR, so the lanes stride over whole rows; exact keys (ego_dist2), ballot
//     compaction of (key, road) into LDS.
//   * SELECT, only when more than K candidates: search on the key bits for the K-th smallest key with three probes
//     per pass (counts packed in one register, one cross-lane reduction); ties at the K-th key go to the lowest road
//     indices (the reference breaks such ties by heap position, the one documented difference of the set).
//   * WRITE-OUT: the selected candidates leave in candidate order, one ballot and one store per 64 of them, straight
//     into the selection scratch of k_map_rows.  (Rounds 1-3 marked them in a bitmap of the world's roads and read it
//     back in ascending road index: a third of the kernel for an order the mode does not promise.)
// The linear scan (AllEntitiesWithRadiusFiltering: first K in index order within the radius) and the rare agent with
// more candidates than the LDS buffer holds take the full-stream path (select_streaming).
template <int A_T>
struct SetSel {
    static constexpr int CAP = GD_SET_CAP;   // (key, road) candidates a wave holds in LDS
    static constexpr int BMW = 256;    // words of the key histogram (and, before it, of the gather's piece list)

    // Full-stream selection (round 1's kernel): every road of the world, 256 per iteration; candidates recomputed from
    // the stream when they do not fit the buffer.  Writes the selected road indices, ascending, to out[0..count).
    static __device__ __forceinline__ int select_streaming(const DevSim &d, const float2 *rxy, int R, bool knn, float ex, float ey, float iw,
                                                           float iz, float *ckey, unsigned short *cidx, unsigned short *out, int lane,
                                                           float &kth) {
        const unsigned long long lower = (1ull << lane) - 1ull;
        const float kmax = d.radius_key_max;
        auto in_radius = [&](float key) -> bool { return knn ? (key <= kmax) : !(key > kmax); };
        int nin = 0;
        constexpr int U = 4;
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
                const float key = ego_dist2(ex, ey, iw, iz, cur[u].x, cur[u].y);
                const bool in = r < R && in_radius(key);
                const unsigned long long b = __ballot(in);
                const int pos = nin + __popcll(b & lower);
                if (in && pos < CAP) { ckey[pos] = key; cidx[pos] = (unsigned short)r; }
                nin += __popcll(b);
            }
        }
        wave_sync();
        int count = 0;
        kth = __builtin_inff();
        if (nin <= K || !knn) {
            count = min(nin, K);
            for (int j = lane; j < count; j += 64) out[j] = cidx[j];
        } else {
            auto key_bits = [&](int j, int r) -> unsigned int {  // candidate j of the buffer, or road r of the stream
                if (nin < CAP) return j < nin ? __float_as_uint(ckey[j]) : 0xffffffffu;
                if (r >= R) return 0xffffffffu;
                const float2 xy = rxy[r];
                const float key = ego_dist2(ex, ey, iw, iz, xy.x, xy.y);
                return in_radius(key) ? __float_as_uint(key) : 0xffffffffu;
            };
            const int n = nin < CAP ? nin : R;
            unsigned int lo = 0u, hi = 0x7f800000u;
            while (lo < hi) {
                const unsigned int mid = lo + (hi - lo + 1) / 2;
                int cnt = 0;
                for (int jb = 0; jb < n; jb += 64) cnt += __popcll(__ballot(key_bits(jb + lane, jb + lane) < mid));
                if (cnt < K) lo = mid; else hi = mid - 1;
            }
            int less = 0;
            for (int jb = 0; jb < n; jb += 64) less += __popcll(__ballot(key_bits(jb + lane, jb + lane) < lo));
            int need_ties = K - less;
            for (int jb = 0; jb < n; jb += 64) {  // buffer and stream are both in road order: earliest ties win
                const int j = jb + lane;
                const unsigned int kb = key_bits(j, j);
                const bool tie = kb == lo;
                const unsigned long long tb = __ballot(tie);
                const bool take = kb < lo || (tie && (int)__popcll(tb & lower) < need_ties);
                need_ties -= min(need_ties, (int)__popcll(tb));
                const unsigned long long kb2 = __ballot(take);
                if (take) out[count + __popcll(kb2 & lower)] = nin < CAP ? cidx[j] : (unsigned short)j;
                count += __popcll(kb2);
            }
            kth = __uint_as_float(lo);
        }
        return count;
    }

    // The selection above lists its roads in ascending road index; the mode's row order is the order of the world's
    // cell-sorted road list (what the grid gather of k_map_obs_set yields: grid cell by grid cell, ascending road index inside a
    // cell).  An agent takes the full-stream path after every reset and the grid path on the steps that follow: both must
    // give ONE order, or a consumer that flattens the rows sees them permuted from one step to the next.  out[0..count) is
    // re-ordered by each road's position in that list (engine.hpp rcell_pos): the positions are marked in a bitmap of the
    // world's roads, and a road's new slot is the number of marked positions below its own.  `words`: GD_MAX_ROAD_ENTITIES / 32
    // + 1 words of LDS, `tmp`: K entries.
    static __device__ __attribute__((noinline)) void to_cell_order(const uint16_t *pos_of, unsigned short *out, int count, unsigned int *words,
                                                         unsigned short *tmp, int lane) {
        constexpr int NWORD = (GD_MAX_ROAD_ENTITIES + 31) / 32, PER = (NWORD + 63) / 64;
        for (int q = lane; q < NWORD; q += 64) words[q] = 0u;
        wave_sync();
        constexpr int NP = (K + 63) / 64;
        unsigned int road[NP], pos[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int j = p * 64 + lane;
            road[p] = j < count ? out[j] : 0u;
            pos[p] = pos_of[road[p]];
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
            if (p * 64 + lane < count) atomicOr(&words[pos[p] >> 5], 1u << (pos[p] & 31u));
        wave_sync();
        // marked positions below each lane's PER consecutive words (exclusive scan over the lanes)
        int own = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) own += lane * PER + k < NWORD ? __popc(words[lane * PER + k]) : 0;
        int incl = own;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
        const int excl = incl - own;
#pragma unroll
        for (int p = 0; p < NP; p++) {  // (every lane runs this: the shuffle reads lanes that hold no road of their own)
            const int wd = (int)(pos[p] >> 5), owner = wd / PER;
            int below = __shfl(excl, owner);
            for (int k = owner * PER; k < wd; k++) below += __popc(words[k]);
            below += __popc(words[wd] & ((1u << (pos[p] & 31u)) - 1u));
            if (p * 64 + lane < count) tmp[below] = (unsigned short)road[p];
        }
        wave_sync();
        for (int j = lane; j < count; j += 64) out[j] = tmp[j];
        wave_sync();
    }
};

#ifndef GD_SET_ABL
#define GD_SET_ABL 0
#endif
#ifndef GD_SET_PB
#define GD_SET_PB 2  // pieces of the grid rows' runs requested together (k_map_obs_set)
#endif
#ifdef GD_CLOCKS
// -DGD_CLOCKS builds (tools/build_expt.sh clk -DGD_CLOCKS; tools/set_phases.py): clock ticks per phase of the set-order
// selection summed over the waves (gd_stat 22..29): 0 prologue, 1 gather, 2 key registers + histogram, 3 K-th key,
// 4 ties + write-out, 5 header; 6 = gather iterations, 7 = candidates
__device__ unsigned long long g_set_clk[8];
#define SET_PHASE(n) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk_sum[n] += (unsigned int)(t_ - clk_prev); clk_prev = t_; } while (0)
#else
#define SET_PHASE(n) do {} while (0)
#endif
#ifndef GD_SET_WPE
#define GD_SET_WPE 4
#endif
template <int A_T, int NW, bool FUSE, bool PACK = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(GD_SET_WPE))) void k_map_obs_set(DevSim d) {
    static_assert(FUSE || !PACK, "the packed rows are written by the fused write-out");
    using S = SetSel<A_T>;
    constexpr int CAP = S::CAP, BMW = S::BMW;
    constexpr int KR_MAX = 5;  // register slots of the K-th-key search: 64 candidates each
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int wg = d.set_groups[blockIdx.x];  // only groups of agent slots that hold a live agent are launched
    const int w = wg >> 8, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int n = d.shape[w * 2 + 0];
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const bool knn = d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST;

    __shared__ float s_ckey[NW][CAP];
    __shared__ unsigned short s_cidx[NW][CAP];
    __shared__ __attribute__((aligned(16))) unsigned int s_bits[NW][BMW];
    __shared__ unsigned short s_sel[NW][K];
    float *ckey = s_ckey[wave];
    unsigned short *cidx = s_cidx[wave];
    unsigned int *bits = s_bits[wave];
    const float2 *rxy = d.road_xy + r0;
    const GridHdr g = d.rgrid[w];
    const int32_t *coff = d.rcell_off + g.cell_base;
    const uint16_t *citems = d.rcell_items + g.item_base;
    const float2 *cxy = d.rcell_xy + g.item_base;
    const float kmax = d.radius_key_max;

    // a workgroup takes NW * set_apw consecutive agents of its world (set_groups), each wave set_apw of them
    const int a_first = (wg & 255) * (NW * d.set_apw), a_end = min(n, a_first + NW * d.set_apw);
#ifdef GD_CLOCKS
    unsigned int clk_sum[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    unsigned long long clk_prev = __builtin_amdgcn_s_memtime();
#endif
    // The selection is a chain of memory round trips (pose -> the grid rows' runs -> the pieces of those runs), and with
    // four waves per SIMD nothing else covers them: what an agent needs is requested while the one before it is still
    // being worked on (pose and last step's bound at the top of the previous agent, the runs before its K-th-key search),
    // and the pieces of a run are requested GD_SET_PB at a time, the next batch before the current one is keyed.
    struct AgentIn { float ex, ey, qw, qz; float4 pv; uint4 st; };
    struct Plan { float bound, floor_key; int nrows, run_lo, run_hi; };
    const bool use_grid = knn && R > 0;
    auto load_agent = [&](int a) -> AgentIn {
        const size_t i = (size_t)w * A_T + a;
        return AgentIn{d.px[i], d.py[i], d.qw[i], d.qz[i], d.knn_prev[i], d.pose_stamp[i]};
    };
    // candidates: within the radius AND within what the previous selection allows (the K-th distance is 1-Lipschitz in the
    // agent's position: it also cannot fall below sqrt(T) - |movement|, which gives the search for the new T a narrow
    // starting interval); the rows of the grid the bound's box crosses (a slightly larger box: cells are chosen in plain
    // float) and the run of every such row (lane q holds row cy0 + q; the grid has at most 64 rows).  Roads left of column
    // 0 / right of the last column were clamped into the border cells at build time; an agent outside the grid reaches
    // them through the clamped cell range.
    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: every use below carries a 1e-4 margin
    auto fast_sqrt = [](float x) -> float { return __builtin_amdgcn_sqrtf(x); };
    const float cell_size = 1.f / g.inv_cell;
    auto plan = [&](const AgentIn &in) -> Plan {
        Plan pl{kmax, 0.f, 0, 0, 0};
        const float4 pv = in.pv;
        if (pv.z < __builtin_inff()) {
            const float dx = in.ex - pv.x, dy = in.ey - pv.y;
            const float move = fast_sqrt(dx * dx + dy * dy);
            const float reach = fast_sqrt(pv.z) * 1.0001f + move * 1.0001f + 1e-3f;
            pl.bound = fminf(pl.bound, reach * reach * 1.0001f);
            const float near = fast_sqrt(pv.z) * 0.9999f - move * 1.0001f - 1e-3f;
            pl.floor_key = near > 0.f ? near * near * 0.9999f : 0.f;
        }
        const float rr = fast_sqrt(pl.bound) * 1.001f + 1e-2f;
        const int cy0 = max(0, min(g.ny - 1, (int)floorf((in.ey - rr - g.oy) * g.inv_cell)));
        const int cy1 = max(0, min(g.ny - 1, (int)floorf((in.ey + rr - g.oy) * g.inv_cell)));
        pl.nrows = cy1 - cy0 + 1;
        // lane q: row cy0 + q, cut to the chord of the bound's circle at the edge of the row nearer to the agent (the box
        // holds 4 / pi times the circle's roads); lanes beyond the box read a row of the grid all the same (no branch
        // around the loads)
        const int row = min(cy0 + lane, g.ny - 1);
        const float band_lo = g.oy + (float)row * cell_size, band_hi = band_lo + cell_size;
        const float dy = fmaxf(fmaxf(band_lo - in.ey, in.ey - band_hi) - 1e-2f, 0.f);
        const float hw = fast_sqrt(fmaxf(rr * rr - dy * dy, 0.f)) * 1.001f + 1e-2f;
        const int cx0 = max(0, min(g.nx - 1, (int)floorf((in.ex - hw - g.ox) * g.inv_cell)));
        const int cx1 = max(0, min(g.nx - 1, (int)floorf((in.ex + hw - g.ox) * g.inv_cell)));
        pl.run_lo = coff[row * g.nx + cx0];
        pl.run_hi = coff[row * g.nx + cx1 + 1];
        return pl;
    };
    AgentIn nx{0.f, 0.f, 1.f, 0.f, make_float4(0.f, 0.f, 0.f, 0.f), make_uint4(0u, 0u, 0u, 0u)};
    int skipped = 0;
    Plan npl{kmax, 0.f, 0, 0, 0};
    if (a_first + wave < a_end) {
        nx = load_agent(a_first + wave);
        if (use_grid) npl = plan(nx);
    }
    for (int a = a_first + wave; a < a_end; a += NW) {  // wave-uniform
        const size_t i = (size_t)w * A_T + a;
        const AgentIn cur_in = nx;
        const Plan pl = npl;
        const bool more = a + NW < a_end;
        if (more) nx = load_agent(a + NW);
        const float ex = cur_in.ex, ey = cur_in.ey;
        const float iw = cur_in.qw, iz = -cur_in.qz;  // the INVERSE rotation
        if (FUSE) {
            // rows already written for exactly this pose (engine.hpp pose_stamp: it dies with the world's roads) are left in
            // place: the agent did not move, so it selects the same roads and sees them in the same place (its bound,
            // knn_prev, stays what it is).  Without the fused write-out k_map_rows decides the same from the header.
            const uint4 st = cur_in.st;
            if (d.pose_skip != 0 && st.x != 0xffffffffu && st.x == __float_as_uint(ex) && st.y == __float_as_uint(ey) &&
                st.z == __float_as_uint(cur_in.qw) && st.w == __float_as_uint(cur_in.qz)) {
                skipped++;
                if (more && use_grid) npl = plan(nx);
                continue;
            }
        }
        unsigned short *out = FUSE ? s_sel[wave] : d.sel_idx + i * K;  // the selected road indices, in the mode's order
        int count = 0;
        float kth = __builtin_inff();
        bool done = false;
        if (use_grid) {
            const float bound = pl.bound, floor_key = pl.floor_key;
            const int nrows = pl.nrows;
            const int run_lo = pl.run_lo, run_hi = pl.run_hi;
            // The rows' runs are cut into units of up to 32 consecutive roads (a row of the box holds about that many on the
            // bench scene: 64-road pieces were half empty); the list of units (first item | length << 16) is laid out in LDS
            // by the lanes that hold the rows (lane q: row q), and a piece is two units, one per half of the wave: the gather
            // loop is a plain count over pieces with nothing to decide in it.  (index, x, y) come from the cell-sorted
            // copies, PB pieces are requested together into one of two register sets, and every load is unconditional (a
            // unit past the end reads item 0) so that the waits count loads instead of draining them.
            constexpr int PB = GD_SET_PB;
            static_assert(GD_MAX_ROAD_ENTITIES < 65536, "a unit's first item in 16 bits");
            constexpr int UMAX = BMW;  // units the list holds (it borrows the histogram's words, which are zeroed after the gather)
            unsigned int *unit = bits;
            int nunits;
            {
                const int len = lane < nrows ? max(run_hi - run_lo, 0) : 0;
                const int nu_row = (len + 31) >> 5;
                int incl = nu_row;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
                nunits = __builtin_amdgcn_readlane(incl, 63);
                const int base = incl - nu_row;
                for (int k = 0; k < nu_row && base + k < UMAX; k++)
                    unit[base + k] = (unsigned int)(run_lo + 32 * k) | (unsigned int)min(32, len - 32 * k) << 16;
            }
            wave_sync();
            const int npieces = (nunits + 1) >> 1;
            const int half = lane >> 5;
            const unsigned int hl = (unsigned int)(lane & 31);
            int nin = 0;
            struct Batch { unsigned int r[PB]; float2 xy[PB]; unsigned int left[PB]; };  // left: roads of this lane's unit, 0 = none
            auto issue = [&](Batch &bt, int p0) {
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    const int ui = 2 * (p0 + u) + half;
                    const unsigned int ds = ui < nunits ? unit[ui] : 0u;
                    bt.left[u] = ds >> 16;
                    const unsigned int j = (ds & 0xffffu) + min(hl, max(ds >> 16, 1u) - 1u);
                    bt.r[u] = citems[j];
                    bt.xy[u] = cxy[j];
                }
            };
            auto take = [&](const Batch &bt) {
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    const float key = ego_dist2(ex, ey, iw, iz, bt.xy[u].x, bt.xy[u].y);
                    const bool in = hl < bt.left[u] && key <= bound;
                    const unsigned long long b = __ballot(in);
                    const int pos = nin + bits_below_lane(b);
                    if (in && pos < CAP) { ckey[pos] = key; cidx[pos] = (unsigned short)bt.r[u]; }
                    nin += __popcll(b);
                }
            };
            SET_PHASE(0);
#if GD_SET_ABL == 2  // timing-only builds: the gather twice (the same candidates again), to read its cost off the counters
            for (int rep = 0; rep < 2; rep++) {
                nin = 0;
                asm volatile("" ::: "memory");
#endif
            if (nunits <= UMAX) {
                Batch ba, bb;
                issue(ba, 0);
                for (int p0 = 0; p0 < npieces; p0 += 2 * PB) {
                    issue(bb, p0 + PB);
                    take(ba);
                    if (p0 + PB >= npieces) break;
                    issue(ba, p0 + 2 * PB);
                    take(bb);
                }
            } else {
                nin = CAP + 1;  // more units than the list holds: the full-stream path
            }
#if GD_SET_ABL == 2
            }
#endif
#ifdef GD_CLOCKS
            clk_sum[6] += (unsigned int)npieces;
#endif
            // the next agent's runs: they arrive while this one's K-th key is searched for
            if (more) npl = plan(nx);
            wave_sync();
            SET_PHASE(1);
#ifdef GD_CLOCKS
            clk_sum[7] += (unsigned int)nin;
#endif
            if (nin <= KR_MAX * 64) {  // (more: the full-stream path below -- a first selection with the radius for a bound, as a rule)
                done = true;
                reinterpret_cast<uint4 *>(bits)[lane] = make_uint4(0u, 0u, 0u, 0u);  // the histogram's words
                wave_sync();
                if (nin < K || GD_SET_ABL == 1) {  // fewer than K roads within the bound: then the bound is the radius, and all of them are selected
                    for (int j = lane; j < min(nin, K); j += 64) out[j] = cidx[j];
                    count = min(nin, K);
                } else {
                  // the K-th smallest key T: largest T with count(key < T) < K.  The candidates' key bits sit in
                  // registers (lane l holds candidates l, l + 64, ...): five slots, 320 candidates -- the coherence bound
                  // leaves about 240 of them on the bench scene.  (Rounds 2-4 also had a sixteen-slot instantiation for up to
                  // CAP candidates: every pass over the slots is unrolled with a guard per slot, and its 141 registers cost
                  // the kernel either a wave per SIMD or a dozen spilled registers on the path everybody takes: 164 -> 153 us
                  // without it.  Agents with more candidates than the slots hold take the full-stream path.)
                  auto select = [&](auto slots_tag) {
                    constexpr int KR = decltype(slots_tag)::value;
                    unsigned int kb[KR];
#pragma unroll
                    for (int u = 0; u < KR; u++) kb[u] = u * 64 + lane < nin ? __float_as_uint(ckey[u * 64 + lane]) : 0xffffffffu;
                    const int nu = (nin + 63) >> 6;
                    auto count_below = [&](unsigned int p) -> int {
                        int c = 0;
#pragma unroll
                        for (int u = 0; u < KR; u++)
                            if (u < nu) c += __popcll(__ballot(kb[u] < p));
                        return c;
                    };
                    // T lies in [floor_key, bound]: count(key < floor_key) < K (fewer than K roads can be that close), and
                    // every candidate is <= bound.  A 256-bucket histogram over that interval (bucket = a monotone function of
                    // the key, LDS atomics on the still-empty bitmap words) names the bucket that holds T and how many keys lie
                    // below it; T is then the m-th smallest key of that bucket, found by stepping through the bucket's distinct
                    // values (usually one or two).  The search on the key bits further down is the fall-back.
                    unsigned int lo = 0u;
                    bool found = false;
                    {
                        unsigned int *hist = bits;  // words 0..255: zero now, zeroed again before the bitmap is used
                        const float span = bound - floor_key;
                        const float scale = span > 0.f ? 256.f / span : 0.f;
                        auto bucket_of = [&](unsigned int kbits) -> int {
                            const float t = (__uint_as_float(kbits) - floor_key) * scale;
                            return min(255, max(0, (int)t));
                        };
                        int bk[KR];
#pragma unroll
                        for (int u = 0; u < KR; u++) {
                            bk[u] = -1;
                            if (u < nu && u * 64 + lane < nin) {
                                bk[u] = bucket_of(kb[u]);
                                atomicAdd(&hist[bk[u]], 1u);
                            }
                        }
                        wave_sync();
                        // lane l owns buckets 4l .. 4l + 3; inclusive prefix over the lanes (DPP), the lane where it crosses K
                        const uint4 h = reinterpret_cast<const uint4 *>(hist)[lane];
                        const int own = (int)(h.x + h.y + h.z + h.w);
                        int incl = own;
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
                        const int excl = incl - own;
                        reinterpret_cast<uint4 *>(hist)[lane] = make_uint4(0u, 0u, 0u, 0u);
                        int jb = 0, before = excl;  // this lane's answer if the crossing is in its four buckets
                        if (before + (int)h.x < K) { before += (int)h.x; jb = 1;
                            if (before + (int)h.y < K) { before += (int)h.y; jb = 2;
                                if (before + (int)h.z < K) { before += (int)h.z; jb = 3; } } }
                        SET_PHASE(2);
                        const unsigned long long cross = __ballot(excl < K && incl >= K);  // exactly one lane: nin >= K
                        const int L = __ffsll((long long)cross) - 1;
                        const int bucket = 4 * L + __builtin_amdgcn_readlane(jb, L);
                        const int m = K - __builtin_amdgcn_readlane(before, L);  // T is the m-th smallest key of the bucket, m >= 1
                        unsigned int inb = 0u;
#pragma unroll
                        for (int u = 0; u < KR; u++) inb |= (bk[u] == bucket ? 1u : 0u) << u;
                        unsigned int prev = 0u;
                        bool first = true;
                        int acc = 0;
                        for (int it = 0; it < 48 && cross != 0ull; it++) {
                            unsigned int cand = 0xffffffffu;
#pragma unroll
                            for (int u = 0; u < KR; u++)
                                if (u < nu && ((inb >> u) & 1u) && (first || kb[u] > prev)) cand = min(cand, kb[u]);
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x111, 0xf, 0xf, false));
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x112, 0xf, 0xf, false));
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x114, 0xf, 0xf, false));
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x118, 0xf, 0xf, false));
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x142, 0xa, 0xf, false));
                            cand = min(cand, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)cand, 0x143, 0xc, 0xf, false));
                            cand = (unsigned int)__builtin_amdgcn_readlane((int)cand, 63);  // the smallest value of the bucket above prev
                            if (cand == 0xffffffffu) break;
                            int c = 0;
#pragma unroll
                            for (int u = 0; u < KR; u++)
                                if (u < nu) c += __popcll(__ballot(((inb >> u) & 1u) && kb[u] == cand));
                            acc += c;
                            if (acc >= m) { lo = cand; found = true; break; }
                            prev = cand;
                            first = false;
                        }
                        wave_sync();  // the histogram words are zero again before anything is marked
                    }
                    if (!found) {
                        unsigned int hi = __float_as_uint(bound);
                        lo = __float_as_uint(floor_key);
                        while (lo < hi) {
                            const unsigned int step = (hi - lo + 3u) / 4u;  // >= 1
                            const unsigned int p1 = lo + step, p2 = min(hi, p1 + step), p3 = min(hi, p2 + step);
                            int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
                            for (int u = 0; u < KR; u++) {
                                if (u < nu) {
                                    c1 += __popcll(__ballot(kb[u] < p1));
                                    c2 += __popcll(__ballot(kb[u] < p2));
                                    c3 += __popcll(__ballot(kb[u] < p3));
                                }
                            }
                            if (c3 < K) lo = p3;
                            else if (c2 < K) { lo = p2; hi = p3 - 1u; }
                            else if (c1 < K) { lo = p1; hi = p2 - 1u; }
                            else hi = p1 - 1u;
                        }
                    }
                    // everything below T, then the lowest road indices among the ties at T
                    SET_PHASE(3);
                    const int less = count_below(lo);
                    // The rows leave in the order the candidates were gathered in (grid cell by grid cell, ascending road index
                    // inside a cell): one ballot and one store per 64 candidates.  (Round 2 marked the selected roads in a bitmap
                    // of the world's roads and read it back in ascending road index: a zeroing pass, 200 LDS atomics and a
                    // bit-by-bit loop that runs as long as the fullest word of the wave -- a third of this kernel's
                    // instructions for an order nobody asked for: the mode's contract is the row SET.)
                    unsigned int rd[KR];
#pragma unroll
                    for (int u = 0; u < KR; u++) rd[u] = u < nu ? (unsigned int)cidx[u * 64 + lane] : 0u;
                    int nties = 0;
#pragma unroll
                    for (int u = 0; u < KR; u++)
                        if (u < nu) nties += __popcll(__ballot(kb[u] == lo));
                    int base = 0;
                    if (less + nties == K) {  // every candidate at T is kept (one candidate at T, as a rule): key <= T is the whole test
#pragma unroll
                        for (int u = 0; u < KR; u++) {
                            if (u < nu) {  // wave-uniform
                                const bool take = kb[u] <= lo;  // (beyond nin: key bits are all ones)
                                const unsigned long long tb = __ballot(take);
                                if (take) out[base + bits_below_lane(tb)] = (unsigned short)rd[u];
                                base += __popcll(tb);
                            }
                        }
                    } else {
                        unsigned int tie_taken = 0u;  // bit u: this lane's candidate u is one of the ties at T that are kept
                        unsigned int floor_idx = 0;   // ties with a road index below this are already taken
                        for (int t = less; t < K; t++) {
                            unsigned int best = 0xffffffffu;
#pragma unroll
                            for (int u = 0; u < KR; u++)
                                if (u < nu && kb[u] == lo && rd[u] >= floor_idx) best = min(best, rd[u]);
                            for (int off = 32; off > 0; off >>= 1) best = min(best, (unsigned int)__shfl_xor((int)best, off));
                            if (best == 0xffffffffu) break;  // cannot happen: at least K candidates have a key <= T
#pragma unroll
                            for (int u = 0; u < KR; u++) tie_taken |= (u < nu && kb[u] == lo && rd[u] == best) ? 1u << u : 0u;
                            floor_idx = best + 1u;
                        }
#pragma unroll
                        for (int u = 0; u < KR; u++) {
                            if (u < nu) {  // wave-uniform
                                const bool take = kb[u] < lo || ((tie_taken >> u) & 1u) != 0u;
                                const unsigned long long tb = __ballot(take);
                                if (take) out[base + bits_below_lane(tb)] = (unsigned short)rd[u];
                                base += __popcll(tb);
                            }
                        }
                    }
                    count = K;
                    kth = __uint_as_float(lo);
                  };
#if GD_SET_ABL == 3  // timing-only builds: the selection twice
                  select(std::integral_constant<int, KR_MAX>{});
                  asm volatile("" ::: "memory");
#endif
                  select(std::integral_constant<int, KR_MAX>{});
                }
                wave_sync();
                SET_PHASE(4);
            }
        }
        if (!done) {
            count = min(S::select_streaming(d, rxy, R, knn, ex, ey, iw, iz, ckey, cidx, out, lane, kth), K);
            // k-NN: into the mode's one row order (the linear scan's order IS ascending road index)
            if (knn) S::to_cell_order(d.rcell_pos + r0, out, count, reinterpret_cast<unsigned int *>(ckey), cidx, lane);
        }
        count = min(count, K);
        if (lane == 0) {
            d.knn_prev[i] = make_float4(ex, ey, kth, 0.f);
            if (!FUSE) write_header(d, i, ex, ey, iw, -iz, count, r0);
        }
        wave_sync();
        SET_PHASE(5);
        if (!FUSE) continue;  // k_map_rows writes the rows
        // ---- FUSE: the agent's K rows, written by the wave that selected them.  With several generations of workgroups
        // (many worlds) the HBM-bound row stores of one workgroup overlap the issue-bound selection of the others; when every
        // workgroup is resident at once the separate, fully occupied k_map_rows is the faster write-out (launch_map_obs) ----
        constexpr int NP = (K + 63) / 64;
        static_assert(K % 4 == 0 && S::CAP >= 64 * 9, "whole 16-byte pieces; the staging block fits the key buffer");
        float4 q0[NP], q1[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {  // every gather of the agent requested at once
            const int sl = p * 64 + lane;
            const int r = r0 + (sl < count ? (int)out[sl] : 0);
            q0[p] = d.road_rec[(size_t)r * 2];
            q1[p] = d.road_rec[(size_t)r * 2 + 1];
        }
        // 64 rows at a time are laid out in LDS (the candidate keys' buffer is free now) and leave as whole 16-byte pieces
        // of one contiguous block: an agent's rows start at a multiple of 7200 bytes and 64 rows are 2304, both multiples of 16
        float *stage = ckey;
        float *rows_out = d.agent_map + i * (size_t)(K * 9);
        typedef float f4 __attribute__((ext_vector_type(4)));
        constexpr int PACK_ROAD0 = 6 + (A_T - 1) * 6, PACK_D = PACK_ROAD0 + K * 13;  // the packed row: ego | partners | road points
        static_assert(S::CAP >= 64 * 13 && PACK_ROAD0 % 4 == 0 && PACK_D % 4 == 0 && ((K % 64) * 13) % 4 == 0, "the packed rows' staging block; whole 16-byte pieces");
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int sl = p * 64 + lane;
            const int nrows = min(64, K - p * 64);
            if (!PACK) {
                road_row(stage + lane * 9, sl < count, knn, ex, ey, iw, -iz, q0[p], q1[p]);
                wave_sync();
                for (int q = lane; q < nrows * 9 / 4; q += 64)
                    __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(rows_out + p * 576 + q * 4));
                wave_sync();
                continue;
            }
            // PACK (gd_attach_packed): the row in registers, stored raw unless pack_only, then in the packed observation's 13
            // normalised columns (pack_cols.hpp)
            float raw[9];
            road_row(raw, sl < count, knn, ex, ey, iw, -iz, q0[p], q1[p]);
            if (!d.pack_only) {
#pragma unroll
                for (int c = 0; c < 9; c++) stage[lane * 9 + c] = raw[c];
                wave_sync();
                for (int q = lane; q < nrows * 9 / 4; q += 64)
                    __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(rows_out + p * 576 + q * 4));
                wave_sync();
            }
            pack_road_row(raw, stage + lane * 13);
            wave_sync();
            float *pout = d.pack + i * (size_t)PACK_D + PACK_ROAD0 + p * (64 * 13);
            for (int q = lane; q < nrows * 13 / 4; q += 64)
                __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(pout + q * 4));
            wave_sync();
        }
        if (lane == 0)
            d.pose_stamp[i] = make_uint4(__float_as_uint(ex), __float_as_uint(ey), __float_as_uint(cur_in.qw), __float_as_uint(cur_in.qz));
    }
    if (FUSE && lane == 0 && skipped) atomicAdd(d.stat_skipped + ((blockIdx.x * NW + wave) & (GD_SKIP_SLOTS - 1)), (unsigned long long)skipped);
#ifdef GD_CLOCKS
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&g_set_clk[k], (unsigned long long)clk_sum[k]);
#endif
}

}  // namespace

#ifdef GD_CLOCKS
void set_clocks_read(unsigned long long *out) {  // and zero them
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_set_clk), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_set_clk), z, sizeof(z));
}
#endif

void launch_map_obs(const DevSim &d, hipStream_t st, bool move) {
    // AllEntitiesWithRadiusFiltering: rows in road-index order whatever knn_order says -- a kernel of its own (map_obs_linear.hip)
    if (d.p.roadObservationAlgorithm != GD_ROADS_K_NEAREST && d.lin_on) {
        launch_map_obs_linear(d, st, move);
        return;
    }
    if (d.knn_order == GD_KNN_SET_ORDER) {
        if (d.set_group_count == 0) return;
        const dim3 grid(d.set_group_count);
        if (d.set_fused_rows) {
            if (d.pack != nullptr) {
                if (d.A == 64) hipLaunchKernelGGL((k_map_obs_set<64, 4, true, true>), grid, dim3(256), 0, st, d);
                else hipLaunchKernelGGL((k_map_obs_set<128, 4, true, true>), grid, dim3(256), 0, st, d);
                return;
            }
            if (d.A == 64) hipLaunchKernelGGL((k_map_obs_set<64, 4, true>), grid, dim3(256), 0, st, d);
            else hipLaunchKernelGGL((k_map_obs_set<128, 4, true>), grid, dim3(256), 0, st, d);
            return;
        }
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs_set<64, 4, false>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs_set<128, 4, false>), grid, dim3(256), 0, st, d);
    } else {
        if (d.rk_on && d.p.roadObservationAlgorithm == GD_ROADS_K_NEAREST) launch_map_obs_rank(d, st);
        const dim3 grid(d.W * (d.A / AW));
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs<64>), grid, dim3(64), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs<128>), grid, dim3(64), 0, st, d);
    }
    const size_t agents = (size_t)d.W * d.A;
    if (d.pack != nullptr) {
        constexpr int AB = RowsGeo<true>::AB;
        const dim3 pgrid((unsigned int)((agents + AB - 1) / AB + 7) / 8 * 8);
        if (d.A == 64) hipLaunchKernelGGL((k_map_rows<64, true>), pgrid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_map_rows<128, true>), pgrid, dim3(256), 0, st, d);
        return;
    }
    constexpr int AB = RowsGeo<false>::AB;
    const dim3 rgrid((unsigned int)((agents + AB - 1) / AB + 7) / 8 * 8);
    if (d.A == 64) hipLaunchKernelGGL((k_map_rows<64>), rgrid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_map_rows<128>), rgrid, dim3(256), 0, st, d);
}

}  // namespace gd
