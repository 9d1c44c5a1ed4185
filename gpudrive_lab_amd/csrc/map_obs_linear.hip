// Road observations, linear mode: FindRoadObservationsWith::AllEntitiesWithRadiusFiltering, the `EnvConfig` default of the
// reference's callers (gpudrive/env/config.py:51 -> gpudrive/env/base_env.py:86-93 -> reference src/sim.cpp:258-279):
//
//     while (roadIdx < numRoads && arrIndex < K) { if (distanceTo(road) > radius) continue; obs[arrIndex++] = observationOf(road); }
//     while (arrIndex < K) obs[arrIndex++] = MapObservation::zero();
//
// The first K roads IN INDEX ORDER within the radius, rows in that order: no heap, no selection -- a prefix scan that stops
// at K, then 7,200 bytes of rows per agent.  The kernel is what the algorithm is:
//
//   * one wave per agent at a time (a workgroup of four waves takes 4 * lin_apw consecutive entries of the agent list).  Roads
//     follow their polylines in index order, so 16 consecutive roads are a short piece of one polyline: the engine keeps the
//     circle around every such block (engine.hpp road_blk), and the wave first drops the blocks that cannot hold a road in
//     reach -- a lane per block, 1024 roads per wave instruction.  On the bench scene 270 of a world's 4096 roads are in
//     reach of an agent and the K-th of them sits at index 2900 on average (3400 at t = 0): the reference's loop visits
//     three quarters of the world, the cull leaves a seventh of it.
//   * the surviving blocks are scanned in index order, four per pass: the lanes read their roads' (x, y), key them exactly
//     as the reference does (gd_math.hpp ego_dist2; the radius test on squared keys, engine.cpp radius_key_max), and compact
//     the road indices that pass into LDS with one ballot per pass.  The scan STOPS at the pass that brings the count to K.
//   * the rows are stored by the wave that selected them: the 32-byte records of the selected roads are gathered (index
//     order: neighbouring lanes read neighbouring records), observationOf is computed 64 rows at a time, the rows are laid
//     out in LDS and leave as whole 16-byte streaming (nt) stores of one contiguous block.
//   * an agent farther than the radius from the bounding box of its world's roads (a finished agent parked at the padding
//     position, reference src/sim.cpp:333-343) scans nothing: its rows are K padding rows.
//   * an agent whose pose bits are the ones its rows were last written for is not touched at all (pose_stamp, engine.hpp):
//     the rows are a function of (pose, the world's roads, the radius), and the stamp dies with the world's roads
//     (rebuild_worlds).  Parked cars (`Static` under the reference's default isStaticAgentControlled = false, reference
//     src/level_gen.cpp:102-113, src/sim.cpp:327-331) and finished agents are most of a Waymo scene.
//   * agents are dealt to the XCDs by world (engine.cpp lin_list: workgroup b runs on XCD b % 8, and the list puts every agent
//     of a world at the same b % 8, in consecutive workgroups), so that a world's road arrays are fetched into one L2, not
//     eight.  Step passes take a second list that leaves out the agents that never move (`Static`).
//
// With DevSim::pack set (gd_attach_packed) the wave also -- or only: pack_only -- writes the agent's 200 x 13 normalised road
// columns of the packed observation (pack_cols.hpp), so that a learner that reads packed_observations() pays no second pass.
//
// Algorithmic bytes per world and pass: 8 B per road of the longest prefix any of its agents visits (HBM delivers a world's roads
// once, the L2 serves its agents) + 16 B of pose + 7,200 B of rows per agent that is rewritten (SURVEY.md 8d's contract counts
// 36 B for every road of the world instead).  Bound: HBM writes -- and of those, streaming stores are the best form by far
// (NOTEBOOK.md: plain stores 174 us against 109).
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "gd_math.hpp"
#include "map_rows.hpp"
#include "pack_cols.hpp"

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr int LST = (GD_MAX_ROAD_ENTITIES + GD_LIN_BLK - 1) / GD_LIN_BLK + 15;  // blocks of the largest world
#ifndef GD_LIN_CU
#define GD_LIN_CU 4
#endif
#ifndef GD_LIN_PB
#define GD_LIN_PB 2
#endif
#ifndef GD_LIN_GH
#define GD_LIN_GH 2
#endif
#ifndef GD_LIN_WPE
#define GD_LIN_WPE 5
#endif
#ifndef GD_LIN_ABL
#define GD_LIN_ABL 0  // timing-only builds (results wrong): 1 = no cull / scan (the first K roads), 2 = no row arithmetic, 3 = no stores
#endif
constexpr int CU = GD_LIN_CU;  // cull: batches of 64 blocks requested together
constexpr int PB = GD_LIN_PB;  // scan: passes (of 64 roads) requested together
static_assert(64 % GD_LIN_BLK == 0, "whole blocks per pass");

template <int A_T, bool PACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GD_LIN_WPE))) void k_map_obs_linear(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int per = 4 * d.lin_apw;
    const int32_t *entries = d.lin_list + (size_t)blockIdx.x * per;  // (world << 8 | agent) or -1: engine.cpp lin_list
    const float kmax = d.radius_key_max;

    __shared__ unsigned short s_sel[4][K + 56];                          // the selected road indices of the wave's agent, index order
    __shared__ unsigned short s_lst[4][LST];                             // the blocks of the world's roads that may hold a road in reach
    __shared__ __attribute__((aligned(16))) float s_stage[4][64 * 13];   // 64 rows on their way out (9 raw or 13 packed columns)
    unsigned short *sel = s_sel[wave];
    unsigned short *lst = s_lst[wave];
    float *stage = s_stage[wave];
    const float reach = d.p.observationRadius * 1.001f + 0.01f;

    // what an agent's turn needs, requested one agent ahead: its pose and stamp, and its world's road ranges and road box
    struct Work { int e, r0, r1, b0; float4 bb; float ex, ey, qw, qz; uint4 st; };
    auto load_work = [&](int k) -> Work {
        Work wk{-1, 0, 0, 0, make_float4(0.f, 0.f, 0.f, 0.f), 0.f, 0.f, 1.f, 0.f, make_uint4(0u, 0u, 0u, 0u)};
        if (k < per) wk.e = entries[k];
        if (wk.e >= 0) {
            const int w = wk.e >> 8;
            const size_t i = (size_t)w * A_T + (wk.e & 255);
            wk.r0 = d.road_off[w]; wk.r1 = d.road_off[w + 1]; wk.b0 = d.blk_off[w]; wk.bb = d.road_bbox[w];
            wk.ex = d.px[i]; wk.ey = d.py[i]; wk.qw = d.qw[i]; wk.qz = d.qz[i]; wk.st = d.pose_stamp[i];
        }
        return wk;
    };
    int skipped = 0;
    Work nx = load_work(wave);
    for (int k = wave; k < per; k += 4) {  // wave-uniform
        const Work p = nx;
        nx = load_work(k + 4);
        if (p.e < 0) continue;  // filler entry (the XCD classes hold different numbers of agents)
        const int w = p.e >> 8;
        const size_t i = (size_t)w * A_T + (p.e & 255);
        const int r0 = p.r0, R = p.r1 - p.r0;
        const float2 *rxy = d.road_xy + r0;
        const int NB = (R + GD_LIN_BLK - 1) / GD_LIN_BLK;
        const float4 *blk = d.road_blk + p.b0;
        const float4 bb = p.bb;  // out of reach of every road: farther than the radius (plus rounding) from the box around the world's roads
        const float ex = p.ex, ey = p.ey;
        const float iw = p.qw, iz = -p.qz;  // the INVERSE rotation
        // rows already written for exactly this pose (and these roads: the stamp is cleared whenever the world is rebuilt)
        const bool same = d.pose_skip != 0 && p.st.x != 0xffffffffu && p.st.x == __float_as_uint(ex) && p.st.y == __float_as_uint(ey) &&
                          p.st.z == __float_as_uint(p.qw) && p.st.w == __float_as_uint(p.qz);
        if (same) {
            skipped++;
            continue;
        }
        int count = 0;
        const float dxo = fmaxf(fmaxf(bb.x - ex, ex - bb.z), 0.f), dyo = fmaxf(fmaxf(bb.y - ey, ey - bb.w), 0.f);
        const bool in_reach = R > 0 && !(dxo * dxo + dyo * dyo > reach * reach);  // (NaN poses scan, like the reference's `dist > radius`)
        if (GD_LIN_ABL == 1) {
            for (int j = lane; j < K; j += 64) sel[j] = (unsigned short)min(j, R - 1);
            count = min(K, R);
        } else if (in_reach) {
            // CULL: lane q takes block b0 + q of the world's road blocks (GD_LIN_BLK consecutive roads inside a circle); a block
            // farther from the agent than its circle's radius + the observation radius holds no road in reach.  The surviving
            // blocks are listed in ascending order (one ballot per 64 blocks = 1024 roads).
            int nsv = 0;
#pragma clang loop unroll(disable)
            for (int b0 = 0; b0 < NB; b0 += 64 * CU) {
                float4 c[CU];
#pragma unroll
                for (int u = 0; u < CU; u++) c[u] = blk[min(b0 + u * 64 + lane, NB - 1)];
#pragma unroll
                for (int u = 0; u < CU; u++) {
                    const int b = b0 + u * 64 + lane;
                    const float dx = c[u].x - ex, dy = c[u].y - ey, rr = reach + c[u].z;
                    const bool sv = b < NB && !(dx * dx + dy * dy > rr * rr);
                    const unsigned long long m = __ballot(sv);
                    if (sv) lst[nsv + bits_below_lane(m)] = (unsigned short)b;
                    nsv += __popcll(m);
                }
            }
            wave_sync();
            // SCAN of the surviving blocks, four per pass (a quarter of the wave each), in index order; exact keys
            // (gd_math.hpp ego_dist2), one ballot per pass; PB passes requested together, the next PB before these are keyed.
            // The loop ends with the passes that bring the count to K (reference src/sim.cpp:261: `arrIndex < K`).
            constexpr int BPP = 64 / GD_LIN_BLK;  // blocks per pass
            const int np = (nsv + BPP - 1) / BPP;
            const int q = lane / GD_LIN_BLK, o = lane % GD_LIN_BLK;
            struct Batch { int r[PB]; float2 xy[PB]; };
            auto issue = [&](Batch &bt, int p0) {
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    const int e = (p0 + u) * BPP + q;
                    const int r = e < nsv ? (int)lst[e] * GD_LIN_BLK + o : R;
                    bt.r[u] = r;
                    bt.xy[u] = rxy[min(r, R - 1)];
                }
            };
            auto take = [&](const Batch &bt) {
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    const float key = ego_dist2(ex, ey, iw, iz, bt.xy[u].x, bt.xy[u].y);
                    const bool in = bt.r[u] < R && !(key > kmax);  // `if (dist > radius) continue;`, reference src/sim.cpp:266-269
                    const unsigned long long b = __ballot(in);
                    const int pos = count + bits_below_lane(b);
                    if (in && pos < K) sel[pos] = (unsigned short)bt.r[u];
                    count += __popcll(b);
                }
            };
            if (np > 0) {
                Batch ba, bb;
                issue(ba, 0);
#pragma clang loop unroll(disable)
                for (int p0 = 0; p0 < np && count < K; p0 += 2 * PB) {
                    issue(bb, p0 + PB);
                    take(ba);
                    if (p0 + PB >= np || count >= K) break;
                    issue(ba, p0 + 2 * PB);
                    take(bb);
                }
            }
            count = min(count, K);
        }
        wave_sync();
        // ---- the agent's K rows ----
        constexpr int NP = (K + 63) / 64;
        constexpr int PACK_ROAD0 = 6 + (A_T - 1) * 6, PACK_D = PACK_ROAD0 + K * 13;  // the packed row: ego | partners | road points
        static_assert(K % 4 == 0 && PACK_ROAD0 % 4 == 0 && PACK_D % 4 == 0 && (64 * 13) % 4 == 0 && ((K % 64) * 13) % 4 == 0, "whole 16-byte pieces");
        float *rows_out = d.agent_map + i * (size_t)(K * 9);
        typedef float f4 __attribute__((ext_vector_type(4)));
        constexpr int GH = GD_LIN_GH < NP ? GD_LIN_GH : NP;  // blocks of 64 rows whose gathers are requested together
#pragma unroll
        for (int h = 0; h < NP; h += GH) {
            float4 q0[GH], q1[GH];
#pragma unroll
            for (int g = 0; g < GH; g++) {
                const int sl = (h + g) * 64 + lane;
                const int r = r0 + (sl < count ? (int)sel[sl] : 0);
                q0[g] = d.road_rec[(size_t)r * 2];
                q1[g] = d.road_rec[(size_t)r * 2 + 1];
            }
#pragma unroll
            for (int g = 0; g < GH; g++) {
                const int pz = h + g;
                if (pz >= NP) break;
                const int sl = pz * 64 + lane;
                float raw[9];
                if (GD_LIN_ABL == 2) {
                    raw[0] = q0[g].x; raw[1] = q0[g].y; raw[2] = q0[g].z; raw[3] = q0[g].w; raw[4] = q1[g].x; raw[5] = q1[g].y; raw[6] = q1[g].z; raw[7] = q1[g].w; raw[8] = ex;
                } else {
                    road_row(raw, sl < count, false, ex, ey, iw, -iz, q0[g], q1[g]);
                }
                const int nrows = min(64, K - pz * 64);
                if (!PACK || !d.pack_only) {
#pragma unroll
                    for (int c = 0; c < 9; c++) stage[lane * 9 + c] = raw[c];
                    wave_sync();
                    for (int q = lane; q < nrows * 9 / 4; q += 64) {
                        if (GD_LIN_ABL == 3) {
                            if (stage[q * 4] == 12345.678f) rows_out[0] = 1.f;  // (keeps the row arithmetic alive)
                            continue;
                        }
                        __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(rows_out + pz * 576 + q * 4));
                    }
                    wave_sync();
                }
                if (PACK) {  // (gd_attach_packed) the same rows in the packed observation's 13 normalised columns (pack_cols.hpp)
                    pack_road_row(raw, stage + lane * 13);
                    wave_sync();
                    float *pout = d.pack + i * (size_t)PACK_D + PACK_ROAD0 + pz * (64 * 13);
                    for (int q = lane; q < nrows * 13 / 4; q += 64)
                        __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(pout + q * 4));
                    wave_sync();
                }
            }
        }
        if (lane == 0)
            d.pose_stamp[i] = make_uint4(__float_as_uint(ex), __float_as_uint(ey), __float_as_uint(p.qw), __float_as_uint(p.qz));
    }
    if (lane == 0 && skipped) atomicAdd(d.stat_skipped + ((blockIdx.x * 4u + wave) & (GD_SKIP_SLOTS - 1)), (unsigned long long)skipped);
}

}  // namespace

void launch_map_obs_linear(const DevSim &d0, hipStream_t st, bool move) {
    // A step pass takes the list without the agents that never move (`Static`: their rows were written by the last reset pass,
    // which follows every rebuild, and their stamps would only confirm it); every other pass takes every live agent.
    DevSim d = d0;
    if (move && d.pose_skip != 0 && d.lin_dyn_off == 0) {
        d.lin_list = d.lin_list_dyn;
        d.lin_blocks = d.lin_blocks_dyn;
    }
    if (d.lin_blocks == 0) return;
    const dim3 grid(d.lin_blocks);
    if (d.pack != nullptr) {
        if (d.A == 64) hipLaunchKernelGGL((k_map_obs_linear<64, true>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_map_obs_linear<128, true>), grid, dim3(256), 0, st, d);
        return;
    }
    if (d.A == 64) hipLaunchKernelGGL((k_map_obs_linear<64, false>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_map_obs_linear<128, false>), grid, dim3(256), 0, st, d);
}

}  // namespace gd
