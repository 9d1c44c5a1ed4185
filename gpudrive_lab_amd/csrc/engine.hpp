// Device-side view of one simulator instance, shared between the host engine and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gpudrive_amd.h"

#ifndef GD_MAP_OBS_AW
#define GD_MAP_OBS_AW 32  // agents per workgroup (= wave) of the reference-order road kernel: 16, 32 or 64
#endif

// Rank replay of the reference-order road selection (map_obs_rank.hip): candidates per agent, checkpoints per agent,
// dwords of the heap array handed from the replay to the finishing kernel (K / 2 + 1 pairs, padded to 16 bytes)
#define GD_RANK_CAP 1280
#define GD_RANK_NCP 40
#define GD_RANK_HEAP_DW 104
#define GD_RANK_CAP_LONG 2560   // candidates of an agent that the long-list instantiation of the ranking takes (map_obs_rank.hip)
#define GD_RANK_KT_LONG 164     // its key-table row (CAP_LONG / 16 entries + the last, padded to 16 bytes)
#define GD_RH_LONG 537          // rk_hist: agents on the long list in this selection
#define GD_RANK_NCH 320  // candidate words per agent: 32 roads each, kMaxRoadEntityCount = 10,000
#define GD_RANK_SPL 256  // sorted slots whose road index k_knn_rank hands on: the K elements the heap ends with have the K smallest
                         // keys, i.e. fewer than K candidates below them and at most 31 equal ones before them: slot < K + 31
#define GD_RANK_KT 84    // floats per agent of the key table: the candidates' key at every 16th sorted slot (CAP / 16 entries), the
                         // largest key behind them; padded to 16 bytes
// rk_hist[GD_RANK_AUDIT]: bounds audit of the rank path (product build too).  Every index that the rank kernels derive from
// data another kernel wrote (a rank turned into a table slot, a ticket into a place of the replay order, a list cursor) is
// checked against its array before it is used; a violation is counted here and the index clamped into the array, so that
// a broken invariant shows up as a failed test (gd_stat 21 must read 0), never as an access outside an allocation.
#define GD_RANK_AUDIT 536
#define GD_LIN_BLK 16       // roads per block of the linear scan's cull (map_obs_linear.hip)
#define GD_SKIP_SLOTS 4096  // counters of DevSim::stat_skipped

// Phase switches for timing experiments exist only in diagnostic builds (-DGD_DIAG); in the product library the
// condition is the constant false and the compiler drops the code.
#ifdef GD_DIAG
#define GD_DIAG_IS(field, value) ((field) == (value))
#else
#define GD_DIAG_IS(field, value) false
#endif

namespace gd {

// the first five are the timed kernels of gd_kernel_timing_read
enum { KERNEL_STATE = 0, KERNEL_MAP_OBS = 1, KERNEL_LIDAR = 2, KERNEL_BEV = 3, KERNEL_PARTNER = 4, KERNEL_RESET = 5, KERNEL_PADDING = 6, KERNEL_TIMED = 5 };

// Per-world broadphase grid header (see HostWorld in scene.hpp).
struct GridHdr {
    float ox, oy, inv_cell;
    int nx, ny;
    int cell_base;  // first entry of this world in cell_off
    int item_base;  // first entry of this world in cell_items
    int pad;
};

// Passed to kernels by value.  HBM layout:
//   exported tensors : the reference's AoS layouts (API contract, src/mgr.cpp:656-902)
//   agent state      : world-major SoA, one [W][A] plane per field
//   roads            : CSR over worlds; (x,y) float2 stream for the scan, 2 x float4 per road
//                      {qw,qz,d0,d1 | d2,type,id,mapType} gathered only for selected rows
//   collidable boxes : CSR over worlds; 5 x float4 per box {cx,cy,radius,type | 14-float OBB}
struct DevSim {
    int W, A;
    int knn_order;    // GD_KNN_*
    int set_fused_rows;  // set-order mode: the selection kernel writes the rows itself (several generations of workgroups, ragged batches)
    int32_t *live_list;  // [W * A] world * A + agent of every live agent, agent-major (rebuilt with the worlds)
    int live_count;
    int32_t *set_groups;  // set-order kernel: (world << 8 | group) of every group of 4 * set_apw agent slots that holds a live agent
    int set_group_count;
    int set_apw;         // set-order mode: agents per wave (a workgroup of 4 waves takes 4 * set_apw consecutive agents of a world)
    float lidar_half_angle;  // 0 -> pi/3 (reference consts::lidarAngle)
    float radius_key_max;    // largest fp32 k with sqrtf(k) <= observationRadius (radiusFilter on squared keys)
    gd_params p;
    // exported
    float *action, *reward, *self_obs, *abs_obs, *partner, *agent_map, *map_obs, *lidar, *bev, *traj, *means;
    int32_t *done, *info, *shape, *controlled, *resp_export, *metadata, *deleted, *map_name, *scenario_id;
    uint32_t *steps;
    // internal agent state [W][A]
    float *px, *py, *pz, *qw, *qz, *vx, *vy, *vz;
    int32_t *collided;
    float *len, *wid, *hgt, *sc0, *sc1, *goal_x, *goal_y;
    int32_t *etype, *agent_id, *resp;
    // road selection scratch: what k_map_obs / k_map_obs_set hand to k_map_rows
    uint16_t *sel_idx;    // [W][A][K] road index (within the world) of every selected slot, in output order
    float4 *sel_hdr;      // [W][A][2] per agent for k_map_rows: (x, y, qw, qz) and, as int bits, (selected rows, first road of the
                          // world, permuted, 0); the count is -1 where this launch selected nothing (padding agents)
    uint8_t *sel_slot;    // [W][A][K] permuted != 0 only: sel_idx lists the selected roads in ASCENDING road index (what the
                          // gathers like) and entry q belongs in output row sel_slot[q]; permuted == 0: entry q is row q
    // per world flags
    int32_t *reset_flags, *rebuilt_flags;
    int32_t *any_reset;    // one int: k_episode_step raised at least one reset flag in this step
    int gate_any;          // reset pass launched without knowing on the host whether anything was flagged:
                           // every workgroup returns at once unless *any_reset is set
    // roads
    const int32_t *road_off;  // [W+1]
    int32_t *wave_order;   // [W * A / GD_MAP_OBS_AW] workgroups of the reference-order road kernel (world * A/AW + part) in launch order: longest first,
                           // by road count at load time, then by the cycles each one took in the previous launch (k_order_waves)
    uint32_t *wave_cost;   // [W * A / GD_MAP_OBS_AW] cycles of each of those workgroups in the last launch that ran them
    const float2 *road_xy;
    const float4 *road_aux;
    const float4 *road_rec;  // [roads][2] what a row of agent_roadmap_tensor needs, in 32 bytes: (x, y, qw, qz), (d0, d1, id, bits: type | (mapType + 1) << 8)
    const int32_t *box_off;   // [W+1]
    const float4 *boxes;
    const GridHdr *grid;        // [W]
    const int32_t *cell_off;    // per world nx*ny+1 entries, local offsets
    const int32_t *cell_items;  // local box indices
    const float4 *cell_hdr;     // (cx, cy, bounding radius, bits: type | local box index << 8) of those boxes, in the same cell order
    // set-order road selection: per-world uniform grid over ALL roads (a road sits in the cell of its (x, y)), CSR of
    // local road indices ascending within a cell, and per agent where and how far the previous selection reached
    const GridHdr *rgrid;          // [W]
    const int32_t *rcell_off;      // per world nx*ny+1 entries, local offsets
    const uint16_t *rcell_items;   // local road indices
    const float2 *rcell_xy;        // the (x, y) of those roads, in the same (cell-sorted) order: one coalesced stream per grid row
    const uint16_t *rcell_pos;     // [roads] the inverse: where road r of its world sits in the world's cell-sorted order (set order IS that order)
    float4 *knn_prev;              // [W][A] {x, y, K-th key of the previous selection or +inf, 0}
    // linear road selection (map_obs_linear.hip): the live agents as (world << 8 | agent) or -1 (filler), 4 * lin_apw entries per
    // workgroup, ordered so that every workgroup that holds agents of a world has the same index modulo 8 (= runs on the same
    // XCD); lin_list_dyn: the same without the agents whose response type is Static (they never move: reference
    // src/sim.cpp:327-331), what a step pass takes; lin_dyn_off: this pass takes the full list all the same
    const int32_t *lin_list, *lin_list_dyn;
    int lin_blocks, lin_blocks_dyn;
    int lin_dyn_off;
    // ... and skip whole blocks of GD_LIN_BLK consecutive roads that cannot hold a road in reach: per block the circle around its
    // roads' points, (cx, cy, radius, 0); the blocks of world w start at blk_off[w] (roads follow their polylines in index order,
    // so a block is a short piece of one polyline as a rule)
    const float4 *road_blk;
    const int32_t *blk_off;        // [W + 1]
    int lin_apw;
    int lin_on;                    // 0: GPUDRIVE_LINEAR_LEGACY=1 -- the linear scan inside k_map_obs / k_map_obs_set (rounds 1-4), for A/B runs
    // rows that cannot have changed are not rewritten: the pose bits (x, y, qw, qz) an agent's road rows were last written for;
    // x = 0xffffffff: none (set for every agent whenever worlds are rebuilt -- roads or agent slots may have changed)
    uint4 *pose_stamp;             // [W][A]
    int pose_skip;                 // 0: GPUDRIVE_NO_POSE_SKIP=1 -- every live agent's rows are rewritten on every step
    // BEV rasters (bev_lidar.hip) are rewritten only for agents whose picture can have changed: the agent moved, or an agent that
    // moved is (or was) within the radius of it.  k_world_step decides (bev_dirty, 1 = rewrite; every live agent on reset passes,
    // with GPUDRIVE_NO_POSE_SKIP=1 and when bev_all_dirty is set), k_bev_list compacts live_list to the dirty agents, k_bev's
    // workgroups stride over that list.
    int32_t *bev_dirty;   // [W][A]
    int32_t *bev_list;    // [W * A] (world * A + agent) of the agents to rasterise in this pass
    int32_t *bev_count;   // [1]
    int bev_all_dirty;    // (also for the LiDAR flags below)
    // The same for the LiDAR returns (k_lidar): an agent's rays see the agents within 200 m (+ their bounding radius), the static
    // roads, and start at its own pose with the head angle of its action row; lidar_head = the head angle its returns were last
    // traced with (0xffffffff: none).  lidar_dirty is k_world_step's verdict for this step.
    int32_t *lidar_dirty;  // [W][A]
    float *lidar_head;     // [W][A]
    // packed observation written where the raw rows are produced (gd_attach_packed): [W][A][6 + (A-1)*6 + K*13], or null.
    // pack_only: the raw partner and road tensors of live agents are NOT written any more (a learner that only reads the
    // packed tensor; the padding agents' rows, written when the worlds are built, stay valid)
    float *pack;
    int pack_only;
    unsigned long long *stat_skipped;  // [GD_SKIP_SLOTS] agents whose rows were left in place since the counters were last read (gd_stat 30
                                       // sums them): every wave adds to the slot of its own index -- one counter for all of them
                                       // serialises ten thousand atomics at one memory channel (measured: 150 us per launch)
    // reference-order road selection, rank replay (map_obs_rank.hip); rk_on = 0: k_map_obs alone selects
    int split_partner;  // the partner rows are written by k_partner_rows (second stream) instead of k_world_step
    int step_dbg;  // -DGD_DIAG builds only (tools/build_expt.sh): k_world_step skips 1 = the road-box loop, 2 = the agent-agent
                   // loop, 3 = the partner rows; timing only, results wrong.  The product build compiles the switches out (GD_DIAG_IS).
    int rk_on;
    int rk_max_roads;  // ... and worlds with more: their agents' candidates (200 ln(R / 200) inserts and more) overflow the buffer too often
    int rk_min_roads;  // worlds with fewer roads are selected by k_map_obs (the rank path's fixed costs do not pay there)
    int rk_dbg;  // -DGD_DIAG builds only: k_knn_rank stops after phase n (timing only; results are wrong)
    uint16_t *rk_E;        // [W][A][CAP] rank of every candidate, candidate (= road) order
    // agents with more candidates than the standard ranking holds (the long list: k_knn_rank<A, CAP_LONG>), rk_nlong slots
    int rk_nlong;
    int32_t *rk_longlist;  // [rk_nlong] agent slot of every entry (count: rk_hist[GD_RH_LONG])
    int32_t *rk_longslot;  // [W][A] the entry an agent holds (valid while bit 29 of its rk_n is set)
    uint16_t *rk_E_long;   // [rk_nlong][CAP_LONG] its ranks
    float *rk_kt_long;     // [rk_nlong][GD_RANK_KT_LONG] its key table
    uint16_t *rk_spc;      // [W][A][GD_RANK_SPL] sorted slot -> road index, the first GD_RANK_SPL slots
    float *rk_kt;          // [W][A][GD_RANK_KT] key at sorted slot 16 j (16 j < n), then the largest key at j = ceil(n / 16)
    uint32_t *rk_heap;     // [W][A][GD_RANK_HEAP_DW] the replayed heap array as rank pairs
    uint16_t *rk_cpe;      // [W][A][NCP] rank on top of the heap at every checkpoint of this selection
    int32_t *rk_hist;      // [544] replay order: 256 bin counts, 256 bin starts, the number of agents on the rank path, selections so far; [514..526] counters of developer builds; [528..535] entries of rk_list; [536] bounds audit (GD_RANK_AUDIT)
    int32_t *rk_ticket;    // [W][A] bin << 20 | place inside the bin (bit 30: fell back after taking it); -1 = not on the rank path this step; < -1: why
    int32_t *rk_order;     // [W][A] agents on the rank path, most candidates first
    int32_t *rk_list;      // [8][W][A] agents on the rank path, one list per XCD that scanned them (counts: rk_hist[528..535])
    uint32_t *rk_words;    // [W][A][NCH] candidate bits of 32 roads, one row per agent (k_knn_scan -> k_knn_rank)
    float *rk_tl;          // [W][A] the last K-th key of the checkpoint set in use (scales the ranking buckets)
    const float4 *road_bbox;  // [W] (min x, min y, max x, max y) over the world's roads
    const float *road_rbmax;  // [W] the largest bounding radius (half diagonal) of a road of the world
    int32_t *rk_n;         // [W][A] candidates | in-radius candidates << 16; 0 = not on the rank path this step; 1 << 30 = too far from every road
    int32_t *rk_fallback;  // [W * A / 32] group of 32 agent slots must be selected by k_map_obs this step
    int32_t *rk_streak;    // [W * A / 32] consecutive selections in which the group needed the fallback: from 3 on the group
                           // bypasses the rank kernels (whose work would be wasted) and retries every 64th selection
    // checkpoints of the previous selection of every agent: the K-th key (cp_T) that the heap held when the scan reached road
    // cp_road; cp_hdr = {x, y, number of checkpoints as int bits (0: none usable), 0} where they were recorded
    // two sets: [0] the previous selection, [1] the selection at the start of the episode (where a reset puts the agent back)
    uint16_t *cp_road;     // [2][W][A][NCP]
    float *cp_T;           // [2][W][A][NCP]
    float4 *cp_hdr;        // [2][W][A]
};

void launch_kernel(const DevSim &d, hipStream_t st, int which, bool move);
void launch_map_obs(const DevSim &d, hipStream_t st, bool move);  // map_obs.hip
void launch_map_obs_rank(const DevSim &d, hipStream_t st);  // map_obs_rank.hip
void launch_map_obs_linear(const DevSim &d, hipStream_t st, bool move);  // map_obs_linear.hip
void launch_bev(const DevSim &d, hipStream_t st);      // bev_lidar.hip
void launch_lidar(const DevSim &d, hipStream_t st);    // bev_lidar.hip
void launch_pack_obs(const DevSim &d, hipStream_t st, float *out);  // pack_obs.hip
void launch_expert_actions(const DevSim &d, hipStream_t st, float *actions, float *pos, float *vel, float *yaw, int *valid);
void launch_set_log_actions(const DevSim &d, hipStream_t st, int t);
void launch_episode_step(const DevSim &d, hipStream_t st, const gd_episode_config &c, const gd_episode_buffers &b);  // episode.hip

}  // namespace gd
