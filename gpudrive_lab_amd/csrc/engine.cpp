// Host engine + C ABI (include/gpudrive_amd.h).  Citations are relative to the reference checkout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <atomic>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"
#include "gd_math.hpp"
#include "scene.hpp"
#ifdef GD_CLOCKS
namespace gd { void set_clocks_read(unsigned long long *out); void step_clocks_read(unsigned long long *out); }
#endif
#ifndef GD_GRID_MIN_CELL
#define GD_GRID_MIN_CELL 16.f
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(e_));               \
    } while (0)

struct TensorSpec {
    int dtype;
    int ndim;
    int64_t dims[5];
};

TensorSpec tensor_spec(int id, int64_t W, int64_t A) {
    switch (id) {
    case GD_T_ACTION: return {GD_DTYPE_F32, 3, {W, A, 10}};
    case GD_T_REWARD: return {GD_DTYPE_F32, 3, {W, A, 1}};
    case GD_T_DONE: return {GD_DTYPE_I32, 3, {W, A, 1}};
    case GD_T_INFO: return {GD_DTYPE_I32, 3, {W, A, 5}};
    case GD_T_SELF_OBS: return {GD_DTYPE_F32, 3, {W, A, 8}};
    case GD_T_ABS_OBS: return {GD_DTYPE_F32, 3, {W, A, 14}};
    case GD_T_PARTNER_OBS: return {GD_DTYPE_F32, 4, {W, A, A - 1, 9}};
    case GD_T_AGENT_MAP_OBS: return {GD_DTYPE_F32, 4, {W, A, GD_MAP_OBS_K, 9}};
    case GD_T_MAP_OBS: return {GD_DTYPE_F32, 3, {W, GD_MAX_ROAD_ENTITIES, 9}};
    case GD_T_LIDAR: return {GD_DTYPE_F32, 5, {W, A, 3, GD_NUM_LIDAR_SAMPLES, 4}};
    case GD_T_BEV: return {GD_DTYPE_F32, 5, {W, A, GD_BEV_RES, GD_BEV_RES, 1}};
    case GD_T_STEPS_REMAINING: return {GD_DTYPE_I32, 3, {W, A, 1}};
    case GD_T_SHAPE: return {GD_DTYPE_I32, 2, {W, 2}};
    case GD_T_CONTROLLED_STATE: return {GD_DTYPE_I32, 3, {W, A, 1}};
    case GD_T_RESPONSE_TYPE: return {GD_DTYPE_I32, 3, {W, A, 1}};
    case GD_T_EXPERT_TRAJECTORY: return {GD_DTYPE_F32, 3, {W, A, GD_TRAJECTORY_FLOATS}};
    case GD_T_WORLD_MEANS: return {GD_DTYPE_F32, 2, {W, 3}};
    case GD_T_METADATA: return {GD_DTYPE_I32, 3, {W, A, 4}};
    case GD_T_DELETED_AGENTS: return {GD_DTYPE_I32, 2, {W, A}};
    case GD_T_MAP_NAME: return {GD_DTYPE_I32, 2, {W, 32}};
    case GD_T_SCENARIO_ID: return {GD_DTYPE_I32, 2, {W, 32}};
    }
    return {-1, 0, {0}};
}

int64_t spec_bytes(const TensorSpec &s) {
    int64_t n = 4;
    for (int i = 0; i < s.ndim; i++) n *= s.dims[i];
    return n;
}

struct EventPair {
    hipEvent_t start, stop;
};

}  // namespace

struct gd_sim {
    gd_config cfg{};
    gd_params params{};
    int W = 0, A = 0;
    hipStream_t stream = nullptr;
    gd::DevSim d{};
    void *exported[GD_T_COUNT] = {};
    bool owned[GD_T_COUNT] = {};
    std::vector<void *> internal;
    std::vector<std::string> scenes;
    std::vector<int32_t> deleted;  // host mirror [W][A]
    // per-world host road data (kept to repack the CSR on set_maps / deleteAgents)
    std::vector<std::vector<float>> w_xy, w_aux;
    std::vector<int> w_agents;  // live agents per world (shape[w][0])
    int cu_count = 256;
    std::vector<std::vector<gd::RoadBox>> w_boxes;
    std::vector<gd::GridHdr> w_grid;
    std::vector<std::vector<int32_t>> w_cell_off, w_cell_items;
    std::vector<gd::GridHdr> w_rgrid;                    // grid over all roads (set-order selection)
    std::vector<std::vector<int32_t>> w_rcell_off;
    std::vector<std::vector<uint16_t>> w_rcell_items;
    size_t rcell_cap = 0, ritem_cap = 0;
    void *d_rcell_off = nullptr, *d_rcell_items = nullptr, *d_rcell_xy = nullptr, *d_rcell_pos = nullptr;
    size_t cell_cap = 0, item_cap = 0;
    void *d_cell_off = nullptr, *d_cell_items = nullptr, *d_cell_hdr = nullptr;
    size_t road_cap = 0, box_cap = 0, blk_cap = 0;
    void *d_road_blk = nullptr;
    void *d_road_xy = nullptr, *d_road_aux = nullptr, *d_road_rec = nullptr, *d_boxes = nullptr;
    // pinned flag staging ring
    static constexpr int kRing = 8;
    int32_t *h_flags[kRing] = {};
    hipEvent_t flag_ev[kRing] = {};
    int ring_pos = 0;
    // kernel timing
    // A fixed ring of event pairs per kernel, created when timing is switched on: a launch re-records the oldest pair
    // after its elapsed time has been read.  (Round 2 kept one pair per launch alive until the read-out; the runtime's
    // signal pool then ran dry in the middle of a timed stretch and one hipLaunchKernel blocked the host for 14 ms --
    // tools/trace_gap.sh -- so that a 20-step wall clock was far above the sum of its kernels.)
    static constexpr size_t kEvRing = 32;
    // second stream for the partner rows (k_partner_rows beside the road kernels): forked and joined with events, also
    // inside the captured step graph
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    size_t lin_cap = 0;        // entries of d.lin_list / d.lin_list_dyn
    int32_t *d_lin_list = nullptr, *d_lin_list_dyn = nullptr;
    std::vector<std::vector<int32_t>> w_resp;  // response type of every agent slot (who can move at all)
    bool full_pass_next = false;  // the next step's road pass takes every live agent (state was written from outside)
    int64_t lin_static_agents = 0;   // live agents that are not on the step passes' list (response type Static)
    int64_t host_skipped = 0;        // ... counted as left in place once per step pass (gd_stat 30 adds the device's counters)
    bool rk_possible = false;  // reference order, k-NN, not switched off: a batch may take the rank replay
    bool rk_alloc = false;     // its buffers exist

    // 5.7 KB per agent slot for ALL W * A slots (0.37 GB at 1024 x 64, 1.5 GB at 4096 x 64), allocated when a batch first takes
    // the rank replay and kept for the life of the simulator.  The path is a scheduling choice, never a result: when the
    // device cannot give the memory, what was allocated is returned, the batch stays on k_map_obs (same rows) and the
    // rank replay is not tried again for this simulator.
    bool ensure_rank_buffers() {
        if (rk_alloc) return true;
        const size_t WA = static_cast<size_t>(W) * A;
        const size_t first = internal.size();
        try {
            // developer switch for the test of the path below: the device "runs out of memory" in the middle of the allocations
            if (std::getenv("GPUDRIVE_RANK_ALLOC_FAIL") != nullptr) {
                d.rk_E = alloc_internal<uint16_t>(WA * GD_RANK_CAP + 64);
                d.rk_spc = alloc_internal<uint16_t>(WA * GD_RANK_SPL);
                throw HipError("GPUDRIVE_RANK_ALLOC_FAIL: simulated allocation failure of the rank replay's buffers");
            }
            d.rk_E = alloc_internal<uint16_t>(WA * GD_RANK_CAP + 64);  // the replay prefetches up to 24 entries past a row
            d.rk_spc = alloc_internal<uint16_t>(WA * GD_RANK_SPL);
            d.rk_kt = alloc_internal<float>(WA * GD_RANK_KT);
            d.rk_heap = alloc_internal<uint32_t>(WA * GD_RANK_HEAP_DW);
            d.rk_cpe = alloc_internal<uint16_t>(WA * GD_RANK_NCP);
            d.rk_n = alloc_internal<int32_t>(WA);
            d.rk_fallback = alloc_internal<int32_t>(WA / 32);
            d.rk_streak = alloc_internal<int32_t>(WA / 32);
            d.cp_road = alloc_internal<uint16_t>(2 * WA * GD_RANK_NCP);
            d.cp_T = alloc_internal<float>(2 * WA * GD_RANK_NCP);
            d.cp_hdr = alloc_internal<float4>(2 * WA);
            d.rk_words = alloc_internal<uint32_t>(WA * GD_RANK_NCH);
            d.rk_tl = alloc_internal<float>(WA);
            d.rk_hist = alloc_internal<int32_t>(544);
            d.rk_ticket = alloc_internal<int32_t>(WA);
            d.rk_order = alloc_internal<int32_t>(WA);
            d.rk_list = alloc_internal<int32_t>(8 * WA);
            // the long list: room for a quarter of the agent slots (an agent beyond that takes the fallback)
            d.rk_nlong = static_cast<int>(std::max<size_t>(WA / 4, 64));
            d.rk_longlist = alloc_internal<int32_t>(d.rk_nlong);
            d.rk_longslot = alloc_internal<int32_t>(WA);
            d.rk_E_long = alloc_internal<uint16_t>(static_cast<size_t>(d.rk_nlong) * GD_RANK_CAP_LONG + 64);  // (+ the replay's prefetch)
            d.rk_kt_long = alloc_internal<float>(static_cast<size_t>(d.rk_nlong) * GD_RANK_KT_LONG);
        } catch (const HipError &) {
            (void)hipGetLastError();
            for (size_t k = first; k < internal.size(); k++) (void)hipFree(internal[k]);
            internal.resize(first);
            rk_possible = false;
            d.rk_on = 0;
            d.rk_nlong = 0;
            // nothing may point at what was just returned
            d.rk_E = nullptr; d.rk_spc = nullptr; d.rk_kt = nullptr; d.rk_heap = nullptr; d.rk_cpe = nullptr; d.rk_n = nullptr;
            d.rk_fallback = nullptr; d.rk_streak = nullptr; d.cp_road = nullptr; d.cp_T = nullptr; d.cp_hdr = nullptr;
            d.rk_words = nullptr; d.rk_tl = nullptr; d.rk_hist = nullptr; d.rk_ticket = nullptr; d.rk_order = nullptr;
            d.rk_list = nullptr; d.rk_longlist = nullptr; d.rk_longslot = nullptr; d.rk_E_long = nullptr; d.rk_kt_long = nullptr;
            return false;
        }
        rk_alloc = true;
        return true;
    }
    bool timing = false;
    std::vector<EventPair> ev_pool[gd::KERNEL_TIMED];
    size_t ev_head[gd::KERNEL_TIMED] = {};  // oldest recorded pair
    size_t ev_used[gd::KERNEL_TIMED] = {};  // recorded pairs not read yet
    double ev_ms[gd::KERNEL_TIMED] = {};
    int64_t ev_launches[gd::KERNEL_TIMED] = {};

    ~gd_sim() {
        (void)hipDeviceSynchronize();
        if (step_graph) (void)hipGraphExecDestroy(step_graph);
        if (side) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (int i = 0; i < GD_T_COUNT; i++)
            if (owned[i] && exported[i]) (void)hipFree(exported[i]);
        for (void *p : internal) (void)hipFree(p);
        if (d_road_xy) (void)hipFree(d_road_xy);
        if (d_road_aux) (void)hipFree(d_road_aux);
        if (d_road_rec) (void)hipFree(d_road_rec);
        if (d_boxes) (void)hipFree(d_boxes);
        if (d_road_blk) (void)hipFree(d_road_blk);
        if (d_cell_off) (void)hipFree(d_cell_off);
        if (d_cell_items) (void)hipFree(d_cell_items);
        if (d_cell_hdr) (void)hipFree(d_cell_hdr);
        if (d_rcell_off) (void)hipFree(d_rcell_off);
        if (d_rcell_items) (void)hipFree(d_rcell_items);
        if (d_rcell_xy) (void)hipFree(d_rcell_xy);
        if (d_rcell_pos) (void)hipFree(d_rcell_pos);
        for (int i = 0; i < kRing; i++) {
            if (h_flags[i]) (void)hipHostFree(h_flags[i]);
            if (flag_ev[i]) (void)hipEventDestroy(flag_ev[i]);
        }
        for (auto &pool : ev_pool)
            for (auto &e : pool) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
    }

    template <typename T>
    T *alloc_internal(size_t count) {
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)));
        internal.push_back(p);  // owned from here on: a failing memset must not leak it
        HIP_CHECK(hipMemset(p, 0, std::max<size_t>(count * sizeof(T), 16)));
        return static_cast<T *>(p);
    }

    void collect_timing(int k, size_t keep = 0) {  // read the oldest pairs until `keep` are left
        while (ev_used[k] > keep) {
            const EventPair &e = ev_pool[k][ev_head[k]];
            float ms = 0.f;
            if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
                ev_ms[k] += ms;
                ev_launches[k]++;
            }
            ev_head[k] = (ev_head[k] + 1) % ev_pool[k].size();
            ev_used[k]--;
        }
    }

    void launch(int which, bool move, hipStream_t stream_override = nullptr) {
        hipStream_t stream = stream_override ? stream_override : this->stream;
        const bool timed = timing && which < gd::KERNEL_TIMED && !d.gate_any;  // gated reset passes are mostly empty launches
        EventPair ep{};
        if (timed) {
            // ring full: read the older half (those launches finished long ago; the host stays well ahead of the GPU)
            if (ev_used[which] == ev_pool[which].size()) collect_timing(which, ev_pool[which].size() / 2);
            ep = ev_pool[which][(ev_head[which] + ev_used[which]) % ev_pool[which].size()];
            ev_used[which]++;
            HIP_CHECK(hipEventRecord(ep.start, stream));
        }
        if (which == gd::KERNEL_BEV) gd::launch_bev(d, stream);
        else if (which == gd::KERNEL_LIDAR) gd::launch_lidar(d, stream);
        else gd::launch_kernel(d, stream, which, move);
        if (timed) HIP_CHECK(hipEventRecord(ep.stop, stream));
        HIP_CHECK(hipGetLastError());
    }

    // The Step task graph as one hipGraph: the kernels of a step are captured once on the engine's
    // stream and replayed with a single hipGraphLaunch (the kernel arguments are the DevSim struct by
    // value, so any change of it -- rebuilt worlds, a new stream, timing mode -- drops the graph).
    hipGraphExec_t step_graph = nullptr;
    int64_t stat_graph_steps = 0, stat_plain_steps = 0, stat_captures = 0;
    bool graph_ok = std::getenv("GPUDRIVE_NO_GRAPH") == nullptr;

    void drop_graph() {
        if (step_graph) {
            (void)hipGraphExecDestroy(step_graph);
            step_graph = nullptr;
        }
    }

    // Set-order road kernel: how the agents are dealt to workgroups and which of its two equivalent write-outs runs
    // (map_obs.hip, launch_map_obs).  Neither changes a result.  Measured (road observation, us; tools/set_schedules.sh),
    // agents per wave 1 / 2 / 4 / 16:
    //   1024 full worlds (synthetic)   row kernel 183 / 181 / 180 / 192     fused 181 / 168 / 167 / 167
    //   1024 ragged worlds (Waymo)     row kernel  83 /  80 /  82 /  84     fused  61 /  70 /  67 /  80
    //   4096 ragged worlds             row kernel 411 / 399 / 392 / 392     fused 249 / 246 / 256 / 274
    // Rounds 2 and 3 chose by batch shape (row kernel for full worlds, sixteen agents per wave for thousands of worlds):
    // the selection then took twice the instructions it takes now (map_obs.hip), and rows stored by the selecting waves
    // had little to hide behind.  Now: always fused, two agents per wave -- enough workgroups for every batch, and the
    // second agent's inputs arrive while the first is selected.
    // GPUDRIVE_SET_FUSED_ROWS=0|1 and GPUDRIVE_SET_AGENTS_PER_WAVE=n pin them (the tests run the combinations).
    void choose_set_schedule() {
        d.set_apw = 2;
        d.set_fused_rows = 1;
        if (const char *e = std::getenv("GPUDRIVE_SET_FUSED_ROWS")) d.set_fused_rows = std::atoi(e) != 0 ? 1 : 0;
        if (const char *e = std::getenv("GPUDRIVE_SET_AGENTS_PER_WAVE")) d.set_apw = std::min(32, std::max(1, std::atoi(e)));
    }

    void step() {
        if (full_pass_next) {  // (gd_debug_set_state moved agents behind the engine's back: nobody is left out of this step's road pass / rasters)
            full_pass_next = false;
            d.lin_dyn_off = 1;
            d.bev_all_dirty = 1;
            try {
                run_rest(true);
            } catch (...) {
                d.lin_dyn_off = 0;
                d.bev_all_dirty = 0;
                throw;
            }
            d.lin_dyn_off = 0;
            d.bev_all_dirty = 0;
            stat_plain_steps++;
            return;
        }
        // (the agents a linear step pass does not even visit are agents whose rows are left in place)
        if (params.roadObservationAlgorithm != GD_ROADS_K_NEAREST && d.lin_on && d.pose_skip && !params.disableClassicalObs)
            host_skipped += lin_static_agents;
        if (!graph_ok || timing || stream == nullptr) {  // the legacy null stream cannot be captured
            run_rest(true);
            stat_plain_steps++;
            return;
        }
        if (!step_graph) {
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();
                graph_ok = false;
                run_rest(true);
                return;
            }
            try {
                run_rest(true);
            } catch (...) {
                (void)hipStreamEndCapture(stream, &g);
                if (g) (void)hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(stream, &g));
            stat_captures++;
            const hipError_t e = hipGraphInstantiate(&step_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) {  // fall back to plain launches for good
                step_graph = nullptr;
                graph_ok = false;
                run_rest(true);
                return;
            }
        }
        HIP_CHECK(hipGraphLaunch(step_graph, stream));
        stat_graph_steps++;
    }

    // setupRestOfTasks, src/sim.cpp:785-943
    void run_rest(bool move) {
        launch(gd::KERNEL_STATE, move);
        // GPUDRIVE_SPLIT_PARTNER=1 (off by default): the partner rows (148 MB of stores at 1024 x 64, nothing downstream of
        // them in the step) on the second stream, beside the road kernels; joined before anything else of the caller's
        // stream can follow.  Measured SLOWER in every workload (step, ms: synthetic 1.48 vs 1.42, Waymo tiles 0.68 vs
        // 0.53, set order 0.43 vs 0.40): beside the road kernels the row kernel takes 86-273 us instead of 28 and the step
        // waits for it at the join.
        const bool fork = d.split_partner && !params.disableClassicalObs;
        if (fork) {
            HIP_CHECK(hipEventRecord(ev_fork, stream));
            HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
            launch(gd::KERNEL_PARTNER, move, side);
            HIP_CHECK(hipEventRecord(ev_join, side));
        }
        if (!params.disableClassicalObs) launch(gd::KERNEL_MAP_OBS, move);
        if (!params.disableClassicalObs && d.bev) {  // collectBevObservationsSystem, src/sim.cpp:879-884 (opt-in)
            launch(gd::KERNEL_BEV, move);
        }
        if (params.enableLidar) {  // lidarSystem, src/sim.cpp:895-913
            launch(gd::KERNEL_LIDAR, move);
        }
        if (fork) HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
    }

    void upload_flags(int32_t *dst, const std::vector<int32_t> &flags) {
        const int slot = ring_pos;
        ring_pos = (ring_pos + 1) % kRing;
        HIP_CHECK(hipEventSynchronize(flag_ev[slot]));
        std::memcpy(h_flags[slot], flags.data(), sizeof(int32_t) * W);
        HIP_CHECK(hipMemcpyAsync(dst, h_flags[slot], sizeof(int32_t) * W, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipEventRecord(flag_ev[slot], stream));
    }

    // (Re)build the listed worlds on the host and upload their init-time rows:
    // MapReader::parseAndWriteOut + createPersistentEntities (src/mgr.cpp:527-535,630-647;
    // src/level_gen.cpp:396-465).
    void rebuild_worlds(const std::vector<int> &worlds) {
        HIP_CHECK(hipStreamSynchronize(stream));
        drop_graph();
        std::map<std::string, std::shared_ptr<const gd::SceneMap>> scene_cache;
        std::map<std::string, std::shared_ptr<gd::HostWorld>> world_cache;
        std::vector<int32_t> rebuilt(W, 0);
        // Worlds are staged in runs of consecutive indices (at most kRun) so that every tensor slice of
        // a run goes up in ONE copy: 1024 worlds need ~150 hipMemcpy calls instead of ~20 per world.
        constexpr int kRun = 128;
        std::vector<int> sorted_worlds(worlds);
        std::sort(sorted_worlds.begin(), sorted_worlds.end());
        struct Staging {
            std::vector<float> traj, map_obs, planes[7], means;
            std::vector<int32_t> etype, agent_id, resp, controlled, metadata, deleted, map_name, scenario_id, shape;
        } st;
        auto flush = [&](int w0, int nw) {
            if (nw == 0) return;
            const size_t o = static_cast<size_t>(w0) * A;
            auto up = [&](void *dst, const void *src, size_t bytes) {
                HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            };
            up(d.traj + o * GD_TRAJECTORY_FLOATS, st.traj.data(), st.traj.size() * 4);
            up(d.map_obs + static_cast<size_t>(w0) * GD_MAX_ROAD_ENTITIES * 9, st.map_obs.data(), st.map_obs.size() * 4);
            float *planes[7] = {d.len, d.wid, d.hgt, d.sc0, d.sc1, d.goal_x, d.goal_y};
            for (int k = 0; k < 7; k++) up(planes[k] + o, st.planes[k].data(), st.planes[k].size() * 4);
            up(d.etype + o, st.etype.data(), st.etype.size() * 4);
            up(d.agent_id + o, st.agent_id.data(), st.agent_id.size() * 4);
            up(d.resp + o, st.resp.data(), st.resp.size() * 4);
            up(d.controlled + o, st.controlled.data(), st.controlled.size() * 4);
            up(d.metadata + o * 4, st.metadata.data(), st.metadata.size() * 4);
            up(d.deleted + o, st.deleted.data(), st.deleted.size() * 4);
            up(d.means + static_cast<size_t>(w0) * 3, st.means.data(), st.means.size() * 4);
            up(d.map_name + static_cast<size_t>(w0) * 32, st.map_name.data(), st.map_name.size() * 4);
            up(d.scenario_id + static_cast<size_t>(w0) * 32, st.scenario_id.data(), st.scenario_id.size() * 4);
            up(d.shape + static_cast<size_t>(w0) * 2, st.shape.data(), st.shape.size() * 4);
            st = Staging();
        };
        // Parse the distinct scenes of this call on a few host threads first: MapReader::parseAndWriteOut runs
        // on ONE thread per set_maps in the reference (src/mgr.cpp:630-647), ~9 ms of JSON per scene here.
        {
            std::vector<std::string> todo;
            for (int w : sorted_worlds)
                if (scene_cache.emplace(scenes[w], nullptr).second) todo.push_back(scenes[w]);
            std::vector<std::shared_ptr<const gd::SceneMap>> parsed(todo.size());
            std::vector<std::exception_ptr> errors(todo.size());
            const unsigned hc = std::thread::hardware_concurrency();
            const size_t nthreads = std::min<size_t>(todo.size(), std::min<size_t>(16, hc ? hc : 1));
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (size_t k; (k = next.fetch_add(1)) < todo.size();) {
                    try {
                        parsed[k] = gd::load_scene(todo[k], params.polylineReductionThreshold);
                    } catch (...) {
                        errors[k] = std::current_exception();
                    }
                }
            };
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nthreads; t++) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
            for (size_t k = 0; k < todo.size(); k++) {
                if (errors[k]) std::rethrow_exception(errors[k]);  // first failing scene in world order
                scene_cache[todo[k]] = parsed[k];
            }
        }
        int run_start = -1, run_len = 0;
        for (int w : sorted_worlds) {
            if (run_len > 0 && (w != run_start + run_len || run_len == kRun)) {
                flush(run_start, run_len);
                run_len = 0;
            }
            if (run_len == 0) run_start = w;
            run_len++;
            const std::string &path = scenes[w];
            const int32_t *del = deleted.data() + static_cast<size_t>(w) * A;
            int ndel = 0;
            std::string key = path;
            for (int i = 0; i < A; i++)
                if (del[i] != -1) { ndel = i + 1; }
            for (int i = 0; i < ndel; i++) key += "|" + std::to_string(del[i]);
            std::shared_ptr<gd::HostWorld> hw;
            auto it = world_cache.find(key);
            if (it != world_cache.end()) {
                hw = it->second;
            } else {
                auto sit = scene_cache.find(path);  // filled above
                hw = std::make_shared<gd::HostWorld>();
                gd::build_host_world(*sit->second, params, A, del, ndel, *hw);
                // the row kernel's 32-byte road record restores the z scale from the entity type (1 for stop signs, 0.1
                // for everything else: scene.cpp put_road callers); checked here, before anything of this call is uploaded
                for (size_t r = 0; r * 8 < hw->road_aux.size(); r++) {
                    const float *a = &hw->road_aux[r * 8];
                    if (a[4] != (static_cast<int>(a[5]) == gd::ET_StopSign ? 1.f : 0.1f))
                        throw std::runtime_error("road record: unexpected z scale for this entity type");
                }
                world_cache.emplace(key, hw);
            }
            st.traj.insert(st.traj.end(), hw->trajectory.begin(), hw->trajectory.end());
            // map_observation_tensor rows + MapObservation::zero() padding (src/level_gen.cpp:331-335)
            const size_t m0 = st.map_obs.size();
            st.map_obs.resize(m0 + static_cast<size_t>(GD_MAX_ROAD_ENTITIES) * 9, 0.f);
            std::memcpy(st.map_obs.data() + m0, hw->map_obs.data(), sizeof(float) * hw->map_obs.size());
            for (int r = hw->num_roads; r < GD_MAX_ROAD_ENTITIES; r++) {
                st.map_obs[m0 + static_cast<size_t>(r) * 9 + 7] = -1.f;
                st.map_obs[m0 + static_cast<size_t>(r) * 9 + 8] = -1.f;
            }
            for (int a = 0; a < A; a++) {  // AoS rows -> SoA planes
                st.planes[0].push_back(hw->size[a * 3 + 0]);
                st.planes[1].push_back(hw->size[a * 3 + 1]);
                st.planes[2].push_back(hw->size[a * 3 + 2]);
                st.planes[3].push_back(hw->scale[a * 2 + 0]);
                st.planes[4].push_back(hw->scale[a * 2 + 1]);
                st.planes[5].push_back(hw->goal[a * 2 + 0]);
                st.planes[6].push_back(hw->goal[a * 2 + 1]);
            }
            st.etype.insert(st.etype.end(), hw->etype.begin(), hw->etype.end());
            st.agent_id.insert(st.agent_id.end(), hw->agent_id.begin(), hw->agent_id.end());
            st.resp.insert(st.resp.end(), hw->resp.begin(), hw->resp.end());
            st.controlled.insert(st.controlled.end(), hw->controlled.begin(), hw->controlled.end());
            st.metadata.insert(st.metadata.end(), hw->metadata.begin(), hw->metadata.end());
            st.deleted.insert(st.deleted.end(), del, del + A);
            st.means.insert(st.means.end(), hw->mean, hw->mean + 3);
            st.map_name.insert(st.map_name.end(), hw->map_name, hw->map_name + 32);
            st.scenario_id.insert(st.scenario_id.end(), hw->scenario_id, hw->scenario_id + 32);
            st.shape.push_back(hw->num_agents);
            st.shape.push_back(hw->num_roads);
            w_xy[w] = hw->road_xy;
            w_agents[w] = hw->num_agents;
            w_resp[w] = hw->resp;
            w_aux[w] = hw->road_aux;
            w_boxes[w] = hw->boxes;
            w_grid[w] = gd::GridHdr{hw->grid_ox, hw->grid_oy, 1.f / hw->grid_cell, hw->grid_nx, hw->grid_ny, 0, 0, 0};
            w_cell_off[w] = hw->cell_off;
            w_cell_items[w] = hw->cell_items;
            build_road_grid(w);
            rebuilt[w] = 1;
        }
        flush(run_start, run_len);
        // repack the road CSR
        std::vector<int32_t> road_off(W + 1, 0), box_off(W + 1, 0);
        for (int w = 0; w < W; w++) {
            road_off[w + 1] = road_off[w] + static_cast<int32_t>(w_xy[w].size() / 2);
            box_off[w + 1] = box_off[w] + static_cast<int32_t>(w_boxes[w].size());
        }
        const size_t nroad = road_off[W], nbox = box_off[W];
        // k_map_obs requests chunks of 32 roads up to 256 roads past a world's last one, and the fused set-order write-out
        // reads road_rec[first road of the world] even for a world without roads: the arrays always end in 320 readable
        // pad entries (also when a rebuild fits the old capacity, and when no world has a road)
        if (nroad + 320 > road_cap || !d_road_xy) {
            if (d_road_xy) (void)hipFree(d_road_xy);
            if (d_road_aux) (void)hipFree(d_road_aux);
            if (d_road_rec) (void)hipFree(d_road_rec);
            road_cap = nroad + nroad / 8 + 640;
            HIP_CHECK(hipMalloc(&d_road_xy, road_cap * sizeof(float) * 2));
            HIP_CHECK(hipMalloc(&d_road_aux, road_cap * sizeof(float) * 8));
            HIP_CHECK(hipMalloc(&d_road_rec, road_cap * sizeof(float) * 8));
            HIP_CHECK(hipMemset(d_road_xy, 0, road_cap * sizeof(float) * 2));
            HIP_CHECK(hipMemset(d_road_aux, 0, road_cap * sizeof(float) * 8));
            HIP_CHECK(hipMemset(d_road_rec, 0, road_cap * sizeof(float) * 8));
        }
        if (nbox > box_cap) {
            if (d_boxes) (void)hipFree(d_boxes);
            box_cap = nbox + nbox / 8 + 64;
            HIP_CHECK(hipMalloc(&d_boxes, box_cap * sizeof(gd::RoadBox)));
        }
        {
            std::vector<float> xy(nroad * 2), aux(nroad * 8);
            std::vector<gd::RoadBox> boxes(nbox);
            for (int w = 0; w < W; w++) {
                std::copy(w_xy[w].begin(), w_xy[w].end(), xy.begin() + static_cast<size_t>(road_off[w]) * 2);
                std::copy(w_aux[w].begin(), w_aux[w].end(), aux.begin() + static_cast<size_t>(road_off[w]) * 8);
                std::copy(w_boxes[w].begin(), w_boxes[w].end(), boxes.begin() + box_off[w]);
            }
            if (nroad) {
                HIP_CHECK(hipMemcpy(d_road_xy, xy.data(), xy.size() * sizeof(float), hipMemcpyHostToDevice));
                HIP_CHECK(hipMemcpy(d_road_aux, aux.data(), aux.size() * sizeof(float), hipMemcpyHostToDevice));
                // the row kernel's 32-byte record: aux is (qw, qz, d0, d1, d2, type, id, mapType); d2 is a function of the
                // type (validated while staging), which the kernel restores
                std::vector<float> rec(nroad * 8);
                for (size_t r = 0; r < nroad; r++) {
                    const float *a = &aux[r * 8];
                    const uint32_t bits = (static_cast<uint32_t>(static_cast<int>(a[5])) & 0xffu) |
                                          (static_cast<uint32_t>(static_cast<int>(a[7]) + 1) << 8);
                    float fb;
                    std::memcpy(&fb, &bits, sizeof(fb));
                    const float row[8] = {xy[r * 2], xy[r * 2 + 1], a[0], a[1], a[2], a[3], a[6], fb};
                    std::copy(row, row + 8, rec.begin() + r * 8);
                }
                HIP_CHECK(hipMemcpy(d_road_rec, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            HIP_CHECK(hipMemsetAsync(d.sel_hdr, 0xff, sizeof(float4) * 2 * static_cast<size_t>(W) * d.A, stream));  // count -1: nothing selected yet (ordered before the kernels of this stream)
            if (nbox) HIP_CHECK(hipMemcpy(d_boxes, boxes.data(), nbox * sizeof(gd::RoadBox), hipMemcpyHostToDevice));
        }
        {
            size_t ncell = 0, nitem = 0;
            for (int w = 0; w < W; w++) {
                w_grid[w].cell_base = static_cast<int>(ncell);
                w_grid[w].item_base = static_cast<int>(nitem);
                ncell += w_cell_off[w].size();
                nitem += w_cell_items[w].size();
            }
            if (ncell > cell_cap) {
                if (d_cell_off) (void)hipFree(d_cell_off);
                cell_cap = ncell + ncell / 8 + 64;
                HIP_CHECK(hipMalloc(&d_cell_off, cell_cap * sizeof(int32_t)));
            }
            if (nitem > item_cap) {
                if (d_cell_items) (void)hipFree(d_cell_items);
                if (d_cell_hdr) (void)hipFree(d_cell_hdr);
                item_cap = nitem + nitem / 8 + 64;
                HIP_CHECK(hipMalloc(&d_cell_items, item_cap * sizeof(int32_t)));
                HIP_CHECK(hipMalloc(&d_cell_hdr, item_cap * sizeof(float) * 4));
            }
            std::vector<int32_t> co(ncell), ci(nitem);
            for (int w = 0; w < W; w++) {
                std::copy(w_cell_off[w].begin(), w_cell_off[w].end(), co.begin() + w_grid[w].cell_base);
                std::copy(w_cell_items[w].begin(), w_cell_items[w].end(), ci.begin() + w_grid[w].item_base);
            }
            if (ncell) HIP_CHECK(hipMemcpy(d_cell_off, co.data(), ncell * sizeof(int32_t), hipMemcpyHostToDevice));
            if (nitem) HIP_CHECK(hipMemcpy(d_cell_items, ci.data(), nitem * sizeof(int32_t), hipMemcpyHostToDevice));
            {
                std::vector<float> ch(nitem * 4);
                for (int w = 0; w < W; w++) {
                    const std::vector<int32_t> &items = w_cell_items[w];
                    for (size_t i = 0; i < items.size(); i++) {
                        const gd::RoadBox &b = w_boxes[w][items[i]];
                        float *o = &ch[(static_cast<size_t>(w_grid[w].item_base) + i) * 4];
                        // (centre, bounding radius, entity type | local box index << 8): what the cull needs and where the box is
                        const uint32_t packed = (static_cast<uint32_t>(static_cast<int>(b.type)) & 0xffu) | (static_cast<uint32_t>(items[i]) << 8);
                        o[0] = b.cx; o[1] = b.cy; o[2] = b.radius;
                        std::memcpy(&o[3], &packed, sizeof(packed));
                    }
                }
                if (nitem) HIP_CHECK(hipMemcpy(d_cell_hdr, ch.data(), ch.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            HIP_CHECK(hipMemcpy(const_cast<gd::GridHdr *>(d.grid), w_grid.data(), sizeof(gd::GridHdr) * W, hipMemcpyHostToDevice));
            d.cell_off = static_cast<const int32_t *>(d_cell_off);
            d.cell_items = static_cast<const int32_t *>(d_cell_items);
            d.cell_hdr = static_cast<const float4 *>(d_cell_hdr);
        }
        HIP_CHECK(hipMemcpy(const_cast<int32_t *>(d.road_off), road_off.data(), sizeof(int32_t) * (W + 1), hipMemcpyHostToDevice));
        upload_road_grids();
        {
            // longest-first launch order of the road kernel: its time per world grows with the road count (the kernel
            // re-sorts by measured cycles after every launch; this is the order of the first one)
            const int parts = d.A / GD_MAP_OBS_AW;
            std::vector<int32_t> order(W);
            for (int w = 0; w < W; w++) order[w] = w;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                return road_off[x + 1] - road_off[x] > road_off[y + 1] - road_off[y];
            });
            std::vector<int32_t> waves(static_cast<size_t>(W) * parts);
            for (int k = 0; k < W; k++)
                for (int q = 0; q < parts; q++) waves[static_cast<size_t>(k) * parts + q] = order[k] * parts + q;
            HIP_CHECK(hipMemcpy(d.wave_order, waves.data(), sizeof(int32_t) * waves.size(), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemsetAsync(d.wave_cost, 0, sizeof(uint32_t) * waves.size(), stream));
        }
        HIP_CHECK(hipMemcpy(const_cast<int32_t *>(d.box_off), box_off.data(), sizeof(int32_t) * (W + 1), hipMemcpyHostToDevice));
        d.road_xy = static_cast<const float2 *>(d_road_xy);
        d.road_aux = static_cast<const float4 *>(d_road_aux);
        d.road_rec = static_cast<const float4 *>(d_road_rec);
        d.boxes = static_cast<const float4 *>(d_boxes);
        HIP_CHECK(hipMemcpy(d.rebuilt_flags, rebuilt.data(), sizeof(int32_t) * W, hipMemcpyHostToDevice));
        choose_set_schedule();
        {
            // the box around every world's roads: an agent farther than the radius from it has no road in reach (linear scan,
            // rank replay)
            std::vector<float> bb(static_cast<size_t>(W) * 4);
            for (int w = 0; w < W; w++) {
                float lo_x = INFINITY, lo_y = INFINITY, hi_x = -INFINITY, hi_y = -INFINITY;
                for (size_t r = 0; r * 2 < w_xy[w].size(); r++) {
                    lo_x = std::min(lo_x, w_xy[w][2 * r]); hi_x = std::max(hi_x, w_xy[w][2 * r]);
                    lo_y = std::min(lo_y, w_xy[w][2 * r + 1]); hi_y = std::max(hi_y, w_xy[w][2 * r + 1]);
                }
                bb[w * 4 + 0] = lo_x; bb[w * 4 + 1] = lo_y; bb[w * 4 + 2] = hi_x; bb[w * 4 + 3] = hi_y;
            }
            HIP_CHECK(hipMemcpy(const_cast<float4 *>(d.road_bbox), bb.data(), bb.size() * sizeof(float), hipMemcpyHostToDevice));
            std::vector<float> rbmax(W, 0.f);  // (road_aux: qw qz d0 d1 | d2 type id mapType)
            for (int w = 0; w < W; w++)
                for (size_t r = 0; r * 8 < w_aux[w].size(); r++)
                    rbmax[w] = std::max(rbmax[w], std::sqrt(w_aux[w][r * 8 + 2] * w_aux[w][r * 8 + 2] + w_aux[w][r * 8 + 3] * w_aux[w][r * 8 + 3]));
            HIP_CHECK(hipMemcpy(const_cast<float *>(d.road_rbmax), rbmax.data(), rbmax.size() * sizeof(float), hipMemcpyHostToDevice));
            // the circle around every GD_LIN_BLK consecutive road points of a world (centre of their bounding box, the largest
            // distance from it to one of them, rounded up)
            std::vector<int32_t> boff(W + 1, 0);
            for (int w = 0; w < W; w++) boff[w + 1] = boff[w] + static_cast<int32_t>((w_xy[w].size() / 2 + GD_LIN_BLK - 1) / GD_LIN_BLK);
            std::vector<float> blk(static_cast<size_t>(boff[W]) * 4 + 4, 0.f);
            for (int w = 0; w < W; w++) {
                const std::vector<float> &xy = w_xy[w];
                const size_t nr = xy.size() / 2;
                for (size_t b = 0; b * GD_LIN_BLK < nr; b++) {
                    const size_t r_lo = b * GD_LIN_BLK, r_hi = std::min(nr, r_lo + GD_LIN_BLK);
                    float lo_x = INFINITY, lo_y = INFINITY, hi_x = -INFINITY, hi_y = -INFINITY;
                    for (size_t r = r_lo; r < r_hi; r++) {
                        lo_x = std::min(lo_x, xy[2 * r]); hi_x = std::max(hi_x, xy[2 * r]);
                        lo_y = std::min(lo_y, xy[2 * r + 1]); hi_y = std::max(hi_y, xy[2 * r + 1]);
                    }
                    const float cx = 0.5f * (lo_x + hi_x), cy = 0.5f * (lo_y + hi_y);
                    float rad = 0.f;
                    for (size_t r = r_lo; r < r_hi; r++)
                        rad = std::max(rad, std::sqrt((xy[2 * r] - cx) * (xy[2 * r] - cx) + (xy[2 * r + 1] - cy) * (xy[2 * r + 1] - cy)));
                    float *o = &blk[(static_cast<size_t>(boff[w]) + b) * 4];
                    o[0] = cx; o[1] = cy; o[2] = rad * 1.0001f + 1e-3f; o[3] = 0.f;
                }
            }
            if (blk.size() / 4 > blk_cap || !d_road_blk) {
                if (d_road_blk) (void)hipFree(d_road_blk);
                blk_cap = blk.size() / 4 + blk.size() / 32 + 64;
                HIP_CHECK(hipMalloc(&d_road_blk, blk_cap * sizeof(float) * 4));
            }
            HIP_CHECK(hipMemcpy(d_road_blk, blk.data(), blk.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(const_cast<int32_t *>(d.blk_off), boff.data(), sizeof(int32_t) * (W + 1), hipMemcpyHostToDevice));
            d.road_blk = static_cast<const float4 *>(d_road_blk);
            // no row written before this call describes the worlds as they are now (roads, agent slots): every pose stamp dies
            HIP_CHECK(hipMemsetAsync(d.pose_stamp, 0xff, sizeof(uint4) * static_cast<size_t>(W) * A, stream));
            HIP_CHECK(hipMemsetAsync(d.bev_dirty, 1, sizeof(int32_t) * static_cast<size_t>(W) * A, stream));
            HIP_CHECK(hipMemsetAsync(d.lidar_dirty, 1, sizeof(int32_t) * static_cast<size_t>(W) * A, stream));
            HIP_CHECK(hipMemsetAsync(d.lidar_head, 0xff, sizeof(float) * static_cast<size_t>(W) * A, stream));
        }
        if (rk_possible) {
            // Which worlds take the rank replay (neither path changes a result).  Measured at 64 agent slots (road
            // observation, ms): 1024 worlds x 4096 roads 1.19 ranked / 1.84 on keys; 1024 Waymo tiles (346-873 roads) 0.44 /
            // 0.47; 4096 Waymo tiles 0.94 / 1.74.  The rank kernels' fixed costs (seven launches, a replay as long as the
            // longest agent's candidate list) pay once k_map_obs's workgroups fill the chip: large worlds always, every
            // world with at least K roads from one full generation of k_map_obs workgroups on.  GPUDRIVE_RANK_MIN_ROADS
            // pins the threshold.
            int groups = 0;
            for (int w = 0; w < W; w++) groups += (w_agents[w] + 31) / 32;
            const char *pin = std::getenv("GPUDRIVE_RANK_MIN_ROADS");
            d.rk_min_roads = pin ? std::atoi(pin) : (groups >= 4 * cu_count ? GD_MAP_OBS_K : 1536);
            // Worlds of every size take it since round 4 (agents whose candidates overflow the standard ranking's 1272 go to the
            // long-list instantiation, 2552; groups that keep overflowing even that bypass the rank kernels on their own:
            // rk_streak).  GPUDRIVE_RANK_MAX_ROADS pins an upper limit for experiments.
            const char *pin_max = std::getenv("GPUDRIVE_RANK_MAX_ROADS");
            d.rk_max_roads = pin_max ? std::atoi(pin_max) : GD_MAX_ROAD_ENTITIES;
            d.rk_on = 0;
            for (int w = 0; w < W; w++) {
                const int R = road_off[w + 1] - road_off[w];
                if (R >= std::max(d.rk_min_roads, GD_MAP_OBS_K) && R <= d.rk_max_roads) d.rk_on = 1;
            }
            if (d.rk_on && ensure_rank_buffers()) reset_rank_state();
        }
        {
            // one BEV workgroup per LIVE agent: a workgroup that only finds out it has no agent still has to be given
            // its 48 KB of LDS and eight waves first
            std::vector<int32_t> live;
            live.reserve(static_cast<size_t>(W) * A);
            for (int a = 0; a < A; a++)  // agent-major: consecutive workgroups belong to different worlds
                for (int w = 0; w < W; w++)
                    if (a < w_agents[w]) live.push_back(w * A + a);
            d.live_count = static_cast<int>(live.size());
            if (!live.empty()) HIP_CHECK(hipMemcpy(d.live_list, live.data(), sizeof(int32_t) * live.size(), hipMemcpyHostToDevice));
            // likewise the set-order road kernel's workgroups (4 waves x set_apw agents each)
            std::vector<int32_t> groups;
            const int per = 4 * d.set_apw;
            for (int g = 0; g * per < A; g++)  // group-major: consecutive workgroups belong to different worlds
                for (int w = 0; w < W; w++)
                    if (g * per < w_agents[w]) groups.push_back(w << 8 | g);
            d.set_group_count = static_cast<int>(groups.size());
            if (!groups.empty()) HIP_CHECK(hipMemcpy(d.set_groups, groups.data(), sizeof(int32_t) * groups.size(), hipMemcpyHostToDevice));
        }
        {
            // The linear scan's work lists: the live agents as (world << 8 | agent), world-major inside eight classes, the classes
            // interleaved workgroup by workgroup (4 * lin_apw entries each): workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md,
            // dispatch), so every workgroup that holds agents of a world -- and the world's road arrays -- stays on one XCD's L2.
            // Classes are filled greedily (fewest entries so far) so that ragged batches leave few filler entries.  Two lists:
            // every live agent (reset passes), and the agents that can move (step passes: a `Static` agent's rows were written
            // by the reset pass that follows every rebuild, and nothing moves it afterwards).
            const int per = 4 * d.lin_apw;
            auto build = [&](bool dyn_only, std::vector<int32_t> &list) -> int {
                std::vector<int32_t> seq[8];
                for (int w = 0; w < W; w++) {
                    std::vector<int32_t> mine;
                    for (int a = 0; a < w_agents[w]; a++)
                        if (!dyn_only || w_resp[w][a] != gd::RESP_Static) mine.push_back(w << 8 | a);
                    if (mine.empty()) continue;
                    int c = w % 8;
                    for (int k = 0; k < 8; k++)
                        if (seq[k].size() + 4 * static_cast<size_t>(A) < seq[c].size()) c = k;  // only when a class runs far ahead
                    seq[c].insert(seq[c].end(), mine.begin(), mine.end());
                }
                size_t blocks = 0;
                for (auto &q : seq) blocks = std::max(blocks, (q.size() + per - 1) / per);
                list.assign(blocks * 8 * per, -1);
                for (int c = 0; c < 8; c++)
                    for (size_t j = 0; j < seq[c].size(); j++) list[((j / per) * 8 + c) * per + j % per] = seq[c][j];
                return static_cast<int>(blocks * 8);
            };
            std::vector<int32_t> full, dyn;
            d.lin_blocks = build(false, full);
            d.lin_blocks_dyn = build(true, dyn);
            lin_static_agents = 0;
            for (int32_t e : full) lin_static_agents += e >= 0;
            for (int32_t e : dyn) lin_static_agents -= e >= 0;
            if (full.size() > lin_cap || dyn.size() > lin_cap) throw std::runtime_error("linear-scan work list: more entries than the list holds");
            if (!full.empty()) HIP_CHECK(hipMemcpy(d_lin_list, full.data(), sizeof(int32_t) * full.size(), hipMemcpyHostToDevice));
            if (!dyn.empty()) HIP_CHECK(hipMemcpy(d_lin_list_dyn, dyn.data(), sizeof(int32_t) * dyn.size(), hipMemcpyHostToDevice));
            d.lin_list = d_lin_list;
            d.lin_list_dyn = d_lin_list_dyn;
        }
        launch(gd::KERNEL_PADDING, false);
        // the packed observation's rows of padding agents come from the raw padding rows just written (the live agents' rows are
        // written by the reset pass that follows every rebuild)
        if (d.pack) gd::launch_pack_obs(d, stream, d.pack);
    }

    // the packed observation can be written where the rows are produced by every road path -- the linear scan, set order
    // (fused write-out), and k_map_rows behind the reference-order selections and the unfused set-order one -- except the
    // linear scan's legacy path (GPUDRIVE_LINEAR_LEGACY=1, an A/B switch)
    bool direct_pack_supported() const {
        if (params.disableClassicalObs) return false;
        if (params.roadObservationAlgorithm != GD_ROADS_K_NEAREST) return d.lin_on != 0;
        return true;
    }

    // Uniform grid over the (x, y) of ALL roads of world w: cells of at least 16 m, at most 64 x 64 of them; a road
    // belongs to the cell its point falls into, and a cell lists its roads in ascending index.
    void build_road_grid(int w) {
        const std::vector<float> &xy = w_xy[w];
        const size_t n = xy.size() / 2;
        gd::GridHdr g{0.f, 0.f, 1.f, 1, 1, 0, 0, 0};
        std::vector<int32_t> off(2, 0);
        std::vector<uint16_t> items(n);
        if (n) {
            float minx = xy[0], maxx = xy[0], miny = xy[1], maxy = xy[1];
            for (size_t r = 1; r < n; r++) {
                minx = std::min(minx, xy[2 * r]); maxx = std::max(maxx, xy[2 * r]);
                miny = std::min(miny, xy[2 * r + 1]); maxy = std::max(maxy, xy[2 * r + 1]);
            }
            const float cell = std::max(GD_GRID_MIN_CELL, std::max(maxx - minx, maxy - miny) / 64.f + 1e-3f);
            g.ox = minx; g.oy = miny; g.inv_cell = 1.f / cell;
            g.nx = std::max(1, std::min(64, static_cast<int>((maxx - minx) * g.inv_cell) + 1));
            g.ny = std::max(1, std::min(64, static_cast<int>((maxy - miny) * g.inv_cell) + 1));
            const int nc = g.nx * g.ny;
            std::vector<int32_t> cell_of(n);
            off.assign(nc + 1, 0);
            for (size_t r = 0; r < n; r++) {
                const int cx = std::max(0, std::min(g.nx - 1, static_cast<int>((xy[2 * r] - g.ox) * g.inv_cell)));
                const int cy = std::max(0, std::min(g.ny - 1, static_cast<int>((xy[2 * r + 1] - g.oy) * g.inv_cell)));
                cell_of[r] = cy * g.nx + cx;
                off[cell_of[r] + 1]++;
            }
            for (int c = 0; c < nc; c++) off[c + 1] += off[c];
            std::vector<int32_t> fill(off.begin(), off.end() - 1);
            for (size_t r = 0; r < n; r++) items[fill[cell_of[r]]++] = static_cast<uint16_t>(r);  // ascending r within a cell
        }
        w_rgrid[w] = g;
        w_rcell_off[w] = std::move(off);
        w_rcell_items[w] = std::move(items);
    }

    void upload_road_grids() {
        size_t ncell = 0, nitem = 0;
        for (int w = 0; w < W; w++) {
            w_rgrid[w].cell_base = static_cast<int>(ncell);
            w_rgrid[w].item_base = static_cast<int>(nitem);
            ncell += w_rcell_off[w].size();
            nitem += w_rcell_items[w].size();
        }
        if (ncell > rcell_cap) {
            if (d_rcell_off) (void)hipFree(d_rcell_off);
            rcell_cap = ncell + ncell / 8 + 64;
            HIP_CHECK(hipMalloc(&d_rcell_off, rcell_cap * sizeof(int32_t)));
        }
        if (nitem + 64 > ritem_cap) {
            if (d_rcell_items) (void)hipFree(d_rcell_items);
            if (d_rcell_xy) (void)hipFree(d_rcell_xy);
            if (d_rcell_pos) (void)hipFree(d_rcell_pos);
            ritem_cap = nitem + nitem / 8 + 128;
            HIP_CHECK(hipMalloc(&d_rcell_items, ritem_cap * sizeof(uint16_t)));
            HIP_CHECK(hipMalloc(&d_rcell_xy, ritem_cap * sizeof(float) * 2));
            HIP_CHECK(hipMalloc(&d_rcell_pos, ritem_cap * sizeof(uint16_t)));
        }
        std::vector<int32_t> co(ncell);
        std::vector<uint16_t> ci(nitem);
        std::vector<float> cxy(nitem * 2);
        std::vector<uint16_t> cpos(nitem);  // (a world's items are its roads, one each: item_base is also its first road)
        for (int w = 0; w < W; w++) {
            std::copy(w_rcell_off[w].begin(), w_rcell_off[w].end(), co.begin() + w_rgrid[w].cell_base);
            std::copy(w_rcell_items[w].begin(), w_rcell_items[w].end(), ci.begin() + w_rgrid[w].item_base);
            for (size_t k = 0; k < w_rcell_items[w].size(); k++) {
                const size_t r = w_rcell_items[w][k], o = (static_cast<size_t>(w_rgrid[w].item_base) + k) * 2;
                cxy[o] = w_xy[w][2 * r];
                cxy[o + 1] = w_xy[w][2 * r + 1];
                cpos[static_cast<size_t>(w_rgrid[w].item_base) + r] = static_cast<uint16_t>(k);
            }
        }
        if (ncell) HIP_CHECK(hipMemcpy(d_rcell_off, co.data(), ncell * sizeof(int32_t), hipMemcpyHostToDevice));
        if (nitem) HIP_CHECK(hipMemcpy(d_rcell_items, ci.data(), nitem * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (nitem) HIP_CHECK(hipMemcpy(d_rcell_xy, cxy.data(), nitem * 2 * sizeof(float), hipMemcpyHostToDevice));
        if (nitem) HIP_CHECK(hipMemcpy(d_rcell_pos, cpos.data(), nitem * sizeof(uint16_t), hipMemcpyHostToDevice));
        d.rcell_xy = static_cast<const float2 *>(d_rcell_xy);
        d.rcell_pos = static_cast<const uint16_t *>(d_rcell_pos);
        HIP_CHECK(hipMemcpy(const_cast<gd::GridHdr *>(d.rgrid), w_rgrid.data(), sizeof(gd::GridHdr) * W, hipMemcpyHostToDevice));
        d.rcell_off = static_cast<const int32_t *>(d_rcell_off);
        d.rcell_items = static_cast<const uint16_t *>(d_rcell_items);
        // the roads changed: no previous selection bounds the next one
        std::vector<float> prev(static_cast<size_t>(W) * A * 4, 0.f);
        for (size_t i = 0; i < static_cast<size_t>(W) * A; i++) prev[i * 4 + 2] = INFINITY;
        HIP_CHECK(hipMemcpy(d.knn_prev, prev.data(), prev.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // No checkpoint of a previous selection survives a change of the worlds' roads or agent slots: in its next selection
    // every agent is bounded afresh inside k_knn_scan (a distance histogram at geometric road counts, map_obs_rank.hip).
    void reset_rank_state() {
        HIP_CHECK(hipMemset(d.cp_hdr, 0, sizeof(float4) * 2 * static_cast<size_t>(W) * A));
        HIP_CHECK(hipMemset(d.rk_fallback, 0, sizeof(int32_t) * (static_cast<size_t>(W) * A / 32)));
        HIP_CHECK(hipMemset(d.rk_streak, 0, sizeof(int32_t) * (static_cast<size_t>(W) * A / 32)));
        HIP_CHECK(hipMemset(d.rk_hist, 0, sizeof(int32_t) * GD_RANK_AUDIT));  // bin counts, the lists of ranked agents: empty (the audit counter behind them keeps counting)
    }

    void do_reset(const std::vector<int32_t> &flags) {
        upload_flags(d.reset_flags, flags);
        reset_flagged(false);
    }

    // resetSystem + the observation half of the task graph (src/sim.cpp:150-166, 960-971) for the worlds
    // whose reset flag is set ON THE DEVICE (by the upload above or by k_episode_step).  Like the
    // reference's Reset graph the observation systems re-run for EVERY world (not idempotent in the
    // reference: the collision system sees the already decremented step counter, so a reset anywhere can
    // raise collision flags elsewhere -- reproduced, and tested).  `gated`: the host does not know whether
    // k_episode_step flagged anything; the kernels are launched regardless and return at once unless
    // *any_reset is set, so a step without finished worlds costs three empty launches and no host sync.
    void reset_flagged(bool gated) {
        d.gate_any = gated ? 1 : 0;
        if (!gated) full_pass_next = false;  // (an ungated reset pass visits every live agent: nothing is owed any more)
        try {
            launch(gd::KERNEL_RESET, false);
            run_rest(false);
        } catch (...) {
            d.gate_any = 0;
            throw;
        }
        d.gate_any = 0;
    }
};

namespace {

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return GD_OK;
    } catch (const HipError &e) {
        return fail(GD_ERR_DEVICE, e.what());
    } catch (const std::invalid_argument &e) {
        const std::string m = e.what();
        return fail(m.find("cannot open") != std::string::npos ? GD_ERR_IO : GD_ERR_INVALID, m);
    } catch (const std::exception &e) {
        return fail(GD_ERR_PARSE, e.what());
    }
}

}  // namespace

extern "C" {

const char *gd_version(void) { return "gpudrive_amd 0.1 (gfx950)"; }
const char *gd_last_error(void) { return g_last_error.c_str(); }

void gd_default_params(gd_params *p) {  // src/init.hpp:111-127
    std::memset(p, 0, sizeof(*p));
    p->collisionBehaviour = GD_COLLISION_AGENT_STOP;
    p->maxNumControlledAgents = 10000;
    p->IgnoreNonVehicles = 0;
    p->roadObservationAlgorithm = GD_ROADS_K_NEAREST;
    p->initOnlyValidAgentsAtFirstStep = 1;
    p->isStaticAgentControlled = 0;
    p->dynamicsModel = GD_DYNAMICS_CLASSIC;
}

int gd_tensor_shape(int32_t id, int32_t W, int32_t A, gd_tensor_desc *out) {
    if (!out || id < 0 || id >= GD_T_COUNT || W < 1 || A < 2) return fail(GD_ERR_INVALID, "gd_tensor_shape: bad argument");
    const TensorSpec s = tensor_spec(id, W, A);
    out->data = nullptr;
    out->dtype = s.dtype;
    out->ndim = s.ndim;
    for (int i = 0; i < 5; i++) out->dims[i] = i < s.ndim ? s.dims[i] : 1;
    out->nbytes = spec_bytes(s);
    return GD_OK;
}

int gd_create(const gd_config *cfg, const gd_params *params, const char *const *scenes, gd_sim **out) {
    if (!cfg || !params || !scenes || !out) return fail(GD_ERR_INVALID, "gd_create: null argument");
    if (cfg->num_worlds < 1) return fail(GD_ERR_INVALID, "gd_create: num_worlds must be >= 1");
    if (cfg->max_agents != 64 && cfg->max_agents != 128)
        return fail(GD_ERR_INVALID, "gd_create: max_agents must be 64 or 128");
    if (params->rewardType == GD_REWARD_DENSE)
        return fail(GD_ERR_UNSUPPORTED, "RewardType::Dense is assert(false) in the reference (src/sim.cpp:579-583)");
    *out = nullptr;
    std::unique_ptr<gd_sim> s(new gd_sim());
    const int rc = guarded([&]() {
        int ndev = 0;
        HIP_CHECK(hipGetDeviceCount(&ndev));
        if (ndev < 1) throw HipError("no HIP device visible: the HIP path is mandatory, there is no CPU fallback");
        HIP_CHECK(hipSetDevice(cfg->device_id));
        s->cfg = *cfg;
        s->params = *params;
        s->W = cfg->num_worlds;
        s->A = cfg->max_agents;
        s->stream = static_cast<hipStream_t>(cfg->stream);
        const int W = s->W, A = s->A;
        for (int w = 0; w < W; w++) {
            if (!scenes[w]) throw std::invalid_argument("gd_create: null scene path");
            s->scenes.emplace_back(scenes[w]);
        }
        s->deleted.assign(static_cast<size_t>(W) * A, -1);  // src/sim.cpp:1003-1006
        s->w_xy.resize(W);
        s->w_agents.assign(W, 0);
        s->w_aux.resize(W);
        s->w_boxes.resize(W);
        s->w_grid.resize(W);
        s->w_cell_off.resize(W);
        s->w_cell_items.resize(W);
        s->w_rgrid.resize(W);
        s->w_rcell_off.resize(W);
        s->w_rcell_items.resize(W);
        for (int id = 0; id < GD_T_COUNT; id++) {
            if (id == GD_T_BEV && !cfg->alloc_bev && !cfg->external[id]) continue;
            const int64_t bytes = spec_bytes(tensor_spec(id, W, A));
            if (cfg->external[id]) {
                s->exported[id] = cfg->external[id];
            } else {
                HIP_CHECK(hipMalloc(&s->exported[id], bytes));
                s->owned[id] = true;
            }
            HIP_CHECK(hipMemset(s->exported[id], 0, bytes));
        }
        gd::DevSim &d = s->d;
        d.W = W;
        d.A = A;
        d.p = *params;
        d.knn_order = cfg->knn_order;
        // (not on the legacy null stream: it synchronises with every blocking stream and cannot be captured)
        d.split_partner = (s->stream != nullptr && std::getenv("GPUDRIVE_SPLIT_PARTNER") != nullptr) ? 1 : 0;
        HIP_CHECK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
        d.step_dbg = 0;
#ifdef GD_DIAG
        if (const char *e = std::getenv("GPUDRIVE_STEP_DBG")) d.step_dbg = std::atoi(e);
#endif
        {
            (void)hipDeviceGetAttribute(&s->cu_count, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
            d.set_fused_rows = 0;  // chosen with the worlds (rebuild_worlds -> choose_set_schedule)
            d.set_apw = 16;
        }
        d.lidar_half_angle = cfg->lidar_half_angle;
        {
            // radiusFilter keeps length() <= radius (src/knn.hpp:88); on squared keys: key <= kmax
            const float r = params->observationRadius;
            float k = r * r;
            while (k > 0.f && sqrtf(k) > r) k = std::nextafterf(k, 0.f);
            while (sqrtf(std::nextafterf(k, INFINITY)) <= r) k = std::nextafterf(k, INFINITY);
            d.radius_key_max = r >= 0.f ? k : -1.f;
        }
        d.action = static_cast<float *>(s->exported[GD_T_ACTION]);
        d.reward = static_cast<float *>(s->exported[GD_T_REWARD]);
        d.done = static_cast<int32_t *>(s->exported[GD_T_DONE]);
        d.info = static_cast<int32_t *>(s->exported[GD_T_INFO]);
        d.self_obs = static_cast<float *>(s->exported[GD_T_SELF_OBS]);
        d.abs_obs = static_cast<float *>(s->exported[GD_T_ABS_OBS]);
        d.partner = static_cast<float *>(s->exported[GD_T_PARTNER_OBS]);
        d.agent_map = static_cast<float *>(s->exported[GD_T_AGENT_MAP_OBS]);
        d.map_obs = static_cast<float *>(s->exported[GD_T_MAP_OBS]);
        d.lidar = static_cast<float *>(s->exported[GD_T_LIDAR]);
        d.bev = static_cast<float *>(s->exported[GD_T_BEV]);
        d.steps = static_cast<uint32_t *>(s->exported[GD_T_STEPS_REMAINING]);
        d.shape = static_cast<int32_t *>(s->exported[GD_T_SHAPE]);
        d.controlled = static_cast<int32_t *>(s->exported[GD_T_CONTROLLED_STATE]);
        d.resp_export = static_cast<int32_t *>(s->exported[GD_T_RESPONSE_TYPE]);
        d.traj = static_cast<float *>(s->exported[GD_T_EXPERT_TRAJECTORY]);
        d.means = static_cast<float *>(s->exported[GD_T_WORLD_MEANS]);
        d.metadata = static_cast<int32_t *>(s->exported[GD_T_METADATA]);
        d.deleted = static_cast<int32_t *>(s->exported[GD_T_DELETED_AGENTS]);
        d.map_name = static_cast<int32_t *>(s->exported[GD_T_MAP_NAME]);
        d.scenario_id = static_cast<int32_t *>(s->exported[GD_T_SCENARIO_ID]);
        const size_t WA = static_cast<size_t>(W) * A;
        d.px = s->alloc_internal<float>(WA); d.py = s->alloc_internal<float>(WA); d.pz = s->alloc_internal<float>(WA);
        d.qw = s->alloc_internal<float>(WA); d.qz = s->alloc_internal<float>(WA);
        d.vx = s->alloc_internal<float>(WA); d.vy = s->alloc_internal<float>(WA); d.vz = s->alloc_internal<float>(WA);
        d.collided = s->alloc_internal<int32_t>(WA);
        d.len = s->alloc_internal<float>(WA); d.wid = s->alloc_internal<float>(WA); d.hgt = s->alloc_internal<float>(WA);
        d.sc0 = s->alloc_internal<float>(WA); d.sc1 = s->alloc_internal<float>(WA);
        d.goal_x = s->alloc_internal<float>(WA); d.goal_y = s->alloc_internal<float>(WA);
        d.etype = s->alloc_internal<int32_t>(WA); d.agent_id = s->alloc_internal<int32_t>(WA);
        d.resp = s->alloc_internal<int32_t>(WA);
        d.sel_idx = s->alloc_internal<uint16_t>(static_cast<size_t>(WA) * GD_MAP_OBS_K);
        d.sel_hdr = s->alloc_internal<float4>(static_cast<size_t>(WA) * 2);
        d.sel_slot = s->alloc_internal<uint8_t>(static_cast<size_t>(WA) * GD_MAP_OBS_K);
        d.reset_flags = s->alloc_internal<int32_t>(W);
        d.rebuilt_flags = s->alloc_internal<int32_t>(W);
        d.any_reset = s->alloc_internal<int32_t>(1);
        d.gate_any = 0;
        d.road_off = s->alloc_internal<int32_t>(W + 1);
        d.live_list = s->alloc_internal<int32_t>(WA);
        d.live_count = 0;
        d.set_groups = s->alloc_internal<int32_t>(WA);
        d.set_group_count = 0;
        d.wave_order = s->alloc_internal<int32_t>(static_cast<size_t>(W) * (A / GD_MAP_OBS_AW));
        d.wave_cost = s->alloc_internal<uint32_t>(static_cast<size_t>(W) * (A / GD_MAP_OBS_AW));
        d.box_off = s->alloc_internal<int32_t>(W + 1);
        d.grid = s->alloc_internal<gd::GridHdr>(W);
        d.rgrid = s->alloc_internal<gd::GridHdr>(W);
        d.knn_prev = s->alloc_internal<float4>(WA);
        d.road_bbox = s->alloc_internal<float4>(W);
        d.road_rbmax = s->alloc_internal<float>(W);
        d.bev_dirty = s->alloc_internal<int32_t>(WA);
        d.bev_list = s->alloc_internal<int32_t>(WA);
        d.bev_count = s->alloc_internal<int32_t>(2);
        d.bev_all_dirty = 0;
        HIP_CHECK(hipMemset(d.bev_dirty, 1, sizeof(int32_t) * WA));  // (non-zero: everything is to be rasterised until k_world_step says otherwise)
        d.lidar_dirty = s->alloc_internal<int32_t>(WA);
        d.lidar_head = s->alloc_internal<float>(WA);
        HIP_CHECK(hipMemset(d.lidar_dirty, 1, sizeof(int32_t) * WA));
        HIP_CHECK(hipMemset(d.lidar_head, 0xff, sizeof(float) * WA));
        d.lin_apw = 2;
        if (const char *e = std::getenv("GPUDRIVE_LIN_AGENTS_PER_WAVE")) d.lin_apw = std::min(A / 4, std::max(1, std::atoi(e)));
        // worst case: every class as long as the longest one, which holds at most W / 8 + a few worlds' agents
        s->lin_cap = (static_cast<size_t>(W) + 64) * static_cast<size_t>(A) + 8 * 4 * static_cast<size_t>(d.lin_apw) + 64;
        s->d_lin_list = s->alloc_internal<int32_t>(s->lin_cap);
        s->d_lin_list_dyn = s->alloc_internal<int32_t>(s->lin_cap);
        d.lin_list = s->d_lin_list; d.lin_list_dyn = s->d_lin_list_dyn;
        d.lin_blocks = 0; d.lin_blocks_dyn = 0; d.lin_dyn_off = 0;
        s->w_resp.resize(W);
        d.lin_on = std::getenv("GPUDRIVE_LINEAR_LEGACY") == nullptr ? 1 : 0;
        d.pose_stamp = s->alloc_internal<uint4>(WA);
        d.pose_skip = std::getenv("GPUDRIVE_NO_POSE_SKIP") == nullptr ? 1 : 0;
        d.stat_skipped = s->alloc_internal<unsigned long long>(GD_SKIP_SLOTS);
        d.blk_off = s->alloc_internal<int32_t>(W + 1);
        // rank replay of the reference-order selection (map_obs_rank.hip): a fallback group is one workgroup of k_map_obs.
        // Its buffers (7.4 KB per agent slot) are allocated when a batch first takes the path (rebuild_worlds).
        d.rk_on = 0;
        d.rk_dbg = 0;
        d.rk_min_roads = 1536;
        s->rk_possible = GD_MAP_OBS_AW == 32 && cfg->knn_order != GD_KNN_SET_ORDER &&
                         params->roadObservationAlgorithm == GD_ROADS_K_NEAREST && std::getenv("GPUDRIVE_NO_RANK_REPLAY") == nullptr;
#ifdef GD_DIAG
        if (const char *e = std::getenv("GPUDRIVE_RANK_DBG")) d.rk_dbg = std::atoi(e);
#endif
        for (int i = 0; i < gd_sim::kRing; i++) {
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s->h_flags[i]), sizeof(int32_t) * W, hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&s->flag_ev[i], hipEventDisableTiming));
        }
        std::vector<int> all(W);
        for (int w = 0; w < W; w++) all[w] = w;
        s->rebuild_worlds(all);
        // Sim::Sim -> initWorld for every world, then Manager::reset({}) (src/sim.cpp:1008-1011, src/mgr.cpp:565)
        s->do_reset(std::vector<int32_t>(W, 1));
        HIP_CHECK(hipStreamSynchronize(s->stream));
    });
    if (rc != GD_OK) return rc;
    *out = s.release();
    return GD_OK;
}

void gd_destroy(gd_sim *sim) { delete sim; }

int gd_step(gd_sim *s) {
    if (!s) return fail(GD_ERR_INVALID, "gd_step: null sim");
    return guarded([&]() { s->step(); });
}

int gd_reset(gd_sim *s, const int32_t *idx, int32_t n) {
    if (!s || (n > 0 && !idx)) return fail(GD_ERR_INVALID, "gd_reset: bad argument");
    std::vector<int32_t> flags(s->W, 0);
    for (int i = 0; i < n; i++) {
        if (idx[i] < 0 || idx[i] >= s->W) return fail(GD_ERR_INVALID, "gd_reset: world index out of range");
        flags[idx[i]] = 1;
    }
    return guarded([&]() { s->do_reset(flags); });
}

int gd_set_maps(gd_sim *s, const char *const *scenes, int32_t n) {
    if (!s || !scenes) return fail(GD_ERR_INVALID, "gd_set_maps: null argument");
    if (n != s->W) return fail(GD_ERR_INVALID, "gd_set_maps: len(maps) must equal the number of worlds");
    const std::vector<std::string> old_scenes = s->scenes;
    const std::vector<int32_t> old_deleted = s->deleted;
    const int rc = guarded([&]() {
        for (int w = 0; w < n; w++) {
            if (!scenes[w]) throw std::invalid_argument("gd_set_maps: null scene path");
            s->scenes[w] = scenes[w];
        }
        std::fill(s->deleted.begin(), s->deleted.end(), -1);  // src/mgr.cpp:613-617
        std::vector<int> all(s->W);
        for (int w = 0; w < s->W; w++) all[w] = w;
        s->rebuild_worlds(all);
        s->do_reset(std::vector<int32_t>(s->W, 1));
    });
    if (rc != GD_OK && rc != GD_ERR_DEVICE) {  // leave the previous worlds in place on a bad file
        s->scenes = old_scenes;
        s->deleted = old_deleted;
    }
    return rc;
}

int gd_delete_agents(gd_sim *s, const int32_t *worlds, const int32_t *offsets, const int32_t *ids, int32_t nw) {
    if (!s || (nw > 0 && (!worlds || !offsets))) return fail(GD_ERR_INVALID, "gd_delete_agents: null argument");
    std::vector<int> touched;
    for (int i = 0; i < nw; i++) {
        const int w = worlds[i], cnt = offsets[i + 1] - offsets[i];
        if (w < 0 || w >= s->W) return fail(GD_ERR_INVALID, "gd_delete_agents: world index out of range");
        if (cnt < 0 || cnt > s->A) return fail(GD_ERR_INVALID, "gd_delete_agents: too many ids for one world");
    }
    return guarded([&]() {
        for (int i = 0; i < nw; i++) {
            const int w = worlds[i], cnt = offsets[i + 1] - offsets[i];
            for (int k = 0; k < cnt; k++) s->deleted[static_cast<size_t>(w) * s->A + k] = ids[offsets[i] + k];
            touched.push_back(w);
        }
        s->rebuild_worlds(touched);
        s->do_reset(std::vector<int32_t>(s->W, 1));  // src/mgr.cpp:712-714: reset(all)
    });
}

int gd_tensor(gd_sim *s, int32_t id, gd_tensor_desc *out) {
    if (!s || !out || id < 0 || id >= GD_T_COUNT) return fail(GD_ERR_INVALID, "gd_tensor: bad argument");
    if (!s->exported[id]) return fail(GD_ERR_UNSUPPORTED, "tensor not allocated (set gd_config.alloc_bev for the BEV tensor)");
    gd_tensor_shape(id, s->W, s->A, out);
    out->data = s->exported[id];
    return GD_OK;
}

int gd_pack_observations(gd_sim *s, float *out, int64_t out_bytes) {
    if (!s || !out) return fail(GD_ERR_INVALID, "gd_pack_observations: null argument");
    const int64_t D = 6 + static_cast<int64_t>(s->A - 1) * 6 + GD_MAP_OBS_K * 13;
    const int64_t need = static_cast<int64_t>(s->W) * s->A * D * 4;
    if (out_bytes < need) return fail(GD_ERR_INVALID, "gd_pack_observations: output buffer too small");
    return guarded([&]() {
        if (s->d.pack) {  // the step already wrote it (gd_attach_packed): nothing to do, or a copy for another buffer
            if (out != s->d.pack) HIP_CHECK(hipMemcpyAsync(out, s->d.pack, need, hipMemcpyDeviceToDevice, s->stream));
            return;
        }
        gd::launch_pack_obs(s->d, s->stream, out);
        HIP_CHECK(hipGetLastError());
    });
}

int gd_attach_packed(gd_sim *s, float *out, int64_t out_bytes, int32_t only) {
    if (!s) return fail(GD_ERR_INVALID, "gd_attach_packed: null sim");
    const int64_t D = 6 + static_cast<int64_t>(s->A - 1) * 6 + GD_MAP_OBS_K * 13;
    if (out && out_bytes < static_cast<int64_t>(s->W) * s->A * D * 4)
        return fail(GD_ERR_INVALID, "gd_attach_packed: output buffer too small");
    if (out && !s->direct_pack_supported())
        return fail(GD_ERR_UNSUPPORTED, "gd_attach_packed: not available with disableClassicalObs or GPUDRIVE_LINEAR_LEGACY=1: use gd_pack_observations");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->drop_graph();
        if (!out) {  // detach: the raw tensors are written again from the next pass on; bring them up to date now
            s->d.pack = nullptr;
            s->d.pack_only = 0;
            HIP_CHECK(hipMemsetAsync(s->d.pose_stamp, 0xff, sizeof(uint4) * static_cast<size_t>(s->W) * s->A, s->stream));
            s->reset_flagged(false);
            return;
        }
        // everything once from the raw tensors (the padding agents' rows never change between rebuilds); then every pass of the
        // step kernels writes the live agents' rows in place.  Every pose stamp dies: the next pass writes every live agent's
        // road columns, whatever was skipped before.
        s->d.pack = nullptr;
        s->d.pack_only = 0;
        HIP_CHECK(hipMemsetAsync(s->d.pose_stamp, 0xff, sizeof(uint4) * static_cast<size_t>(s->W) * s->A, s->stream));
        s->reset_flagged(false);               // raw tensors up to date (a previous pack_only attachment left them stale)
        gd::launch_pack_obs(s->d, s->stream, out);
        HIP_CHECK(hipGetLastError());
        s->d.pack = out;
        s->d.pack_only = only ? 1 : 0;
    });
}

int gd_expert_actions(gd_sim *s, float *actions, int32_t action_cols, float *pos_xy, float *vel_xy, float *yaw,
                      int32_t *valids) {
    if (!s) return fail(GD_ERR_INVALID, "gd_expert_actions: null sim");
    const int cols = s->d.p.dynamicsModel == GD_DYNAMICS_STATE ? 10 : 3;
    if (actions && action_cols != cols)
        return fail(GD_ERR_INVALID, "gd_expert_actions: action_cols does not match the dynamics model (10 for State, else 3)");
    return guarded([&]() {
        gd::launch_expert_actions(s->d, s->stream, actions, pos_xy, vel_xy, yaw, valids);
        HIP_CHECK(hipGetLastError());
    });
}

int gd_advance_log_playback(gd_sim *s, int32_t init_steps) {
    if (!s) return fail(GD_ERR_INVALID, "gd_advance_log_playback: null sim");
    if (init_steps < 0 || init_steps >= GD_EPISODE_LEN)
        return fail(GD_ERR_INVALID, "gd_advance_log_playback: the expert trajectory has 91 steps, init_steps must be < 91");
    return guarded([&]() {
        for (int t = 0; t < init_steps; t++) {
            gd::launch_set_log_actions(s->d, s->stream, t);
            HIP_CHECK(hipGetLastError());
            s->step();
        }
    });
}

int gd_episode_step(gd_sim *s, const gd_episode_config *cfg, const gd_episode_buffers *b) {
    if (!s || !cfg || !b) return fail(GD_ERR_INVALID, "gd_episode_step: null argument");
    if (!b->controlled_mask || !b->agent_episode_returns || !b->episode_lengths || !b->collided_in_episode ||
        !b->offroad_in_episode || !b->live_agent_mask || !b->reward_out || !b->terminal_out || !b->truncated_out ||
        !b->mask_out || !b->done_worlds || !b->stats || !b->world_stats)
        return fail(GD_ERR_INVALID, "gd_episode_step: every buffer is required");
    if (cfg->reward_type != GD_EPISODE_REWARD_WEIGHTED && cfg->reward_type != GD_EPISODE_REWARD_SPARSE)
        return fail(GD_ERR_INVALID, "gd_episode_step: unknown reward_type");
    return guarded([&]() {
        HIP_CHECK(hipMemsetAsync(s->d.any_reset, 0, sizeof(int32_t), s->stream));
        gd::launch_episode_step(s->d, s->stream, *cfg, *b);
        HIP_CHECK(hipGetLastError());
        if (cfg->auto_reset) s->reset_flagged(true);
    });
}

int gd_sync(gd_sim *s) {
    if (!s) return fail(GD_ERR_INVALID, "gd_sync: null sim");
    return guarded([&]() { HIP_CHECK(hipStreamSynchronize(s->stream)); });
}

int gd_set_stream(gd_sim *s, void *stream) {
    if (!s) return fail(GD_ERR_INVALID, "gd_set_stream: null sim");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->drop_graph();
        HIP_CHECK(hipStreamSynchronize(s->side));
        s->stream = static_cast<hipStream_t>(stream);
        s->d.split_partner = (s->stream != nullptr && std::getenv("GPUDRIVE_SPLIT_PARTNER") != nullptr) ? 1 : 0;
    });
}

int gd_attach_bev(gd_sim *s, float *bev) {
    if (!s || !bev) return fail(GD_ERR_INVALID, "gd_attach_bev: null argument");
    return guarded([&]() {
        if (s->d.bev && s->d.bev != bev) throw std::runtime_error("gd_attach_bev: a BEV tensor is already attached");
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->drop_graph();
        s->exported[GD_T_BEV] = bev;
        s->d.bev = bev;
        HIP_CHECK(hipMemsetAsync(s->d.bev_dirty, 1, sizeof(int32_t) * static_cast<size_t>(s->W) * s->A, s->stream));
        if (!s->params.disableClassicalObs) s->launch(gd::KERNEL_BEV, false);  // the rasters of the current state
    });
}

int gd_stat(gd_sim *s, int32_t which, int64_t *out) {
#ifdef GD_CLOCKS
    if (s && out && which >= 22 && which <= 29) {  // the set-order selection's phase clocks (map_obs.hip g_set_clk): 22 reads and zeroes all
        static unsigned long long clk[8];
        if (which == 22) gd::set_clocks_read(clk);
        *out = (int64_t)clk[which - 22];
        return GD_OK;
    }
#endif
#ifdef GD_CLOCKS
    if (s && out && which >= 32 && which <= 43) {  // k_world_step's phase clocks and counters (kernels.hip g_step_clk / g_step_cnt): 32 reads and zeroes all
        static unsigned long long clk[12];
        if (which == 32) gd::step_clocks_read(clk);
        *out = (int64_t)clk[which - 32];
        return GD_OK;
    }
#endif
#if defined(GD_DIAG) || defined(GD_CLOCKS)
    constexpr int32_t kLastStat = 20;
#else
    constexpr int32_t kLastStat = 7;
#endif
    if (s && out && which == 30) {  // agents whose road rows were left in place (pose unchanged) since the last read
        std::vector<unsigned long long> v(GD_SKIP_SLOTS, 0);
        (void)hipStreamSynchronize(s->stream);
        if (hipMemcpy(v.data(), s->d.stat_skipped, sizeof(unsigned long long) * v.size(), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemset(s->d.stat_skipped, 0, sizeof(unsigned long long) * v.size()) != hipSuccess)
            return fail(GD_ERR_DEVICE, "gd_stat: reading the skip counters failed");
        unsigned long long sum = 0;
        for (unsigned long long x : v) sum += x;
        *out = static_cast<int64_t>(sum) + s->host_skipped;
        s->host_skipped = 0;
        return GD_OK;
    }
    if (s && out && which == 31) {  // BEV rasters painted by the last pass that rasterised (bev_lidar.hip k_bev_list's count)
        int32_t c = 0;
        (void)hipStreamSynchronize(s->stream);
        if (hipMemcpy(&c, s->d.bev_count, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(GD_ERR_DEVICE, "gd_stat: reading the raster count failed");
        *out = c;
        return GD_OK;
    }
    if (s && out && which == 44) {  // agents whose LiDAR returns the last pass marked for tracing (the others were left in place)
        std::vector<int32_t> f(static_cast<size_t>(s->W) * s->A), n(static_cast<size_t>(s->W) * 2);
        (void)hipStreamSynchronize(s->stream);
        if (hipMemcpy(f.data(), s->d.lidar_dirty, sizeof(int32_t) * f.size(), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(n.data(), s->d.shape, sizeof(int32_t) * n.size(), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(GD_ERR_DEVICE, "gd_stat: reading the LiDAR flags failed");
        int64_t c = 0;
        for (int w = 0; w < s->W; w++)
            for (int a = 0; a < n[static_cast<size_t>(w) * 2] && a < s->A; a++) c += f[static_cast<size_t>(w) * s->A + a] != 0;
        *out = c;
        return GD_OK;
    }
    if (s && out && which == 21) {  // bounds audit of the rank path (engine.hpp GD_RANK_AUDIT): violations since the buffers exist
        *out = 0;
        if (s->rk_alloc) {
            int32_t v = 0;
            (void)hipStreamSynchronize(s->stream);
            if (hipMemcpy(&v, s->d.rk_hist + GD_RANK_AUDIT, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
                return fail(GD_ERR_DEVICE, "gd_stat: reading the audit counter failed");
            *out = v;
        }
        return GD_OK;
    }
    if (!s || !out || which < 0 || which > kLastStat) return fail(GD_ERR_INVALID, "gd_stat: bad argument");
#if defined(GD_DIAG) || defined(GD_CLOCKS)
    if (which >= 8) {  // 8 = most crowded ranking bucket (-DGD_DIAG, GPUDRIVE_RANK_DBG=9), 10..17 = clock ticks / 256 per phase of
                       // k_knn_rank summed over its waves, 18..20 = k_knn_replay's rounds of its first wave / candidates beyond K /
                       // inserts (-DGD_CLOCKS); since the last read
        *out = 0;
        if (s->rk_alloc) {
            int32_t v = 0;
            (void)hipStreamSynchronize(s->stream);
            (void)hipMemcpy(&v, s->d.rk_hist + 514 + (which - 8), sizeof(v), hipMemcpyDeviceToHost);
            (void)hipMemset(s->d.rk_hist + 514 + (which - 8), 0, sizeof(v));
            *out = v;
        }
        return GD_OK;
    }
#endif
    if (which == 7) {  // 1: the reference-order road selection takes the rank replay (map_obs_rank.hip)
        *out = s->d.rk_on;
        return GD_OK;
    }
    *out = which == 0 ? s->stat_graph_steps : which == 1 ? s->stat_plain_steps : which == 2 ? s->stat_captures
         : which == 3 ? s->d.set_fused_rows : which == 4 ? s->d.set_apw : which == 5 ? s->d.live_count : GD_MAP_OBS_AW;
    return GD_OK;
}

int gd_kernel_timing_enable(gd_sim *s, int32_t enable) {
    if (!s) return fail(GD_ERR_INVALID, "null sim");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        for (int k = 0; k < gd::KERNEL_TIMED; k++) {
            s->collect_timing(k);
            s->ev_ms[k] = 0;
            s->ev_launches[k] = 0;
            while (enable && s->ev_pool[k].size() < gd_sim::kEvRing) {  // every event exists before the first timed launch
                EventPair n{};
                HIP_CHECK(hipEventCreate(&n.start));
                HIP_CHECK(hipEventCreate(&n.stop));
                s->ev_pool[k].push_back(n);
            }
        }
        s->timing = enable != 0;
    });
}

int gd_kernel_timing_read(gd_sim *s, int32_t kernel, double *total_ms, int64_t *launches) {
    if (!s || kernel < 0 || kernel >= gd::KERNEL_TIMED) return fail(GD_ERR_INVALID, "gd_kernel_timing_read: bad argument");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->collect_timing(kernel);
        if (total_ms) *total_ms = s->ev_ms[kernel];
        if (launches) *launches = s->ev_launches[kernel];
    });
}

int gd_debug_get_state(gd_sim *s, float *out) {
    if (!s || !out) return fail(GD_ERR_INVALID, "null argument");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        const size_t WA = static_cast<size_t>(s->W) * s->A;
        std::vector<float> plane(WA);
        std::vector<int32_t> iplane(WA);
        const float *src[8] = {s->d.px, s->d.py, s->d.pz, s->d.qw, s->d.qz, s->d.vx, s->d.vy, s->d.vz};
        const int dstcol[8] = {0, 1, 2, 3, 6, 7, 8, 9};
        for (size_t i = 0; i < WA * 11; i++) out[i] = 0.f;
        for (int k = 0; k < 8; k++) {
            HIP_CHECK(hipMemcpy(plane.data(), src[k], WA * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < WA; i++) out[i * 11 + dstcol[k]] = plane[i];
        }
        for (size_t i = 0; i < WA; i++) {  // x, y = 0 * z as produced by angleAxis
            out[i * 11 + 4] = 0.f * out[i * 11 + 6];
            out[i * 11 + 5] = 0.f * out[i * 11 + 6];
        }
        HIP_CHECK(hipMemcpy(iplane.data(), s->d.collided, WA * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < WA; i++) out[i * 11 + 10] = static_cast<float>(iplane[i]);
    });
}

int gd_debug_road_path(gd_sim *s, int32_t *out) {
    if (!s || !out) return fail(GD_ERR_INVALID, "null argument");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        const size_t WA = static_cast<size_t>(s->W) * s->A;
        if (!s->rk_alloc || !s->d.rk_on) {
            for (size_t i = 0; i < WA; i++) out[i] = -2;
            return;
        }
        HIP_CHECK(hipMemcpy(out, s->d.rk_n, WA * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<int32_t> why(WA);
        HIP_CHECK(hipMemcpy(why.data(), s->d.rk_ticket, WA * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < WA; i++) {
            if (out[i] > 0) out[i] = out[i] == (1 << 30) ? -3 : (out[i] & 0xffff);
            else if (out[i] == -1 && why[i] < -1) out[i] = -10 + (why[i] + 2);  // -10 no checkpoints / small world, -11 overflow, -13 bypass
            else if (out[i] == -1 && why[i] >= 0 && ((why[i] >> 30) & 1)) out[i] = -12;  // more than 32 candidates with one key
        }
    });
}

int gd_debug_set_state(gd_sim *s, const float *in) {
    if (!s || !in) return fail(GD_ERR_INVALID, "null argument");
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->stream));
        const size_t WA = static_cast<size_t>(s->W) * s->A;
        std::vector<float> plane(WA);
        std::vector<int32_t> iplane(WA);
        float *dst[8] = {s->d.px, s->d.py, s->d.pz, s->d.qw, s->d.qz, s->d.vx, s->d.vy, s->d.vz};
        const int srccol[8] = {0, 1, 2, 3, 6, 7, 8, 9};
        for (int k = 0; k < 8; k++) {
            for (size_t i = 0; i < WA; i++) plane[i] = in[i * 11 + srccol[k]];
            HIP_CHECK(hipMemcpy(dst[k], plane.data(), WA * 4, hipMemcpyHostToDevice));
        }
        for (size_t i = 0; i < WA; i++) iplane[i] = in[i * 11 + 10] != 0.f;
        HIP_CHECK(hipMemcpy(s->d.collided, iplane.data(), WA * 4, hipMemcpyHostToDevice));
        // agents that never move are not on the linear scan's step-pass list: the next road pass must visit them all the same
        // ... and k_world_step's "who moved" (the BEV's dirty flags) compares the poses before and after its own movement only
        s->full_pass_next = (s->params.roadObservationAlgorithm != GD_ROADS_K_NEAREST && s->d.lin_on != 0) || s->d.bev != nullptr ||
                            (s->d.lidar != nullptr && s->params.enableLidar);
    });
}

int gd_host_world_build(const char *scene, const gd_params *params, int32_t A, const int32_t *deleted, int32_t ndel,
                        gd_host_world *out) {
    if (!scene || !params || !out || A < 2 || A > GD_MAX_AGENTS_LIMIT) return fail(GD_ERR_INVALID, "gd_host_world_build: bad argument");
    std::memset(out, 0, sizeof(*out));
    return guarded([&]() {
        auto map = gd::load_scene(scene, params->polylineReductionThreshold);
        gd::HostWorld hw;
        gd::build_host_world(*map, *params, A, deleted, ndel, hw);
        out->num_agents = hw.num_agents;
        out->num_roads = hw.num_roads;
        out->num_collidable_roads = static_cast<int32_t>(hw.boxes.size());
        out->max_agents = A;
        std::memcpy(out->mean, hw.mean, sizeof(hw.mean));
        std::memcpy(out->map_name, hw.map_name, sizeof(hw.map_name));
        std::memcpy(out->scenario_id, hw.scenario_id, sizeof(hw.scenario_id));
        auto dupf = [](const std::vector<float> &v, size_t n) {
            float *p = static_cast<float *>(std::calloc(std::max<size_t>(n, 1), sizeof(float)));
            std::memcpy(p, v.data(), std::min(n, v.size()) * sizeof(float));
            return p;
        };
        auto dupi = [](const std::vector<int32_t> &v) {
            int32_t *p = static_cast<int32_t *>(std::calloc(std::max<size_t>(v.size(), 1), sizeof(int32_t)));
            std::memcpy(p, v.data(), v.size() * sizeof(int32_t));
            return p;
        };
        out->map_obs = dupf(hw.map_obs, static_cast<size_t>(GD_MAX_ROAD_ENTITIES) * 9);
        for (int r = hw.num_roads; r < GD_MAX_ROAD_ENTITIES; r++) { out->map_obs[r * 9 + 7] = -1.f; out->map_obs[r * 9 + 8] = -1.f; }
        out->trajectory = dupf(hw.trajectory, hw.trajectory.size());
        out->vehicle_size = dupf(hw.size, hw.size.size());
        out->goal = dupf(hw.goal, hw.goal.size());
        out->controlled = dupi(hw.controlled);
        out->response_type = dupi(hw.resp);
        out->agent_id = dupi(hw.agent_id);
        out->entity_type = dupi(hw.etype);
        out->metadata = dupi(hw.metadata);
    });
}

int gd_scene_cache_write(const char *scene, float polyline_reduction_threshold, const char *out_path) {
    if (!scene || !out_path) return fail(GD_ERR_INVALID, "gd_scene_cache_write: null argument");
    if (!gd::is_scene_cache_path(out_path)) return fail(GD_ERR_INVALID, "gd_scene_cache_write: the cache path must end in .gdsm");
    return guarded([&]() {
        auto map = gd::load_scene(scene, polyline_reduction_threshold);
        gd::write_scene_cache(*map, polyline_reduction_threshold, out_path);
    });
}

void gd_host_world_free(gd_host_world *w) {
    if (!w) return;
    std::free(w->map_obs); std::free(w->trajectory); std::free(w->vehicle_size); std::free(w->goal);
    std::free(w->controlled); std::free(w->response_type); std::free(w->agent_id); std::free(w->entity_type);
    std::free(w->metadata);
    std::memset(w, 0, sizeof(*w));
}

}  // extern "C"
