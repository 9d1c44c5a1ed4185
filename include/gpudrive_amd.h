/*
 * gpudrive_amd.h -- C ABI of the MI355X-native GPUDrive step engine (libgpudrive_amd.so).
 *
 * This is the drop-in boundary for ONE path of CILAB-MA/gpudrive_lab: the batched per-world
 * simulation step behind `madrona_gpudrive.SimManager`.  Every entry point names the reference
 * interface it replaces (paths relative to the reference checkout).  Plain pointers and sizes
 * only; no torch / Python types.  All functions return GD_OK (0) or a negative error code and
 * never abort the process; gd_last_error() returns the message of the calling thread's last
 * failure.
 *
 * Buffers: every exported tensor lives in device (HBM) memory for the lifetime of the sim.
 * Either the caller hands in device pointers (gd_config.external[...], e.g. torch-allocated
 * storage so that `.to_torch()` is a zero-copy alias) or the engine allocates them itself.
 * The exported buffers ARE the live simulation storage, exactly like Madrona's exported ECS
 * columns: actions are written in place by the caller (gpudrive/env/env_torch.py:645-664) and
 * `controlled_state`, `done`, `expert_trajectory` ... are read back by the next step.
 */
#ifndef GPUDRIVE_AMD_H
#define GPUDRIVE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_OK 0
#define GD_ERR_INVALID (-1)   /* bad argument */
#define GD_ERR_IO (-2)        /* scene file missing / unreadable (reference: assert, src/MapReader.cpp:40) */
#define GD_ERR_PARSE (-3)     /* malformed scene JSON (reference: nlohmann exception -> abort) */
#define GD_ERR_DEVICE (-4)    /* HIP runtime failure, no gfx950 device */
#define GD_ERR_UNSUPPORTED (-5)

/* src/consts.hpp:11-13,34,37 */
#define GD_MAX_AGENTS_LIMIT 128
#define GD_MAX_ROAD_ENTITIES 10000
#define GD_MAP_OBS_K 200
#define GD_EPISODE_LEN 91
#define GD_NUM_LIDAR_SAMPLES 50
#define GD_TRAJECTORY_FLOATS 1456   /* src/types.hpp:373 */
#define GD_BEV_RES 200
#define GD_VEHICLE_SCALE 0.7f       /* src/consts.hpp:25 */

/* src/init.hpp:76-109 enums (values as seen through src/bindings.cpp:31-88) */
enum { GD_REWARD_DISTANCE_BASED = 0, GD_REWARD_ON_GOAL_ACHIEVED = 1, GD_REWARD_DENSE = 2 };
enum { GD_COLLISION_AGENT_STOP = 0, GD_COLLISION_AGENT_REMOVED = 1, GD_COLLISION_IGNORE = 2 };
enum { GD_DYNAMICS_CLASSIC = 0, GD_DYNAMICS_INVERTIBLE_BICYCLE = 1, GD_DYNAMICS_DELTA_LOCAL = 2,
       GD_DYNAMICS_STATE = 3 };
enum { GD_ROADS_K_NEAREST = 0, GD_ROADS_ALL_WITHIN_RADIUS = 1 };

/* src/init.hpp:111-127 `Parameters` (+ RewardParams :83-88), field for field. */
typedef struct gd_params {
    float polylineReductionThreshold;
    float observationRadius;
    int32_t rewardType;
    float distanceToGoalThreshold;
    float distanceToExpertThreshold;
    int32_t collisionBehaviour;          /* default AgentStop */
    uint32_t maxNumControlledAgents;     /* default 10000 */
    int32_t IgnoreNonVehicles;           /* default 0 */
    int32_t roadObservationAlgorithm;    /* default K nearest */
    int32_t initOnlyValidAgentsAtFirstStep; /* default 1 */
    int32_t isStaticAgentControlled;     /* default 0 */
    int32_t enableLidar;                 /* default 0 */
    int32_t disableClassicalObs;         /* default 0 */
    int32_t dynamicsModel;               /* default Classic */
    int32_t readFromTracksToPredict;     /* default 0 */
} gd_params;

/* Export slots: src/sim.hpp:17-45 `ExportID`, in the order of the getters of src/bindings.cpp:109-149. */
enum {
    GD_T_ACTION = 0,          /* action_tensor                   f32 [W,A,10]        mgr.cpp:718 */
    GD_T_REWARD,              /* reward_tensor                   f32 [W,A,1]         mgr.cpp:729 */
    GD_T_DONE,                /* done_tensor                     i32 [W,A,1]         mgr.cpp:749 */
    GD_T_INFO,                /* info_tensor                     i32 [W,A,5]         mgr.cpp:759 */
    GD_T_SELF_OBS,            /* self_observation_tensor         f32 [W,A,8]         mgr.cpp:769 */
    GD_T_ABS_OBS,             /* absolute_self_observation_tensor f32 [W,A,14]       mgr.cpp:865 */
    GD_T_PARTNER_OBS,         /* partner_observations_tensor     f32 [W,A,A-1,9]     mgr.cpp:792 */
    GD_T_AGENT_MAP_OBS,       /* agent_roadmap_tensor            f32 [W,A,200,9]     mgr.cpp:804 */
    GD_T_MAP_OBS,             /* map_observation_tensor          f32 [W,10000,9]     mgr.cpp:780 */
    GD_T_LIDAR,               /* lidar_tensor                    f32 [W,A,3,50,4]    mgr.cpp:817 */
    GD_T_BEV,                 /* bev_observation_tensor          f32 [W,A,200,200,1] mgr.cpp:829 (lazy) */
    GD_T_STEPS_REMAINING,     /* steps_remaining_tensor          i32 [W,A,1]         mgr.cpp:841 */
    GD_T_SHAPE,               /* shape_tensor                    i32 [W,2]           mgr.cpp:852 */
    GD_T_CONTROLLED_STATE,    /* controlled_state_tensor         i32 [W,A,1]         mgr.cpp:857 */
    GD_T_RESPONSE_TYPE,       /* response_type_tensor            i32 [W,A,1]         mgr.cpp:861 */
    GD_T_EXPERT_TRAJECTORY,   /* expert_trajectory_tensor        f32 [W,A,1456]      mgr.cpp:877 */
    GD_T_WORLD_MEANS,         /* world_means_tensor              f32 [W,3]           mgr.cpp:739 */
    GD_T_METADATA,            /* metadata_tensor                 i32 [W,A,4]         mgr.cpp:897 */
    GD_T_DELETED_AGENTS,      /* deleted_agents_tensor           i32 [W,A]           mgr.cpp:656 */
    GD_T_MAP_NAME,            /* map_name_tensor                 i32 [W,32]          mgr.cpp:883 */
    GD_T_SCENARIO_ID,         /* scenario_id_tensor              i32 [W,32]          mgr.cpp:890 */
    GD_T_COUNT
};

enum { GD_DTYPE_F32 = 0, GD_DTYPE_I32 = 1 };

typedef struct gd_tensor_desc {
    void *data;        /* device pointer (NULL from gd_tensor_shape) */
    int32_t dtype;     /* GD_DTYPE_* */
    int32_t ndim;
    int64_t dims[5];
    int64_t nbytes;
} gd_tensor_desc;

/* How the k-NN road observation orders its rows. */
enum {
    GD_KNN_REFERENCE_ORDER = 0, /* rows in the reference's SGI-heap array order (src/knn.hpp:103-158) */
    GD_KNN_SET_ORDER = 1        /* same row SET, in a fixed order of the engine's own (grid cell by grid cell); NOT elementwise identical */
};

/* Manager::Config (src/mgr.hpp:30-44) minus the render fields, plus engine knobs. */
typedef struct gd_config {
    int32_t num_worlds;         /* = len(scenes) */
    int32_t max_agents;         /* consts::kMaxAgentCount: 64 (benchmark configs) or 128 (this fork) */
    int32_t device_id;          /* Config::gpuID */
    void *stream;               /* hipStream_t to launch on; NULL = the null stream */
    int32_t knn_order;          /* GD_KNN_* */
    int32_t alloc_bev;          /* 1: allocate + compute the BEV tensor (160 KB/agent) */
    float lidar_half_angle;     /* consts::lidarAngle; 0 -> pi/3 (reference), pi -> 360 degrees */
    void *external[GD_T_COUNT]; /* caller-owned device buffers per export slot, or NULL */
} gd_config;

typedef struct gd_sim gd_sim;

/* Library identity / capability probes (no device access). */
const char *gd_version(void);
const char *gd_last_error(void);
void gd_default_params(gd_params *out);                   /* src/init.hpp:111-127 defaults */

/* Shape/dtype of export slot `id` for (num_worlds, max_agents); data = NULL.  Lets the caller
 * allocate storage before gd_create.  Replaces the dims lists of src/mgr.cpp:656-902. */
int gd_tensor_shape(int32_t id, int32_t num_worlds, int32_t max_agents, gd_tensor_desc *out);

/* Manager::Manager (src/mgr.cpp:565; bindings.cpp:93-108): parse `scenes`, build every world,
 * upload, and run the Reset task graph once so that all tensors hold t = 0 state. */
int gd_create(const gd_config *cfg, const gd_params *params, const char *const *scenes, gd_sim **out);
/* Manager::~Manager */
void gd_destroy(gd_sim *sim);

/* Manager::step (src/mgr.cpp:569-580): movement -> collision -> reward -> --t -> done -> obs.
 * Launches on cfg.stream and returns without synchronising (stream order = data order). */
int gd_step(gd_sim *sim);
/* Manager::reset (src/mgr.cpp:582-588): flag the listed worlds, run the Reset graph for ALL worlds. */
int gd_reset(gd_sim *sim, const int32_t *world_indices, int32_t n);
/* Manager::setMaps (src/mgr.cpp:590-654): n must equal num_worlds. */
int gd_set_maps(gd_sim *sim, const char *const *scenes, int32_t n);
/* Manager::deleteAgents (src/mgr.cpp:665-715): CSR lists; ids[offsets[i]..offsets[i+1]) for worlds[i]. */
int gd_delete_agents(gd_sim *sim, const int32_t *worlds, const int32_t *offsets, const int32_t *ids,
                     int32_t n_worlds);
/* Manager::*Tensor() (src/mgr.cpp:656-902). */
int gd_tensor(gd_sim *sim, int32_t id, gd_tensor_desc *out);
/* Fused observation pack: writes what GPUDriveTorchEnv.get_obs() concatenates with
 * norm_obs=True (gpudrive/env/env_torch.py:756-896,1172-1216) for every agent slot:
 * out[W][A][6 + (A-1)*6 + 200*13] f32 = ego | partners | road points (type one-hot over 7).
 * `out` is a device pointer of at least out_bytes bytes. */
int gd_pack_observations(gd_sim *sim, float *out, int64_t out_bytes);
/* The same tensor written WHERE THE ROWS ARE PRODUCED instead of by a second pass over the exported tensors: from this call on
 * every step / reset pass writes the live agents' packed rows straight into `out` (k_world_step: ego + partner columns; the
 * road kernel: the 200 x 13 road columns), bit-identical to gd_pack_observations on the same state; `out` must stay valid
 * until it is detached (out = NULL) or the simulator is destroyed.  only != 0: the raw partner_observations and
 * agent_roadmap rows of live agents are no longer written (for a learner that reads nothing but the packed tensor --
 * gpudrive/env/env_torch.py:756-896 is the only consumer of those rows in the reference's PPO loop); only = 0 keeps them.
 * Every road path writes them (the linear scan, the fused set-order kernel, k_map_rows behind the reference-order
 * selections); GD_ERR_UNSUPPORTED only with disableClassicalObs or the developer switch GPUDRIVE_LINEAR_LEGACY=1.  With a
 * buffer attached gd_pack_observations is a no-op for that buffer and a device copy for any other. */
int gd_attach_packed(gd_sim *sim, float *out, int64_t out_bytes, int32_t only);
/* Expert-action export: GPUDriveTorchEnv.get_expert_actions()
 * (gpudrive/env/env_torch.py:1445-1509 over gpudrive/datatypes/trajectory.py:24-41) in one pass over the
 * expert trajectory rows.  Device pointers, any of them may be NULL:
 *   actions f32 [W][A][91][cols]  cols = 10 for DynamicsModel::State, else 3 (clamped per model)
 *   pos_xy  f32 [W][A][91][2], vel_xy f32 [W][A][91][2], yaw f32 [W][A][91][1], valids i32 [W][A][91][1]
 * `action_cols` must match the simulator's dynamics model (GD_ERR_INVALID otherwise). */
int gd_expert_actions(gd_sim *sim, float *actions, int32_t action_cols, float *pos_xy, float *vel_xy, float *yaw,
                      int32_t *valids);
/* GPUDriveTorchEnv.advance_sim_with_log_playback(init_steps) (gpudrive/env/env_torch.py:1274-1293): for
 * t = 0 .. init_steps-1 write the expert action of step t into action[:, :, :cols] of every agent slot
 * (env_torch.py:645-664) and step.  init_steps >= 91 is GD_ERR_INVALID (the reference raises ValueError). */
int gd_advance_log_playback(gd_sim *sim, int32_t init_steps);
/* Episode bookkeeping on the device: PufferGPUDrive.step()'s tracking of live agents,
 * episode returns / lengths / collision and off-road counts, finished worlds and their asynchronous reset
 * (gpudrive/env/env_puffer.py:250-403; rewards gpudrive/env/env_torch.py:469-505) without a host round trip.
 * Call after gd_step.  All pointers are device pointers owned by the caller, [W][A] unless noted. */
enum { GD_EPISODE_REWARD_WEIGHTED = 0,  /* "weighted_combination": cw*collided + gw*goal_achieved + ow*off_road */
       GD_EPISODE_REWARD_SPARSE = 1 };  /* "sparse_on_goal_achieved": the simulator's reward tensor */
enum { GD_EPISODE_STAT_EPISODES = 0,    /* finished worlds */
       GD_EPISODE_STAT_FINISHED_AGENTS, /* controlled agents in them */
       GD_EPISODE_STAT_RETURN_SUM,      /* sum of agent_episode_returns over those agents */
       GD_EPISODE_STAT_OFF_ROAD_AGENTS, /* agents with offroad_in_episode > 0 */
       GD_EPISODE_STAT_COLLIDED_AGENTS, /* agents with collided_in_episode > 0 */
       GD_EPISODE_STAT_GOAL_ACHIEVED,   /* sum of info.goal_achieved */
       GD_EPISODE_STAT_TRUNCATED_AGENTS,
       GD_EPISODE_STAT_LENGTH_SUM,      /* sum of episode_lengths over ALL slots of the finished worlds */
       GD_EPISODE_STAT_TOTAL_COLLISIONS, GD_EPISODE_STAT_TOTAL_OFF_ROAD, /* sums over all slots */
       GD_EPISODE_STATS = 12 };
typedef struct gd_episode_config {
    float collision_weight, goal_achieved_weight, off_road_weight;
    int32_t reward_type;  /* GD_EPISODE_REWARD_* */
    int32_t auto_reset;   /* raise the reset flag of finished worlds and reset them (resetSystem + observations) */
} gd_episode_config;
typedef struct gd_episode_buffers {
    const uint8_t *controlled_mask;  /* cont_agent_mask captured at t = 0 (bool) */
    /* running state, read and written */
    float *agent_episode_returns, *episode_lengths, *collided_in_episode, *offroad_in_episode;
    uint8_t *live_agent_mask;
    /* per-step outputs */
    float *reward_out;
    uint8_t *terminal_out, *truncated_out, *mask_out;
    int32_t *done_worlds;  /* [W] 1 for worlds whose episode ended in this step */
    float *stats;          /* [GD_EPISODE_STATS] running sums over finished episodes (the caller zeroes them) */
    float *world_stats;    /* [W][GD_EPISODE_STATS] the same for the last finished episode of each world */
} gd_episode_buffers;
int gd_episode_step(gd_sim *sim, const gd_episode_config *cfg, const gd_episode_buffers *buffers);

/* Block until everything launched so far has finished (the reference's step() is synchronous). */
int gd_sync(gd_sim *sim);
/* Change the launch stream (e.g. torch's current stream). */
int gd_set_stream(gd_sim *sim, void *stream);
/* Attach the BEV tensor after construction (replaces the reference's always-on BevObservations export,
 * src/mgr.cpp:870-880 + src/sim.cpp:462-555: there the raster exists from the start; here it is created on the first
 * bev_observation_tensor() call).  `bev` is a device buffer of gd_tensor_shape(GD_T_BEV) floats that the
 * caller keeps alive.  The rasters of the current state are computed at once; every later step / reset refreshes them. */
int gd_attach_bev(gd_sim *sim, float *bev);
/* Engine counters (tests and diagnostics).  which: 0 = steps replayed from the captured hipGraph,
 * 1 = steps launched kernel by kernel, 2 = hipGraph captures; the schedule the engine chose for this batch (it never
 * changes a result): 3 = set-order road kernel stores its rows itself (0 / 1), 4 = its agents per wave, 5 = live agents,
 * 6 = agents per wave of the reference-order road kernel (compile-time GD_MAP_OBS_AW of this build), 7 = the
 * reference-order road selection takes the rank replay (0 / 1).  Developer builds (tools/build_expt.sh) add 8 = the most
 * crowded ranking bucket (-DGD_DIAG with GPUDRIVE_RANK_DBG=9) and 10..17 = clock ticks per phase of k_knn_rank, then
 * 18..20 = k_knn_replay's rounds of its first wave / candidates beyond K / inserts (-DGD_CLOCKS; tools/rank_spikes.py),
 * all since the last read.  Every build: 21 = accesses the rank replay's bounds audit found out of range since its buffers
 * exist (must stay 0), 30 = agents whose road rows were left in place because their pose bits had not changed, since the last
 * read, 31 = BEV rasters painted by the last pass that rasterised (the others could not have changed and were left in
 * place), 44 = agents whose LiDAR returns the last pass marked for tracing (likewise).  Otherwise GD_ERR_INVALID. */
int gd_stat(gd_sim *sim, int32_t which, int64_t *out);

/* Timing hooks for the bench: HIP events around the named kernel on the engine's stream (a fixed ring of event pairs,
 * created by the enable call; steps run kernel by kernel, not from the hipGraph, while it is on).  Enabling again
 * while enabled zeroes the sums.  kernel: 0 = state step, 1 = road observation, 2 = LiDAR, 3 = BEV,
 * 4 = partner rows (on the engine's second stream, beside the road observation; no launches when they are written by the
 * state step, which is the default; GPUDRIVE_SPLIT_PARTNER=1 moves them there). */
int gd_kernel_timing_enable(gd_sim *sim, int32_t enable);
int gd_kernel_timing_read(gd_sim *sim, int32_t kernel, double *total_ms, int64_t *launches);

/* Internal agent state for teacher-forced parity tests: 11 floats per agent slot
 * {pos xyz, quat wxyz, vel xyz, collided}.  Device-to-host / host-to-device copies. */
int gd_debug_get_state(gd_sim *sim, float *host_out);
int gd_debug_set_state(gd_sim *sim, const float *host_in);
/* Which kernel selected every agent slot's roads in the last reference-order selection: > 0 the rank replay
 * (map_obs_rank.hip; the value is the agent's candidate count), -1 the history replay on keys (k_map_obs, the fallback:
 * first selection after a map change, jumps, overflow; -1 because another agent of its group of 32 needed it, -10 no usable
 * checkpoints or a world too small for the rank path, -11 more candidates than the buffer holds, -12 more than 32 equal keys, -13 the group bypasses the rank kernels
 * after three fallbacks in a row and retries every 64th selection), -3 no road of the world within reach of the radius (no rows),
 * 0 no selection (padding slot), -2 the rank replay is not in use (set order, linear scan, GPUDRIVE_NO_RANK_REPLAY).  out: [W][A] int32 on the host. */
int gd_debug_road_path(gd_sim *sim, int32_t *out);

/* Host-only scene pipeline (no device): parse + build world `scene` exactly as gd_create would
 * and return the init-time rows, for CPU tests of the host logic.
 * Replaces MapReader::parseAndWriteOut + createPersistentEntities (src/MapReader.cpp:55-61,
 * src/level_gen.cpp:396-465).  Caller frees with gd_host_world_free. */
typedef struct gd_host_world {
    int32_t num_agents, num_roads, num_collidable_roads, max_agents;
    float mean[3];
    int32_t map_name[32], scenario_id[32];
    float *map_obs;          /* [10000][9] */
    float *trajectory;       /* [A][1456] */
    int32_t *controlled, *response_type, *agent_id, *entity_type, *metadata /* [A][4] */;
    float *vehicle_size;     /* [A][3] */
    float *goal;             /* [A][2] */
} gd_host_world;
int gd_host_world_build(const char *scene, const gd_params *params, int32_t max_agents,
                        const int32_t *deleted_ids, int32_t n_deleted, gd_host_world *out);
void gd_host_world_free(gd_host_world *w);

/* Binary scene cache: parse `scene` (JSON) once the way MapReader::parseAndWriteOut +
 * from_json(Map) do (src/MapReader.cpp:46-61, src/json_serialization.hpp) and write the parsed, polyline-reduced
 * map to `out_path`, which must end in ".gdsm".  A ".gdsm" path is accepted wherever a scene path is (gd_create,
 * gd_set_maps, gd_host_world_build) and loads ~100x faster than the JSON; the worlds built from it are bit-identical.
 * The threshold is stored: asking for another one later is GD_ERR_INVALID.  Host only, no device needed. */
int gd_scene_cache_write(const char *scene, float polyline_reduction_threshold, const char *out_path);

#ifdef __cplusplus
}
#endif
#endif /* GPUDRIVE_AMD_H */
