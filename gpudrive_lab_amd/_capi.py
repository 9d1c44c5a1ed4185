"""ctypes binding of include/gpudrive_amd.h.  There is no CPU fallback: if the HIP library is
missing or no gfx950 device is visible, calls fail loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgpudrive_amd.so")
# Developer experiments only (tools/build_expt.sh writes them to build/expt/, outside this package): another build of the
# library is loaded only when GPUDRIVE_DEV=1 says so as well -- a stray GPUDRIVE_AMD_LIB alone must never make a bench or
# a test run a diagnostic build.
if os.environ.get("GPUDRIVE_AMD_LIB"):
    if os.environ.get("GPUDRIVE_DEV") != "1":
        raise ImportError("GPUDRIVE_AMD_LIB is set but GPUDRIVE_DEV=1 is not: refusing to load a developer build of the "
                          "library (unset GPUDRIVE_AMD_LIB, or set GPUDRIVE_DEV=1 for an experiment)")
    _SO = os.environ["GPUDRIVE_AMD_LIB"]
_LIB = None

GD_OK = 0
GD_ERR_INVALID, GD_ERR_IO, GD_ERR_PARSE, GD_ERR_DEVICE, GD_ERR_UNSUPPORTED = -1, -2, -3, -4, -5

(T_ACTION, T_REWARD, T_DONE, T_INFO, T_SELF_OBS, T_ABS_OBS, T_PARTNER_OBS, T_AGENT_MAP_OBS, T_MAP_OBS,
 T_LIDAR, T_BEV, T_STEPS_REMAINING, T_SHAPE, T_CONTROLLED_STATE, T_RESPONSE_TYPE, T_EXPERT_TRAJECTORY,
 T_WORLD_MEANS, T_METADATA, T_DELETED_AGENTS, T_MAP_NAME, T_SCENARIO_ID, T_COUNT) = range(22)

DTYPE_F32, DTYPE_I32 = 0, 1


class GdParams(C.Structure):
    _fields_ = [
        ("polylineReductionThreshold", C.c_float),
        ("observationRadius", C.c_float),
        ("rewardType", C.c_int32),
        ("distanceToGoalThreshold", C.c_float),
        ("distanceToExpertThreshold", C.c_float),
        ("collisionBehaviour", C.c_int32),
        ("maxNumControlledAgents", C.c_uint32),
        ("IgnoreNonVehicles", C.c_int32),
        ("roadObservationAlgorithm", C.c_int32),
        ("initOnlyValidAgentsAtFirstStep", C.c_int32),
        ("isStaticAgentControlled", C.c_int32),
        ("enableLidar", C.c_int32),
        ("disableClassicalObs", C.c_int32),
        ("dynamicsModel", C.c_int32),
        ("readFromTracksToPredict", C.c_int32),
    ]


class GdTensorDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 5), ("nbytes", C.c_int64)]


class GdConfig(C.Structure):
    _fields_ = [("num_worlds", C.c_int32), ("max_agents", C.c_int32), ("device_id", C.c_int32),
                ("stream", C.c_void_p), ("knn_order", C.c_int32), ("alloc_bev", C.c_int32),
                ("lidar_half_angle", C.c_float), ("external", C.c_void_p * T_COUNT)]


class GdHostWorld(C.Structure):
    _fields_ = [("num_agents", C.c_int32), ("num_roads", C.c_int32), ("num_collidable_roads", C.c_int32),
                ("max_agents", C.c_int32), ("mean", C.c_float * 3), ("map_name", C.c_int32 * 32),
                ("scenario_id", C.c_int32 * 32), ("map_obs", C.POINTER(C.c_float)),
                ("trajectory", C.POINTER(C.c_float)), ("controlled", C.POINTER(C.c_int32)),
                ("response_type", C.POINTER(C.c_int32)), ("agent_id", C.POINTER(C.c_int32)),
                ("entity_type", C.POINTER(C.c_int32)), ("metadata", C.POINTER(C.c_int32)),
                ("vehicle_size", C.POINTER(C.c_float)), ("goal", C.POINTER(C.c_float))]


EPISODE_STATS = 12
EPISODE_STAT_NAMES = ("episodes", "finished_agents", "return_sum", "off_road_agents", "collided_agents", "goal_achieved",
                      "truncated_agents", "length_sum", "total_collisions", "total_off_road")


class GdEpisodeConfig(C.Structure):
    _fields_ = [("collision_weight", C.c_float), ("goal_achieved_weight", C.c_float), ("off_road_weight", C.c_float),
                ("reward_type", C.c_int32), ("auto_reset", C.c_int32)]


class GdEpisodeBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "controlled_mask", "agent_episode_returns", "episode_lengths", "collided_in_episode", "offroad_in_episode",
        "live_agent_mask", "reward_out", "terminal_out", "truncated_out", "mask_out", "done_worlds", "stats", "world_stats")]


# every symbol include/gpudrive_amd.h declares
SYMBOLS = [
    "gd_version", "gd_last_error", "gd_default_params", "gd_tensor_shape", "gd_create", "gd_destroy",
    "gd_step", "gd_reset", "gd_set_maps", "gd_delete_agents", "gd_tensor", "gd_pack_observations", "gd_attach_packed",
    "gd_expert_actions", "gd_advance_log_playback", "gd_episode_step", "gd_sync",
    "gd_set_stream", "gd_attach_bev", "gd_stat",
    "gd_kernel_timing_enable", "gd_kernel_timing_read", "gd_debug_get_state", "gd_debug_set_state", "gd_debug_road_path",
    "gd_host_world_build", "gd_host_world_free", "gd_scene_cache_write",
]


def lib_path():
    return _SO


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcdir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", srcdir]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise ImportError(
            "gpudrive_lab_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the step path." % _SO)
    # torch-ROCm bundles its own libamdhip64.so.7; load it first so that this library binds to the
    # SAME HIP runtime instance (two runtimes in one process cannot both own the device, and the
    # engine is handed torch-allocated device pointers).
    import torch  # noqa: F401
    L = C.CDLL(_SO)
    L.gd_version.restype = C.c_char_p
    L.gd_last_error.restype = C.c_char_p
    L.gd_default_params.argtypes = [C.POINTER(GdParams)]
    L.gd_default_params.restype = None
    L.gd_tensor_shape.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(GdTensorDesc)]
    L.gd_create.argtypes = [C.POINTER(GdConfig), C.POINTER(GdParams), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
    L.gd_destroy.argtypes = [C.c_void_p]
    L.gd_destroy.restype = None
    L.gd_step.argtypes = [C.c_void_p]
    L.gd_reset.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    L.gd_set_maps.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32]
    L.gd_delete_agents.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
    L.gd_tensor.argtypes = [C.c_void_p, C.c_int32, C.POINTER(GdTensorDesc)]
    L.gd_sync.argtypes = [C.c_void_p]
    L.gd_pack_observations.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.gd_attach_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.gd_attach_packed.restype = C.c_int
    L.gd_episode_step.argtypes = [C.c_void_p, C.POINTER(GdEpisodeConfig), C.POINTER(GdEpisodeBuffers)]
    L.gd_scene_cache_write.argtypes = [C.c_char_p, C.c_float, C.c_char_p]
    L.gd_expert_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gd_advance_log_playback.argtypes = [C.c_void_p, C.c_int32]
    L.gd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.gd_attach_bev.argtypes = [C.c_void_p, C.c_void_p]
    L.gd_stat.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    L.gd_kernel_timing_enable.argtypes = [C.c_void_p, C.c_int32]
    L.gd_kernel_timing_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.gd_debug_get_state.argtypes = [C.c_void_p, C.c_void_p]
    L.gd_debug_set_state.argtypes = [C.c_void_p, C.c_void_p]
    L.gd_debug_road_path.argtypes = [C.c_void_p, C.c_void_p]
    L.gd_host_world_build.argtypes = [C.c_char_p, C.POINTER(GdParams), C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                      C.POINTER(GdHostWorld)]
    L.gd_host_world_free.argtypes = [C.POINTER(GdHostWorld)]
    L.gd_host_world_free.restype = None
    _LIB = L
    return L


def check(rc, what="call"):
    if rc == GD_OK:
        return
    msg = lib().gd_last_error().decode("utf-8", "replace")
    if rc == GD_ERR_IO:
        raise FileNotFoundError(msg)
    if rc == GD_ERR_INVALID:
        raise ValueError(msg)
    if rc == GD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))
