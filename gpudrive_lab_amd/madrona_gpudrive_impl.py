"""Host-side mirror of the reference's `madrona_gpudrive` nanobind module (reference
src/bindings.cpp:14-152): same names, argument meaning and tensor shapes, running on the
MI355X-native engine through the C ABI of include/gpudrive_amd.h.

Exported tensors are torch CUDA(HIP) tensors allocated here and handed to the engine as raw
device pointers, so `.to_torch()` is a zero-copy alias of live simulator storage, like Madrona's
exported ECS columns (reference src/mgr.cpp:158-164,198-204).  Launches go to torch's current
stream: the reference's step() is synchronous (src/mgr.cpp:154-156); here stream order gives the same
data order for every torch consumer on that stream, `SimManager.sync()` blocks explicitly, and
`SimManager(..., sync=True)` or GPUDRIVE_SYNC_STEP=1 restores the reference's blocking step / reset /
set_maps for consumers on other streams or outside torch (DLPack, JAX).

Build-specific options that the reference's constructor does not have can also be given through the
environment, so that an unchanged `gpudrive.env.base_env` (which passes none of them,
gpudrive/env/base_env.py:176-190) reaches them: GPUDRIVE_MAX_AGENTS (64 | 128), GPUDRIVE_KNN_ORDER
(0 reference heap order | 1 set order), GPUDRIVE_LIDAR_HALF_ANGLE (radians; unset = the reference's
pi/3), GPUDRIVE_SYNC_STEP (1 = blocking), GPUDRIVE_DIRECT_PACK (1 | 2: the packed observation written by the step itself,
`SimManager.direct_pack`).  The BEV tensor needs no option: it is created on the first
`bev_observation_tensor()` call.
"""
import ctypes as C
import enum
import os

import numpy as np

from . import _capi

# ---- constants: reference src/bindings.cpp:23-28 ----
# kMaxAgentCount is a compile-time constant in the reference (128 in this fork, src/consts.hpp:11;
# 64 in the benchmark configs).  Here it is fixed at import time from GPUDRIVE_MAX_AGENTS.
kMaxAgentCount = int(os.environ.get("GPUDRIVE_MAX_AGENTS", "128"))
if kMaxAgentCount not in (64, 128):
    raise ImportError("GPUDRIVE_MAX_AGENTS must be 64 or 128")
kMaxRoadEntityCount = 10000
kMaxAgentMapObservationsCount = 200
episodeLen = 91
numLidarSamples = 50
vehicleScale = 0.7


# ---- enums: reference src/bindings.cpp:31-88 ----
class RewardType(enum.IntEnum):
    DistanceBased = 0
    OnGoalAchieved = 1
    Dense = 2


class FindRoadObservationsWith(enum.IntEnum):
    KNearestEntitiesWithRadiusFiltering = 0
    AllEntitiesWithRadiusFiltering = 1


class CollisionBehaviour(enum.IntEnum):
    AgentStop = 0
    AgentRemoved = 1
    Ignore = 2


class DynamicsModel(enum.IntEnum):
    Classic = 0
    InvertibleBicycle = 1
    DeltaLocal = 2
    State = 3


class EntityType(enum.IntEnum):
    _None = 0
    RoadEdge = 1
    RoadLine = 2
    RoadLane = 3
    CrossWalk = 4
    SpeedBump = 5
    StopSign = 6
    Vehicle = 7
    Pedestrian = 8
    Cyclist = 9
    Padding = 10
    NumTypes = 11


class ExecMode(enum.IntEnum):
    """madrona.ExecMode.  `CUDA` is the device path; on this build that is HIP on gfx950 (torch-ROCm
    also calls the device "cuda", gpudrive/env/base_env.py:170-174)."""
    CPU = 0
    CUDA = 1


class RewardParams:
    """reference src/init.hpp:83-88; default-constructible, read/write fields."""

    def __init__(self):
        self.rewardType = RewardType.DistanceBased
        self.distanceToGoalThreshold = 0.0
        self.distanceToExpertThreshold = 0.0


class Parameters:
    """reference src/init.hpp:111-127 via src/bindings.cpp:48-62."""

    def __init__(self):
        self.polylineReductionThreshold = 0.0
        self.observationRadius = 0.0
        self.rewardParams = RewardParams()
        self.collisionBehaviour = CollisionBehaviour.AgentStop
        self.maxNumControlledAgents = 10000
        self.IgnoreNonVehicles = False
        self.roadObservationAlgorithm = FindRoadObservationsWith.KNearestEntitiesWithRadiusFiltering
        self.initOnlyValidAgentsAtFirstStep = True
        self.isStaticAgentControlled = False
        self.enableLidar = False
        self.disableClassicalObs = False
        self.dynamicsModel = DynamicsModel.Classic
        self.readFromTracksToPredict = False

    def _to_c(self):
        p = _capi.GdParams()
        p.polylineReductionThreshold = float(self.polylineReductionThreshold)
        p.observationRadius = float(self.observationRadius)
        p.rewardType = int(self.rewardParams.rewardType)
        p.distanceToGoalThreshold = float(self.rewardParams.distanceToGoalThreshold)
        p.distanceToExpertThreshold = float(self.rewardParams.distanceToExpertThreshold)
        p.collisionBehaviour = int(self.collisionBehaviour)
        p.maxNumControlledAgents = int(self.maxNumControlledAgents)
        p.IgnoreNonVehicles = int(bool(self.IgnoreNonVehicles))
        p.roadObservationAlgorithm = int(self.roadObservationAlgorithm)
        p.initOnlyValidAgentsAtFirstStep = int(bool(self.initOnlyValidAgentsAtFirstStep))
        p.isStaticAgentControlled = int(bool(self.isStaticAgentControlled))
        p.enableLidar = int(bool(self.enableLidar))
        p.disableClassicalObs = int(bool(self.disableClassicalObs))
        p.dynamicsModel = int(self.dynamicsModel)
        p.readFromTracksToPredict = int(bool(self.readFromTracksToPredict))
        return p


class Tensor:
    """madrona.Tensor: a non-owning view of engine storage (reference madrona::py::Tensor)."""

    def __init__(self, torch_tensor):
        self._t = torch_tensor

    def to_torch(self):
        return self._t

    def to_jax(self):
        try:
            import jax.dlpack as jdl  # noqa: F401
        except Exception as e:  # pragma: no cover - jax is not in this image
            raise ImportError("jax is not installed; use .to_torch()") from e
        import jax
        return jax.dlpack.from_dlpack(self._t)

    @property
    def shape(self):
        return tuple(self._t.shape)


_SLOT_GETTERS = {
    "action_tensor": _capi.T_ACTION,
    "reward_tensor": _capi.T_REWARD,
    "done_tensor": _capi.T_DONE,
    "info_tensor": _capi.T_INFO,
    "self_observation_tensor": _capi.T_SELF_OBS,
    "absolute_self_observation_tensor": _capi.T_ABS_OBS,
    "partner_observations_tensor": _capi.T_PARTNER_OBS,
    "agent_roadmap_tensor": _capi.T_AGENT_MAP_OBS,
    "map_observation_tensor": _capi.T_MAP_OBS,
    "lidar_tensor": _capi.T_LIDAR,
    "steps_remaining_tensor": _capi.T_STEPS_REMAINING,
    "shape_tensor": _capi.T_SHAPE,
    "controlled_state_tensor": _capi.T_CONTROLLED_STATE,
    "response_type_tensor": _capi.T_RESPONSE_TYPE,
    "expert_trajectory_tensor": _capi.T_EXPERT_TRAJECTORY,
    "world_means_tensor": _capi.T_WORLD_MEANS,
    "metadata_tensor": _capi.T_METADATA,
    "deleted_agents_tensor": _capi.T_DELETED_AGENTS,
    "map_name_tensor": _capi.T_MAP_NAME,
    "scenario_id_tensor": _capi.T_SCENARIO_ID,
}


class SimManager:
    """reference src/bindings.cpp:91-149 (`Manager`, src/mgr.hpp:27-100)."""

    def __init__(self, exec_mode, gpu_id, scenes, params, enable_batch_renderer=False,
                 batch_render_view_width=64, batch_render_view_height=64, max_agents=None,
                 knn_order=None, enable_bev=False, lidar_half_angle=None, sync=None):
        import torch

        if knn_order is None:
            knn_order = int(os.environ.get("GPUDRIVE_KNN_ORDER", "0"))
        if lidar_half_angle is None:
            lidar_half_angle = float(os.environ.get("GPUDRIVE_LIDAR_HALF_ANGLE", "0"))
        if sync is None:
            sync = os.environ.get("GPUDRIVE_SYNC_STEP", "0") not in ("", "0")
        self._sync = bool(sync)

        if int(exec_mode) != int(ExecMode.CUDA):
            raise RuntimeError(
                "madrona_gpudrive (MI355X build): only the device path exists; ExecMode.CPU is not "
                "provided and there is no CPU fallback. Use ExecMode.CUDA (HIP on gfx950).")
        if enable_batch_renderer:
            raise NotImplementedError("the Madrona batch renderer is outside this engine's scope")
        if not torch.cuda.is_available():
            raise RuntimeError("madrona_gpudrive (MI355X build): no HIP device visible to torch")
        scenes = [os.fspath(s) for s in scenes]
        if len(scenes) < 1:
            raise ValueError("SimManager: scenes must not be empty")
        self._L = _capi.lib()
        self._W = len(scenes)
        self._A = int(max_agents) if max_agents is not None else kMaxAgentCount
        self._device = torch.device("cuda", int(gpu_id))
        self._params = params
        self._tensors = {}
        cfg = _capi.GdConfig()
        cfg.num_worlds = self._W
        cfg.max_agents = self._A
        cfg.device_id = int(gpu_id)
        cfg.knn_order = int(knn_order)
        cfg.alloc_bev = 1 if enable_bev else 0
        cfg.lidar_half_angle = float(lidar_half_angle)
        self._enable_bev = bool(enable_bev)
        with torch.cuda.device(self._device):
            cfg.stream = torch.cuda.current_stream(self._device).cuda_stream
            desc = _capi.GdTensorDesc()
            for slot in range(_capi.T_COUNT):
                if slot == _capi.T_BEV and not enable_bev:
                    continue
                _capi.check(self._L.gd_tensor_shape(slot, self._W, self._A, C.byref(desc)), "gd_tensor_shape")
                dims = [int(desc.dims[i]) for i in range(desc.ndim)]
                dt = torch.float32 if desc.dtype == _capi.DTYPE_F32 else torch.int32
                t = torch.zeros(dims, dtype=dt, device=self._device)
                self._tensors[slot] = t
                cfg.external[slot] = t.data_ptr()
            torch.cuda.synchronize(self._device)
            arr = (C.c_char_p * self._W)(*[s.encode("utf-8") for s in scenes])
            cparams = params._to_c()
            h = C.c_void_p()
            _capi.check(self._L.gd_create(C.byref(cfg), C.byref(cparams), arr, C.byref(h)), "gd_create")
            self._stream = cfg.stream  # the stream the engine launches on until _bind_stream() sees another one
        self._h = h
        self._direct = None
        dp = os.environ.get("GPUDRIVE_DIRECT_PACK", "0")
        if dp not in ("", "0"):
            self.direct_pack(only=dp == "2")
        if self._sync:
            self.sync()

    # ---- lifetime ----
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.gd_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bind_stream(self):
        """Launch on torch's current stream.  A change of stream drains the old one first (gd_set_stream), so work
        already queued there is ordered before everything launched from now on."""
        import torch
        st = torch.cuda.current_stream(self._device).cuda_stream
        if st != self._stream:
            _capi.check(self._L.gd_set_stream(self._h, st), "gd_set_stream")
            self._stream = st

    def _after(self):
        if self._sync:  # reference semantics: the call returns after the task graph has finished (src/mgr.cpp:154-156)
            _capi.check(self._L.gd_sync(self._h), "gd_sync")

    # ---- control: reference src/mgr.cpp:569-588 ----
    def step(self):
        self._bind_stream()
        _capi.check(self._L.gd_step(self._h), "gd_step")
        self._after()

    def reset(self, worlds):
        """Accepts an int (gpudrive sb3_wrapper.py:163), a list, a numpy array (env_puffer.py:376)
        or a torch tensor."""
        if hasattr(worlds, "detach"):
            worlds = worlds.detach().cpu().numpy()
        idx = np.atleast_1d(np.asarray(worlds)).astype(np.int32).ravel()
        idx = np.ascontiguousarray(idx)
        self._bind_stream()
        _capi.check(self._L.gd_reset(self._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx)), "gd_reset")
        self._after()

    def set_maps(self, maps):
        maps = [os.fspath(m) for m in maps]
        self._bind_stream()
        arr = (C.c_char_p * len(maps))(*[m.encode("utf-8") for m in maps])
        _capi.check(self._L.gd_set_maps(self._h, arr, len(maps)), "gd_set_maps")
        self._after()

    def deleteAgents(self, agents_to_delete):
        worlds, offsets, ids = [], [0], []
        for w, lst in dict(agents_to_delete).items():
            worlds.append(int(w))
            ids.extend(int(x) for x in lst)
            offsets.append(len(ids))
        wa = np.asarray(worlds, np.int32)
        oa = np.asarray(offsets, np.int32)
        ia = np.asarray(ids if ids else [0], np.int32)
        self._bind_stream()
        P = C.POINTER(C.c_int32)
        _capi.check(self._L.gd_delete_agents(self._h, wa.ctypes.data_as(P), oa.ctypes.data_as(P),
                                             ia.ctypes.data_as(P), len(worlds)), "gd_delete_agents")
        self._after()

    def sync(self):
        _capi.check(self._L.gd_sync(self._h), "gd_sync")

    def packed_observations(self, out=None):
        """Extension (not in the reference module): the flattened, normalised observation that
        `GPUDriveTorchEnv.get_obs()` assembles from the raw tensors (ego 6 | partners (A-1)*6 |
        road points 200*13), written by one fused kernel.  Returns a [W, A, D] float32 tensor."""
        import torch
        D = 6 + (self._A - 1) * 6 + kMaxAgentMapObservationsCount * 13
        if out is None:
            out = getattr(self, "_packed", None)
            if out is None:
                out = self._packed = torch.empty((self._W, self._A, D), dtype=torch.float32, device=self._device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == self._W * self._A * D
        self._bind_stream()
        _capi.check(self._L.gd_pack_observations(self._h, out.data_ptr(), out.numel() * 4), "gd_pack_observations")
        return out

    def direct_pack(self, only=True, out=None):
        """Extension: from now on the step itself writes the packed observation (ego + partner columns by the state kernel,
        the 200 x 13 road columns by the road kernel) instead of `packed_observations()` making a second pass over the raw
        tensors; `packed_observations()` then just returns that tensor.  `only=True`: the raw partner / road rows of live
        agents are no longer written (a learner that reads nothing else).  Returns False -- and changes nothing -- where it
        is not available (disableClassicalObs; the developer switch GPUDRIVE_LINEAR_LEGACY=1): the second-pass
        `packed_observations()` keeps working there.  GPUDRIVE_DIRECT_PACK=1 (raw rows kept) / 2 (only) at construction does
        the same for an unchanged caller."""
        import torch
        D = 6 + (self._A - 1) * 6 + kMaxAgentMapObservationsCount * 13
        if out is None:
            out = getattr(self, "_packed", None)
            if out is None:
                out = torch.empty((self._W, self._A, D), dtype=torch.float32, device=self._device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == self._W * self._A * D
        self._bind_stream()
        rc = self._L.gd_attach_packed(self._h, out.data_ptr(), out.numel() * 4, 1 if only else 0)
        if rc == _capi.GD_ERR_UNSUPPORTED:
            return False
        _capi.check(rc, "gd_attach_packed")
        self._packed = self._direct = out
        return True

    def direct_pack_off(self):
        """Back to the raw tensors + the second-pass `packed_observations()`."""
        if getattr(self, "_direct", None) is not None:
            self._bind_stream()
            _capi.check(self._L.gd_attach_packed(self._h, None, 0, 0), "gd_attach_packed")
            self._direct = None

    def expert_actions(self):
        """Extension: what `GPUDriveTorchEnv.get_expert_actions()` returns (reference
        gpudrive/env/env_torch.py:1445-1509), from one kernel over the expert trajectory rows:
        (inferred_actions [W,A,91,3 or 10], pos_xy [W,A,91,2], vel_xy [W,A,91,2], yaw [W,A,91,1],
        valids [W,A,91,1] int32), inferred actions clamped per dynamics model."""
        import torch
        cols = 10 if int(self._params.dynamicsModel) == int(DynamicsModel.State) else 3
        mk = lambda c, dt=torch.float32: torch.empty((self._W, self._A, episodeLen, c), dtype=dt, device=self._device)
        act, pos, vel, yaw, valid = mk(cols), mk(2), mk(2), mk(1), mk(1, torch.int32)
        self._bind_stream()
        _capi.check(self._L.gd_expert_actions(self._h, act.data_ptr(), cols, pos.data_ptr(), vel.data_ptr(), yaw.data_ptr(),
                                              valid.data_ptr()), "gd_expert_actions")
        return act, pos, vel, yaw, valid

    def advance_log_playback(self, init_steps):
        """Extension: `GPUDriveTorchEnv.advance_sim_with_log_playback(init_steps)` (reference
        gpudrive/env/env_torch.py:1274-1293) without leaving the device: every step feeds the logged
        (inferred) action of that time step to every agent slot and steps.  ValueError if init_steps >= 91."""
        self._bind_stream()
        _capi.check(self._L.gd_advance_log_playback(self._h, int(init_steps)), "gd_advance_log_playback")

    # ---- dead / out-of-scope API kept for attribute compatibility ----
    def bev_observation_tensor(self):
        """[W, A, 200, 200, 1] f32 (reference src/mgr.cpp:870-880).  The reference rasterises the BEV of every agent
        on every step whether anyone reads it or not (160 KB per agent, 10.5 GB at 1024 x 64).  Here the
        tensor is created by the first call (or by enable_bev=True at construction): that call computes the rasters of
        the current state, and from then on every step / reset refreshes them like the reference does, so the
        unchanged caller (gpudrive/env/env_torch.py:926-945) sees the same data."""
        import torch
        if _capi.T_BEV not in self._tensors:
            desc = _capi.GdTensorDesc()
            _capi.check(self._L.gd_tensor_shape(_capi.T_BEV, self._W, self._A, C.byref(desc)), "gd_tensor_shape")
            dims = [int(desc.dims[i]) for i in range(desc.ndim)]
            with torch.cuda.device(self._device):
                t = torch.zeros(dims, dtype=torch.float32, device=self._device)
                torch.cuda.current_stream(self._device).synchronize()
                self._bind_stream()
                _capi.check(self._L.gd_attach_bev(self._h, t.data_ptr()), "gd_attach_bev")
            self._tensors[_capi.T_BEV] = t
            self._enable_bev = True
            self._after()
        return Tensor(self._tensors[_capi.T_BEV])

    def valid_state_tensor(self):
        raise NotImplementedError("valid_state_tensor: nothing is exported under ExportID::ValidState in the reference either")

    def rgb_tensor(self):
        raise NotImplementedError("rgb_tensor: the Madrona batch renderer is outside this engine's scope")

    def depth_tensor(self):
        raise NotImplementedError("depth_tensor: the Madrona batch renderer is outside this engine's scope")

    # ---- engine hooks used by bench.py / tests ----
    def stat(self, which):
        """0 = steps replayed from the captured hipGraph, 1 = steps launched kernel by kernel, 2 = graph captures,
        3 / 4 = the set-order road kernel's schedule for this batch (rows fused 0/1, agents per wave), 5 = live agents."""
        out = C.c_int64()
        _capi.check(self._L.gd_stat(self._h, int(which), C.byref(out)), "gd_stat")
        return out.value

    def kernel_timing(self, enable):
        _capi.check(self._L.gd_kernel_timing_enable(self._h, int(bool(enable))))

    def kernel_timing_read(self, kernel):
        ms = C.c_double()
        n = C.c_int64()
        _capi.check(self._L.gd_kernel_timing_read(self._h, int(kernel), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_get_state(self):
        out = np.zeros((self._W, self._A, 11), np.float32)
        _capi.check(self._L.gd_debug_get_state(self._h, out.ctypes.data))
        return out

    def debug_road_path(self):
        """[W, A] int32: > 0 rank replay (candidate count), -1 history replay on keys (fallback), 0 padding slot,
        -2 rank replay not in use."""
        out = np.zeros((self._W, self._A), np.int32)
        _capi.check(self._L.gd_debug_road_path(self._h, out.ctypes.data))
        return out

    def debug_set_state(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (self._W, self._A, 11)
        _capi.check(self._L.gd_debug_set_state(self._h, st.ctypes.data))


def _make_getter(slot):
    def getter(self):
        return Tensor(self._tensors[slot])
    return getter


for _name, _slot in _SLOT_GETTERS.items():
    setattr(SimManager, _name, _make_getter(_slot))
