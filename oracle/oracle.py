"""ORACLE (test infrastructure, NOT product code): ctypes front-end of oracle/gd_oracle.c.

`OracleSim` mirrors the reference's `madrona_gpudrive.SimManager` call surface
(src/bindings.cpp:91-149) on numpy arrays so parity tests read like the reference's own
tests.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import scene as _scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

K_MAP = 200
MAX_ROADS = 10000
TRAJ_F = 16 * 91


class OrcParams(C.Structure):
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
        ("enableBev", C.c_int32),
        ("lidarHalfAngle", C.c_float),
    ]


class OrcMap(C.Structure):
    _fields_ = [
        ("n_obj", C.c_int32),
        ("obj_pos", C.c_void_p), ("obj_vel", C.c_void_p), ("obj_head", C.c_void_p),
        ("obj_valid", C.c_void_p), ("obj_npos", C.c_void_p), ("obj_size", C.c_void_p),
        ("obj_goal", C.c_void_p), ("obj_type", C.c_void_p), ("obj_id", C.c_void_p),
        ("obj_expert", C.c_void_p), ("obj_meta", C.c_void_p),
        ("n_road", C.c_int32),
        ("road_off", C.c_void_p), ("road_pts", C.c_void_p), ("road_type", C.c_void_p),
        ("road_id", C.c_void_p), ("road_maptype", C.c_void_p),
        ("mean", C.c_float * 2),
        ("name", C.c_char * 32),
        ("scenario_id", C.c_char * 32),
    ]


def build(force=False, archflags=""):
    """Compile oracle/gd_oracle.c -> oracle/liboracle.so (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gd_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        cmd = ["make", "-C", _HERE, "-B", "liboracle.so"]
        if archflags:
            cmd.append("ARCHFLAGS=" + archflags)
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return so


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    so = path or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        so = build()
    L = C.CDLL(so)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcMap), C.c_int]
    L.orc_delete_agents.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_init.argtypes = [C.c_void_p]
    L.orc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_step.argtypes = [C.c_void_p]
    L.orc_tensor.restype = C.c_void_p
    L.orc_tensor.argtypes = [C.c_void_p, C.c_int]
    L.orc_knn_inserts.restype = C.c_int64
    L.orc_knn_inserts.argtypes = [C.c_void_p]
    L.orc_get_state.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_set_threads.argtypes = [C.c_int32]
    L.orc_set_threads.restype = C.c_int32
    L.orc_debug_road_obs.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.orc_debug_road_obs.restype = C.c_int32
    L.orc_set_state.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_test_obb_collide.restype = C.c_int
    L.orc_test_obb_collide.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_test_reference_frame.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    if path is None:
        _LIB = L
    return L


def expert_actions_of(raw, dynamics_model):
    """GPUDriveTorchEnv.get_expert_actions() (reference gpudrive/env/env_torch.py:1445-1509) over
    LogTrajectory's slicing of the trajectory rows (gpudrive/datatypes/trajectory.py:24-41).
    raw: [W, A, 1456] float32; dynamics_model: 0 classic, 1 bicycle, 2 delta_local, 3 state.
    torch.clamp with Python scalars on a float32 tensor clamps against the float32 bounds."""
    T = 91
    W, A = raw.shape[:2]
    pos = raw[:, :, :2 * T].reshape(W, A, T, 2)
    vel = raw[:, :, 2 * T:4 * T].reshape(W, A, T, 2)
    yaw = raw[:, :, 4 * T:5 * T].reshape(W, A, T, 1)
    with np.errstate(invalid="ignore"):
        valids = raw[:, :, 5 * T:6 * T].reshape(W, A, T, 1).astype(np.int32)
    inferred = raw[:, :, 6 * T:16 * T].reshape(W, A, T, 10)
    f32 = np.float32
    if dynamics_model == 3:    # state: (x, y, 1, yaw, vx, vy, 0, 0, 0, 0), :1470-1487
        act = np.concatenate([pos, np.ones((W, A, T, 1), f32), yaw, vel, np.zeros((W, A, T, 4), f32)], -1)
    elif dynamics_model == 2:  # delta_local, :1460-1469
        act = inferred[..., :3].copy()
        act[..., 0] = np.clip(act[..., 0], f32(-6), f32(6))
        act[..., 1] = np.clip(act[..., 1], f32(-6), f32(6))
        act[..., 2] = np.clip(act[..., 2], f32(-np.pi), f32(np.pi))
    else:                      # classic / bicycle, :1488-1499
        act = inferred[..., :3].copy()
        act[..., 0] = np.clip(act[..., 0], f32(-6), f32(6))
        act[..., 1] = np.clip(act[..., 1], f32(-0.3), f32(0.3))
    return act.astype(f32), pos.copy(), vel.copy(), yaw.copy(), valids


def default_params(**kw):
    """Defaults of src/init.hpp:111-127 (polylineReductionThreshold / observationRadius /
    rewardParams have no default there; zero-initialised like `Parameters()` from Python)."""
    p = OrcParams()
    p.polylineReductionThreshold = 0.0
    p.observationRadius = 0.0
    p.rewardType = 0
    p.distanceToGoalThreshold = 0.0
    p.distanceToExpertThreshold = 0.0
    p.collisionBehaviour = 0
    p.maxNumControlledAgents = 10000
    p.IgnoreNonVehicles = 0
    p.roadObservationAlgorithm = 0
    p.initOnlyValidAgentsAtFirstStep = 1
    p.isStaticAgentControlled = 0
    p.enableLidar = 0
    p.disableClassicalObs = 0
    p.dynamicsModel = 0
    p.readFromTracksToPredict = 0
    p.enableBev = 0
    p.lidarHalfAngle = 0.0
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


_T = dict(action=0, reward=1, done=2, info=3, self_obs=4, abs_obs=5, partner=6, agent_map=7,
          map=8, lidar=9, bev=10, steps=11, shape=12, controlled=13, resp=14, traj=15, means=16,
          meta=17, deleted=18, map_name=19, scenario_id=20, agent_id=21)

_SCENE_CACHE = {}


def _packed_scene(path, thr):
    key = (os.path.abspath(path), float(np.float32(thr)))
    if key not in _SCENE_CACHE:
        _SCENE_CACHE[key] = _scene.pack_map(_scene.parse_scene(path, thr))
    return _SCENE_CACHE[key]


def _to_orc_map(pk):
    m = OrcMap()
    m.n_obj = pk["n_obj"]
    for k in ("obj_pos", "obj_vel", "obj_head", "obj_valid", "obj_npos", "obj_size", "obj_goal",
              "obj_type", "obj_id", "obj_expert", "obj_meta", "road_off", "road_pts", "road_type",
              "road_id", "road_maptype"):
        setattr(m, k, pk[k].ctypes.data)
    m.n_road = pk["n_road"]
    m.mean[0] = float(pk["mean"][0])
    m.mean[1] = float(pk["mean"][1])
    m.name = pk["name"]
    m.scenario_id = pk["scenario_id"]
    return m


class OracleSim:
    """CPU oracle with the SimManager surface: step / reset / set_maps / deleteAgents and
    `*_tensor()` getters returning numpy views that alias the oracle's buffers."""

    def __init__(self, scenes, params, max_agents=128, lib_path=None):
        self.L = lib(lib_path)
        self.p = params
        self.W = len(scenes)
        self.A = int(max_agents)
        self.h = self.L.orc_create(C.byref(params), self.W, self.A)
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.scenes = list(scenes)
        for w, path in enumerate(scenes):
            self._set_map(w, path, 0)
        self.L.orc_init(self.h)

    def _set_map(self, w, path, clear_deleted):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        pk = _packed_scene(path, self.p.polylineReductionThreshold)
        m = _to_orc_map(pk)
        if self.L.orc_set_map(self.h, w, C.byref(m), clear_deleted) != 0:
            raise RuntimeError("orc_set_map failed")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- control ---
    def step(self):
        self.L.orc_step(self.h)

    def reset(self, worlds):
        if isinstance(worlds, (int, np.integer)):
            worlds = [int(worlds)]
        arr = np.ascontiguousarray(np.asarray(list(worlds), dtype=np.int32))
        self.L.orc_reset(self.h, arr.ctypes.data, len(arr))

    def set_maps(self, maps):
        if len(maps) != self.W:
            raise ValueError("set_maps: len(maps) must equal the number of worlds")
        for w, path in enumerate(maps):
            self._set_map(w, path, 1)
        self.scenes = list(maps)
        self.reset(list(range(self.W)))

    def deleteAgents(self, d):
        for w, ids in d.items():
            arr = np.ascontiguousarray(np.asarray(ids, dtype=np.int32))
            if self.L.orc_delete_agents(self.h, int(w), arr.ctypes.data, len(arr)) != 0:
                raise RuntimeError("orc_delete_agents failed")
        self.reset(list(range(self.W)))

    # --- tensors ---
    def _view(self, name, shape, dtype):
        ptr = self.L.orc_tensor(self.h, _T[name])
        if not ptr:
            raise RuntimeError("tensor %s not allocated" % name)
        n = int(np.prod(shape))
        ct = {np.float32: C.c_float, np.int32: C.c_int32, np.uint32: C.c_uint32}[dtype]
        buf = (ct * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def action_tensor(self): return self._view("action", (self.W, self.A, 10), np.float32)
    def reward_tensor(self): return self._view("reward", (self.W, self.A, 1), np.float32)
    def done_tensor(self): return self._view("done", (self.W, self.A, 1), np.int32)
    def info_tensor(self): return self._view("info", (self.W, self.A, 5), np.int32)
    def self_observation_tensor(self): return self._view("self_obs", (self.W, self.A, 8), np.float32)
    def absolute_self_observation_tensor(self): return self._view("abs_obs", (self.W, self.A, 14), np.float32)
    def partner_observations_tensor(self): return self._view("partner", (self.W, self.A, self.A - 1, 9), np.float32)
    def agent_roadmap_tensor(self): return self._view("agent_map", (self.W, self.A, K_MAP, 9), np.float32)
    def map_observation_tensor(self): return self._view("map", (self.W, MAX_ROADS, 9), np.float32)
    def lidar_tensor(self): return self._view("lidar", (self.W, self.A, 3, 50, 4), np.float32)
    def bev_observation_tensor(self): return self._view("bev", (self.W, self.A, 200, 200, 1), np.float32)
    def steps_remaining_tensor(self): return self._view("steps", (self.W, self.A, 1), np.int32)
    def shape_tensor(self): return self._view("shape", (self.W, 2), np.int32)
    def controlled_state_tensor(self): return self._view("controlled", (self.W, self.A, 1), np.int32)
    def response_type_tensor(self): return self._view("resp", (self.W, self.A, 1), np.int32)
    def expert_trajectory_tensor(self): return self._view("traj", (self.W, self.A, TRAJ_F), np.float32)
    def world_means_tensor(self): return self._view("means", (self.W, 3), np.float32)
    def metadata_tensor(self): return self._view("meta", (self.W, self.A, 4), np.int32)
    def deleted_agents_tensor(self): return self._view("deleted", (self.W, self.A), np.int32)
    def map_name_tensor(self): return self._view("map_name", (self.W, 32), np.int32)
    def scenario_id_tensor(self): return self._view("scenario_id", (self.W, 32), np.int32)
    def agent_id_tensor(self): return self._view("agent_id", (self.W, self.A), np.int32)

    # --- test hooks ---
    # ---- expert-action export and log playback (callers of the path) ----
    def expert_actions(self):
        return expert_actions_of(np.array(self.expert_trajectory_tensor()), int(self.p.dynamicsModel))

    def advance_log_playback(self, init_steps):
        """GPUDriveTorchEnv.advance_sim_with_log_playback (env_torch.py:1274-1293): step t copies
        log_playback_traj[:, :, t, :] into action[:, :, :cols] (env_torch.py:645-664) and steps."""
        if init_steps >= 91:
            raise ValueError("The length of the expert trajectory is 91, so init_steps should be < 91.")
        act = self.expert_actions()[0]
        cols = act.shape[-1]
        for t in range(init_steps):
            self.action_tensor()[:, :, :cols] = act[:, :, t, :]
            self.step()

    def road_obs_of(self, w, a):
        """observationOf of every road of world w in agent a's frame, in road order: [R, 9]."""
        out = np.zeros((MAX_ROADS, 9), np.float32)
        n = self.L.orc_debug_road_obs(self.h, int(w), int(a), out.ctypes.data)
        return out[:n].copy()

    def knn_inserts(self):
        return int(self.L.orc_knn_inserts(self.h))

    def get_state(self):
        out = np.zeros((self.W, self.A, 11), np.float32)
        self.L.orc_get_state(self.h, out.ctypes.data)
        return out

    def set_state(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (self.W, self.A, 11)
        self.L.orc_set_state(self.h, st.ctypes.data)


def obb_collide(posA, yawA, scaleA, posB, yawB, scaleB):
    L = lib()
    a = np.asarray(posA, np.float32); sa = np.asarray(scaleA, np.float32)
    b = np.asarray(posB, np.float32); sb = np.asarray(scaleB, np.float32)
    return bool(L.orc_test_obb_collide(a.ctypes.data, C.c_float(np.float32(yawA)), sa.ctypes.data,
                                       b.ctypes.data, C.c_float(np.float32(yawB)), sb.ctypes.data))


def reference_frame_obs(ref_pos, ref_yaw, pos, yaw, scale):
    L = lib()
    rp = np.asarray(ref_pos, np.float32); p = np.asarray(pos, np.float32); sc = np.asarray(scale, np.float32)
    out = np.zeros(9, np.float32)
    L.orc_test_reference_frame(rp.ctypes.data, C.c_float(np.float32(ref_yaw)), p.ctypes.data,
                               C.c_float(np.float32(yaw)), sc.ctypes.data, out.ctypes.data)
    return out
