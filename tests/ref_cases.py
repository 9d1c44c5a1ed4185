"""Known-answer cases restated from the reference's own tests (paths relative to /root/reference).

Each `check_*` takes a factory `make_sim(scenes, **params)` returning an object with the
`madrona_gpudrive.SimManager` call surface, so the same case pins the CPU oracle (CPU suite)
and the HIP path (GPU suite).  `as_np` accepts numpy arrays, torch tensors and the drop-in
`Tensor` wrapper.
"""
import json
import math

import numpy as np

f32 = np.float32


def as_np(t):
    if hasattr(t, "to_torch"):
        t = t.to_torch()
    if hasattr(t, "detach"):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def write_actions(sim, arr):
    """In-place write into the exported action buffer (gpudrive/env/env_torch.py:645-664)."""
    t = sim.action_tensor()
    if hasattr(t, "to_torch"):
        import torch
        tt = t.to_torch()
        tt.copy_(torch.as_tensor(arr, dtype=tt.dtype).to(tt.device))
    else:
        np.copyto(t, arr)


# ---------------------------------------------------------------------------------------
# std::default_random_engine (libstdc++: minstd_rand0) + uniform_real_distribution<float>
# as used by tests/bicyclemodel.cpp:77-79
# ---------------------------------------------------------------------------------------
class MinStdRand0:
    def __init__(self, seed):
        self.m = 2147483647
        self.x = seed % self.m or 1

    def __call__(self):
        self.x = (self.x * 16807) % self.m
        return self.x


def uniform_float(rng, a, b):
    r = float(rng.m - 1)  # max - min + 1 = 2^31 - 2
    ret = f32(f32(rng() - 1) / f32(r))
    if ret >= f32(1.0):
        ret = np.nextafter(f32(1.0), f32(0.0))
    return f32(f32(f32(b) - f32(a)) * ret + f32(a))


def calc_mean(raw):
    """tests/test_utils.cpp:21-53."""
    mx = f32(0); my = f32(0); n = 0
    for obj in raw["objects"]:
        for i, pos in enumerate(obj["position"]):
            if obj["valid"][i] is False:
                continue
            n += 1
            mx = f32(mx + f32(f32(f32(pos["x"]) - mx) / f32(n)))
            my = f32(my + f32(f32(f32(pos["y"]) - my) / f32(n)))
    for road in raw["roads"]:
        for p in road["geometry"]:
            n += 1
            mx = f32(mx + f32(f32(f32(p["x"]) - mx) / f32(n)))
            my = f32(my + f32(f32(f32(p["y"]) - my) / f32(n)))
    return mx, my


def step_bicycle_model(x, y, theta, speed, acc, steer, dt=0.1, L=1.0):
    """tests/bicyclemodel.cpp:84-100 (closed form, float results)."""
    x, y, theta, speed, acc, steer, dt, L = map(float, (f32(x), f32(y), f32(theta), f32(speed), f32(acc), f32(steer), f32(dt), f32(L)))
    v = float(f32(speed + 0.5 * acc * dt))
    beta = float(f32(math.atan(math.tan(steer) * (L / 2) / L)))
    w = float(f32(v * math.cos(beta) * math.tan(steer) / L))
    xn = f32(x + v * math.cos(theta + beta) * dt)
    yn = f32(y + v * math.sin(theta + beta) * dt)
    tn = float(f32(math.fmod(theta + w * dt, math.pi * 2)))
    tn = tn - math.pi * 2 if tn > math.pi else (tn + math.pi * 2 if tn < -math.pi else tn)
    sn = abs(f32(speed + acc * dt))
    return f32(xn), f32(yn), f32(tn), f32(sn)


EPS = 1e-3  # tests/test_utils.hpp:11


def check_bicycle_model(make_sim, test_json):
    """tests/bicyclemodel.cpp:20-81,187-242: Classic model, 10 steps, agent 0, seeded actions,
    closed form fed with the simulator's own previous state, tolerance 1e-3."""
    sim = make_sim([test_json], polylineReductionThreshold=0.0, observationRadius=100.0,
                   collisionBehaviour=2, initOnlyValidAgentsAtFirstStep=0, dynamicsModel=0)
    raw = json.load(open(test_json))
    mean = calc_mean(raw)
    means = as_np(sim.world_means_tensor())
    assert means[0, 0] == mean[0] and means[0, 1] == mean[1]
    # agent row 0 is the SDC (src/json_serialization.hpp:318-346)
    sdc = raw["objects"][raw["metadata"]["sdc_track_index"]]
    length = f32(sdc["length"])
    abs_obs = as_np(sim.absolute_self_observation_tensor())
    self_obs = as_np(sim.self_observation_tensor())
    assert abs(abs_obs[0, 0, 0] - (f32(sdc["position"][0]["x"]) - mean[0])) < EPS
    assert abs(abs_obs[0, 0, 1] - (f32(sdc["position"][0]["y"]) - mean[1])) < EPS
    th = float(f32(sdc["heading"][0]))
    th = th - 2 * math.pi if th > math.pi else (th + 2 * math.pi if th < -math.pi else th)
    assert abs(abs_obs[0, 0, 7] - th) < EPS
    sp = math.hypot(f32(sdc["velocity"][0]["x"]), f32(sdc["velocity"][0]["y"]))
    assert abs(self_obs[0, 0, 0] - sp) < EPS
    assert as_np(sim.controlled_state_tensor())[0, 0, 0] == 1

    rng = MinStdRand0(42)
    for _ in range(10):
        abs_obs = as_np(sim.absolute_self_observation_tensor()).copy()
        self_obs = as_np(sim.self_observation_tensor()).copy()
        prev = (abs_obs[0, 0, 0], abs_obs[0, 0, 1], abs_obs[0, 0, 7], self_obs[0, 0, 0])
        acc = uniform_float(rng, -3.0, 2.0)
        steer = uniform_float(rng, -0.7, 0.7)
        act = as_np(sim.action_tensor()).copy()
        act[0, 0, :3] = (acc, steer, 0.0)
        write_actions(sim, act)
        exp = step_bicycle_model(*prev, acc, steer, 0.1, length)
        sim.step()
        abs_obs = as_np(sim.absolute_self_observation_tensor())
        self_obs = as_np(sim.self_observation_tensor())
        assert abs(abs_obs[0, 0, 0] - exp[0]) <= EPS
        assert abs(abs_obs[0, 0, 1] - exp[1]) <= EPS
        assert abs(abs_obs[0, 0, 7] - exp[2]) <= EPS
        assert abs(self_obs[0, 0, 0] - exp[3]) <= EPS
    return sim


_ROAD_T = {"road_edge": 1, "road_line": 2, "lane": 3, "crosswalk": 4, "speed_bump": 5, "stop_sign": 6}


def check_map_observation(make_sim, test_json):
    """tests/observationTest.cpp:87-139 (intent): with polylineReductionThreshold = 0 the global
    map rows are, in JSON road order (types that create no entity skipped): mean-centred segment
    midpoints / 4-corner centroid / stop-sign point; type column exact; tolerance 1e-3."""
    sim = make_sim([test_json], polylineReductionThreshold=0.0, observationRadius=100.0,
                   collisionBehaviour=2, roadObservationAlgorithm=0)
    raw = json.load(open(test_json))
    mean = calc_mean(raw)
    obs = as_np(sim.map_observation_tensor())[0]
    idx = 0
    for road in raw["roads"]:
        t = _ROAD_T.get(road["type"], 0)
        g = [(f32(p["x"]), f32(p["y"])) for p in road["geometry"]]
        if t == 0:
            continue
        if 3 < t < 6:
            x = (g[0][0] + g[1][0] + g[2][0] + g[3][0]) / 4 - mean[0]
            y = (g[0][1] + g[1][1] + g[2][1] + g[3][1]) / 4 - mean[1]
            assert abs(obs[idx, 0] - x) < EPS and abs(obs[idx, 1] - y) < EPS
            assert obs[idx, 6] == t
            idx += 1
        elif t == 6:
            assert abs(obs[idx, 0] - (g[0][0] - mean[0])) < EPS
            assert abs(obs[idx, 1] - (g[0][1] - mean[1])) < EPS
            assert obs[idx, 6] == t
            idx += 1
        else:
            for j in range(len(g) - 1):
                if idx >= obs.shape[0]:
                    break
                x1 = g[j][0] - mean[0]; y1 = g[j][1] - mean[1]
                x2 = g[j + 1][0] - mean[0]; y2 = g[j + 1][1] - mean[1]
                assert abs(obs[idx, 0] - (x2 + x1) / 2) < EPS
                assert abs(obs[idx, 1] - (y2 + y1) / 2) < EPS
                assert obs[idx, 6] == t
                idx += 1
        if idx >= obs.shape[0]:
            break
    shape = as_np(sim.shape_tensor())
    assert shape[0, 1] == idx
    # padding rows: MapObservation::zero() (src/level_gen.cpp:331-335)
    if idx < obs.shape[0]:
        assert (obs[idx:, 7] == -1).all() and (obs[idx:, 8] == -1).all() and (obs[idx:, :7] == 0).all()
    return sim


def _forward_inverse(make_sim, test_json, model):
    sim = make_sim([test_json], polylineReductionThreshold=0.5, observationRadius=10.0,
                   collisionBehaviour=0, rewardType=0, distanceToGoalThreshold=1.0,
                   distanceToExpertThreshold=1.0, maxNumControlledAgents=2, IgnoreNonVehicles=1,
                   dynamicsModel=model)
    idx = 1
    traj = as_np(sim.expert_trajectory_tensor())[0, idx]
    pos = traj[:2 * 91].reshape(91, 2)
    vel = traj[2 * 91:4 * 91].reshape(91, 2)
    head = traj[4 * 91:5 * 91]
    inv = traj[6 * 91:16 * 91].reshape(91, 10)
    abs_obs = as_np(sim.absolute_self_observation_tensor())
    self_obs = as_np(sim.self_observation_tensor())
    assert np.allclose(abs_obs[0, idx, :2], pos[0], atol=1e-2)
    assert abs(abs_obs[0, idx, 7] - head[0]) <= 1e-2
    assert abs(self_obs[0, idx, 0] - np.linalg.norm(vel[0])) <= 1e-2
    act = np.zeros_like(as_np(sim.action_tensor()))
    act[:, idx, :3] = inv[0, :3]
    write_actions(sim, act)
    sim.step()
    abs_obs = as_np(sim.absolute_self_observation_tensor())
    self_obs = as_np(sim.self_observation_tensor())
    assert np.allclose(abs_obs[0, idx, :2], pos[1], atol=2e-2), (abs_obs[0, idx, :2], pos[1])
    if model == 1:
        # This fork sets consts::useEstimatedYaw = true (src/consts.hpp:15): the inverse bicycle
        # model targets atan2(v_y, v_x) of the next step, not the logged heading
        # (src/dynamics.hpp:133-136), so the reference test's heading assertion
        # (tests/test_waymax_model.py:58) cannot hold on this scene (|logged - estimated| = 8.9e-3).
        # The pin is restated against the yaw the source actually targets.
        assert abs(abs_obs[0, idx, 7] - math.atan2(vel[1][1], vel[1][0])) <= 3e-3
    else:
        assert abs(abs_obs[0, idx, 7] - head[1]) <= 3e-3
    assert abs(self_obs[0, idx, 0] - np.linalg.norm(vel[1])) <= 1e-3
    return sim


def check_delta_model(make_sim, test_json):
    """tests/test_delta_model.py:29-60."""
    return _forward_inverse(make_sim, test_json, 2)


def check_waymax_model(make_sim, test_json):
    """tests/test_waymax_model.py:29-59."""
    return _forward_inverse(make_sim, test_json, 1)


def check_expert_replay(make_sim, test_json):
    """tests/test_expert.py:6-60: all-expert replay until all done; every vehicle reaches its
    goal; zero collision flags."""
    sim = make_sim([test_json], polylineReductionThreshold=0.5, observationRadius=10.0,
                   collisionBehaviour=0, rewardType=0, distanceToGoalThreshold=1.0,
                   distanceToExpertThreshold=1.0, maxNumControlledAgents=0, IgnoreNonVehicles=1,
                   isStaticAgentControlled=0)
    n = 0
    while not as_np(sim.done_tensor()).all():
        sim.step()
        n += 1
        assert n <= 91
    info = as_np(sim.info_tensor())
    shape = as_np(sim.shape_tensor())
    valid = info[info[:, :, -1] == 7]
    assert valid[:, -2].sum() == shape[:, 0].sum()
    assert valid[:, :3].sum() == 0
    return sim
