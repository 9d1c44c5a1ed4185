"""GPU suite (-m gpu): the HIP path, called through the drop-in `madrona_gpudrive` module and the
C ABI, against the oracle on the same seeded inputs, plus the reference's own known-answer cases
and size-independent properties at the BASELINE.json sizes.  Nothing here reads /root/reference."""
import zlib

import numpy as np
import pytest

from tests import parity as P
from tests import ref_cases as RC
from tests.conftest import SCENE_4, SCENE_407, TEST_JSON

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def make_gpu():
    def _mk(scenes, max_agents=128, **kw):
        return P.make_gpu_sim(scenes, max_agents=max_agents, **kw)
    return _mk


def test_native_library_is_loaded():
    import madrona_gpudrive  # noqa: F401
    from gpudrive_lab_amd import _capi
    _capi.lib()
    assert "libgpudrive_amd.so" in open("/proc/self/maps").read()


def test_cpu_exec_mode_is_refused():
    import madrona_gpudrive as mg
    with pytest.raises(RuntimeError):
        mg.SimManager(mg.madrona.ExecMode.CPU, 0, [TEST_JSON], mg.Parameters())


def test_missing_scene_raises_not_aborts():
    import madrona_gpudrive as mg
    with pytest.raises(FileNotFoundError):
        mg.SimManager(mg.madrona.ExecMode.CUDA, 0, ["/nonexistent.json"], mg.Parameters(), max_agents=64)


# ---- the reference's own tests, run on the HIP path ----
def test_ref_bicycle_model(make_gpu):
    RC.check_bicycle_model(make_gpu, TEST_JSON)


def test_ref_map_observation(make_gpu):
    RC.check_map_observation(make_gpu, TEST_JSON)


def test_ref_delta_model(make_gpu):
    RC.check_delta_model(make_gpu, TEST_JSON)


def test_ref_waymax_model(make_gpu):
    RC.check_waymax_model(make_gpu, TEST_JSON)


def test_ref_expert_replay(make_gpu):
    RC.check_expert_replay(make_gpu, TEST_JSON)


# ---- lockstep parity against the oracle ----
ALL_OBJECTS = dict(isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)

LOCKSTEP = [
    # BASELINE configs[0]: 1 scene, 1 world, delta dynamics (tests/test_delta_model.py:7-27)
    ("cfg0_delta", [TEST_JSON], 128, 40, dict(polylineReductionThreshold=0.5, observationRadius=10.0, collisionBehaviour=0,
                                               rewardType=0, distanceToGoalThreshold=1.0, maxNumControlledAgents=2,
                                               IgnoreNonVehicles=1, dynamicsModel=2)),
    # configs[1]: classic, k-NN roads, radius 50, threshold 0.1, collisions ignored
    ("cfg1_classic_knn", [TEST_JSON, SCENE_407, SCENE_4, SCENE_4], 64, 95,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)),
    # configs[2]: collisions on (AgentStop), goal-reach reward
    ("cfg2_agent_stop", [SCENE_4, SCENE_407, TEST_JSON], 64, 95,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)),
    ("agent_removed_bicycle", [SCENE_4, SCENE_407], 64, 60,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=1, rewardType=0,
          distanceToGoalThreshold=2.0, dynamicsModel=1, **ALL_OBJECTS)),
    ("linear_roads_fork128", [SCENE_4, TEST_JSON], 128, 30,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
          distanceToGoalThreshold=2.0, roadObservationAlgorithm=1, dynamicsModel=0, **ALL_OBJECTS)),
    # this fork's kMaxAgentCount with more than 64 live agents (81 objects): the road kernel runs two workgroups per world
    ("knn_fork128", [SCENE_4, SCENE_407, SCENE_4], 128, 25,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)),
    ("state_model", [SCENE_407], 64, 10,
     dict(polylineReductionThreshold=0.5, observationRadius=50.0, collisionBehaviour=2, rewardType=0,
          distanceToGoalThreshold=2.0, dynamicsModel=3, **ALL_OBJECTS)),
    # unreduced polylines: R = 9899 > K, long heap histories
    ("knn_unreduced", [TEST_JSON], 64, 4,
     dict(polylineReductionThreshold=0.0, observationRadius=100.0, collisionBehaviour=2, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, initOnlyValidAgentsAtFirstStep=0)),
    # expert replay only (no controlled agents), default init rules
    ("experts_only", [SCENE_4], 64, 95,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
          distanceToGoalThreshold=2.0, maxNumControlledAgents=0, IgnoreNonVehicles=1)),
]


@pytest.mark.parametrize("name,scenes,A,steps,kw", LOCKSTEP, ids=[c[0] for c in LOCKSTEP])
def test_lockstep_parity(oracle_mod, name, scenes, A, steps, kw):
    gpu = P.make_gpu_sim(scenes, max_agents=A, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=A, **kw)
    # t = 0: everything the constructor leaves behind
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, steps, kw.get("dynamicsModel", 0), seed=zlib.crc32(name.encode()) % 1000)
    gpu.close()


SET_ORDER = [
    ("set_classic", [TEST_JSON, SCENE_407, SCENE_4], 64, 20,
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)),
    # unreduced polylines and a 100 m radius: far more than K roads in radius -> exact selection path
    ("set_unreduced", [TEST_JSON], 64, 4,
     dict(polylineReductionThreshold=0.0, observationRadius=100.0, collisionBehaviour=2, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, initOnlyValidAgentsAtFirstStep=0)),
    ("set_fork128", [SCENE_4, SCENE_407], 128, 6,
     dict(polylineReductionThreshold=0.0, observationRadius=60.0, collisionBehaviour=0, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)),
]


@pytest.mark.parametrize("rows", ["rows_kernel", "rows_fused", "rows_fused_16_per_wave", "rows_kernel_16_per_wave"])
@pytest.mark.parametrize("name,scenes,A,steps,kw", SET_ORDER, ids=[c[0] for c in SET_ORDER])
def test_set_order_mode_matches_reference_rows_as_a_set(oracle_mod, monkeypatch, name, scenes, A, steps, kw, rows):
    """gd_config.knn_order = GD_KNN_SET_ORDER: every other tensor identical, road rows equal to the
    oracle's as a set (the reference's order is a heap-history artefact).  Both write-outs of that mode
    (k_map_rows, and rows stored by the selecting wave: the engine picks by batch size) are run."""
    monkeypatch.setenv("GPUDRIVE_SET_FUSED_ROWS", "1" if "fused" in rows else "0")
    if "16_per_wave" in rows:  # the engine's choice for thousands of worlds
        monkeypatch.setenv("GPUDRIVE_SET_AGENTS_PER_WAVE", "16")
    gpu = P.make_gpu_sim(scenes, max_agents=A, knn_order=1, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=A, **kw)
    names = [n for n in P.OBS_TENSORS if n != "agent_roadmap_tensor"]
    rng = np.random.default_rng(5)
    for k in range(steps):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
        P.compare_ints(gpu, orc, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
        P.compare_obs(gpu, orc, names=names)
        P.compare_roadmap_as_set(gpu, orc)
    gpu.close()


@pytest.mark.parametrize("half_angle", [0.0, float(np.pi)], ids=["cone120", "full360"])
def test_lidar_parity(oracle_mod, half_angle):
    """BASELINE configs[4]: LiDAR 3 x 50 rays (120 degree cone = reference, and the 360 degree
    variant), mixed vehicle / cyclist / pedestrian agents.  Parity is against the oracle's
    restatement of the mesh extents (the reference's BVH is absent: unpinned)."""
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, enableLidar=1, **ALL_OBJECTS)
    scenes = [SCENE_4, SCENE_407, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, lidar_half_angle=half_angle, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, lidarHalfAngle=half_angle, **kw)
    rng = np.random.default_rng(11)
    for k in range(12):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        act[..., 2] = rng.uniform(-0.5, 0.5, act.shape[:2])  # head angle (ClassicAction.headAngle)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
        frac = P.compare_lidar(gpu, orc)
        assert frac > 0.05  # the scene is not empty
    gpu.close()


BEV_CASES = [
    ("waymo64", [SCENE_4, TEST_JSON], 64, dict(polylineReductionThreshold=0.1, observationRadius=50.0)),
    # unreduced polylines, 100 m radius: the 200-road cap binds, boxes overlap heavily (paint order matters everywhere)
    ("unreduced_r100", [TEST_JSON, SCENE_407], 64, dict(polylineReductionThreshold=0.0, observationRadius=100.0)),
    # this fork's 128 agent slots, small radius (large cells relative to the boxes)
    ("fork128_r20", [SCENE_407, SCENE_4], 128, dict(polylineReductionThreshold=0.1, observationRadius=20.0)),
]


@pytest.mark.parametrize("name,scenes,A,over", BEV_CASES, ids=[c[0] for c in BEV_CASES])
def test_bev_parity(oracle_mod, name, scenes, A, over):
    kw = dict(collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    kw.update(over)
    gpu = P.make_gpu_sim(scenes, max_agents=A, enable_bev=True, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=A, enableBev=1, **kw)
    rng = np.random.default_rng(12)
    for k in range(4):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
        painted = P.compare_bev(gpu, orc)
        assert painted > 0.0003  # something was painted (200 half-metre segments cover very little of a 200 m view)
    gpu.close()


def test_packed_observations_match_reference_python_golden():
    """the fused pack kernel against golden vectors produced by the reference's own
    gpudrive/datatypes code (tests/golden/make_obs_pack_golden.py, run in the authoring container)."""
    import os
    import torch
    from tests.conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "obs_pack_golden.npz"))
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    gpu = P.make_gpu_sim([SCENE_4], max_agents=64, **kw)
    n = g["self_obs"].shape[0]
    dev = gpu.self_observation_tensor().to_torch().device
    # the exported tensors are the live storage: overwrite the first n agent rows with the golden inputs
    gpu.self_observation_tensor().to_torch()[0, :n] = torch.from_numpy(g["self_obs"]).to(dev)
    gpu.partner_observations_tensor().to_torch()[0, :n] = torch.from_numpy(g["partner"]).to(dev)
    gpu.agent_roadmap_tensor().to_torch()[0, :n] = torch.from_numpy(g["roadmap"]).to(dev)
    out = gpu.packed_observations()
    assert out.shape == (1, 64, 6 + 63 * 6 + 200 * 13)
    got = out[0, :n].cpu().numpy()
    # torch CPU divides, the kernel divides: bit-exact expected; allow one ulp of a [-1,1] value
    assert np.allclose(got, g["expected"], atol=1.2e-7, rtol=0), np.abs(got - g["expected"]).max()
    # and against a torch restatement on the device for every agent slot (incl. padding rows)
    so = gpu.self_observation_tensor().to_torch()
    po = gpu.partner_observations_tensor().to_torch()
    ro = gpu.agent_roadmap_tensor().to_torch()
    nm = lambda x: 2 * ((x + 1000) / 2000) - 1
    ego = torch.stack([so[..., 0] / 100, so[..., 1] * 0.7 / 30, so[..., 2] * 0.7 / 15, nm(so[..., 4]), nm(so[..., 5]),
                       so[..., 6]], -1)
    part = torch.stack([po[..., 0] / 100, nm(po[..., 1]), nm(po[..., 2]), po[..., 3] / (2 * np.pi),
                        po[..., 4] * 0.7 / 30, po[..., 5] * 0.7 / 15], -1).flatten(2)
    road = torch.cat([nm(ro[..., 0:1]), nm(ro[..., 1:2]), ro[..., 2:5] / 100, ro[..., 5:6] / (2 * np.pi),
                      torch.nn.functional.one_hot(ro[..., 6].long(), 7).float()], -1).flatten(2)
    ref = torch.cat([ego, part, road], -1)
    assert torch.allclose(out, ref, atol=3e-7, rtol=1e-6)  # GPU torch multiplies by 1/scalar: last-bit differences
    gpu.close()


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_expert_actions_match_golden_and_oracle(oracle_mod, model):
    """gd_expert_actions bit for bit against (a) the golden vectors produced with the
    reference's own LogTrajectory and (b) the oracle on every agent slot of two real scenes."""
    import os
    import torch
    from tests.conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "expert_actions_golden.npz"))
    kw = dict(polylineReductionThreshold=0.1, dynamicsModel=model, **ALL_OBJECTS)
    gpu = P.make_gpu_sim([SCENE_4, SCENE_407], max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, [SCENE_4, SCENE_407], max_agents=64, **kw)
    same = lambda a, b: a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for got, exp in zip(gpu.expert_actions(), orc.expert_actions()):
        assert same(got.cpu().numpy(), exp)
    # the exported trajectory tensor is live storage: overwrite rows with the golden input
    traj = gpu.expert_trajectory_tensor().to_torch()
    n = g["raw"].shape[1]
    traj[0, :n] = torch.from_numpy(g["raw"][0]).to(traj.device)
    act, pos, vel, yaw, valids = [t.cpu().numpy() for t in gpu.expert_actions()]
    name = {0: "classic", 1: "classic", 2: "delta_local", 3: "state"}[model]
    assert same(act[:1, :n], g[name + "_actions"])
    assert same(pos[:1, :n], g["pos_xy"]) and same(vel[:1, :n], g["vel_xy"]) and same(yaw[:1, :n], g["yaw"])
    assert np.array_equal(valids[:1, :n], g["valids"])
    gpu.close()


@pytest.mark.parametrize("model", [0, 1, 2, 3])
def test_log_playback_matches_oracle(oracle_mod, model):
    """advance_log_playback(k) on the device == the reference's Python loop (copy the step's expert action
    into the action tensor, step) restated on the oracle."""
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, dynamicsModel=model, collisionBehaviour=2,
              **ALL_OBJECTS)
    gpu = P.make_gpu_sim([SCENE_4, TEST_JSON], max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, [SCENE_4, TEST_JSON], max_agents=64, **kw)
    k = 12
    gpu.advance_log_playback(k)
    orc.advance_log_playback(k)
    gpu.sync()
    assert np.array_equal(gpu.action_tensor().to_torch().cpu().numpy().view(np.uint32),
                          np.array(orc.action_tensor()).view(np.uint32))
    P.compare_ints(gpu, orc)
    P.compare_state(gpu, orc)
    P.inject_and_compare(gpu, orc)
    with pytest.raises(ValueError):
        gpu.advance_log_playback(91)
    gpu.close()


def test_scene_cache_gives_identical_simulator(tmp_path):
    """a simulator built (and re-mapped with set_maps) from .gdsm caches holds the same
    bytes in every exported tensor as one built from the JSON scenes."""
    from gpudrive_lab_amd import scene_cache
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, **ALL_OBJECTS)
    scenes = [SCENE_4, SCENE_407, TEST_JSON, SCENE_4]
    cached = scene_cache.build_cache(scenes, 0.1, out_dir=str(tmp_path))
    a = P.make_gpu_sim(scenes, max_agents=64, **kw)
    b = P.make_gpu_sim(cached, max_agents=64, **kw)
    import torch
    names = list(P.INT_TENSORS) + list(P.OBS_TENSORS) + list(P.STATIC_FLOAT_TENSORS)

    def same():
        for n in names:
            x, y = getattr(a, n)().to_torch(), getattr(b, n)().to_torch()
            assert x.shape == y.shape and torch.equal(x.reshape(-1).view(torch.int32), y.reshape(-1).view(torch.int32)), n
    same()
    for sim in (a, b):
        sim.step()
    same()
    a.set_maps(scenes[::-1])
    b.set_maps(cached[::-1])
    same()
    a.close(); b.close()


@pytest.mark.parametrize("reward_type,behaviour", [("weighted_combination", 0), ("sparse_on_goal_achieved", 2)])
def test_episode_tracker_matches_oracle(oracle_mod, reward_type, behaviour):
    """the device-side episode bookkeeping (one kernel + device-driven reset per step)
    against the statement-by-statement restatement of PufferGPUDrive.step (oracle/episode.py) on the same
    seeded actions: per-step outputs, running trackers and per-world episode statistics bit-exact, the
    simulator tensors after the asynchronous resets within the usual parity bounds."""
    from gpudrive_lab_amd.episode import EpisodeTracker
    from oracle.episode import OracleEpisodeTracker
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=behaviour, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, maxNumControlledAgents=3, isStaticAgentControlled=0,
              initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
    scenes = [SCENE_4, SCENE_407, TEST_JSON, SCENE_4, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    gt = EpisodeTracker(gpu, reward_type=reward_type, collision_weight=-0.75, goal_achieved_weight=1.0, off_road_weight=-0.5)
    ot = OracleEpisodeTracker(orc, reward_type=reward_type, collision_weight=-0.75, goal_achieved_weight=1.0,
                              off_road_weight=-0.5)
    assert np.array_equal(gt.controlled_agent_mask.cpu().numpy(), ot.controlled_agent_mask)
    rng = np.random.default_rng(3)
    finished = 0
    bits = lambda x: np.ascontiguousarray(x, np.float32).view(np.uint32)
    for k in range(130):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        act[..., 0] = np.abs(act[..., 0])  # accelerate: collisions, goals and off-road events all happen
        P.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        o_rew, o_term, o_trunc, o_mask, o_done = ot.step()
        g_rew, g_term, g_trunc, g_mask = [t.cpu().numpy() for t in gt.step()]
        try:
            assert np.array_equal(bits(g_rew), bits(o_rew)), "rewards"
            assert np.array_equal(g_term, o_term) and np.array_equal(g_trunc, o_trunc) and np.array_equal(g_mask, o_mask)
            assert np.array_equal(gt.done_worlds.cpu().numpy(), o_done), "done worlds"
            for n in ("agent_episode_returns", "episode_lengths", "collided_in_episode", "offroad_in_episode"):
                assert np.array_equal(bits(getattr(gt, n).cpu().numpy()), bits(getattr(ot, n))), n
            assert np.array_equal(gt.live_agent_mask.cpu().numpy(), ot.live_agent_mask)
            done = np.flatnonzero(o_done)
            finished += len(done)
            assert np.array_equal(bits(gt.world_stats.cpu().numpy()[done]), bits(ot.world_stats[done])), "episode statistics"
            # the simulator after the device-driven reset of the finished worlds
            P.compare_ints(gpu, orc, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
            P.compare_state(gpu, orc)
            P.inject_and_compare(gpu, orc)
        except AssertionError as e:
            raise AssertionError("step %d: %s" % (k + 1, e))
    assert finished >= 5  # 91-step timeouts at least; early terminations with AgentStop
    stats = gt.pop_stats()
    tot = ot.world_stats  # last finished episode per world only; totals are checked through the counters
    assert stats["num_completed_episodes"] == finished and 0.0 <= stats["perc_truncated"] <= 1.0
    assert stats["total_controlled_agents"] == int(ot.controlled_agent_mask.sum()) and tot.shape[1] == 12
    gpu.close()


def test_cross_tensor_invariants_on_the_hip_path():
    """The column contract (gpudrive_lab_amd/columns.py) on the device tensors: partner and road rows are the
    ego-frame image of the absolute rows (tests/test_columns.py, no oracle involved)."""
    from tests.test_columns import check_cross_tensor_invariants, check_road_selection_by_brute_force
    kw = dict(polylineReductionThreshold=0.1, observationRadius=40.0, collisionBehaviour=2, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    gpu = P.make_gpu_sim([TEST_JSON, SCENE_4], max_agents=64, **kw)
    rng = np.random.default_rng(9)
    for _ in range(5):
        P.write_actions(gpu, P.random_actions(rng, 2, 64, 0))
        gpu.step()
    as_np = lambda t: t.to_torch().cpu().numpy()
    n_pairs, n_roads = check_cross_tensor_invariants(gpu, as_numpy=as_np)
    assert n_pairs > 500 and n_roads > 1500
    assert check_road_selection_by_brute_force(gpu, 40.0, as_numpy=as_np) == 25 + 64  # k-NN set, no oracle
    gpu.close()


EDGE_SCENES = [
    # (name, n_agents, n_polylines, pts_per_polyline)  ->  road entities = n_polylines * (pts - 1)
    ("roads_below_K", 5, 3, 51),         # R = 150 < K: no heap, radius filter + zero fill only
    ("roads_equal_K", 7, 4, 51),         # R = 200 == K: make_heap, no inserts
    ("roads_K_plus_1", 7, 1, 202),       # R = 201: exactly one candidate
    ("no_roads", 4, 0, 2),               # empty road list
    ("single_agent", 1, 8, 65),          # N = 1: all partner rows are the id -2 padding
    ("more_objects_than_slots", 90, 6, 65),   # 90 objects, 64 agent slots
    ("road_cap", 6, 40, 301),            # 12,000 segments: capped at kMaxRoadEntityCount = 10,000
    ("fork128_full", 128, 16, 129),      # A = 128 slots all live (140 objects): both 64-agent workgroups of a world busy
]


@pytest.mark.parametrize("name,n_agents,n_poly,pts", EDGE_SCENES, ids=[c[0] for c in EDGE_SCENES])
@pytest.mark.parametrize("road_alg", [0, 1], ids=["knn", "linear"])
def test_edge_case_scenes(oracle_mod, tmp_path, name, n_agents, n_poly, pts, road_alg):
    """Ragged / empty / maximum-size inputs, both road algorithms, reference row order."""
    import json
    from gpudrive_lab_amd import synth
    path = tmp_path / (name + ".json")
    A = 128 if name.startswith("fork128") else 64
    n_objects = n_agents + 12 if name.startswith("fork128") else n_agents
    path.write_text(json.dumps(synth.make_scene(17, n_agents=n_objects, n_polylines=n_poly, pts_per_polyline=pts)))
    kw = dict(polylineReductionThreshold=0.0, observationRadius=40.0, collisionBehaviour=0, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, roadObservationAlgorithm=road_alg, **ALL_OBJECTS)
    scenes = [str(path), SCENE_407]  # ragged batch: the edge world next to an ordinary one
    gpu = P.make_gpu_sim(scenes, max_agents=A, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=A, **kw)
    shape = np.asarray(orc.shape_tensor())
    assert shape[0, 0] == min(n_agents, A) and shape[0, 1] == min(n_poly * (pts - 1), 10000)
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 4, 0, seed=21)
    gpu.close()


def test_parity_on_a_side_stream_with_graph_replay(oracle_mod):
    """On a non-default torch stream the engine captures the step into a hipGraph and replays it;
    results must not change (and the stream hand-over through gd_set_stream must work)."""
    import torch
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    scenes = [SCENE_4, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)      # built on the default stream
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    with torch.cuda.stream(torch.cuda.Stream()):
        P.lockstep(gpu, orc, 12, 0, seed=31)               # first step captures, the rest replay
        gpu.set_maps([TEST_JSON, SCENE_407])               # rebuild drops the graph
        orc.set_maps([TEST_JSON, SCENE_407])
        P.compare_fresh(gpu, orc)
        P.lockstep(gpu, orc, 4, 0, seed=32)
    P.lockstep(gpu, orc, 3, 0, seed=33)                    # back on the default stream
    gpu.close()


def test_free_running_flags_stay_exact(oracle_mod):
    """No teacher forcing: 91 steps + reset + 30 steps; int tensors must stay bit-exact."""
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    scenes = [SCENE_4, SCENE_407, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    P.lockstep(gpu, orc, 91, 0, seed=7, teacher_force=False)
    gpu.reset([0, 2])
    orc.reset([0, 2])
    P.compare_ints(gpu, orc)
    P.compare_state(gpu, orc)
    P.lockstep(gpu, orc, 30, 0, seed=8, teacher_force=False)
    gpu.close()


def test_partial_reset_set_maps_delete_agents(oracle_mod):
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    scenes = [SCENE_4, SCENE_407, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    P.lockstep(gpu, orc, 5, 0, seed=1)
    # reset(int), reset(ndarray) (gpudrive sb3_wrapper.py:163, env_puffer.py:376)
    gpu.reset(1)
    orc.reset(1)
    P.compare_fresh(gpu, orc)
    gpu.reset(np.array([0, 2]))
    orc.reset([0, 2])
    P.compare_fresh(gpu, orc)
    # set_maps: len must match; worlds rebuilt; views stay valid
    view = gpu.shape_tensor().to_torch()
    with pytest.raises(ValueError):
        gpu.set_maps([TEST_JSON])
    new = [TEST_JSON, SCENE_4, SCENE_407]
    gpu.set_maps(new)
    orc.set_maps(new)
    assert view.data_ptr() == gpu.shape_tensor().to_torch().data_ptr()
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 3, 0, seed=2)
    # deleteAgents
    ids = np.asarray(orc.agent_id_tensor())
    victims = {1: [int(ids[1, 0]), int(ids[1, 5])], 2: [int(ids[2, 1])]}
    gpu.deleteAgents(victims)
    orc.deleteAgents(victims)
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 3, 0, seed=3)
    with pytest.raises(FileNotFoundError):
        gpu.set_maps(["/nonexistent.json"] * 3)
    P.lockstep(gpu, orc, 2, 0, seed=4)  # engine still usable after a failed set_maps
    gpu.close()


def test_in_place_action_writes_and_aliasing():
    """Python mutates the exported action tensor in place (env_torch.py:645-664)."""
    import torch
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    gpu = P.make_gpu_sim([SCENE_407], max_agents=64, **kw)
    a1 = gpu.action_tensor().to_torch()
    a2 = gpu.action_tensor().to_torch()
    assert a1.data_ptr() == a2.data_ptr() and a1.is_cuda
    before = gpu.absolute_self_observation_tensor().to_torch().clone()
    a1[:, :, :3].copy_(torch.tensor([2.0, 0.0, 0.0], device=a1.device).expand(1, 64, 3))
    gpu.step()
    after = gpu.absolute_self_observation_tensor().to_torch()
    ctrl = gpu.controlled_state_tensor().to_torch()[0, :, 0] == 1
    assert ctrl.any()
    assert (before[0, ctrl, :2] != after[0, ctrl, :2]).any()
    gpu.close()


# ---- BASELINE.json full sizes: size-independent properties ----
def _tiled(n):
    base = [TEST_JSON, SCENE_407, SCENE_4]
    return [base[i % 3] for i in range(n)]


@pytest.mark.parametrize("W,collision", [(1024, 2), (4096, 0)])
def test_full_size_properties(W, collision):
    """configs[1] (1024 worlds) and configs[2] (4096 worlds, collisions on): worlds tiled from the
    same scene and fed the same actions must stay identical replicas (checksum of checksums);
    k-NN rows must be within the radius, unique, and no unselected road may be closer than a
    selected one."""
    import torch
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=collision, rewardType=1,
              distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)
    gpu = P.make_gpu_sim(_tiled(W), max_agents=64, **kw)
    g = torch.Generator(device="cpu").manual_seed(0)
    act3 = torch.zeros(3, 64, 10)
    for step in range(12):
        act3[..., 0] = torch.rand(3, 64, generator=g) * 5 - 3
        act3[..., 1] = torch.rand(3, 64, generator=g) * 1.4 - 0.7
        a = gpu.action_tensor().to_torch()
        a.copy_(act3.repeat((W + 2) // 3, 1, 1)[:W].to(a.device))
        gpu.step()
    names = ["done_tensor", "info_tensor", "reward_tensor", "self_observation_tensor",
             "absolute_self_observation_tensor", "partner_observations_tensor", "agent_roadmap_tensor"]
    for name in names:
        t = getattr(gpu, name)().to_torch()
        flat = t.reshape(W, -1)
        for r in range(3):
            grp = flat[r::3]
            assert torch.equal(grp, grp[0:1].expand_as(grp)), "%s: replicas of scene %d diverged" % (name, r)
    # k-NN selection properties on world 0..2
    am = gpu.agent_roadmap_tensor().to_torch()[:3].cpu().numpy()
    shape = gpu.shape_tensor().to_torch()[:3].cpu().numpy()
    mo = gpu.map_observation_tensor().to_torch()[:3].cpu().numpy()
    ab = gpu.absolute_self_observation_tensor().to_torch()[:3].cpu().numpy()
    for w in range(3):
        n, R = shape[w]
        for a in range(n):
            rows = am[w, a]
            valid = rows[:, 6] != 0
            d = np.hypot(rows[valid, 0], rows[valid, 1])
            assert (d <= 50.0 + 1e-4).all()
            # padding rows of the k-NN path are all-zero (src/knn.hpp:19-28)
            assert (rows[~valid] == 0).all()
            # every road within the radius that is closer than the farthest selected one is selected
            dist_all = np.hypot(mo[w, :R, 0] - ab[w, a, 0], mo[w, :R, 1] - ab[w, a, 1])
            within = np.sort(dist_all[dist_all <= 50.0 - 1e-3])
            if len(within) <= 200:
                assert valid.sum() >= len(within)
            else:
                assert valid.sum() == 200 and d.max() <= within[199] + 1e-3
    gpu.close()
