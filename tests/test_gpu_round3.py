"""GPU suite, round 3: the rank replay of the reference-order road selection (csrc/map_obs_rank.hip) against the history
replay on keys it falls back to and against the oracle; oracle cross-checks at BASELINE.json's full sizes (first and last
worlds of the batch, both workgroup generations, config 5); the road arrays' slack when a rebuild fits the old capacity;
the set-order kernel's rule for ties at the K-th key; the reference's OBB known-answer cases as worlds of two parked vehicles.
Everything goes through `madrona_gpudrive` -> ctypes -> the C ABI."""
import collections
import json

import numpy as np
import pytest

from gpudrive_lab_amd import synth
from tests import parity as P
from tests import ref_cases as RC
from tests.conftest import SCENE_4, SCENE_407, TEST_JSON

pytestmark = pytest.mark.gpu

ALL_OBJECTS = dict(isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
BENCH = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
             roadObservationAlgorithm=0, polylineReductionThreshold=0.0, **ALL_OBJECTS)
CLASSIC = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
               distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)


@pytest.fixture(scope="module")
def bench_scenes(tmp_path_factory):
    return synth.write_scenes(str(tmp_path_factory.mktemp("bench_scenes3")), list(range(8)))


def _dup_scene(tmp_path):
    """Three polylines twice: exactly equal keys among the candidates (tests/test_heap_pin.py builds the same scene)."""
    sc = synth.make_scene(5, n_agents=12, n_polylines=10, pts_per_polyline=60)
    sc["roads"] = sc["roads"] + [dict(r, id=100 + i) for i, r in enumerate(sc["roads"][:3])]
    p = tmp_path / "dup.json"
    p.write_text(json.dumps(sc))
    return str(p)


def _bits(t):
    return RC.as_np(t).view(np.uint32)


@pytest.mark.parametrize("which", ["bench", "waymo", "waymo_128_slots", "waymo_agent_stop", "waymo_unreduced", "equal_keys"])
def test_rank_replay_equals_history_replay_bit_for_bit(bench_scenes, tmp_path, monkeypatch, which):
    """Two engines on the same scenes and actions, free running: one selects the roads with the rank replay, the other
    with GPUDRIVE_NO_RANK_REPLAY=1 (k_map_obs alone, round 2's kernel).  agent_roadmap_tensor must be bit-identical at
    every step -- through a reset of some worlds, a jump of a few agents (their bounds no longer hold) and agents that
    finish and are parked at the padding position -- and the rank replay must actually have been taken."""
    min_taken, slots = 0.6, 64
    if which == "bench":
        scenes, kw = bench_scenes[:3], BENCH
    elif which == "waymo":
        scenes, kw = [TEST_JSON, SCENE_407, SCENE_4], CLASSIC   # a few hundred roads per world: every road may be a candidate
    elif which == "waymo_128_slots":
        scenes, kw, slots = [TEST_JSON, SCENE_407, SCENE_4], CLASSIC, 128   # the kernels' 128-slot instantiations (two halves per world)
    elif which == "waymo_agent_stop":
        # BASELINE configs[2]'s rules.  Agents come back from the padding position with checkpoints that were recorded out
        # there (every road a candidate, all of them far below the recorded K-th key): the ranking's buckets then go by
        # the smallest and largest key (k_knn_rank, `jumped`)
        scenes, kw = [TEST_JSON, SCENE_407, SCENE_4], dict(CLASSIC, collisionBehaviour=0)
    elif which == "waymo_unreduced":
        # thousands of roads with repeated points (equal keys) and more inserts than the candidate buffer holds for many
        # agents: most groups take the fallback, all of it must still be bit-identical
        scenes, kw, min_taken = [TEST_JSON, SCENE_407, SCENE_4], dict(CLASSIC, polylineReductionThreshold=0.0, observationRadius=60.0), 0.0
    else:
        # SCENE_407 unreduced next to it: its agents mostly overflow the candidate buffer (thousands of roads, long insert
        # histories) and take the fallback, the duplicated-polyline world is ranked with its equal keys
        scenes, kw, min_taken = [_dup_scene(tmp_path), SCENE_407], dict(BENCH, observationRadius=200.0), 0.4
    monkeypatch.setenv("GPUDRIVE_RANK_MIN_ROADS", "200")   # small worlds too (by default k_map_obs keeps those)
    monkeypatch.setenv("GPUDRIVE_RANK_MAX_ROADS", "20000")  # and the largest (by default left to k_map_obs as well)
    fast = P.make_gpu_sim(scenes, max_agents=slots, **kw)
    monkeypatch.setenv("GPUDRIVE_NO_RANK_REPLAY", "1")
    slow = P.make_gpu_sim(scenes, max_agents=slots, **kw)
    monkeypatch.delenv("GPUDRIVE_NO_RANK_REPLAY")
    assert fast.stat(7) == 1 and slow.stat(7) == 0
    assert (slow.debug_road_path() == -2).all()
    W = len(scenes)
    rng = np.random.default_rng(5)
    taken = fell_back = 0
    why = collections.Counter()
    live = RC.as_np(fast.shape_tensor())[:, 0]
    n_live = int(live.sum())

    def same(tag):
        a, b = _bits(fast.agent_roadmap_tensor()), _bits(slow.agent_roadmap_tensor())
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError("%s: agent_roadmap differs at %d places, first %s" % (tag, len(bad), bad[0]))
        for name in ("done_tensor", "info_tensor", "self_observation_tensor"):
            assert np.array_equal(_bits(getattr(fast, name)()), _bits(getattr(slow, name)())), (tag, name)
    same("t=0")
    for k in range(24):
        act = P.random_actions(rng, W, slots, 0)
        RC.write_actions(fast, act)
        RC.write_actions(slow, act)
        fast.step()
        slow.step()
        same("step %d" % (k + 1))
        path = fast.debug_road_path()
        taken += int(((path > 0) | (path == -3)).sum())   # ranked, or parked out of reach of every road (no rows)
        fell_back += int(((path == -1) | (path <= -10)).sum())
        why.update(path[path <= -10].tolist())
        if k == 8:  # some worlds go back to their start poses
            fast.reset([0])
            slow.reset([0])
            same("reset")
        if k == 14:  # a few agents jump 30 m: last step's bounds say nothing about where they are now
            st = fast.debug_get_state()
            st[:, :3, 0] += 30.0
            fast.debug_set_state(st)
            slow.debug_set_state(st)
            fast.reset([])
            slow.reset([])
            same("jump")
    print("rank replay taken for %d of %d agent-steps, fallback for %d (own reasons: %s)" % (taken, 24 * n_live, fell_back, dict(why)))
    assert taken >= min_taken * 24 * n_live, "the rank replay was taken for only %d of %d agent-steps" % (taken, 24 * n_live)
    fast.close()
    slow.close()


def test_rank_replay_lockstep_with_the_oracle_through_a_whole_episode(oracle_mod, bench_scenes):
    """91 free-running steps on the bench scenes (agents finish and are parked along the way), then the reset: ints exact at
    every step, every observation against the oracle under teacher forcing every 7th step."""
    scenes = bench_scenes[:2]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **BENCH)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **BENCH)
    P.lockstep(gpu, orc, 91, 0, seed=77, check_every=7)
    path = gpu.debug_road_path()
    assert (path[path != 0] != -1).all(), "no agent should need the fallback in the steady state"
    gpu.reset([0, 1])
    orc.reset([0, 1])
    P.compare_fresh(gpu, orc)
    assert (gpu.debug_road_path()[:, :64] > 0).all(), "a reset goes back to the episode's first checkpoints, not to the fallback"
    P.lockstep(gpu, orc, 5, 0, seed=78)
    gpu.close()


def test_partner_rows_on_the_second_stream(oracle_mod, monkeypatch):
    """GPUDRIVE_SPLIT_PARTNER=1 (off by default: measured slower): the partner rows are written by
    k_partner_rows on the engine's second stream, forked and joined inside the captured step graph.  Same rows, on the null
    stream (where the switch is ignored) and on a side stream through graph replay and plain reset passes."""
    import torch
    monkeypatch.setenv("GPUDRIVE_SPLIT_PARTNER", "1")
    scenes = [SCENE_4, TEST_JSON, SCENE_407]
    with torch.cuda.stream(torch.cuda.Stream()):
        gpu = P.make_gpu_sim(scenes, max_agents=64, **CLASSIC)
        orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **CLASSIC)
        P.compare_fresh(gpu, orc)
        P.lockstep(gpu, orc, 6, 0, seed=41)
        assert gpu.stat(0) > 0, "the step graph (with its fork and join) was not replayed"
        gpu.reset([1])
        orc.reset([1])
        P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 2, 0, seed=42)   # back on the null stream: written by the state step again
    gpu.close()


# ---- BASELINE.json full sizes against the oracle ----
def _tiled(n):
    base = [TEST_JSON, SCENE_407, SCENE_4]
    return [base[i % 3] for i in range(n)]


class _Worlds:
    """A few worlds of a big simulator seen as a small one (what tests/parity.py's comparisons take)."""

    def __init__(self, sim, idx):
        self._sim, self._idx = sim, list(idx)

    def __getattr__(self, name):
        if not name.endswith("_tensor"):
            raise AttributeError(name)
        full = getattr(self._sim, name)

        def getter():
            return RC.as_np(full())[self._idx]
        return getter


@pytest.mark.parametrize("W,collision,knn_order", [(1024, 2, 0), (4096, 0, 0), (1024, 2, 1)])
def test_full_size_first_and_last_worlds_equal_the_oracle(oracle_mod, W, collision, knn_order):
    """configs[1] / configs[2] at full size: the worlds are replicas of three scenes fed the same per-scene actions, so
    three oracle worlds describe all of them.  Worlds 0-2 and the LAST three worlds of the batch are compared with the
    oracle elementwise (ints exact, observations under teacher forcing), and every replica must equal its scene's first
    world bit for bit -- which carries the comparison to all W worlds."""
    import torch
    kw = dict(CLASSIC, collisionBehaviour=collision)
    gpu = P.make_gpu_sim(_tiled(W), max_agents=64, knn_order=knn_order, **kw)
    orc = P.make_oracle_sim(oracle_mod, _tiled(3), max_agents=64, **kw)
    rng = np.random.default_rng(3)
    last = [W - 3, W - 2, W - 1]
    last_scene = [w % 3 for w in last]
    views = [(_Worlds(gpu, [0, 1, 2]), [0, 1, 2]), (_Worlds(gpu, last), last_scene)]
    for step in range(13):
        act3 = P.random_actions(rng, 3, 64, 0)
        a = gpu.action_tensor().to_torch()
        a.copy_(torch.as_tensor(act3).repeat((W + 2) // 3, 1, 1)[:W].to(a.device))
        np.copyto(orc.action_tensor(), act3)
        gpu.step()
        orc.step()
        for view, scene in views:
            sub = _Worlds(orc, scene)
            P.compare_ints(view, sub, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        if step % 4 == 0:  # teacher forcing: the oracle's state into every replica, both recompute, observations compared
            st = orc.get_state()
            gpu.debug_set_state(np.tile(st, ((W + 2) // 3, 1, 1))[:W])
            gpu.reset([])
            orc.reset([])
            for view, scene in views:
                sub = _Worlds(orc, scene)
                sub.W, sub.A = 3, 64
                if knn_order == 0:
                    P.compare_obs(view, sub)
                else:
                    P.compare_obs(view, sub, names=[n for n in P.OBS_TENSORS if n != "agent_roadmap_tensor"])
                    P.compare_roadmap_as_set(view, sub)
    for name in ["done_tensor", "info_tensor", "reward_tensor", "self_observation_tensor", "partner_observations_tensor",
                 "agent_roadmap_tensor"]:
        flat = getattr(gpu, name)().to_torch().reshape(W, -1)
        for r in range(3):
            grp = flat[r::3]
            assert torch.equal(grp, grp[0:1].expand_as(grp)), "%s: replicas of scene %d diverged" % (name, r)
    gpu.close()


@pytest.mark.parametrize("knn_order", [0, 1], ids=["reference_order", "set_order"])
def test_1024_bench_worlds_spread_over_both_generations_equal_the_oracle(oracle_mod, bench_scenes, knn_order):
    """What bench.py times, at its size: 1024 worlds tiled from the 8 synthetic scenes, 14 steps (the reference-order
    kernels re-sort their launch orders after every launch), the oracle on the 8 distinct scenes, 16 worlds spread over the
    whole batch compared elementwise (set order: the road rows as a set)."""
    import torch
    W = 1024
    scenes = [bench_scenes[i % 8] for i in range(W)]
    gpu = P.make_gpu_sim(scenes, max_agents=64, knn_order=knn_order, **BENCH)
    orc = P.make_oracle_sim(oracle_mod, bench_scenes, max_agents=64, **BENCH)
    picks = sorted(set(int(x) for x in np.linspace(0, W - 1, 16)))
    view, sub_scene = _Worlds(gpu, picks), [w % 8 for w in picks]
    rng = np.random.default_rng(9)
    for step in range(14):
        act8 = P.random_actions(rng, 8, 64, 0)
        a = gpu.action_tensor().to_torch()
        a.copy_(torch.as_tensor(act8).repeat(W // 8, 1, 1).to(a.device))
        np.copyto(orc.action_tensor(), act8)
        gpu.step()
        orc.step()
        sub = _Worlds(orc, sub_scene)
        P.compare_ints(view, sub, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        if step in (0, 5, 13):
            gpu.debug_set_state(np.tile(orc.get_state(), (W // 8, 1, 1)))
            gpu.reset([])
            orc.reset([])
            sub = _Worlds(orc, sub_scene)
            sub.W, sub.A = len(picks), 64
            if knn_order == 0:
                P.compare_obs(view, sub)
            else:
                P.compare_obs(view, sub, names=[n for n in P.OBS_TENSORS if n != "agent_roadmap_tensor"])
                P.compare_roadmap_as_set(view, sub)
    if knn_order == 0:
        path = gpu.debug_road_path()
        assert (path > 0).mean() > 0.9, "the rank replay should carry the bench scenes"
    gpu.close()


def test_config5_lidar_at_1024_worlds(oracle_mod):
    """configs[4] at its size: LiDAR rows of replicas identical, worlds 0-2 and the last three equal to the (modelled,
    parity unpinned) oracle."""
    import torch
    W = 1024
    kw = dict(CLASSIC, enableLidar=1)
    gpu = P.make_gpu_sim(_tiled(W), max_agents=64, lidar_half_angle=float(np.pi), **kw)
    orc = P.make_oracle_sim(oracle_mod, _tiled(3), max_agents=64, lidarHalfAngle=float(np.pi), **kw)
    rng = np.random.default_rng(11)
    last = [W - 3, W - 2, W - 1]
    for step in range(4):
        act3 = P.random_actions(rng, 3, 64, 0)
        a = gpu.action_tensor().to_torch()
        a.copy_(torch.as_tensor(act3).repeat((W + 2) // 3, 1, 1)[:W].to(a.device))
        np.copyto(orc.action_tensor(), act3)
        gpu.step()
        orc.step()
        gpu.debug_set_state(np.tile(orc.get_state(), ((W + 2) // 3, 1, 1))[:W])
        gpu.reset([])
        orc.reset([])
    for idx in ([0, 1, 2], last):
        sub = _Worlds(orc, [w % 3 for w in idx])
        sub.W, sub.A = 3, 64
        P.compare_lidar(_Worlds(gpu, idx), sub)
    flat = gpu.lidar_tensor().to_torch().reshape(W, -1)
    live = torch.as_tensor(np.arange(64)[None, :] < np.asarray(orc.shape_tensor())[:, 0:1])  # [3, 64]
    per_agent = gpu.lidar_tensor().to_torch().reshape(W, 64, -1)
    for r in range(3):
        grp = per_agent[r::3][:, live[r]]
        assert torch.equal(grp, grp[0:1].expand_as(grp)), "lidar rows of scene %d's replicas diverged" % r
    assert flat.shape[0] == W
    gpu.close()


# ---- road arrays: readable slack behind the last world on every rebuild ----
def test_rebuild_into_existing_capacity_keeps_the_pad(oracle_mod, tmp_path):
    """k_map_obs prefetches up to 256 roads past a world's last one and the fused set-order write-out reads the first
    record of a world without roads: the arrays must end in readable pad entries also when set_maps / deleteAgents
    rebuild into the capacity of an earlier, larger road set, and when no world has a road at all."""
    def scene(name, n_poly, pts, seed):
        p = tmp_path / (name + ".json")
        p.write_text(json.dumps(synth.make_scene(seed, n_agents=6, n_polylines=n_poly, pts_per_polyline=pts)))
        return str(p)
    big = scene("big", 8, 129, 1)        # 1024 roads
    small = scene("small", 2, 65, 2)     # 128 roads: fits the old capacity with room to spare
    edge = scene("edge", 8, 141, 3)      # 1120 roads: above `big`, inside its capacity + slack
    none = scene("none", 0, 2, 4)        # no roads at all
    kw = dict(BENCH, observationRadius=60.0)
    for order in (0, 1):
        gpu = P.make_gpu_sim([big, big], max_agents=64, knn_order=order, **kw)
        orc = P.make_oracle_sim(oracle_mod, [big, big], max_agents=64, **kw)
        for new in ([small, small], [edge, edge], [none, small], [none, none], [big, none]):
            gpu.set_maps(new)
            orc.set_maps(new)
            if order == 0:
                P.compare_fresh(gpu, orc)
                P.lockstep(gpu, orc, 2, 0, seed=5)
            else:
                P.compare_ints(gpu, orc)
                gpu.debug_set_state(orc.get_state())
                gpu.reset([])
                orc.reset([])
                P.compare_roadmap_as_set(gpu, orc)
        gpu.close()
    zero = P.make_gpu_sim([none], max_agents=64, **kw)   # constructed without a single road
    assert (RC.as_np(zero.agent_roadmap_tensor())[0, :6, :, 6] == 0).all()
    zero.step()
    zero.close()


# ---- set order: ties at the K-th key ----
def test_set_order_ties_at_the_kth_key(oracle_mod, tmp_path):
    """Duplicated polylines put pairs of roads at exactly the same distance.  When such a pair straddles the K-th place
    (both have the K-th key, only one fits) the reference keeps whichever the heap's history left in the array
    (src/knn.hpp:15-17, 138-151: an arriving road with a key EQUAL to the top is not inserted, but two equal keys already
    inside are evicted in heap order, which is neither index order) and the set-order kernel keeps the LOWEST ROAD INDEX.
    What a consumer can see of that, pinned here: the two row sets are equal in every column except the id of such a tied
    row (column 7); a differing id always belongs to a road whose key equals the agent's K-th key, i.e. to the twin of the
    road the reference kept -- same position, size, heading, type, map type.  Both outcomes occur (the reference keeps the
    earlier road for some agents, the later one for others)."""
    sc = synth.make_scene(5, n_agents=16, n_polylines=12, pts_per_polyline=60)
    sc["roads"] = sc["roads"] + [dict(r, id=100 + i) for i, r in enumerate(sc["roads"][:6])]  # six polylines twice
    p = tmp_path / "dup_set.json"
    p.write_text(json.dumps(sc))
    kw = dict(BENCH, observationRadius=300.0)   # the radius keeps everything: K binds for every agent
    gpu = P.make_gpu_sim([str(p)], max_agents=64, knn_order=1, **kw)
    orc = P.make_oracle_sim(oracle_mod, [str(p)], max_agents=64, **kw)
    rng = np.random.default_rng(2)
    straddles = same_choice = other_choice = 0
    n = int(np.asarray(orc.shape_tensor())[0, 0])
    for step in range(6):
        if step:
            act = P.random_actions(rng, 1, 64, 0)
            RC.write_actions(gpu, act)
            np.copyto(orc.action_tensor(), act)
            gpu.step()
            orc.step()
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
        g_all = P._sorted_rows(RC.as_np(gpu.agent_roadmap_tensor()))[0]
        o_all = P._sorted_rows(np.asarray(orc.agent_roadmap_tensor()))[0]
        for a in range(n):
            g, o = g_all[a], o_all[a]
            cols = [0, 1, 2, 3, 4, 5, 6, 8]
            assert np.allclose(g[:, cols], o[:, cols], atol=P.OBS_ATOL, rtol=0), "agent %d: the row sets differ beyond a twin's id" % a
            obs = orc.road_obs_of(0, a)
            keys = obs[:, 0] * obs[:, 0] + obs[:, 1] * obs[:, 1]
            d2 = o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1]
            kth = d2[o[:, 6] != 0].max()
            tied = np.flatnonzero(keys == kth)
            if len(tied) > 1:
                straddles += 1
            diff = np.flatnonzero(g[:, 7] != o[:, 7])
            for r in diff:   # only the tied road's id may differ, and only to its twin's
                assert len(tied) > 1 and d2[r] == kth, "agent %d row %d: ids differ away from a tie at the K-th key" % (a, r)
                assert g[r, 7] in obs[tied, 7] and o[r, 7] in obs[tied, 7]
                assert g[r, 7] == obs[tied, 7].min() or g[r, 7] == obs[tied[0], 7]   # the kernel's rule: lowest road index
            if len(tied) > 1:
                other_choice += len(diff) > 0
                same_choice += len(diff) == 0
    assert straddles > 0, "no agent had duplicated roads at its K-th distance: the scene does not exercise the tie rule"
    print("ties at the K-th key: %d agent-steps; the reference kept the lowest road index in %d of them, the twin in %d"
          % (straddles, same_choice, other_choice))
    gpu.close()


# ---- the reference's OBB known-answer cases (tests/CollisionDetectionTests.cpp:11-85) on the device ----
def _two_boxes(tmp_path, name, pos_b, yaw_b, half_a, half_b):
    """Two parked vehicles whose boxes have the given half extents (the engine scales length / 2 and width / 2 by 0.7,
    src/level_gen.cpp), the first at the origin with yaw 0; one far road so that the world has a map."""
    def obj(i, x, y, yaw, half):
        return {"position": [{"x": 100.0 + x, "y": 50.0 + y, "z": 0.0}] * 91, "width": 2.0 * half[1] / 0.7,
                "length": 2.0 * half[0] / 0.7, "height": 1.6, "heading": [yaw] * 91, "velocity": [{"x": 0.0, "y": 0.0}] * 91,
                "valid": [True] * 91, "goalPosition": {"x": 400.0, "y": 400.0, "z": 0.0}, "type": "vehicle", "id": i,
                "mark_as_expert": False}
    sc = {"name": name, "scenario_id": name, "objects": [obj(0, 0.0, 0.0, 0.0, half_a), obj(1, pos_b[0], pos_b[1], yaw_b, half_b)],
          "roads": [{"geometry": [{"x": 300.0 + k, "y": 300.0, "z": 0.0} for k in range(8)], "type": "road_edge",
                     "map_element_id": 15, "id": 0}],
          "tl_states": {}, "metadata": {"sdc_track_index": 0, "objects_of_interest": [], "tracks_to_predict": []}}
    p = tmp_path / (name + ".json")
    p.write_text(json.dumps(sc))
    return str(p)


def test_obb_known_answer_cases_on_the_hip_path(oracle_mod, tmp_path):
    """Each case is a world of two parked vehicles; one step with zero actions; the collision flags of `info_tensor` must be
    what the reference's test expects and what the oracle computes from the same scene.  (The touching-corners case is
    compared with the oracle only: the half extents go through length / 2 * 0.7 in float.)"""
    import math
    cases = [("aligned", (1.0, 1.0), 0.0, (1.0, 1.0), (1.0, 1.0), True),
             ("apart", (2.0, 2.0), 0.0, (0.5, 0.5), (0.5, 0.5), False),
             ("corner", (1.0, 1.0), 0.0, (0.5, 0.5), (0.5, 0.5), None),
             ("inside", (0.0, 0.0), 0.0, (1.0, 1.0), (0.5, 0.5), True)]
    cases += [("rot%03d" % deg, (0.5, 0.5), math.radians(deg), (1.0, 1.0), (1.0, 1.0), True) for deg in range(0, 360, 15)]
    scenes = [_two_boxes(tmp_path, n, pb, yb, ha, hb) for n, pb, yb, ha, hb, _ in cases]
    kw = dict(CLASSIC, collisionBehaviour=2)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    zero = np.zeros((len(scenes), 64, 10), np.float32)
    RC.write_actions(gpu, zero)
    RC.write_actions(orc, zero)
    gpu.step()
    orc.step()
    P.compare_ints(gpu, orc)
    info = RC.as_np(gpu.info_tensor())
    for w, (name, _, _, _, _, expect) in enumerate(cases):
        hit = info[w, :2, 1] != 0   # column 1: collided with a vehicle (src/types.hpp Info)
        assert hit[0] == hit[1], name
        if expect is not None:
            assert bool(hit[0]) == expect, "%s: the device says %s" % (name, bool(hit[0]))
    gpu.close()
