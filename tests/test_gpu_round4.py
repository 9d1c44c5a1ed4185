"""GPU suite, round 4: the rank replay (csrc/map_obs_rank.hip) where the product uses it but no test had it --
world rebuilds (`set_maps` big -> small -> big, `deleteAgents`: `reset_rank_state` / `ensure_rank_buffers` into live rank
buffers), the device-driven reset of the episode tracker (the `gate_any` passes of all the rank kernels), and the ends of its
arrays (last world, last agent slot, a candidate list that fills the buffer exactly, a replay wave with one active lane).
Everything goes through `madrona_gpudrive` -> ctypes -> the C ABI; reference: src/knn.hpp:103-158, src/mgr.cpp:590-715,
gpudrive/env/env_puffer.py:250-403."""
import json
import math

import numpy as np
import pytest

from gpudrive_lab_amd import synth
from tests import parity as P
from tests import ref_cases as RC

pytestmark = pytest.mark.gpu

ALL_OBJECTS = dict(isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
BENCH = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
             roadObservationAlgorithm=0, polylineReductionThreshold=0.0, **ALL_OBJECTS)


def _scene(tmp_path, name, n_agents, n_poly, pts, seed):
    p = tmp_path / (name + ".json")
    p.write_text(json.dumps(synth.make_scene(seed, n_agents=n_agents, n_polylines=n_poly, pts_per_polyline=pts)))
    return str(p)


def _ranked(gpu):
    """(agents ranked in the last selection, agents that needed the fallback although their world has K roads or more)"""
    path = gpu.debug_road_path()
    big = (RC.as_np(gpu.shape_tensor())[:, 1] >= 200)[:, None]
    return int((path > 0).sum()), int((((path == -1) | (path <= -10)) & big).sum())


# ---- world rebuilds on the rank path ----
def test_rank_path_through_set_maps_and_delete_agents(oracle_mod, tmp_path, monkeypatch):
    """`set_maps` big -> small -> big and `deleteAgents` with the rank replay forced on for every world that has K roads:
    after each rebuild the first selection takes the fallback (no checkpoint survives), the following ones the rank path
    again -- in lockstep with the oracle all the way (ints exact, observations under teacher forcing)."""
    monkeypatch.setenv("GPUDRIVE_RANK_MIN_ROADS", "200")
    big_a = _scene(tmp_path, "big_a", 20, 12, 129, 11)    # 1536 roads
    big_b = _scene(tmp_path, "big_b", 64, 16, 129, 12)    # 2048 roads, every agent slot live
    mid = _scene(tmp_path, "mid", 9, 10, 60, 13)          # 590 roads
    small = _scene(tmp_path, "small", 6, 2, 65, 14)       # 128 roads: below K, never ranked
    kw = dict(BENCH, observationRadius=60.0)
    scenes = [big_a, big_b, mid]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    assert gpu.stat(7) == 1
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 4, 0, seed=1)
    ranked, fell = _ranked(gpu)
    assert ranked > 0 and fell == 0, (ranked, fell)
    seed = 2
    for new in ([small, mid, small], [big_b, big_a, big_b], [small, small, small], [mid, big_b, big_a]):
        gpu.set_maps(new)
        orc.set_maps(new)
        P.compare_fresh(gpu, orc)
        P.lockstep(gpu, orc, 4, 0, seed=seed)
        seed += 1
        ranked, fell = _ranked(gpu)
        has_k = any(s is not small for s in new)
        assert (ranked > 0) == has_k and fell == 0, (new, ranked, fell)
    # deleteAgents rebuilds two of the three worlds into the live rank buffers
    ids = np.asarray(orc.agent_id_tensor())
    victims = {1: [int(ids[1, 0]), int(ids[1, 63]), int(ids[1, 17])], 2: [int(ids[2, 3])]}
    gpu.deleteAgents(victims)
    orc.deleteAgents(victims)
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 5, 0, seed=9)
    ranked, fell = _ranked(gpu)
    assert ranked > 0 and fell == 0
    # a partial reset in between, then on
    gpu.reset([1])
    orc.reset([1])
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 3, 0, seed=10)
    gpu.close()


# ---- the device-driven reset (gate_any passes) on the rank path ----
def test_episode_tracker_auto_reset_on_the_rank_path(oracle_mod, tmp_path):
    """`EpisodeTracker(auto_reset)` -- what bench.py's `rl_loop` line runs every step -- on two bench-sized worlds (4096 and
    2048 roads: rank path by default) through more than two episode ends per world: every step launches the gated reset
    pass (all six rank kernels return at once unless a world finished), a finished world is reset on the device and its
    agents go back to the episode's first checkpoints.  Bookkeeping bit-exact against oracle/episode.py, the simulator in
    lockstep with the oracle."""
    from gpudrive_lab_amd.episode import EpisodeTracker
    from oracle.episode import OracleEpisodeTracker
    kw = dict(BENCH, collisionBehaviour=0, maxNumControlledAgents=4, isStaticAgentControlled=0)
    scenes = [synth.write_scenes(str(tmp_path), [3])[0], _scene(tmp_path, "half", 40, 16, 129, 21)]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    assert gpu.stat(7) == 1
    gt = EpisodeTracker(gpu, collision_weight=-0.75, goal_achieved_weight=1.0, off_road_weight=-0.5)
    ot = OracleEpisodeTracker(orc, collision_weight=-0.75, goal_achieved_weight=1.0, off_road_weight=-0.5)
    rng = np.random.default_rng(4)
    bits = lambda x: np.ascontiguousarray(x, np.float32).view(np.uint32)
    finished = np.zeros(2, np.int64)
    ranked_steps = 0
    for k in range(200):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        act[..., 0] = np.abs(act[..., 0])
        P.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        o_rew, o_term, o_trunc, o_mask, o_done = ot.step()
        g_rew, g_term, g_trunc, g_mask = [t.cpu().numpy() for t in gt.step()]
        try:
            assert np.array_equal(bits(g_rew), bits(o_rew)), "rewards"
            assert np.array_equal(g_term, o_term) and np.array_equal(g_trunc, o_trunc) and np.array_equal(g_mask, o_mask)
            assert np.array_equal(gt.done_worlds.cpu().numpy(), o_done), "done worlds"
            finished += o_done != 0
            P.compare_ints(gpu, orc, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
            P.compare_state(gpu, orc)
            if o_done.any() or k % 5 == 0:
                P.inject_and_compare(gpu, orc)
        except AssertionError as e:
            raise AssertionError("step %d: %s" % (k + 1, e))
        ranked_steps += _ranked(gpu)[0] > 0
    assert (finished >= 2).all(), finished
    assert ranked_steps > 150, "the rank replay should carry these worlds (%d of 200 steps)" % ranked_steps
    gpu.close()


# ---- the ends of the rank path's arrays ----
def _spiral_scene(tmp_path, name, n_roads, n_agents=64):
    """One polyline that winds inwards around a point, every segment closer to it than all the segments before: for an agent
    parked at that point EVERY road is an insert (src/knn.hpp:138-151), i.e. a candidate, so the candidate count is the road
    count.  The agent is the LAST of the scene's agents; all the others are parked far outside the map."""
    cx, cy = 200.0, -100.0
    pts = []
    for j in range(n_roads + 1):
        r = 48.0 - 44.0 * j / n_roads
        th = 0.11 * j
        pts.append({"x": cx + r * math.cos(th), "y": cy + r * math.sin(th), "z": 0.0})

    def obj(i, x, y):
        return {"position": [{"x": x, "y": y, "z": 0.0}] * 91, "width": 2.0, "length": 4.5, "height": 1.6, "heading": [0.3] * 91,
                "velocity": [{"x": 0.0, "y": 0.0}] * 91, "valid": [True] * 91, "goalPosition": {"x": x + 500.0, "y": y, "z": 0.0},
                "type": "vehicle", "id": i, "mark_as_expert": False}
    objects = [obj(i, 3000.0 + 20.0 * i, 2000.0) for i in range(n_agents - 1)] + [obj(n_agents - 1, cx, cy)]
    sc = {"name": name, "scenario_id": name, "objects": objects,
          # (the scene format caps a polyline at 1746 points: longer spirals continue in a second and third polyline)
          "roads": [{"geometry": pts[lo:lo + 1501], "type": "road_edge", "map_element_id": 15, "id": lo // 1500}
                    for lo in range(0, n_roads, 1500)],
          "tl_states": {}, "metadata": {"sdc_track_index": 0, "objects_of_interest": [], "tracks_to_predict": []}}
    p = tmp_path / (name + ".json")
    p.write_text(json.dumps(sc))
    return str(p)


def test_rank_path_at_the_ends_of_its_arrays(oracle_mod, tmp_path, monkeypatch):
    """The worst case at the end of every rank array: the LAST agent slot of the LAST world is the only ranked agent of the
    batch's last replay wave (its 63 neighbours are out of reach of every road), its candidate list is the longest the
    standard ranking takes (1272 = GD_RANK_CAP - 8: the replay's block prefetch runs into the slack behind the last row of
    rk_E), and all of its roads are inside the radius.  The world before it has eighteen roads too many for that (the
    long-list instantiation ranks it: 1290 candidates, entry 0 of the long rows), the one before that more than even the
    long list holds (4000 roads: overflow, fallback, then the bypass streak).  Rows against the oracle at every step, and the
    device-side bounds audit must stay at zero."""
    monkeypatch.setenv("GPUDRIVE_RANK_MIN_ROADS", "200")
    cap = _spiral_scene(tmp_path, "cap", 1272)
    over = _spiral_scene(tmp_path, "over", 1290)
    huge = _spiral_scene(tmp_path, "huge", 4000)
    plain = _scene(tmp_path, "plain", 5, 4, 80, 31)   # 316 roads, a handful of agents
    scenes = [plain, huge, over, cap]
    kw = dict(BENCH, observationRadius=60.0)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    assert gpu.stat(7) == 1
    P.compare_fresh(gpu, orc)
    zero = np.zeros((4, 64, 10), np.float32)
    seen_cap = seen_over = seen_huge = 0
    for k in range(8):
        P.write_actions(gpu, zero)   # parked: the candidate sets stay what they are
        np.copyto(orc.action_tensor(), zero)
        gpu.step()
        orc.step()
        P.compare_ints(gpu, orc, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        P.inject_and_compare(gpu, orc)
        path = gpu.debug_road_path()
        seen_cap += path[3, 63] == 1272
        seen_over += path[2, 63] == 1290            # ranked by the long-list instantiation
        seen_huge += path[1, 63] in (-11, -13)      # overflow, then the group bypasses the rank kernels
        # the parked agents are out of reach of every road (-3); those that share the overflowing agent's group of 32 are
        # selected with it by k_map_obs (-1), to the same empty rows
        assert (path[3, :63] == -3).all() and (path[2, :63] == -3).all() and np.isin(path[1, :63], (-3, -1)).all(), path[1:, :63]
    assert seen_cap >= 6, "the last agent slot should be ranked with GD_RANK_CAP - 8 candidates (%d of 8 steps)" % seen_cap
    assert seen_over >= 6, "1290 candidates go to the long list (%d of 8 steps)" % seen_over
    assert seen_huge >= 6, "the 4000-road spiral overflows every buffer (%d of 8 steps)" % seen_huge
    rows = RC.as_np(gpu.agent_roadmap_tensor())
    for w in (1, 2, 3):
        assert (rows[w, 63, :, 6] != 0).all(), "200 roads in reach of the spiral's centre"
    assert gpu.stat(21) == 0, "device-side bounds audit of the rank path: %d indices out of range" % gpu.stat(21)
    # the same worlds after a rebuild into the live buffers, the spirals now FIRST (agent slot 63 of worlds 0 and 1)
    new = [cap, over, plain, huge]
    gpu.set_maps(new)
    orc.set_maps(new)
    P.compare_fresh(gpu, orc)
    for k in range(3):
        P.write_actions(gpu, zero)
        np.copyto(orc.action_tensor(), zero)
        gpu.step()
        orc.step()
        P.inject_and_compare(gpu, orc)
    path = gpu.debug_road_path()
    assert path[0, 63] == 1272 and path[1, 63] == 1290
    assert gpu.stat(21) == 0
    gpu.close()


# ---- road rows: headings of roads exactly along the axes of the agent's frame ----
def _axis_scene(tmp_path, yaw):
    """One parked vehicle at (100, 50) with the given yaw and four straight polylines through points around it: along +x, -x,
    +y and -y of the WORLD frame, i.e. (for yaw a multiple of pi / 2) exactly aligned with, opposite to and perpendicular to
    the agent's heading."""
    def obj(i, x, y, yaw):
        return {"position": [{"x": x, "y": y, "z": 0.0}] * 91, "width": 2.0, "length": 4.5, "height": 1.6, "heading": [yaw] * 91,
                "velocity": [{"x": 0.0, "y": 0.0}] * 91, "valid": [True] * 91, "goalPosition": {"x": 400.0, "y": 400.0, "z": 0.0},
                "type": "vehicle", "id": i, "mark_as_expert": False}
    def line(rid, x0, y0, dx, dy):
        return {"geometry": [{"x": x0 + dx * k, "y": y0 + dy * k, "z": 0.0} for k in range(6)], "type": "road_edge",
                "map_element_id": 15, "id": rid}
    sc = {"name": "axes", "scenario_id": "axes", "objects": [obj(0, 100.0, 50.0, yaw)],
          "roads": [line(0, 104.0, 53.0, 2.0, 0.0), line(1, 96.0, 47.0, -2.0, 0.0), line(2, 103.0, 54.0, 0.0, 2.0),
                    line(3, 97.0, 46.0, 0.0, -2.0)],
          "tl_states": {}, "metadata": {"sdc_track_index": 0, "objects_of_interest": [], "tracks_to_predict": []}}
    p = tmp_path / ("axes_%d.json" % round(yaw * 100))
    p.write_text(json.dumps(sc))
    return str(p)


@pytest.mark.gpu
@pytest.mark.parametrize("knn_order", [0, 1], ids=["reference_order", "set_order"])
def test_headings_of_roads_along_the_axes_of_the_agents_frame(oracle_mod, tmp_path, knn_order):
    """`road_row` (map_obs.hip) uses the yaw-only forms of the reference's rotateVec and quaternion product, which can differ
    from the general forms in the SIGN OF A ZERO only -- and a zero decides between +pi and -pi for a road exactly opposite to
    the agent (atan2f(+-0, negative); reference tests/EgocentricRoadObservationTests.cpp pins +pi).  That case keeps the general
    product; here: agents at yaw 0, pi / 2, pi, -pi / 2 with roads exactly along both axes in both directions, every row
    against the oracle, headings within 1e-6 (so that +pi against -pi is a failure)."""
    scenes = [_axis_scene(tmp_path, y) for y in (0.0, math.pi / 2, math.pi, -math.pi / 2)]
    kw = dict(polylineReductionThreshold=0.0, observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0,
              dynamicsModel=0, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
    gpu = P.make_gpu_sim(scenes, max_agents=64, knn_order=knn_order, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    g = RC.as_np(gpu.agent_roadmap_tensor())[:, 0]
    o = np.asarray(orc.agent_roadmap_tensor())[:, 0]
    if knn_order == 1:
        g, o = P._sorted_rows(g), P._sorted_rows(o)
    live = o[:, :, 6] != 0
    assert live.sum() >= 4 * 16, "the scenes' roads did not reach the observation"
    assert np.abs(np.abs(o[live][:, 5]) - math.pi).min() < 1e-6, "no road exactly opposite to an agent: the scene does not exercise the case"
    assert np.allclose(g[..., 5], o[..., 5], atol=1e-6, rtol=0), "headings differ (a sign of pi?)"
    assert np.allclose(g, o, atol=P.OBS_ATOL, rtol=0)
    gpu.close()
