"""GPU suite, round 5: the linear road selection (`AllEntitiesWithRadiusFiltering`, the reference's `EnvConfig` default:
gpudrive/env/config.py:51 -> src/sim.cpp:258-279) in its own kernel (csrc/map_obs_linear.hip) against the oracle on the
scenes bench.py times and at BASELINE.json's full size, against the kernels that carried the mode in rounds 1-4, and the rule
that rows which cannot have changed are not rewritten (pose stamps) through resets, map changes and agent deletions.
Everything goes through `madrona_gpudrive` -> ctypes -> the C ABI."""
import os

import numpy as np
import pytest

from gpudrive_lab_amd import synth
from tests import parity as P
from tests import ref_cases as RC
from tests.conftest import SCENE_4, SCENE_407, TEST_JSON
from tests.test_gpu_round3 import _Worlds, _tiled

pytestmark = pytest.mark.gpu

ALL_OBJECTS = dict(isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
BENCH_LINEAR = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
                    roadObservationAlgorithm=1, polylineReductionThreshold=0.0, **ALL_OBJECTS)
WAYMO_LINEAR = dict(BENCH_LINEAR, polylineReductionThreshold=0.1)
# what an unchanged gpudrive/env/base_env.py:96-159 builds from the EnvConfig defaults + baselines/ppo/config/ppo_base_puffer.yaml
PPO_DEFAULT = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
                   roadObservationAlgorithm=1, polylineReductionThreshold=0.1, isStaticAgentControlled=0,
                   initOnlyValidAgentsAtFirstStep=1, IgnoreNonVehicles=1)


@pytest.fixture(scope="module")
def bench_scenes(tmp_path_factory):
    return synth.write_scenes(str(tmp_path_factory.mktemp("bench_scenes5")), list(range(8)))


def _bits(t):
    return RC.as_np(t).view(np.uint32)


def test_linear_mode_on_the_bench_scenes_in_lockstep_with_the_oracle(oracle_mod, bench_scenes):
    """Two of the worlds bench.py's `synthetic_linear` times (64 live agents, 4096 road edges): t = 0, then a third of an
    episode in lockstep -- rows elementwise (index order IS the mode's order), ints exact."""
    scenes = bench_scenes[:2]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **BENCH_LINEAR)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **BENCH_LINEAR)
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 30, 0, seed=11)
    rows = RC.as_np(gpu.agent_roadmap_tensor())
    assert (rows[..., 6] > 0).sum(-1).max() == 200, "some agent should fill all K rows on this scene (the early exit)"
    gpu.close()


@pytest.mark.parametrize("slots", [64, 128])
def test_linear_mode_full_size_first_and_last_worlds_equal_the_oracle(oracle_mod, slots):
    """1024 Waymo tiles (what `waymo_linear` / `ppo_default` run on): worlds 0-2 and the last three against the oracle,
    every replica bit-identical to its scene's first world."""
    import torch
    W = 1024
    kw = WAYMO_LINEAR if slots == 64 else PPO_DEFAULT
    gpu = P.make_gpu_sim(_tiled(W), max_agents=slots, **kw)
    orc = P.make_oracle_sim(oracle_mod, _tiled(3), max_agents=slots, **kw)
    rng = np.random.default_rng(4)
    last = [W - 3, W - 2, W - 1]
    views = [(_Worlds(gpu, [0, 1, 2]), [0, 1, 2]), (_Worlds(gpu, last), [w % 3 for w in last])]
    for step in range(9):
        act3 = P.random_actions(rng, 3, slots, 0)
        a = gpu.action_tensor().to_torch()
        a.copy_(torch.as_tensor(act3).repeat((W + 2) // 3, 1, 1)[:W].to(a.device))
        np.copyto(orc.action_tensor(), act3)
        gpu.step()
        orc.step()
        for view, scene in views:
            P.compare_ints(view, _Worlds(orc, scene), ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        if step % 4 == 0:
            gpu.debug_set_state(np.tile(orc.get_state(), ((W + 2) // 3, 1, 1))[:W])
            gpu.reset([])
            orc.reset([])
            for view, scene in views:
                sub = _Worlds(orc, scene)
                sub.W, sub.A = 3, slots
                P.compare_obs(view, sub)
    flat = gpu.agent_roadmap_tensor().to_torch().reshape(W, -1)
    for r in range(3):
        grp = flat[r::3]
        assert torch.equal(grp, grp[0:1].expand_as(grp)), "agent_roadmap_tensor: replicas of scene %d diverged" % r
    gpu.close()


@pytest.mark.parametrize("which", ["bench", "waymo_128", "ppo_default"])
def test_linear_kernel_equals_the_kernels_that_carried_the_mode_before(bench_scenes, monkeypatch, which):
    """Free running, same scenes and actions: the linear kernel (with and without the pose stamps) against
    GPUDRIVE_LINEAR_LEGACY=1 (the linear branch of k_map_obs + k_map_rows, rounds 1-4).  Bit-identical rows at every step,
    through a reset of some worlds and a teleport of a few agents."""
    if which == "bench":
        scenes, kw, slots = bench_scenes[:3], BENCH_LINEAR, 64
    elif which == "waymo_128":
        scenes, kw, slots = [TEST_JSON, SCENE_407, SCENE_4], WAYMO_LINEAR, 128
    else:
        scenes, kw, slots = [TEST_JSON, SCENE_407, SCENE_4, SCENE_407], PPO_DEFAULT, 128
    new = P.make_gpu_sim(scenes, max_agents=slots, **kw)
    monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
    noskip = P.make_gpu_sim(scenes, max_agents=slots, **kw)
    monkeypatch.delenv("GPUDRIVE_NO_POSE_SKIP")
    monkeypatch.setenv("GPUDRIVE_LINEAR_LEGACY", "1")
    old = P.make_gpu_sim(scenes, max_agents=slots, **kw)
    monkeypatch.delenv("GPUDRIVE_LINEAR_LEGACY")
    sims = [new, noskip, old]
    W = len(scenes)
    rng = np.random.default_rng(8)
    new.stat(30)
    for step in range(40):
        act = P.random_actions(rng, W, slots, 0)
        for s in sims:
            RC.write_actions(s, act)
            s.step()
        if step == 12:
            for s in sims:
                s.reset([1])
        if step == 20:  # a few agents 30 m away and turned: nothing about their previous rows holds
            st = new.debug_get_state()
            st[:, :3, 0] += 30.0
            st[:, :3, 3], st[:, :3, 6] = np.cos(0.4), np.sin(0.4)
            for s in sims:
                s.debug_set_state(st)
                s.reset([])
        ref = _bits(old.agent_roadmap_tensor())
        assert np.array_equal(_bits(new.agent_roadmap_tensor()), ref), "step %d: linear kernel differs from the legacy path" % step
        assert np.array_equal(_bits(noskip.agent_roadmap_tensor()), ref), "step %d: (no pose skip) differs from the legacy path" % step
    skipped = new.stat(30)
    assert noskip.stat(30) == 0
    if which == "ppo_default":
        assert skipped > 0, "parked cars never move: their rows must have been left in place"
    for s in sims:
        s.close()


def test_pose_stamps_die_with_the_world(oracle_mod, tmp_path):
    """`ppo_default`-style worlds (parked cars stay Static) in lockstep with the oracle for a whole episode and into the next,
    with a reset, a set_maps to other scenes (same poses may meet other roads), a set_maps back and a deleteAgents in the
    middle.  Before every rebuild the road rows are overwritten with garbage from outside: rows that survive are rows the
    engine skipped although the world changed."""
    import torch
    scenes = [TEST_JSON, SCENE_407, SCENE_4]
    gpu = P.make_gpu_sim(scenes, max_agents=128, **PPO_DEFAULT)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=128, **PPO_DEFAULT)
    P.compare_fresh(gpu, orc)
    resp = np.asarray(orc.response_type_tensor())[..., 0]
    live = P._live_mask(orc)
    assert ((resp == 2) & live).sum() > 0, "the scenes should hold parked (Static) agents"
    gpu.stat(30)
    P.lockstep(gpu, orc, 40, 0, seed=2)
    assert gpu.stat(30) > 0, "parked agents' rows should have been left in place"
    gpu.reset([0, 2])
    orc.reset([0, 2])
    P.compare_obs(gpu, orc, atol=P.FREE_OBS_ATOL)
    P.lockstep(gpu, orc, 10, 0, seed=3)
    def scribble(value):
        # (only the rows of live agents: padding agents' rows are written when their world is built, like in the reference,
        # src/level_gen.cpp:308-336, and deleteAgents rebuilds only the worlds it names)
        rows = gpu.agent_roadmap_tensor().to_torch()
        n = gpu.shape_tensor().to_torch()[:, 0]
        mask = torch.arange(rows.shape[1], device=rows.device)[None, :] < n[:, None]
        rows[mask] = value

    for new_scenes in ([SCENE_4, TEST_JSON, SCENE_407], scenes):
        scribble(123.0)
        gpu.set_maps(new_scenes)
        orc.set_maps(new_scenes)
        P.compare_fresh(gpu, orc)
        P.lockstep(gpu, orc, 6, 0, seed=4)
    scribble(-7.0)
    ids = np.asarray(orc.agent_id_tensor())
    victims = {1: [int(ids[1, 0]), int(ids[1, 2])], 2: [int(ids[2, 1])]}
    gpu.deleteAgents(victims)
    orc.deleteAgents(victims)
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 60, 0, seed=5)   # through the episode's end: finished agents are parked at the padding position
    torch.cuda.synchronize()
    gpu.close()


DIRECT_PACK = [
    ("set_order_64", [TEST_JSON, SCENE_407, SCENE_4], 64, 1, dict(WAYMO_LINEAR, roadObservationAlgorithm=0)),
    ("set_order_128", [SCENE_4, SCENE_407], 128, 1, dict(WAYMO_LINEAR, roadObservationAlgorithm=0, collisionBehaviour=0)),
    ("linear_ppo_default", [TEST_JSON, SCENE_407, SCENE_4], 128, 0, PPO_DEFAULT),
    ("linear_64_set_order_flag", [SCENE_407, SCENE_4], 64, 1, WAYMO_LINEAR),
    # k_map_rows behind the reference-order selections (rank replay forced on for these small worlds; history replay) and behind
    # the unfused set-order selection
    ("reference_order_rank_replay", [TEST_JSON, SCENE_407, SCENE_4], 64, 0, dict(WAYMO_LINEAR, roadObservationAlgorithm=0)),
    ("reference_order_history_replay_128", [SCENE_4, SCENE_407], 128, 0, dict(WAYMO_LINEAR, roadObservationAlgorithm=0)),
    ("set_order_row_kernel", [TEST_JSON, SCENE_4], 64, 1, dict(WAYMO_LINEAR, roadObservationAlgorithm=0)),
]


@pytest.mark.parametrize("name,scenes,slots,knn_order,kw", DIRECT_PACK, ids=[c[0] for c in DIRECT_PACK])
def test_packed_observation_written_by_the_step_equals_the_second_pass(monkeypatch, name, scenes, slots, knn_order, kw):
    """gd_attach_packed: the packed observation written where the rows are produced (k_world_step: ego + partner columns;
    the road kernel: 200 x 13 road columns) must be bit-identical to k_pack_obs's second pass over the raw tensors -- which
    tests/golden/obs_pack_golden.npz pins to the reference's own gpudrive/datatypes code -- on three simulators fed the
    same actions: raw tensors + second pass, direct with the raw rows kept, direct ONLY.  Through steps, a partial reset, a
    set_maps and a deleteAgents; padding agents' rows included."""
    if name == "reference_order_rank_replay":
        monkeypatch.setenv("GPUDRIVE_RANK_MIN_ROADS", "200")
    if name == "reference_order_history_replay_128":
        monkeypatch.setenv("GPUDRIVE_NO_RANK_REPLAY", "1")
    if name == "set_order_row_kernel":
        monkeypatch.setenv("GPUDRIVE_SET_FUSED_ROWS", "0")
    ref = P.make_gpu_sim(scenes, max_agents=slots, knn_order=knn_order, **kw)
    both = P.make_gpu_sim(scenes, max_agents=slots, knn_order=knn_order, **kw)
    only = P.make_gpu_sim(scenes, max_agents=slots, knn_order=knn_order, **kw)
    assert both.direct_pack(only=False) and only.direct_pack(only=True)
    sims = [ref, both, only]
    W = len(scenes)
    rng = np.random.default_rng(3)

    def check(tag):
        want = _bits(ref.packed_observations())
        for s, nm in ((both, "raw rows kept"), (only, "packed only")):
            got = _bits(s.packed_observations())
            assert got.shape == want.shape
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                raise AssertionError("%s (%s): %d packed elements differ, first at %s" % (tag, nm, len(bad), bad[0]))
        for t in ("partner_observations_tensor", "agent_roadmap_tensor", "self_observation_tensor"):
            assert np.array_equal(_bits(getattr(both, t)()), _bits(getattr(ref, t)())), "%s: %s differs with the raw rows kept" % (tag, t)

    check("t = 0")
    for step in range(24):
        act = P.random_actions(rng, W, slots, 0)
        for s in sims:
            RC.write_actions(s, act)
            s.step()
        if step == 7:
            for s in sims:
                s.reset([W - 1])
        if step == 12:
            new = scenes[1:] + scenes[:1]
            for s in sims:
                s.set_maps(new)
        if step == 17:
            ids = RC.as_np(ref.absolute_self_observation_tensor())[0, :2, 13].astype(int).tolist()
            for s in sims:
                s.deleteAgents({0: ids})
        check("step %d" % (step + 1))
    # switching it off again: the raw rows are brought up to date at once
    only.direct_pack_off()
    for t in ("partner_observations_tensor", "agent_roadmap_tensor"):
        assert np.array_equal(_bits(getattr(only, t)()), _bits(getattr(ref, t)())), t
    for s in sims:
        s.close()


def test_direct_pack_is_refused_where_it_is_not_available(monkeypatch):
    """disableClassicalObs (no rows at all) and the developer switch GPUDRIVE_LINEAR_LEGACY=1: gd_attach_packed answers
    GD_ERR_UNSUPPORTED, nothing changes, and the second-pass packed_observations() keeps working."""
    monkeypatch.setenv("GPUDRIVE_LINEAR_LEGACY", "1")
    gpu = P.make_gpu_sim([SCENE_407], max_agents=64, **WAYMO_LINEAR)
    before = _bits(gpu.packed_observations()).copy()
    assert gpu.direct_pack(only=True) is False
    assert np.array_equal(_bits(gpu.packed_observations()), before)
    gpu.close()


@pytest.mark.parametrize("mode", ["reference_order_rank_path", "reference_order_history_replay", "set_order_fused", "set_order_row_kernel"])
def test_rows_left_in_place_are_the_rows_a_rewrite_would_produce(monkeypatch, mode):
    """Pose stamps in the k-NN modes (k_map_rows for the reference's row order and the unfused set-order write-out, the fused
    set-order kernel): two simulators on Waymo scenes under the default init rules (parked cars stay Static, finished agents
    are parked at the padding position), one with GPUDRIVE_NO_POSE_SKIP=1, same actions, free running through a partial
    reset and a teleport of a few agents: bit-identical road rows at every step, and rows must actually have been skipped."""
    kw = dict(PPO_DEFAULT, roadObservationAlgorithm=0, collisionBehaviour=0)
    knn_order = 1 if mode.startswith("set") else 0
    if mode == "reference_order_rank_path":
        monkeypatch.setenv("GPUDRIVE_RANK_MIN_ROADS", "200")
    if mode == "reference_order_history_replay":
        monkeypatch.setenv("GPUDRIVE_NO_RANK_REPLAY", "1")
    if mode == "set_order_row_kernel":
        monkeypatch.setenv("GPUDRIVE_SET_FUSED_ROWS", "0")
    scenes = [TEST_JSON, SCENE_407, SCENE_4, SCENE_407]
    skip = P.make_gpu_sim(scenes, max_agents=128, knn_order=knn_order, **kw)
    monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
    plain = P.make_gpu_sim(scenes, max_agents=128, knn_order=knn_order, **kw)
    monkeypatch.delenv("GPUDRIVE_NO_POSE_SKIP")
    rng = np.random.default_rng(6)
    skip.stat(30)
    assert np.array_equal(_bits(skip.agent_roadmap_tensor()), _bits(plain.agent_roadmap_tensor()))
    for step in range(45):
        act = P.random_actions(rng, len(scenes), 128, 0)
        for s in (skip, plain):
            RC.write_actions(s, act)
            s.step()
        if step == 15:
            for s in (skip, plain):
                s.reset([0, 3])
        if step == 25:
            st = skip.debug_get_state()
            st[:, 1:4, 0] -= 25.0
            st[:, 1:4, 3], st[:, 1:4, 6] = np.cos(-0.7), np.sin(-0.7)
            for s in (skip, plain):
                s.debug_set_state(st)
                s.reset([])
        a, b = _bits(skip.agent_roadmap_tensor()), _bits(plain.agent_roadmap_tensor())
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError("step %d: %d elements differ, first at %s" % (step + 1, len(bad), bad[0]))
    assert skip.stat(30) > 0 and plain.stat(30) == 0
    skip.close()
    plain.close()


@pytest.mark.parametrize("rows", ["fused", "row_kernel"])
def test_set_order_is_one_order_on_both_selection_paths(monkeypatch, rows):
    """Set order promises ONE row order (the world's cell-sorted road list).  An agent's first selection after a reset has only
    the radius for a bound and, with thousands of roads in reach, takes the full-stream path; the next one is bounded by the
    first and takes the grid path.  A parked car sees the same roads in the same places on both steps: its rows must be
    bit-identical, order included (the pose stamps are switched off so that both steps really write them)."""
    monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
    if rows == "row_kernel":
        monkeypatch.setenv("GPUDRIVE_SET_FUSED_ROWS", "0")
    kw = dict(PPO_DEFAULT, roadObservationAlgorithm=0, polylineReductionThreshold=0.0, observationRadius=100.0)
    gpu = P.make_gpu_sim([TEST_JSON, SCENE_407], max_agents=64, knn_order=1, **kw)
    resp = RC.as_np(gpu.response_type_tensor())[..., 0]
    n = RC.as_np(gpu.shape_tensor())[:, 0]
    parked = (resp == 2) & (np.arange(64)[None, :] < n[:, None])
    first = RC.as_np(gpu.agent_roadmap_tensor()).copy()
    full = (first[..., 6] > 0).sum(-1) == 200
    assert (parked & full).sum() >= 3, "the scenes should hold parked cars with K roads in reach"
    for _ in range(3):
        gpu.step()
        now = RC.as_np(gpu.agent_roadmap_tensor())
        assert np.array_equal(now[parked].view(np.uint32), first[parked].view(np.uint32)), "a parked car's rows changed their order"
    gpu.reset([0, 1])   # the full-stream path again
    assert np.array_equal(RC.as_np(gpu.agent_roadmap_tensor())[parked].view(np.uint32), first[parked].view(np.uint32))
    gpu.close()


def test_rank_buffers_that_cannot_be_allocated_leave_the_batch_on_the_history_replay(oracle_mod, bench_scenes, monkeypatch):
    """engine.cpp ensure_rank_buffers: when the device cannot give the rank replay its scratch (simulated: the third allocation
    throws), what was allocated is returned, nothing points at it any more, the batch is selected by k_map_obs -- the same
    rows -- and the rank replay is not tried again for this simulator, through a rebuild too."""
    monkeypatch.setenv("GPUDRIVE_RANK_ALLOC_FAIL", "1")
    scenes = bench_scenes[:2]
    kw = dict(BENCH_LINEAR, roadObservationAlgorithm=0)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    assert gpu.stat(7) == 0, "the rank replay must be off after the failed allocation"
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 3, 0, seed=1)
    gpu.set_maps(scenes[::-1])
    orc.set_maps(scenes[::-1])
    assert gpu.stat(7) == 0 and gpu.stat(21) == 0
    P.compare_fresh(gpu, orc)
    gpu.close()


def test_linear_selection_by_brute_force_on_the_hip_path():
    """The oracle-independent pin of linear mode (tests/test_columns.py): float64 brute force over the global road rows -- the
    first K roads in index order within the radius, in that order, each row the road's ego-frame image -- on the HIP path,
    unreduced polylines (K binds: the scan's early exit) and the reduced scenes."""
    from tests.test_columns import check_linear_selection_by_brute_force
    for thr, radius in ((0.0, 60.0), (0.1, 50.0)):
        kw = dict(BENCH_LINEAR, polylineReductionThreshold=thr, observationRadius=radius)
        gpu = P.make_gpu_sim([TEST_JSON, SCENE_4], max_agents=64, **kw)
        rng = np.random.default_rng(4)
        for _ in range(3):
            RC.write_actions(gpu, P.random_actions(rng, 2, 64, 0))
            gpu.step()
        checked, full = check_linear_selection_by_brute_force(gpu, radius, as_numpy=RC.as_np)
        assert checked == 25 + 64 and (full > 10 or thr > 0)
        gpu.close()


def test_partner_rows_beyond_the_worlds_agents_are_written_by_reset_passes_only(oracle_mod):
    """A step pass leaves the rows of partner slots beyond the world's agents (zero_nonexist(), id -2: a function of the agent
    count alone) where the last reset pass put them and writes only the n (n - 1) real rows of a ragged world; a reset pass
    writes every row.  Free-running steps (no teacher forcing in between, which would be a reset pass) must still show the
    oracle's partner tensor; garbage written from outside survives a step exactly in those rows and nowhere else, and is
    gone after any reset pass."""
    import torch
    scenes = [TEST_JSON, SCENE_407, SCENE_4]
    gpu = P.make_gpu_sim(scenes, max_agents=128, **PPO_DEFAULT)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=128, **PPO_DEFAULT)
    rng = np.random.default_rng(12)
    for _ in range(6):
        act = P.random_actions(rng, 3, 128, 0)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
    P.compare_obs(gpu, orc, atol=P.FREE_OBS_ATOL, names=["partner_observations_tensor"])
    part = gpu.partner_observations_tensor().to_torch()
    n = gpu.shape_tensor().to_torch()[:, 0]
    live = torch.arange(part.shape[1], device=part.device)[None, :] < n[:, None]
    part[live] = 55.0
    act = P.random_actions(rng, 3, 128, 0)
    RC.write_actions(gpu, act)
    np.copyto(orc.action_tensor(), act)
    gpu.step()
    orc.step()
    g = RC.as_np(gpu.partner_observations_tensor())
    o = np.asarray(orc.partner_observations_tensor())
    lv = RC.as_np(live)
    survived = (g == 55.0).all(-1) & lv[:, :, None]
    assert survived.any(), "a step pass should have left the constant rows alone"
    assert (o[survived][:, 8] == -2.0).all(), "only rows of partner slots beyond the world's agents may be left in place"
    real = lv[:, :, None] & ~survived
    assert np.allclose(g[real], o[real], atol=P.FREE_OBS_ATOL)
    assert (o[real][:, 8] != -2.0).all(), "every real partner row must have been written"
    gpu.reset([])
    orc.reset([])
    P.compare_obs(gpu, orc, atol=P.FREE_OBS_ATOL, names=["partner_observations_tensor"])
    gpu.close()


def test_a_static_agent_moved_from_outside_is_visited_by_the_next_step(monkeypatch):
    """Step passes of the linear scan leave out the agents that never move (`Static`).  gd_debug_set_state can move one all the
    same: the road pass of the NEXT step -- a step, not a reset pass -- must then visit every live agent (engine.cpp
    full_pass_next), and the step after that is back on the captured graph."""
    import torch
    scenes = [TEST_JSON, SCENE_407]
    with torch.cuda.stream(torch.cuda.Stream()):
        skip = P.make_gpu_sim(scenes, max_agents=128, **PPO_DEFAULT)
        monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
        plain = P.make_gpu_sim(scenes, max_agents=128, **PPO_DEFAULT)
        monkeypatch.delenv("GPUDRIVE_NO_POSE_SKIP")
        resp = RC.as_np(skip.response_type_tensor())[..., 0]
        n = RC.as_np(skip.shape_tensor())[:, 0]
        w, a = [(w, a) for w in range(2) for a in range(n[w]) if resp[w, a] == 2][0]
        rng = np.random.default_rng(1)
        for step in range(8):
            if step == 3:
                st = skip.debug_get_state()
                st[w, a, 0] += 12.0
                st[w, a, 1] -= 7.0
                for s in (skip, plain):
                    s.debug_set_state(st)
            act = P.random_actions(rng, 2, 128, 0)
            for s in (skip, plain):
                RC.write_actions(s, act)
                s.step()
            assert np.array_equal(_bits(skip.agent_roadmap_tensor()), _bits(plain.agent_roadmap_tensor())), "step %d" % (step + 1)
        assert skip.stat(0) >= 5, "the steps around the full pass should have been graph replays"
        skip.close()
        plain.close()


def test_bev_rasters_left_in_place_are_the_rasters_a_repaint_would_produce(monkeypatch, oracle_mod):
    """k_world_step marks the agents whose BEV raster can have changed (own pose bits changed, or an agent that moved is or was
    within reach) and k_bev repaints only those.  Two simulators on Waymo scenes under the default init rules (parked cars
    Static), one with GPUDRIVE_NO_POSE_SKIP=1 (every live agent repainted every step), same actions through a partial reset, a
    teleport from outside the engine and a set_maps: the same bytes at every step, rasters must really have been left alone,
    and the skipping simulator still matches the oracle's rasters at the end."""
    kw = dict(PPO_DEFAULT, roadObservationAlgorithm=1)
    scenes = [TEST_JSON, SCENE_407, SCENE_4, SCENE_407]
    skip = P.make_gpu_sim(scenes, max_agents=128, enable_bev=True, **kw)
    monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
    plain = P.make_gpu_sim(scenes, max_agents=128, enable_bev=True, **kw)
    monkeypatch.delenv("GPUDRIVE_NO_POSE_SKIP")
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=128, enableBev=1, **kw)
    rng = np.random.default_rng(16)
    live = RC.as_np(skip.shape_tensor())[:, 0]
    painted_total = 0

    def same(step):
        a, b = RC.as_np(skip.bev_observation_tensor()), RC.as_np(plain.bev_observation_tensor())
        for w in range(len(scenes)):
            n = int(RC.as_np(skip.shape_tensor())[w, 0])
            if not np.array_equal(a[w, :n].view(np.uint32), b[w, :n].view(np.uint32)):
                bad = np.argwhere(a[w, :n] != b[w, :n])
                raise AssertionError("step %d world %d: %d cells differ, first at %s" % (step, w, len(bad), bad[0]))

    same(0)
    for step in range(40):
        act = P.random_actions(rng, len(scenes), 128, 0)
        act[:, 5:] = 0.0  # most agents stand still: their neighbourhoods are the ones that may be left alone
        for s in (skip, plain):
            RC.write_actions(s, act)
            s.step()
        if step < 20:
            np.copyto(orc.action_tensor(), act)
            orc.step()
        painted_total += skip.stat(31)
        if step == 12:
            for s in (skip, plain, orc):
                s.reset([0, 3])
        if step == 19:
            assert P.compare_bev(skip, orc) > 0.001   # (the oracle has no debug_set_state: it leaves here)
        if step == 20:
            st = skip.debug_get_state()
            st[:, 1:3, 0] += 8.0
            for s in (skip, plain):
                s.debug_set_state(st)
                s.reset([])
        if step == 30:
            for s in (skip, plain):
                s.set_maps(scenes[::-1])
        same(step + 1)
    assert 0 < painted_total < 40 * int(live.sum()), "rasters were repainted %d times for %d live agents" % (painted_total, live.sum())
    skip.close()
    plain.close()


@pytest.mark.parametrize("half_angle", [0.0, 3.14159265], ids=["120deg", "360deg"])
def test_lidar_returns_left_in_place_are_the_returns_a_retrace_would_produce(monkeypatch, half_angle):
    """k_world_step marks the agents whose LiDAR returns can have changed (own pose or the head angle of the action row changed, or
    an agent that moved is or was within the rays' reach) and k_lidar traces only those.  Twin simulators on Waymo scenes under the
    default init rules, one with GPUDRIVE_NO_POSE_SKIP=1 (every live agent traced on every step), same actions -- a few agents
    drive, some only turn their heads, most stand still -- through a partial reset, a teleport from outside the engine and a
    set_maps: the same bytes at every step."""
    kw = dict(PPO_DEFAULT, roadObservationAlgorithm=1, enableLidar=1)
    scenes = [TEST_JSON, SCENE_407, SCENE_4, SCENE_407]
    skip = P.make_gpu_sim(scenes, max_agents=128, lidar_half_angle=half_angle, **kw)
    monkeypatch.setenv("GPUDRIVE_NO_POSE_SKIP", "1")
    plain = P.make_gpu_sim(scenes, max_agents=128, lidar_half_angle=half_angle, **kw)
    monkeypatch.delenv("GPUDRIVE_NO_POSE_SKIP")
    rng = np.random.default_rng(26)
    traced = 0

    def same(step):
        a, b = RC.as_np(skip.lidar_tensor()), RC.as_np(plain.lidar_tensor())
        for w in range(len(scenes)):
            n = int(RC.as_np(skip.shape_tensor())[w, 0])
            if not np.array_equal(a[w, :n].view(np.uint32), b[w, :n].view(np.uint32)):
                bad = np.argwhere(a[w, :n].view(np.uint32) != b[w, :n].view(np.uint32))
                raise AssertionError("step %d world %d: %d values differ, first at %s" % (step, w, len(bad), bad[0]))

    same(0)
    for step in range(40):
        act = P.random_actions(rng, len(scenes), 128, 0)
        act[:, 3:] = 0.0              # three agents per world drive ...
        if step % 3 == 0:
            act[:, 6:9, 2] = rng.uniform(-0.5, 0.5, (len(scenes), 3)).astype(np.float32)   # ... three others turn their heads now and then
        for s in (skip, plain):
            RC.write_actions(s, act)
            s.step()
        traced += skip.stat(44)
        if step == 12:
            for s in (skip, plain):
                s.reset([0, 3])
        if step == 20:
            st = skip.debug_get_state()
            st[:, 1:3, 0] += 8.0
            for s in (skip, plain):
                s.debug_set_state(st)
                s.reset([])
        if step == 30:
            for s in (skip, plain):
                s.set_maps(scenes[::-1])
        same(step + 1)
    live = int(RC.as_np(skip.shape_tensor())[:, 0].sum())
    assert 0 < traced < 40 * live, "returns were traced %d times for %d live agents over 40 steps" % (traced, live)
    skip.close()
    plain.close()


def test_gym_wrapper_episode_calls_on_the_device(oracle_mod):
    """The calls GPUDriveTorchEnv makes around an episode, through gpudrive_lab_amd/harness.py on the real module: the expert
    actions sliced from the exported trajectory equal the kernel's (`sim.expert_actions()`), a reset with warm-up steps leaves
    the simulator where the device-side log playback leaves a twin, infos / controlled mask / file names read what the oracle
    holds, `remove_agents_by_id` and `swap_data_batch` go through deleteAgents / set_maps and re-read the mask."""
    import torch
    from gpudrive_lab_amd.harness import TorchCallSequence
    scenes = [SCENE_4, SCENE_407, TEST_JSON]
    kw = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1, distanceToGoalThreshold=2.0,
              dynamicsModel=0, **ALL_OBJECTS)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    twin = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    env = TorchCallSequence(gpu, dynamics_model="classic", init_steps=4)
    for a, b in zip(env.get_expert_actions(), gpu.expert_actions()):
        assert a.shape == b.shape and torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                                  b.view(torch.int32) if b.dtype == torch.float32 else b)
    obs = env.reset()
    twin.reset(list(range(3)))
    twin.advance_log_playback(4)
    for name in ("self_observation_tensor", "partner_observations_tensor", "agent_roadmap_tensor", "done_tensor", "info_tensor",
                 "steps_remaining_tensor"):
        x, y = RC.as_np(getattr(gpu, name)()), RC.as_np(getattr(twin, name)())
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y), name
    assert torch.allclose(obs, twin.packed_observations(), atol=3e-7, rtol=1e-6)
    ctrl = np.asarray(orc.controlled_state_tensor())[..., 0] == 1
    assert np.array_equal(env.get_controlled_agents_mask().cpu().numpy(), ctrl)
    assert env.num_valid_controlled_agents_across_worlds == int(ctrl.sum())
    names = env.get_env_filenames()   # (the scene's own "name" field, src/MapReader.cpp -> map_name_tensor)
    exp_names = ["".join(chr(c) for c in row if c != 0) for row in np.asarray(orc.map_name_tensor()).tolist()]
    assert [names[w] for w in range(3)] == exp_names and all(n.endswith(".json") for n in exp_names)
    info = env.get_infos()
    assert info.shape == (3, 64) and np.array_equal(info.goal_achieved.cpu().numpy(), RC.as_np(gpu.info_tensor())[..., 3])
    before = int(RC.as_np(gpu.shape_tensor())[:, 0].sum())
    env.remove_agents_by_id(0.25, remove_controlled_agents=True, generator=torch.Generator().manual_seed(4))
    after = int(RC.as_np(gpu.shape_tensor())[:, 0].sum())
    assert after < before and env.num_valid_controlled_agents_across_worlds < int(ctrl.sum())
    env.swap_data_batch(scenes[::-1])
    orc.set_maps(scenes[::-1])
    assert np.array_equal(env.get_controlled_agents_mask().cpu().numpy(), np.asarray(orc.controlled_state_tensor())[..., 0] == 1)
    P.compare_fresh(gpu, orc)
    gpu.close()
    twin.close()
