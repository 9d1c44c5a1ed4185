"""GPU suite, round 2: parity on the scenes bench.py times, the row order pinned by libstdc++'s heap, the
oracle-independent partner check on the HIP path, the parameter modes no test covered, and the boundary pieces an
unchanged caller depends on (lazy BEV tensor, blocking step, stream hand-over, environment fall-backs, the torch-only
call sequence of the gym wrapper).  Everything goes through `madrona_gpudrive` -> ctypes -> the C ABI."""
import os

import numpy as np
import pytest

from gpudrive_lab_amd import synth
from tests import heap_pin as HP
from tests import parity as P
from tests import ref_cases as RC
from tests.conftest import SCENE_4, SCENE_407, TEST_JSON
from tests.test_columns import check_partner_rows_by_brute_force

pytestmark = pytest.mark.gpu

ALL_OBJECTS = dict(isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
# bench.py params_for("synthetic")
BENCH = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
             roadObservationAlgorithm=0, polylineReductionThreshold=0.0, **ALL_OBJECTS)
CLASSIC = dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
               distanceToGoalThreshold=2.0, dynamicsModel=0, **ALL_OBJECTS)


@pytest.fixture(scope="module")
def bench_scenes(tmp_path_factory):
    """The worlds bench.py times on rank 0: synth.write_scenes seeds 0 and 1 (64 live agents, 4096 road edges)."""
    return synth.write_scenes(str(tmp_path_factory.mktemp("bench_scenes")), [0, 1])


def test_lockstep_parity_on_the_bench_scenes(oracle_mod, bench_scenes):
    """What bench.py times is compared with the oracle on what it is timed on: reference row order, elementwise."""
    gpu = P.make_gpu_sim(bench_scenes, max_agents=64, **BENCH)
    orc = P.make_oracle_sim(oracle_mod, bench_scenes, max_agents=64, **BENCH)
    assert np.asarray(orc.shape_tensor()).tolist() == [[64, 4096], [64, 4096]]
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 12, 0, seed=1234)
    gpu.reset([0, 1])
    orc.reset([0, 1])
    P.compare_fresh(gpu, orc)
    gpu.close()


@pytest.mark.parametrize("rows", ["rows_kernel", "rows_fused"])
def test_set_order_on_the_bench_scenes(oracle_mod, bench_scenes, monkeypatch, rows):
    monkeypatch.setenv("GPUDRIVE_SET_FUSED_ROWS", "1" if rows == "rows_fused" else "0")
    gpu = P.make_gpu_sim(bench_scenes, max_agents=64, knn_order=1, **BENCH)
    orc = P.make_oracle_sim(oracle_mod, bench_scenes, max_agents=64, **BENCH)
    rng = np.random.default_rng(5)
    names = [n for n in P.OBS_TENSORS if n != "agent_roadmap_tensor"]
    for _ in range(4):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
        P.compare_obs(gpu, orc, names=names)
        P.compare_roadmap_as_set(gpu, orc)
    gpu.close()


@pytest.mark.parametrize("which", ["waymo", "bench"])
def test_row_order_equals_libstdcxx_heap_on_the_hip_path(oracle_mod, bench_scenes, which):
    """knn.hpp's loop on libstdc++'s make_heap / pop_heap / push_heap (tests/heap_pin.cpp), fed the oracle's key
    sequence, decides the row order; the HIP path must produce those rows (an order mismatch shows up as rows that
    differ by metres, far beyond the 1e-5 bound)."""
    scenes, kw = ([TEST_JSON, SCENE_407, SCENE_4], CLASSIC) if which == "waymo" else (bench_scenes[:1], BENCH)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    rng = np.random.default_rng(2)
    for _ in range(3):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        RC.write_actions(gpu, act)
        np.copyto(orc.action_tensor(), act)
        gpu.step()
        orc.step()
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    got = RC.as_np(gpu.agent_roadmap_tensor())
    shape = np.asarray(orc.shape_tensor())
    checked = 0
    for w in range(orc.W):
        n = int(shape[w, 0])
        for a in sorted(set(np.linspace(0, n - 1, 10).astype(int))):
            rows, _, _ = HP.expected_rows(orc, w, a, kw["observationRadius"])
            assert np.allclose(got[w, a], rows, atol=P.OBS_ATOL, rtol=0), (w, a, np.abs(got[w, a] - rows).max())
            checked += 1
    assert checked >= 10
    gpu.close()


def test_partner_rows_by_brute_force_on_the_hip_path():
    """collectPartnerObsSystem recomputed in numpy from the absolute rows (tests/test_columns.py), no oracle involved."""
    kw = dict(CLASSIC, observationRadius=40.0)
    gpu = P.make_gpu_sim([TEST_JSON, SCENE_4], max_agents=64, **kw)
    rng = np.random.default_rng(9)
    for _ in range(5):
        RC.write_actions(gpu, P.random_actions(rng, 2, 64, 0))
        gpu.step()
    n_in, n_out = check_partner_rows_by_brute_force(gpu, 40.0, as_numpy=RC.as_np)
    assert n_in > 500 and n_out > 500
    gpu.close()


MODES = [
    # src/sim.cpp:194,247,468: the partner, road and BEV systems return at once; self / absolute observations and
    # everything else run as usual
    ("disable_classical_obs", [SCENE_4, TEST_JSON], dict(CLASSIC, disableClassicalObs=1, collisionBehaviour=0)),
    # src/level_gen.cpp:102-129, 357-370 with readFromTracksToPredict: no object is filtered out by type or validity,
    # none is Static, and an object is controlled iff its isTrackToPredict flag is not -1 (the flag of a real object is
    # 0 or 1, so every created agent is controlled, up to maxNumControlledAgents).  The reference reads the MetaData
    # of the new entity before assigning it; pinned here is the evident intent (the object's own
    # metadata), which is what oracle and engine implement.
    ("read_from_tracks_to_predict", [SCENE_4, SCENE_407, TEST_JSON],
     dict(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=0, rewardType=1,
          distanceToGoalThreshold=2.0, dynamicsModel=0, readFromTracksToPredict=1, IgnoreNonVehicles=0)),
]


@pytest.mark.parametrize("name,scenes,kw", MODES, ids=[m[0] for m in MODES])
def test_lockstep_parity_parameter_modes(oracle_mod, name, scenes, kw):
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    if name == "read_from_tracks_to_predict":
        ctrl = np.asarray(orc.controlled_state_tensor())[..., 0]
        meta = np.asarray(orc.metadata_tensor())
        resp = np.asarray(orc.response_type_tensor())[..., 0]
        live = np.arange(64)[None, :] < np.asarray(orc.shape_tensor())[:, :1]
        assert (ctrl[live] == 1).all() and (meta[..., 2][live] != -1).all() and (resp[live] == 0).all()
        assert (ctrl[~live] == 0).all()
    P.compare_fresh(gpu, orc)
    P.lockstep(gpu, orc, 20, 0, seed=77)
    if name == "disable_classical_obs":
        # the observation tensors are never written: still the zeros they were allocated with
        for t in ("partner_observations_tensor", "agent_roadmap_tensor"):
            live = np.asarray(orc.shape_tensor())[:, 0]
            g = RC.as_np(getattr(gpu, t)())
            assert all((g[w, :live[w]] == 0).all() for w in range(len(live))), t
    gpu.close()


# ---- boundary ----
def test_bev_tensor_is_created_by_the_first_getter_call(oracle_mod):
    """An unchanged caller never passes enable_bev (gpudrive/env/base_env.py:176-190) and just calls
    sim.bev_observation_tensor() (env_torch.py:936): the call creates the tensor, holds the rasters of the current state,
    and every later step refreshes them."""
    scenes = [SCENE_4, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **CLASSIC)          # no enable_bev
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, enableBev=1, **CLASSIC)
    rng = np.random.default_rng(12)

    def advance(k):
        for _ in range(k):
            act = P.random_actions(rng, orc.W, orc.A, 0)
            RC.write_actions(gpu, act)
            np.copyto(orc.action_tensor(), act)
            gpu.step()
            orc.step()
        gpu.debug_set_state(orc.get_state())
        gpu.reset([])
        orc.reset([])
    advance(3)
    t = gpu.bev_observation_tensor().to_torch()                      # first call: allocates, rasterises the current state
    assert tuple(t.shape) == (2, 64, 200, 200, 1)
    assert P.compare_bev(gpu, orc) > 0.001
    advance(2)                                                       # later steps refresh the same storage
    assert gpu.bev_observation_tensor().to_torch().data_ptr() == t.data_ptr()
    assert P.compare_bev(gpu, orc) > 0.001
    gpu.close()


def test_blocking_step_is_visible_from_another_stream(oracle_mod):
    """SimManager(sync=True) / GPUDRIVE_SYNC_STEP=1: step() returns after the graph has finished like the reference's
    (src/mgr.cpp:154-156), so a reader on ANOTHER stream that was never ordered against the launch stream sees it."""
    import torch
    scenes = [SCENE_4, TEST_JSON]
    launch, reader = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(launch):
        gpu = P.make_gpu_sim(scenes, max_agents=64, sync=True, **CLASSIC)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **CLASSIC)
    rng = np.random.default_rng(4)
    for k in range(6):
        act = P.random_actions(rng, orc.W, orc.A, 0)
        with torch.cuda.stream(launch):
            RC.write_actions(gpu, act)
            gpu.step()
        with torch.cuda.stream(reader):                               # no wait_stream / event between the two
            steps = gpu.steps_remaining_tensor().to_torch().clone()
            done = gpu.done_tensor().to_torch().clone()
        reader.synchronize()
        np.copyto(orc.action_tensor(), act)
        orc.step()
        assert np.array_equal(steps.cpu().numpy(), np.asarray(orc.steps_remaining_tensor()))
        assert np.array_equal(done.cpu().numpy(), np.asarray(orc.done_tensor()))
    gpu.close()
    os.environ["GPUDRIVE_SYNC_STEP"] = "1"
    try:
        assert P.make_gpu_sim([TEST_JSON], max_agents=64, **CLASSIC)._sync is True
    finally:
        del os.environ["GPUDRIVE_SYNC_STEP"]


def test_stream_hand_over_and_graph_replay_are_real(oracle_mod):
    """Built on the default stream, first stepped on a side stream: the engine must move to that stream (so that torch's
    action writes and tensor reads there are ordered against the kernels), capture the step once and replay it.  Reads
    happen on the side stream itself, without the synchronising debug calls."""
    import torch
    scenes = [SCENE_4, TEST_JSON]
    gpu = P.make_gpu_sim(scenes, max_agents=64, **CLASSIC)            # default (legacy) stream: plain launches
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **CLASSIC)
    rng = np.random.default_rng(8)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                act = P.random_actions(rng, orc.W, orc.A, 0)
                RC.write_actions(gpu, act)                             # H2D + copy_ on `stream`
                gpu.step()
                got_done = gpu.done_tensor().to_torch().clone()       # read on `stream`, no device sync
                got_info = gpu.info_tensor().to_torch().clone()
                got_steps = gpu.steps_remaining_tensor().to_torch().clone()
                np.copyto(orc.action_tensor(), act)
                orc.step()
                assert np.array_equal(got_done.cpu().numpy(), np.asarray(orc.done_tensor()))
                assert np.array_equal(got_info.cpu().numpy(), np.asarray(orc.info_tensor()))
                assert np.array_equal(got_steps.cpu().numpy(), np.asarray(orc.steps_remaining_tensor()))
    assert gpu.stat(0) == 0
    run(s1, 5)
    assert gpu.stat(2) == 1 and gpu.stat(0) == 5, (gpu.stat(0), gpu.stat(1), gpu.stat(2))   # one capture, five replays
    run(s2, 4)                                                         # built on one stream, stepped on another
    assert gpu.stat(2) == 2 and gpu.stat(0) == 9
    P.compare_state(gpu, orc)
    gpu.close()


def test_environment_fallbacks_reach_the_build_specific_options(oracle_mod):
    """knn order and LiDAR cone through the environment, for callers that cannot pass constructor keywords."""
    import madrona_gpudrive as mg
    os.environ["GPUDRIVE_KNN_ORDER"] = "1"
    os.environ["GPUDRIVE_LIDAR_HALF_ANGLE"] = repr(float(np.pi))
    try:
        p = mg.Parameters()
        kw = dict(CLASSIC, enableLidar=1)
        for k, v in kw.items():
            if k in ("rewardType", "distanceToGoalThreshold"):
                setattr(p.rewardParams, k, v)
            else:
                setattr(p, k, v)
        scenes = [SCENE_4, TEST_JSON]
        gpu = mg.SimManager(mg.madrona.ExecMode.CUDA, 0, scenes, p, max_agents=64)   # the reference's own signature
    finally:
        del os.environ["GPUDRIVE_KNN_ORDER"], os.environ["GPUDRIVE_LIDAR_HALF_ANGLE"]
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, lidarHalfAngle=float(np.pi), **kw)
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    P.compare_roadmap_as_set(gpu, orc)          # set order: same rows ...
    g, o = RC.as_np(gpu.agent_roadmap_tensor()), np.asarray(orc.agent_roadmap_tensor())
    assert not np.allclose(g, o, atol=1e-3)      # ... in a different order
    assert P.compare_lidar(gpu, orc) > 0.05      # 360 degree cone
    gpu.close()


def test_gym_wrapper_call_sequence_on_the_device(oracle_mod):
    """SURVEY 8-b last row: action INDICES -> 7 x 13 table -> in-place [:, :, :3].copy_() -> step -> rewards / dones ->
    get_obs, the way GPUDriveTorchEnv drives the module (gpudrive_lab_amd/harness.py mirrors its methods); get_obs must
    equal the fused pack kernel and the oracle stepped with the table's values."""
    import torch
    from gpudrive_lab_amd.harness import TorchCallSequence
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "action_table_golden.npz"))
    scenes = [SCENE_4, SCENE_407, TEST_JSON]
    kw = dict(CLASSIC, collisionBehaviour=0)
    gpu = P.make_gpu_sim(scenes, max_agents=64, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    env = TorchCallSequence(gpu, dynamics_model="classic", reward_type="weighted_combination")
    assert np.array_equal(env.action_keys_tensor.cpu().numpy(), g["classic"])
    gen = torch.Generator().manual_seed(3)
    for k in range(15):
        idx = torch.randint(0, 91, (3, 64), generator=gen)
        env.step_dynamics(idx)
        act = np.asarray(orc.action_tensor())
        act[..., :3] = g["classic"][idx.numpy()]
        orc.step()
        assert np.array_equal(gpu.action_tensor().to_torch().cpu().numpy().view(np.uint32), act.view(np.uint32))
        P.compare_ints(gpu, orc, ["done_tensor", "info_tensor", "steps_remaining_tensor"])
        info = np.asarray(orc.info_tensor()).astype(np.float32)
        exp_r = -0.5 * info[..., 1:3].sum(-1) + 1.0 * info[..., 3] - 0.5 * info[..., 0]
        assert np.array_equal(env.get_rewards().cpu().numpy(), exp_r)
        assert np.array_equal(env.get_dones().cpu().numpy(), np.asarray(orc.done_tensor())[..., 0].astype(np.float32))
        obs = env.get_obs()
        packed = gpu.packed_observations()
        assert obs.shape == packed.shape == (3, 64, 6 + 63 * 6 + 200 * 13)
        assert torch.allclose(obs, packed, atol=3e-7, rtol=1e-6)
        gpu.debug_set_state(orc.get_state())   # keep the two simulators on the same state (teacher forcing)
    gpu.close()


def test_set_order_bounds_survive_resets_and_map_changes(oracle_mod):
    """The set-order selection prunes with last step's K-th distance (a bound that moves with the agent).  Free-running
    steps carry it over; a reset teleports agents (the bound must open up to the radius again) and set_maps changes the
    roads under them (the bound must be dropped): the rows must stay the reference's as a set through all of it."""
    scenes = [TEST_JSON, SCENE_407, SCENE_4]
    kw = dict(CLASSIC, polylineReductionThreshold=0.0, observationRadius=60.0)   # thousands of roads: K binds
    gpu = P.make_gpu_sim(scenes, max_agents=64, knn_order=1, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, **kw)
    rng = np.random.default_rng(17)

    def advance(k):
        for _ in range(k):
            act = P.random_actions(rng, orc.W, orc.A, 0)
            RC.write_actions(gpu, act)
            np.copyto(orc.action_tensor(), act)
            gpu.step()
            orc.step()
            gpu.debug_set_state(orc.get_state())   # same state on both sides, then one more coherent recompute
            gpu.reset([])
            orc.reset([])
            P.compare_roadmap_as_set(gpu, orc)
    advance(6)
    gpu.reset([0, 2])                                # agents jump back to their logged start poses
    orc.reset([0, 2])
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    P.compare_roadmap_as_set(gpu, orc)
    advance(3)
    new = [SCENE_4, TEST_JSON, SCENE_407]            # other roads under the same agent slots
    gpu.set_maps(new)
    orc.set_maps(new)
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    P.compare_roadmap_as_set(gpu, orc)
    advance(3)
    gpu.close()


def test_live_agent_lists_follow_delete_agents_and_set_maps(oracle_mod, monkeypatch):
    """BEV launches one workgroup per LIVE agent and the set-order road kernel one per live group of agent slots: both
    lists are rebuilt with the worlds.  After deleteAgents and set_maps (other agent counts per world) the rasters, the
    LiDAR rows and the road rows must still be the oracle's."""
    kw = dict(CLASSIC, enableLidar=1)
    scenes = [SCENE_407, TEST_JSON, SCENE_4]
    gpu = P.make_gpu_sim(scenes, max_agents=64, knn_order=1, enable_bev=True, **kw)
    orc = P.make_oracle_sim(oracle_mod, scenes, max_agents=64, enableBev=1, **kw)
    rng = np.random.default_rng(23)

    def check(steps):
        for _ in range(steps):
            act = P.random_actions(rng, orc.W, orc.A, 0)
            RC.write_actions(gpu, act)
            np.copyto(orc.action_tensor(), act)
            gpu.step()
            orc.step()
            gpu.debug_set_state(orc.get_state())
            gpu.reset([])
            orc.reset([])
            P.compare_roadmap_as_set(gpu, orc)
            P.compare_bev(gpu, orc)
            P.compare_lidar(gpu, orc)
            P.compare_ints(gpu, orc, ["shape_tensor", "done_tensor", "info_tensor"])
    check(2)
    ids = np.asarray(orc.agent_id_tensor())
    n0 = int(np.asarray(orc.shape_tensor())[0, 0])
    victims = {0: [int(ids[0, k]) for k in range(0, n0, 2)], 2: [int(ids[2, 1])]}   # half of world 0 goes
    gpu.deleteAgents(victims)
    orc.deleteAgents(victims)
    assert int(np.asarray(orc.shape_tensor())[0, 0]) < n0
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    check(2)
    new = [TEST_JSON, SCENE_4, SCENE_407]
    gpu.set_maps(new)
    orc.set_maps(new)
    gpu.debug_set_state(orc.get_state())
    gpu.reset([])
    orc.reset([])
    check(2)
    gpu.close()
