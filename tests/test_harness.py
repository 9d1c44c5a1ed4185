"""The torch-only call-sequence harness (gpudrive_lab_amd/harness.py, SURVEY.md 8-b last row) against golden vectors:
the discrete action tables built from the reference's own value lists, and the observation assembly against the
vectors the reference's gpudrive/datatypes classes produced (tests/golden/make_obs_pack_golden.py)."""
import os

import numpy as np
import torch

from gpudrive_lab_amd.harness import TorchCallSequence
from tests.conftest import ROOT


class _T:
    def __init__(self, t):
        self.t = t

    def to_torch(self):
        return self.t


class FakeSim:
    """Holds exported tensors like SimManager does (aliasing views), steps nothing."""

    def __init__(self, W=2, A=64):
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt)
        self.t = dict(action=z(W, A, 10), done=z(W, A, 1, dt=torch.int32), info=z(W, A, 5, dt=torch.int32), reward=z(W, A, 1),
                      self_obs=z(W, A, 8), partner=z(W, A, A - 1, 9), roadmap=z(W, A, 200, 9))
        self.t.update(controlled=z(W, A, 1, dt=torch.int32), traj=z(W, A, 16 * 91), map_name=z(W, 32, dt=torch.int32),
                      scenario=z(W, 32, dt=torch.int32))
        self.t.update(resp=z(W, A, 1, dt=torch.int32), lidar=z(W, A, 3, 30, 4), bev=z(W, A, 8, 8, 1))
        self.steps = 0
        self.calls = []

    def step(self):
        self.steps += 1
        self.calls.append(("step", self.t["action"][:, :, :3].clone()))

    def reset(self, worlds):
        self.calls.append(("reset", list(worlds)))

    def set_maps(self, maps):
        self.calls.append(("set_maps", list(maps)))

    def deleteAgents(self, d):
        self.calls.append(("deleteAgents", dict(d)))

    response_type_tensor = lambda self: _T(self.t["resp"])
    lidar_tensor = lambda self: _T(self.t["lidar"])
    bev_observation_tensor = lambda self: _T(self.t["bev"])
    controlled_state_tensor = lambda self: _T(self.t["controlled"])
    expert_trajectory_tensor = lambda self: _T(self.t["traj"])
    map_name_tensor = lambda self: _T(self.t["map_name"])
    scenario_id_tensor = lambda self: _T(self.t["scenario"])

    action_tensor = lambda self: _T(self.t["action"])
    done_tensor = lambda self: _T(self.t["done"])
    info_tensor = lambda self: _T(self.t["info"])
    reward_tensor = lambda self: _T(self.t["reward"])
    self_observation_tensor = lambda self: _T(self.t["self_obs"])
    partner_observations_tensor = lambda self: _T(self.t["partner"])
    agent_roadmap_tensor = lambda self: _T(self.t["roadmap"])


def test_action_tables_match_the_reference_value_lists():
    g = np.load(os.path.join(ROOT, "tests", "golden", "action_table_golden.npz"))
    for model in ("classic", "delta_local"):
        h = TorchCallSequence(FakeSim(), dynamics_model=model)
        tab = h.action_keys_tensor.numpy()
        assert tab.shape == g[model].shape and np.array_equal(tab.view(np.uint32), g[model].view(np.uint32)), model
    h = TorchCallSequence(FakeSim(), dynamics_model="classic")
    assert h.action_keys_tensor.shape == (91, 3)  # 7 accelerations x 13 steering angles x 1 head tilt
    assert h.values_to_action_key[(0.0, 0.0, 0.0)] == 3 * 13 + 6


def test_indices_go_through_the_table_into_the_action_tensor_in_place():
    sim = FakeSim()
    h = TorchCallSequence(sim, dynamics_model="classic")
    before = sim.action_tensor().to_torch()
    idx = torch.randint(0, 91, (2, 64), generator=torch.Generator().manual_seed(0))
    idx_f = idx.to(torch.float32)
    idx_f[0, 0] = float("nan")  # nan_to_num -> index 0 (env_torch.py:626-628)
    h.step_dynamics(idx_f)
    after = sim.action_tensor().to_torch()
    assert after.data_ptr() == before.data_ptr() and sim.steps == 1
    exp = h.action_keys_tensor[idx]
    exp[0, 0] = h.action_keys_tensor[0]
    assert torch.equal(after[:, :, :3], exp) and torch.equal(after[:, :, 3:], torch.zeros(2, 64, 7))
    h.step_dynamics(idx.unsqueeze(-1))  # [W, A, 1] indices
    assert torch.equal(sim.action_tensor().to_torch()[:, :, :3], h.action_keys_tensor[idx])
    vals = torch.rand(2, 64, 3)
    h.step_dynamics(vals)  # [W, A, 3] values pass through
    assert torch.equal(sim.action_tensor().to_torch()[:, :, :3], vals)
    assert (h.world_time_steps == 3).all()


def test_observation_assembly_matches_the_reference_python_golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "obs_pack_golden.npz"))
    n = g["self_obs"].shape[0]
    sim = FakeSim(W=1)
    sim.t["self_obs"][0, :n] = torch.from_numpy(g["self_obs"])
    sim.t["partner"][0, :n] = torch.from_numpy(g["partner"])
    sim.t["roadmap"][0, :n] = torch.from_numpy(g["roadmap"])
    h = TorchCallSequence(sim, dynamics_model="classic")
    obs = h.get_obs()
    assert obs.shape == (1, 64, 6 + 63 * 6 + 200 * 13)
    got = obs[0, :n].numpy()
    assert np.array_equal(got.view(np.uint32), g["expected"].view(np.uint32)), np.abs(got - g["expected"]).max()


def test_rewards_and_dones_follow_the_wrapper():
    sim = FakeSim(W=1)
    sim.t["info"][0, 0] = torch.tensor([1, 0, 1, 0, 7], dtype=torch.int32)  # off road + hit a non-vehicle
    sim.t["info"][0, 1] = torch.tensor([0, 1, 0, 1, 7], dtype=torch.int32)  # hit a vehicle + goal
    sim.t["reward"][0, 1, 0] = 1.0
    sim.t["done"][0, 1, 0] = 1
    h = TorchCallSequence(sim, reward_type="weighted_combination")
    r = h.get_rewards(collision_weight=-0.75, goal_achieved_weight=1.0, off_road_weight=-0.5)
    assert r[0, 0].item() == -1.25 and r[0, 1].item() == 0.25
    assert TorchCallSequence(sim).get_rewards()[0, 1].item() == 1.0
    d = h.get_dones()
    assert d.dtype == torch.float32 and d[0, 1].item() == 1.0 and d.sum().item() == 1.0


def _fill_traj(sim, seed=5):
    g = torch.Generator().manual_seed(seed)
    sim.t["traj"].copy_(torch.rand(sim.t["traj"].shape, generator=g) * 20 - 10)
    sim.t["traj"][:, :, 5 * 91:6 * 91] = (torch.rand(sim.t["traj"][:, :, 5 * 91:6 * 91].shape, generator=g) < 0.7).float()


def test_expert_actions_follow_the_trajectory_layout_and_the_clamps():
    """gpudrive/datatypes/trajectory.py:21-40 + env_torch.py:1445-1509, restated independently with numpy strides."""
    sim = FakeSim(W=2, A=64)
    _fill_traj(sim)
    raw = sim.t["traj"].numpy().copy()
    T = 91
    inferred = raw[:, :, 6 * T:].reshape(2, 64, T, 10)
    for model in ("classic", "delta_local", "state"):
        act, pos, vel, yaw, valid = TorchCallSequence(sim, dynamics_model=model).get_expert_actions()
        assert np.array_equal(pos.numpy(), raw[:, :, :2 * T].reshape(2, 64, T, 2))
        assert np.array_equal(vel.numpy(), raw[:, :, 2 * T:4 * T].reshape(2, 64, T, 2))
        assert np.array_equal(yaw.numpy(), raw[:, :, 4 * T:5 * T].reshape(2, 64, T, 1))
        assert valid.dtype == torch.int32 and np.array_equal(valid.numpy(), raw[:, :, 5 * T:6 * T].reshape(2, 64, T, 1).astype(np.int32))
        if model == "classic":
            exp = inferred[..., :3].copy()
            exp[..., 0] = np.clip(exp[..., 0], -6, 6); exp[..., 1] = np.clip(exp[..., 1], -0.3, 0.3)
        elif model == "delta_local":
            exp = inferred[..., :3].copy()
            exp[..., 0] = np.clip(exp[..., 0], -6, 6); exp[..., 1] = np.clip(exp[..., 1], -6, 6)
            exp[..., 2] = np.clip(exp[..., 2], -np.float32(np.pi), np.float32(np.pi))
        else:
            exp = np.concatenate([pos.numpy(), np.ones((2, 64, T, 1), np.float32), yaw.numpy(), vel.numpy(),
                                  np.zeros((2, 64, T, 4), np.float32)], axis=-1)
        assert act.shape == exp.shape and np.array_equal(act.numpy(), exp), model
    assert np.array_equal(sim.t["traj"].numpy(), raw), "the exported tensor must not be clamped in place"


def test_reset_zeroes_the_clocks_and_plays_the_log_for_the_warm_up_steps():
    sim = FakeSim(W=2, A=64)
    _fill_traj(sim)
    h = TorchCallSequence(sim, dynamics_model="classic", init_steps=3)
    h.world_time_steps += 7
    obs = h.reset()
    assert obs.shape == (2, 64, 6 + 63 * 6 + 200 * 13)
    assert sim.calls[0] == ("reset", [0, 1])
    steps = [c for c in sim.calls if c[0] == "step"]
    assert len(steps) == 3 and (h.world_time_steps == 3).all()   # (zeroed, then three logged steps)
    exp = h.get_expert_actions()[0]
    for k in range(3):
        assert torch.equal(steps[k][1], exp[:, :, k, :])
    sim.calls.clear()
    mask = torch.zeros(2, 64, dtype=torch.bool)
    mask[1, :5] = True
    assert TorchCallSequence(sim).reset(mask=mask, env_idx_list=[1]).shape == (5, 6 + 63 * 6 + 200 * 13)
    assert sim.calls[0] == ("reset", [1])
    try:
        h.advance_sim_with_log_playback(init_steps=91)
        raise AssertionError("init_steps = 91 must be refused")
    except ValueError:
        pass


def test_infos_masks_names_and_scene_changes():
    sim = FakeSim(W=2, A=64)
    sim.t["info"][0, 0] = torch.tensor([1, 1, 1, 0, 7], dtype=torch.int32)
    sim.t["info"][1, 3] = torch.tensor([0, 0, 0, 1, 7], dtype=torch.int32)
    sim.t["controlled"][0, :4, 0] = 1
    sim.t["controlled"][1, 2:5, 0] = 1
    sim.t["self_obs"][:, :, 7] = -1
    sim.t["self_obs"][0, :6, 7] = torch.arange(100, 106).float()
    sim.t["self_obs"][1, :6, 7] = torch.arange(200, 206).float()
    for w, name in enumerate(("tfrecord-00001.json", "x.json")):
        sim.t["map_name"][w, :len(name)] = torch.tensor([ord(c) for c in name], dtype=torch.int32)
        sim.t["scenario"][w, :3] = torch.tensor([ord(c) for c in "ab%d" % w], dtype=torch.int32)
    h = TorchCallSequence(sim)
    info = h.get_infos()
    assert info.shape == (2, 64) and info.off_road[0, 0] == 1 and info.collided[0, 0] == 2 and info.goal_achieved[1, 3] == 1
    m = h.get_controlled_agents_mask()
    assert m.dtype == torch.bool and m.shape == (2, 64) and m.sum().item() == 7 and h.num_valid_controlled_agents_across_worlds == 7
    assert h.get_env_filenames() == {0: "tfrecord-00001.json", 1: "x.json"} and h.get_scenario_ids() == {0: "ab0", 1: "ab1"}
    # half of the controlled agents of every world, by id, at least one
    h.remove_agents_by_id(0.5, remove_controlled_agents=True, generator=torch.Generator().manual_seed(1))
    dels = [c[1] for c in sim.calls if c[0] == "deleteAgents"]
    assert len(dels) == 2 and len(dels[0][0]) == 2 and len(dels[1][1]) == 1
    assert set(dels[0][0]) <= {100, 101, 102, 103} and set(dels[1][1]) <= {202, 203, 204}
    sim.calls.clear()
    h.remove_agents_by_id(0.34, remove_controlled_agents=False, generator=torch.Generator().manual_seed(2))
    dels = [c[1] for c in sim.calls if c[0] == "deleteAgents"]
    assert set(dels[0][0]) <= {104, 105} and set(dels[1][1]) <= {200, 201, 205}   # uncontrolled agents with an id
    h.remove_agents_by_id(0.0)
    sim.calls.clear()
    try:
        h.swap_data_batch(["a.json"])
        raise AssertionError("a batch of the wrong size must be refused")
    except ValueError:
        pass
    sim.t["controlled"][0, :4, 0] = 0
    h.swap_data_batch(["a.json", "b.json"])
    assert sim.calls == [("set_maps", ["a.json", "b.json"])] and h.num_valid_controlled_agents_across_worlds == 3


def test_expert_actions_match_the_reference_datatypes_golden():
    """tests/golden/expert_actions_golden.npz was produced by the reference's own LogTrajectory class and the wrapper's
    clamps (make_expert_actions_golden.py): the harness's restatement gives the same bits."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "expert_actions_golden.npz"))
    n = g["raw"].shape[1]
    sim = FakeSim(W=1, A=64)
    sim.t["traj"][0, :n] = torch.from_numpy(g["raw"][0])
    for model in ("classic", "delta_local", "state"):
        act, pos, vel, yaw, valid = TorchCallSequence(sim, dynamics_model=model).get_expert_actions()
        assert np.array_equal(act[:, :n].numpy().view(np.uint32), g[model + "_actions"].view(np.uint32)), model
        assert np.array_equal(pos[:, :n].numpy(), g["pos_xy"]) and np.array_equal(vel[:, :n].numpy(), g["vel_xy"])
        assert np.array_equal(yaw[:, :n].numpy(), g["yaw"]) and np.array_equal(valid[:, :n].numpy(), g["valids"])


def test_masks_sensors_and_frame_stacking():
    sim = FakeSim(W=1, A=64)
    sim.t["partner"][..., 8] = -2            # nobody anywhere ...
    sim.t["partner"][0, 0, 0] = torch.tensor([3.0, 1.0, 2.0, 0.1, 4.5, 2.0, 1.5, 1.0, 11.0])   # ... but ego 0 sees agent 1 (slot 0)
    sim.t["partner"][0, 0, 1] = torch.tensor([0.0, 5.0, 6.0, 0.2, 4.5, 2.0, 1.5, 1.0, 12.0])   # ... and the parked agent 2 (slot 1)
    sim.t["partner"][0, 2, 0] = torch.tensor([3.0, -5.0, -6.0, 0.0, 4.5, 2.0, 1.5, 1.0, 10.0])  # ego 2 sees agent 0
    sim.t["resp"][0, 2, 0] = 2               # agent 2 is Static
    sim.t["resp"][0, 3:, 0] = 2              # padding slots are Static too, with all-zero rows
    sim.t["roadmap"][..., 7] = -1
    sim.t["roadmap"][0, 0, :5, 7] = torch.arange(5).float()
    h = TorchCallSequence(sim, num_stack=3)
    first = h.get_obs(reset=True)
    D = 6 + 63 * 6 + 200 * 13
    assert first.shape == (1, 64, 3 * D) and (first[..., :2 * D] == 0).all()
    m = h.get_partner_mask()
    assert m.shape == (1, 64, 63)
    assert m[0, 0, 0] == 0 and m[0, 0, 1] == 1 and m[0, 0, 2] == 2      # acting partner, parked partner, nobody
    assert m[0, 2, 0] == 0 and (m[0, 1] == 2).all()
    rm = h.get_road_mask()
    assert rm.shape == (1, 64, 200) and rm[0, 0, :5].sum() == 0 and rm[0, 0, 5:].all() and rm[0, 1].all()
    sim.t["self_obs"][0, 0, 0] = 50.0        # the next frame differs
    second = h.get_obs()
    assert torch.equal(second[..., D:2 * D], first[..., 2 * D:]) and (second[..., :D] == 0).all()
    assert second[0, 0, 2 * D].item() == 0.5 and first[0, 0, 2 * D].item() == 0.0
    assert h.get_obs(reset=True)[..., :2 * D].abs().sum() == 0              # a reset forgets the earlier frames
    # LiDAR: the three planes side by side per ray; BEV: one-hot over the 11 entity types
    sim.t["lidar"][0, 0] = torch.arange(3 * 30 * 4).float().view(3, 30, 4)
    lo = h._get_lidar_obs()
    assert lo.shape == (1, 64, 30 * 12)
    assert torch.equal(lo[0, 0].view(30, 12)[:, 4:8], sim.t["lidar"][0, 0, 1])
    parts = h._get_lidar_obs(mask=torch.tensor([[True] + [False] * 63]))
    assert len(parts) == 3 and parts[2].shape == (1, 30, 4)
    sim.t["bev"][0, 0, 2, 3, 0] = 7.0
    bo = h._get_bev_obs()
    assert bo.shape == (1, 64, 8 * 8 * 11) and bo[0, 0].view(8, 8, 11)[2, 3, 7] == 1 and bo[0, 0].view(8, 8, 11)[0, 0, 0] == 1
