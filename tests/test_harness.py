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
        self.steps = 0

    def step(self):
        self.steps += 1

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
