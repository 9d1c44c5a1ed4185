#!/usr/bin/env python3
"""Generates tests/golden/obs_pack_golden.npz: golden input/output vectors of the reference's
observation assembly (gpudrive/env/env_torch.py:756-896,1172-1216 via gpudrive/datatypes/
observation.py and roadgraph.py), produced by IMPORTING THE REFERENCE'S OWN PYTHON CODE.

Run in the authoring container only (the reference never travels to the GPU box):

    GPUDRIVE_MAX_AGENTS=64 PYTHONPATH=/root/repo:/root/reference python tests/golden/make_obs_pack_golden.py

Inputs are the raw self / partner / road-map tensors of 8 agents of a real scene (from the CPU oracle,
any plausible values would do); the expected output is what `GPUDriveTorchEnv.get_obs()` concatenates
for them with norm_obs=True: ego(6) + partners((A-1)*6) + road points(200*13).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gpudrive.datatypes.observation import LocalEgoState, PartnerObs  # noqa: E402  (reference code)
from gpudrive.datatypes.roadgraph import LocalRoadGraphPoints  # noqa: E402  (reference code)

from oracle import oracle as O  # noqa: E402


class _T:
    """Stands in for madrona.Tensor: the reference only calls .to_torch()."""

    def __init__(self, t):
        self.t = t

    def to_torch(self):
        return self.t


def main():
    scene = os.path.join(ROOT, "tests", "data", "tfrecord-00000-of-01000_4.json")
    p = O.default_params(polylineReductionThreshold=0.1, observationRadius=50.0, collisionBehaviour=2, rewardType=1,
                         distanceToGoalThreshold=2.0, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)
    sim = O.OracleSim([scene], p, max_agents=64)
    for _ in range(5):
        sim.step()
    n = 8
    self_obs = np.array(sim.self_observation_tensor()[:, :n])
    partner = np.array(sim.partner_observations_tensor()[:, :n])
    roadmap = np.array(sim.agent_roadmap_tensor()[:, :n])

    # gpudrive/env/env_torch.py:756-800 (_get_ego_state, norm_obs, not reward_conditioned)
    ego = LocalEgoState.from_tensor(_T(torch.from_numpy(self_obs.copy())), backend="torch", device="cpu")
    ego.normalize()
    ego_t = torch.stack([ego.speed, ego.vehicle_length, ego.vehicle_width, ego.rel_goal_x, ego.rel_goal_y,
                         ego.is_collided], dim=-1)
    # :828-858 (_get_partner_obs)
    po = PartnerObs.from_tensor(_T(torch.from_numpy(partner.copy())), backend="torch", device="cpu")
    po.normalize()
    part_t = torch.concat([po.speed, po.rel_pos_x, po.rel_pos_y, po.orientation, po.vehicle_length,
                           po.vehicle_width], dim=-1).flatten(start_dim=2)
    # :860-896 (_get_road_map_obs)
    rg = LocalRoadGraphPoints.from_tensor(_T(torch.from_numpy(roadmap.copy())), backend="torch", device="cpu")
    rg.one_hot_encode_road_point_types()
    rg.normalize()
    road_t = torch.cat([rg.x.unsqueeze(-1), rg.y.unsqueeze(-1), rg.segment_length.unsqueeze(-1),
                        rg.segment_width.unsqueeze(-1), rg.segment_height.unsqueeze(-1),
                        rg.orientation.unsqueeze(-1), rg.type], dim=-1).flatten(start_dim=2)
    # :1172-1201 (get_obs concatenation)
    obs = torch.cat((ego_t, part_t, road_t), dim=-1).numpy().astype(np.float32)
    out = os.path.join(ROOT, "tests", "golden", "obs_pack_golden.npz")
    np.savez_compressed(out, self_obs=self_obs[0], partner=partner[0], roadmap=roadmap[0], expected=obs[0])
    print("wrote", out, obs.shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
