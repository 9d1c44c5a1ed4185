#!/usr/bin/env python3
"""Generates tests/golden/columns_golden.npz: the exported tensors of a small real scene (built and stepped
by the CPU oracle) decoded BY THE REFERENCE'S OWN PYTHON CLASSES (gpudrive/datatypes/*.py).  The test
(tests/test_columns.py) requires gpudrive_lab_amd/columns.py to name the same columns.

    GPUDRIVE_MAX_AGENTS=64 PYTHONPATH=/root/repo:/root/reference python tests/golden/make_columns_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gpudrive.datatypes.control import ResponseType  # noqa: E402  (reference code)
from gpudrive.datatypes.info import Info  # noqa: E402
from gpudrive.datatypes.metadata import Metadata  # noqa: E402
from gpudrive.datatypes.observation import GlobalEgoState, LocalEgoState, PartnerObs  # noqa: E402
from gpudrive.datatypes.roadgraph import GlobalRoadGraphPoints, LocalRoadGraphPoints  # noqa: E402

from oracle import oracle as O  # noqa: E402


class _T:
    def __init__(self, t):
        self.t = t

    def to_torch(self):
        return self.t


def main():
    scene = os.path.join(ROOT, "tests", "data", "test.json")
    p = O.default_params(polylineReductionThreshold=0.5, observationRadius=30.0, collisionBehaviour=0, rewardType=1,
                         distanceToGoalThreshold=2.0, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)
    sim = O.OracleSim([scene], p, max_agents=64)
    rng = np.random.default_rng(5)
    for _ in range(12):
        act = sim.action_tensor()
        act[..., 0] = rng.uniform(0, 3, act.shape[:2]); act[..., 1] = rng.uniform(-0.3, 0.3, act.shape[:2])
        sim.step()
    n = 6  # agents kept (partner rows keep all 63 slots)
    raw = dict(self_obs=np.array(sim.self_observation_tensor())[:, :n], abs_obs=np.array(sim.absolute_self_observation_tensor())[:, :n],
               partner=np.array(sim.partner_observations_tensor())[:, :n], roadmap=np.array(sim.agent_roadmap_tensor())[:, :n, :40],
               map_obs=np.array(sim.map_observation_tensor())[:, :60], info=np.array(sim.info_tensor())[:, :n],
               metadata=np.array(sim.metadata_tensor())[:, :n], response=np.array(sim.response_type_tensor())[:, :n])
    t = lambda x: _T(torch.from_numpy(x.copy()))
    out = {"raw_" + k: v for k, v in raw.items()}

    def dump(prefix, obj, names):
        for nme in names:
            out[prefix + "." + nme] = getattr(obj, nme).numpy()

    dump("SELF_OBS", LocalEgoState.from_tensor(t(raw["self_obs"]), backend="torch", device="cpu"),
         ["speed", "vehicle_length", "vehicle_width", "vehicle_height", "rel_goal_x", "rel_goal_y", "is_collided", "id"])
    dump("ABS_OBS", GlobalEgoState.from_tensor(t(raw["abs_obs"]), backend="torch", device="cpu"),
         ["pos_x", "pos_y", "pos_z", "rotation_as_quaternion", "rotation_angle", "goal_x", "goal_y", "vehicle_length",
          "vehicle_width", "vehicle_height", "id"])
    dump("PARTNER_OBS", PartnerObs.from_tensor(t(raw["partner"]), backend="torch", device="cpu"),
         ["speed", "rel_pos_x", "rel_pos_y", "orientation", "vehicle_length", "vehicle_width", "vehicle_height", "agent_type", "ids"])
    dump("ROAD_ROW_local", LocalRoadGraphPoints.from_tensor(t(raw["roadmap"]), backend="torch", device="cpu"),
         ["x", "y", "segment_length", "segment_width", "segment_height", "orientation", "type", "id"])
    dump("ROAD_ROW_global", GlobalRoadGraphPoints.from_tensor(t(raw["map_obs"]), backend="torch", device="cpu"),
         ["x", "y", "segment_length", "segment_width", "segment_height", "orientation", "type", "id", "vbd_type"])
    dump("INFO", Info.from_tensor(t(raw["info"]), backend="torch", device="cpu"), ["off_road", "collided", "goal_achieved"])
    dump("METADATA", Metadata.from_tensor(t(raw["metadata"]), backend="torch"),
         ["is_sdc", "objects_of_interest", "tracks_to_predict", "difficulty"])
    rt = ResponseType.from_tensor(t(raw["response"]), backend="torch", device="cpu")
    dump("RESPONSE_TYPE", rt, ["moving", "kinematic", "static"])
    path = os.path.join(ROOT, "tests", "golden", "columns_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
