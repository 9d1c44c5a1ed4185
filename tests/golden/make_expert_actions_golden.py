#!/usr/bin/env python3
"""Generates tests/golden/expert_actions_golden.npz: golden vectors of the reference's expert-action
export (GPUDriveTorchEnv.get_expert_actions, gpudrive/env/env_torch.py:1445-1509).

The trajectory slicing is done by IMPORTING THE REFERENCE'S OWN `LogTrajectory`
(gpudrive/datatypes/trajectory.py); `env_torch` itself cannot be imported here (it needs gymnasium),
so the per-dynamics-model clamps of :1460-1499 are applied with the same torch calls, statement by
statement.  Run in the authoring container only (the reference never travels to the GPU box):

    GPUDRIVE_MAX_AGENTS=64 PYTHONPATH=/root/repo:/root/reference python tests/golden/make_expert_actions_golden.py

Input: the raw expert trajectory rows of the first 6 agents of a real scene (built by the CPU oracle;
any rows would do, they are data).  Extreme values are injected into a few inferred actions so that
every clamp bound is exercised.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gpudrive.datatypes.trajectory import LogTrajectory  # noqa: E402  (reference code)

from oracle import oracle as O  # noqa: E402


class _T:
    """Stands in for madrona.Tensor: the reference only calls .to_torch()."""

    def __init__(self, t):
        self.t = t

    def to_torch(self):
        return self.t


def expert_actions(raw, model):
    """env_torch.py:1445-1509 for one dynamics model ('classic', 'bicycle', 'delta_local', 'state')."""
    W, A = raw.shape[:2]
    log_trajectory = LogTrajectory.from_tensor(_T(torch.from_numpy(raw.copy())), W, A, backend="torch")
    if model == "delta_local":
        inferred_actions = log_trajectory.inferred_actions[..., :3]
        inferred_actions[..., 0] = torch.clamp(inferred_actions[..., 0], -6, 6)
        inferred_actions[..., 1] = torch.clamp(inferred_actions[..., 1], -6, 6)
        inferred_actions[..., 2] = torch.clamp(inferred_actions[..., 2], -torch.pi, torch.pi)
    elif model == "state":
        inferred_actions = torch.cat(
            (log_trajectory.pos_xy, torch.ones((*log_trajectory.pos_xy.shape[:-1], 1)), log_trajectory.yaw,
             log_trajectory.vel_xy, torch.zeros((*log_trajectory.pos_xy.shape[:-1], 4))), dim=-1)
    else:
        inferred_actions = log_trajectory.inferred_actions[..., :3]
        inferred_actions[..., 0] = torch.clamp(inferred_actions[..., 0], -6, 6)
        inferred_actions[..., 1] = torch.clamp(inferred_actions[..., 1], -0.3, 0.3)
    return (inferred_actions.numpy().copy(), log_trajectory.pos_xy.numpy().copy(), log_trajectory.vel_xy.numpy().copy(),
            log_trajectory.yaw.numpy().copy(), log_trajectory.valids.numpy().copy())


def main():
    scene = os.path.join(ROOT, "tests", "data", "tfrecord-00000-of-01000_4.json")
    sim = O.OracleSim([scene], O.default_params(polylineReductionThreshold=0.1), max_agents=64)
    raw = np.array(sim.expert_trajectory_tensor())[:, :6].copy()
    T = 91
    inf = raw[:, :, 6 * T:].reshape(1, 6, T, 10)
    inf[0, 0, 3, 0] = 9.5; inf[0, 0, 4, 0] = -7.25; inf[0, 1, 5, 1] = 0.31; inf[0, 1, 6, 1] = -4.0
    inf[0, 2, 7, 2] = 3.5; inf[0, 2, 8, 2] = -3.1415927; inf[0, 3, 9, 1] = 6.5; inf[0, 3, 10, 0] = np.nan
    out = {"raw": raw}
    for model in ("classic", "delta_local", "state"):
        act, pos, vel, yaw, valids = expert_actions(raw, model)
        out[model + "_actions"] = act
    out["pos_xy"], out["vel_xy"], out["yaw"], out["valids"] = pos, vel, yaw, valids
    path = os.path.join(ROOT, "tests", "golden", "expert_actions_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
