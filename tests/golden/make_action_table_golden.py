#!/usr/bin/env python3
"""Generates tests/golden/action_table_golden.npz: the discrete action tables the reference's gym wrapper builds,
from THE REFERENCE'S OWN value lists (gpudrive/env/config.py:62-77, imported here) enumerated the way
GPUDriveTorchEnv._set_discrete_action_space does (itertools.product over (accel, steer, head) resp. (dx, dy, dyaw),
index = enumeration order; gpudrive/env/env_torch.py:666-724 -- that module itself needs gymnasium, which this image
lacks, so its three-line enumeration is restated below and cited).

    GPUDRIVE_MAX_AGENTS=64 PYTHONPATH=/root/repo:/root/reference python tests/golden/make_action_table_golden.py
"""
import os
import sys
from itertools import product

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gpudrive.env.config import EnvConfig  # noqa: E402  (reference code)


def table(a1, a2, a3):
    # env_torch.py:702-721: action_key_to_values[idx] = [a.item(), b.item(), c.item()] in product order,
    # action_keys_tensor = torch.tensor([... for key in sorted(keys)])
    d = {i: [x.item(), y.item(), z.item()] for i, (x, y, z) in enumerate(product(a1, a2, a3))}
    return torch.tensor([d[k] for k in sorted(d.keys())]).numpy()


def main():
    c = EnvConfig()
    out = dict(classic=table(c.accel_actions, c.steer_actions, c.head_tilt_actions),
               delta_local=table(c.dx, c.dy, c.dyaw),
               steer_actions=c.steer_actions.numpy(), accel_actions=c.accel_actions.numpy(),
               head_tilt_actions=c.head_tilt_actions.numpy(), dx=c.dx.numpy(), dy=c.dy.numpy(), dyaw=c.dyaw.numpy())
    path = os.path.join(ROOT, "tests", "golden", "action_table_golden.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
