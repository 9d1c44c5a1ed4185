"""Runs tests/heap_pin.cpp (knn.hpp's loop on libstdc++'s heap functions) and turns its road order into the rows
agent_roadmap_tensor must hold.  Shared by the CPU suite (oracle vs libstdc++) and the GPU suite (HIP path vs libstdc++)."""
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_BIN = None


def binary():
    global _BIN
    if _BIN is None:
        out = os.path.join(tempfile.gettempdir(), "gd_heap_pin_%d" % os.getuid())
        src = os.path.join(HERE, "heap_pin.cpp")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", out, src])
        _BIN = out
    return _BIN


def libstdcxx_order(keys, radius, K=200):
    """Road index per output slot (-1 = zero-filled) for one agent's key sequence."""
    keys = np.ascontiguousarray(keys, np.float32)
    blob = struct.pack("<iif", K, len(keys), float(radius)) + keys.tobytes()
    out = subprocess.run([binary()], input=blob, stdout=subprocess.PIPE, check=True).stdout
    return np.frombuffer(out, np.int32).copy()


def expected_rows(orc, w, a, radius, K=200):
    """agent_roadmap rows of agent (w, a) as the libstdc++ run orders them: the oracle's observationOf of every road
    (position keys in float32, x*x + y*y like Vector2::length2) -> order -> rows; zero-filled rows are fillZeros
    (id 0, mapType 0: knn.hpp:19-28)."""
    obs = orc.road_obs_of(w, a)
    keys = obs[:, 0] * obs[:, 0] + obs[:, 1] * obs[:, 1]  # float32 arithmetic, no contraction
    order = libstdcxx_order(keys, radius, K)
    rows = np.zeros((K, 9), np.float32)
    sel = order >= 0
    rows[sel] = obs[order[sel]]
    return rows, order, keys
