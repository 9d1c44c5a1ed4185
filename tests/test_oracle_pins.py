"""CPU suite: pin the oracle against every known-answer test the reference holds for the
step path.  The same cases run against the HIP path in test_gpu_parity.py."""
import math

import numpy as np
import pytest

from tests import ref_cases as RC
from tests.conftest import TEST_JSON

f32 = np.float32


@pytest.fixture(scope="module")
def make_sim(oracle_mod):
    O = oracle_mod

    def _mk(scenes, max_agents=128, **kw):
        return O.OracleSim(scenes, O.default_params(**kw), max_agents=max_agents)
    return _mk


# ---- tests/CollisionDetectionTests.cpp:11-85 ----
def test_obb_axis_aligned_colliding(oracle_mod):
    assert oracle_mod.obb_collide([0, 0, 0], 0, [1, 1, 1], [1, 1, 1], 0, [1, 1, 1])


def test_obb_axis_aligned_not_colliding(oracle_mod):
    assert not oracle_mod.obb_collide([0, 0, 0], 0, [.5, .5, 1], [2, 2, 0], 0, [.5, .5, 1])


def test_obb_point_intersection_counts(oracle_mod):
    assert oracle_mod.obb_collide([0, 0, 0], 0, [.5, .5, .5], [1, 1, 0], 0, [.5, .5, .5])


def test_obb_one_inside_other(oracle_mod):
    assert oracle_mod.obb_collide([0, 0, 0], 0, [1, 1, 1], [0, 0, 0], 0, [.5, .5, .5])


def test_obb_exhaustive_rotations(oracle_mod):
    deg = f32(0)
    while deg < 360:
        rad = f32(deg) * f32(f32(math.pi) / f32(180))
        assert oracle_mod.obb_collide([0, 0, 0], 0, [1, 1, 1], [.5, .5, 0], rad, [1, 1, 1])
        deg = f32(deg + 15)


# ---- tests/EgocentricRoadObservationTests.cpp:9-22 ----
def test_reference_frame_relative(oracle_mod):
    to_rad = lambda d: f32(f32(d) * f32(f32(math.pi) / f32(180)))
    obs = oracle_mod.reference_frame_obs([3, 0], to_rad(90), [3, 3, 0], to_rad(270), [10, .1, .1])
    assert obs[0] - 3 < 1e-6
    assert obs[1] - 0 < 1e-6
    assert obs[5] == to_rad(180)  # EXPECT_EQ: exactly +pi
    # magnitude as well (the reference only checks one side)
    assert abs(obs[0] - 3) < 1e-5 and abs(obs[1]) < 1e-5


# ---- integration cases ----
def test_bicycle_model(make_sim):
    RC.check_bicycle_model(make_sim, TEST_JSON)


def test_map_observation(make_sim):
    RC.check_map_observation(make_sim, TEST_JSON)


def test_delta_model(make_sim):
    RC.check_delta_model(make_sim, TEST_JSON)


def test_waymax_model(make_sim):
    RC.check_waymax_model(make_sim, TEST_JSON)


def test_expert_replay(make_sim):
    RC.check_expert_replay(make_sim, TEST_JSON)
