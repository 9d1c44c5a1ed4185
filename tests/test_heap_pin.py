"""The k-NN row order pinned independently of the oracle's heap: knn.hpp's loop on libstdc++'s make_heap / pop_heap /
push_heap (tests/heap_pin.cpp) must order the rows exactly as oracle/gd_oracle.c does, ties included."""
import numpy as np
import pytest

from gpudrive_lab_amd import synth
from tests import heap_pin as HP
from tests.conftest import SCENE_4, SCENE_407, TEST_JSON

KNN = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0, dynamicsModel=0,
           roadObservationAlgorithm=0, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)


def _check(orc, radius, agents_per_world=None):
    shape = orc.shape_tensor()
    got = np.asarray(orc.agent_roadmap_tensor())
    checked = ties = 0
    for w in range(orc.W):
        n = int(shape[w, 0])
        picks = range(n) if agents_per_world is None else sorted(set(np.linspace(0, n - 1, agents_per_world).astype(int)))
        for a in picks:
            rows, order, keys = HP.expected_rows(orc, w, a, radius)
            assert np.array_equal(got[w, a].view(np.uint32), rows.view(np.uint32)), (w, a)
            sel = order[order >= 0]
            ties += len(sel) - len(np.unique(keys[sel]))
            checked += 1
    return checked, ties


@pytest.mark.parametrize("thr,radius", [(0.1, 50.0), (0.0, 100.0)])
def test_oracle_order_equals_libstdcxx_on_waymo_scenes(oracle_mod, tmp_path, thr, radius):
    O = oracle_mod
    scenes = [TEST_JSON, SCENE_407, SCENE_4] if thr > 0 else [TEST_JSON]
    kw = dict(KNN, polylineReductionThreshold=thr, observationRadius=radius)
    orc = O.OracleSim(scenes, O.default_params(**kw), max_agents=64)
    rng = np.random.default_rng(3)
    for _ in range(3):  # a few steps so that agents have moved off their logged start poses
        act = orc.action_tensor()
        act[..., 0] = rng.uniform(-3, 2, act.shape[:2])
        act[..., 1] = rng.uniform(-0.7, 0.7, act.shape[:2])
        orc.step()
    checked, _ = _check(orc, radius, agents_per_world=None if thr > 0 else 6)
    assert checked >= 6
    orc.close()


def test_oracle_order_equals_libstdcxx_on_the_bench_scene(oracle_mod, tmp_path):
    O = oracle_mod
    paths = synth.write_scenes(str(tmp_path), [0])
    kw = dict(KNN, polylineReductionThreshold=0.0)
    orc = O.OracleSim(paths, O.default_params(**kw), max_agents=64)
    checked, _ = _check(orc, 50.0, agents_per_world=8)
    assert checked == 8
    orc.close()


def test_equal_keys_are_ordered_like_libstdcxx(oracle_mod, tmp_path):
    """Duplicate road geometry gives exactly equal keys; the heap's strict `<` decides where they end up."""
    O = oracle_mod
    import json
    sc = synth.make_scene(5, n_agents=4, n_polylines=6, pts_per_polyline=60)
    sc["roads"] = sc["roads"] + [dict(r, id=100 + i) for i, r in enumerate(sc["roads"][:3])]  # three polylines twice
    p = tmp_path / "dup.json"
    p.write_text(json.dumps(sc))
    kw = dict(KNN, polylineReductionThreshold=0.0, observationRadius=200.0)
    orc = O.OracleSim([str(p)], O.default_params(**kw), max_agents=64)
    checked, ties = _check(orc, 200.0)
    assert checked == 4 and ties > 0
    orc.close()
