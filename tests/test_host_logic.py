"""CPU suite: the C-ABI library loads and exports every declared symbol, and the product's host
world builder (C++) produces the same init-time rows as the oracle (numpy parser + C builder).
No compute entry point is called here (no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gpudrive_lab_amd import _capi
from tests.conftest import ROOT, SCENE_4, SCENE_407, TEST_JSON


@pytest.fixture(scope="module")
def L():
    _capi.build()
    return _capi.lib()


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "gpudrive_amd.h")).read()
    declared = set(re.findall(r"\b(gd_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.gd_version().startswith(b"gpudrive_amd")


def test_struct_sizes_match_header(L):
    # gd_params: 15 4-byte fields; gd_tensor_desc: ptr + 2 int32 + 5 int64 + int64
    assert C.sizeof(_capi.GdParams) == 60
    assert C.sizeof(_capi.GdTensorDesc) == 8 + 8 + 40 + 8


def test_default_params(L):
    p = _capi.GdParams()
    L.gd_default_params(C.byref(p))
    assert p.collisionBehaviour == 0 and p.maxNumControlledAgents == 10000
    assert p.initOnlyValidAgentsAtFirstStep == 1 and p.dynamicsModel == 0 and p.roadObservationAlgorithm == 0


def test_tensor_shapes(L):
    d = _capi.GdTensorDesc()
    exp = {
        _capi.T_ACTION: (3, [7, 64, 10]), _capi.T_PARTNER_OBS: (4, [7, 64, 63, 9]),
        _capi.T_AGENT_MAP_OBS: (4, [7, 64, 200, 9]), _capi.T_MAP_OBS: (3, [7, 10000, 9]),
        _capi.T_LIDAR: (5, [7, 64, 3, 50, 4]), _capi.T_EXPERT_TRAJECTORY: (3, [7, 64, 1456]),
        _capi.T_SHAPE: (2, [7, 2]), _capi.T_DELETED_AGENTS: (2, [7, 64]), _capi.T_INFO: (3, [7, 64, 5]),
    }
    for slot, (nd, dims) in exp.items():
        assert L.gd_tensor_shape(slot, 7, 64, C.byref(d)) == 0
        assert d.ndim == nd and [d.dims[i] for i in range(nd)] == dims
        assert d.nbytes == 4 * int(np.prod(dims))
    assert L.gd_tensor_shape(99, 7, 64, C.byref(d)) == _capi.GD_ERR_INVALID


def test_errors_do_not_abort(L):
    p = _capi.GdParams()
    L.gd_default_params(C.byref(p))
    hw = _capi.GdHostWorld()
    assert L.gd_host_world_build(b"/nonexistent/scene.json", C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_IO
    assert b"cannot open" in L.gd_last_error()
    bad = os.path.join(ROOT, "tests", "golden", "_bad_scene.json")
    os.makedirs(os.path.dirname(bad), exist_ok=True)
    with open(bad, "w") as fh:
        fh.write('{"name": "x", "objects": [')
    try:
        assert L.gd_host_world_build(bad.encode(), C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_PARSE
    finally:
        os.remove(bad)


def _host_world(L, scene, A, deleted=(), **kw):
    p = _capi.GdParams()
    L.gd_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    hw = _capi.GdHostWorld()
    dl = np.asarray(list(deleted) or [0], np.int32)
    rc = L.gd_host_world_build(scene.encode(), C.byref(p), A, dl.ctypes.data_as(C.POINTER(C.c_int32)),
                               len(deleted), C.byref(hw))
    assert rc == 0, L.gd_last_error()
    out = dict(
        num_agents=hw.num_agents, num_roads=hw.num_roads, mean=np.array(list(hw.mean), np.float32),
        map_name=np.array(list(hw.map_name), np.int32), scenario_id=np.array(list(hw.scenario_id), np.int32),
        map_obs=np.ctypeslib.as_array(hw.map_obs, (10000, 9)).copy(),
        trajectory=np.ctypeslib.as_array(hw.trajectory, (A, 1456)).copy(),
        controlled=np.ctypeslib.as_array(hw.controlled, (A,)).copy(),
        response_type=np.ctypeslib.as_array(hw.response_type, (A,)).copy(),
        agent_id=np.ctypeslib.as_array(hw.agent_id, (A,)).copy(),
        entity_type=np.ctypeslib.as_array(hw.entity_type, (A,)).copy(),
        metadata=np.ctypeslib.as_array(hw.metadata, (A, 4)).copy(),
        vehicle_size=np.ctypeslib.as_array(hw.vehicle_size, (A, 3)).copy(),
    )
    L.gd_host_world_free(C.byref(hw))
    return out


CASES = [
    (TEST_JSON, 128, dict(polylineReductionThreshold=0.0, observationRadius=100.0, initOnlyValidAgentsAtFirstStep=0)),
    (TEST_JSON, 64, dict(polylineReductionThreshold=0.5, maxNumControlledAgents=2, IgnoreNonVehicles=1, dynamicsModel=2)),
    (TEST_JSON, 64, dict(polylineReductionThreshold=0.5, maxNumControlledAgents=2, IgnoreNonVehicles=1, dynamicsModel=1)),
    (SCENE_407, 64, dict(polylineReductionThreshold=0.1, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)),
    (SCENE_4, 64, dict(polylineReductionThreshold=0.1, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)),
    (SCENE_4, 128, dict(polylineReductionThreshold=1.0, readFromTracksToPredict=1, dynamicsModel=3)),
    (SCENE_4, 64, dict(polylineReductionThreshold=0.1, maxNumControlledAgents=5)),
]


@pytest.mark.parametrize("scene,A,kw", CASES)
def test_host_world_matches_oracle_bit_for_bit(L, oracle_mod, scene, A, kw):
    O = oracle_mod
    hw = _host_world(L, scene, A, **kw)
    sim = O.OracleSim([scene], O.default_params(**kw), max_agents=A)
    assert hw["num_agents"] == sim.shape_tensor()[0, 0]
    assert hw["num_roads"] == sim.shape_tensor()[0, 1]
    n = hw["num_agents"]
    assert np.array_equal(hw["mean"].view(np.uint32), sim.world_means_tensor()[0].view(np.uint32))
    assert np.array_equal(hw["map_obs"].view(np.uint32), sim.map_observation_tensor()[0].view(np.uint32))
    assert np.array_equal(hw["trajectory"].view(np.uint32), sim.expert_trajectory_tensor()[0].view(np.uint32))
    assert np.array_equal(hw["controlled"], sim.controlled_state_tensor()[0, :, 0])
    assert np.array_equal(hw["response_type"], sim.response_type_tensor()[0, :, 0])
    assert np.array_equal(hw["agent_id"], sim.agent_id_tensor()[0])
    assert np.array_equal(hw["metadata"], sim.metadata_tensor()[0])
    assert np.array_equal(hw["entity_type"][:n], sim.info_tensor()[0, :n, 4])
    assert np.array_equal(hw["vehicle_size"][:n], sim.self_observation_tensor()[0, :n, 1:4])
    assert np.array_equal(hw["map_name"], sim.map_name_tensor()[0])
    assert np.array_equal(hw["scenario_id"], sim.scenario_id_tensor()[0])


def test_host_world_deleted_agents(L, oracle_mod):
    O = oracle_mod
    kw = dict(polylineReductionThreshold=0.1, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)
    base = _host_world(L, SCENE_4, 64, **kw)
    victims = [int(base["agent_id"][0]), int(base["agent_id"][3])]
    hw = _host_world(L, SCENE_4, 64, deleted=victims, **kw)
    sim = O.OracleSim([SCENE_4], O.default_params(**kw), max_agents=64)
    sim.deleteAgents({0: victims})
    assert not set(victims) & set(hw["agent_id"].tolist())
    assert np.array_equal(hw["agent_id"], sim.agent_id_tensor()[0])
    assert np.array_equal(hw["controlled"], sim.controlled_state_tensor()[0, :, 0])
    assert np.array_equal(hw["trajectory"].view(np.uint32), sim.expert_trajectory_tensor()[0].view(np.uint32))


# ---- binary scene cache ----
@pytest.mark.parametrize("scene,A,kw", CASES)
def test_scene_cache_worlds_are_bit_identical(L, tmp_path, scene, A, kw):
    """A world built from the .gdsm cache equals the world built from the JSON, byte for byte, for every
    parameter combination above and with deleted agents (deleteAgents re-builds from the same map)."""
    from gpudrive_lab_amd import scene_cache
    (cached,) = scene_cache.build_cache([scene], kw["polylineReductionThreshold"], out_dir=str(tmp_path))
    assert cached.endswith(".gdsm") and os.path.getsize(cached) < os.path.getsize(scene)
    for deleted in ((), (3, 7)):
        a = _host_world(L, scene, A, deleted=deleted, **kw)
        b = _host_world(L, cached, A, deleted=deleted, **kw)
        for k in a:
            x, y = np.asarray(a[k]), np.asarray(b[k])
            assert x.shape == y.shape and x.tobytes() == y.tobytes(), k


def test_scene_cache_guards(L, tmp_path):
    from gpudrive_lab_amd import scene_cache
    (cached,) = scene_cache.build_cache([SCENE_4], 0.1, out_dir=str(tmp_path))
    p = _capi.GdParams()
    L.gd_default_params(C.byref(p))
    hw = _capi.GdHostWorld()
    p.polylineReductionThreshold = 0.5  # another threshold than the one baked into the file
    assert L.gd_host_world_build(cached.encode(), C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_INVALID
    assert b"polylineReductionThreshold" in L.gd_last_error()
    p.polylineReductionThreshold = 0.1
    blob = open(cached, "rb").read()
    trunc = str(tmp_path / "trunc.gdsm")
    open(trunc, "wb").write(blob[:len(blob) // 2])
    assert L.gd_host_world_build(trunc.encode(), C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_PARSE
    junk = str(tmp_path / "junk.gdsm")
    open(junk, "wb").write(b"\0" * 4096)
    assert L.gd_host_world_build(junk.encode(), C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_PARSE
    assert L.gd_host_world_build(str(tmp_path / "missing.gdsm").encode(), C.byref(p), 64, None, 0, C.byref(hw)) == _capi.GD_ERR_IO
    assert L.gd_scene_cache_write(SCENE_4.encode(), 0.1, str(tmp_path / "x.bin").encode()) == _capi.GD_ERR_INVALID
    # a cache can be re-written from a cache, and build_cache reuses fresh files
    again = str(tmp_path / "again.gdsm")
    assert L.gd_scene_cache_write(cached.encode(), 0.1, again.encode()) == 0
    assert open(again, "rb").read() == blob
    t0 = os.path.getmtime(cached)
    assert scene_cache.build_cache([SCENE_4, SCENE_4], 0.1, out_dir=str(tmp_path)) == [cached, cached]
    assert os.path.getmtime(cached) == t0


def test_scene_cache_is_much_faster_than_json(L, tmp_path):
    import time
    from gpudrive_lab_amd import scene_cache
    (cached,) = scene_cache.build_cache([SCENE_4], 0.1, out_dir=str(tmp_path))
    def best(path):
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            _host_world(L, path, 64, polylineReductionThreshold=0.1)
            ts.append(time.perf_counter() - t)
        return min(ts)
    tj, tc = best(SCENE_4), best(cached)
    print("host world build: json %.2f ms, cache %.2f ms" % (tj * 1e3, tc * 1e3))
    assert tc < tj / 2
