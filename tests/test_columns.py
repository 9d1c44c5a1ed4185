"""Column contract of the exported tensors (gpudrive_lab_amd/columns.py).

(1) against golden vectors decoded by the reference's own Python classes (tests/golden/make_columns_golden.py);
(2) cross-tensor invariants that only hold if the columns mean what the reference says they mean: the
    partner / road rows of an ego must be the ego-frame image of the other entity's ABSOLUTE row.
The invariant check is shared with the GPU suite (tests/test_gpu_parity.py runs it on the HIP path)."""
import os

import numpy as np

from gpudrive_lab_amd import columns as COL
from tests.conftest import ROOT, SCENE_4, TEST_JSON

AGENT_SCALE = np.float32(0.7)  # madrona_gpudrive.vehicleScale


def test_columns_match_reference_decoders():
    g = np.load(os.path.join(ROOT, "tests", "golden", "columns_golden.npz"))
    raws = dict(SELF_OBS="raw_self_obs", ABS_OBS="raw_abs_obs", PARTNER_OBS="raw_partner", ROAD_ROW_local="raw_roadmap",
                ROAD_ROW_global="raw_map_obs", INFO="raw_info", METADATA="raw_metadata")
    checked = 0
    for key in g.files:
        if key.startswith("raw_") or key.startswith("RESPONSE_TYPE"):
            continue
        table, name = key.split(".")
        cmap = getattr(COL, table.split("_local")[0].split("_global")[0])
        raw = g[raws[table]]
        sel = raw[..., cmap[name]]
        if name in COL.SCALED_BY_AGENT_SCALE.get(table, ()):
            sel = sel * AGENT_SCALE
        if table == "INFO" and name == "collided":
            sel = sel.sum(-1)
        exp = g[key]
        if table == "PARTNER_OBS":
            exp = exp[..., 0]  # the reference unsqueezes every partner field
        if table == "ROAD_ROW_local" and name == "type":
            sel = sel.astype(np.int64)
        assert sel.shape == exp.shape and np.array_equal(sel, exp), key
        checked += 1
    assert checked == 52
    resp = g["raw_response"][..., 0]
    for name, val in COL.RESPONSE_TYPE.items():
        assert np.array_equal(g["RESPONSE_TYPE." + name], resp == val)


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def check_cross_tensor_invariants(sim, as_numpy=np.array, atol=2e-4):
    """Partner and road rows are the ego-frame image of the absolute rows; self rows agree with absolute rows."""
    ab = as_numpy(sim.absolute_self_observation_tensor()).astype(np.float64)
    so = as_numpy(sim.self_observation_tensor()).astype(np.float64)
    po = as_numpy(sim.partner_observations_tensor()).astype(np.float64)
    rm = as_numpy(sim.agent_roadmap_tensor()).astype(np.float64)
    mo = as_numpy(sim.map_observation_tensor()).astype(np.float64)
    shape = as_numpy(sim.shape_tensor())
    A, C, P, R = COL.ABS_OBS, COL.SELF_OBS, COL.PARTNER_OBS, COL.ROAD_ROW
    n_pairs = n_roads = 0
    for w in range(ab.shape[0]):
        n, nr = int(shape[w, 0]), int(shape[w, 1])
        pos = ab[w, :n][:, [A["pos_x"], A["pos_y"]]]
        yaw = ab[w, :n, A["rotation_angle"]]
        q = ab[w, :n, A["rotation_as_quaternion"]]
        assert np.allclose(2 * np.arctan2(q[:, 3], q[:, 0]), yaw, atol=1e-5) or np.allclose(wrap(2 * np.arctan2(q[:, 3], q[:, 0]) - yaw), 0, atol=1e-5)
        c, s = np.cos(yaw), np.sin(yaw)
        to_ego = lambda i, v: np.stack([c[i] * v[..., 0] + s[i] * v[..., 1], -s[i] * v[..., 0] + c[i] * v[..., 1]], -1)
        # self rows: goal in the ego frame, sizes and ids shared with the absolute row
        goal = ab[w, :n][:, [A["goal_x"], A["goal_y"]]]
        for i in range(n):
            assert np.allclose(to_ego(i, goal[i] - pos[i]), so[w, i, [C["rel_goal_x"], C["rel_goal_y"]]], atol=atol)
        assert np.array_equal(so[w, :n, C["id"]], ab[w, :n, A["id"]])
        assert np.allclose(so[w, :n, C["vehicle_length"]], ab[w, :n, A["vehicle_length"]])
        ids = ab[w, :n, A["id"]]
        for i in range(n):
            others = [j for j in range(n) if j != i]  # OtherAgents order: agent index order skipping self
            for k, j in enumerate(others):
                row = po[w, i, k]
                if row[P["ids"]] < 0:
                    continue  # beyond the observation radius: zero() row with id -1
                assert row[P["ids"]] == ids[j]
                assert np.allclose(to_ego(i, pos[j] - pos[i]), row[[P["rel_pos_x"], P["rel_pos_y"]]], atol=atol)
                assert abs(wrap(row[P["orientation"]] - (yaw[j] - yaw[i]))) < 1e-4
                assert np.isclose(row[P["vehicle_length"]], ab[w, j, A["vehicle_length"]]) and np.isclose(row[P["speed"]], so[w, j, C["speed"]], atol=1e-5)
                n_pairs += 1
            assert (po[w, i, n - 1:, P["ids"]] == -2).all()  # zero_nonexist() rows
            # road rows: matched to the global rows through (id, type, length) and the ego-frame position
            gl = mo[w, :nr]
            for row in rm[w, i]:
                if row[R["type"]] == 0 and row[R["id"]] <= 0 and row[R["segment_length"]] == 0:
                    continue  # padding
                cand = gl[(gl[:, R["id"]] == row[R["id"]]) & (gl[:, R["type"]] == row[R["type"]])]
                img = to_ego(i, cand[:, [R["x"], R["y"]]] - pos[i])
                dist = np.abs(img - row[[R["x"], R["y"]]]).max(-1)
                m = int(np.argmin(dist))
                assert dist[m] < atol, (w, i, row)
                assert np.isclose(cand[m, R["segment_length"]], row[R["segment_length"]])
                assert abs(wrap(row[R["orientation"]] - (cand[m, R["orientation"]] - yaw[i]))) < 1e-4
                n_roads += 1
    return n_pairs, n_roads


def check_road_selection_by_brute_force(sim, radius, as_numpy=np.array, K=200, margin=1e-3):
    """k-NN mode, independent of the oracle: the non-padding road rows of an agent are exactly the global roads
    that are among its K nearest AND within the radius (roads within `margin` of either boundary may go both ways)."""
    ab = as_numpy(sim.absolute_self_observation_tensor()).astype(np.float64)
    rm = as_numpy(sim.agent_roadmap_tensor()).astype(np.float64)
    mo = as_numpy(sim.map_observation_tensor()).astype(np.float64)
    shape = as_numpy(sim.shape_tensor())
    A, R = COL.ABS_OBS, COL.ROAD_ROW
    checked = 0
    for w in range(ab.shape[0]):
        n, nr = int(shape[w, 0]), int(shape[w, 1])
        gl = mo[w, :nr]
        for i in range(n):
            pos = ab[w, i, [A["pos_x"], A["pos_y"]]]
            yaw = ab[w, i, A["rotation_angle"]]
            d = np.hypot(gl[:, R["x"]] - pos[0], gl[:, R["y"]] - pos[1])
            kth = np.sort(d)[min(K, nr) - 1] if nr else 0.0
            must = (d < min(kth, radius) - margin)                      # certainly selected
            may = (d <= min(kth, radius) + margin)                      # possibly selected
            rows = rm[w, i]
            rows = rows[~((rows[:, R["type"]] == 0) & (rows[:, R["segment_length"]] == 0) & (rows[:, R["id"]] <= 0))]
            # map every row back to a global road through its ego-frame position
            c, s_ = np.cos(yaw), np.sin(yaw)
            gx = pos[0] + c * rows[:, R["x"]] - s_ * rows[:, R["y"]]
            gy = pos[1] + s_ * rows[:, R["x"]] + c * rows[:, R["y"]]
            hit = np.zeros(nr, int)
            for x, y, t in zip(gx, gy, rows[:, R["type"]]):
                m = int(np.argmin(np.hypot(gl[:, R["x"]] - x, gl[:, R["y"]] - y) + 1e3 * (gl[:, R["type"]] != t)))
                assert np.hypot(gl[m, R["x"]] - x, gl[m, R["y"]] - y) < 1e-3
                hit[m] += 1
            assert len(rows) <= K
            assert (hit[must] >= 1).all(), (w, i, "a road that must be selected is missing")
            assert (hit[~may] == 0).all(), (w, i, "a road outside the K nearest / the radius was selected")
            checked += 1
    return checked


def check_linear_selection_by_brute_force(sim, radius, as_numpy=np.array, K=200, margin=1e-3):
    """Linear mode (AllEntitiesWithRadiusFiltering, reference src/sim.cpp:258-279), independent of the oracle: an agent's
    non-padding rows are, IN THIS ORDER, the first K global roads in index order that lie within the radius -- computed here in
    float64 from the absolute poses and the global road rows (roads within `margin` of the radius may go either way, so the
    expected sequence is built from the roads the kernel chose among those); every row must be that road's ego-frame image;
    the rows behind them are MapObservation::zero() (id -1, mapType -1)."""
    ab = as_numpy(sim.absolute_self_observation_tensor()).astype(np.float64)
    rm = as_numpy(sim.agent_roadmap_tensor()).astype(np.float64)
    mo = as_numpy(sim.map_observation_tensor()).astype(np.float64)
    shape = as_numpy(sim.shape_tensor())
    A, R = COL.ABS_OBS, COL.ROAD_ROW
    checked = full = 0
    for w in range(ab.shape[0]):
        n, nr = int(shape[w, 0]), int(shape[w, 1])
        gl = mo[w, :nr]
        for i in range(n):
            pos = ab[w, i, [A["pos_x"], A["pos_y"]]]
            yaw = ab[w, i, A["rotation_angle"]]
            d = np.hypot(gl[:, R["x"]] - pos[0], gl[:, R["y"]] - pos[1])
            sure, maybe = d < radius - margin, d <= radius + margin
            rows = rm[w, i]
            live = ~((rows[:, R["type"]] == 0) & (rows[:, R["id"]] == -1))
            k = int(live.sum())
            assert live[:k].all() and not live[k:].any(), (w, i, "padding rows must follow the road rows")
            assert (rows[k:, :6] == 0).all() and (rows[k:, R["id"]] == -1).all() and (rows[k:, R["vbd_type"]] == -1).all(), (w, i)
            # the kernel's rows -> global road indices, through the ego-frame position and the type, in ascending order
            c, s_ = np.cos(yaw), np.sin(yaw)
            gx = pos[0] + c * rows[:k, R["x"]] - s_ * rows[:k, R["y"]]
            gy = pos[1] + s_ * rows[:k, R["x"]] + c * rows[:k, R["y"]]
            chosen, last = [], -1
            for x, y, t in zip(gx, gy, rows[:k, R["type"]]):
                cand = np.where((np.hypot(gl[:, R["x"]] - x, gl[:, R["y"]] - y) < 1e-3) & (gl[:, R["type"]] == t))[0]
                cand = cand[cand > last]   # (duplicated points: the next one in index order)
                assert len(cand), (w, i, "a row is no road of this world, or the rows are not in index order")
                last = int(cand[0])
                chosen.append(last)
            chosen = np.array(chosen, int)
            assert maybe[chosen].all(), (w, i, "a road beyond the radius was selected")
            # every road that is certainly in reach and comes before the last selected one (or anywhere, if fewer than K were
            # selected) must be among the selected
            limit = chosen[-1] if k == K else nr
            missing = [r for r in np.where(sure)[0] if r <= limit and r not in set(chosen.tolist())]
            assert not missing, (w, i, "roads in reach were skipped", missing[:5])
            assert k <= K
            checked += 1
            full += k == K
    return checked, full


def check_partner_rows_by_brute_force(sim, radius, as_numpy=np.array, margin=1e-3, atol=2e-4):
    """collectPartnerObsSystem (reference src/sim.cpp:188-240) recomputed in float64 numpy from the ABSOLUTE rows,
    independently of the oracle: slot k of ego i is agent j = k-th other agent in index order; within the radius
    the row is {j's speed, ego-frame position, relative heading, j's size, j's type, j's id}, beyond it the
    all-zero row with id -1 (a pair within `margin` of the radius may go either way), and slots >= n - 1 are the
    all-zero row with id -2."""
    ab = as_numpy(sim.absolute_self_observation_tensor()).astype(np.float64)
    so = as_numpy(sim.self_observation_tensor()).astype(np.float64)
    po = as_numpy(sim.partner_observations_tensor()).astype(np.float64)
    info = as_numpy(sim.info_tensor())
    shape = as_numpy(sim.shape_tensor())
    A, C, P = COL.ABS_OBS, COL.SELF_OBS, COL.PARTNER_OBS
    zero_row = np.zeros(9)
    n_in = n_out = 0
    for w in range(ab.shape[0]):
        n = int(shape[w, 0])
        pos = ab[w, :n][:, [A["pos_x"], A["pos_y"]]]
        yaw = ab[w, :n, A["rotation_angle"]]
        size = ab[w, :n][:, [A["vehicle_length"], A["vehicle_width"], A["vehicle_height"]]]
        for i in range(n):
            c, s_ = np.cos(yaw[i]), np.sin(yaw[i])
            others = [j for j in range(n) if j != i]
            for k, j in enumerate(others):
                d = pos[j] - pos[i]
                rel = np.array([c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1]])
                dist = np.hypot(*rel)
                row = po[w, i, k]
                if row[P["ids"]] == -1:  # zero(): claimed to be beyond the radius
                    assert dist > radius - margin, (w, i, j, dist)
                    z = zero_row.copy(); z[P["ids"]] = -1
                    assert np.array_equal(row, z), (w, i, j, row)
                    n_out += 1
                    continue
                assert dist <= radius + margin, (w, i, j, dist)
                exp = np.zeros(9)
                exp[P["speed"]] = so[w, j, C["speed"]]
                exp[[P["rel_pos_x"], P["rel_pos_y"]]] = rel
                exp[P["orientation"]] = row[P["orientation"]]  # compared modulo 2 pi below
                exp[[P["vehicle_length"], P["vehicle_width"], P["vehicle_height"]]] = size[j]
                exp[P["agent_type"]] = info[w, j, 4]
                exp[P["ids"]] = ab[w, j, A["id"]]
                assert np.allclose(row, exp, atol=atol, rtol=0), (w, i, j, row, exp)
                assert abs(wrap(row[P["orientation"]] - (yaw[j] - yaw[i]))) < 1e-4
                n_in += 1
            tail = po[w, i, max(n - 1, 0):]
            z = zero_row.copy(); z[P["ids"]] = -2
            assert (tail == z).all(), (w, i, "rows beyond the last other agent must be zero_nonexist()")
    return n_in, n_out


def test_cross_tensor_invariants_on_the_oracle(oracle_mod):
    O = oracle_mod
    p = O.default_params(polylineReductionThreshold=0.1, observationRadius=40.0, collisionBehaviour=2, rewardType=1,
                         distanceToGoalThreshold=2.0, isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)
    sim = O.OracleSim([TEST_JSON, SCENE_4], p, max_agents=64)
    rng = np.random.default_rng(9)
    for _ in range(5):
        act = sim.action_tensor()
        act[..., 0] = rng.uniform(-1, 2, act.shape[:2]); act[..., 1] = rng.uniform(-0.5, 0.5, act.shape[:2])
        sim.step()
    n_pairs, n_roads = check_cross_tensor_invariants(sim)
    assert n_pairs > 500 and n_roads > 1500
    assert check_road_selection_by_brute_force(sim, 40.0) == 25 + 64
    n_in, n_out = check_partner_rows_by_brute_force(sim, 40.0)
    assert n_in > 500 and n_out > 500


def test_road_selection_by_brute_force_when_K_binds(oracle_mod):
    """Unreduced polylines (9,899 roads) and a 100 m radius: far more than K roads are in radius, so the K-th
    distance, not the radius, decides."""
    O = oracle_mod
    p = O.default_params(polylineReductionThreshold=0.0, observationRadius=100.0, collisionBehaviour=2,
                         initOnlyValidAgentsAtFirstStep=0)
    sim = O.OracleSim([TEST_JSON], p, max_agents=64)
    assert int(sim.shape_tensor()[0, 1]) == 9899
    assert check_road_selection_by_brute_force(sim, 100.0) == 25


def test_linear_selection_by_brute_force_on_the_oracle(oracle_mod):
    """Linear mode pinned without the oracle's own loop: float64 brute force over the global rows (and the same on the HIP
    path, tests/test_gpu_round5.py).  Unreduced polylines and a 60 m radius: K binds for many agents (the early exit)."""
    O = oracle_mod
    p = O.default_params(polylineReductionThreshold=0.0, observationRadius=60.0, collisionBehaviour=2, roadObservationAlgorithm=1,
                         isStaticAgentControlled=1, initOnlyValidAgentsAtFirstStep=0)
    sim = O.OracleSim([TEST_JSON, SCENE_4], p, max_agents=64)
    rng = np.random.default_rng(4)
    for _ in range(3):
        act = sim.action_tensor()
        act[..., 0] = rng.uniform(-1, 2, act.shape[:2]); act[..., 1] = rng.uniform(-0.5, 0.5, act.shape[:2])
        sim.step()
    checked, full = check_linear_selection_by_brute_force(sim, 60.0)
    assert checked == 25 + 64 and full > 10
