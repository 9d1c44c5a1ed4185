"""ORACLE (test infrastructure, NOT product code): scene JSON -> packed Map.

A numpy/float32 restatement of the reference's scene parser.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Follows (paths relative to /root/reference):
  * src/json_serialization.hpp:12-16     MapVector2 from_json (double -> float narrowing)
  * src/json_serialization.hpp:18-108    MapObject from_json (91-position cap, metadata zeroed)
  * src/json_serialization.hpp:110-244   MapRoad from_json (type names, polyline reduction, mapType)
  * src/json_serialization.hpp:246-279   calc_mean (float incremental mean, int64 count)
  * src/json_serialization.hpp:281-414   Map from_json (SDC -> tracks_to_predict -> objects_of_interest -> rest)
  * src/init.hpp:9-12                    MAX_OBJECTS/MAX_ROADS/MAX_POSITIONS/MAX_GEOMETRY

Third-party arithmetic not in /root/reference: nlohmann/json (un-vendored submodule,
version unknown).  `get<float>()` parses a JSON number as double and narrows with a
static_cast; Python's json module parses the same decimal string to the same double
and np.float32() narrows the same way (round-to-nearest-even).
"""
import json

import numpy as np

MAX_OBJECTS = 515
MAX_ROADS = 956
MAX_POSITIONS = 91
MAX_GEOMETRY = 1746

# src/types.hpp:24-38
ENTITY = dict(none=0, road_edge=1, road_line=2, lane=3, crosswalk=4, speed_bump=5,
              stop_sign=6, vehicle=7, pedestrian=8, cyclist=9, padding=10)
_OBJ_TYPES = {"vehicle": 7, "pedestrian": 8, "cyclist": 9}
_ROAD_TYPES = {"road_edge": 1, "road_line": 2, "lane": 3, "crosswalk": 4,
               "speed_bump": 5, "stop_sign": 6}

f32 = np.float32


def calc_mean(j):
    """src/json_serialization.hpp:246-279."""
    mx = f32(0)
    my = f32(0)
    n = 0
    for obj in j["objects"]:
        valid = obj["valid"]
        for i, pos in enumerate(obj["position"]):
            if valid[i] == False:  # noqa: E712  (json bool compare, as in the reference)
                continue
            n += 1
            nx = f32(pos["x"])
            ny = f32(pos["y"])
            mx = f32(mx + f32(f32(nx - mx) / f32(n)))
            my = f32(my + f32(f32(ny - my) / f32(n)))
    for road in j["roads"]:
        for p in road["geometry"]:
            n += 1
            nx = f32(p["x"])
            ny = f32(p["y"])
            mx = f32(mx + f32(f32(nx - mx) / f32(n)))
            my = f32(my + f32(f32(ny - my) / f32(n)))
    return mx, my


def _parse_object(o):
    """src/json_serialization.hpp:18-108."""
    n = min(len(o["position"]), MAX_POSITIONS)
    pos = np.zeros((MAX_POSITIONS, 2), f32)
    vel = np.zeros((MAX_POSITIONS, 2), f32)
    head = np.zeros((MAX_POSITIONS,), f32)
    valid = np.zeros((MAX_POSITIONS,), np.int32)
    for i in range(n):
        pos[i, 0] = f32(o["position"][i]["x"])
        pos[i, 1] = f32(o["position"][i]["y"])
    nh = min(len(o["heading"]), MAX_POSITIONS)
    for i in range(nh):
        head[i] = f32(o["heading"][i])
    nv = min(len(o["velocity"]), MAX_POSITIONS)
    for i in range(nv):
        vel[i, 0] = f32(o["velocity"][i]["x"])
        vel[i, 1] = f32(o["velocity"][i]["y"])
    nva = min(len(o["valid"]), MAX_POSITIONS)
    for i in range(nva):
        valid[i] = 1 if o["valid"][i] else 0
    return dict(
        pos=pos, vel=vel, head=head, valid=valid, num_positions=n,
        size=np.array([f32(o["length"]), f32(o["width"]), f32(o["height"])], f32),
        goal=np.array([f32(o["goalPosition"]["x"]), f32(o["goalPosition"]["y"])], f32),
        type=_OBJ_TYPES.get(o["type"], 0),
        id=int(o["id"]),
        mark_as_expert=1 if o.get("mark_as_expert", False) else 0,
        metadata=[0, 0, 0, 0],  # isSdc, isObjectOfInterest, isTrackToPredict, difficulty
    )


def _reduce_polyline(pts, threshold):
    """src/json_serialization.hpp:142-204 (sample_every_n_ == 1)."""
    num_segments = len(pts) - 1
    npts = num_segments + 1
    skip = [False] * npts
    changed = True
    thr = f32(threshold)
    while changed:
        changed = False
        k = 0
        while k < npts - 1:
            k1 = k + 1
            while k1 < npts - 1 and skip[k1]:
                k1 += 1
            if k1 >= npts - 1:
                break
            k2 = k1 + 1
            while k2 < npts and skip[k2]:
                k2 += 1
            if k2 >= npts:
                break
            p1, p2, p3 = pts[k], pts[k1], pts[k2]
            # float_t area = 0.5 * std::abs((p1.x-p3.x)*(p2.y-p1.y) - (p1.x-p2.x)*(p3.y-p1.y));
            a = f32(f32(f32(p1[0] - p3[0]) * f32(p2[1] - p1[1])) -
                    f32(f32(p1[0] - p2[0]) * f32(p3[1] - p1[1])))
            area = f32(0.5 * float(abs(a)))
            if area < thr:
                skip[k1] = True
                k = k2
                changed = True
            else:
                k = k1
    skip[0] = False
    skip[npts - 1] = False
    return [p for p, s in zip(pts, skip) if not s]


def _parse_road(r, threshold):
    """src/json_serialization.hpp:110-244."""
    rtype = _ROAD_TYPES.get(r["type"], 0)
    pts = [(f32(p["x"]), f32(p["y"])) for p in r["geometry"]]
    num_segments = len(pts) - 1
    if num_segments >= 10 and rtype in (1, 2, 3):
        pts = _reduce_polyline(pts, threshold)
    # The reference writes at most MAX_GEOMETRY points but keeps the unclamped count
    # (reads past the array for longer polylines); the restatement clamps the count.
    pts = pts[:MAX_GEOMETRY]
    rid = int(r["id"]) if "id" in r else 0
    if "map_element_id" in r:
        m = int(r["map_element_id"])
        if m == 4 or m >= 21 or m < -1:
            m = -1
    else:
        m = -1
    return dict(type=rtype, id=rid, map_type=m,
                geometry=np.array(pts, f32).reshape(-1, 2))


def parse_scene(path, polyline_reduction_threshold=0.0):
    """src/json_serialization.hpp:281-414.  Returns the packed Map as a dict."""
    with open(path, "r") as fh:
        j = json.load(fh)
    name = j["name"]
    scenario_id = j["scenario_id"]
    mean = calc_mean(j)
    objs_j = j["objects"]
    num_objects = min(len(objs_j), MAX_OBJECTS)
    meta = j["metadata"]
    sdc_index = int(meta["sdc_track_index"])

    ttp = {}
    for tr in meta["tracks_to_predict"]:
        ti = int(tr["track_index"])
        if 0 <= ti < len(objs_j):
            ttp.setdefault(ti, int(tr["difficulty"]))  # first match wins (`break`)
    ttp_indices = set(ttp.keys())
    ooi_ids = set(int(x) for x in meta["objects_of_interest"])

    objects = []
    seen_ids = {}
    if 0 <= sdc_index < len(objs_j):
        o = _parse_object(objs_j[sdc_index])
        o["metadata"][0] = 1
        if sdc_index in ttp_indices:
            o["metadata"][2] = 1
            o["metadata"][3] = ttp[sdc_index]
        if o["id"] in ooi_ids:
            o["metadata"][1] = 1
        seen_ids[o["id"]] = 0
        objects.append(o)
        ttp_indices.discard(sdc_index)
        ooi_ids.discard(o["id"])

    for i in range(len(objs_j)):
        if len(objects) >= num_objects:
            break
        if i == sdc_index:
            continue
        if i in ttp_indices:
            o = _parse_object(objs_j[i])
            o["metadata"][2] = 1
            o["metadata"][3] = ttp[i]
            if o["id"] in ooi_ids:
                o["metadata"][1] = 1
                ooi_ids.discard(o["id"])
            seen_ids[o["id"]] = len(objects)
            objects.append(o)

    for i in range(len(objs_j)):
        if len(objects) >= num_objects:
            break
        if i == sdc_index:
            continue
        if int(objs_j[i]["id"]) in ooi_ids:
            o = _parse_object(objs_j[i])
            o["metadata"][1] = 1
            seen_ids[o["id"]] = len(objects)
            objects.append(o)

    for i in range(len(objs_j)):
        if len(objects) >= num_objects:
            break
        if i == sdc_index:
            continue
        if int(objs_j[i]["id"]) not in seen_ids:
            o = _parse_object(objs_j[i])
            seen_ids[o["id"]] = len(objects)
            objects.append(o)

    roads = []
    for r in j["roads"][:MAX_ROADS]:
        roads.append(_parse_road(r, polyline_reduction_threshold))

    return dict(name=name, scenario_id=scenario_id, mean=np.array(mean, f32),
                objects=objects, roads=roads)


def pack_map(m):
    """Flatten the parsed Map into contiguous numpy arrays for the C oracle."""
    objs = m["objects"]
    n = len(objs)
    out = dict(
        n_obj=n,
        obj_pos=np.ascontiguousarray(np.stack([o["pos"] for o in objs]) if n else np.zeros((0, 91, 2), f32), f32),
        obj_vel=np.ascontiguousarray(np.stack([o["vel"] for o in objs]) if n else np.zeros((0, 91, 2), f32), f32),
        obj_head=np.ascontiguousarray(np.stack([o["head"] for o in objs]) if n else np.zeros((0, 91), f32), f32),
        obj_valid=np.ascontiguousarray(np.stack([o["valid"] for o in objs]) if n else np.zeros((0, 91), np.int32), np.int32),
        obj_npos=np.array([o["num_positions"] for o in objs], np.int32),
        obj_size=np.ascontiguousarray(np.stack([o["size"] for o in objs]) if n else np.zeros((0, 3), f32), f32),
        obj_goal=np.ascontiguousarray(np.stack([o["goal"] for o in objs]) if n else np.zeros((0, 2), f32), f32),
        obj_type=np.array([o["type"] for o in objs], np.int32),
        obj_id=np.array([o["id"] for o in objs], np.int32),
        obj_expert=np.array([o["mark_as_expert"] for o in objs], np.int32),
        obj_meta=np.array([o["metadata"] for o in objs], np.int32).reshape(n, 4),
    )
    roads = m["roads"]
    offs = [0]
    for r in roads:
        offs.append(offs[-1] + len(r["geometry"]))
    out.update(
        n_road=len(roads),
        road_off=np.array(offs, np.int32),
        road_pts=np.ascontiguousarray(
            np.concatenate([r["geometry"] for r in roads]) if roads else np.zeros((0, 2), f32), f32),
        road_type=np.array([r["type"] for r in roads], np.int32),
        road_id=np.array([r["id"] for r in roads], np.int32),
        road_maptype=np.array([r["map_type"] for r in roads], np.int32),
        mean=np.ascontiguousarray(m["mean"], f32),
        name=m["name"].encode("utf-8")[:32].ljust(32, b"\0"),
        scenario_id=m["scenario_id"].encode("utf-8")[:32].ljust(32, b"\0"),
    )
    return out
