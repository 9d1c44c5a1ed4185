// Host-side scene pipeline (see scene.hpp).  Citations are relative to the reference checkout.
#include "scene.hpp"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>

#include "gd_math.hpp"
#include "json_min.hpp"

namespace gd {

namespace {

int object_type(const std::string &s) {  // src/json_serialization.hpp:86-94
    if (s == "vehicle") return ET_Vehicle;
    if (s == "pedestrian") return ET_Pedestrian;
    if (s == "cyclist") return ET_Cyclist;
    return ET_None;
}

int road_type(const std::string &s) {  // src/json_serialization.hpp:113-127
    if (s == "road_edge") return ET_RoadEdge;
    if (s == "road_line") return ET_RoadLine;
    if (s == "lane") return ET_RoadLane;
    if (s == "crosswalk") return ET_CrossWalk;
    if (s == "speed_bump") return ET_SpeedBump;
    if (s == "stop_sign") return ET_StopSign;
    return ET_None;
}

// src/json_serialization.hpp:18-108
void parse_object(const JValue &j, SceneObject &o) {
    std::memset(&o, 0, sizeof(o));
    int i = 0;
    for (const auto &p : j.at("position").arr) {
        if (i >= kMaxPositions) break;
        o.pos[i][0] = p.at("x").f32();
        o.pos[i][1] = p.at("y").f32();
        ++i;
    }
    o.num_positions = i;
    o.width = j.at("width").f32();
    o.length = j.at("length").f32();
    o.height = j.at("height").f32();
    o.id = static_cast<uint32_t>(j.at("id").i64());
    i = 0;
    for (const auto &h : j.at("heading").arr) {
        if (i >= kMaxPositions) break;
        o.heading[i++] = h.f32();
    }
    i = 0;
    for (const auto &v : j.at("velocity").arr) {
        if (i >= kMaxPositions) break;
        o.vel[i][0] = v.at("x").f32();
        o.vel[i][1] = v.at("y").f32();
        ++i;
    }
    i = 0;
    for (const auto &v : j.at("valid").arr) {
        if (i >= kMaxPositions) break;
        o.valid[i++] = v.boolean() ? 1 : 0;
    }
    o.goal[0] = j.at("goalPosition").at("x").f32();
    o.goal[1] = j.at("goalPosition").at("y").f32();
    o.type = object_type(j.at("type").string());
    if (const JValue *m = j.find("mark_as_expert")) o.mark_as_expert = m->boolean();
}

// Triangle-area polyline thinning, src/json_serialization.hpp:142-204.
void reduce_polyline(std::vector<float> &pts, float threshold) {
    const int64_t n = static_cast<int64_t>(pts.size() / 2);
    std::vector<uint8_t> skip(n, 0);
    bool changed = true;
    while (changed) {
        changed = false;
        int64_t k = 0;
        while (k < n - 1) {
            int64_t k1 = k + 1;
            while (k1 < n - 1 && skip[k1]) k1++;
            if (k1 >= n - 1) break;
            int64_t k2 = k1 + 1;
            while (k2 < n && skip[k2]) k2++;
            if (k2 >= n) break;
            const float p1x = pts[2 * k], p1y = pts[2 * k + 1];
            const float p2x = pts[2 * k1], p2y = pts[2 * k1 + 1];
            const float p3x = pts[2 * k2], p3y = pts[2 * k2 + 1];
            const float cross2 = (p1x - p3x) * (p2y - p1y) - (p1x - p2x) * (p3y - p1y);
            const float area = static_cast<float>(0.5 * static_cast<double>(std::fabs(cross2)));
            if (area < threshold) {
                skip[k1] = 1;
                k = k2;
                changed = true;
            } else {
                k = k1;
            }
        }
    }
    skip[0] = 0;
    skip[n - 1] = 0;
    std::vector<float> kept;
    kept.reserve(pts.size());
    for (int64_t k = 0; k < n; k++)
        if (!skip[k]) { kept.push_back(pts[2 * k]); kept.push_back(pts[2 * k + 1]); }
    pts.swap(kept);
}

// src/json_serialization.hpp:110-244
void parse_road(const JValue &j, SceneRoad &r, float threshold) {
    r.type = road_type(j.at("type").string());
    const auto &geom = j.at("geometry").arr;
    r.pts.clear();
    r.pts.reserve(geom.size() * 2);
    for (const auto &p : geom) {
        r.pts.push_back(p.at("x").f32());
        r.pts.push_back(p.at("y").f32());
    }
    const int64_t num_segments = static_cast<int64_t>(geom.size()) - 1;
    if (num_segments >= 10 && (r.type == ET_RoadLane || r.type == ET_RoadEdge || r.type == ET_RoadLine))
        reduce_polyline(r.pts, threshold);
    // The reference stores at most MAX_GEOMETRY points yet keeps the larger count; reading the
    // excess is out of bounds there.  Clamp the count instead.
    if (r.num_points() > kMaxGeometry) r.pts.resize(static_cast<size_t>(kMaxGeometry) * 2);
    r.id = j.contains("id") ? static_cast<uint32_t>(j.at("id").i64()) : 0u;
    if (const JValue *m = j.find("map_element_id")) {
        int32_t id = static_cast<int32_t>(m->i64());
        r.map_type = (id == 4 || id >= 21 || id < -1) ? -1 : id;  // MapType::UNKNOWN
    } else {
        r.map_type = -1;
    }
}

// Float incremental mean over valid object positions then all raw road points,
// src/json_serialization.hpp:246-279.
void scene_mean(const JValue &j, float out[2]) {
    float mx = 0.f, my = 0.f;
    int64_t n = 0;
    for (const auto &obj : j.at("objects").arr) {
        const auto &valid = obj.at("valid").arr;
        size_t i = 0;
        for (const auto &pos : obj.at("position").arr) {
            const JValue &v = valid.at(i++);
            if (v.t == JValue::Bool && v.b == false) continue;
            n++;
            const float nx = pos.at("x").f32(), ny = pos.at("y").f32();
            mx += (nx - mx) / n;
            my += (ny - my) / n;
        }
    }
    for (const auto &road : j.at("roads").arr)
        for (const auto &p : road.at("geometry").arr) {
            n++;
            const float nx = p.at("x").f32(), ny = p.at("y").f32();
            mx += (nx - mx) / n;
            my += (ny - my) / n;
        }
    out[0] = mx;
    out[1] = my;
}

// src/json_serialization.hpp:281-414: SDC -> tracks_to_predict -> objects_of_interest -> rest.
void parse_map(const JValue &j, SceneMap &map, float threshold) {
    std::memset(map.name, 0, sizeof(map.name));
    std::memset(map.scenario_id, 0, sizeof(map.scenario_id));
    std::strncpy(map.name, j.at("name").string().c_str(), sizeof(map.name));
    std::strncpy(map.scenario_id, j.at("scenario_id").string().c_str(), sizeof(map.scenario_id));
    scene_mean(j, map.mean);

    const auto &objs = j.at("objects").arr;
    const size_t num_objects = std::min(objs.size(), static_cast<size_t>(kMaxObjects));
    const JValue &meta = j.at("metadata");
    const int64_t sdc_index = meta.at("sdc_track_index").i64();

    std::unordered_map<int64_t, int32_t> track_difficulty;  // first entry per index wins
    std::unordered_set<int64_t> ttp;
    for (const auto &tr : meta.at("tracks_to_predict").arr) {
        const int64_t ti = tr.at("track_index").i64();
        if (ti >= 0 && ti < static_cast<int64_t>(objs.size())) {
            ttp.insert(ti);
            track_difficulty.emplace(ti, static_cast<int32_t>(tr.at("difficulty").i64()));
        } else {
            std::fprintf(stderr, "Warning: Invalid track_index %lld in scene %s\n", static_cast<long long>(ti),
                         j.at("name").string().c_str());
        }
    }
    std::unordered_set<int64_t> ooi;
    for (const auto &v : meta.at("objects_of_interest").arr) ooi.insert(v.i64());

    map.objects.clear();
    map.objects.reserve(num_objects);
    std::unordered_set<int64_t> placed_ids;
    auto push = [&](size_t i) -> SceneObject & {
        map.objects.emplace_back();
        parse_object(objs[i], map.objects.back());
        placed_ids.insert(static_cast<int32_t>(map.objects.back().id));
        return map.objects.back();
    };

    if (sdc_index >= 0 && sdc_index < static_cast<int64_t>(objs.size())) {
        SceneObject &o = push(static_cast<size_t>(sdc_index));
        o.metadata[0] = 1;
        const int64_t sdc_id = static_cast<int32_t>(o.id);
        if (ttp.count(sdc_index)) {
            o.metadata[2] = 1;
            o.metadata[3] = track_difficulty[sdc_index];
        }
        if (ooi.count(sdc_id)) o.metadata[1] = 1;
        ttp.erase(sdc_index);
        ooi.erase(sdc_id);
    }
    for (size_t i = 0; i < objs.size() && map.objects.size() < num_objects; i++) {
        if (static_cast<int64_t>(i) == sdc_index) continue;
        if (!ttp.count(static_cast<int64_t>(i))) continue;
        SceneObject &o = push(i);
        o.metadata[2] = 1;
        o.metadata[3] = track_difficulty[static_cast<int64_t>(i)];
        const int64_t id = static_cast<int32_t>(o.id);
        if (ooi.count(id)) { o.metadata[1] = 1; ooi.erase(id); }
    }
    for (size_t i = 0; i < objs.size() && map.objects.size() < num_objects; i++) {
        if (static_cast<int64_t>(i) == sdc_index) continue;
        const int64_t id = static_cast<int32_t>(objs[i].at("id").i64());
        if (!ooi.count(id)) continue;
        push(i).metadata[1] = 1;
    }
    for (size_t i = 0; i < objs.size() && map.objects.size() < num_objects; i++) {
        if (static_cast<int64_t>(i) == sdc_index) continue;
        const int64_t id = static_cast<int32_t>(objs[i].at("id").i64());
        if (placed_ids.count(id)) continue;
        push(i);
    }

    const auto &roads = j.at("roads").arr;
    const size_t num_roads = std::min(roads.size(), static_cast<size_t>(kMaxRoads));
    map.roads.resize(num_roads);
    for (size_t i = 0; i < num_roads; i++) parse_road(roads[i], map.roads[i], threshold);
}

// ---- inverse dynamics for the expert action columns (src/dynamics.hpp:117-184) ----
void inverse_bicycle(Quat rot, const float vel[2], const float tvel[2], float *out) {
    const float dt = 0.1f;
    for (int i = 0; i < 10; i++) out[i] = 0.f;
    const float speed = len_3(vel[0], vel[1], 0.f);
    const float target_speed = len_3(tvel[0], tvel[1], 0.f);
    out[0] = (target_speed - speed) / dt;
    const float yaw = normalize_angle(quat_to_yaw(rot));
    const float target_yaw = atan2f(tvel[1], tvel[0]);  // consts::useEstimatedYaw (src/consts.hpp:15)
    const float denominator = static_cast<float>(static_cast<double>(speed * dt) + 0.5 * out[0] * dt * dt);
    out[1] = denominator != 0 ? (target_yaw - yaw) / denominator : 0.f;
}

void inverse_delta(Quat rot, const float pos[2], Quat trot, const float tpos[2], float *out) {
    for (int i = 0; i < 10; i++) out[i] = 0.f;
    const float yaw = quat_to_yaw(rot), target_yaw = quat_to_yaw(trot);
    float dx = tpos[0] - pos[0], dy = tpos[1] - pos[1];
    const float dyaw = target_yaw - yaw;
    dx = fmaxf(-6.0f, fminf(dx, 6.0f));
    dy = fmaxf(-6.0f, fminf(dy, 6.0f));
    const float c = cosf(-yaw), s = sinf(-yaw);
    const float ldx = dx * c - dy * s;
    const float ldy = dx * s + dy * c;
    out[0] = fmaxf(-6.0f, fminf(ldx, 6.0f));
    out[1] = fmaxf(-6.0f, fminf(ldy, 6.0f));
    out[2] = normalize_angle(dyaw);
}

inline float *tr_pos(float *t, int i) { return t + 2 * i; }
inline float *tr_vel(float *t, int i) { return t + 2 * kTrajLen + 2 * i; }
inline float &tr_head(float *t, int i) { return t[4 * kTrajLen + i]; }
inline float &tr_valid(float *t, int i) { return t[5 * kTrajLen + i]; }
inline float *tr_inv(float *t, int i) { return t + 6 * kTrajLen + 10 * i; }

void zero_action(int model, float *a) {  // src/level_gen.hpp:16-42
    for (int i = 0; i < 10; i++) a[i] = 0.f;
    if (model == GD_DYNAMICS_STATE) a[2] = 1.f;
}

// src/level_gen.cpp:56-100
void fill_trajectory(const SceneObject &o, const float mean[3], int model, float *t) {
    float za[10];
    zero_action(model, za);
    for (int i = 0; i < o.num_positions; i++) {
        tr_pos(t, i)[0] = o.pos[i][0] - mean[0];
        tr_pos(t, i)[1] = o.pos[i][1] - mean[1];
        tr_vel(t, i)[0] = o.vel[i][0];
        tr_vel(t, i)[1] = o.vel[i][1];
        tr_head(t, i) = o.heading[i];
        tr_valid(t, i) = o.valid[i] ? 1.f : 0.f;
        std::memcpy(tr_inv(t, i), za, sizeof(za));
    }
    if (model == GD_DYNAMICS_CLASSIC || model == GD_DYNAMICS_STATE) return;
    for (int i = o.num_positions - 2; i >= 0; i--) {
        const Quat rot = quat_yaw(tr_head(t, i));
        const Quat trot = quat_yaw(tr_head(t, i + 1));
        if (model == GD_DYNAMICS_INVERTIBLE_BICYCLE) inverse_bicycle(rot, tr_vel(t, i), tr_vel(t, i + 1), tr_inv(t, i));
        else inverse_delta(rot, tr_pos(t, i), trot, tr_pos(t, i + 1), tr_inv(t, i));
    }
}

bool should_create(const SceneObject &o, const gd_params &p, const int32_t *deleted, int n_deleted) {
    auto is_deleted = [&]() {
        for (int i = 0; i < n_deleted; i++)
            if (deleted[i] == static_cast<int32_t>(o.id)) return true;
        return false;
    };
    if (p.readFromTracksToPredict) return !is_deleted();  // src/level_gen.cpp:357-370
    if (p.IgnoreNonVehicles && (o.type == ET_Pedestrian || o.type == ET_Cyclist)) return false;
    if (p.initOnlyValidAgentsAtFirstStep && !o.valid[0]) return false;
    return !is_deleted();
}

void put_road(HostWorld &w, float x, float y, Quat rot, float d0, float d1, float d2, int type, uint32_t id,
              int map_type) {
    // setRoadEntitiesProps, src/level_gen.hpp:44-65
    const float row[9] = {x, y, d0, d1, d2, quat_to_yaw(rot), static_cast<float>(type), static_cast<float>(id),
                          static_cast<float>(map_type)};
    w.map_obs.insert(w.map_obs.end(), row, row + 9);
    w.road_xy.push_back(x);
    w.road_xy.push_back(y);
    // RoadMapId is int32; the k-NN row casts it back to float (src/knn.hpp:117)
    const float aux[8] = {rot.w, rot.z, d0, d1, d2, static_cast<float>(type),
                          static_cast<float>(static_cast<int32_t>(id)), static_cast<float>(map_type)};
    w.road_aux.insert(w.road_aux.end(), aux, aux + 8);
    if (type == ET_RoadEdge || type == ET_StopSign) {
        RoadBox b{};
        b.cx = x; b.cy = y; b.type = static_cast<float>(type);
        b.radius = sqrtf(d0 * d0 + d1 * d1);
        const Obb o = obb_from(x, y, rot, d0, d1);
        std::memcpy(b.obb, &o, sizeof(o));
        w.boxes.push_back(b);
    }
    w.num_roads++;
}

}  // namespace

std::shared_ptr<const SceneMap> parse_scene_file(const std::string &path, float threshold) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw std::invalid_argument("cannot open scene file: " + path);
    std::string buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    JParser parser(buf.c_str(), buf.c_str() + buf.size());
    JValue root = parser.parse();
    auto map = std::make_shared<SceneMap>();
    try {
        parse_map(root, *map, threshold);
    } catch (const std::out_of_range &) {
        throw std::runtime_error("scene JSON: array index out of range in " + path);
    }
    return map;
}

// createPersistentEntities (src/level_gen.cpp:396-465) for one world, on the host.
void build_host_world(const SceneMap &map, const gd_params &p, int A, const int32_t *deleted, int n_deleted,
                      HostWorld &w) {
    w = HostWorld();
    w.max_agents = A;
    for (int i = 0; i < 32; i++) {
        w.map_name[i] = static_cast<int32_t>(static_cast<char32_t>(map.name[i]));
        w.scenario_id[i] = static_cast<int32_t>(static_cast<char32_t>(map.scenario_id[i]));
    }
    w.mean[0] = map.mean[0]; w.mean[1] = map.mean[1]; w.mean[2] = 0.f;
    w.trajectory.assign(static_cast<size_t>(A) * kTrajFloats, 0.f);
    w.size.assign(static_cast<size_t>(A) * 3, 0.f);
    w.scale.assign(static_cast<size_t>(A) * 2, 0.f);
    w.goal.assign(static_cast<size_t>(A) * 2, 0.f);
    w.etype.assign(A, ET_None);
    w.agent_id.assign(A, -1);
    w.resp.assign(A, RESP_Static);
    w.controlled.assign(A, 0);
    w.metadata.assign(static_cast<size_t>(A) * 4, -1);

    int a = 0;
    for (size_t oi = 0; oi < map.objects.size() && a < A; ++oi) {
        const SceneObject &o = map.objects[oi];
        if (!should_create(o, p, deleted, n_deleted)) continue;
        // createAgent, src/level_gen.cpp:131-164
        w.size[a * 3 + 0] = o.length; w.size[a * 3 + 1] = o.width; w.size[a * 3 + 2] = o.height;
        float d0 = o.length / 2, d1 = o.width / 2;
        d0 *= GD_VEHICLE_SCALE; d1 *= GD_VEHICLE_SCALE;
        w.scale[a * 2 + 0] = d0; w.scale[a * 2 + 1] = d1;
        w.etype[a] = o.type;
        w.goal[a * 2 + 0] = o.goal[0] - w.mean[0];
        w.goal[a * 2 + 1] = o.goal[1] - w.mean[1];
        w.agent_id[a] = static_cast<int32_t>(o.id);
        float *t = w.trajectory.data() + static_cast<size_t>(a) * kTrajFloats;
        fill_trajectory(o, w.mean, p.dynamicsModel, t);
        // isAgentStatic (src/level_gen.cpp:102-113).  readFromTracksToPredict reads the interface
        // row's MetaData before it is assigned in the reference; the evident intent (the object's
        // own metadata) is implemented.
        bool is_static;
        if (p.readFromTracksToPredict && o.metadata[2] != -1) {
            is_static = false;
        } else {
            const float d = len_2(w.goal[a * 2 + 0] - tr_pos(t, 0)[0], w.goal[a * 2 + 1] - tr_pos(t, 0)[1]);
            is_static = !p.isStaticAgentControlled && d < 0.2f;  // consts::staticThreshold
        }
        w.resp[a] = is_static ? RESP_Static : RESP_Dynamic;
        // isAgentControllable (src/level_gen.cpp:115-129)
        bool ctrl;
        if (p.readFromTracksToPredict)
            ctrl = static_cast<uint32_t>(w.num_controlled) < p.maxNumControlledAgents && o.metadata[2] != -1;
        else
            ctrl = static_cast<uint32_t>(w.num_controlled) < p.maxNumControlledAgents && tr_valid(t, 0) != 0.f &&
                   w.resp[a] == RESP_Dynamic && !o.mark_as_expert;
        w.controlled[a] = ctrl ? 1 : 0;
        w.num_controlled += ctrl ? 1 : 0;
        std::memcpy(&w.metadata[a * 4], o.metadata, sizeof(o.metadata));
        a++;
    }
    w.num_agents = a;

    // createRoadEntities, src/level_gen.cpp:258-300
    for (const SceneRoad &r : map.roads) {
        if (w.num_roads >= kMaxRoadEntities) break;
        const int n = r.num_points();
        if (r.type == ET_RoadEdge || r.type == ET_RoadLine || r.type == ET_RoadLane) {
            for (int j = 0; j + 1 < n; j++) {  // makeRoadEdge, :166-185
                const float z = 1 + (r.type == ET_RoadEdge ? 0.1f : -0.1f);
                const float sx = r.pts[2 * j] - w.mean[0], sy = r.pts[2 * j + 1] - w.mean[1];
                const float ex = r.pts[2 * j + 2] - w.mean[0], ey = r.pts[2 * j + 3] - w.mean[1];
                const float px = (sx + ex) / 2, py = (sy + ey) / 2;
                const Quat rot = quat_yaw(atan2f(ey - sy, ex - sx));
                const float dist = len_3(sx - ex, sy - ey, z - z);
                put_road(w, px, py, rot, dist / 2, 0.1f, 0.1f, r.type, r.id, r.map_type);
                if (w.num_roads >= kMaxRoadEntities) break;
            }
        } else if (r.type == ET_CrossWalk || r.type == ET_SpeedBump) {  // makeCube, :191-241
            if (n < 4) continue;
            float len[4];
            for (int i = 0; i < 4; i++) {
                const int k = (i + 1) % 4;
                const double dx = static_cast<double>(r.pts[2 * k] - r.pts[2 * i]);
                const double dy = static_cast<double>(r.pts[2 * k + 1] - r.pts[2 * i + 1]);
                len[i] = static_cast<float>(std::sqrt(std::pow(dx, 2) + std::pow(dy, 2)));
            }
            int mx = 0, mn = 0;
            for (int i = 1; i < 4; i++) {
                if (len[i] > len[mx]) mx = i;
                if (len[i] < len[mn]) mn = i;
            }
            const int e = (mx + 1) % 4;
            const float angle = atan2f(r.pts[2 * e + 1] - r.pts[2 * mx + 1], r.pts[2 * e] - r.pts[2 * mx]);
            float sum_x = 0.f, sum_y = 0.f;
            for (int i = 0; i < 4; i++) { sum_x += r.pts[2 * i]; sum_y += r.pts[2 * i + 1]; }
            put_road(w, sum_x / 4 - w.mean[0], sum_y / 4 - w.mean[1], quat_yaw(angle), len[mx] / 2, len[mn] / 2, 0.1f,
                     r.type, r.id, r.map_type);
        } else if (r.type == ET_StopSign) {  // makeStopSign, :243-256
            if (n < 1) continue;
            put_road(w, r.pts[0] - w.mean[0], r.pts[1] - w.mean[1], quat_yaw(0.f), 0.2f, 0.2f, 1.f, ET_StopSign, r.id,
                     r.map_type);
        }
    }

    // Broadphase grid.  A box belongs to a cell if its centre is within (box radius + largest agent
    // radius, both with the kernels' 1.001 / +0.01 slack) of the cell's square, so the per-pair
    // bounding-circle test in the kernel never sees fewer boxes than a full scan would.
    if (!w.boxes.empty()) {
        float max_agent_r = 0.f;
        for (int i = 0; i < w.num_agents; i++)
            max_agent_r = std::max(max_agent_r, sqrtf(w.scale[i * 2] * w.scale[i * 2] + w.scale[i * 2 + 1] * w.scale[i * 2 + 1]));
        float minx = FLT_MAX, miny = FLT_MAX, maxx = -FLT_MAX, maxy = -FLT_MAX, max_r = 0.f;
        for (const RoadBox &b : w.boxes) {
            minx = std::min(minx, b.cx); maxx = std::max(maxx, b.cx);
            miny = std::min(miny, b.cy); maxy = std::max(maxy, b.cy);
            max_r = std::max(max_r, b.radius);
        }
        const float reach_pad = (max_agent_r + max_r) * 1.002f + 0.05f;
        minx -= reach_pad; miny -= reach_pad; maxx += reach_pad; maxy += reach_pad;
        const float extent = std::max(maxx - minx, maxy - miny);
        const float cell = std::max(16.f, extent / 64.f);
        w.grid_ox = minx; w.grid_oy = miny; w.grid_cell = cell;
        w.grid_nx = std::max(1, static_cast<int>(std::ceil((maxx - minx) / cell)));
        w.grid_ny = std::max(1, static_cast<int>(std::ceil((maxy - miny) / cell)));
        const int ncell = w.grid_nx * w.grid_ny;
        std::vector<std::vector<int32_t>> lists(ncell);
        for (size_t bi = 0; bi < w.boxes.size(); bi++) {
            const RoadBox &b = w.boxes[bi];
            const float reach = (max_agent_r + b.radius) * 1.002f + 0.05f;
            const int x0 = std::max(0, static_cast<int>(std::floor((b.cx - reach - minx) / cell)) - 1);
            const int x1 = std::min(w.grid_nx - 1, static_cast<int>(std::floor((b.cx + reach - minx) / cell)) + 1);
            const int y0 = std::max(0, static_cast<int>(std::floor((b.cy - reach - miny) / cell)) - 1);
            const int y1 = std::min(w.grid_ny - 1, static_cast<int>(std::floor((b.cy + reach - miny) / cell)) + 1);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    const float qx0 = minx + x * cell, qy0 = miny + y * cell;
                    const float dx = std::max(std::max(qx0 - b.cx, 0.f), b.cx - (qx0 + cell));
                    const float dy = std::max(std::max(qy0 - b.cy, 0.f), b.cy - (qy0 + cell));
                    if (dx * dx + dy * dy <= reach * reach) lists[y * w.grid_nx + x].push_back(static_cast<int32_t>(bi));
                }
        }
        w.cell_off.assign(ncell + 1, 0);
        for (int c = 0; c < ncell; c++) w.cell_off[c + 1] = w.cell_off[c] + static_cast<int32_t>(lists[c].size());
        w.cell_items.reserve(w.cell_off[ncell]);
        for (int c = 0; c < ncell; c++) w.cell_items.insert(w.cell_items.end(), lists[c].begin(), lists[c].end());
    } else {
        w.grid_nx = w.grid_ny = 0;
        w.cell_off.assign(1, 0);
    }
}

}  // namespace gd
