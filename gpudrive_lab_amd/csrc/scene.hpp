// Host-side scene pipeline: scene JSON -> SceneMap -> per-world init rows.
// Replaces MapReader + json_serialization.hpp + createPersistentEntities of the reference
// (src/MapReader.cpp:46-61, src/json_serialization.hpp, src/level_gen.cpp:56-185,308-465).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gpudrive_amd.h"

namespace gd {

constexpr int kMaxObjects = 515;    // reference src/init.hpp:9
constexpr int kMaxRoads = 956;      // :10
constexpr int kMaxPositions = 91;   // :11
constexpr int kMaxGeometry = 1746;  // :12
constexpr int kTrajLen = 91;
constexpr int kTrajFloats = GD_TRAJECTORY_FLOATS;
constexpr int kMapK = GD_MAP_OBS_K;
constexpr int kMaxRoadEntities = GD_MAX_ROAD_ENTITIES;

struct SceneObject {
    float pos[kMaxPositions][2];
    float vel[kMaxPositions][2];
    float heading[kMaxPositions];
    uint8_t valid[kMaxPositions];
    int num_positions;
    float length, width, height;
    float goal[2];
    int type;
    uint32_t id;
    bool mark_as_expert;
    int32_t metadata[4];  // isSdc, isObjectOfInterest, isTrackToPredict, difficulty
};

struct SceneRoad {
    std::vector<float> pts;  // x0,y0,x1,y1,...
    int type;
    uint32_t id;
    int map_type;
    int num_points() const { return static_cast<int>(pts.size() / 2); }
};

// The parsed `Map` (reference src/init.hpp:50-69), without its 14 MB fixed-size layout.
struct SceneMap {
    std::vector<SceneObject> objects;
    std::vector<SceneRoad> roads;
    float mean[2];
    char name[32];
    char scenario_id[32];
};

// Throws std::runtime_error (parse) / std::invalid_argument with "cannot open" (io).
std::shared_ptr<const SceneMap> parse_scene_file(const std::string &path, float polyline_reduction_threshold);

// Binary scene cache (scene_cache.cpp): a ".gdsm" file is the parsed, polyline-reduced SceneMap.
bool is_scene_cache_path(const std::string &path);
void write_scene_cache(const SceneMap &map, float polyline_reduction_threshold, const std::string &out_path);
// Throws std::invalid_argument when the cache was built with another threshold.
std::shared_ptr<const SceneMap> read_scene_cache(const std::string &path, float polyline_reduction_threshold);
// parse_scene_file or read_scene_cache, by extension: every scene path of the API goes through here.
std::shared_ptr<const SceneMap> load_scene(const std::string &path, float polyline_reduction_threshold);

// One collidable road box (RoadEdge / StopSign ...), precomputed on the host.
struct RoadBox {
    float cx, cy, radius, type;  // centre, bounding-circle radius, EntityType as float
    float obb[14];               // Obb: cx[4] cy[4] ax[2] ay[2] origin[2]
    float pad[2];
};
static_assert(sizeof(RoadBox) == 80, "RoadBox is five float4");

// Init-time rows of one world (what createPersistentEntities leaves behind).
struct HostWorld {
    int max_agents = 0;
    int num_agents = 0, num_roads = 0, num_controlled = 0;
    float mean[3] = {0, 0, 0};
    int32_t map_name[32], scenario_id[32];
    // per agent slot [A]
    std::vector<float> trajectory;   // [A][1456]
    std::vector<float> size;         // [A][3] length,width,height
    std::vector<float> scale;        // [A][2] collision half extents (x 0.7)
    std::vector<float> goal;         // [A][2]
    std::vector<int32_t> etype, agent_id, resp, controlled;
    std::vector<int32_t> metadata;   // [A][4]
    // roads
    std::vector<float> map_obs;      // [num_roads][9] rows of map_observation_tensor
    std::vector<float> road_xy;      // [num_roads][2]
    std::vector<float> road_aux;     // [num_roads][8] qw,qz,d0,d1,d2,type,id,mapType
    std::vector<RoadBox> boxes;      // collidable subset
    // uniform broadphase grid over the collidable boxes (stands in for Madrona's BVH,
    // reference src/sim.cpp:792-797): cell c lists every box whose bounding circle can touch an
    // agent whose centre lies in c
    float grid_ox = 0, grid_oy = 0, grid_cell = 1;
    int grid_nx = 0, grid_ny = 0;
    std::vector<int32_t> cell_off;   // [nx*ny + 1]
    std::vector<int32_t> cell_items; // local box indices
};

void build_host_world(const SceneMap &map, const gd_params &params, int max_agents,
                      const int32_t *deleted, int n_deleted, HostWorld &out);

}  // namespace gd
