// Binary scene cache.
//
// The reference re-parses W x 1-4 MB of scene JSON on one host thread every time maps are
// (re)sampled (MapReader::parseAndWriteOut, reference src/MapReader.cpp:46-61 via
// src/mgr.cpp:630-647); parsing is ~99 % of the host-side world build here (8-9 ms of JSON DOM per
// scene against 0.1 ms for createPersistentEntities' restatement).  A ".gdsm" file holds the parsed,
// polyline-reduced SceneMap (what from_json(Map) leaves behind, src/json_serialization.hpp) as one
// flat little-endian blob that is mmap-ed and copied: header | SceneObject[n] | road table | points.
// The only parse-time parameter is polylineReductionThreshold, which is stored and checked; every
// other Parameters field still takes effect at world-build time, and deleteAgents keeps working.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <type_traits>

#include "gd_math.hpp"
#include "scene.hpp"

namespace gd {

namespace {

struct CacheHeader {
    char magic[4];  // "GDSM"
    uint32_t version;
    uint32_t header_bytes, object_bytes;  // sizeof(CacheHeader), sizeof(SceneObject): layout guards
    float threshold;
    uint32_t n_objects, n_roads;
    uint64_t n_points;  // sum over roads
    float mean[2];
    char name[32], scenario_id[32];
    uint8_t pad[16];
};
static_assert(sizeof(CacheHeader) == 128, "header is 128 bytes");
static_assert(std::is_trivially_copyable<SceneObject>::value, "SceneObject is written as raw bytes");

struct RoadEntry {
    int32_t type;
    uint32_t id;
    int32_t map_type;
    uint32_t n_points;
};

constexpr uint32_t kVersion = 1;

}  // namespace

bool is_scene_cache_path(const std::string &path) {
    return path.size() > 5 && path.compare(path.size() - 5, 5, ".gdsm") == 0;
}

void write_scene_cache(const SceneMap &map, float threshold, const std::string &out_path) {
    CacheHeader h{};
    std::memcpy(h.magic, "GDSM", 4);
    h.version = kVersion;
    h.header_bytes = sizeof(CacheHeader);
    h.object_bytes = sizeof(SceneObject);
    h.threshold = threshold;
    h.n_objects = static_cast<uint32_t>(map.objects.size());
    h.n_roads = static_cast<uint32_t>(map.roads.size());
    for (const SceneRoad &r : map.roads) h.n_points += static_cast<uint64_t>(r.num_points());
    h.mean[0] = map.mean[0]; h.mean[1] = map.mean[1];
    std::memcpy(h.name, map.name, 32);
    std::memcpy(h.scenario_id, map.scenario_id, 32);
    const std::string tmp = out_path + ".tmp" + std::to_string(static_cast<long>(getpid()));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::invalid_argument("cannot open scene cache for writing: " + out_path);
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
    if (ok && h.n_objects) ok = std::fwrite(map.objects.data(), sizeof(SceneObject), h.n_objects, f) == h.n_objects;
    for (const SceneRoad &r : map.roads) {
        const RoadEntry e{r.type, r.id, r.map_type, static_cast<uint32_t>(r.num_points())};
        ok = ok && std::fwrite(&e, sizeof(e), 1, f) == 1;
    }
    for (const SceneRoad &r : map.roads)
        if (!r.pts.empty()) ok = ok && std::fwrite(r.pts.data(), sizeof(float), r.pts.size(), f) == r.pts.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), out_path.c_str()) != 0) {
        std::remove(tmp.c_str());
        throw std::runtime_error("failed to write scene cache: " + out_path);
    }
}

std::shared_ptr<const SceneMap> read_scene_cache(const std::string &path, float threshold) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::invalid_argument("cannot open scene file: " + path);
    struct stat sb;
    if (::fstat(fd, &sb) != 0 || static_cast<size_t>(sb.st_size) < sizeof(CacheHeader)) {
        ::close(fd);
        throw std::runtime_error("scene cache is truncated: " + path);
    }
    const size_t size = static_cast<size_t>(sb.st_size);
    void *mem = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (mem == MAP_FAILED) throw std::runtime_error("cannot map scene cache: " + path);
    struct Unmap {
        void *p; size_t n;
        ~Unmap() { ::munmap(p, n); }
    } guard{mem, size};
    const uint8_t *base = static_cast<const uint8_t *>(mem);
    CacheHeader h;
    std::memcpy(&h, base, sizeof(h));
    if (std::memcmp(h.magic, "GDSM", 4) != 0 || h.version != kVersion || h.header_bytes != sizeof(CacheHeader) ||
        h.object_bytes != sizeof(SceneObject))
        throw std::runtime_error("not a scene cache of this build (magic / version / layout): " + path);
    if (std::memcmp(&h.threshold, &threshold, sizeof(float)) != 0)
        throw std::invalid_argument("scene cache " + path + " was built with polylineReductionThreshold " +
                                std::to_string(h.threshold) + ", the simulator asks for " + std::to_string(threshold));
    if (h.n_objects > static_cast<uint32_t>(kMaxObjects) || h.n_roads > static_cast<uint32_t>(kMaxRoads))
        throw std::runtime_error("scene cache has impossible counts: " + path);
    const size_t off_roads = sizeof(CacheHeader) + static_cast<size_t>(h.n_objects) * sizeof(SceneObject);
    const size_t off_pts = off_roads + static_cast<size_t>(h.n_roads) * sizeof(RoadEntry);
    // bound n_points by what the file can hold BEFORE multiplying (a huge value must not wrap the size check)
    if (off_pts > size || h.n_points > (size - off_pts) / (2 * sizeof(float)) || off_pts + h.n_points * 2 * sizeof(float) != size)
        throw std::runtime_error("scene cache is truncated: " + path);
    auto map = std::make_shared<SceneMap>();
    map->mean[0] = h.mean[0]; map->mean[1] = h.mean[1];
    std::memcpy(map->name, h.name, 32);
    std::memcpy(map->scenario_id, h.scenario_id, 32);
    map->objects.resize(h.n_objects);
    if (h.n_objects) std::memcpy(map->objects.data(), base + sizeof(CacheHeader), static_cast<size_t>(h.n_objects) * sizeof(SceneObject));
    // the object blobs are raw structs: refuse values the world builder would index arrays with
    for (const SceneObject &o : map->objects) {
        const bool type_ok = o.type == ET_None || o.type == ET_Vehicle || o.type == ET_Pedestrian || o.type == ET_Cyclist;
        if (o.num_positions < 0 || o.num_positions > kMaxPositions || !type_ok)
            throw std::runtime_error("scene cache holds a corrupt object record: " + path);
    }
    map->roads.resize(h.n_roads);
    const float *pts = reinterpret_cast<const float *>(base + off_pts);
    uint64_t used = 0;
    for (uint32_t i = 0; i < h.n_roads; i++) {
        RoadEntry e;
        std::memcpy(&e, base + off_roads + static_cast<size_t>(i) * sizeof(RoadEntry), sizeof(e));
        if (e.n_points > static_cast<uint32_t>(kMaxGeometry) || used + e.n_points > h.n_points)
            throw std::runtime_error("scene cache road table is inconsistent: " + path);
        SceneRoad &r = map->roads[i];
        r.type = e.type; r.id = e.id; r.map_type = e.map_type;
        r.pts.assign(pts + used * 2, pts + (used + e.n_points) * 2);
        used += e.n_points;
    }
    if (used != h.n_points) throw std::runtime_error("scene cache road table is inconsistent: " + path);
    return map;
}

std::shared_ptr<const SceneMap> load_scene(const std::string &path, float threshold) {
    return is_scene_cache_path(path) ? read_scene_cache(path, threshold) : parse_scene_file(path, threshold);
}

}  // namespace gd
