// BEV rasteriser (reference src/sim.cpp:462-555 + src/rasterizer.hpp) and LiDAR (src/sim.cpp:394-460).
// One workgroup per live agent (grid = agents x worlds); both are opt-in (gd_config.alloc_bev,
// Parameters.enableLidar).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine.hpp"
#include "gd_math.hpp"

#ifdef GD_STAMPS
// diagnostic build only (tools/stamps.sh bev): per-workgroup cycle counts of the phases of k_bev
__device__ unsigned long long g_bev_stamps[8192][8];
extern "C" int gd_debug_read_bev_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bev_stamps), sizeof(unsigned long long) * 8 * n);
}
#define BEV_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define BEV_STAMP(var)
#endif

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr int RES = GD_BEV_RES;
#ifndef GD_LIDAR_GROUP
#define GD_LIDAR_GROUP 4  // agents per workgroup of k_lidar = one per wave (16 per workgroup: 350 us on the Waymo tiles, 8: 322, 4: 312)
#endif
#ifndef GD_BEV_BANDS
#define GD_BEV_BANDS 4  // bands of 50 grid rows: 40,000 bytes of cells (measured: 4 bands 2.04 ms, 5 2.16, 8 2.42)
#endif

// ------------------------------------------------------------------------------------------
// BEV.  The reference paints the first <= 200 in-radius roads in road order, then the in-radius partners in
// OtherAgents order; later paints overwrite earlier ones.  Here a cell keeps the entity with the highest
// position in that order (LDS atomic max), which is the same picture painted in any order.
// ------------------------------------------------------------------------------------------
struct BevEnt {
    float cx, cy, cosy, siny, half_l, half_w;
    unsigned int box;  // cell range to test (inclusive): x0 | x1 << 8 | y0 << 16 | y1 << 24; an empty range is stored as x0 = 1, x1 = 0
    int type;
    __device__ __forceinline__ int x0() const { return (int)(box & 0xff); }
    __device__ __forceinline__ int x1() const { return (int)((box >> 8) & 0xff); }
    __device__ __forceinline__ int y0() const { return (int)((box >> 16) & 0xff); }
    __device__ __forceinline__ int y1() const { return (int)(box >> 24); }
};
static_assert(sizeof(BevEnt) == 32 && RES <= 256, "32-byte entities, 8-bit cell coordinates");

__device__ __forceinline__ BevEnt bev_entity(float cx, float cy, float yaw, float length, float width, int type,
                                             float radius) {
    // rasterizeRotatedRectangle prologue, src/rasterizer.hpp:27-50
    BevEnt e;
    e.cx = cx; e.cy = cy;
    e.half_w = width / 2.0f;
    e.half_l = length / 2.0f;
    const float scale_px = (2 * radius) / RES;
    const float scale_m = RES / (2 * radius);
    int gx = (int)((cx + radius) * scale_m), gy = (int)((cy + radius) * scale_m);
    gx = min(max(0, gx), RES - 1);
    gy = min(max(0, gy), RES - 1);
    const float max_side = fmaxf(e.half_w, e.half_l);
    const int br = (int)ceilf(sqrtf(2 * (max_side * max_side)) / scale_px);
    e.cosy = p_cos(-yaw);
    e.siny = p_sin(-yaw);
    e.type = type;
    // The reference tests every cell of the square (gx, gy) +- br (the circumscribed circle's box around the
    // CLAMPED centre cell).  The cells that can pass the rotated-rectangle test also lie in the rectangle's
    // axis-aligned box (one cell of margin): testing the intersection paints the identical cell set.
    const float hx = e.half_l * fabsf(e.cosy) + e.half_w * fabsf(e.siny) + 2e-3f;
    const float hy = e.half_l * fabsf(e.siny) + e.half_w * fabsf(e.cosy) + 2e-3f;
    const int tx0 = (int)floorf((cx - hx + radius) * scale_m) - 1, tx1 = (int)ceilf((cx + hx + radius) * scale_m) + 1;
    const int ty0 = (int)floorf((cy - hy + radius) * scale_m) - 1, ty1 = (int)ceilf((cy + hy + radius) * scale_m) + 1;
    const int x0 = max(max(gx - br, 0), tx0), x1 = min(min(gx + br, RES - 1), tx1);
    const int y0 = max(max(gy - br, 0), ty0), y1 = min(min(gy + br, RES - 1), ty1);
    e.box = (x1 < x0 || y1 < y0) ? 1u : (unsigned int)x0 | (unsigned int)x1 << 8 | (unsigned int)y0 << 16 | (unsigned int)y1 << 24;
    return e;
}

#ifndef GD_BEV_WAVES_PER_SIMD
#define GD_BEV_WAVES_PER_SIMD 6  // three workgroups of 512 threads per CU (LDS: 48.5 KB each)
#endif
template <int A_T, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(GD_BEV_WAVES_PER_SIMD, 8))) void k_bev(DevSim d) {
    constexpr int NWV = NT / 64;  // waves
    static_assert(NT >= 128 && NT % 64 == 0, "geometry");
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    // The workgroups stride over the list of agents whose raster can have changed (k_bev_list below; every live agent on a reset
    // pass): a fixed grid of a few workgroups per CU instead of one workgroup per live agent -- a workgroup that only finds out it
    // has nothing to do still has to be handed its 48 KB of LDS and eight waves first (round 2: 0.68 ms for 30 thousand of them).
    // Rows of padding agents are never written (src/level_gen.cpp:308-336).
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const float radius = d.p.observationRadius;
    constexpr int MAXE = K + A_T;  // <= K roads and A_T - 1 partners
    __shared__ __attribute__((aligned(16))) unsigned int s_cells[RES * RES / GD_BEV_BANDS];  // one band of rows, 32-bit cells
    __shared__ BevEnt s_ent[MAXE];
    __shared__ int s_wcnt[NWV];
    __shared__ int s_ne;
    for (int c = tid; c < RES * RES / GD_BEV_BANDS; c += NT) s_cells[c] = 0u;  // (every band is zeroed again on its way out)
    __shared__ int s_item;
    const int items = d.bev_count[0];
    // (the next unclaimed item, not a stride: rasters cost between a few and a few hundred painted cells each)
#pragma clang loop unroll(disable)
    for (int item = blockIdx.x;;) {
    if (item >= items) break;
    const int wa = d.bev_list[item];
    const int w = wa / A_T, a = wa - w * A_T;
    const int n = d.shape[w * 2 + 0];
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;
    const size_t i = (size_t)w * A_T + a;

    BEV_STAMP(t_begin);
    const float ex = d.px[i], ey = d.py[i];
    const Quat rot = quat_from_wz(d.qw[i], d.qz[i]);
    const Quat inv = quat_inv(rot);

    // ---- roads: first K in-radius in road order (src/sim.cpp:484-523) ----
    int count = 0;
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int base = 0; base < R && count < K; base += NT) {
        const int r = base + tid;
        bool in = false;
        V2 rel{0.f, 0.f};
        if (r < R) {
            const float2 xy = d.road_xy[(size_t)r0 + r];
            rel = ego_relative(ex, ey, inv, xy.x, xy.y);
            in = !(len_2(rel.x, rel.y) > radius);
        }
        const unsigned long long b = __ballot(in);
        if (lane == 0) s_wcnt[wave] = __popcll(b);
        __syncthreads();
        int before = count;
        for (int v = 0; v < wave; v++) before += s_wcnt[v];
        int total = 0;
        for (int v = 0; v < NWV; v++) total += s_wcnt[v];
        const int pos = before + __popcll(b & lower);
        if (in && pos < K) {
            const float4 a0 = d.road_aux[(size_t)(r0 + r) * 2], a1 = d.road_aux[(size_t)(r0 + r) * 2 + 1];
            const float yaw = quat_to_yaw(quat_mul(inv, quat_from_wz(a0.x, a0.y)));
            const float minw = (2 * radius / RES);
            s_ent[pos] = bev_entity(rel.x, rel.y, yaw, a0.z, fmaxf(a0.w, minw), (int)(size_t)a1.y, radius);
        }
        count = min(count + total, K);
        __syncthreads();
    }
    BEV_STAMP(t_roads);
    // ---- partners in OtherAgents order (src/sim.cpp:526-554) ----
    {
        const int j = tid;  // A_T <= 128 <= NT
        bool in = false;
        V2 rel{0.f, 0.f};
        if (j < n && j != a) {
            const size_t oi = (size_t)w * A_T + j;
            rel = ego_relative(ex, ey, inv, d.px[oi], d.py[oi]);
            in = !(len_2(rel.x, rel.y) > radius);
        }
        const unsigned long long b = __ballot(in);
        if (lane == 0) s_wcnt[wave] = __popcll(b);
        __syncthreads();
        int before = count;
        for (int v = 0; v < wave; v++) before += s_wcnt[v];
        int total = 0;
        for (int v = 0; v < NWV; v++) total += s_wcnt[v];
        if (in) {
            const size_t oi = (size_t)w * A_T + j;
            const float yaw = quat_to_yaw(quat_mul(inv, quat_from_wz(d.qw[oi], d.qz[oi])));
            s_ent[before + __popcll(b & lower)] = bev_entity(rel.x, rel.y, yaw, d.len[oi], d.wid[oi], d.etype[oi], radius);
        }
        if (tid == 0) s_ne = count + total;
        __syncthreads();
    }
    const int ne = s_ne;
    BEV_STAMP(t_ents);

    // ---- paint and write out, one band of BR grid rows at a time ----
    // A cell holds (entity position + 1) << 8 | type and is updated with an LDS atomic max, so "the last entity in the
    // reference's order wins" holds whatever the order in which lanes get there: the entities are painted concurrently,
    // one per group of 16 lanes (road segments cover a few dozen cells; with one entity per wave and the rows dealt to
    // waves in bands, the wave that owned the rows around the agent did nearly all the work).  Bands keep the 32-bit
    // cells within the 40,000 bytes the byte grid used to take.
    constexpr int NB = GD_BEV_BANDS, BR = RES / NB, NG = NT / 16;
    static_assert(BR * NB == RES && (BR * RES) % 4 == 0, "bands");
    const int grp = tid >> 4, sl = tid & 15;
    const float scale_px = (2 * radius) / RES;
    float4 *out = reinterpret_cast<float4 *>(d.bev + i * (size_t)(RES * RES));
#pragma clang loop unroll(disable)
    for (int band = 0; band < NB; band++) {
        const int row_lo = band * BR, row_hi = row_lo + BR - 1;
#pragma clang loop unroll(disable)
        for (int e = grp; e < ne; e += NG) {
            const BevEnt en = s_ent[e];
            const int ex0 = en.x0(), ex1 = en.x1();
            const int y0 = max(en.y0(), row_lo), y1 = min(en.y1(), row_hi);
            if (y1 < y0 || ex1 < ex0) continue;
            const int nx = ex1 - ex0 + 1, cells = nx * (y1 - y0 + 1);
            const float inv_nx = 1.f / (float)nx;
            const unsigned int val = ((unsigned int)(e + 1) << 8) | (unsigned int)(en.type & 0xff);
            for (int c = sl; c < cells; c += 16) {
                int q = (int)(((float)c + 0.5f) * inv_nx);  // c / nx for c < 40,000 (checked and corrected below)
                q -= q * nx > c ? 1 : 0;
                q += (q + 1) * nx <= c ? 1 : 0;
                const int y = y0 + q, x = ex0 + c - q * nx;
                const float px = x * scale_px - radius, py = y * scale_px - radius;
                const float ldx = px - en.cx, ldy = py - en.cy;
                const float lx = ldx * en.cosy - ldy * en.siny;
                const float ly = ldx * en.siny + ldy * en.cosy;
                const float epsilon = 1e-3f;
                if (fabsf(lx) <= en.half_l + epsilon && fabsf(ly) <= en.half_w + epsilon) atomicMax(&s_cells[(y - row_lo) * RES + x], val);
            }
        }
        __syncthreads();
        // the band leaves as floats (float4 stores) and is zeroed for the next one on the way
#pragma clang loop unroll_count(2)
        for (int c = tid; c < BR * RES / 4; c += NT) {
            const uint4 v = reinterpret_cast<const uint4 *>(s_cells)[c];
            reinterpret_cast<uint4 *>(s_cells)[c] = make_uint4(0u, 0u, 0u, 0u);
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 cellv = {(float)(v.x & 0xff), (float)(v.y & 0xff), (float)(v.z & 0xff), (float)(v.w & 0xff)};
            __builtin_nontemporal_store(cellv, reinterpret_cast<f4 *>(out + band * (BR * RES / 4) + c));  // written once, not read by the step
        }
        __syncthreads();
    }
    if (tid == 0) s_item = (int)gridDim.x + atomicAdd(d.bev_count + 1, 1);
    __syncthreads();
    item = s_item;
    }  // (the list)
#ifdef GD_STAMPS
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0 && wg < 8192) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_bev_stamps[wg];
        o[0] = t_end; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0;
        o[5] = 0; o[6] = (unsigned long long)items; o[7] = 0;  // (per-phase stamps were per workgroup = per agent before the list)
    }
#endif
}

// ------------------------------------------------------------------------------------------
// LiDAR: 3 planes x 50 rays per agent.  PARITY UNPINNED: the reference traces rays through
// Madrona's absent 3-D BVH; the geometry is restated from the collision meshes' extents exactly as
// in oracle/gd_oracle.c (lidar_system): horizontal rays see the entities whose scaled z-range
// contains the ray height, as 2-D boxes; a box containing the origin is not hit.
// One workgroup per world, one wave per agent at a time.  Entity-major: in batches of 64 every lane
// takes one entity (agent or road), culls it by plane, by range and by the angular interval its
// bounding circle subtends, and the wave then runs the slab tests one (entity, ray) PAIR per lane:
// entities that subtend at most 16 rays queue their pairs in LDS (prefix sum over the lanes), wider
// (close) ones go to a per-wave list that is traced afterwards with one lane per ray.  The nearest
// hit per (plane, ray) is kept with a 64-bit LDS atomicMin on (t bits, entity row): ties go to the
// lowest entity row like the oracle's scan order.
// ------------------------------------------------------------------------------------------
// Slab test of the ray o + t d against the box (centre c, yaw quaternion (qw, qz), half extents hx, hy) in the
// box frame.  The oracle divides; here each axis multiplies by one reciprocal (v_rcp_f32, 1 ulp): hit
// distances move by an ulp or two, far inside the 1e-4 test tolerance.
__device__ __forceinline__ bool ray_box(float ox, float oy, float dx, float dy, float cx, float cy, Quat rot, float hx,
                                        float hy, float &t_out) {
    const V2 lo = rotate_yaw(rot.w, -rot.z, ox - cx, oy - cy);  // inverse rotation
    const V2 ld = rotate_yaw(rot.w, -rot.z, dx, dy);
    float tmin = -INFINITY, tmax = INFINITY;
    if (ld.x == 0.f) {
        if (lo.x < -hx || lo.x > hx) return false;
    } else {
        const float inv = __builtin_amdgcn_rcpf(ld.x);
        const float t1 = (-hx - lo.x) * inv, t2 = (hx - lo.x) * inv;
        tmin = fminf(t1, t2);
        tmax = fmaxf(t1, t2);
    }
    if (ld.y == 0.f) {
        if (lo.y < -hy || lo.y > hy) return false;
    } else {
        const float inv = __builtin_amdgcn_rcpf(ld.y);
        const float t1 = (-hy - lo.y) * inv, t2 = (hy - lo.y) * inv;
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    if (!(tmax >= tmin) || !(tmin > 0.f)) return false;
    t_out = tmin;
    return true;
}

// Culling only (never the hit arithmetic): atan2 to ~1e-5 rad (Abramowitz & Stegun 4.4.49 on the octant
// ratio) and an upper bound of asin (x + (pi/2 - 1) x^3 >= asin x on [0, 1]); the ray windows derived
// from them carry 0.02 rad and one ray of slack on either side.
__device__ __forceinline__ float cull_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float r = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float r2 = r * r;
    float t = r * (0.9998660f + r2 * (-0.3302995f + r2 * (0.1801410f + r2 * (-0.0851330f + 0.0208351f * r2))));
    t = ay > ax ? 1.57079632679f - t : t;
    t = x < 0.f ? 3.14159265359f - t : t;
    return y < 0.f ? -t : t;
}
__device__ __forceinline__ float cull_asin_upper(float x) { return x + 0.5707963268f * x * x * x; }

template <int A_T>
__global__ __launch_bounds__(256) void k_lidar(DevSim d) {
    // One workgroup per 16 agents of a world, one wave per agent at a time: a grid of (agents x worlds)
    // one-agent workgroups spent most of its time in workgroup launch/teardown, one workgroup per world
    // left the worlds with many agents as a long tail.
    constexpr int NS = GD_NUM_LIDAR_SAMPLES;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int wave = tid >> 6, lane = tid & 63;
    const int n = d.shape[w * 2 + 0];
    const int r0 = d.road_off[w];
    const int R = d.road_off[w + 1] - r0;

    constexpr int HEAVY_CAP = 256;
    __shared__ unsigned long long s_best[4][3 * NS];
    __shared__ int s_heavy[4][HEAVY_CAP];
    // per 64-entity batch: the culled (entity, ray) pairs and the entities' boxes, so that the slab tests run
    // with one pair per lane whatever the per-entity ray counts are
#ifndef GD_LIDAR_LIGHT_MAX
#define GD_LIDAR_LIGHT_MAX 16
#endif
    constexpr int LIGHT_MAX = GD_LIDAR_LIGHT_MAX;  // entities that subtend more rays go to the lane-per-ray list
    constexpr int ITEM_CAP = 64 * LIGHT_MAX;
    __shared__ unsigned short s_items[4][ITEM_CAP];
    __shared__ float4 s_ent[4][64][2];  // {cx, cy, qw, qz}, {hx, hy, planes, entity row}
    __shared__ float s_x[4][NS], s_y[4][NS], s_dx[4][NS], s_dy[4][NS];

    const float half = d.lidar_half_angle > 0.f ? d.lidar_half_angle : kPi / 3;
    const float step = 2.f * half / (float)NS;  // angle between neighbouring rays
    const float inv_step = 1.f / step;
    const float offs[3] = {0.5f, 0.1f, -0.1f};  // src/consts.hpp:42-44

    // blockIdx.y: a group of GROUP agents of the world (worlds with many agents get several workgroups)
    constexpr int GROUP = GD_LIDAR_GROUP;
    const int a_end = min(n, (int)(blockIdx.y + 1) * GROUP);
    for (int a = blockIdx.y * GROUP + wave; a < a_end; a += 4) {
        const size_t i = (size_t)w * A_T + a;
        // returns that cannot have changed since they were last traced are left in place (k_world_step's verdict: engine.hpp
        // lidar_dirty; only wave-level synchronisation inside this loop, so a wave may skip an agent on its own)
        if (d.lidar_dirty[i] == 0) continue;
        const float ox = d.px[i], oy = d.py[i], oz = d.pz[i];
        const Quat rot = quat_from_wz(d.qw[i], d.qz[i]);
        const Quat inv = quat_inv(rot);
        const float head_angle = d.controlled[i] ? d.action[i * 10 + 2] : 0.f;
        unsigned long long *best = s_best[wave];
        if (lane < NS) {
            const float theta = half * (2 * (float)lane / (float)NS - 1) + head_angle;
            const float x = p_cos(theta), y = p_sin(theta);
            const V3 fwd = quat_rotate(rot, V3{0.f, 1.f, 0.f}), right = quat_rotate(rot, V3{1.f, 0.f, 0.f});
            V3 rd{x * right.x + y * fwd.x, x * right.y + y * fwd.y, x * right.z + y * fwd.z};
            const float invl = 1.f / sqrtf(rd.x * rd.x + rd.y * rd.y + rd.z * rd.z);
            s_x[wave][lane] = x; s_y[wave][lane] = y;
            s_dx[wave][lane] = rd.x * invl; s_dy[wave][lane] = rd.y * invl;
        }
        for (int t = lane; t < 3 * NS; t += 64) best[t] = ~0ull;
        int nheavy = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // An agent farther from the box around its world's roads than the rays reach plus the largest road's bounding radius can
        // hit no road: only the agents are looked at (an object that is not valid at t = 0 sits ten kilometres from the map, a
        // finished one at the padding position: four agents in five on the Waymo tiles, nine entity batches each instead of one).
        const float4 bb = d.road_bbox[w];
        const float far = 200.f + d.road_rbmax[w] + 1.f;
        const float dxo = fmaxf(fmaxf(bb.x - ox, ox - bb.z), 0.f), dyo = fmaxf(fmaxf(bb.y - oy, oy - bb.w), 0.f);
        const int e_end = (R > 0 && !(dxo * dxo + dyo * dyo > far * far)) ? n + R : n;  // (NaN poses look at everything)
        for (int eb = 0; eb < e_end; eb += 64) {  // uniform trip count: the ballot below needs every lane
            const int e = eb + lane;
            float cx = 0.f, cy = 0.f, hx = 0.f, hy = 0.f, zlo = 1.f, zhi = 0.f;
            Quat q{1.f, 0.f, 0.f, 0.f};
            bool valid = e < e_end && e != a;
            if (valid && e < n) {
                const size_t oi = (size_t)w * A_T + e;
                cx = d.px[oi]; cy = d.py[oi];
                q = quat_from_wz(d.qw[oi], d.qz[oi]);
                hx = d.sc0[oi]; hy = d.sc1[oi];
                zlo = d.pz[oi]; zhi = d.pz[oi] + 2 * GD_VEHICLE_SCALE;  // agent mesh z in [0,2], Scale d2 = 0.7
            } else if (valid) {
                const int r = r0 + (e - n);
                const float2 xy = d.road_xy[r];
                const float4 a0 = d.road_aux[(size_t)r * 2], a1 = d.road_aux[(size_t)r * 2 + 1];
                cx = xy.x; cy = xy.y;
                q = quat_from_wz(a0.x, a0.y);
                hx = a0.z; hy = a0.w;
                const int type = (int)a1.y;
                const float zc = type == ET_RoadEdge ? 1 + 0.1f : (type == ET_StopSign ? 1.f : 1 + -0.1f);
                zlo = zc - a1.x; zhi = zc + a1.x;  // cube mesh z in [-1,1] scaled by d2
            }
            int planes = 0;
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const float rz = oz + offs[p];
                if (rz >= zlo && rz <= zhi) planes |= 1 << p;
            }
            valid = valid && planes != 0;
            const V2 rel = rotate_yaw(inv.w, inv.z, cx - ox, cy - oy);
            const float rho = len_2(rel.x, rel.y), rb = sqrtf(hx * hx + hy * hy);
            valid = valid && !(rho > 200.f + rb);
            // candidate rays: theta_idx = -half + idx*step + head_angle within phi +- alpha (+ slack), all wraps
            const float phi = cull_atan2(rel.y, rel.x) - head_angle;
            const float alpha = (rho <= rb ? kPi : cull_asin_upper(fminf(1.f, rb / rho))) + 0.02f;
            int lo_w[3], hi_w[3], nrays = 0;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const float c = phi + (kw - 1) * kPiM2;
                lo_w[kw] = max((int)floorf((c - alpha + half) * inv_step) - 1, 0);
                hi_w[kw] = min((int)ceilf((c + alpha + half) * inv_step) + 1, NS - 1);
                nrays += max(hi_w[kw] - lo_w[kw] + 1, 0);
            }
            // Entities that subtend many rays (close ones) would make the whole wave loop 50 times for a
            // few lanes: they go to a per-wave list and are traced afterwards with one LANE PER RAY.
            const bool heavy = valid && nrays > LIGHT_MAX;
            const unsigned long long hb = __ballot(heavy);
            if (heavy) {
                const int pos = nheavy + __popcll(hb & ((1ull << lane) - 1ull));
                if (pos < HEAVY_CAP) s_heavy[wave][pos] = (planes << 28) | e;
            }
            const bool overflow = heavy && nheavy + __popcll(hb & ((1ull << lane) - 1ull)) >= HEAVY_CAP;
            nheavy = min(nheavy + __popcll(hb), HEAVY_CAP);
            // light entities (<= LIGHT_MAX rays, or the overflow of the heavy list): queue their (entity, ray) pairs
            const bool light = valid && !(heavy && !overflow);
            s_ent[wave][lane][0] = make_float4(cx, cy, q.w, q.z);
            s_ent[wave][lane][1] = make_float4(hx, hy, __int_as_float(planes), __int_as_float(e));
            int mine = light ? nrays : 0;
            if (mine > LIGHT_MAX) mine = 0;  // an overflowing heavy entity is traced right here, lane-serial (rare)
            int off = mine;          // inclusive prefix sum over the wave
#pragma unroll
            for (int st = 1; st < 64; st <<= 1) {
                const int v = __shfl_up(off, st);
                if (lane >= st) off += v;
            }
            const int total = __shfl(off, 63);
            off -= mine;
            if (mine > 0) {
#pragma unroll
                for (int kw = 0; kw < 3; kw++)
                    for (int idx = lo_w[kw]; idx <= hi_w[kw]; idx++) s_items[wave][off++] = (unsigned short)((lane << 6) | idx);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int j = lane; j < total; j += 64) {
                const int item = s_items[wave][j];
                const int el = item >> 6, idx = item & 63;
                const float4 e0 = s_ent[wave][el][0], e1 = s_ent[wave][el][1];
                float t;
                if (!ray_box(ox, oy, s_dx[wave][idx], s_dy[wave][idx], e0.x, e0.y, quat_from_wz(e0.z, e0.w), e1.x, e1.y, t)) continue;
                if (!(t <= 200.f)) continue;
                const int pl = __float_as_int(e1.z);
                const unsigned long long packed = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned int)__float_as_int(e1.w);
#pragma unroll
                for (int p = 0; p < 3; p++)
                    if (pl & (1 << p)) atomicMin(&best[p * NS + idx], packed);
            }
            if (light && nrays > LIGHT_MAX) {  // heavy-list overflow: same test, this lane alone
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    for (int idx = lo_w[kw]; idx <= hi_w[kw]; idx++) {
                        float t;
                        if (!ray_box(ox, oy, s_dx[wave][idx], s_dy[wave][idx], cx, cy, q, hx, hy, t)) continue;
                        if (!(t <= 200.f)) continue;
                        const unsigned long long packed = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned int)e;
#pragma unroll
                        for (int p = 0; p < 3; p++)
                            if (planes & (1 << p)) atomicMin(&best[p * NS + idx], packed);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // heavy entities: lane = ray
        for (int h = 0; h < nheavy; h++) {
            const int packed_e = s_heavy[wave][h];
            const int e = packed_e & 0x0fffffff, planes = (packed_e >> 28) & 7;
            float cx, cy, hx, hy;
            Quat q;
            if (e < n) {
                const size_t oi = (size_t)w * A_T + e;
                cx = d.px[oi]; cy = d.py[oi];
                q = quat_from_wz(d.qw[oi], d.qz[oi]);
                hx = d.sc0[oi]; hy = d.sc1[oi];
            } else {
                const int r = r0 + (e - n);
                const float2 xy = d.road_xy[r];
                const float4 a0 = d.road_aux[(size_t)r * 2];
                cx = xy.x; cy = xy.y;
                q = quat_from_wz(a0.x, a0.y);
                hx = a0.z; hy = a0.w;
            }
            if (lane < NS) {
                float t;
                if (ray_box(ox, oy, s_dx[wave][lane], s_dy[wave][lane], cx, cy, q, hx, hy, t) && t <= 200.f) {
                    const unsigned long long packed = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned int)e;
#pragma unroll
                    for (int p = 0; p < 3; p++)
                        if ((planes & (1 << p)) && packed < best[p * NS + lane]) best[p * NS + lane] = packed;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int t = lane; t < 3 * NS; t += 64) {
            const unsigned long long b = best[t];
            float *o = d.lidar + (i * 3 * NS + t) * 4;
            if (b == ~0ull) {
                o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
            } else {
                const float tt = __uint_as_float((unsigned int)(b >> 32));
                const int e = (int)(b & 0xffffffffu);
                const int type = e < n ? d.etype[(size_t)w * A_T + e] : (int)d.road_aux[(size_t)(r0 + e - n) * 2 + 1].y;
                const int idx = t % NS;
                o[0] = tt; o[1] = (float)type; o[2] = tt * s_x[wave][idx]; o[3] = tt * s_y[wave][idx];
            }
        }
        if (lane == 0) d.lidar_head[i] = head_angle;  // (what k_world_step compares the next action row's head angle with)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// live_list compacted to the agents whose bev_dirty flag is set (the order inside the list is the order the workgroups' atomic
// adds arrive in: it decides nothing but who rasterises whom)
__global__ __launch_bounds__(1024) void k_bev_list(DevSim d) {
    if (d.gate_any && *d.any_reset == 0) return;
    __shared__ int s_cnt[16], s_base;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int idx = blockIdx.x * 1024 + tid;
    const int wa = idx < d.live_count ? d.live_list[idx] : 0;
    const bool on = idx < d.live_count && d.bev_dirty[wa] != 0;
    const unsigned long long b = __ballot(on);
    if (lane == 0) s_cnt[wave] = __popcll(b);
    __syncthreads();
    if (tid == 0) {
        int total = 0;
        for (int v = 0; v < 16; v++) { const int c = s_cnt[v]; s_cnt[v] = total; total += c; }
        s_base = total ? atomicAdd(d.bev_count, total) : 0;
    }
    __syncthreads();
    if (on) d.bev_list[s_base + s_cnt[wave] + __popcll(b & ((1ull << lane) - 1ull))] = wa;
}

}  // namespace

void launch_bev(const DevSim &d, hipStream_t st) {
    if (d.live_count == 0) return;
    (void)hipMemsetAsync(d.bev_count, 0, 2 * sizeof(int32_t), st);  // [0] the list's length, [1] items claimed beyond the grid's first
    hipLaunchKernelGGL(k_bev_list, dim3((d.live_count + 1023) / 1024), dim3(1024), 0, st, d);
    // a fixed grid: up to 12 workgroups' worth of agents per CU slot (three 512-thread workgroups fit a CU), fewer for small batches
    const dim3 grid((unsigned int)std::min(d.live_count, 3 * 256 * 4));
    constexpr int NT = 512;  // 8 waves x 25 grid rows (measured: 256 threads 3.5 ms, 512 3.1 ms, 640 5.4 ms)
    if (d.A == 64) hipLaunchKernelGGL((k_bev<64, NT>), grid, dim3(NT), 0, st, d);
    else hipLaunchKernelGGL((k_bev<128, NT>), grid, dim3(NT), 0, st, d);
}

void launch_lidar(const DevSim &d, hipStream_t st) {
    const dim3 grid(d.W, d.A / GD_LIDAR_GROUP);
    if (d.A == 64) hipLaunchKernelGGL(k_lidar<64>, grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL(k_lidar<128>, grid, dim3(256), 0, st, d);
}

}  // namespace gd
