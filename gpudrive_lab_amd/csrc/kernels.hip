// HIP kernels of the MI355X-native GPUDrive step engine (gfx950, wave64).
//
// One workgroup per world.  Agent state is world-major SoA ([W][A] per field) so a wave's 64
// lanes load 64 consecutive floats; exported observation tensors keep the reference's AoS layout
// (API contract) and are written by thread->row mappings that keep a wave's stores contiguous.
// Citations are relative to the reference checkout.
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "gd_math.hpp"
#include "map_rows.hpp"
#include "pack_cols.hpp"

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr int TRAJ = GD_TRAJECTORY_FLOATS;
constexpr int EPISODE = GD_EPISODE_LEN;

__device__ __forceinline__ int current_step(uint32_t t) {  // src/sim.cpp:23-25, kept inside the row
    long long k = (long long)EPISODE - (long long)t;
    if (k < 0 || k > EPISODE) k = EPISODE;
    return (int)k;
}

// ------------------------------------------------------------------------------------------
// dynamics, src/dynamics.hpp
// ------------------------------------------------------------------------------------------
struct Body {
    float px, py, pz, qw, qz, vx, vy, vz;
};

__device__ __forceinline__ void forward_classic(const float *act, float length, Body &b) {  // :11-50
    const float maxSpeed = FLT_MAX;
    const float dt = 0.1f;
    const float speed = len_3(b.vx, b.vy, b.vz);
    const float yaw = quat_to_yaw(quat_from_wz(b.qw, b.qz));
    const float v = fmaxf(fminf(speed + 0.5f * act[0] * dt, maxSpeed), -maxSpeed);
    const float tanDelta = p_tan(act[1]);
    const float beta = p_atan(0.5f * tanDelta);
    float sn, cs;
    p_sincos(yaw + beta, sn, cs);
    const float dx = v * cs, dy = v * sn;
    const float w = v * p_cos(beta) * tanDelta / length;
    const float new_yaw = angle_add(yaw, w * dt);
    const float new_speed = fmaxf(fminf(speed + act[0] * dt, maxSpeed), -maxSpeed);
    b.px += dx * dt;
    b.py += dy * dt;
    b.pz = 1.f;
    p_sincos(new_yaw / 2.f, b.qz, b.qw);
    p_sincos(new_yaw, sn, cs);
    b.vx = new_speed * cs;
    b.vy = new_speed * sn;
    b.vz = 0.f;
}

__device__ __forceinline__ void forward_bicycle(float *act, Body &b) {  // :52-81 (clamps written back)
    act[0] = fmaxf(-6.0f, fminf(act[0], 6.0f));
    act[1] = fmaxf(-3.0f, fminf(act[1], 3.0f));
    const float dt = 0.1f;
    const float yaw = quat_to_yaw(quat_from_wz(b.qw, b.qz));
    const float speed = len_3(b.vx, b.vy, b.vz);
    float sn, cs;
    p_sincos(yaw, sn, cs);
    b.px = (float)((double)(b.px + b.vx * dt) + 0.5 * act[0] * cs * dt * dt);
    b.py = (float)((double)(b.py + b.vy * dt) + 0.5 * act[0] * sn * dt * dt);
    const float delta_yaw = (float)(act[1] * ((double)(speed * dt) + 0.5 * act[0] * dt * dt));
    const float new_yaw = angle_add(yaw, delta_yaw);
    const float new_speed = speed + act[0] * dt;
    p_sincos(new_yaw, sn, cs);
    b.vx = new_speed * cs;
    b.vy = new_speed * sn;
    b.vz = 0.f;
    p_sincos(new_yaw / 2.f, b.qz, b.qw);
}

__device__ __forceinline__ void forward_delta(const float *act, Body &b) {  // :83-115
    const float dt = 0.1f;
    const float yaw = quat_to_yaw(quat_from_wz(b.qw, b.qz));
    float c, s;
    p_sincos(yaw, s, c);
    const float dx = act[0] * c - act[1] * s;
    const float dy = act[0] * s + act[1] * c;
    b.px = b.px + dx;
    b.py = b.py + dy;
    b.vx = dx / dt;
    b.vy = dy / dt;
    b.vz = 0.f;
    const float new_yaw = angle_add(yaw, act[2]);
    p_sincos(new_yaw / 2.f, b.qz, b.qw);
}

__device__ __forceinline__ void forward_state(const float *act, Body &b) {  // :186-194
    b.px = act[0]; b.py = act[1]; b.pz = act[2];
    b.vx = act[4]; b.vy = act[5]; b.vz = act[6];
    p_sincos(act[3] / 2.f, b.qz, b.qw);
}

// ------------------------------------------------------------------------------------------
// reset of flagged worlds: resetAgent / resetAgentInterface, src/level_gen.cpp:23-54
// ------------------------------------------------------------------------------------------
template <int A_T>
__global__ __launch_bounds__(A_T) void k_reset_worlds(DevSim d) {
    const int w = blockIdx.x, a = threadIdx.x;
    if (d.reset_flags[w] == 0) return;
    const int n = d.shape[w * 2 + 0];
    const size_t i = (size_t)w * A_T + a;
    if (a < n) {
        const float *t = d.traj + i * TRAJ;
        d.px[i] = t[0];
        d.py[i] = t[1];
        d.pz[i] = 1.f;
        const float heading = t[4 * 91];
        d.qw[i] = p_cos(heading / 2.f);
        d.qz[i] = p_sin(heading / 2.f);
        const bool is_static = d.resp[i] == RESP_Static;
        d.vx[i] = is_static ? 0.f : t[2 * 91 + 0];
        d.vy[i] = is_static ? 0.f : t[2 * 91 + 1];
        d.vz[i] = 0.f;
        float *act = d.action + i * 10;
#pragma unroll
        for (int k = 0; k < 10; k++) act[k] = 0.f;
        if (d.p.dynamicsModel == GD_DYNAMICS_STATE) act[2] = 1.f;  // getZeroAction, level_gen.hpp:31-34
        d.steps[i] = EPISODE;
        d.done[i] = 0;
        d.reward[i] = 0.f;
        int32_t *info = d.info + i * 5;
        info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0;
        info[4] = d.etype[i];
        d.resp_export[i] = d.resp[i];
        d.collided[i] = 0;
    }
    __syncthreads();
    if (a == 0) d.reset_flags[w] = 0;
}

// Rows that only change when a world is (re)built: padding agents, createPaddingEntities
// (src/level_gen.cpp:308-336).  Runs for worlds whose `rebuilt` flag is set.
template <int A_T>
__global__ __launch_bounds__(A_T) void k_init_padding_rows(DevSim d) {
    const int w = blockIdx.x;
    if (d.rebuilt_flags[w] == 0) return;
    const int n = d.shape[w * 2 + 0];
    const int npad = A_T - n;
    // scalar per-agent columns
    for (int a = n + threadIdx.x; a < A_T; a += blockDim.x) {
        const size_t i = (size_t)w * A_T + a;
        d.steps[i] = 0;
        d.done[i] = 1;
        d.reward[i] = 0.f;
        int32_t *info = d.info + i * 5;
        info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; info[4] = ET_None;
        d.resp_export[i] = RESP_Static;
        d.collided[i] = 0;
        float *so = d.self_obs + i * 8;
        so[0] = 0; so[1] = 0; so[2] = 0; so[3] = 0; so[4] = 0; so[5] = 0; so[6] = 0; so[7] = -1.f;
    }
    // partner rows: PartnerObservation::zero() (id -1)
    {
        float *base = d.partner + ((size_t)w * A_T + n) * (A_T - 1) * 9;
        const int total = npad * (A_T - 1) * 9;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int c = e % 9;
            base[e] = c == 8 ? -1.f : 0.f;
        }
    }
    // map rows: MapObservation::zero() (id -1, mapType -1)
    {
        float *base = d.agent_map + ((size_t)w * A_T + n) * K * 9;
        const int total = npad * K * 9;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int c = e % 9;
            base[e] = c >= 7 ? -1.f : 0.f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) d.rebuilt_flags[w] = 0;
}

// ------------------------------------------------------------------------------------------
// state step: movement -> collision -> reward -> --t -> done -> self/abs/partner observations
// (src/sim.cpp:294-383, 628-747, 560-626, 168-240, 769-783; task order :785-958)
// ------------------------------------------------------------------------------------------
constexpr int STEP_THREADS = 256;  // agents live on threads [0, A); all threads write partner rows

#ifdef GD_CLOCKS
// -DGD_CLOCKS builds (tools/build_expt.sh clk -DGD_CLOCKS; tools/step_clocks.py): s_memtime ticks per phase of k_world_step as seen by
// thread 0 of every workgroup, summed (gd_stat 32..39): 0 loads, 1 movement, 2 publish + OBB, 3 agent pairs, 4 road boxes,
// 5 flags + reward / done + write-back + self / absolute rows, 6 partner rows, 7 = workgroups
__device__ unsigned long long g_step_clk[8];
__device__ unsigned long long g_step_cnt[4];  // road-box items, items that pass the cull, hits (read with the clocks: gd_stat 40..42)
#define STEP_PHASE(n) do { if (a == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_step_clk[n], t_ - clk_prev); clk_prev = t_; } } while (0)
#else
#define STEP_PHASE(n) do {} while (0)
#endif

// collectPartnerObsSystem, src/sim.cpp:188-240, for one world by STEP_THREADS threads.  One thread per (ego, slot) row; a chunk
// of STEP_THREADS consecutive 36-byte rows is assembled in LDS (row stride 9 floats: conflict-free) and leaves as whole
// 16-byte pieces with streaming stores (a world's block starts at a multiple of 16 bytes and so does every chunk; only the
// piece that straddles the end of the live egos' rows goes element by element).
// One partner row into o[0..9): ego's view of the partner in slot k (OtherAgents order, src/level_gen.cpp:450-464), k < n - 1.
// Every rotation is a yaw rotation: rotateVec and the Hamilton product with the terms that multiply the zero x / y components
// dropped (gd_math.hpp rotate_yaw; map_rows.hpp road_row has the argument: every non-zero result is the same float, a zero may
// change its sign, which only the heading of an exactly opposite partner can see -- that case keeps the full product).
// `length() > radius` on the squared length (engine.cpp radius_key_max).
__device__ __forceinline__ void partner_row(float *o, const DevSim &d, int ego, int k, const float *s_px, const float *s_py, const float *s_qw,
                                            const float *s_qz, const float *s_speed, const float *s_len, const float *s_wid,
                                            const float *s_hgt, const int *s_etype, const int *s_id) {
    const int j = k < ego ? k : k + 1;
    const float ew = s_qw[ego], ez = s_qz[ego];
    const V2 r = rotate_yaw(ew, -ez, s_px[j] - s_px[ego], s_py[j] - s_py[ego]);
    if (r.x * r.x + r.y * r.y > d.radius_key_max) {  // zero(): id -1
        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = (float)ET_None; o[8] = -1.f;
        return;
    }
    const float rw = s_qw[j], rz = s_qz[j], iz = -ez;
    const float pw = ew * rw - iz * rz, pz = ew * rz + iz * rw;  // (w, z) of inverse(ego) * other
    const float wz = pw * pz;
    float heading;
    if (wz != 0.f) heading = atan2f(2.0f * wz, 1.0f - 2.0f * (pz * pz));
    else heading = quat_to_yaw_row(quat_mul(quat_inv(quat_from_wz(ew, ez)), quat_from_wz(rw, rz)));
    o[0] = s_speed[j];
    o[1] = r.x; o[2] = r.y;
    o[3] = heading;
    o[4] = s_len[j]; o[5] = s_wid[j]; o[6] = s_hgt[j];
    o[7] = (float)s_etype[j];
    o[8] = (float)s_id[j];
}

// `write_const` = false (step passes) leaves out what cannot have changed since the last pass that wrote everything: the rows of
// partner slots beyond the world's agents (zero_nonexist(), id -2: a function of the world's agent count alone -- 104 of an
// agent's 127 rows in a Waymo scene of 24 vehicles under this fork's 128 slots).  Reset passes follow every rebuild of the worlds
// (gd_create, set_maps, deleteAgents) and write everything.  In a ragged world a step pass therefore computes and stores n (n - 1)
// rows instead of n (A - 1): an ego's real rows are one contiguous piece of its block (not 16-byte aligned: they leave float by
// float, consecutive threads to consecutive addresses); a full world (n == A) has no such rows and takes the aligned path.
template <int A_T>
__device__ __forceinline__ void partner_rows(const DevSim &d, int w, int n, int a, const float *s_px, const float *s_py,
                                             const float *s_qw, const float *s_qz, const float *s_speed, const float *s_len,
                                             const float *s_wid, const float *s_hgt, const int *s_etype, const int *s_id,
                                             bool write_const, float *s_rows) {  // s_rows: STEP_THREADS * 9 floats of LDS, 16-byte aligned
    float *base = d.partner + (size_t)w * A_T * (A_T - 1) * 9;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (!write_const && n < A_T) {
        const int m = n - 1;  // real partners per ego
        const int rows = n * m;
        for (int p0 = 0; p0 < rows; p0 += STEP_THREADS) {
            if (GD_DIAG_IS(d.step_dbg, 3)) break;
            const int p = p0 + a;
            if (p < rows) {
                const int ego = p / m;
                partner_row(s_rows + a * 9, d, ego, p - ego * m, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id);
            }
            __syncthreads();
            const int nf = min(STEP_THREADS, rows - p0) * 9;  // floats of this chunk: float f of it is float p0 * 9 + f of the real rows
            for (int f = a; f < nf; f += STEP_THREADS) {
                const int gf = p0 * 9 + f, ego = gf / (m * 9);
                __builtin_nontemporal_store(s_rows[f], base + (size_t)ego * (A_T - 1) * 9 + (gf - ego * m * 9));
            }
            __syncthreads();
        }
        return;
    }
    const int rows = n * (A_T - 1);
    for (int p0 = 0; p0 < rows; p0 += STEP_THREADS) {
        if (GD_DIAG_IS(d.step_dbg, 3)) break;
        const int p = p0 + a;
        if (p < rows) {
            const int ego = p / (A_T - 1), k = p - ego * (A_T - 1);
            float *o = s_rows + a * 9;
            if (k >= n - 1) {  // zero_nonexist(): id -2
                o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = (float)ET_None; o[8] = -2.f;
            } else {
                partner_row(o, d, ego, k, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id);
            }
        }
        __syncthreads();
        const int nf = min(STEP_THREADS, rows - p0) * 9;  // floats of this chunk
        float *out = base + (size_t)p0 * 9;
        for (int q = a; q * 4 < nf; q += STEP_THREADS) {
            if (q * 4 + 4 <= nf) {
                __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(s_rows + q * 4), reinterpret_cast<f4 *>(out + q * 4));
            } else {
                for (int e = q * 4; e < nf; e++) out[e] = s_rows[e];
            }
        }
        __syncthreads();
    }
}


// The head of the packed observation (gd_attach_packed; pack_cols.hpp): ego (6) | partners (A - 1) x 6 of every live agent,
// written where the raw rows are produced.  An agent's head is A rows of 6 floats (row 0 the ego columns, row r the partner in
// slot r - 1) in one contiguous, 16-byte aligned block: a wave takes an agent at a time, 64 rows per pass -- lane r computes row
// r exactly as partner_rows does and normalises it, the block is laid out in LDS and leaves as whole 16-byte pieces.
// `s_self`: the agents' raw self-observation columns 0..6 (speed, length, width, -, goal x, goal y, collided), [A][8].
template <int A_T>
__device__ __forceinline__ void packed_head(const DevSim &d, int w, int n, int a, const float *s_px, const float *s_py,
                                            const float *s_qw, const float *s_qz, const float *s_speed, const float *s_len,
                                            const float *s_wid, const float *s_self) {
    constexpr int D = 6 + (A_T - 1) * 6 + K * 13;
    __shared__ __attribute__((aligned(16))) float s_head[STEP_THREADS / 64][64 * 6];
    const int wave = a >> 6, lane = a & 63;
    float *stage = s_head[wave];
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int ego = wave; ego < n; ego += STEP_THREADS / 64) {
        float *out = d.pack + ((size_t)w * A_T + ego) * D;
#pragma unroll
        for (int h = 0; h < A_T / 64; h++) {
            const int r = h * 64 + lane;
            float *o = stage + lane * 6;
            if (r == 0) {
#pragma unroll
                for (int c = 0; c < 6; c++) o[c] = pack_ego_col(s_self + ego * 8, c);
            } else {
                const int k = r - 1;
                float raw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // zero_nonexist() / zero(): the columns the pack reads are all zero
                if (k < n - 1) {
                    const int j = k < ego ? k : k + 1;
                    const float ew = s_qw[ego], ez = s_qz[ego];
                    const V2 rel = rotate_yaw(ew, -ez, s_px[j] - s_px[ego], s_py[j] - s_py[ego]);
                    if (!(rel.x * rel.x + rel.y * rel.y > d.radius_key_max)) {
                        const float rw = s_qw[j], rz = s_qz[j], iz = -ez;
                        const float pw = ew * rw - iz * rz, pz = ew * rz + iz * rw;
                        const float wz = pw * pz;
                        float heading;
                        if (wz != 0.f) heading = atan2f(2.0f * wz, 1.0f - 2.0f * (pz * pz));
                        else heading = quat_to_yaw_row(quat_mul(quat_inv(quat_from_wz(ew, ez)), quat_from_wz(rw, rz)));
                        raw[0] = s_speed[j]; raw[1] = rel.x; raw[2] = rel.y; raw[3] = heading; raw[4] = s_len[j]; raw[5] = s_wid[j];
                    }
                }
                pack_partner_row(raw, o);
            }
            wave_sync();
            for (int q = lane; q < 64 * 6 / 4; q += 64)
                __builtin_nontemporal_store(*reinterpret_cast<const f4 *>(stage + q * 4), reinterpret_cast<f4 *>(out + h * (64 * 6) + q * 4));
            wave_sync();
        }
    }
}

template <int A_T, bool MOVE>
__global__ __launch_bounds__(STEP_THREADS) void k_world_step(DevSim d) {
    const int w = blockIdx.x, a = threadIdx.x;
    if (!MOVE && d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int n = d.shape[w * 2 + 0];
    const size_t i = (size_t)w * A_T + a;
    const bool live = a < n;
    const GridHdr gh = d.grid[w];  // (uniform: requested with the first loads, long before the collision phase needs it)
    const int bbase = d.box_off[w];

    __shared__ float s_px[A_T], s_py[A_T], s_qw[A_T], s_qz[A_T], s_speed[A_T];
    __shared__ float s_len[A_T], s_wid[A_T], s_hgt[A_T], s_rad[A_T];
    __shared__ int s_etype[A_T], s_id[A_T], s_flags[A_T];  // flags: bit0 active, bit1 static
    __shared__ float s_obb[14][A_T];
    __shared__ int s_hit[A_T];  // collision flags found by the threads sharing an agent
    __shared__ float s_self[A_T * 8];  // the self-observation rows (packed_head reads them)
    // One staging buffer for three phases that follow each other with workgroup barriers between them: the positions before the
    // movement and who moved at all (the BEV's dirty flags, right after the publish), the road boxes that passed the cull
    // (agent | local box index << 8 | entity type << 28), and the partner rows on their way out.  (Each with LDS of its own,
    // k_world_step<128> took 42,000 bytes: three workgroups per CU instead of four, a second generation for 1024 worlds.)
    __shared__ __attribute__((aligned(16))) float s_stage[STEP_THREADS * 9];
    float *const s_opx = s_stage, *const s_opy = s_stage + A_T;
    unsigned long long *const s_moved = reinterpret_cast<unsigned long long *>(s_stage + 2 * A_T);  // a bit per agent slot
    // candidates looked at per trip of the road-box phase: six per thread (2,285 (agent, candidate) items per world on the bench
    // scene with 128 slots are two trips; with eight or nine per thread and one trip the kernel took the same 185 us)
    constexpr int SVCAP = 6 * STEP_THREADS;
    static_assert(SVCAP <= STEP_THREADS * 9 && 3 * A_T <= STEP_THREADS * 9, "the staging buffer holds each of its tenants");
    unsigned int *const s_sv = reinterpret_cast<unsigned int *>(s_stage);
    __shared__ int s_nsv[2];
    __shared__ int s_c0[A_T], s_coff[A_T], s_wtot[A_T / 64];  // road-box candidates: first entry, offset in the world's item list, per-wave totals

#ifdef GD_CLOCKS
    unsigned long long clk_prev = __builtin_amdgcn_s_memtime();
    if (a == 0) atomicAdd(&g_step_clk[7], 1ull);
#endif
    Body b{};
    int collided = 0, done = 0, resp = RESP_Static, controlled = 0, etype = 0;
    uint32_t steps = 0;
    int32_t info0 = 0, info1 = 0, info2 = 0, info3 = 0;
    float sc0 = 0.f, sc1 = 0.f, length = 0.f, width = 0.f, height = 0.f, gx = 0.f, gy = 0.f;

    // (every agent slot of the world is readable: the loads do not wait for the world's agent count)
    if (a < A_T) {
        b.px = d.px[i]; b.py = d.py[i]; b.pz = d.pz[i];
        b.qw = d.qw[i]; b.qz = d.qz[i];
        b.vx = d.vx[i]; b.vy = d.vy[i]; b.vz = d.vz[i];
        collided = d.collided[i];
        done = d.done[i];
        resp = d.resp[i];
        controlled = d.controlled[i];
        etype = d.etype[i];
        steps = d.steps[i];
        const int32_t *info = d.info + i * 5;
        info0 = info[0]; info1 = info[1]; info2 = info[2]; info3 = info[3];
        sc0 = d.sc0[i]; sc1 = d.sc1[i];
        length = d.len[i]; width = d.wid[i]; height = d.hgt[i];
        gx = d.goal_x[i]; gy = d.goal_y[i];
    }

#ifdef GD_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEP_PHASE(0);
    // ---- movementSystem, src/sim.cpp:294-383 ----
    if (MOVE && live) {
        if (collided) {
            if (d.p.collisionBehaviour == GD_COLLISION_AGENT_STOP) {
                done = 1;
                b.vx = 0.f; b.vy = 0.f; b.vz = 0.f;
            } else if (d.p.collisionBehaviour == GD_COLLISION_AGENT_REMOVED) {
                done = 1;
                b.px = kPadX; b.py = kPadY; b.pz = kPadZ;
                b.vx = 0.f; b.vy = 0.f; b.vz = 0.f;
            } else {
                collided = 0;
                info0 = 0; info1 = 0; info2 = 0;
            }
        }
        if (resp != RESP_Static) {
            if (done) {
                b.px = kPadX; b.py = kPadY; b.pz = kPadZ;
                b.vx = 0.f; b.vy = 0.f; b.vz = 0.f;
            } else if (controlled) {
                float act[10];
                float *ap = d.action + i * 10;
#pragma unroll
                for (int k = 0; k < 10; k++) act[k] = ap[k];
                switch (d.p.dynamicsModel) {
                case GD_DYNAMICS_INVERTIBLE_BICYCLE:
                    forward_bicycle(act, b);
                    ap[0] = act[0];
                    ap[1] = act[1];
                    break;
                case GD_DYNAMICS_DELTA_LOCAL: forward_delta(act, b); break;
                case GD_DYNAMICS_STATE: forward_state(act, b); break;
                default: forward_classic(act, length, b); break;
                }
            } else {
                const float *t = d.traj + i * TRAJ;
                const int k = current_step(steps);
                b.px = t[2 * k]; b.py = t[2 * k + 1]; b.pz = 1.f;
                b.vx = t[2 * 91 + 2 * k]; b.vy = t[2 * 91 + 2 * k + 1]; b.vz = 0.f;
                const float heading = t[4 * 91 + k];
                p_sincos(heading / 2.f, b.qz, b.qw);
            }
        }
    }

    STEP_PHASE(1);
    // ---- publish per-agent geometry for the pair phases ----
    bool active = false, moved = false;
    const bool track_moves = d.bev != nullptr || (d.lidar != nullptr && d.p.enableLidar != 0);  // (BEV rasters / LiDAR returns left in place where nothing changed)
    float theta = 0.f;  // quat_to_yaw of the pose after the movement: the agent's box and its absolute row both need it
    if (a < A_T) s_hit[a] = 0;
    if (live) {
        theta = quat_to_yaw(quat_from_wz(b.qw, b.qz));
        // isInvalidExpertOrDone, src/sim.cpp:631-662; agents parked at kPaddingPosition overlap nothing
        bool invalid;
        if (!controlled) {
            const float *t = d.traj + i * TRAJ;
            invalid = !(t[5 * 91 + current_step(steps)] != 0.f);
        } else {
            invalid = done && !collided;
        }
        active = !(b.pz == kPadZ) && !invalid;
        s_px[a] = b.px; s_py[a] = b.py; s_qw[a] = b.qw; s_qz[a] = b.qz;
        s_speed[a] = len_3(b.vx, b.vy, b.vz);
        s_len[a] = length; s_wid[a] = width; s_hgt[a] = height;
        s_etype[a] = etype;
        s_id[a] = d.agent_id[i];
        s_rad[a] = sqrtf(sc0 * sc0 + sc1 * sc1);
        s_flags[a] = (active ? 1 : 0) | (resp == RESP_Static ? 2 : 0);
        if (track_moves) {
            // (the pose before the movement is still what the state arrays hold: they are written back further down)
            const float old_px = d.px[i], old_py = d.py[i], old_pz = d.pz[i], old_qw = d.qw[i], old_qz = d.qz[i];
            s_opx[a] = old_px; s_opy[a] = old_py;
            moved = __float_as_uint(old_px) != __float_as_uint(b.px) || __float_as_uint(old_py) != __float_as_uint(b.py) ||
                    __float_as_uint(old_pz) != __float_as_uint(b.pz) ||
                    __float_as_uint(old_qw) != __float_as_uint(b.qw) || __float_as_uint(old_qz) != __float_as_uint(b.qz);
        }
        if (active) {
            const Obb o = obb_from_yaw(b.px, b.py, theta, sc0, sc1);
            const float *of = reinterpret_cast<const float *>(&o);
#pragma unroll
            for (int k = 0; k < 14; k++) s_obb[k][a] = of[k];
        }
    }
    if (track_moves && a < A_T) {  // (whole waves: the agents that moved, a bit per agent slot)
        const unsigned long long mv = __ballot(moved);
        if ((a & 63) == 0) s_moved[a >> 6] = mv;
    }
    __syncthreads();
    STEP_PHASE(2);
    // ---- which BEV rasters can have changed (collectBevObservationsSystem paints the in-radius roads and partners around the
    // agent, src/sim.cpp:462-555): the agent's own pose changed, or an agent whose pose changed is within the radius of it now or
    // was before it moved (a little more than the radius: whoever is marked without need is merely rasterised again) ----
    if (d.bev != nullptr && live) {
        int dirty = (!MOVE || d.pose_skip == 0 || d.bev_all_dirty != 0 || moved) ? 1 : 0;
        if (!dirty) {
            const float rr = d.p.observationRadius * 1.001f + 0.05f, r2 = rr * rr;
            const float mx = s_px[a], my = s_py[a];  // (this agent did not move: its old position is its new one)
#pragma unroll
            for (int h = 0; h < A_T / 64; h++) {
                for (unsigned long long m = s_moved[h]; m != 0ull && !dirty; m &= m - 1ull) {
                    const int j = h * 64 + __ffsll((long long)m) - 1;
                    const float dx = s_px[j] - mx, dy = s_py[j] - my, ox = s_opx[j] - mx, oy = s_opy[j] - my;
                    if (!(dx * dx + dy * dy > r2) || !(ox * ox + oy * oy > r2)) dirty = 1;
                }
            }
        }
        d.bev_dirty[i] = dirty;
    }
    // ---- the same for the LiDAR returns (lidarSystem, src/sim.cpp:394-460, 895-913): the agent's own pose or the head angle of its
    // action row changed, or an agent that moved is or was within the rays' 200 m plus its own bounding radius ----
    if (d.lidar != nullptr && d.p.enableLidar != 0 && live) {
        const float head = controlled ? d.action[i * 10 + 2] : 0.f;
        int dirty = (!MOVE || d.pose_skip == 0 || d.bev_all_dirty != 0 || moved || __float_as_uint(head) != __float_as_uint(d.lidar_head[i])) ? 1 : 0;
        if (!dirty) {
            const float mx = s_px[a], my = s_py[a];
#pragma unroll
            for (int h = 0; h < A_T / 64; h++) {
                for (unsigned long long m = s_moved[h]; m != 0ull && !dirty; m &= m - 1ull) {
                    const int j = h * 64 + __ffsll((long long)m) - 1;
                    const float rr = (200.f + s_rad[j]) * 1.001f + 0.1f, r2 = rr * rr;
                    const float dx = s_px[j] - mx, dy = s_py[j] - my, ox = s_opx[j] - mx, oy = s_opy[j] - my;
                    if (!(dx * dx + dy * dy > r2) || !(ox * ox + oy * oy > r2)) dirty = 1;
                }
            }
        }
        d.lidar_dirty[i] = dirty;
    }

    // ---- collisionDetectionSystem over broadphase candidates, src/sim.cpp:628-747, 792-801 ----
    // All STEP_THREADS threads work here: P = STEP_THREADS / A threads per agent share its candidate
    // agents and the road boxes of its broadphase cell; the flags they find are OR-ed through LDS.
    {
        constexpr int P = STEP_THREADS / A_T;
        const int ag = a % A_T, part = a / A_T;
        const int my_fl = ag < n ? s_flags[ag] : 0;
        if (my_fl & 1) {
            Obb me;
            {
                float *mf = reinterpret_cast<float *>(&me);
#pragma unroll
                for (int k = 0; k < 14; k++) mf[k] = s_obb[k][ag];
            }
            const float mx = s_px[ag], my = s_py[ag], my_rad = s_rad[ag];
            const int my_type = s_etype[ag];
            const bool me_static = (my_fl & 2) != 0;
            int hit = 0;  // bit 0 collided, bits 1..3 info0..info2
            for (int j = part; j < n; j += P) {
                if (GD_DIAG_IS(d.step_dbg, 2)) break;
                if (j == ag) continue;
                const int fl = s_flags[j];
                if (!(fl & 1)) continue;
                if (me_static && (fl & 2)) continue;  // static-static pairs are never candidates
                const float dx = mx - s_px[j], dy = my - s_py[j];
                const float rr = (my_rad + s_rad[j]) * 1.001f + 0.01f;
                if (dx * dx + dy * dy > rr * rr) continue;
                Obb ot;
                float *of = reinterpret_cast<float *>(&ot);
#pragma unroll
                for (int k = 0; k < 14; k++) of[k] = s_obb[k][j];
                if (!obb_collided(me, ot)) continue;
                const int otype = s_etype[j];
                if (collision_pair_filtered(my_type, otype)) continue;
                hit |= 1;
                if (otype > ET_None && otype <= ET_StopSign) hit |= 2;
                else if (otype == ET_Vehicle) hit |= 4;
                else if (otype <= ET_Cyclist) hit |= 8;
            }
            if (hit) atomicOr(&s_hit[ag], hit);
        }
        STEP_PHASE(3);
        // Road boxes of the broadphase cell under each agent's centre, as ONE list of (agent, candidate) items dealt evenly to
        // all STEP_THREADS threads.  (Rounds 1-4 gave every agent's candidates to the P threads that share the agent: the phase
        // lasted as long as the fullest cell of the world -- 36 % of a workgroup's time on the bench scene, per-phase clocks of a
        // -DGD_CLOCKS build -- while most threads had finished.)  The agent threads publish their cell's run of the cell-ordered
        // candidate arrays and its length, a scan over the agents turns the lengths into offsets, and item t belongs to the
        // agent whose offset range holds t.  Two items per thread and trip, so that their loads are in flight together.
        {
            int cnt = 0, c0 = 0;
            if (a < A_T && (my_fl & 1) && !(my_fl & 2) && !GD_DIAG_IS(d.step_dbg, 1)) {
                const float fx = (s_px[a] - gh.ox) * gh.inv_cell, fy = (s_py[a] - gh.oy) * gh.inv_cell;
                if (gh.nx > 0 && fx >= 0.f && fy >= 0.f && fx < (float)gh.nx && fy < (float)gh.ny) {
                    const int cell = (int)fy * gh.nx + (int)fx;
                    c0 = d.cell_off[gh.cell_base + cell];
                    cnt = d.cell_off[gh.cell_base + cell + 1] - c0;
                    c0 += gh.item_base;
                }
            }
            // inclusive scan over the lanes of a wave (DPP), the waves of agent threads chained through LDS
            int incl = cnt;
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);  // row_shr:1
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);  // row_shr:2
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);  // row_shr:4
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);  // row_shr:8
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
            constexpr int AW = A_T / 64;  // waves of agent threads
            if (a < A_T && (a & 63) == 63) s_wtot[a >> 6] = incl;
            if (a == 0) s_nsv[0] = 0;
            __syncthreads();
            if (a < A_T) {
                int base = 0;
#pragma unroll
                for (int k = 0; k < AW; k++) base += (k < (a >> 6)) ? s_wtot[k] : 0;
                s_c0[a] = c0;
                s_coff[a] = base + incl - cnt;
            }
            int total = 0;
#pragma unroll
            for (int k = 0; k < AW; k++) total += s_wtot[k];
            __syncthreads();
#ifdef GD_CLOCKS
            if (a == 0) atomicAdd(&g_step_cnt[0], (unsigned long long)total);
#endif
            // the exact test of one (agent, box) pair: the box's 14 floats (a gather), the agent's from LDS
            auto box_test = [&](unsigned int entry) {
#ifdef GD_CLOCKS
                atomicAdd(&g_step_cnt[1], 1ull);
#endif
                const int g = (int)(entry & 0xffu), rtype = (int)(entry >> 28);
                const size_t r = (size_t)(bbase + (int)((entry >> 8) & 0xfffffu));
                const float4 q1 = d.boxes[r * 5 + 1], q2 = d.boxes[r * 5 + 2], q3 = d.boxes[r * 5 + 3], q4 = d.boxes[r * 5 + 4];
                const float tmp[16] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w};
                Obb ro, me;
                float *of = reinterpret_cast<float *>(&ro), *mf = reinterpret_cast<float *>(&me);
#pragma unroll
                for (int k = 0; k < 14; k++) { of[k] = tmp[k]; mf[k] = s_obb[k][g]; }
                if (obb_collided(me, ro)) atomicOr(&s_hit[g], 1 | ((rtype > ET_None && rtype <= ET_StopSign) ? 2 : 0));
            };
            // U items per thread and trip, their candidate records requested together: the records of a world's cells are spread
            // over ~0.7 MB per world (every box is listed in the ~9 cells it can reach), 0.7 GB per batch, so each trip is a miss
            // all the way to HBM and the phase is as long as the number of dependent trips (1,280 items per world on the bench
            // scene: one trip).  A candidate that passes the cull goes on the workgroup's list of (agent, box) pairs -- one in
            // fifteen does, and which threads hold them is a matter of luck -- and the exact tests are dealt out again from the list.
            constexpr int U = SVCAP / STEP_THREADS;
            int trip = 0;
#pragma clang loop unroll(disable)
            for (int t0 = 0; t0 < total; t0 += SVCAP, trip ^= 1) {  // (uniform: the barriers inside are reached by every thread)
                int agu[U];
                float4 hdr[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int t = t0 + u * STEP_THREADS + a;
                    on[u] = t < total;
                    // the LAST agent whose offset is <= t: agents without candidates share the offset of the next agent that has
                    // some, and "last" picks that one
                    int lo = 0;
#pragma unroll
                    for (int st = A_T / 2; st > 0; st >>= 1)
                        if (s_coff[lo + st] <= t) lo += st;
                    agu[u] = lo;
                    hdr[u] = d.cell_hdr[on[u] ? s_c0[lo] + (t - s_coff[lo]) : 0];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int g = agu[u];
                    const unsigned int packed = __float_as_uint(hdr[u].w);  // type | local box index << 8 (engine.cpp)
                    const int rtype = (int)(packed & 0xffu);
                    const float dx = s_px[g] - hdr[u].x, dy = s_py[g] - hdr[u].y;
                    const float rr = (s_rad[g] + hdr[u].z) * 1.001f + 0.01f;
                    if (on[u] && !collision_pair_filtered(s_etype[g], rtype) && !(dx * dx + dy * dy > rr * rr))
                        s_sv[atomicAdd(&s_nsv[trip], 1)] = (unsigned int)g | (packed >> 8) << 8 | (packed & 0xfu) << 28;  // at most SVCAP items per trip
                }
                __syncthreads();
                const int nsv = s_nsv[trip];
                if (a == 0) s_nsv[trip ^ 1] = 0;  // (the next trip's counter: nobody touches it before the barrier below)
                for (int e = a; e < nsv; e += STEP_THREADS) box_test(s_sv[e]);
                __syncthreads();
            }
        }
        __syncthreads();
        STEP_PHASE(4);
        if (live && active) {
            const int hit = s_hit[a];
            if (hit & 1) collided = 1;
            if (hit & 2) info0 = 1;
            if (hit & 4) info1 = 1;
            if (hit & 8) info2 = 1;
        }
    }

    if (live) {
        // ---- rewardSystem, src/sim.cpp:560-587 ----
        const float dist = len_2(b.px - gx, b.py - gy);
        if (d.p.rewardType == GD_REWARD_DISTANCE_BASED) d.reward[i] = -dist;
        else if (d.p.rewardType == GD_REWARD_ON_GOAL_ACHIEVED) d.reward[i] = dist < d.p.distanceToGoalThreshold ? 1.f : 0.f;
        // ---- stepTrackerSystem, :589-592 ----
        if (MOVE) --steps;
        // ---- doneSystem, :597-626 ----
        const int32_t num_remaining = (int32_t)steps;
        if (num_remaining == EPISODE && done != 1) {
            done = 0;
        } else {
            if (num_remaining == 0) done = 1;
            if (done != 1 || info3 != 1) {
                if (dist < d.p.distanceToGoalThreshold) { done = 1; info3 = 1; }
            }
        }
        // ---- write back ----
        d.px[i] = b.px; d.py[i] = b.py; d.pz[i] = b.pz;
        d.qw[i] = b.qw; d.qz[i] = b.qz;
        d.vx[i] = b.vx; d.vy[i] = b.vy; d.vz[i] = b.vz;
        d.collided[i] = collided;
        d.done[i] = done;
        d.steps[i] = steps;
        int32_t *info = d.info + i * 5;
        info[0] = info0; info[1] = info1; info[2] = info2; info[3] = info3;

        // ---- collectSelfObsSystem, :168-186 ----
        const Quat rot = quat_from_wz(b.qw, b.qz);
        const V3 g = quat_rotate(quat_inv(rot), V3{gx - b.px, gy - b.py, 0.f});
        float *so = d.self_obs + i * 8;
        so[0] = s_speed[a];
        so[1] = length; so[2] = width; so[3] = height;
        so[4] = g.x; so[5] = g.y;
        so[6] = collided ? 1.f : 0.f;
        so[7] = (float)s_id[a];
        if (d.pack != nullptr) {
            float *ss = s_self + a * 8;
            ss[0] = s_speed[a]; ss[1] = length; ss[2] = width; ss[3] = height; ss[4] = g.x; ss[5] = g.y; ss[6] = collided ? 1.f : 0.f;
        }
        // ---- collectAbsoluteObservationsSystem, :769-783 ----
        float *ao = d.abs_obs + i * 14;
        ao[0] = b.px; ao[1] = b.py; ao[2] = b.pz;
        ao[3] = rot.w; ao[4] = rot.x; ao[5] = rot.y; ao[6] = rot.z;
        ao[7] = theta;
        ao[8] = gx; ao[9] = gy;
        ao[10] = length; ao[11] = width; ao[12] = height;
        ao[13] = (float)s_id[a];
    }

#ifdef GD_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEP_PHASE(5);
    // ---- collectPartnerObsSystem, :188-240: here, or in k_partner_rows on a stream of its own beside the road kernels ----
    if (!d.p.disableClassicalObs && !d.split_partner && !d.pack_only)
        partner_rows<A_T>(d, w, n, a, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id, !MOVE || d.pose_skip == 0, s_stage);
    if (d.pack != nullptr && !d.p.disableClassicalObs) {
        __syncthreads();  // the agent threads' self columns
        packed_head<A_T>(d, w, n, a, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_self);
    }
#ifdef GD_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEP_PHASE(6);
}

// The same rows as a kernel of their own: the engine runs it on a second stream while the road kernels (which do not read the
// partner rows) run on the first -- 148 MB of stores at 1024 x 64 that otherwise sit between two compute-bound phases.
template <int A_T>
__global__ __launch_bounds__(STEP_THREADS) void k_partner_rows(DevSim d) {
    const int w = blockIdx.x, a = threadIdx.x;
    if (d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int n = d.shape[w * 2 + 0];
    __shared__ float s_px[A_T], s_py[A_T], s_qw[A_T], s_qz[A_T], s_speed[A_T], s_len[A_T], s_wid[A_T], s_hgt[A_T];
    __shared__ int s_etype[A_T], s_id[A_T];
    if (a < n) {
        const size_t i = (size_t)w * A_T + a;
        s_px[a] = d.px[i]; s_py[a] = d.py[i]; s_qw[a] = d.qw[i]; s_qz[a] = d.qz[i];
        s_speed[a] = len_3(d.vx[i], d.vy[i], d.vz[i]);
        s_len[a] = d.len[i]; s_wid[a] = d.wid[i]; s_hgt[a] = d.hgt[i];
        s_etype[a] = d.etype[i];
        s_id[a] = d.agent_id[i];
    }
    __syncthreads();
    __shared__ __attribute__((aligned(16))) float s_rows[STEP_THREADS * 9];
    partner_rows<A_T>(d, w, n, a, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id, true, s_rows);
}

}  // namespace

#ifdef GD_CLOCKS
void step_clocks_read(unsigned long long *out) {  // and zero them; out[8..11] = the counters
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_clk), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_clk), z, sizeof(z));
    (void)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_step_cnt), 4 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_cnt), z, 4 * sizeof(unsigned long long));
}
#endif

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int A_T>
static void launch_all(const DevSim &d, hipStream_t st, int which, bool move) {
    const dim3 grid(d.W), block(A_T);
    switch (which) {
    case KERNEL_RESET: hipLaunchKernelGGL(k_reset_worlds<A_T>, grid, block, 0, st, d); break;
    case KERNEL_PADDING: hipLaunchKernelGGL(k_init_padding_rows<A_T>, grid, block, 0, st, d); break;
    case KERNEL_STATE:
        if (move) hipLaunchKernelGGL((k_world_step<A_T, true>), grid, dim3(STEP_THREADS), 0, st, d);
        else hipLaunchKernelGGL((k_world_step<A_T, false>), grid, dim3(STEP_THREADS), 0, st, d);
        break;
    case KERNEL_MAP_OBS: launch_map_obs(d, st, move); break;
    case KERNEL_PARTNER: hipLaunchKernelGGL(k_partner_rows<A_T>, grid, dim3(STEP_THREADS), 0, st, d); break;
    }
}

void launch_kernel(const DevSim &d, hipStream_t st, int which, bool move) {
    if (d.A == 64) launch_all<64>(d, st, which, move);
    else launch_all<128>(d, st, which, move);
}

}  // namespace gd
