// HIP kernels of the MI355X-native GPUDrive step engine (gfx950, wave64).
//
// One workgroup per world.  Agent state is world-major SoA ([W][A] per field) so a wave's 64
// lanes load 64 consecutive floats; exported observation tensors keep the reference's AoS layout
// (API contract) and are written by thread->row mappings that keep a wave's stores contiguous.
// Citations are relative to the reference checkout.
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "gd_math.hpp"

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
    const float dx = v * p_cos(yaw + beta), dy = v * p_sin(yaw + beta);
    const float w = v * p_cos(beta) * tanDelta / length;
    const float new_yaw = angle_add(yaw, w * dt);
    const float new_speed = fmaxf(fminf(speed + act[0] * dt, maxSpeed), -maxSpeed);
    b.px += dx * dt;
    b.py += dy * dt;
    b.pz = 1.f;
    b.qw = p_cos(new_yaw / 2.f);
    b.qz = p_sin(new_yaw / 2.f);
    b.vx = new_speed * p_cos(new_yaw);
    b.vy = new_speed * p_sin(new_yaw);
    b.vz = 0.f;
}

__device__ __forceinline__ void forward_bicycle(float *act, Body &b) {  // :52-81 (clamps written back)
    act[0] = fmaxf(-6.0f, fminf(act[0], 6.0f));
    act[1] = fmaxf(-3.0f, fminf(act[1], 3.0f));
    const float dt = 0.1f;
    const float yaw = quat_to_yaw(quat_from_wz(b.qw, b.qz));
    const float speed = len_3(b.vx, b.vy, b.vz);
    b.px = (float)((double)(b.px + b.vx * dt) + 0.5 * act[0] * p_cos(yaw) * dt * dt);
    b.py = (float)((double)(b.py + b.vy * dt) + 0.5 * act[0] * p_sin(yaw) * dt * dt);
    const float delta_yaw = (float)(act[1] * ((double)(speed * dt) + 0.5 * act[0] * dt * dt));
    const float new_yaw = angle_add(yaw, delta_yaw);
    const float new_speed = speed + act[0] * dt;
    b.vx = new_speed * p_cos(new_yaw);
    b.vy = new_speed * p_sin(new_yaw);
    b.vz = 0.f;
    b.qw = p_cos(new_yaw / 2.f);
    b.qz = p_sin(new_yaw / 2.f);
}

__device__ __forceinline__ void forward_delta(const float *act, Body &b) {  // :83-115
    const float dt = 0.1f;
    const float yaw = quat_to_yaw(quat_from_wz(b.qw, b.qz));
    const float c = p_cos(yaw), s = p_sin(yaw);
    const float dx = act[0] * c - act[1] * s;
    const float dy = act[0] * s + act[1] * c;
    b.px = b.px + dx;
    b.py = b.py + dy;
    b.vx = dx / dt;
    b.vy = dy / dt;
    b.vz = 0.f;
    const float new_yaw = angle_add(yaw, act[2]);
    b.qw = p_cos(new_yaw / 2.f);
    b.qz = p_sin(new_yaw / 2.f);
}

__device__ __forceinline__ void forward_state(const float *act, Body &b) {  // :186-194
    b.px = act[0]; b.py = act[1]; b.pz = act[2];
    b.vx = act[4]; b.vy = act[5]; b.vz = act[6];
    b.qw = p_cos(act[3] / 2.f);
    b.qz = p_sin(act[3] / 2.f);
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

// collectPartnerObsSystem, src/sim.cpp:188-240, for one world by STEP_THREADS threads.  One thread per (ego, slot) row; a chunk
// of STEP_THREADS consecutive 36-byte rows is assembled in LDS (row stride 9 floats: conflict-free) and leaves as whole
// 16-byte pieces with streaming stores (a world's block starts at a multiple of 16 bytes and so does every chunk; only the
// piece that straddles the end of the live egos' rows goes element by element).
template <int A_T>
__device__ __forceinline__ void partner_rows(const DevSim &d, int w, int n, int a, const float *s_px, const float *s_py,
                                             const float *s_qw, const float *s_qz, const float *s_speed, const float *s_len,
                                             const float *s_wid, const float *s_hgt, const int *s_etype, const int *s_id) {
    {
        __shared__ __attribute__((aligned(16))) float s_rows[STEP_THREADS * 9];
        const int rows = n * (A_T - 1);
        float *base = d.partner + (size_t)w * A_T * (A_T - 1) * 9;
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int p0 = 0; p0 < rows; p0 += STEP_THREADS) {
            if (GD_DIAG_IS(d.step_dbg, 3)) break;
            const int p = p0 + a;
            if (p < rows) {
                const int ego = p / (A_T - 1), k = p - ego * (A_T - 1);
                float *o = s_rows + a * 9;
                if (k >= n - 1) {  // zero_nonexist(): id -2
                    o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = (float)ET_None; o[8] = -2.f;
                } else {
                    const int j = k < ego ? k : k + 1;  // OtherAgents order, src/level_gen.cpp:450-464
                    const Quat ego_inv = quat_inv(quat_from_wz(s_qw[ego], s_qz[ego]));
                    const V3 r = quat_rotate(ego_inv, V3{s_px[j] - s_px[ego], s_py[j] - s_py[ego], 0.f});
                    if (len_2(r.x, r.y) > d.p.observationRadius) {  // zero(): id -1
                        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = (float)ET_None; o[8] = -1.f;
                    } else {
                        const float heading = quat_to_yaw_row(quat_mul(ego_inv, quat_from_wz(s_qw[j], s_qz[j])));
                        o[0] = s_speed[j];
                        o[1] = r.x; o[2] = r.y;
                        o[3] = heading;
                        o[4] = s_len[j]; o[5] = s_wid[j]; o[6] = s_hgt[j];
                        o[7] = (float)s_etype[j];
                        o[8] = (float)s_id[j];
                    }
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
}


template <int A_T, bool MOVE>
__global__ __launch_bounds__(STEP_THREADS) void k_world_step(DevSim d) {
    const int w = blockIdx.x, a = threadIdx.x;
    if (!MOVE && d.gate_any && *d.any_reset == 0) return;  // device-driven reset pass: nothing was flagged this step
    const int n = d.shape[w * 2 + 0];
    const size_t i = (size_t)w * A_T + a;
    const bool live = a < n;

    __shared__ float s_px[A_T], s_py[A_T], s_qw[A_T], s_qz[A_T], s_speed[A_T];
    __shared__ float s_len[A_T], s_wid[A_T], s_hgt[A_T], s_rad[A_T];
    __shared__ int s_etype[A_T], s_id[A_T], s_flags[A_T];  // flags: bit0 active, bit1 static
    __shared__ float s_obb[14][A_T];
    __shared__ int s_hit[A_T];  // collision flags found by the threads sharing an agent

    Body b{};
    int collided = 0, done = 0, resp = RESP_Static, controlled = 0, etype = 0;
    uint32_t steps = 0;
    int32_t info0 = 0, info1 = 0, info2 = 0, info3 = 0;
    float sc0 = 0.f, sc1 = 0.f, length = 0.f, width = 0.f, height = 0.f, gx = 0.f, gy = 0.f;

    if (live) {
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
                b.qw = p_cos(heading / 2.f);
                b.qz = p_sin(heading / 2.f);
            }
        }
    }

    // ---- publish per-agent geometry for the pair phases ----
    bool active = false;
    if (a < A_T) s_hit[a] = 0;
    if (live) {
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
        if (active) {
            const Obb o = obb_from(b.px, b.py, quat_from_wz(b.qw, b.qz), sc0, sc1);
            const float *of = reinterpret_cast<const float *>(&o);
#pragma unroll
            for (int k = 0; k < 14; k++) s_obb[k][a] = of[k];
        }
    }
    __syncthreads();

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
            if (!me_static && !GD_DIAG_IS(d.step_dbg, 1)) {
                // road boxes of the broadphase cell under the agent's centre
                const GridHdr gh = d.grid[w];
                const float fx = (mx - gh.ox) * gh.inv_cell, fy = (my - gh.oy) * gh.inv_cell;
                if (gh.nx > 0 && fx >= 0.f && fy >= 0.f && fx < (float)gh.nx && fy < (float)gh.ny) {
                    const int cell = (int)fy * gh.nx + (int)fx;
                    const int c0 = d.cell_off[gh.cell_base + cell], c1 = d.cell_off[gh.cell_base + cell + 1];
                    const int bbase = d.box_off[w];
                    // the cull reads (centre, radius, type) from the cell-ordered copy, one candidate ahead; only a box that
                    // passes it is fetched through its index
                    const float4 *chdr = d.cell_hdr + gh.item_base;
                    const int32_t *citem = d.cell_items + gh.item_base;
                    int c = c0 + part;
                    float4 hdr_n = make_float4(0.f, 0.f, 0.f, 0.f);
                    int item_n = 0;
                    if (c < c1) { hdr_n = chdr[c]; item_n = citem[c]; }
                    for (; c < c1; c += P) {
                        const float4 hdr = hdr_n;
                        const size_t r = (size_t)(bbase + item_n);
                        if (c + P < c1) { hdr_n = chdr[c + P]; item_n = citem[c + P]; }
                        const int rtype = (int)hdr.w;
                        if (collision_pair_filtered(my_type, rtype)) continue;
                        const float dx = mx - hdr.x, dy = my - hdr.y;
                        const float rr = (my_rad + hdr.z) * 1.001f + 0.01f;
                        if (dx * dx + dy * dy > rr * rr) continue;
                        const float4 q1 = d.boxes[r * 5 + 1], q2 = d.boxes[r * 5 + 2], q3 = d.boxes[r * 5 + 3],
                                     q4 = d.boxes[r * 5 + 4];
                        const float tmp[16] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w,
                                               q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w};
                        Obb ro;
                        float *of = reinterpret_cast<float *>(&ro);
#pragma unroll
                        for (int k = 0; k < 14; k++) of[k] = tmp[k];
                        if (!obb_collided(me, ro)) continue;
                        hit |= 1;
                        if (rtype > ET_None && rtype <= ET_StopSign) hit |= 2;
                    }
                }
            }
            if (hit) atomicOr(&s_hit[ag], hit);
        }
        __syncthreads();
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
        // ---- collectAbsoluteObservationsSystem, :769-783 ----
        float *ao = d.abs_obs + i * 14;
        ao[0] = b.px; ao[1] = b.py; ao[2] = b.pz;
        ao[3] = rot.w; ao[4] = rot.x; ao[5] = rot.y; ao[6] = rot.z;
        ao[7] = quat_to_yaw(rot);
        ao[8] = gx; ao[9] = gy;
        ao[10] = length; ao[11] = width; ao[12] = height;
        ao[13] = (float)s_id[a];
    }

    // ---- collectPartnerObsSystem, :188-240: here, or in k_partner_rows on a stream of its own beside the road kernels ----
    if (!d.p.disableClassicalObs && !d.split_partner)
        partner_rows<A_T>(d, w, n, a, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id);
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
    partner_rows<A_T>(d, w, n, a, s_px, s_py, s_qw, s_qz, s_speed, s_len, s_wid, s_hgt, s_etype, s_id);
}

}  // namespace

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
    case KERNEL_MAP_OBS: launch_map_obs(d, st); break;
    case KERNEL_PARTNER: hipLaunchKernelGGL(k_partner_rows<A_T>, grid, dim3(STEP_THREADS), 0, st, d); break;
    }
}

void launch_kernel(const DevSim &d, hipStream_t st, int which, bool move) {
    if (d.A == 64) launch_all<64>(d, st, which, move);
    else launch_all<128>(d, st, which, move);
}

}  // namespace gd
