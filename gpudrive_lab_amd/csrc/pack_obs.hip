// Fused observation pack: what GPUDriveTorchEnv.get_obs() assembles
// with ~20 torch ops and a clone per tensor (reference gpudrive/env/env_torch.py:756-896,1172-1216;
// normalisation in gpudrive/datatypes/observation.py:71-90,229-262, gpudrive/datatypes/roadgraph.py:
// 329-364; constants gpudrive/env/constants.py:6-21) in one pass over the exported tensors:
//   out[w][a] = ego(6) | partners (A-1) x 6 | road points 200 x 13      (norm_obs = True)
// Divisions are true IEEE divisions like torch's CPU kernels (torch's CUDA kernels multiply by the
// reciprocal of a scalar divisor, which may differ in the last bit).
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "pack_cols.hpp"

namespace gd {

namespace {

#ifndef GD_PACK_PARTS
#define GD_PACK_PARTS 64  // workgroups per world of k_pack_obs: one agent slot each (4 or 16 per world measured the same within noise)
#endif
constexpr int K = GD_MAP_OBS_K;
// Per agent: the source rows (63 x 9 partner floats, 200 x 9 road floats: 9.5 KB) are staged in LDS with
// coalesced loads (16-byte loads for the road rows), then one thread per FOUR consecutive output floats
// (rows are 2984 floats, so float4 groups never straddle a row) computes from LDS and issues one 16-byte
// store: both directions of the 1.4 GB this pass moves per step at 1024 x 64 are fully coalesced.  Every
// output element costs ONE true division: numerator and divisor are selected per column first (pack_cols.hpp).
template <int A_T>
__device__ __forceinline__ float pack_element(const float *self, const float *partner, const float *road, int j) {
    if (j < 6) return pack_ego_col(self, j);
    if (j < 6 + (A_T - 1) * 6) {
        const int p = j - 6, k = p / 6, c = p - k * 6;
        return pack_partner_col(partner[k * 9 + c], c);
    }
    const int p = j - 6 - (A_T - 1) * 6, k = p / 13, c = p - k * 13;
    return pack_road_col(road[k * 9 + (c < 6 ? c : 6)], c);
}

template <int A_T>
__global__ __launch_bounds__(256) void k_pack_obs(DevSim d, float *out) {
    constexpr int D = 6 + (A_T - 1) * 6 + K * 13;
    static_assert(D % 4 == 0 && (K * 9) % 4 == 0, "rows are whole float4 groups");
    constexpr int Q = D / 4, NP = (A_T - 1) * 9, NR = K * 9;
    constexpr int GROUP = A_T / GD_PACK_PARTS;  // blockIdx.y: a part of the world's agent slots
    __shared__ float s_self[8];
    __shared__ float s_partner[NP];
    __shared__ __attribute__((aligned(16))) float s_road[NR];
    const int w = blockIdx.x, tid = threadIdx.x;
    for (int al = 0; al < GROUP; al++) {
        const size_t agent = (size_t)w * A_T + blockIdx.y * GROUP + al;
        if (tid < 8) s_self[tid] = d.self_obs[agent * 8 + tid];
        for (int t = tid; t < NP; t += 256) s_partner[t] = d.partner[agent * NP + t];
        const float4 *rsrc = reinterpret_cast<const float4 *>(d.agent_map + agent * NR);
        for (int t = tid; t < NR / 4; t += 256) reinterpret_cast<float4 *>(s_road)[t] = rsrc[t];
        __syncthreads();
        for (int q = tid; q < Q; q += 256) {
            float4 v;
            v.x = pack_element<A_T>(s_self, s_partner, s_road, 4 * q + 0);
            v.y = pack_element<A_T>(s_self, s_partner, s_road, 4 * q + 1);
            v.z = pack_element<A_T>(s_self, s_partner, s_road, 4 * q + 2);
            v.w = pack_element<A_T>(s_self, s_partner, s_road, 4 * q + 3);
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, reinterpret_cast<f4 *>(out + agent * D) + q);  // 782 MB written once: keep it out of the caches
        }
        __syncthreads();
    }
}

// ---- expert-action export and log playback ----
// GPUDriveTorchEnv.get_expert_actions() (reference gpudrive/env/env_torch.py:1445-1509) slices the
// expert trajectory rows ([pos 182 | vel 182 | yaw 91 | valid 91 | inferred action 910],
// gpudrive/datatypes/trajectory.py:24-41) and clamps the inferred actions per dynamics model:
//   classic / bicycle : columns 0..2, accel in [-6, 6], steer in [-0.3, 0.3]
//   delta_local       : columns 0..2, dx, dy in [-6, 6], dyaw in [-pi, pi]
//   state             : (x, y, 1, yaw, vx, vy, 0, 0, 0, 0)
// torch.clamp = min(max(x, lo), hi) with NaN propagated.
constexpr int T = GD_EPISODE_LEN;
constexpr float kPiF = 3.14159265358979323846f;  // torch.pi rounded to fp32

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }

// columns of the action the caller would feed for time step t of agent row `tr` (1456 floats)
__device__ __forceinline__ void expert_action(const float *tr, int t, int model, float *act /*3 or 10*/) {
    const float *inf = tr + 6 * T + t * 10;
    if (model == GD_DYNAMICS_STATE) {
        act[0] = tr[2 * t]; act[1] = tr[2 * t + 1]; act[2] = 1.f; act[3] = tr[4 * T + t];
        act[4] = tr[2 * T + 2 * t]; act[5] = tr[2 * T + 2 * t + 1];
        act[6] = 0.f; act[7] = 0.f; act[8] = 0.f; act[9] = 0.f;
    } else if (model == GD_DYNAMICS_DELTA_LOCAL) {
        act[0] = clampf(inf[0], -6.f, 6.f); act[1] = clampf(inf[1], -6.f, 6.f); act[2] = clampf(inf[2], -kPiF, kPiF);
    } else {
        act[0] = clampf(inf[0], -6.f, 6.f); act[1] = clampf(inf[1], -0.3f, 0.3f); act[2] = inf[2];
    }
}

__global__ __launch_bounds__(256) void k_expert_actions(DevSim d, float *actions, float *pos, float *vel, float *yaw, int *valid) {
    const size_t n = (size_t)d.W * d.A * T;  // one thread per (world, agent, time step)
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const size_t row = g / T;
    const int t = (int)(g - row * T);
    const float *tr = d.traj + row * GD_TRAJECTORY_FLOATS;
    const int model = d.p.dynamicsModel;
    if (actions) {
        float act[10];
        expert_action(tr, t, model, act);
        const int cols = model == GD_DYNAMICS_STATE ? 10 : 3;
        for (int c = 0; c < cols; c++) actions[g * cols + c] = act[c];
    }
    if (pos) { pos[g * 2] = tr[2 * t]; pos[g * 2 + 1] = tr[2 * t + 1]; }
    if (vel) { vel[g * 2] = tr[2 * T + 2 * t]; vel[g * 2 + 1] = tr[2 * T + 2 * t + 1]; }
    if (yaw) yaw[g] = tr[4 * T + t];
    if (valid) valid[g] = (int)tr[5 * T + t];  // .to(torch.int32) truncates
}

// advance_sim_with_log_playback (env_torch.py:1274-1293): step t feeds log_playback_traj[:, :, t, :]
// into action[:, :, :cols] for EVERY agent slot (env_torch.py:645-664), then steps the simulator.
__global__ __launch_bounds__(256) void k_set_log_actions(DevSim d, int t) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= d.W * d.A) return;
    const int model = d.p.dynamicsModel;
    float act[10];
    expert_action(d.traj + (size_t)g * GD_TRAJECTORY_FLOATS, t, model, act);
    const int cols = model == GD_DYNAMICS_STATE ? 10 : 3;
    for (int c = 0; c < cols; c++) d.action[(size_t)g * 10 + c] = act[c];
}

}  // namespace

void launch_pack_obs(const DevSim &d, hipStream_t st, float *out) {
    if (d.A == 64) hipLaunchKernelGGL(k_pack_obs<64>, dim3(d.W, GD_PACK_PARTS), dim3(256), 0, st, d, out);
    else hipLaunchKernelGGL(k_pack_obs<128>, dim3(d.W, GD_PACK_PARTS), dim3(256), 0, st, d, out);
}

void launch_expert_actions(const DevSim &d, hipStream_t st, float *actions, float *pos, float *vel, float *yaw, int *valid) {
    const size_t n = (size_t)d.W * d.A * GD_EPISODE_LEN;
    hipLaunchKernelGGL(k_expert_actions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, actions, pos, vel, yaw, valid);
}

void launch_set_log_actions(const DevSim &d, hipStream_t st, int t) {
    hipLaunchKernelGGL(k_set_log_actions, dim3((d.W * d.A + 255) / 256), dim3(256), 0, st, d, t);
}

}  // namespace gd
