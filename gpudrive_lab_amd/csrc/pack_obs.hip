// Fused observation pack: what GPUDriveTorchEnv.get_obs() assembles
// with ~20 torch ops and a clone per tensor (reference gpudrive/env/env_torch.py:756-896,1172-1216;
// normalisation in gpudrive/datatypes/observation.py:71-90,229-262, gpudrive/datatypes/roadgraph.py:
// 329-364; constants gpudrive/env/constants.py:6-21) in one pass over the exported tensors:
//   out[w][a] = ego(6) | partners (A-1) x 6 | road points 200 x 13      (norm_obs = True)
// Divisions are true IEEE divisions like torch's CPU kernels (torch's CUDA kernels multiply by the
// reciprocal of a scalar divisor, which may differ in the last bit).
#include <hip/hip_runtime.h>

#include "engine.hpp"

namespace gd {

namespace {

constexpr int K = GD_MAP_OBS_K;
constexpr float kAgentScale = GD_VEHICLE_SCALE;  // madrona_gpudrive.vehicleScale
constexpr float kTwoPi = 6.283185307179586f;     // constants.MAX_ORIENTATION_RAD = 2 * np.pi

__device__ __forceinline__ float norm_min_max(float x, float lo, float hi) {  // gpudrive/utils/geometry.py:15-26
    return 2.f * ((x - lo) / (hi - lo)) - 1.f;
}

template <int A_T>
__global__ __launch_bounds__(256) void k_pack_obs(DevSim d, float *out) {
    constexpr int D = 6 + (A_T - 1) * 6 + K * 13;
    const int w = blockIdx.x, tid = threadIdx.x;
    float *ow = out + (size_t)w * A_T * D;
    // ego, env_torch.py:756-800
    for (int a = tid; a < A_T; a += 256) {
        const float *s = d.self_obs + ((size_t)w * A_T + a) * 8;
        float *o = ow + (size_t)a * D;
        o[0] = s[0] / 100.f;
        o[1] = (s[1] * kAgentScale) / 30.f;
        o[2] = (s[2] * kAgentScale) / 15.f;
        o[3] = norm_min_max(s[4], -1000.f, 1000.f);
        o[4] = norm_min_max(s[5], -1000.f, 1000.f);
        o[5] = s[6];
    }
    // partners, env_torch.py:828-858
    for (int p = tid; p < A_T * (A_T - 1); p += 256) {
        const int a = p / (A_T - 1), k = p - a * (A_T - 1);
        const float *s = d.partner + ((size_t)w * A_T * (A_T - 1) + p) * 9;
        float *o = ow + (size_t)a * D + 6 + k * 6;
        o[0] = s[0] / 100.f;
        o[1] = norm_min_max(s[1], -1000.f, 1000.f);
        o[2] = norm_min_max(s[2], -1000.f, 1000.f);
        o[3] = s[3] / kTwoPi;
        o[4] = (s[4] * kAgentScale) / 30.f;
        o[5] = (s[5] * kAgentScale) / 15.f;
    }
    // road points, env_torch.py:860-896 (one-hot over 7 road point types)
    for (int p = tid; p < A_T * K; p += 256) {
        const int a = p / K, k = p - a * K;
        const float *s = d.agent_map + ((size_t)w * A_T * K + p) * 9;
        float *o = ow + (size_t)a * D + 6 + (A_T - 1) * 6 + k * 13;
        o[0] = norm_min_max(s[0], -1000.f, 1000.f);
        o[1] = norm_min_max(s[1], -1000.f, 1000.f);
        o[2] = s[2] / 100.f;
        o[3] = s[3] / 100.f;
        o[4] = s[4] / 100.f;
        o[5] = s[5] / kTwoPi;
        const int type = (int)(long long)s[6];
#pragma unroll
        for (int c = 0; c < 7; c++) o[6 + c] = type == c ? 1.f : 0.f;
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
    if (d.A == 64) hipLaunchKernelGGL(k_pack_obs<64>, dim3(d.W), dim3(256), 0, st, d, out);
    else hipLaunchKernelGGL(k_pack_obs<128>, dim3(d.W), dim3(256), 0, st, d, out);
}

void launch_expert_actions(const DevSim &d, hipStream_t st, float *actions, float *pos, float *vel, float *yaw, int *valid) {
    const size_t n = (size_t)d.W * d.A * GD_EPISODE_LEN;
    hipLaunchKernelGGL(k_expert_actions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, actions, pos, vel, yaw, valid);
}

void launch_set_log_actions(const DevSim &d, hipStream_t st, int t) {
    hipLaunchKernelGGL(k_set_log_actions, dim3((d.W * d.A + 255) / 256), dim3(256), 0, st, d, t);
}

}  // namespace gd
