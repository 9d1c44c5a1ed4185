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

}  // namespace

void launch_pack_obs(const DevSim &d, hipStream_t st, float *out) {
    if (d.A == 64) hipLaunchKernelGGL(k_pack_obs<64>, dim3(d.W), dim3(256), 0, st, d, out);
    else hipLaunchKernelGGL(k_pack_obs<128>, dim3(d.W), dim3(256), 0, st, d, out);
}

}  // namespace gd
