// The columns of the packed observation (what GPUDriveTorchEnv.get_obs() assembles with norm_obs = True, reference
// gpudrive/env/env_torch.py:756-896, 1172-1216; normalisation in gpudrive/datatypes/observation.py:71-90, 229-262 and
// gpudrive/datatypes/roadgraph.py:329-364; constants gpudrive/env/constants.py:6-21) from one raw row of the exported tensors.
// One definition for k_pack_obs (the second pass over the exported tensors) and for the kernels that write the packed rows
// where the raw rows are produced (kernels.hip, map_obs.hip, map_obs_linear.hip), so that the two are bit-identical by
// construction.  Divisions are true IEEE divisions like torch's CPU kernels (torch's CUDA kernels multiply by the reciprocal
// of a scalar divisor, which may differ in the last bit).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gpudrive_amd.h"

namespace gd {

constexpr float kPackAgentScale = GD_VEHICLE_SCALE;  // madrona_gpudrive.vehicleScale
constexpr float kPackTwoPi = 6.283185307179586f;     // constants.MAX_ORIENTATION_RAD = 2 * np.pi

__device__ __forceinline__ float pack_norm_min_max(float x, float lo, float hi) {  // gpudrive/utils/geometry.py:15-26
    return 2.f * ((x - lo) / (hi - lo)) - 1.f;
}

// ego, env_torch.py:756-800: column c (0..5) of the packed row from the self-observation row (8 floats)
__device__ __forceinline__ float pack_ego_col(const float *self, int c) {
    switch (c) {
        case 0: return self[0] / 100.f;
        case 1: return (self[1] * kPackAgentScale) / 30.f;
        case 2: return (self[2] * kPackAgentScale) / 15.f;
        case 3: return pack_norm_min_max(self[4], -1000.f, 1000.f);
        case 4: return pack_norm_min_max(self[5], -1000.f, 1000.f);
        default: return self[6];
    }
}

// partners, env_torch.py:828-858: column c (0..5) from the raw value of the same column of the partner row (9 floats)
__device__ __forceinline__ float pack_partner_col(float x, int c) {
    const bool nm = c == 1 || c == 2;
    const float num = nm ? x - (-1000.f) : (c >= 4 ? x * kPackAgentScale : x);
    const float den = c == 0 ? 100.f : (nm ? 1000.f - (-1000.f) : (c == 3 ? kPackTwoPi : (c == 4 ? 30.f : 15.f)));
    const float q = num / den;
    return nm ? 2.f * q - 1.f : q;
}

// road points, env_torch.py:860-896: column c (0..12; 6..12 = one-hot over 7 road point types) from the raw road row (9 floats:
// x is raw column c for c < 6, the type column otherwise)
__device__ __forceinline__ float pack_road_col(float x, int c) {
    if (c >= 6) return (int)(long long)x == c - 6 ? 1.f : 0.f;
    const bool nm = c < 2;
    const float num = nm ? x - (-1000.f) : x;
    const float den = nm ? 1000.f - (-1000.f) : (c < 5 ? 100.f : kPackTwoPi);
    const float q = num / den;
    return nm ? 2.f * q - 1.f : q;
}

// the 13 packed columns of one raw road row, o[0..13)
__device__ __forceinline__ void pack_road_row(const float *raw, float *o) {
#pragma unroll
    for (int c = 0; c < 6; c++) o[c] = pack_road_col(raw[c], c);
#pragma unroll
    for (int c = 6; c < 13; c++) o[c] = pack_road_col(raw[6], c);
}
// the 6 packed columns of one raw partner row
__device__ __forceinline__ void pack_partner_row(const float *raw, float *o) {
#pragma unroll
    for (int c = 0; c < 6; c++) o[c] = pack_partner_col(raw[c], c);
}

}  // namespace gd
