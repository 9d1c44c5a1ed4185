// What every road-observation kernel needs to write a row of agent_roadmap_tensor: ReferenceFrame::observationOf
// (reference src/utils.hpp:36-49) of one road entity seen from one agent, or the padding row; and, for a learner that only
// reads the packed observation, the same row in the 13 normalised columns of GPUDriveTorchEnv.get_obs()
// (reference gpudrive/env/env_torch.py:860-896).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "gd_math.hpp"

namespace gd {

// Intra-wave ordering point for LDS traffic between lanes of one wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Set bits of `b` below this lane's own (v_mbcnt_lo / v_mbcnt_hi: two instructions; popc(b & lower_mask) is four).
__device__ __forceinline__ int bits_below_lane(unsigned long long b) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b, 0u));
}

// One row of agent_roadmap_tensor into o[0..9): the selected road (its 32-byte record, engine.hpp road_rec) seen from the agent
// at (ex, ey) with rotation (ew, ez), or the padding row.
__device__ __forceinline__ void road_row(float *o, bool selected, bool knn, float ex, float ey, float ew, float ez, float4 q0, float4 q1) {
    if (!selected) {
        // k-NN pads with fillZeros (id 0, mapType 0: src/knn.hpp:19-28); the linear scan pads with
        // MapObservation::zero() (id -1, mapType -1: src/sim.cpp:277-279)
        const float pad = knn ? 0.f : -1.f;
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; o[5] = 0.f; o[6] = (float)ET_None; o[7] = pad; o[8] = pad;
        return;
    }
    const unsigned int bits = __float_as_uint(q1.w);
    const int type = (int)(bits & 0xffu), map_type = (int)(bits >> 8) - 1;
    // Every rotation here is a yaw rotation (x = y = +-0): the reference's rotateVec and Hamilton product with the terms
    // that multiply those zeros dropped (gd_math.hpp rotate_yaw) -- a dropped term only ever adds a zero, so every non-zero
    // result is the same float, and a zero result may differ in its sign, which matters in one place: the heading of a road
    // exactly opposite to the agent (w z = 0: atan2f(+-0, negative) = +-pi, reference
    // tests/EgocentricRoadObservationTests.cpp), so that case keeps the full product.  A row is 110 vector instructions
    // with the general forms and 65 with these; the rows stored by the selecting waves are bound by exactly that.
    const V2 rel = rotate_yaw(ew, -ez, q0.x - ex, q0.y - ey);
    o[0] = rel.x; o[1] = rel.y; o[2] = q1.x; o[3] = q1.y;
    o[4] = type == ET_StopSign ? 1.f : 0.1f;  // the z scale of the road entity (scene.cpp put_road)
    const float rw = q0.z, rz = q0.w, iz = -ez;
    const float pw = ew * rw - iz * rz, pz = ew * rz + iz * rw;  // (w, z) of inverse(ego) * road
    const float wz = pw * pz;
    if (wz != 0.f) o[5] = atan2f(2.0f * wz, 1.0f - 2.0f * (pz * pz));
    else o[5] = quat_to_yaw_row(quat_mul(quat_inv(quat_from_wz(ew, ez)), quat_from_wz(rw, rz)));
    o[6] = (float)type; o[7] = q1.z; o[8] = (float)map_type;
}

}  // namespace gd
