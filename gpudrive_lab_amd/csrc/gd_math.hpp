// Host/device math shared by the world builder (host) and the HIP kernels (device).
//
// The quaternion/vector primitives come from the reference's absent Madrona submodule
// (math.hpp); they are written here from their published definitions, in the operation order
// the reference's call sites imply (reference src/utils.hpp:20-65, src/dynamics.hpp, src/obb.hpp).
// Everything is compiled with -ffp-contract=off so that the host build and the gfx950 build
// evaluate the same IEEE operations (the reference's x86-64 CPU build has no FMA contraction).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GD_HD __host__ __device__ __forceinline__
#else
#define GD_HD inline
#endif

namespace gd {

constexpr float kPi = 3.14159265358979323846f;  // madrona::math::pi
constexpr float kPiM2 = kPi * 2.f;              // madrona::math::pi_m2
constexpr float kPadX = -11000.f, kPadY = -11000.f, kPadZ = FLT_MAX;  // reference src/consts.hpp:64-65

enum : int { ET_None = 0, ET_RoadEdge, ET_RoadLine, ET_RoadLane, ET_CrossWalk, ET_SpeedBump, ET_StopSign,
             ET_Vehicle, ET_Pedestrian, ET_Cyclist, ET_Padding };
enum : int { RESP_Dynamic = 0, RESP_Kinematic = 1, RESP_Static = 2 };

// Transcendentals of per-agent state (pose, dynamics, agent boxes).  The reference evaluates them
// with the host libm in float.  On the device they are evaluated in double and rounded once
// (fp64 is cheap on CDNA4 and there are only O(agents) such calls per step): the result is the
// correctly rounded float, which is what glibc's sinf/cosf return in all but rare cases, whereas
// OCML's float versions differ from glibc by 1-2 ulp far more often.  One ulp of a quaternion
// component moves an egocentric coordinate at 50 m by ~1e-5, the whole parity budget.
// Observation-row headings keep the float atan2f (their error, <1e-6 rad, is harmless).
#if defined(__HIP_DEVICE_COMPILE__)
GD_HD float p_sin(float x) { return (float)sin((double)x); }
GD_HD float p_cos(float x) { return (float)cos((double)x); }
GD_HD float p_tan(float x) { return (float)tan((double)x); }
GD_HD float p_atan(float x) { return (float)atan((double)x); }
GD_HD float p_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
// sine and cosine of one angle share the argument reduction (half the instructions of two calls)
GD_HD void p_sincos(float x, float &s, float &c) {
    double sd, cd;
    sincos((double)x, &sd, &cd);
    s = (float)sd;
    c = (float)cd;
}
#else
GD_HD float p_sin(float x) { return sinf(x); }
GD_HD float p_cos(float x) { return cosf(x); }
GD_HD float p_tan(float x) { return tanf(x); }
GD_HD float p_atan(float x) { return atanf(x); }
GD_HD float p_atan2(float y, float x) { return atan2f(y, x); }
GD_HD void p_sincos(float x, float &s, float &c) { s = sinf(x); c = cosf(x); }
#endif

struct Quat { float w, x, y, z; };
struct V3 { float x, y, z; };
struct V2 { float x, y; };

// Quat::angleAxis(a, up): {cos(a/2), up * sin(a/2)} with up = (0,0,1)
GD_HD Quat quat_yaw(float a) {
    float c = p_cos(a / 2.f), s = p_sin(a / 2.f);
    return Quat{c, 0.f * s, 0.f * s, 1.f * s};
}
// Every rotation in the simulator is a yaw rotation; it is stored as (w, z) and the x/y
// components are re-derived exactly as angleAxis produced them (0 * sin(a/2): signed zeros).
GD_HD Quat quat_from_wz(float w, float z) { return Quat{w, 0.f * z, 0.f * z, z}; }
GD_HD Quat quat_inv(Quat q) { return Quat{q.w, -q.x, -q.y, -q.z}; }
GD_HD Quat quat_mul(Quat a, Quat o) {  // Hamilton product
    return Quat{(a.w * o.w - a.x * o.x - a.y * o.y - a.z * o.z),
                (a.w * o.x + a.x * o.w + a.y * o.z - a.z * o.y),
                (a.w * o.y - a.x * o.z + a.y * o.w + a.z * o.x),
                (a.w * o.z + a.x * o.y - a.y * o.x + a.z * o.w)};
}
GD_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Quat::rotateVec: v + 2 * ((pure x v) * w + pure x (pure x v))
GD_HD V3 quat_rotate(Quat q, V3 v) {
    V3 pure{q.x, q.y, q.z};
    float scalar = q.w;
    V3 pxv = cross(pure, v);
    V3 pxpxv = cross(pure, pxv);
    return V3{v.x + 2.f * ((pxv.x * scalar) + pxpxv.x), v.y + 2.f * ((pxv.y * scalar) + pxpxv.y),
              v.z + 2.f * ((pxv.z * scalar) + pxpxv.z)};
}
GD_HD float len2_2(float x, float y) { return x * x + y * y; }
GD_HD float len_2(float x, float y) { return sqrtf(x * x + y * y); }
GD_HD float len_3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// reference src/utils.hpp:11-25
GD_HD float normalize_angle(float angle) {
    const float ret = fmodf(angle, kPiM2);
    return ret > kPi ? ret - kPiM2 : (ret < -kPi ? ret + kPiM2 : ret);
}
GD_HD float angle_add(float a, float b) { return normalize_angle(a + b); }
GD_HD float quat_to_yaw(Quat q) {
    return p_atan2(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
}
// observation-row variant (hundreds per agent per step): plain float atan2f
GD_HD float quat_to_yaw_row(Quat q) {
    return atan2f(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
}

// ReferenceFrame::relative(position) (reference src/utils.hpp:51-57): ego-frame xy of an absolute point
GD_HD V2 ego_relative(float ref_x, float ref_y, Quat ref_inv, float abs_x, float abs_y) {
    V3 r = quat_rotate(ref_inv, V3{abs_x - ref_x, abs_y - ref_y, 0.f});
    return V2{r.x, r.y};
}

// Squared ego-frame distance of an absolute point for a yaw-only frame: the value of
// len2_2(ego_relative(...)) with the terms that multiply the quaternion's zero x/y components
// dropped.  With pure = (0, 0, qz) and v = (vx, vy, 0) the general rotateVec reduces, operation
// for operation, to  r.x = vx - 2 (qz vy w + qz (qz vx)),  r.y = vy + 2 (qz vx w - qz (qz vy));
// a dropped term only ever adds a zero, so every finite result is identical (the sign of a zero
// result may differ, which a squared distance cannot see).  2*A is exact, so the final
// multiply-add is written as one fma.  `qw`, `qz` are the components of the INVERSE rotation.
// The same reduction for the rotated vector itself: rotateVec of (vx, vy, 0) by the yaw-only quaternion
// (qw, 0, 0, qz).  Identical to quat_rotate for every finite input except for the sign of a zero.
GD_HD V2 rotate_yaw(float qw, float qz, float vx, float vy) {
    const float t = qz * vy, u = qz * vx;
    const float A = t * qw + qz * u;
    const float B = u * qw - qz * t;
    return V2{__builtin_fmaf(-2.f, A, vx), __builtin_fmaf(2.f, B, vy)};
}
GD_HD float ego_dist2(float ref_x, float ref_y, float qw, float qz, float abs_x, float abs_y) {
    const float vx = abs_x - ref_x, vy = abs_y - ref_y;
    const float t = qz * vy, u = qz * vx;
    const float A = t * qw + qz * u;
    const float B = u * qw - qz * t;
    const float rx = __builtin_fmaf(-2.f, A, vx), ry = __builtin_fmaf(2.f, B, vy);
    return rx * rx + ry * ry;
}

// 2-D oriented box, reference src/obb.hpp:12-50.  14 floats.
struct Obb {
    float cx[4], cy[4];  // corners
    float ax[2], ay[2];  // axes scaled by 1/len^2
    float origin[2];
};
GD_HD Obb obb_from_yaw(float px, float py, float theta, float d0, float d1);
GD_HD Obb obb_from(float px, float py, Quat rot, float d0, float d1) { return obb_from_yaw(px, py, quat_to_yaw(rot), d0, d1); }
// ... with theta = quat_to_yaw(rot) already at hand
GD_HD Obb obb_from_yaw(float px, float py, float theta, float d0, float d1) {
    float sn, cs;
    p_sincos(theta, sn, cs);
    float Xx = cs, Xy = sn;
    float Yx = -sn, Yy = cs;
    Xx *= d0; Xy *= d0;
    Yx *= d1; Yy *= d1;
    Obb o;
    o.cx[0] = px - Xx - Yx; o.cy[0] = py - Xy - Yy;
    o.cx[1] = px + Xx - Yx; o.cy[1] = py + Xy - Yy;
    o.cx[2] = px + Xx + Yx; o.cy[2] = py + Xy + Yy;
    o.cx[3] = px - Xx + Yx; o.cy[3] = py - Xy + Yy;
    o.ax[0] = o.cx[1] - o.cx[0]; o.ay[0] = o.cy[1] - o.cy[0];
    o.ax[1] = o.cx[3] - o.cx[0]; o.ay[1] = o.cy[3] - o.cy[0];
    for (int a = 0; a < 2; ++a) {
        float inv = 1.f / len2_2(o.ax[a], o.ay[a]);  // Vector2::operator/= multiplies by the reciprocal
        o.ax[a] *= inv; o.ay[a] *= inv;
        o.origin[a] = o.cx[0] * o.ax[a] + o.cy[0] * o.ay[a];
    }
    return o;
}
// reference src/obb.hpp:51-82: strict separating test; touching boxes collide.
GD_HD bool obb_overlaps(const Obb &self, const Obb &other) {
    for (int a = 0; a < 2; ++a) {
        float t = other.cx[0] * self.ax[a] + other.cy[0] * self.ay[a];
        float tMin = t, tMax = t;
        for (int c = 1; c < 4; ++c) {
            t = other.cx[c] * self.ax[a] + other.cy[c] * self.ay[a];
            if (t < tMin) tMin = t;
            else if (t > tMax) tMax = t;
        }
        if ((tMin > 1 + self.origin[a]) || (tMax < self.origin[a])) return false;
    }
    return true;
}
GD_HD bool obb_collided(const Obb &a, const Obb &b) { return obb_overlaps(a, b) && obb_overlaps(b, a); }

// reference src/sim.hpp:88-102 `collisionPairs`: type pairs whose overlap is NOT a collision.
GD_HD bool collision_pair_filtered(int a, int b) {
    if (a > b) { int t = a; a = b; b = t; }
    if (a == ET_None && b == ET_None) return true;  // the six value-initialised tail entries
    if (b == ET_Pedestrian || b == ET_Cyclist) return a >= ET_RoadEdge && a <= ET_SpeedBump;
    if (b == ET_Vehicle) return a == ET_CrossWalk || a == ET_SpeedBump || a == ET_RoadLine || a == ET_RoadLane;
    return false;
}

}  // namespace gd
