// Device-side f32 vector / colour math for the gfx950 kernels.
// Operation order follows glam 0.20 as used by the reference (dot = (x*x'+y*y')+z*z',
// Mat3*v = (c0*x+c1*y)+c2*z, normalize = v / sqrt(dot)); the build disables FP
// contraction, so each expression is exactly the f32 sequence written here.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/spt_abi.h"
#include "../../../include/spt_detmath.h"

#define SPT_DEV __device__ __forceinline__

// 1 only in libspt_hip_bez.so: the CubicBezier primitive (bezier.h) in the walkers, the hit reconstruction and the
// medium probe.  The plain library never sees a scene with patches (spt_scene_create hands those over).
#ifndef SPT_WITH_BEZIER
#define SPT_WITH_BEZIER 0
#endif

struct f3 {
    float x, y, z;
};
SPT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
SPT_DEV f3 mk3(const float* p) { return f3{p[0], p[1], p[2]}; }
SPT_DEV f3 mk3(float4 v) { return f3{v.x, v.y, v.z}; }
SPT_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
SPT_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SPT_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
SPT_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
SPT_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
SPT_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // Color * Color
SPT_DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }      // Vec3A / f32 (true division)
SPT_DEV f3 cdiv(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }       // Color / Color
SPT_DEV f3 crcp(f3 a, float s) { return a * (1.0f / s); }                        // Color / f32 = * (1/s) (color.rs:145-151)
SPT_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SPT_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
SPT_DEV float length(f3 a) { return spt_sqrt(dot(a, a)); }
SPT_DEV f3 normalize(f3 a) { return a / length(a); }
SPT_DEV f3 gray(float v) { return f3{v, v, v}; }
SPT_DEV float luminance(f3 c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }
SPT_DEV float cavg(f3 c) { return (c.x + c.y + c.z) / 3.0f; }
SPT_DEV bool all_finite(f3 c) { return spt_is_finite(c.x) && spt_is_finite(c.y) && spt_is_finite(c.z); }
SPT_DEV f3 csqrt(f3 c) { return {spt_sqrt(c.x), spt_sqrt(c.y), spt_sqrt(c.z)}; }
SPT_DEV f3 cexp(f3 c) { return {spt_exp(c.x), spt_exp(c.y), spt_exp(c.z)}; }
SPT_DEV float pow2(float x) { return x * x; }

// glam Affine3A as 3 columns + translation (spt_instance::inv / fwd)
SPT_DEV f3 xf_vector(const float* m, f3 v) { return (mk3(m) * v.x + mk3(m + 3) * v.y) + mk3(m + 6) * v.z; }
SPT_DEV f3 xf_point(const float* m, f3 p) { return xf_vector(m, p) + mk3(m + 9); }

struct DRay {
    f3 o, d;
    float t_min;
};
SPT_DEV f3 point_at(const DRay& r, float t) { return r.o + r.d * t; }

constexpr float kTMinEps = 0.0001f;
