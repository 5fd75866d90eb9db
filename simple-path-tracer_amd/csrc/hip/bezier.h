// CubicBezier primitive (reference src/primitive/bezier.rs, Bezier-clipping build) for the gfx950 kernels.
// Only compiled into libspt_hip_bez.so (SPT_WITH_BEZIER): the patch test needs a 16-frame subdivision stack in
// scratch memory, which the kernels of scenes without patches must not pay for (see spt_hip.hip: scenes with
// Bezier instances are handed to that library by spt_scene_create).
//
// The reference recursion (bezier.rs:239-422) is walked depth-first with an explicit stack, left half before right
// half, so candidate (u, v) pairs appear in the order of the reference's result Vec and "first strictly smaller t
// wins" (bezier.rs:121-131) selects the same one.  Arithmetic: glam 0.20 Vec2 (dot = x*x + y*y, normalize =
// v * (1 / length), Vec2 / f32 per component), everything else as in device_math.h.
#pragma once
#include "device_math.h"

#if SPT_WITH_BEZIER

struct v2 {
    float x, y;
};
SPT_DEV v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
SPT_DEV v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
SPT_DEV v2 operator*(v2 a, float s) { return {a.x * s, a.y * s}; }
SPT_DEV v2 operator/(v2 a, float s) { return {a.x / s, a.y / s}; }
SPT_DEV v2 normalize2(v2 a) { return a * (1.0f / spt_sqrt(a.x * a.x + a.y * a.y)); }
// x / 3.0f, correctly rounded, without the division sequence (24 of the 44 divisions of one clipping call divide by 3):
// q = x * rn(1/3), one fma for the exact residual, one for the correction.  Checked against x / 3.0f for ALL 2^32 bit
// patterns on the host (tools/check_div3.c): identical except for -0 (gives +0) and the infinities (NaN), which are
// passed through here.
SPT_DEV float div3(float x) {
    const float r = 0x1.555556p-2f;
    const float q = x * r;
    const float e = __builtin_fmaf(-3.0f, q, x);
    const float res = __builtin_fmaf(e, r, q);
    return (x == 0.0f || spt_abs(x) == spt_inf()) ? x : res;
}
SPT_DEV v2 div3(v2 a) { return {div3(a.x), div3(a.y)}; }

constexpr uint32_t kClippingMaxTimes = 16;   // bezier.rs:14-17
constexpr float kClippingEps = 0.00001f;

SPT_DEV void cubic_bezier_at(float u, float* b) {      // bezier.rs:206-209
    const float iu = 1.0f - u;
    b[0] = iu * iu * iu; b[1] = 3.0f * iu * iu * u; b[2] = 3.0f * u * u * iu; b[3] = u * u * u;
}
SPT_DEV void cubic_bezier_du_at(float u, float* b) {   // bezier.rs:211-219
    const float iu = 1.0f - u;
    b[0] = -3.0f * iu * iu;
    b[1] = 3.0f * iu * iu - 6.0f * iu * u;
    b[2] = 6.0f * u * iu - 3.0f * u * u;
    b[3] = 3.0f * u * u;
}
// bezier.rs:222-236: cp[4 * i + j] = control_points[i][j]
SPT_DEV f3 cubic_bezier_sum(const float4* cp, const float* bu, const float* bv) {
    f3 result = mk3(0, 0, 0);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) result = result + mk3(cp[4 * i + j]) * (bu[j] * bv[i]);
    return result;
}
SPT_DEV f3 bezier_point_at(const float4* cp, float u, float v) {
    float bu[4], bv[4];
    cubic_bezier_at(u, bu); cubic_bezier_at(v, bv);
    return cubic_bezier_sum(cp, bu, bv);
}
SPT_DEV f3 bezier_tangent_at(const float4* cp, float u, float v) {
    float bu[4], bv[4];
    cubic_bezier_du_at(u, bu); cubic_bezier_at(v, bv);
    return cubic_bezier_sum(cp, bu, bv);
}
SPT_DEV f3 bezier_bitangent_at(const float4* cp, float u, float v) {
    float bu[4], bv[4];
    cubic_bezier_at(u, bu); cubic_bezier_du_at(v, bv);
    return cubic_bezier_sum(cp, bu, bv);
}
SPT_DEV void clip_bezier_by(const v2* pt, float u_min, float u_max, v2* out) {     // bezier.rs:425-455
    float b[4];
    cubic_bezier_at(u_min, b);
    const v2 p_min = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(u_min, b);
    v2 d_min = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_min = d_min * (u_max - u_min);
    cubic_bezier_at(u_max, b);
    const v2 p_max = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(u_max, b);
    v2 d_max = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_max = d_max * (u_max - u_min);
    out[0] = p_min; out[1] = p_min + div3(d_min); out[2] = p_max - div3(d_max); out[3] = p_max;
}
SPT_DEV void clip_bezier_at_midpoint(const v2* pt, v2* l, v2* r) {                // bezier.rs:458-485
    float b[4];
    cubic_bezier_at(0.5f, b);
    const v2 p_mid = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    cubic_bezier_du_at(0.5f, b);
    v2 d_mid = ((pt[0] * b[0] + pt[1] * b[1]) + pt[2] * b[2]) + pt[3] * b[3];
    d_mid = div3(d_mid * 0.5f);
    l[0] = pt[0]; l[1] = (pt[0] + pt[1]) * 0.5f; l[2] = p_mid - d_mid; l[3] = p_mid;
    r[0] = p_mid; r[1] = p_mid + d_mid; r[2] = (pt[2] + pt[3]) * 0.5f; r[3] = pt[3];
}

// one call of bezier_clipping: its arguments
struct BzFrame {
    v2 p[4][4];
    v2 lu, lv;
    float au0, au1, av0, av1;   // affine_u, affine_v
    float calc;                 // `calculated` when has_calc
    uint32_t real_u, has_calc, times;
};

// CubicBezier::intersect_ray (bezier.rs:105-134): (u, v, t) of the nearest accepted candidate.  Inlined into its three
// call sites: 239 -> 173 ms on t_bezier.json against a __noinline__ call (measured).  A conservative early-out against the
// padded box of the control points was tried and measured no gain (173.8 ms): the time goes into the clipping of the rays
// that do reach the patch, run by a few lanes per wave, not into rays that miss its hull.
// CubicBezier::intersect_ray of the `bezier_ni` build (bezier.rs:58-103): Newton's iteration on (t, u, v) from the middle of
// the patch and of the ray's stretch inside the patch's bounding box (Bbox::intersect_ray, bbox.rs:63-85: IEEE divisions,
// NaN-ignoring min / max), at most 16 steps, accepted when the ray point and the patch point are closer than sqrt(1e-9).
constexpr uint32_t kNewtonMaxTimes = 16;
constexpr float kNewtonEps = 0.000000001f;
SPT_DEV bool bezier_intersect_ray_newton(const float4* cp, const DRay& ray, float* u_out, float* v_out, float* t_out) {
    f3 lo = mk3(cp[0]), hi = lo;      // CubicBezier::new (bezier.rs:26-38): the box of the control points
    for (int k = 1; k < 16; ++k) {
        const f3 p = mk3(cp[k]);
        lo = mk3(spt_min(lo.x, p.x), spt_min(lo.y, p.y), spt_min(lo.z, p.z));
        hi = mk3(spt_max(hi.x, p.x), spt_max(hi.y, p.y), spt_max(hi.z, p.z));
    }
    float x0 = (lo.x - ray.o.x) / ray.d.x, x1 = (hi.x - ray.o.x) / ray.d.x;
    float y0 = (lo.y - ray.o.y) / ray.d.y, y1 = (hi.y - ray.o.y) / ray.d.y;
    float z0 = (lo.z - ray.o.z) / ray.d.z, z1 = (hi.z - ray.o.z) / ray.d.z;
    const float t0 = spt_max(spt_min(x0, x1), spt_max(spt_min(y0, y1), spt_min(z0, z1)));
    const float t1 = spt_min(spt_max(x0, x1), spt_min(spt_max(y0, y1), spt_max(z0, z1)));
    if (!(t0 <= t1)) return false;
    float t = 0.5f * (t0 + t1), u = 0.5f, v = 0.5f;
    for (uint32_t it = 0; it < kNewtonMaxTimes; ++it) {
        const f3 point = bezier_point_at(cp, u, v);
        const f3 diff = (ray.o + ray.d * t) - point;
        if (!spt_is_finite(t) || !spt_is_finite(u) || !spt_is_finite(v)) break;
        if (dot(diff, diff) < kNewtonEps) {
            if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f && t > ray.t_min) {
                *u_out = u; *v_out = v; *t_out = t;
                return true;
            }
            break;
        }
        const f3 dpdu = bezier_tangent_at(cp, u, v), dpdv = bezier_bitangent_at(cp, u, v);
        const f3 n = cross(dpdu, dpdv);
        float det = dot(ray.d, n);
        if (det == 0.0f) break;
        det = 1.0f / det;
        const float dt = dot(diff, n) * det;
        const f3 q = cross(ray.d, diff);
        const float du = -dot(dpdv, q) * det;
        const float dv = dot(dpdu, q) * det;
        t -= dt; u -= du; v -= dv;
    }
    return false;
}

SPT_DEV bool bezier_intersect_ray(const float4* cp_global, const DRay& ray, float* u_out, float* v_out, float* t_out) {
    if (cp_global[0].w != 0.0f) return bezier_intersect_ray_newton(cp_global, ray, u_out, v_out, t_out);   // SPT_BEZIER_NEWTON (ABI v12)
    float4 cp[16];
    for (int k = 0; k < 16; ++k) cp[k] = cp_global[k];
    const f3 n1 = normalize(mk3(-ray.d.y, ray.d.x, 0.0f));
    const f3 n2 = normalize(mk3(0.0f, -ray.d.z, ray.d.y));
    BzFrame cur;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const f3 diff = mk3(cp[4 * i + j]) - ray.o;
            cur.p[i][j] = v2{dot(diff, n1), dot(diff, n2)};
        }
    cur.lu = normalize2((cur.p[3][0] - cur.p[0][0]) + (cur.p[3][3] - cur.p[0][3]));
    cur.lv = normalize2((cur.p[0][3] - cur.p[0][0]) + (cur.p[3][3] - cur.p[3][0]));
    cur.au0 = 1.0f; cur.au1 = 0.0f; cur.av0 = 1.0f; cur.av1 = 0.0f;
    cur.calc = 0.0f; cur.real_u = 1u; cur.has_calc = 0u; cur.times = 0u;

    BzFrame stack[kClippingMaxTimes];   // one pending right half per level at most
    uint32_t sp = 0;
    float best_t = SPT_F32_MAX;
    bool found = false;
    auto candidate = [&](float u, float v) {   // bezier.rs:121-131
        const f3 p = bezier_point_at(cp, u, v);
        const f3 diff = p - ray.o;
        const f3 c = cross(diff, ray.d);
        if (dot(c, c) < kClippingEps) {
            const float t = dot(diff, ray.d) / dot(ray.d, ray.d);
            if (t > ray.t_min && t < best_t) {
                best_t = t;
                *u_out = u; *v_out = v; *t_out = t;
                found = true;
            }
        }
    };
    for (;;) {
        bool descend = false;   // true: `cur` was replaced by a child and is processed next
        if (cur.times == kClippingMaxTimes) {
            const float u = 0.5f * cur.au0 + cur.au1;
            const float v = cur.has_calc ? cur.calc : 0.5f * cur.av0 + cur.av1;
            if (cur.real_u) candidate(u, v); else candidate(v, u);
        } else {
            float upper[4], lower[4];
            for (int j = 0; j < 4; ++j) {
                upper[j] = 0.0f; lower[j] = 0.0f;
                for (int i = 0; i < 4; ++i) {
                    const float dist = cur.p[i][j].x * cur.lu.y - cur.p[i][j].y * cur.lu.x;
                    if (i == 0 || dist > upper[j]) upper[j] = dist;
                    if (i == 0 || dist < lower[j]) lower[j] = dist;
                }
            }
            float u_min = (upper[0] >= 0.0f && lower[0] <= 0.0f) ? 0.0f : 1.0f;
            float u_max = (upper[3] >= 0.0f && lower[3] <= 0.0f) ? 1.0f : 0.0f;
            for (int a = 0; a < 3; ++a)
                for (int b = a + 1; b < 4; ++b) {   // pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
                    if (upper[a] * upper[b] <= 0.0f) {
                        const float diff = upper[b] - upper[a];
                        if (diff == 0.0f) {
                            u_min = spt_min(u_min, (float)a / 3.0f);
                            u_max = spt_max(u_max, (float)b / 3.0f);
                        } else {
                            const float k = (float)(b - a) / 3.0f / diff;
                            const float c = (float)a / 3.0f - k * upper[a];
                            u_min = spt_min(u_min, c);
                            u_max = spt_max(u_max, c);
                        }
                    }
                    if (lower[a] * lower[b] <= 0.0f) {
                        const float diff = lower[b] - lower[a];
                        if (diff == 0.0f) {
                            u_min = spt_min(u_min, (float)a / 3.0f);
                            u_max = spt_max(u_max, (float)b / 3.0f);
                        } else {
                            const float k = (float)(b - a) / 3.0f / diff;
                            const float c = (float)b / 3.0f - k * lower[b];
                            u_min = spt_min(u_min, c);
                            u_max = spt_max(u_max, c);
                        }
                    }
                }
            if (!(u_max < u_min)) {
                const bool swap = cur.has_calc == 0u;
                if (u_max - u_min > 0.8f) {
                    // both halves: the right one waits on the stack
                    v2 l[4][4], r[4][4];
                    for (int k = 0; k < 4; ++k) clip_bezier_at_midpoint(cur.p[k], l[k], r[k]);
                    BzFrame& right = stack[sp++];
                    const float half = cur.au0 * 0.5f;
                    if (swap) {
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) { right.p[a][b] = r[b][a]; }
                        right.lu = cur.lv; right.lv = cur.lu;
                        right.au0 = cur.av0; right.au1 = cur.av1; right.av0 = half; right.av1 = half + cur.au1;
                        right.real_u = cur.real_u ^ 1u; right.has_calc = 0u; right.calc = 0.0f;
                        right.times = cur.times + 1u;
                        BzFrame left;
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) left.p[a][b] = l[b][a];
                        left.lu = cur.lv; left.lv = cur.lu;
                        left.au0 = cur.av0; left.au1 = cur.av1; left.av0 = half; left.av1 = cur.au1;
                        left.real_u = cur.real_u ^ 1u; left.has_calc = 0u; left.calc = 0.0f;
                        left.times = cur.times + 1u;
                        cur = left;
                    } else {
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) right.p[a][b] = r[a][b];
                        right.lu = cur.lu; right.lv = cur.lv;
                        right.au0 = half; right.au1 = half + cur.au1; right.av0 = cur.av0; right.av1 = cur.av1;
                        right.real_u = cur.real_u; right.has_calc = cur.has_calc; right.calc = cur.calc;
                        right.times = cur.times + 1u;
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) cur.p[a][b] = l[a][b];
                        cur.au0 = half;
                        cur.times += 1u;
                    }
                    descend = true;
                } else {
                    const float u_len = u_max - u_min;
                    const bool stop = u_len * cur.au0 < kClippingEps;
                    bool finished = false;
                    if (stop) {
                        const float u = 0.5f * (u_max + u_min) * cur.au0 + cur.au1;
                        if (cur.has_calc) {
                            if (cur.real_u) candidate(u, cur.calc); else candidate(cur.calc, u);
                            finished = true;
                        } else {
                            cur.has_calc = 1u;
                            cur.calc = u;
                        }
                    }
                    if (!finished) {
                        v2 n[4][4];
                        for (int k = 0; k < 4; ++k) clip_bezier_by(cur.p[k], u_min, u_max, n[k]);
                        const float na0 = cur.au0 * u_len, na1 = cur.au0 * u_min + cur.au1;
                        if (swap) {
                            for (int a = 0; a < 4; ++a)
                                for (int b = 0; b < 4; ++b) cur.p[a][b] = n[b][a];
                            const v2 t = cur.lu; cur.lu = cur.lv; cur.lv = t;
                            cur.au0 = cur.av0; cur.au1 = cur.av1; cur.av0 = na0; cur.av1 = na1;
                            cur.real_u ^= 1u;
                        } else {
                            for (int a = 0; a < 4; ++a)
                                for (int b = 0; b < 4; ++b) cur.p[a][b] = n[a][b];
                            cur.au0 = na0; cur.au1 = na1;
                        }
                        cur.times += 1u;
                        descend = true;
                    }
                }
            }
        }
        if (descend) continue;
        if (sp == 0u) break;
        cur = stack[--sp];
    }
    return found;
}

// The same test as a resumable walk, for the pair kernel of bezier_pairs.h ONLY (opt-in, SPT_BEZ_DEFER=1): the walkers keep
// the function above, which the compiler turns into faster code (t_catmull.json: k_primary 27 ms against 52 ms through this
// struct).  Identical arithmetic, call for call; tests/test_gpu_baseline_configs.py runs both against the oracle.
// begin() projects the control points, every step()
// is one call of the reference's bezier_clipping (depth-first, the pending right halves on `stack`), done() says when the
// recursion has unwound; (u, v, t) of the nearest accepted candidate are left in the struct.
// Why resumable: the clipping needs ~4 calls for most rays and 30 - 40 for a few, so a wave that runs it to the end inside
// a traversal step waits for its slowest lane at EVERY patch; as one more phase of the stepping Walker (trace.h) a lane
// whose test is over goes on with its traversal while the others still clip.
// The 16-frame stack lives in the caller's scratch memory and is passed to step(): a struct that holds its own array is
// moved to scratch as a whole (measured for the traversal walker: 2.4x slower).
struct BzWalk {
    BzFrame cur;
    const float4* cp;
    f3 ro, rd;
    float t_min, best_t, u, v, t;
    uint32_t sp;
    bool found, finished;

    SPT_DEV void candidate(float cu, float cv) {   // bezier.rs:121-131
        const f3 p = bezier_point_at(cp, cu, cv);
        const f3 diff = p - ro;
        const f3 c = cross(diff, rd);
        if (dot(c, c) < kClippingEps) {
            const float tt = dot(diff, rd) / dot(rd, rd);
            if (tt > t_min && tt < best_t) {
                best_t = tt;
                u = cu; v = cv; t = tt;
                found = true;
            }
        }
    }
    SPT_DEV void begin(const float4* cp_global, const DRay& ray) {
        cp = cp_global;
        ro = ray.o; rd = ray.d; t_min = ray.t_min;
        const f3 n1 = normalize(mk3(-ray.d.y, ray.d.x, 0.0f));
        const f3 n2 = normalize(mk3(0.0f, -ray.d.z, ray.d.y));
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const f3 diff = mk3(cp_global[4 * i + j]) - ray.o;
                cur.p[i][j] = v2{dot(diff, n1), dot(diff, n2)};
            }
        cur.lu = normalize2((cur.p[3][0] - cur.p[0][0]) + (cur.p[3][3] - cur.p[0][3]));
        cur.lv = normalize2((cur.p[0][3] - cur.p[0][0]) + (cur.p[3][3] - cur.p[3][0]));
        cur.au0 = 1.0f; cur.au1 = 0.0f; cur.av0 = 1.0f; cur.av1 = 0.0f;
        cur.calc = 0.0f; cur.real_u = 1u; cur.has_calc = 0u; cur.times = 0u;
        sp = 0u;
        best_t = SPT_F32_MAX;
        found = false;
        finished = false;
    }
    // one call of bezier_clipping (bezier.rs:239-422)
    SPT_DEV void step(BzFrame* stack) {
        bool descend = false;   // true: `cur` was replaced by a child and is processed next
        if (cur.times == kClippingMaxTimes) {
            const float u = 0.5f * cur.au0 + cur.au1;
            const float v = cur.has_calc ? cur.calc : 0.5f * cur.av0 + cur.av1;
            if (cur.real_u) candidate(u, v); else candidate(v, u);
        } else {
            float upper[4], lower[4];
            for (int j = 0; j < 4; ++j) {
                upper[j] = 0.0f; lower[j] = 0.0f;
                for (int i = 0; i < 4; ++i) {
                    const float dist = cur.p[i][j].x * cur.lu.y - cur.p[i][j].y * cur.lu.x;
                    if (i == 0 || dist > upper[j]) upper[j] = dist;
                    if (i == 0 || dist < lower[j]) lower[j] = dist;
                }
            }
            float u_min = (upper[0] >= 0.0f && lower[0] <= 0.0f) ? 0.0f : 1.0f;
            float u_max = (upper[3] >= 0.0f && lower[3] <= 0.0f) ? 1.0f : 0.0f;
            for (int a = 0; a < 3; ++a)
                for (int b = a + 1; b < 4; ++b) {   // pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
                    if (upper[a] * upper[b] <= 0.0f) {
                        const float diff = upper[b] - upper[a];
                        if (diff == 0.0f) {
                            u_min = spt_min(u_min, (float)a / 3.0f);
                            u_max = spt_max(u_max, (float)b / 3.0f);
                        } else {
                            const float k = (float)(b - a) / 3.0f / diff;
                            const float c = (float)a / 3.0f - k * upper[a];
                            u_min = spt_min(u_min, c);
                            u_max = spt_max(u_max, c);
                        }
                    }
                    if (lower[a] * lower[b] <= 0.0f) {
                        const float diff = lower[b] - lower[a];
                        if (diff == 0.0f) {
                            u_min = spt_min(u_min, (float)a / 3.0f);
                            u_max = spt_max(u_max, (float)b / 3.0f);
                        } else {
                            const float k = (float)(b - a) / 3.0f / diff;
                            const float c = (float)b / 3.0f - k * lower[b];
                            u_min = spt_min(u_min, c);
                            u_max = spt_max(u_max, c);
                        }
                    }
                }
            if (!(u_max < u_min)) {
                const bool swap = cur.has_calc == 0u;
                if (u_max - u_min > 0.8f) {
                    // both halves: the right one waits on the stack
                    v2 l[4][4], r[4][4];
                    for (int k = 0; k < 4; ++k) clip_bezier_at_midpoint(cur.p[k], l[k], r[k]);
                    BzFrame& right = stack[sp++];
                    const float half = cur.au0 * 0.5f;
                    if (swap) {
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) { right.p[a][b] = r[b][a]; }
                        right.lu = cur.lv; right.lv = cur.lu;
                        right.au0 = cur.av0; right.au1 = cur.av1; right.av0 = half; right.av1 = half + cur.au1;
                        right.real_u = cur.real_u ^ 1u; right.has_calc = 0u; right.calc = 0.0f;
                        right.times = cur.times + 1u;
                        BzFrame left;
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) left.p[a][b] = l[b][a];
                        left.lu = cur.lv; left.lv = cur.lu;
                        left.au0 = cur.av0; left.au1 = cur.av1; left.av0 = half; left.av1 = cur.au1;
                        left.real_u = cur.real_u ^ 1u; left.has_calc = 0u; left.calc = 0.0f;
                        left.times = cur.times + 1u;
                        cur = left;
                    } else {
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) right.p[a][b] = r[a][b];
                        right.lu = cur.lu; right.lv = cur.lv;
                        right.au0 = half; right.au1 = half + cur.au1; right.av0 = cur.av0; right.av1 = cur.av1;
                        right.real_u = cur.real_u; right.has_calc = cur.has_calc; right.calc = cur.calc;
                        right.times = cur.times + 1u;
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) cur.p[a][b] = l[a][b];
                        cur.au0 = half;
                        cur.times += 1u;
                    }
                    descend = true;
                } else {
                    const float u_len = u_max - u_min;
                    const bool stop = u_len * cur.au0 < kClippingEps;
                    bool finished = false;
                    if (stop) {
                        const float u = 0.5f * (u_max + u_min) * cur.au0 + cur.au1;
                        if (cur.has_calc) {
                            if (cur.real_u) candidate(u, cur.calc); else candidate(cur.calc, u);
                            finished = true;
                        } else {
                            cur.has_calc = 1u;
                            cur.calc = u;
                        }
                    }
                    if (!finished) {
                        v2 n[4][4];
                        for (int k = 0; k < 4; ++k) clip_bezier_by(cur.p[k], u_min, u_max, n[k]);
                        const float na0 = cur.au0 * u_len, na1 = cur.au0 * u_min + cur.au1;
                        if (swap) {
                            for (int a = 0; a < 4; ++a)
                                for (int b = 0; b < 4; ++b) cur.p[a][b] = n[b][a];
                            const v2 t = cur.lu; cur.lu = cur.lv; cur.lv = t;
                            cur.au0 = cur.av0; cur.au1 = cur.av1; cur.av0 = na0; cur.av1 = na1;
                            cur.real_u ^= 1u;
                        } else {
                            for (int a = 0; a < 4; ++a)
                                for (int b = 0; b < 4; ++b) cur.p[a][b] = n[a][b];
                            cur.au0 = na0; cur.au1 = na1;
                        }
                        cur.times += 1u;
                        descend = true;
                    }
                }
            }
        }
        if (descend) return;
        if (sp == 0u) { finished = true; return; }
        cur = stack[--sp];
    }
};


#endif  // SPT_WITH_BEZIER
