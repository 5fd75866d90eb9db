/*
 * spt_detmath.h — deterministic scalar f32 building blocks shared by the HIP
 * kernels and by the CPU oracle: elementary functions, the per-path random
 * number generator and the pixel-sampler closed form.
 *
 * Why this file exists.  The reference (Rust) calls the platform libm through
 * f32::sin_cos / ln / exp / acos / atan2 (e.g. src/core/rng.rs:72-80,
 * src/medium/homogeneous.rs:47-49, src/light/environment.rs:111-135) and seeds a
 * Xoshiro generator from OS entropy (src/core/rng.rs:8-12).  Neither is
 * reproducible across CPU and GPU.  This header pins both as a *specification*:
 * every function below uses only IEEE-754 f32 +,-,*,/,sqrt, comparisons, integer
 * ops and bit casts, in a fixed order, so a build with FP contraction disabled
 * (-ffp-contract=off, no fast-math) returns identical bits on x86-64 and gfx950.
 * That makes the GPU-vs-oracle parity tests bit-exact instead of statistical.
 *
 * Accuracy vs libm is pinned separately by tests/test_detmath.py (<= 2 ulp over
 * the argument ranges the path tracer uses).
 *
 * Everything here is header-only, C++11, no dependencies.
 */
#ifndef SPT_DETMATH_H
#define SPT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SPT_HD __host__ __device__ inline
#else
#define SPT_HD inline
#endif

SPT_HD uint32_t spt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
SPT_HD float spt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

#define SPT_PI 3.14159265358979323846f
#define SPT_FRAC_1_PI 0.318309886183790671538f
#define SPT_F32_MAX 3.40282346638528859812e+38f

SPT_HD float spt_inf() { return spt_u2f(0x7f800000u); }

SPT_HD float spt_abs(float x) { return spt_u2f(spt_f2u(x) & 0x7fffffffu); }

/* Rust f32::is_finite */
SPT_HD bool spt_is_finite(float x) { return (spt_f2u(x) & 0x7f800000u) != 0x7f800000u; }

/* Rust f32::max / f32::min: a NaN operand is ignored (IEEE maxNum/minNum).
 * On gfx950 this is one v_max_f32 / v_min_f32.  The two sides may differ in the
 * SIGN of a zero result (max(-0,+0)); no caller lets that sign reach a division
 * or an output, only comparisons and sums, so results stay bit-identical. */
#if defined(__HIP_DEVICE_COMPILE__)
SPT_HD float spt_max(float a, float b) { return __builtin_fmaxf(a, b); }
SPT_HD float spt_min(float a, float b) { return __builtin_fminf(a, b); }
#else
SPT_HD float spt_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
SPT_HD float spt_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
#endif

/* Rust f32::clamp(lo, hi): NaN passes through. */
SPT_HD float spt_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* Rust `f as usize` / `as i32`: saturating, NaN -> 0 (only the ranges we need). */
SPT_HD uint32_t spt_f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u; /* NaN, negatives, zero */
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
SPT_HD int32_t spt_f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}

/* floor; values of magnitude >= 2^23 are integral already, inf and NaN pass through (as f32::floor) */
SPT_HD float spt_floor(float x) {
    if (!(spt_abs(x) < 8388608.0f)) return x;
    float t = (float)(int32_t)x;
    return (t > x) ? t - 1.0f : t;
}

/* Rust f32::round: half away from zero; NaN stays NaN. */
SPT_HD float spt_round(float x) {
    float a = spt_abs(x);
    float f = spt_floor(a);
    float r = (a - f >= 0.5f) ? f + 1.0f : f; /* a - f is exact; a + 0.5f could round up */
    return (x < 0.0f) ? -r : r;
}

/* ---- sin / cos ------------------------------------------------------------
 * Cody-Waite reduction to [-pi/4, pi/4] with a three-part pi/2, then the
 * classic single-precision minimax kernels.  Valid for |x| <= ~1e4 (callers
 * pass angles in [0, 2pi]). */
SPT_HD void spt_sincos(float x, float* s_out, float* c_out) {
    float kf = spt_floor(x * 0.636619772367581343076f + 0.5f);
    int32_t k = (int32_t)kf;
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.549790126404332e-8f;
    float z = r * r;
    float sp = -1.9515295891e-4f;
    sp = sp * z + 8.3321608736e-3f;
    sp = sp * z - 1.6666654611e-1f;
    float s = r + r * z * sp;
    float cp = 2.443315711809948e-5f;
    cp = cp * z - 1.388731625493765e-3f;
    cp = cp * z + 4.166664568298827e-2f;
    float c = (1.0f - 0.5f * z) + z * z * cp;
    float ss, cc;
    switch (k & 3) {
    case 0: ss = s; cc = c; break;
    case 1: ss = c; cc = -s; break;
    case 2: ss = -s; cc = -c; break;
    default: ss = -c; cc = s; break;
    }
    *s_out = ss;
    *c_out = cc;
}
SPT_HD float spt_sin(float x) { float s, c; spt_sincos(x, &s, &c); return s; }
SPT_HD float spt_cos(float x) { float s, c; spt_sincos(x, &s, &c); return c; }

/* ---- log (natural) ---------------------------------------------------------
 * x > 0 normal; x == 0 -> -inf; x < 0 or NaN -> NaN. */
SPT_HD float spt_log(float x) {
    if (x != x || x < 0.0f) return spt_u2f(0x7fc00000u);
    if (x == 0.0f) return -spt_inf();
    uint32_t ux = spt_f2u(x);
    int32_t e = 0;
    if (ux < 0x00800000u) { /* subnormal: scale up */
        x = x * 8388608.0f;
        ux = spt_f2u(x);
        e = -23;
    }
    if (ux >= 0x7f800000u) return x; /* +inf */
    e += (int32_t)(ux >> 23) - 126;
    float m = spt_u2f((ux & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = p * m - 1.1514610310e-1f;
    p = p * m + 1.1676998740e-1f;
    p = p * m - 1.2420140846e-1f;
    p = p * m + 1.4249322787e-1f;
    p = p * m - 1.6668057665e-1f;
    p = p * m + 2.0000714765e-1f;
    p = p * m - 2.4999993993e-1f;
    p = p * m + 3.3333331174e-1f;
    float y = m * z * p;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    float r = m + y;
    r = r + fe * 0.693359375f;
    return r;
}

/* ---- exp --------------------------------------------------------------------
 * Flushes to 0 below -87.3 and saturates to +inf above 88.7. */
SPT_HD float spt_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return spt_inf();
    if (x < -87.3f) return 0.0f;
    float nf = spt_floor(x * 1.44269504088896341f + 0.5f);
    int32_t n = (int32_t)nf;
    float r = x - nf * 0.693359375f;
    r = r - nf * -2.12194440e-4f;
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float y = (p * z + r) + 1.0f;
    /* scale by 2^n in two steps so n in [-126, 128] stays in range */
    int32_t n1 = n / 2, n2 = n - n1;
    y = y * spt_u2f((uint32_t)(n1 + 127) << 23);
    y = y * spt_u2f((uint32_t)(n2 + 127) << 23);
    return y;
}

/* ---- pow / log2 / trunc (image textures: sRGB decode, mip level, wrap modes) ---
 * pow(x, y) for x >= 0 as exp(y * ln x): within ~1e-6 relative of a correctly rounded powf
 * for the exponents used here (2.4), and the same bits on host and device. */
SPT_HD float spt_pow(float x, float y) {
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : spt_inf());
    return spt_exp(y * spt_log(x));
}
SPT_HD float spt_log2(float x) { return spt_log(x) * 1.44269504088896341f; }
SPT_HD float spt_trunc(float x) {
    if (!(spt_abs(x) < 8388608.0f)) return x; /* already integral, inf or NaN */
    float t = (float)(int32_t)x;
    return (t == 0.0f) ? spt_u2f(spt_f2u(x) & 0x80000000u) : t; /* keep the sign of zero like Rust's trunc */
}
/* f32::fract = x - trunc(x) */
SPT_HD float spt_fract(float x) { return x - spt_trunc(x); }

/* ---- BSSRDF radius table (src/bxdf/substrate.rs:187-196, SS_CDF_TABLE) -------------
 * entry i of 512: x = -2 ln(1 - i/512), y = 1 - e^-x / 4 - 3 e^(-x/3) / 4 (the CDF of the normalised diffusion
 * profile); host and device build the table from this one definition. */
#define SPT_SS_CDF_SIZE 512
SPT_HD void spt_ss_cdf_entry(uint32_t i, float* x_out, float* y_out) {
    float x = (float)i / 512.0f;
    x = -2.0f * spt_log(1.0f - x);
    *x_out = x;
    *y_out = 1.0f - spt_exp(-x) * 0.25f - spt_exp(-x / 3.0f) * 0.75f;
}

/* ---- atan / atan2 / asin / acos --------------------------------------------- */
SPT_HD float spt_atan_pos(float x) { /* x >= 0 */
    float y;
    if (x > 2.414213562373095f) { /* tan 3pi/8 */
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan pi/8 */
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    y = y + (p * z * x + x);
    return y;
}
SPT_HD float spt_atan(float x) { return (x < 0.0f) ? -spt_atan_pos(-x) : spt_atan_pos(x); }

/* Rust y.atan2(x): angle of the point (x, y). */
SPT_HD float spt_atan2(float y, float x) {
    if (x != x || y != y) return spt_u2f(0x7fc00000u);
    bool yneg = (spt_f2u(y) >> 31) != 0;
    bool xneg = (spt_f2u(x) >> 31) != 0;
    if (y == 0.0f) {
        if (xneg) return yneg ? -SPT_PI : SPT_PI;
        return y;
    }
    if (x == 0.0f) return yneg ? -1.5707963267948966f : 1.5707963267948966f;
    float ay = spt_abs(y), ax = spt_abs(x);
    float a = spt_atan_pos(ay / ax);
    if (xneg) a = SPT_PI - a;
    return yneg ? -a : a;
}

SPT_HD float spt_sqrt(float x); /* provided per platform below */

SPT_HD float spt_asin(float x) {
    float a = spt_abs(x);
    if (a > 1.0f) return spt_u2f(0x7fc00000u);
    bool big = a > 0.5f;
    float z, r;
    if (big) {
        z = 0.5f * (1.0f - a);
        r = spt_sqrt(z);
    } else {
        r = a;
        z = r * r;
    }
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    float y = p * z * r + r;
    if (big) {
        y = y + y;
        y = 1.5707963267948966f - y;
    }
    return (x < 0.0f) ? -y : y;
}

SPT_HD float spt_acos(float x) {
    if (x != x || x > 1.0f || x < -1.0f) return spt_u2f(0x7fc00000u);
    if (x < -0.5f) return SPT_PI - 2.0f * spt_asin(spt_sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * spt_asin(spt_sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966f - spt_asin(x);
}

/* correctly rounded sqrt on both sides: x86 sqrtss; gfx950 llvm.sqrt.f32 under hipcc's
 * default -fhip-fp32-correctly-rounded-divide-sqrt.  (HIP's __fsqrt_rn is NOT usable: without
 * OCML_BASIC_ROUNDED_OPERATIONS it is the approximate native sqrt.) */
SPT_HD float spt_sqrt(float x) { return __builtin_sqrtf(x); }

/* ---- per-path random numbers --------------------------------------------------
 * Replaces SmallRng::from_entropy (src/core/rng.rs:8-12, one generator per
 * worker thread) by one PCG32 (XSH-RR 64/32) stream per camera sample, keyed by
 * (seed, pixel index, sample index); uniform_1d keeps rand 0.8's f32 recipe,
 * (u32 >> 8) * 2^-24 in [0,1) (src/core/rng.rs:14-16). */
typedef struct spt_rng { uint64_t state; } spt_rng;

SPT_HD uint64_t spt_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* pixel = j*width + i with j counted from the top row (src/renderer/pt.rs:263-264) */
SPT_HD spt_rng spt_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample) {
    spt_rng r;
    uint64_t key = ((uint64_t)pixel << 32) | (uint64_t)sample;
    r.state = spt_splitmix64(spt_splitmix64(seed) ^ key);
    return r;
}

SPT_HD uint32_t spt_rng_u32(spt_rng* r) {
    uint64_t old = r->state;
    r->state = old * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}

SPT_HD float spt_rng_f32(spt_rng* r) { return (float)(spt_rng_u32(r) >> 8) * 5.9604644775390625e-8f; }

/* ---- additive-recurrence (R2) pixel sampler, closed form -----------------------
 * src/pixel_sampler/recurrence.rs:37-56 adds alpha / alpha^2 (f32) per sample and
 * wraps at 1, carrying state from pixel to pixel inside a thread band.  Closed
 * form: sample s of pixel p is element k = p*spp + s + 1 of that sequence,
 * x = frac(0.5 + k*alpha), evaluated exactly in 24-bit fixed point with
 * alpha = f32(0.754877666246571) = 0xC13FAA * 2^-24 and
 * alpha^2 (f32 product) = 0x91E10E * 2^-24. */
SPT_HD void spt_r2_offset(uint32_t pixel, uint32_t spp, uint32_t sample, float* ox, float* oy) {
    uint32_t k = pixel * spp + sample + 1u;
    uint32_t xi = (0x800000u + k * 0xC13FAAu) & 0xFFFFFFu;
    uint32_t yi = (0x800000u + k * 0x91E10Eu) & 0xFFFFFFu;
    *ox = (float)xi * 5.9604644775390625e-8f;
    *oy = (float)yi * 5.9604644775390625e-8f;
}

#endif /* SPT_DETMATH_H */
