/*
 * mirt_oracle_math.h — scalar "mirt-math v1" for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product (weekend-raytracer-wgpu_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Why this exists: the path-traced mode restates the behaviours of the reference's WGSL shader
 * (src/raytracer/raytracer.wgsl), which calls sin/cos/acos/atan2/pow/exp/sqrt whose precision
 * WGSL leaves implementation-defined.  To make CPU and gfx950 results bit-identical, every
 * elementary function is DEFINED here as a fixed sequence of IEEE-754 binary32 operations
 * (+, -, *, /, sqrt, fma — each correctly rounded on both machines) and evaluated in exactly
 * this order.  The device code re-implements the same sequences independently
 * (csrc/mirt_device_math.h); coefficients come from tools/fit_poly.py.
 *
 * Build rule: -ffp-contract=off, no -ffast-math.  The only fused operations are the explicit
 * MFMA() calls below.
 */
#ifndef MIRT_ORACLE_MATH_H
#define MIRT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define MFMA(a, b, c) __builtin_fmaf((a), (b), (c))

/* constants as the WGSL source spells them (raytracer.wgsl:1-8) */
#define OM_EPSILON   0.001f
#define OM_PI        3.1415927f
#define OM_FRAC_1_PI 0.31830987f
#define OM_FRAC_PI_2 1.5707964f
#define OM_TWO_PI    6.2831855f   /* 2f * PI (exact doubling of the f32 constant) */
#define OM_MIN_T     0.001f
#define OM_MAX_T     1000.0f

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- polynomial coefficient tables (tools/fit_poly.py) ---- */
static const float OM_SIN[4]  = { -0x1.5555560000000p-3f, 0x1.11110e0000000p-7f, -0x1.a013a80000000p-13f, 0x1.6dbe080000000p-19f };
static const float OM_COS[4]  = { 0x1.5555560000000p-5f, -0x1.6c16c00000000p-10f, 0x1.a015c80000000p-16f, -0x1.2524f20000000p-22f };
static const float OM_ASIN[6] = { 0x1.5555540000000p-3f, 0x1.3334300000000p-4f, 0x1.6d5bba0000000p-5f, 0x1.fd8da20000000p-6f, 0x1.18f91e0000000p-6f, 0x1.13fed40000000p-5f };
static const float OM_ATAN[9] = { -0x1.5555540000000p-2f, 0x1.99983c0000000p-3f, -0x1.246cd20000000p-3f, 0x1.c3f1680000000p-4f, -0x1.6295f80000000p-4f, 0x1.0001c60000000p-4f, -0x1.25dc120000000p-5f, 0x1.ba9f660000000p-7f, -0x1.38de560000000p-9f };
static const float OM_LOG2[10] = { 0x1.7154760000000p+0f, -0x1.7154700000000p-1f, 0x1.ec70aa0000000p-2f, -0x1.715a700000000p-2f, 0x1.277a520000000p-2f, -0x1.eab7aa0000000p-3f, 0x1.a38c680000000p-3f, -0x1.87f6a20000000p-3f, 0x1.7a63a00000000p-3f, -0x1.b84fb60000000p-4f };
static const float OM_EXP2[7] = { 0x1.62e4300000000p-1f, 0x1.ebfbe00000000p-3f, 0x1.c6b08e0000000p-5f, 0x1.3b2a1c0000000p-7f, 0x1.5d879e0000000p-10f, 0x1.4440000000000p-13f, 0x1.00a5800000000p-16f };

/* Horner with one fma per step, highest power first. */
static inline float om_horner(const float* c, int n, float z)
{
    float acc = c[n - 1];
    for (int i = n - 2; i >= 0; --i) acc = MFMA(acc, z, c[i]);
    return acc;
}

/* ---- sin & cos ----
 * k  = nearest integer to x*(2/pi) by the add-magic/subtract-magic trick (|k| < 2^22)
 * r  = x - k*pi/2 with a three-part pi/2 (Cody-Waite), three fmas
 * sin(r) = r + r*z*S(z), cos(r) = 1 - z/2 + z*z*C(z), z = r*r; quadrant fix-up by k mod 4. */
#define OM_TWO_OVER_PI 0.63661975f
#define OM_RND_MAGIC   12582912.0f              /* 1.5 * 2^23 */
#define OM_PIO2_HI     1.5703125f               /* pi/2, high 8 bits */
#define OM_PIO2_MD     4.837512969970703125e-4f
#define OM_PIO2_LO     7.54978995489188216e-8f

static inline void om_sincos(float x, float* s_out, float* c_out)
{
    if (!(fabsf(x) <= 1048576.0f)) x = 0.0f;   /* outside the reduction's domain (and NaN): defined as 0 */
    float kf = (x * OM_TWO_OVER_PI + OM_RND_MAGIC) - OM_RND_MAGIC;
    int32_t q = (int32_t)kf;
    float r = MFMA(-kf, OM_PIO2_HI, x);
    r = MFMA(-kf, OM_PIO2_MD, r);
    r = MFMA(-kf, OM_PIO2_LO, r);
    float z = r * r;
    float sp = om_horner(OM_SIN, 4, z);
    float cp = om_horner(OM_COS, 4, z);
    float sr = MFMA(r * z, sp, r);
    float cr = MFMA(z * z, cp, MFMA(-0.5f, z, 1.0f));
    float s, c;
    switch (q & 3) {
        case 0:  s = sr;  c = cr;  break;
        case 1:  s = cr;  c = -sr; break;
        case 2:  s = -sr; c = -cr; break;
        default: s = -cr; c = sr;  break;
    }
    *s_out = s;
    *c_out = c;
}
static inline float om_sin(float x) { float s, c; om_sincos(x, &s, &c); return s; }

/* Sign of sin(x) in {-1, 0, +1} from the SAME argument reduction as om_sincos, without the
 * polynomials: for |r| <= pi/4, cos(r) > 0 and sin(r) has the sign of r (0 iff r == 0). */
static inline int om_sin_sign(float x)
{
    if (!(fabsf(x) <= 1048576.0f)) x = 0.0f;
    float kf = (x * OM_TWO_OVER_PI + OM_RND_MAGIC) - OM_RND_MAGIC;
    int32_t q = (int32_t)kf;
    float r = MFMA(-kf, OM_PIO2_HI, x);
    r = MFMA(-kf, OM_PIO2_MD, r);
    r = MFMA(-kf, OM_PIO2_LO, r);
    int sr = (r > 0.0f) - (r < 0.0f);
    int base = (q & 1) ? 1 : sr;            /* odd quadrant: +-cos(r), cos(r) > 0 */
    return (q & 2) ? -base : base;
}
static inline float om_cos(float x) { float s, c; om_sincos(x, &s, &c); return c; }

/* ---- asin kernel on [0, 0.5]: asin(x) = x + x*z*A(z), z = x*x ---- */
static inline float om_asin_core(float x, float z) { return MFMA(x * z, om_horner(OM_ASIN, 6, z), x); }

/* ---- acos on [-1,1]; the argument is clamped first (NaN -> treated as 1 -> 0) ---- */
static inline float om_acos(float x)
{
    if (!(x < 1.0f)) x = 1.0f;          /* also catches NaN */
    if (x < -1.0f) x = -1.0f;
    float ax = fabsf(x);
    if (ax <= 0.5f) {
        float z = x * x;
        return OM_FRAC_PI_2 - om_asin_core(x, z);
    }
    float z = (1.0f - ax) * 0.5f;
    float s = sqrtf(z);
    float t = 2.0f * om_asin_core(s, z);
    return (x > 0.0f) ? t : (OM_PI - t);
}

/* ---- atan2: one division; a = min/max in [0,1]; atan(a) = a + a*z*T(z), z = a*a ----
 * octant fix-ups: |y|>|x| -> pi/2 - r;  x<0 -> pi - r;  sign of y applied last.
 * atan2(+-0, +-0) is defined as 0.  y = -0.0 counts as non-negative. */
static inline float om_atan2(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = (ax > ay) ? ax : ay;
    float mn = (ax > ay) ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;      /* both zero (or NaN) */
    float a = mn / mx;
    float z = a * a;
    float r = MFMA(a * z, om_horner(OM_ATAN, 9, z), a);
    if (ay > ax) r = OM_FRAC_PI_2 - r;
    if (x < 0.0f) r = OM_PI - r;
    return (y < 0.0f) ? -r : r;
}

/* ---- log2 for finite x > 0 (subnormals are scaled first); x <= 0 -> -inf/NaN not needed:
 * callers guard.  m in [sqrt(1/2), sqrt(2)), log2(x) = e + f*L(f), f = m - 1 ---- */
static inline float om_log2(float x)
{
    uint32_t u = om_f2u(x);
    int32_t eadj = 0;
    if (u < 0x00800000u) {              /* subnormal: scale by 2^24 */
        x = x * 16777216.0f;
        u = om_f2u(x);
        eadj = -24;
    }
    int32_t e = (int32_t)(u >> 23) - 127;
    uint32_t mbits = (u & 0x007fffffu) | 0x3f800000u;   /* m in [1,2) */
    if (mbits >= 0x3fb504f3u) {         /* m >= sqrt(2): halve */
        mbits -= 0x00800000u;
        e += 1;
    }
    float f = om_u2f(mbits) - 1.0f;
    float p = om_horner(OM_LOG2, 10, f);
    return MFMA(f, p, (float)(e + eadj));
}

/* ---- exp2 for any finite y: n = nearest int, f = y - n in [-0.5,0.5],
 * 2^f = 1 + f*E(f); result scaled by 2^n through the exponent field.
 * y >= 128 -> +inf, y < -126 -> 0 (subnormal results are flushed by definition). ---- */
static inline float om_exp2(float y)
{
    if (!(y < 128.0f)) return INFINITY;
    if (!(y >= -126.0f)) return 0.0f;   /* also NaN -> 0 */
    float nf = (y + OM_RND_MAGIC) - OM_RND_MAGIC;
    float f = y - nf;
    int32_t n = (int32_t)nf;
    float p = MFMA(f, om_horner(OM_EXP2, 7, f), 1.0f);   /* in [0.70, 1.42] */
    /* n in [-126,128]; p*2^n: when n == 128 p < 1 is guaranteed?  no: handle by two steps */
    int32_t n1 = n >> 1, n2 = n - n1;
    float s1 = om_u2f((uint32_t)(n1 + 127) << 23);
    float s2 = om_u2f((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* pow(x, y) for x >= 0 as WGSL uses it (raytracer.wgsl:482): x == 0 -> 0 */
static inline float om_pow_pos(float x, float y)
{
    if (!(x > 0.0f)) return 0.0f;
    return om_exp2(y * om_log2(x));
}

#define OM_LOG2_E 1.44269504f
static inline float om_exp(float x) { return om_exp2(x * OM_LOG2_E); }

/* ---- 3-vectors (path-traced mode: fma-friendly definitions) ---- */
typedef struct { float x, y, z; } ov3;

static inline ov3 ov(float x, float y, float z) { ov3 r = { x, y, z }; return r; }
static inline ov3 ov_add(ov3 a, ov3 b) { return ov(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ov3 ov_sub(ov3 a, ov3 b) { return ov(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ov3 ov_scale(float s, ov3 a) { return ov(s * a.x, s * a.y, s * a.z); }
static inline ov3 ov_neg(ov3 a) { return ov(-a.x, -a.y, -a.z); }
/* s*a + b, one fma per component */
static inline ov3 ov_fma(float s, ov3 a, ov3 b) { return ov(MFMA(s, a.x, b.x), MFMA(s, a.y, b.y), MFMA(s, a.z, b.z)); }
/* PT dot: fma(az,bz, fma(ay,by, ax*bx)) */
static inline float ov_dot(ov3 a, ov3 b) { return MFMA(a.z, b.z, MFMA(a.y, b.y, a.x * b.x)); }
/* PT normalize: v * (1 / sqrt(dot(v,v))) */
static inline ov3 ov_normalize(ov3 a) { float inv = 1.0f / sqrtf(ov_dot(a, a)); return ov_scale(inv, a); }

#endif /* MIRT_ORACLE_MATH_H */
