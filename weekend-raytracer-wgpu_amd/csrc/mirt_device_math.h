// mirt_device_math.h — "mirt-math v1" on gfx950.
//
// The path-traced kernel must produce bit-identical fp32 results on the device and in any
// independent CPU check, so every elementary function is a fixed sequence of correctly rounded
// IEEE binary32 operations (+ - * / sqrt fma) — no OCML, no v_sin/v_cos/v_rcp approximations.
// The translation unit is compiled with -ffp-contract=off; the only fused operations are the
// explicit __builtin_fmaf calls (v_fma_f32).  Division and sqrt use hipcc's correctly rounded
// expansions (-fhip-fp32-correctly-rounded-divide-sqrt, the default).
//
// Coefficients: tools/fit_poly.py (tests/test_math_spec.py checks this file against it).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIRT_DEV __device__ __forceinline__

// The kernels exist in two builds of the same source (mirt_kernels.hip is compiled twice):
//   exact_build  the default and the only one parity is claimed for: the bit-exact arithmetic described above;
//   fast_build   opt-in (MIRT_FLAG_FAST_MATH): hardware v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp / v_log
//                (about 1 ulp), contraction allowed, approximate division.  Same RNG streams, same integer
//                accumulation; pixels may differ from the exact build by a few units in the last place.
#ifndef MIRT_KNS
#define MIRT_KNS exact_build
#endif

namespace mirt {
namespace MIRT_KNS {

// constants exactly as the reference shader spells them (raytracer.wgsl:1-8)
constexpr float kEpsilon  = 0.001f;
constexpr float kPi       = 3.1415927f;
constexpr float kFrac1Pi  = 0.31830987f;
constexpr float kFracPi2  = 1.5707964f;
constexpr float kTwoPi    = 6.2831855f;
constexpr float kMinT     = 0.001f;
constexpr float kMaxT     = 1000.0f;

MIRT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Correctly rounded sqrt(x) and 1/x in 5 and 3 instructions instead of hipcc's ~15 / ~10:
// the hardware seed (v_rsq_f32 / v_rcp_f32, 1 ulp) plus ONE fma residual correction is already the
// correctly rounded result for every binary32 input with 2^-100 <= |x| <= 2^100 — verified
// EXHAUSTIVELY against the compiler's IEEE expansion on gfx950 (all 2^32 bit patterns,
// mirt_ctx_selftest_math / tests/test_gpu_api.py::test_fast_sqrt_rcp_are_correctly_rounded).
// Anything outside that range (0, subnormals, huge, inf, NaN) takes the IEEE expansion.
MIRT_DEV float sqrt_ieee(float a) { return __builtin_sqrtf(a); }
MIRT_DEV float rcp_ieee(float a) { return 1.0f / a; }

#ifdef MIRT_FAST_MATH
MIRT_DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
MIRT_DEV float sqrt_unit(float x) { return __builtin_amdgcn_sqrtf(x); }
MIRT_DEV float sqrt_unit_where(float x, bool) { return __builtin_amdgcn_sqrtf(x); }
MIRT_DEV float rcp_in_range(float x) { return __builtin_amdgcn_rcpf(x); }
MIRT_DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
#else
MIRT_DEV float sqrt_(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    const bool odd = !(x >= 0x1p-100f && x <= 0x1p100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0ull, 0)) {        // wave-uniform and rare
        asm volatile("; sqrt_: IEEE expansion" ::);           // keeps this a real branch (hipcc would otherwise
        s = odd ? sqrt_ieee(x) : s;                           // if-convert it and run BOTH sequences on every lane)
    }
    return s;
}

// sqrt_ for arguments known to lie in [0, 1] (a uniform variate, 1 - variate): the upper range check is moot.
MIRT_DEV float sqrt_unit(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    const bool odd = !(x >= 0x1p-100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0ull, 0)) {
        asm volatile("; sqrt_unit: IEEE expansion" ::);
        s = odd ? sqrt_ieee(x) : s;
    }
    return s;
}

// sqrt_ for callers that only use the result on lanes where `want` holds, with 0 < x <= 1 there (the
// refraction discriminant): the other lanes may carry negative or NaN arguments without sending the wave
// down the IEEE expansion.
MIRT_DEV float sqrt_unit_where(float x, bool want)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    const bool odd = want && !(x >= 0x1p-100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0ull, 0)) {
        asm volatile("; sqrt_unit_where: IEEE expansion" ::);
        s = odd ? sqrt_ieee(x) : s;
    }
    return s;
}

// 1/x for 2^-100 <= |x| <= 2^100 guaranteed by the caller: the fast sequence alone (no range check).
MIRT_DEV float rcp_in_range(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}

MIRT_DEV float rcp_(float x)           // == 1.0f / x bit for bit
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    float y = __builtin_fmaf(y0, e, y0);
    const float ax = __builtin_fabsf(x);
    const bool odd = !(ax >= 0x1p-100f && ax <= 0x1p100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0ull, 0)) {
        asm volatile("; rcp_: IEEE expansion" ::);
        y = odd ? rcp_ieee(x) : y;
    }
    return y;
}
#endif  // MIRT_FAST_MATH
MIRT_DEV float abs_(float a) { return __builtin_fabsf(a); }
MIRT_DEV uint32_t bits(float f) { return __builtin_bit_cast(uint32_t, f); }
MIRT_DEV float from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- polynomial kernels: Horner, one fma per step, highest power first ----
template <int N>
MIRT_DEV float horner(const float (&c)[N], float z)
{
    float acc = c[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) acc = fma_(acc, z, c[i]);
    return acc;
}

struct SinCos { float s, c; };

// sin & cos together: nearest-integer quadrant by the magic-number trick, three-part pi/2
// reduction (three fmas), two degree-3 polynomials in r*r, quadrant fix-up.
#ifdef MIRT_FAST_MATH
// v_sin_f32 / v_cos_f32 take their argument in revolutions and are specified for |x| <= 256 revolutions
template <bool CLAMP>
MIRT_DEV SinCos sincos_impl(float x)
{
    if constexpr (CLAMP) x = (abs_(x) <= 1024.0f) ? x : 0.0f;
    const float turns = x * 0.15915494f;
    SinCos o;
    o.s = __builtin_amdgcn_sinf(turns);
    o.c = __builtin_amdgcn_cosf(turns);
    return o;
}
#else
template <bool CLAMP>
MIRT_DEV SinCos sincos_impl(float x)
{
    constexpr float kSin[4] = { -0x1.5555560000000p-3f, 0x1.11110e0000000p-7f, -0x1.a013a80000000p-13f, 0x1.6dbe080000000p-19f };
    constexpr float kCos[4] = { 0x1.5555560000000p-5f, -0x1.6c16c00000000p-10f, 0x1.a015c80000000p-16f, -0x1.2524f20000000p-22f };
    constexpr float kTwoOverPi = 0.63661975f, kMagic = 12582912.0f;
    constexpr float kHi = 1.5703125f, kMd = 4.837512969970703125e-4f, kLo = 7.54978995489188216e-8f;
    if constexpr (CLAMP) x = (abs_(x) <= 1048576.0f) ? x : 0.0f;   // outside the reduction domain / NaN -> 0
    const float kf = (x * kTwoOverPi + kMagic) - kMagic;
    const int q = (int)kf;
    float r = fma_(-kf, kHi, x);
    r = fma_(-kf, kMd, r);
    r = fma_(-kf, kLo, r);
    const float z = r * r;
    const float sr = fma_(r * z, horner(kSin, z), r);
    const float cr = fma_(z * z, horner(kCos, z), fma_(-0.5f, z, 1.0f));
    // q&1 swaps, q&2 negates sin, (q+1)&2 negates cos
    const float s0 = (q & 1) ? cr : sr;
    const float c0 = (q & 1) ? sr : cr;
    SinCos o;
    o.s = (q & 2) ? -s0 : s0;
    o.c = ((q + 1) & 2) ? -c0 : c0;
    return o;
}
#endif  // MIRT_FAST_MATH
MIRT_DEV SinCos sincos_(float x) { return sincos_impl<true>(x); }
// sincos_ for arguments the caller knows to be finite with |x| <= 2^20 (a variate times pi or 2 pi): the domain
// clamp never acts there, so leaving it out gives the same bits.
MIRT_DEV SinCos sincos_small(float x) { return sincos_impl<false>(x); }

// Sign of sin(x) in {-1, 0, +1}: the argument reduction of sincos_ without the polynomials.
// For |r| <= pi/4 cos(r) > 0 and sin(r) has the sign of r (zero iff r == 0).
MIRT_DEV int sin_sign(float x)
{
    constexpr float kTwoOverPi = 0.63661975f, kMagic = 12582912.0f;
    constexpr float kHi = 1.5703125f, kMd = 4.837512969970703125e-4f, kLo = 7.54978995489188216e-8f;
    x = (abs_(x) <= 1048576.0f) ? x : 0.0f;
    const float kf = (x * kTwoOverPi + kMagic) - kMagic;
    const int q = (int)kf;
    float r = fma_(-kf, kHi, x);
    r = fma_(-kf, kMd, r);
    r = fma_(-kf, kLo, r);
    const int sr = (r > 0.0f) - (r < 0.0f);
    const int base = (q & 1) ? 1 : sr;
    return (q & 2) ? -base : base;
}

// The reduction of sin_sign as bits, for products of sines: returns z = (bits(r) << 1) | (q & 1), which is 0 exactly
// when sin is 0 (q even and r == +-0), and sets `neg` bit 0 to "sin < 0" (meaningless when sin is 0).
MIRT_DEV uint32_t sin_sign_bits(float x, uint32_t& neg)
{
    constexpr float kTwoOverPi = 0.63661975f, kMagic = 12582912.0f;
    constexpr float kHi = 1.5703125f, kMd = 4.837512969970703125e-4f, kLo = 7.54978995489188216e-8f;
    x = (abs_(x) <= 1048576.0f) ? x : 0.0f;
    const float kf = (x * kTwoOverPi + kMagic) - kMagic;
    const uint32_t q = (uint32_t)(int)kf;
    float r = fma_(-kf, kHi, x);
    r = fma_(-kf, kMd, r);
    r = fma_(-kf, kLo, r);
    const uint32_t br = bits(r);
    // q odd: |sin| = cos(r) > 0, sign + ; q even: the sign of r.  Then q & 2 negates.
    neg = (((br >> 31) & ~q) ^ (q >> 1));
    return (br << 1) | (q & 1u);
}

// sin(a) sin(b) sin(c) < 0, decided exactly as sin_sign(a) * sin_sign(b) * sin_sign(c) < 0
MIRT_DEV bool sin_product_negative(float a, float b, float c)
{
    uint32_t n0, n1, n2;
    const uint32_t z0 = sin_sign_bits(a, n0), z1 = sin_sign_bits(b, n1), z2 = sin_sign_bits(c, n2);
    const uint32_t zmin = (z0 < z1) ? ((z0 < z2) ? z0 : z2) : ((z1 < z2) ? z1 : z2);      // v_min3_u32
    return (((n0 ^ n1 ^ n2) & 1u) != 0u) && (zmin != 0u);
}

MIRT_DEV float asin_core(float x, float z)
{
    constexpr float kAsin[6] = { 0x1.5555540000000p-3f, 0x1.3334300000000p-4f, 0x1.6d5bba0000000p-5f, 0x1.fd8da20000000p-6f, 0x1.18f91e0000000p-6f, 0x1.13fed40000000p-5f };
    return fma_(x * z, horner(kAsin, z), x);
}

// acos with the argument clamped to [-1,1] (NaN behaves as 1).
MIRT_DEV float acos_(float x)
{
    x = (x < 1.0f) ? x : 1.0f;
    x = (x < -1.0f) ? -1.0f : x;
    const float ax = abs_(x);
    const bool small = ax <= 0.5f;
    // both branches share one asin_core evaluation: pick its argument first
    const float zb = (1.0f - ax) * 0.5f;
    const float z = small ? x * x : zb;
    const float a = small ? x : sqrt_(zb);
    const float core = asin_core(a, z);
    const float t = 2.0f * core;
    const float big = (x > 0.0f) ? t : (kPi - t);
    return small ? (kFracPi2 - core) : big;
}

// atan2 with one division: a = min/max; atan(a) = a + a*z*T(z); octant fix-ups.
MIRT_DEV float atan2_(float y, float x)
{
    constexpr float kAtan[9] = { -0x1.5555540000000p-2f, 0x1.99983c0000000p-3f, -0x1.246cd20000000p-3f, 0x1.c3f1680000000p-4f, -0x1.6295f80000000p-4f, 0x1.0001c60000000p-4f, -0x1.25dc120000000p-5f, 0x1.ba9f660000000p-7f, -0x1.38de560000000p-9f };
    const float ax = abs_(x), ay = abs_(y);
    const bool xbig = ax > ay;
    const float mx = xbig ? ax : ay;
    const float mn = xbig ? ay : ax;
    const float a = mn / mx;                               // 0/0 -> NaN, discarded by the last select
    const float z = a * a;
    float r = fma_(a * z, horner(kAtan, z), a);
    r = (ay > ax) ? (kFracPi2 - r) : r;
    r = (x < 0.0f) ? (kPi - r) : r;
    r = (y < 0.0f) ? -r : r;
    return (mx > 0.0f) ? r : 0.0f;                         // branch-free: selects only
}

#ifdef MIRT_FAST_MATH
MIRT_DEV float log2_(float x) { return __builtin_amdgcn_logf(x); }      // v_log_f32 = log2
MIRT_DEV float exp2_(float y) { return __builtin_amdgcn_exp2f(y); }     // v_exp_f32 = 2^y
MIRT_DEV float pow_unit(float x, float y) { return (x > 0.0f) ? exp2_(y * log2_(x)) : 0.0f; }
#else
// log2 of a finite positive float.
MIRT_DEV float log2_(float x)
{
    constexpr float kLog2[10] = { 0x1.7154760000000p+0f, -0x1.7154700000000p-1f, 0x1.ec70aa0000000p-2f, -0x1.715a700000000p-2f, 0x1.277a520000000p-2f, -0x1.eab7aa0000000p-3f, 0x1.a38c680000000p-3f, -0x1.87f6a20000000p-3f, 0x1.7a63a00000000p-3f, -0x1.b84fb60000000p-4f };
    const uint32_t u0 = bits(x);
    const bool sub = u0 < 0x00800000u;                     // subnormal: scale by 2^24 first (select, no branch)
    const uint32_t u = sub ? bits(x * 16777216.0f) : u0;
    const int eadj = sub ? -24 : 0;
    const int e0 = (int)(u >> 23) - 127;
    const uint32_t m0 = (u & 0x007fffffu) | 0x3f800000u;
    const bool big = m0 >= 0x3fb504f3u;                    // m >= sqrt(2): halve
    const uint32_t m = big ? (m0 - 0x00800000u) : m0;
    const int e = big ? (e0 + 1) : e0;
    const float f = from_bits(m) - 1.0f;
    return fma_(f, horner(kLog2, f), (float)(e + eadj));
}

// 2^y: nearest-integer split, degree-7 polynomial, two-step exponent scaling.
MIRT_DEV float exp2_(float y)
{
    constexpr float kExp2[7] = { 0x1.62e4300000000p-1f, 0x1.ebfbe00000000p-3f, 0x1.c6b08e0000000p-5f, 0x1.3b2a1c0000000p-7f, 0x1.5d879e0000000p-10f, 0x1.4440000000000p-13f, 0x1.00a5800000000p-16f };
    constexpr float kMagic = 12582912.0f;
    const bool hi = !(y < 128.0f);                          // -> +inf
    const bool lo = !(y >= -126.0f);                        // -> 0 (also NaN)
    const float yc = (hi || lo) ? 0.0f : y;                 // keep the main path in range; selects, no branches
    const float nf = (yc + kMagic) - kMagic;
    const float f = yc - nf;
    const int n = (int)nf;
    const float p = fma_(f, horner(kExp2, f), 1.0f);
    const int n1 = n >> 1, n2 = n - n1;
    const float s1 = from_bits((uint32_t)(n1 + 127) << 23);
    const float s2 = from_bits((uint32_t)(n2 + 127) << 23);
    const float r = (p * s1) * s2;
    return hi ? __builtin_inff() : (lo ? 0.0f : r);
}

// pow_pos(x, y) for 2^-32 <= x <= 1 (a non-zero uniform variate; 0 gives 0) and 0 < y <= 1: the same bits as
// pow_pos with the branches that cannot act removed -- x is never subnormal (no 2^24 rescale), y * log2(x) lies in
// [-32, 0] (no overflow / underflow clamps), and the two exact power-of-two scalings of exp2_ collapse into ONE
// exact scaling (v_ldexp_f32) because p * 2^n with n >= -32 cannot round.
MIRT_DEV float pow_unit(float x, float y)
{
    constexpr float kLog2[10] = { 0x1.7154760000000p+0f, -0x1.7154700000000p-1f, 0x1.ec70aa0000000p-2f, -0x1.715a700000000p-2f, 0x1.277a520000000p-2f, -0x1.eab7aa0000000p-3f, 0x1.a38c680000000p-3f, -0x1.87f6a20000000p-3f, 0x1.7a63a00000000p-3f, -0x1.b84fb60000000p-4f };
    constexpr float kExp2[7] = { 0x1.62e4300000000p-1f, 0x1.ebfbe00000000p-3f, 0x1.c6b08e0000000p-5f, 0x1.3b2a1c0000000p-7f, 0x1.5d879e0000000p-10f, 0x1.4440000000000p-13f, 0x1.00a5800000000p-16f };
    constexpr float kMagic = 12582912.0f;
    const bool pos = x > 0.0f;
    const uint32_t u = bits(pos ? x : 1.0f);
    const int e0 = (int)(u >> 23) - 127;
    const uint32_t m0 = (u & 0x007fffffu) | 0x3f800000u;
    const bool big = m0 >= 0x3fb504f3u;
    const uint32_t m = big ? (m0 - 0x00800000u) : m0;
    const int e = big ? (e0 + 1) : e0;
    const float f = from_bits(m) - 1.0f;
    const float lg = fma_(f, horner(kLog2, f), (float)e);
    const float yy = y * lg;
    const float nf = (yy + kMagic) - kMagic;
    const float g = yy - nf;
    const float pw = fma_(g, horner(kExp2, g), 1.0f);
    const float r = __builtin_amdgcn_ldexpf(pw, (int)nf);
    return pos ? r : 0.0f;
}
#endif  // MIRT_FAST_MATH

MIRT_DEV float pow_pos(float x, float y)
{
    const float r = exp2_(y * log2_((x > 0.0f) ? x : 1.0f));      // evaluated for every lane, selected below
    return (x > 0.0f) ? r : 0.0f;
}
MIRT_DEV float exp_(float x) { return exp2_(x * 1.44269504f); }

// ---- 3-vectors ----
struct f3 { float x, y, z; };
MIRT_DEV f3 mk(float x, float y, float z) { return f3{ x, y, z }; }
MIRT_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
MIRT_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MIRT_DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
MIRT_DEV f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
MIRT_DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// s*a + b with one fma per component
MIRT_DEV f3 fma3(float s, f3 a, f3 b) { return mk(fma_(s, a.x, b.x), fma_(s, a.y, b.y), fma_(s, a.z, b.z)); }
// path-traced dot: fma(az,bz, fma(ay,by, ax*bx))
MIRT_DEV float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// 1 / sqrt(x) as two correctly rounded steps, == rcp_(sqrt_(x)) bit for bit, behind ONE range check: for
// 2^-100 <= x <= 2^100 the square root lies in [2^-50, 2^50], inside the reciprocal's fast range too.
#ifdef MIRT_FAST_MATH
MIRT_DEV float inv_sqrt_2step(float x) { return __builtin_amdgcn_rsqf(x); }
#else
MIRT_DEV float inv_sqrt_2step(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    const float s = __builtin_fmaf(d, h, g);                   // sqrt_(x)
    const float y0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    float r = __builtin_fmaf(y0, e, y0);                       // rcp_(s)
    const bool odd = !(x >= 0x1p-100f && x <= 0x1p100f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0ull, 0)) {
        asm volatile("; inv_sqrt_2step: IEEE expansion" ::);
        r = odd ? rcp_ieee(sqrt_ieee(x)) : r;
    }
    return r;
}
#endif  // MIRT_FAST_MATH
MIRT_DEV f3 normalize(f3 a) { return inv_sqrt_2step(dot(a, a)) * a; }
// parity-mode dot (nalgebra, no fusion): (a0*b0 + a1*b1) + a2*b2
MIRT_DEV float dot_nofma(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

}  // namespace MIRT_KNS
}  // namespace mirt
