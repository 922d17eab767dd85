/*
 * mirt_oracle_pt.c — CPU restatement of the path-traced mode: the behaviours of the
 * reference's WGSL fragment shader (src/raytracer/raytracer.wgsl:50-521), one (pixel, sample)
 * work item at a time.
 *
 * TEST INFRASTRUCTURE ONLY (see mirt_oracle.h).  PARITY UNPINNED: the WGSL cannot run here
 * (no wgpu/naga), the reference holds no images or vectors, and WGSL leaves the precision of
 * sin/cos/acos/atan2/pow/exp implementation-defined.  What this file pins is the BEHAVIOUR
 * (which formulas, which RNG draws in which order, which quirks) with arithmetic that is fully
 * specified (mirt_oracle_math.h), so that the HIP kernel can be checked bit for bit.
 *
 * Spec decisions (DESIGN.md "PT spec"):
 *   S1  sample s of pixel (x,y) uses the WGSL RNG stream of frame_number = s + 1 at 1 spp per
 *       frame (mod.rs:284, 626-670; wgsl:498-502): state0 = jenkins((x + y*W) ^ jenkins(s+1) ^ mix(seed)),
 *       mix(0) = 0.  The image is therefore independent of tiling, GPU count and scheduling.
 *   S2  every sample's radiance is converted to unsigned fixed point (2^-20 units, clamped to
 *       [0, 4096)) and pixels are summed in 64-bit integers: the sum is exact and
 *       order-independent (the WGSL accumulates f32 frame by frame, wgsl:64-73).
 *   S3  the dielectric keeps the reference's quirk (wgsl:266-273): the Schlick branch draws one
 *       random number and discards the reflection, so the ray always refracts when it can.
 *   S4  WGSL `discriminant > 0`, `t < tmax && t > tmin`, MIN_T = 0.001, MAX_T = 1000, outward
 *       normal without face flip (wgsl:407-440).
 *   S5  resolve: mean -> uncharted2 (wgsl:83-103) -> sRGB OETF (the Bgra8UnormSrgb surface,
 *       main.rs:465) -> round-to-nearest unorm8; flags can switch the two curves off.
 *   S6  sky: RTIOW gradient by default (the Hosek-Wilkie tables live in the hw-skymodel crate,
 *       which is not available); with MIRT_FLAG_SKY_HOSEK the 144-byte state is evaluated
 *       exactly as wgsl:154-166, 316-343.
 *   S7  texture index: offset + i*width + j, clamped to the table (wgsl:377-387; WGSL's robust
 *       buffer access makes out-of-range reads implementation-defined).
 *   S9  checkerboard: the sign of sin(5x)*sin(5y)*sin(5z) (wgsl:301-302) is taken from the signs
 *       of the three sines (exact argument reduction, no polynomial); it differs from the float
 *       product only if that product would underflow to zero.
 */
#define _GNU_SOURCE
#include <math.h>
#include <string.h>

#include "mirt_oracle_internal.h"
#include "mirt_oracle_math.h"

/* ---------------- RNG (wgsl:493-521) ---------------- */

static inline uint32_t jenkins_hash(uint32_t x)
{
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}

static inline uint32_t seed_mix(uint64_t seed)
{
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    return jenkins_hash(lo ^ jenkins_hash(hi));
}

static inline uint32_t rng_init(uint32_t pixel_index, uint32_t sample, uint32_t mix)
{
    uint32_t frame = sample + 1u;                                /* S1 */
    return jenkins_hash((pixel_index ^ jenkins_hash(frame)) ^ mix);
}

static inline float rng_next(uint32_t* state)
{
    uint32_t old = *state + 747796405u + 2891336453u;            /* add, not multiply-add (wgsl:508) */
    uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
    *state = (word >> 22) ^ word;
    return (float)(*state) * 0x1p-32f;                           /* f32(state) / f32(0xffffffff) == / 2^32 */
}

void mirt_oracle_rng_stream(uint32_t pixel_index, uint32_t sample, uint64_t seed, float* out, size_t n)
{
    uint32_t st = rng_init(pixel_index, sample, seed_mix(seed));
    for (size_t i = 0; i < n; ++i) out[i] = rng_next(&st);
}

/* ---------------- scene access ---------------- */

typedef struct {
    const MirtGpuCamera* cam;
    const MirtSphere* spheres;
    uint32_t n_spheres;
    const MirtMaterial* mats;
    const float* texels;
    uint64_t n_texels;
    const MirtSkyState* sky;
    uint32_t W, H, num_bounces, flags, mix;
    float inv_w, inv_h;
    MirtStats* st;
} tctx_t;

typedef struct { ov3 o, d; } tray_t;
typedef struct { ov3 p, n; float u, v, t; } thit_t;

static inline float om_max(float a, float b) { return (b > a) ? b : a; }   /* max(a, b), NaN b -> a */

static inline uint32_t sat_u32f(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* textureLookup wgsl:377-387 (+S7) */
static inline ov3 tex_lookup(const tctx_t* T, MirtTextureDescriptor d, float u, float v)
{
    float uc = (u < 0.0f) ? 0.0f : ((u > 1.0f) ? 1.0f : u);
    float vc = (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v);
    float vf = 1.0f - vc;
    uint32_t j = sat_u32f(uc * (float)d.width);
    uint32_t i = sat_u32f(vf * (float)d.height);
    uint32_t idx = i * d.width + j;
    uint64_t g = (uint64_t)d.offset + (uint64_t)idx;
    if (g >= T->n_texels) g = T->n_texels - 1;
    const float* e = T->texels + 3 * g;
    return ov(e[0], e[1], e[2]);
}

/* rngNextVec3InUnitSphere wgsl:480-491 */
static inline ov3 rand_in_unit_sphere(uint32_t* st)
{
    float r = om_pow_pos(rng_next(st), 0.33333f);
    float theta = OM_PI * rng_next(st);
    float phi = OM_TWO_PI * rng_next(st);
    float sth, cth, sph, cph;
    om_sincos(theta, &sth, &cth);
    om_sincos(phi, &sph, &cph);
    float rs = r * sth;
    return ov(rs * cph, rs * sph, r * cth);
}

/* reflect(v, n) = v - 2*dot(v,n)*n */
static inline ov3 reflect3(ov3 v, ov3 n) { return ov_fma(-(2.0f * ov_dot(v, n)), n, v); }

/* cameraMakeRay wgsl:456-478 */
static inline tray_t camera_make_ray(const tctx_t* T, uint32_t* st, float u, float v)
{
    const MirtGpuCamera* c = T->cam;
    float r = sqrtf(rng_next(st));
    float alpha = OM_TWO_PI * rng_next(st);
    float sa, ca;
    om_sincos(alpha, &sa, &ca);
    float px = c->lens_radius * (r * ca);
    float py = c->lens_radius * (r * sa);
    ov3 cu = ov(c->u[0], c->u[1], c->u[2]), cv = ov(c->v[0], c->v[1], c->v[2]);
    ov3 off = ov_fma(py, cv, ov_scale(px, cu));
    tray_t ray;
    ray.o = ov_add(ov(c->eye[0], c->eye[1], c->eye[2]), off);
    ov3 llc = ov(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    ov3 hor = ov(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    ov3 ver = ov(c->vertical[0], c->vertical[1], c->vertical[2]);
    ray.d = ov_sub(ov_fma(v, ver, ov_fma(u, hor, llc)), ray.o);
    return ray;
}

/* nearest hit over the sphere list, wgsl:135-145 + rayIntersectSphere wgsl:407-429 */
static inline int nearest_hit(const tctx_t* T, const tray_t* ray, thit_t* hit, uint32_t* mat_idx)
{
    float a = ov_dot(ray->d, ray->d);
    float inv_a = 1.0f / a;
    float closest = OM_MAX_T;
    int32_t best = -1;
    T->st->rays++;
    for (uint32_t i = 0; i < T->n_spheres; ++i) {
        const MirtSphere* s = &T->spheres[i];
        ov3 oc = ov_sub(ray->o, ov(s->center[0], s->center[1], s->center[2]));
        float b = ov_dot(oc, ray->d);
        float cq = ov_dot(oc, oc) - s->radius * s->radius;
        float disc = MFMA(b, b, -(a * cq));
        T->st->sphere_tests++;
        if (disc > 0.0f) {
            float sq = sqrtf(disc);
            float t = (-b - sq) * inv_a;
            T->st->roots++;
            if (t < closest && t > OM_MIN_T) {
                closest = t; best = (int32_t)i;
            } else {
                t = (-b + sq) * inv_a;
                T->st->roots++;
                if (t < closest && t > OM_MIN_T) { closest = t; best = (int32_t)i; }
            }
        }
    }
    if (best < 0) return 0;
    /* sphereIntersection wgsl:431-440 */
    const MirtSphere* s = &T->spheres[best];
    ov3 c = ov(s->center[0], s->center[1], s->center[2]);
    hit->t = closest;
    hit->p = ov_fma(closest, ray->d, ray->o);
    hit->n = ov_scale(1.0f / s->radius, ov_sub(hit->p, c));
    float theta = om_acos(-hit->n.y);
    float phi = om_atan2(-hit->n.z, hit->n.x) + OM_PI;
    hit->u = (0.5f * OM_FRAC_1_PI) * phi;
    hit->v = OM_FRAC_1_PI * theta;
    *mat_idx = s->material_idx;
    T->st->hits++;
    return 1;
}

/* scatterLambertian wgsl:204-242 */
static inline void scatter_lambertian(const tctx_t* T, const thit_t* hit, MirtTextureDescriptor tex,
                                      uint32_t* st, tray_t* out_ray, ov3* atten)
{
    float r1 = rng_next(st);
    float r2 = rng_next(st);
    float sqrt_r2 = sqrtf(r2);
    float z = sqrtf(1.0f - r2);
    float phi = OM_TWO_PI * r1;
    float sp, cp;
    om_sincos(phi, &sp, &cp);
    float lx = cp * sqrt_r2;
    float ly = sp * sqrt_r2;
    /* pixarOnb wgsl:233-242 */
    ov3 n = hit->n;
    float sg = (n.z >= 0.0f) ? 1.0f : -1.0f;
    float aa = -1.0f / (sg + n.z);
    float bb = n.x * n.y * aa;
    ov3 U = ov(MFMA(sg * n.x, n.x * aa, 1.0f), sg * bb, -(sg * n.x));
    ov3 V = ov(bb, MFMA(n.y, n.y * aa, sg), -n.y);
    ov3 wi = ov_fma(z, n, ov_fma(ly, V, ov_scale(lx, U)));
    float dn = ov_dot(n, wi);
    float k = (OM_FRAC_1_PI * om_max(OM_EPSILON, dn)) / om_max(OM_EPSILON, dn * OM_FRAC_1_PI);
    ov3 t = tex_lookup(T, tex, hit->u, hit->v);
    *atten = ov_scale(k, t);
    out_ray->o = hit->p;
    out_ray->d = wi;
}

static inline void scatter(const tctx_t* T, const tray_t* in, const thit_t* hit, const MirtMaterial* m,
                           uint32_t* st, tray_t* out_ray, ov3* atten)
{
    switch (m->id) {
    case 0:
        T->st->scatter[0]++;
        scatter_lambertian(T, hit, m->desc1, st, out_ray, atten);
        return;
    case 1: {   /* scatterMetal wgsl:244-248 */
        T->st->scatter[1]++;
        ov3 refl = reflect3(in->d, hit->n);
        ov3 rs = rand_in_unit_sphere(st);
        out_ray->o = hit->p;
        out_ray->d = ov_fma(m->x, rs, refl);
        *atten = tex_lookup(T, m->desc1, hit->u, hit->v);
        return;
    }
    case 2: {   /* scatterDielectric wgsl:250-292 (+S3) */
        T->st->scatter[2]++;
        ov3 wo = in->d;
        float dn = ov_dot(wo, hit->n);
        ov3 uv = ov_normalize(wo);
        ov3 outn;
        float ratio;
        if (dn > 0.0f) { outn = ov_neg(hit->n); ratio = m->x; }
        else           { outn = hit->n;         ratio = 1.0f / m->x; }
        float dt = ov_dot(uv, outn);
        float disc = MFMA(-(ratio * ratio), MFMA(-dt, dt, 1.0f), 1.0f);
        out_ray->o = hit->p;
        if (disc > 0.0f) {
            float sq = sqrtf(disc);
            ov3 q = ov_fma(-dt, outn, uv);
            ov3 w = ov_fma(-sq, outn, ov_scale(ratio, q));
            out_ray->d = ov_normalize(w);
            (void)rng_next(st);           /* the discarded Schlick draw */
        } else {
            out_ray->d = reflect3(wo, hit->n);
        }
        *atten = ov(1.0f, 1.0f, 1.0f);
        return;
    }
    case 3: {   /* scatterCheckerboard wgsl:300-307 */
        T->st->scatter[3]++;
        /* S9: `sin(5x)*sin(5y)*sin(5z) < 0` decided from the signs of the three sines */
        int sines = om_sin_sign(5.0f * hit->p.x) * om_sin_sign(5.0f * hit->p.y) * om_sin_sign(5.0f * hit->p.z);
        scatter_lambertian(T, hit, (sines < 0) ? m->desc1 : m->desc2, st, out_ray, atten);
        return;
    }
    default: {  /* scatterMissingMaterial wgsl:309-314 */
        T->st->scatter[4]++;
        ov3 rs = rand_in_unit_sphere(st);
        out_ray->o = hit->p;
        out_ray->d = ov_add(hit->n, rs);
        *atten = ov(0.9921f, 0.24705f, 0.57254f);
        return;
    }
    }
}

/* radiance() wgsl:316-343 for one channel */
static inline float hosek_radiance(const MirtSkyState* S, float theta, float gamma, int ch)
{
    const float* p = S->params + 9 * ch;
    float r = S->radiances[ch];
    float cg = om_cos(gamma);
    float cg2 = cg * cg;
    float ct = fabsf(om_cos(theta));
    float expm = om_exp(p[4] * gamma);
    float base = MFMA(-(2.0f * p[8]), cg, MFMA(p[8], p[8], 1.0f));
    float mie = (1.0f + cg2) / (base * sqrtf(base));      /* pow(base, 1.5) */
    float zenith = sqrtf(ct);
    float lhs = MFMA(p[0], om_exp(p[1] / (ct + 0.01f)), 1.0f);
    float rhs = MFMA(p[7], zenith, MFMA(p[6], mie, MFMA(p[5], cg2, MFMA(p[3], expm, p[2]))));
    return r * (lhs * rhs);
}

static inline ov3 sky_color(const tctx_t* T, ov3 d)
{
    ov3 v = ov_normalize(d);
    if (T->flags & MIRT_FLAG_SKY_HOSEK) {
        const MirtSkyState* S = T->sky;
        ov3 s = ov(S->sun_direction[0], S->sun_direction[1], S->sun_direction[2]);
        float theta = om_acos(v.y);
        float gamma = om_acos(ov_dot(v, s));              /* om_acos clamps to [-1,1] */
        return ov(hosek_radiance(S, theta, gamma, 0), hosek_radiance(S, theta, gamma, 1),
                  hosek_radiance(S, theta, gamma, 2));
    }
    /* RTIOW background: (1-t)*white + t*(0.5,0.7,1.0), t = 0.5*(unit.y + 1) */
    float t = 0.5f * (v.y + 1.0f);
    float omt = 1.0f - t;
    return ov(MFMA(t, 0.5f, omt), MFMA(t, 0.7f, omt), MFMA(t, 1.0f, omt));
}

static inline uint32_t to_fixed(float c)
{
    if (!(c > 0.0f)) return 0;                       /* negatives and NaN */
    float s = c * 1048576.0f;
    if (s >= 4294967040.0f) s = 4294967040.0f;        /* largest f32 below 2^32 */
    return (uint32_t)s;
}

/* samplePixel + rayColor for ONE sample (wgsl:105-172) -> fixed-point rgb */
/* `rng` = NULL: the sample's own stream (S1).  Otherwise the caller's running stream of the current frame
 * (MirtParams.frame_spp > 0: the reference's samplePixel loop, wgsl:105-122, draws all samples of a frame from one). */
static void trace_sample(const tctx_t* T, uint32_t x, uint32_t y, uint32_t sample, uint32_t q[3], uint32_t* rng)
{
    uint32_t st = rng ? *rng : rng_init(x + y * T->W, sample, T->mix);
    float u = ((float)x + rng_next(&st)) * T->inv_w;
    float v = ((float)y + rng_next(&st)) * T->inv_h;
    tray_t ray = camera_make_ray(T, &st, u, 1.0f - v);
    ov3 color = ov(0, 0, 0);
    ov3 thr = ov(1, 1, 1);
    for (uint32_t bounce = 0; bounce < T->num_bounces; ++bounce) {
        thit_t hit;
        uint32_t mi = 0;
        if (nearest_hit(T, &ray, &hit, &mi)) {
            tray_t next;
            ov3 att;
            scatter(T, &ray, &hit, &T->mats[mi], &st, &next, &att);
            ray = next;
            thr = ov(thr.x * att.x, thr.y * att.y, thr.z * att.z);
        } else {
            color = sky_color(T, ray.d);
            T->st->sky_misses++;
            break;
        }
    }
    q[0] = to_fixed(thr.x * color.x);
    q[1] = to_fixed(thr.y * color.y);
    q[2] = to_fixed(thr.z * color.z);
    if (rng) *rng = st;              /* the frame's stream continues with the next sample */
}

/* ---------------- resolve (S5) ---------------- */

static inline float uncharted2_tonemap(float x)
{
    const float A = 0.15f, B = 0.50f, CB = 0.05f, DE = 0.004f, DF = 0.06f;
    const float EF = 0.02f / 0.30f;
    float num = MFMA(x, MFMA(A, x, CB), DE);
    float den = MFMA(x, MFMA(A, x, B), DF);
    return num / den - EF;
}

static inline float uncharted2(float x)
{
    float curr = uncharted2_tonemap(0.246f * x);
    float white = 1.0f / uncharted2_tonemap(11.2f);
    return white * curr;
}

static inline float srgb_oetf(float x)
{
    if (!(x > 0.0031308f)) return 12.92f * x;
    return MFMA(1.055f, om_pow_pos(x, 0.41666666f), -0.055f);
}

static inline uint8_t quantise(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)MFMA(x, 255.0f, 0.5f);
}

static void resolve_pixel(const uint64_t sum[3], uint32_t n_samples, uint32_t flags, uint8_t px[4])
{
    double denom = (double)n_samples * 1048576.0;
    for (int k = 0; k < 3; ++k) {
        float m = (float)((double)sum[k] / denom);
        if (!(flags & MIRT_FLAG_NO_TONEMAP)) m = uncharted2(m);
        if (!(flags & MIRT_FLAG_NO_SRGB)) m = srgb_oetf(m);
        px[k] = quantise(m);
    }
    px[3] = 255;
}

int mirt_oracle_render_pt(const MirtScene* scene, const MirtParams* params, uint8_t* out_rgba8,
                          uint64_t* out_sums, int n_threads, MirtStats* total)
{
    const uint32_t rows = mirt_params_out_rows_impl(params);
    const uint32_t W = params->width;
    int nt = mirt_oracle_pick_threads(n_threads);
#pragma omp parallel num_threads(nt)
    {
        MirtStats st;
        memset(&st, 0, sizeof st);
        tctx_t T;
        T.cam = scene->camera; T.spheres = scene->spheres; T.n_spheres = scene->n_spheres;
        T.mats = scene->materials; T.texels = scene->texels; T.n_texels = scene->n_texels;
        T.sky = scene->sky;
        T.W = params->width; T.H = params->height;
        T.num_bounces = params->num_bounces; T.flags = params->flags;
        T.mix = seed_mix(params->seed);
        T.inv_w = 1.0f / (float)params->width;
        T.inv_h = 1.0f / (float)params->height;
        T.st = &st;
#pragma omp for schedule(dynamic, 1)
        for (uint32_t i = 0; i < rows; ++i) {
            uint32_t y = mirt_params_out_row_index_impl(params, i);
            for (uint32_t x = 0; x < W; ++x) {
                uint64_t sum[3] = { 0, 0, 0 };
                uint32_t frame_rng = 0;
                for (uint32_t s = 0; s < params->spp; ++s) {
                    uint32_t q[3];
                    const uint32_t sample = params->sample_begin + s;
                    if (params->frame_spp) {
                        /* frame f = sample / n + 1 seeds once: rng_init's `sample + 1` is the frame number */
                        if (sample % params->frame_spp == 0) frame_rng = rng_init(x + y * T.W, params->frame_begin + sample / params->frame_spp, T.mix);   /* frame_number survives resets: mod.rs:284, 350, 385 */
                        trace_sample(&T, x, y, sample, q, &frame_rng);
                    } else {
                        trace_sample(&T, x, y, sample, q, NULL);
                    }
                    sum[0] += q[0]; sum[1] += q[1]; sum[2] += q[2];
                }
                size_t pi = (size_t)i * W + x;
                if (out_sums) { out_sums[3 * pi] = sum[0]; out_sums[3 * pi + 1] = sum[1]; out_sums[3 * pi + 2] = sum[2]; }
                if (out_rgba8) resolve_pixel(sum, params->spp, params->flags, out_rgba8 + 4 * pi);
            }
        }
        st.lane_iterations = st.rays;
#pragma omp critical
        mirt_oracle_stats_add(total, &st);
    }
    return MIRT_OK;
}
