/*
 * mirt_oracle_parity.c — CPU restatement of the reference's CPU pixel loop
 * (src/raytracer/layer.rs:264-444 and callees), operation for operation.
 *
 * TEST INFRASTRUCTURE ONLY (see mirt_oracle.h).  PARITY UNPINNED (no reference fixtures exist).
 *
 * Arithmetic rules (Rust/LLVM semantics): IEEE binary32, no fused multiply-add, no
 * reassociation; nalgebra's 3-vector dot is (a0*b0 + a1*b1) + a2*b2; `normalize` divides each
 * component by sqrt(dot(v,v)); `f32 as u8` truncates toward zero and saturates, NaN -> 0.
 * Built with -ffp-contract=off.
 *
 * Every quirk of the reference is kept on purpose:
 *   Q1  jitter is numerically zero: random_f32() = rand / (f32::MAX + 1.0) <= 2.94e-39
 *       (math.rs:107-110).  It is DEFINED here as +0.0f, so the image needs no RNG.
 *   Q2  v = y/h grows downward while make_ray adds v*vertical upward: image is flipped.
 *   Q3  depth = 20 is per pixel, not per path (layer.rs:318,333-337).
 *   Q4  the world scan returns the LAST sphere hit in list order (layer.rs:426-439:
 *       `closest_hit = old_hit` and nobody ever writes `t`).
 *   Q5  the material is hard-wired to index 2 and the texture is looked up with the
 *       SCREEN-space (uu,vv), not the hit (u,v) (layer.rs:345-351).
 *   Q6  unit_vertor divides by 3 (element count), not by the norm (math.rs:147-149).
 *   Q7  texture_lookup on a 1x1 texture reads texel offset+1 on image row 0 (mod.rs:1008-1014).
 *   Q8  scatter_metal applies no fuzz; `fuzz` only scales the colour (layer.rs:349,366-370).
 *   Q9  closest_hit_raw accepts discriminant == 0 (`< 0.0` rejects, mod.rs:1139).
 */
#define _GNU_SOURCE
#include "mirt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mirt_oracle_internal.h"

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_scale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }   /* f32 * Vec3 */
static inline v3 v_mul_s(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }   /* Vec3 * f32 */
static inline v3 v_div_s(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }   /* Vec3 / f32 */
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* nalgebra dot for 3-vectors: a = a0*b0; a += a1*b1; a += a2*b2 */
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* nalgebra normalize: self / self.norm(), norm = sqrt(norm_squared) */
static inline v3 v_normalize(v3 a) { return v_div_s(a, sqrtf(v_dot(a, a))); }

typedef struct { v3 origin, direction; } ray_t;

/* `Intersection` mod.rs:1058-1093 */
typedef struct {
    v3 p, n;
    float u, v, t;
    int f;
    uint32_t m;
} isect_t;

static inline isect_t isect_new(void)
{
    isect_t r;
    r.p = V(0, 0, 0); r.n = V(0, 0, 0);
    r.u = 0.0f; r.v = 0.0f; r.t = FLT_MAX; r.f = 0; r.m = 0;
    return r;
}

/* `vec3_to_rgb8` math.rs:15-17 — Rust `as u8` */
static inline uint8_t sat_u8(float f)
{
    if (!(f > 0.0f)) return 0;      /* negatives, -0, NaN */
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
static inline uint32_t sat_u32(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* `clamp` math.rs:94-106 */
static inline float clampf(float x, float lo, float hi)
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

typedef struct {
    const MirtGpuCamera* cam;
    const MirtSphere* spheres;
    uint32_t n_spheres;
    const MirtMaterial* mats;
    uint32_t n_mats;
    const float* texels;
    uint64_t n_texels;
    float wf, hf;
    uint32_t spp;
    int faithful;
    MirtStats* st;
} pctx_t;

/* `GpuCamera::make_ray` mod.rs:745-754: ((llc + u*horizontal) + v*vertical) - eye */
static inline ray_t make_ray(const MirtGpuCamera* c, float u, float v)
{
    v3 eye = V(c->eye[0], c->eye[1], c->eye[2]);
    v3 llc = V(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    v3 hor = V(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    v3 ver = V(c->vertical[0], c->vertical[1], c->vertical[2]);
    ray_t r;
    r.origin = eye;
    r.direction = v_sub(v_add(v_add(llc, v_scale(u, hor)), v_scale(v, ver)), eye);
    return r;
}

/* `Intersection::set_face_normal` mod.rs:1095-1110 */
static inline void set_face_normal(isect_t* h, const ray_t* ray, v3 outward)
{
    h->f = v_dot(ray->direction, outward) < 0.0f;
    h->n = h->f ? outward : v_neg(outward);
}

/* `Sphere::update_ray_hit_info` mod.rs:1217-1243 (never writes hit.t) */
static inline void update_ray_hit_info(const MirtSphere* s, const ray_t* ray, float t, isect_t* hit)
{
    if (t < 0.0f) return;
    hit->m = s->material_idx;
    hit->p = v_add(ray->origin, v_mul_s(ray->direction, t));
    v3 c = V(s->center[0], s->center[1], s->center[2]);
    v3 n = v_scale(1.0f / s->radius, v_sub(hit->p, c));
    set_face_normal(hit, ray, n);
    /* `acos(&-n.yy()).len() as f32` is the element count of a Vec2 = 2 (mod.rs:1237-1240) */
    const float PI_F = 3.14159265358979323846f, FRAC_1_PI_F = 0.318309886183790671538f;
    float theta = 2.0f;
    float phi = 2.0f + PI_F;
    hit->u = 0.5f * FRAC_1_PI_F * phi;
    hit->v = FRAC_1_PI_F * theta;
}

/* `Sphere::closest_hit_raw` mod.rs:1121-1157 */
static inline int closest_hit_raw(const MirtSphere* s, const ray_t* ray, float tmin, float tmax,
                                  isect_t* rec, MirtStats* st)
{
    v3 c = V(s->center[0], s->center[1], s->center[2]);
    v3 oc = v_sub(ray->origin, c);
    float a = v_dot(ray->direction, ray->direction);
    float half_b = v_dot(oc, ray->direction);
    float cc = v_dot(oc, oc) - s->radius * s->radius;
    float discriminant = half_b * half_b - a * cc;
    st->sphere_tests++;
    if (discriminant < 0.0f) return 0;
    float closest_t = (-half_b - sqrtf(discriminant)) / a;
    st->roots++;
    if (closest_t < tmin || tmax < closest_t) {
        closest_t = (-half_b + sqrtf(discriminant)) / a;
        st->roots++;
        if (closest_t < tmin || tmax < closest_t) return 0;
    }
    update_ray_hit_info(s, ray, closest_t, rec);
    st->hits++;
    return 1;
}

/* `Layer::ray_hit_world_raw` layer.rs:413-444 */
static int ray_hit_world_raw(const pctx_t* P, const ray_t* ray, float tmin, float tmax, isect_t* rec)
{
    /* FAITHFUL variant: `self.world.clone()` = one Vec allocation + one Box per sphere */
    void** clone = NULL;
    if (P->faithful) {
        clone = (void**)malloc(sizeof(void*) * (P->n_spheres ? P->n_spheres : 1));
        for (uint32_t i = 0; i < P->n_spheres; ++i) {
            clone[i] = malloc(sizeof(MirtSphere));
            memcpy(clone[i], &P->spheres[i], sizeof(MirtSphere));
        }
    }
    isect_t temp_rec = isect_new();
    int hit_anything = 0;
    float closest_hit = tmax;
    float old_hit = rec->t;
    P->st->rays++;
    for (uint32_t i = 0; i < P->n_spheres; ++i) {
        const MirtSphere* object = P->faithful ? (const MirtSphere*)clone[i] : &P->spheres[i];
        if (closest_hit_raw(object, ray, tmin, closest_hit, &temp_rec, P->st)) {
            hit_anything = 1;
            closest_hit = old_hit;
            *rec = temp_rec;
        }
    }
    if (P->faithful) {
        for (uint32_t i = 0; i < P->n_spheres; ++i) free(clone[i]);
        free(clone);
    }
    return hit_anything;
}

/* `texture_lookup` mod.rs:1000-1019; the final index is clamped to the table (the reference
 * would panic / read out of bounds there; both sides of this build clamp). */
static inline v3 texture_lookup(const pctx_t* P, MirtTextureDescriptor desc, float u, float v)
{
    u = clampf(u, 0.0f, 1.0f);
    v = 1.0f - clampf(v, 0.0f, 1.0f);
    uint32_t j = sat_u32(u * (float)desc.width);
    uint32_t i = sat_u32(v * (float)desc.height);
    uint32_t idx = i * desc.width + j;
    uint64_t g = (uint64_t)desc.offset + (uint64_t)idx;
    if (g >= P->n_texels) g = P->n_texels - 1;
    const float* e = P->texels + 3 * g;
    return V(e[0], e[1], e[2]);
}

/* `reflect` math.rs:154-159: v - (2.0 * dot(v,n)) * n */
static inline v3 reflect(v3 v, v3 n) { return v_sub(v, v_scale(2.0f * v_dot(v, n), n)); }
/* `unit_vertor` math.rs:147-149: v / 3 */
static inline v3 unit_vertor(v3 v) { return v_div_s(v, 3.0f); }

/* `scatter_metal` mod.rs:1292-1315 */
static inline int scatter_metal(const ray_t* ray, const isect_t* rec, ray_t* scattered)
{
    v3 reflected = reflect(unit_vertor(ray->direction), rec->n);
    scattered->origin = rec->p;
    scattered->direction = reflected;
    return v_dot(scattered->direction, rec->n) > 0.0f;
}

/* `Layer::ray_color_per_pixel` layer.rs:304-381 */
static void ray_color_per_pixel(const pctx_t* P, uint32_t x, uint32_t y, uint8_t out[3])
{
    float u = (float)x / P->wf;           /* coord_to_color math.rs:4-9 */
    float v = (float)y / P->hf;
    uint32_t depth = 20;
    v3 pixel_color = V(0, 0, 0);
    for (uint32_t s = 0; s < P->spp; ++s) {
        float uu = u + 0.0f, vv = v + 0.0f;            /* Q1 */
        P->st->lane_iterations++;                      /* samples the sequential loop executes before it returns */
        ray_t ray = make_ray(P->cam, uu, vv);
        isect_t rec = isect_new();                     /* Box::into_raw(Box::new(Intersection::new())) */
        if (ray_hit_world_raw(P, &ray, 0.001f, FLT_MAX, &rec)) {
            if (depth <= 0) { out[0] = out[1] = out[2] = 0; return; }
            depth -= 1;
            ray_t scattered;
            const uint32_t index = 2;                  /* Q5 */
            MirtTextureDescriptor texture = P->mats[index].desc1;
            float fuzzy = P->mats[2].x;
            v3 albedo = texture_lookup(P, texture, uu, vv);
            P->st->scatter[1]++;
            if (!scatter_metal(&ray, &rec, &scattered)) { out[0] = out[1] = out[2] = 0; return; }
            if (ray_hit_world_raw(P, &scattered, 0.001f, FLT_MAX, &rec)) {
                v3 sampled = v_div_s(v_mul_s(v_normalize(rec.n), 255.0f), 2.0f);
                sampled.x *= albedo.x * fuzzy;
                sampled.y *= albedo.y * fuzzy;
                sampled.z *= albedo.z * fuzzy;
                pixel_color = v_add(pixel_color, sampled);
                out[0] = sat_u8(pixel_color.x);
                out[1] = sat_u8(pixel_color.y);
                out[2] = sat_u8(pixel_color.z);
                return;
            }
        }
    }
    out[0] = sat_u8(v * 255.0f);
    out[1] = sat_u8(u * 255.0f);
    out[2] = sat_u8(255.0f);
}

/* `Layer::set_data` layer.rs:264-282 over the rows this call owns */
int mirt_oracle_render_parity(const MirtScene* scene, const MirtParams* params, uint8_t* out,
                              int n_threads, int variant, MirtStats* total)
{
    const uint32_t rows = mirt_params_out_rows_impl(params);
    const uint32_t W = params->width;
    int nt = mirt_oracle_pick_threads(n_threads);
#pragma omp parallel num_threads(nt)
    {
        MirtStats st;
        memset(&st, 0, sizeof st);
        pctx_t P;
        P.cam = scene->camera; P.spheres = scene->spheres; P.n_spheres = scene->n_spheres;
        P.mats = scene->materials; P.n_mats = scene->n_materials;
        P.texels = scene->texels; P.n_texels = scene->n_texels;
        P.wf = (float)params->width; P.hf = (float)params->height;
        P.spp = params->spp; P.faithful = (variant == MIRT_ORACLE_FAITHFUL); P.st = &st;
#pragma omp for schedule(dynamic, 1)
        for (uint32_t i = 0; i < rows; ++i) {
            uint32_t y = mirt_params_out_row_index_impl(params, i);
            for (uint32_t x = 0; x < W; ++x) {
                uint8_t rgb[3];
                ray_color_per_pixel(&P, x, y, rgb);
                uint8_t* px = out + ((size_t)i * W + x) * 4;
                px[0] = rgb[0]; px[1] = rgb[1]; px[2] = rgb[2]; px[3] = 255;
            }
        }
#pragma omp critical
        mirt_oracle_stats_add(total, &st);
    }
    return MIRT_OK;
}
