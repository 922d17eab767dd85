/*
 * mirt_oracle_host.c — oracle entry points, parameter handling and the host-side set-up
 * arithmetic of the reference (camera construction, Angle, parameter validation).
 *
 * TEST INFRASTRUCTURE ONLY (see mirt_oracle.h).  PARITY UNPINNED except for the five Angle
 * KATs of src/raytracer/angle.rs:52-93.
 */
#define _GNU_SOURCE
#include "mirt_oracle_internal.h"
#include "mirt_oracle_math.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static _Thread_local MirtStats g_last_stats;

/* ---------------- row selection (MirtParams contract, include/mirt.h) ---------------- */

static int rows_valid(const MirtParams* p, uint32_t* rb, uint32_t* re)
{
    uint32_t b = p->row_begin, e = p->row_end == 0 ? p->height : p->row_end;
    if (b >= e || e > p->height) return 0;
    if (p->tile_rows != 0 && p->n_parts > 1 && p->part >= p->n_parts) return 0;
    *rb = b; *re = e;
    return 1;
}

uint32_t mirt_params_out_rows_impl(const MirtParams* p)
{
    uint32_t rb, re;
    if (!p || p->width == 0 || p->height == 0 || !rows_valid(p, &rb, &re)) return 0;
    uint32_t band = re - rb;
    if (p->tile_rows == 0 || p->n_parts <= 1) return band;
    uint32_t tr = p->tile_rows;
    uint32_t tiles = (band + tr - 1) / tr;
    if (tiles <= p->part) return 0;
    uint32_t owned = (tiles - p->part + p->n_parts - 1) / p->n_parts;
    uint32_t rows = owned * tr;
    if ((tiles - 1) % p->n_parts == p->part) rows -= tiles * tr - band;   /* ragged last tile */
    return rows;
}

uint32_t mirt_params_out_row_index_impl(const MirtParams* p, uint32_t i)
{
    uint32_t rb, re;
    if (!p || !rows_valid(p, &rb, &re)) return UINT32_MAX;
    if (i >= mirt_params_out_rows_impl(p)) return UINT32_MAX;
    if (p->tile_rows == 0 || p->n_parts <= 1) return rb + i;
    uint32_t tr = p->tile_rows;
    uint32_t t = p->part + (i / tr) * p->n_parts;
    return rb + t * tr + i % tr;
}

/* ---------------- validation: same status codes as the product for the same inputs -------- */

static int desc_ok(const MirtTextureDescriptor* d, uint64_t n_texels)
{
    if (d->width == 0 || d->height == 0) return 0;
    uint64_t end = (uint64_t)d->offset + (uint64_t)d->width * (uint64_t)d->height;
    return end <= n_texels;
}

int mirt_oracle_check(const MirtScene* scene, const MirtParams* params)
{
    if (!scene || !params || !scene->camera) return MIRT_ERR_NULL_POINTER;
    if (scene->n_spheres && !scene->spheres) return MIRT_ERR_NULL_POINTER;
    if (scene->n_materials && !scene->materials) return MIRT_ERR_NULL_POINTER;
    if (scene->n_texels && !scene->texels) return MIRT_ERR_NULL_POINTER;
    if (params->width == 0 || params->height == 0) return MIRT_ERR_VIEWPORT_SIZE;
    if (params->spp == 0) return MIRT_ERR_SPP_ZERO;
    if (params->spp > MIRT_MAX_SPP_PER_CALL || (uint64_t)params->sample_begin + params->spp > 0xffffffffull) return MIRT_ERR_SPP_RANGE;
    if (params->mode != MIRT_MODE_PARITY && params->mode != MIRT_MODE_PT) return MIRT_ERR_BAD_MODE;
    uint32_t rb, re;
    if (!rows_valid(params, &rb, &re)) return MIRT_ERR_BAD_ROWS;
    if (params->mode == MIRT_MODE_PT && params->frame_spp != 0 &&
        (params->spp % params->frame_spp != 0 || params->sample_begin % params->frame_spp != 0)) return MIRT_ERR_FRAME_SPP;
    if (params->mode == MIRT_MODE_PARITY) {
        if (scene->n_spheres > 0) {
            /* layer.rs:345-349 reads material_data[2] on every primary hit */
            if (scene->n_materials < 3) return MIRT_ERR_MATERIAL_INDEX;
            if (!desc_ok(&scene->materials[2].desc1, scene->n_texels)) return MIRT_ERR_TEXEL_RANGE;
        }
    } else {
        for (uint32_t i = 0; i < scene->n_spheres; ++i)
            if (scene->spheres[i].material_idx >= scene->n_materials) return MIRT_ERR_MATERIAL_INDEX;
        for (uint32_t i = 0; i < scene->n_materials; ++i) {
            const MirtMaterial* m = &scene->materials[i];
            if (m->id == 0 || m->id == 1 || m->id == 3)
                if (!desc_ok(&m->desc1, scene->n_texels)) return MIRT_ERR_TEXEL_RANGE;
            if (m->id == 3)
                if (!desc_ok(&m->desc2, scene->n_texels)) return MIRT_ERR_TEXEL_RANGE;
        }
        if ((params->flags & MIRT_FLAG_SKY_HOSEK) && !scene->sky) return MIRT_ERR_SKY;
    }
    return MIRT_OK;
}

int mirt_oracle_pick_threads(int n_threads)
{
    int maxt = omp_get_num_procs();
    if (n_threads <= 0 || n_threads > maxt) return maxt;
    return n_threads;
}

void mirt_oracle_stats_add(MirtStats* t, const MirtStats* p)
{
    t->rays += p->rays; t->sphere_tests += p->sphere_tests; t->roots += p->roots;
    t->hits += p->hits; t->sky_misses += p->sky_misses;
    t->lane_iterations += p->lane_iterations; t->wave_iterations += p->wave_iterations;
    for (int i = 0; i < 5; ++i) t->scatter[i] += p->scatter[i];
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int mirt_oracle_render(const MirtScene* scene, const MirtParams* params, uint8_t* out_rgba8,
                       size_t out_len, int n_threads, int variant)
{
    int rc = mirt_oracle_check(scene, params);
    if (rc != MIRT_OK) return rc;
    if (!out_rgba8) return MIRT_ERR_NULL_POINTER;
    size_t need = (size_t)mirt_params_out_rows_impl(params) * params->width * 4;
    if (out_len < need) return MIRT_ERR_OUT_BUFFER;
    MirtStats st;
    memset(&st, 0, sizeof st);
    double t0 = now_ms();
    if (params->mode == MIRT_MODE_PARITY)
        rc = mirt_oracle_render_parity(scene, params, out_rgba8, n_threads, variant, &st);
    else
        rc = mirt_oracle_render_pt(scene, params, out_rgba8, NULL, n_threads, &st);
    st.kernel_ms = now_ms() - t0;
    st.samples = (uint64_t)mirt_params_out_rows_impl(params) * params->width * params->spp;
    g_last_stats = st;
    return rc;
}

int mirt_oracle_render_pt_sums(const MirtScene* scene, const MirtParams* params, uint64_t* out_sums,
                               size_t out_len_u64, int n_threads)
{
    int rc = mirt_oracle_check(scene, params);
    if (rc != MIRT_OK) return rc;
    if (params->mode != MIRT_MODE_PT) return MIRT_ERR_BAD_MODE;
    if (!out_sums) return MIRT_ERR_NULL_POINTER;
    size_t need = (size_t)mirt_params_out_rows_impl(params) * params->width * 3;
    if (out_len_u64 < need) return MIRT_ERR_OUT_BUFFER;
    MirtStats st;
    memset(&st, 0, sizeof st);
    double t0 = now_ms();
    rc = mirt_oracle_render_pt(scene, params, NULL, out_sums, n_threads, &st);
    st.kernel_ms = now_ms() - t0;
    st.samples = (uint64_t)mirt_params_out_rows_impl(params) * params->width * params->spp;
    g_last_stats = st;
    return rc;
}

int mirt_oracle_get_stats(MirtStats* out)
{
    if (!out) return MIRT_ERR_NULL_POINTER;
    *out = g_last_stats;
    return MIRT_OK;
}

/* ---------------- Angle (angle.rs:8-22) ---------------- */

#define RUST_PI 3.14159265358979323846f   /* std::f32::consts::PI */

float mirt_oracle_degrees_to_radians(float degrees) { return degrees * RUST_PI / 180.0f; }
float mirt_oracle_radians_to_degrees(float radians) { return radians * 180.0f / RUST_PI; }

/* ---------------- RenderParams::validate (mod.rs:450-484) ---------------- */

int mirt_oracle_validate_render_params(const MirtCamera* camera, const MirtSamplingParams* sampling,
                                       uint32_t w, uint32_t h)
{
    if (!camera || !sampling) return MIRT_ERR_NULL_POINTER;
    if (sampling->num_samples_per_pixel == 0) return MIRT_ERR_SPP_ZERO;   /* Rust would panic on `% 0` */
    if (sampling->max_samples_per_pixel % sampling->num_samples_per_pixel != 0)
        return MIRT_ERR_MAX_SAMPLES_MULTIPLE;
    if (w == 0 || h == 0) return MIRT_ERR_VIEWPORT_SIZE;
    float lo = mirt_oracle_degrees_to_radians(0.0f), hi = mirt_oracle_degrees_to_radians(90.0f);
    if (!(camera->vfov_radians >= lo && camera->vfov_radians <= hi)) return MIRT_ERR_VFOV_RANGE;
    if (!(camera->aperture >= 0.0f && camera->aperture <= 1.0f)) return MIRT_ERR_APERTURE_RANGE;
    if (camera->focus_distance < 0.0f) return MIRT_ERR_FOCUS_DISTANCE;
    return MIRT_OK;
}

/* ---------------- GpuCamera::new (mod.rs:700-741), nalgebra arithmetic ---------------- */

typedef struct { float x, y, z; } hv3;
static inline hv3 H(float x, float y, float z) { hv3 r = { x, y, z }; return r; }
static inline float h_dot(hv3 a, hv3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline hv3 h_normalize(hv3 a) { float n = sqrtf(h_dot(a, a)); return H(a.x / n, a.y / n, a.z / n); }
static inline hv3 h_cross(hv3 a, hv3 b)
{
    return H(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline hv3 h_scale(float s, hv3 a) { return H(s * a.x, s * a.y, s * a.z); }
static inline hv3 h_add(hv3 a, hv3 b) { return H(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline hv3 h_sub(hv3 a, hv3 b) { return H(a.x - b.x, a.y - b.y, a.z - b.z); }

int mirt_oracle_camera_new(const MirtCamera* camera, uint32_t vw, uint32_t vh, MirtGpuCamera* out)
{
    if (!camera || !out) return MIRT_ERR_NULL_POINTER;
    if (vw == 0 || vh == 0) return MIRT_ERR_VIEWPORT_SIZE;
    float lens_radius = 0.5f * camera->aperture;
    float aspect = (float)vw / (float)vh;
    float theta = camera->vfov_radians;
    float half_height = camera->focus_distance * tanf(0.5f * theta);
    float half_width = aspect * half_height;
    hv3 eye = H(camera->eye_pos[0], camera->eye_pos[1], camera->eye_pos[2]);
    hv3 w = h_normalize(H(camera->eye_dir[0], camera->eye_dir[1], camera->eye_dir[2]));
    hv3 v = h_normalize(H(camera->up[0], camera->up[1], camera->up[2]));
    hv3 u = h_cross(w, v);
    hv3 llc = h_sub(h_sub(h_add(eye, h_scale(camera->focus_distance, w)), h_scale(half_width, u)),
                    h_scale(half_height, v));
    hv3 horizontal = h_scale(2.0f * half_width, u);
    hv3 vertical = h_scale(2.0f * half_height, v);
    memset(out, 0, sizeof *out);
    out->eye[0] = eye.x; out->eye[1] = eye.y; out->eye[2] = eye.z;
    out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    out->lens_radius = lens_radius;
    out->lower_left_corner[0] = llc.x; out->lower_left_corner[1] = llc.y; out->lower_left_corner[2] = llc.z;
    return MIRT_OK;
}

/* camera_orientation + renderer_camera (fly_camera.rs:52-64, 227-241) */
int mirt_oracle_camera_from_fly_pose(const float position[3], float yaw, float pitch,
                                     float vfov_degrees, float aperture, float focus_distance,
                                     MirtCamera* out)
{
    if (!position || !out) return MIRT_ERR_NULL_POINTER;
    hv3 forward = h_normalize(H(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)));
    hv3 world_up = H(0.0f, 1.0f, 0.0f);
    hv3 right = h_cross(forward, world_up);
    hv3 up = h_cross(right, forward);
    out->eye_pos[0] = position[0]; out->eye_pos[1] = position[1]; out->eye_pos[2] = position[2];
    out->eye_dir[0] = forward.x; out->eye_dir[1] = forward.y; out->eye_dir[2] = forward.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
    out->vfov_radians = mirt_oracle_degrees_to_radians(vfov_degrees);
    out->aperture = aperture;
    out->focus_distance = focus_distance;
    return MIRT_OK;
}

/* ---------------- math-spec exports ---------------- */

void mirt_oracle_math_sincos(const float* x, float* s, float* c, size_t n)
{
    for (size_t i = 0; i < n; ++i) om_sincos(x[i], &s[i], &c[i]);
}
void mirt_oracle_math_acos(const float* x, float* y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = om_acos(x[i]);
}
void mirt_oracle_math_atan2(const float* y, const float* x, float* r, size_t n)
{
    for (size_t i = 0; i < n; ++i) r[i] = om_atan2(y[i], x[i]);
}
void mirt_oracle_math_log2(const float* x, float* y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = om_log2(x[i]);
}
void mirt_oracle_math_exp2(const float* x, float* y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = om_exp2(x[i]);
}
void mirt_oracle_math_pow(const float* x, const float* y, float* r, size_t n)
{
    for (size_t i = 0; i < n; ++i) r[i] = om_pow_pos(x[i], y[i]);
}
