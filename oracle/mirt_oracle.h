/*
 * mirt_oracle.h — CPU oracle for the per-pixel ray-trace path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (weekend-raytracer-wgpu_amd/) never does and has no
 * CPU fallback.
 *
 * PARITY UNPINNED: the reference (linuxing3/weekend-raytracer-wgpu) holds no golden vectors,
 * known-answer tests or fixtures for this path — its only tests are five `Angle` unit tests
 * (src/raytracer/angle.rs:52-93) — and it cannot be compiled here (Rust; no toolchain, no
 * vendored crates, no network).  This oracle is therefore a restatement written from the
 * source text, pinned only by (i) those five Angle KATs and (ii) analytic properties derived
 * from the source (tests/test_oracle_kats.py).
 *
 * It shares include/mirt.h (the wire structs / params of the C ABI) with the product so that
 * both sides are driven with the same bytes; it shares no code.
 */
#ifndef MIRT_ORACLE_H
#define MIRT_ORACLE_H

#include "../include/mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* variant of the timed CPU loop */
enum {
    MIRT_ORACLE_CLEAN    = 0, /* no allocation in the loop: the fair algorithmic CPU number */
    MIRT_ORACLE_FAITHFUL = 1  /* parity mode only: reproduces the per-sample `world.clone()` x2
                                 malloc/free traffic of layer.rs:332,359 (leaks omitted) */
};

/* Same contract as mirt_ctx_render (host output, compact rows).  n_threads <= 0 = all cores. */
int mirt_oracle_render(const MirtScene* scene, const MirtParams* params, uint8_t* out_rgba8,
                       size_t out_len, int n_threads, int variant);

/* PT mode only: the exact per-pixel fixed-point radiance sums (3 x u64 per pixel, 2^-20 units)
 * that the resolve step averages — the fp32-intermediate check of the parity tests. */
int mirt_oracle_render_pt_sums(const MirtScene* scene, const MirtParams* params, uint64_t* out_sums,
                               size_t out_len_u64, int n_threads);

/* Work counters of the last mirt_oracle_render* call on this thread (same meaning as MirtStats;
 * kernel_ms = wall time of the pixel loop). */
int mirt_oracle_get_stats(MirtStats* out);

/* host-side set-up restatements */
int   mirt_oracle_camera_new(const MirtCamera* camera, uint32_t w, uint32_t h, MirtGpuCamera* out);
int   mirt_oracle_camera_from_fly_pose(const float position[3], float yaw_radians, float pitch_radians,
                                       float vfov_degrees, float aperture, float focus_distance,
                                       MirtCamera* out);
float mirt_oracle_degrees_to_radians(float degrees);
float mirt_oracle_radians_to_degrees(float radians);
int   mirt_oracle_validate_render_params(const MirtCamera* camera, const MirtSamplingParams* sampling,
                                         uint32_t w, uint32_t h);

/* elementary functions of mirt-math v1, vectorised over n (for the math-spec tests) */
void mirt_oracle_math_sincos(const float* x, float* s, float* c, size_t n);
void mirt_oracle_math_acos(const float* x, float* y, size_t n);
void mirt_oracle_math_atan2(const float* y, const float* x, float* r, size_t n);
void mirt_oracle_math_log2(const float* x, float* y, size_t n);
void mirt_oracle_math_exp2(const float* x, float* y, size_t n);
void mirt_oracle_math_pow(const float* x, const float* y, float* r, size_t n);
/* the PT RNG (raytracer.wgsl:493-521): first `n` floats of the stream of (pixel, sample, seed) */
void mirt_oracle_rng_stream(uint32_t pixel_index, uint32_t sample, uint64_t seed, float* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
