/*
 * mirt.h — C ABI of the MI355X-native per-pixel sphere ray tracer.
 *
 * This is the drop-in boundary for ONE hot path of linuxing3/weekend-raytracer-wgpu:
 * the body of `Layer::set_data` (reference src/raytracer/layer.rs:264-282) and everything it
 * calls.  A Rust `Layer` keeps its method surface and replaces the pixel loop with one call
 * into this library (binding shown in INTEGRATION.md).  All structs below are byte-identical
 * to the reference's `#[repr(C)] + bytemuck::Pod` types, so a Rust `&[T]` can be passed as
 * `(ptr, len)` with no marshalling.
 *
 * Conventions
 *   - every entry point returns 0 (MIRT_OK) or a negative MirtStatus; nothing unwinds or
 *     aborts across the ABI; mirt_last_error() returns a thread-local message.
 *   - the caller owns every input pointer for the duration of the call only; the library
 *     never retains caller pointers.  Output buffers are caller-allocated.
 *   - images are row-major, top row first, RGBA8 (A = 255): the format the reference's
 *     consumer `Layer::register_texture` uploads (layer.rs:150-176, `to_rgba8`).
 *   - there is NO CPU fallback behind these symbols: without a HIP device every render call
 *     fails with MIRT_ERR_NO_DEVICE.
 *   - threading: a context is used from one host thread at a time.  Its render calls may be queued
 *     on DIFFERENT HIP streams and overlap on the device: every launch owns its work dispenser and
 *     work counters, and carries the camera it was issued with.  Two exceptions, both about the
 *     context's single accumulation buffer: mirt_ctx_accum_add calls must be ordered with respect to
 *     each other by the caller (same stream, or events), and mirt_ctx_set_scene waits for the device.
 *   - tuning knobs (MIRT_POOL_CONFIG, MIRT_GSS_*, ...) are read from the environment once, in
 *     mirt_ctx_create; render calls never consult the environment.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_VERSION_MAJOR 0
#define MIRT_VERSION_MINOR 4
#define MIRT_VERSION_PATCH 0

/* ------------------------------------------------------------------------------------------
 * Wire structs (reference file:line of the Rust type each one mirrors)
 * ---------------------------------------------------------------------------------------- */

/* `pub struct Sphere(glm::Vec4, f32, u32, [u32; 2])` — src/raytracer/mod.rs:418-431. 32 B. */
typedef struct MirtSphere {
    float    center[4];     /* xyz + w(=0), `glm::vec3_to_vec4(&center)` mod.rs:429 */
    float    radius;
    uint32_t material_idx;
    uint32_t _pad[2];
} MirtSphere;

/* `pub struct TextureDescriptor { width, height, offset }` — mod.rs:869-876. 12 B.
 * `offset` indexes the flat texel table in units of texels ([f32;3]).
 * The "empty" descriptor is {0, 0, 0xffffffff} (mod.rs:878-886). */
typedef struct MirtTextureDescriptor {
    uint32_t width;
    uint32_t height;
    uint32_t offset;
} MirtTextureDescriptor;

/* `struct GpuMaterial { id, desc1, desc2, x }` — mod.rs:757-765. 32 B.
 * id: 0 lambertian, 1 metal, 2 dielectric, 3 checkerboard (mod.rs:768-813).
 * x:  fuzz (metal) or refraction index (dielectric). */
typedef struct MirtMaterial {
    uint32_t              id;
    MirtTextureDescriptor desc1;
    MirtTextureDescriptor desc2;
    float                 x;
} MirtMaterial;

/* `pub struct GpuCamera` — mod.rs:681-697. 96 B. Produced on the host by
 * `GpuCamera::new` (mod.rs:700-741) == mirt_camera_new() below. */
typedef struct MirtGpuCamera {
    float eye[3];               float _padding1;
    float horizontal[3];        float _padding2;
    float vertical[3];          float _padding3;
    float u[3];                 float _padding4;
    float v[3];                 float lens_radius;
    float lower_left_corner[3]; float _padding5;
} MirtGpuCamera;

/* `struct GpuSkyState` — mod.rs:888-896. 144 B. Opaque Hosek-Wilkie state produced by the
 * hw-skymodel crate on the Rust side (mod.rs:567-595); consumed by raytracer.wgsl:316-343. */
typedef struct MirtSkyState {
    float    params[27];
    float    radiances[3];
    uint32_t _padding[2];
    float    sun_direction[4];
} MirtSkyState;

/* `pub struct Camera { eye_pos, eye_dir, up, vfov: Angle, aperture, focus_distance }` —
 * mod.rs:489-499 (Angle is a newtype over f32 radians, angle.rs:1-4). 48 B. */
typedef struct MirtCamera {
    float eye_pos[3];
    float eye_dir[3];
    float up[3];
    float vfov_radians;
    float aperture;
    float focus_distance;
} MirtCamera;

/* `pub struct SamplingParams` — mod.rs:597-603. 12 B. */
typedef struct MirtSamplingParams {
    uint32_t max_samples_per_pixel;
    uint32_t num_samples_per_pixel;
    uint32_t num_bounces;
} MirtSamplingParams;

/* The state `Layer` holds when `set_data` runs (layer.rs:37-46): camera, world (gathered from
 * `Vec<Box<Sphere>>` into one contiguous slice), material_data, global_texture_data. */
typedef struct MirtScene {
    const MirtGpuCamera* camera;
    const MirtSphere*    spheres;
    uint32_t             n_spheres;
    const MirtMaterial*  materials;
    uint32_t             n_materials;
    const float*         texels;      /* [n_texels][3] f32, `Vec<[f32;3]>` layer.rs:44 */
    uint64_t             n_texels;
    const MirtSkyState*  sky;         /* nullable; required only with MIRT_FLAG_SKY_HOSEK */
} MirtScene;

/* Render modes. */
enum {
    /* `layer.rs` semantics, bit-faithful, every quirk preserved. */
    MIRT_MODE_PARITY = 0,
    /* Full path tracer with the behaviours of raytracer.wgsl:105-521. */
    MIRT_MODE_PT = 1
};

/* Most samples per pixel ONE render / accum_add call takes (MIRT_ERR_SPP_RANGE above): the kernels count the work items of a unit --
 * up to 64 pixels x spp -- in 32 bits.  The reference's UI offers 1..=64 per frame and accumulates to max_samples_per_pixel (mod.rs:599-613). */
#define MIRT_MAX_SPP_PER_CALL (1u << 24)

/* Flags (MIRT_MODE_PT only, except MIRT_FLAG_COUNT_WORK which both modes honour). */
enum {
    MIRT_FLAG_SKY_HOSEK      = 1u << 0, /* sky = Hosek-Wilkie blob (wgsl:154-166,316-343); default: RTIOW gradient */
    MIRT_FLAG_NO_TONEMAP     = 1u << 1, /* skip uncharted2 (wgsl:83-103) */
    MIRT_FLAG_NO_SRGB        = 1u << 2, /* skip the sRGB OETF the Bgra8UnormSrgb surface applies (main.rs:465) */
    MIRT_FLAG_COUNT_WORK     = 1u << 3, /* run the counting build of the kernel: fills MirtStats work counters */
    /* Scheduling of the path-traced kernel (the image is bit-identical either way).  Default: the pooled kernel
     * (paths queued by shading routine in LDS) from a measured number of samples per pixel on -- 32 for scenes with
     * several shading routines, 16 for many-sphere scenes; scenes with ONE routine only on frames below 1 Mpixel, from
     * 800 (csrc/mirt_kernels.h, kPoolMinSpp*) -- and the strip kernel below: lane = pixel (from 16 samples per pixel on
     * in flat scenes its streaming build, render_pt_stream_kernel), lane = sample for counting launches.
     * In parity mode MIRT_FLAG_KERNEL_STRIP forces the lane = sample schedule (default: lane = pixel, the shape of the
     * reference's 2-spp operating point, at every sample count); the image is the same. */
    MIRT_FLAG_KERNEL_STRIP   = 1u << 4, /* force the strip kernel (wave = 64 samples of one pixel) */
    MIRT_FLAG_KERNEL_POOL    = 1u << 5, /* force the pooled kernel */
    MIRT_FLAG_NO_GRID        = 1u << 6, /* many-sphere scenes: scan the flat sphere list instead of the uniform grid */
    /* With MIRT_FLAG_COUNT_WORK on a many-sphere scene: count the work of the grid build that renders such
     * scenes in production (sphere_tests = tests a lane really performs, grid_cells) instead of falling back to
     * the reference's flat scan, whose counters are the oracle's. */
    MIRT_FLAG_COUNT_GRID     = 1u << 7,
    /* OPT-IN fast-math build of the path-traced kernels: hardware v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp /
     * v_log (about 1 ulp) and contraction instead of the bit-exact arithmetic.  Same RNG streams and exact integer
     * accumulation, so the image is deterministic, but pixels may differ from the default build by a few units in
     * the last place: NO parity claim holds with this flag.  Ignored by counting launches and in parity mode. */
    MIRT_FLAG_FAST_MATH      = 1u << 8,
    /* OPT-IN "LDS texel tiles" (BASELINE configs[3]): scenes with an image texture run the pooled kernel's tile build
     * (render_pt_pool_tile_kernel): every wave keeps a 16 x 2 texel window of the texture in LDS, centred on the first
     * image fetch of its strip, and serves the fetches that fall into it from there.  Same texel values, so the image
     * is IDENTICAL to the default build's; measured 3 % slower on config 4, hence opt-in.  A hint: it has
     * no effect where the pooled kernel does not run (few samples per pixel, many-sphere scenes, no image texture).
     * With MIRT_FLAG_COUNT_WORK it fills MirtStats.texel_fetches / texel_tile_hits. */
    MIRT_FLAG_TEXEL_TILES    = 1u << 9
};

/* What one render call computes.  The image is `width x height`; this call renders the rows
 * selected by (row_begin,row_end) and the tile interleave, into a COMPACT buffer of
 * mirt_params_out_rows() rows.  With tile_rows == 0 (or n_parts <= 1) the rows are the
 * contiguous band [row_begin,row_end).  With n_parts > 1 the band is cut into tiles of
 * `tile_rows` rows and this call renders tiles t with t % n_parts == part, in order — the
 * partition used to balance the 8 GPUs of a node. */
typedef struct MirtParams {
    uint32_t width;        /* `vp_size[0] as u32` layer.rs:270-275 */
    uint32_t height;
    uint32_t spp;          /* `sampling.num_samples_per_pixel` layer.rs:316; 1 .. MIRT_MAX_SPP_PER_CALL (progressive accumulation adds calls up) */
    uint32_t num_bounces;  /* PT mode; SamplingParams.num_bounces mod.rs:602 */
    uint32_t mode;         /* MIRT_MODE_* */
    uint32_t flags;        /* MIRT_FLAG_* */
    uint64_t seed;         /* PT mode RNG seed; 0 reproduces the WGSL stream (wgsl:498-502) */
    uint32_t row_begin;    /* band; row_end == 0 means `height` */
    uint32_t row_end;
    uint32_t tile_rows;    /* 0 = no interleave */
    uint32_t n_parts;
    uint32_t part;
    uint32_t sample_begin; /* PT: first sample index (progressive accumulation); normally 0 */
    /* PT: 0 (default) = every sample has its own RNG stream, that of frame_number = sample + 1 at one sample per
     * frame -- the image then does not depend on how samples are grouped into launches.  n > 0 = the REFERENCE's
     * stream for `num_samples_per_pixel` = n: frame f = sample / n + 1 seeds the pixel's RNG once (initRng,
     * wgsl:498-502) and its n samples draw from that one stream in turn (samplePixel, wgsl:105-122), exactly as
     * Raytracer::render_frame advances (mod.rs:303-351, 626-670).  Needs spp % n == 0 and sample_begin % n == 0;
     * runs on the lane-per-pixel schedule (a pixel's samples are then sequentially dependent).  Frame numbers count from
     * sample 0 of the accumulation plus `frame_begin` (below): the reference's frame_number survives
     * render_progress.reset() (mod.rs:284, 385), so accumulations after the first start at a later frame. */
    uint32_t frame_spp;
    /* PT with frame_spp > 0: the frames the reference's Raytracer had rendered BEFORE sample 0 of this accumulation, i.e. its
     * `frame_number - 1` at the reset -- `frame_number` starts at 1, advances with every render_frame call (completed frames
     * included) and survives `render_progress.reset()` (mod.rs:284, 350, 385).  Sample s then draws from the stream of frame
     * frame_begin + s / frame_spp + 1.  0 = a freshly constructed Raytracer.  Ignored when frame_spp == 0. */
    uint32_t frame_begin;
} MirtParams;

/* Work counters of the last render on a context (rocprof-independent).  The ray/test/scatter
 * counters are filled only when MIRT_FLAG_COUNT_WORK was set.  Parity mode counts what the reference's
 * sequential sample loop executes (it returns at the first terminating sample, layer.rs:333-374):
 * lane_iterations = samples executed, scatter[1] = scatter_metal calls. */
typedef struct MirtStats {
    double   kernel_ms;        /* hipEvent time of the render kernel of the LAST render call */
    double   kernel_ms_total;  /* summed over every render call since the previous mirt_ctx_get_stats */
    uint64_t launches;         /* number of render calls in kernel_ms_total */
    uint64_t samples;          /* pixels x spp rendered by the last call */
    uint64_t rays;             /* rays traced (primary + scattered) */
    uint64_t sphere_tests;     /* ray-sphere discriminant evaluations */
    uint64_t roots;            /* quadratic roots evaluated */
    uint64_t hits;             /* hit records built */
    uint64_t scatter[5];       /* lambertian, metal, dielectric, checkerboard, missing-material */
    uint64_t sky_misses;       /* rays that left the scene */
    uint64_t lane_iterations;  /* PT: active lanes summed over bounce-loop iterations */
    uint64_t wave_iterations;  /* PT: bounce-loop iterations summed over waves (x64 = lane slots) */
    uint64_t grid_cells;       /* MIRT_FLAG_COUNT_GRID: grid cells visited, summed over lanes */
    uint64_t grid_wave_cells;  /* MIRT_FLAG_COUNT_GRID: cell-loop iterations summed over waves (x64 = lane slots) */
    uint64_t texel_fetches[2];   /* MIRT_FLAG_TEXEL_TILES: image-texel fetches at [0] camera-ray hits, [1] later bounces */
    uint64_t texel_tile_hits[2]; /* ... of those, served by the wave's LDS tile */
} MirtStats;

typedef enum MirtStatus {
    MIRT_OK                      =  0,
    /* mirrors RenderParamsValidationError, mod.rs:396-411 / RenderParams::validate mod.rs:450-484 */
    MIRT_ERR_MAX_SAMPLES_MULTIPLE = -1, /* MaxSampleCountNotMultiple */
    MIRT_ERR_VIEWPORT_SIZE        = -2, /* ViewportSize */
    MIRT_ERR_VFOV_RANGE           = -3, /* VfovOutOfRange */
    MIRT_ERR_APERTURE_RANGE       = -4, /* ApertureOutOfRange */
    MIRT_ERR_FOCUS_DISTANCE       = -5, /* FocusDistanceOutOfRange */
    MIRT_ERR_SKY                  = -6, /* HwSkyModelValidationError / sky blob missing */
    /* library-level */
    MIRT_ERR_NULL_POINTER         = -10,
    MIRT_ERR_SPP_ZERO             = -11,
    MIRT_ERR_BAD_MODE             = -12,
    MIRT_ERR_BAD_ROWS             = -13,
    MIRT_ERR_MATERIAL_INDEX       = -14, /* sphere.material_idx >= n_materials, or parity mode with n_materials < 3 (layer.rs:345-349 reads material_data[2]) */
    MIRT_ERR_TEXEL_RANGE          = -15, /* a descriptor reaches past n_texels */
    MIRT_ERR_OUT_BUFFER           = -16, /* out_len too small */
    MIRT_ERR_NO_SCENE             = -17,
    /* The kernels keep the scene in LDS (160 KB per CU, 120 KB = 122 880 B of it for the scene): a scene is accepted if one
     * of two layouts fits, else mirt_ctx_set_scene fails with this code.  (The reference's own buffers are bounded by wgpu's
     * 512 MiB storage-buffer limit, main.rs:421-481; BASELINE's largest scene has 484 spheres.)
     *   flat layout (every mode and flag): 96 + 32 * n_spheres + 48 * n_materials + 144 <= 122 880 bytes
     *       -- e.g. 3 831 spheres with one material;
     *   grid layout (MIRT_MODE_PT only, scenes of >= 32 spheres of which at most 64 exceed 4 median radii): n_spheres <= 4 095
     *       (12-bit sphere ids in a path's state word), <= 4 096 cells and <= 65 535 cell entries (the host starts at a cell of
     *       2.5 median radii and coarsens it, x 1.26 per step up to 4, while either limit is exceeded or the blob would cost the
     *       pooled kernel its 152-slot geometry), and 240 + blob <= 122 880 bytes with blob ~ 17 * n_spheres + 8 * cells +
     *       2 * (entries beyond the first two of each cell).  A blob that leaves the pooled kernel no
     *       room for its path pools (mirt_grid_plan: pool_slots = 0) runs the strip kernel's grid build.  mirt_grid_plan() answers "which grid would this scene get" without a device.
     * A scene that fits ONLY the grid layout renders in path-traced mode with default flags; a render call in parity mode, or
     * with MIRT_FLAG_COUNT_WORK without MIRT_FLAG_COUNT_GRID, or with MIRT_FLAG_NO_GRID, returns this code. */
    MIRT_ERR_SCENE_TOO_LARGE      = -18,
    MIRT_ERR_FRAME_SPP            = -19, /* frame_spp does not divide spp / sample_begin */
    MIRT_ERR_NO_DEVICE            = -20,
    MIRT_ERR_HIP                  = -21,
    MIRT_ERR_ALLOC                = -22,
    MIRT_ERR_SPP_RANGE            = -24, /* spp > MIRT_MAX_SPP_PER_CALL, or sample_begin (the samples accumulated so far) + spp reaches 2^32 */
    MIRT_ERR_IMAGE_DECODE         = -23  /* mirt_jpeg_*: not a JPEG this decoder handles (TextureError::ImageLoadError, texture.rs:193-199) */
} MirtStatus;

typedef struct MirtContext MirtContext; /* opaque: one per device; owns all device memory */

/* ------------------------------------------------------------------------------------------
 * Entry points
 * ---------------------------------------------------------------------------------------- */

/* Packed version (major<<16 | minor<<8 | patch). */
uint32_t    mirt_version(void);
/* Thread-local description of the last failure on this thread ("" if none). */
const char* mirt_last_error(void);
/* Static name of a status code. */
const char* mirt_status_string(int status);

/* `RenderParams::validate` (mod.rs:450-484) over the fields that reach this path. */
int mirt_validate_render_params(const MirtCamera* camera, const MirtSamplingParams* sampling,
                                uint32_t viewport_w, uint32_t viewport_h);

/* `GpuCamera::new(&camera, viewport_size)` (mod.rs:700-741).  Host-side set-up arithmetic in
 * f32 exactly as the reference does it; runs once per image, not part of the hot loop. */
int mirt_camera_new(const MirtCamera* camera, uint32_t viewport_w, uint32_t viewport_h,
                    MirtGpuCamera* out);

/* `camera_orientation` + `FlyCameraController::renderer_camera` (fly_camera.rs:52-64,227-241):
 * pose (position, yaw, pitch in radians) -> Camera. */
int mirt_camera_from_fly_pose(const float position[3], float yaw_radians, float pitch_radians,
                              float vfov_degrees, float aperture, float focus_distance,
                              MirtCamera* out);

/* `Angle::degrees(d).as_radians()` (angle.rs:8-12): d * PI / 180 in f32. */
float mirt_degrees_to_radians(float degrees);
/* `Angle::as_degrees` (angle.rs:20-22): r * 180 / PI in f32. */
float mirt_radians_to_degrees(float radians);

/* Number of rows a render call with these params writes (0 on invalid params). */
uint32_t mirt_params_out_rows(const MirtParams* params);
/* Absolute image row of compact output row `i` (UINT32_MAX if out of range). */
uint32_t mirt_params_out_row_index(const MirtParams* params, uint32_t i);

/* The uniform grid mirt_ctx_set_scene would build over these spheres (many-sphere scenes, path-traced mode): HOST-ONLY -- no
 * device, no context.  `lds_bytes_per_block` = LDS a workgroup may use (0 = gfx950's 163 840).  cell_factor = the cell edge in
 * median radii the coarsening rule settles on (0 = no grid: fewer than 32 spheres, nothing small enough to bin, more than 64
 * big spheres, or more than 65 535 cell entries even at the coarsest cell); pool_slots = path slots per wave the pooled kernel's
 * grid build would run beside that blob (0 = it does not fit: strip kernel). */
typedef struct MirtGridPlan {
    float    cell_factor;
    uint32_t blob_bytes;
    uint32_t n_cells;
    uint32_t n_entries;
    uint32_t n_big;
    uint32_t pool_slots;
} MirtGridPlan;
int mirt_grid_plan(const MirtSphere* spheres, uint32_t n_spheres, uint64_t lds_bytes_per_block, MirtGridPlan* out);

/* Create / destroy a context on HIP device `device` (ordinal as seen by this process). */
int  mirt_ctx_create(int device, MirtContext** out);
void mirt_ctx_destroy(MirtContext* ctx);

/* Validate and upload the scene into device memory owned by the context (what
 * `Layer::new` + `set_global_data` leave in `self`, layer.rs:49-148).  The scene stays
 * resident across render calls until replaced. */
int mirt_ctx_set_scene(MirtContext* ctx, const MirtScene* scene);
/* Replace only the camera (`Layer::update_camera`, layer.rs:188-193; `Raytracer::set_render_params`,
 * mod.rs:353-388 — every interactive frame in the reference).  Host-side only: the camera travels by
 * value with each launch, so this neither copies to the device nor synchronises; launches already
 * queued keep their camera, the next render call uses the new one. */
int mirt_ctx_set_camera(MirtContext* ctx, const MirtGpuCamera* camera);

/* Render into HOST memory: kernel + D2H copy, blocking.  `out_rgba8` receives
 * mirt_params_out_rows() * width * 4 bytes.  This is the `set_data` replacement. */
int mirt_ctx_render(MirtContext* ctx, const MirtParams* params, uint8_t* out_rgba8, size_t out_len);

/* Render into DEVICE memory `d_out_rgba8` (same layout) asynchronously on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the context's own non-blocking stream; to run on the
 * default stream pass hipStreamLegacy, not 0).  No host sync: the caller orders later work on
 * the same stream (RCCL gather, D2H).  Used by bench.py and
 * the multi-GPU path so the framebuffer never leaves HBM before the collective. */
int mirt_ctx_render_device(MirtContext* ctx, const MirtParams* params, void* d_out_rgba8,
                           size_t out_len, void* hip_stream);

/* Name of the render kernel the LAST render call on this context launched (the schedule is chosen per call from
 * mode, spp, scene and flags), e.g. "render_pt_pool_kernel<256,112,6,false,false,3,false>" — the template
 * arguments as they appear in rocprofv3 kernel traces, without the `u` suffixes.  "" before the first render.
 * The pointer stays valid until the next render call or mirt_ctx_destroy. */
const char* mirt_ctx_last_kernel(const MirtContext* ctx);

/* Block until everything the context queued has finished. */
int mirt_ctx_synchronize(MirtContext* ctx);
/* Two streams of the context on DIFFERENT hardware queues, for hosts that keep two frames in flight -- a swap chain's double buffering:
 * queue frame k on stream k & 1 into framebuffer k & 1 (mirt_ctx_render_device) and the head of frame k + 1 fills the GPU while the tail
 * of frame k drains: 1080p, 2 spp 98.9 -> 83.9 us per frame; `Layer::set_data` at 800x600, 2 spp 12.8 -> 7.7 us.  index 0 is the context's
 * own stream (what a NULL hip_stream means), index 1 a stream of the device's highest priority, created on first use -- HIP keeps
 * hardware queues per priority, so the pair never shares one, which two streams of the caller's may (they then run one after the other).
 * *out_hip_stream is a hipStream_t owned by the context: pass it as `hip_stream`, wait on it with hipStreamSynchronize or
 * mirt_ctx_synchronize; it dies with the context.  Frames in flight must not share a framebuffer; mirt_ctx_accum_add calls stay ordered
 * among themselves (one stream).  MIRT_ERR_BAD_ROWS for index > 1. */
int mirt_ctx_frame_stream(MirtContext* ctx, uint32_t index, void** out_hip_stream);
/* Kernel timing on (the default) or off.  ON: every launch carries a start and an end event (on the kernel dispatch itself) and
 * mirt_ctx_get_stats reports kernel times.  OFF: a launch carries no event unless the context itself must learn that it has finished
 * (launches that take work units from the dispenser, counting launches): mirt_ctx_get_stats then counts launches but reports 0 ms for
 * them.  For hosts that queue frame after frame (the reference's render loop: `Raytracer::render_frame`, mod.rs:303-351;
 * `Layer::set_data` per resize): one `set_data` at the reference's operating point (800x600, 2 spp) takes 12.9 us per launch instead
 * of 17.9.  mirt_ctx_synchronize / mirt_ctx_set_scene / mirt_ctx_destroy still wait for such launches (through their streams). */
int mirt_ctx_set_timing(MirtContext* ctx, int enabled);
/* Stats of the last completed render call (synchronises the context first). */
int mirt_ctx_get_stats(MirtContext* ctx, MirtStats* out);

/* ---- progressive accumulation: what `RenderProgress::next_frame` (mod.rs:615-679) and the f32
 * image buffer of fsMain (wgsl:61-80) do across frames.  The context owns one accumulation buffer of
 * EXACT 64-bit fixed-point sums (3 per pixel, 2^-20 units), so frames add up bit-identically to
 * a single launch with the total sample count.  Path-traced mode only. ---- */

/* (Re)size the buffer for the rows `params` selects and clear it (`clear_accumulated_samples`). */
int mirt_ctx_accum_reset(MirtContext* ctx, const MirtParams* params);
/* Render params->spp further samples of every pixel (samples accumulated so far .. +spp of the RNG
 * stream; params->sample_begin is ignored) and add them to the buffer.  Asynchronous on `hip_stream`
 * (NULL = the context's stream). */
int mirt_ctx_accum_add(MirtContext* ctx, const MirtParams* params, void* hip_stream);
/* Samples per pixel accumulated since the last reset (`accumulated_samples_per_pixel`). */
uint32_t mirt_ctx_accum_samples(const MirtContext* ctx);
/* Resolve the buffer (mean over the accumulated samples, tone curves per params->flags) into
 * host memory, RGBA8; blocking.  Ordered after every mirt_ctx_accum_add issued so far, whatever
 * stream it was queued on (so is mirt_ctx_accum_read). */
int mirt_ctx_accum_resolve(MirtContext* ctx, const MirtParams* params, uint8_t* out_rgba8, size_t out_len);
/* Copy the raw sums to the host: pixels x 3 uint64 (the fp32-intermediate view used by tests). */
int mirt_ctx_accum_read(MirtContext* ctx, uint64_t* out_sums, size_t out_len_u64);

/* Device self-test of the arithmetic contract: the kernels' fast sqrt and reciprocal sequences
 * (hardware seed + one fma correction) are compared with the compiler's correctly rounded IEEE
 * expansions over ALL 2^32 binary32 bit patterns.  Writes the two mismatch counts (sqrt, 1/x);
 * both must be 0 for the bit-parity claim to hold on this device. */
int mirt_ctx_selftest_math(MirtContext* ctx, uint64_t out_mismatches[2]);

/* One-shot convenience: create context on `device`, set scene, render to host, destroy.
 * Signature a Rust `set_data` binds when it does not keep a context. */
int mirt_render(const MirtScene* scene, const MirtParams* params, int device,
                uint8_t* out_rgba8, size_t out_len);

/* RGBA8 -> RGB8 view for `Layer::imgbuf()` (layer.rs:182-186; `ImageBuffer<Rgb<u8>>`). Host-side
 * repack of n_pixels pixels. */
int mirt_rgba8_to_rgb8(const uint8_t* rgba, size_t n_pixels, uint8_t* rgb);

/* `Texture::new_from_image` (texture.rs:21-46): JPEG file bytes -> RGB8 -> `inv_255 * (p as f32)` texels, host side, so
 * that a host gets from the reference's `assets/earthmap.jpeg` / `assets/moon.jpeg` to MirtScene.texels with this
 * library alone.  Baseline and progressive Huffman JPEG, 8 bit, grey or 3 components, 4:4:4 / 4:2:2 / 4:2:0 (csrc/mirt_jpeg.cpp);
 * other files fail with MIRT_ERR_IMAGE_DECODE and a message in mirt_jpeg_last_error().  The decode is bit-identical to
 * libjpeg-turbo's (tests/test_jpeg.py); against the `image` crate it is decoder-unpinned.  No device involved. */
int mirt_jpeg_info(const uint8_t* data, size_t len, uint32_t* width, uint32_t* height);
int mirt_jpeg_decode_rgb8(const uint8_t* data, size_t len, uint8_t* rgb, size_t rgb_len);   /* rgb_len >= width * height * 3 */
int mirt_rgb8_to_texels(const uint8_t* rgb, size_t n_pixels, float* texels);               /* texels: n_pixels x [f32;3] */
const char* mirt_jpeg_last_error(void);

/* Reassemble a full image from the `n_parts` compact buffers produced with the tile
 * interleave (root side of the multi-GPU gather).  `parts` holds n_parts buffers
 * back-to-back, each padded to `part_stride` bytes.  Device-side: both pointers are
 * device memory; runs on `hip_stream`. */
int mirt_ctx_deinterleave_device(MirtContext* ctx, const MirtParams* params, const void* d_parts,
                                 size_t part_stride, void* d_out_rgba8, size_t out_len,
                                 void* hip_stream);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(MirtSphere) == 32, "Sphere is 32 B (mod.rs:418-421)");
static_assert(sizeof(MirtTextureDescriptor) == 12, "TextureDescriptor is 12 B (mod.rs:869-876)");
static_assert(sizeof(MirtMaterial) == 32, "GpuMaterial is 32 B (mod.rs:757-765)");
static_assert(sizeof(MirtGpuCamera) == 96, "GpuCamera is 96 B (mod.rs:681-697)");
static_assert(sizeof(MirtSkyState) == 144, "GpuSkyState is 144 B (mod.rs:888-896)");
static_assert(sizeof(MirtCamera) == 48, "Camera is 12 f32 (mod.rs:489-499)");
#else
_Static_assert(sizeof(MirtSphere) == 32, "Sphere is 32 B (mod.rs:418-421)");
_Static_assert(sizeof(MirtTextureDescriptor) == 12, "TextureDescriptor is 12 B (mod.rs:869-876)");
_Static_assert(sizeof(MirtMaterial) == 32, "GpuMaterial is 32 B (mod.rs:757-765)");
_Static_assert(sizeof(MirtGpuCamera) == 96, "GpuCamera is 96 B (mod.rs:681-697)");
_Static_assert(sizeof(MirtSkyState) == 144, "GpuSkyState is 144 B (mod.rs:888-896)");
_Static_assert(sizeof(MirtCamera) == 48, "Camera is 12 f32 (mod.rs:489-499)");
#endif

#endif /* MIRT_H */
