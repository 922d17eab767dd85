// mirt_kernels.h — launch interface between the C ABI (mirt_api.hip) and the gfx950 kernels
// (mirt_kernels.hip).  Plain structs, passed by value as kernel arguments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirt.h"

namespace mirt {

// A sphere as the kernels read it from LDS: the reference's 32-byte `Sphere`
// (src/raytracer/mod.rs:418-431) with its two padding words replaced by values every
// intersection needs.  rr = radius*radius and inv_r = 1.0f/radius are computed once on the host
// in IEEE binary32 — the same roundings the per-test expressions `sphere.radius * sphere.radius`
// (mod.rs:1135, wgsl:411) and `1.0 / radius` (mod.rs:1233, wgsl:433) produce.
struct PreparedSphere {
    float    cx, cy, cz, rr;
    float    inv_r, radius;
    uint32_t material_idx;
    uint32_t op;            // pool kernel: queue of its material's shading routine (RenderArgs.queue_routine)
};
static_assert(sizeof(PreparedSphere) == 32, "PreparedSphere must stay 2 x 16 B for ds_read_b128");

// A material as the path-traced kernels read it from LDS: `GpuMaterial` (mod.rs:757-765) plus
// values derived once on the host: 1/x (the dielectric's `1f / refractionIndex`, wgsl:261) and,
// for 1x1 textures (every colour material of the reference's scenes), the texel itself so that
// the common lookup needs no global-memory access.
//   tex[k] for a 1x1 texture: { r, g, b, bits(offset) };  otherwise { bits(w), bits(h), bits(offset), 0 }.
struct PreparedMaterial {
    uint32_t id;
    float    x;
    float    inv_x;
    uint32_t flags;            // bit 0: desc1 is 1x1, bit 1: desc2 is 1x1
    float    tex[2][4];
};
static_assert(sizeof(PreparedMaterial) == 48, "PreparedMaterial is 3 x 16 B");

// Many-sphere scenes (grid builds): what shading a hit on sphere i needs besides what the grid blob already holds in LDS (centre,
// routine id): 1/r and the sphere's material, in ONE 64-byte line of global memory -- no sphere -> material_idx -> material chain.
// The scatter step loads `a` and `t0` (two independent 16-byte loads; `t1` only in the checkerboard's case) and uses them after
// the routine's random draws (mirt_kernels.hip: shade_grid_hit).
struct ShadeRec {
    float a[4];        // {1/r, x, 1/x, bits(flags)}
    float t0[4];       // first texture (PreparedMaterial.tex[0])
    float t1[4];       // second texture (checkerboard only)
    float c[4];        // {centre, bits(id)}: not read by the kernels (centre and routine id come from the grid blob in LDS)
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec is one 64-byte line");

// Uniform grid over the small spheres of a many-sphere scene (>= kGridMinSpheres): the nearest-hit scan
// visits only the cells a ray crosses (3D-DDA) plus a short list of "big" spheres.  The grid is
// CONSERVATIVE (every sphere is listed in all cells its slightly enlarged bounding box touches) and the
// hit rule breaks ties by sphere index, so the result is identical to the reference's flat scan.
// Layout of the blob (all of it is staged into LDS; the kernels of a grid build read NO other sphere data from LDS):
//   GridHeader | big ids [n_big] u16 | FURTHER item ids [n_items] u16 (a cell's third, fourth ... sphere, ascending)
//   | routine of every sphere [n_spheres] u8 = min(GpuMaterial.id, 4): what a scatter step switches on
//   | 8-byte aligned: cell table [ncells] of two u32 = {first further item | item count << 16, id0 | id1 << 16}: the first two sphere ids
//     of a cell arrive WITH its entry, so their records are one LDS round trip away (entry -> records) instead of two (entry -> ids ->
//     records) -- most cells list at most two spheres (RTIOW: 0.84 tests per visited cell) -- round 4, -2 % on RTIOW
//   | 16-byte aligned: test records {cx, cy, cz, r^2} of the big spheres [n_big], in list order (a COPY: the always-tested list
//     needs no id -> record indirection) | of EVERY sphere [n_spheres], by sphere id.
// A record is the first half of the sphere's PreparedSphere.  (Round 2 stored a record per list ENTRY -- a sphere listed in k cells stored
// k times, 17 KB for RTIOW; stored once per sphere the cells can be made smaller without the blob outgrowing LDS: cell = 2.5 median radii
// instead of 4, -9 % on RTIOW.)
struct GridHeader {
    float    org[3];        // lower corner
    float    cell[3];       // cell size per axis
    float    inv_cell[3];
    uint32_t dims[3];
    uint32_t n_big;         // spheres tested for every ray (too large for the grid)
    uint32_t off_big;       // offsets in uint16 units from the start of the blob
    uint32_t off_cells;     // BYTE offset (multiple of 4) of the cell table
    uint32_t off_items;     // sphere ids, ascending within a cell
    uint32_t total_bytes;   // multiple of 16
    uint32_t off_ops;       // BYTE offset of the per-sphere routine-queue table
    uint32_t off_big_recs;  // BYTE offsets (multiples of 16) of the test records: the big spheres' copies ...
    uint32_t off_recs;      // ... and every sphere's, by id
    uint32_t n_entries;     // cell entries in total (a sphere counts once per cell that lists it)
    float    inv_dim_x;     // 1 / dims[0] and 1 / (dims[0] * dims[1]), IEEE quotients from the host: a parked walk's linear cell index is
    float    inv_dim_xy;    // taken apart with them (floor((n + 0.5) * inv) == n / d exactly for n, d <= 8192: tests/test_abi.py)
    uint32_t pad_;
};
static_assert(sizeof(GridHeader) % 16 == 0, "GridHeader is staged with 16-byte copies");
#ifndef MIRT_DISPENSER_STRIDE
#define MIRT_DISPENSER_STRIDE 1024           // u32 words between a launch's eight dispenser words: 4 KB, so that they sit in different memory channels
#endif
constexpr uint32_t kGridMinSpheres = 32;
constexpr uint32_t kGridMaxCells   = 4096;   // x 8 bytes per cell entry = the 32 KB of LDS that round 3's 8 192 four-byte entries could take: the pools keep their room
                                            // (a soup of 700 small spheres over a wide volume lost every pool geometry to a 64 KB cell table: soak seed 3150)

constexpr uint32_t kStripLevels   = 5;    // strip widths 16, 8, 4, 2, 1
constexpr uint32_t kStripPixels   = 16;   // strip kernels: pixels one wave owns per work unit (64-B RGBA8 store)
constexpr uint32_t kBlockThreads  = 256;  // strip kernels: 4 waves
#if defined(MIRT_DIAG_STEPS) || defined(MIRT_DIAG_STAMPS)   // diagnosis builds (tools/step_stats.py): ten more counters, printed by mirt_ctx_get_stats
constexpr uint32_t kNumCounters   = 31;
#else
constexpr uint32_t kNumCounters   = 18;   // u64 work counters (MirtStats order)
#endif
constexpr uint32_t kMaxLdsBytes   = 120 * 1024; // scene budget in LDS (of 160 KB per CU); sphere ids are 12 bit in the pool kernel

// pooled path-traced kernel: every wave keeps a pool of paths in LDS, queued by pending shading routine
constexpr uint32_t kDefaultPoolConfig = 0;
// MIRT_FLAG_TEXEL_TILES: the pool geometry with an LDS texel window per wave (render_pt_pool_tile_kernel)
#ifndef MIRT_TILE_SLOTS
#define MIRT_TILE_SLOTS 112
#endif
constexpr uint32_t kTilePoolConfig = 5, kTilePoolSlots = MIRT_TILE_SLOTS;
constexpr uint32_t kByPixelMaxSpp = 64;   // strip kernel: below this many samples per pixel a wave takes 64 pixels, lane = pixel (scenes with
                                          // one shading routine: at any sample count -- nothing diverges, 1.51 ms against 2.27 on config 2)
// pool kernel, grid build (many-sphere scenes): ONE 1024-thread block per CU, so that the grid blob (~20 KB for RTIOW)
// is staged once for all 16 waves and the rest of the 160 KB goes to the path pools.  Measured on RTIOW 1080p x 128 spp:
// 512 threads x 112 slots, two blocks 17.3 ms; x 128 slots 16.6; 1024 x 144 15.9; 1024 x 152 15.7; one 512-thread
// block per CU (8 waves) with 160-256 slots 26.3 -- the pools want to be as large as 16 resident waves allow.
// 160 slots are the first choice where a small blob leaves room for them (they cost nothing there); on RTIOW a blob shrunk until they
// fit measured +-0 against 152 (profiles/r04_c5_ab.txt block 4), which is why plan_grid coarsens cells only until 152 fit, not 160.
// (experiment builds: -DMIRT_GRID_THREADS=640 -DMIRT_GRID_BLOCKS=2 -DMIRT_GRID_MINW=5 "-DMIRT_GRID_SLOTS=104,96,88,80" = 20 waves per CU)
#ifndef MIRT_GRID_THREADS
#define MIRT_GRID_THREADS 1024
#define MIRT_GRID_BLOCKS 1
#define MIRT_GRID_MINW 4
#define MIRT_GRID_SLOTS 160, 152, 128, 96
#endif
constexpr uint32_t kGridPoolThreads = MIRT_GRID_THREADS;
constexpr uint32_t kGridPoolBlocksPerCu = MIRT_GRID_BLOCKS;    // blocks of the grid build that share a CU's LDS
constexpr uint32_t kGridPoolMinWaves = MIRT_GRID_MINW;         // waves per SIMD the build is held to (launch bounds)
constexpr uint32_t kGridPoolSlotChoices[4] = { MIRT_GRID_SLOTS };   // the largest geometry whose block fits LDS is taken
// Samples per pixel from which the pooled kernel is the default (below: the strip kernel, lane = pixel).  A 16-pixel strip
// of few samples cannot keep a pool full (about 25 steps of fill and drain per strip whatever it holds), so the thresholds are
// measured crossovers (tools/ab_libs.py with MIRT_FLAG_KERNEL_STRIP / _POOL, 1080p):
//   several shading routines (config 3): strip 1.58 / 1.73 / 1.94 ms at 36 / 40 / 44 spp, pool 1.61 / 1.67 / 1.74   -> 40 (round 2)
//     round 3 (sample groups; eight-word dispenser in both kernels): strip 0.98 / 1.13 / 1.24 / 1.37 / 1.54 ms at 24 / 28 / 32 / 36 / 40 spp, pool
//     1.00 / 1.11 / 1.21 / 1.30 / 1.41 (main.rs 5-sphere scene: 1.32 / 1.65 against 1.33 / 1.55 at 24 / 32 spp)                                   -> 28
//   ONE routine (config 2, no divergence for the lane-per-pixel kernel to lose): round 2 184; round 3: strip 0.93 / 1.78 / 3.55 / 5.21 / 6.98 /
//     8.48 ms at 100 / 200 / 400 / 600 / 800 / 1000 spp, pool 1.81 / 2.42 / 3.85 / 5.17 / 6.52 / 7.96                                           -> 600
//   many-sphere scenes (grid build, RTIOW): round 2: strip 2.09 / 4.11 / 7.69 ms at 8 / 16 / 32 spp, pool 2.31 / 3.14 / 4.66 -> 16;
//     round 3: strip 1.54 / 2.19 / 2.70 ms at 8 / 12 / 16 spp, pool 1.99 / 2.28 / 2.61                                                          -> 16
//   round 4 (lane-per-pixel launches run one unit per wave; profiles/r04_lowspp_ab.txt block 5): several routines: strip 0.96 / 1.10 / 1.27 /
//     1.52 ms at 24 / 28 / 32 / 40 spp, pool 1.01 / 1.12 / 1.20 / 1.39 (main.rs scene 1.26 / 1.66 against 1.32 / 1.56 at 24 / 32)             -> 32
//     ONE routine: strip 3.36 / 4.81 / 6.49 / 8.11 ms at 400 / 600 / 800 / 1000 spp, pool 3.80 / 5.25 / 6.52 / 7.93                           -> 800
//     with the streaming build of lane = pixel (from 16 spp on; block 9): several routines: 0.94 / 1.16 / 1.28 / 1.61 ms at 28 / 32 / 36 / 48 spp, pool
//     - / 1.17 / 1.24 / 1.50                                                                                                                  -> 32
//     ONE routine: 5.38 / 13.3 / 26.5 ms at 800 / 2000 / 4000 spp, pool 6.34 / 15.5 / 31.2; 3840x2160 x 1000 spp 26.3 against 30.5 -> frames of
//     >= 1 Mpixel never take the pool (mirt_api.hip: stream_any_spp); 800x600 x 1000 spp 2.31 against the pool's 2.17                          -> 800 below that
constexpr uint32_t kPoolMinSpp           = 32;
constexpr uint32_t kPoolMinSppOneRoutine = 800;
constexpr uint32_t kPoolMinSppGrid       = 16;

enum CounterSlot : uint32_t {
    kCntRays = 0, kCntTests, kCntRoots, kCntHits,
    kCntScatter0, kCntScatter1, kCntScatter2, kCntScatter3, kCntScatter4,
    kCntSky, kCntLaneIters, kCntWaveIters,
    kCntCells, kCntWaveCells,           // grid builds: cells visited by lanes / cell-loop iterations of waves
    kCntTexelsPrimary, kCntTileHitsPrimary,   // tile build: image-texel fetches of camera-ray hits / of those, served by the LDS tile
    kCntTexelsLater, kCntTileHitsLater        // ... of later bounces
};

struct RenderArgs {
    // The camera travels BY VALUE with every launch (96 B of kernel arguments, staged into LDS from the kernarg
    // segment): `Layer::update_camera` / `set_render_params` (layer.rs:188-193, mod.rs:353-388) change it every
    // interactive frame, and this way mirt_ctx_set_camera needs no device copy and no synchronisation, and a
    // launch already queued keeps the camera it was issued with.  MUST stay the first member (stage_scene).
    MirtGpuCamera           cam;
    const PreparedSphere*   spheres;
    const MirtMaterial*     mats;          // reference layout (parity kernel)
    const PreparedMaterial* pmats;         // derived layout (path-traced kernels)
    const float*            texels;
    const MirtSkyState*     sky;
    uint32_t*               out;           // compact RGBA8, one u32 per pixel
    unsigned long long*     counters;      // [kNumCounters] of THIS launch (one block per event slot), COUNT builds only
    const unsigned char*    grid;          // nullable: GridHeader + lists (strip kernel, GRID build)
    const ShadeRec*         shade;         // grid builds: [n_spheres] shading records (nullable otherwise)
    uint32_t                grid_bytes;
    uint32_t                grid_pool_slots;   // pool kernel, grid build: slots per wave of the geometry chosen on the host
    uint32_t                strip_cand;        // pool kernel, grid build: 1 = camera rays scan their strip's candidate list (0: A/B runs)
    uint32_t                grid_flat_y;       // grid builds: 1 = the grid is ONE cell high (dims[1] == 1: spheres on a ground plane) -> the 2-D walk
    uint32_t*               work_counter;  // dynamic work dispenser: THIS launch's own word (one per event slot), preset before the launch
    unsigned long long*     accum;         // nullable: [pixels][3] exact fixed-point sums to ADD into instead of resolving
    uint64_t                n_texels;
    uint32_t n_spheres, n_mats;
    uint32_t width, height, spp, num_bounces, flags, seed_mix, sample_begin;
    uint32_t frame_begin;                  // frames before this accumulation (MirtParams.frame_begin)
    uint32_t frame_spp;                    // > 0: the reference's per-frame RNG stream (MirtParams.frame_spp); lane-per-pixel schedule
    uint32_t row_begin, tile_rows, n_parts, part;
    uint32_t out_rows;                     // rows this launch writes
    uint32_t n_units;                      // work units (strips) the dispenser hands out
    uint32_t first_dispensed;              // the dispenser's words start at ZERO (pre-zeroed per launch slot, no memset node in front of the kernel):
                                           // units below this one -- one per launched wave -- are taken by wave index (first_unit)
    uint32_t disp_taken[8];                // eight-word dispenser: how many of word x's units (u = 8 k + x) those are
    uint32_t static_units;                 // lane-per-pixel kernels: no dispenser -- unit = wave index (+ the grid's waves, if the host launched fewer waves than units)
    uint32_t spread_units;                 // strip-type kernels: units from eight dispenser words (unit u <-> word u mod 8; next_unit_any)
    uint32_t px_groups_log2;               // lane-per-pixel strip kernel: a unit is 64 >> g pixels, their samples dealt to 1 << g groups of lanes
    // pool kernel, guided self-scheduling: level l = strips of (kStripPixels >> l) pixels; it starts at unit
    // lvl_unit[l] / pixel lvl_pix[l] (entry kStripLevels = end).  Strips shrink 16 -> 1 pixels towards the end of
    // the frame so that all waves finish within a fraction of a strip of each other.
    uint32_t lvl_unit[6];
    uint32_t lvl_pix[6];
    uint32_t queue_routine[5];             // pool kernel: scatter queue q runs routine queue_routine[q] = min(GpuMaterial.id, 4)
    uint32_t lds_bytes;
    uint32_t launch_threads;               // strip-type kernels: threads per block of THIS launch (0: kBlockThreads)
    uint32_t stream_samples;               // lane-per-pixel strip launch in a flat scene: 1 = render_pt_stream_kernel (a lane starts its next sample without waiting for the wave)
    float    sph3[12];                     // scenes of exactly three spheres: their {centre, r^2} records as kernel arguments (scalar loads)
};

struct DeinterleaveArgs {
    const uint32_t* parts;      // n_parts buffers back-to-back, part_stride_px pixels apart
    uint32_t*       out;        // band image [band_rows][width]
    uint64_t        part_stride_px;
    uint32_t        width, band_rows, tile_rows, n_parts;
};

// Where a render kernel is dispatched: the stream, and optionally the event pair of the launch.  Non-null events ride on the kernel
// dispatch itself (hipExtLaunchKernel attaches start / stop timestamps to the dispatch) instead of two event-record packets in the stream.
struct LaunchOn {
    hipStream_t stream = nullptr;
    hipEvent_t  begin = nullptr, end = nullptr;
    LaunchOn(hipStream_t s = nullptr, hipEvent_t b = nullptr, hipEvent_t e = nullptr) : stream(s), begin(b), end(e) {}
};

// launchers (mirt_kernels.hip).  That file is compiled twice: namespace exact_build (the default, bit-exact arithmetic;
// everything below) and namespace fast_build (mirt_kernels_fast.hip: MIRT_FLAG_FAST_MATH, hardware transcendentals;
// path-traced launchers and the host helpers only).
struct PoolConfig { uint32_t threads, slots, lds_bytes; };
namespace fast_build {
hipError_t launch_pt_strip(const RenderArgs& a, uint32_t grid_blocks, bool count, bool use_grid, bool by_pixel, LaunchOn stream);
uint32_t   strip_blocks_per_cu(bool hosek, bool count, bool use_grid, bool by_pixel, uint32_t lds_bytes, bool stream);
hipError_t launch_pt_pool(const RenderArgs& a, uint32_t grid_blocks, uint32_t cfg, bool count, uint32_t nq, LaunchOn stream);
}
namespace exact_build {
hipError_t launch_parity(const RenderArgs& a, uint32_t grid_blocks, bool count, bool by_pixel, LaunchOn stream);
hipError_t launch_pt_strip(const RenderArgs& a, uint32_t grid_blocks, bool count, bool use_grid, bool by_pixel, LaunchOn stream);
// blocks of the kernel such a launch runs that are resident per CU at once (hipOccupancyMaxActiveBlocksPerMultiprocessor)
uint32_t   strip_blocks_per_cu(bool hosek, bool count, bool use_grid, bool by_pixel, uint32_t lds_bytes, bool stream);
uint32_t   parity_blocks_per_cu(bool count, bool by_pixel, uint32_t lds_bytes);
uint32_t   pool_config_count();
PoolConfig pool_config(uint32_t i, uint32_t nq);
PoolConfig pool_config_grid(size_t lds_for_pools);   // grid build: slots by the LDS left beside scene + grid (slots = 0: none fits)
hipError_t launch_pt_pool(const RenderArgs& a, uint32_t grid_blocks, uint32_t cfg, bool count, uint32_t nq, LaunchOn stream);
// template arguments of the pool kernel launch_pt_pool would start: <threads, slots, min waves, COUNT, HOSEK, NQ, GRID, FLATY>
void       pool_kernel_name(const RenderArgs& a, uint32_t cfg, bool count, uint32_t nq, char* out, size_t out_len);
uint32_t pool_scatter_queues(uint32_t n_routines, bool count);
hipError_t launch_resolve(const unsigned long long* accum, uint32_t* out, uint64_t n_pixels, uint32_t n_samples,
                          uint32_t flags, hipStream_t stream);
hipError_t launch_selftest_math(unsigned long long* d_mismatches, hipStream_t stream);
hipError_t launch_deinterleave(const DeinterleaveArgs& a, hipStream_t stream);
size_t     scene_lds_bytes(uint32_t n_spheres, uint32_t n_mats, bool pt, bool hosek);
size_t     scene_lds_bytes_grid(uint32_t n_spheres, bool hosek);
}  // namespace exact_build

}  // namespace mirt
