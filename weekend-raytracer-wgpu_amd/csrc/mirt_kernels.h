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
    uint32_t _pad;
};
static_assert(sizeof(PreparedSphere) == 32, "PreparedSphere must stay 2 x 16 B for ds_read_b128");

constexpr uint32_t kStripPixels   = 16;   // pixels one wave owns per work unit -> one 64-B coalesced RGBA8 store
constexpr uint32_t kBlockThreads  = 256;  // 4 waves
constexpr uint32_t kNumCounters   = 16;   // u64 work counters (MirtStats order)
constexpr uint32_t kMaxLdsBytes   = 64 * 1024;  // scene budget in LDS (2 blocks/CU stay resident)

enum CounterSlot : uint32_t {
    kCntRays = 0, kCntTests, kCntRoots, kCntHits,
    kCntScatter0, kCntScatter1, kCntScatter2, kCntScatter3, kCntScatter4,
    kCntSky, kCntLaneIters, kCntWaveIters
};

struct RenderArgs {
    const MirtGpuCamera*  cam;
    const PreparedSphere* spheres;
    const MirtMaterial*   mats;
    const float*          texels;
    const MirtSkyState*   sky;
    uint32_t*             out;           // compact RGBA8, one u32 per pixel
    unsigned long long*   counters;      // [kNumCounters], COUNT builds only
    uint32_t*             work_counter;  // dynamic strip dispenser, zeroed before every launch
    uint64_t              n_texels;
    uint32_t n_spheres, n_mats;
    uint32_t width, height, spp, num_bounces, flags, seed_mix, sample_begin;
    uint32_t row_begin, tile_rows, n_parts, part;
    uint32_t out_rows;                   // rows this launch writes
    uint32_t n_strips;                   // ceil(out_rows*width / kStripPixels)
    uint32_t lds_bytes;
};

struct DeinterleaveArgs {
    const uint32_t* parts;      // n_parts buffers back-to-back, part_stride_px pixels apart
    uint32_t*       out;        // band image [band_rows][width]
    uint64_t        part_stride_px;
    uint32_t        width, band_rows, tile_rows, n_parts;
};

// launchers (mirt_kernels.hip)
hipError_t launch_parity(const RenderArgs& a, uint32_t grid_blocks, hipStream_t stream);
hipError_t launch_pt(const RenderArgs& a, uint32_t grid_blocks, bool count, hipStream_t stream);
hipError_t launch_deinterleave(const DeinterleaveArgs& a, hipStream_t stream);
size_t     scene_lds_bytes(uint32_t n_spheres, uint32_t n_mats, bool hosek);

}  // namespace mirt
