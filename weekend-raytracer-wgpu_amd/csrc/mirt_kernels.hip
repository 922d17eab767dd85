// mirt_kernels.hip — the gfx950 (CDNA4, wave64) kernels of the per-pixel ray-trace path.
//
// A LANE CARRIES ONE PIXEL-SAMPLE AT A TIME.  The scene (camera, sphere list, material table) is
// staged once per block into LDS; all lanes read the same sphere at the same time, which LDS
// serves as a broadcast.  Wave ballots give the uniform loop exits and resolve the sequential
// semantics of the reference's sample loop.
//
//   render_parity_kernel   reference src/raytracer/layer.rs:264-444 semantics, bit-faithful, no fma.  Two schedules
//                          (BY_PIXEL): lane = pixel below 64 samples per pixel (the reference runs 2), lane = sample above.
//   render_pt_strip_kernel behaviours of src/raytracer/raytracer.wgsl:50-521; a wave owns a unit of pixels.  BY_PIXEL:
//                          lane = pixel, every lane walks its pixel's samples (few samples per pixel; any count in scenes
//                          with one shading routine); else the 64 lanes take samples lane, lane+64, … of one pixel.
//   render_pt_pool_kernel  the same arithmetic, scheduled differently: every wave keeps a pool of
//                          paths in LDS, queued by the shading routine they wait for, and always
//                          runs ONE routine on up to 64 of them — the material switch no longer
//                          serialises inside a wave (default from 28 / 600 / 16 samples per pixel on for scenes
//                          with several / one shading routine / many spheres: mirt_kernels.h, kPoolMinSpp*).
//
// Every pixel's radiance is summed in 64-bit fixed point (exact, order-independent), so the three
// schedules — and any GPU count — give bit-identical images.
//
// Compiled with -ffp-contract=off: every fused multiply-add in this file is an explicit fma_().
#include <cstdio>
#include <mutex>

#include <hip/hip_ext.h>
#include "mirt_kernels.h"
#include "mirt_device_math.h"

namespace mirt {
namespace MIRT_KNS {

// ------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------

struct SceneLds {
    const float*            cam;      // 24 floats, MirtGpuCamera layout
    const PreparedSphere*   spheres;
    const MirtMaterial*     mats;     // parity kernel
    const PreparedMaterial* pmats;    // path-traced kernels
    const float*            sky;      // 36 floats (MirtSkyState) or nullptr
    unsigned char*          end;      // first byte after the staged scene (16-B aligned)
};

// Cooperative global -> LDS copy of the scene tables, 16 B per thread per step.
// PT = false stages the reference-layout material table, PT = true the prepared one.
// GRID builds (MATS_IN_LDS = false) stage neither the materials nor the spheres: what their sphere tests read lives in
// the grid blob (stage_grid), and a hit reads its sphere and material from global memory / L2 once.
template <bool PT, bool MATS_IN_LDS = true>
MIRT_DEV SceneLds stage_scene(const RenderArgs& A, unsigned char* smem, bool hosek)
{
    constexpr uint32_t kMatQuads = PT ? sizeof(PreparedMaterial) / 16 : sizeof(MirtMaterial) / 16;
    const uint32_t n_cam = sizeof(MirtGpuCamera) / 16;
    const uint32_t n_sph = MATS_IN_LDS ? A.n_spheres * 2 : 0u;
    const uint32_t n_mat = MATS_IN_LDS ? A.n_mats * kMatQuads : 0u;
    const uint32_t n_sky = hosek ? sizeof(MirtSkyState) / 16 : 0;
    uint4* dst = reinterpret_cast<uint4*>(smem);
    // the camera is the first kernel argument, by value: read it from the kernarg segment (constant address
    // space, vector loads) so that thread i copies quad i without a private copy of the struct
    static_assert(offsetof(RenderArgs, cam) == 0, "camera must be the first kernel argument");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(4))) const u32x4 kernarg_quad;
    kernarg_quad* src_cam = (kernarg_quad*)__builtin_amdgcn_kernarg_segment_ptr();
    const uint4* src_sph = reinterpret_cast<const uint4*>(A.spheres);
    const uint4* src_mat = PT ? reinterpret_cast<const uint4*>(A.pmats) : reinterpret_cast<const uint4*>(A.mats);
    const uint4* src_sky = reinterpret_cast<const uint4*>(A.sky);
    const uint32_t total = n_cam + n_sph + n_mat + n_sky;
#ifdef MIRT_PROBE_NOSTAGE      // experiment builds only: what does staging cost a short launch?  (the frame is garbage)
    if (A.width == 0xffffffffu)
#endif
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        uint4 v;
        if (i < n_cam) { const u32x4 q = src_cam[i]; v = make_uint4(q.x, q.y, q.z, q.w); }
        else if (i < n_cam + n_sph) v = src_sph[i - n_cam];
        else if (i < n_cam + n_sph + n_mat) v = src_mat[i - n_cam - n_sph];
        else v = src_sky[i - n_cam - n_sph - n_mat];
        dst[i] = v;
    }
    __syncthreads();
    SceneLds S;
    S.cam = reinterpret_cast<const float*>(smem);
    S.spheres = MATS_IN_LDS ? reinterpret_cast<const PreparedSphere*>(smem + 16 * n_cam) : A.spheres;
    S.mats = reinterpret_cast<const MirtMaterial*>(smem + 16 * (n_cam + n_sph));
    S.pmats = MATS_IN_LDS ? reinterpret_cast<const PreparedMaterial*>(smem + 16 * (n_cam + n_sph)) : A.pmats;
    S.sky = hosek ? reinterpret_cast<const float*>(smem + 16 * (n_cam + n_sph + n_mat)) : nullptr;
    S.end = smem + 16 * total;
    return S;
}

// The kernel's arguments re-read from the kernarg segment through a pointer the compiler cannot see through: values
// that are needed once per strip (the strip levels, the output and accumulation pointers, the tone-curve flags) are
// then loaded where they are used instead of sitting in SGPRs across the whole hot loop, which needs all 102 of them.
MIRT_DEV const RenderArgs& per_strip_args()
{
    typedef __attribute__((address_space(4))) const RenderArgs kernarg_args;
    kernarg_args* p = (kernarg_args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const RenderArgs*)p;
}

MIRT_DEV size_t scene_lds_bytes_dev(uint32_t n_spheres, uint32_t n_mats, bool hosek, bool mats_in_lds = true)
{
    return sizeof(MirtGpuCamera) + (mats_in_lds ? (size_t)n_spheres * sizeof(PreparedSphere) + (size_t)n_mats * sizeof(PreparedMaterial) : 0) +
           (hosek ? sizeof(MirtSkyState) : 0);
}

// compact output row -> absolute image row (MirtParams contract, include/mirt.h)
MIRT_DEV uint32_t abs_row(const RenderArgs& A, uint32_t i)
{
    if (A.tile_rows == 0 || A.n_parts <= 1) return A.row_begin + i;
    const uint32_t t = A.part + (i / A.tile_rows) * A.n_parts;
    return A.row_begin + t * A.tile_rows + i % A.tile_rows;
}

// The first unit of every wave is its own index in the grid (the dispenser continues at RenderArgs.first_dispensed = the
// number of waves launched): otherwise all waves of a launch queue up on one atomic address at 14 ns each
// before any of them has work -- 86 us for 6 144 waves.
MIRT_DEV uint32_t first_unit() { return __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); }

MIRT_DEV uint32_t next_unit(const RenderArgs& A, uint32_t lane)
{
    uint32_t s = 0;
    if (lane == 0) s = atomicAdd(A.work_counter, 1u);
    return __builtin_amdgcn_readfirstlane(s) + A.first_dispensed;
}

// The strip kernels' launches of many small units -- 64 800 units of 32 pixels in a 1 ms launch (config 2) -- run into the rate of ONE
// dispenser word: a returning device-scope atomic on one address saturates at ~88 per microsecond (MI355X_MICROARCH.md, "dequeue").  Such
// launches (A.spread_units) use EIGHT words, 4 KB apart so that they sit in different memory channels (64 bytes apart they gained nothing);
// word x hands out the units u = 8 k + x.  A wave asks the word of its block's XCD first (blockIdx mod 8: workgroups are dealt to the XCDs
// round-robin, so the eight words see equal traffic; nothing depends on that for correctness) and the others in turn once it has run
// dry: every unit is taken exactly once, and a wave gets "no unit" (>= n_units) only after all eight words have run dry -- the loop that
// calls this ends for every wave.
MIRT_DEV uint32_t next_unit_any(const RenderArgs& A_unused, uint32_t lane)
{
    (void)A_unused;
    const RenderArgs& A = per_strip_args();        // once per unit: read from the kernarg segment here, not held in scalar registers across the hot loop
    if (A.spread_units == 0u) return next_unit(A, lane);
    const uint32_t home = blockIdx.x & 7u;
#pragma unroll 1
    for (uint32_t j = 0; j < 8u; ++j) {                    // wave-uniform; kept a loop (unrolled it costs the strip kernels ~45 scalar registers)
        const uint32_t x = (home + j) & 7u;
        const uint32_t share = A.n_units > x ? (A.n_units - x + 7u) >> 3 : 0u;      // units of word x: u = 8 k + x < n_units
        const uint32_t taken = A.disp_taken[x];                 // of word x's units, those taken by wave index
        uint32_t k = 0u;
        if (lane == 0) k = atomicAdd(A.work_counter + (size_t)MIRT_DISPENSER_STRIDE * x, 1u);
        k = __builtin_amdgcn_readfirstlane(k) + taken;
        if (k < share) return 8u * k + x;
    }
    return 0xffffffffu;
}

MIRT_DEV uint32_t sat_u32(float f)   // Rust `as u32` / WGSL u32(): truncate, saturate, NaN -> 0
{
    const float c = (f > 0.0f) ? f : 0.0f;                  // NaN and negatives -> 0
    const uint32_t v = (uint32_t)((c >= 4294967296.0f) ? 0.0f : c);
    return (c >= 4294967296.0f) ? 0xffffffffu : v;
}

MIRT_DEV uint32_t sat_u8(float f)    // Rust `as u8` (math.rs:15-17)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 255.0f) return 255u;
    return (uint32_t)f;
}

// wave ballot of a predicate.  HIP's __ballot(int) compares a materialised 0/1 with zero (v_cndmask + v_cmp per call when the
// predicate is a loop-carried lane mask); the builtin takes the lane mask itself.
MIRT_DEV unsigned long long ballot_(bool p) { return __builtin_amdgcn_ballot_w64(p); }

MIRT_DEV float clamp01(float x) { return (x < 0.0f) ? 0.0f : ((x > 1.0f) ? 1.0f : x); }

MIRT_DEV f3 texel_at(const RenderArgs& A_unused, uint64_t g)
{
    // the table's address and size are re-read from the kernel arguments here (see per_strip_args): image-texture
    // lookups are rare in the path-traced kernels' hot loop, and four SGPRs held for them across it are not
    (void)A_unused;
    const RenderArgs& A = per_strip_args();
    if (g >= A.n_texels) g = A.n_texels - 1;
    const float* e = A.texels + 3 * g;
    return mk(e[0], e[1], e[2]);
}

// LDS texel tile of the pool kernel's tile build (MIRT_FLAG_TEXEL_TILES; BASELINE configs[3] "LDS texel tiles"): a
// kTileW x kTileH window of ONE image texture per wave and strip, in three colour planes, centred on the texel of the
// strip's first image fetch -- the camera rays of a strip hit neighbouring texels (a 16-pixel strip of config 4 covers
// about 9 x 2 texels at the centre of the earth sphere, more towards the limb), later bounces do not.  Lanes whose texel
// lies in the window read LDS, the others the global table: the VALUES are the same, so the image cannot change.
// Geometry, measured on config 4 (tools/texel_tiles.py, profiles/r02_texel_tile_build.json): the window has to fit
// beside the pool in the 6.6 KB a wave has at 6 blocks per CU, and slots are worth more than texels --
//   88 slots + 32 x 4 texels: camera-ray hits 97.6 % from the tile, all fetches 77 %, kernel time +11.4 % vs the default build
//  104 slots + 16 x 4:        92.3 % / 73 %, +5.1 %
//  112 slots + 16 x 2:        83.9 % / 66 %, +3.1 %   <- built (the default build's pool, window in the spare LDS)
#ifndef MIRT_TILE_W
#define MIRT_TILE_W 16
#endif
#ifndef MIRT_TILE_H
#define MIRT_TILE_H 2
#endif
constexpr uint32_t kTileW = MIRT_TILE_W, kTileH = MIRT_TILE_H, kTileTexels = kTileW * kTileH;
constexpr uint32_t kTileBytes = 16 + kTileTexels * 12;      // header {j0, i0, table offset, texture width (0 = empty)} + planes
struct TexelTile {
    uint32_t* hdr;          // LDS, wave-private: where the lanes that place the window publish it
    float*    plane;        // LDS [3][kTileTexels]
    uint32_t  j0, i0, off, w;          // the window, wave-uniform registers (w == 0: none yet); see refresh()
    uint32_t  fetched, from_tile;      // per lane, counting builds: image-texel fetches of the current routine / served by the tile
    MIRT_DEV void reset(uint32_t lane) { if (lane == 0) *reinterpret_cast<uint4*>(hdr) = make_uint4(0u, 0u, 0u, 0u); j0 = i0 = off = w = 0u; }
    // The window is placed inside a divergent region (by the lanes that fetch), so it reaches the wave's scalar registers
    // through LDS: the pool kernel calls this in uniform control flow after every scatter routine until a window exists.
    MIRT_DEV void refresh()
    {
        if (w != 0u) return;
        const uint4 h = *reinterpret_cast<const uint4*>(hdr);
        j0 = __builtin_amdgcn_readfirstlane(h.x); i0 = __builtin_amdgcn_readfirstlane(h.y);
        off = __builtin_amdgcn_readfirstlane(h.z); w = __builtin_amdgcn_readfirstlane(h.w);
    }
};

// texture_lookup (mod.rs:1000-1019 == wgsl:377-387); final index clamped to the table.
MIRT_DEV f3 texture_lookup(const RenderArgs& A, uint32_t width, uint32_t height, uint32_t offset, float u, float v, TexelTile* T = nullptr)
{
    const float uc = clamp01(u);
    const float vf = 1.0f - clamp01(v);
    const uint32_t j = sat_u32(uc * (float)width);
    const uint32_t i = sat_u32(vf * (float)height);
    const uint32_t idx = i * width + j;
#define MIRT_PROBE_SITE 0           // experiment builds only: expands to nothing in libmirt.so
#include "mirt_pool_probes.inc"
#undef MIRT_PROBE_SITE
    if (T == nullptr) return texel_at(A, (uint64_t)offset + (uint64_t)idx);
    // ---- tile build: runs in the lanes of a wave that need an image texel (a divergent region) ----
    uint4 hd = make_uint4(T->j0, T->i0, T->off, T->w);                          // scalar registers
    if (T->w == 0u) {
        // first image fetch of this strip: centre the window on the first lane's texel and load it with the lanes at hand
        const uint32_t wf = __builtin_amdgcn_readfirstlane(width), hf = __builtin_amdgcn_readfirstlane(height);
        const uint32_t of = __builtin_amdgcn_readfirstlane(offset);
        if (wf >= kTileW && hf >= kTileH) {
            const uint32_t jf = __builtin_amdgcn_readfirstlane(j), i_f = __builtin_amdgcn_readfirstlane(i);
            uint32_t j0 = jf > kTileW / 2 ? jf - kTileW / 2 : 0u, i0 = i_f > kTileH / 2 ? i_f - kTileH / 2 : 0u;
            j0 = j0 > wf - kTileW ? wf - kTileW : j0;
            i0 = i0 > hf - kTileH ? hf - kTileH : i0;
            const unsigned long long act = ballot_(true);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
            const uint32_t n_act = (uint32_t)__popcll(act);
            for (uint32_t t = rank; t < kTileTexels; t += n_act) {
                const f3 e = texel_at(A, (uint64_t)of + (uint64_t)((i0 + t / kTileW) * wf + j0 + t % kTileW));
                T->plane[t] = e.x; T->plane[kTileTexels + t] = e.y; T->plane[2 * kTileTexels + t] = e.z;
            }
            hd = make_uint4(j0, i0, of, wf);
            if (rank == 0u) *reinterpret_cast<uint4*>(T->hdr) = hd;
            __builtin_amdgcn_wave_barrier();       // LDS serves a wave's requests in order: the reads below see the planes
        }
    }
    const uint32_t dj = j - hd.x, di = i - hd.y;
    const bool in = (hd.w != 0u) & (hd.w == width) & (hd.z == offset) & (dj < kTileW) & (di < kTileH);
    T->fetched += 1u;
    T->from_tile += in ? 1u : 0u;
    f3 c = mk(0, 0, 0);
    if (ballot_(in) != 0ull) {
        if (in) { const uint32_t t = di * kTileW + dj; c = mk(T->plane[t], T->plane[kTileTexels + t], T->plane[2 * kTileTexels + t]); }
    }
    if (ballot_(!in) != 0ull) {
        if (!in) c = texel_at(A, (uint64_t)offset + (uint64_t)idx);
    }
    return c;
}

MIRT_DEV unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

MIRT_DEV uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16) | 0xff000000u; }

// per-lane work counters of the counting builds (MIRT_FLAG_COUNT_WORK); compiled away otherwise
template <bool COUNT>
struct Work {
    uint32_t c[kNumCounters];
    MIRT_DEV void clear() { if constexpr (COUNT) { for (uint32_t i = 0; i < kNumCounters; ++i) c[i] = 0; } }
    MIRT_DEV void add(uint32_t slot, uint32_t n = 1) { if constexpr (COUNT) c[slot] += n; }
    MIRT_DEV void flush(unsigned long long* g, uint32_t lane)
    {
        if constexpr (COUNT) {
            for (uint32_t i = 0; i < kNumCounters; ++i) {
                const unsigned long long t = wave_sum_u64(c[i]);
                if (lane == 0 && t) atomicAdd(&g[i], t);
                c[i] = 0;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// render_parity — layer.rs semantics
// ------------------------------------------------------------------------------------------

struct PRay { f3 o, d; };
struct PHit { f3 p, n; };

// `Layer::ray_hit_world_raw` (layer.rs:413-444) over `Sphere::closest_hit_raw` (mod.rs:1121-1157)
// and `update_ray_hit_info` / `set_face_normal` (mod.rs:1217-1243, 1095-1110).
// Returns the LAST sphere in list order with a root in [tmin, tmax]: `closest_hit` is reset to
// `old_hit` = rec.t, which nobody ever writes (it stays f32::MAX).
#ifndef MIRT_FAST_MATH       // parity mode exists in the exact build only
struct PCount { uint32_t tests, roots, hits; };     // what ONE ray_hit_world_raw call costs the reference

template <bool COUNT>
MIRT_DEV bool parity_world_hit(const SceneLds& S, uint32_t n_spheres, const PRay& ray, float tmin, float tmax,
                               const float rec_t, PHit& rec, PCount& pc)
{
    if constexpr (COUNT) { pc.tests = n_spheres; pc.roots = 0; pc.hits = 0; }
    bool hit_anything = false;
    float closest_hit = tmax;
    const float old_hit = rec_t;
    const float a = dot_nofma(ray.d, ray.d);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const PreparedSphere sp = S.spheres[i];
        const f3 c = mk(sp.cx, sp.cy, sp.cz);
        const f3 oc = ray.o - c;
        const float half_b = dot_nofma(oc, ray.d);
        const float cc = dot_nofma(oc, oc) - sp.rr;
        const float disc = half_b * half_b - a * cc;
        if (disc < 0.0f) continue;                       // `< 0.0` rejects: disc == 0 and NaN go on
        const float sq = sqrt_(disc);
        float t = (-half_b - sq) / a;
        if constexpr (COUNT) pc.roots += 1;
        if (t < tmin || closest_hit < t) {
            t = (-half_b + sq) / a;
            if constexpr (COUNT) pc.roots += 1;
            if (t < tmin || closest_hit < t) continue;
        }
        if constexpr (COUNT) pc.hits += 1;
        // update_ray_hit_info: `if t < 0.0 { return false }` cannot trigger (t >= tmin > 0)
        const f3 p = ray.o + mk(ray.d.x * t, ray.d.y * t, ray.d.z * t);
        const f3 n = sp.inv_r * (p - c);
        const bool front = dot_nofma(ray.d, n) < 0.0f;
        rec.p = p;
        rec.n = front ? n : -n;
        hit_anything = true;
        closest_hit = old_hit;
    }
    return hit_anything;
}

// COUNT = true additionally tallies the work the REFERENCE's sequential sample loop performs (it returns at
// the first terminating sample): only the samples up to and including that one are counted, although all 64
// lanes of a batch compute theirs.  The CPU check in tests/test_gpu_parity.py tallies the same quantities.
// BY_PIXEL = false: a wave owns a strip of 16 pixels and takes them one at a time, lane = sample (64 samples of a pixel per batch).
// BY_PIXEL = true : a wave owns 64 consecutive pixels, lane = pixel, and every lane runs the reference's sample loop for its pixel
//                   in order -- the launch shape for the reference's operating point, 2 samples per pixel (mod.rs:605-613), where
//                   lane = sample would leave 62 of 64 lanes idle.  The host takes it below kByPixelMaxSpp samples per pixel.
template <bool COUNT, bool BY_PIXEL = false>
__global__ __launch_bounds__(kBlockThreads) void render_parity_kernel(RenderArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const SceneLds S = stage_scene<false>(A, smem, false);
    const uint32_t lane = threadIdx.x & 63u;
    const float wf = (float)A.width, hf = (float)A.height;
    const uint32_t npix = A.out_rows * A.width;
    const float FMAX = 3.40282347e+38f;

    const f3 eye = mk(S.cam[0], S.cam[1], S.cam[2]);
    const f3 hor = mk(S.cam[4], S.cam[5], S.cam[6]);
    const f3 ver = mk(S.cam[8], S.cam[9], S.cam[10]);
    const f3 llc = mk(S.cam[20], S.cam[21], S.cam[22]);
    Work<COUNT> work;
    work.clear();

    const uint32_t grid_waves = gridDim.x * (kBlockThreads / 64u);
    for (uint32_t strip = first_unit(); strip < A.n_units; strip = (BY_PIXEL && A.static_units) ? strip + grid_waves : next_unit_any(A, lane)) {
        if constexpr (BY_PIXEL) {
            const uint32_t pi = strip * 64u + lane;
            const bool inside = pi < npix;
            const uint32_t ci = (inside ? pi : 0u) / A.width;
            const uint32_t x = (inside ? pi : 0u) - ci * A.width;
            const uint32_t y = abs_row(A, ci);
            const float u = (float)x / wf;               // coord_to_color math.rs:4-9
            const float v = (float)y / hf;
            uint32_t hits_before = 0;                    // primary hits of this pixel's earlier samples: `depth` = 20 - hits_before
            bool done = !inside;
            uint32_t rgba = pack_rgba(sat_u8(v * 255.0f), sat_u8(u * 255.0f), sat_u8(255.0f));   // loop exhausted: layer.rs:380
            for (uint32_t s = 0; s < A.spp && ballot_(!done) != 0ull; ++s) {      // the sample loop layer.rs:320-378, every lane its own
                const bool active = !done;
                const float uu = u + 0.0f, vv = v + 0.0f;                          // jitter == +0.0f (see the lane = sample loop below)
                PRay ray;
                ray.o = eye;
                ray.d = ((llc + uu * hor) + vv * ver) - eye;                       // GpuCamera::make_ray mod.rs:745-754
                PHit rec;
                rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0);
                bool prim = false, scat_ok = false, sec = false;
                f3 colour = mk(0, 0, 0);
                PCount pc1{}, pc2{};
                if (active) prim = parity_world_hit<COUNT>(S, A.n_spheres, ray, 0.001f, FMAX, FMAX, rec, pc1);
                const bool exhausted = prim && (hits_before >= 20u);                // `if depth <= 0 { return black }` layer.rs:333-335
                if (prim) {
                    const MirtMaterial m2 = S.mats[2];                              // index 2, screen-space (uu, vv): layer.rs:345-351
                    const float fuzzy = m2.x;
                    const f3 albedo = texture_lookup(A, m2.desc1.width, m2.desc1.height, m2.desc1.offset, uu, vv);
                    const f3 unit = mk(ray.d.x / 3.0f, ray.d.y / 3.0f, ray.d.z / 3.0f);     // unit_vertor math.rs:147-149
                    const float k = 2.0f * dot_nofma(unit, rec.n);                  // reflect math.rs:154-159
                    PRay sc;
                    sc.o = rec.p;
                    sc.d = unit - k * rec.n;
                    scat_ok = dot_nofma(sc.d, rec.n) > 0.0f;                        // scatter_metal mod.rs:1292-1315
                    if (scat_ok && !exhausted) {
                        sec = parity_world_hit<COUNT>(S, A.n_spheres, sc, 0.001f, FMAX, FMAX, rec, pc2);
                        if (sec) {
                            const float norm = sqrt_(dot_nofma(rec.n, rec.n));      // layer.rs:364-372
                            f3 c = mk(rec.n.x / norm, rec.n.y / norm, rec.n.z / norm);
                            c = mk((c.x * 255.0f) / 2.0f, (c.y * 255.0f) / 2.0f, (c.z * 255.0f) / 2.0f);
                            c = mk(c.x * (albedo.x * fuzzy), c.y * (albedo.y * fuzzy), c.z * (albedo.z * fuzzy));
                            colour = mk(0.0f + c.x, 0.0f + c.y, 0.0f + c.z);
                        }
                    }
                }
                const bool term = prim && (exhausted || !scat_ok || sec);          // this sample returns from the pixel
                if (term) {
                    rgba = (exhausted || !scat_ok) ? pack_rgba(0, 0, 0) : pack_rgba(sat_u8(colour.x), sat_u8(colour.y), sat_u8(colour.z));
                    done = true;
                }
                hits_before += prim ? 1u : 0u;
                if constexpr (COUNT) {
                    if (active) {
                        const bool second = prim && !exhausted && scat_ok;
                        work.add(kCntLaneIters);
                        work.add(kCntRays, second ? 2u : 1u);
                        work.add(kCntTests, pc1.tests + (second ? pc2.tests : 0u));
                        work.add(kCntRoots, pc1.roots + (second ? pc2.roots : 0u));
                        work.add(kCntHits, pc1.hits + (second ? pc2.hits : 0u));
                        if (prim && !exhausted) work.add(kCntScatter1);
                    }
                }
            }
            if (inside) A.out[pi] = rgba;
            work.flush(A.counters, lane);
            continue;
        }
        const uint32_t base = strip * kStripPixels;
        uint32_t my_px = 0;
        for (uint32_t p = 0; p < kStripPixels; ++p) {
            const uint32_t pi = base + p;
            if (pi >= npix) break;
            const uint32_t ci = pi / A.width;
            const uint32_t x = pi - ci * A.width;
            const uint32_t y = abs_row(A, ci);
            const float u = (float)x / wf;               // coord_to_color math.rs:4-9
            const float v = (float)y / hf;

            uint32_t hits_before = 0;                    // primary hits of earlier samples (depth bookkeeping)
            bool done = false;
            uint32_t rgba = 0;
            for (uint32_t s0 = 0; s0 < A.spp && !done; s0 += 64) {
                const bool active = (s0 + lane) < A.spp;
                // jitter: random_f32() <= 2.94e-39 (math.rs:107-110) is defined as +0.0f
                const float uu = u + 0.0f, vv = v + 0.0f;
                // GpuCamera::make_ray mod.rs:745-754
                PRay ray;
                ray.o = eye;
                ray.d = ((llc + uu * hor) + vv * ver) - eye;
                PHit rec;
                rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0);
                bool prim = false, scat_ok = false, sec = false;
                f3 colour = mk(0, 0, 0);
                PCount pc1{}, pc2{};
                if (active) prim = parity_world_hit<COUNT>(S, A.n_spheres, ray, 0.001f, FMAX, FMAX, rec, pc1);
                if (prim) {
                    // material hard-wired to index 2; texture looked up with SCREEN-space (uu,vv) (layer.rs:345-351)
                    const MirtMaterial m2 = S.mats[2];
                    const float fuzzy = m2.x;
                    const f3 albedo = texture_lookup(A, m2.desc1.width, m2.desc1.height, m2.desc1.offset, uu, vv);
                    // scatter_metal mod.rs:1292-1315; unit_vertor divides by 3 (math.rs:147-149)
                    const f3 unit = mk(ray.d.x / 3.0f, ray.d.y / 3.0f, ray.d.z / 3.0f);
                    const float k = 2.0f * dot_nofma(unit, rec.n);      // reflect math.rs:154-159
                    PRay sc;
                    sc.o = rec.p;
                    sc.d = unit - k * rec.n;
                    scat_ok = dot_nofma(sc.d, rec.n) > 0.0f;
                    if (scat_ok) {
                        sec = parity_world_hit<COUNT>(S, A.n_spheres, sc, 0.001f, FMAX, FMAX, rec, pc2);
                        if (sec) {
                            // (n.normalize() * 255.0 / 2.0) * (albedo * fuzzy)  layer.rs:364-372
                            const float norm = sqrt_(dot_nofma(rec.n, rec.n));
                            f3 c = mk(rec.n.x / norm, rec.n.y / norm, rec.n.z / norm);
                            c = mk((c.x * 255.0f) / 2.0f, (c.y * 255.0f) / 2.0f, (c.z * 255.0f) / 2.0f);
                            c = mk(c.x * (albedo.x * fuzzy), c.y * (albedo.y * fuzzy), c.z * (albedo.z * fuzzy));
                            colour = mk(0.0f + c.x, 0.0f + c.y, 0.0f + c.z);   // pixel_color += sampled_color
                        }
                    }
                }
                // sequential semantics of the sample loop, resolved with ballots:
                // sample s terminates the pixel if it has a primary hit and (depth exhausted |
                // scatter rejected | secondary hit); the first such sample in order wins.
                const unsigned long long prim_mask = ballot_(prim);
                const uint32_t k_before = hits_before + __popcll(prim_mask & ((1ull << lane) - 1ull));
                const bool exhausted = prim && (k_before >= 20u);
                const bool term = prim && (exhausted || !scat_ok || sec);
                const unsigned long long term_mask = ballot_(term);
                if (term_mask) {
                    const int first = __builtin_ctzll(term_mask);
                    const bool black = exhausted || !scat_ok;
                    const uint32_t mine = black ? pack_rgba(0, 0, 0)
                                                : pack_rgba(sat_u8(colour.x), sat_u8(colour.y), sat_u8(colour.z));
                    rgba = __shfl(mine, first, 64);
                    done = true;
                }
                hits_before += (uint32_t)__popcll(prim_mask);
                if constexpr (COUNT) {
                    // the sequential loop reaches sample s only if no earlier sample terminated the pixel
                    const bool reached = active && (term_mask == 0ull || lane <= (uint32_t)__builtin_ctzll(term_mask));
                    if (reached) {
                        const bool second = prim && !exhausted && scat_ok;      // the scattered ray is traced
                        work.add(kCntLaneIters);                               // samples executed
                        work.add(kCntRays, second ? 2u : 1u);
                        work.add(kCntTests, pc1.tests + (second ? pc2.tests : 0u));
                        work.add(kCntRoots, pc1.roots + (second ? pc2.roots : 0u));
                        work.add(kCntHits, pc1.hits + (second ? pc2.hits : 0u));
                        if (prim && !exhausted) work.add(kCntScatter1);       // scatter_metal (layer.rs:353)
                    }
                }
            }
            if (!done) rgba = pack_rgba(sat_u8(v * 255.0f), sat_u8(u * 255.0f), sat_u8(255.0f));   // layer.rs:380
            if (lane == p) my_px = rgba;
        }
        if (lane < kStripPixels && base + lane < npix) A.out[base + lane] = my_px;
        work.flush(A.counters, lane);
    }
}

#endif  // !MIRT_FAST_MATH

// ------------------------------------------------------------------------------------------
// path-traced mode: building blocks shared by the strip and the pool kernel
// ------------------------------------------------------------------------------------------

MIRT_DEV uint32_t jenkins_hash(uint32_t x)
{
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}

struct Rng {
    uint32_t state;
    // rngNextInt + rngNextFloat (wgsl:493-511): `+ 747796405 + 2891336453` is an add, not a multiply-add
    MIRT_DEV float next()
    {
        const uint32_t old = state + 747796405u + 2891336453u;
        const uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
        state = (word >> 22) ^ word;
        return (float)state * 0x1p-32f;
    }
    // c * next() for a constant c in one multiplication: next() = f * 2^-32 is exact (a power-of-two scaling of the
    // converted integer), so c * (f * 2^-32) and (c * 2^-32) * f round the same real number, c * f * 2^-32, once
    // each -- identical bits as long as nothing underflows (|c| >= 2^-60 is ample).  The caller passes c * 2^-32.
    MIRT_DEV float next_scaled(float c_times_2m32)
    {
        const uint32_t old = state + 747796405u + 2891336453u;
        const uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
        state = (word >> 22) ^ word;
        return (float)state * c_times_2m32;
    }
    // a draw whose value is not used: the state advances, nothing is converted
    MIRT_DEV void skip()
    {
        const uint32_t old = state + 747796405u + 2891336453u;
        const uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
        state = (word >> 22) ^ word;
    }
};

// camera constants hoisted into registers once per thread
struct CamRegs {
    f3 eye, hor, ver, llc;
    f3 cam_u, cam_v;           // lens basis and radius: loaded only for cameras with a lens
    float lens_radius;
    bool pinhole;              // wave-uniform: the lens offset is exactly zero for every draw (see generate_primary)
    float inv_w, inv_h;
};

MIRT_DEV CamRegs load_camera(const SceneLds& S, const RenderArgs& A)
{
    CamRegs c;
    // (reading the camera with scalar loads from the kernel arguments instead -- it is RenderArgs' first member -- was measured:
    //  +5.4 % on config 3, the scan's three sphere records are the only data for which that pays)
    c.eye = mk(S.cam[0], S.cam[1], S.cam[2]);
    c.hor = mk(S.cam[4], S.cam[5], S.cam[6]);
    c.ver = mk(S.cam[8], S.cam[9], S.cam[10]);
    c.llc = mk(S.cam[20], S.cam[21], S.cam[22]);
    // RenderArgs.cam._padding5 is written by the host for every launch (launch_render): 1 = pinhole.  All lanes read the
    // same LDS word; readfirstlane tells the compiler so.  (All six reads are issued together, lens or not: one LDS round trip.)
    c.cam_u = mk(S.cam[12], S.cam[13], S.cam[14]);
    c.cam_v = mk(S.cam[16], S.cam[17], S.cam[18]);
    c.lens_radius = S.cam[19];
    c.pinhole = __builtin_amdgcn_readfirstlane(bits(S.cam[23])) != 0u;
    c.inv_w = 1.0f / (float)A.width;
    c.inv_h = 1.0f / (float)A.height;
    return c;
}

// The pooled kernel's flat builds are held to 80 VGPRs (6 waves per SIMD) and use 70: nine of the idle ones keep hor / ver / llc for the whole
// kernel, so a GEN step re-reads three rows of the camera from LDS instead of six (config 3: -0.4 %, 79 VGPRs; round 4).  Builds that would
// spill for it (Hosek sky, texel tiles, counting) keep reading all six.
struct CamHeld { f3 hor, ver, llc; };
MIRT_DEV CamHeld hold_camera(const SceneLds& S) { return CamHeld{ mk(S.cam[4], S.cam[5], S.cam[6]), mk(S.cam[8], S.cam[9], S.cam[10]), mk(S.cam[20], S.cam[21], S.cam[22]) }; }
MIRT_DEV CamRegs load_camera(const SceneLds& S, const RenderArgs& A, const CamHeld& H)
{
    CamRegs c;
    c.eye = mk(S.cam[0], S.cam[1], S.cam[2]);
    c.hor = H.hor; c.ver = H.ver; c.llc = H.llc;
    c.cam_u = mk(S.cam[12], S.cam[13], S.cam[14]);
    c.cam_v = mk(S.cam[16], S.cam[17], S.cam[18]);
    c.lens_radius = S.cam[19];
    c.pinhole = __builtin_amdgcn_readfirstlane(bits(S.cam[23])) != 0u;
    c.inv_w = 1.0f / (float)A.width;
    c.inv_h = 1.0f / (float)A.height;
    return c;
}

// initRng (wgsl:498-502, frame = sample + 1) + samplePixel (wgsl:114-117) + cameraMakeRay (wgsl:456-478)
// RESEED = false continues the caller's stream: the reference's samplePixel loop draws all samples of one frame from
// the stream initRng seeded once for that frame (MirtParams.frame_spp > 0; lane-per-pixel schedule only).
// PINHOLE cameras (aperture 0: BASELINE configs 2 and 4, every camera of the reference's CPU path) skip the lens, EXACTLY:
// with lens_radius == +-0 both offsets lpx, lpy are +-0 (lr in [0, 1], sin / cos finite), so for a finite lens basis
// fma(lpy, v_k, lpx * u_k) = +-0, and eye_k + (+-0) == eye_k bit for bit unless eye_k is -0 (then the sum's sign would
// depend on the draw).  The host sets the flag only when lens_radius == 0, eye / u / v are finite and no eye component is
// -0 (mirt_api.hip: camera_is_pinhole); the two lens draws are still consumed, so the stream is the reference's.
template <bool RESEED = true>
MIRT_DEV void generate_primary(const RenderArgs& A, const CamRegs& C, uint32_t x, uint32_t y, uint32_t sample,
                               Rng& rng, f3& ro, f3& rd)
{
    if constexpr (RESEED) {
        const uint32_t pixel_index = x + y * A.width;
        rng.state = jenkins_hash((pixel_index ^ jenkins_hash(sample + 1u)) ^ A.seed_mix);
    }
    const float u = ((float)x + rng.next()) * C.inv_w;
    const float v = 1.0f - ((float)y + rng.next()) * C.inv_h;
    if (C.pinhole) {
        asm volatile("; generate_primary: pinhole camera" ::);       // keeps this a real (scalar) branch
        rng.skip();
        rng.skip();
        ro = C.eye;
    } else {
        const float lr = sqrt_unit(rng.next());
        const SinCos la = sincos_small(rng.next_scaled(kTwoPi * 0x1p-32f));
        const float lpx = C.lens_radius * (lr * la.c);
        const float lpy = C.lens_radius * (lr * la.s);
        ro = C.eye + fma3(lpy, C.cam_v, lpx * C.cam_u);
    }
    rd = fma3(v, C.ver, fma3(u, C.hor, C.llc)) - ro;
}

// nearest hit (wgsl:135-145, 407-429): every lane walks the same sphere list in LDS.
// One sphere: straight-line code, no divergent branch — lanes without a real root are masked out of the
// two comparisons instead (their sqrt argument is negative or NaN and never used).
template <bool COUNT>
MIRT_DEV void hit_sphere(const float4 s4, uint32_t i, f3 ro, f3 rd, float a, float inv_a, bool alive, float& closest, int& best,
                         Work<COUNT>& work)
{
    const f3 oc = ro - mk(s4.x, s4.y, s4.z);
    const float b = dot(oc, rd);
    const float cq = dot(oc, oc) - s4.w;
    const float disc = fma_(b, b, -(a * cq));
    if (alive && disc > 0.0f) {                              // skipped when no lane of the wave can hit
        const float sq = sqrt_(disc);
        const float t0 = (-b - sq) * inv_a;
        const float t1 = (-b + sq) * inv_a;
        // first root, else second (wgsl:415-425): `t0 < closest && t0 > MIN_T ? t0 : (t1 < closest && t1 > MIN_T ? t1 : none)`.
        // t0 <= t1 always (sq >= 0, inv_a > 0), so that is the smaller of the roots above MIN_T, accepted if below
        // `closest`: two compare+select pairs, one min and one compare, and NO scalar mask arithmetic (the and/and/or of
        // four compare masks costs three SALU instructions per test, each dearer than a VALU one here).
        if constexpr (COUNT) work.add(kCntRoots, ((t0 < closest) && (t0 > kMinT)) ? 1u : 2u);
        const float kInf = __builtin_inff();
        const float c0 = (t0 > kMinT) ? t0 : kInf;
        const float c1 = (t1 > kMinT) ? t1 : kInf;
        const float t = __builtin_fminf(c0, c1);
        const bool ok = t < closest;
        closest = ok ? t : closest;
        best = ok ? (int)i : best;
    }
}

template <bool COUNT>
MIRT_DEV int nearest_hit(const SceneLds& S, uint32_t n_spheres, f3 ro, f3 rd, bool alive, float& closest_out,
                         Work<COUNT>& work)
{
    const float a = dot(rd, rd);
    const float inv_a = rcp_(a);
    float closest = kMaxT;
    int best = -1;
    if (alive) { work.add(kCntRays); work.add(kCntTests, n_spheres); }
    const float4* sph = reinterpret_cast<const float4*>(S.spheres);     // {centre, radius^2} is the first half of a PreparedSphere
    // Three spheres -- the headline scene (src/main.rs:539-541) -- are straight-line code: all reads issued together, no
    // loop bookkeeping.  Measured on config 3 (tools/ab_libs.py, interleaved rounds): -1.5 %; the same treatment for
    // lists of 1, 2 and 4 spheres behind one switch cost the three-sphere case 1.7 % and was dropped.
    if (n_spheres == 3u) {
        // the three records come from the kernel arguments (RenderArgs.sph3, filled by the host for such scenes): scalar
        // loads into SGPRs that live only for this scan -- no LDS round trip, no VGPRs for the records (-0.5 %)
        const RenderArgs& AK = per_strip_args();
        const float4 s0 = make_float4(AK.sph3[0], AK.sph3[1], AK.sph3[2], AK.sph3[3]);
        const float4 s1 = make_float4(AK.sph3[4], AK.sph3[5], AK.sph3[6], AK.sph3[7]);
        const float4 s2 = make_float4(AK.sph3[8], AK.sph3[9], AK.sph3[10], AK.sph3[11]);
        hit_sphere<COUNT>(s0, 0, ro, rd, a, inv_a, alive, closest, best, work);
        hit_sphere<COUNT>(s1, 1, ro, rd, a, inv_a, alive, closest, best, work);
        hit_sphere<COUNT>(s2, 2, ro, rd, a, inv_a, alive, closest, best, work);
        closest_out = closest;
        return best;
    }
    uint32_t i = 0;
    for (; i + 3 <= n_spheres; i += 3) {                                 // three LDS reads in flight per round trip
        const float4 s0 = sph[2 * i], s1 = sph[2 * i + 2], s2 = sph[2 * i + 4];
        hit_sphere<COUNT>(s0, i, ro, rd, a, inv_a, alive, closest, best, work);
        hit_sphere<COUNT>(s1, i + 1, ro, rd, a, inv_a, alive, closest, best, work);
        hit_sphere<COUNT>(s2, i + 2, ro, rd, a, inv_a, alive, closest, best, work);
    }
    for (; i < n_spheres; ++i) hit_sphere<COUNT>(sph[2 * i], i, ro, rd, a, inv_a, alive, closest, best, work);
    closest_out = closest;
    return best;
}


MIRT_DEV float max_(float a, float b) { return (b > a) ? b : a; }

// pool kernel, grid builds: cells a path may visit in the step that starts its walk / in an OP_WALK step that resumes it
#ifndef MIRT_WALK_FRESH
#define MIRT_WALK_FRESH 3
#endif
#ifndef MIRT_WALK_RESUME
#define MIRT_WALK_RESUME 3
#endif
constexpr uint32_t kWalkCellsFresh = MIRT_WALK_FRESH, kWalkCellsResume = MIRT_WALK_RESUME;

// In-kernel timing of the pooled kernel's sections (diagnosis builds only, -DMIRT_DIAG_STAMPS; tools/step_stats.py): s_memtime at
// the section boundaries, summed per wave and added to the launch's counters 18.. at the end of every strip.
#ifdef MIRT_DIAG_STAMPS
struct Stamps {
    unsigned long long acc[10];
    unsigned long long prev;
    MIRT_DEV void start()
    {
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0;
        prev = __builtin_amdgcn_s_memtime();
    }
    MIRT_DEV void mark(int i) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc[i] += t - prev; prev = t; }
    MIRT_DEV void flush(unsigned long long* g, uint32_t lane)
    {
#pragma unroll
        for (int i = 0; i < 10; ++i) { if (lane == 0) atomicAdd(&g[18 + i], acc[i]); acc[i] = 0; }
    }
};
#else
struct Stamps {
    MIRT_DEV void start() {}
    MIRT_DEV void mark(int) {}
    MIRT_DEV void flush(unsigned long long*, uint32_t) {}
};
#endif

// ---- nearest hit through the uniform grid (many-sphere scenes) ----
//
// Equivalence with the flat scan (nearest_hit above).  For one sphere let f = its first root above
// MIN_T (t0 if t0 > MIN_T, else t1 if t1 > MIN_T).  The scan replaces `closest` exactly when
// f < closest, so whatever the visiting order the result is min f over the spheres, and on equal f
// the flat scan keeps the LOWER index (strict `<`).  Here spheres are visited in cell order and
// possibly several times, so the update rule carries the tie-break explicitly.  A cell walk may
// stop as soon as closest <= the parameter at which the ray leaves the current cell: any sphere
// with a smaller f is intersected inside a cell already visited, and is listed there because every
// sphere is registered in all cells its (slightly enlarged) bounding box overlaps.
struct GridLds {
    const GridHeader*     h;
    const unsigned short* big;        // ids of the big spheres
    const uint2*          cells;      // per cell: {first FURTHER item | item count << 16, id0 | id1 << 16}: the first two ids arrive with the header
    const unsigned short* items;      // ids of the cells' items
    const unsigned char*  ops;        // routine queue of every sphere (PreparedSphere.op)
    const float4*         big_recs;   // {centre, r^2} of the big spheres, in list order
    const float4*         recs;       // ... of every sphere, by sphere id (the cells' items are ids)
};

// What every step of the pooled kernel's grid build reads from the blob's header, and the always-tested spheres' records when there are
// at most four of them (RTIOW: the ground and three heroes), held in REGISTERS for the whole kernel (round 4).  That build runs one
// 1024-thread block per CU -- four waves per SIMD, whatever it does -- so the 128 VGPRs a wave may use are there for the taking: the walk
// of round 4's block 9 used 96; the header's fields (11 registers: -1.2 %) and four records with their ids (-1.2 % more) fill the rest,
// and a fresh step starts its sphere tests without waiting for six broadcast LDS reads (profiles/r04_c5_ab.txt block 13).
struct GridConsts { f3 org, cell, inv_cell, hi; int dx, dy, dz; float inv_dim_x, inv_dim_xy;
                    float4 bigr[4]; uint32_t bigid[4]; uint32_t n_big; };
MIRT_DEV GridConsts load_grid_consts(const GridLds& G)
{
    const GridHeader& H = *G.h;
    GridConsts c;
    c.org = mk(H.org[0], H.org[1], H.org[2]);
    c.cell = mk(H.cell[0], H.cell[1], H.cell[2]);
    c.inv_cell = mk(H.inv_cell[0], H.inv_cell[1], H.inv_cell[2]);
    c.dx = (int)H.dims[0]; c.dy = (int)H.dims[1]; c.dz = (int)H.dims[2];
    c.hi = mk(fma_((float)c.dx, c.cell.x, c.org.x), fma_((float)c.dy, c.cell.y, c.org.y), fma_((float)c.dz, c.cell.z, c.org.z));
    c.inv_dim_x = H.inv_dim_x; c.inv_dim_xy = H.inv_dim_xy;
    c.n_big = __builtin_amdgcn_readfirstlane(H.n_big);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {                     // (entries past n_big: copies of entry 0 when there is one -- never tested)
        const bool on = j < c.n_big;
        c.bigr[j] = c.n_big ? G.big_recs[on ? j : 0u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        c.bigid[j] = c.n_big ? (uint32_t)G.big[on ? j : 0u] : 0u;
    }
    return c;
}

// one sphere test from its record s4 = {centre, r^2} (a copy of the first half of PreparedSphere i)
template <bool COUNT>
MIRT_DEV void test_sphere(const float4 s4, uint32_t i, f3 ro, f3 rd, float a, float inv_a, bool alive, float& closest, int& best,
                          Work<COUNT>& work)
{
    if (alive) work.add(kCntTests);                              // grid builds count the tests a lane really performs
    const f3 oc = ro - mk(s4.x, s4.y, s4.z);
    const float b = dot(oc, rd);
    const float cq = dot(oc, oc) - s4.w;
    const float disc = fma_(b, b, -(a * cq));
    if (alive && disc > 0.0f) {
        const float sq = sqrt_(disc);
        const float t0 = (-b - sq) * inv_a;
        const float t1 = (-b + sq) * inv_a;
        const bool first = t0 > kMinT;
        work.add(kCntRoots, first ? 1u : 2u);
        const float f = first ? t0 : t1;                          // first root above MIN_T
        // mask logic and two selects, no short-circuit: nested exec-mask regions around the tie-break cost more scalar
        // instructions than the two compares they skip (-1.1 % on RTIOW)
        const bool valid = first | (t1 > kMinT);
        const bool take = valid & ((f < closest) | ((f == closest) & ((int)i < best)));
        closest = take ? f : closest;
        best = take ? (int)i : best;
    }
}

// copy the grid (header + 16-bit CSR lists) behind the scene tables in LDS; all threads of the block take part
MIRT_DEV GridLds stage_grid(const RenderArgs& A, unsigned char* gdst)
{
    for (uint32_t i = threadIdx.x; i < A.grid_bytes / 16; i += blockDim.x)
        reinterpret_cast<uint4*>(gdst)[i] = reinterpret_cast<const uint4*>(A.grid)[i];
    __syncthreads();
    GridLds G;
    G.h = reinterpret_cast<const GridHeader*>(gdst);
    const unsigned short* base = reinterpret_cast<const unsigned short*>(gdst);
    G.big = base + G.h->off_big;
    G.cells = reinterpret_cast<const uint2*>(gdst + G.h->off_cells);
    G.items = base + G.h->off_items;
    G.ops = gdst + G.h->off_ops;
    G.big_recs = reinterpret_cast<const float4*>(gdst + G.h->off_big_recs);
    G.recs = reinterpret_cast<const float4*>(gdst + G.h->off_recs);
    return G;
}

// the big spheres, four records in flight per LDS round trip (addresses are wave-uniform: broadcast reads)
template <bool COUNT>
MIRT_DEV void test_big_spheres(const GridLds& G, f3 ro, f3 rd, float a, float inv_a, bool alive, float& closest, int& best, Work<COUNT>& work)
{
    const uint32_t n_big = G.h->n_big;
    uint32_t j = 0;
    for (; j + 4 <= n_big; j += 4) {
        const float4 r0 = G.big_recs[j], r1 = G.big_recs[j + 1], r2 = G.big_recs[j + 2], r3 = G.big_recs[j + 3];
        const uint32_t i0 = G.big[j], i1 = G.big[j + 1], i2 = G.big[j + 2], i3 = G.big[j + 3];
        test_sphere<COUNT>(r0, i0, ro, rd, a, inv_a, alive, closest, best, work);
        test_sphere<COUNT>(r1, i1, ro, rd, a, inv_a, alive, closest, best, work);
        test_sphere<COUNT>(r2, i2, ro, rd, a, inv_a, alive, closest, best, work);
        test_sphere<COUNT>(r3, i3, ro, rd, a, inv_a, alive, closest, best, work);
    }
    for (; j < n_big; ++j) test_sphere<COUNT>(G.big_recs[j], G.big[j], ro, rd, a, inv_a, alive, closest, best, work);
}

// A cell's items from its table entry `cw` (count = 0 for lanes that are not walking): the first two ids came with the entry, so their
// records are ONE round trip away (entry -> records) instead of two (entry -> ids -> records) -- and most cells list at most two
// spheres (RTIOW: 0.84 tests per visited cell).  Further items as before: ids, then records, two per round trip.
template <bool COUNT>
MIRT_DEV void test_cell_entry(const GridLds& G, uint2 cw, uint32_t count, f3 ro, f3 rd, float a, float inv_a, float& closest, int& best, Work<COUNT>& work)
{
    const uint32_t first = cw.x & 0xffffu;
    {
        const bool on0 = count > 0u, on1 = count > 1u;
        const uint32_t i0 = cw.y & 0xffffu, i1 = cw.y >> 16;                 // absent ids are stored as 0: a valid record, not tested
        const float4 r0 = G.recs[i0], r1 = G.recs[i1];
        if (ballot_(on0)) test_sphere<COUNT>(r0, i0, ro, rd, a, inv_a, on0, closest, best, work);
        if (ballot_(on1)) test_sphere<COUNT>(r1, i1, ro, rd, a, inv_a, on1, closest, best, work);
    }
    for (uint32_t n = 2; ballot_(n < count); n += 2) {
        const bool on0 = n < count, on1 = n + 1 < count;
        const uint32_t k0 = on0 ? first + n - 2u : 0u, k1 = on1 ? first + n - 1u : 0u;
        const uint32_t i0 = G.items[k0], i1 = G.items[k1];
        const float4 r0 = G.recs[i0], r1 = G.recs[i1];
        test_sphere<COUNT>(r0, i0, ro, rd, a, inv_a, on0, closest, best, work);
        if (ballot_(on1)) test_sphere<COUNT>(r1, i1, ro, rd, a, inv_a, on1, closest, best, work);
    }
}

// the items [first, first + count) of each lane's cell, two records in flight per LDS round trip
template <bool COUNT>
MIRT_DEV void test_cell_items(const GridLds& G, uint32_t first, uint32_t count, f3 ro, f3 rd, float a, float inv_a, float& closest, int& best,
                              Work<COUNT>& work)
{
    for (uint32_t n = 0; ballot_(n < count); n += 2) {
        const bool on0 = n < count, on1 = n + 1 < count;
        const uint32_t k0 = on0 ? first + n : 0u, k1 = on1 ? first + n + 1 : 0u;
        const uint32_t i0 = G.items[k0], i1 = G.items[k1];
        const float4 r0 = G.recs[i0], r1 = G.recs[i1];                    // records by sphere id
        test_sphere<COUNT>(r0, i0, ro, rd, a, inv_a, on0, closest, best, work);
        if (ballot_(on1)) test_sphere<COUNT>(r1, i1, ro, rd, a, inv_a, on1, closest, best, work);
    }
}

// FLATY: the grid is one cell high -> the two-dimensional walk (see grid_walk below for the argument)
template <bool COUNT, bool FLATY>
MIRT_DEV int nearest_hit_grid(const SceneLds& S, const GridLds& G, f3 ro, f3 rd, bool alive, float& closest_out, Work<COUNT>& work,
                              uint32_t lane)
{
    const float a = dot(rd, rd);
    const float inv_a = rcp_(a);
    float closest = kMaxT;
    int best = -1;
    const GridHeader& H = *G.h;
    if (alive) work.add(kCntRays);
    test_big_spheres<COUNT>(G, ro, rd, a, inv_a, alive, closest, best, work);

    // clip the ray against the grid's box
    const f3 org = mk(H.org[0], H.org[1], H.org[2]);
    const f3 cell = mk(H.cell[0], H.cell[1], H.cell[2]);
    const f3 inv_cell = mk(H.inv_cell[0], H.inv_cell[1], H.inv_cell[2]);
    const int dx = (int)H.dims[0], dy = (int)H.dims[1], dz = (int)H.dims[2];
    const f3 hi = mk(fma_((float)dx, cell.x, org.x), fma_((float)dy, cell.y, org.y), fma_((float)dz, cell.z, org.z));
    const float kHuge = 3.0e38f;
    const f3 inv_d = mk(rd.x != 0.0f ? rcp_(rd.x) : kHuge, rd.y != 0.0f ? rcp_(rd.y) : kHuge, rd.z != 0.0f ? rcp_(rd.z) : kHuge);
    float tmin = 0.0f, tmax = kMaxT;
    bool inside = alive;
    {
        const float o[3] = { ro.x, ro.y, ro.z }, d[3] = { rd.x, rd.y, rd.z }, id[3] = { inv_d.x, inv_d.y, inv_d.z };
        const float lo3[3] = { org.x, org.y, org.z }, hi3[3] = { hi.x, hi.y, hi.z };
#pragma unroll
        for (int k = 0; k < 3; ++k) {       // mask logic and selects, no divergent if / else per axis
            const bool nz = d[k] != 0.0f;
            const float ta = (lo3[k] - o[k]) * id[k], tb = (hi3[k] - o[k]) * id[k];
            const float t_lo = (ta < tb) ? ta : tb, t_hi = (ta < tb) ? tb : ta;
            tmin = nz ? max_(tmin, t_lo) : tmin;
            tmax = (nz & (t_hi < tmax)) ? t_hi : tmax;
            inside = inside & (nz | ((o[k] >= lo3[k]) & (o[k] <= hi3[k])));
        }
    }
    // a little slack on both ends: the walk below is clamped to the grid anyway
    bool walking = inside && (tmin <= tmax) && (tmin < closest);
    const f3 p0 = fma3(tmin, rd, ro);
    int cx = (int)((p0.x - org.x) * inv_cell.x), cy = 0, cz = (int)((p0.z - org.z) * inv_cell.z);
    cx = cx < 0 ? 0 : (cx >= dx ? dx - 1 : cx);
    cz = cz < 0 ? 0 : (cz >= dz ? dz - 1 : cz);
    if constexpr (!FLATY) { cy = (int)((p0.y - org.y) * inv_cell.y); cy = cy < 0 ? 0 : (cy >= dy ? dy - 1 : cy); }
    const int sx = rd.x > 0.0f ? 1 : -1, sy = rd.y > 0.0f ? 1 : -1, sz = rd.z > 0.0f ? 1 : -1;
    // parameter at which the ray crosses the next cell boundary on each axis, and the per-cell increment
    float tx = rd.x != 0.0f ? (fma_((float)(cx + (sx > 0 ? 1 : 0)), cell.x, org.x) - ro.x) * inv_d.x : kHuge;
    float tz = rd.z != 0.0f ? (fma_((float)(cz + (sz > 0 ? 1 : 0)), cell.z, org.z) - ro.z) * inv_d.z : kHuge;
    const float ddx = rd.x != 0.0f ? abs_(cell.x * inv_d.x) : kHuge;
    const float ddz = rd.z != 0.0f ? abs_(cell.z * inv_d.z) : kHuge;
    float ty = kHuge, ddy = kHuge;
    if constexpr (!FLATY) {
        ty = rd.y != 0.0f ? (fma_((float)(cy + (sy > 0 ? 1 : 0)), cell.y, org.y) - ro.y) * inv_d.y : kHuge;
        ddy = rd.y != 0.0f ? abs_(cell.y * inv_d.y) : kHuge;
    }

    while (ballot_(walking)) {
        uint32_t count = 0;
        if constexpr (COUNT) { if (walking) work.add(kCntCells); if (lane == 0) work.add(kCntWaveCells); }
        {
            const uint32_t c = walking ? (FLATY ? (uint32_t)(cz * dx + cx) : (uint32_t)((cz * dy + cy) * dx + cx)) : 0u;
            const uint2 cw = G.cells[c];
            count = walking ? cw.x >> 16 : 0u;
            test_cell_entry<COUNT>(G, cw, count, ro, rd, a, inv_a, closest, best, work);
        }
        if constexpr (FLATY) {   // one DDA step in two dimensions: the slab's end stops the walk through tmax
            const float t_exit = (tx < tz) ? tx : tz;
            const bool stop = (closest <= t_exit) | (t_exit > tmax);
            const bool ax = tx <= tz;
            cx += ax ? sx : 0; cz += ax ? 0 : sz;
            tx = ax ? tx + ddx : tx; tz = ax ? tz : tz + ddz;
            const bool inside_grid = ((uint32_t)cx < (uint32_t)dx) & ((uint32_t)cz < (uint32_t)dz);
            walking = walking & !stop & inside_grid;
        } else {   // one DDA step, branch-free (see grid_walk)
            const float t_exit = (tx < ty) ? ((tx < tz) ? tx : tz) : ((ty < tz) ? ty : tz);
            const bool stop = (closest <= t_exit) | (t_exit > tmax);        // nearest hit is final, or the ray left the grid
            const bool ax = (tx <= ty) & (tx <= tz);
            const bool ay = !ax & (ty <= tz);
            const bool az = !ax & !ay;
            cx += ax ? sx : 0; cy += ay ? sy : 0; cz += az ? sz : 0;
            tx = ax ? tx + ddx : tx; ty = ay ? ty + ddy : ty; tz = az ? tz + ddz : tz;
            const bool inside_grid = ((uint32_t)cx < (uint32_t)dx) & ((uint32_t)cy < (uint32_t)dy) & ((uint32_t)cz < (uint32_t)dz);
            walking = walking & !stop & inside_grid;
        }
    }
    closest_out = closest;
    return best;
}

// The same walk in instalments, for the pool kernel: a FRESH call tests the big spheres, clips the ray against the grid
// and enters it; every call then visits at most `budget` cells per lane and reports the lanes that are not finished
// (`walking`), whose state -- closest, best, the LINEAR index of the cell (16 bit) -- the pool keeps in the path's
// slot until a later OP_WALK step RESUMES it.  Lanes finish after very different numbers of cells (3.4 on average on
// the RTIOW scene, dozens for rays that graze the ground): cutting the walk into instalments lets the pool re-compact
// the unfinished paths instead of idling the finished lanes (36 % lane use in the un-cut walk of nearest_hit_grid).
// A resumed instalment recomputes the per-axis crossing parameters from the cell index with the formula the fresh
// walk starts from; the cells visited may differ from the uncut walk's in the last bit of a crossing parameter,
// which the conservative binning absorbs exactly as it absorbs the rounding of `tx += ddx` (see GridLds above).

template <bool COUNT, bool FLATY>
MIRT_DEV void grid_walk(const SceneLds& S, const GridLds& G, f3 ro, f3 rd, bool active, bool resume /* wave-uniform */, uint32_t budget,
                        float& closest, int& best, uint32_t& cellp, bool& walking, Work<COUNT>& work, uint32_t lane, Stamps& stamps,
                        const GridConsts& GC)
{
    const float a = dot(rd, rd);
    const float inv_a = rcp_(a);
    const f3 org = GC.org, cell = GC.cell;                 // the header's fields: registers (GridConsts), not LDS reads per step
    const int dx = GC.dx, dy = GC.dy, dz = GC.dz;
    const float kHuge = 3.0e38f;
    // The crossing parameters only STEER the walk (which cells, when to stop); the hit itself is decided by the exact
    // sphere tests.  An error of a few ulp in them is absorbed by the conservative binning (cells are enlarged by
    // 1e-3 cell, GridLds above) exactly like the rounding of `tx += ddx`, so the hardware reciprocal (1 ulp, one
    // instruction, no range guard) is enough here; every build and the resumed instalments use the same one.
    const f3 inv_d = mk(rd.x != 0.0f ? __builtin_amdgcn_rcpf(rd.x) : kHuge, rd.y != 0.0f ? __builtin_amdgcn_rcpf(rd.y) : kHuge,
                        rd.z != 0.0f ? __builtin_amdgcn_rcpf(rd.z) : kHuge);
    const f3 hi = GC.hi;
    // the ray's parameter range inside the grid's box (both modes: tmax ends the walk)
    float tmin = 0.0f, tmax = kMaxT;
    bool inside = active;
    {
        const float o[3] = { ro.x, ro.y, ro.z }, d[3] = { rd.x, rd.y, rd.z }, id[3] = { inv_d.x, inv_d.y, inv_d.z };
        const float lo3[3] = { org.x, org.y, org.z }, hi3[3] = { hi.x, hi.y, hi.z };
#pragma unroll
        for (int k = 0; k < 3; ++k) {       // mask logic and selects, no divergent if / else per axis
            const bool nz = d[k] != 0.0f;
            const float ta = (lo3[k] - o[k]) * id[k], tb = (hi3[k] - o[k]) * id[k];
            const float t_lo = (ta < tb) ? ta : tb, t_hi = (ta < tb) ? tb : ta;
            tmin = nz ? max_(tmin, t_lo) : tmin;
            tmax = (nz & (t_hi < tmax)) ? t_hi : tmax;
            inside = inside & (nz | ((o[k] >= lo3[k]) & (o[k] <= hi3[k])));
        }
    }
    int cx, cy, cz;
    if (!resume) {
        closest = kMaxT;
        best = -1;
        if (active) work.add(kCntRays);
        stamps.mark(4);                                       // (diagnosis builds) the clip against the grid's box
        if (GC.n_big <= 4u) {                                 // wave-uniform: the records live in registers
            if (GC.n_big > 0u) test_sphere<COUNT>(GC.bigr[0], GC.bigid[0], ro, rd, a, inv_a, active, closest, best, work);
            if (GC.n_big > 1u) test_sphere<COUNT>(GC.bigr[1], GC.bigid[1], ro, rd, a, inv_a, active, closest, best, work);
            if (GC.n_big > 2u) test_sphere<COUNT>(GC.bigr[2], GC.bigid[2], ro, rd, a, inv_a, active, closest, best, work);
            if (GC.n_big > 3u) test_sphere<COUNT>(GC.bigr[3], GC.bigid[3], ro, rd, a, inv_a, active, closest, best, work);
        } else test_big_spheres<COUNT>(G, ro, rd, a, inv_a, active, closest, best, work);
        stamps.mark(3);
        walking = inside && (tmin <= tmax) && (tmin < closest);
        const f3 inv_cell = GC.inv_cell;
        const f3 p0 = fma3(tmin, rd, ro);
        cx = (int)((p0.x - org.x) * inv_cell.x); cz = (int)((p0.z - org.z) * inv_cell.z);
        cx = cx < 0 ? 0 : (cx >= dx ? dx - 1 : cx);
        cz = cz < 0 ? 0 : (cz >= dz ? dz - 1 : cz);
        if constexpr (FLATY) cy = 0;
        else { cy = (int)((p0.y - org.y) * inv_cell.y); cy = cy < 0 ? 0 : (cy >= dy ? dy - 1 : cy); }
    } else {
        walking = active;
        // the parked LINEAR index taken apart: floor((n + 0.5) * (1 / d)) == n / d exactly for n, d <= 8192
        const int nxy = FLATY ? dx : dx * dy;
        cz = (int)(((float)cellp + 0.5f) * GC.inv_dim_xy);
        const int rem = (int)cellp - cz * nxy;
        if constexpr (FLATY) { cy = 0; cx = rem; }
        else { cy = (int)(((float)rem + 0.5f) * GC.inv_dim_x); cx = rem - cy * dx; }
    }
    const int sx = rd.x > 0.0f ? 1 : -1, sy = rd.y > 0.0f ? 1 : -1, sz = rd.z > 0.0f ? 1 : -1;
    // parameter at which the ray crosses the next cell boundary on each axis, and the per-cell increment
    float tx = rd.x != 0.0f ? (fma_((float)(cx + (sx > 0 ? 1 : 0)), cell.x, org.x) - ro.x) * inv_d.x : kHuge;
    float tz = rd.z != 0.0f ? (fma_((float)(cz + (sz > 0 ? 1 : 0)), cell.z, org.z) - ro.z) * inv_d.z : kHuge;
    const float ddx = rd.x != 0.0f ? abs_(cell.x * inv_d.x) : kHuge;
    const float ddz = rd.z != 0.0f ? abs_(cell.z * inv_d.z) : kHuge;
    float ty = kHuge, ddy = kHuge;
    if constexpr (!FLATY) {
        ty = rd.y != 0.0f ? (fma_((float)(cy + (sy > 0 ? 1 : 0)), cell.y, org.y) - ro.y) * inv_d.y : kHuge;
        ddy = rd.y != 0.0f ? abs_(cell.y * inv_d.y) : kHuge;
    }

    // linear cell index, advanced with the walk (one multiply-add pair per WALK instead of per cell; the strides are selects)
    uint32_t cidx = FLATY ? (uint32_t)(cz * dx + cx) : (uint32_t)((cz * dy + cy) * dx + cx);
    const int nxy_ = FLATY ? dx : dx * dy;
    const int stride_y = sy > 0 ? dx : -dx, stride_z = sz > 0 ? nxy_ : -nxy_;
    if (resume) stamps.mark(7); else stamps.mark(4);
    // (Letting an instalment run past its budget while most lanes are still walking was measured: the fuller instalments gain 1-2 %,
    //  but the loop header it needs -- ballot, population count and two compares instead of `it < budget && any` -- costs this loop 5 %.)
    // (The budget is a compile-time constant when both kinds of instalment have the same one, and the loop is then unrolled: a
    //  run-time bound -- different budgets for fresh and resumed walks -- was measured 5 % slower at every pair of values tried.)
    // Software-pipelined walk: the DDA step does not depend on the sphere tests (only `stop` does), so the NEXT cell's table entry is
    // requested before the current cell's records are tested -- one exposed LDS round trip per cell (the records) instead of two.
    uint2 cw = G.cells[walking ? cidx : 0u];
#pragma unroll
    for (uint32_t it = 0; it < budget && ballot_(walking); ++it) {
        if constexpr (COUNT) { if (walking) work.add(kCntCells); if (lane == 0) work.add(kCntWaveCells); }
        const uint32_t count = walking ? cw.x >> 16 : 0u;
        const uint2 cw_now = cw;
        float t_exit;
        bool inside_grid;
        if constexpr (FLATY) {
            // A grid ONE cell high (spheres on a ground plane: RTIOW): the walk is two-dimensional.  The y slab ends the walk through
            // tmax -- the clip's far parameter on y is the very expression the 3-D walk's `ty` starts from (cy = 0, dims[1] = 1), so
            // "y is the nearest crossing" there is "t_exit > tmax" here; on an exact tie of a z crossing with the slab's end this walk
            // may test one more cell, which can only add candidates the flat scan tests as well.
            t_exit = (tx < tz) ? tx : tz;
            const bool ax = tx <= tz;
            cx += ax ? sx : 0; cz += ax ? 0 : sz;
            cidx += (uint32_t)(ax ? sx : stride_z);
            tx = ax ? tx + ddx : tx; tz = ax ? tz : tz + ddz;
            inside_grid = ((uint32_t)cx < (uint32_t)dx) & ((uint32_t)cz < (uint32_t)dz);
        } else {
            t_exit = (tx < ty) ? ((tx < tz) ? tx : tz) : ((ty < tz) ? ty : tz);
            const bool ax = (tx <= ty) & (tx <= tz);
            const bool ay = !ax & (ty <= tz);
            const bool az = !ax & !ay;
            cx += ax ? sx : 0; cy += ay ? sy : 0; cz += az ? sz : 0;
            cidx += (uint32_t)(ax ? sx : (ay ? stride_y : stride_z));
            tx = ax ? tx + ddx : tx; ty = ay ? ty + ddy : ty; tz = az ? tz + ddz : tz;
            inside_grid = ((uint32_t)cx < (uint32_t)dx) & ((uint32_t)cy < (uint32_t)dy) & ((uint32_t)cz < (uint32_t)dz);
        }
        const bool may_go_on = walking & inside_grid & !(t_exit > tmax);
        if (it + 1 < budget) cw = G.cells[may_go_on ? cidx : 0u];                // in flight while the tests below run
        test_cell_entry<COUNT>(G, cw_now, count, ro, rd, a, inv_a, closest, best, work);
        walking = may_go_on & !(closest <= t_exit);                              // the nearest hit is final once it lies inside the cells visited
    }
    if (resume) stamps.mark(8); else stamps.mark(5);
    cellp = cidx;                                                             // meaningful where `walking` is still set
}

MIRT_DEV f3 rand_in_unit_sphere(Rng& rng)     // wgsl:480-491
{
    const float r = pow_unit(rng.next(), 0.33333f);
    const float theta = rng.next_scaled(kPi * 0x1p-32f);
    const float phi = rng.next_scaled(kTwoPi * 0x1p-32f);
    const SinCos t = sincos_small(theta), p = sincos_small(phi);
    const float rs = r * t.s;
    return mk(rs * p.c, rs * p.s, r * t.c);
}

MIRT_DEV f3 reflect3(f3 v, f3 n) { return fma3(-(2.0f * dot(v, n)), n, v); }

// A material as the shading routines see it: either a pointer to the PreparedMaterial table (LDS in the flat builds,
// global memory in the strip kernel's grid build) or MatRegs, a copy in registers that arrived with the hit sphere's
// ShadeRec (pool kernel, grid build).
MIRT_DEV float4   mat_tex(const PreparedMaterial* m, int k) { return *reinterpret_cast<const float4*>(&m->tex[k][0]); }    // one ds_read_b128
MIRT_DEV uint32_t mat_flags(const PreparedMaterial* m) { return m->flags; }
MIRT_DEV float    mat_x(const PreparedMaterial* m) { return m->x; }
MIRT_DEV float    mat_inv_x(const PreparedMaterial* m) { return m->inv_x; }
struct MatRegs { uint32_t id; float x, inv_x; uint32_t flags; float4 t0, t1; };
MIRT_DEV float4   mat_tex(const MatRegs& m, int k) { return make_float4(k ? m.t1.x : m.t0.x, k ? m.t1.y : m.t0.y, k ? m.t1.z : m.t0.z, k ? m.t1.w : m.t0.w); }
MIRT_DEV uint32_t mat_flags(const MatRegs& m) { return m.flags; }
MIRT_DEV float    mat_x(const MatRegs& m) { return m.x; }
MIRT_DEV float    mat_inv_x(const MatRegs& m) { return m.inv_x; }

// Albedo of texture k of a material at a hit with outward normal n: sphereIntersection's (u,v)
// (wgsl:434-437) fed to textureLookup (wgsl:377-387).
// 1x1 textures — every colour material in the reference's scenes — take a shortcut that is
// EXACT: the lookup returns texel `offset` unless u rounds to 1.0 (j = 1) or 1-v rounds to 1.0
// (i = 1).  u = phi/(2pi) reaches 1.0 only for atan2(-n.z, n.x) within ~4e-7 of +pi, i.e.
// n.x < 0 and |n.z| <= ~5e-7 |n.x|; 1-v reaches 1.0 only for acos(-n.y) < 1e-7, i.e. n.y <= -1.
// The guards below are 20x / 1e-6 wider than that (and false for NaN), and the rare lanes they
// catch run the full formula.
template <class M>
MIRT_DEV f3 albedo_at(const RenderArgs& A, const M& m, int k, f3 n, TexelTile* T = nullptr)
{
    const float4 t4 = mat_tex(m, k);
    const bool one = (mat_flags(m) >> k) & 1u;
    // plain mask logic (no short-circuit) so that the shortcut is straight-line code; the full lookup sits
    // behind ONE wave-uniform, rarely taken branch
    const bool fast = one & (n.y > -0.999999f) & ((n.x >= 0.0f) | (abs_(n.z) > 1.0e-5f * abs_(n.x)));
    f3 c = mk(t4.x, t4.y, t4.z);
    if (__builtin_expect(ballot_(!fast) != 0ull, 0)) {
        asm volatile("; albedo_at: full texture lookup" ::);
        if (!fast) {
            const float theta = acos_(-n.y);
            const float phi = atan2_(-n.z, n.x) + kPi;
            const float u = (0.5f * kFrac1Pi) * phi;
            const float v = kFrac1Pi * theta;
            const uint32_t w = one ? 1u : bits(t4.x);
            const uint32_t h = one ? 1u : bits(t4.y);
            const uint32_t off = one ? bits(t4.w) : bits(t4.z);
            c = texture_lookup(A, w, h, off, u, v, T);
        }
    }
    return c;
}

// scatterLambertian (wgsl:204-242): cosine-weighted direction around n through the Pixar ONB
template <class M>
MIRT_DEV void scatter_lambertian(const RenderArgs& A, const M& m, int k, f3 n, Rng& rng, f3& dir, f3& atten, TexelTile* T = nullptr)
{
    const float phi = rng.next_scaled(kTwoPi * 0x1p-32f);      // 2 pi r1 (r1 is used nowhere else)
    const float r2 = rng.next();
    const float sqrt_r2 = sqrt_unit(r2);
    const float z = sqrt_unit(1.0f - r2);
    const SinCos sc = sincos_small(phi);
    const float lx = sc.c * sqrt_r2;
    const float ly = sc.s * sqrt_r2;
    const float sg = (n.z >= 0.0f) ? 1.0f : -1.0f;
    const float aa = -rcp_in_range(sg + n.z);          // -(1/x) == -1/x exactly; |sg + n.z| lies in [1, 2] for a unit normal
    const float bb = n.x * n.y * aa;
    const f3 U = mk(fma_(sg * n.x, n.x * aa, 1.0f), sg * bb, -(sg * n.x));
    const f3 V = mk(bb, fma_(n.y, n.y * aa, sg), -n.y);
    const f3 wi = fma3(z, n, fma3(ly, V, lx * U));
    const float dn = dot(n, wi);
    // eval/pdf (wgsl:206): (FRAC_1_PI*max(eps,dn)) / max(eps, dn*FRAC_1_PI).  When dn*FRAC_1_PI > eps both
    // max() pick their second argument and the quotient is x/x = 1 exactly; only grazing directions divide.
    const float dnc = dn * kFrac1Pi;
    float kk = 1.0f;
    const bool grazing = !(dnc > kEpsilon);
    if (__builtin_expect(ballot_(grazing) != 0ull, 0)) {
        asm volatile("; scatter_lambertian: grazing direction" ::);
        if (grazing) kk = (kFrac1Pi * max_(kEpsilon, dn)) / max_(kEpsilon, dnc);
    }
    atten = albedo_at(A, m, k, n, T);
    if (__builtin_expect(ballot_(grazing) != 0ull, 0)) {      // kk == 1.0f elsewhere, and 1.0f * x == x
        asm volatile("; scatter_lambertian: grazing attenuation" ::);
        atten = kk * atten;
    }
    dir = wi;
}

// The five scatter routines of scatterRay (wgsl:174-314), one function each so that the pool
// kernel can run exactly one of them per wave.  rd = incoming direction, hp/hn = hit point/normal.
template <class M>
MIRT_DEV void shade_lambertian(const RenderArgs& A, const M& m, f3 hn, Rng& rng, f3& ndir, f3& att, TexelTile* T = nullptr)
{
    scatter_lambertian(A, m, 0, hn, rng, ndir, att, T);
}

template <class M>
MIRT_DEV void shade_metal(const RenderArgs& A, const M& m, f3 rd, f3 hn, Rng& rng, f3& ndir, f3& att, TexelTile* T = nullptr)
{   // scatterMetal wgsl:244-248
    const f3 refl = reflect3(rd, hn);
    const f3 rs = rand_in_unit_sphere(rng);
    ndir = fma3(mat_x(m), rs, refl);
    att = albedo_at(A, m, 0, hn, T);
}

template <class M>
MIRT_DEV void shade_dielectric(const M& m, f3 rd, f3 hn, Rng& rng, f3& ndir, f3& att)
{   // scatterDielectric wgsl:250-292; the Schlick draw is consumed and the reflection discarded (wgsl:266-273)
    const float dn = dot(rd, hn);
    const f3 uvn = normalize(rd);
    const bool inside = dn > 0.0f;
    const f3 outn = inside ? -hn : hn;
    const float ratio = inside ? mat_x(m) : mat_inv_x(m);
    const float dt = dot(uvn, outn);
    const float disc = fma_(-(ratio * ratio), fma_(-dt, dt, 1.0f), 1.0f);
    // both outcomes in straight-line code and a select (no divergent if/else): total internal reflection
    // keeps the reflected direction and does NOT consume the Schlick draw
    const bool refracts = disc > 0.0f;
    const float sq = refracts ? sqrt_unit_where(disc, refracts) : 0.0f;   // disc <= 1; 0 keeps the unused lanes finite
    const f3 q = fma3(-dt, outn, uvn);
    const f3 refr = normalize(fma3(-sq, outn, ratio * q));
    const f3 refl = reflect3(rd, hn);
    Rng drawn = rng;
    (void)drawn.next();
    rng.state = refracts ? drawn.state : rng.state;
    ndir = mk(refracts ? refr.x : refl.x, refracts ? refr.y : refl.y, refracts ? refr.z : refl.z);
    att = mk(1, 1, 1);
}

template <class M>
MIRT_DEV void shade_checkerboard(const RenderArgs& A, const M& m, f3 hp, f3 hn, Rng& rng, f3& ndir, f3& att, TexelTile* T = nullptr)
{   // scatterCheckerboard wgsl:300-307: sign of sin(5x)sin(5y)sin(5z) from the signs of the factors
    const bool negative = sin_product_negative(5.0f * hp.x, 5.0f * hp.y, 5.0f * hp.z);
    scatter_lambertian(A, m, negative ? 0 : 1, hn, rng, ndir, att, T);
}

MIRT_DEV void shade_missing(f3 hn, Rng& rng, f3& ndir, f3& att)
{   // scatterMissingMaterial wgsl:309-314
    const f3 rs = rand_in_unit_sphere(rng);
    ndir = hn + rs;
    att = mk(0.9921f, 0.24705f, 0.57254f);
}

// scatterRay's switch (wgsl:174-202) for kernels that shade per lane (strip kernel; pool kernel, grid build).  `counted` = this lane
// holds a real path (the pool kernel runs the routines on idle lanes too).
struct Scattered { f3 dir, att; };
template <bool COUNT, class M>
MIRT_DEV Scattered shade_by_id(const RenderArgs& A, const M& m, uint32_t id, bool counted, f3 rd, f3 hp, f3 hn, Rng& rng, Work<COUNT>& work)
{
    f3 ndir = rd, att = mk(1, 1, 1);
    switch (id) {
    case 0u: if (counted) work.add(kCntScatter0); shade_lambertian(A, m, hn, rng, ndir, att); break;
    case 1u: if (counted) work.add(kCntScatter1); shade_metal(A, m, rd, hn, rng, ndir, att); break;
    case 2u: if (counted) work.add(kCntScatter2); shade_dielectric(m, rd, hn, rng, ndir, att); break;
    case 3u: if (counted) work.add(kCntScatter3); shade_checkerboard(A, m, hp, hn, rng, ndir, att); break;
    default: if (counted) work.add(kCntScatter4); shade_missing(hn, rng, ndir, att); break;
    }
    return Scattered{ ndir, att };
}

// Pool kernel, grid build: scatterRay's switch with everything a hit needs arriving as late as it is needed.  The routine id and the
// sphere's centre come from LDS (GridLds.ops / recs, the tables the trace reads anyway), so the switch and the routines' random draws --
// which need neither the normal nor the material -- run while the two 16-byte global loads of the ShadeRec (`ra` = {1/r, x, 1/x, flags},
// `t0` = first texture; the checkerboard's second texture is a third load inside its case) are still in flight.  Same arithmetic in the
// same order of RNG draws as shade_by_id: the routines only take their normal later.
template <bool COUNT>
MIRT_DEV Scattered shade_grid_hit(const RenderArgs& A, uint32_t id, const float4* rec, const float4 ra, const float4 t0, f3 centre, bool counted,
                                  f3 rd, f3 hp, Rng& rng, Work<COUNT>& work)
{
    f3 ndir = rd, att = mk(1, 1, 1);
    switch (id) {
    case 0u: {
        if (counted) work.add(kCntScatter0);
        // scatter_lambertian with the normal taken after the draws
        const float phi = rng.next_scaled(kTwoPi * 0x1p-32f);
        const float r2 = rng.next();
        const float sqrt_r2 = sqrt_unit(r2);
        const float z = sqrt_unit(1.0f - r2);
        const SinCos sc = sincos_small(phi);
        const float lx = sc.c * sqrt_r2;
        const float ly = sc.s * sqrt_r2;
        const f3 n = ra.x * (hp - centre);
        const MatRegs m{ 0u, ra.y, ra.z, bits(ra.w), t0, t0 };
        const float sg = (n.z >= 0.0f) ? 1.0f : -1.0f;
        const float aa = -rcp_in_range(sg + n.z);
        const float bb = n.x * n.y * aa;
        const f3 U = mk(fma_(sg * n.x, n.x * aa, 1.0f), sg * bb, -(sg * n.x));
        const f3 V = mk(bb, fma_(n.y, n.y * aa, sg), -n.y);
        const f3 wi = fma3(z, n, fma3(ly, V, lx * U));
        const float dn = dot(n, wi);
        const float dnc = dn * kFrac1Pi;
        float kk = 1.0f;
        const bool grazing = !(dnc > kEpsilon);
        if (__builtin_expect(ballot_(grazing) != 0ull, 0)) {
            asm volatile("; shade_grid_hit: grazing direction" ::);
            if (grazing) kk = (kFrac1Pi * max_(kEpsilon, dn)) / max_(kEpsilon, dnc);
        }
        att = albedo_at(A, m, 0, n);
        if (__builtin_expect(ballot_(grazing) != 0ull, 0)) {
            asm volatile("; shade_grid_hit: grazing attenuation" ::);
            att = kk * att;
        }
        ndir = wi;
        break;
    }
    case 1u: {
        if (counted) work.add(kCntScatter1);
        const f3 rs = rand_in_unit_sphere(rng);                 // the draws first: they need neither normal nor material
        const f3 n = ra.x * (hp - centre);
        const MatRegs m{ 1u, ra.y, ra.z, bits(ra.w), t0, t0 };
        const f3 refl = reflect3(rd, n);
        ndir = fma3(mat_x(m), rs, refl);
        att = albedo_at(A, m, 0, n);
        break;
    }
    case 2u: {
        if (counted) work.add(kCntScatter2);
        const f3 n = ra.x * (hp - centre);
        const MatRegs m{ 2u, ra.y, ra.z, bits(ra.w), t0, t0 };
        shade_dielectric(m, rd, n, rng, ndir, att);
        break;
    }
    case 3u: {
        if (counted) work.add(kCntScatter3);
        const float4 t1 = rec[2];
        const f3 n = ra.x * (hp - centre);
        const MatRegs m{ 3u, ra.y, ra.z, bits(ra.w), t0, t1 };
        shade_checkerboard(A, m, hp, n, rng, ndir, att);
        break;
    }
    default: {
        if (counted) work.add(kCntScatter4);
        const f3 rs = rand_in_unit_sphere(rng);
        const f3 n = ra.x * (hp - centre);
        ndir = n + rs;
        att = mk(0.9921f, 0.24705f, 0.57254f);
        break;
    }
    }
    return Scattered{ ndir, att };
}


// radiance() wgsl:316-343, one channel of the Hosek-Wilkie state held in LDS
MIRT_DEV float hosek_radiance(const float* sky, float theta, float gamma, int ch)
{
    const float* p = sky + 9 * ch;
    const float r = sky[27 + ch];
    const float cg = sincos_(gamma).c;
    const float cg2 = cg * cg;
    const float ct = abs_(sincos_(theta).c);
    const float expm = exp_(p[4] * gamma);
    const float base = fma_(-(2.0f * p[8]), cg, fma_(p[8], p[8], 1.0f));
    const float mie = (1.0f + cg2) / (base * sqrt_(base));
    const float zenith = sqrt_(ct);
    const float lhs = fma_(p[0], exp_(p[1] / (ct + 0.01f)), 1.0f);
    const float rhs = fma_(p[7], zenith, fma_(p[6], mie, fma_(p[5], cg2, fma_(p[3], expm, p[2]))));
    return r * (lhs * rhs);
}

template <bool HOSEK>
MIRT_DEV f3 sky_color(const SceneLds& S, f3 d)
{
    const f3 v = normalize(d);
    if constexpr (HOSEK) {
        const f3 s = mk(S.sky[32], S.sky[33], S.sky[34]);
        const float theta = acos_(v.y);
        const float gamma = acos_(dot(v, s));
        return mk(hosek_radiance(S.sky, theta, gamma, 0), hosek_radiance(S.sky, theta, gamma, 1),
                  hosek_radiance(S.sky, theta, gamma, 2));
    } else {
        const float t = 0.5f * (v.y + 1.0f);
        const float omt = 1.0f - t;
        return mk(fma_(t, 0.5f, omt), fma_(t, 0.7f, omt), fma_(t, 1.0f, omt));
    }
}

MIRT_DEV uint32_t to_fixed(float c)          // 2^-20 units, clamped to [0, 4096)
{
    const float p = (c > 0.0f) ? c : 0.0f;                  // NaN and negatives -> 0 (select, no branch)
    const float s = __builtin_fminf(p * 1048576.0f, 4294967040.0f);    // p is never NaN here: one v_min_f32
    return (uint32_t)s;
}

MIRT_DEV float uncharted2_tonemap(float x)   // wgsl:94-103
{
    const float A_ = 0.15f, B_ = 0.50f, CB = 0.05f, DE = 0.004f, DF = 0.06f;
    const float EF = 0.02f / 0.30f;
    const float num = fma_(x, fma_(A_, x, CB), DE);
    const float den = fma_(x, fma_(A_, x, B_), DF);
    return num / den - EF;
}

MIRT_DEV uint32_t resolve_channel(unsigned long long sum, uint32_t n_samples, uint32_t flags)
{
    // mean = sum / (n * 2^20), in double, rounded once to float (mirt-math v1's definition of the mean).  A power-of-two sample count -- the reference adds
    // 2 per frame -- makes the divisor a power of two: the quotient is an exact scaling of (double)sum (one v_ldexp_f64), without the f64
    // division (about 20 of the ~105 instructions of a channel).  Wave-uniform choice; n_samples >= 1.
    const double denom = (double)n_samples * 1048576.0;
    float m;
    if ((n_samples & (n_samples - 1u)) == 0u) m = (float)__builtin_ldexp((double)sum, -(int)(20u + (uint32_t)__builtin_ctz(n_samples)));
    else m = (float)((double)sum / denom);
    if (!(flags & MIRT_FLAG_NO_TONEMAP)) {   // uncharted2 wgsl:83-92
        const float curr = uncharted2_tonemap(0.246f * m);
        const float white = rcp_(uncharted2_tonemap(11.2f));
        m = white * curr;
    }
    if (!(flags & MIRT_FLAG_NO_SRGB))        // the Bgra8UnormSrgb surface's transfer curve
        m = (m > 0.0031308f) ? fma_(1.055f, pow_pos(m, 0.41666666f), -0.055f) : 12.92f * m;
    if (!(m > 0.0f)) return 0u;
    m = (m > 1.0f) ? 1.0f : m;
    return (uint32_t)fma_(m, 255.0f, 0.5f);
}

// Grid builds: the camera rays of one strip -- up to 16 pixels x spp samples, 39 % of all rays of the RTIOW scene -- leave a lens of a
// few centimetres through a footprint of a few pixels on the focal plane: one thin bundle.  The wave therefore selects, ONCE per strip, the
// spheres that bundle can touch (strip_candidates below: a conservative bound, a superset is all that is needed), and the primary rays of
// the strip scan that short list -- typically the ground and two or three spheres -- instead of paying the always-tested big spheres, the
// clip, the grid entry and a walk of 4-5 cells each (camera rays graze the slab of small spheres: they are the longest walkers).  The
// nearest hit over a superset of the spheres a ray can hit is the nearest hit over all spheres, test by test the same arithmetic
// (test_sphere) and tie-break: identical images.  More than kMaxCand candidates, a strip that wraps into the next row or a bundle that is
// not thin: the strip's camera rays take the grid like every other ray.
constexpr uint32_t kMaxCand = 16, kNoCand = 0xffffffffu;
// (Flat scenes -- a handful of spheres, no grid -- were given the lists too, so that a strip that sees only the ground tests one sphere and a
//  sky strip none: measured +2.5 % on config 3 and +2 % on config 4, whose three / two sphere records already sit in scalar registers;
//  -1.7 % on the five-sphere main.rs scene.  Removed: profiles/r03_flat_ab.txt block 7.)
constexpr uint32_t kCandBytes = 48;       // u16 ids [kMaxCand] | u32 count (kNoCand: use the grid) | pad

// The candidate list of a strip's camera rays (see above).  Pixels [x0, x0 + n) of image row y; all lanes take part.
// A camera ray (wgsl:456-478) starts at O = eye + e, |e| <= R (the lens), and aims at a point F of the strip's footprint on the
// focal plane, |F - Fc| <= delta (Fc its centre): P(t) = O + t (F - O) = eye + t (Fc - eye) + w with |w| <= t (delta + R) + R.
// With a = Fc - eye, D = |a|, and for a sphere (c, r): s = (c - eye) . a / D, d = distance of c from the axis.  A hit needs
// |c - P(t)| <= r for some t >= 0, hence |s - t D| <= r + R + t g and d <= r + R + t g with g = delta + R; the first gives
// t <= tmax = (s + r + R) / (D - g) (and s + r + R >= 0), the second then d <= r + R + tmax g.  Everything is evaluated with slack
// (2 % on g, 5 % + 0.01 + 0.001 |c - eye| on the bound) that dwarfs the rounding of these few float operations, and every
// comparison is written so that a NaN keeps the sphere: the list is a superset by construction.
MIRT_DEV uint32_t strip_candidates(const RenderArgs& AP, const SceneLds& S, uint32_t x0, uint32_t n, uint32_t y, unsigned short* ids, uint32_t lane)
{
    const f3 eye = mk(S.cam[0], S.cam[1], S.cam[2]), hor = mk(S.cam[4], S.cam[5], S.cam[6]), ver = mk(S.cam[8], S.cam[9], S.cam[10]);
    const f3 cam_u = mk(S.cam[12], S.cam[13], S.cam[14]), cam_v = mk(S.cam[16], S.cam[17], S.cam[18]), llc = mk(S.cam[20], S.cam[21], S.cam[22]);
    const float lens_radius = S.cam[19];
    const float inv_w = 1.0f / (float)AP.width, inv_h = 1.0f / (float)AP.height;
    const float u0 = (float)x0 * inv_w, u1 = (float)(x0 + n) * inv_w;
    const float v1 = 1.0f - (float)y * inv_h, v0 = 1.0f - (float)(y + 1u) * inv_h;
    const float uc = 0.5f * (u0 + u1), vc = 0.5f * (v0 + v1), du = 0.5f * (u1 - u0), dv = 0.5f * (v1 - v0);
    const f3 axis = fma3(vc, ver, fma3(uc, hor, llc)) - eye;
    const float D = __builtin_sqrtf(dot(axis, axis));
    const float delta = abs_(du) * __builtin_sqrtf(dot(hor, hor)) + abs_(dv) * __builtin_sqrtf(dot(ver, ver));     // triangle inequality over the corners
    const float R = 1.5f * abs_(lens_radius) * max_(__builtin_sqrtf(dot(cam_u, cam_u)), __builtin_sqrtf(dot(cam_v, cam_v)));
    const float g = 1.02f * (delta + R) + 1.0e-6f * D;
    if (!(D > 0.0f && D < 1.0e30f && g < 0.25f * D)) return kNoCand;          // not a thin bundle (or NaN): no list
    const float inv_D = 1.0f / D;
    const f3 ah = inv_D * axis;
    const float inv_den = 1.0f / (D - g);
    uint32_t count = 0;
    for (uint32_t base = 0; base < AP.n_spheres; base += 64u) {
        const uint32_t i = base + lane;
        const bool valid = i < AP.n_spheres;
        const PreparedSphere sp = AP.spheres[valid ? i : 0u];
        const f3 q = mk(sp.cx, sp.cy, sp.cz) - eye;
        const float r = abs_(sp.radius);
        const float s = dot(q, ah), qq = dot(q, q);
        const float d2 = qq - s * s;
        const float dperp = __builtin_sqrtf(d2 > 0.0f ? d2 : 0.0f);
        const float len_q = __builtin_sqrtf(qq);
        const float margin = 0.01f + 1.0e-3f * len_q;
        const float reach = s + r + R;
        const float tmax = (reach > 0.0f ? reach : 0.0f) * inv_den;
        const float bound = 1.05f * (r + R + tmax * g) + margin;
        const bool out = (reach < -margin) | (dperp > bound);                    // false for NaN: the sphere stays
        const bool cand = valid & !out;
        const unsigned long long mask = ballot_(cand);
        const uint32_t more = (uint32_t)__popcll(mask);
        if (count + more > kMaxCand) return kNoCand;                             // wave-uniform
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (cand) ids[count + rank] = (unsigned short)i;
        count += more;
    }
    return count;
}

// ------------------------------------------------------------------------------------------
// render_pt_strip — one path per lane, per-lane material switch
// ------------------------------------------------------------------------------------------

// rayColor wgsl:124-172 for the 64 paths of a wave: returns throughput x sky colour (0 if the bounce limit
// ended the path).  The loop leaves as soon as no lane of the wave has a live path.
template <bool COUNT, bool HOSEK, bool GRID>
// `cand` / n_cand (GRID builds, lane = pixel): the candidate list of this unit's camera rays (strip_candidates), scanned instead of the
// grid at bounce 0; n_cand = kNoCand: none.
MIRT_DEV f3 path_radiance(const RenderArgs& A, const SceneLds& S, const GridLds& G, bool alive, Rng& rng, f3 ro, f3 rd,
                          Work<COUNT>& work, uint32_t lane, const unsigned short* cand = nullptr, uint32_t n_cand = kNoCand)
{
    f3 thr = mk(1, 1, 1);
    f3 color = mk(0, 0, 0);
    for (uint32_t bounce = 0; bounce < A.num_bounces; ++bounce) {
        if (!ballot_(alive)) break;
        if constexpr (COUNT) {
            if (lane == 0) work.add(kCntWaveIters);
            if (alive) work.add(kCntLaneIters);
        }
        float closest;
        int best;
        if constexpr (GRID) {
            if (bounce == 0u && n_cand != kNoCand) {          // wave-uniform: camera rays scan their unit's candidate list
                closest = kMaxT;
                best = -1;
                if (alive) work.add(kCntRays);
                const float a = dot(rd, rd);
                const float inv_a = rcp_(a);
                for (uint32_t j = 0; j < n_cand; ++j) {
                    const uint32_t id = cand[j];
                    test_sphere<COUNT>(G.recs[id], id, ro, rd, a, inv_a, alive, closest, best, work);
                }
            } else {
                // a grid ONE cell high (spheres on a ground plane) takes the two-dimensional walk; the flag is wave-uniform (host: dims[1] == 1)
                best = A.grid_flat_y ? nearest_hit_grid<COUNT, true>(S, G, ro, rd, alive, closest, work, lane)
                                     : nearest_hit_grid<COUNT, false>(S, G, ro, rd, alive, closest, work, lane);
            }
        } else best = nearest_hit<COUNT>(S, A.n_spheres, ro, rd, alive, closest, work);
        if (alive) {
            if (best >= 0) {
                work.add(kCntHits);
                // sphereIntersection wgsl:431-440
                PreparedSphere sp;
                if constexpr (GRID) sp = A.spheres[best];           // grid builds keep no sphere table in LDS: one global read per hit
                else sp = S.spheres[best];
                const f3 hp = fma3(closest, rd, ro);
                const f3 hn = sp.inv_r * (hp - mk(sp.cx, sp.cy, sp.cz));
                const PreparedMaterial* m = &S.pmats[sp.material_idx];
                const Scattered sc = shade_by_id<COUNT>(A, m, m->id, true, rd, hp, hn, rng, work);
                ro = hp;
                rd = sc.dir;
                thr = thr * sc.att;
            } else {
                work.add(kCntSky);
                color = sky_color<HOSEK>(S, rd);
                alive = false;
            }
        }
    }
    return thr * color;
}

// BY_PIXEL = false: a wave owns a strip of 16 pixels and takes them one at a time, lane = sample
//                   (s = lane, lane + 64, ...), 64-lane integer reduction per pixel.
// BY_PIXEL = true : a wave owns 64 consecutive pixels, lane = pixel, and every lane walks its pixel's samples
//                   in turn -- the launch shape of the reference's interactive loop, which adds 2 samples per
//                   pixel and frame (mod.rs:606-611): all 64 lanes carry a path whatever spp is, no reduction.
// The plain lane-per-pixel build is held to 64 VGPRs (8 waves per SIMD instead of 7 at 71): -11 % on the single-sphere
// scene (config 2: 1.54 -> 1.37 ms), -1 % on 2-spp frames, +1 ... +3 % on divergent scenes at 16-32 spp; the Hosek build
// would spill and keeps its registers.  The grid build of the same schedule (many-sphere scenes below 16 spp) runs 7 waves
// per SIMD at 72 VGPRs instead of 5 at 90, a few spilled registers included: RTIOW 2 / 8 spp 0.66 -> 0.62 / 1.92 -> 1.74 ms
// (6 waves -2 / -6 %, 8 waves -5 / -8 %).
//
// STREAM (lane = pixel, flat scenes; render_pt_stream_kernel): the lanes do not wait for each other between samples.  A lane whose path has
// ended takes its pixel's NEXT sample as soon as kStreamRefillMin lanes of the wave want one (or none has a live path), so a wave lasts as long
// as its lane with the longest SUM of path lengths, not the sum over samples of the longest path.  It pays from 8 samples per lane on
// (profiles/r04_lowspp_ab.txt block 9: config 2 -7 %, three spheres 16 / 24 spp -3 / -7 %; below that the fresh camera rays it mixes with old
// paths make a step run more shading routines than it saves steps: 2 / 8 spp +10 / +4 %).  Same samples, same exact integer sums.
constexpr uint32_t kStreamRefillMin = 16;
template <bool COUNT, bool HOSEK, bool GRID, bool BY_PIXEL, bool STREAM>
MIRT_DEV void strip_kernel_body(const RenderArgs& A)
{
    static_assert(!STREAM || (BY_PIXEL && !GRID && !COUNT), "the streaming schedule exists for lane = pixel in flat scenes");
    extern __shared__ __align__(16) unsigned char smem[];
    // many-sphere scenes (GRID build): the material table (one 48-byte read per hit) stays in global memory / L2
    // so that LDS holds only what every sphere TEST reads, and more waves fit a CU
    const SceneLds S = stage_scene<true, !GRID>(A, smem, HOSEK);
    GridLds G{};
    if constexpr (GRID) G = stage_grid(A, smem + scene_lds_bytes_dev(A.n_spheres, A.n_mats, HOSEK, false));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t npix = A.out_rows * A.width;

    Work<COUNT> work;
    work.clear();

    // BY_PIXEL units are small (16-64 pixels x a few samples).  With A.static_units they take no dispenser: a wave's unit is its index,
    // and the host launches as many waves as units (round 4: the hardware's workgroup dispatcher is the load balancer; a grid of fewer
    // waves -- rounds 2-3, MIRT_STATIC_GRID -- deals the units round-robin).  One dispenser atomic per unit serialises on its address
    // (measured: 14 ns each, 0.47 ms for the 32 400 units of a 1080p frame whatever the work).
    const uint32_t grid_waves = gridDim.x * (kBlockThreads / 64u);
    // (Dispensing these units in batches -- a static first batch per wave, then guided batches of (units left) / (waves) per atomic -- was
    //  measured: 16 ... 110 % SLOWER than one unit per atomic at every unit size; eight dispenser words instead of one: 1 ... 9 % slower.)
    for (uint32_t strip = first_unit(); strip < A.n_units;
         strip = (BY_PIXEL && A.static_units) ? strip + grid_waves : next_unit_any(A, lane)) {
        if constexpr (BY_PIXEL) {
            // arguments needed once per unit are read from the kernarg segment here and now (per_strip_args) instead of
            // living in SGPRs across the sample loop, which the 8-waves-per-SIMD build has too few of
            const RenderArgs& AP = per_strip_args();
            // A unit is 64 >> g pixels x all samples, its samples dealt to 1 << g groups of lanes (AP.px_groups_log2, chosen on the
            // host): units of a quarter of the pixels are four times as many, so that the waves of a short launch -- config 2 lasts
            // 1.2 ms and had 4 units per wave, each a quarter of a wave's life -- finish within a small fraction of each other.  The sums
            // are exact integers: any split of a pixel's samples over lanes gives the same total.
            const uint32_t g = AP.px_groups_log2;
            const uint32_t unit_px = 64u >> g;
            const uint32_t grp = lane >> (6u - g);               // which share of the samples this lane takes
            const uint32_t pi = strip * unit_px + (lane & (unit_px - 1u));
            const bool inside = pi < AP.out_rows * AP.width;
            const uint32_t ci = (inside ? pi : 0u) / AP.width;
            const uint32_t x = (inside ? pi : 0u) - ci * AP.width;
            const uint32_t y = abs_row(AP, ci);
            const uint32_t n_mine = AP.spp >> g;                 // wave-uniform: the host deals shares only when they are equal (spp % (1 << g) == 0)
            const uint32_t s_first = AP.sample_begin + grp * n_mine;
            unsigned long long acc_r = 0, acc_g = 0, acc_b = 0;
            Rng rng;
            rng.state = 0;
            // grid builds: the spheres the camera rays of these pixels can touch (strip_candidates), if the unit lies in one image row
            uint32_t n_cand = kNoCand;
            const unsigned short* cand = nullptr;
            if constexpr (GRID) {
                unsigned short* my_cand = reinterpret_cast<unsigned short*>(smem + scene_lds_bytes_dev(A.n_spheres, A.n_mats, HOSEK, false) + A.grid_bytes) +
                                          (threadIdx.x >> 6) * (kCandBytes / 2u);
                const uint32_t left = AP.out_rows * AP.width - strip * unit_px;
                const uint32_t unit_x0 = __builtin_amdgcn_readfirstlane(x), unit_n = left < unit_px ? left : unit_px;
                const uint32_t unit_y = __builtin_amdgcn_readfirstlane(y);
                if (AP.strip_cand != 0u && unit_x0 + unit_n <= AP.width)
                    n_cand = strip_candidates(AP, S, unit_x0, unit_n, unit_y, my_cand, lane);
                n_cand = __builtin_amdgcn_readfirstlane(n_cand);
                cand = my_cand;
            }
            if constexpr (STREAM) {
                uint32_t j = 0;
                const uint32_t n_lane = (inside && A.num_bounces != 0u) ? n_mine : 0u;
                bool alive = false;
                f3 ro = mk(0, 0, 0), rd = mk(0, 0, 1), thr = mk(1, 1, 1);
                uint32_t bounce = 0;
                for (;;) {
                    const bool want = !alive && j < n_lane;
                    const unsigned long long wm = ballot_(want), am = ballot_(alive);
                    if ((wm | am) == 0ull) break;
                    if (am == 0ull || (uint32_t)__popcll(wm) >= kStreamRefillMin) {
                        if (want) {
                            const CamRegs C = load_camera(S, A);
                            const uint32_t sample = s_first + j;
                            if (A.frame_spp == 0u) {
                                generate_primary(A, C, x, y, sample, rng, ro, rd);
                            } else {
                                if (sample % A.frame_spp == 0u)
                                    rng.state = jenkins_hash(((x + y * A.width) ^ jenkins_hash(A.frame_begin + sample / A.frame_spp + 1u)) ^ A.seed_mix);
                                generate_primary<false>(A, C, x, y, sample, rng, ro, rd);
                            }
                            thr = mk(1, 1, 1);
                            bounce = 0;
                            alive = true;
                            ++j;
                        }
                    }
                    float closest;
                    const int best = nearest_hit<COUNT>(S, A.n_spheres, ro, rd, alive, closest, work);
                    if (alive) {
                        if (best >= 0) {
                            const PreparedSphere sp = S.spheres[best];
                            const f3 hp = fma3(closest, rd, ro);
                            const f3 hn = sp.inv_r * (hp - mk(sp.cx, sp.cy, sp.cz));
                            const PreparedMaterial* m = &S.pmats[sp.material_idx];
                            const Scattered sc = shade_by_id<COUNT>(A, m, m->id, true, rd, hp, hn, rng, work);
                            ro = hp;
                            rd = sc.dir;
                            thr = thr * sc.att;
                            ++bounce;
                            if (bounce >= A.num_bounces) alive = false;
                        } else {
                            const f3 c = thr * sky_color<HOSEK>(S, rd);
                            acc_r += to_fixed(c.x);
                            acc_g += to_fixed(c.y);
                            acc_b += to_fixed(c.z);
                            alive = false;
                        }
                    }
                }
            } else
            for (uint32_t j = 0; j < n_mine; ++j) {
                f3 ro, rd;
                {
                    const CamRegs C = load_camera(S, A);
                    const uint32_t sample = s_first + j;
                    if (A.frame_spp == 0u) {
                        generate_primary(A, C, x, y, sample, rng, ro, rd);
                    } else {                     // the reference's stream: frame = sample / n + 1 seeds once, its n samples share it (g == 0)
                        if (sample % A.frame_spp == 0u)
                            rng.state = jenkins_hash(((x + y * A.width) ^ jenkins_hash(A.frame_begin + sample / A.frame_spp + 1u)) ^ A.seed_mix);
                        generate_primary<false>(A, C, x, y, sample, rng, ro, rd);
                    }
                }
                const f3 c = path_radiance<COUNT, HOSEK, GRID>(A, S, G, inside, rng, ro, rd, work, lane, cand, n_cand);
                acc_r += to_fixed(c.x);
                acc_g += to_fixed(c.y);
                acc_b += to_fixed(c.z);
            }
            for (uint32_t off = unit_px; off < 64u; off <<= 1) {                 // add the shares: lanes l, l + unit_px, ... hold one pixel
                acc_r += __shfl_xor(acc_r, (int)off, 64);
                acc_g += __shfl_xor(acc_g, (int)off, 64);
                acc_b += __shfl_xor(acc_b, (int)off, 64);
            }
            if (inside && grp == 0u) {
                const RenderArgs& AS = per_strip_args();
                if (AS.accum) {                  // progressive mode: add the exact sums, resolve later
                    AS.accum[3ull * pi + 0] += acc_r; AS.accum[3ull * pi + 1] += acc_g; AS.accum[3ull * pi + 2] += acc_b;
                } else {
#ifdef MIRT_PROBE_NOSTORE      // experiment builds only
                    if (acc_r == 0x123456789abcull)
#endif
                    AS.out[pi] = pack_rgba(resolve_channel(acc_r, AS.spp, AS.flags), resolve_channel(acc_g, AS.spp, AS.flags),
                                           resolve_channel(acc_b, AS.spp, AS.flags));
                }
            }
        } else {
            const uint32_t base = strip * kStripPixels;
            uint32_t my_px = 0;
            for (uint32_t p = 0; p < kStripPixels; ++p) {
                const uint32_t pi = base + p;
                if (pi >= npix) break;
                const uint32_t ci = pi / A.width;
                const uint32_t x = pi - ci * A.width;
                const uint32_t y = abs_row(A, ci);

                unsigned long long acc_r = 0, acc_g = 0, acc_b = 0;
                for (uint32_t s0 = 0; s0 < A.spp; s0 += 64) {
                    const uint32_t s = s0 + lane;
                    Rng rng;
                    f3 ro, rd;
                    {   // camera constants are re-read from LDS per batch instead of living in 21 VGPRs
                        const CamRegs C = load_camera(S, A);
                        generate_primary(A, C, x, y, A.sample_begin + s, rng, ro, rd);
                    }
                    const f3 c = path_radiance<COUNT, HOSEK, GRID>(A, S, G, s < A.spp, rng, ro, rd, work, lane);
                    if (s < A.spp) {
                        acc_r += to_fixed(c.x);
                        acc_g += to_fixed(c.y);
                        acc_b += to_fixed(c.z);
                    }
                }
                acc_r = wave_sum_u64(acc_r);
                acc_g = wave_sum_u64(acc_g);
                acc_b = wave_sum_u64(acc_b);
                if (A.accum) {                   // progressive mode: add the exact sums, resolve later
                    if (lane == 0) { A.accum[3ull * pi + 0] += acc_r; A.accum[3ull * pi + 1] += acc_g; A.accum[3ull * pi + 2] += acc_b; }
                } else {
                    const uint32_t rgba = pack_rgba(resolve_channel(acc_r, A.spp, A.flags), resolve_channel(acc_g, A.spp, A.flags),
                                                    resolve_channel(acc_b, A.spp, A.flags));
                    if (lane == p) my_px = rgba;
                }
            }
            if (!A.accum && lane < kStripPixels && base + lane < npix) A.out[base + lane] = my_px;
        }
        work.flush(A.counters, lane);
    }
}

template <bool COUNT, bool HOSEK, bool GRID, bool BY_PIXEL = false>
__global__ __launch_bounds__(kBlockThreads, (BY_PIXEL && !COUNT && !GRID && !HOSEK) ? 8 : ((BY_PIXEL && !COUNT && GRID) ? 7 : 1)) void render_pt_strip_kernel(RenderArgs A)
{
    strip_kernel_body<COUNT, HOSEK, GRID, BY_PIXEL, false>(A);
}

template <bool HOSEK>
__global__ __launch_bounds__(kBlockThreads, HOSEK ? 6 : 8) void render_pt_stream_kernel(RenderArgs A)
{
    strip_kernel_body<false, HOSEK, false, true, true>(A);
}

// Shading routines a path can wait for: the five scatter routines of scatterRay (wgsl:174-314), identified by
// min(GpuMaterial.id, 4), and OP_GEN = finish the path (add throughput x sky) + start the next work item.
// A scene's routines are numbered densely on the host (queue q runs routine RenderArgs.queue_routine[q],
// PreparedSphere.op holds q), so a kernel built for NQ scatter queues has NQ + 1 queues: 0..NQ-1 scatter,
// NQ = OP_GEN.
constexpr uint32_t RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2, RT_CHECKER = 3;
constexpr uint32_t OP_NONE = 7, kMaxQueues = 6;
// pool kernel: a step whose paths all move on to one routine continues with it directly (no store / push / pop /
// gather) if it holds at least this many paths; 65 switches fast-forwarding off (experiment builds)
#ifndef MIRT_FF_MIN
#define MIRT_FF_MIN 64
#endif
constexpr uint32_t kFastForwardMin = MIRT_FF_MIN;

// ------------------------------------------------------------------------------------------
// render_pt_pool — wave-private path pool: every wave-instruction runs ONE shading routine
// ------------------------------------------------------------------------------------------
//
// The strip kernel loses half its lanes to divergence: a wave's 64 samples start on the same
// pixel (same material) but scatter to different materials or leave the scene, and the per-lane
// material switch runs every routine present.  Here a wave instead keeps SLOTS paths in LDS:
//   q0 = {hit point (or nothing after a miss), sphere | pixel << 12 | bounce << 16 | has_miss << 24}
//   q1 = {incoming direction rd, rng}   q2 = {throughput, -}                 (48 B, ds_*_b128)
// and every slot id sits in exactly one queue — that of the routine its path waits for.  Each step the wave
// pops up to 64 ids from its DEEPEST queue, gathers those paths, runs that one routine for all lanes, then
// the common tail (bounce limit, nearest hit, next queue) and pushes each id to the queue of its next
// routine.  The queues are stacks of 8-bit ids: the order of service cannot change the image (exact integer
// accumulation), so a queue is described by its depth alone.  A wave owns its pool outright: a strip of up to
// kStripPixels pixels = strip_pixels x spp work items (item = sample * strip_pixels + pixel), their 64-bit
// accumulators, the item counter and all queue depths in SGPRs.  Nothing is shared between waves after the
// scene has been staged: no barrier, no inter-wave atomic.  (A block-level pool with barriers was measured
// 20 % slower)
template <uint32_t SLOTS, uint32_t NQ, bool GRID = false, bool TILE = false>
struct WavePoolLayout {
    static constexpr uint32_t kQueues   = NQ + 1 + (GRID ? 1 : 0);            // scatter queues, OP_GEN, and OP_WALK in grid builds
    static constexpr uint32_t kRing     = SLOTS;                              // per-queue stack capacity: every slot could sit in one queue
    static constexpr uint32_t kOffState = 0;                                  // [SLOTS][3] uint4
    static constexpr uint32_t kOffAcc   = kOffState + SLOTS * 48;             // [kStripPixels][3] u64
    static constexpr uint32_t kOffCell  = kOffAcc + kStripPixels * 3 * 8;     // [SLOTS] u32: grid cell of a path whose walk is cut (GRID)
    static constexpr uint32_t kOffRing  = kOffCell + (GRID ? SLOTS * 2 : 0);  // ([SLOTS] u16 cells: the LINEAR index of a parked walk's cell)  [kQueues][kRing] u8
    static constexpr uint32_t kOffCand  = ((kOffRing + kQueues * kRing + 15) / 16) * 16;     // camera-ray candidates of the strip (GRID): ids + count
    static constexpr uint32_t kOffTile  = kOffCand + (GRID ? kCandBytes : 0);                // TexelTile: header + 3 colour planes (TILE)
    static constexpr uint32_t kBytes    = kOffTile + (TILE ? ((kTileBytes + 15) / 16) * 16 : 0);
};

// GRID = true (many-sphere scenes): nearest hit through the uniform grid staged behind the spheres; the material
// table stays in global memory / L2 as in the strip kernel's grid build, and the pools follow the grid in LDS.
// The kernel's text lives in mirt_pool_kernel.inc and is compiled twice: the default build and the tile build.
#define MIRT_POOL_KERNEL_NAME render_pt_pool_kernel
#define MIRT_POOL_KERNEL_TILE false
#include "mirt_pool_kernel.inc"
#undef MIRT_POOL_KERNEL_NAME
#undef MIRT_POOL_KERNEL_TILE
#define MIRT_POOL_KERNEL_NAME render_pt_pool_tile_kernel
#define MIRT_POOL_KERNEL_TILE true
#include "mirt_pool_kernel.inc"
#undef MIRT_POOL_KERNEL_NAME
#undef MIRT_POOL_KERNEL_TILE

#ifdef MIRT_ISA_PROBES
#include "mirt_isa_probes.inc"       // tools/isa_mix.py; never part of libmirt.so
#endif

#ifndef MIRT_FAST_MATH       // self-test, resolve and de-interleave live in the exact build only
// ------------------------------------------------------------------------------------------
// self-test: the fast sqrt_/rcp_ against the IEEE expansions over ALL 2^32 binary32 patterns
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void selftest_math_kernel(unsigned long long* mismatches)
{
    unsigned long long bad_sqrt = 0, bad_rcp = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = from_bits((uint32_t)i);
        bad_sqrt += bits(sqrt_(x)) != bits(sqrt_ieee(x));
        bad_rcp += bits(rcp_(x)) != bits(rcp_ieee(x));
        // the derived forms, on their domains: sqrt_unit on [0, 1]; inv_sqrt_2step == rcp(sqrt(x)) on [0, +inf]
        if (x >= 0.0f && x <= 1.0f) bad_sqrt += bits(sqrt_unit(x)) != bits(sqrt_ieee(x));
        if (x >= 0.0f) bad_rcp += bits(inv_sqrt_2step(x)) != bits(rcp_ieee(sqrt_ieee(x)));
    }
    bad_sqrt = wave_sum_u64(bad_sqrt);
    bad_rcp = wave_sum_u64(bad_rcp);
    if ((threadIdx.x & 63u) == 0) {
        if (bad_sqrt) atomicAdd(&mismatches[0], bad_sqrt);
        if (bad_rcp) atomicAdd(&mismatches[1], bad_rcp);
    }
}

hipError_t launch_selftest_math(unsigned long long* d_mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_math_kernel, dim3(4096), dim3(256), 0, stream, d_mismatches);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// resolve of the progressive accumulation buffer: exact sums -> mean -> tonemap -> RGBA8
// (the `invN * pixel` + uncharted2 tail of fsMain, wgsl:75-80)
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void resolve_accum_kernel(const unsigned long long* accum, uint32_t* out, uint64_t n_pixels,
                                                            uint32_t n_samples, uint32_t flags)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = pack_rgba(resolve_channel(accum[3 * i + 0], n_samples, flags), resolve_channel(accum[3 * i + 1], n_samples, flags),
                           resolve_channel(accum[3 * i + 2], n_samples, flags));
}

hipError_t launch_resolve(const unsigned long long* accum, uint32_t* out, uint64_t n_pixels, uint32_t n_samples,
                          uint32_t flags, hipStream_t stream)
{
    uint32_t blocks = (uint32_t)((n_pixels + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(resolve_accum_kernel, dim3(blocks), dim3(256), 0, stream, accum, out, n_pixels, n_samples, flags);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// de-interleave: root side of the multi-GPU gather (tile-interleaved parts -> band image)
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void deinterleave_kernel(DeinterleaveArgs D)
{
    const uint64_t total = (uint64_t)D.band_rows * D.width;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / D.width);
        const uint32_t x = (uint32_t)(i - (uint64_t)row * D.width);
        const uint32_t tile = row / D.tile_rows;
        const uint32_t part = tile % D.n_parts;
        const uint32_t local_row = (tile / D.n_parts) * D.tile_rows + row % D.tile_rows;
        D.out[i] = D.parts[(uint64_t)part * D.part_stride_px + (uint64_t)local_row * D.width + x];
    }
}

#endif  // !MIRT_FAST_MATH

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

size_t scene_lds_bytes(uint32_t n_spheres, uint32_t n_mats, bool pt, bool hosek)
{
    return sizeof(MirtGpuCamera) + (size_t)n_spheres * sizeof(PreparedSphere) +
           (size_t)n_mats * (pt ? sizeof(PreparedMaterial) : sizeof(MirtMaterial)) + (hosek ? sizeof(MirtSkyState) : 0);
}

size_t scene_lds_bytes_grid(uint32_t n_spheres, bool hosek)      // GRID build: camera (+ sky) only; the grid blob holds the rest
{
    (void)n_spheres;
    return sizeof(MirtGpuCamera) + (hosek ? sizeof(MirtSkyState) : 0);
}


template <typename K>
static hipError_t launch_with_lds(K kernel, dim3 g, dim3 b, const RenderArgs& a, LaunchOn stream)
{
    if (a.lds_bytes > 48u * 1024u) {        // more than the default dynamic-LDS window: ask for it
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (stream.end != nullptr) {                   // the launch's event pair rides on the dispatch
        hipExtLaunchKernelGGL(kernel, g, b, a.lds_bytes, stream.stream, stream.begin, stream.end, 0u, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kernel, g, b, a.lds_bytes, stream.stream, a);
    return hipGetLastError();
}

#ifndef MIRT_FAST_MATH
using ParityKernel = void (*)(RenderArgs);
static ParityKernel parity_kernel(bool count, bool by_pixel)
{
    if (by_pixel) return count ? render_parity_kernel<true, true> : render_parity_kernel<false, true>;
    return count ? render_parity_kernel<true, false> : render_parity_kernel<false, false>;
}

hipError_t launch_parity(const RenderArgs& a, uint32_t grid_blocks, bool count, bool by_pixel, LaunchOn stream)
{
    return launch_with_lds(parity_kernel(count, by_pixel), dim3(grid_blocks), dim3(a.launch_threads ? a.launch_threads : kBlockThreads), a, stream);
}

#endif

using StripKernel = void (*)(RenderArgs);

// the build of the strip kernel a launch runs (nullptr: no such build -- counting launches of the fast-math library)
static StripKernel strip_kernel(bool count, bool hosek, bool use_grid, bool by_pixel, bool stream = false)
{
    if (stream && by_pixel && !count && !use_grid) return hosek ? render_pt_stream_kernel<true> : render_pt_stream_kernel<false>;
#ifdef MIRT_FAST_MATH
    if (count) return nullptr;                           // the counting builds exist in the exact build only
#else
    if (count && use_grid) return hosek ? render_pt_strip_kernel<true, true, true> : render_pt_strip_kernel<true, false, true>;
    if (count && by_pixel) return hosek ? render_pt_strip_kernel<true, true, false, true> : render_pt_strip_kernel<true, false, false, true>;
    if (count) return hosek ? render_pt_strip_kernel<true, true, false> : render_pt_strip_kernel<true, false, false>;
#endif
    if (by_pixel) {
        if (use_grid) return hosek ? render_pt_strip_kernel<false, true, true, true> : render_pt_strip_kernel<false, false, true, true>;
        return hosek ? render_pt_strip_kernel<false, true, false, true> : render_pt_strip_kernel<false, false, false, true>;
    }
    if (use_grid) return hosek ? render_pt_strip_kernel<false, true, true> : render_pt_strip_kernel<false, false, true>;
    return hosek ? render_pt_strip_kernel<false, true, false> : render_pt_strip_kernel<false, false, false>;
}

hipError_t launch_pt_strip(const RenderArgs& a, uint32_t grid_blocks, bool count, bool use_grid, bool by_pixel, LaunchOn stream)
{
    const StripKernel k = strip_kernel(count, (a.flags & MIRT_FLAG_SKY_HOSEK) != 0, use_grid, by_pixel, a.stream_samples != 0u);
    if (!k) return hipErrorInvalidValue;
    return launch_with_lds(k, dim3(grid_blocks), dim3(a.launch_threads ? a.launch_threads : kBlockThreads), a, stream);
}

// Blocks of a kernel that are resident per CU at once (registers and LDS decide: the strip kernels use 59-106 VGPRs).  The
// host sizes the persistent grids with it: a block that is launched but not resident holds its first work unit until some
// other block exits, i.e. until the dispenser has run dry, and then runs that unit alone at the end of the launch.
static uint32_t blocks_per_cu(const void* kernel, uint32_t threads, uint32_t lds_bytes)
{
    // asked once per (kernel, LDS size): the interactive loop launches every frame
    struct Entry { const void* kernel; uint32_t lds, blocks; };
    static Entry cache[64];
    static uint32_t used = 0;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    for (uint32_t i = 0; i < used; ++i)
        if (cache[i].kernel == kernel && cache[i].lds == lds_bytes) return cache[i].blocks;
    uint32_t blocks = 1u;
    int n = 0;
    if ((lds_bytes <= 48u * 1024u ||
         hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) == hipSuccess) &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, (int)threads, lds_bytes) == hipSuccess && n >= 1)
        blocks = (uint32_t)n;
    if (used < 64u) cache[used++] = Entry{ kernel, lds_bytes, blocks };
    return blocks;
}

uint32_t strip_blocks_per_cu(bool hosek, bool count, bool use_grid, bool by_pixel, uint32_t lds_bytes, bool stream)
{
    const StripKernel k = strip_kernel(count, hosek, use_grid, by_pixel, stream);
    return k ? blocks_per_cu(reinterpret_cast<const void*>(k), kBlockThreads, lds_bytes) : 1u;
}

#ifndef MIRT_FAST_MATH
uint32_t parity_blocks_per_cu(bool count, bool by_pixel, uint32_t lds_bytes)
{
    return blocks_per_cu(reinterpret_cast<const void*>(parity_kernel(count, by_pixel)), kBlockThreads, lds_bytes);
}
#endif

// builds of the pool kernel by scatter queues: 3 and 5 for every geometry; 1, 2 and 4 as well (FEW = true) for the default and the
// tile geometry -- a scene with fewer shading routines than queues would carry empty queues through every pick and push
// (two routines, config 4: -1.2 %)
template <uint32_t T, uint32_t SL, uint32_t MW = 1, bool FEW = false>
static hipError_t launch_pool_cfg(const RenderArgs& a, uint32_t grid_blocks, bool count, bool hosek, uint32_t nq, LaunchOn stream)
{
    const dim3 g(grid_blocks), b(T);
#ifdef MIRT_FAST_MATH
    if (count) return hipErrorInvalidValue;
#else
    if (count) return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, 1, true, true>, g, b, a, stream)
                            : launch_with_lds(render_pt_pool_kernel<T, SL, 1, true, false>, g, b, a, stream);
#endif
    if constexpr (FEW) {
        if (nq == 1) return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, true, 1>, g, b, a, stream)
                                  : launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, false, 1>, g, b, a, stream);
        if (nq == 2) return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, true, 2>, g, b, a, stream)
                                  : launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, false, 2>, g, b, a, stream);
        if (nq == 4) return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, true, 4>, g, b, a, stream)
                                  : launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, false, 4>, g, b, a, stream);
    }
    if (nq <= 3) return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, true, 3>, g, b, a, stream)
                              : launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, false, 3>, g, b, a, stream);
    return hosek ? launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, true>, g, b, a, stream)
                 : launch_with_lds(render_pt_pool_kernel<T, SL, MW, false, false>, g, b, a, stream);
}

// the tile build (MIRT_FLAG_TEXEL_TILES): fewer slots, a texel window per wave
template <uint32_t T, uint32_t SL, uint32_t MW>
static hipError_t launch_pool_tile(const RenderArgs& a, uint32_t grid_blocks, bool count, bool hosek, uint32_t nq, LaunchOn stream)
{
    const dim3 g(grid_blocks), b(T);
#ifdef MIRT_FAST_MATH
    if (count) return hipErrorInvalidValue;
#else
    if (count) return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, 1, true, true>, g, b, a, stream)
                            : launch_with_lds(render_pt_pool_tile_kernel<T, SL, 1, true, false>, g, b, a, stream);
#endif
    if (nq == 1) return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, true, 1>, g, b, a, stream)
                              : launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, false, 1>, g, b, a, stream);
    if (nq == 2) return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, true, 2>, g, b, a, stream)
                              : launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, false, 2>, g, b, a, stream);
    if (nq == 4) return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, true, 4>, g, b, a, stream)
                              : launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, false, 4>, g, b, a, stream);
    if (nq <= 3) return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, true, 3>, g, b, a, stream)
                              : launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, false, 3>, g, b, a, stream);
    return hosek ? launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, true>, g, b, a, stream)
                 : launch_with_lds(render_pt_pool_tile_kernel<T, SL, MW, false, false>, g, b, a, stream);
}

// grid build of the default pool geometry: LDS (scene + grid + pools) bounds it to a few blocks per CU, so the
// register budget is not the limit
template <bool FLATY>
static hipError_t launch_pool_grid_(const RenderArgs& a, uint32_t grid_blocks, bool count, bool hosek, LaunchOn stream)
{
    const dim3 g(grid_blocks), b(kGridPoolThreads);
    const uint32_t slots = a.grid_pool_slots;
#ifdef MIRT_FAST_MATH
    if (count) return hipErrorInvalidValue;
#else
    if (count) {                                // counting builds exist for the two largest geometries
        if (slots == kGridPoolSlotChoices[0])
            return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[0], 1, true, true, 1, true, FLATY>, g, b, a, stream)
                         : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[0], 1, true, false, 1, true, FLATY>, g, b, a, stream);
        if (slots == kGridPoolSlotChoices[1])
            return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[1], 1, true, true, 1, true, FLATY>, g, b, a, stream)
                         : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[1], 1, true, false, 1, true, FLATY>, g, b, a, stream);
        return hipErrorInvalidValue;
    }
#endif
    if (slots == kGridPoolSlotChoices[0])
        return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[0], kGridPoolMinWaves, false, true, 1, true, FLATY>, g, b, a, stream)
                     : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[0], kGridPoolMinWaves, false, false, 1, true, FLATY>, g, b, a, stream);
    if (slots == kGridPoolSlotChoices[1])
        return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[1], kGridPoolMinWaves, false, true, 1, true, FLATY>, g, b, a, stream)
                     : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[1], kGridPoolMinWaves, false, false, 1, true, FLATY>, g, b, a, stream);
    if (slots == kGridPoolSlotChoices[2])
        return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[2], kGridPoolMinWaves, false, true, 1, true, FLATY>, g, b, a, stream)
                     : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[2], kGridPoolMinWaves, false, false, 1, true, FLATY>, g, b, a, stream);
    return hosek ? launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[3], kGridPoolMinWaves, false, true, 1, true, FLATY>, g, b, a, stream)
                 : launch_with_lds(render_pt_pool_kernel<kGridPoolThreads, kGridPoolSlotChoices[3], kGridPoolMinWaves, false, false, 1, true, FLATY>, g, b, a, stream);
}

// grid builds have ONE scatter queue (per-lane material switch); FLATY = the grid is one cell high (RenderArgs.grid_flat_y)
static hipError_t launch_pool_grid(const RenderArgs& a, uint32_t grid_blocks, bool count, bool hosek, uint32_t nq, LaunchOn stream)
{
    (void)nq;
    return a.grid_flat_y ? launch_pool_grid_<true>(a, grid_blocks, count, hosek, stream) : launch_pool_grid_<false>(a, grid_blocks, count, hosek, stream);
}

// pool geometries {threads per block, slots per wave}; [0] is the default.  The kernel is held to 80 VGPRs
// (6 waves per SIMD = 24 per CU); 112 slots x 48 B + rings + accumulators = 6.4 KB per wave keeps all 24
// resident in a CU's 160 KB of LDS.  Measured on config 3: 128 slots (5 waves/SIMD)
// +2.5 % time; 64 slots -> 68 % lane use, 1.4x; 256 slots -> 8 waves per CU, 1.7x; 88 slots at 7
// waves per SIMD (72 VGPRs, spills) +10 %.
// [kTilePoolConfig] is the tile build: the default geometry + a texel window per wave in the spare LDS of its 6 blocks per CU.
static const PoolConfig kPoolConfigs[] = { { 256, 112, 0 }, { 256, 128, 0 }, { 256, 64, 0 }, { 256, 256, 0 }, { 256, 96, 0 }, { 256, kTilePoolSlots, 0 } };

uint32_t pool_config_count() { return (uint32_t)(sizeof(kPoolConfigs) / sizeof(kPoolConfigs[0])); }

template <uint32_t NQ>
static uint32_t pool_bytes_per_wave(uint32_t slots)
{
    return (slots == 64) ? WavePoolLayout<64, NQ>::kBytes : (slots == 96) ? WavePoolLayout<96, NQ>::kBytes
         : (slots == 112) ? WavePoolLayout<112, NQ>::kBytes : (slots == 128) ? WavePoolLayout<128, NQ>::kBytes : WavePoolLayout<256, NQ>::kBytes;
}

// the build that runs for a geometry: 1 / 2 queues exist for the default and the tile geometry only
static uint32_t built_queues(uint32_t cfg, uint32_t nq)
{
    const bool few = cfg == kDefaultPoolConfig || cfg == kTilePoolConfig;
    if (few && nq >= 1 && nq <= 4) return nq;
    return nq <= 3 ? 3u : 5u;
}

// nq = scatter queues the scene needs (pool_scatter_queues); the LDS layout is that of the build that runs
PoolConfig pool_config(uint32_t i, uint32_t nq)
{
    if (i >= pool_config_count()) i = 0;
    PoolConfig c = kPoolConfigs[i];
    const uint32_t q = built_queues(i, nq);
    const uint32_t per_wave = q == 1 ? pool_bytes_per_wave<1>(c.slots) : q == 2 ? pool_bytes_per_wave<2>(c.slots)
                            : q == 3 ? pool_bytes_per_wave<3>(c.slots) : q == 4 ? pool_bytes_per_wave<4>(c.slots) : pool_bytes_per_wave<5>(c.slots);
    c.lds_bytes = per_wave * (c.threads / 64);
    if (i == kTilePoolConfig) {
        const uint32_t tw = q == 1 ? WavePoolLayout<kTilePoolSlots, 1, false, true>::kBytes : q == 2 ? WavePoolLayout<kTilePoolSlots, 2, false, true>::kBytes
                          : q == 3 ? WavePoolLayout<kTilePoolSlots, 3, false, true>::kBytes : q == 4 ? WavePoolLayout<kTilePoolSlots, 4, false, true>::kBytes
                          : WavePoolLayout<kTilePoolSlots, 5, false, true>::kBytes;
        c.lds_bytes = tw * (c.threads / 64);
    }
    return c;
}

// the grid build's geometry: 512 threads, 112 slots, one more queue (OP_WALK) and a cell word per slot
PoolConfig pool_config_grid(size_t lds_for_pools)
{
    const uint32_t waves = kGridPoolThreads / 64;
    const uint32_t bytes[4] = { WavePoolLayout<kGridPoolSlotChoices[0], 1, true>::kBytes * waves, WavePoolLayout<kGridPoolSlotChoices[1], 1, true>::kBytes * waves,
                                WavePoolLayout<kGridPoolSlotChoices[2], 1, true>::kBytes * waves, WavePoolLayout<kGridPoolSlotChoices[3], 1, true>::kBytes * waves };
    for (int i = 0; i < 4; ++i)
        if (bytes[i] <= lds_for_pools) return PoolConfig{ kGridPoolThreads, kGridPoolSlotChoices[i], bytes[i] };
    return PoolConfig{ kGridPoolThreads, 0, 0 };
}

// scatter queues a scene needs: one per shading routine (builds exist for 1 ... 5); the counting build always has 5
uint32_t pool_scatter_queues(uint32_t n_routines, bool count)
{
    if (count || n_routines > 4) return 5u;
    return n_routines < 1 ? 1u : n_routines;
}

hipError_t launch_pt_pool(const RenderArgs& a, uint32_t grid_blocks, uint32_t cfg, bool count, uint32_t nq, LaunchOn stream)
{
    const bool hosek = (a.flags & MIRT_FLAG_SKY_HOSEK) != 0;
    if (a.grid) return launch_pool_grid(a, grid_blocks, count, hosek, nq, stream);     // host: default geometry only
    switch (cfg) {
    case 1:  return launch_pool_cfg<256, 128>(a, grid_blocks, count, hosek, nq, stream);
    case 2:  return launch_pool_cfg<256, 64>(a, grid_blocks, count, hosek, nq, stream);
    case 3:  return launch_pool_cfg<256, 256>(a, grid_blocks, count, hosek, nq, stream);
    case 4:  return launch_pool_cfg<256, 96, 7>(a, grid_blocks, count, hosek, nq, stream);     // 7 waves per SIMD: <= 72 VGPRs
    case kTilePoolConfig: return launch_pool_tile<256, kTilePoolSlots, 6>(a, grid_blocks, count, hosek, nq, stream);
    default: return launch_pool_cfg<256, 112, 6, true>(a, grid_blocks, count, hosek, nq, stream);   // 6 waves per SIMD: <= 80 VGPRs
    }
}

// mirrors the dispatch of launch_pt_pool / launch_pool_cfg / launch_pool_grid above
void pool_kernel_name(const RenderArgs& a, uint32_t cfg, bool count, uint32_t nq, char* out, size_t out_len)
{
    const bool hosek = (a.flags & MIRT_FLAG_SKY_HOSEK) != 0;
    const char* tf[2] = { "false", "true" };
    uint32_t slots = 112, minw = 6, threads = 256;
    if (a.grid) { threads = kGridPoolThreads; slots = a.grid_pool_slots; minw = count ? 1 : kGridPoolMinWaves; nq = 1; }
    else {
        switch (cfg) {
        case 1: slots = 128; minw = 1; break;
        case 2: slots = 64; minw = 1; break;
        case 3: slots = 256; minw = 1; break;
        case 4: slots = 96; minw = 7; break;
        case kTilePoolConfig: slots = kTilePoolSlots; break;
        default: break;
        }
        if (count) { minw = 1; nq = 5; } else nq = built_queues(cfg, nq);
        if (cfg == kTilePoolConfig) {
            snprintf(out, out_len, "render_pt_pool_tile_kernel<%u,%u,%u,%s,%s,%u,false>", threads, slots, minw, tf[count], tf[hosek], nq);
            return;
        }
    }
    snprintf(out, out_len, "render_pt_pool_kernel<%u,%u,%u,%s,%s,%u,%s,%s>", threads, slots, minw, tf[count], tf[hosek], nq, tf[a.grid != nullptr],
             tf[a.grid != nullptr && a.grid_flat_y != 0u]);
}

#ifndef MIRT_FAST_MATH
hipError_t launch_deinterleave(const DeinterleaveArgs& a, hipStream_t stream)
{
    const uint64_t total = (uint64_t)a.band_rows * a.width;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
#endif

}  // namespace MIRT_KNS
}  // namespace mirt
