// mirt_api.hip — the C ABI of include/mirt.h: host-side validation, scene residency, kernel
// launches and timing.  No CPU rendering path exists here: without a HIP device every render
// entry point fails with MIRT_ERR_NO_DEVICE.
//
// Host arithmetic that the reference also does on the host (GpuCamera::new, camera_orientation,
// Angle, RenderParams::validate) is restated here in f32 without contraction
// (-ffp-contract=off applies to host code too).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mirt.h"
#include "mirt_kernels.h"

namespace kx = mirt::exact_build;     // the kernels of the bit-exact build (default)
namespace kf = mirt::fast_build;      // MIRT_FLAG_FAST_MATH: the same kernels with hardware transcendentals

namespace {

thread_local char g_err[512] = "";

int fail(int status, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(MIRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- MirtParams row selection ----
bool rows_valid(const MirtParams* p, uint32_t* rb, uint32_t* re)
{
    const uint32_t b = p->row_begin, e = p->row_end == 0 ? p->height : p->row_end;
    if (b >= e || e > p->height) return false;
    if (p->tile_rows != 0 && p->n_parts > 1 && p->part >= p->n_parts) return false;
    *rb = b;
    *re = e;
    return true;
}

uint32_t out_rows(const MirtParams* p)
{
    uint32_t rb, re;
    if (!p || p->width == 0 || p->height == 0 || !rows_valid(p, &rb, &re)) return 0;
    const uint32_t band = re - rb;
    if (p->tile_rows == 0 || p->n_parts <= 1) return band;
    const uint32_t tr = p->tile_rows;
    const uint32_t tiles = (band + tr - 1) / tr;
    if (tiles <= p->part) return 0;
    const uint32_t owned = (tiles - p->part + p->n_parts - 1) / p->n_parts;
    uint32_t rows = owned * tr;
    if ((tiles - 1) % p->n_parts == p->part) rows -= tiles * tr - band;
    return rows;
}

bool desc_ok(const MirtTextureDescriptor& d, uint64_t n_texels)
{
    if (d.width == 0 || d.height == 0) return false;
    return (uint64_t)d.offset + (uint64_t)d.width * (uint64_t)d.height <= n_texels;
}

// jenkinsHash (raytracer.wgsl:513-521) — used to fold the 64-bit seed into the stream key
uint32_t jenkins_hash(uint32_t x)
{
    x += x << 10; x ^= x >> 6; x += x << 3; x ^= x >> 11; x += x << 15;
    return x;
}

constexpr int kEventPool = 64;
constexpr int kDispenserStride = MIRT_DISPENSER_STRIDE;             // u32 words between the eight dispenser words of a launch (mirt_kernels.h)
constexpr int kDispenserWords = 8 * kDispenserStride;

constexpr float kRustPi = 3.14159265358979323846f;   // std::f32::consts::PI

struct V3 { float x, y, z; };
// nalgebra 3-vector arithmetic (no fusion): dot = (a0*b0 + a1*b1) + a2*b2; normalize = v / sqrt(dot)
float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
V3 normalize3(V3 a) { const float n = std::sqrt(dot3(a, a)); return V3{ a.x / n, a.y / n, a.z / n }; }
V3 cross3(V3 a, V3 b) { return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
V3 scale3(float s, V3 a) { return V3{ s * a.x, s * a.y, s * a.z }; }
V3 add3(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
V3 sub3(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }

// Build the uniform grid of mirt_kernels.h over the spheres of a many-sphere scene.  Returns an empty
// blob when a grid would not help (few spheres, or nothing small enough to bin) -- or when THIS cell size lists more
// than 65 535 items (`*overflow` = true: a coarser cell may still work, mirt_ctx_set_scene retries).
std::vector<unsigned char> build_grid(const MirtSphere* sph, uint32_t n, double cell_factor_knob, double big_factor_knob, bool* overflow = nullptr)
{
    std::vector<unsigned char> blob;
    if (overflow) *overflow = false;
    if (n < mirt::kGridMinSpheres || n > 65535u) return blob;
    std::vector<float> radii(n);
    for (uint32_t i = 0; i < n; ++i) radii[i] = std::fabs(sph[i].radius);
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float r_med = sorted[n / 2];
    if (!(r_med > 0.0f) || !std::isfinite(r_med)) return blob;
    // spheres up to `big_factor` median radii go into the grid; the rest (ground planes) are tested for every ray
    const float big_factor = (float)(big_factor_knob > 0.0 ? big_factor_knob : 4.0);
    std::vector<uint16_t> big, small;
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (uint32_t i = 0; i < n; ++i) {
        const bool finite = std::isfinite(sph[i].center[0]) && std::isfinite(sph[i].center[1]) && std::isfinite(sph[i].center[2]) && std::isfinite(radii[i]);
        if (!finite || radii[i] > big_factor * r_med) { big.push_back((uint16_t)i); continue; }
        small.push_back((uint16_t)i);
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], (double)sph[i].center[k] - radii[i]);
            hi[k] = std::max(hi[k], (double)sph[i].center[k] + radii[i]);
        }
    }
    if (small.size() < mirt::kGridMinSpheres / 2 || big.size() > 64) return blob;
    // cell = 2.5 median radii (a binned sphere of up to 4 median radii spans at most 5 cells per axis).  Measured on RTIOW, 1080p x 128 spp,
    // records stored once per sphere: 4 -> 14.31 ms, 3.5 -> 13.95, 3 -> 13.35, 2.75 -> 13.38, 2.5 -> 13.18, 2.25 -> 13.11, 2 -> 19.7 (the cell
    // table outgrows LDS: 128-slot pools); fewer tests per ray against more cells per ray (profiles/r03_c5_ab.txt blocks 4 and 7)
    const double cell_factor = cell_factor_knob > 0.0 ? cell_factor_knob : 2.5;   // mirt_ctx_set_scene passes the factor it settles on
    double cell = cell_factor * r_med;
    uint32_t dims[3];
    for (;;) {
        uint64_t total = 1;
        for (int k = 0; k < 3; ++k) { dims[k] = (uint32_t)std::max(1.0, std::ceil((hi[k] - lo[k]) / cell + 1e-6)); total *= dims[k]; }
        if (total <= mirt::kGridMaxCells) break;
        cell *= 1.26;
    }
    const double eps = 1e-3 * cell;                               // enlargement that makes the binning conservative
    const uint32_t ncells = dims[0] * dims[1] * dims[2];
    std::vector<std::vector<uint16_t>> lists(ncells);
    for (uint16_t i : small) {
        int c0[3], c1[3];
        for (int k = 0; k < 3; ++k) {
            c0[k] = (int)std::floor(((double)sph[i].center[k] - radii[i] - eps - lo[k]) / cell);
            c1[k] = (int)std::floor(((double)sph[i].center[k] + radii[i] + eps - lo[k]) / cell);
            c0[k] = std::max(0, std::min((int)dims[k] - 1, c0[k]));
            c1[k] = std::max(0, std::min((int)dims[k] - 1, c1[k]));
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int x = c0[0]; x <= c1[0]; ++x) lists[((size_t)z * dims[1] + y) * dims[0] + x].push_back(i);   // ascending ids
    }
    size_t n_items = 0, n_entries = 0;
    for (auto& l : lists) {
        n_entries += l.size();
        n_items += l.size() > 2 ? l.size() - 2 : 0;             // the first two ids of a cell live in its table entry
    }
    if (n_entries > 65535u) { if (overflow) *overflow = true; return blob; }     // cell_start is 16 bit
    mirt::GridHeader h{};
    for (int k = 0; k < 3; ++k) {
        h.org[k] = (float)lo[k];
        h.cell[k] = (float)cell;
        h.inv_cell[k] = (float)(1.0 / cell);
        h.dims[k] = dims[k];
    }
    h.n_big = (uint32_t)big.size();
    const size_t hdr_u16 = sizeof(mirt::GridHeader) / 2;
    h.off_big = (uint32_t)hdr_u16;
    h.off_items = h.off_big + (uint32_t)big.size();
    size_t bytes = ((size_t)h.off_items + n_items) * 2;
    h.off_ops = (uint32_t)bytes;                                  // u8 per sphere, filled by the caller (needs the materials)
    bytes = (bytes + n + 7) & ~size_t(7);
    h.off_cells = (uint32_t)bytes;                                // two u32 per cell: first further item | count << 16, id0 | id1 << 16
    bytes = (bytes + 8 * (size_t)ncells + 15) & ~size_t(15);
    h.n_entries = (uint32_t)n_entries;
    h.inv_dim_x = 1.0f / (float)dims[0];
    h.inv_dim_xy = 1.0f / (float)(dims[0] * dims[1]);
    h.off_big_recs = (uint32_t)bytes;
    bytes += big.size() * 16;
    h.off_recs = (uint32_t)bytes;
    bytes += (size_t)n * 16;                                      // one record per SPHERE, indexed by sphere id
    h.total_bytes = (uint32_t)bytes;
    blob.assign(h.total_bytes, 0);
    uint16_t* u = reinterpret_cast<uint16_t*>(blob.data());
    uint32_t* cells = reinterpret_cast<uint32_t*>(blob.data() + h.off_cells);
    std::memcpy(blob.data(), &h, sizeof h);
    // test record of sphere i: centre and r*r, the first half of its PreparedSphere (same IEEE product as set_scene's)
    auto put_rec = [&](size_t byte_off, uint16_t i) {
        const float rec[4] = { sph[i].center[0], sph[i].center[1], sph[i].center[2], sph[i].radius * sph[i].radius };
        std::memcpy(blob.data() + byte_off, rec, 16);
    };
    for (size_t i = 0; i < big.size(); ++i) { u[h.off_big + i] = big[i]; put_rec(h.off_big_recs + 16 * i, big[i]); }
    uint32_t pos = 0;
    for (uint32_t c = 0; c < ncells; ++c) {
        const auto& l = lists[c];
        cells[2 * c] = pos | ((uint32_t)l.size() << 16);
        cells[2 * c + 1] = (l.size() > 0 ? (uint32_t)l[0] : 0u) | ((l.size() > 1 ? (uint32_t)l[1] : 0u) << 16);   // absent: sphere 0 (valid, never tested)
        for (size_t k = 2; k < l.size(); ++k) u[h.off_items + pos++] = l[k];
    }
    for (uint32_t i = 0; i < n; ++i) put_rec(h.off_recs + 16 * (size_t)i, (uint16_t)i);
    return blob;
}

// The grid mirt_ctx_set_scene keeps: the default cell of 2.5 median radii, coarsened (x 1.26 per step, up to the 4 median radii of
// round 2 and no further) while the blob would not leave the pooled kernel the 152-slot geometry RTIOW runs (kGridPoolSlotChoices:
// 160 only fits beside smaller blobs and measured no better) -- or while the cell is so fine that the lists overflow their 16-bit
// index (bimodal radii: a sphere of 4 median radii spans 5 cells per axis at 2.5, 3 at 4): such a scene built a grid at round 2's
// cell of 4 and must not lose it to the finer default.  `cell_knob` > 0 (MIRT_GRID_CELL) forces one cell size.
std::vector<unsigned char> plan_grid(const MirtSphere* sph, uint32_t n, double cell_knob, double big_knob, size_t lds_per_block, double* factor_out)
{
    std::vector<unsigned char> grid;
    double f = cell_knob > 0.0 ? cell_knob : 2.5;
    for (;; f = std::min(f * 1.26, 4.0)) {
        bool overflow = false;
        grid = build_grid(sph, n, f, big_knob, &overflow);
        if (cell_knob > 0.0 || f >= 4.0) break;                                // a forced cell, or the coarsest one: take what it gives
        if (grid.empty()) { if (overflow) continue; break; }                   // too many entries: coarser; a grid would not help: none
        const size_t beside = kx::scene_lds_bytes_grid(n, true) + grid.size();
        if (beside < lds_per_block && kx::pool_config_grid(lds_per_block - beside).slots >= 152u) break;
    }
    if (factor_out) *factor_out = grid.empty() ? 0.0 : f;
    return grid;
}

// Tuning / experiment knobs.  They are read from the environment ONCE, when a context is created
// (mirt_ctx_create), never inside a render call: a shipped library must not change its schedule
// because a variable appeared mid-run, and getenv has no place in a sub-millisecond launch path.
struct Tuning {
    int      pool_config = -1;        // MIRT_POOL_CONFIG: geometry of the path pool (index into kPoolConfigs)
    int      pool_grid = -1;          // MIRT_POOL_GRID=0: never take the pool kernel's grid build
    bool     fixed_strips = false;    // MIRT_STRIP_MODE=16: fixed 16-pixel strips (no guided self-scheduling)
    uint32_t gss_min_width = 0;       // MIRT_GSS_MINW: narrowest strip of the guided schedule
    double   gss_keep = 0.0;          // MIRT_GSS_KEEP: strips per resident wave kept for the narrower levels
    int      by_pixel = -1;           // MIRT_BY_PIXEL=0/1: force lane = sample / lane = pixel in the strip kernel
    uint32_t pool_blocks_per_cu = 0;  // MIRT_POOL_BLOCKS_PER_CU: fewer resident pool blocks (occupancy experiments)
    double   grid_cell = 0.0;         // MIRT_GRID_CELL: cell size of the uniform grid in median radii
    double   grid_big = 0.0;          // MIRT_GRID_BIG: spheres above this many median radii stay outside the grid
    int      pinhole = -1;            // MIRT_PINHOLE=0: never take the pinhole-camera shortcut (A/B runs)
    int      spread_units = -1;       // MIRT_SPREAD_UNITS=0: strip-type launches use one dispenser word (A/B runs)
    int      static_units = -1;       // MIRT_STATIC_UNITS=0/1: lane-per-pixel units dispensed to a persistent grid / one unit per wave (A/B runs, tests)
    int      pool_min_spp = -1;       // MIRT_POOL_MIN_SPP=n: flat scenes take the pooled kernel from n samples per pixel on (A/B runs: the crossovers kPoolMinSpp*)
    int      stream = -1;             // MIRT_STREAM=0/1: lane-per-pixel launches in flat scenes never / always run the streaming build (A/B runs)
    int      px_groups = -1;          // MIRT_PX_GROUPS=0: lane-per-pixel units are always 64 pixels; 1 / 2 / 3: force 1 / 2 / 4 sample groups (A/B runs)
    int      strip_cand = -1;         // MIRT_STRIP_CAND=0: camera rays of grid builds take the grid like every other ray (A/B runs)
    int      grid_flat_y = -1;        // MIRT_GRID_FLAT_Y=0: a grid one cell high is walked in three dimensions like any other (A/B runs, tests)
    bool     debug_slots = false;     // MIRT_DEBUG_SLOTS=1: every launch checks (synchronously) that its slot's dispenser words are zero
    uint32_t static_block = 0;        // MIRT_STATIC_BLOCK=64/128/256 (A/B runs): threads per block of the launches that run one unit per wave
    int      static_grid = -1;        // MIRT_STATIC_GRID=k (A/B runs): launches with one unit per wave run k x the resident blocks instead, units dealt round-robin
    int      timing = -1;             // MIRT_TIMING=0/1: the context's initial mirt_ctx_set_timing state (A/B runs; default 1)
    bool     ext_events = true;       // MIRT_EXT_EVENTS=0: the event pair as two records in the stream instead of riding on the kernel dispatch (A/B runs)
};

Tuning read_tuning()
{
    Tuning t;
    if (const char* e = std::getenv("MIRT_POOL_CONFIG")) { const long v = std::strtol(e, nullptr, 10); if (v >= 0 && (uint32_t)v < kx::pool_config_count()) t.pool_config = (int)v; }
    if (const char* e = std::getenv("MIRT_POOL_GRID")) t.pool_grid = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_STRIP_MODE")) t.fixed_strips = e[0] == '1';
    if (const char* e = std::getenv("MIRT_GSS_MINW")) { const uint32_t v = (uint32_t)std::atoi(e); if (v >= 1 && v <= 16) t.gss_min_width = v; }
    if (const char* e = std::getenv("MIRT_GSS_KEEP")) { const double v = std::atof(e); if (v > 0.0 && v < 64.0) t.gss_keep = v; }
    if (const char* e = std::getenv("MIRT_BY_PIXEL")) t.by_pixel = (e[0] == '1') ? 1 : 0;
    if (const char* e = std::getenv("MIRT_POOL_BLOCKS_PER_CU")) { const uint32_t v = (uint32_t)std::atoi(e); if (v >= 1) t.pool_blocks_per_cu = v; }
    if (const char* e = std::getenv("MIRT_GRID_CELL")) { const double v = std::atof(e); if (v >= 1.0 && v <= 64.0) t.grid_cell = v; }
    if (const char* e = std::getenv("MIRT_PINHOLE")) t.pinhole = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_SPREAD_UNITS")) t.spread_units = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_STATIC_UNITS")) t.static_units = (e[0] == '1') ? 1 : 0;
    if (const char* e = std::getenv("MIRT_POOL_MIN_SPP")) { const int v = std::atoi(e); if (v >= 1) t.pool_min_spp = v; }
    if (const char* e = std::getenv("MIRT_STREAM")) t.stream = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("MIRT_PX_GROUPS")) { const int v = std::atoi(e); if (v >= 0 && v <= 3) t.px_groups = v; }
    if (const char* e = std::getenv("MIRT_STRIP_CAND")) t.strip_cand = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_GRID_FLAT_Y")) t.grid_flat_y = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_DEBUG_SLOTS")) t.debug_slots = e[0] == '1';
    if (const char* e = std::getenv("MIRT_STATIC_BLOCK")) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) t.static_block = (uint32_t)v; }
    if (const char* e = std::getenv("MIRT_STATIC_GRID")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) t.static_grid = v; }
    if (const char* e = std::getenv("MIRT_TIMING")) t.timing = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("MIRT_EXT_EVENTS")) t.ext_events = e[0] != '0';
    if (const char* e = std::getenv("MIRT_GRID_BIG")) { const double v = std::atof(e); if (v >= 1.0 && v <= 1024.0) t.grid_big = v; }
    return t;
}

// A camera whose thin-lens offset is exactly zero for every lens draw, so that `origin = eye + lens_radius * (...)`
// (wgsl:468-472) is `eye` bit for bit: lens_radius == +-0 (aperture 0: GpuCamera::new mod.rs:705), a finite lens basis and a
// finite eye without a -0 component (see generate_primary in mirt_kernels.hip for the argument).
bool camera_is_pinhole(const MirtGpuCamera& cam)
{
    if (cam.lens_radius != 0.0f) return false;
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(cam.u[k]) || !std::isfinite(cam.v[k]) || !std::isfinite(cam.eye[k])) return false;
        if (cam.eye[k] == 0.0f && std::signbit(cam.eye[k])) return false;
    }
    return true;
}

template <typename T>
int ensure_capacity(T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap && *ptr) return MIRT_OK;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    const size_t bytes = (need ? need : 1) * sizeof(T);
    if (hipMalloc(ptr, bytes) != hipSuccess) return fail(MIRT_ERR_ALLOC, "hipMalloc(%zu bytes) failed", bytes);
    *cap = need ? need : 1;
    return MIRT_OK;
}

}  // namespace

struct MirtContext {
    int         device = -1;
    int         cu_count = 0;
    size_t      lds_per_block = 65536, lds_per_cu = 65536;
    hipStream_t stream = nullptr;
    // hipEvent pairs around every render kernel, recorded on the stream the kernel runs on;
    // drained (summed) by mirt_ctx_get_stats so that no host sync sits inside a timed loop.
    // The slots form a RING: a launch takes the next slot; the slots half a ring ahead are retired in batches of 16 (launches 17 to 32
    // launches old, normally long finished: waits that return at once) -- no drain of the whole pool, nothing that stalls a caller who
    // queues launches back to back (the reference's interactive loop: one set_data / render_frame per displayed frame).
    std::vector<hipEvent_t> ev_begin, ev_end;
    std::vector<hipEvent_t> ev_zeroed;            // per slot: its dispenser words are zero again (recorded on zero_stream)
    std::vector<unsigned char> slot_busy;         // a launch is recorded in the slot and not yet folded into ms_folded
    std::vector<unsigned char> slot_dirty;        // the slot's launch took units from its dispenser words: they must be re-zeroed before the next use
    std::vector<unsigned char> slot_zeroing;      // a re-zeroing of the slot's dispenser words is queued (ev_zeroed says when it is done)
    std::vector<unsigned char> slot_zero_event;   // ... and which ev_zeroed: the slot's own, or its batch's first slot (retire_slots)
    std::vector<unsigned char> slot_timed;        // the slot's launch carries a start AND an end event (kernel time is folded into the statistics)
    bool        timing = true;                    // mirt_ctx_set_timing: launches carry an event pair (mirt_ctx_get_stats reports kernel times)
    std::vector<hipStream_t> untimed_streams;     // caller streams that carried launches without an end event (mirt_ctx_synchronize covers them)
    size_t      ev_next = 0;          // slot of the next launch
    size_t      ev_in_flight = 0;     // busy slots: the ring positions ev_next - ev_in_flight .. ev_next - 1
    hipStream_t zero_stream = nullptr;            // re-zeroes the dispenser words of retired slots, off the callers' streams
    hipStream_t frame_stream_b = nullptr;         // mirt_ctx_frame_stream(ctx, 1): a second stream on another hardware queue (created on first use)
    double      ms_folded = 0.0;      // time of the launches already retired
    uint64_t    launches_folded = 0;
    double      last_ms = 0.0;

    // resident scene
    bool     have_scene = false;
    uint32_t n_spheres = 0, n_mats = 0;
    uint64_t n_texels = 0;
    bool     have_sky = false;
    float    sph3[12] = {};               // {centre, r^2} of the spheres of a three-sphere scene (RenderArgs.sph3)
    bool     has_image_texture = false;   // a material refers to a texture larger than 1x1
    uint32_t n_shading_routines = 0;      // distinct scatter routines the spheres' materials select
    uint32_t queue_routine[5] = {0, 1, 2, 3, 4};   // dense numbering of the routines present (pool kernel queues)
    int      pt_scene_status = MIRT_OK;
    int      parity_scene_status = MIRT_OK;
    MirtGpuCamera         cam{};          // host copy; travels by value with every launch (RenderArgs.cam)
    mirt::PreparedSphere* d_spheres = nullptr;
    MirtMaterial*         d_mats = nullptr;
    mirt::PreparedMaterial* d_pmats = nullptr;
    size_t cap_pmats = 0;
    unsigned char* d_grid = nullptr;        // uniform grid blob (many-sphere scenes), see build_grid
    size_t cap_grid = 0;
    mirt::ShadeRec* d_shade = nullptr;      // [n_spheres] shading records of the grid builds (sphere + its material in one line)
    size_t cap_shade = 0;
    uint32_t grid_bytes = 0;
    bool     have_shade = false;             // d_shade holds this scene's records
    bool     grid_packable = false;          // the grid has at most kGridMaxCells cells: a parked walk's linear cell index fits 16 bits (always, as built)
    bool     grid_flat_y = false;            // the grid is ONE cell high (dims[1] == 1: spheres on a ground plane): the pooled kernel walks it in two dimensions
    bool     fits_flat = true;               // spheres + materials fit the LDS budget (flat kernels usable)
    float*                d_texels = nullptr;
    MirtSkyState*         d_sky = nullptr;
    size_t cap_spheres = 0, cap_mats = 0, cap_texels = 0;

    Tuning tuning;

    // per-launch state: every launch owns the dispenser word and the counter block of its event slot, so
    // launches of one context that overlap on different streams cannot disturb each other
    unsigned long long* d_counters = nullptr;      // [kEventPool][kNumCounters]
    uint32_t*           d_work_counter = nullptr;  // [kEventPool][kDispenserWords]: word 0 = the launch's dispenser; words kDispenserStride k = the 8 dispensers of a lane-per-pixel launch
    size_t              last_slot = 0;             // event slot of the last launch (its counters feed MirtStats)
    hipEvent_t          ev_accum = nullptr;        // end of the last launch that adds into d_accum
    bool                accum_pending = false;
    uint32_t*           d_out = nullptr;     // scratch framebuffer for host-output renders
    size_t              cap_out = 0;
    unsigned long long* d_accum = nullptr;   // progressive accumulation: [pixels][3] exact sums
    size_t              cap_accum = 0;
    uint64_t            accum_pixels = 0;
    uint32_t            accum_samples = 0;
    uint32_t            accum_width = 0, accum_rows = 0;

    MirtStats stats{};
    bool      stats_counted = false;
    char      last_kernel[128] = "";
};

extern "C" {

uint32_t mirt_version(void) { return (MIRT_VERSION_MAJOR << 16) | (MIRT_VERSION_MINOR << 8) | MIRT_VERSION_PATCH; }

const char* mirt_last_error(void) { return g_err; }

const char* mirt_status_string(int status)
{
    switch (status) {
    case MIRT_OK: return "MIRT_OK";
    case MIRT_ERR_MAX_SAMPLES_MULTIPLE: return "MIRT_ERR_MAX_SAMPLES_MULTIPLE";
    case MIRT_ERR_VIEWPORT_SIZE: return "MIRT_ERR_VIEWPORT_SIZE";
    case MIRT_ERR_VFOV_RANGE: return "MIRT_ERR_VFOV_RANGE";
    case MIRT_ERR_APERTURE_RANGE: return "MIRT_ERR_APERTURE_RANGE";
    case MIRT_ERR_FOCUS_DISTANCE: return "MIRT_ERR_FOCUS_DISTANCE";
    case MIRT_ERR_SKY: return "MIRT_ERR_SKY";
    case MIRT_ERR_NULL_POINTER: return "MIRT_ERR_NULL_POINTER";
    case MIRT_ERR_SPP_ZERO: return "MIRT_ERR_SPP_ZERO";
    case MIRT_ERR_BAD_MODE: return "MIRT_ERR_BAD_MODE";
    case MIRT_ERR_BAD_ROWS: return "MIRT_ERR_BAD_ROWS";
    case MIRT_ERR_MATERIAL_INDEX: return "MIRT_ERR_MATERIAL_INDEX";
    case MIRT_ERR_TEXEL_RANGE: return "MIRT_ERR_TEXEL_RANGE";
    case MIRT_ERR_OUT_BUFFER: return "MIRT_ERR_OUT_BUFFER";
    case MIRT_ERR_NO_SCENE: return "MIRT_ERR_NO_SCENE";
    case MIRT_ERR_SCENE_TOO_LARGE: return "MIRT_ERR_SCENE_TOO_LARGE";
    case MIRT_ERR_FRAME_SPP: return "MIRT_ERR_FRAME_SPP";
    case MIRT_ERR_NO_DEVICE: return "MIRT_ERR_NO_DEVICE";
    case MIRT_ERR_HIP: return "MIRT_ERR_HIP";
    case MIRT_ERR_ALLOC: return "MIRT_ERR_ALLOC";
    case MIRT_ERR_IMAGE_DECODE: return "MIRT_ERR_IMAGE_DECODE";
    case MIRT_ERR_SPP_RANGE: return "MIRT_ERR_SPP_RANGE";
    default: return "MIRT_ERR_UNKNOWN";
    }
}

// Angle::degrees / as_degrees (reference src/raytracer/angle.rs:8-22)
float mirt_degrees_to_radians(float degrees) { return degrees * kRustPi / 180.0f; }
float mirt_radians_to_degrees(float radians) { return radians * 180.0f / kRustPi; }

// RenderParams::validate (reference src/raytracer/mod.rs:450-484), same order of checks
int mirt_validate_render_params(const MirtCamera* camera, const MirtSamplingParams* sampling, uint32_t w, uint32_t h)
{
    if (!camera || !sampling) return fail(MIRT_ERR_NULL_POINTER, "camera/sampling is null");
    if (sampling->num_samples_per_pixel == 0)
        return fail(MIRT_ERR_SPP_ZERO, "num_samples_per_pixel is zero");
    if (sampling->max_samples_per_pixel % sampling->num_samples_per_pixel != 0)
        return fail(MIRT_ERR_MAX_SAMPLES_MULTIPLE, "max_samples_per_pixel (%u) is not a multiple of num_samples_per_pixel (%u)",
                    sampling->max_samples_per_pixel, sampling->num_samples_per_pixel);
    if (w == 0 || h == 0) return fail(MIRT_ERR_VIEWPORT_SIZE, "viewport_size elements cannot be zero: (%u, %u)", w, h);
    const float lo = mirt_degrees_to_radians(0.0f), hi = mirt_degrees_to_radians(90.0f);
    if (!(camera->vfov_radians >= lo && camera->vfov_radians <= hi))
        return fail(MIRT_ERR_VFOV_RANGE, "vfov must be between 0..=90 degrees");
    if (!(camera->aperture >= 0.0f && camera->aperture <= 1.0f))
        return fail(MIRT_ERR_APERTURE_RANGE, "aperture must be between 0..=1");
    if (camera->focus_distance < 0.0f)
        return fail(MIRT_ERR_FOCUS_DISTANCE, "focus_distance must be greater than zero");
    return MIRT_OK;
}

// GpuCamera::new (reference src/raytracer/mod.rs:700-741)
int mirt_camera_new(const MirtCamera* camera, uint32_t vw, uint32_t vh, MirtGpuCamera* out)
{
    if (!camera || !out) return fail(MIRT_ERR_NULL_POINTER, "camera/out is null");
    if (vw == 0 || vh == 0) return fail(MIRT_ERR_VIEWPORT_SIZE, "viewport_size elements cannot be zero: (%u, %u)", vw, vh);
    const float lens_radius = 0.5f * camera->aperture;
    const float aspect = (float)vw / (float)vh;
    const float half_height = camera->focus_distance * std::tan(0.5f * camera->vfov_radians);
    const float half_width = aspect * half_height;
    const V3 eye{ camera->eye_pos[0], camera->eye_pos[1], camera->eye_pos[2] };
    const V3 w = normalize3(V3{ camera->eye_dir[0], camera->eye_dir[1], camera->eye_dir[2] });
    const V3 v = normalize3(V3{ camera->up[0], camera->up[1], camera->up[2] });
    const V3 u = cross3(w, v);
    const V3 llc = sub3(sub3(add3(eye, scale3(camera->focus_distance, w)), scale3(half_width, u)), scale3(half_height, v));
    const V3 horizontal = scale3(2.0f * half_width, u);
    const V3 vertical = scale3(2.0f * half_height, v);
    std::memset(out, 0, sizeof *out);
    out->eye[0] = eye.x; out->eye[1] = eye.y; out->eye[2] = eye.z;
    out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    out->lens_radius = lens_radius;
    out->lower_left_corner[0] = llc.x; out->lower_left_corner[1] = llc.y; out->lower_left_corner[2] = llc.z;
    return MIRT_OK;
}

// camera_orientation + FlyCameraController::renderer_camera (reference src/fly_camera.rs:52-64, 227-241)
int mirt_camera_from_fly_pose(const float position[3], float yaw, float pitch, float vfov_degrees, float aperture,
                              float focus_distance, MirtCamera* out)
{
    if (!position || !out) return fail(MIRT_ERR_NULL_POINTER, "position/out is null");
    const V3 forward = normalize3(V3{ std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch) });
    const V3 right = cross3(forward, V3{ 0.0f, 1.0f, 0.0f });
    const V3 up = cross3(right, forward);
    out->eye_pos[0] = position[0]; out->eye_pos[1] = position[1]; out->eye_pos[2] = position[2];
    out->eye_dir[0] = forward.x; out->eye_dir[1] = forward.y; out->eye_dir[2] = forward.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
    out->vfov_radians = mirt_degrees_to_radians(vfov_degrees);
    out->aperture = aperture;
    out->focus_distance = focus_distance;
    return MIRT_OK;
}

uint32_t mirt_params_out_rows(const MirtParams* params) { return out_rows(params); }

uint32_t mirt_params_out_row_index(const MirtParams* p, uint32_t i)
{
    uint32_t rb, re;
    if (!p || !rows_valid(p, &rb, &re) || i >= out_rows(p)) return UINT32_MAX;
    if (p->tile_rows == 0 || p->n_parts <= 1) return rb + i;
    const uint32_t t = p->part + (i / p->tile_rows) * p->n_parts;
    return rb + t * p->tile_rows + i % p->tile_rows;
}

int mirt_grid_plan(const MirtSphere* spheres, uint32_t n_spheres, uint64_t lds_bytes_per_block, MirtGridPlan* out)
{
    if (!out || (n_spheres && !spheres)) return fail(MIRT_ERR_NULL_POINTER, "spheres/out is null");
    *out = MirtGridPlan{};
    const size_t lds = lds_bytes_per_block ? (size_t)lds_bytes_per_block : (size_t)160 * 1024;
    double f = 0.0;
    const Tuning tune = read_tuning();                 // the experiment knobs MIRT_GRID_CELL / MIRT_GRID_BIG apply as in mirt_ctx_create
    const std::vector<unsigned char> grid = plan_grid(spheres, n_spheres, tune.grid_cell, tune.grid_big, lds, &f);
    if (grid.empty() || n_spheres > 4095u || kx::scene_lds_bytes_grid(n_spheres, true) + grid.size() > mirt::kMaxLdsBytes) return MIRT_OK;
    const mirt::GridHeader* gh = reinterpret_cast<const mirt::GridHeader*>(grid.data());
    out->cell_factor = (float)f;
    out->blob_bytes = (uint32_t)grid.size();
    out->n_cells = gh->dims[0] * gh->dims[1] * gh->dims[2];
    out->n_entries = gh->n_entries;
    out->n_big = gh->n_big;
    const size_t beside = kx::scene_lds_bytes_grid(n_spheres, true) + grid.size();
    out->pool_slots = beside < lds ? kx::pool_config_grid(lds - beside).slots : 0u;
    return MIRT_OK;
}

int mirt_ctx_create(int device, MirtContext** out)
{
    if (!out) return fail(MIRT_ERR_NULL_POINTER, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MIRT_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    MirtContext* c = new (std::nothrow) MirtContext();
    if (!c) return fail(MIRT_ERR_ALLOC, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < kEventPool && e == hipSuccess; ++i) {
        hipEvent_t a = nullptr, b = nullptr;
        e = hipEventCreate(&a);
        if (e == hipSuccess) e = hipEventCreate(&b);
        if (a) c->ev_begin.push_back(a);
        if (b) c->ev_end.push_back(b);
        hipEvent_t z = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&z, hipEventDisableTiming);
        if (z) c->ev_zeroed.push_back(z);
    }
    c->slot_busy.assign(kEventPool, 0);
    c->slot_zeroing.assign(kEventPool, 0);
    c->slot_dirty.assign(kEventPool, 0);
    c->slot_zero_event.assign(kEventPool, 0);
    c->slot_timed.assign(kEventPool, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->zero_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_accum, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->d_sky, sizeof(MirtSkyState));
    if (e == hipSuccess) e = hipMalloc(&c->d_counters, sizeof(unsigned long long) * mirt::kNumCounters * kEventPool);
    if (e == hipSuccess) e = hipMalloc(&c->d_work_counter, sizeof(uint32_t) * kEventPool * kDispenserWords);
    if (e == hipSuccess) e = hipMemset(c->d_work_counter, 0, sizeof(uint32_t) * kEventPool * kDispenserWords);     // see retire_slots / launch_render
    if (e != hipSuccess) {
        const int rc = fail(MIRT_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        mirt_ctx_destroy(c);
        return rc;
    }
    c->cu_count = prop.multiProcessorCount;
    c->lds_per_block = prop.sharedMemPerBlock;            // 160 KiB on gfx950
    c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    c->tuning = read_tuning();
    c->timing = c->tuning.timing != 0;
    *out = c;
    return MIRT_OK;
}

void mirt_ctx_destroy(MirtContext* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->zero_stream) (void)hipStreamSynchronize(c->zero_stream);
    if (c->frame_stream_b) (void)hipStreamSynchronize(c->frame_stream_b);
    if (!c->untimed_streams.empty()) (void)hipDeviceSynchronize();      // launches without an event may still read the tables freed below
    (void)hipFree(c->d_spheres); (void)hipFree(c->d_mats); (void)hipFree(c->d_pmats); (void)hipFree(c->d_grid); (void)hipFree(c->d_shade); (void)hipFree(c->d_texels);
    (void)hipFree(c->d_sky); (void)hipFree(c->d_counters); (void)hipFree(c->d_work_counter); (void)hipFree(c->d_out); (void)hipFree(c->d_accum);
    for (hipEvent_t ev : c->ev_begin) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->ev_end) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->ev_zeroed) (void)hipEventDestroy(ev);
    if (c->zero_stream) (void)hipStreamDestroy(c->zero_stream);
    if (c->frame_stream_b) (void)hipStreamDestroy(c->frame_stream_b);
    if (c->ev_accum) (void)hipEventDestroy(c->ev_accum);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mirt_ctx_set_scene(MirtContext* c, const MirtScene* s)
{
    if (!c || !s || !s->camera) return fail(MIRT_ERR_NULL_POINTER, "ctx/scene/camera is null");
    if (s->n_spheres && !s->spheres) return fail(MIRT_ERR_NULL_POINTER, "spheres is null");
    if (s->n_materials && !s->materials) return fail(MIRT_ERR_NULL_POINTER, "materials is null");
    if (s->n_texels && !s->texels) return fail(MIRT_ERR_NULL_POINTER, "texels is null");
    // The flat kernels stage spheres AND materials in LDS; the grid build (path-traced mode, many spheres)
    // only the spheres + the grid.  A scene is accepted if at least one of the two layouts fits.
    const bool fits_flat = kx::scene_lds_bytes(s->n_spheres, s->n_materials, true, true) <= mirt::kMaxLdsBytes;
    // The cell size trades sphere tests against cells walked AND against LDS: the blob shares the CU's 160 KB with the path pools,
    // and a pool geometry lost to a large cell table costs more than finer cells bring (RTIOW: cell = 2 median radii drops the
    // pools from 152 to 128 slots per wave and the frame from 13.2 to 19.7 ms): plan_grid.
    double grid_factor = 0.0;
    std::vector<unsigned char> grid = plan_grid(s->spheres, s->n_spheres, c->tuning.grid_cell, c->tuning.grid_big, c->lds_per_block, &grid_factor);
    const bool fits_grid = !grid.empty() && s->n_spheres <= 4095u &&
                           kx::scene_lds_bytes_grid(s->n_spheres, true) + grid.size() <= mirt::kMaxLdsBytes;   // camera (+ sky) + the blob
    if (!fits_flat && !fits_grid)
        return fail(MIRT_ERR_SCENE_TOO_LARGE, "%u spheres + %u materials exceed the %u-byte LDS budget", s->n_spheres,
                    s->n_materials, mirt::kMaxLdsBytes);
    HIP_TRY(hipSetDevice(c->device));
    // Failure-atomic: from here until the last copy has succeeded the context holds NO scene, so an allocation or
    // copy that fails half-way leaves render calls answering MIRT_ERR_NO_SCENE instead of launching on freed or
    // half-written tables.
    c->have_scene = false;
    c->fits_flat = fits_flat;

    // mode-specific validity is decided here once and reported by the render call that needs it
    c->pt_scene_status = MIRT_OK;
    for (uint32_t i = 0; i < s->n_spheres && c->pt_scene_status == MIRT_OK; ++i)
        if (s->spheres[i].material_idx >= s->n_materials) c->pt_scene_status = MIRT_ERR_MATERIAL_INDEX;
    for (uint32_t i = 0; i < s->n_materials && c->pt_scene_status == MIRT_OK; ++i) {
        const MirtMaterial& m = s->materials[i];
        if ((m.id == 0 || m.id == 1 || m.id == 3) && !desc_ok(m.desc1, s->n_texels)) c->pt_scene_status = MIRT_ERR_TEXEL_RANGE;
        if (m.id == 3 && !desc_ok(m.desc2, s->n_texels)) c->pt_scene_status = MIRT_ERR_TEXEL_RANGE;
    }
    uint32_t routine_queue[5] = {0, 1, 2, 3, 4};    // routine id -> queue of the pool kernel
    {   // which scatter routines can a path meet?  (decides the path-traced schedule; the pool kernel
        // numbers the routines present densely so that scenes with <= 3 of them run a 4-queue build)
        uint32_t seen = 0;
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const uint32_t mi = s->spheres[i].material_idx;
            if (mi < s->n_materials) { const uint32_t id = s->materials[mi].id; seen |= 1u << (id < 4u ? id : 4u); }
        }
        c->n_shading_routines = (uint32_t)__builtin_popcount(seen);
        uint32_t q = 0;
        for (uint32_t r = 0; r < 5; ++r) if (seen & (1u << r)) { routine_queue[r] = q; c->queue_routine[q] = r; ++q; }
        for (uint32_t r = 0; r < 5; ++r) if (!(seen & (1u << r))) { routine_queue[r] = q; c->queue_routine[q] = r; ++q; }
    }
    c->parity_scene_status = MIRT_OK;
    if (s->n_spheres > 0) {   // layer.rs:345-349 reads material_data[2] on every primary hit
        if (s->n_materials < 3) c->parity_scene_status = MIRT_ERR_MATERIAL_INDEX;
        else if (!desc_ok(s->materials[2].desc1, s->n_texels)) c->parity_scene_status = MIRT_ERR_TEXEL_RANGE;
    }

    std::vector<mirt::PreparedSphere> prep(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const MirtSphere& in = s->spheres[i];
        mirt::PreparedSphere& o = prep[i];
        o.cx = in.center[0]; o.cy = in.center[1]; o.cz = in.center[2];
        o.rr = in.radius * in.radius;
        o.inv_r = 1.0f / in.radius;
        o.radius = in.radius;
        o.material_idx = in.material_idx;
        o.op = routine_queue[(in.material_idx < s->n_materials && s->materials[in.material_idx].id < 4u) ? s->materials[in.material_idx].id : 4u];
    }
    // PreparedMaterial: GpuMaterial + 1/x + the texel of every 1x1 texture (see mirt_kernels.h)
    std::vector<mirt::PreparedMaterial> pmats(s->n_materials);
    std::vector<unsigned char> mat_has_image(s->n_materials, 0);     // a texture larger than 1x1: what MIRT_FLAG_TEXEL_TILES caches in LDS
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const MirtMaterial& in = s->materials[i];
        mirt::PreparedMaterial& o = pmats[i];
        std::memset(&o, 0, sizeof o);
        o.id = in.id;
        o.x = in.x;
        o.inv_x = 1.0f / in.x;
        const MirtTextureDescriptor* d[2] = { &in.desc1, &in.desc2 };
        for (int k = 0; k < 2; ++k) {
            uint32_t w = d[k]->width, h = d[k]->height, off = d[k]->offset;
            if (w == 1 && h == 1 && (uint64_t)off < s->n_texels) {
                o.flags |= (1u << k);
                o.tex[k][0] = s->texels[3 * (size_t)off + 0];
                o.tex[k][1] = s->texels[3 * (size_t)off + 1];
                o.tex[k][2] = s->texels[3 * (size_t)off + 2];
                std::memcpy(&o.tex[k][3], &off, 4);
            } else {
                if ((uint64_t)w * h > 1u) mat_has_image[i] = 1;
                std::memcpy(&o.tex[k][0], &w, 4);
                std::memcpy(&o.tex[k][1], &h, 4);
                std::memcpy(&o.tex[k][2], &off, 4);
            }
        }
    }
    if (!grid.empty()) {                         // routine queue of every sphere, for the grid builds (GridHeader.off_ops)
        const mirt::GridHeader* gh = reinterpret_cast<const mirt::GridHeader*>(grid.data());
        for (uint32_t i = 0; i < s->n_spheres; ++i) {            // the ROUTINE itself, min(GpuMaterial.id, 4): what the scatter step switches on
            const uint32_t mi = s->spheres[i].material_idx;
            grid[gh->off_ops + i] = (unsigned char)((mi < s->n_materials && s->materials[mi].id < 4u) ? s->materials[mi].id : 4u);
        }
    }
    // grid builds: sphere centre, 1/r and a copy of the sphere's material in one 64-byte record (mirt_kernels.h: ShadeRec)
    std::vector<mirt::ShadeRec> shade;
    if (fits_grid && s->n_materials) {
        shade.resize(s->n_spheres);
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const uint32_t mi = prep[i].material_idx < s->n_materials ? prep[i].material_idx : 0u;   // out of range: pt_scene_status refuses the launch
            const mirt::PreparedMaterial& pm = pmats[mi];
            float fl, idf;
            std::memcpy(&fl, &pm.flags, 4);
            std::memcpy(&idf, &pm.id, 4);
            shade[i] = mirt::ShadeRec{ { prep[i].inv_r, pm.x, pm.inv_x, fl }, { pm.tex[0][0], pm.tex[0][1], pm.tex[0][2], pm.tex[0][3] },
                                       { pm.tex[1][0], pm.tex[1][1], pm.tex[1][2], pm.tex[1][3] }, { prep[i].cx, prep[i].cy, prep[i].cz, idf } };
        }
    }
    int rc;
    if (!shade.empty() && (rc = ensure_capacity(&c->d_shade, &c->cap_shade, shade.size())) != MIRT_OK) return rc;
    if ((rc = ensure_capacity(&c->d_pmats, &c->cap_pmats, (size_t)s->n_materials)) != MIRT_OK) return rc;
    if ((rc = ensure_capacity(&c->d_spheres, &c->cap_spheres, (size_t)s->n_spheres)) != MIRT_OK) return rc;
    if ((rc = ensure_capacity(&c->d_mats, &c->cap_mats, (size_t)s->n_materials)) != MIRT_OK) return rc;
    if ((rc = ensure_capacity(&c->d_texels, &c->cap_texels, (size_t)s->n_texels * 3)) != MIRT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());      // renders may be in flight on caller streams (mirt_ctx_render_device)
    c->cam = *s->camera;
    if (s->n_spheres) HIP_TRY(hipMemcpy(c->d_spheres, prep.data(), prep.size() * sizeof(mirt::PreparedSphere), hipMemcpyHostToDevice));
    if (s->n_materials) HIP_TRY(hipMemcpy(c->d_mats, s->materials, (size_t)s->n_materials * sizeof(MirtMaterial), hipMemcpyHostToDevice));
    if (s->n_materials) HIP_TRY(hipMemcpy(c->d_pmats, pmats.data(), pmats.size() * sizeof(mirt::PreparedMaterial), hipMemcpyHostToDevice));
    if (s->n_texels) HIP_TRY(hipMemcpy(c->d_texels, s->texels, (size_t)s->n_texels * 3 * sizeof(float), hipMemcpyHostToDevice));
    if (!shade.empty()) HIP_TRY(hipMemcpy(c->d_shade, shade.data(), shade.size() * sizeof(mirt::ShadeRec), hipMemcpyHostToDevice));
    c->have_shade = !shade.empty();
    {
        c->grid_bytes = fits_grid ? (uint32_t)grid.size() : 0u;
        c->grid_packable = false;
        c->grid_flat_y = false;
        if (fits_grid) {
            const mirt::GridHeader* gh = reinterpret_cast<const mirt::GridHeader*>(grid.data());
            c->grid_packable = gh->dims[0] * gh->dims[1] * gh->dims[2] <= mirt::kGridMaxCells;      // a parked walk keeps its linear cell index (16 bit)
            c->grid_flat_y = gh->dims[1] == 1u && c->tuning.grid_flat_y != 0;
        }
        if (fits_grid) {
            if ((rc = ensure_capacity(&c->d_grid, &c->cap_grid, grid.size())) != MIRT_OK) return rc;
            HIP_TRY(hipMemcpy(c->d_grid, grid.data(), grid.size(), hipMemcpyHostToDevice));
        }
    }
    c->have_sky = s->sky != nullptr;
    if (s->sky) HIP_TRY(hipMemcpy(c->d_sky, s->sky, sizeof(MirtSkyState), hipMemcpyHostToDevice));
    c->n_spheres = s->n_spheres;
    c->n_mats = s->n_materials;
    c->n_texels = s->n_texels;
    for (uint32_t i = 0; i < 3; ++i)
        for (int k = 0; k < 4; ++k) c->sph3[4 * i + k] = (s->n_spheres == 3) ? (&prep[i].cx)[k] : 0.0f;
    c->has_image_texture = false;                // ... on a material some sphere uses
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (s->spheres[i].material_idx < s->n_materials && mat_has_image[s->spheres[i].material_idx]) c->has_image_texture = true;
    c->have_scene = true;
    return MIRT_OK;
}

int mirt_ctx_set_camera(MirtContext* c, const MirtGpuCamera* cam)
{
    if (!c || !cam) return fail(MIRT_ERR_NULL_POINTER, "ctx/camera is null");
    if (!c->have_scene) return fail(MIRT_ERR_NO_SCENE, "set_scene has not been called");
    // No device work at all: the camera is a kernel argument of every launch (RenderArgs.cam).  Launches already
    // queued keep the camera they were issued with; the next render call uses this one.  The reference updates the
    // camera every interactive frame (layer.rs:188-193, mod.rs:353-388), so this must cost nothing.
    c->cam = *cam;
    return MIRT_OK;
}

static int check_params(const MirtContext* c, const MirtParams* p)
{
    if (!c || !p) return fail(MIRT_ERR_NULL_POINTER, "ctx/params is null");
    if (!c->have_scene) return fail(MIRT_ERR_NO_SCENE, "set_scene has not been called");
    if (p->width == 0 || p->height == 0)
        return fail(MIRT_ERR_VIEWPORT_SIZE, "viewport_size elements cannot be zero: (%u, %u)", p->width, p->height);
    if (p->spp == 0) return fail(MIRT_ERR_SPP_ZERO, "spp is zero");
    // the kernels count a unit's work items (up to 64 pixels x spp) and the samples of a pixel in 32 bits
    if (p->spp > MIRT_MAX_SPP_PER_CALL || (uint64_t)p->sample_begin + p->spp > 0xffffffffull)
        return fail(MIRT_ERR_SPP_RANGE, "spp %u (from sample %u) is out of range: at most %u samples per pixel in one call, sample indices below 2^32",
                    p->spp, p->sample_begin, (unsigned)MIRT_MAX_SPP_PER_CALL);
    if (p->mode != MIRT_MODE_PARITY && p->mode != MIRT_MODE_PT) return fail(MIRT_ERR_BAD_MODE, "unknown mode %u", p->mode);
    uint32_t rb, re;
    if (!rows_valid(p, &rb, &re)) return fail(MIRT_ERR_BAD_ROWS, "invalid row selection [%u,%u) of %u, part %u/%u", p->row_begin, p->row_end, p->height, p->part, p->n_parts);
    if (p->mode == MIRT_MODE_PT && p->frame_spp != 0 && (p->spp % p->frame_spp != 0 || p->sample_begin % p->frame_spp != 0))
        return fail(MIRT_ERR_FRAME_SPP, "frame_spp %u must divide spp %u and sample_begin %u", p->frame_spp, p->spp, p->sample_begin);
    if (p->mode == MIRT_MODE_PARITY) {
        if (c->parity_scene_status != MIRT_OK)
            return fail(c->parity_scene_status, "parity mode needs material_data[2] with a valid texture (layer.rs:345-351)");
    } else {
        if (c->pt_scene_status != MIRT_OK) return fail(c->pt_scene_status, "scene tables are inconsistent: %s", mirt_status_string(c->pt_scene_status));
        if ((p->flags & MIRT_FLAG_SKY_HOSEK) && !c->have_sky) return fail(MIRT_ERR_SKY, "MIRT_FLAG_SKY_HOSEK needs scene.sky");
    }
    if ((uint64_t)out_rows(p) * p->width > 0xffffffffull) return fail(MIRT_ERR_BAD_ROWS, "more than 2^32 pixels in one call");
    return MIRT_OK;
}

// ---- the ring of launch slots ----
// Wait for the launch recorded in slot i (if any) and add its kernel time to ms_folded.
static int fold_slot(MirtContext* c, size_t i)
{
    if (!c->slot_busy[i]) return MIRT_OK;
    HIP_TRY(hipEventSynchronize(c->ev_end[i]));
    float ms = 0.0f;
    if (c->slot_timed[i]) HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin[i], c->ev_end[i]));    // (untimed launches carry an end event only)
    c->ms_folded += ms;
    c->launches_folded += 1;
    c->last_ms = ms;
    c->slot_busy[i] = 0;
    c->ev_in_flight -= 1;
    return MIRT_OK;
}

// Wait until a re-zeroing queued for slot i has landed.
static int await_zero(MirtContext* c, size_t i)
{
    if (!c->slot_zeroing[i]) return MIRT_OK;
    HIP_TRY(hipEventSynchronize(c->ev_zeroed[c->slot_zero_event[i]]));
    c->slot_zeroing[i] = 0;
    return MIRT_OK;
}

// Retire slots [first, first + n): fold their launches' times and -- if any of their kernels took units from the dispenser
// (slot_dirty) -- re-zero the dispenser words of all of them with ONE memset on the context's zero_stream (the eight words of a slot
// sit 4 KB apart: 32 KB per slot).  The host has seen every kernel's end before it queues the memset, so no stream ever waits on
// another stream's event; ev_zeroed[first] says when the memset has landed (await_zero).
static int retire_slots(MirtContext* c, size_t first, size_t n)
{
    bool dirty = false;
    for (size_t i = first; i < first + n; ++i) {
        int rc = await_zero(c, i);                                  // (an earlier memset of this slot: let it land first)
        if (rc == MIRT_OK) rc = fold_slot(c, i);
        if (rc != MIRT_OK) return rc;
        dirty = dirty || c->slot_dirty[i];
    }
    if (!dirty) return MIRT_OK;
    HIP_TRY(hipMemsetAsync(c->d_work_counter + first * kDispenserWords, 0, sizeof(uint32_t) * kDispenserWords * n, c->zero_stream));
    HIP_TRY(hipEventRecord(c->ev_zeroed[first], c->zero_stream));
    for (size_t i = first; i < first + n; ++i) {
        c->slot_dirty[i] = 0;
        c->slot_zeroing[i] = 1;
        c->slot_zero_event[i] = (unsigned char)first;
    }
    return MIRT_OK;
}

// The ring retires its slots in batches of 16, half a ring ahead of the slot in use (launch_render): one small fill kernel per 16
// launches instead of one per launch (per-launch fills cost the dispensed 4-spp frame 2.4 %, profiles/r04_ring_ab.txt).
constexpr size_t kRingBatch = 16;

// The blocking path (mirt_ctx_get_stats): fold every launch in flight, oldest first (last_ms = the newest launch's time), and leave
// every slot clean.
static int fold_events(MirtContext* c)
{
    const size_t n = c->ev_begin.size();
    for (size_t k = c->ev_in_flight; k > 0; --k) {
        const int rc = fold_slot(c, (c->ev_next + n - k) % n);
        if (rc != MIRT_OK) return rc;
    }
    for (size_t b = 0; b < n / kRingBatch; ++b) {
        const int rc = retire_slots(c, b * kRingBatch, kRingBatch);
        if (rc != MIRT_OK) return rc;
    }
    for (size_t i = 0; i < n; ++i) {
        const int rc = await_zero(c, i);
        if (rc != MIRT_OK) return rc;
    }
    return MIRT_OK;
}

// A launch went wrong after its kernel was enqueued: make the slot safe to reuse whatever state the stream is in.
static void poison_slot(MirtContext* c, size_t i, hipStream_t stream)
{
    (void)hipStreamSynchronize(stream);
    (void)hipStreamSynchronize(c->zero_stream);
    (void)hipMemset(c->d_work_counter + i * kDispenserWords, 0, sizeof(uint32_t) * kDispenserWords);
    if (c->slot_busy[i]) { c->slot_busy[i] = 0; c->ev_in_flight -= 1; }
    c->slot_zeroing[i] = 0;
    c->slot_dirty[i] = 0;
}

static int launch_render(MirtContext* c, const MirtParams* p, uint32_t* d_out, hipStream_t stream,
                         unsigned long long* d_accum = nullptr)
{
    // The ring's next slot.  When the ring enters a batch of 16 slots, the batch HALF A RING AHEAD is retired (retire_slots): its launches
    // -- 32 to 17 launches old -- have normally finished long ago (the waits return at once; a caller more than 17 launches ahead of the
    // GPU is held here, which is what a bounded queue should do), their times are folded and their dispenser words re-zeroed on the
    // context's own stream.  By the time the ring reaches those slots they are clean: the pipeline is never drained.
    static_assert(kEventPool % (2 * kRingBatch) == 0, "the ring is a whole number of batch pairs");
    const size_t ev = c->ev_next;
    {
        int rc = MIRT_OK;
        if (ev % kRingBatch == 0) rc = retire_slots(c, (ev + kEventPool / 2) % kEventPool, kRingBatch);
        if (rc == MIRT_OK && (c->slot_busy[ev] || c->slot_dirty[ev])) rc = retire_slots(c, ev, 1);   // (not in the normal flow: the batch ahead was retired 32 launches ago)
        if (rc == MIRT_OK) rc = await_zero(c, ev);               // at most one wait for the batch's memset, 32 launches old
        if (rc != MIRT_OK) return rc;
        if (c->tuning.debug_slots) {                 // MIRT_DEBUG_SLOTS=1 (tests): the slot's eight dispenser words ARE zero now
            uint32_t wds[8];
            HIP_TRY(hipMemcpy2D(wds, sizeof(uint32_t), c->d_work_counter + ev * kDispenserWords, sizeof(uint32_t) * kDispenserStride, sizeof(uint32_t), 8, hipMemcpyDeviceToHost));
            for (int k = 0; k < 8; ++k)
                if (wds[k] != 0u) return fail(MIRT_ERR_HIP, "launch slot %zu: dispenser word %d holds %u before the launch", ev, k, wds[k]);
        }
    }
    const uint32_t rows = out_rows(p);
    const uint64_t npix = (uint64_t)rows * p->width;
    const bool count = (p->flags & MIRT_FLAG_COUNT_WORK) != 0;
    const bool hosek = p->mode == MIRT_MODE_PT && (p->flags & MIRT_FLAG_SKY_HOSEK);

    const bool pt = p->mode == MIRT_MODE_PT;
    const size_t scene_lds = kx::scene_lds_bytes(c->n_spheres, c->n_mats, pt, hosek);
    // kernel choice (path-traced mode): the pooled kernel needs enough samples per tile to keep
    // its path pool full, 8-bit bounce counters and room for the pool beside the scene in LDS
    const Tuning& tune = c->tuning;
    // MIRT_FLAG_TEXEL_TILES: flat scenes with an image texture run the pool geometry that keeps a texel window per wave
    const bool tiles = pt && (p->flags & MIRT_FLAG_TEXEL_TILES) && c->has_image_texture && tune.pool_config < 0;
    const uint32_t pool_cfg = tune.pool_config >= 0 ? (uint32_t)tune.pool_config : (tiles ? mirt::kTilePoolConfig : mirt::kDefaultPoolConfig);
    const uint32_t pool_nq = kx::pool_scatter_queues(c->n_shading_routines, count);
    const mirt::PoolConfig pc = kx::pool_config(pool_cfg, pool_nq);
    // Default schedule: the pooled kernel pays off when a strip holds enough samples to keep the pool full and the
    // pools still leave >= 16 waves per CU resident beside the scene tables; the sample counts from which it does are
    // measured crossovers against the strip kernel's lane-per-pixel schedule (mirt_kernels.h: kPoolMinSpp*): 28 for scenes
    // with several shading routines, 600 for single-routine scenes (nothing diverges there, so lane = pixel is hard to
    // beat: single metal sphere, 1080p x 100 spp, 1.38 ms against the pool's 1.82 and the lane-per-sample schedule's 2.27),
    // 16 for many-sphere scenes, where the pool's re-compaction of grid walks is worth most.
    const size_t lds_pool_block = scene_lds + pc.lds_bytes;
    const uint32_t pool_waves_per_cu = (uint32_t)(c->lds_per_cu / (lds_pool_block ? lds_pool_block : 1)) * (pc.threads / 64u);
    // (round 4, the streaming build of lane = pixel: a single-routine scene on a frame of at least eight rounds of 16-pixel units per resident
    //  wave -- 1 Mpixel on 256 CUs -- never takes the pool: single sphere 1080p x 800 / 2000 / 4000 spp 5.38 / 13.3 / 26.5 ms against the pool's
    //  6.34 / 15.5 / 31.2, 3840x2160 x 1000 spp 26.3 / 30.5; at 800x600 x 1000 spp the pool still wins, 2.17 against 2.31.  Scenes with several
    //  routines: streaming 1.16 ms at 32 spp, pool 1.17; 36 / 48 spp 1.28 / 1.61 against 1.24 / 1.50 -- the crossover stays.  r04_lowspp_ab.txt block 9)
    const bool stream_any_spp = c->n_shading_routines <= 1 && !hosek && npix >= 16ull * 8u * 32u * (uint64_t)c->cu_count;
    const uint32_t pool_min_spp = tune.pool_min_spp > 0 ? (uint32_t)tune.pool_min_spp
                                : stream_any_spp ? UINT32_MAX
                                : c->n_shading_routines <= 1 ? mirt::kPoolMinSppOneRoutine : mirt::kPoolMinSpp;
    bool pool = pt && p->spp >= pool_min_spp && c->n_shading_routines >= 1 && pool_waves_per_cu >= 16;
    if (p->flags & MIRT_FLAG_KERNEL_STRIP) pool = false;
    if (p->flags & MIRT_FLAG_KERNEL_POOL) pool = pt;
    // the reference's per-frame RNG stream makes a pixel's samples sequentially dependent: lane-per-pixel strip kernel only
    const bool frame_stream = pt && p->frame_spp != 0;
    if (frame_stream) pool = false;
    if (p->num_bounces > 255u || scene_lds + pc.lds_bytes > (size_t)c->lds_per_block || !c->fits_flat) pool = false;
    // Many-sphere scenes: the pool kernel's grid build keeps spheres + grid + pools in LDS (materials in L2).  It
    // is taken when the flat pool is not (too few waves per CU beside a big sphere table) and >= 12 waves fit.
    const size_t scene_lds_g = kx::scene_lds_bytes_grid(c->n_spheres, hosek);
    const bool grid_ok = pt && (!count || ((p->flags & MIRT_FLAG_COUNT_GRID) && !frame_stream)) && c->grid_bytes != 0 &&
                         !(p->flags & MIRT_FLAG_NO_GRID);
    // one 1024-thread block per CU; its path pools take what scene + grid leave of the block's LDS (counting launches:
    // the largest geometry only)
    const size_t lds_beside = scene_lds_g + c->grid_bytes;
    const size_t lds_grid_block = (size_t)c->lds_per_block / mirt::kGridPoolBlocksPerCu;
    mirt::PoolConfig pcg = kx::pool_config_grid(lds_grid_block > lds_beside ? lds_grid_block - lds_beside : 0);
    if (count && pcg.slots != mirt::kGridPoolSlotChoices[0] && pcg.slots != mirt::kGridPoolSlotChoices[1]) pcg.slots = 0;
    const size_t lds_pool_grid_block = lds_beside + pcg.lds_bytes;
    const uint32_t pool_grid_waves_per_cu = pcg.slots ? (uint32_t)(c->lds_per_cu / lds_pool_grid_block) * (pcg.threads / 64u) : 0u;
    bool pool_grid = grid_ok && c->grid_packable && pcg.slots != 0 && tune.pool_config < 0 && p->num_bounces <= 255u &&
                     !(p->flags & MIRT_FLAG_KERNEL_STRIP) && !frame_stream &&
                     ((p->flags & MIRT_FLAG_KERNEL_POOL) ||
                      (!pool && p->spp >= mirt::kPoolMinSppGrid && c->n_shading_routines >= 1 && pool_grid_waves_per_cu >= 16));
    if (tune.pool_grid == 0) pool_grid = false;
    if (pool_grid) pool = true;
    const mirt::PoolConfig pcu = pool_grid ? pcg : pc;                 // the geometry of the pool kernel that will run
    const uint32_t resident_pool_waves = pool_grid ? pool_grid_waves_per_cu : pool_waves_per_cu;

    mirt::RenderArgs a{};
    a.cam = c->cam;
    // the kernels' pinhole shortcut (generate_primary): the flag travels in the padding word beside lower_left_corner of
    // THIS launch's by-value copy of the camera; the caller's struct is never touched
    { const uint32_t flag = (pt && tune.pinhole != 0 && camera_is_pinhole(c->cam)) ? 1u : 0u; std::memcpy(&a.cam._padding5, &flag, 4); }
    a.spheres = c->d_spheres; a.mats = c->d_mats; a.pmats = c->d_pmats; a.texels = c->d_texels; a.sky = c->d_sky;
    a.out = d_out; a.counters = c->d_counters + ev * mirt::kNumCounters; a.work_counter = c->d_work_counter + ev * kDispenserWords; a.accum = d_accum;
    a.n_texels = c->n_texels; a.n_spheres = c->n_spheres; a.n_mats = c->n_mats;
    a.width = p->width; a.height = p->height; a.spp = p->spp; a.num_bounces = p->num_bounces; a.flags = p->flags;
    a.seed_mix = jenkins_hash((uint32_t)p->seed ^ jenkins_hash((uint32_t)(p->seed >> 32)));
    a.sample_begin = p->sample_begin;
    a.frame_spp = pt ? p->frame_spp : 0u;
    a.frame_begin = pt ? p->frame_begin : 0u;
    a.row_begin = p->row_begin; a.tile_rows = p->tile_rows; a.n_parts = p->n_parts; a.part = p->part;
    a.out_rows = rows;
    for (int q = 0; q < 5; ++q) a.queue_routine[q] = c->queue_routine[q];
    for (int k = 0; k < 12; ++k) a.sph3[k] = c->sph3[k];
    a.n_units = (uint32_t)((npix + mirt::kStripPixels - 1) / mirt::kStripPixels);
    for (uint32_t l = 0; l <= mirt::kStripLevels; ++l) { a.lvl_unit[l] = a.n_units; a.lvl_pix[l] = (uint32_t)npix; }
    a.lvl_unit[0] = 0;
    a.lvl_pix[0] = 0;
    if (pool) {
        // Guided self-scheduling for the pool kernel: 16-pixel strips while more than 4 strips per resident
        // wave remain, then 8- and 4-pixel strips under the same rule, so that the waves finish within a
        // fraction of a strip of each other.  Measured on config 3 (tools/part_timing.py, one rank's share of
        // an N-way partition vs 1/N of the whole frame): fixed 16-pixel strips 95 / 83 / 67 % at N = 2 / 4 / 8,
        // this schedule 99 / 95 / 90 %; strips narrower than 4 pixels lose more to pool fill/drain than they
        // gain.  Narrow strips also need width x spp >= 512 work items to keep a wave's pool busy.
        const uint64_t waves = (uint64_t)c->cu_count * (resident_pool_waves < 32u ? resident_pool_waves : 32u);   // resident waves (LDS-bound)
        uint32_t min_width = 4;
        while (min_width < mirt::kStripPixels && (uint64_t)min_width * p->spp < 512u) min_width *= 2;
        if (tune.fixed_strips) min_width = mirt::kStripPixels;
        if (tune.gss_min_width) min_width = tune.gss_min_width;
        const double keep_factor = tune.gss_keep > 0.0 ? tune.gss_keep : 4.0;
        uint64_t pix = 0, unit = 0;
        for (uint32_t l = 0; l < mirt::kStripLevels; ++l) {
            const uint32_t width = mirt::kStripPixels >> l;
            a.lvl_unit[l] = (uint32_t)unit;
            a.lvl_pix[l] = (uint32_t)pix;
            uint64_t level_px;
            if (width <= min_width) {
                level_px = npix - pix;                               // last usable level takes the rest
            } else {
                const uint64_t keep = (uint64_t)(keep_factor * (double)(waves * width));   // pixels left for the narrower levels
                const uint64_t left = npix - pix;
                level_px = left > keep ? (left - keep) / width * width : 0;
            }
            pix += level_px;
            unit += (level_px + width - 1) / width;
        }
        a.lvl_unit[mirt::kStripLevels] = (uint32_t)unit;
        a.lvl_pix[mirt::kStripLevels] = (uint32_t)npix;
        a.n_units = (uint32_t)unit;
    }
    // many-sphere scenes: nearest hit through the uniform grid (strip kernel, non-counting build only:
    // the counting build keeps the reference's flat scan so that its work counters stay comparable)
    const bool use_grid = pool_grid || (grid_ok && !pool && scene_lds_g + c->grid_bytes + 4u * 48u <= (size_t)c->lds_per_block);
    a.grid = use_grid ? c->d_grid : nullptr;
    a.grid_bytes = use_grid ? c->grid_bytes : 0u;
    a.shade = (use_grid && c->have_shade) ? c->d_shade : nullptr;
    a.grid_pool_slots = pool_grid ? pcg.slots : 0u;
    // camera-ray candidate lists (mirt_kernels.hip: strip_candidates): always in the pooled kernel (a list serves 16 pixels x spp rays);
    // in the strip kernel's lane-per-pixel units (64 pixels x spp) from 4 samples per pixel on -- measured on RTIOW 1080p: 2 spp +2 %
    // (selecting among 484 spheres costs more than 128 camera rays save), 8 spp -6 %
    a.strip_cand = (use_grid && tune.strip_cand != 0 && (pool_grid || p->spp >= 4u)) ? 1u : 0u;
    a.grid_flat_y = (use_grid && c->grid_flat_y) ? 1u : 0u;
    // (the strip kernel's grid build keeps a camera-ray candidate list per wave behind the blob: 4 waves x 48 bytes)
    a.lds_bytes = use_grid ? (uint32_t)(scene_lds_g + a.grid_bytes + (pool ? pcu.lds_bytes : 4u * 48u)) : (uint32_t)(scene_lds + (pool ? pcu.lds_bytes : 0));
    if (!use_grid && !c->fits_flat)
        return fail(MIRT_ERR_SCENE_TOO_LARGE, "this scene only fits LDS in the grid build of the path-traced mode "
                    "(no parity mode, no MIRT_FLAG_COUNT_WORK / MIRT_FLAG_NO_GRID / MIRT_FLAG_KERNEL_POOL)");

    // few samples per pixel (the reference's interactive loop adds 2 per frame): lane = pixel instead of lane = sample
    // (single-routine scenes: lane = pixel at any sample count, but only for frames that give every CU a few 64-pixel units;
    //  a tiny frame with many samples per pixel needs the lanes of a wave on the samples)
    bool by_pixel = pt && !pool && !count && (p->spp < mirt::kByPixelMaxSpp || (c->n_shading_routines <= 1 && npix >= 64ull * 4u * c->cu_count));
    if (tune.by_pixel >= 0) by_pixel = pt && !pool && !count && tune.by_pixel == 1;
    if (frame_stream) by_pixel = true;                    // also for counting launches (flat scan) and any spp
    // parity mode: lane = pixel at EVERY sample count, counting or not (round 3: below 64 spp).  The reference's loop returns at the first
    // terminating sample (layer.rs:320-378), which a lane that walks its own pixel's samples does too, while lane = sample computes 64
    // samples of a pixel to find it: with one unit per wave 1080p x 64 / 100 / 1000 spp take 0.29 / 0.34 / 1.89 ms against 1.14 / 1.17 /
    // 2.08 (profiles/r04_lowspp_ab.txt block 7).  MIRT_FLAG_KERNEL_STRIP still forces lane = sample (same image, same work counters).
    if (!pt) by_pixel = (p->flags & MIRT_FLAG_KERNEL_STRIP) ? false : (tune.by_pixel >= 0 ? tune.by_pixel == 1 : true);
    a.static_units = 0;
    a.px_groups_log2 = 0;
    // lane = pixel in a flat scene from 16 spp on: the streaming build -- a lane starts its pixel's next sample without waiting for the wave
    // (mirt_kernels.hip: strip_kernel_body<STREAM>; MIRT_STREAM=0/1 for A/B runs)
    a.stream_samples = (pt && by_pixel && !count && !use_grid && p->spp >= 16u) ? 1u : 0u;
    if (tune.stream >= 0 && pt && by_pixel && !count && !use_grid) a.stream_samples = (uint32_t)tune.stream;
    if (by_pixel) {
        // Path-traced lane-per-pixel units: 64 pixels x all samples -- or 32 / 16 pixels with the samples dealt to 2 / 4 groups of lanes:
        // smaller units are more units, and the last unit of a wave is then a smaller part of its life (config 2, 1080p x 100 spp, had 4
        // units per wave: 1.21 ms; 16-pixel units, with the eight-word dispenser they need: 0.86).  Measured (tools/ab_libs.py with
        // MIRT_PX_GROUPS, profiles/r03_flat_ab.txt block 9; 1080p unless stated): as many groups as leave each >= 8 samples -- three spheres
        // at 32 / 16 / 8 spp: 1 / 2 / 4 groups 1.33 / 1.20 / 1.16, 0.68 / 0.63 / 0.66, 0.36 / 0.37 / 0.43 ms; config 2 1.17 / 0.93 / 0.86;
        // the same at 3840x2160 3.20 / 2.96 / 2.89, at 800x600 0.71 / 0.41 / 0.33 -- and one more for many-sphere scenes (RTIOW 8 spp:
        // 1.54 / 1.40).  The samples must divide evenly; never with the reference's per-frame stream (sequentially dependent samples).
        if (pt && !frame_stream && tune.px_groups != 0) {
            // (round 4, one unit per wave, profiles/r04_lowspp_ab.txt block 4: two groups from shares of 4 -- three spheres 8 spp 0.337 ms against
            //  0.359, main.rs scene 0.434 / 0.459 --, four groups from shares of 6: 16 spp 0.610 with two groups, 0.617 with four; 24 spp 0.884 / 0.881)
            // (streaming launches, below: shares of at least 8 samples -- three spheres 16 spp 0.588 ms with two groups, 0.643 with four; 24 spp
            //  0.822 / 0.886; config 2 at 25 per lane, four groups: 0.750, two: 0.829, eight (128 spp): +-0; block 9)
            const uint32_t min_share[2] = { a.stream_samples ? 8u : 4u, a.stream_samples ? 8u : (use_grid ? 4u : 6u) };
            while (a.px_groups_log2 < 2u && p->spp % (2u << a.px_groups_log2) == 0u && (p->spp >> (a.px_groups_log2 + 1u)) >= min_share[a.px_groups_log2])
                a.px_groups_log2 += 1u;
            if (tune.px_groups > 0 && p->spp % (1u << (tune.px_groups - 1)) == 0u) a.px_groups_log2 = (uint32_t)tune.px_groups - 1u;
        }
        a.n_units = (uint32_t)((npix + (64u >> a.px_groups_log2) - 1u) / (64u >> a.px_groups_log2));
        // ONE UNIT PER WAVE instead of units dispensed to a persistent grid (a.static_units; round 4): the launch has as many waves as units, and
        // the hardware's workgroup dispatcher -- which starts a block wherever a CU has room -- is the load balancer: no dispenser atomic
        // (one returning device-scope atomic per unit, 14 ns each on one address, was the whole kernel time of a 2-spp frame), and none of
        // the imbalance of dealing units round-robin to resident waves (rounds 2-3: a wave's four units are a quarter of its life each).
        // Measured against both (profiles/r04_lowspp_ab.txt block 3; 1080p, three spheres): 1 / 2 / 4 / 8 / 16 / 24 spp -14 / -16 / -13 / -4 / -2 /
        // +-0 %; main.rs scene 2 spp -20 %; config 2 -1.6 %; parity mode's lane = pixel 2 spp (1080p) -11 %, 32 spp -21 %.  Every block stages
        // the scene itself, which is why many-sphere scenes (a 25 KB grid blob per block) keep the dispenser from 4 spp on (RTIOW: 2 spp -21 %,
        // 4 / 8 / 12 spp +7 / +8 / +7 %).
        a.static_units = (!pt || !use_grid || p->spp < 4u) ? 1u : 0u;
        if (tune.static_units >= 0) a.static_units = (uint32_t)tune.static_units;
    }

    uint32_t blocks;
    if (pool) {
        uint32_t per_cu = (uint32_t)(c->lds_per_cu / (a.lds_bytes ? a.lds_bytes : 1));
        const uint32_t by_waves = 32u / (pcu.threads / 64u);   // upper bound; LDS decides (6 blocks of 4 waves with 112-slot pools)
        if (per_cu > by_waves) per_cu = by_waves;
        if (tune.pool_blocks_per_cu >= 1 && tune.pool_blocks_per_cu < per_cu) per_cu = tune.pool_blocks_per_cu;
        if (per_cu == 0u) per_cu = 1u;
        blocks = (uint32_t)c->cu_count * per_cu;
        const uint32_t units_per_block = pcu.threads / 64u;
        const uint32_t need = (a.n_units + units_per_block - 1) / units_per_block;
        if (blocks > need) blocks = need;
    } else {
        // One unit per wave in a flat scene: one WAVE per block, too -- a block leaves when its slowest wave is done, and the scene a block
        // stages is a few hundred bytes (1080p, three spheres 2 / 4 / 8 / 16 spp -1.6 / -2.2 / -1.5 / -1.3 %, config 2 -1.9 %, parity mode at
        // 1000 spp nominal -15 %; profiles/r04_lowspp_ab.txt block 8).  Many-sphere scenes stage a 25 KB grid blob per block: 256 threads
        // (RTIOW 2 spp with 64-thread blocks: 2.4 x the time).  MIRT_STATIC_BLOCK=64/128/256 for A/B runs.
        a.launch_threads = 0u;
        if (a.static_units != 0u && tune.static_grid <= 0) a.launch_threads = tune.static_block ? tune.static_block : (use_grid ? 0u : 64u);
        const uint32_t waves_per_block = (a.launch_threads ? a.launch_threads : mirt::kBlockThreads) / 64;
        blocks = (a.n_units + waves_per_block - 1) / waves_per_block;
        // a persistent grid of exactly the blocks that are resident at once: registers and LDS decide (4-8 per CU for these
        // kernels).  A block beyond that would hold its first unit until the dispenser has run dry and run it alone at the end.
        const bool fast_strip = pt && !count && (p->flags & MIRT_FLAG_FAST_MATH);
        uint32_t per_cu = !pt ? kx::parity_blocks_per_cu(count, by_pixel, a.lds_bytes)
                        : fast_strip ? kf::strip_blocks_per_cu(hosek, count, use_grid, by_pixel, a.lds_bytes, a.stream_samples != 0u)
                                     : kx::strip_blocks_per_cu(hosek, count, use_grid, by_pixel, a.lds_bytes, a.stream_samples != 0u);
        if (per_cu > 8u) per_cu = 8u;                                 // 2048 threads per CU / 256
        const uint32_t resident = (uint32_t)c->cu_count * per_cu;
        // ... unless the launch runs one unit per wave (a.static_units): then the grid is the units (MIRT_STATIC_GRID=k, A/B runs: k x the
        // resident blocks with the units dealt round-robin, rounds 2-3's schedule at k = 1)
        uint32_t cap = resident;
        if (a.static_units != 0u) cap = tune.static_grid > 0 ? resident * (uint32_t)tune.static_grid : blocks;
        if (blocks > cap) blocks = cap;
    }
    if (blocks == 0) blocks = 1;

    // the dispenser continues after the units the waves take by their own index (first_unit() in the kernels)
    const uint32_t launched_waves = blocks * ((pool ? pcu.threads : (a.launch_threads ? a.launch_threads : mirt::kBlockThreads)) / 64u);
    // kernels with dispensed units take them from eight dispenser words (mirt_kernels.hip: next_unit_any): the strip-type kernels
    // (path-traced strip kernel, both schedules; parity kernel) always.
    // The pooled kernel's strips last long at high sample counts (config 3: 9 atomics per microsecond); below 128 spp they do not
    // (three spheres, 1080p, pool forced: 48 / 64 / 100 / 200 spp -13.5 / -8.5 / -2 / +1 % with eight words; RTIOW 16 spp -4.5 %).
    a.spread_units = (a.static_units == 0u && tune.spread_units != 0 && (!pool || p->spp < 128u)) ? 1u : 0u;
    a.first_dispensed = launched_waves;          // the words themselves are zero (retire_slots): no memset node in front of the kernel
    for (uint32_t x = 0; x < 8u; ++x) a.disp_taken[x] = launched_waves > x ? (launched_waves - x + 7u) / 8u : 0u;
#ifdef MIRT_DIAG_STAMPS
    HIP_TRY(hipMemsetAsync(a.counters, 0, sizeof(unsigned long long) * mirt::kNumCounters, stream));
#endif
    if (count) HIP_TRY(hipMemsetAsync(a.counters, 0, sizeof(unsigned long long) * mirt::kNumCounters, stream));
    // Events.  A TIMED launch (mirt_ctx_set_timing, the default) carries a start and an end event for mirt_ctx_get_stats; they ride on
    // the kernel dispatch itself (hipExtLaunchKernel), not as two record packets in the stream: per launch of the reference's own loop
    // (parity mode, 2 spp, 800x600) 20.6 us with records, 17.9 with the events on the dispatch, 12.9 with none (profiles/r04_ring_ab.txt
    // block 3).  An UNTIMED launch carries an end event only if something must learn that it has finished: its dispenser words have to
    // be re-zeroed (dispensed units), or its counters will be read (counting launch).  The reference's interactive frames -- units
    // dealt round-robin -- carry none.
    const bool timed = c->timing;
    const bool need_end = timed || count || a.static_units == 0u;
    const bool ext_events = tune.ext_events && need_end;
    if (timed && !ext_events) HIP_TRY(hipEventRecord(c->ev_begin[ev], stream));
    const mirt::LaunchOn on = ext_events ? mirt::LaunchOn(stream, timed ? c->ev_begin[ev] : nullptr, c->ev_end[ev]) : mirt::LaunchOn(stream);
    // the opt-in fast-math build of the path-traced kernels; counting launches always run the exact build
    const bool fast = pt && !count && (p->flags & MIRT_FLAG_FAST_MATH);
    const char* tf[2] = { "false", "true" };
    char kname[112] = "";
    if (p->mode == MIRT_MODE_PARITY) {
        HIP_TRY(kx::launch_parity(a, blocks, count, by_pixel, on));
        snprintf(c->last_kernel, sizeof c->last_kernel, "render_parity_kernel<%s,%s>", tf[count], tf[by_pixel]);
    } else if (pool) {
        HIP_TRY(fast ? kf::launch_pt_pool(a, blocks, pool_cfg, count, pool_nq, on) : kx::launch_pt_pool(a, blocks, pool_cfg, count, pool_nq, on));
        kx::pool_kernel_name(a, pool_cfg, count, pool_nq, kname, sizeof kname);
        snprintf(c->last_kernel, sizeof c->last_kernel, "%s%s", fast ? "fast_build::" : "", kname);
    } else {
        HIP_TRY(fast ? kf::launch_pt_strip(a, blocks, count, use_grid, by_pixel, on) : kx::launch_pt_strip(a, blocks, count, use_grid, by_pixel, on));
        if (a.stream_samples) snprintf(c->last_kernel, sizeof c->last_kernel, "%srender_pt_stream_kernel<%s>", fast ? "fast_build::" : "", tf[hosek]);
        else snprintf(c->last_kernel, sizeof c->last_kernel, "%srender_pt_strip_kernel<%s,%s,%s,%s>", fast ? "fast_build::" : "", tf[count], tf[hosek],
                      tf[use_grid], tf[by_pixel]);
    }
    // The kernel is enqueued: from here on the slot is in use, whatever happens below (a failure after this point must not hand the
    // slot -- its dispenser words no longer zero -- to the next launch: poison_slot).
    c->ev_next = (ev + 1) % c->ev_begin.size();
    hipError_t he = hipSuccess;
    if (need_end) {
        c->slot_busy[ev] = 1;
        c->slot_timed[ev] = timed ? 1 : 0;
        c->ev_in_flight += 1;
        if (!ext_events) he = hipEventRecord(c->ev_end[ev], stream);
    } else {                                      // nothing will ever wait for this launch through the context ...
        c->launches_folded += 1;
        c->last_ms = 0.0;
        if (stream != c->stream && std::find(c->untimed_streams.begin(), c->untimed_streams.end(), stream) == c->untimed_streams.end())
            c->untimed_streams.push_back(stream);   // ... except mirt_ctx_synchronize / _destroy, through its stream
    }
    // Kernels that take units from the dispenser leave its words non-zero: retire_slots re-zeroes them before the slot's next use, on the
    // context's zero_stream -- never a memset node on the caller's stream, where it costs 3 us between kernels.  Launches whose units are
    // dealt round-robin (the reference's 2-spp frames, parity mode's lane = pixel) never touch the words.
    if (a.static_units == 0u) c->slot_dirty[ev] = 1;
    if (he == hipSuccess && d_accum && stream != c->stream) {   // resolve/read (on the context's stream) must see sums added on a caller stream
        he = hipEventRecord(c->ev_accum, stream);
        if (he == hipSuccess) c->accum_pending = true;
    }
    if (he != hipSuccess) {
        poison_slot(c, ev, stream);
        return fail(MIRT_ERR_HIP, "%s", hipGetErrorString(he));
    }
    c->last_slot = ev;
    c->stats_counted = count;
    c->stats = MirtStats{};
    c->stats.samples = npix * p->spp;
    return MIRT_OK;
}

int mirt_ctx_render_device(MirtContext* c, const MirtParams* p, void* d_out, size_t out_len, void* hip_stream)
{
    int rc = check_params(c, p);
    if (rc != MIRT_OK) return rc;
    if (!d_out) return fail(MIRT_ERR_NULL_POINTER, "d_out_rgba8 is null");
    const size_t need = (size_t)out_rows(p) * p->width * 4;
    if (out_len < need) return fail(MIRT_ERR_OUT_BUFFER, "output buffer holds %zu bytes, %zu needed", out_len, need);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return launch_render(c, p, (uint32_t*)d_out, st);
}

int mirt_ctx_render(MirtContext* c, const MirtParams* p, uint8_t* out, size_t out_len)
{
    int rc = check_params(c, p);
    if (rc != MIRT_OK) return rc;
    if (!out) return fail(MIRT_ERR_NULL_POINTER, "out_rgba8 is null");
    const size_t npix = (size_t)out_rows(p) * p->width;
    if (out_len < npix * 4) return fail(MIRT_ERR_OUT_BUFFER, "output buffer holds %zu bytes, %zu needed", out_len, npix * 4);
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = ensure_capacity(&c->d_out, &c->cap_out, npix)) != MIRT_OK) return rc;
    if ((rc = launch_render(c, p, c->d_out, c->stream)) != MIRT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->d_out, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MIRT_OK;
}

const char* mirt_ctx_last_kernel(const MirtContext* c) { return c ? c->last_kernel : ""; }

int mirt_ctx_synchronize(MirtContext* c)
{
    if (!c) return fail(MIRT_ERR_NULL_POINTER, "ctx is null");
    HIP_TRY(hipSetDevice(c->device));
    for (size_t i = 0; i < c->ev_begin.size(); ++i) if (c->slot_busy[i]) HIP_TRY(hipEventSynchronize(c->ev_end[i]));
    HIP_TRY(hipStreamSynchronize(c->zero_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->frame_stream_b) HIP_TRY(hipStreamSynchronize(c->frame_stream_b));
    for (hipStream_t st : c->untimed_streams) HIP_TRY(hipStreamSynchronize(st));      // launches that carried no event (mirt_ctx_set_timing(0))
    c->untimed_streams.clear();
    return MIRT_OK;
}

// Two streams of the context for hosts that keep two frames in flight.  Stream 0 is the context's own stream (normal priority); stream 1 is
// created here, on first use, with the device's HIGHEST priority: HIP keeps its hardware queues per priority, so the two can never share
// one -- two normal-priority streams share a queue whenever the process holds more streams than the runtime has queues (4 by default),
// and kernels of streams on one queue run strictly one after the other (measured with torch's pool streams: 1080p, 2 spp: 98.6 us per
// frame on a shared queue, 77.8 on two queues, 83.4 for this pair; profiles/r04_ring_ab.txt block 7).
int mirt_ctx_frame_stream(MirtContext* c, uint32_t index, void** out_hip_stream)
{
    if (!c || !out_hip_stream) return fail(MIRT_ERR_NULL_POINTER, "ctx/out_hip_stream is null");
    if (index > 1u) return fail(MIRT_ERR_BAD_ROWS, "a context has frame streams 0 and 1, not %u", index);
    HIP_TRY(hipSetDevice(c->device));
    if (index == 1u && !c->frame_stream_b) {
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->frame_stream_b, hipStreamNonBlocking, greatest));
    }
    *out_hip_stream = index == 0u ? (void*)c->stream : (void*)c->frame_stream_b;
    return MIRT_OK;
}

int mirt_ctx_set_timing(MirtContext* c, int enabled)
{
    if (!c) return fail(MIRT_ERR_NULL_POINTER, "ctx is null");
    c->timing = enabled != 0;
    return MIRT_OK;
}

int mirt_ctx_get_stats(MirtContext* c, MirtStats* out)
{
    if (!c || !out) return fail(MIRT_ERR_NULL_POINTER, "ctx/out is null");
    HIP_TRY(hipSetDevice(c->device));
    const bool had_launch = c->ev_in_flight > 0 || c->launches_folded > 0;
    const int rc = fold_events(c);
    if (rc != MIRT_OK) return rc;
    c->stats.kernel_ms = c->last_ms;
    c->stats.kernel_ms_total = c->ms_folded;
    c->stats.launches = c->launches_folded;
    c->ms_folded = 0.0;
    c->launches_folded = 0;
#ifdef MIRT_DIAG_STAMPS
    if (had_launch) {                            // wave-cycles per section of the pooled kernel (mirt_kernels.hip: Stamps)
        unsigned long long h[mirt::kNumCounters];
        HIP_TRY(hipMemcpy(h, c->d_counters + c->last_slot * mirt::kNumCounters, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "MIRT_STAMPS pop %llu gen %llu scatter %llu big %llu setup_fresh %llu cells_fresh %llu tail %llu setup_resume %llu cells_resume %llu strip_end %llu\n",
                h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27]);
    }
#endif
    if (had_launch && c->stats_counted) {
        unsigned long long h[mirt::kNumCounters];
        HIP_TRY(hipMemcpy(h, c->d_counters + c->last_slot * mirt::kNumCounters, sizeof h, hipMemcpyDeviceToHost));
        c->stats.rays = h[mirt::kCntRays];
        c->stats.sphere_tests = h[mirt::kCntTests];
        c->stats.roots = h[mirt::kCntRoots];
        c->stats.hits = h[mirt::kCntHits];
        for (int i = 0; i < 5; ++i) c->stats.scatter[i] = h[mirt::kCntScatter0 + i];

        c->stats.sky_misses = h[mirt::kCntSky];
        c->stats.lane_iterations = h[mirt::kCntLaneIters];
        c->stats.wave_iterations = h[mirt::kCntWaveIters];
        c->stats.grid_cells = h[mirt::kCntCells];
        c->stats.grid_wave_cells = h[mirt::kCntWaveCells];
        c->stats.texel_fetches[0] = h[mirt::kCntTexelsPrimary];
        c->stats.texel_fetches[1] = h[mirt::kCntTexelsLater];
        c->stats.texel_tile_hits[0] = h[mirt::kCntTileHitsPrimary];
        c->stats.texel_tile_hits[1] = h[mirt::kCntTileHitsLater];
#ifdef MIRT_DIAG_STEPS
        fprintf(stderr, "MIRT_DIAG steps scatter/gen/walk %llu %llu %llu paths %llu %llu %llu ff_steps %llu traces cut/hit/miss %llu %llu %llu strips %llu with_list %llu candidates %llu\n",
                h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27], h[28], h[29], h[30]);
#endif
#ifdef MIRT_PROBE_TEXELS
        c->stats.grid_cells = h[12]; c->stats.grid_wave_cells = h[13]; c->stats.lane_iterations = h[14]; c->stats.wave_iterations = h[15];
#endif
#ifdef MIRT_PROBE_UNIFORM_NEXT
        c->stats.grid_cells = h[14]; c->stats.grid_wave_cells = h[15]; c->stats.scatter[4] = h[12]; c->stats.roots = h[13];
#endif
        c->stats_counted = false;
    }
    *out = c->stats;
    return MIRT_OK;
}

int mirt_ctx_accum_reset(MirtContext* c, const MirtParams* p)
{
    int rc = check_params(c, p);
    if (rc != MIRT_OK) return rc;
    if (p->mode != MIRT_MODE_PT) return fail(MIRT_ERR_BAD_MODE, "progressive accumulation exists in path-traced mode only");
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t npix = (uint64_t)out_rows(p) * p->width;
    if (c->accum_pending) { HIP_TRY(hipEventSynchronize(c->ev_accum)); c->accum_pending = false; }   // no add may still be running
    if ((rc = ensure_capacity(&c->d_accum, &c->cap_accum, (size_t)npix * 3)) != MIRT_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->d_accum, 0, (size_t)npix * 3 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->accum_pixels = npix;
    c->accum_samples = 0;
    c->accum_width = p->width;
    c->accum_rows = out_rows(p);
    return MIRT_OK;
}

int mirt_ctx_accum_add(MirtContext* c, const MirtParams* p, void* hip_stream)
{
    int rc = check_params(c, p);
    if (rc != MIRT_OK) return rc;
    if (p->mode != MIRT_MODE_PT) return fail(MIRT_ERR_BAD_MODE, "progressive accumulation exists in path-traced mode only");
    if (!c->d_accum || c->accum_width != p->width || c->accum_rows != out_rows(p))
        return fail(MIRT_ERR_OUT_BUFFER, "accumulation buffer does not match these params: call mirt_ctx_accum_reset first");
    HIP_TRY(hipSetDevice(c->device));
    MirtParams q = *p;
    q.sample_begin = c->accum_samples;                  // continue the RNG stream where the last frame stopped
    // check_params saw the CALLER's sample_begin; the one that counts is this one.  A caller that changes frame_spp between adds
    // without a reset would start mid-frame: the lane-per-pixel kernel seeds only at frame boundaries, so the sums would be wrong silently.
    if ((uint64_t)q.sample_begin + q.spp > 0xffffffffull)
        return fail(MIRT_ERR_SPP_RANGE, "%u samples accumulated so far + %u would pass 2^32: reset first", q.sample_begin, q.spp);
    if (q.frame_spp != 0 && q.sample_begin % q.frame_spp != 0)
        return fail(MIRT_ERR_FRAME_SPP, "frame_spp %u does not divide the %u samples accumulated so far: reset first", q.frame_spp, q.sample_begin);
    rc = launch_render(c, &q, nullptr, hip_stream ? (hipStream_t)hip_stream : c->stream, c->d_accum);
    if (rc == MIRT_OK) c->accum_samples += p->spp;
    return rc;
}

uint32_t mirt_ctx_accum_samples(const MirtContext* c) { return c ? c->accum_samples : 0u; }

int mirt_ctx_accum_resolve(MirtContext* c, const MirtParams* p, uint8_t* out, size_t out_len)
{
    if (!c || !p || !out) return fail(MIRT_ERR_NULL_POINTER, "ctx/params/out is null");
    if (!c->d_accum || c->accum_samples == 0) return fail(MIRT_ERR_NO_SCENE, "nothing accumulated yet");
    if (out_len < c->accum_pixels * 4) return fail(MIRT_ERR_OUT_BUFFER, "output buffer holds %zu bytes, %llu needed", out_len,
                                                    (unsigned long long)c->accum_pixels * 4);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_capacity(&c->d_out, &c->cap_out, (size_t)c->accum_pixels)) != MIRT_OK) return rc;
    // adds may sit on a caller stream: the HOST waits for the last of them (this call blocks anyway).  Never hipStreamWaitEvent here: with the
    // event recorded on the legacy default stream (torch's current stream when no side stream is set) that call crashes inside the HIP
    // runtime of ROCm 7.2 (tests/test_gpu_api.py::test_accumulation_on_the_default_stream_then_resolve).
    if (c->accum_pending) { HIP_TRY(hipEventSynchronize(c->ev_accum)); c->accum_pending = false; }
    HIP_TRY(kx::launch_resolve(c->d_accum, c->d_out, c->accum_pixels, c->accum_samples, p->flags, c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->d_out, (size_t)c->accum_pixels * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MIRT_OK;
}

int mirt_ctx_accum_read(MirtContext* c, uint64_t* out_sums, size_t out_len_u64)
{
    if (!c || !out_sums) return fail(MIRT_ERR_NULL_POINTER, "ctx/out is null");
    if (!c->d_accum) return fail(MIRT_ERR_NO_SCENE, "nothing accumulated yet");
    if (out_len_u64 < c->accum_pixels * 3) return fail(MIRT_ERR_OUT_BUFFER, "output buffer too small");
    HIP_TRY(hipSetDevice(c->device));
    if (c->accum_pending) HIP_TRY(hipEventSynchronize(c->ev_accum));                 // adds may sit on a caller stream
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out_sums, c->d_accum, (size_t)c->accum_pixels * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MIRT_OK;
}

int mirt_ctx_selftest_math(MirtContext* c, uint64_t out[2])
{
    if (!c || !out) return fail(MIRT_ERR_NULL_POINTER, "ctx/out is null");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = mirt_ctx_synchronize(c); if (rc != MIRT_OK) return rc; }    // borrows counter block 0
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), c->stream));
    HIP_TRY(kx::launch_selftest_math(c->d_counters, c->stream));
    unsigned long long h[2] = { 0, 0 };
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out[0] = h[0];
    out[1] = h[1];
    return MIRT_OK;
}

int mirt_render(const MirtScene* scene, const MirtParams* params, int device, uint8_t* out, size_t out_len)
{
    MirtContext* c = nullptr;
    int rc = mirt_ctx_create(device, &c);
    if (rc != MIRT_OK) return rc;
    rc = mirt_ctx_set_scene(c, scene);
    if (rc == MIRT_OK) rc = mirt_ctx_render(c, params, out, out_len);
    mirt_ctx_destroy(c);
    return rc;
}

int mirt_rgba8_to_rgb8(const uint8_t* rgba, size_t n_pixels, uint8_t* rgb)
{
    if (!rgba || !rgb) return fail(MIRT_ERR_NULL_POINTER, "rgba/rgb is null");
    for (size_t i = 0; i < n_pixels; ++i) {
        rgb[3 * i + 0] = rgba[4 * i + 0];
        rgb[3 * i + 1] = rgba[4 * i + 1];
        rgb[3 * i + 2] = rgba[4 * i + 2];
    }
    return MIRT_OK;
}

int mirt_ctx_deinterleave_device(MirtContext* c, const MirtParams* p, const void* d_parts, size_t part_stride,
                                 void* d_out, size_t out_len, void* hip_stream)
{
    if (!c || !p || !d_parts || !d_out) return fail(MIRT_ERR_NULL_POINTER, "ctx/params/buffers is null");
    uint32_t rb, re;
    if (p->width == 0 || p->height == 0 || !rows_valid(p, &rb, &re)) return fail(MIRT_ERR_BAD_ROWS, "invalid row selection");
    if (p->tile_rows == 0 || p->n_parts <= 1) return fail(MIRT_ERR_BAD_ROWS, "params describe no tile interleave");
    const uint32_t band = re - rb;
    if (out_len < (size_t)band * p->width * 4) return fail(MIRT_ERR_OUT_BUFFER, "output buffer too small");
    if (part_stride % 4 != 0) return fail(MIRT_ERR_OUT_BUFFER, "part_stride must be a multiple of 4 bytes");
    MirtParams q = *p;
    uint32_t max_rows = 0;
    for (uint32_t i = 0; i < p->n_parts; ++i) { q.part = i; const uint32_t r = out_rows(&q); if (r > max_rows) max_rows = r; }
    if (part_stride < (size_t)max_rows * p->width * 4) return fail(MIRT_ERR_OUT_BUFFER, "part_stride smaller than the largest part");
    HIP_TRY(hipSetDevice(c->device));
    mirt::DeinterleaveArgs d{};
    d.parts = (const uint32_t*)d_parts; d.out = (uint32_t*)d_out; d.part_stride_px = part_stride / 4;
    d.width = p->width; d.band_rows = band; d.tile_rows = p->tile_rows; d.n_parts = p->n_parts;
    HIP_TRY(kx::launch_deinterleave(d, hip_stream ? (hipStream_t)hip_stream : c->stream));
    return MIRT_OK;
}

}  // extern "C"
