// mirt_kernels.hip — the gfx950 (CDNA4, wave64) kernels of the per-pixel ray-trace path.
//
// Work decomposition (both kernels): ONE WORKITEM PER PIXEL-SAMPLE.  A wave owns a strip of
// kStripPixels consecutive pixels; for each pixel its 64 lanes take samples s = lane, lane+64, …
// so a wave-instruction advances 64 samples of one pixel.  Strips are handed out by a global
// atomic dispenser (cost per pixel varies 10x between sky and glass).  The scene (camera, sphere
// list, material table) is staged once per block into LDS; all lanes read the same sphere at the
// same time, which LDS serves as a broadcast.  Wave ballots give the uniform loop exits.  Each
// wave finishes its strip with one coalesced 64-byte RGBA8 store.
//
//   render_parity : reference src/raytracer/layer.rs:264-444 semantics, bit-faithful (no fma).
//   render_pt     : behaviours of reference src/raytracer/raytracer.wgsl:50-521 (explicit fma).
//
// Compiled with -ffp-contract=off: every fused multiply-add in this file is an explicit fma_().
#include "mirt_kernels.h"
#include "mirt_device_math.h"

namespace mirt {

// ------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------

struct SceneLds {
    const float*          cam;      // 24 floats, MirtGpuCamera layout
    const PreparedSphere* spheres;
    const MirtMaterial*   mats;
    const float*          sky;      // 36 floats (MirtSkyState) or nullptr
};

MIRT_DEV size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// Cooperative global -> LDS copy of the scene tables, 16 B per thread per step.
MIRT_DEV SceneLds stage_scene(const RenderArgs& A, unsigned char* smem, bool hosek)
{
    const uint32_t n_cam = sizeof(MirtGpuCamera) / 16;
    const uint32_t n_sph = A.n_spheres * 2;
    const uint32_t n_mat = A.n_mats * 2;
    const uint32_t n_sky = hosek ? sizeof(MirtSkyState) / 16 : 0;
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint4* src_cam = reinterpret_cast<const uint4*>(A.cam);
    const uint4* src_sph = reinterpret_cast<const uint4*>(A.spheres);
    const uint4* src_mat = reinterpret_cast<const uint4*>(A.mats);
    const uint4* src_sky = reinterpret_cast<const uint4*>(A.sky);
    const uint32_t total = n_cam + n_sph + n_mat + n_sky;
    for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
        uint4 v;
        if (i < n_cam) v = src_cam[i];
        else if (i < n_cam + n_sph) v = src_sph[i - n_cam];
        else if (i < n_cam + n_sph + n_mat) v = src_mat[i - n_cam - n_sph];
        else v = src_sky[i - n_cam - n_sph - n_mat];
        dst[i] = v;
    }
    __syncthreads();
    SceneLds S;
    S.cam = reinterpret_cast<const float*>(smem);
    S.spheres = reinterpret_cast<const PreparedSphere*>(smem + 16 * n_cam);
    S.mats = reinterpret_cast<const MirtMaterial*>(smem + 16 * (n_cam + n_sph));
    S.sky = hosek ? reinterpret_cast<const float*>(smem + 16 * (n_cam + n_sph + n_mat)) : nullptr;
    return S;
}

// compact output row -> absolute image row (MirtParams contract, include/mirt.h)
MIRT_DEV uint32_t abs_row(const RenderArgs& A, uint32_t i)
{
    if (A.tile_rows == 0 || A.n_parts <= 1) return A.row_begin + i;
    const uint32_t t = A.part + (i / A.tile_rows) * A.n_parts;
    return A.row_begin + t * A.tile_rows + i % A.tile_rows;
}

MIRT_DEV uint32_t next_strip(const RenderArgs& A, uint32_t lane)
{
    uint32_t s = 0;
    if (lane == 0) s = atomicAdd(A.work_counter, 1u);
    return __builtin_amdgcn_readfirstlane(s);
}

MIRT_DEV uint32_t sat_u32(float f)   // Rust `as u32` / WGSL u32(): truncate, saturate, NaN -> 0
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

MIRT_DEV uint32_t sat_u8(float f)    // Rust `as u8` (math.rs:15-17)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 255.0f) return 255u;
    return (uint32_t)f;
}

MIRT_DEV float clamp01(float x) { return (x < 0.0f) ? 0.0f : ((x > 1.0f) ? 1.0f : x); }

// texture_lookup (mod.rs:1000-1019 == wgsl:377-387); final index clamped to the table.
MIRT_DEV f3 texture_lookup(const RenderArgs& A, const MirtTextureDescriptor d, float u, float v)
{
    const float uc = clamp01(u);
    const float vf = 1.0f - clamp01(v);
    const uint32_t j = sat_u32(uc * (float)d.width);
    const uint32_t i = sat_u32(vf * (float)d.height);
    const uint32_t idx = i * d.width + j;
    uint64_t g = (uint64_t)d.offset + (uint64_t)idx;
    if (g >= A.n_texels) g = A.n_texels - 1;
    const float* e = A.texels + 3 * g;
    return mk(e[0], e[1], e[2]);
}

MIRT_DEV unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

MIRT_DEV uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16) | 0xff000000u; }

// ------------------------------------------------------------------------------------------
// render_parity — layer.rs semantics
// ------------------------------------------------------------------------------------------

struct PRay { f3 o, d; };
struct PHit { f3 p, n; };

// `Layer::ray_hit_world_raw` (layer.rs:413-444) over `Sphere::closest_hit_raw` (mod.rs:1121-1157)
// and `update_ray_hit_info` / `set_face_normal` (mod.rs:1217-1243, 1095-1110).
// Returns the LAST sphere in list order with a root in [tmin, tmax]: `closest_hit` is reset to
// `old_hit` = rec.t, which nobody ever writes (it stays f32::MAX).
MIRT_DEV bool parity_world_hit(const SceneLds& S, uint32_t n_spheres, const PRay& ray, float tmin, float tmax,
                               const float rec_t, PHit& rec)
{
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
        if (t < tmin || closest_hit < t) {
            t = (-half_b + sq) / a;
            if (t < tmin || closest_hit < t) continue;
        }
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

__global__ __launch_bounds__(kBlockThreads) void render_parity_kernel(RenderArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const SceneLds S = stage_scene(A, smem, false);
    const uint32_t lane = threadIdx.x & 63u;
    const float wf = (float)A.width, hf = (float)A.height;
    const uint32_t npix = A.out_rows * A.width;
    const float FMAX = 3.40282347e+38f;

    const f3 eye = mk(S.cam[0], S.cam[1], S.cam[2]);
    const f3 hor = mk(S.cam[4], S.cam[5], S.cam[6]);
    const f3 ver = mk(S.cam[8], S.cam[9], S.cam[10]);
    const f3 llc = mk(S.cam[20], S.cam[21], S.cam[22]);

    for (;;) {
        const uint32_t strip = next_strip(A, lane);
        if (strip >= A.n_strips) break;
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
                if (active) prim = parity_world_hit(S, A.n_spheres, ray, 0.001f, FMAX, FMAX, rec);
                if (prim) {
                    // material hard-wired to index 2; texture looked up with SCREEN-space (uu,vv) (layer.rs:345-351)
                    const MirtMaterial m2 = S.mats[2];
                    const float fuzzy = m2.x;
                    const f3 albedo = texture_lookup(A, m2.desc1, uu, vv);
                    // scatter_metal mod.rs:1292-1315; unit_vertor divides by 3 (math.rs:147-149)
                    const f3 unit = mk(ray.d.x / 3.0f, ray.d.y / 3.0f, ray.d.z / 3.0f);
                    const float k = 2.0f * dot_nofma(unit, rec.n);      // reflect math.rs:154-159
                    PRay sc;
                    sc.o = rec.p;
                    sc.d = unit - k * rec.n;
                    scat_ok = dot_nofma(sc.d, rec.n) > 0.0f;
                    if (scat_ok) {
                        sec = parity_world_hit(S, A.n_spheres, sc, 0.001f, FMAX, FMAX, rec);
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
                const unsigned long long prim_mask = __ballot(prim);
                const uint32_t k_before = hits_before + __popcll(prim_mask & ((1ull << lane) - 1ull));
                const bool exhausted = prim && (k_before >= 20u);
                const bool term = prim && (exhausted || !scat_ok || sec);
                const unsigned long long term_mask = __ballot(term);
                if (term_mask) {
                    const int first = __builtin_ctzll(term_mask);
                    const bool black = exhausted || !scat_ok;
                    const uint32_t mine = black ? pack_rgba(0, 0, 0)
                                                : pack_rgba(sat_u8(colour.x), sat_u8(colour.y), sat_u8(colour.z));
                    rgba = __shfl(mine, first, 64);
                    done = true;
                }
                hits_before += (uint32_t)__popcll(prim_mask);
            }
            if (!done) rgba = pack_rgba(sat_u8(v * 255.0f), sat_u8(u * 255.0f), sat_u8(255.0f));   // layer.rs:380
            if (lane == p) my_px = rgba;
        }
        if (lane < kStripPixels && base + lane < npix) A.out[base + lane] = my_px;
    }
}

// ------------------------------------------------------------------------------------------
// render_pt — WGSL behaviours, one (pixel, sample) per lane
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
};

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

MIRT_DEV f3 rand_in_unit_sphere(Rng& rng)     // wgsl:480-491
{
    const float r = pow_pos(rng.next(), 0.33333f);
    const float theta = kPi * rng.next();
    const float phi = kTwoPi * rng.next();
    const SinCos t = sincos_(theta), p = sincos_(phi);
    const float rs = r * t.s;
    return mk(rs * p.c, rs * p.s, r * t.c);
}

MIRT_DEV f3 reflect3(f3 v, f3 n) { return fma3(-(2.0f * dot(v, n)), n, v); }

MIRT_DEV float max_(float a, float b) { return (b > a) ? b : a; }

struct HitUV { float u, v; };
MIRT_DEV HitUV sphere_uv(f3 n)               // wgsl:434-437
{
    const float theta = acos_(-n.y);
    const float phi = atan2_(-n.z, n.x) + kPi;
    HitUV r;
    r.u = (0.5f * kFrac1Pi) * phi;
    r.v = kFrac1Pi * theta;
    return r;
}

// scatterLambertian (wgsl:204-242): cosine-weighted direction around n through the Pixar ONB
MIRT_DEV void scatter_lambertian(const RenderArgs& A, const MirtTextureDescriptor tex, f3 n, HitUV uv, Rng& rng,
                                 f3& dir, f3& atten)
{
    const float r1 = rng.next();
    const float r2 = rng.next();
    const float sqrt_r2 = sqrt_(r2);
    const float z = sqrt_(1.0f - r2);
    const SinCos sc = sincos_(kTwoPi * r1);
    const float lx = sc.c * sqrt_r2;
    const float ly = sc.s * sqrt_r2;
    const float sg = (n.z >= 0.0f) ? 1.0f : -1.0f;
    const float aa = -1.0f / (sg + n.z);
    const float bb = n.x * n.y * aa;
    const f3 U = mk(fma_(sg * n.x, n.x * aa, 1.0f), sg * bb, -(sg * n.x));
    const f3 V = mk(bb, fma_(n.y, n.y * aa, sg), -n.y);
    const f3 wi = fma3(z, n, fma3(ly, V, lx * U));
    const float dn = dot(n, wi);
    const float k = (kFrac1Pi * max_(kEpsilon, dn)) / max_(kEpsilon, dn * kFrac1Pi);
    atten = k * texture_lookup(A, tex, uv.u, uv.v);
    dir = wi;
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
    if (!(c > 0.0f)) return 0u;
    float s = c * 1048576.0f;
    s = (s >= 4294967040.0f) ? 4294967040.0f : s;
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
    const double denom = (double)n_samples * 1048576.0;
    float m = (float)((double)sum / denom);
    if (!(flags & MIRT_FLAG_NO_TONEMAP)) {   // uncharted2 wgsl:83-92
        const float curr = uncharted2_tonemap(0.246f * m);
        const float white = 1.0f / uncharted2_tonemap(11.2f);
        m = white * curr;
    }
    if (!(flags & MIRT_FLAG_NO_SRGB))        // the Bgra8UnormSrgb surface's transfer curve
        m = (m > 0.0031308f) ? fma_(1.055f, pow_pos(m, 0.41666666f), -0.055f) : 12.92f * m;
    if (!(m > 0.0f)) return 0u;
    m = (m > 1.0f) ? 1.0f : m;
    return (uint32_t)fma_(m, 255.0f, 0.5f);
}

template <bool COUNT, bool HOSEK>
__global__ __launch_bounds__(kBlockThreads) void render_pt_kernel(RenderArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const SceneLds S = stage_scene(A, smem, HOSEK);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t npix = A.out_rows * A.width;
    const float inv_w = 1.0f / (float)A.width;
    const float inv_h = 1.0f / (float)A.height;

    const f3 eye = mk(S.cam[0], S.cam[1], S.cam[2]);
    const f3 hor = mk(S.cam[4], S.cam[5], S.cam[6]);
    const f3 ver = mk(S.cam[8], S.cam[9], S.cam[10]);
    const f3 cam_u = mk(S.cam[12], S.cam[13], S.cam[14]);
    const f3 cam_v = mk(S.cam[16], S.cam[17], S.cam[18]);
    const float lens_radius = S.cam[19];
    const f3 llc = mk(S.cam[20], S.cam[21], S.cam[22]);

    Work<COUNT> work;
    work.clear();

    for (;;) {
        const uint32_t strip = next_strip(A, lane);
        if (strip >= A.n_strips) break;
        const uint32_t base = strip * kStripPixels;
        uint32_t my_px = 0;
        for (uint32_t p = 0; p < kStripPixels; ++p) {
            const uint32_t pi = base + p;
            if (pi >= npix) break;
            const uint32_t ci = pi / A.width;
            const uint32_t x = pi - ci * A.width;
            const uint32_t y = abs_row(A, ci);
            const uint32_t pixel_index = x + y * A.width;

            unsigned long long acc_r = 0, acc_g = 0, acc_b = 0;
            for (uint32_t s0 = 0; s0 < A.spp; s0 += 64) {
                const uint32_t s = s0 + lane;
                bool alive = s < A.spp;
                // initRng wgsl:498-502 with frame = sample + 1
                Rng rng;
                rng.state = jenkins_hash((pixel_index ^ jenkins_hash(A.sample_begin + s + 1u)) ^ A.seed_mix);
                // samplePixel wgsl:114-117
                const float u = ((float)x + rng.next()) * inv_w;
                const float v = 1.0f - ((float)y + rng.next()) * inv_h;
                // cameraMakeRay wgsl:456-478
                const float lr = sqrt_(rng.next());
                const SinCos la = sincos_(kTwoPi * rng.next());
                const float lpx = lens_radius * (lr * la.c);
                const float lpy = lens_radius * (lr * la.s);
                f3 ro = eye + fma3(lpy, cam_v, lpx * cam_u);
                f3 rd = fma3(v, ver, fma3(u, hor, llc)) - ro;
                f3 thr = mk(1, 1, 1);
                f3 color = mk(0, 0, 0);

                // rayColor wgsl:124-172
                for (uint32_t bounce = 0; bounce < A.num_bounces; ++bounce) {
                    const unsigned long long alive_mask = __ballot(alive);
                    if (!alive_mask) break;
                    if constexpr (COUNT) {
                        if (lane == 0) work.add(kCntWaveIters);
                        if (alive) { work.add(kCntLaneIters); work.add(kCntRays); work.add(kCntTests, A.n_spheres); }
                    }
                    // nearest hit (wgsl:135-145, 407-429): every lane walks the same sphere list in LDS
                    const float a = dot(rd, rd);
                    const float inv_a = 1.0f / a;
                    float closest = kMaxT;
                    int best = -1;
                    for (uint32_t i = 0; i < A.n_spheres; ++i) {
                        const float4 s4 = reinterpret_cast<const float4*>(S.spheres)[2 * i];
                        const f3 oc = ro - mk(s4.x, s4.y, s4.z);
                        const float b = dot(oc, rd);
                        const float cq = dot(oc, oc) - s4.w;
                        const float disc = fma_(b, b, -(a * cq));
                        if (alive && disc > 0.0f) {
                            const float sq = sqrt_(disc);
                            float t = (-b - sq) * inv_a;
                            bool ok = (t < closest) && (t > kMinT);
                            work.add(kCntRoots);
                            if (!ok) {
                                t = (-b + sq) * inv_a;
                                ok = (t < closest) && (t > kMinT);
                                work.add(kCntRoots);
                            }
                            if (ok) { closest = t; best = (int)i; }
                        }
                    }
                    if (alive) {
                        if (best >= 0) {
                            work.add(kCntHits);
                            // sphereIntersection wgsl:431-440
                            const PreparedSphere sp = S.spheres[best];
                            const f3 hp = fma3(closest, rd, ro);
                            const f3 hn = sp.inv_r * (hp - mk(sp.cx, sp.cy, sp.cz));
                            const MirtMaterial m = S.mats[sp.material_idx];
                            f3 ndir, att;
                            switch (m.id) {     // scatterRay wgsl:174-202
                            case 0u: {
                                work.add(kCntScatter0);
                                scatter_lambertian(A, m.desc1, hn, sphere_uv(hn), rng, ndir, att);
                                break;
                            }
                            case 1u: {          // scatterMetal wgsl:244-248
                                work.add(kCntScatter1);
                                const f3 refl = reflect3(rd, hn);
                                const f3 rs = rand_in_unit_sphere(rng);
                                ndir = fma3(m.x, rs, refl);
                                const HitUV uv = sphere_uv(hn);
                                att = texture_lookup(A, m.desc1, uv.u, uv.v);
                                break;
                            }
                            case 2u: {          // scatterDielectric wgsl:250-292; always refracts when it can
                                work.add(kCntScatter2);
                                const float dn = dot(rd, hn);
                                const f3 uvn = normalize(rd);
                                const bool inside = dn > 0.0f;
                                const f3 outn = inside ? -hn : hn;
                                const float ratio = inside ? m.x : (1.0f / m.x);
                                const float dt = dot(uvn, outn);
                                const float disc = fma_(-(ratio * ratio), fma_(-dt, dt, 1.0f), 1.0f);
                                if (disc > 0.0f) {
                                    const float sq = sqrt_(disc);
                                    const f3 q = fma3(-dt, outn, uvn);
                                    ndir = normalize(fma3(-sq, outn, ratio * q));
                                    (void)rng.next();          // the discarded Schlick draw (wgsl:269)
                                } else {
                                    ndir = reflect3(rd, hn);
                                }
                                att = mk(1, 1, 1);
                                break;
                            }
                            case 3u: {          // scatterCheckerboard wgsl:300-307
                                work.add(kCntScatter3);
                                const float sx = sincos_(5.0f * hp.x).s, sy = sincos_(5.0f * hp.y).s,
                                            sz = sincos_(5.0f * hp.z).s;
                                const float sines = (sx * sy) * sz;
                                scatter_lambertian(A, (sines < 0.0f) ? m.desc1 : m.desc2, hn, sphere_uv(hn), rng,
                                                   ndir, att);
                                break;
                            }
                            default: {          // scatterMissingMaterial wgsl:309-314
                                work.add(kCntScatter4);
                                const f3 rs = rand_in_unit_sphere(rng);
                                ndir = hn + rs;
                                att = mk(0.9921f, 0.24705f, 0.57254f);
                                break;
                            }
                            }
                            ro = hp;
                            rd = ndir;
                            thr = thr * att;
                        } else {
                            work.add(kCntSky);
                            color = sky_color<HOSEK>(S, rd);
                            alive = false;
                        }
                    }
                }
                if (s < A.spp) {
                    acc_r += to_fixed(thr.x * color.x);
                    acc_g += to_fixed(thr.y * color.y);
                    acc_b += to_fixed(thr.z * color.z);
                }
            }
            acc_r = wave_sum_u64(acc_r);
            acc_g = wave_sum_u64(acc_g);
            acc_b = wave_sum_u64(acc_b);
            const uint32_t rgba = pack_rgba(resolve_channel(acc_r, A.spp, A.flags), resolve_channel(acc_g, A.spp, A.flags),
                                            resolve_channel(acc_b, A.spp, A.flags));
            if (lane == p) my_px = rgba;
        }
        if (lane < kStripPixels && base + lane < npix) A.out[base + lane] = my_px;
        work.flush(A.counters, lane);
    }
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

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

size_t scene_lds_bytes(uint32_t n_spheres, uint32_t n_mats, bool hosek)
{
    return sizeof(MirtGpuCamera) + (size_t)n_spheres * sizeof(PreparedSphere) + (size_t)n_mats * sizeof(MirtMaterial) +
           (hosek ? sizeof(MirtSkyState) : 0);
}

hipError_t launch_parity(const RenderArgs& a, uint32_t grid_blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(render_parity_kernel, dim3(grid_blocks), dim3(kBlockThreads), a.lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_pt(const RenderArgs& a, uint32_t grid_blocks, bool count, hipStream_t stream)
{
    const bool hosek = (a.flags & MIRT_FLAG_SKY_HOSEK) != 0;
    const dim3 g(grid_blocks), b(kBlockThreads);
    if (count) {
        if (hosek) hipLaunchKernelGGL((render_pt_kernel<true, true>), g, b, a.lds_bytes, stream, a);
        else       hipLaunchKernelGGL((render_pt_kernel<true, false>), g, b, a.lds_bytes, stream, a);
    } else {
        if (hosek) hipLaunchKernelGGL((render_pt_kernel<false, true>), g, b, a.lds_bytes, stream, a);
        else       hipLaunchKernelGGL((render_pt_kernel<false, false>), g, b, a.lds_bytes, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_deinterleave(const DeinterleaveArgs& a, hipStream_t stream)
{
    const uint64_t total = (uint64_t)a.band_rows * a.width;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace mirt
