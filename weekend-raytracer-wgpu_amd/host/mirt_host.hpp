// mirt_host.hpp — C++ mirror of the reference's Rust interface for the per-pixel path
// (src/raytracer/{mod,layer,texture,angle}.rs, src/fly_camera.rs), sitting directly on the C ABI
// of include/mirt.h.  Same names, argument meaning and error behaviour; `Layer::set_data` is ONE
// FFI call.  Header-only; link with -lmirt.  No pixel is computed on the host.
#pragma once

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/mirt.h"

namespace mirt_host {

struct Vec3 { float x, y, z; };

// RenderParamsValidationError (mod.rs:396-411) and every other negative status
struct MirtError : std::runtime_error {
    int status;
    MirtError(int s, const std::string& what) : std::runtime_error(std::string(mirt_status_string(s)) + ": " + what), status(s) {}
};
inline void check(int rc) { if (rc != MIRT_OK) throw MirtError(rc, mirt_last_error()); }

// Angle — angle.rs:1-50
class Angle {
    float radians_;
    explicit Angle(float r) : radians_(r) {}
public:
    static Angle degrees(float d) { return Angle(mirt_degrees_to_radians(d)); }
    static Angle radians(float r) { return Angle(r); }
    float as_degrees() const { return mirt_radians_to_degrees(radians_); }
    float as_radians() const { return radians_; }
    Angle clamp(Angle lo, Angle hi) const {
        float r = radians_;
        if (r < lo.radians_) r = lo.radians_;
        if (r > hi.radians_) r = hi.radians_;
        return Angle(r);
    }
    Angle operator+(Angle rhs) const { return Angle(radians_ + rhs.radians_); }
    bool operator==(Angle o) const { return radians_ == o.radians_; }
};

// Camera — mod.rs:489-499
struct Camera {
    Vec3 eye_pos, eye_dir, up;
    Angle vfov = Angle::degrees(30.0f);
    float aperture = 0.0f, focus_distance = 1.0f;
    MirtCamera to_c() const {
        return MirtCamera{ { eye_pos.x, eye_pos.y, eye_pos.z }, { eye_dir.x, eye_dir.y, eye_dir.z }, { up.x, up.y, up.z },
                           vfov.as_radians(), aperture, focus_distance };
    }
};

// SamplingParams — mod.rs:597-613
struct SamplingParams { uint32_t max_samples_per_pixel = 128, num_samples_per_pixel = 2, num_bounces = 8; };

// RenderParams — mod.rs:442-485
struct RenderParams {
    Camera camera;
    SamplingParams sampling;
    uint32_t viewport_w = 800, viewport_h = 600;
    void validate() const {
        const MirtCamera c = camera.to_c();
        const MirtSamplingParams s{ sampling.max_samples_per_pixel, sampling.num_samples_per_pixel, sampling.num_bounces };
        check(mirt_validate_render_params(&c, &s, viewport_w, viewport_h));
    }
};

// GpuCamera::new — mod.rs:700-741
inline MirtGpuCamera gpu_camera_new(const Camera& cam, uint32_t w, uint32_t h)
{
    MirtGpuCamera out;
    const MirtCamera c = cam.to_c();
    check(mirt_camera_new(&c, w, h, &out));
    return out;
}

// FlyCameraController::default + renderer_camera — fly_camera.rs:24-64
inline Camera default_fly_camera()
{
    const float pos[3] = { -10.0f, 2.0f, -4.0f };
    const float dx = 10.0f, dy = -1.0f, dz = 4.0f;                 // look_at - look_from
    const float focus = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    MirtCamera c;
    check(mirt_camera_from_fly_pose(pos, mirt_degrees_to_radians(25.0f), mirt_degrees_to_radians(-10.0f), 30.0f, 0.8f, focus, &c));
    Camera out;
    out.eye_pos = { c.eye_pos[0], c.eye_pos[1], c.eye_pos[2] };
    out.eye_dir = { c.eye_dir[0], c.eye_dir[1], c.eye_dir[2] };
    out.up = { c.up[0], c.up[1], c.up[2] };
    out.vfov = Angle::radians(c.vfov_radians);
    out.aperture = c.aperture;
    out.focus_distance = c.focus_distance;
    return out;
}

// Sphere::new — mod.rs:423-431
inline MirtSphere sphere_new(Vec3 center, float radius, uint32_t material_idx)
{
    return MirtSphere{ { center.x, center.y, center.z, 0.0f }, radius, material_idx, { 0u, 0u } };
}

// Texture — texture.rs:9-78.  Decoded RGB8 sources only (binary PPM "P6" or raw .rgb8 with explicit
// size): JPEG decoding belongs to the `image` crate on the Rust side and is not rebuilt here.
class Texture {
    uint32_t w_ = 0, h_ = 0;
    std::vector<float> data_;      // [w*h][3]
public:
    static Texture new_from_color(Vec3 c) { Texture t; t.w_ = t.h_ = 1; t.data_ = { c.x, c.y, c.z }; return t; }
    static Texture new_from_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h) {
        Texture t; t.w_ = w; t.h_ = h; t.data_.resize((size_t)w * h * 3);
        const float inv_255 = 1.0f / 255.0f;                       // texture.rs:30
        for (size_t i = 0; i < t.data_.size(); ++i) t.data_[i] = inv_255 * (float)rgb[i];
        return t;
    }
    // texture.rs:21-46.  JPEG files (what the reference loads: assets/moon.jpeg, assets/earthmap.jpeg) go through the
    // library's own decoder (mirt_jpeg_decode_rgb8); binary PPM is accepted beside it for already-decoded texels.
    static Texture new_from_image(const std::string& path) {       // TextureError::FileIoError / ImageLoadError on failure
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("TextureError::IoError: cannot open " + path);
        if (f.peek() == 0xff) {
            std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            uint32_t jw = 0, jh = 0;
            if (mirt_jpeg_info(bytes.data(), bytes.size(), &jw, &jh) != MIRT_OK)
                throw std::runtime_error(std::string("TextureError::ImageLoadError: ") + mirt_jpeg_last_error() + ": " + path);
            std::vector<uint8_t> rgb((size_t)jw * jh * 3);
            if (mirt_jpeg_decode_rgb8(bytes.data(), bytes.size(), rgb.data(), rgb.size()) != MIRT_OK)
                throw std::runtime_error(std::string("TextureError::ImageLoadError: ") + mirt_jpeg_last_error() + ": " + path);
            return new_from_rgb8(rgb.data(), jw, jh);
        }
        std::string magic; uint32_t w = 0, h = 0, maxv = 0;
        f >> magic >> w >> h >> maxv;
        if (magic != "P6" || maxv != 255 || w == 0 || h == 0) throw std::runtime_error("TextureError::ImageLoadError: not a binary PPM: " + path);
        f.get();
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        f.read(reinterpret_cast<char*>(rgb.data()), (std::streamsize)rgb.size());
        if ((size_t)f.gcount() != rgb.size()) throw std::runtime_error("TextureError::ImageLoadError: truncated " + path);
        return new_from_rgb8(rgb.data(), w, h);
    }
    const std::vector<float>& as_slice() const { return data_; }
    uint32_t width() const { return w_; }
    uint32_t height() const { return h_; }
};

// enum Material — mod.rs:433-438
struct Lambertian { Texture albedo; };
struct Metal { Texture albedo; float fuzz; };
struct Dielectric { float refraction_index; };
struct Checkerboard { Texture even, odd; };
using Material = std::variant<Lambertian, Metal, Dielectric, Checkerboard>;

struct Scene { std::vector<MirtSphere> spheres; std::vector<Material> materials; };

// GpuMaterial::* + append_to_global_texture_data — mod.rs:767-830
namespace gpu_material {
inline MirtTextureDescriptor empty() { return MirtTextureDescriptor{ 0u, 0u, 0xffffffffu }; }   // mod.rs:878-886
inline MirtTextureDescriptor append(const Texture& t, std::vector<float>& gtd) {
    const uint32_t offset = (uint32_t)(gtd.size() / 3);
    gtd.insert(gtd.end(), t.as_slice().begin(), t.as_slice().end());
    return MirtTextureDescriptor{ t.width(), t.height(), offset };
}
inline MirtMaterial lambertian(const Texture& a, std::vector<float>& g) { return MirtMaterial{ 0u, append(a, g), empty(), 0.0f }; }
inline MirtMaterial metal(const Texture& a, float fuzz, std::vector<float>& g) { return MirtMaterial{ 1u, append(a, g), empty(), fuzz }; }
inline MirtMaterial dielectric(float ior) { return MirtMaterial{ 2u, empty(), empty(), ior }; }
inline MirtMaterial checkerboard(const Texture& even, const Texture& odd, std::vector<float>& g) {
    const MirtTextureDescriptor d1 = append(even, g);
    const MirtTextureDescriptor d2 = append(odd, g);
    return MirtMaterial{ 3u, d1, d2, 0.0f };
}
}  // namespace gpu_material

// Layer — layer.rs:37-282 (imgui/wgpu upload paths are UI and out of scope)
class Layer {
public:
    float vp_size[2];
    MirtGpuCamera camera;
    std::vector<MirtSphere> world;          // Vec<Box<Sphere>> gathered contiguously
    std::vector<Material> materials;

    Layer(const float size[2], const RenderParams& rp, Scene scene, int device = 0)     // layer.rs:49-88
        : world(std::move(scene.spheres)), materials(std::move(scene.materials)), device_(device)
    {
        vp_size[0] = size[0]; vp_size[1] = size[1];
        camera = gpu_camera_new(rp.camera, (uint32_t)size[0], (uint32_t)size[1]);
    }
    ~Layer() { if (ctx_) mirt_ctx_destroy(ctx_); }
    Layer(const Layer&) = delete;
    Layer& operator=(const Layer&) = delete;

    // Layer::scene — layer.rs:90-123; the two image textures come from decoded files
    static Scene scene(const std::string& moon_ppm, const std::string& earth_ppm)
    {
        Scene s;
        s.materials.emplace_back(Checkerboard{ Texture::new_from_color({ 0.5f, 0.7f, 0.8f }), Texture::new_from_color({ 0.9f, 0.9f, 0.9f }) });
        s.materials.emplace_back(Lambertian{ Texture::new_from_image(moon_ppm) });
        s.materials.emplace_back(Metal{ Texture::new_from_color({ 1.0f, 0.85f, 0.57f }), 0.4f });
        s.materials.emplace_back(Dielectric{ 1.5f });
        s.materials.emplace_back(Lambertian{ Texture::new_from_image(earth_ppm) });
        s.spheres = { sphere_new({ 5.0f, 1.2f, -1.5f }, 1.2f, 4), sphere_new({ 0.0f, -500.0f, -1.0f }, 500.0f, 0),
                      sphere_new({ 0.0f, 1.0f, 0.0f }, 1.0f, 3),   sphere_new({ -5.0f, 1.0f, 0.0f }, 1.0f, 2),
                      sphere_new({ 2.0f, -1.0f, 0.0f }, 2.0f, 3),  sphere_new({ 5.0f, 0.8f, 1.5f }, 0.8f, 1) };
        return s;
    }

    // layer.rs:125-148; the call sites pass (odd, even) into a (even, odd) signature
    bool set_global_data()
    {
        material_data_.clear();
        global_texture_data_.clear();
        for (const Material& m : materials) {
            if (auto* l = std::get_if<Lambertian>(&m)) material_data_.push_back(gpu_material::lambertian(l->albedo, global_texture_data_));
            else if (auto* me = std::get_if<Metal>(&m)) material_data_.push_back(gpu_material::metal(me->albedo, me->fuzz, global_texture_data_));
            else if (auto* d = std::get_if<Dielectric>(&m)) material_data_.push_back(gpu_material::dielectric(d->refraction_index));
            else { const auto& c = std::get<Checkerboard>(m); material_data_.push_back(gpu_material::checkerboard(c.odd, c.even, global_texture_data_)); }
        }
        return true;
    }

    // layer.rs:264-282 — the hot path: one FFI call
    void set_data(const RenderParams& rp)
    {
        const uint32_t w = (uint32_t)vp_size[0], h = (uint32_t)vp_size[1];
        if (!ctx_) check(mirt_ctx_create(device_, &ctx_));
        MirtScene sc{};
        sc.camera = &camera;
        sc.spheres = world.data(); sc.n_spheres = (uint32_t)world.size();
        sc.materials = material_data_.data(); sc.n_materials = (uint32_t)material_data_.size();
        sc.texels = global_texture_data_.data(); sc.n_texels = global_texture_data_.size() / 3;
        check(mirt_ctx_set_scene(ctx_, &sc));
        MirtParams p{};
        p.width = w; p.height = h; p.spp = rp.sampling.num_samples_per_pixel; p.mode = MIRT_MODE_PARITY;
        rgba_.assign((size_t)w * h * 4, 0);
        check(mirt_ctx_render(ctx_, &p, rgba_.data(), rgba_.size()));
    }

    // layer.rs:182-186: ImageBuffer<Rgb<u8>> view
    std::vector<uint8_t> imgbuf() const
    {
        std::vector<uint8_t> rgb(rgba_.size() / 4 * 3);
        if (!rgba_.empty()) check(mirt_rgba8_to_rgb8(rgba_.data(), rgba_.size() / 4, rgb.data()));
        return rgb;
    }
    const std::vector<uint8_t>& register_texture() const { return rgba_; }    // the RGBA8 bytes imgui would get (layer.rs:150-176)
    void update_camera(const RenderParams& rp) { camera = gpu_camera_new(rp.camera, rp.viewport_w, rp.viewport_h); }   // layer.rs:188-193
    void resize(const RenderParams& rp)                                        // layer.rs:240-262
    {
        if (vp_size[0] != (float)rp.viewport_w || vp_size[1] != (float)rp.viewport_h) {
            vp_size[0] = (float)rp.viewport_w; vp_size[1] = (float)rp.viewport_h;
            update_camera(rp);
            set_data(rp);
        }
    }
    const std::vector<MirtMaterial>& material_data() const { return material_data_; }
    const std::vector<float>& global_texture_data() const { return global_texture_data_; }

private:
    int device_;
    MirtContext* ctx_ = nullptr;
    std::vector<MirtMaterial> material_data_;
    std::vector<float> global_texture_data_;
    std::vector<uint8_t> rgba_;
};

// Raytracer — the path-traced mode behind the names of src/raytracer/mod.rs:20-394 (wgpu plumbing
// omitted).  render_frame() = RenderProgress::next_frame (mod.rs:626-670) + the shader's accumulation.
class Raytracer {
public:
    Raytracer(const Scene& scene, const RenderParams& rp, int device = 0, const MirtSkyState* sky = nullptr)
        : rp_(rp), spheres_(scene.spheres)
    {
        rp.validate();                                                     // mod.rs:44-47
        for (const Material& m : scene.materials) {                        // mod.rs:160-183 (same (odd, even) swap)
            if (auto* l = std::get_if<Lambertian>(&m)) material_data_.push_back(gpu_material::lambertian(l->albedo, texels_));
            else if (auto* me = std::get_if<Metal>(&m)) material_data_.push_back(gpu_material::metal(me->albedo, me->fuzz, texels_));
            else if (auto* d = std::get_if<Dielectric>(&m)) material_data_.push_back(gpu_material::dielectric(d->refraction_index));
            else { const auto& c = std::get<Checkerboard>(m); material_data_.push_back(gpu_material::checkerboard(c.odd, c.even, texels_)); }
        }
        if (sky) { sky_ = *sky; have_sky_ = true; }
        camera_ = gpu_camera_new(rp.camera, rp.viewport_w, rp.viewport_h);
        check(mirt_ctx_create(device, &ctx_));
        upload();
    }
    ~Raytracer() { if (ctx_) mirt_ctx_destroy(ctx_); }
    Raytracer(const Raytracer&) = delete;
    Raytracer& operator=(const Raytracer&) = delete;

    void set_render_params(const RenderParams& rp)                         // mod.rs:353-388
    {
        rp.validate();
        rp_ = rp;
        camera_ = gpu_camera_new(rp.camera, rp.viewport_w, rp.viewport_h);
        check(mirt_ctx_set_camera(ctx_, &camera_));
        accumulated_ = -1;                                                 // render_progress.reset()
    }

    // all max_samples_per_pixel samples in one launch
    std::vector<uint8_t> render(uint64_t seed = 0, uint32_t flags = 0)
    {
        MirtParams p = params(rp_.sampling.max_samples_per_pixel, seed, flags);
        std::vector<uint8_t> out((size_t)p.width * p.height * 4);
        check(mirt_ctx_render(ctx_, &p, out.data(), out.size()));
        return out;
    }

    // one progressive frame: add num_samples_per_pixel until max_samples_per_pixel, return the estimate
    std::vector<uint8_t> render_frame(uint64_t seed = 0, uint32_t flags = 0)
    {
        MirtParams p = params(rp_.sampling.num_samples_per_pixel, seed, flags);
        if (accumulated_ < 0) { check(mirt_ctx_accum_reset(ctx_, &p)); accumulated_ = 0; }
        if ((uint32_t)accumulated_ + p.spp <= rp_.sampling.max_samples_per_pixel) {
            check(mirt_ctx_accum_add(ctx_, &p, nullptr));
            accumulated_ += (int)p.spp;
        }
        std::vector<uint8_t> out((size_t)p.width * p.height * 4);
        check(mirt_ctx_accum_resolve(ctx_, &p, out.data(), out.size()));
        return out;
    }
    float progress() const { return (accumulated_ < 0 ? 0.0f : (float)accumulated_) / (float)rp_.sampling.max_samples_per_pixel; }   // mod.rs:390-393

private:
    MirtParams params(uint32_t spp, uint64_t seed, uint32_t flags) const
    {
        MirtParams p{};
        p.width = rp_.viewport_w; p.height = rp_.viewport_h; p.spp = spp; p.num_bounces = rp_.sampling.num_bounces;
        p.mode = MIRT_MODE_PT; p.flags = flags | (have_sky_ ? (uint32_t)MIRT_FLAG_SKY_HOSEK : 0u); p.seed = seed;
        return p;
    }
    void upload()
    {
        MirtScene sc{};
        sc.camera = &camera_;
        sc.spheres = spheres_.data(); sc.n_spheres = (uint32_t)spheres_.size();
        sc.materials = material_data_.data(); sc.n_materials = (uint32_t)material_data_.size();
        sc.texels = texels_.data(); sc.n_texels = texels_.size() / 3;
        sc.sky = have_sky_ ? &sky_ : nullptr;
        check(mirt_ctx_set_scene(ctx_, &sc));
    }
    RenderParams rp_;
    std::vector<MirtSphere> spheres_;
    std::vector<MirtMaterial> material_data_;
    std::vector<float> texels_;
    MirtGpuCamera camera_{};
    MirtSkyState sky_{};
    bool have_sky_ = false;
    MirtContext* ctx_ = nullptr;
    int accumulated_ = -1;
};

}  // namespace mirt_host
