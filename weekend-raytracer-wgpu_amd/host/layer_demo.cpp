// layer_demo.cpp — what the reference's main() does around the CPU layer (src/main.rs:117-125):
//   Layer::new -> set_global_data -> set_data -> (register_texture), written against the C++ mirror.
//
//   layer_demo --host-only MOON.ppm EARTH.ppm            checks table layout only (no GPU needed)
//   layer_demo MOON.ppm EARTH.ppm W H SPP OUT.rgba       renders Layer::scene on device 0, writes raw RGBA8
//   layer_demo --pt MOON.ppm EARTH.ppm W H NUM MAX OUT   path-traced main.rs scene: MAX/NUM progressive frames of
//                                                        NUM spp (Raytracer::render_frame), writes the final RGBA8
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mirt_host.hpp"

using namespace mirt_host;

int main(int argc, char** argv)
{
    try {
        if (argc > 1 && std::strcmp(argv[1], "--pt") == 0) {
            if (argc < 9) { std::fprintf(stderr, "usage: layer_demo --pt MOON.ppm EARTH.ppm W H NUM MAX OUT\n"); return 2; }
            RenderParams rp;
            rp.camera = default_fly_camera();
            rp.viewport_w = (uint32_t)std::atoi(argv[4]); rp.viewport_h = (uint32_t)std::atoi(argv[5]);
            rp.sampling.num_samples_per_pixel = (uint32_t)std::atoi(argv[6]);
            rp.sampling.max_samples_per_pixel = (uint32_t)std::atoi(argv[7]);
            Scene sc = Layer::scene(argv[2], argv[3]);
            // src/main.rs:538-544: the wgpu path's sphere list (same materials, different spheres)
            sc.spheres = { sphere_new({ 0.0f, -500.0f, -1.0f }, 500.0f, 0), sphere_new({ 0.0f, 1.0f, 0.0f }, 1.0f, 3),
                           sphere_new({ -5.0f, 1.0f, 0.0f }, 1.0f, 2),     sphere_new({ 5.0f, 0.8f, 1.5f }, 0.8f, 1),
                           sphere_new({ 5.0f, 1.2f, -1.5f }, 1.2f, 4) };
            Raytracer rt(sc, rp);
            std::vector<uint8_t> img;
            int frames = 0;
            while (rt.progress() < 1.0f) { img = rt.render_frame(); ++frames; }
            img = rt.render_frame();                                    // one more: must change nothing
            FILE* f = std::fopen(argv[8], "wb");
            if (!f) return 1;
            std::fwrite(img.data(), 1, img.size(), f);
            std::fclose(f);
            std::printf("pt: %d frames, progress %.2f, %zu bytes\n", frames, rt.progress(), img.size());
            return 0;
        }
        const bool host_only = argc > 1 && std::strcmp(argv[1], "--host-only") == 0;
        const int a = host_only ? 2 : 1;
        if (argc < a + 2) { std::fprintf(stderr, "usage: layer_demo [--host-only] MOON.ppm EARTH.ppm [W H SPP OUT]\n"); return 2; }
        const uint32_t w = host_only ? 800 : (uint32_t)std::atoi(argv[a + 2]);
        const uint32_t h = host_only ? 600 : (uint32_t)std::atoi(argv[a + 3]);
        RenderParams rp;
        rp.camera = default_fly_camera();
        rp.viewport_w = w; rp.viewport_h = h;
        if (!host_only) rp.sampling.num_samples_per_pixel = (uint32_t)std::atoi(argv[a + 4]);
        rp.sampling.max_samples_per_pixel = rp.sampling.num_samples_per_pixel * 4;
        rp.validate();
        const float size[2] = { (float)w, (float)h };
        Layer layer(size, rp, Layer::scene(argv[a], argv[a + 1]));
        if (!layer.set_global_data()) return 1;
        const auto& md = layer.material_data();
        std::printf("materials %zu texels %zu offsets %u %u %u %u %u ids %u %u %u %u %u\n", md.size(),
                    layer.global_texture_data().size() / 3, md[0].desc1.offset, md[0].desc2.offset, md[1].desc1.offset,
                    md[2].desc1.offset, md[4].desc1.offset, md[0].id, md[1].id, md[2].id, md[3].id, md[4].id);
        std::printf("camera llc %.9g %.9g %.9g\n", layer.camera.lower_left_corner[0], layer.camera.lower_left_corner[1],
                    layer.camera.lower_left_corner[2]);
        // RenderParams::validate error path (mod.rs:451-456)
        RenderParams bad = rp;
        bad.sampling.max_samples_per_pixel = 7; bad.sampling.num_samples_per_pixel = 2;
        try { bad.validate(); std::printf("validate: no error?!\n"); return 1; }
        catch (const MirtError& e) { std::printf("validate: %s\n", mirt_status_string(e.status)); }
        {   // a checksum of the texel table's bits (the JPEG and the PPM route must build the same table)
            uint64_t sum = 0;
            for (float t : layer.global_texture_data()) { uint32_t u; std::memcpy(&u, &t, 4); sum = sum * 1099511628211ull + u; }
            std::printf("texels fnv %016llx\n", (unsigned long long)sum);
        }
        if (host_only) return 0;
        layer.set_data(rp);
        const auto& rgba = layer.register_texture();
        FILE* f = std::fopen(argv[a + 5], "wb");
        if (!f) return 1;
        std::fwrite(rgba.data(), 1, rgba.size(), f);
        std::fclose(f);
        std::printf("rendered %ux%u spp %u -> %s (%zu bytes), imgbuf %zu bytes\n", w, h, rp.sampling.num_samples_per_pixel,
                    argv[a + 5], rgba.size(), layer.imgbuf().size());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
