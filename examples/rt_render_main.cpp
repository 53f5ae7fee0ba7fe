// Headless stand-in for the reference's main() (src/main.rs:418-488) on top of the C ABI: the
// same job (cornell_box.obj, 400 x 400, 400 samples per pixel, 6 bounces: main.rs:20-24, 322),
// rendered on one MI355X instead of a pool of CPU threads, with the frame written as a PPM instead
// of being blitted into a window.  With --progressive the frame is delivered the way the window
// receives it: 64 x 64 blocks after 10, 20, 40 ... samples (RenderEvent::BlockComplete,
// main.rs:26-29, 356-412); the callback below is what the event loop's handler (main.rs:468-483)
// does with a block.
//
//   rt_render [--scene cornell|cornell_plane|world1|world2|world2_glass] [--obj FILE] [--size WxH]
//             [--spp N] [--depth N] [--seed N] [--device N | --gpus N | --devices a,b,..] [--sah]
//             [--wide | --wide-device] [--progressive] [--out FILE.ppm]
//
// --gpus N renders on GPUs 0 .. N-1 of the node through one multi-GPU context (the reference's one
// render() owns all its workers, main.rs:319-336, 443): the scene is built once and uploaded to every
// GPU, each renders its share of the frame and one RCCL sum-reduce combines them on GPU 0.
//
// Plain C++17 against include/rt_abi.h and include/rt_host.h; links librt_amd.so only.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.h"

namespace {

// main.rs:20-24
constexpr uint32_t SAMPLES_PER_PIXEL = 400, MAX_BOUNCES = 6, IMAGE_SIZE = 400;

struct Window {                                // the framebuffer the event loop keeps (main.rs:446-449)
    uint32_t width = 0, height = 0;
    std::vector<uint32_t> buffer;
    uint64_t blocks = 0;
};

// main.rs:468-483: copy the block's rows into the framebuffer
int on_block(void* user, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const uint32_t* block, uint32_t, uint32_t) {
    Window* win = static_cast<Window*>(user);
    for (uint32_t row = 0; row < h; row++)
        std::memcpy(&win->buffer[(size_t)(y + row) * win->width + x], block + (size_t)row * w, (size_t)w * 4);
    win->blocks++;
    return 0;
}

int die(const char* what) {
    std::fprintf(stderr, "rt_render: %s: %s\n", what, rt_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    std::string scene = "cornell", obj, out = "frame.ppm";
    uint32_t width = IMAGE_SIZE, height = IMAGE_SIZE, spp = SAMPLES_PER_PIXEL, depth = MAX_BOUNCES;
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    int device = 0;
    std::vector<int> devices;
    bool progressive = false, sah = false;
    int wide = 0;                              // 1: RT_TRAVERSAL_WIDE on the tree collapsed on the host, 2: on the tree built on the GPU
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&]() -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "rt_render: %s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--scene") scene = value();
        else if (a == "--obj") obj = value();
        else if (a == "--out") out = value();
        else if (a == "--size") { if (std::sscanf(value(), "%ux%u", &width, &height) != 2) { std::fprintf(stderr, "rt_render: --size WxH\n"); return 2; } }
        else if (a == "--spp") spp = (uint32_t)std::strtoul(value(), nullptr, 10);
        else if (a == "--depth") depth = (uint32_t)std::strtoul(value(), nullptr, 10);
        else if (a == "--seed") seed = std::strtoull(value(), nullptr, 0);
        else if (a == "--device") device = std::atoi(value());
        else if (a == "--gpus") { const int n = std::atoi(value()); devices.clear(); for (int d = 0; d < n; d++) devices.push_back(d); }
        else if (a == "--devices") {
            devices.clear();
            for (const char* c = value(); *c;) { devices.push_back((int)std::strtol(c, const_cast<char**>(&c), 10)); if (*c == ',') c++; }
        }
        else if (a == "--progressive") progressive = true;
        else if (a == "--sah") sah = true;
        else if (a == "--wide") wide = 1;
        else if (a == "--wide-device") wide = 2;
        else { std::fprintf(stderr, "rt_render: unknown argument %s\n", a.c_str()); return 2; }
    }

    const auto now = std::chrono::steady_clock::now();          // main.rs:436
    rt_context* ctx = nullptr;
    if (devices.empty()) { if (rt_context_create(device, &ctx)) return die("rt_context_create"); }
    else if (rt_context_create_multi((int)devices.size(), devices.data(), &ctx)) return die("rt_context_create_multi");
    if (sah && rt_set_default_build(RT_BUILD_SAH)) return die("rt_set_default_build");
    rt_world* world = nullptr;
    int rc;
    if (scene == "world1") rc = rt_world_create_world1(ctx, &world);
    else if (scene == "world2" || scene == "world2_glass") {
        if (obj.empty()) { std::fprintf(stderr, "rt_render: --scene %s needs --obj thing.obj\n", scene.c_str()); return 2; }
        rc = rt_world_create_world2(ctx, obj.c_str(), scene == "world2_glass" ? 1 : 0, &world);
    } else if (scene == "cornell" || scene == "cornell_plane") {
        if (obj.empty()) obj = scene == "cornell" ? "cornell_box.obj" : "cornell_box_plane.obj";   // main.rs:322
        rc = rt_world_create_cornell_box(ctx, obj.c_str(), &world);
    } else { std::fprintf(stderr, "rt_render: unknown scene %s\n", scene.c_str()); return 2; }
    if (rc) return die("scene");
    if (wide) {                                // the opt-in eight-child tree: collapsed on the host, or built on the GPU
        if (wide == 2 && rt_scene_set_wide_build(rt_world_scene(world), RT_WIDE_BUILD_DEVICE)) return die("rt_scene_set_wide_build");
        if (rt_scene_set_traversal(rt_world_scene(world), RT_TRAVERSAL_WIDE)) return die("rt_scene_set_traversal");
    }
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();

    Window win;
    win.width = width; win.height = height;
    win.buffer.assign((size_t)width * height, 0x00FFFFFFu);     // main.rs:446-448
    rt_stats st{};
    if (progressive) {
        if (rt_world_render_progressive(world, width, height, spp, depth, seed, on_block, &win, nullptr, &st))
            return die("rt_world_render_progressive");
    } else {
        std::vector<float> accum((size_t)width * height * 3, 0.0f);
        if (rt_world_render(world, width, height, spp, depth, seed, accum.data(), &st)) return die("rt_world_render");
        if (rt_tonemap(ctx, accum.data(), spp, win.buffer.data(), (size_t)width * height)) return die("rt_tonemap");
    }
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
    std::printf("Finished in %.2fs.\n", elapsed);               // main.rs:454 (scene load + tree build + render)
    std::printf("scene %.3f s, render %.2f ms on %u GPU(s): %llu paths, %llu rays, %.1f Mrays/s%s\n", build_s, st.ms_render,
                st.n_devices ? st.n_devices : 1u,
                (unsigned long long)st.paths, (unsigned long long)st.rays_traced,
                st.ms_render > 0 ? (double)st.rays_traced / st.ms_render * 1e-3 : 0.0,
                progressive ? (", " + std::to_string(win.blocks) + " block events").c_str() : "");
    if (rt_write_ppm(out.c_str(), win.buffer.data(), width, height)) return die("rt_write_ppm");
    rt_world_destroy(world);
    rt_context_destroy(ctx);
    return 0;
}
