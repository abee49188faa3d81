// rtrace — command-line driver with the reference's interface (src/main.rs:25-88): same flags
// (README.md:21-43), same scene DSL, same console lines, writes `out.png` (ACES + sRGB, 8-bit).
// The one thing that changed is line 75 of the reference's main.rs: instead of
// `camera.render(world, lights, &mut buf)` the frame comes from librt_mi355.so (rt_render).
//
// New, optional flags (ignored by the reference's parser, so command lines stay compatible):
//   --seed=<u64>  --gpus=<n>  --precision=f64|f32  --pipeline=auto|mega|wavefront  --bvh=host|device
// With --gpus=n the frame is row-tiled in interleaved bands (rth_band_rows: 16 rows, or finer when that balances the GPUs), one
// host thread per GPU; the tiles are assembled on the host here (bench.py shows the RCCL gather path used for the
// multi-process launch).  RT_RTRACE_ONE_DEVICE=1 (tests on a one-GPU box): every part renders on device 0.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rt_host.h"

static std::string fmt_duration(double seconds) {  // Rust `{:.2?}` of a Duration
    char buf[64];
    if (seconds >= 1.0) std::snprintf(buf, sizeof buf, "%.2fs", seconds);
    else if (seconds >= 1e-3) std::snprintf(buf, sizeof buf, "%.2fms", seconds * 1e3);
    else if (seconds >= 1e-6) std::snprintf(buf, sizeof buf, "%.2f\xC2\xB5s", seconds * 1e6);
    else std::snprintf(buf, sizeof buf, "%.2fns", seconds * 1e9);
    return buf;
}

int main(int argc, char** argv) {
    using clock = std::chrono::steady_clock;
    auto t0 = clock::now();
    auto since = [&]() { return std::chrono::duration<double>(clock::now() - t0).count(); };

    RtHost* host = nullptr;
    if (rth_load(argc, argv, &host) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rth_last_error());
        return 1;
    }
    std::fputs(rth_log(host), stdout);  // "Loaded N tris", loader warnings
    const RtCameraDesc* cam = rth_camera(host);
    const RtRenderParams* params = rth_params(host);
    std::printf("Ready: %s\n", fmt_duration(since()).c_str());  // main.rs:62
    uint32_t spp = rth_samples_per_pixel(host);
    std::printf("Rendering: %ux%u @%uspp on %u threads (%u samples/thread)\n", cam->image_width, cam->image_height, spp,
                params->thread_count, spp / params->thread_count);  // main.rs:68-71
    std::fflush(stdout);

    uint32_t gpus = rth_gpus(host);
    int available = rt_device_count();
    if (available < 1) {
        std::fprintf(stderr, "Error: no HIP device (the render path has no CPU fallback)\n");
        return 1;
    }
    const char* one_dev = std::getenv("RT_RTRACE_ONE_DEVICE");
    const bool rehearsal = one_dev && std::atoi(one_dev) != 0;
    if (!rehearsal && int(gpus) > available) gpus = uint32_t(available);
    const uint32_t W = cam->image_width, H = cam->image_height;
    const uint32_t band = gpus > 1 ? rth_band_rows(H, gpus) : 16;  // interleaved row bands
    std::vector<double> frame(size_t(W) * H * 4, 0.0);  // camera.create_buffer(), main.rs:74
    std::vector<std::string> errors(gpus);
    std::vector<std::thread> workers;
    for (uint32_t g = 0; g < gpus; g++) {
        workers.emplace_back([&, g]() {
            auto tg = clock::now();
            RtScene* scene = nullptr;
            if (rt_scene_create(rth_scene(host), rehearsal ? 0 : int(g), &scene) != RT_OK) {
                errors[g] = rt_last_error();
                return;
            }
            RtRenderParams p = *params;
            if (gpus > 1) {
                p.band_rows = band;
                p.n_parts = gpus;
                p.part = g;
            }
            uint32_t rows = rt_owned_rows(H, &p);
            std::vector<double> part(size_t(rows) * W * 4);
            if (rows && rt_render(scene, cam, &p, part.data()) != RT_OK) errors[g] = rt_last_error();
            rt_scene_destroy(scene);
            if (!errors[g].empty()) return;
            uint32_t r = 0;
            for (uint32_t y = 0; y < H; y++) {
                bool mine = gpus == 1 || (y / band) % gpus == g;
                if (!mine) continue;
                std::memcpy(&frame[size_t(y) * W * 4], &part[size_t(r) * W * 4], size_t(W) * 4 * sizeof(double));
                r++;
            }
            // the reference prints one line per render thread (camera.rs:236); here: one per GPU
            std::printf("GPU %u finished in %s\n", g, fmt_duration(std::chrono::duration<double>(clock::now() - tg).count()).c_str());
            std::fflush(stdout);
        });
    }
    for (auto& t : workers) t.join();
    for (auto& e : errors)
        if (!e.empty()) {
            std::fprintf(stderr, "Error: %s\n", e.c_str());
            return 1;
        }
    std::printf("Done: %s. Writing output to file...\n", fmt_duration(since()).c_str());  // main.rs:78
    if (rth_save_png("out.png", frame.data(), W, H) != RT_OK) {                           // main.rs:23,80-82
        std::fprintf(stderr, "Error: %s\n", rth_last_error());
        return 1;
    }
    std::printf("Done! Took %s. Goodbye :)\n", fmt_duration(since()).c_str());  // main.rs:85
    rth_destroy(host);
    return 0;
}
