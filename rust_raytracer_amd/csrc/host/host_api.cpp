// extern "C" entry points of librt_host.so (include/rt_host.h).
#include <cstdlib>
#include <cstring>

#include "host_internal.h"

static thread_local std::string g_err;

static int fail(const std::string& msg) {
    g_err = msg;
    return RT_E_INVALID;
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// reference src/main.rs:26-59
int rth_load(int argc, const char* const* argv, RtHost** out) {
    if (!out) return fail("rth_load: out is NULL");
    *out = nullptr;
    auto host = std::make_unique<RtHost>();
    std::string err;
    if (!rth::config_from_args(argc, argv, &host->config, &err)) return fail(err);
    const std::string& scene = host->config.scene_name;
    rth::SceneRng rng(host->config.seed);

    if (scene.empty() || scene == "golden_monkey") {
        if (!rth::load_default_scene(rng, &host->scene, &host->log, &err)) return fail(err);
    } else if (scene == "earth" || scene == "perlin" || scene == "light_test" || scene == "cornell" ||
               scene == "cornell_smoke" || scene == "tonemap_test") {
        // main.rs:32-37: the reference's other hard-coded Rust scenes are scene DATA outside the render path; here the
        // names select their DSL twins (scenes/<name>, relative to the current directory like every asset path).
        const std::string path = "scenes/" + scene;
        if (!rth::load_dsl_scene(path, "scenes/", rng, &host->scene, &host->log, &err))
            return fail("built-in scene '" + scene + "' is provided as the DSL file " + path + ": " + err);
    } else if (scene.rfind("model:", 0) == 0) {
        return fail("model: loader (Assimp) is not available");  // main.rs:38-41, out of scope
    } else {
        std::string asset_path;
        size_t slash = scene.rfind('/');
        if (slash != std::string::npos) asset_path = scene.substr(0, slash) + "/";  // main.rs:46-54
        if (!rth::load_dsl_scene(scene, asset_path, rng, &host->scene, &host->log, &err)) return fail(err);
    }

    // SceneConfig::merge(scene defaults, CLI) then Camera::new (scene.rs:144-150, golden_monkey.rs:35-46)
    rth::SceneConfig merged = rth::merge(host->scene.scene_config, host->config.scene);
    rth::make_camera(merged, &host->camera);
    host->desc = host->scene.builder.finish(host->scene.world, host->scene.lights,
                                            host->config.bvh_on_device ? RT_SCENE_BVH_ON_DEVICE : 0u);

    RtRenderParams& p = host->params;
    std::memset(&p, 0, sizeof p);
    p.sqrt_spt = uint32_t(host->config.sqrt_samples_per_thread);
    p.thread_count = uint32_t(host->config.thread_count);
    p.max_depth = uint32_t(host->config.max_depth);
    p.light_bias = host->config.light_bias;
    p.has_background = merged.background ? 1u : 0u;
    if (merged.background) {
        p.background[0] = merged.background->x;
        p.background[1] = merged.background->y;
        p.background[2] = merged.background->z;
    }
    p.seed = host->config.seed;
    p.band_rows = 0;
    p.n_parts = 1;
    p.part = 0;
    p.precision = host->config.precision;
    p.pipeline = host->config.pipeline;
    *out = host.release();
    return RT_OK;
}

void rth_destroy(RtHost* host) { delete host; }
const RtSceneDesc* rth_scene(const RtHost* host) { return host->desc; }
const RtCameraDesc* rth_camera(const RtHost* host) { return &host->camera; }
const RtRenderParams* rth_params(const RtHost* host) { return &host->params; }
uint32_t rth_gpus(const RtHost* host) { return host->config.gpus; }
uint32_t rth_band_rows(uint32_t height, uint32_t n_parts) {
    if (n_parts <= 1) return 0;
    uint32_t band = 16, best_rows = 0xFFFFFFFFu;
    for (uint32_t b : {16u, 8u, 4u, 2u, 1u}) {
        uint32_t most = 0;
        for (uint32_t g = 0; g < n_parts; g++) {
            uint32_t rows = 0;
            for (uint32_t y = 0; y < height; y++) rows += ((y / b) % n_parts == g);
            most = rows > most ? rows : most;
        }
        if (most < best_rows) { best_rows = most; band = b; }
    }
    return band;
}
uint32_t rth_samples_per_pixel(const RtHost* host) {
    return host->params.sqrt_spt * host->params.sqrt_spt * host->params.thread_count;  // camera.rs:50-51
}
const char* rth_log(const RtHost* host) { return host->log.c_str(); }

int rth_make_camera(uint32_t width, double aspect_ratio, double focal_length, double f_number,
                    double focus_distance, const double position[3], const double look_at[3],
                    RtCameraDesc* out) {
    if (!position || !look_at || !out) return fail("rth_make_camera: NULL argument");
    rth::SceneConfig sc;
    sc.output_width = width;
    sc.aspect_ratio = aspect_ratio;
    sc.focal_length = focal_length;
    if (f_number >= 0) sc.f_number = f_number;
    if (focus_distance >= 0) sc.focus_distance = focus_distance;
    sc.camera_pos = rth::point(position[0], position[1], position[2]);
    sc.camera_target = rth::point(look_at[0], look_at[1], look_at[2]);
    rth::make_camera(sc, out);
    return RT_OK;
}

int rth_tonemap_rgb8(const double* rgba, uint32_t w, uint32_t h, uint8_t* rgb_out) {
    if (!rgba || !rgb_out) return fail("rth_tonemap_rgb8: NULL argument");
    rth::tonemap_rgb8(rgba, w, h, rgb_out);
    return RT_OK;
}

int rth_save_png(const char* path, const double* rgba, uint32_t w, uint32_t h) {
    if (!path || !rgba) return fail("rth_save_png: NULL argument");
    std::vector<uint8_t> rgb(size_t(w) * h * 3);
    rth::tonemap_rgb8(rgba, w, h, rgb.data());
    std::string err;
    if (!rth::write_png_rgb8(path, rgb.data(), w, h, &err)) return fail(err);
    return RT_OK;
}

int rth_load_image(const char* path, float** rgb_out, uint32_t* w_out, uint32_t* h_out) {
    if (!path || !rgb_out || !w_out || !h_out) return fail("rth_load_image: NULL argument");
    std::vector<float> rgb;
    std::string err;
    if (!rth::load_image_rgb32f(path, &rgb, w_out, h_out, &err)) return fail(err);
    float* p = static_cast<float*>(std::malloc(rgb.size() * sizeof(float)));
    if (!p) return fail("rth_load_image: out of memory");
    std::memcpy(p, rgb.data(), rgb.size() * sizeof(float));
    *rgb_out = p;
    return RT_OK;
}
void rth_free_image(float* rgb) { std::free(rgb); }

}  // extern "C"
