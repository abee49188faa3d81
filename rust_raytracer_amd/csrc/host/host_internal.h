// Internal declarations shared by the host-side translation units.
#pragma once
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "hmath.h"
#include "scene_builder.h"

namespace rth {

// reference src/config.rs:8-18
struct SceneConfig {
    std::optional<size_t> output_width;
    std::optional<double> aspect_ratio, focal_length, f_number, focus_distance;
    std::optional<V4> camera_pos, camera_target, background;
};
SceneConfig default_scene_config();                                    // config.rs:20-29
SceneConfig merge(const SceneConfig& base, const SceneConfig& over);   // config.rs:32-43

// reference src/config.rs:46-59 (+ the new keys)
struct Config {
    SceneConfig scene;
    size_t thread_count = 1;
    size_t sqrt_samples_per_thread = 15;
    size_t max_depth = 20;
    double light_bias = 0.25;
    std::string scene_name;
    // additions
    uint64_t seed = 1;
    uint32_t gpus = 1;
    uint32_t precision = RT_PRECISION_F64;
    uint32_t pipeline = RT_PIPELINE_AUTO;
    bool bvh_on_device = false;  // --bvh=device: RT_SCENE_BVH_ON_DEVICE
};
bool config_from_args(int argc, const char* const* argv, Config* out, std::string* err);  // config.rs:62-176

bool parse_f64(const std::string& s, double* out);        // str::parse::<f64>
bool parse_usize(const std::string& s, size_t* out);      // str::parse::<usize>
bool parse_vec3(const std::string& s, double out[3], std::string* err);  // utils.rs:39-50
// `^([^=\s]+)=([^=\s]+)$` (config.rs:63 without the leading '-', loaders/scene.rs:756)
bool split_key_value(const std::string& s, std::string* key, std::string* value);

void make_camera(const SceneConfig& sc, RtCameraDesc* out);  // camera.rs:47-130

// texture/image.rs + buffer.rs:30-48 (PNG, baseline JPEG -> RGB f32)
bool load_image_rgb32f(const std::string& path, std::vector<float>* rgb, uint32_t* width, uint32_t* height, std::string* err);

std::unique_ptr<MeshData> load_obj(const std::string& path, std::string* log, std::string* err);

struct LoadedScene {
    SceneBuilder builder;
    int world = -1, lights = -1;
    SceneConfig scene_config;  // scene-level defaults (@config / hard-coded), before the CLI merge
};
// loaders/scene.rs:80-156
bool load_dsl_scene(const std::string& file_path, const std::string& asset_path, SceneRng& rng,
                    LoadedScene* out, std::string* log, std::string* err);
// scene/golden_monkey.rs:23-139
bool load_default_scene(SceneRng& rng, LoadedScene* out, std::string* log, std::string* err);

void tonemap_rgb8(const double* rgba, uint32_t w, uint32_t h, uint8_t* rgb);  // output.rs + aces.rs
bool write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h, std::string* err);

}  // namespace rth

struct RtHost {
    rth::Config config;
    rth::LoadedScene scene;
    const RtSceneDesc* desc = nullptr;
    RtCameraDesc camera{};
    RtRenderParams params{};
    std::string log;
};
