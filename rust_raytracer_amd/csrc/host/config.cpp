// CLI flags and camera set-up — behavioural restatement of reference src/config.rs and
// src/camera.rs:47-130.  Where the reference panics on a malformed value (`.expect(...)`,
// config.rs:89-139) these functions return an error message.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_internal.h"

namespace rth {

bool parse_f64(const std::string& s, double* out) {
    if (s.empty() || s[0] == ' ' || s[0] == '\t' || s[0] == '\n') return false;
    // Rust's f64::from_str has no hex floats; strtod does — reject them.
    for (char c : s)
        if (c == 'x' || c == 'X') return false;
    char* end = nullptr;
    *out = std::strtod(s.c_str(), &end);
    return end == s.c_str() + s.size();
}

bool parse_usize(const std::string& s, size_t* out) {
    if (s.empty()) return false;
    size_t i = 0;
    if (s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    unsigned long long v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        unsigned long long nv = v * 10 + (unsigned long long)(s[i] - '0');
        if (nv < v) return false;
        v = nv;
    }
    *out = size_t(v);
    return true;
}

bool parse_vec3(const std::string& s, double out[3], std::string* err) {
    // utils.rs:39-50: split(",") then parse each component; exactly three are required.
    std::vector<double> comps;
    size_t b = 0;
    for (;;) {
        size_t c = s.find(',', b);
        std::string part = s.substr(b, c == std::string::npos ? std::string::npos : c - b);
        double x;
        if (!parse_f64(part, &x)) {
            if (err) *err = "Vector component must be a number";
            return false;
        }
        comps.push_back(x);
        if (c == std::string::npos) break;
        b = c + 1;
    }
    if (comps.size() != 3) {
        if (err) *err = "ParseError: Vector must have three components";
        return false;
    }
    out[0] = comps[0]; out[1] = comps[1]; out[2] = comps[2];
    return true;
}

static bool is_ws(char c) {
    // regex \s (ASCII subset is enough for command lines and scene files)
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v';
}

bool split_key_value(const std::string& s, std::string* key, std::string* value) {
    size_t eq = s.find('=');
    if (eq == std::string::npos || eq == 0 || eq + 1 >= s.size()) return false;
    if (s.find('=', eq + 1) != std::string::npos) return false;
    for (char c : s)
        if (is_ws(c)) return false;
    *key = s.substr(0, eq);
    *value = s.substr(eq + 1);
    return true;
}

SceneConfig default_scene_config() {
    SceneConfig c;
    c.output_width = 600;
    c.aspect_ratio = 1.5;
    c.focal_length = 50.0;
    c.camera_pos = point(0, 0, 1);
    c.camera_target = point(0, 0, 0);
    c.background = V4{0, 0, 0, 1};
    return c;
}

SceneConfig merge(const SceneConfig& base, const SceneConfig& over) {
    SceneConfig r;
    r.output_width = over.output_width ? over.output_width : base.output_width;
    r.aspect_ratio = over.aspect_ratio ? over.aspect_ratio : base.aspect_ratio;
    r.focal_length = over.focal_length ? over.focal_length : base.focal_length;
    r.f_number = over.f_number ? over.f_number : base.f_number;
    r.focus_distance = over.focus_distance ? over.focus_distance : base.focus_distance;
    r.camera_pos = over.camera_pos ? over.camera_pos : base.camera_pos;
    r.camera_target = over.camera_target ? over.camera_target : base.camera_target;
    r.background = over.background ? over.background : base.background;
    return r;
}

bool config_from_args(int argc, const char* const* argv, Config* out, std::string* err) {
    Config cfg;
    size_t samples_per_pixel = 250;  // config.rs:75
    for (int i = 1; i < argc; i++) {
        std::string arg = argv[i];
        if (arg.empty() || arg[0] != '-') {
            cfg.scene_name = arg;  // config.rs:150 (last one wins)
            continue;
        }
        std::string key, value;
        if (!split_key_value(arg.substr(1), &key, &value)) continue;  // no regex match: ignored
        double d;
        size_t u;
        double v3[3];
        auto need_f64 = [&](const char* what) {
            if (!parse_f64(value, &d)) { *err = std::string(what) + " must be a number"; return false; }
            return true;
        };
        auto need_usize = [&](const char* what) {
            if (!parse_usize(value, &u)) { *err = std::string(what) + " must be a positive integer"; return false; }
            return true;
        };
        if (key == "w" || key == "-width") {
            if (!need_usize("Output width")) return false;
            cfg.scene.output_width = u;
        } else if (key == "r" || key == "-aspect-ratio") {
            if (!need_f64("Aspect ratio")) return false;
            cfg.scene.aspect_ratio = d;
        } else if (key == "f" || key == "-focal-length") {
            if (!need_f64("Focal length")) return false;
            cfg.scene.focal_length = d;
        } else if (key == "a" || key == "-aperture") {
            if (!need_f64("Aperture")) return false;
            cfg.scene.f_number = d;
        } else if (key == "d" || key == "-focus-dist") {
            if (!need_f64("Focus distance")) return false;
            cfg.scene.focus_distance = d;
        } else if (key == "c" || key == "-camera-position") {
            if (!parse_vec3(value, v3, err)) return false;
            cfg.scene.camera_pos = point(v3[0], v3[1], v3[2]);
        } else if (key == "l" || key == "-look-at") {
            if (!parse_vec3(value, v3, err)) return false;
            cfg.scene.camera_target = point(v3[0], v3[1], v3[2]);
        } else if (key == "t" || key == "-threads") {
            if (!need_usize("Thread count")) return false;
            cfg.thread_count = u;
        } else if (key == "s" || key == "-samples") {
            if (!need_usize("Sample count")) return false;
            samples_per_pixel = u;
        } else if (key == "b" || key == "-background-color") {
            if (!parse_vec3(value, v3, err)) return false;
            cfg.scene.background = point(v3[0], v3[1], v3[2]);
        } else if (key == "-max-depth") {
            if (!need_usize("Max ray depth")) return false;
            cfg.max_depth = u;
        } else if (key == "-light-bias") {
            if (!need_f64("Light bias")) return false;
            if (!(d >= 0.0 && d <= 1.0)) { *err = "Light bias must be in range [0; 1]"; return false; }
            cfg.light_bias = d;
        } else if (key == "-seed") {  // new
            if (!need_usize("Seed")) return false;
            cfg.seed = u;
        } else if (key == "-gpus") {  // new
            if (!need_usize("GPU count") || u == 0) { *err = "GPU count must be a positive integer"; return false; }
            cfg.gpus = uint32_t(u);
        } else if (key == "-precision") {  // new
            if (value == "f64") cfg.precision = RT_PRECISION_F64;
            else if (value == "f32") cfg.precision = RT_PRECISION_F32;
            else { *err = "Precision must be f64 or f32"; return false; }
        } else if (key == "-pipeline") {  // new
            if (value == "auto") cfg.pipeline = RT_PIPELINE_AUTO;
            else if (value == "mega") cfg.pipeline = RT_PIPELINE_MEGAKERNEL;
            else if (value == "wavefront") cfg.pipeline = RT_PIPELINE_WAVEFRONT;
            else { *err = "Pipeline must be auto, mega or wavefront"; return false; }
        } else if (key == "-bvh") {  // new: where the mesh BVHs are built
            if (value == "host") cfg.bvh_on_device = false;
            else if (value == "device") cfg.bvh_on_device = true;
            else { *err = "BVH builder must be host or device"; return false; }
        }
        // unknown keys: ignored (config.rs:146)
    }
    if (cfg.thread_count == 0) {
        *err = "Thread count must be a positive integer";  // the reference divides by zero here
        return false;
    }
    size_t samples_per_thread = samples_per_pixel / cfg.thread_count;           // config.rs:154
    cfg.sqrt_samples_per_thread = size_t(std::sqrt(double(samples_per_thread)));  // config.rs:155
    *out = cfg;
    return true;
}

static void put3(double dst[3], V4 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// Camera::new + Camera::init, camera.rs:47-130.
void make_camera(const SceneConfig& sc, RtCameraDesc* out) {
    size_t image_width = *sc.output_width;
    double aspect_ratio = *sc.aspect_ratio;
    double focal_length = *sc.focal_length;
    V4 position = *sc.camera_pos;
    V4 look_at = *sc.camera_target;
    V4 v_up = vec(0, 1, 0);

    double hf = double(image_width) / aspect_ratio;
    size_t image_height;
    if (!(hf > 0.0)) image_height = 0;                       // `as usize` saturates, NaN -> 0
    else if (hf >= 18446744073709551615.0) image_height = SIZE_MAX;
    else image_height = size_t(hf);
    if (image_height < 1) image_height = 1;                  // camera.rs:87

    V4 direction = position - look_at;
    double focus_dist = sc.focus_distance ? *sc.focus_distance : length(direction);
    double h = 24.0 / focal_length;
    double real_aspect_ratio = double(image_width) / double(image_height);
    double viewport_height = focus_dist * h;
    double viewport_width = viewport_height * real_aspect_ratio;

    V4 w = to_unit(direction);
    V4 u = cross(v_up, w);
    V4 v = cross(w, u);

    V4 viewport_u = u * viewport_width;
    V4 viewport_v = (-v) * viewport_height;
    V4 pdu = viewport_u / double(image_width);
    V4 pdv = viewport_v / double(image_height);
    V4 upper_left = position - w * focus_dist - viewport_u / 2.0 - viewport_v / 2.0;
    V4 first_pixel = upper_left + (pdu + pdv) * 0.5;

    RtCameraDesc c{};
    c.image_width = uint32_t(image_width);
    c.image_height = uint32_t(image_height);
    put3(c.position, position);
    put3(c.first_pixel, first_pixel);
    put3(c.pixel_delta_u, pdu);
    put3(c.pixel_delta_v, pdv);
    put3(c.basis_u, u);
    put3(c.basis_v, v);
    c.has_aperture = sc.f_number ? 1u : 0u;
    c.aperture_radius = sc.f_number ? (focal_length / 1000.0) / *sc.f_number : 0.0;  // camera.rs:125-129
    *out = c;
}

}  // namespace rth
