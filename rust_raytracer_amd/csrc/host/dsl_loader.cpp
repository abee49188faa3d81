// Scene-DSL loader — behavioural restatement of reference src/loaders/scene.rs (the
// language is specified in the reference's docs/scene_dsl.md).  It produces the flat Hit
// tree of include/rt_mi355.h through SceneBuilder instead of `Arc<dyn Hit>` objects.
//
// Kept behaviours: line forms and skipping rules (scene.rs:83-136), `@config` keys
// (:158-212), parameter splitting on single spaces outside parentheses (:214-245),
// `$label` references resolved per entity kind with later labels overwriting earlier
// ones (:108-125), inline `( ... )` declarations (:292-452), warnings printed for bad
// lines (:93-96,127-135), "No world/lights object" (:153-155).
// `image` decodes PNG / baseline JPEG itself (image_loader.cpp); `perlin` fills its tables from the
// scene RNG (the reference: entropy-seeded, perlin.rs:21-36).
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>

#include "host_internal.h"

namespace rth {
namespace {

enum class Kind { Object, Material, TexColor, TexFloat, Noise };
struct Entity {
    Kind kind;
    int id;
};

struct Loader {
    SceneBuilder& b;
    SceneRng& rng;
    std::string asset_path;
    std::string* log;
    std::map<std::string, int> objects, materials, color_tex, float_tex, noises;
    SceneConfig scene_config = default_scene_config();

    using Params = std::vector<std::string>;

    static std::string trim(const std::string& s) {
        size_t b = 0, e = s.size();
        auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
        while (b < e && ws(s[b])) b++;
        while (e > b && ws(s[e - 1])) e--;
        return s.substr(b, e - b);
    }

    // scene.rs:214-245
    static Params parse_params(const std::string& decl) {
        Params params;
        std::string cur;
        int nest = 0;
        for (char c : decl) {
            if (c == '(') { cur.push_back(c); nest++; }
            else if (c == ')') { cur.push_back(c); nest--; }
            else if (c == ' ') {
                if (nest > 0) cur.push_back(' ');
                else { params.push_back(cur); cur.clear(); }
            } else cur.push_back(c);
        }
        params.push_back(cur);
        return params;
    }

    bool get_entity(const std::string& expr, Kind want, const std::map<std::string, int>& table,
                    const char* what, int* out, std::string* err) {
        bool is_ref = !expr.empty() && expr[0] == '$';
        bool is_inline = expr.size() >= 2 && expr.front() == '(' && expr.back() == ')';
        if (is_ref) {
            auto it = table.find(expr.substr(1));
            if (it == table.end()) {
                *err = std::string("ParseError: Invalid ") + what + " reference " + expr.substr(1);
                return false;
            }
            *out = it->second;
            return true;
        }
        if (is_inline) {
            Entity e;
            if (!parse_declaration(expr.substr(1, expr.size() - 2), &e, err)) return false;
            if (e.kind != want) {
                *err = "ParseError: Expression evaluates to a different entity type";
                return false;
            }
            *out = e.id;
            return true;
        }
        *err = "ParseError: Expected a reference or inline declaration";
        return false;
    }
    bool get_color_tex(const std::string& e, int* out, std::string* err) { return get_entity(e, Kind::TexColor, color_tex, "texture", out, err); }
    bool get_float_tex(const std::string& e, int* out, std::string* err) { return get_entity(e, Kind::TexFloat, float_tex, "texture", out, err); }
    bool get_material(const std::string& e, int* out, std::string* err) { return get_entity(e, Kind::Material, materials, "material", out, err); }
    bool get_object(const std::string& e, int* out, std::string* err) { return get_entity(e, Kind::Object, objects, "object", out, err); }
    // scene.rs:352-362: colour first, then float
    bool get_texture(const std::string& e, int* out, bool* is_color, std::string* err) {
        std::string ignored;
        if (get_color_tex(e, out, &ignored)) { *is_color = true; return true; }
        *is_color = false;
        return get_float_tex(e, out, err);
    }

    bool parse_declaration(const std::string& decl, Entity* out, std::string* err) {
        Params p = parse_params(decl);
        size_t i = 0;
        auto next = [&](std::string* s) { if (i < p.size()) { *s = p[i++]; return true; } return false; };
        std::string type;
        next(&type);
        std::string a, bb, c, d;
        double v3[3];

        // ---- textures ----
        if (type == "constant") {
            if (!next(&a)) { *err = "ParseError: Constant texture missing parameters"; return false; }
            // scene.rs:460-473: vector first; a single number falls through to f64.
            std::string verr;
            if (a.find(',') != std::string::npos) {
                if (!parse_vec3(a, v3, &verr)) { *err = verr; return false; }
                *out = {Kind::TexColor, b.tex_const_color(v3[0], v3[1], v3[2])};
                return true;
            }
            double k;
            if (!parse_f64(a, &k)) { *err = "invalid float literal"; return false; }
            *out = {Kind::TexFloat, b.tex_const_float(k)};
            return true;
        }
        if (type == "checker" || type == "checker_solid") {
            bool solid = type == "checker_solid";
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Checkerboard texture missing parameters"; return false; }
            double scale = 1.0;
            if (next(&c) && !parse_f64(c, &scale)) { *err = "invalid float literal"; return false; }
            int t1, t2;
            bool is_color;
            if (!get_texture(a, &t1, &is_color, err)) return false;
            if (is_color) { if (!get_color_tex(bb, &t2, err)) return false; }
            else { if (!get_float_tex(bb, &t2, err)) return false; }
            *out = {is_color ? Kind::TexColor : Kind::TexFloat, b.tex_checker(t1, t2, scale, solid)};
            return true;
        }
        if (type == "lerp") {
            if (!next(&a) || !next(&bb) || !next(&c)) { *err = "ParseError: Interpolate texture missing parameters"; return false; }
            int t, t1, t2;
            bool is_color;
            if (!get_float_tex(c, &t, err)) return false;
            if (!get_texture(a, &t1, &is_color, err)) return false;
            if (is_color) { if (!get_color_tex(bb, &t2, err)) return false; }
            else { if (!get_float_tex(bb, &t2, err)) return false; }
            *out = {is_color ? Kind::TexColor : Kind::TexFloat, b.tex_lerp(t1, t2, t)};
            return true;
        }
        if (type == "channel") {
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Channel texture missing parameters"; return false; }
            int t;
            size_t ch;
            if (!get_color_tex(a, &t, err)) return false;
            if (!parse_usize(bb, &ch)) { *err = "invalid digit found in string"; return false; }
            *out = {Kind::TexFloat, b.tex_channel(t, uint32_t(ch))};
            return true;
        }
        if (type == "uv_debug") { *out = {Kind::TexColor, b.tex_uv_debug()}; return true; }
        if (type == "noise") { *err = "ParseError: Not implemented"; return false; }  // scene.rs:257
        if (type == "image") {  // scene.rs:542-553
            if (!next(&a)) { *err = "ParseError: Image texture missing parameters"; return false; }
            std::vector<float> rgb;
            uint32_t w = 0, h = 0;
            if (!load_image_rgb32f(asset_path + a, &rgb, &w, &h, err)) return false;
            *out = {Kind::TexColor, b.tex_image(std::move(rgb), w, h)};
            return true;
        }
        if (type == "noise_solid") {  // scene.rs:555-570
            if (!next(&a)) { *err = "ParseError: Noise texture missing parameters"; return false; }
            int gen;
            if (!get_entity(a, Kind::Noise, noises, "noise gen", &gen, err)) return false;
            double scale = 1.0;
            size_t samples = 7;
            if (next(&bb) && !parse_f64(bb, &scale)) { *err = "invalid float literal"; return false; }
            if (next(&c) && !parse_usize(c, &samples)) { *err = "invalid digit found in string"; return false; }
            *out = {Kind::TexFloat, b.tex_noise_solid(gen, scale, uint32_t(samples))};
            return true;
        }
        if (type == "perlin") { *out = {Kind::Noise, b.noise_perlin(rng)}; return true; }  // scene.rs:282

        // ---- materials ----
        if (type == "lambertian") {
            int t;
            if (!next(&a)) { *err = "ParseError: LambertianDiffuse material missing parameters"; return false; }
            if (!get_color_tex(a, &t, err)) return false;
            *out = {Kind::Material, b.mat_lambertian(t)};
            return true;
        }
        if (type == "metal") {
            int t1, t2;
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Metal material missing parameters"; return false; }
            if (!get_color_tex(a, &t1, err) || !get_float_tex(bb, &t2, err)) return false;
            *out = {Kind::Material, b.mat_metal(t1, t2)};
            return true;
        }
        if (type == "glass") {
            double ior = 1.5;
            if (next(&a) && !parse_f64(a, &ior)) { *err = "invalid float literal"; return false; }
            *out = {Kind::Material, b.mat_dielectric(ior)};
            return true;
        }
        if (type == "glossy") {
            int t1, t2, nm = -1;
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Glossy material missing parameters"; return false; }
            if (!get_color_tex(a, &t1, err) || !get_float_tex(bb, &t2, err)) return false;
            double ior = 1.5;
            if (next(&c) && !parse_f64(c, &ior)) { *err = "invalid float literal"; return false; }
            if (next(&d) && !get_color_tex(d, &nm, err)) return false;
            *out = {Kind::Material, b.mat_glossy(t1, t2, ior, nm)};
            return true;
        }
        if (type == "emissive") {
            int t;
            if (!next(&a)) { *err = "ParseError: Emissive material missing parameters"; return false; }
            if (!get_color_tex(a, &t, err)) return false;
            *out = {Kind::Material, b.mat_emissive(t)};
            return true;
        }
        if (type == "isotropic") {
            int t;
            if (!next(&a)) { *err = "ParseError: Isotropic material missing parameters"; return false; }
            if (!get_color_tex(a, &t, err)) return false;
            *out = {Kind::Material, b.mat_isotropic(t)};
            return true;
        }
        if (type == "normal_debug") {
            int nm = -1;
            if (next(&a) && !get_color_tex(a, &nm, err)) return false;
            *out = {Kind::Material, b.mat_normal_debug(nm)};
            return true;
        }

        // ---- objects ----
        if (type == "sphere") {
            if (!next(&a) || !next(&bb) || !next(&c)) { *err = "ParseError: Sphere missing parameters"; return false; }
            double r;
            int m;
            if (!parse_vec3(a, v3, err)) return false;
            if (!parse_f64(bb, &r)) { *err = "invalid float literal"; return false; }
            if (!get_material(c, &m, err)) return false;
            *out = {Kind::Object, b.sphere(point(v3[0], v3[1], v3[2]), r, m)};
            return true;
        }
        if (type == "plane") {
            if (!next(&a) || !next(&bb) || !next(&c) || !next(&d)) { *err = "ParseError: Plane missing parameters"; return false; }
            double u3[3], w3[3];
            int m;
            if (!parse_vec3(a, v3, err) || !parse_vec3(bb, u3, err) || !parse_vec3(c, w3, err)) return false;
            if (!get_material(d, &m, err)) return false;
            std::string flag;
            bool backface = next(&flag) && flag == "backface";
            int id = b.plane(point(v3[0], v3[1], v3[2]), vec(u3[0], u3[1], u3[2]), vec(w3[0], w3[1], w3[2]), m, backface, err);
            if (id < 0) return false;
            *out = {Kind::Object, id};
            return true;
        }
        if (type == "box") {
            if (!next(&a) || !next(&bb) || !next(&c)) { *err = "ParseError: Box missing parameters"; return false; }
            double s3[3];
            int m;
            if (!parse_vec3(a, v3, err) || !parse_vec3(bb, s3, err)) return false;
            if (!get_material(c, &m, err)) return false;
            *out = {Kind::Object, b.box(point(v3[0], v3[1], v3[2]), point(s3[0], s3[1], s3[2]), m)};
            return true;
        }
        if (type == "mesh") {
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Mesh missing parameters"; return false; }
            // scene.rs:744-748: the file is opened before the material is resolved
            std::string path = asset_path + a;
            {
                FILE* f = std::fopen(path.c_str(), "rb");
                if (!f) { *err = "No such file or directory: " + path; return false; }
                std::fclose(f);
            }
            int m;
            if (!get_material(bb, &m, err)) return false;
            auto data = load_obj(path, log, err);
            if (!data) return false;
            *out = {Kind::Object, b.mesh(std::move(data), m)};
            return true;
        }
        if (type == "transform") {
            if (!next(&a)) { *err = "ParseError: Transform missing parameters"; return false; }
            int obj;
            if (!get_object(a, &obj, err)) return false;
            int t = b.transform_new(obj);
            std::string param;
            while (next(&param)) {
                std::string key, value;
                if (!split_key_value(param, &key, &value)) continue;  // regex does not match: ignored
                double x;
                if (key == "t") {
                    if (!parse_vec3(value, v3, err)) return false;
                    b.transform_translate(t, v3[0], v3[1], v3[2]);
                } else if (key == "s") {
                    // scene.rs:769-776: a vector if it parses as one, else a uniform factor
                    if (value.find(',') != std::string::npos) {
                        if (!parse_vec3(value, v3, err)) return false;
                        b.transform_scale(t, v3[0], v3[1], v3[2]);
                    } else {
                        if (!parse_f64(value, &x)) { *err = "invalid float literal"; return false; }
                        b.transform_scale(t, x, x, x);
                    }
                } else if (key == "rx" || key == "ry" || key == "rz") {
                    if (!parse_f64(value, &x)) { *err = "invalid float literal"; return false; }
                    double rad = x / 180.0 * 3.14159265358979323846;  // utils.rs:5-7 deg_to_rad
                    b.transform_rotate(t, key == "rx" ? 0 : key == "ry" ? 1 : 2, rad);
                }
            }
            *out = {Kind::Object, t};
            return true;
        }
        if (type == "list") {
            int list = b.list_new();
            std::string e;
            while (next(&e)) {
                int obj;
                if (!get_object(e, &obj, err)) return false;
                b.list_add(list, obj);
            }
            *out = {Kind::Object, list};
            return true;
        }
        if (type == "bvh") {
            if (!next(&a)) { *err = "ParseError: BVH missing parameters"; return false; }
            bool axes[3] = {a.find('x') != std::string::npos, a.find('y') != std::string::npos,
                            a.find('z') != std::string::npos};
            if (!axes[0] && !axes[1] && !axes[2]) { *err = "ParseError: BVH needs at least one axis"; return false; }  // reference loops forever
            std::vector<int> objs;
            std::string e;
            while (next(&e)) {
                int obj;
                if (!get_object(e, &obj, err)) return false;
                objs.push_back(obj);
            }
            if (objs.empty()) { *err = "ParseError: BVH needs at least one object"; return false; }  // reference overflows its stack
            *out = {Kind::Object, b.bvh(objs, axes, rng)};
            return true;
        }
        if (type == "sky") {
            int t;
            if (!next(&a)) { *err = "ParseError: Sky missing parameters"; return false; }
            if (!get_color_tex(a, &t, err)) return false;
            *out = {Kind::Object, b.sky(t)};
            return true;
        }
        if (type == "sun") {
            int t;
            if (!next(&a) || !next(&bb)) { *err = "ParseError: Sun missing parameters"; return false; }
            if (!parse_vec3(a, v3, err)) return false;
            if (!get_color_tex(bb, &t, err)) return false;
            *out = {Kind::Object, b.sun(t, point(v3[0], v3[1], v3[2]))};
            return true;
        }
        if (type == "volume") {
            if (!next(&a) || !next(&bb) || !next(&c)) { *err = "ParseError: Volume missing parameters"; return false; }
            int obj, m;
            double density;
            if (!get_object(a, &obj, err) || !get_material(bb, &m, err)) return false;
            if (!parse_f64(c, &density)) { *err = "invalid float literal"; return false; }
            *out = {Kind::Object, b.volume(obj, m, density)};
            return true;
        }
        *err = "ParseError: Unknown object type";
        return false;
    }

    // scene.rs:158-212
    bool parse_config_directive(const std::string& content, std::string* err) {
        size_t eq = content.find('=');
        if (eq == std::string::npos) { *err = "ParseError: @config " + content; return false; }
        std::string key = trim(content.substr(0, eq));
        std::string value = trim(content.substr(eq + 1));
        double d, v3[3];
        size_t u;
        if (key == "output_width") {
            if (!parse_usize(value, &u)) { *err = "invalid digit found in string"; return false; }
            scene_config.output_width = u;
        } else if (key == "aspect_ratio") {
            size_t slash = value.find('/');
            if (slash != std::string::npos) {
                double a, b2;
                if (!parse_f64(trim(value.substr(0, slash)), &a) || !parse_f64(trim(value.substr(slash + 1)), &b2)) {
                    *err = "invalid float literal";
                    return false;
                }
                d = a / b2;
            } else if (!parse_f64(value, &d)) { *err = "invalid float literal"; return false; }
            scene_config.aspect_ratio = d;
        } else if (key == "focal_length") {
            if (!parse_f64(value, &d)) { *err = "invalid float literal"; return false; }
            scene_config.focal_length = d;
        } else if (key == "f_number") {
            if (!parse_f64(value, &d)) { *err = "invalid float literal"; return false; }
            scene_config.f_number = d;
        } else if (key == "focus_distance") {
            if (!parse_f64(value, &d)) { *err = "invalid float literal"; return false; }
            scene_config.focus_distance = d;
        } else if (key == "camera_pos") {
            if (!parse_vec3(value, v3, err)) return false;
            scene_config.camera_pos = point(v3[0], v3[1], v3[2]);
        } else if (key == "camera_target") {
            if (!parse_vec3(value, v3, err)) return false;
            scene_config.camera_target = point(v3[0], v3[1], v3[2]);
        }
        return true;
    }
};

}  // namespace

bool load_dsl_scene(const std::string& file_path, const std::string& asset_path, SceneRng& rng,
                    LoadedScene* out, std::string* log, std::string* err) {
    std::ifstream in(file_path, std::ios::binary);
    if (!in) {
        *err = "No such file or directory: " + file_path;  // main.rs:43 `File::open(file_path)?`
        return false;
    }
    Loader L{out->builder, rng, asset_path, log, {}, {}, {}, {}, {}, default_scene_config()};
    std::string line;
    size_t line_number = 0;
    for (; std::getline(in, line); line_number++) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        if (line[0] == '@') {
            size_t sp = line.find(' ');
            if (sp != std::string::npos && line.substr(1, sp - 1) == "config") {
                std::string e;
                if (!L.parse_config_directive(line.substr(sp + 1), &e))
                    *log += "Warning: invalid @config directive\n\t" + e + "\n\n";
            }
            continue;
        }
        // scene.rs:104-105: split(":"), first two parts, trimmed
        size_t c1 = line.find(':');
        if (c1 == std::string::npos) {
            *log += "Warning: parse failed on line " + std::to_string(line_number) + ", skipped\n\t" + line + "\n\n";
            continue;
        }
        size_t c2 = line.find(':', c1 + 1);
        std::string label = Loader::trim(line.substr(0, c1));
        std::string decl = Loader::trim(line.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1));
        Entity ent;
        std::string e;
        if (!L.parse_declaration(decl, &ent, &e)) {
            *log += "Warning: error on line " + std::to_string(line_number) + ", skipped\n\t" + e + "\n\n";
            continue;
        }
        switch (ent.kind) {
            case Kind::Object: L.objects[label] = ent.id; break;
            case Kind::Material: L.materials[label] = ent.id; break;
            case Kind::TexColor: L.color_tex[label] = ent.id; break;
            case Kind::TexFloat: L.float_tex[label] = ent.id; break;
            case Kind::Noise: L.noises[label] = ent.id; break;
        }
    }
    auto w = L.objects.find("world"), l = L.objects.find("lights");
    if (w == L.objects.end() || l == L.objects.end()) {
        *err = "ParseError: No world/lights object";
        return false;
    }
    out->world = w->second;
    out->lights = l->second;
    out->scene_config = L.scene_config;
    return true;
}

}  // namespace rth
