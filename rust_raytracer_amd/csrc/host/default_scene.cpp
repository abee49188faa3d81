// The built-in default scene ("golden_monkey" or no scene argument) — behavioural
// restatement of reference src/scene/golden_monkey.rs:23-139: checker ground quad, Suzanne
// in brushed gold, a 21x21 field of small glossy/glass spheres in an object BVH over XZ,
// sky + sun as world members and as the light list.
// The reference draws the sphere field from an entropy-seeded generator (:83); here the
// draws come from the --seed keyed scene stream, in the same order.
#include "host_internal.h"

namespace rth {

bool load_default_scene(SceneRng& rng, LoadedScene* out, std::string* log, std::string* err) {
    SceneBuilder& b = out->builder;

    SceneConfig defaults;  // golden_monkey.rs:24-33
    defaults.output_width = 600;
    defaults.aspect_ratio = 1.5;
    defaults.focal_length = 50.0;
    defaults.f_number = 2.8;
    defaults.camera_pos = point(5.0, 2.0, 9.0);
    defaults.camera_target = point(0.0, 0.5, 0.0);
    out->scene_config = merge(default_scene_config(), defaults);
    // background: None in scene_defaults, so DEFAULT_SCENE_CONFIG's black stays (config.rs:28)

    int mat_ground = b.mat_lambertian(b.tex_checker(b.tex_const_color(0.2, 0.3, 0.1), b.tex_const_color(0.9, 0.9, 0.9), 0.02, false));
    int mat_metal = b.mat_metal(b.tex_const_color(0.8, 0.6, 0.2), b.tex_const_float(0.05));
    int mat_glass = b.mat_dielectric(1.5);

    int sky = b.sky(b.tex_const_color(0.2, 0.6, 2.0));
    int sun = b.sun(b.tex_const_color(20.0, 20.0, 20.0), vec(-1.0, 1.0, 0.0));

    int floor = b.plane(point(0, 0, 0), vec(20, 0, 0), vec(0, 0, -20), mat_ground, false, err);
    if (floor < 0) return false;

    auto mesh_data = load_obj("scenes/resource/monkey.obj", log, err);  // relative to CWD, golden_monkey.rs:77
    if (!mesh_data) return false;
    int mesh = b.transform_new(b.mesh(std::move(mesh_data), mat_metal));
    b.transform_translate(mesh, 0.0, 1.0, 0.0);

    std::vector<int> spheres;
    for (int i = -10; i < 11; i++) {
        for (int j = -10; j < 11; j++) {
            double x = double(i), z = double(j);
            double cx = x + rng.range(0.0, 0.9);
            double cz = z + rng.range(0.0, 0.9);
            V4 center = point(cx, 0.2, cz);
            if (length_squared(center - vec(0.0, 0.2, 0.0)) < 1.0) continue;
            double mat_type = rng.range(0.0, 1.0);
            if (mat_type < 0.95) {
                double a0 = rng.uniform(), a1 = rng.uniform(), a2 = rng.uniform();
                double b0 = rng.uniform(), b1 = rng.uniform(), b2 = rng.uniform();
                int material = b.mat_glossy(b.tex_const_color(a0 * b0, a1 * b1, a2 * b2), b.tex_const_float(0.1), 1.5, -1);
                spheres.push_back(b.sphere(center, 0.2, material));
            } else {
                spheres.push_back(b.sphere(center, 0.2, mat_glass));
                spheres.push_back(b.sphere(center, -0.18, mat_glass));
            }
        }
    }
    const bool axes_xz[3] = {true, false, true};  // bvh.rs:21 AXES_XZ
    int spheres_bvh = b.bvh(spheres, axes_xz, rng);

    int world = b.list_new();
    b.list_add(world, mesh);
    b.list_add(world, floor);
    b.list_add(world, spheres_bvh);
    b.list_add(world, sky);
    b.list_add(world, sun);

    int lights = b.list_new();
    b.list_add(lights, sky);
    b.list_add(lights, sun);

    out->world = world;
    out->lights = lights;
    return true;
}

}  // namespace rth
