/*
 * rt_host.h — C ABI of the host side that stays *around* the render path
 * (librt_host.so, CPU only, no HIP): the reference's CLI flag parser, scene
 * DSL loader, OBJ loader, default scene, camera set-up and the output stage,
 * restated in C++ because no Rust toolchain exists in the build image.
 *
 *   rth_load        = reference src/main.rs:26-59  (Config::from_args + scene dispatch
 *                     + SceneLoader::load / GoldenMonkeyScene::init + Camera::new)
 *   rth_scene/...   = the flattened `(Camera, world, lights)` SceneData (src/scene.rs:30)
 *                     in the form rt_mi355.h takes
 *   rth_tonemap_rgb8, rth_save_png = Writer::save with tonemap_aces
 *                     (src/output.rs:23-49, src/tonemapping/aces.rs:27-33, main.rs:80-82)
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtHost RtHost;

/* argv[0] is skipped like env::args().skip(1) (config.rs:81).  Flags are the
 * reference's (README.md:21-43) plus --seed=<u64>, --gpus=<n>,
 * --precision=f64|f32, --pipeline=auto|mega|wavefront, --bvh=host|device (unknown keys are
 * ignored by the reference, config.rs:146, so these are compatible).
 * Relative scene/asset paths resolve against the current directory, as in
 * the reference (main.rs:43, golden_monkey.rs:77). */
int rth_load(int argc, const char* const* argv, RtHost** out);
void rth_destroy(RtHost* host);

const RtSceneDesc* rth_scene(const RtHost* host);
const RtCameraDesc* rth_camera(const RtHost* host);
const RtRenderParams* rth_params(const RtHost* host);
uint32_t rth_gpus(const RtHost* host);            /* --gpus, default 1 */
uint32_t rth_samples_per_pixel(const RtHost* host); /* Camera::samples_per_pixel() */
/* Row partition of `rtrace --gpus=N` (replaces the per-thread full-frame buffers of src/camera.rs:243-255): band height
 * for `height` image rows over `n_parts` GPUs = the largest of 16, 8, 4, 2, 1 rows that gives the most loaded part as few
 * rows as any of them does (the slowest GPU sets the time of the frame); 0 for n_parts <= 1.  The ONE definition of the
 * rule on the native side; rust_raytracer_amd/dist.py (band_rows_for) is its Python twin and a test compares the two. */
uint32_t rth_band_rows(uint32_t height, uint32_t n_parts);
/* Everything the reference would have printed while loading ("Loaded N tris",
 * loader warnings), newline separated. */
const char* rth_log(const RtHost* host);

/* Camera::new + init for explicit settings (used by tests of camera.rs:47-130).
 * f_number / focus_distance < 0 mean None. */
int rth_make_camera(uint32_t width, double aspect_ratio, double focal_length,
                    double f_number, double focus_distance,
                    const double position[3], const double look_at[3],
                    RtCameraDesc* out);

/* ACES fit + sRGB OETF + `(x * 255.999) as u8` on w*h RGBA f64 pixels -> w*h*3 bytes. */
int rth_tonemap_rgb8(const double* rgba, uint32_t w, uint32_t h, uint8_t* rgb_out);
/* The same followed by an 8-bit RGB PNG file (zlib deflate). */
int rth_save_png(const char* path, const double* rgba, uint32_t w, uint32_t h);

/* Buffer::from_image (src/buffer.rs:30-48): decodes a PNG or baseline-JPEG file into w*h RGB f32
 * triples (8-bit: x/255, 16-bit: x/65535), row 0 first.  The caller frees with rth_free_image. */
int rth_load_image(const char* path, float** rgb_out, uint32_t* w_out, uint32_t* h_out);
void rth_free_image(float* rgb);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
