"""Zero-weight path vertices (reference src/camera.rs:310-314).

The reference keeps tracing a path whose weight `att * s_pdf / pdf` is exactly 0 and multiplies what the
continuation returns by it: a later vertex with a 0/0 weight makes the sample NaN (0 * NaN).  The HIP path may end
such a path early only in scenes whose light set cannot produce an infinite or NaN weight; the scene compiler
decides that (rt_scene_info, host only).  These CPU tests pin (1) the classification and (2) that
tests/scenes/zero_weight_nan really contains NaN samples that are NaN ONLY through a zero-weight prefix — the
property the GPU parity test on that scene relies on."""
import numpy as np
import pytest

from oracle import pyoracle
from rust_raytracer_amd import api

ZW = api.RT_SCENE_INFO_ZERO_WEIGHT_STOP


def info(args):
    hs = api.HostScene(args)
    return api.scene_info(hs.desc)


def write(tmp_path, name, text):
    f = tmp_path / name
    f.write_text(text)
    return str(f)


BASE = ("@config output_width = 16\nfloor: plane 0,0,0 4,0,0 0,0,-4 (lambertian (constant 0.7,0.7,0.7))\n"
        "ball: sphere 0,1,0 1 (lambertian (constant 0.3,0.4,0.8))\n")


def test_reference_scenes_classification():
    # two-sided quad light (+ the glass ball, which does not scatter with a pdf and holds nothing inside)
    assert info(["scenes/cornell"]) & ZW
    assert info(["scenes/light_test"]) & ZW            # two emissive spheres
    assert info(["-w=40"]) & ZW                        # default scene: sun + sky
    assert not info(["tests/scenes/zero_weight_nan"]) & ZW   # one-sided quad light


def test_classification_rules(tmp_path):
    two_sided = BASE + "lamp: plane 0,3,0 -1,0,0 0,0,-1 (emissive (constant 9,9,9)) backface\nworld: list $floor $ball $lamp\nlights: list $lamp\n"
    one_sided = two_sided.replace(" backface", "")
    assert info([write(tmp_path, "a", two_sided)]) & ZW
    assert not info([write(tmp_path, "b", one_sided)]) & ZW
    # empty light list: ObjectList::random returns (1,0,0) with pdf 0 (list.rs:93-95)
    assert not info([write(tmp_path, "c", BASE + "world: list $floor $ball\nlights: list\n")]) & ZW
    # a light that scatters with a pdf itself (points ON it sample it from its own surface)
    lamb_light = BASE + "world: list $floor $ball\nlights: list $ball\n"
    assert not info([write(tmp_path, "d", lamb_light)]) & ZW
    # a diffuse object inside a sphere light's ball: Sphere::random takes sqrt(1 - r^2/d^2) of a negative number
    inside = (BASE + "glow: sphere 0,1,0 3 (glass 1.5)\nworld: list $floor $ball $glow\nlights: list $glow\n")
    assert not info([write(tmp_path, "e", inside)]) & ZW
    outside = (BASE + "glow: sphere 0,9,0 3 (emissive (constant 5,5,5))\nworld: list $floor $ball $glow\nlights: list $glow\n")
    assert info([write(tmp_path, "f", outside)]) & ZW
    # a mesh / transform as a light has pdf_value 0 (object.rs default)
    xf = BASE + "lamp: transform (sphere 0,5,0 1 (emissive (constant 5,5,5))) t=0,1,0\nworld: list $floor $ball $lamp\nlights: list $lamp\n"
    assert not info([write(tmp_path, "g", xf)]) & ZW


def test_scene_has_nan_samples_only_reachable_through_zero_weight_vertices():
    hs = api.HostScene(["tests/scenes/zero_weight_nan", "-w=32", "-s=16", "--seed=18"])
    S = hs.params.sqrt_spt
    black = {1, 2}            # material indices of `black` and `gloss` (albedo 0) in the scene file
    via_albedo = via_pdf = 0
    pixels_only_via_zero = 0
    for y in range(hs.height):
        for x in range(hs.width):
            direct = via_zero = False
            for st in range(S * S):
                rgb, tr = pyoracle.trace_sample(hs.desc, hs.camera, hs.params, 0, x, y, st % S, st // S)
                zero_a = zero_p = hit = False
                for k in range(len(tr)):
                    mat, kind, pdf, s_pdf = tr[k, 4:8]
                    if kind != 0:        # ScatterKind::WithPDF only
                        continue
                    if pdf == 0 and (zero_a or zero_p):
                        via_albedo += zero_a
                        via_pdf += zero_p
                        hit = True
                        break
                    if pdf > 0 and int(mat) in black:
                        zero_a = True
                    if pdf > 0 and s_pdf == 0:
                        zero_p = True
                if hit:
                    assert np.isnan(rgb).all()      # 0 * NaN: the reference's value for this sample
                    via_zero = True
                elif np.isnan(rgb).any():
                    direct = True
            pixels_only_via_zero += via_zero and not direct
    assert via_albedo >= 20 and via_pdf >= 1
    assert pixels_only_via_zero >= 20     # pixels an early stop on zero weight would get wrong
