"""The scene compiler's output for the round-3 features, without a GPU (rt_scene_program compiles like rt_scene_create and
returns the op program plus the wavefront scheduler's kernel plan): mesh-op lists, the two forms of a re-built primitive
group (OP_GROUP in front of its skip-pointer ops), nested volumes, nested light lists, and which scenes leave the fast kernels."""
import os

import numpy as np
import pytest

from rust_raytracer_amd import api

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rt_scene.h OpType
OP_END, OP_BOUNDS, OP_XFORM_PUSH, OP_XFORM_POP, OP_SPHERE, OP_PLANE, OP_MESH, OP_SKY, OP_SUN, OP_VOL_BEGIN, OP_VOL_MID, OP_VOL_END, OP_GROUP = range(13)


def program(args):
    hs = api.HostScene(args)
    return api.scene_program(hs.desc)


def test_default_scene_has_one_group_in_two_forms():
    ops, info = program(["-w=60", "-s=16", "--seed=5"])
    groups = np.flatnonzero(ops[:, 0] == OP_GROUP)
    assert len(groups) == 1 and info["groups"] == 1 and info["group_bvh"] and info["split"] and not info["multi_mesh"]
    g = int(groups[0])
    skip = int(ops[g, 2])
    assert g < skip <= len(ops) and ops[g + 1, 0] == OP_BOUNDS            # the op form follows the OP_GROUP ...
    inside = ops[g + 1:skip]
    assert set(np.unique(inside[:, 0])) <= {OP_BOUNDS, OP_SPHERE, OP_PLANE}  # ... and holds nothing but boxes and primitives,
    assert np.all(inside[inside[:, 0] == OP_BOUNDS, 2] <= skip)            # whose skip pointers stay inside it
    n_prims = int(np.isin(inside[:, 0], (OP_SPHERE, OP_PLANE)).sum())
    assert n_prims == info["group_prims"] >= 400                           # the 22 x 22 sphere field (entropy of --seed decides how many)
    outside = np.concatenate([ops[:g], ops[skip:]])
    assert int((outside[:, 0] == OP_SPHERE).sum()) <= 8                    # the few big spheres of the scene are not in the group
    ranks = inside[np.isin(inside[:, 0], (OP_SPHERE, OP_PLANE)), 2]
    assert len(set(ranks.tolist())) == n_prims                             # every primitive keeps its own rank (tie rule)
    assert info["group_nodes"] * 64 <= 24 * 1024 and info["group_stack"] <= 16
    assert ops[-1, 0] == OP_END


def test_group_form_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("RT_PRIM_REBUILD", "0")
    ops, info = program(["-w=60", "-s=16", "--seed=5"])
    assert info["groups"] == 0 and not info["group_bvh"] and not (ops[:, 0] == OP_GROUP).any()


def test_mesh_ops_and_kernel_plan():
    ops, info = program([os.path.join(REPO, "tests/scenes/two_meshes")])
    assert info["mesh_ops"] == 2 == int((ops[:, 0] == OP_MESH).sum()) and info["split"] and info["multi_mesh"] and not info["vol_prims"]
    ops, info = program([os.path.join(REPO, "scenes/light_test")])
    assert info["mesh_ops"] == 1 and info["split"] and not info["multi_mesh"]
    ops, info = program([os.path.join(REPO, "scenes/cornell")])
    assert info["mesh_ops"] == 0 and info["split"]


def test_volume_scenes_and_the_fast_path():
    # boxes as boundaries: the volumes run inside k_wf_prims
    ops, info = program([os.path.join(REPO, "scenes/cornell_smoke")])
    assert info["volumes"] == 2 and info["split"] and info["vol_prims"]
    # a mesh inside a boundary: the combined kernel
    ops, info = program([os.path.join(REPO, "tests/scenes/smoke")])
    assert info["volumes"] == 4 and not info["split"]
    # a volume inside a volume's boundary, then a mesh BEHIND the volumes: still the split kernels
    ops, info = program([os.path.join(REPO, "tests/scenes/nested_volumes")])
    assert info["split"] and info["vol_prims"] and info["mesh_ops"] == 1
    types = ops[:, 0]
    depth = max_depth = 0
    for t in types:
        if t == OP_VOL_BEGIN:
            depth += 1
            max_depth = max(max_depth, depth)
        elif t == OP_VOL_END:
            depth -= 1
    assert depth == 0 and max_depth == 2
    # the inner volume is compiled once per search of the outer boundary (entry search, exit search), each with its own two searches
    assert int((types == OP_VOL_BEGIN).sum()) == int((types == OP_VOL_END).sum()) == 2 + 2
    mids = np.flatnonzero(types == OP_VOL_MID)
    assert all(ops[m, 2] > m for m in mids)                                   # a missed volume jumps forward, behind its VOL_END
    last_vol = int(np.flatnonzero(types == OP_VOL_END).max())
    assert int(np.flatnonzero(types == OP_MESH)[0]) > last_vol


def test_nested_light_lists_flatten_to_a_tree():
    _, info = program([os.path.join(REPO, "tests/scenes/nested_lights")])
    # lights: list $middle $lamp_quad (list $inner $ball); middle = list $lamp_ball $inner (list $lamp_quad); inner = list $lamp_ball2 $lamp_box;
    # lamp_box = a list of six quads.  Entries: 3 (top) + 3 (middle) + 2 (inner) + 6 (box) + 1 + 2 (last top list) + 2 (inner again) + 6 (box again)
    assert info["lights"] == 3 + 3 + 2 + 6 + 1 + 2 + 2 + 6
