"""Parity of the HIP path against the CPU oracle, through the C-ABI.

Bar: the primary ray's hit id equal on >= 99.9 % of pixels and 8-bit RGB
within 1/255.  The kernels use the oracle's fp32 operation order without
contraction, so in practice both are exact; the exact fractions are asserted
where that has been observed to hold.
"""
import os

import numpy as np
import pytest

from oracle import orc
from oracle.gen_golden import image_scenes
from raytracing_amd import scenes
from raytracing_amd import types as T
from raytracing_amd.pipe import (RT_HIP_HITIDS, RT_HIP_STATS, RT_PARAM_COOP_LIVE, RT_PARAM_COOP_STACK, RT_PARAM_GPU_BUILD,
                                 RT_PARAM_MAX_DEPTH, RT_PARAM_MAX_PRIMS_IN_NODE, RT_PARAM_STACK_SIZE, RT_PARAM_TREE_CACHE,
                                 FlatTree, HipRenderer, RenderPipe)
from tests import known_answers
from tests.helpers import GOLD, random_soup

pytestmark = pytest.mark.gpu

IMG = image_scenes()


def hip_render(sc, depth=10, prims=7, flags=RT_HIP_HITIDS | RT_HIP_STATS, tile=None, params=(), prune=0.0):
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, depth, prims, prune_leaves=prune)
    r = HipRenderer(0)
    r.load_scene(sc, tree)
    if tile:
        r.set_tile(*tile)
    for key, value in params:
        r.set_param(key, value)
    r.render(flags)
    img, ids, st = r.read_frame(), r.read_hit_ids(), r.stats()
    r.close()
    return img, ids, st, tree


def assert_parity(img, ids, oimg, oid, mask=None):
    if mask is None:
        mask = np.ones(ids.shape, bool)
    same = (ids == oid)[mask].mean()
    assert same >= 0.999, f"hit ids agree on {same:.5f}"
    d = np.abs(img.astype(int) - oimg.astype(int))[mask]
    assert d.max() <= 1, f"rgb differs by {d.max()}"
    return same, (d.max(axis=-1) == 0).mean()


@pytest.mark.parametrize("case", sorted(known_answers.all_cases()))
@pytest.mark.parametrize("depth,prims", [(10, 7), (10, 1)])
def test_hand_derived_shading_cases_on_the_gpu(case, depth, prims):
    """Centre-pixel colours worked out by hand from the reference's formulas
    (tests/known_answers.py), compared with the HIP path directly - no oracle
    in between - and then the whole frame with the oracle."""
    sc, want = known_answers.expected(case, prims)
    img, ids, _, tree = hip_render(sc, depth, prims)
    c = known_answers.CENTRE
    assert (case.startswith("sphere") or ids[c, c] == 0) and img[c, c].tolist() == want
    nodes, idx = tree.expand()
    oimg, oid, _ = orc.raytrace(sc, tree.bbox, nodes, idx, threads=2)
    assert np.array_equal(ids, oid) and np.array_equal(img, oimg)
    tree.release()


@pytest.mark.parametrize("kind", ["mirror", "refraction"])
def test_hand_derived_oblique_child_rays_on_the_gpu(kind):
    sc, want, _ = known_answers.oblique_case(kind)
    img, ids, _, tree = hip_render(sc)
    tree.release()
    c = known_answers.CENTRE
    assert ids[c, c] == 0
    assert np.abs(img[c, c].astype(int) - np.array(want)).max() <= 1     # expectation worked out in double precision


def test_hand_derived_camera_rays_on_the_gpu():
    """Pixel -> ray through a turned and moved camera: every target sits where
    the reference's ray generation sends that one pixel."""
    sc, want = known_answers.camera_case()
    _, ids, _, tree = hip_render(sc, 10, 2)
    tree.release()
    for (x, y), k in want.items():
        assert ids[y, x] == k
    assert (ids >= 0).sum() == len(want)


def test_hand_derived_refraction_falloff_on_the_gpu():
    sc, want = known_answers.refraction_case(falloff=0.05)
    img, _, _, tree = hip_render(sc)
    tree.release()
    c = known_answers.CENTRE
    assert np.abs(img[c, c].astype(int) - np.array(want)).max() <= 1     # the distance between the two hit points carries float32 rounding the hand calculation ignores


@pytest.mark.parametrize("name", sorted(IMG))
def test_fixture_scenes_match_committed_vectors(name):
    sc, depth, prims = IMG[name]
    z = np.load(os.path.join(GOLD, f"orc_{name}.npz"))
    img, ids, st, _ = hip_render(sc, depth, prims)
    same, exact = assert_parity(img, ids, z["img"], z["hit"])
    assert same == 1.0 and exact == 1.0
    # same rays and hits as the restatement.  The work differs both ways: queries skip the cells
    # behind the ray origin and beyond the light and stop at full opacity; rays walked
    # cooperatively scan the leaves behind the one that ends them.  Bounded, not equal.
    zs = dict(zip(("raysClosest", "raysShadow", "nodeVisits", "triTests", "hits"), z["stats"].tolist()))
    assert [st[k] for k in ("raysClosest", "raysShadow", "hits")] == [zs[k] for k in ("raysClosest", "raysShadow", "hits")]
    assert st["nodeVisits"] <= 1.5 * zs["nodeVisits"] and st["triTests"] <= 1.5 * zs["triTests"]


@pytest.mark.parametrize("seed,depth", [(1, 10), (2, 18), (3, 26)])
def test_random_soup_with_spheres_and_two_lights(seed, depth):
    rng = np.random.RandomState(seed)
    sc = scenes.Scene(f"soup{seed}", 160, 120, scenes.make_camera(160, 120, hv=0.9))
    v, t = random_soup(4000, seed)
    sc.meshes.append(scenes.Mesh(v, t, 0))
    sc.flatten = lambda: (v, t)                # keep per-triangle materials
    for k in range(3):
        a = float(rng.choice([1.0, 0.6]))
        sc.materials[k] = T.make_material((a, *rng.uniform(0.2, 1, 3)), (0.0, 0.05, 0.05, 0.05),
                                          (0.0, 0.6, 0.6, 0.6), (float(rng.uniform(2, 30)), 0.4, 0.4, 0.4),
                                          (0.0, *([float(rng.choice([0.0, 0.4]))] * 3)), 0.02, float(rng.choice([1.0, 1.4])))
    sc.spheres += [scenes.make_sphere(rng.uniform(-20, 20, 3) + (0, 0, 60), float(rng.uniform(3, 9)), int(rng.randint(3)))
                   for _ in range(4)]
    sc.lights += [scenes.make_light((-20.0, -35.0, 0.0), 100.0), scenes.make_light((30.0, 10.0, 20.0), 60.0, (1, .9, .7, .5))]
    img, ids, st, tree = hip_render(sc, depth, 5)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact >= 0.999
    assert st["raysClosest"] == ost["raysClosest"] and st["raysShadow"] == ost["raysShadow"]
    # traversal work of the same order as the restatement's (cells behind the ray origin and
    # beyond the light are skipped; cooperative walks scan some leaves in vain)
    assert st["triTests"] - st["triTestsShadow"] <= 2 * (ost["triTests"] - ost["triTestsShadow"])
    assert st["nodeVisits"] - st["nodeVisitsShadow"] <= 2 * (ost["nodeVisits"] - ost["nodeVisitsShadow"])
    assert st["triTestsShadow"] <= 2 * ost["triTestsShadow"] and st["nodeVisitsShadow"] <= 2 * ost["nodeVisitsShadow"]


def _coop_scene(kind):
    if kind == "glass":
        return IMG["sphere16k_glass"]
    if kind == "wavy_deep":
        return IMG["wavy32_deep"]
    if kind == "default":
        return IMG["default_small"]
    if kind == "soup_negative_alpha":
        return _soup_scene(11, 3000, (1.0, -0.5, 0.7)), 14, 6
    if kind == "big_leaves":
        return _soup_scene(12, 700, (1.0, 0.6, 1.0), w=96, h=64), 3, 7
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["glass", "wavy_deep", "default", "soup_negative_alpha", "big_leaves"])
def test_cooperative_walk_gives_the_lane_per_ray_image(kind):
    """A wave whose rays have thinned out lets the remaining rays share its lanes
    (several leaves of a ray per round, sub-trees descended side by side).  Where
    that sets in (RT_PARAM_COOP_LIVE: never, at 4 rays, at 32) and how much stack
    it may use (RT_PARAM_COOP_STACK: 12 entries, 2, none - then every round
    restarts the walk at the root) must not change a bit of the result, closest
    hits and shadow sums (with negative opacities: no early exit) alike."""
    sc, depth, prims = _coop_scene(kind)
    want = None
    for live, stack in [(0, 12), (4, 12), (32, 12), (32, 2), (32, 0), (9, 1)]:
        img, ids, st, tree = hip_render(sc, depth, prims, params=((RT_PARAM_COOP_LIVE, live), (RT_PARAM_COOP_STACK, stack)))
        if want is None:
            nodes, idx = tree.expand()
            oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
            assert np.array_equal(ids, oid) and np.abs(img.astype(int) - oimg.astype(int)).max() <= 1
            want = (img, ids, st)
        assert np.array_equal(img, want[0]) and np.array_equal(ids, want[1]), (live, stack)
        assert [st[k] for k in ("raysClosest", "raysShadow", "hits")] == [want[2][k] for k in ("raysClosest", "raysShadow", "hits")]
        tree.release()


def test_short_ray_lists_are_spread_over_the_chip():
    """Lists too short to give every resident wave 64 rays are dealt out in
    smaller chunks and walked cooperatively from the first round: a 1080p frame
    of a mirror-in-mirror scene has such lists in its deep slots.  Lattice check
    against the oracle and equality with the lane-per-ray walk."""
    sc = scenes.sphere_scene(250, 200, 1920, 1080, reflect=0.5, stack_size=6)
    a = hip_render(sc, 22, 8, flags=RT_HIP_HITIDS)
    b = hip_render(sc, 22, 8, flags=RT_HIP_HITIDS, params=((RT_PARAM_COOP_LIVE, 0),))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2]["slotRays"] == b[2]["slotRays"] and 0 < a[2]["slotRays"][5] < 200000
    nodes, idx = a[3].expand()
    oimg, oid, _ = orc.raytrace(sc, a[3].bbox, nodes, idx, x0=3, x_step=8, y0=1, y_step=8)
    m = np.zeros(a[1].shape, bool)
    m[1::8, 3::8] = True
    same, exact = assert_parity(a[0], a[1], oimg, oid, m)
    assert same == 1.0 and exact == 1.0


def test_render_pipe_api_matches_and_consumes_the_scene():
    sc, depth, prims = IMG["default_small"]
    z = np.load(os.path.join(GOLD, "orc_default_small.npz"))
    rp = RenderPipe(sc.w, sc.h)
    rp.set_camera(sc.camera)
    rp.add_scene(sc)
    assert rp.counts == dict(triangles=sc.n_tris, vertices=sum(len(m.verts) for m in sc.meshes), prims=10, lights=1)
    frame = rp.draw()
    assert np.array_equal(frame, z["img"])
    assert rp.counts == dict(triangles=0, vertices=0, prims=0, lights=0)    # geometry is consumed by draw
    # materials and the camera persist: re-add everything but the materials
    for l in sc.lights:
        rp.add_light(l)
    for s in sc.spheres:
        rp.add_primitive(s)
    for m in sc.meshes:
        rp.mesh_add(m)
    assert np.array_equal(rp.draw(), z["img"])
    # tree cache on: same geometry again, then changed geometry
    rp.set_param(RT_PARAM_TREE_CACHE, 1)
    for rep in range(2):
        rp.add_scene(sc)
        assert np.array_equal(rp.draw(), z["img"])
    sc2 = scenes.default_scene(320, 192, t=0.4)
    rp.add_scene(sc2)
    v, t = sc2.flatten()
    o2 = orc.raytrace(sc2, *orc.kdtree_build(v, t))[0]
    assert np.array_equal(rp.draw(), o2)
    rp.cleanup()


def test_tree_cache_notices_every_kind_of_change():
    """RT_PARAM_TREE_CACHE keeps the tree while the geometry handed to the pipe is the same, word for word; its change
    detector is fed while the triangles are added.  Every frame of a pipe with the cache on equals the frame of a
    fresh pipe without it: unchanged geometry, some vertices moved, back again, one index
    changed, the material slot changed, the same mesh added in two calls, an odd number of triangles, and the cache
    switched on when the geometry had already been added."""
    import copy
    base = scenes.sphere_scene(40, 30, 200, 120, reflect=0.4)
    base.materials[1] = T.make_material((1.0, 0.2, 0.7, 0.3), (0.0, 0.05, 0.05, 0.05), (0.0, 0.6, 0.6, 0.6), (20.0, 0.3, 0.3, 0.3),
                                        (0.0, 0.0, 0.0, 0.0), 0.05, 1.0)

    def variant(kind):
        sc = copy.deepcopy(base)
        m = sc.meshes[0]
        if kind == "vertex":
            m.verts["pos"]["y"][600:1000] -= 4.0
        elif kind == "index":
            m.tris["pV2"][1000:1400] = m.tris["pV2"][0]
        elif kind == "group":
            m.group = 1
        elif kind == "odd":
            m.tris = m.tris[:-1].copy()
        return sc

    def add(rp, sc, two_calls=False):
        for k in sorted(sc.materials):
            rp.add_material(sc.materials[k], k)
        for l in sc.lights:
            rp.add_light(l)
        for m in sc.meshes:
            if two_calls:
                half = len(m.tris) // 2 + 1
                rp.add_triangles(m.verts, m.tris, 0, 0, len(m.verts), half, m.group)
                rp.add_triangles(m.verts, m.tris, 0, half, len(m.verts), len(m.tris) - half, m.group)
            else:
                rp.mesh_add(m)

    def fresh(sc, two_calls=False):
        rp = RenderPipe(sc.w, sc.h)
        rp.set_camera(sc.camera)
        add(rp, sc, two_calls)
        img = rp.draw().copy()
        rp.cleanup()
        return img

    rp = RenderPipe(base.w, base.h)
    rp.set_camera(base.camera)
    rp.set_param(RT_PARAM_TREE_CACHE, 1)
    seen = []
    for kind, two in (("base", False), ("base", False), ("vertex", False), ("base", False), ("index", False), ("group", False),
                      ("base", True), ("base", True), ("odd", False), ("odd", False), ("base", False)):
        sc = variant(kind)
        add(rp, sc, two)
        img = rp.draw().copy()
        assert np.array_equal(img, fresh(sc, two)), (kind, two)
        seen.append(img)
    assert not np.array_equal(seen[0], seen[2]) and not np.array_equal(seen[0], seen[4]) and not np.array_equal(seen[0], seen[5])
    rp.cleanup()
    # switched on with the geometry already there, and off again
    rp = RenderPipe(base.w, base.h)
    rp.set_camera(base.camera)
    for kind in ("base", "vertex", "vertex"):
        sc = variant(kind)
        add(rp, sc)
        rp.set_param(RT_PARAM_TREE_CACHE, 1)
        assert np.array_equal(rp.draw(), fresh(sc)), kind
        rp.set_param(RT_PARAM_TREE_CACHE, 0)
    rp.cleanup()


def test_geometry_that_goes_to_the_device_while_it_is_added():
    """From 4 MiB of geometry on, the pipe hands what has been added to the GPU builder piece by piece
    (rt_hip_stage_geometry), when the tree is going to be built there.  Same frame as with the host builder, which
    reads the host arrays: a mesh added in three calls, the next frame with other vertices, and a frame whose
    builder is changed between adding and drawing (what went ahead is dropped)."""
    base = scenes.sphere_scene(8, 6, 120, 90, reflect=0.3)
    v, t = random_soup(120_000, 31, extent=25.0, size=1.5)
    assert v.nbytes + t.nbytes > 3 * (4 << 20)

    def add(rp, verts):
        for k in sorted(base.materials):
            rp.add_material(base.materials[k], k)
        for l in base.lights:
            rp.add_light(l)
        # (a piece's worth of vertices that nothing refers to in front, triangles in three calls over the same vertices)
        rp.add_triangles(verts, t, 0, 0, 150_000, 0, 0)
        for first, count, group in ((0, 50_001, 0), (50_001, 49_999, 1), (100_000, 20_000, 0)):
            rp.add_triangles(verts, t, 0, first, len(verts), count, group)

    def pipe(gpu_build):
        rp = RenderPipe(base.w, base.h)
        rp.set_camera(base.camera)
        rp.set_param(RT_PARAM_MAX_DEPTH, 16)
        rp.set_param(RT_PARAM_GPU_BUILD, gpu_build)
        return rp

    host, dev = pipe(0), pipe(1)
    v2 = v.copy()
    v2["pos"]["x"] += 3.0
    frames = []
    for verts in (v, v2, v):
        add(host, verts)
        add(dev, verts)
        frames.append(host.draw().copy())
        assert np.array_equal(dev.draw(), frames[-1])
    assert not np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    add(dev, v2)
    dev.set_param(RT_PARAM_GPU_BUILD, 0)
    assert np.array_equal(dev.draw(), frames[1])
    dev.set_param(RT_PARAM_GPU_BUILD, 1)
    add(dev, v)
    assert np.array_equal(dev.draw(), frames[0])
    # two pipes of one thread that take turns, call by call: each loses what went ahead to the other (the builder's
    # input buffers belong to the thread), stops sending ahead for the frame, and its draw copies everything
    dev2 = pipe(1)
    for rp in (dev, dev2):
        for k in sorted(base.materials):
            rp.add_material(base.materials[k], k)
        for l in base.lights:
            rp.add_light(l)
    for first, count, group in ((0, 0, 0), (0, 50_001, 0), (50_001, 49_999, 1), (100_000, 20_000, 0)):
        for rp, verts in ((dev, v), (dev2, v2)):
            rp.add_triangles(verts, t, 0, first, 150_000 if count == 0 else len(verts), count, group)
    assert np.array_equal(dev2.draw(), frames[1]) and np.array_equal(dev.draw(), frames[0])
    host.cleanup()
    dev.cleanup()
    dev2.cleanup()


def test_staged_geometry_through_the_hip_layer():
    """rt_hip_stage_geometry + rt_hip_build_tree called directly (what the render pipe does underneath): a frame from
    geometry that went ahead in two pieces, a frame after geometry that went ahead was dropped for other arrays
    (rt_hip_unstage_geometry), and a frame into page-locked host memory - all equal to the frames from uploaded host trees."""
    import torch
    sc = scenes.sphere_scene(60, 40, 160, 120, reflect=0.4)
    v, t = sc.flatten()
    v2 = v.copy()
    v2["pos"]["y"] += 2.0

    def reference(verts):
        r = HipRenderer(0)
        tree = r.load_scene(sc, FlatTree.build(verts, t, 14, 6))
        r.render(0)
        img = r.read_frame().copy()
        tree.release()
        r.close()
        return img

    want, want2 = reference(v), reference(v2)
    assert not np.array_equal(want, want2)
    r = HipRenderer(0)
    r.upload_materials(sc.material_table())
    r.upload_spheres(sc.sphere_table())
    r.upload_lights(sc.light_table())
    r.set_camera(sc.camera)
    r.set_image(sc.w, sc.h)
    r.set_param(RT_PARAM_STACK_SIZE, sc.stack_size)
    r.stage_geometry(v, len(v) // 3, t, 0)
    r.stage_geometry(v, len(v), t, len(t) // 2)
    r.build_tree(v, t, 14, 6)
    r.render(0)
    assert np.array_equal(r.read_frame(), want)
    r.stage_geometry(v, len(v), t, len(t))
    r.unstage_geometry()
    r.build_tree(v2, t, 14, 6)
    host = torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8).pin_memory()
    r.render(0, d_out=host.data_ptr())
    r.sync()
    assert np.array_equal(host.numpy(), want2)
    r.close()


def test_render_pipe_kdtree_build_and_params():
    sc, _, _ = IMG["sphere16k_reflect"]
    rp = RenderPipe(sc.w, sc.h)
    rp.set_camera(sc.camera)
    rp.add_scene(sc)
    v, t = sc.flatten()
    bbox, nodes, idx = rp.kdtree_build()
    ob, on, oi = orc.kdtree_build(v, t)
    assert bbox.tobytes() == ob.tobytes() and np.array_equal(idx, oi) and np.array_equal(nodes["sep"], on["sep"])
    a = rp.draw()
    rp.set_param(RT_PARAM_MAX_DEPTH, 22)
    rp.set_param(RT_PARAM_MAX_PRIMS_IN_NODE, 4)
    rp.add_scene(sc)
    b = rp.draw()
    assert (np.abs(a.astype(int) - b.astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    rp.set_param(RT_PARAM_STACK_SIZE, 1)       # primary rays only
    rp.add_scene(sc)
    c = rp.draw()
    sc1 = scenes.sphere_scene(100, 80, 160, 120, reflect=0.5, stack_size=1)
    assert np.array_equal(c, orc.raytrace(sc1, *orc.kdtree_build(v, t, 22, 4))[0])
    rp.cleanup()


def test_deep_tree_gives_the_depth10_image():
    sc, _, _ = IMG["sphere16k_glass"]
    a = hip_render(sc, 10, 7)
    b = hip_render(sc, 30, 3)
    assert (a[1] == b[1]).mean() >= 0.999
    assert (np.abs(a[0].astype(int) - b[0].astype(int)).max(axis=-1) <= 1).mean() >= 0.999
    assert b[3].depth > 16 and b[2]["maxStack"] > 0


@pytest.mark.parametrize("band,parts", [(16, 2), (16, 3), (8, 8), (5, 4)])
def test_band_split_assembles_to_the_full_frame(band, parts):
    from raytracing_amd.dist import band_rows
    sc, depth, prims = IMG["default_small"]
    z = np.load(os.path.join(GOLD, "orc_default_small.npz"))
    full = np.zeros_like(z["img"])
    ids = np.zeros_like(z["hit"])
    rays = 0
    for p in range(parts):
        img, hid, st, _ = hip_render(sc, depth, prims, tile=(band, parts, p))
        rows = band_rows(sc.h, band, parts, p)
        assert img.shape[0] == len(rows)
        full[rows], ids[rows] = img, hid
        rays += st["raysClosest"]
    assert np.array_equal(full, z["img"]) and np.array_equal(ids, z["hit"])
    assert rays == int(z["stats"][0])


def edge_scene(kind):
    sc = scenes.Scene(kind, 64, 48, scenes.make_camera(64, 48, hv=0.8))
    sc.materials[0] = T.make_material((1.0, 0.8, 0.6, 0.4), (0.0, 0.1, 0.1, 0.1), (0.0, 0.7, 0.7, 0.7),
                                      (8.0, 0.3, 0.3, 0.3), (0.0, 0.3, 0.3, 0.3), 0.0, 1.0)
    if kind != "no_lights":
        sc.lights.append(scenes.make_light((-5.0, -10.0, 0.0), 30.0))
    if kind in ("spheres_only", "no_lights", "inside_box"):
        sc.spheres += [scenes.make_sphere((0.0, 0.0, 20.0), 4.0, 0), scenes.make_sphere((5.0, 2.0, 14.0), 1.5, 0)]
    if kind in ("few_tris", "no_lights", "inside_box"):
        v, t = random_soup(5, 9, extent=4.0, size=5.0, center=(0, 0, 18))
        t["mat"] = 0
        sc.meshes.append(scenes.Mesh(v, t, 0))
    if kind == "inside_box":
        v, t = random_soup(300, 10, extent=25.0, size=4.0, center=(0, 0, 0))   # camera sits inside the scene box
        t["mat"] = 0
        sc.meshes.append(scenes.Mesh(v, t, 0))
    return sc


@pytest.mark.parametrize("kind", ["empty", "spheres_only", "few_tris", "no_lights", "inside_box"])
def test_edge_scenes(kind):
    sc = edge_scene(kind)
    img, ids, st, tree = hip_render(sc)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact == 1.0
    if kind == "empty":
        assert img.max() == 0 and (ids == -1).all()


def test_errors_are_reported_not_swallowed():
    from raytracing_amd._lib import RtError
    r = HipRenderer(0)
    with pytest.raises(RtError):
        r.render()                                   # nothing uploaded
    sc = edge_scene("few_tris")
    v, t = sc.flatten()
    t = t.copy()
    t["mat"] = 7                                     # beyond the one-entry material table
    tree = FlatTree.build(v, t)
    r.load_scene(sc, tree)
    with pytest.raises(RtError):
        r.render()
    with pytest.raises(RtError):
        r.set_param(RT_PARAM_STACK_SIZE, 99)
    r.close()
    with pytest.raises(RtError):
        HipRenderer(99)


def test_malformed_trees_are_rejected_on_the_host():
    """A link or record range outside the arrays, or a tree deeper than the
    traversal stack (whatever depth the caller declares), must never reach
    the GPU."""
    import types
    from raytracing_amd._lib import RtError
    from raytracing_amd.pipe import kdnode
    sc = edge_scene("few_tris")
    v, t = sc.flatten()
    good = FlatTree.build(v, t)
    r = HipRenderer(0)

    def variant(nodes, depth=None):
        return types.SimpleNamespace(nodes=np.ascontiguousarray(nodes, kdnode), recs=good.recs, shade=good.shade,
                                     bbox=good.bbox, depth=good.depth if depth is None else depth)

    r.upload_tree(variant(good.nodes.copy()))                      # the unmodified copy is accepted
    leaf = int(np.flatnonzero((good.nodes["b"] & 3) == 3)[0])
    bad = good.nodes.copy()
    bad["b"][leaf] = ((len(good.recs) + 1) << 2) | 3               # leaf range beyond the records
    with pytest.raises(RtError):
        r.upload_tree(variant(bad))
    inner = np.zeros(3, kdnode)
    inner["b"][0] = (7 << 2) | 0                                   # right child beyond the array
    inner["b"][1:] = 3
    with pytest.raises(RtError):
        r.upload_tree(variant(inner))
    # a chain of 70 inner nodes, declared as depth 10
    n = 70
    chain = np.zeros(2 * n + 1, kdnode)
    for i in range(n):
        chain["b"][i] = ((2 * n - i) << 2) | 0                     # left = i + 1, right = a leaf at the end
    chain["b"][n:] = 3                                             # empty leaves
    with pytest.raises(RtError, match="deeper"):
        r.upload_tree(variant(chain, depth=10))
    r.close()


def test_config2_full_size_sampled_against_oracle():
    """100k triangles at 1920x1080 on the deep tree: every 12th pixel of every
    12th row against the oracle, plus image-wide invariants."""
    sc = scenes.config2()
    img, ids, st, tree = hip_render(sc, 28, 7)
    nodes, idx = tree.expand()
    oimg, oid, _ = orc.raytrace(sc, tree.bbox, nodes, idx, x0=5, x_step=12, y0=7, y_step=12)
    m = np.zeros(ids.shape, bool)
    m[7::12, 5::12] = True
    same, exact = assert_parity(img, ids, oimg, oid, m)
    assert same == 1.0
    assert st["raysClosest"] == sc.w * sc.h            # reflect = 0: primary rays only
    assert st["raysShadow"] == st["hits"] == int((ids != -1).sum())
    assert img[..., 3].max() == 0 and (img[ids == -1] == 0).all()


@pytest.mark.parametrize("depth,prims,flags", [(32, 7, RT_HIP_HITIDS | RT_HIP_STATS), (26, 12, RT_HIP_HITIDS)])
def test_config3_4k_sampled_against_oracle(depth, prims, flags):
    """The benchmark workload itself (1M triangles, 3840x2160, 5 rays per pixel):
    a 1-in-24 pixel lattice against the oracle, and the ray-count identities.
    The second case is exactly what bench.py times: its tree limits (depth 26,
    leaves of at most 12) and the kernels without the counting code, which for
    frames of this size are a separately compiled build."""
    sc = scenes.config3()
    img, ids, st, tree = hip_render(sc, depth, prims, flags=flags)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx, x0=11, x_step=24, y0=3, y_step=24)
    m = np.zeros(ids.shape, bool)
    m[3::24, 11::24] = True
    same, exact = assert_parity(img, ids, oimg, oid, m)
    assert same == 1.0 and exact == 1.0
    assert sc.w * sc.h <= st["raysClosest"] <= 5 * sc.w * sc.h
    assert st["raysShadow"] == st["hits"] <= st["raysClosest"]
    tree.release()


@pytest.mark.parametrize("which", ["config3", "config5", "glass", "default"])
def test_pruned_leaves_give_the_frame_of_the_reference_leaves(which):
    """bench.py renders a tree whose leaves hold only the triangles that reach
    their cell (pruneLeaves = 1e-5, same nodes): every pixel and every primary
    hit id must be what the reference's leaf lists give, at full size; for the
    benchmark frame additionally a lattice against the oracle walking the
    REFERENCE leaves."""
    sc, depth, prims = {"config3": (scenes.config3(), 26, 12), "config5": (scenes.config5_frame(101), 26, 12),
                        "glass": IMG["sphere16k_glass"][:1] + (20, 3), "default": (scenes.default_scene(), 18, 6)}[which]
    a = hip_render(sc, depth, prims, flags=RT_HIP_HITIDS)
    b = hip_render(sc, depth, prims, flags=RT_HIP_HITIDS, prune=1e-5)           # (the kernels bench.py times: no counting code)
    assert len(b[3].recs) < len(a[3].recs) and len(b[3].nodes) == len(a[3].nodes)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if which == "config3":
        nodes, idx = a[3].expand()
        oimg, oid, ost = orc.raytrace(sc, a[3].bbox, nodes, idx, x0=5, x_step=24, y0=17, y_step=24)
        m = np.zeros(b[1].shape, bool)
        m[17::24, 5::24] = True
        same, exact = assert_parity(b[0], b[1], oimg, oid, m)
        assert same == 1.0 and exact == 1.0


def test_config5_4k_frame_sampled_against_oracle():
    """A fly-through frame at full size: 1M triangles, 3840x2160, reflective and
    refractive material, six ray slots filled breadth first, on the benchmark's
    tree limits and the kernels without counting code; 1-in-24 lattice."""
    sc = scenes.config5_frame(37)
    img, ids, st, tree = hip_render(sc, 26, 12, flags=RT_HIP_HITIDS)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx, x0=7, x_step=24, y0=13, y_step=24)
    m = np.zeros(ids.shape, bool)
    m[13::24, 7::24] = True
    same, exact = assert_parity(img, ids, oimg, oid, m)
    assert same == 1.0 and exact == 1.0
    assert sc.w * sc.h < st["raysClosest"] <= 6 * sc.w * sc.h and st["raysShadow"] == st["hits"]
    assert st["slotRays"][5] > 0                       # the budget of six is really used
    tree.release()


def test_config4_shares_of_the_benchmark_frame_assemble_to_the_full_frame():
    """Config 4 as far as one GPU can take it: the 1M-triangle 4K frame on the benchmark's tree (depth 26, leaves of
    at most 12, pruned lists), cut into bands of 16 rows for 2, 4 and 8 ranks.  Every rank's share is rendered with
    the schedule bench.py gives a rank (three contexts in flight on one shared scene, overlapped passes, 8 waves per
    CU and pass for half a frame, 6 below that), the shares are laid out as the exchange lays them out and put
    together by the library's own de-interleave kernel: the result is the full frame rendered alone, bit for bit,
    and the shares together trace exactly its rays."""
    import ctypes as C

    import torch
    from raytracing_amd import dist as rdist
    from raytracing_amd._lib import check, lib
    from raytracing_amd.pipe import RT_PARAM_LAUNCH_WAVES, RT_PARAM_OVERLAP
    sc = scenes.config3()
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, 26, 12, prune_leaves=1e-5)
    plain = FlatTree.build(v, t, 26, 12)
    assert len(tree.recs) < len(plain.recs)
    # the full frame, one at a time, passes in order, from the reference's leaf lists
    alone = HipRenderer(0)
    alone.load_scene(sc, plain)
    alone.set_param(RT_PARAM_OVERLAP, 0)
    full = torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    alone.render(0, d_out=full.data_ptr())
    want_rays = alone.stats()
    alone.close()
    plain.release()
    assert bool(full.any())

    streams = [torch.cuda.Stream() for _ in range(3)]
    ctx = []
    for k, s in enumerate(streams):
        r = HipRenderer(0, stream=s.cuda_stream)
        if k == 0:
            r.load_scene(sc, tree)
        else:
            r.share_scene(ctx[0], sc)
        ctx.append(r)
    L = lib()
    L.rt_hip_deinterleave.restype = C.c_int
    for parts in (2, 4, 8):
        pad = rdist.padded_rows(sc.h, rdist.BAND_H, parts)
        assert pad == L.rt_hip_gather_rows(sc.h, rdist.BAND_H, parts)
        stage = torch.zeros((parts, pad, sc.w, 4), dtype=torch.uint8, device="cuda")
        got = torch.zeros_like(full)
        for r in ctx:
            r.set_param(RT_PARAM_LAUNCH_WAVES, 8 if parts <= 2 else 6)
        torch.cuda.synchronize()
        # shares in flight: part p on context p mod 3, nothing waited for in between (two rounds, so that every
        # context renders with the others' frames beside it from start to end)
        for rnd in range(2):
            for p in range(parts):
                r = ctx[p % 3]
                r.set_tile(rdist.BAND_H, parts, p)
                r.render(0, d_out=stage[p].data_ptr())
        torch.cuda.synchronize()
        check(L.rt_hip_deinterleave(C.c_void_p(stage.data_ptr()), C.c_void_p(got.data_ptr()), sc.w, sc.h, rdist.BAND_H, parts,
                                    None), "rt_hip_deinterleave")
        torch.cuda.synchronize()
        assert torch.equal(got, full), parts
        rays = dict(raysClosest=0, raysShadow=0, hits=0)
        for p in range(parts):
            ctx[0].set_tile(rdist.BAND_H, parts, p)
            ctx[0].render(0, d_out=stage[p].data_ptr())
            st = ctx[0].stats()
            for k in rays:
                rays[k] += st[k]
        assert rays == {k: want_rays[k] for k in rays}, parts
    for r in reversed(ctx):
        r.close()
    tree.release()


def test_default_scene_at_its_built_in_size():
    """The reference's demo frame (main.c:19-31: 1280x768, 150x150 water quads,
    ten spheres) at the reference's tree limits, every pixel against the oracle."""
    sc = scenes.default_scene()
    img, ids, st, tree = hip_render(sc, 10, 7, flags=RT_HIP_HITIDS)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx, threads=16)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact == 1.0
    assert (st["raysClosest"], st["raysShadow"]) == (ost["raysClosest"], ost["raysShadow"])
    tree.release()


def test_primitives_other_than_spheres_are_counted_but_never_drawn():
    """rt_render_pipe_add_primitive (rt_funcs_render_pipe.c:209-259) counts every
    primitive and stores only spheres."""
    sc, depth, prims = IMG["default_small"]
    z = np.load(os.path.join(GOLD, "orc_default_small.npz"))
    rp = RenderPipe(sc.w, sc.h)
    rp.set_camera(sc.camera)
    rp.add_scene(sc)
    plane = np.zeros(48, np.uint8)
    end = rp._rp.primsEnd
    rp._L.rt_render_pipe_add_primitive(__import__("ctypes").byref(rp._rp), plane.ctypes.data_as(__import__("ctypes").c_void_p), 1)
    assert rp.counts["prims"] == 11 and rp._rp.primsEnd == end
    assert np.array_equal(rp.draw(), z["img"])
    rp.cleanup()


def test_exact_reciprocal_shortcut_equals_ieee_division_for_every_float():
    """The triangle test computes 1/det as hardware reciprocal + one Newton
    step where the estimate is a normal number and as the IEEE division
    elsewhere (the oracle and the reference's host code divide).  All 2^32 bit
    patterns are evaluated on the device: the two must never differ."""
    r = HipRenderer(0)
    bad, fast = r.selftest_rcp()
    r.close()
    assert bad == 0, f"{bad} inputs round differently"
    assert fast > 0.95 * 2 ** 32 * (252 / 256), "the short path should cover nearly all normal inputs"


def test_overlapped_and_in_order_schedules_give_the_same_frame():
    """RT_PARAM_OVERLAP only changes on which stream the shadow / lighting passes run."""
    from raytracing_amd.pipe import RT_PARAM_OVERLAP
    sc = scenes.wavy_grid_scene(48, 160, 96)
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, 14, 7)
    r = HipRenderer(0)
    r.load_scene(sc, tree)
    out = []
    for ov in (1, 0, 1):
        r.set_param(RT_PARAM_OVERLAP, ov)
        r.render(RT_HIP_HITIDS)
        out.append((r.read_frame().copy(), r.read_hit_ids().copy(), r.stats()["raysClosest"]))
    r.close()
    for img, ids, n in out[1:]:
        assert np.array_equal(img, out[0][0]) and np.array_equal(ids, out[0][1]) and n == out[0][2]


def test_waves_per_pass_do_not_change_the_frame():
    """RT_PARAM_LAUNCH_WAVES only sizes the persistent grids of the traversal passes (how many waves of a CU
    one pass occupies); values outside 1..20 are refused."""
    from raytracing_amd._lib import RtError
    from raytracing_amd.pipe import RT_PARAM_LAUNCH_WAVES
    sc, depth, prims = IMG["sphere16k_glass"]
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, 20, 4)
    r = HipRenderer(0)
    r.load_scene(sc, tree)
    out = []
    for waves in (16, 1, 6, 8, 20, 16):
        r.set_param(RT_PARAM_LAUNCH_WAVES, waves)
        r.render(RT_HIP_HITIDS | RT_HIP_STATS)
        st = r.stats()
        # (node visits are not compared: how many leaves the cooperative walk scans ahead depends on how a list is dealt out)
        out.append((r.read_frame().copy(), r.read_hit_ids().copy(), st["raysClosest"], st["raysShadow"], st["hits"]))
    for bad in (0, 21, -3):
        with pytest.raises(RtError, match="RT_PARAM_LAUNCH_WAVES"):
            r.set_param(RT_PARAM_LAUNCH_WAVES, bad)
    r.close()
    for img, ids, *counts in out[1:]:
        assert np.array_equal(img, out[0][0]) and np.array_equal(ids, out[0][1]) and counts == list(out[0][2:])


@pytest.mark.parametrize("frame", [0, 37, 121])
def test_fly_through_frames_match_the_oracle(frame):
    """Config-5 style: camera on the orbit, reflective + refractive sphere, six rays per pixel."""
    sc = scenes.config5_frame(frame, 192, 108, lon=60, lat=30)
    img, ids, st, tree = hip_render(sc, 16, 7)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact == 1.0
    assert st["raysClosest"] == ost["raysClosest"] and st["raysShadow"] == ost["raysShadow"]
    tree.release()


def _read_png(path):
    """Decoder for the files apps/flythrough.py writes (8-bit RGB, filter type 0 on every row)."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data)
        if tag == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("mode", ["pipelined", "serial"])
def test_fly_through_driver_single_rank(tmp_path, mode):
    """Frames in flight + encoder threads (and the one-at-a-time mode) write, for every camera of the orbit, the
    frame that camera gives when rendered on its own."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 5
    cmd = [sys.executable, os.path.join(root, "apps", "flythrough.py"), "--frames", str(n), "--size", "160x96",
           "--lon", "40", "--lat", "20", "--depth", "14", "--leaf", "7", "--out", str(tmp_path / "pngs")]
    out = subprocess.run(cmd + (["--serial"] if mode == "serial" else ["--encoders", "3"]),
                         check=True, capture_output=True, text=True, timeout=300).stdout
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["frames"] == n and rec["rays_last_frame"] > 160 * 96
    assert rec["frames_in_flight"] == (1 if mode == "serial" else 3)
    assert sorted(os.listdir(tmp_path / "pngs"), key=lambda x: int(x[:-4])) == [f"{f + 8}.png" for f in range(n)]
    sc = scenes.config5_frame(0, 160, 96, 40, 20)
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, 14, 7)
    r = HipRenderer(0)
    r.load_scene(sc, tree)
    seen = []
    for f in range(n):
        r.set_camera(scenes.make_camera(160, 96, world=scenes.orbit_world(f, n)))
        r.render(0)
        want = r.read_frame()[..., [2, 1, 0]]
        assert np.array_equal(_read_png(tmp_path / "pngs" / f"{f + 8}.png"), want), f"frame {f}"
        seen.append(want)
    r.close()
    tree.release()
    assert not np.array_equal(seen[0], seen[1])


def _soup_scene(seed, n_tris, alphas, w=128, h=96):
    rng = np.random.RandomState(seed)
    sc = scenes.Scene(f"soup{seed}", w, h, scenes.make_camera(w, h, hv=0.9))
    v, t = random_soup(n_tris, seed)
    sc.meshes.append(scenes.Mesh(v, t, 0))
    sc.flatten = lambda: (v, t)                # keep per-triangle materials
    for k, a in enumerate(alphas):
        sc.materials[k] = T.make_material((a, *rng.uniform(0.2, 1, 3)), (0.0, 0.05, 0.05, 0.05),
                                          (0.0, 0.6, 0.6, 0.6), (8.0, 0.4, 0.4, 0.4), (0.0, 0.3, 0.3, 0.3), 0.02, 1.2)
    sc.lights.append(scenes.make_light((-20.0, -35.0, 0.0), 100.0))
    return sc


def test_negative_opacity_keeps_the_full_shadow_sum():
    """With a negative material opacity the shadow sum can fall below 1 again,
    so the early exit of the shadow query must be off; the result still has
    to be the reference's sum over every triangle before the light."""
    sc = _soup_scene(11, 3000, (1.0, -0.5, 0.7))
    img, ids, st, tree = hip_render(sc, 14, 6)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact >= 0.999
    assert st["raysShadow"] == ost["raysShadow"]


@pytest.mark.parametrize("depth,prims", [(0, 7), (3, 7), (8, 100)])
def test_very_large_leaves(depth, prims):
    """Leaves of hundreds of triangles (more than one wave's worth in the
    cooperative scan, long per-lane scans otherwise)."""
    sc = _soup_scene(12, 700, (1.0, 0.6, 1.0), w=96, h=64)
    img, ids, st, tree = hip_render(sc, depth, prims)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    same, exact = assert_parity(img, ids, oimg, oid)
    assert same == 1.0 and exact >= 0.999
    assert st["raysClosest"] == ost["raysClosest"] and st["raysShadow"] == ost["raysShadow"]


def test_mesh_add_group_feeds_the_pipe_like_explicit_adds(tmp_path):
    """rt_mesh_add_group (rt_meshes.c:352-378) = the group's material into the
    slot + the group's vertex / triangle range into the pipe."""
    import ctypes as C
    from oracle import obj_cases as oc
    from raytracing_amd._lib import lib
    oc.write_case(str(tmp_path))
    L = lib()
    m = oc.Mesh()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert L.rt_mesh_load_from_obj(C.byref(m), b"case.obj") == 0
    finally:
        os.chdir(cwd)
    rc, data = oc.load(L, str(tmp_path), "case.obj")
    v = np.frombuffer((C.c_char * (m.vc * T.vertex.itemsize)).from_address(m.v), T.vertex).copy()
    t = np.frombuffer((C.c_char * (m.tc * T.triangle.itemsize)).from_address(m.t), T.triangle).copy()
    mats = np.frombuffer((C.c_char * (m.mc * T.material.itemsize)).from_address(m.mat), T.material).copy()
    world = np.eye(4, dtype=np.float32)
    world[3, :3] = (1.5, 0.5, -6.0)             # row-vector convention: translation in the last row
    cam = scenes.make_camera(96, 64, hv=0.5, world=world)
    frames = []
    for mode in ("groups", "explicit"):
        rp = RenderPipe(96, 64)
        rp.set_camera(cam)
        rp.add_light(scenes.make_light((2.0, -3.0, -4.0), 20.0))
        if mode == "groups":
            for g in range(m.gc):
                L.rt_mesh_add_group(C.byref(rp._rp), C.byref(m), C.c_uint64(g), C.c_uint64(g))
        else:
            for g, (mi, vo, to, vc, tc) in enumerate(data["groups"]):
                rp.add_material(mats[mi], g)
                rp.add_triangles(v, t, vo, to, vc, tc, g)
        assert rp.counts["triangles"] == 7 and rp.counts["vertices"] == 9
        frames.append(rp.draw())
        rp.cleanup()
    L.rt_mesh_release(C.byref(m))
    assert np.array_equal(frames[0], frames[1]) and frames[0].max() > 0


def test_block_counters_grow_and_reset_like_the_reference():
    """The block accounting of rt_add_block (rt_funcs_render_pipe.c:186-207) and
    its call sites (:215-367): which add makes which counter grow, and
    rt_render_pipe_reset_blocks (:111-125) puts everything back to one block."""
    rp = RenderPipe(32, 24)
    s = rp._rp
    blocks = lambda: (s.primBlocks, s.primDecsBlocks, s.lightBlocks, s.lightDecsBlocks, s.triangleBlocks, s.vertexBlocks, s.materialBlocks)
    assert blocks() == (1,) * 7
    # lights: 48-byte records in 1024-byte blocks grow when end + 48 >= 1024 * blocks (the 22nd light: 21 * 48 + 48 = 1056),
    # descriptors in blocks of 32 grow with the 33rd
    for i in range(33):
        rp.add_light(scenes.make_light((float(i), -30.0, 0.0), 50.0))
        want = (2 if i >= 21 else 1, 2 if i >= 32 else 1)
        assert (s.lightBlocks, s.lightDecsBlocks) == want, i
    assert s.lightsCount == 33 and s.lightsEnd == 33 * T.point_light.itemsize
    # materials: slot 127 fits the first block, slot 128 needs the second
    mat = T.make_material((1.0, 0.5, 0.5, 0.5), (0, .1, .1, .1), (0, .5, .5, .5), (5.0, .2, .2, .2), (0, 0, 0, 0), 0.0, 1.0)
    rp.add_material(mat, 127)
    assert s.materialBlocks == 1
    rp.add_material(mat, 128)
    assert s.materialBlocks == 2
    # triangles / vertices: grow when count + added >= 1048576 * blocks, by added / 1048576 blocks (at least one)
    v, t = random_soup(4, 1)
    big_t = np.zeros(1048575, T.triangle)
    big_t["pV1"], big_t["pV2"] = 1, 2
    rp.add_triangles(v, big_t, 0, 0, 3, len(big_t), 0)
    assert (s.triangleBlocks, s.vertexBlocks, s.trianglesCount, s.vertexCount) == (1, 1, 1048575, 3)
    rp.add_triangles(v, t, 0, 0, 3, 1, 0)
    assert (s.triangleBlocks, s.trianglesCount, s.vertexCount) == (2, 1048576, 6)
    rp.add_triangles(v, np.concatenate([big_t, big_t, big_t[:5]]), 0, 0, 3, 2 * 1048575 + 5, 0)
    assert s.triangleBlocks == 4                       # 2 + (2097155 / 1048576 = 2)
    # spheres: 32-byte records in 32 MiB blocks, descriptors in blocks of 1048576
    rp.add_primitive(scenes.make_sphere((0.0, 0.0, 50.0), 3.0, 0))
    assert (s.primBlocks, s.primDecsBlocks, s.primsCount, s.primsEnd) == (1, 1, 1, T.sphere.itemsize)
    import ctypes
    rp._L.rt_render_pipe_reset_blocks(ctypes.byref(s))
    assert blocks() == (1,) * 7
    assert (s.trianglesCount, s.vertexCount, s.lightsCount, s.primsCount, s.lightsEnd, s.primsEnd) == (0,) * 6
    rp.cleanup()


def test_material_slots_beyond_the_first_block_and_many_lights_render_like_the_oracle():
    sc = edge_scene("few_tris")
    sc.materials[200] = sc.materials[0]
    v, t = sc.flatten()
    t = t.copy()
    t["mat"] = 200
    sc.meshes = [scenes.Mesh(v, t, 200)]
    sc.lights = [scenes.make_light((-5.0 + 0.3 * i, -10.0, 0.1 * i), 30.0 / 40.0) for i in range(40)]
    rp = RenderPipe(sc.w, sc.h)
    rp.set_camera(sc.camera)
    rp.add_scene(sc)
    assert rp._rp.materialBlocks == 2 and rp._rp.lightDecsBlocks == 2
    img = rp.draw()
    rp.cleanup()
    vv, tt = sc.flatten()
    tree = FlatTree.build(vv, tt)
    nodes, idx = tree.expand()
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    assert np.abs(img.astype(int) - oimg.astype(int)).max() <= 1 and img.max() > 0


def test_get_camera_lends_the_pipe_camera_and_draw_device_fills_a_device_buffer():
    """rt_render_pipe_get_camera / _free_camera (rt_funcs_render_pipe.c:369-387):
    the caller edits the pipe's own camera in place.  rt_render_pipe_draw_device
    (no reference counterpart) leaves the frame in HBM."""
    import ctypes
    import torch
    sc, depth, prims = IMG["default_small"]
    frames = []
    for mode in ("set", "edit_in_place", "device"):
        rp = RenderPipe(sc.w, sc.h)
        cam = np.array(sc.camera, T.camera)
        moved = cam.copy()
        moved["world"][3, 0] += 3.0
        if mode == "edit_in_place":
            rp.set_camera(cam)
            p = ctypes.c_void_p()
            rp._L.rt_render_pipe_get_camera(ctypes.byref(rp._rp), ctypes.byref(p))
            lent = np.frombuffer((ctypes.c_char * T.camera.itemsize).from_address(p.value), T.camera)
            assert lent.tobytes() == cam.tobytes()
            lent["world"][0][3, 0] += 3.0
            rp._L.rt_render_pipe_free_camera(ctypes.byref(rp._rp), p)
        else:
            rp.set_camera(moved)
        rp.add_scene(sc)
        if mode == "device":
            out = torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda:0")
            rp._L.rt_render_pipe_draw_device(ctypes.byref(rp._rp), ctypes.c_void_p(out.data_ptr()))
            rp._L.rt_hip_sync(ctypes.c_void_p(rp._L.rt_render_pipe_hip(ctypes.byref(rp._rp))))
            frames.append(out.cpu().numpy())
            assert rp.counts == dict(triangles=0, vertices=0, prims=0, lights=0)     # consumed like draw
        else:
            frames.append(rp.draw())
        rp.cleanup()
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    z = np.load(os.path.join(GOLD, "orc_default_small.npz"))
    assert not np.array_equal(frames[0], z["img"])          # the camera really moved


def test_page_locked_host_block_and_builder_scratch_trim():
    import ctypes
    from raytracing_amd._lib import lib
    from raytracing_amd.pipe import DeviceTree
    L = lib()
    r = HipRenderer(0)
    L.rt_hip_alloc_host.restype = ctypes.c_void_p
    p = L.rt_hip_alloc_host(r._c, ctypes.c_size_t(1 << 20))
    assert p
    ctypes.memset(p, 0x5a, 1 << 20)
    L.rt_hip_free_host(r._c, ctypes.c_void_p(p))
    v, t = random_soup(3000, 2)
    a = DeviceTree.build(v, t, 12, 7)
    n = (a.n_nodes, a.n_recs)
    a.release()
    L.rt_kdtree_device_trim()                                   # frees the builder's kept work buffers
    b = DeviceTree.build(v, t, 12, 7)
    assert (b.n_nodes, b.n_recs) == n
    b.release()
    L.rt_kdtree_device_trim()
    r.close()


def test_resizing_the_image_and_tiling_through_the_pipe_api():
    """rt_render_pipe_set_image_size on a live pipe (the reference's GTK resize
    path, rt_funcs_render_pipe.c:96-109) and rt_render_pipe_set_tile."""
    from raytracing_amd.dist import band_rows
    sc = edge_scene("inside_box")
    big = scenes.make_camera(96, 64, hv=0.8)

    def frame(rp, cam):
        rp.set_camera(cam)
        rp.add_scene(sc)
        return rp.draw()

    fresh = RenderPipe(96, 64)
    want = frame(fresh, big)
    fresh.cleanup()
    rp = RenderPipe(sc.w, sc.h)
    small = frame(rp, sc.camera)
    assert small.shape == (sc.h, sc.w, 4)
    rp.set_image_size(96, 64)
    assert np.array_equal(frame(rp, big), want)
    # two halves in bands of 8 rows: each draw leaves its rows at the top of screenData
    full = np.zeros_like(want)
    for part in range(2):
        rp.set_tile(8, 2, part)
        rows = band_rows(64, 8, 2, part)
        full[rows] = frame(rp, big)[:len(rows)]
    assert np.array_equal(full, want)
    rp.set_tile(16, 1, 0)
    rp.set_image_size(sc.w, sc.h)
    assert np.array_equal(frame(rp, sc.camera), small)
    rp.cleanup()


def test_8k_frame_sampled_against_oracle_and_size_limits():
    """33 Mpixel, 5 rays per pixel (87 M rays, every list longer than 2^24
    entries): a 1-in-96 lattice against the oracle; sizes beyond the 32-bit
    work-item range are refused, not wrapped."""
    from raytracing_amd._lib import RtError
    sc = scenes.sphere_scene(250, 200, 7680, 4320, reflect=0.5, stack_size=5)
    img, ids, st, tree = hip_render(sc, 24, 12)
    nodes, idx = tree.expand()
    oimg, oid, _ = orc.raytrace(sc, tree.bbox, nodes, idx, x0=17, x_step=96, y0=5, y_step=96)
    m = np.zeros(ids.shape, bool)
    m[5::96, 17::96] = True
    same, exact = assert_parity(img, ids, oimg, oid, m)
    assert same == 1.0 and exact == 1.0
    assert sc.w * sc.h <= st["raysClosest"] <= 5 * sc.w * sc.h and st["raysShadow"] == st["hits"]
    r = HipRenderer(0)
    with pytest.raises(RtError):
        r.set_image(40000, 40000)                       # 1.6 G pixels
    r.load_scene(sc, tree)
    r.upload_lights(np.array([scenes.make_light((0.0, -50.0, 0.0), 50.0)] * 64, T.point_light))
    with pytest.raises(RtError, match="work-item range"):
        r.render()                                      # 33 M pixels x 64 lights >= 2^31 shadow items
    r.close()


@pytest.mark.parametrize("prune", [0.0, 1e-5])
@pytest.mark.parametrize("seed", [0, 1, 3, 10, 11])
def test_rays_parallel_to_an_axis_with_the_origin_in_a_split_plane(seed, prune):
    """Odd image sizes give the middle row / column a direction component of
    exactly 0, integer geometry symmetric about the camera puts split planes
    through the origin: the split distance 0 * inf is a NaN, which the
    reference turns into 'both children' and into an exit distance that
    rejects every hit.  Such rays must be walked literally (these seeds are
    the ones a walk with shortcuts got wrong) - and over the reference's leaf
    lists: with pruned lists (prune > 0) they scan the complete lists kept
    behind the pruned ones, the oracle walks the reference's tree."""
    rng = np.random.RandomState(seed)
    w, h = int(rng.choice([65, 97, 33])), int(rng.choice([49, 71, 33]))
    sc = scenes.Scene(f"axis{seed}", w, h, scenes.make_camera(w, h, hv=float(rng.choice([0.5, 0.9]))))
    n = int(rng.choice([200, 2000]))
    v, t = random_soup(n, seed, extent=14.0, size=3.0, center=(0.0, 0.0, 0.0))
    for a in "xyz":
        v["pos"][a] = np.round(v["pos"][a])
    # box = [-17, 17]^3 or [-18, 18]^3: the middle candidate plane of the root is exactly 0 = the camera position
    v["pos"]["x"][:2], v["pos"]["y"][:2], v["pos"]["z"][:2] = (-16, 16), (-16, 16), (-16, 16)
    sc.meshes.append(scenes.Mesh(v, t, 0))
    sc.flatten = lambda: (v, t)
    for m in range(3):
        sc.materials[m] = T.make_material((float(rng.choice([1.0, 0.5])), 0.8, 0.7, 0.6), (0, .05, .05, .05), (0, .6, .6, .6),
                                          (10.0, .4, .4, .4), (0, .4, .4, .4), 0.02, 1.3)
    sc.lights += [scenes.make_light((0.0, -10.0, 0.0), 40.0), scenes.make_light((5.0, 0.0, 0.0), 40.0)]
    sc.stack_size = 6
    depth, prims = int(rng.choice([10, 20])), int(rng.choice([2, 7]))
    img, ids, st, tree = hip_render(sc, depth, prims, prune=prune)
    if prune:
        # (where the reference's partition dropped a triangle that lies in a split plane the builder refuses to prune)
        assert tree.leaf_full is None or len(tree.recs_all) > len(tree.recs)
        tree.release()
        tree = FlatTree.build(v, t, depth, prims)
    nodes, idx = tree.expand()
    assert np.frombuffer(nodes["sep"][:1].tobytes(), np.float32)[0] == 0.0
    oimg, oid, ost = orc.raytrace(sc, tree.bbox, nodes, idx)
    assert np.array_equal(ids, oid) and np.array_equal(img, oimg)
    assert (st["raysClosest"], st["raysShadow"]) == (ost["raysClosest"], ost["raysShadow"])


def test_pruned_lists_are_exact_for_the_reference_nan_walks():
    """Scene 7383 of the randomised sweep: the camera in the centre of integer geometry, so that rays of the middle
    column start in split planes with a direction component of exactly zero.  Under the NaN split distances the
    reference computes there, whole sub-trees cannot confirm a hit, and it is confirmed instead by a leaf that lists
    the triangle without containing it.  With pruned lists alone the frame differs; with the complete lists kept
    behind them (which the rays parallel to an axis scan) it is the oracle's frame on the reference's tree."""
    from tests.helpers import stress_scene
    sc, depth, prims, _, kind, n = stress_scene(7383)
    assert (kind, n) == (3, 400)
    v, t = sc.flatten()
    plain = FlatTree.build(v, t, depth, prims)
    nodes, idx = plain.expand()
    oimg, oid, _ = orc.raytrace(sc, plain.bbox, nodes, idx)
    plain.release()
    tree = FlatTree.build(v, t, depth, prims, prune_leaves=1e-5)
    assert tree.leaf_full is not None
    frames = []
    for lists in ("both", "pruned only"):
        if lists == "pruned only":
            tree.leaf_full = None                  # (upload_tree then hands over the pruned lists alone)
        r = HipRenderer(0)
        r.load_scene(sc, tree)
        r.render(RT_HIP_HITIDS)
        frames.append((r.read_frame().copy(), r.read_hit_ids().copy()))
        r.close()
    assert np.array_equal(frames[0][0], oimg) and np.array_equal(frames[0][1], oid)
    assert not np.array_equal(frames[1][1], oid)          # the case is a real one
    tree.release()


def test_a_frame_can_be_captured_in_a_hip_graph_and_replayed():
    """rt_hip_render only enqueues (kernels, memsets, event fork / join between
    its streams): a caller may capture it in a graph.  The replayed graph must
    write the frame the direct call writes, every time."""
    import torch
    sc = scenes.sphere_scene(100, 80, 320, 200, reflect=0.5)
    v, t = sc.flatten()
    tree = FlatTree.build(v, t, 18, 7)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        r = HipRenderer(0, stream=s.cuda_stream)
        r.load_scene(sc, tree)
        out = torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda")
        r.render(0, d_out=out.data_ptr())
        s.synchronize()
        direct = out.cpu().numpy().copy()
        out.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            r.render(0, d_out=out.data_ptr())
        for _ in range(2):
            g.replay()
            s.synchronize()
            assert np.array_equal(out.cpu().numpy(), direct)
            out.zero_()
        del g
        r.close()
    tree.release()
    assert direct.any()


def test_two_contexts_interleaved_and_from_two_threads():
    """Contexts are independent: frames enqueued alternately on two of them
    (no synchronisation in between) and frames rendered by two host threads at
    once, each with its own context and its own GPU tree build, are the frames
    each scene gives on its own."""
    import threading
    from raytracing_amd.pipe import DeviceTree
    scs = [scenes.sphere_scene(100, 80, 200, 150, reflect=0.5), scenes.sphere_scene(60, 50, 160, 128, reflect=0.0)]
    alone = []
    for sc in scs:
        img, ids, _, tree = hip_render(sc, 16, 7, flags=RT_HIP_HITIDS)
        tree.release()
        alone.append((img, ids))

    rs = []
    for sc in scs:
        v, t = sc.flatten()
        r = HipRenderer(0)
        r.load_scene(sc, FlatTree.build(v, t, 16, 7))
        rs.append(r)
    for _ in range(3):
        for r in rs:
            r.render(RT_HIP_HITIDS)
    for r, (img, ids) in zip(rs, alone):
        assert np.array_equal(r.read_frame(), img) and np.array_equal(r.read_hit_ids(), ids)
        r.close()

    out, errs = [None, None], []

    def worker(i):
        try:
            sc = scs[i]
            v, t = sc.flatten()
            r = HipRenderer(0)
            for _ in range(3):
                r.load_scene(sc, DeviceTree.build(v, t, 16, 7))
                r.render(RT_HIP_HITIDS)
            out[i] = (r.read_frame(), r.read_hit_ids())
            r.close()
        except Exception as e:          # noqa: BLE001 - reported below
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(120)
    assert not errs, errs
    for i in range(2):
        assert out[i] is not None
        assert np.array_equal(out[i][0], alone[i][0]) and np.array_equal(out[i][1], alone[i][1])


def test_frames_in_flight_on_contexts_that_share_one_scene():
    """Two contexts on streams of their own render the scene resident in the
    first (rt_hip_share_scene: no second copy of tree, materials, lights), each
    with its own camera, frames enqueued alternately without synchronisation in
    between.  Every frame is the one the scene gives on its own for that camera."""
    import torch
    sc = scenes.sphere_scene(100, 80, 256, 160, reflect=0.5)
    cams = []
    for k in range(2):
        cam = sc.camera.copy()
        cam["world"][3, 0] += np.float32(6.0 * k)          # row 3 of the world matrix is the eye
        cam["world"][3, 1] -= np.float32(4.0 * k)
        cams.append(cam)
    alone = []
    for cam in cams:
        sc.camera = cam
        img, ids, _, tree = hip_render(sc, 18, 7, flags=RT_HIP_HITIDS)
        tree.release()
        alone.append((img, ids))
    assert not np.array_equal(alone[0][0], alone[1][0])

    v, t = sc.flatten()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    first = HipRenderer(0, stream=streams[0].cuda_stream)
    sc.camera = cams[0]
    first.load_scene(sc, FlatTree.build(v, t, 18, 7))
    second = HipRenderer(0, stream=streams[1].cuda_stream)
    sc.camera = cams[1]
    second.share_scene(first, sc)
    outs = [torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(4):
        for r, o in zip((first, second), outs):
            r.render(RT_HIP_HITIDS, d_out=o.data_ptr())
    for r, o, (img, ids) in zip((first, second), outs, alone):
        r.sync()
        assert np.array_equal(o.cpu().numpy(), img)          # (a frame given a destination is written there only)
        assert np.array_equal(r.read_hit_ids(), ids)
    # an upload into the lender may move its buffers: the borrower has to share again before its next frame
    first.upload_materials(sc.material_table())
    with pytest.raises(Exception, match="rt_hip_share_scene again"):
        second.render(0, d_out=outs[1].data_ptr())
    second.share_scene(first, sc)                       # (sc.camera is the second camera)
    second.render(RT_HIP_HITIDS, d_out=outs[1].data_ptr())
    second.sync()
    assert np.array_equal(outs[1].cpu().numpy(), alone[1][0])
    # and so does the lender's destruction
    third = HipRenderer(0)
    third.share_scene(second, sc)                       # a borrower lends what it borrowed
    third.render(0)
    third.sync()
    # the borrower goes first; the owner's scene is still whole afterwards
    second.close()
    with pytest.raises(Exception, match="rt_hip_share_scene again"):
        third.render(0)
    third.close()
    first.render(RT_HIP_HITIDS)
    assert np.array_equal(first.read_frame(), alone[0][0])
    first.close()
    with pytest.raises(Exception, match="share_scene"):
        a, b = HipRenderer(0), HipRenderer(0)
        try:
            a.share_scene(b, sc)          # b holds no scene
        finally:
            a.close()
            b.close()


def test_launch_intervals_of_two_contexts_share_one_clock():
    """rt_hip_kernel_intervals: every launch of the timed frames of a context, on the clock of another context's
    first timed frame.  Intervals are well-formed, a frame's launches lie inside the frame's own interval, and the
    second context's frames start after the first context's first frame did."""
    import torch
    from raytracing_amd.pipe import RT_HIP_TIMING
    sc = scenes.sphere_scene(100, 80, 256, 160, reflect=0.5)
    v, t = sc.flatten()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    a = HipRenderer(0, stream=streams[0].cuda_stream)
    a.load_scene(sc, FlatTree.build(v, t, 18, 7))
    b = HipRenderer(0, stream=streams[1].cuda_stream)
    b.share_scene(a, sc)
    with pytest.raises(Exception, match="RT_HIP_TIMING"):
        a.kernel_intervals()
    n_frames = 3
    for _ in range(n_frames):
        a.render(RT_HIP_TIMING)
        b.render(RT_HIP_TIMING)
    ia, ib = a.kernel_intervals(), b.kernel_intervals(origin=a)
    slots = sc.stack_size
    for iv in (ia, ib):
        assert len(iv) == n_frames * (3 * slots + 2)
        assert all(e >= s for _, s, e in iv)
        frames = [x for x in iv if x[0] == "frame"]
        assert len(frames) == n_frames and frames == sorted(frames, key=lambda x: x[1])
        per = 3 * slots + 2
        for f in range(n_frames):
            chunk = iv[f * per:(f + 1) * per]
            fr = [x for x in chunk if x[0] == "frame"][0]
            assert all(fr[1] - 1e-3 <= s and e <= fr[2] + 1e-3 for k, s, e in chunk if k != "frame")
            assert sorted(k for k, _, _ in chunk) == sorted(["closest", "shadow", "shade"] * slots + ["fold", "frame"])
    assert min(s for _, s, _ in ia) >= -1e-3 and min(s for _, s, _ in ib) >= -1e-3
    ms, launches, wall = a.kernel_times_wall()          # the read-out that resets the timings
    assert launches["closest"] == slots and wall["closest"] <= ms["closest"] * 1.001
    with pytest.raises(Exception, match="RT_HIP_TIMING"):
        a.kernel_intervals()
    b.close()
    a.close()


def test_c_program_on_the_hip_layer_renders_the_frame_the_python_mirror_renders():
    """apps/abi_example.c: the HIP layer called from plain C (host tree build, uploads, a second context on the first
    one's resident scene, two frames in flight, read-back).  It checks by itself that both contexts agree; here its
    frame hash is compared with the same scene rendered through the ctypes mirror."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "abi_example")
    if not os.path.exists(exe):
        pytest.fail("apps/abi_example is missing: run `make -C raytracing_amd/csrc app` (build() does)")
    w, h, n = 200, 120, 24
    out = subprocess.run([exe, str(w), str(h)], check=True, capture_output=True, text=True, timeout=120).stdout.split()
    assert out[0] == f"{w}x{h}" and int(out[2]) == 2 * n * n
    f32 = np.float32
    i = np.arange(n + 1, dtype=np.float32)
    x = f32(-60.0) + f32(120.0) * i / f32(n)
    z = f32(40.0) + f32(120.0) * i / f32(n)
    X, Z = np.meshgrid(x, z)                              # row j = z index, column i = x index
    dx, dz = X / f32(30.0), (Z - f32(100.0)) / f32(30.0)
    Y = f32(25.0) - f32(12.0) / (f32(1.0) + dx * dx + dz * dz)
    assert Y.dtype == np.float32
    v = np.zeros((n + 1) * (n + 1), T.vertex)
    v["pos"]["x"], v["pos"]["y"], v["pos"]["z"] = X.ravel(), Y.ravel(), Z.ravel()
    v["norm"]["y"] = -1.0
    t = np.zeros(2 * n * n, T.triangle)
    o = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).ravel()
    t["pV0"][0::2], t["pV1"][0::2], t["pV2"][0::2] = o, o + 1, o + n + 1
    t["pV0"][1::2], t["pV1"][1::2], t["pV2"][1::2] = o + 1, o + n + 2, o + n + 1
    sc = scenes.Scene("abi_example", w, h, scenes.make_camera(w, h))
    rest = ((0.0, 0.05, 0.05, 0.05), (0.0, 0.6, 0.6, 0.6), (20.0, 0.3, 0.3, 0.3), (0.0, 0.4, 0.4, 0.4), 0.0, 1.0)
    sc.materials[0] = T.make_material((1.0, 0.8, 0.7, 0.5), *rest)
    sc.materials[1] = T.make_material((1.0, 0.2, 0.4, 0.9), *rest)
    sc.spheres.append(scenes.make_sphere((10.0, 0.0, 90.0), 9.0, 1))
    sc.lights.append(scenes.make_light((-20.0, -35.0, 0.0), 100.0))
    sc.meshes.append(scenes.Mesh(v, t, 0))
    sc.stack_size = 4
    img, _, _, tree = hip_render(sc, 12, 4, flags=RT_HIP_HITIDS)
    tree.release()
    hsh = 1469598103934665603
    for byte in img.tobytes():
        hsh = ((hsh ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(out[6]) == int(np.count_nonzero(img[..., :3].any(axis=-1)))
    assert out[8] == f"{hsh:016x}"


def test_fly_through_frames_are_dealt_round_robin_to_the_ranks(tmp_path):
    """Frame f belongs to rank f mod N and no rank needs the others: two ranks started one after the other (as
    two GPUs would run side by side) together write every frame of the sequence exactly once."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 5
    counts = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
        out = subprocess.run([sys.executable, os.path.join(root, "apps", "flythrough.py"), "--frames", str(n), "--size", "96x64",
                              "--lon", "30", "--lat", "16", "--depth", "12", "--leaf", "7", "--out", str(tmp_path / f"r{rank}")],
                             check=True, capture_output=True, text=True, timeout=300, env=env).stdout
        rec = json.loads(out.strip().splitlines()[-1])
        assert (rec["rank"], rec["world"]) == (rank, 2)
        counts.append(rec["frames"])
        assert sorted(os.listdir(tmp_path / f"r{rank}")) == sorted(f"{f + 8}.png" for f in range(rank, n, 2))
    assert counts == [3, 2]
