"""Render-pipe bookkeeping without a GPU: only meaningful against the sanitizer
build of the host sources (tools/run_asan.sh), where the HIP layer is a
test-only stand-in.  Adds, block growth, draw (consumes the scene), resize,
tiling, reset and cleanup must touch no memory they do not own."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import scenes
from raytracing_amd import types as T
from raytracing_amd._lib import SO

if "asan" not in os.path.basename(SO):
    pytest.skip("needs RT_AMD_LIB = the sanitizer build (tools/run_asan.sh)", allow_module_level=True)

from raytracing_amd.pipe import RT_PARAM_GPU_BUILD, RT_PARAM_MAX_DEPTH, RT_PARAM_TREE_CACHE, RenderPipe
from tests.helpers import random_soup


def test_call_sequences_of_the_render_pipe():
    sc = scenes.default_scene(64, 48, squares=24)
    rp = RenderPipe(sc.w, sc.h)
    rp.set_param(RT_PARAM_GPU_BUILD, 0)
    rp.set_camera(sc.camera)
    for rep in range(3):
        rp.add_scene(sc)
        assert rp.counts["triangles"] == sc.n_tris
        img = rp.draw()
        assert img.shape == (sc.h, sc.w, 4) and rp.counts == dict(triangles=0, vertices=0, prims=0, lights=0)
    rp.set_param(RT_PARAM_TREE_CACHE, 1)
    rp.set_param(RT_PARAM_MAX_DEPTH, 18)
    for rep in range(2):
        rp.add_scene(sc)
        rp.draw()
    rp.set_image_size(96, 40)
    rp.set_tile(8, 3, 1)
    rp.add_scene(sc)
    rp.draw()
    bb, nodes, idx = (rp.add_scene(sc), rp.kdtree_build())[1]
    assert len(nodes) > 1 and len(idx) >= sc.n_tris
    rp.draw()
    # block growth: more than a block of triangles, lights, material slots
    v, t = random_soup(4, 1)
    big = np.zeros(1048580, T.triangle)
    big["pV1"], big["pV2"] = 1, 2
    rp.add_triangles(v, big, 0, 0, 3, len(big), 0)
    for i in range(40):
        rp.add_light(scenes.make_light((float(i), -30.0, 0.0), 50.0))
    rp.add_material(sc.materials[0], 300)
    assert rp._rp.triangleBlocks >= 2 and rp._rp.lightBlocks >= 2 and rp._rp.materialBlocks >= 3
    rp._L.rt_render_pipe_reset_blocks(C.byref(rp._rp))
    assert rp.counts == dict(triangles=0, vertices=0, prims=0, lights=0)
    rp.cleanup()


@pytest.mark.parametrize("cache", [0, 1])
def test_triangles_added_in_pieces_around_the_staging_size(cache):
    """add_triangles copies, rebases (and, with the tree cache, hashes) a mesh in pieces of 131072 triangles and hands
    what it has copied to the GPU builder's staging call: counts on both sides of a piece, odd counts, offsets into the
    caller's arrays, two frames (the second reuses the buffers)."""
    v, t = random_soup(3, 2)
    tri = np.zeros(600_000, T.triangle)
    tri["pV0"], tri["pV1"], tri["pV2"] = 5, 6, 7
    verts = np.zeros(200_000, T.vertex)
    rp = RenderPipe(32, 24)
    rp.set_param(RT_PARAM_GPU_BUILD, 1)
    rp.set_param(RT_PARAM_TREE_CACHE, cache)
    for frame in range(2):
        total = 0
        for off, cnt in ((0, 131071), (7, 131072), (1, 131073), (3, 1), (0, 0), (11, 262145)):
            rp.add_triangles(verts, tri, 5, off, 150_000, cnt, 1)
            total += cnt
        assert rp.counts["triangles"] == total and rp.counts["vertices"] == 6 * 150_000
        rp.draw()
        assert rp.counts == dict(triangles=0, vertices=0, prims=0, lights=0)
    rp.add_triangles(verts, tri, 5, 0, 150_000, 300_000, 1)
    rp.cleanup()          # (with geometry handed over and not drawn)
