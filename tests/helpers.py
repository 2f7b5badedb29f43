"""Shared test helpers: golden loading and tree comparison."""
import hashlib
import os

import numpy as np

from raytracing_amd import types as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def geom_hash(v, t):
    return hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest()


def load_tree(name):
    z = np.load(os.path.join(GOLD, f"ref_tree_{name}.npz"))
    return {k: z[k] for k in z.files}


def tree_fields(bbox, nodes, idx):
    """Reduce a flat tree to the fields that carry meaning per node kind."""
    last = nodes["isLast"] != 0
    return dict(
        bbox=np.frombuffer(np.asarray(bbox).tobytes(), np.float32),
        last=last.astype(np.int8),
        left=np.where(last, 0, nodes["leftNode"]).astype(np.uint32),
        right=np.where(last, 0, nodes["rightNode"]).astype(np.uint32),
        sep=np.where(last, np.float32(0), nodes["sep"]).astype(np.float32),
        axis=np.where(last, 0, nodes["axis"]).astype(np.int8),
        count=np.where(last, nodes["primsCount"], 0).astype(np.uint32),
        prims=np.where(last, nodes["prims"], 0).astype(np.uint32),
        idx=np.asarray(idx).astype(np.uint32))


def golden_fields(g):
    last = g["last"] != 0
    return dict(bbox=g["bbox"], last=g["last"], left=np.where(last, 0, g["left"]).astype(np.uint32),
                right=np.where(last, 0, g["right"]).astype(np.uint32),
                sep=np.where(last, np.float32(0), g["sep"]).astype(np.float32),
                axis=np.where(last, 0, g["axis"]).astype(np.int8),
                count=np.where(last, g["count"], 0).astype(np.uint32),
                prims=np.where(last, g["prims"], 0).astype(np.uint32), idx=g["idx"])


def assert_tree_equal(a, b):
    for k in ("bbox", "last", "left", "right", "sep", "axis", "count", "prims", "idx"):
        assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
        assert a[k].tobytes() == b[k].tobytes(), f"tree field {k} differs"


def random_soup(n_tris, seed, extent=30.0, size=6.0, center=(0.0, 0.0, 60.0)):
    """Seeded random triangle soup: T.vertex[3n], T.triangle[n] (mat 0..2)."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-extent, extent, (n_tris, 1, 3)) + np.asarray(center)
    p = (c + rng.uniform(-size, size, (n_tris, 3, 3))).astype(np.float32).reshape(-1, 3)
    v = np.zeros(3 * n_tris, T.vertex)
    v["pos"]["x"], v["pos"]["y"], v["pos"]["z"] = p[:, 0], p[:, 1], p[:, 2]
    nrm = rng.normal(size=(3 * n_tris, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v["norm"]["x"], v["norm"]["y"], v["norm"]["z"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    t = np.zeros(n_tris, T.triangle)
    t["pV0"], t["pV1"], t["pV2"] = np.arange(n_tris) * 3, np.arange(n_tris) * 3 + 1, np.arange(n_tris) * 3 + 2
    t["mat"] = rng.randint(0, 3, n_tris)
    return v, t


def stress_scene(seed):
    """Scene `seed` of the randomised parity sweep (tools/stress_parity.py): a random triangle soup with spheres and
    lights, random materials (opaque / transparent / reflective), tree limits and ray budget.
    -> (scene, kd depth limit, leaf size limit, rng, kind, n)"""
    from raytracing_amd import scenes
    rng = np.random.RandomState(seed)
    w, h = int(rng.choice([64, 96, 130, 65, 33])), int(rng.choice([48, 70, 101, 49]))
    sc = scenes.Scene(f"stress{seed}", w, h, scenes.make_camera(w, h, hv=float(rng.uniform(0.3, 0.9))))
    n = int(rng.choice([1, 30, 400, 3000]))
    v, t = random_soup(n, seed, extent=float(rng.uniform(5, 30)), size=float(rng.uniform(1, 8)))
    kind = int(rng.randint(6))
    if kind == 1:                                   # integer coordinates: exact ties, hits on edges and vertices
        for a in "xyz":
            v["pos"][a] = np.round(v["pos"][a])
    elif kind == 2 and n > 4:                       # zero-area and repeated triangles
        t["pV1"][::3] = t["pV0"][::3]
        t[n // 2:] = t[:n - n // 2]
    elif kind == 3:                                 # camera inside integer geometry centred on it
        for a in "xyz":
            v["pos"][a] = np.round(v["pos"][a] - (60.0 if a == "z" else 0.0))
    elif kind == 4:                                 # everything in one plane facing the camera
        v["pos"]["z"] = 40.0
    sc.meshes.append(scenes.Mesh(v, t, 0))
    sc.flatten = lambda v=v, t=t: (v, t)        # keep the per-triangle materials 0..2
    for m in range(3):
        alpha = float(rng.choice([1.0, 1.0, 0.6, 0.3, 0.0]))
        refl = float(rng.choice([0.0, 0.4, 0.8]))
        sc.materials[m] = T.make_material((alpha, *rng.uniform(0.2, 1.0, 3)), (0.0, 0.05, 0.05, 0.05), (0.0, 0.6, 0.6, 0.6),
                                          (float(rng.uniform(2, 30)), 0.4, 0.4, 0.4), (0.0, refl, refl, refl),
                                          float(rng.choice([0.0, 0.02, 0.2])), float(rng.choice([1.0, 1.33, 1.6])))
    sc.spheres += [scenes.make_sphere(rng.uniform(-20, 20, 3) + (0, 0, 60), float(rng.uniform(2, 9)), int(rng.randint(3)))
                   for _ in range(int(rng.randint(0, 4)))]
    if kind == 5:                                   # the camera sits inside a sphere (hits from the inside, side = -1)
        sc.spheres.append(scenes.make_sphere((1.0, -2.0, 10.0), float(rng.uniform(60, 140)), int(rng.randint(3))))
    sc.lights += [scenes.make_light(rng.uniform(-40, 40, 3), float(rng.uniform(30, 120))) for _ in range(int(rng.randint(0, 4)))]
    sc.stack_size = int(rng.randint(1, 9))
    depth, prims = int(rng.choice([0, 4, 10, 20, 30])), int(rng.choice([1, 3, 7, 12, 40]))
    vv, tt = sc.flatten()
    if n >= 3000:
        depth = min(depth, 20 if prims > 3 else 16)      # overlapping triangles cannot be separated: references double per level
    elif n >= 400 and prims <= 3:
        depth = min(depth, 20)
    return sc, depth, prims, rng, kind, n
