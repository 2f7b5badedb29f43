"""Generates the committed fixtures under tests/golden/ (TEST INFRASTRUCTURE).

Run in the build container only (`python -m oracle.gen_golden`), after
`make -C oracle ref`.  Two kinds of fixture are written:

ref_*   outputs of the REFERENCE's own host code (oracle/_ref/librt_ref.so,
        compiled from /root/reference): struct layout, kd-trees of four
        scenes, SAH sweeps, host matrix / camera constructors.  These pin
        the oracle's and the product's tree builder and host math.
orc_*   outputs of this repo's CPU restatement of the device kernel on small
        scenes (hit ids + 8-bit image + counters).  The reference kernel
        cannot run here, so these are regression vectors of the
        restatement, not reference outputs ("parity unpinned").
"""
import hashlib
import json
import math
import os
import sys
import subprocess

import numpy as np

from oracle import orc, ref
from raytracing_amd import scenes
from raytracing_amd import types as T

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def tree_scenes():
    return {
        "wavy32": scenes.wavy_grid_scene(32, 96, 64),
        "sphere640": scenes.sphere_scene(20, 16, 64, 48),
        "sphere16k": scenes.sphere_scene(100, 80, 320, 240, reflect=0.5),
        "default45k": scenes.default_scene(),
    }


def image_scenes():
    return {
        "wavy32": (scenes.wavy_grid_scene(32, 96, 64), 10, 7),
        "wavy32_deep": (scenes.wavy_grid_scene(32, 96, 64), 20, 4),
        "sphere16k_reflect": (scenes.sphere_scene(100, 80, 160, 120, reflect=0.5), 10, 7),
        "sphere16k_glass": (scenes.sphere_scene(100, 80, 160, 120, reflect=0.5, alpha=0.5, opt_dens=1.33), 24, 7),
        "default_small": (scenes.default_scene(320, 192), 10, 7),
    }


def geom_hash(v, t):
    return hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest()


def pack_tree(bbox, nodes, idx):
    return dict(bbox=np.frombuffer(bbox.tobytes(), np.float32).copy(),
                left=nodes["leftNode"].astype(np.uint32), right=nodes["rightNode"].astype(np.uint32),
                count=nodes["primsCount"].astype(np.uint32), prims=nodes["prims"].astype(np.uint32),
                sep=nodes["sep"].copy(), axis=nodes["axis"].astype(np.int8),
                last=nodes["isLast"].astype(np.int8), idx=idx.astype(np.uint32))


def main():
    os.makedirs(GOLD, exist_ok=True)
    assert ref.available(), "run `make -C oracle ref` first"

    # 1. struct layout of the reference's own header
    out = subprocess.run([os.path.join(HERE, "_ref", "layout_probe")], check=True, capture_output=True, text=True)
    json.loads(out.stdout)
    with open(os.path.join(GOLD, "ref_layout.json"), "w") as f:
        f.write(out.stdout)

    # 2. kd-trees built by the reference's rt_kdtree_make_childs / pack_to_buffer
    for name, sc in tree_scenes().items():
        v, t = sc.flatten()
        bbox, nodes, idx = ref.kdtree_build(v, t)
        np.savez_compressed(os.path.join(GOLD, f"ref_tree_{name}.npz"), geom=np.array(geom_hash(v, t)),
                            **pack_tree(bbox, nodes, idx))
        print("ref tree", name, len(t), "tris ->", len(nodes), "nodes", len(idx), "refs")

    # 3. SAH sweeps of the reference's rt_kdtree_compute_sah on sub-lists
    sc = scenes.sphere_scene(20, 16, 64, 48)
    v, t = sc.flatten()
    bbox, _, _ = ref.kdtree_build(v, t)
    rng = np.random.RandomState(7)
    sah = []
    for k in range(12):
        prims = np.sort(rng.choice(len(t), size=rng.randint(8, len(t)), replace=False)).astype(np.uint64)
        axis = k % 3
        bx = bbox.copy()
        if k >= 6:                       # shrunken boxes exercise off-centre planes
            for a in "xyz":
                bx["extents"][a] *= np.float32(0.37 + 0.05 * k)
        nl, nr, sep = ref.compute_sah(prims, t, v, axis, bx)
        sah.append(dict(prims=prims.tolist(), axis=axis, box=np.frombuffer(bx.tobytes(), np.float32).tolist(),
                        nl=nl, nr=nr, sep_bits=int(np.float32(sep).view(np.uint32))))
    with open(os.path.join(GOLD, "ref_sah.json"), "w") as f:
        json.dump(sah, f)

    # 4. host math and constructors of the reference
    import ctypes as C
    hm = {"camera": [], "rotate": [], "projection": []}
    for w, h, hv in ((1280, 768, 0.125 * math.pi), (1920, 1080, 0.125 * math.pi), (3840, 2160, 0.125 * math.pi),
                     (96, 64, 0.3), (500, 500, 1.0)):
        asp = float(np.float32(np.float32(w) / np.float32(h)))
        cam = ref.camera_create(asp, hv)
        hm["camera"].append(dict(w=w, h=h, hv=hv, bits=np.frombuffer(cam.tobytes(), np.uint32).tolist()))
    for ang in (0.0, 0.1, -0.7, 1.5707964, 3.0):
        for ax in (0, 1, 2):
            m = ref.matrix4("rt_matrix4_create_rotate", C.c_float(ang), C.c_int(ax))
            hm["rotate"].append(dict(ang=ang, axis=ax, bits=m.view(np.uint32).ravel().tolist()))
    a = ref.matrix4("rt_matrix4_create_rotate", C.c_float(0.3), C.c_int(1))
    b = ref.matrix4("rt_matrix4_create_translate", C.c_float(1.5), C.c_float(-2.0), C.c_float(7.25))
    prod = np.zeros((4, 4), np.float32)
    ref.lib().rt_matrix4_mult(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), prod.ctypes.data_as(C.c_void_p))
    inv = np.zeros((4, 4), np.float32)
    ref.lib().rt_matrix4_inverse(prod.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))
    hm["mult"] = dict(a=a.view(np.uint32).ravel().tolist(), b=b.view(np.uint32).ravel().tolist(),
                      prod=prod.view(np.uint32).ravel().tolist(), inv=inv.view(np.uint32).ravel().tolist())
    vec = np.zeros(1, T.vec3)
    vec["x"], vec["y"], vec["z"] = 0.25, -3.0, 9.5
    res = {}
    for fn in ("rt_vector3_matrix4_mult", "rt_vector3_matrix4_mult_dir"):
        o = np.zeros(1, T.vec3)
        getattr(ref.lib(), fn)(vec.ctypes.data_as(C.c_void_p), prod.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        res[fn] = [int(np.float32(o[k][0]).view(np.uint32)) for k in "xyz"]
    nrm = np.zeros(1, T.vec3)
    ref.lib().rt_vector3_normalize(vec.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p))
    res["rt_vector3_normalize"] = [int(np.float32(nrm[k][0]).view(np.uint32)) for k in "xyz"]
    n2 = np.zeros(1, T.vec3)
    n2["x"], n2["y"], n2["z"] = 0.0, -0.6, 0.8
    rf = np.zeros(1, T.vec3)
    ref.lib().rt_vector3_reflect(vec.ctypes.data_as(C.c_void_p), n2.ctypes.data_as(C.c_void_p), rf.ctypes.data_as(C.c_void_p))
    res["rt_vector3_reflect"] = [int(np.float32(rf[k][0]).view(np.uint32)) for k in "xyz"]
    hm["vec"] = res
    with open(os.path.join(GOLD, "ref_host_math.json"), "w") as f:
        json.dump(hm, f)

    # 4a. every function of rt_funcs_math.h / rt_funcs_primitives.h on the seeded call table
    write_math_golden()

    # 4b. OBJ / MTL loader of the reference on the inputs of oracle/obj_cases.py
    write_obj_golden()

    # 4c. the reference's vertex-normal rule on the demo water mesh and a closed sphere
    write_normals_golden()

    # 5. regression vectors of the kernel restatement
    for name, (sc, depth, prims) in image_scenes().items():
        v, t = sc.flatten()
        bbox, nodes, idx = orc.kdtree_build(v, t, depth, prims)
        img, hit, st = orc.raytrace(sc, bbox, nodes, idx, threads=8)
        np.savez_compressed(os.path.join(GOLD, f"orc_{name}.npz"), geom=np.array(geom_hash(v, t)),
                            depth=depth, prims=prims, img=img, hit=hit,
                            stats=np.array([st[k] for k in ("raysClosest", "raysShadow", "nodeVisits", "triTests", "hits")], np.uint64))
        print("orc image", name, img.shape, st)


def restore_normals_by_reference(verts, tris):
    """rt_restore_normals of the reference (rt_meshes.c:77-107) on a copy of
    `verts`: adds every triangle's normal to its three vertices in turn, on top
    of the normals already stored, then normalises.  main.c:398-423 applies the
    same per-triangle order to the water mesh."""
    import ctypes as C
    from oracle import obj_cases as oc
    v = np.ascontiguousarray(verts, T.vertex).copy()
    t = np.ascontiguousarray(tris, T.triangle)
    m = oc.Mesh()
    m.v, m.t, m.vc, m.tc = v.ctypes.data, t.ctypes.data, len(v), len(t)
    ref.lib().rt_restore_normals(C.byref(m))
    return v


def normals_cases():
    """name -> (start normals, mesh) for the normal fixtures"""
    water = scenes.default_scene(64, 48, t=0.37, squares=24).meshes[1]
    ball = scenes.displaced_sphere_mesh(20, 16, 0)
    return {"water24": ((0.0, -1.0, 0.0), water), "sphere640": ((0.0, 0.0, 0.0), ball)}


def write_normals_golden():
    """tests/golden/ref_normals.npz: vertex normals the reference's
    rt_restore_normals gives for the positions / triangles of two meshes of
    raytracing_amd/scenes.py (bit patterns)."""
    out = {}
    for name, (start, mesh) in normals_cases().items():
        v = mesh.verts.copy()
        v["norm"]["x"], v["norm"]["y"], v["norm"]["z"] = start
        r = restore_normals_by_reference(v, mesh.tris)
        out[name] = np.stack([r["norm"][k] for k in "xyz"], axis=1).astype(np.float32).view(np.uint32)
    np.savez_compressed(os.path.join(GOLD, "ref_normals.npz"), **out)
    print("normals golden:", {k: a.shape for k, a in out.items()})


def write_math_golden():
    """tests/golden/ref_math_table.json: outputs of the reference's host math and
    constructors on oracle/math_cases.py's call table (bit patterns)."""
    from oracle import math_cases as mc
    out = mc.run_all(ref.lib(), skip=mc.NOT_RUNNABLE_IN_REFERENCE)
    with open(os.path.join(GOLD, "ref_math_table.json"), "w") as f:
        json.dump(dict(cases=mc.CASES, results=out), f)
    print("math table golden:", len(out), "functions x", mc.CASES, "cases")


def write_obj_golden():
    """tests/golden/ref_obj_loader.json: what the reference's rt_mesh_load_from_obj
    (rt_meshes.c:136-340) returns for oracle/obj_cases.py's OBJ + MTL pair
    (floats as bit patterns).  The inputs are this repository's own text."""
    import tempfile
    from oracle import obj_cases as oc
    d = tempfile.mkdtemp()
    oc.write_case(d)
    rc, m = oc.load(ref.lib(), d, "case.obj")
    assert rc == 0
    bits = lambda a: np.asarray(a, np.float32).view(np.uint32).tolist()
    out = dict(pos=bits(m["pos"]), norm=bits(m["norm"]), tris=m["tris"].tolist(), groups=m["groups"],
               mats=[{k: (bits(v) if isinstance(v, list) else bits([v])[0]) for k, v in mt.items()} for mt in m["mats"]])
    with open(os.path.join(GOLD, "ref_obj_loader.json"), "w") as f:
        json.dump(out, f)
    print("obj loader golden:", len(m["tris"]), "triangles", len(m["groups"]), "groups", len(m["mats"]), "materials")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "obj":
        write_obj_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "math":
        write_math_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "normals":
        write_normals_golden()
    else:
        main()
