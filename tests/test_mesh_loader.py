"""OBJ / MTL loader and mesh container (reference rt_meshes.c) of the product
library: against what the reference's own loader returned for the same text
(tests/golden/ref_obj_loader.json, written by oracle/gen_golden.py), against
the reference run live where its library exists, and the documented
extensions (files without normals or without an 'o' line)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import obj_cases as oc
from raytracing_amd._lib import lib
from tests.helpers import GOLD


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32).tolist()


def as_golden(m):
    return dict(pos=bits(m["pos"]), norm=bits(m["norm"]), tris=m["tris"].tolist(), groups=m["groups"],
                mats=[{k: (bits(v) if isinstance(v, list) else bits([v])[0]) for k, v in mt.items()} for mt in m["mats"]])


def test_loader_gives_what_the_reference_loader_gave(tmp_path):
    oc.write_case(str(tmp_path))
    rc, m = oc.load(lib(), str(tmp_path), "case.obj")
    assert rc == 0
    want = json.load(open(os.path.join(GOLD, "ref_obj_loader.json")))
    got = as_golden(m)
    for k in ("tris", "groups", "pos", "norm", "mats"):
        assert got[k] == want[k], k
    # the quirks by name: a quad (a b c d) is (a b c) + (d a c); a vertex keeps the normal of the LAST
    # face corner naming it; the opacity is the number on the last line that has one after two characters
    assert m["tris"][:2].tolist() == [[0, 1, 2], [3, 0, 2]]
    assert np.allclose(m["norm"][0], [0.0, 0.6, 0.8]) and np.allclose(m["norm"][6], [0.0, 0.0, 1.0])
    assert [mt["color"][3] for mt in m["mats"]] == [0.75, 0.25]
    assert all(mt["color"][:3] == [1.0, 1.0, 1.0] and mt["reflect"] == [0.0, 0.0, 0.0] for mt in m["mats"])


def test_loader_against_the_live_reference(tmp_path):
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/librt_ref.so is only built where the reference sources are present")
    oc.write_case(str(tmp_path))
    a = oc.load(lib(), str(tmp_path), "case.obj")[1]
    b = oc.load(ref.lib(), str(tmp_path), "case.obj")[1]
    assert as_golden(a) == as_golden(b)


@pytest.mark.parametrize("seed", range(12))
def test_random_well_formed_files_against_the_live_reference(tmp_path, seed):
    """Random objects, tables, triangles and quads, number formats, materials
    (oracle/obj_cases.random_case; tools/fuzz_obj.py runs hundreds of them).
    The reference never writes the normal of a vertex no face names, so only
    the named ones are compared."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/librt_ref.so is only built where the reference sources are present")
    obj, mtl = oc.random_case(seed)
    (tmp_path / f"fuzz{seed}.mtl").write_text(mtl)
    (tmp_path / "f.obj").write_text(obj)
    out = []
    for L in (lib(), ref.lib()):
        rc, m = oc.load(L, str(tmp_path), "f.obj")
        assert rc == 0
        used = np.zeros(len(m["pos"]), bool)
        used[m["tris"].ravel()] = True
        out.append(as_golden(dict(m, norm=np.where(used[:, None], m["norm"], 0.0).astype(np.float32))))
    assert out[0] == out[1]


def test_file_without_normals_gets_them_from_the_faces(tmp_path):
    """The reference dereferences its (absent) normal table for such a file;
    here the normals are the normalised sums of (v1-v0) x (v2-v0) over the
    faces at a vertex, the rule of rt_restore_normals (rt_meshes.c:77-107)."""
    oc.write_case(str(tmp_path))
    rc, m = oc.load(lib(), str(tmp_path), "plain.obj", with_materials=False)
    assert rc == 0 and m["groups"] == [[-1, 0, 0, 4, 4]]
    assert m["tris"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 1], [2, 0, 1]]
    acc = np.zeros((4, 3), np.float32)
    for a, b, c in m["tris"]:
        n = np.cross(m["pos"][b] - m["pos"][a], m["pos"][c] - m["pos"][a]).astype(np.float32)
        for k in (a, b, c):
            acc[k] += n
    want = acc / np.linalg.norm(acc, axis=1, keepdims=True)
    assert np.allclose(m["norm"], want, atol=1e-6)


def test_faces_before_any_object_line_form_one_group(tmp_path):
    text = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n"
    (tmp_path / "nogroup.obj").write_text(text)
    rc, m = oc.load(lib(), str(tmp_path), "nogroup.obj", with_materials=False)
    assert rc == 0 and m["groups"] == [[-1, 0, 0, 3, 1]] and m["tris"].tolist() == [[0, 1, 2]]
    assert m["norm"].tolist() == [[0.0, 0.0, 1.0]] * 3


def test_other_corner_forms_relative_and_wild_indices(tmp_path):
    """Corners written "v", "v/vt" or "v//vn" and negative (relative) indices are
    read; a face that names a vertex outside the vertex list is dropped instead of
    being stored with a wild index (the reference stops parsing after the first
    corner of such a line and keeps whatever its stack held)."""
    text = ("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvn 0 1 0\n"
            "f 1 2 3\n"                 # plain
            "f 1/7 2/8 4/9\n"           # v/vt
            "f 1//1 2//1 3//2\n"        # v//vn
            "f -4 -3 -1\n"              # relative: 1 2 4
            "f 1 2 9\n"                 # 9 > 4 vertices: dropped
            "f 0 1 2\n"                 # 0 is not an index: dropped
            "f 1 2 -7\n"                # before the first vertex: dropped
            "f 2 3 4 1\n")              # quad of plain corners
    (tmp_path / "forms.obj").write_text(text)
    rc, m = oc.load(lib(), str(tmp_path), "forms.obj", with_materials=False)
    assert rc == 0
    assert m["tris"].tolist() == [[0, 1, 2], [0, 1, 3], [0, 1, 2], [0, 1, 3], [1, 2, 3], [0, 1, 3]]
    assert m["groups"] == [[-1, 0, 0, 4, 6]]
    assert m["norm"][2].tolist() == [0.0, 1.0, 0.0] and m["norm"][0].tolist() == [0.0, 0.0, 1.0]
    # without normals the same faces go through rt_restore_normals: nothing out of range reaches it
    (tmp_path / "forms_plain.obj").write_text(text.replace("vn 0 0 1\nvn 0 1 0\n", "").replace("//1", "").replace("//2", ""))
    rc, m = oc.load(lib(), str(tmp_path), "forms_plain.obj", with_materials=False)
    assert rc == 0 and len(m["tris"]) == 6 and np.isfinite(m["norm"]).all()


@pytest.mark.parametrize("seed", range(8))
def test_random_malformed_faces_never_store_a_wild_index(tmp_path, seed):
    obj, mtl = oc.malformed_case(seed)
    (tmp_path / f"fuzz{seed}.mtl").write_text(mtl)
    (tmp_path / "f.obj").write_text(obj)
    rc, m = oc.load(lib(), str(tmp_path), "f.obj")
    assert rc == 0
    assert len(m["tris"]) == 0 or (m["tris"].min() >= 0 and m["tris"].max() < len(m["pos"]))
    assert sum(g[4] for g in m["groups"]) == len(m["tris"])


def test_restore_normals_skips_triangles_with_wild_indices():
    from raytracing_amd import types as T
    v = np.zeros(3, T.vertex)
    v["pos"]["x"][1], v["pos"]["y"][2] = 1.0, 1.0
    t = np.zeros(2, T.triangle)
    t["pV1"][0], t["pV2"][0] = 1, 2
    t["pV0"][1], t["pV1"][1], t["pV2"][1] = 0, 1, 2 ** 40
    m = oc.Mesh()
    m.v, m.t, m.vc, m.tc = v.ctypes.data, t.ctypes.data, 3, 2
    lib().rt_restore_normals(C.byref(m))
    assert [v["norm"][k].tolist() for k in "xyz"] == [[0.0] * 3, [0.0] * 3, [1.0] * 3]


@pytest.mark.parametrize("style", ["trailing_blanks", "crlf", "tabs"])
def test_line_ends_and_blanks_an_exporter_may_write(tmp_path, style):
    """Trailing blanks and CRLF line ends drive the reference's parser into
    memory it does not own (tools/fuzz_obj.py); the product reads such files
    as the same file without them."""
    oc.write_case(str(tmp_path))
    want = oc.load(lib(), str(tmp_path), "case.obj")[1]
    lines = oc.OBJ.split("\n")[:-1]
    mtl = oc.MTL
    if style == "trailing_blanks":
        text = "\n".join(l + "  " for l in lines) + "\n"
    elif style == "crlf":
        text = "\r\n".join(lines) + "\r\n"
        mtl = mtl.replace("\n", "\r\n")
    else:
        text = "\n".join(l.replace(" ", "\t") for l in lines) + "\n"
        mtl = mtl.replace(" ", "\t")
    d = tmp_path / style
    d.mkdir()
    (d / "case.mtl").write_text(mtl, newline="")
    (d / "case.obj").write_text(text, newline="")
    rc, got = oc.load(lib(), str(d), "case.obj")
    assert rc == 0 and as_golden(got) == as_golden(want)


def test_missing_file_returns_minus_one(tmp_path):
    rc, m = oc.load(lib(), str(tmp_path), "absent.obj")
    assert rc == -1 and m is None


def test_reserve_and_release():
    L = lib()
    m = oc.Mesh()
    L.rt_mesh_reserve(C.byref(m), C.c_uint64(5), C.c_uint64(3), C.c_uint64(2))
    assert (m.vc, m.tc, m.gc) == (5, 3, 2) and m.v and m.t and bool(m.g)
    L.rt_mesh_release(C.byref(m))
    assert not m.v and not m.t and not bool(m.g)
