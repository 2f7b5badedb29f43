"""The headless C application (apps/rt_headless.c, reference API only) against
the Python restatement of the same demo scene rendered through the HIP layer."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from raytracing_amd import scenes
from raytracing_amd.pipe import FlatTree, HipRenderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "apps", "rt_headless")


def read_ppm(path):
    raw = open(path, "rb").read()
    magic, dims, mx, body = raw.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"P6" and mx == b"255"
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


def read_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data)
        if tag == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", data[:10])
            assert (depth, colour) == (8, 2)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_app_is_built_and_links_the_product_library():
    assert os.access(APP, os.X_OK), "run `make -C raytracing_amd/csrc app`"
    out = subprocess.run(["ldd", APP], capture_output=True, text=True).stdout
    assert "librt_amd.so" in out


@pytest.mark.gpu
def test_headless_app_renders_the_demo_scene_and_resumes(tmp_path):
    w, h, sq, dt = 320, 192, 40, 0.25
    run = lambda *a: subprocess.run([APP, "--size", f"{w}x{h}", "--squares", str(sq), "--dt", str(dt), *a],
                                    cwd=tmp_path, check=True, capture_output=True, text=True, timeout=120)
    run("--frames", "2", "--format", "ppm", "--out", "pngs")
    frames = [read_ppm(tmp_path / "pngs" / f"{i}.ppm") for i in range(2)]
    t, dt_saved, n = struct.unpack("<ffQ", open(tmp_path / "save", "rb").read())
    assert (n, dt_saved) == (2, np.float32(dt)) and t == np.float32(dt)

    # the same two frames through the HIP layer from the Python scene description
    for i, tt in enumerate((0.0, float(np.float32(dt)))):
        sc = scenes.default_scene(w, h, t=tt, squares=sq)
        v, tr = sc.flatten()
        tree = FlatTree.build(v, tr, 10, 7)
        r = HipRenderer(0)
        r.load_scene(sc, tree)
        r.render(0)
        img = r.read_frame()
        r.close()
        rgb = np.stack([img[..., 2], img[..., 1], img[..., 0]], axis=-1)      # BGRA -> RGB
        d = np.abs(rgb.astype(int) - frames[i].astype(int))
        assert d.max() <= 1 and (d == 0).mean() > 0.999, (i, d.max(), (d == 0).mean())
    assert not np.array_equal(frames[0], frames[1])

    # resume: continues at t + dT with the next file number, PNG this time
    run("--frames", "1", "--format", "png", "--out", "pngs", "--resume")
    t2, _, n2 = struct.unpack("<ffQ", open(tmp_path / "save", "rb").read())
    assert n2 == 3 and t2 == np.float32(np.float32(dt) + np.float32(dt))
    png = read_png(tmp_path / "pngs" / "2.png")
    assert png.shape == (h, w, 3) and png.any()


@pytest.mark.gpu
def test_headless_app_adds_an_obj_mesh(tmp_path):
    """--obj: rt_mesh_load_from_obj + rt_mesh_add on top of the demo scene."""
    from oracle import obj_cases as oc
    oc.write_case(str(tmp_path))
    # put the little mesh where the demo camera sees it: scaled up, above the water in front of the big sphere
    text = open(tmp_path / "case.obj").read().splitlines()
    moved = []
    for line in text:
        if line.startswith("v "):
            x, y, z = (float(s) for s in line.split()[1:4])
            line = f"v {8 * x - 12:.3f} {8 * y - 8:.3f} {8 * z + 5:.3f}"
        moved.append(line)
    (tmp_path / "big.obj").write_text("\n".join(moved) + "\n")
    run = lambda *a: subprocess.run([APP, "--size", "256x160", "--squares", "30", "--frames", "1", "--format", "ppm", *a],
                                    cwd=tmp_path, check=True, capture_output=True, text=True, timeout=120)
    a = run("--out", "plain")
    b = run("--out", "withobj", "--obj", "big.obj")
    assert "1810 triangles" in a.stdout and "1817 triangles" in b.stdout, (a.stdout, b.stdout)
    fa, fb = read_ppm(tmp_path / "plain" / "0.ppm"), read_ppm(tmp_path / "withobj" / "0.ppm")
    changed = (fa != fb).any(axis=-1).mean()
    assert 0.001 < changed < 0.5, changed
    c = run("--out", "missing", "--obj", "absent.obj")          # unreadable file: message, scene without it
    assert "obj load error" in c.stderr and np.array_equal(read_ppm(tmp_path / "missing" / "0.ppm"), fa)
