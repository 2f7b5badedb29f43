"""ABI: struct layouts against the reference's own header (golden dump) and
the exported symbols of the C-ABI library against include/*.h."""
import ctypes
import glob
import json
import os
import re

import numpy as np

from raytracing_amd import types as T
from tests.helpers import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NUMPY = {
    "rt_vector3": T.vec3, "rt_vertex": T.vertex, "rt_triangle": T.triangle, "rt_box": T.box,
    "rt_sphere": T.sphere, "rt_point_light": T.point_light, "rt_color": T.color,
    "rt_material": T.material, "rt_argb": T.argb, "rt_camera": T.camera,
    "rt_raytrace_args": T.raytrace_args, "rt_prim_desc": T.prim_desc, "rt_light_desc": T.light_desc,
    "rt_cl_kdtree_node": T.cl_kdtree_node,
}


def test_numpy_layouts_match_reference_header():
    ref = json.load(open(os.path.join(GOLD, "ref_layout.json")))
    for name, dt in NUMPY.items():
        assert dt.itemsize == ref[name]["size"], name
        for field, off in ref[name].items():
            if field == "size" or field not in (dt.names or ()):
                continue
            assert dt.fields[field][1] == off, (name, field)


def test_c_header_layouts_match_reference_header(tmp_path):
    """Compile a probe against include/rt_types.h and compare with the dump
    taken from the reference's header."""
    import subprocess
    src = open(os.path.join(ROOT, "oracle", "layout_probe.c")).read()
    exe = tmp_path / "probe"
    (tmp_path / "probe.c").write_text(src)
    subprocess.run(["gcc", "-std=gnu11", "-I", os.path.join(ROOT, "include"), str(tmp_path / "probe.c"), "-o", str(exe)],
                   check=True)
    mine = json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)
    ref = json.load(open(os.path.join(GOLD, "ref_layout.json")))
    assert mine == ref


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"\b((?:rt_|minF|maxF|is_power_of_two)\w*)\s*\(", text):
            names.add(m.group(1))
    names -= {"rt_pipe_impl"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "raytracing_amd", "librt_amd.so")
    assert os.path.exists(so), "librt_amd.so not built"
    lib = ctypes.CDLL(so)
    names = declared_functions()
    assert len(names) > 80
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    for g in ("rt_cl_include_path", "rt_cl_raytrace_kernel_path"):
        ctypes.c_char_p.in_dll(lib, g)


def test_hbm_record_sizes():
    from raytracing_amd import pipe
    assert pipe.kdnode.itemsize == 8 and pipe.trirec.itemsize == 48 and pipe.trishade.itemsize == 48


def test_frame_kernels_keep_their_state_in_registers(tmp_path):
    """No kernel of a frame may spill registers to scratch memory: spill traffic
    is HBM traffic in the hottest loops, and builds of the traversal kernels
    that spilled heavily have produced wrong pixels.  Read from the metadata
    of the device code inside the built library."""
    import shutil
    import subprocess
    import pytest
    from raytracing_amd._lib import SO
    tools = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{tools}/llvm-objdump") and os.path.exists(f"{tools}/llvm-readelf")):
        pytest.skip("LLVM binary tools not installed")
    so = shutil.copy(SO, tmp_path / "lib.so")
    subprocess.run([f"{tools}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=tmp_path)
    seen = {}
    for obj in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([f"{tools}/llvm-readelf", "--notes", str(obj)], check=True, capture_output=True, text=True).stdout
        for name, scratch in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            seen[name] = int(scratch)
    frame = {k: v for k, v in seen.items() if re.search(r"k_closest|k_shadow|k_light|k_fold", k)}
    assert len(frame) >= 6, sorted(seen)
    assert all(v == 0 for v in frame.values()), frame
