"""Loader of the product's C-ABI library (librt_amd.so).

There is no fallback: if the library is missing or a device call fails the
caller gets an exception, never a CPU result.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB selects an alternative build of the same library (kernel tuning experiments)
SO = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "librt_amd.so")
_lib = None


class RtError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("raysClosest", "raysShadow", "hits", "nodeVisits", "triTests",
                 "spillPushes", "maxStack", "nodeVisitsShadow", "triTestsShadow",
                 "nodeSlots", "triSlots", "nodeSlotsShadow", "triSlotsShadow")] + [
                ("slotRays", C.c_uint64 * 8), ("slotHits", C.c_uint64 * 8)]

    def as_dict(self):
        d = {}
        for n, t in self._fields_:
            v = getattr(self, n)
            d[n] = int(v) if t is C.c_uint64 else [int(x) for x in v]
        return d


class FlatTree(C.Structure):      # rt_kdtree_flat
    _fields_ = [("nodes", C.c_void_p), ("nNodes", C.c_uint64),
                ("recs", C.c_void_p), ("nRecs", C.c_uint64),
                ("shade", C.c_void_p), ("nTris", C.c_uint64),
                ("bbox", C.c_float * 8), ("depth", C.c_int),
                ("leafFull", C.c_void_p), ("nRecsFull", C.c_uint64)]


class DeviceTreeStruct(C.Structure):   # rt_kdtree_device
    _fields_ = [("nodes", C.c_void_p), ("nNodes", C.c_uint64),
                ("recs", C.c_void_p), ("nRecs", C.c_uint64),
                ("shade", C.c_void_p), ("nTris", C.c_uint64),
                ("bbox", C.c_float * 8), ("depth", C.c_int), ("maxMaterial", C.c_uint32),
                ("leafFull", C.c_void_p), ("nRecsFull", C.c_uint64)]


class BuildOpts(C.Structure):     # rt_kdtree_build_opts
    _fields_ = [("maxDepth", C.c_int), ("maxPrims", C.c_int), ("threads", C.c_int),
                ("sahStop", C.c_float), ("allAxes", C.c_int), ("pruneLeaves", C.c_float),
                ("stageTicket", C.c_uint64)]


class RenderPipeStruct(C.Structure):   # rt_render_pipe
    _fields_ = [("impl", C.c_void_p)] + [(n, C.c_uint64) for n in
                ("primBlocks", "primDecsBlocks", "lightBlocks", "lightDecsBlocks",
                 "triangleBlocks", "vertexBlocks", "materialBlocks")] + [
                ("primsEnd", C.c_size_t), ("primsCount", C.c_uint64),
                ("lightsEnd", C.c_size_t), ("lightsCount", C.c_uint64),
                ("trianglesCount", C.c_uint64), ("vertexCount", C.c_uint64),
                ("screenData", C.c_void_p), ("w", C.c_int), ("h", C.c_int)]


def build():
    """Compile librt_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    subprocess.run(["make", "-s", "-j8", "-C", os.path.join(HERE, "csrc")], check=True)


def _hip_soname(path):
    """The versioned libamdhip64 name an ELF file carries: for librt_amd.so the runtime it was linked against
    (DT_NEEDED), for a copy of the runtime its own DT_SONAME."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(rb"libamdhip64\.so\.\d+", f.read())
        return m.group(0).decode() if m else None
    except OSError:
        return None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so.  Two HIP runtimes in one
    process do not work: the one that comes second finds no device.  A process that
    uses torch as well (torch already imported, or RT_AMD_SHARE_TORCH_HIP=1 for one
    that imports it later) therefore maps torch's copy first, so that librt_amd.so's
    own dependency binds to it by soname.  Only a copy with the soname the library
    was linked against is taken; a process that never uses torch gets the system
    runtime (RT_AMD_SHARE_TORCH_HIP=0 forces that)."""
    import glob
    import importlib.util
    import sys
    import warnings
    want = os.environ.get("RT_AMD_SHARE_TORCH_HIP")
    if want == "0" or (want != "1" and "torch" not in sys.modules):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    soname = _hip_soname(SO)
    for d in spec.submodule_search_locations:
        found = sorted(glob.glob(os.path.join(d, "lib", "libamdhip64.so*")))
        names = {so: _hip_soname(so) for so in found}
        match = [so for so in found if soname and names[so] == soname]
        if found and not match:
            warnings.warn(f"torch ships {sorted(set(names.values()), key=str)} but {os.path.basename(SO)} was linked against "
                          f"{soname}: not sharing torch's HIP runtime (two runtimes in one process: the second finds no device)")
        for so in match:
            try:
                C.CDLL(so, mode=C.RTLD_GLOBAL)
                return
            except OSError:
                continue


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RtError(f"{SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C raytracing_amd/csrc`) first")
        _share_hip_runtime_with_torch()
        L = C.CDLL(SO)
        L.rt_hip_last_error.restype = C.c_char_p
        L.rt_hip_create.restype = C.c_void_p
        L.rt_hip_create.argtypes = [C.c_int]
        L.rt_hip_destroy.argtypes = [C.c_void_p]
        L.rt_render_pipe_hip.restype = C.c_void_p
        L.rt_render_pipe_draw.restype = C.c_void_p
        for name in ("rt_hip_set_stream", "rt_hip_upload_tree", "rt_hip_upload_tree_lists", "rt_hip_upload_materials",
                     "rt_hip_upload_spheres", "rt_hip_upload_lights", "rt_hip_set_camera",
                     "rt_hip_set_image", "rt_hip_set_tile", "rt_hip_share_scene", "rt_hip_render_noalloc", "rt_hip_kernel_times_ex", "rt_hip_kernel_intervals", "rt_hip_local_rows", "rt_hip_set_param",
                     "rt_hip_render", "rt_hip_sync", "rt_hip_read_frame", "rt_hip_read_hit_ids",
                     "rt_hip_get_stats", "rt_hip_kernel_times", "rt_kdtree_build_flat", "rt_kdtree_build_flat_ex",
                     "rt_kdtree_flat_expand", "rt_kdtree_flatten", "rt_hip_device_count"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RtError(f"{what}: {lib().rt_hip_last_error().decode()} (code {rc})")
