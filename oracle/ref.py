"""Driver for the reference's own host library (TEST INFRASTRUCTURE ONLY).

oracle/_ref/librt_ref.so is compiled by `make -C oracle ref` from the
reference's sources where they lie in /root/reference; it exists only in the
build container.  This module calls its tree-builder and host-math entry
points directly (none of them needs an OpenCL device) so that golden vectors
can be produced by the reference itself: see oracle/gen_golden.py.
"""
import ctypes as C
import os

import numpy as np

from raytracing_amd import types as T

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "librt_ref.so")


def available():
    # RT_NO_LIVE_REFERENCE: sanitizer runs of the product's host code (tools/run_asan.sh) leave the
    # reference's own library out - its allocations are not the sanitizer's business
    return os.path.exists(SO) and not os.environ.get("RT_NO_LIVE_REFERENCE")


class CountInfo(C.Structure):
    _fields_ = [("primsCount", C.c_uint64), ("nodesCount", C.c_uint64)]


class KdNode(C.Structure):          # reference rt_types.h:191-200
    pass


KdNode._fields_ = [("leftNode", C.POINTER(KdNode)), ("rightNode", C.POINTER(KdNode)),
                   ("primsCount", C.c_uint64), ("prims", C.POINTER(C.c_uint64)),
                   ("sep", C.c_float), ("axis", C.c_int), ("isLast", C.c_int)]

_lib = None
_libc = None


def lib():
    global _lib, _libc
    if _lib is None:
        _lib = C.CDLL(SO)
        _libc = C.CDLL("libc.so.6")
        _libc.malloc.restype = C.c_void_p
        _libc.malloc.argtypes = [C.c_size_t]
        _libc.free.argtypes = [C.c_void_p]
        _lib.rt_kdtree_make_childs.restype = CountInfo
        _lib.rt_kdtree_make_childs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KdNode), C.c_void_p, C.c_int]
        _lib.rt_kdtree_pack_to_buffer.restype = CountInfo
        _lib.rt_kdtree_pack_to_buffer.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KdNode),
                                                  C.c_uint64, C.c_uint64]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def scene_box(verts):
    """Preamble of rt_kdtree_build (reference rt_funcs_render_pipe.c:1114-1137):
    min/max over all vertices, +-1.0f, then the reference's own
    rt_vector3_add / _sub / _scalar_mult for centre and half extents."""
    L = lib()
    if len(verts):
        mn = np.array([verts["pos"][k].min() for k in "xyz"], np.float32) - np.float32(1.0)
        mx = np.array([verts["pos"][k].max() for k in "xyz"], np.float32) + np.float32(1.0)
    else:
        mn = np.full(3, -1.0, np.float32)
        mx = np.full(3, 1.0, np.float32)
    vmn, vmx = np.zeros(1, T.vec3), np.zeros(1, T.vec3)
    for i, k in enumerate("xyz"):
        vmn[k], vmx[k] = mn[i], mx[i]
    bb = np.zeros(1, T.box)
    c, e = np.zeros(1, T.vec3), np.zeros(1, T.vec3)
    L.rt_vector3_add(_p(vmx), _p(vmn), _p(c))
    L.rt_vector3_scalar_mult(_p(c), C.c_float(0.5), _p(c))
    L.rt_vector3_sub(_p(vmn), _p(vmx), _p(e))
    L.rt_vector3_scalar_mult(_p(e), C.c_float(0.5), _p(e))
    bb["center"], bb["extents"] = c[0], e[0]
    return bb


def kdtree_build(verts, tris, builder=None):
    """The reference's rt_kdtree_make_childs + rt_kdtree_pack_to_buffer on host
    arrays, driven exactly like rt_kdtree_build drives them (:1139-1198).
    Depth and leaf size are the reference's compile-time 10 and 7.
    `builder`: another library exporting the same two functions (the product's
    reference-signature builder), driven the same way; the scene box always
    comes from the reference.
    -> (bbox, nodes T.cl_kdtree_node[], idx uint64[])"""
    lib()
    L = builder if builder is not None else lib()
    L.rt_kdtree_make_childs.restype = CountInfo
    L.rt_kdtree_make_childs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KdNode), C.c_void_p, C.c_int]
    L.rt_kdtree_pack_to_buffer.restype = CountInfo
    L.rt_kdtree_pack_to_buffer.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KdNode), C.c_uint64, C.c_uint64]
    verts = np.ascontiguousarray(verts, T.vertex)
    tris = np.ascontiguousarray(tris, T.triangle)
    n = len(tris)
    bb = scene_box(verts)
    root = C.cast(_libc.malloc(C.sizeof(KdNode)), C.POINTER(KdNode))
    prims = C.cast(_libc.malloc(8 * max(n, 1)), C.POINTER(C.c_uint64))
    for i in range(n):
        prims[i] = i
    root.contents.prims = prims
    root.contents.primsCount = n
    if n < 7:
        root.contents.isLast = 1
        cnt = CountInfo(n, 1)
    else:
        root.contents.isLast = 0
        cnt = L.rt_kdtree_make_childs(_p(verts), _p(tris), root, _p(bb), 0)
    nodes = np.zeros(cnt.nodesCount, T.cl_kdtree_node)
    idx = np.zeros(max(cnt.primsCount, 1), np.uint64)
    L.rt_kdtree_pack_to_buffer(_p(nodes), _p(idx), root, 0, 0)
    return bb[0], nodes, idx[:cnt.primsCount]


def compute_sah(prims, tris, verts, axis, box):
    """The reference's rt_kdtree_compute_sah (:774-857). -> (nL, nR, sep)"""
    L = lib()
    prims = np.ascontiguousarray(prims, np.uint64)
    bb = np.array([box], T.box)
    nl, nr, sep = C.c_uint64(), C.c_uint64(), C.c_float()
    L.rt_kdtree_compute_sah(_p(prims), _p(np.ascontiguousarray(tris, T.triangle)),
                            _p(np.ascontiguousarray(verts, T.vertex)), C.c_uint64(len(prims)),
                            C.c_int(axis), _p(bb), C.byref(nl), C.byref(nr), C.byref(sep))
    return nl.value, nr.value, sep.value


def camera_create(asp, hv):
    """The reference's rt_camera_create (rt_funcs_primitives.c:53-69)."""
    cam = np.zeros(1, T.camera)
    lib().rt_camera_create(_p(cam), C.c_float(asp), C.c_float(hv))
    return cam[0]


def matrix4(fn, *args):
    """Call one of the reference's rt_matrix4_create_* functions."""
    m = np.zeros((4, 4), np.float32)
    getattr(lib(), fn)(_p(m), *args)
    return m
