"""Python mirror of the C API (ctypes over librt_amd.so).

RenderPipe  - the reference's rt_render_pipe_* call sequence (create, add
              materials / lights / spheres / meshes, set camera, draw).
HipRenderer - the rt_hip_* layer underneath for callers that keep a scene
              resident in HBM (benchmark, multi-GPU driver, parity tests).
FlatTree    - the host kd builder's output in the HBM layout.

Every call goes to the native library; nothing here computes pixels.
"""
import ctypes as C

import numpy as np

from . import types as T
from ._lib import FlatTree as _FlatTreeStruct
from ._lib import DeviceTreeStruct, RenderPipeStruct, RtError, Stats, check, lib

RT_HIP_STATS, RT_HIP_HITIDS, RT_HIP_TIMING = 1, 2, 4
(RT_PARAM_STACK_SIZE, RT_PARAM_MAX_DEPTH, RT_PARAM_MAX_PRIMS_IN_NODE,
 RT_PARAM_BUILD_THREADS, RT_PARAM_TREE_CACHE, RT_PARAM_OVERLAP, RT_PARAM_GPU_BUILD,
 RT_PARAM_COOP_LIVE, RT_PARAM_COOP_STACK, RT_PARAM_LAUNCH_WAVES, RT_PARAM_PRUNE_LEAVES) = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11

kdnode = np.dtype([("a", "<u4"), ("b", "<u4")])
leaf_full = np.dtype([("first", "<u4"), ("count", "<u4")])
trirec = np.dtype([("nv0x", "<f4"), ("e1x", "<f4"), ("nv0y", "<f4"), ("e1y", "<f4"),
                   ("nv0z", "<f4"), ("e1z", "<f4"), ("e2x", "<f4"), ("e2y", "<f4"),
                   ("e2z", "<f4"), ("tri", "<u4"), ("mat", "<u4"), ("_pad", "<u4")])


def rec_vectors(recs):
    """(v0, e1, e2) of leaf records as [n, 3] float32 arrays."""
    v0 = -np.stack([recs["nv0x"], recs["nv0y"], recs["nv0z"]], axis=1)
    e1 = np.stack([recs["e1x"], recs["e1y"], recs["e1z"]], axis=1)
    e2 = np.stack([recs["e2x"], recs["e2y"], recs["e2z"]], axis=1)
    return v0, e1, e2
trishade = np.dtype([("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
                     ("mat", "<u4"), ("_pad", "<u4", 2)])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class DeviceTree:
    """kd-tree built on the GPU (rt_kdtree_device): lives in device memory
    until a renderer adopts it or release() frees it."""

    def __init__(self, struct):
        self._s = struct

    @staticmethod
    def stage(ticket, verts, tris, n_verts, n_tris, stream=None):
        """rt_kdtree_device_stage: the leading n_verts / n_tris entries go to the device ahead of the build that is
        given the ticket returned (0 on the first call; 0 comes back when the staging was lost).  The arrays must be
        the ones the build gets (no copies are made here) and stay as they are until then."""
        assert verts.dtype == T.vertex and tris.dtype == T.triangle and verts.flags.c_contiguous and tris.flags.c_contiguous
        L = lib()
        L.rt_kdtree_device_last_error.restype = C.c_char_p
        t = C.c_uint64(ticket)
        rc = L.rt_kdtree_device_stage(C.byref(t), _p(verts), C.c_uint64(n_verts), _p(tris), C.c_uint64(n_tris), C.c_void_p(stream))
        if rc:
            raise RtError(f"rt_kdtree_device_stage failed with code {rc}: {L.rt_kdtree_device_last_error().decode()}")
        return int(t.value)

    @classmethod
    def build(cls, verts, tris, max_depth=10, max_prims=7, stream=None, prune_leaves=0.0, ticket=0):
        from ._lib import BuildOpts
        verts = np.ascontiguousarray(verts, T.vertex)
        tris = np.ascontiguousarray(tris, T.triangle)
        s = DeviceTreeStruct()
        L = lib()
        L.rt_kdtree_device_last_error.restype = C.c_char_p
        o = BuildOpts(max_depth, max_prims, 0, 0.0, 0, prune_leaves, ticket)
        rc = L.rt_kdtree_build_device_ex(_p(verts), _p(tris), C.c_uint64(len(verts)), C.c_uint64(len(tris)),
                                         C.byref(o), C.c_void_p(stream), C.byref(s))
        if rc:
            raise RtError(f"rt_kdtree_build_device failed with code {rc}: {L.rt_kdtree_device_last_error().decode()}")
        return cls(s)

    @property
    def n_nodes(self):
        return int(self._s.nNodes)

    @property
    def n_recs(self):
        return int(self._s.nRecs)

    @property
    def depth(self):
        return int(self._s.depth)

    def download(self):
        """-> FlatTree copy on the host (for comparisons)."""
        s = _FlatTreeStruct()
        rc = lib().rt_kdtree_device_download(C.byref(self._s), C.byref(s))
        if rc:
            raise RtError(f"rt_kdtree_device_download failed with code {rc}")
        return FlatTree(s)

    def release(self):
        if self._s is not None:
            lib().rt_kdtree_device_release(C.byref(self._s))
            self._s = None


class FlatTree:
    """kd-tree in the HBM layout (rt_kdtree_flat).  Arrays are views into
    memory owned by the native library until release()."""

    def __init__(self, struct):
        self._s = struct
        self.nodes = _view(struct.nodes, struct.nNodes, kdnode)
        self.recs = _view(struct.recs, struct.nRecs, trirec)                  # the lists the nodes point to
        self.shade = _view(struct.shade, struct.nTris, trishade)
        # pruned leaf lists only: every leaf's complete list, stored behind `recs`, and where it lies per node
        self.leaf_full = _view(struct.leafFull, struct.nNodes, leaf_full) if struct.leafFull else None
        self.recs_all = _view(struct.recs, struct.nRecs + (struct.nRecsFull if struct.leafFull else 0), trirec)
        self.bbox = np.frombuffer(bytes(struct.bbox), T.box)[0]
        self.depth = struct.depth

    @classmethod
    def build(cls, verts, tris, max_depth=10, max_prims=7, threads=0, sah_stop=0.0, all_axes=0, prune_leaves=0.0):
        """sah_stop / all_axes / prune_leaves = 0 gives the reference's tree for the limits;
        prune_leaves > 0 (1e-4) keeps its nodes and drops leaf references that cannot be hit."""
        from ._lib import BuildOpts
        verts = np.ascontiguousarray(verts, T.vertex)
        tris = np.ascontiguousarray(tris, T.triangle)
        s = _FlatTreeStruct()
        o = BuildOpts(max_depth, max_prims, threads, sah_stop, all_axes, prune_leaves)
        rc = lib().rt_kdtree_build_flat_ex(_p(verts), _p(tris), C.c_uint64(len(verts)),
                                           C.c_uint64(len(tris)), C.byref(o), C.byref(s))
        if rc:
            raise RtError(f"rt_kdtree_build_flat failed with code {rc}")
        return cls(s)

    @classmethod
    def from_interchange(cls, nodes, idx, verts, tris, bbox):
        nodes = np.ascontiguousarray(nodes, T.cl_kdtree_node)
        idx = np.ascontiguousarray(idx, np.uint64)
        verts = np.ascontiguousarray(verts, T.vertex)
        tris = np.ascontiguousarray(tris, T.triangle)
        bb = np.array([bbox], T.box)
        s = _FlatTreeStruct()
        rc = lib().rt_kdtree_flatten(_p(nodes), C.c_uint64(len(nodes)), _p(idx), C.c_uint64(len(idx)),
                                     _p(verts), _p(tris), C.c_uint64(len(verts)), C.c_uint64(len(tris)),
                                     _p(bb), C.byref(s))
        if rc:
            raise RtError(f"rt_kdtree_flatten failed with code {rc}")
        return cls(s)

    def expand(self):
        """-> (nodes T.cl_kdtree_node[], idx uint64[]) in the reference's flat format."""
        pn, pi = C.c_void_p(), C.c_void_p()
        nn, ni = C.c_uint64(), C.c_uint64()
        rc = lib().rt_kdtree_flat_expand(C.byref(self._s), C.byref(pn), C.byref(nn), C.byref(pi), C.byref(ni))
        if rc:
            raise RtError(f"rt_kdtree_flat_expand failed with code {rc}")
        nodes = _view(pn.value, nn.value, T.cl_kdtree_node).copy()
        idx = _view(pi.value, ni.value, np.dtype("<u8")).copy()
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(pn)
        libc.free(pi)
        return nodes, idx

    def release(self):
        if self._s is not None:
            lib().rt_kdtree_flat_release(C.byref(self._s))
            self._s = None
            self.nodes = self.recs = self.shade = self.leaf_full = self.recs_all = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class HipRenderer:
    """One rt_hip_ctx: a scene resident on one GPU."""

    def __init__(self, device=0, stream=None):
        self._L = lib()
        self._c = self._L.rt_hip_create(device)
        if not self._c:
            raise RtError("rt_hip_create: " + self._L.rt_hip_last_error().decode())
        self._c = C.c_void_p(self._c)
        self.w = self.h = 0
        if stream is not None:
            check(self._L.rt_hip_set_stream(self._c, C.c_void_p(stream)), "rt_hip_set_stream")

    def close(self):
        if self._c:
            self._L.rt_hip_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ----------------------------------------------------------------
    def upload_tree(self, tree):
        full = getattr(tree, "leaf_full", None)
        if full is None:
            check(self._L.rt_hip_upload_tree(self._c, _p(tree.nodes), C.c_uint64(len(tree.nodes)),
                                             _p(tree.recs), C.c_uint64(len(tree.recs)),
                                             _p(tree.shade), C.c_uint64(len(tree.shade)),
                                             _p(np.array([tree.bbox], T.box)), C.c_int(tree.depth)),
                  "rt_hip_upload_tree")
            return
        check(self._L.rt_hip_upload_tree_lists(self._c, _p(tree.nodes), C.c_uint64(len(tree.nodes)),
                                               _p(tree.recs_all), C.c_uint64(len(tree.recs)),
                                               C.c_uint64(len(tree.recs_all) - len(tree.recs)), _p(full),
                                               _p(tree.shade), C.c_uint64(len(tree.shade)),
                                               _p(np.array([tree.bbox], T.box)), C.c_int(tree.depth)),
              "rt_hip_upload_tree_lists")

    def upload_materials(self, mats):
        mats = np.ascontiguousarray(mats, T.material)
        check(self._L.rt_hip_upload_materials(self._c, _p(mats), C.c_uint64(len(mats))), "rt_hip_upload_materials")

    def upload_spheres(self, sph):
        sph = np.ascontiguousarray(sph, T.sphere)
        keep = sph if len(sph) else np.zeros(1, T.sphere)
        check(self._L.rt_hip_upload_spheres(self._c, _p(keep), C.c_uint64(len(sph))), "rt_hip_upload_spheres")

    def upload_lights(self, lts):
        lts = np.ascontiguousarray(lts, T.point_light)
        keep = lts if len(lts) else np.zeros(1, T.point_light)
        check(self._L.rt_hip_upload_lights(self._c, _p(keep), C.c_uint64(len(lts))), "rt_hip_upload_lights")

    def set_camera(self, cam):
        cam = np.array([cam], T.camera)
        check(self._L.rt_hip_set_camera(self._c, _p(cam)), "rt_hip_set_camera")

    def set_image(self, w, h):
        check(self._L.rt_hip_set_image(self._c, C.c_int(w), C.c_int(h)), "rt_hip_set_image")
        self.w, self.h = w, h

    def set_tile(self, band_h, n_parts, part):
        check(self._L.rt_hip_set_tile(self._c, C.c_int(band_h), C.c_int(n_parts), C.c_int(part)), "rt_hip_set_tile")

    def set_param(self, key, value):
        check(self._L.rt_hip_set_param(self._c, C.c_int(key), C.c_int(value)), "rt_hip_set_param")

    @property
    def local_rows(self):
        return self._L.rt_hip_local_rows(self._c)

    def adopt_tree(self, dtree):
        """Install a DeviceTree without a copy; the renderer owns its buffers afterwards."""
        check(self._L.rt_hip_adopt_tree(self._c, C.byref(dtree._s)), "rt_hip_adopt_tree")
        dtree._s = None

    def stage_geometry(self, verts, n_verts, tris, n_tris):
        """rt_hip_stage_geometry: the leading entries of the arrays the next build_tree is given go to the device now
        (asynchronously; the arrays are used as they are and must stay unchanged until that build)."""
        assert verts.dtype == T.vertex and tris.dtype == T.triangle and verts.flags.c_contiguous and tris.flags.c_contiguous
        rc = self._L.rt_hip_stage_geometry(self._c, _p(verts), C.c_uint64(n_verts), _p(tris), C.c_uint64(n_tris))
        if rc == 2:
            return False          # what had gone ahead was lost to other geometry: the build copies everything
        check(rc, "rt_hip_stage_geometry")
        return True

    def unstage_geometry(self):
        self._L.rt_hip_unstage_geometry(self._c)

    def build_tree(self, verts, tris, max_depth=10, max_prims=7):
        """rt_hip_build_tree: the tree of this geometry, built on the GPU and installed."""
        verts = np.ascontiguousarray(verts, T.vertex)
        tris = np.ascontiguousarray(tris, T.triangle)
        check(self._L.rt_hip_build_tree(self._c, _p(verts), _p(tris), C.c_uint64(len(verts)), C.c_uint64(len(tris)),
                                        C.c_int(max_depth), C.c_int(max_prims)), "rt_hip_build_tree")

    def share_scene(self, other, scene):
        """Render the scene resident in `other` (same GPU) without a copy of it: tree, materials, spheres, lights
        are read from there; camera, image size and ray budget are set from `scene`."""
        check(self._L.rt_hip_share_scene(self._c, other._c), "rt_hip_share_scene")
        self.set_camera(scene.camera)
        self.set_image(scene.w, scene.h)
        self.set_param(RT_PARAM_STACK_SIZE, scene.stack_size)

    def load_scene(self, scene, tree=None, max_depth=10, max_prims=7, threads=0):
        """Upload everything `scene` holds; builds the tree unless given. -> tree"""
        if tree is None:
            v, t = scene.flatten()
            tree = FlatTree.build(v, t, max_depth, max_prims, threads)
        if isinstance(tree, DeviceTree):
            self.adopt_tree(tree)
        else:
            self.upload_tree(tree)
        self.upload_materials(scene.material_table())
        self.upload_spheres(scene.sphere_table())
        self.upload_lights(scene.light_table())
        self.set_camera(scene.camera)
        self.set_image(scene.w, scene.h)
        self.set_param(RT_PARAM_STACK_SIZE, scene.stack_size)
        return tree

    # -- frame ----------------------------------------------------------------
    def render(self, flags=0, d_out=None):
        """Enqueue one frame (asynchronous).  d_out: device pointer (int) or None."""
        check(self._L.rt_hip_render(self._c, C.c_void_p(d_out) if d_out else None, C.c_uint(flags)), "rt_hip_render")

    def sync(self):
        check(self._L.rt_hip_sync(self._c), "rt_hip_sync")

    def read_frame(self, out=None):
        """-> uint8 [local_rows, w, 4] in b,g,r,a order (into `out` when given: same shape, contiguous)"""
        if out is None:
            out = np.zeros((self.local_rows, self.w, 4), np.uint8)
        elif out.shape != (self.local_rows, self.w, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise RtError("read_frame: `out` must be a contiguous uint8 array of shape (local_rows, w, 4)")
        check(self._L.rt_hip_read_frame(self._c, _p(out)), "rt_hip_read_frame")
        return out

    def alloc_host_frame(self):
        """A page-locked uint8 [local_rows, w, 4] block for read_frame(out=...): read-backs into it run at link speed
        and need no staging copy.  Stays valid until free_host_frame() or close()."""
        n = self.local_rows * self.w * 4
        self._L.rt_hip_alloc_host.restype = C.c_void_p
        self._L.rt_hip_alloc_host.argtypes = [C.c_void_p, C.c_size_t]
        p = self._L.rt_hip_alloc_host(self._c, n)
        if not p:
            raise RtError("rt_hip_alloc_host: " + self._L.rt_hip_last_error().decode())
        a = np.frombuffer((C.c_uint8 * n).from_address(p), np.uint8).reshape(self.local_rows, self.w, 4)
        self._host_blocks = getattr(self, "_host_blocks", {})
        self._host_blocks[a.ctypes.data] = p
        return a

    def free_host_frame(self, a):
        p = self._host_blocks.pop(a.ctypes.data)
        self._L.rt_hip_free_host.argtypes = [C.c_void_p, C.c_void_p]
        self._L.rt_hip_free_host(self._c, C.c_void_p(p))

    def read_hit_ids(self):
        out = np.zeros((self.local_rows, self.w), np.int32)
        check(self._L.rt_hip_read_hit_ids(self._c, _p(out)), "rt_hip_read_hit_ids")
        return out

    def stats(self):
        st = Stats()
        check(self._L.rt_hip_get_stats(self._c, C.byref(st)), "rt_hip_get_stats")
        return st.as_dict()

    def kernel_times(self):
        """After render(RT_HIP_TIMING): ({closest, shadow, shade, fold, frame} ms, launch counts)."""
        ms = (C.c_float * 5)()
        n = (C.c_int * 5)()
        check(self._L.rt_hip_kernel_times(self._c, ms, n), "rt_hip_kernel_times")
        names = ("closest", "shadow", "shade", "fold", "frame")
        return dict(zip(names, [float(x) for x in ms])), dict(zip(names, [int(x) for x in n]))

    def kernel_times_wall(self):
        """kernel_times() plus, per kernel, the wall-clock ms per frame during which at least one of its launches ran."""
        ms = (C.c_float * 5)()
        n = (C.c_int * 5)()
        wall = (C.c_float * 4)()
        check(self._L.rt_hip_kernel_times_ex(self._c, ms, n, wall), "rt_hip_kernel_times_ex")
        names = ("closest", "shadow", "shade", "fold", "frame")
        return (dict(zip(names, [float(x) for x in ms])), dict(zip(names, [int(x) for x in n])),
                dict(zip(names[:4], [float(x) for x in wall])))

    def kernel_intervals(self, origin=None):
        """[(kind, start_ms, end_ms)] of every launch of the timed frames since the last kernel_times*() call, on the
        clock of `origin` (a renderer on the same GPU; default: this one).  kind: closest, shadow, shade, fold, frame."""
        cap = 32 * (3 * 8 + 2)
        iv = (C.c_float * (2 * cap))()
        kind = (C.c_int * cap)()
        n = C.c_int(0)
        check(self._L.rt_hip_kernel_intervals(self._c, origin._c if origin is not None else None, iv, kind, cap, C.byref(n)),
              "rt_hip_kernel_intervals")
        names = ("closest", "shadow", "shade", "fold", "frame")
        return [(names[kind[i]], float(iv[2 * i]), float(iv[2 * i + 1])) for i in range(n.value)]

    def selftest_rcp(self):
        """(mismatches, fast-path inputs) of the exact-reciprocal shortcut over all 2^32 floats."""
        bad, fast = C.c_uint64(0), C.c_uint64(0)
        check(self._L.rt_hip_selftest_rcp(self._c, C.byref(bad), C.byref(fast)), "rt_hip_selftest_rcp")
        return int(bad.value), int(fast.value)


class RenderPipe:
    """The reference's render-pipe call sequence (rt_funcs_render_pipe.h)."""

    def __init__(self, w, h, device=0):
        self._L = lib()
        self._rp = RenderPipeStruct()
        self._L.rt_render_pipe_select_device(C.c_int(device))
        self._L.rt_render_pipe_create(C.byref(self._rp), C.c_int(w), C.c_int(h))
        self._open = True

    def set_image_size(self, w, h):
        self._L.rt_render_pipe_set_image_size(C.byref(self._rp), C.c_int(w), C.c_int(h))

    def set_param(self, key, value):
        self._L.rt_render_pipe_set_param(C.byref(self._rp), C.c_int(key), C.c_int(value))

    def set_tile(self, band_h, n_parts, part):
        self._L.rt_render_pipe_set_tile(C.byref(self._rp), C.c_int(band_h), C.c_int(n_parts), C.c_int(part))

    def set_camera(self, cam):
        cam = np.array([cam], T.camera)
        self._L.rt_render_pipe_set_camera(C.byref(self._rp), _p(cam))

    def add_material(self, mat, ind):
        m = np.array([mat], T.material)
        self._L.rt_render_pipe_add_material(C.byref(self._rp), _p(m), C.c_ulong(ind))

    def add_light(self, light, lt=T.RT_LT_POINT):
        l = np.array([light], T.point_light)
        self._L.rt_render_pipe_add_light(C.byref(self._rp), _p(l), C.c_int(lt))

    def add_primitive(self, prim, pt=T.RT_PT_SPHERE):
        s = np.array([prim], T.sphere)
        self._L.rt_render_pipe_add_primitive(C.byref(self._rp), _p(s), C.c_int(pt))

    def add_triangles(self, verts, tris, v_offset, t_offset, v_count, t_count, group):
        verts = np.ascontiguousarray(verts, T.vertex)
        tris = np.ascontiguousarray(tris, T.triangle)
        self._L.rt_render_pipe_add_triangles(C.byref(self._rp), _p(verts), _p(tris), C.c_uint64(v_offset),
                                             C.c_uint64(t_offset), C.c_uint64(v_count), C.c_uint64(t_count),
                                             C.c_uint64(group))

    def mesh_add(self, mesh):
        """rt_mesh_add: the whole mesh with its material slot."""
        self.add_triangles(mesh.verts, mesh.tris, 0, 0, len(mesh.verts), len(mesh.tris), mesh.group)

    def add_scene(self, scene):
        """The per-frame add sequence of the reference's draw callback
        (main.c:674-716): materials, lights, spheres, meshes."""
        for k in sorted(scene.materials):
            self.add_material(scene.materials[k], k)
        for l in scene.lights:
            self.add_light(l)
        for s in scene.spheres:
            self.add_primitive(s)
        for m in scene.meshes:
            self.mesh_add(m)

    @property
    def counts(self):
        r = self._rp
        return dict(triangles=r.trianglesCount, vertices=r.vertexCount, prims=r.primsCount, lights=r.lightsCount)

    def kdtree_build(self):
        """rt_kdtree_build over the current geometry -> (bbox, nodes, idx)."""
        bb = np.zeros(1, T.box)
        pn, pi = C.c_void_p(), C.c_void_p()
        nn, ni = C.c_uint64(), C.c_uint64()
        self._L.rt_kdtree_build(C.byref(self._rp), _p(bb), C.byref(pn), C.byref(nn), C.byref(pi), C.byref(ni))
        nodes = _view(pn.value, nn.value, T.cl_kdtree_node).copy()
        idx = _view(pi.value, ni.value, np.dtype("<u8")).copy()
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(pn)
        libc.free(pi)
        return bb[0], nodes, idx

    def draw(self):
        """rt_render_pipe_draw -> uint8 [h, w, 4] (b,g,r,a); consumes the scene."""
        ptr = self._L.rt_render_pipe_draw(C.byref(self._rp))
        w, h = self._rp.w, self._rp.h
        return _view(ptr, w * h * 4, np.dtype("u1")).reshape(h, w, 4).copy()

    def cleanup(self):
        if self._open:
            self._L.rt_render_pipe_cleanup(C.byref(self._rp))
            self._open = False

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
