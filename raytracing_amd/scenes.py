"""Deterministic synthetic scenes for the benchmark configurations.

A Scene holds exactly what a caller of the render-pipe API passes in:
materials by slot, point lights, spheres, meshes (local vertex indices + the
material slot they are added with) and a camera.  `flatten()` applies the
index rebasing that rt_render_pipe_add_triangles performs
(reference rt_funcs_render_pipe.c:292-298).

No random numbers come from the C library: colours and orbit coefficients of
the default scene use a fixed LCG instead of rand() (the reference seeds with
time(0), main.c:765), and the water uses the 7 waves the reference defines
(its loop bound of 10 reads past the array, main.c:310-311).
"""
from dataclasses import dataclass, field
import math

import numpy as np

from . import types as T


@dataclass
class Mesh:
    verts: np.ndarray          # T.vertex[]
    tris: np.ndarray           # T.triangle[], indices local to verts
    group: int                 # material slot passed to rt_mesh_add


@dataclass
class Scene:
    name: str
    w: int
    h: int
    camera: np.ndarray                      # T.camera scalar
    materials: dict = field(default_factory=dict)   # slot -> T.material scalar
    lights: list = field(default_factory=list)      # T.point_light scalars
    spheres: list = field(default_factory=list)     # T.sphere scalars
    meshes: list = field(default_factory=list)      # Mesh, in add order
    stack_size: int = 6                     # rays per pixel (STACK_SIZE)

    def flatten(self):
        """-> (verts, tris) with global vertex indices and mat = group."""
        vs, ts, base = [], [], 0
        for m in self.meshes:
            t = m.tris.copy()
            for k in ("pV0", "pV1", "pV2"):
                t[k] += base
            t["mat"] = m.group
            vs.append(m.verts)
            ts.append(t)
            base += len(m.verts)
        if not vs:
            return np.zeros(0, T.vertex), np.zeros(0, T.triangle)
        return np.concatenate(vs), np.concatenate(ts)

    def material_table(self):
        n = (max(self.materials) + 1) if self.materials else 1
        tab = np.zeros(n, T.material)
        for k, v in self.materials.items():
            tab[k] = v
        return tab

    def sphere_table(self):
        return np.array(self.spheres, dtype=T.sphere) if self.spheres else np.zeros(0, T.sphere)

    def light_table(self):
        return np.array(self.lights, dtype=T.point_light) if self.lights else np.zeros(0, T.point_light)

    @property
    def n_tris(self):
        return sum(len(m.tris) for m in self.meshes)


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------

def make_camera(w, h, hv=0.125 * math.pi, world=None):
    """rt_camera_create (reference rt_funcs_primitives.c:53-69) +
    rt_matrix4_create_projection (rt_funcs_math.c:127-137): float arguments,
    tan() evaluated in double, result rounded to float."""
    cam = np.zeros((), T.camera)
    asp = np.float32(np.float32(w) / np.float32(h))
    hvf = np.float32(hv)
    t = math.tan(float(np.float32(hvf / np.float32(2.0))))
    p = np.zeros((4, 4), np.float32)
    p[0, 0] = np.float32(2.0 * t * float(asp))
    p[1, 1] = np.float32(2.0 * t)
    p[2, 2] = p[3, 3] = 1.0
    cam["viewToPersp"] = p
    cam["world"] = np.eye(4, dtype=np.float32) if world is None else np.asarray(world, np.float32)
    return cam


def make_light(pos, rad, col=(1.0, 1.0, 1.0, 1.0)):
    l = np.zeros((), T.point_light)
    l["col"] = T.make_color(*col)
    l["pos"]["x"], l["pos"]["y"], l["pos"]["z"] = pos
    l["rad"] = rad
    return l


def make_sphere(pos, rad, mat):
    s = np.zeros((), T.sphere)
    s["pos"]["x"], s["pos"]["y"], s["pos"]["z"] = pos
    s["rad"] = rad
    s["mat"] = mat
    return s


def _set_xyz(dst, xyz):
    dst["x"], dst["y"], dst["z"] = xyz[..., 0], xyz[..., 1], xyz[..., 2]


def accumulate_normals(pos, tri_idx, start):
    """The rule of rt_restore_normals / buildPlaneOfTriangles
    (reference rt_meshes.c:77-107, main.c:397-423): triangle by triangle, add
    cross(v1-v0, v2-v0) to the triangle's three vertices in turn (fp32, so the
    per-vertex summation order is ascending triangle index), then normalise."""
    n = np.array(start, dtype=np.float32).copy()
    p = pos.astype(np.float32)
    e1 = p[tri_idx[:, 1]] - p[tri_idx[:, 0]]
    e2 = p[tri_idx[:, 2]] - p[tri_idx[:, 0]]
    c = np.cross(e1, e2).astype(np.float32)
    # ufunc.at applies the additions one by one in the order given
    np.add.at(n, tri_idx.ravel(), np.repeat(c, 3, axis=0))
    ln = np.sqrt((n * n).sum(axis=1, dtype=np.float32)).astype(np.float32)
    return (n / ln[:, None]).astype(np.float32)


def grid_triangles(rows, cols):
    """Two triangles per quad of a (rows+1) x (cols+1) vertex lattice, in the
    winding of the reference's water mesh (main.c:378-395)."""
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    i, j = i.ravel(), j.ravel()
    pr = cols + 1
    a = np.stack([pr * i + j, pr * (i + 1) + j, pr * i + (j + 1)], axis=1)
    b = np.stack([pr * (i + 1) + (j + 1), pr * i + (j + 1), pr * (i + 1) + j], axis=1)
    out = np.empty((2 * len(i), 3), np.int64)
    out[0::2], out[1::2] = a, b
    return out


def _mesh_from(pos, nrm, tri_idx, group):
    v = np.zeros(len(pos), T.vertex)
    _set_xyz(v["pos"], pos.astype(np.float32))
    _set_xyz(v["norm"], nrm.astype(np.float32))
    t = np.zeros(len(tri_idx), T.triangle)
    t["pV0"], t["pV1"], t["pV2"] = tri_idx[:, 0], tri_idx[:, 1], tri_idx[:, 2]
    t["mat"] = group
    return Mesh(v, t, group)


# ---------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------

def displaced_sphere_mesh(lon, lat, group, radius=40.0, center=(0.0, 0.0, 110.0)):
    """UV sphere with `lon` x `lat` quads (2*lon*lat triangles), latitude in
    [0.02pi, 0.98pi], analytic two-octave displacement; the seam column is a
    bit-exact copy of column 0."""
    th = np.linspace(0.0, 2.0 * math.pi, lon + 1)
    ph = np.linspace(0.02 * math.pi, 0.98 * math.pi, lat + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")          # [lon+1, lat+1]
    r = radius * (1.0 + 0.04 * np.sin(9 * TH) * np.sin(7 * PH)
                  + 0.02 * np.sin(23 * TH + 1.0) * np.sin(19 * PH + 2.0))
    pos = np.stack([center[0] + r * np.sin(PH) * np.cos(TH),
                    center[1] + r * np.cos(PH),
                    center[2] + r * np.sin(PH) * np.sin(TH)], axis=-1).astype(np.float32)
    pos[lon] = pos[0]
    pos = pos.reshape(-1, 3)
    tri = grid_triangles(lon, lat)
    nrm = accumulate_normals(pos, tri, np.zeros_like(pos))
    return _mesh_from(pos, nrm, tri, group)


def floor_mesh(group, y=45.0, half=120.0, zc=110.0):
    pos = np.array([[half, y, zc + half], [half, y, zc - half],
                    [-half, y, zc + half], [-half, y, zc - half]], np.float32)
    nrm = np.tile(np.array([0.0, -1.0, 0.0], np.float32), (4, 1))
    tri = np.array([[0, 1, 2], [3, 1, 2]], np.int64)
    return _mesh_from(pos, nrm, tri, group)


def sphere_scene(lon, lat, w, h, reflect=0.0, alpha=1.0, opt_dens=1.0, stack_size=6, name=None,
                 world=None):
    """Displaced sphere + floor, one white light, one material in slot 0."""
    sc = Scene(name or f"sphere_{2 * lon * lat}", w, h, make_camera(w, h, world=world),
               stack_size=stack_size)
    sc.materials[0] = T.make_material((alpha, 0.8, 0.8, 0.8), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 0.6, 0.6, 0.6), (20.0, 0.3, 0.3, 0.3),
                                      (0.0, reflect, reflect, reflect), 0.05, opt_dens)
    sc.lights.append(make_light((-20.0, -35.0, 0.0), 100.0))
    sc.meshes.append(displaced_sphere_mesh(lon, lat, 0))
    sc.meshes.append(floor_mesh(0))
    return sc


def config2(w=1920, h=1080):
    """100 000-triangle sphere, primary + shadow rays only."""
    return sphere_scene(250, 200, w, h, reflect=0.0, name="c2_sphere100k")


def config3(w=3840, h=2160):
    """1 000 000-triangle sphere, opaque reflective, 5 rays per pixel
    (= primary + 4 bounces)."""
    return sphere_scene(1000, 500, w, h, reflect=0.5, stack_size=5, name="c3_sphere1m")


def orbit_world(frame, n_frames=240, radius=110.0, height=-20.0, target=(0.0, 0.0, 110.0)):
    """Camera `frame` of the fly-through: on a circle around `target`,
    looking at it.  Rows of `world` are the camera's right/down/forward axes
    (row-vector convention, +y is down the image)."""
    ang = 2.0 * math.pi * frame / n_frames
    eye = np.array([target[0] + radius * math.sin(ang), height, target[2] - radius * math.cos(ang)])
    fwd = np.array(target) - eye
    fwd /= np.linalg.norm(fwd)
    down = np.array([0.0, 1.0, 0.0])
    right = np.cross(down, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3] = right, down, fwd, eye
    return m


def config5_frame(frame, w=3840, h=2160, lon=1000, lat=500):
    """Fly-through frame: same mesh, reflective + refractive material."""
    return sphere_scene(lon, lat, w, h, reflect=0.5, alpha=0.5, opt_dens=1.33, stack_size=6,
                        name=f"c5_frame{frame}", world=orbit_world(frame))


def config5(w=3840, h=2160):
    """One representative frame of the fly-through (camera a quarter of the way round)."""
    return config5_frame(60, w, h)


def wavy_grid_scene(n=32, w=96, h=64, with_sphere=True, reflect=0.3, alpha=0.5):
    """Small fixture scene: n x n wavy quad grid (2 n^2 triangles, reflective
    and refractive like the reference's water), a matte back wall, one
    sphere, one light."""
    sc = Scene(f"wavy{n}", w, h, make_camera(w, h))
    sc.materials[0] = T.make_material((alpha, 0.223, 0.345, 0.474), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 0.5, 0.5, 0.5), (3.5, 0.25, 0.25, 0.25),
                                      (0.0, reflect, reflect, reflect), 0.25, 1.33)
    sc.materials[1] = T.make_material((0.5, 0.0, 0.5, 1.0), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 0.2, 0.2, 0.2), (15.0, 0.2, 0.2, 0.2),
                                      (0.0, 0.5, 0.5, 0.5), 0.0, 1.33)
    sc.materials[2] = T.make_material((1.0, 1.0, 1.0, 1.0), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 1.0, 1.0, 1.0), (5.0, 1.0, 1.0, 1.0),
                                      (0.0, 0.0, 0.0, 0.0), 0.0, 1.0)
    sc.lights.append(make_light((-20.0, -35.0, 0.0), 100.0))
    if with_sphere:
        sc.spheres.append(make_sphere((0.0, 0.0, 50.0), 8.0, 1))
    u = np.linspace(0.0, 1.0, n + 1)
    U, V = np.meshgrid(u, u, indexing="ij")
    y = 12.0 + 1.5 * np.sin(9.0 * U + 0.3) * np.cos(7.0 * V) + 0.5 * np.sin(23.0 * V + 1.0)
    pos = np.stack([-40.0 + 80.0 * U, y, 20.0 + 80.0 * V], axis=-1).reshape(-1, 3).astype(np.float32)
    tri = grid_triangles(n, n)
    nrm = accumulate_normals(pos, tri, np.tile(np.array([0.0, -1.0, 0.0], np.float32), (len(pos), 1)))
    sc.meshes.append(_mesh_from(pos, nrm, tri, 0))
    wall = np.array([[-60.0, -40.0, 105.0], [60.0, -40.0, 105.0], [-60.0, 30.0, 105.0],
                     [60.0, 30.0, 105.0]], np.float32)
    sc.meshes.append(_mesh_from(wall, np.tile(np.array([0.0, 0.0, -1.0], np.float32), (4, 1)),
                                np.array([[0, 1, 2], [3, 2, 1]], np.int64), 2))
    return sc


class _LCG:
    """Fixed-seed stand-in for rand() so the default scene is reproducible."""

    def __init__(self, seed=12345):
        self.s = seed

    def rand(self):
        self.s = (self.s * 1103515245 + 12345) & 0x7FFFFFFF
        return self.s >> 8


WAVES = [  # period, amplitude, angle, speed (reference main.c:330-349)
    (100.0, 1.5, math.pi * 0.5, 500.0), (79.16677, 0.25, 2.9428474, 256.9786),
    (144.04169, 0.25, -1.8729727, 160.6270), (79.15634, 0.25, 1.8583484, 239.9041),
    (121.51403, 0.5, -1.5343947, 230.2120), (72.40281, 0.5, 1.4335813, 300.9681),
    (81.57271, 0.5, -1.2057098, 280.5781),
]


def default_scene(w=1280, h=768, t=0.0, squares=150, seed=12345):
    """Headless restatement of the reference demo frame (main.c:306-463,
    465-611, 656-716, 810-837): water mesh, open box, ten spheres, one light."""
    world = np.array([[1, 0, 0, 0], [0, 0.968582, -0.248690, 0],
                      [0, 0.248690, 0.968582, 0], [0, -12.0, -43.0, 1]], np.float32)
    sc = Scene("default", w, h, make_camera(w, h, world=world))
    rng = _LCG(seed)
    pl_sz, pl_ht = 150.0, 20.0

    # materials: 0 water, 1 centre sphere, 2..10 orbiting spheres, 11 box
    sc.materials[0] = T.make_material((0.5, 0.223, 0.345, 0.474), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 0.5, 0.5, 0.5), (3.5, 0.25, 0.25, 0.25),
                                      (0.0, 0.3, 0.3, 0.3), 0.25, 1.33)
    sc.materials[1] = T.make_material((0.5, 0.0, 0.5, 1.0), (0.0, 0.05, 0.05, 0.05),
                                      (0.0, 0.2, 0.2, 0.2), (15.0, 0.2, 0.2, 0.2),
                                      (0.0, 0.5, 0.5, 0.5), 0.0, 1.33)
    koef = [0.0]
    for i in range(1, 10):
        c = [(rng.rand() % 16365) / 16384.0 for _ in range(3)]
        sc.materials[1 + i] = T.make_material((1.0, c[0], c[1], c[2]), (0.0, 0.05, 0.05, 0.05),
                                              (0.0, 0.5, 0.5, 0.5), (15.0, 0.5, 0.5, 0.5),
                                              (0.0, 0.5, 0.5, 0.5), 0.0, 1.0)
        koef.append(float(rng.rand() % 101))
    sc.materials[11] = T.make_material((1.0, 1.0, 1.0, 1.0), (0.0, 0.05, 0.05, 0.05),
                                       (0.0, 1.0, 1.0, 1.0), (5.0, 1.0, 1.0, 1.0),
                                       (0.0, 0.0, 0.0, 0.0), 0.0, 1.0)
    sc.lights.append(make_light((-20.0, -35.0, 0.0), 100.0))

    # spheres (main.c:686-708)
    sc.spheres.append(make_sphere((0.0, 0.0, 50.0), 10.0, 1))
    radius = 25.0 + 25.0 * (0.5 + math.sin(10.0 * t) * 0.5)
    for i in range(1, 10):
        a = t + (i - 1) * 2.0 / 9.0 * math.pi
        sc.spheres.append(make_sphere((math.sin(a) * radius, koef[i] * 0.5 * math.sin(koef[i] * 0.5 * t),
                                       50.0 + math.cos(a) * radius), 3.66541, 1 + i))

    # box, added first with material slot 11 (main.c:465-611, 711)
    top, bot = pl_ht - 3.0, pl_ht - 3.0 + 25.0
    hx, z0, z1 = pl_sz * 0.5, -pl_sz * 0.5 + 50.0, pl_sz * 0.5 + 50.0
    bpos = np.array([
        [hx, bot, z1], [hx, bot, z0], [-hx, bot, z1], [-hx, bot, z0],
        [-hx, bot, z1], [-hx, bot, z0], [-hx, top, z1], [-hx, top, z0],
        [hx, bot, z1], [hx, bot, z0], [hx, top, z1], [hx, top, z0],
        [-hx, bot, z1], [hx, bot, z1], [-hx, top, z1], [hx, top, z1],
        [-hx, bot, z0], [hx, bot, z0], [-hx, top, z0], [hx, top, z0]], np.float32)
    bn = np.repeat(np.array([[0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, -1], [0, 0, 1]], np.float32), 4, axis=0)
    btri = np.array([[0, 1, 2], [3, 1, 2], [4, 5, 6], [7, 6, 5], [8, 9, 10], [11, 10, 9],
                     [12, 13, 14], [15, 14, 13], [16, 17, 18], [19, 18, 17]], np.int64)
    sc.meshes.append(_mesh_from(bpos, bn, btri, 11))

    # water (main.c:351-423), slot 0
    if squares:
        pr = squares + 1
        i, j = np.meshgrid(np.arange(pr), np.arange(pr), indexing="ij")
        u = (i / pr).astype(np.float32).astype(np.float64)
        v = (j / pr).astype(np.float32).astype(np.float64)
        y = np.full(u.shape, pl_ht, np.float32)
        for per, amp, ang, spd in WAVES:
            per, amp, ang, spd = (float(np.float32(x)) for x in (per, amp, ang, spd))
            # t * speed is a float * float product in the reference (main.c:362-364)
            ts = float(np.float32(np.float32(t) * np.float32(spd)))
            y = (y.astype(np.float64) + np.sin(per * (-math.cos(ang) * u + math.sin(ang) * v)
                                               + ts) * amp).astype(np.float32)
        dt = np.float32(pl_sz / squares)
        pos = np.stack([np.float32(-pl_sz * 0.5) + i.astype(np.float32) * dt, y,
                        np.float32(50.0 - pl_sz * 0.5) + j.astype(np.float32) * dt],
                       axis=-1).reshape(-1, 3).astype(np.float32)
        tri = grid_triangles(squares, squares)
        nrm = accumulate_normals(pos, tri, np.tile(np.array([0.0, -1.0, 0.0], np.float32), (len(pos), 1)))
        sc.meshes.append(_mesh_from(pos, nrm, tri, 0))
    return sc
