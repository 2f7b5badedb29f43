"""numpy mirrors of the POD structs in include/rt_types.h.

Layouts follow the reference ABI (reference rt_types.h:43-200); the ABI test
compares every itemsize/offset here with the C header and with the layout
dumped from the reference's own header.
"""
import numpy as np

vec3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("_pad", "<f4")])
vertex = np.dtype([("pos", vec3), ("norm", vec3)])
triangle = np.dtype([("pV0", "<u8"), ("pV1", "<u8"), ("pV2", "<u8"), ("mat", "<u8")])
color = np.dtype([("b", "<f4"), ("g", "<f4"), ("r", "<f4"), ("a", "<f4")])
material = np.dtype([
    ("color", color), ("ambient", color), ("diffuse", color), ("specular", color),
    ("reflect", color), ("lightFalloff", "<f4"), ("optDens", "<f4"),
])
sphere = np.dtype({"names": ["pos", "rad", "mat"], "formats": [vec3, "<f4", "<u8"],
                   "offsets": [0, 16, 24], "itemsize": 32})
point_light = np.dtype({"names": ["col", "pos", "rad"], "formats": [color, vec3, "<f4"],
                        "offsets": [0, 16, 32], "itemsize": 48})
box = np.dtype([("center", vec3), ("extents", vec3)])
matrix4 = np.dtype(("<f4", (4, 4)))
camera = np.dtype([("world", "<f4", (4, 4)), ("viewToPersp", "<f4", (4, 4))])
argb = np.dtype([("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1")])
raytrace_args = np.dtype([
    ("boundingBox", box), ("primsCount", "<u8"), ("trianglesCount", "<u8"),
    ("vertexCount", "<u8"), ("lightsCount", "<u8"),
    ("w", "<i4"), ("h", "<i4"), ("xdelta", "<i4"), ("ydelta", "<i4"),
])
prim_desc = np.dtype({"names": ["offset", "pt"], "formats": ["<u8", "<i4"],
                      "offsets": [0, 8], "itemsize": 16})
light_desc = np.dtype({"names": ["offset", "lt"], "formats": ["<u8", "<i4"],
                       "offsets": [0, 8], "itemsize": 16})
cl_kdtree_node = np.dtype({
    "names": ["leftNode", "rightNode", "primsCount", "prims", "sep", "axis", "isLast"],
    "formats": ["<u8", "<u8", "<u8", "<u8", "<f4", "<i4", "<i4"],
    "offsets": [0, 8, 16, 24, 32, 36, 40], "itemsize": 48})

RT_PT_SPHERE, RT_PT_PLANE, RT_PT_TRIANGLE = 0, 1, 2
RT_LT_POINT, RT_LT_DIRECTED = 0, 1


def make_color(a, r, g, b):
    """Argument order (a, r, g, b) as rt_color_create (reference rt_funcs_math.c:14-24)."""
    c = np.zeros((), dtype=color)
    c["a"], c["r"], c["g"], c["b"] = a, r, g, b
    return c


def make_material(col, ambient, diffuse, specular, reflect, falloff, opt_dens):
    """Each colour is an (a, r, g, b) tuple; mirrors rt_material_create."""
    m = np.zeros((), dtype=material)
    for name, v in (("color", col), ("ambient", ambient), ("diffuse", diffuse),
                    ("specular", specular), ("reflect", reflect)):
        m[name] = make_color(*v)
    m["lightFalloff"] = falloff
    m["optDens"] = opt_dens
    return m
