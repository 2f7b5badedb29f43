"""Scenes whose centre pixel can be worked out by hand from the reference's
shading code, and the worked-out values.

Every scene has one receiver triangle A in the plane z = 20 that the centre
ray (origin 0, direction +z) meets at P = (0, 0, 20) with barycentrics
u = v = 1/4 (all dyadic, so the intersection arithmetic is exact), vertex
normals (0, 0, -1), and further small triangles placed so that the centre
pixel's secondary and shadow rays meet them head-on.  The expected colours are
computed below in float32, operation by operation as the reference does
(CL/rt_funcs_primitives.cl:450-564 lighting, :643-677 child rays, :688-725
fold, CL/rt_raytrace.cl store), NOT by calling the restatement under test.
"""
import numpy as np

from raytracing_amd import scenes
from raytracing_amd import types as T

f32 = np.float32
W = H = 33
CENTRE = 16

# (a, r, g, b) as rt_color_create takes them
COL_A = (1.0, 0.55, 0.3, 0.95)
AMB, DIF = 0.1, 0.8


def _tri(sc, pts, normal, group):
    pos = np.array(pts, np.float32)
    nrm = np.tile(np.array(normal, np.float32), (3, 1))
    sc.meshes.append(scenes._mesh_from(pos, nrm, np.array([[0, 1, 2]]), group))


def _receiver(sc, col=COL_A, spec=(1.0, 0.0, 0.0, 0.0), reflect=(0.0, 0.0, 0.0, 0.0), falloff=0.0, opt_dens=1.0,
              normal=(0, 0, -1)):
    sc.materials[0] = T.make_material(col, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), spec, reflect, falloff, opt_dens)
    _tri(sc, [(-4, -4, 20), (12, -4, 20), (-4, 12, 20)], normal, 0)


def _scene(name, light):
    sc = scenes.Scene(name, W, H, scenes.make_camera(W, H))
    sc.lights.append(scenes.make_light(light, 10.0))
    return sc


def _plain(col_a):
    """material that only has an opacity (a blocker) or a colour"""
    z = (0.0, 0.0, 0.0, 0.0)
    return T.make_material(col_a, z, z, (1.0, 0.0, 0.0, 0.0), z, 0.0, 1.0)


def lit(col_bgr, n_dot_l, rad_over_d_num, d, shadowed, amb=AMB, dif=DIF, spec_bgr=(0, 0, 0), spec_fact=0.0):
    """One light's contribution per channel, :506-562: (ambient + diffuse +
    specular) * (1 - shadowed), clamped; light colour is white."""
    out = []
    for c, s in zip(col_bgr, spec_bgr):
        ambient = f32(1.0) * f32(amb)
        diffuse = (f32(1.0) * f32(dif)) * f32(c)
        diffuse = diffuse * (f32(n_dot_l) * f32(rad_over_d_num) / f32(d))
        specular = (f32(s) * f32(1.0)) * (f32(spec_fact) * f32(rad_over_d_num) / f32(d))
        v = ((ambient + diffuse) + specular) * (f32(1.0) - f32(shadowed))
        v = f32(0.0) + v
        out.append(clamp(v))
    return out


def clamp(v):
    return v if f32(0.0) < v < f32(1.0) else (f32(1.0) if v >= f32(1.0) else f32(0.0))


def to_bytes(bgr):
    """CL/rt_raytrace.cl store: 255 * channel, truncated; alpha byte 0.  Also
    checks that the case does not sit on a rounding edge (it would not be a
    hand-checkable case then)."""
    out = []
    for v in bgr:
        x = float(f32(255.0) * v)
        assert x == 0.0 or x == 255.0 or 0.02 < x - int(x) < 0.98, "known-answer case too close to an 8-bit step"
        out.append(int(x))
    return out + [0]


BGR_A = (COL_A[3], COL_A[2], COL_A[1])


def shadow_case(kind):
    """Light at (0,-10,10): the shadow ray of P leaves along (0,-1,-1)/sqrt 2,
    d = sqrt 200, N.L = 10/d.  Blockers sit on that ray."""
    sc = _scene("ka_shadow_" + kind, (0.0, -10.0, 10.0))
    _receiver(sc)

    def blocker(y, z, group):
        _tri(sc, [(-1, y - 1, z), (1, y - 1, z), (0, y + 1, z)], (0, 0, -1), group)

    shadowed = 0.0
    if kind == "opaque":
        sc.materials[1] = _plain((1.0, 0, 0, 0)); blocker(-5, 15, 1); shadowed = 1.0
    elif kind == "half":
        sc.materials[1] = _plain((0.5, 0, 0, 0)); blocker(-5, 15, 1); shadowed = 0.5
    elif kind == "two_quarters":
        sc.materials[1] = _plain((0.25, 0, 0, 0)); blocker(-3, 17, 1); blocker(-7, 13, 1); shadowed = 0.5
    elif kind == "three_halves_clamp":
        sc.materials[1] = _plain((0.5, 0, 0, 0)); blocker(-2, 18, 1); blocker(-5, 15, 1); blocker(-8, 12, 1); shadowed = 1.0
    elif kind == "beyond_the_light":
        sc.materials[1] = _plain((1.0, 0, 0, 0)); blocker(-15, 5, 1); shadowed = 0.0
    elif kind != "none":
        raise ValueError(kind)
    d = np.sqrt(f32(200.0))
    n_dot_l = f32(-1.0) * (f32(-10.0) / d)
    return sc, to_bytes(lit(BGR_A, n_dot_l, 10.0, d, shadowed))


def specular_case():
    """Light at the eye: N.L = 1, the mirrored light direction points at the
    eye, pow(1, e) = 1."""
    sc = _scene("ka_specular", (0.0, 0.0, 0.0))
    _receiver(sc, spec=(7.0, 0.3, 0.2, 0.1))
    return sc, to_bytes(lit(BGR_A, 1.0, 10.0, 20.0, 0.0, spec_bgr=(0.1, 0.2, 0.3), spec_fact=1.0))


COL_B = (1.0, 0.9, 0.7, 0.3)
BGR_B = (COL_B[3], COL_B[2], COL_B[1])


def _fold(parent, child, weight):
    return [clamp(p + c * f32(w)) for p, c, w in zip(parent, child, weight)]


def reflection_case():
    """A mirrors half of what lies behind the camera: triangle B at z = -10
    faces +z; light at the eye (distance 10 from B, 20 from A)."""
    sc = _scene("ka_reflection", (0.0, 0.0, 0.0))
    _receiver(sc, reflect=(0.0, 0.5, 0.25, 0.125))
    sc.materials[1] = T.make_material(COL_B, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0), (0, 0, 0, 0), 0.0, 1.0)
    _tri(sc, [(-4, -4, -10), (12, -4, -10), (-4, 12, -10)], (0, 0, 1), 1)
    a = lit(BGR_A, 1.0, 10.0, 20.0, 0.0)
    b = lit(BGR_B, 1.0, 10.0, 10.0, 0.0)
    return sc, to_bytes(_fold(a, b, (0.125, 0.25, 0.5)))


def refraction_case(falloff=0.0, with_reflection=False):
    """A is half transparent with optical density 1: at normal incidence the
    ray goes straight on (:660-676) and meets C at z = 30.  C's shadow ray to
    the light at the eye passes through A (opacity 1/2).  With falloff 0 the
    fold weight is exp(0) = 1 (:704-719)."""
    sc = _scene("ka_refraction" + ("_reflection" if with_reflection else "") + (f"_{falloff}" if falloff else ""), (0.0, 0.0, 0.0))
    col_a = (0.5,) + COL_A[1:]
    _receiver(sc, col=col_a, reflect=(0.0, 0.5, 0.25, 0.125) if with_reflection else (0, 0, 0, 0), falloff=falloff)
    sc.materials[2] = T.make_material(COL_B, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0), (0, 0, 0, 0), 0.0, 1.0)
    _tri(sc, [(-4, -4, 30), (12, -4, 30), (-4, 12, 30)], (0, 0, -1), 2)
    a = lit(BGR_A, 1.0, 10.0, 20.0, 0.0)
    c = lit(BGR_B, 1.0, 10.0, 30.0, 0.5)
    w = [f32(np.exp(np.float64(f32(ch) * f32(falloff) * f32(-10.0)))) for ch in BGR_A]
    out = _fold(a, c, w)                       # the refraction child (slot 2, or 1 without a mirror) folds first
    if with_reflection:
        sc.materials[1] = sc.materials[2]
        _tri(sc, [(-4, -4, -10), (12, -4, -10), (-4, 12, -10)], (0, 0, 1), 1)
        b = lit(BGR_B, 1.0, 10.0, 10.0, 0.0)
        out = _fold(out, b, (0.125, 0.25, 0.5))
    return sc, to_bytes(out)


def _target(sc, q, facing, group, size=2.0):
    """small triangle centred on q in the plane through q perpendicular to
    `facing`, vertex normals = facing"""
    f = np.asarray(facing, np.float64)
    e1 = np.cross(f, (0.0, 1.0, 0.0)); e1 /= np.linalg.norm(e1)
    e2 = np.cross(f, e1)
    q = np.asarray(q, np.float64)
    _tri(sc, [q - size * e1 - size * e2, q + size * e1 - size * e2, q + size * e2], f, group)


def _lit64(col_bgr, n, p, light, rad, shadowed):
    to_l = np.asarray(light, np.float64) - p
    d = np.linalg.norm(to_l)
    ang = max(0.0, float(np.dot(n, to_l / d)))
    return [min(1.0, (AMB + DIF * c * ang * rad / d) * (1.0 - shadowed)) for c in col_bgr]


def oblique_case(kind):
    """Vertex normals of A tilted to (0.6, 0, -0.8) (the geometry stays in the
    plane z = 20): the mirror direction d - 2(d.n)n (CL/rt_funcs_math.cl reflect)
    and the refracted direction of :660-670 for optical density 1.5 are worked
    out in double precision and a small target is put 25 / 15 units along
    them.  Expected values carry one 8-bit step of slack (float32 vs double)."""
    n = np.array((0.6, 0.0, -0.8))
    d = np.array((0.0, 0.0, 1.0))
    p = np.array((0.0, 0.0, 20.0))
    eye = (0.0, 0.0, 0.0)
    sc = _scene("ka_oblique_" + kind, eye)
    weights = (0.125, 0.25, 0.5)
    if kind == "mirror":
        _receiver(sc, reflect=(0.0, 0.5, 0.25, 0.125), normal=n)
        r = d - 2.0 * np.dot(d, n) * n                      # (0.96, 0, -0.28)
        q = p + 25.0 * r
        shadow_t = 0.0
        alpha_a = 1.0
    else:
        alpha_a = 0.5
        _receiver(sc, col=(alpha_a,) + COL_A[1:], opt_dens=1.5, normal=n)
        n_rel = 1.0 / 1.5
        cos_i = float(np.dot(n, d))
        sin_t = n_rel * n_rel * (1.0 - cos_i * cos_i)
        r = d * n_rel - n * (n_rel * cos_i + np.sqrt(1.0 - sin_t))
        r /= np.linalg.norm(r)
        # Snell: the sines of the angles to the normal are in the ratio of the densities
        assert abs(np.linalg.norm(np.cross(r, -n)) * 1.5 - np.linalg.norm(np.cross(d, -n))) < 1e-12
        q = p + 15.0 * r
        shadow_t = alpha_a                                  # the target's shadow ray to the eye crosses A
        weights = (1.0, 1.0, 1.0)                           # falloff 0
    sc.materials[1] = T.make_material(COL_B, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0), (0, 0, 0, 0), 0.0, 1.0)
    _target(sc, q, -r, 1)
    # make_material rounds the tilted normal to float32; A's own shading uses it through N.L only
    a = _lit64(BGR_A, n, p, eye, 10.0, 0.0)
    t = _lit64(BGR_B, -r, q, eye, 10.0, shadow_t)
    out = [min(1.0, x + y * w) for x, y, w in zip(a, t, weights)]
    return sc, [int(255.0 * v) for v in out] + [0], q


def sphere_case(kind):
    """The receiver is a sphere of radius 10 about (0,0,30): the centre ray
    meets it at P = (0,0,20), t = 20 exactly, normal (P - c)/r = (0,0,-1)
    (:42-46), so the numbers are those of the triangle receiver.  A unit sphere
    of opacity 0.3 sits on the shadow ray.  The shadow query adds the spheres'
    opacities (:418-448), clamps, and then the triangles' - but the triangle
    query returns before resetting its sum when the shadow ray misses the scene
    box (:289-293), so the sphere sum is counted a second time."""
    sc = _scene("ka_sphere_" + kind, (0.0, -10.0, 10.0))
    sc.materials[0] = T.make_material(COL_A, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0), (0, 0, 0, 0), 0.0, 1.0)
    sc.materials[1] = _plain((0.3, 0, 0, 0))
    sc.materials[2] = _plain((1.0, 0, 0, 0))
    sc.spheres.append(scenes.make_sphere((0.0, 0.0, 30.0), 10.0, 0))
    if kind == "lit":
        shadowed = 0.0
    else:
        sc.spheres.append(scenes.make_sphere((0.0, -5.0, 15.0), 1.0, 1))
        shadowed = 0.3
    if kind == "blocker_counted_twice":
        # the only triangle is far to the side: every ray of the centre pixel misses the scene box
        _tri(sc, [(50, 0, 20), (52, 0, 20), (51, 2, 20)], (0, 0, -1), 2)
        shadowed = float(f32(0.3) + f32(0.3))
    else:
        # two walls left and right put the whole shadow ray inside the scene box without touching it
        _tri(sc, [(-6, -20, 0), (-6, 20, 0), (-6, 0, 40)], (1, 0, 0), 2)
        _tri(sc, [(6, -20, 0), (6, 20, 0), (6, 0, 40)], (-1, 0, 0), 2)
    d = np.sqrt(f32(200.0))
    n_dot_l = f32(-1.0) * (f32(-10.0) / d)
    return sc, to_bytes(lit(BGR_A, n_dot_l, 10.0, d, shadowed))


def camera_case(w=48, h=32):
    """Ray generation, CL/rt_raytrace.cl:27-28,60-79: pixel (x, y) looks along
    (ndc_x * P11, ndc_y * P22, 1) with ndc = 2 * pixel / (size - 1) - 1, taken
    through the camera's world matrix as a ROW vector (CL/rt_funcs_math.cl:126-147:
    rows 1-3 are the images of the view axes, row 4 the position).  The world
    matrix here turns view +z into world +x and view +x into world -z and moves
    the eye to (5,-3,2).  One small target per chosen pixel sits 20 units down
    its ray; the pixel pitch there is 0.5, the targets are 0.2 wide.
    -> (scene, {(x, y): triangle id})"""
    world = np.array([[0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0], [5, -3, 2, 1]], np.float32)
    sc = scenes.Scene("ka_camera", w, h, scenes.make_camera(w, h, world=world))
    sc.materials[0] = _plain((1.0, 0, 0, 0))
    p11 = 2.0 * np.tan(0.125 * np.pi / 2.0) * (w / h)
    p22 = 2.0 * np.tan(0.125 * np.pi / 2.0)
    want = {}
    for k, (x, y) in enumerate([(0, 0), (w - 1, h - 1), (0, h - 1), (w - 1, 0), (5, 20), (w // 2, h // 2), (w - 2, 1)]):
        vx = (2.0 * x / (w - 1) - 1.0) * p11
        vy = (2.0 * y / (h - 1) - 1.0) * p22
        direction = np.array((1.0, vy, -vx))            # view (vx, vy, 1) as a row vector times the world rows
        q = np.array((5.0, -3.0, 2.0)) + 20.0 * direction
        _target(sc, q, -direction / np.linalg.norm(direction), 0, size=0.2)
        want[(x, y)] = k
    return sc, want


def facing_mirrors_case(stack_size):
    """Two mirrors face each other along the centre ray (A at z = 20, B at
    z = -10, the eye and the light between them): the ray bounces A, B, A, ...
    until the ray budget is used up - a hit only spawns a child while
    stackPos + 1 < STACK_SIZE (:643) - and the colours fold back child into
    parent with the PARENT's mirror weights, clamped at every step (:688-725)."""
    sc = _scene(f"ka_mirrors_{stack_size}", (0.0, 0.0, 0.0))
    sc.stack_size = stack_size
    wa, wb = (0.5, 0.5, 0.75), (0.5, 0.5, 0.5)                 # b, g, r
    _receiver(sc, reflect=(0.0, wa[2], wa[1], wa[0]))
    sc.materials[1] = T.make_material(COL_B, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0),
                                      (0.0, wb[2], wb[1], wb[0]), 0.0, 1.0)
    _tri(sc, [(-4, -4, -10), (12, -4, -10), (-4, 12, -10)], (0, 0, 1), 1)
    a = lit(BGR_A, 1.0, 10.0, 20.0, 0.0)
    b = lit(BGR_B, 1.0, 10.0, 10.0, 0.0)
    cols = [list(a if i % 2 == 0 else b) for i in range(stack_size)]
    for i in range(stack_size - 1, 0, -1):
        cols[i - 1] = _fold(cols[i - 1], cols[i], wa if (i - 1) % 2 == 0 else wb)
    return sc, to_bytes(cols[0])


def misc_case(kind):
    sc = _scene("ka_" + kind, (0.0, -10.0, 10.0))
    d = np.sqrt(f32(200.0))
    side = f32(-1.0) * (f32(-10.0) / d)                 # N.L for the light at (0,-10,10)
    mat_b = T.make_material(COL_B, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0), (0, 0, 0, 0), 0.0, 1.0)
    if kind == "two_lights":
        # every light adds its clamped term to the running colour, which is clamped again (:561-562)
        _receiver(sc)
        sc.lights.append(scenes.make_light((0.0, 0.0, 0.0), 10.0))
        first = lit(BGR_A, side, 10.0, d, 0.0)
        second = lit(BGR_A, 1.0, 10.0, 20.0, 0.0)
        return sc, to_bytes([clamp(x + y) for x, y in zip(first, second)])
    if kind == "light_behind":
        # max(0, N.L): only the ambient term is left (:516)
        sc.lights[0] = scenes.make_light((0.0, 0.0, 30.0), 10.0)
        _receiver(sc)
        return sc, to_bytes(lit(BGR_A, 0.0, 10.0, 10.0, 0.0))
    if kind == "total_internal_reflection":
        # optical density 1/2 and a normal tilted by 36.9 degrees: sinT = 4 * 0.36 >= 1, no refracted ray (:664-666);
        # the bright target straight behind A stays unseen
        sc.lights[0] = scenes.make_light((0.0, 0.0, 0.0), 9.0)
        _receiver(sc, col=(0.5,) + COL_A[1:], opt_dens=0.5, normal=(0.6, 0.0, -0.8))
        sc.materials[1] = mat_b
        _tri(sc, [(-4, -4, 30), (12, -4, 30), (-4, 12, 30)], (0, 0, -1), 1)
        return sc, to_bytes(lit(BGR_A, 0.8, 9.0, 20.0, 0.0))
    if kind == "inside_a_mirror_sphere":
        # seen from inside, a sphere reports -1: the normal is flipped (:45-46) and no mirror ray is spawned (:644)
        sc.materials[0] = T.make_material(COL_A, (0.0, AMB, AMB, AMB), (0.0, DIF, DIF, DIF), (1.0, 0, 0, 0),
                                          (0.0, 0.5, 0.5, 0.5), 0.0, 1.0)
        sc.materials[1] = _plain((1.0, 0, 0, 0))
        sc.spheres.append(scenes.make_sphere((0.0, 0.0, 0.0), 20.0, 0))
        _tri(sc, [(50, 0, 20), (52, 0, 20), (51, 2, 20)], (0, 0, -1), 1)
        return sc, to_bytes(lit(BGR_A, side, 10.0, d, 0.0))
    if kind == "sphere_wins_a_tie":
        # triangle and sphere at the same distance 20: the triangle is only taken when strictly nearer (:613)
        _receiver(sc)
        sc.materials[1] = mat_b
        sc.spheres.append(scenes.make_sphere((0.0, 0.0, 30.0), 10.0, 1))
        return sc, to_bytes(lit(BGR_B, side, 10.0, d, 0.0))
    raise ValueError(kind)


MISC = ("two_lights", "light_behind", "total_internal_reflection", "inside_a_mirror_sphere", "sphere_wins_a_tie")


def all_cases():
    cases = {}
    for k in MISC:
        cases["sphere_" + k if "sphere" in k else k] = misc_case(k)
    for n in (1, 2, 3, 5, 6):
        cases[f"facing_mirrors_budget_{n}"] = facing_mirrors_case(n)
    for k in ("lit", "blocker_in_the_box", "blocker_counted_twice"):
        cases["sphere_" + k] = sphere_case(k)
    for k in ("none", "opaque", "half", "two_quarters", "three_halves_clamp", "beyond_the_light"):
        cases["shadow_" + k] = shadow_case(k)
    cases["specular"] = specular_case()
    cases["reflection"] = reflection_case()
    cases["refraction"] = refraction_case()
    cases["refraction_and_reflection"] = refraction_case(with_reflection=True)
    return cases


def expected(case, max_prims):
    """-> (scene, centre pixel).  One combination meets a builder rule of the
    reference: with one triangle per leaf the two-triangle mirror scenes are split
    at z = -10, the plane the mirrored triangle lies in, and a triangle with
    min == max == sep goes to neither child (rt_funcs_render_pipe.c:1024-1027):
    it vanishes from the tree and the mirror shows nothing."""
    sc, want = all_cases()[case]
    if (case == "reflection" or case.startswith("facing_mirrors")) and max_prims == 1:
        want = to_bytes(lit(BGR_A, 1.0, 10.0, 20.0, 0.0))
    return sc, want
