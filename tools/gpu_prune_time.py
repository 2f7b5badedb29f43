import time, numpy as np
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, DeviceTree, RT_HIP_TIMING, RT_HIP_HITIDS
sc = scenes.config3(); v, t = sc.flatten()
r = HipRenderer(0)
ref = None
for label, mk in (("host plain", lambda: FlatTree.build(v, t, 26, 12)), ("host pruned", lambda: FlatTree.build(v, t, 26, 12, prune_leaves=1e-5)),
                  ("gpu plain", lambda: DeviceTree.build(v, t, 26, 12)), ("gpu pruned", lambda: DeviceTree.build(v, t, 26, 12, prune_leaves=1e-5))):
    t0 = time.perf_counter(); tr = mk(); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); tr2 = mk(); tb = time.perf_counter() - t0
    if isinstance(tr2, DeviceTree): tr2.release()
    r.load_scene(sc, tr)
    best = 1e9
    for _ in range(6):
        r.render(RT_HIP_TIMING); ms, _ = r.kernel_times(); best = min(best, ms["frame"])
    r.render(RT_HIP_HITIDS)
    img, ids = r.read_frame().copy(), r.read_hit_ids().copy()
    if ref is None: ref = (img, ids)
    print(f"{label}: build {tb*1e3:.1f} ms, frame {best:.3f} ms, same image {np.array_equal(img, ref[0])}, same ids {np.array_equal(ids, ref[1])}", flush=True)
r.close()
