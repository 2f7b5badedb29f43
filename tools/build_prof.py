import time
from raytracing_amd import scenes
from raytracing_amd.pipe import DeviceTree
sc = scenes.config3(); v, t = sc.flatten()
for _ in range(4):
    t0 = time.perf_counter(); d = DeviceTree.build(v, t, 26, 12); print((time.perf_counter() - t0) * 1e3, "ms"); d.release()
