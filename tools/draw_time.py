"""What a call of rt_render_pipe_draw costs for a scene that does not change, with and without RT_PARAM_TREE_CACHE
(the time to hand the geometry to the pipe is given beside it: the cache's change detector runs there).  The C entry point
is timed by itself: RenderPipe.draw() would add its own copy of the frame (1.1 ms for a 4K frame) to it.
python tools/draw_time.py [config3|config2|config5]"""
import ctypes as C, sys, time
import numpy as np
from raytracing_amd import scenes
from raytracing_amd.pipe import RenderPipe, RT_PARAM_MAX_DEPTH, RT_PARAM_MAX_PRIMS_IN_NODE, RT_PARAM_TREE_CACHE
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
sc = getattr(scenes, name)()
for cache in (0, 1):
    rp = RenderPipe(sc.w, sc.h)
    rp.set_param(RT_PARAM_MAX_DEPTH, 26 if name != "config2" else 22)
    rp.set_param(RT_PARAM_MAX_PRIMS_IN_NODE, 12)
    rp.set_param(RT_PARAM_TREE_CACHE, cache)
    rp.set_camera(sc.camera)
    ta, td = [], []
    for f in range(12):
        t0 = time.perf_counter(); rp.add_scene(sc); t1 = time.perf_counter(); rp._L.rt_render_pipe_draw(C.byref(rp._rp)); t2 = time.perf_counter()
        ta.append(t1 - t0); td.append(t2 - t1)
    print(f"{name}: tree cache {cache}: add {sorted(ta[2:])[5] * 1e3:.2f} ms, draw {sorted(td[2:])[5] * 1e3:.2f} ms (medians of 10)", flush=True)
    rp.cleanup()
