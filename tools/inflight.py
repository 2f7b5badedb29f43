"""Experiment: frames in flight.  K contexts on one GPU render consecutive frames of the same share
(part 0 of E of the benchmark frame) without waiting for each other; prints ms per frame."""
import os, sys, time
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_OVERLAP, RT_HIP_TIMING
FLAGS = RT_HIP_TIMING if os.environ.get('TIMING') else 0
SHARE = bool(os.environ.get('SHARE'))
DEFAULT0 = bool(os.environ.get('DEFAULT0'))

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ov = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc = scenes.config3()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26, 12, prune_leaves=float(os.environ.get("RT_PRUNE", "1e-5")))
if os.environ.get('BUILD_FIRST'):
    from raytracing_amd.pipe import DeviceTree
    for _ in range(int(os.environ['BUILD_FIRST'])):
        DeviceTree.build(v, t, 26, 12).release()
LAZY = bool(os.environ.get('LAZY'))
OWN = bool(os.environ.get('OWN'))
streams = [] if LAZY else [torch.cuda.current_stream() if (DEFAULT0 and k == 0) else torch.cuda.Stream() for k in range(K)]
rs = []
for k in range(K):
    if LAZY:
        streams.append(torch.cuda.Stream())
    r = HipRenderer(0, stream=None if OWN else streams[k].cuda_stream)
    if SHARE and k > 0:
        r.share_scene(rs[0], sc)
    else:
        r.load_scene(sc, tree)
    r.set_param(RT_PARAM_OVERLAP, ov)
    r.set_tile(16, E, 0)
    rs.append(r)
outs = [torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device='cuda') for _ in range(K)]
torch.cuda.synchronize()
DOUT = bool(os.environ.get('DOUT'))
BUILD = bool(os.environ.get('BUILD'))
if BUILD:
    from raytracing_amd.pipe import DeviceTree
    DeviceTree.build(v, t, 26, 12).release()
def go(it):
    rs[it % K].render(FLAGS, d_out=outs[it % K].data_ptr() if DOUT else None)
for it in range(3 * K):
    go(it)
torch.cuda.synchronize()
best = 1e9
n = 24
for rep in range(5):
    t0 = time.perf_counter()
    for it in range(n):
        go(it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'   rep {rep}: {dt*1e3:.3f} ms', flush=True)
    best = min(best, dt)
print(f"in flight {K}, share 1/{E}, overlap {ov}, timing {FLAGS}, shared scene {SHARE}, default stream first {DEFAULT0}, d_out {DOUT}, gpu build before {BUILD}, streams made per context {LAZY}, contexts' own streams {OWN}: {best*1e3:.3f} ms per frame", flush=True)
for r in rs: r.close()
