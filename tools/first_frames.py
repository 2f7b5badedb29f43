"""Are the very first frames of a process right?  Four contexts on one scene, their first frames enqueued side by
side before anything else has run on the GPU in this process (with kernels that use scratch memory the runtime
sizes its scratch arena at these first launches, on a dozen streams at once), then two more rounds; every frame
is compared with the frame a fifth context renders alone afterwards, passes in order.
python tools/first_frames.py [config3|config5] [parts]"""
import os, sys
os.environ["GPU_MAX_HW_QUEUES"] = "16"
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_LAUNCH_WAVES, RT_PARAM_OVERLAP

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = getattr(scenes, name)()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26, 12, prune_leaves=1e-5)
K = 4
streams = [torch.cuda.Stream() for _ in range(K)]
rs = []
for k in range(K):
    r = HipRenderer(0, stream=streams[k].cuda_stream)
    if k:
        r.share_scene(rs[0], sc)
    else:
        r.load_scene(sc, tree)
    r.set_param(RT_PARAM_LAUNCH_WAVES, 8)
    r.set_tile(16, parts, 0)
    rs.append(r)
rows = rs[0].local_rows
outs = [[torch.zeros((rows, sc.w, 4), dtype=torch.uint8, device="cuda") for _ in range(K)] for _ in range(3)]
torch.cuda.synchronize()
for rnd in range(3):
    for k in range(K):
        rs[k].render(0, d_out=outs[rnd][k].data_ptr())
torch.cuda.synchronize()
ck = HipRenderer(0)
ck.share_scene(rs[0], sc)
ck.set_param(RT_PARAM_OVERLAP, 0)
ck.set_tile(16, parts, 0)
want = torch.zeros_like(outs[0][0])
ck.render(0, d_out=want.data_ptr())
ck.sync()
bad = [(rnd, k, int((outs[rnd][k] != want).any(dim=-1).sum())) for rnd in range(3) for k in range(K) if not torch.equal(outs[rnd][k], want)]
print(f"{name} 1/{parts}: {3 * K} frames, lib {os.environ.get('RT_AMD_LIB', 'product')}: differing (round, context, pixels): {bad}", flush=True)
ck.close()
for r in reversed(rs):
    r.close()
sys.exit(1 if bad else 0)
