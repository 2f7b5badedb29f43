"""Timing experiment: what the chain of dependent closest-hit passes costs.

Needs the development build with RT_EXP_REPLAY (tools/variant.sh replay raytracing_amd/csrc/rt_hip.hip
-DRT_EXP_REPLAY; RT_AMD_LIB=raytracing_amd/librt_amd_replay.so).  For rank 0's share of an N-way split of the
benchmark frame, K contexts in flight:
  normal   the frame as the product renders it (slot k's closest-hit pass waits for slot k - 1's)
  replay   every slot's passes again over the ray lists the normal frame left behind, all slots side by side:
           the same work with no order between slots - a lower bound for every schedule of these passes,
           a persistent pass fed by a queue included
python tools/deep_bound.py [parts ...]"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
import torch
from raytracing_amd import scenes
from raytracing_amd.pipe import HipRenderer, FlatTree, RT_PARAM_LAUNCH_WAVES

K = int(os.environ.get("K", "3"))
parts_list = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
sc = scenes.config3()
v, t = sc.flatten()
tree = FlatTree.build(v, t, 26, 12, prune_leaves=1e-5)
streams = [torch.cuda.Stream() for _ in range(K)]
rs = []
for k in range(K):
    r = HipRenderer(0, stream=streams[k].cuda_stream)
    if k:
        r.share_scene(rs[0], sc)
    else:
        r.load_scene(sc, tree)
    rs.append(r)
want = None
for parts in parts_list:
    outs = [torch.zeros((sc.h, sc.w, 4), dtype=torch.uint8, device="cuda") for _ in range(K)]
    for waves in ([8, 16] if parts <= 2 else [6, 8, 16]):
        res = {}
        for mode in ("normal", "replay"):
            for r in rs:
                r.set_param(100, 0)
                r.set_param(RT_PARAM_LAUNCH_WAVES, waves)
                r.set_tile(16, parts, 0)
            for i in range(2 * K):                      # (every context's lists are those of this share)
                rs[i % K].render(0, d_out=outs[i % K].data_ptr())
            torch.cuda.synchronize()
            ref = outs[0].clone()
            if mode == "replay":
                for r in rs:
                    r.set_param(100, 1)
                for i in range(K):
                    rs[i % K].render(0, d_out=outs[i % K].data_ptr())
                torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for i in range(36):
                    rs[i % K].render(0, d_out=outs[i % K].data_ptr())
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 36)
            same = all(bool(torch.equal(o, ref)) for o in outs)
            res[mode] = best * 1e3
            print(f"   1/{parts} frame, {K} in flight, {waves} waves per CU and pass, {mode}: {best * 1e3:.3f} ms per frame (frames equal: {same})", flush=True)
        print(f"1/{parts} frame, {waves} waves: normal {res['normal']:.3f} ms, without order between slots {res['replay']:.3f} ms", flush=True)
for r in rs:
    r.set_param(100, 0)
for r in reversed(rs):
    r.close()
