"""Band split + gather of the multi-GPU path on CPU: two gloo ranks, each
"renders" its bands (row number painted into the pixels), rank 0 assembles."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import dist as rdist


@pytest.mark.parametrize("h,band,parts", [(2160, 16, 8), (1080, 16, 8), (768, 16, 3), (50, 16, 2), (7, 16, 4), (33, 8, 2)])
def test_band_rows_partition_the_image(h, band, parts):
    rows = [rdist.band_rows(h, band, parts, p) for p in range(parts)]
    allr = np.concatenate(rows)
    assert sorted(allr.tolist()) == list(range(h))
    assert max(len(r) for r in rows) == rdist.max_local_rows(h, band, parts)
    assert 0 <= rdist.padded_rows(h, band, parts) - rdist.max_local_rows(h, band, parts) < band
    for r in rows:            # local order is increasing, bands stay contiguous
        assert (np.diff(r) > 0).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, band, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = rdist.band_rows(h, band, world, rank)
    local = torch.zeros((len(rows), w, 4), dtype=torch.uint8)
    local[:, :, 0] = torch.as_tensor(rows % 251, dtype=torch.uint8)[:, None]
    local[:, :, 1] = rank
    local[:, :, 2] = torch.arange(w, dtype=torch.int64).remainder(256).to(torch.uint8)[None, :]
    ok = True
    for call in range(2):         # the second call reuses the receive buffer; its input is already padded
        if call == 1:
            local = torch.cat([local, local.new_zeros((rdist.padded_rows(h, band, world) - len(rows), w, 4))])
            local[:len(rows), :, 3] = 7
        frame = rdist.gather_frame(local, h, w, band)
        if rank == 0:
            ok = ok and frame.shape == (h, w, 4)
            ok = ok and bool((frame[:, :, 0].to(torch.int64) == (torch.arange(h) % 251)[:, None]).all())
            owner = torch.as_tensor((np.arange(h) // band) % world)
            ok = ok and bool((frame[:, :, 1].to(torch.int64) == owner[:, None]).all())
            ok = ok and bool((frame[:, :, 2].to(torch.int64) == (torch.arange(w) % 256)[None, :]).all())
            ok = ok and bool((frame[:, :, 3] == 7 * call).all())
        else:
            assert frame is None
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w,band", [(70, 40, 16), (64, 33, 16), (9, 5, 4), (3, 4, 16), (40, 8, 16)])
def test_gather_frame_two_ranks_gloo(h, w, band):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, h, w, band, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("n,parts", [(240, 8), (7, 3), (2, 4), (0, 2)])
def test_sequence_frames_partition(n, parts):
    got = [rdist.sequence_frames(n, parts, p) for p in range(parts)]
    assert sorted(f for g in got for f in g) == list(range(n))
    assert all(f % parts == p for p, g in enumerate(got) for f in g)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,band,parts", [(64, 40, 16, 2), (70, 33, 16, 3), (135, 17, 8, 8), (50, 64, 5, 4), (16, 8, 16, 1)])
def test_device_deinterleave_is_the_band_mapping(h, w, band, parts):
    """rt_hip_deinterleave (the second half of rt_hip_gather_frame) puts local band b of share r at band b * N + r."""
    import torch
    from raytracing_amd import dist as rdist
    pad = rdist.padded_rows(h, band, parts)
    rng = np.random.RandomState(h * w + parts)
    full = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    stage = rng.randint(0, 256, (parts, pad, w, 4)).astype(np.uint8)      # rows beyond a share: never shown
    for p in range(parts):
        rows = rdist.band_rows(h, band, parts, p)
        stage[p, :len(rows)] = full[rows]
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    rdist.deinterleave(torch.from_numpy(stage).cuda(), out, band, parts)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full)


@pytest.mark.gpu
def test_rccl_gather_through_the_c_abi_single_rank():
    """The whole C-ABI path with a one-rank communicator (the box has one GPU; RCCL refuses two ranks on one device):
    unique id, communicator, gather = own share through the staging buffer and the de-interleave kernel."""
    import torch
    from raytracing_amd import dist as rdist
    h, w = 72, 48
    comm = rdist.HipComm(0, 1, 0, rdist.HipComm.unique_id())
    local = torch.randint(0, 256, (rdist.padded_rows(h, 16, 1), w, 4), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        comm.gather(local, frame, h, w, 16, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(frame, local[:h])
    comm.close()
