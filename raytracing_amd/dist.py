"""Multi-GPU frame split: one process per GPU, the scene replicated, the image
cut into bands of `band_h` rows dealt round-robin to the ranks, and the
per-rank frame buffers gathered with one collective (RCCL over xGMI when the
tensors live on GPUs, gloo on CPU in the tests).

The reference renders the whole frame in one NDRange on one device
(rt_funcs_render_pipe.c:534-565); this module has no counterpart there.
Pixels are independent, so the only exchange is the gather of finished
8-bit rows: h*w*4 bytes per frame in total.
"""
import numpy as np
import torch
import torch.distributed as dist

BAND_H = 16


def band_rows(h, band_h, n_parts, part):
    """Global row indices rendered by `part`, in its local storage order
    (the mapping of rt_hip_set_tile)."""
    rows = []
    n_bands = (h + band_h - 1) // band_h
    for b in range(part, n_bands, n_parts):
        rows.extend(range(b * band_h, min((b + 1) * band_h, h)))
    return np.asarray(rows, dtype=np.int64)


def local_rows(h, band_h, n_parts, part):
    return len(band_rows(h, band_h, n_parts, part))


def max_local_rows(h, band_h, n_parts):
    return max(local_rows(h, band_h, n_parts, p) for p in range(n_parts))


def padded_rows(h, band_h, n_parts):
    """Rows of the per-rank buffer the gather exchanges: the largest share,
    rounded up to whole bands (every rank sends the same amount)."""
    n_bands = (h + band_h - 1) // band_h
    return ((n_bands + n_parts - 1) // n_parts) * band_h


def assemble(parts, h, w, band_h):
    """parts[r]: [>= local_rows(r), w, 4] tensor of rank r -> [h, w, 4]."""
    n = len(parts)
    out = torch.empty((h, w, 4), dtype=parts[0].dtype, device=parts[0].device)
    for r in range(n):
        rows = torch.as_tensor(band_rows(h, band_h, n, r), device=out.device)
        out[rows] = parts[r][:len(rows)]
    return out


_recv = {}      # receive buffer of the destination rank, kept between frames


def gather_frame(local, h, w, band_h=BAND_H, dst=0, group=None):
    """All ranks pass their [rows, w, 4] uint8 tensor (any row count from
    their own share up; padded_rows() avoids a copy).  Rank `dst` returns the
    assembled [h, w, 4] frame, the others None.

    Band b lives on rank b mod N as local band b div N, so with every share
    padded to B whole bands the received [N, B, band_h, w, 4] block becomes
    the frame by swapping its first two axes: one strided copy, no index
    arrays, nothing allocated per frame (the receive buffer is reused, so the
    returned frame is valid until the next call with the same geometry)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local[:h]
    pad = padded_rows(h, band_h, world)
    if local.shape[0] < pad:
        local = torch.cat([local, local.new_zeros((pad - local.shape[0], w, 4))])
    elif local.shape[0] > pad:
        local = local[:pad]
    local = local.contiguous()
    # gloo (CPU rehearsals of the multi-rank path) cannot gather device tensors: stage through the host
    device = local.device
    if device.type != "cpu" and dist.get_backend(group) == "gloo":
        local = local.cpu()
    # all-to-one: every source uses its own direct link to dst
    bufs = None
    if rank == dst:
        key = (world, pad, w, local.dtype, local.device)
        if key not in _recv:
            _recv.clear()
            _recv[key] = torch.empty((world, pad, w, 4), dtype=local.dtype, device=local.device)
        big = _recv[key]
        bufs = [big[i] for i in range(world)]
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    nb = pad // band_h
    if big.device != device:
        big = big.to(device)
    return big.view(world, nb, band_h, w, 4).permute(1, 0, 2, 3, 4).reshape(nb * world * band_h, w, 4)[:h]


def sequence_frames(n_frames, n_parts, part, start=0):
    """Frames of a sequence rendered by `part`: frame f goes to rank f mod
    n_parts.  Frames are independent given the scene (only the camera
    changes), so a sequence needs no collective at all."""
    first = start + ((part - start) % n_parts)
    return list(range(first, n_frames, n_parts))


# ---------------------------------------------------------------------------
# the same exchange through the library's own C-ABI (RCCL send / receive +
# de-interleave kernel, raytracing_amd/csrc/gather.hip): what a C caller of the
# render-pipe API uses, and what bench.py times on GPUs
# ---------------------------------------------------------------------------

class HipComm:
    """One rt_hip_comm: this rank's end of an RCCL communicator over `n_ranks` GPUs."""

    ID_BYTES = 128

    def __init__(self, device, n_ranks, rank, unique_id):
        import ctypes as C
        from ._lib import RtError, lib
        self._L = lib()
        self._L.rt_hip_comm_create.restype = C.c_void_p
        self._L.rt_hip_comm_last_error.restype = C.c_char_p
        self.n_ranks, self.rank = n_ranks, rank
        buf = (C.c_char * self.ID_BYTES).from_buffer_copy(bytes(unique_id))
        h = self._L.rt_hip_comm_create(C.c_int(device), C.c_int(n_ranks), C.c_int(rank), buf)
        if not h:
            raise RtError("rt_hip_comm_create: " + self._L.rt_hip_comm_last_error().decode())
        self._h = C.c_void_p(h)

    @staticmethod
    def unique_id():
        """bytes of a fresh communicator id (call on one rank, hand to the others)"""
        import ctypes as C
        from ._lib import RtError, lib
        L = lib()
        L.rt_hip_comm_last_error.restype = C.c_char_p
        buf = (C.c_char * HipComm.ID_BYTES)()
        if L.rt_hip_comm_unique_id(buf):
            raise RtError("rt_hip_comm_unique_id: " + L.rt_hip_comm_last_error().decode())
        return bytes(buf)

    def gather(self, local, frame, h, w, band_h=BAND_H, dst=0, stream=None):
        """local: device tensor of at least padded_rows() rows; frame: [h, w, 4] device tensor on rank dst (else None)."""
        import ctypes as C
        from ._lib import RtError
        assert local.shape[0] >= padded_rows(h, band_h, self.n_ranks) and local.is_contiguous()
        rc = self._L.rt_hip_gather_frame(self._h, C.c_void_p(local.data_ptr()),
                                         C.c_void_p(frame.data_ptr()) if frame is not None else None,
                                         C.c_int(w), C.c_int(h), C.c_int(band_h), C.c_int(dst), C.c_void_p(stream))
        if rc:
            raise RtError("rt_hip_gather_frame: " + self._L.rt_hip_comm_last_error().decode())

    def close(self):
        if self._h:
            self._L.rt_hip_comm_destroy(self._h)
            self._h = None


def deinterleave(stage, frame, band_h, n_parts, stream=None):
    """rt_hip_deinterleave: stage [n_parts, padded_rows, w, 4] -> frame [h, w, 4], both device tensors."""
    import ctypes as C
    from ._lib import RtError, lib
    h, w = frame.shape[0], frame.shape[1]
    L = lib()
    L.rt_hip_comm_last_error.restype = C.c_char_p
    if L.rt_hip_deinterleave(C.c_void_p(stage.data_ptr()), C.c_void_p(frame.data_ptr()), C.c_int(w), C.c_int(h),
                             C.c_int(band_h), C.c_int(n_parts), C.c_void_p(stream)):
        raise RtError("rt_hip_deinterleave: " + L.rt_hip_comm_last_error().decode())
