"""Frame sharding across the GPUs of one node and re-assembly of the demodulated bit-stream.

Frames are independent (each is one reference work() buffer: own sync, own channel estimate), so rank r simply owns
frames [r*n/W, (r+1)*n/W) and there is no data-path collective.  The only exchange is the all-gather of the packed hard
bits (RCCL over xGMI on GPUs; gloo in the CPU tests), issued per sub-batch so that it overlaps the demod of the next one.
"""
from __future__ import annotations


def shard_frames(n_frames_total: int, world: int, rank: int):
    """Contiguous, equal-size shard of whole frames: (first_frame, n_frames).  n_frames_total must divide evenly
    (all-gather needs equal counts); callers pad the batch to a multiple of `world`."""
    if n_frames_total % world:
        raise ValueError("n_frames_total=%d is not a multiple of world=%d" % (n_frames_total, world))
    per = n_frames_total // world
    return rank * per, per


def sub_batches(n_frames: int, n_chunks: int):
    n_chunks = max(1, min(n_chunks, n_frames))
    return [(i * n_frames // n_chunks, (i + 1) * n_frames // n_chunks) for i in range(n_chunks)]


def alloc_gather_buffers(torch, world: int, bounds, row_bytes: int, device):
    """One contiguous [world, n_rows, row_bytes] receive buffer per sub-batch: all_gather_into_tensor then needs no
    staging copy (a list of strided views would make c10d gather into a temporary and copy out)."""
    # 2-D [world*n_rows, row_bytes]: the concatenation-along-dim-0 form every c10d backend accepts
    return [torch.empty((world * (f1 - f0), row_bytes), dtype=torch.uint8, device=device) for f0, f1 in bounds]


def all_gather_bits(dist, recv, local_bits, f0: int, f1: int, async_op: bool = True):
    """Gather rows [f0,f1) of every rank's `local_bits` [n_frames, row_bytes] into recv ([world*(f1-f0), row_bytes], rank-major).
    Returns the work handle (or None)."""
    return dist.all_gather_into_tensor(recv, local_bits[f0:f1].contiguous(), async_op=async_op)


def reassemble(torch, recv_list, world: int):
    """[world, n_frames, row_bytes] copy of the whole bit-stream from the per-sub-batch receive buffers."""
    return torch.cat([r.view(world, r.shape[0] // world, r.shape[1]) for r in recv_list], dim=1)
