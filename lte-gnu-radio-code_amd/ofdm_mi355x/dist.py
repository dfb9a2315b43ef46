"""Frame sharding across the GPUs of one node and re-assembly of the demodulated bit-stream.

Frames are independent (each is one reference work() buffer: own sync, own channel estimate), so rank r simply owns
frames [r*n/W, (r+1)*n/W) and there is no data-path collective.  The only exchange is the all-gather of the packed hard
bits (RCCL over xGMI on GPUs; gloo in the CPU tests), issued per sub-batch so that it overlaps the demod of the next one --
as one collective (`all_gather_into_tensor`) or as a group of point-to-point transfers (`direct_gather_bits`).
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


class _WorkGroup:
    """The works of one grouped point-to-point exchange behind the `wait()` of a single collective's work handle."""

    def __init__(self, works, keep=None):
        self.works = list(works or [])
        self.keep = keep                     # the source tensor stays alive until the exchange is waited for

    def wait(self):
        for w in self.works:
            w.wait()


def direct_gather_bits(dist, recv, src, rank: int, world: int):
    """The same re-assembly as `all_gather_bits`, spelled as point-to-point transfers: every rank sends its rows to each of its
    W-1 peers and receives each peer's rows straight into its place in `recv` ([world*rows, row_bytes], rank-major); its own
    rows are a local copy.  One group of 2(W-1) transfers (`batch_isend_irecv`: one ncclGroup under RCCL), peers visited at
    distance 1, 2, ... so that at every distance each rank talks to a different pair.  On a node whose GPUs are fully connected
    by point-to-point xGMI links this puts ONE transfer per direction on every link and takes one hop -- which
    form is faster on a given node is a measurement: `bench.py --gather auto` times both before the timed loop and keeps the
    faster.  Returns a handle with `wait()`."""
    rows = src.shape[0]
    slots = recv.view(world, rows, recv.shape[1])
    slots[rank].copy_(src, non_blocking=True)
    ops = []
    for k in range(1, world):
        to, frm = (rank + k) % world, (rank - k) % world
        ops.append(dist.P2POp(dist.isend, src, to))
        ops.append(dist.P2POp(dist.irecv, slots[frm], frm))
    return _WorkGroup(dist.batch_isend_irecv(ops) if ops else [], keep=src)


def reassemble(torch, recv_list, world: int):
    """[world, n_frames, row_bytes] copy of the whole bit-stream from the per-sub-batch receive buffers."""
    return torch.cat([r.view(world, r.shape[0] // world, r.shape[1]) for r in recv_list], dim=1)


class GatherPipeline:
    """Step loop of the N > 1 path: per sub-batch `produce` (the demod launch) followed by an asynchronous all-gather of the
    rows it wrote, with `generations` sets of bit / receive buffers.  A buffer set is reused every `generations` steps and the
    launch stream waits for the gather that last used it only then, so the gathers of step i run under the demod of step i+1:
    the loop is bound by max(demod, fabric), not by their sum.  `drain()` completes everything outstanding.

    With world == 1 nothing is gathered (one generation, `produce` only).  `host_staging` gathers host copies of the rows
    (gloo rehearsal of the control flow on a box with fewer GPUs than ranks)."""

    def __init__(self, dist, torch, world, bounds, n_rows, row_bytes, device, recv_device=None, generations=2, host_staging=False,
                 gather_at_world_1=False, algo="collective", rank=0):
        self.dist, self.torch, self.world, self.bounds = dist, torch, world, list(bounds)
        if algo not in ("collective", "direct"):
            raise ValueError("algo must be 'collective' (all_gather_into_tensor) or 'direct' (grouped point-to-point transfers)")
        self.algo, self.rank = algo, rank
        # a one-rank group still runs the collectives when asked to (the only way to exercise the RCCL calls on a 1-GPU box)
        self.gather = world > 1 or (gather_at_world_1 and dist is not None)
        self.gen = generations if self.gather else 1
        self.host_staging = host_staging
        self.bits = [torch.empty((n_rows, row_bytes), dtype=torch.uint8, device=device) for _ in range(self.gen)]
        self.recv = None
        if self.gather:
            self.recv = [alloc_gather_buffers(torch, world, self.bounds, row_bytes, recv_device or device) for _ in range(self.gen)]
        self.pending = [[None] * len(self.bounds) for _ in range(self.gen)]
        self.steps = 0

    def step(self, produce):
        """produce(bits, f0, f1): fill rows [f0, f1) of `bits` (asynchronously on the current stream)."""
        g = self.steps % self.gen
        bits = self.bits[g]
        for ci, (f0, f1) in enumerate(self.bounds):
            if self.pending[g][ci] is not None:
                self.pending[g][ci].wait()          # stream-ordered: the gather that read bits[f0:f1] `generations` steps ago
                self.pending[g][ci] = None
            produce(bits, f0, f1)
            if self.gather:
                src = bits[f0:f1].cpu() if self.host_staging else bits[f0:f1].contiguous()
                self.pending[g][ci] = self.gather_rows(self.recv[g][ci], src)
        self.steps += 1

    def gather_rows(self, recv, src):
        """Start the re-assembly of one sub-batch (asynchronous; returns a handle with wait())."""
        if self.algo == "direct":
            return direct_gather_bits(self.dist, recv, src, self.rank, self.world)
        return self.dist.all_gather_into_tensor(recv, src, async_op=True)

    def drain(self):
        for row in self.pending:
            for ci, w in enumerate(row):
                if w is not None:
                    w.wait()
                    row[ci] = None

    def last(self):
        """(bits, receive buffers) of the most recent step."""
        g = (self.steps - 1) % self.gen
        return self.bits[g], (self.recv[g] if self.recv is not None else None)
