"""N>1 path on CPU: 2 gloo ranks, frame shards + sub-batched all-gather of the packed bit-stream.
The per-rank bits come from the CPU oracle here (the HIP path needs a GPU); what is tested is the sharding /
re-assembly bookkeeping bench.py uses with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ofdm_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_bits(frame_index):
    """packed hard bits of one frame, deterministic in the GLOBAL frame index"""
    N, cp, Kd, n_sym = 64, 16, 60, 8
    rng = np.random.default_rng(1000 + frame_index)
    bits = rng.integers(0, 2, 6 * Kd * 2)
    iq = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym).astype(np.complex64)
    rx = orc.RxOracle(n_sym, N, cp, N - 2, [1, 3], Kd, 100, 0.7)
    rx.work(iq, np.zeros(len(iq), np.complex64))
    rows = [r for r in range(n_sym) if r % 4 != 3]
    hb = orc.demap_hard(rx.est_data_freq[rows].ravel(), "QPSK")
    assert np.array_equal(hb, bits)
    return np.packbits(hb)


def _worker(rank, world, port, n_total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "lte-gnu-radio-code_amd")]
    from ofdm_mi355x import dist as od
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = od.shard_frames(n_total, world, rank)
    local = torch.from_numpy(np.stack([_rank_bits(first + f) for f in range(n)]))
    bounds = od.sub_batches(n, 3)
    recv = od.alloc_gather_buffers(torch, world, bounds, local.shape[1], "cpu")
    works = [od.all_gather_bits(dist, recv[i], local, f0, f1, async_op=True) for i, (f0, f1) in enumerate(bounds)]
    for w in works:
        w.wait()
    dist.barrier()
    q.put((rank, od.reassemble(torch, recv, world).numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_shard_and_allgather():
    world, n_total = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([_rank_bits(f) for f in range(n_total)]).reshape(world, n_total // world, -1)
    for r in range(world):
        assert np.array_equal(res[r], expect)          # every rank holds the whole re-assembled stream, in frame order


def test_shard_helpers():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "lte-gnu-radio-code_amd")]
    from ofdm_mi355x import dist as od
    assert od.shard_frames(16, 8, 3) == (6, 2)
    with pytest.raises(ValueError):
        od.shard_frames(10, 4, 0)
    assert od.sub_batches(10, 4) == [(0, 2), (2, 5), (5, 7), (7, 10)]
    assert od.sub_batches(2, 4) == [(0, 1), (1, 2)]


def _pipeline_worker(rank, world, port, q, algo="collective"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "lte-gnu-radio-code_amd")]
    from ofdm_mi355x import dist as od
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_rows, row_bytes, steps = 11, 96, 5
    bounds = od.sub_batches(n_rows, 4)
    pipe = od.GatherPipeline(dist, torch, world, bounds, n_rows, row_bytes, "cpu", algo=algo, rank=rank)
    history = []

    def make(step):
        def produce(bits, f0, f1):
            # the "demod" of this rank: a pattern that depends on (rank, step, row, byte)
            r = torch.arange(f0, f1).view(-1, 1)
            c = torch.arange(row_bytes).view(1, -1)
            bits[f0:f1] = ((rank * 131 + step * 17 + r * 7 + c) % 251).to(torch.uint8)
        return produce

    for s in range(steps):
        pipe.step(make(s))
    pipe.drain()
    bits, recv = pipe.last()
    full = od.reassemble(torch, recv, world)
    q.put((rank, full.numpy().copy(), bits.numpy().copy(), pipe.steps, pipe.gen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,algo", [(2, "collective"), (2, "direct"), (3, "direct")])
def test_two_rank_gather_pipeline_two_generations(world, algo):
    """The step loop bench.py runs at N > 1 (GatherPipeline: per sub-batch produce + async re-assembly, two buffer generations,
    deferred waits, drain) with gloo ranks on CPU: after 5 steps every rank holds every rank's rows of the LAST step -- with the
    re-assembly as one collective and as a group of point-to-point transfers (three ranks: two peers per rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_rows, row_bytes, last = 11, 96, 4
    r = np.arange(n_rows).reshape(-1, 1)
    c = np.arange(row_bytes).reshape(1, -1)
    expect = np.stack([((rk * 131 + last * 17 + r * 7 + c) % 251).astype(np.uint8) for rk in range(world)])
    for rank, full, bits, steps, gen in res:
        assert steps == 5 and gen == 2
        assert np.array_equal(full, expect)
        assert np.array_equal(bits, expect[rank])


def test_gather_pipeline_single_rank_is_a_plain_loop():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "lte-gnu-radio-code_amd")]
    from ofdm_mi355x import dist as od
    pipe = od.GatherPipeline(None, torch, 1, od.sub_batches(6, 1), 6, 8, "cpu")
    calls = []
    pipe.step(lambda bits, f0, f1: calls.append((f0, f1)))
    pipe.drain()
    assert calls == [(0, 6)] and pipe.gen == 1 and pipe.last()[1] is None
