"""N > 1: frame-parallel sharding with no data-path collective, exercised with two gloo ranks
on the CPU (the per-rank detection itself is GPU work and is stubbed by a deterministic
function of the frame index here; what is tested is the partitioning and the result gather)."""
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from lfd_amd import _native, batch


def _fake_records(k0, k1):
    rec = np.zeros(k1 - k0, _native.RESULT_DTYPE)
    ks = np.arange(k0, k1)
    rec["found"] = ks % 3
    rec["rho"] = (20 * ks + 10).astype(np.float32)
    rec["theta"] = (ks % 180).astype(np.float32)
    rec["x1"] = -ks
    rec["y2"] = ks * 7
    return rec


def _worker(rank, world, n_frames, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = batch.shard_range(n_frames, rank, world)
    full = batch.gather_results(_fake_records(a, b), n_frames)
    q.put((rank, a, b, full.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, n_frames, port=None):
    port = port or _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, n_frames, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


def test_two_ranks_even_split():
    out = _run(2, 64)
    assert [(r, a, b) for r, a, b, _ in out] == [(0, 0, 32), (1, 32, 64)]
    want = _fake_records(0, 64).tobytes()
    assert all(buf == want for *_, buf in out)


def test_two_ranks_ragged_split():
    out = _run(2, 7)
    assert [(a, b) for _, a, b, _ in out] == [(0, 4), (4, 7)]
    want = _fake_records(0, 7).tobytes()
    assert all(buf == want for *_, buf in out)


def test_single_process_passthrough():
    rec = _fake_records(0, 5)
    assert batch.gather_results(rec, 5) is rec


def test_eight_ranks_config3_blocks():
    """BASELINE configs[3]'s placement: 8 192 frames over eight ranks, 1 024 each (and a ragged 8 190), the records of all ranks
    gathered on every rank -- eight gloo processes on the CPU (the one part of the eight-GPU run that needs no GPU)."""
    out = _run(8, 8192)
    assert [(a, b) for _, a, b, _ in out] == [(1024 * r, 1024 * (r + 1)) for r in range(8)]
    want = _fake_records(0, 8192).tobytes()
    assert all(buf == want for *_, buf in out)
    out = _run(8, 8190)
    assert [(a, b) for _, a, b, _ in out][-1] == (7168, 8190)
    want = _fake_records(0, 8190).tobytes()
    assert all(buf == want for *_, buf in out)
