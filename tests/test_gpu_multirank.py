"""Two real ranks (two processes, each with its own BatchDetector) on the one GPU of the test box: contiguous frame
blocks per rank, no data-path collective, a gloo gather of the result records -- the replacement of the reference's
PBS job fan-out (lfd/createjobs/createjobs.py:173-202).  The gathered records equal a single-process run and the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, world, n, extra=(), timeout=600):
    """Start ``world`` rank programs on device 0, wait for all of them, return the array rank 0 gathered.  Whatever
    happens -- a rank failing, a timeout -- no rank outlives this function (a rank left blocked in the gloo gather would
    keep the GPU), and a failure reports every rank's stderr."""
    port = _free_port()
    out = str(tmp_path / "gathered.npy")
    logs = [open(tmp_path / f"rank{r}.err", "w+") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(r), str(world), str(port),
                               str(n), "0", out, *[str(x) for x in extra]], env=dict(os.environ), cwd=ROOT, stderr=logs[r])
             for r in range(world)]
    try:
        codes = []
        for p in procs:
            try:
                codes.append(p.wait(timeout=timeout))
            except subprocess.TimeoutExpired:
                codes.append("timeout")
                break                                                 # (the others are stuck behind it: killed below)
        if codes != [0] * world:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            tails = []
            for r, f in enumerate(logs):
                f.flush()
                f.seek(0)
                tails.append(f"--- rank {r} ---\n" + f.read()[-3000:])
            pytest.fail(f"rank exit codes {codes}\n" + "\n".join(tails))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for f in logs:
            f.close()
    return np.load(out)


def test_two_ranks_on_one_device(tmp_path, oracle):
    from lfd_amd import _native, batch, synth
    from lfd_amd.detecttrails import default_params
    n, world = 11, 2                                                  # ragged: 6 + 5 frames
    got = run_ranks(tmp_path, world, n)
    assert got.dtype == _native.RESULT_DTYPE and len(got) == n
    assert batch.shard_bounds(n, world) == [(0, 6), (6, 11)]
    pb, pd, prs = default_params()
    kw = {k: v for k, v in prs.items() if k != "debug"}
    rs_g, rs_o = _native.make_rs_params("r", **kw), oracle.rs_params("r", **kw)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(n)])
    with _native.Context(0, 1489, 2048, n) as ctx:
        single = ctx.detect_batch(np.stack(frames), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
    assert got.tobytes() == single.tobytes()
    for i in (0, 5, 6, 10):                                           # both sides of the block boundary
        want = oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o)
        assert all(got[i][k].item() == v for k, v in want.items()), (i, want, got[i])
