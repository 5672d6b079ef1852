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


def test_two_ranks_on_one_device(tmp_path, oracle):
    from lfd_amd import _native, batch, synth
    from lfd_amd.detecttrails import default_params
    n, world = 11, 2                                                  # ragged: 6 + 5 frames
    port = _free_port()
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(r), str(world), str(port),
                               str(n), "0", out], env=env, cwd=ROOT) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
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
