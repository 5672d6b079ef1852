"""The per-GPU shares of the BASELINE configurations that need eight GPUs as a whole:

  configs[3]  8 192 SDSS frames frame-parallel over 8 GPUs  -> 1 024 frames per GPU, four 256-frame chunks of one BatchDetector
                                                              (single process, and as two ranks of 512 on the one device);
  configs[4]  2 048 frames of 4096 x 4096 over 8 GPUs       -> 256 frames per GPU in ONE launch sequence of a 256-slot workspace,
                                                              dim pass with a 9 x 9 erosion, HoughLines at rho 20 / 10 / 5.

What cannot be tested on one GPU is the placement of eight such shares on eight devices; there is no data-path
collective between them (lfd/createjobs/createjobs.py:173-202: the reference runs disjoint run lists as separate jobs).
Frame i of a share is synthetic frame i % DISTINCT: determinism, permutation equivariance, identical records for the
repeats, the oracle on a sample, and no frame spilled to the worst-case workspace."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from test_gpu_multirank import run_ranks

pytestmark = pytest.mark.gpu

SHARE, DISTINCT = 1024, 128


def params():
    from lfd_amd.detecttrails import default_params
    return default_params()


def same(rec_gpu, rec_oracle):
    return all(rec_gpu[k].item() == v for k, v in rec_oracle.items())


@pytest.fixture(scope="module")
def sdss_share():
    """(128 distinct host frames, their catalogues, the records of the 1 024-frame share from one BatchDetector)."""
    import torch
    from lfd_amd import _native, synth
    from lfd_amd.batch import BatchDetector
    pb, pd, prs = params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    base, cats = synth.make_frames(0, DISTINCT, synth.SDSS_SHAPE)
    idx = np.arange(SHARE) % DISTINCT
    dbase = torch.from_numpy(base).cuda()
    packed = synth.pack_catalogs([cats[i] for i in idx])
    dcat = {k: torch.from_numpy(v).cuda() for k, v in packed.items()}
    det = BatchDetector(0, synth.SDSS_SHAPE, inflight=256)            # four chunks of 256
    try:
        frames = dbase[torch.from_numpy(idx).cuda()].contiguous()     # 12.5 GB, device-resident
        torch.cuda.synchronize()
        res = det.detect(frames, pb, pd, dcat, rs)
        res2 = det.detect(frames, pb, pd, dcat, rs)                   # (remove_stars has blotted the frames: idempotent)
        assert res.tobytes() == res2.tobytes(), "not deterministic"
        # permutation equivariance: another order of the same frames, through other slots and chunks
        perm = np.random.default_rng(5).permutation(SHARE)
        tperm = torch.from_numpy(perm).cuda()
        pframes = frames[tperm].contiguous()
        del frames
        pcat = {k: v[tperm].contiguous() for k, v in dcat.items()}
        torch.cuda.synchronize()
        resp = det.detect(pframes, pb, pd, pcat, rs)
        assert resp.tobytes() == res[perm].tobytes(), "records depend on the position in the batch"
        assert det.spill_count() == 0
        assert len(det.ctxs) == 1 and det.ctx.max_inflight == 256
    finally:
        det.close()
        del dbase
        torch.cuda.empty_cache()
    return base, cats, res


def test_config3_share_single_process(sdss_share, oracle):
    base, cats, res = sdss_share
    from lfd_amd import _native
    pb, pd, prs = params()
    rs_o = oracle.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    assert res.dtype == _native.RESULT_DTYPE and len(res) == SHARE
    assert (res["status"] == 0).all()
    for rep in range(1, SHARE // DISTINCT):                           # the eight copies of a frame give one record
        assert res[rep * DISTINCT:(rep + 1) * DISTINCT].tobytes() == res[:DISTINCT].tobytes()
    sample = list(range(0, DISTINCT, 8))                              # 16 of the distinct frames against the oracle

    def cpu(i):
        return oracle.detect_frame(base[i].copy(), pb, pd, cats[i], rs_o)

    with ThreadPoolExecutor(8) as ex:
        want = list(ex.map(cpu, sample))
    for i, w in zip(sample, want):
        assert same(res[i], w), (i, w, res[i])
    assert {int(f) for f in res["found"]} == {0, 1, 2}


def test_config3_share_as_two_ranks(sdss_share, tmp_path):
    """The same 1 024-frame share as two ranks of 512 device-resident frames (two processes on the one device, 256 slots
    each, gloo gather): the gathered records equal the single-process run."""
    _, _, res = sdss_share
    got = run_ranks(tmp_path, 2, SHARE, extra=(DISTINCT, 256), timeout=900)
    assert got.tobytes() == res.tobytes()


def test_config4_share_all_256_slots(oracle):
    """256 frames of 4096 x 4096 (32 distinct x 8) in one launch sequence of a 256-slot workspace, three Hough scales; the
    first eight distinct frames are checked against the oracle at every scale."""
    import torch
    from lfd_amd import synth
    from lfd_amd.batch import BatchDetector
    _, pd, _ = params()
    pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    rhos = [20.0, 10.0, 5.0]
    n, distinct = 256, 32
    base, _ = synth.make_frames(0, distinct, synth.LSST_SHAPE, with_catalog=False)

    def cpu(job):
        i, rho = job
        return oracle.process_dim(base[i].copy(), dict(pd, houghMethod=rho), flip=True)

    jobs = [(i, r) for i in range(8) for r in rhos]
    ex = ThreadPoolExecutor(12)
    futs = {j: ex.submit(cpu, j) for j in jobs}                        # the CPU works while the GPU does
    idx = torch.arange(n).cuda() % distinct
    dbase = torch.from_numpy(base).cuda()
    frames = dbase[idx].contiguous()                                  # 17.2 GB
    del dbase
    torch.cuda.synchronize()
    det = BatchDetector(0, synth.LSST_SHAPE, inflight=n)
    try:
        res = det.multiscale(frames, pd, rhos, dim=True, flip=True)
        res2 = det.multiscale(frames, pd, rhos, dim=True, flip=True)
        assert res.tobytes() == res2.tobytes()
        assert det.spill_count() == 0
        assert det.workspace_bytes() < n * 170e6
    finally:
        det.close()
        del frames
        torch.cuda.empty_cache()
    assert res.shape == (3, n) and (res["status"] == 0).all()
    for rep in range(1, n // distinct):
        assert res[:, rep * distinct:(rep + 1) * distinct].tobytes() == res[:, :distinct].tobytes()
    for (i, rho), f in futs.items():
        w = f.result()
        assert same(res[rhos.index(rho), i], w), (rho, i, w, res[rhos.index(rho), i])
    ex.shutdown()
