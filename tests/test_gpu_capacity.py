"""Workspace sizing (include/lfdmi.h: lfdmi_caps), the worst-case spill path, and BASELINE configs[4] at batch:
4096x4096 float32 frames through the dim pass with a 9x9 erosion and HoughLines at rho 20 / 10 / 5, against the oracle."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def params():
    from lfd_amd.detecttrails import default_params
    return default_params()


def same(rec_gpu, rec_oracle):
    return all(rec_gpu[k].item() == v for k, v in rec_oracle.items())


def test_lsst_batch_multiscale_vs_oracle(oracle):
    """configs[4]: 8 LSST-size frames in ONE launch sequence of a compact 8-slot workspace; every record at every
    Hough scale equals the oracle's process_dim with houghMethod = rho (reference call sites: processfield.py:453-506;
    the reference always uses one rho, so the multi-scale part is GPU == oracle self-consistency)."""
    import torch
    from lfd_amd import _native, synth
    _, pd, _ = params()
    pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    rhos = [20.0, 10.0, 5.0]
    n = 8
    frames, _ = synth.make_frames(0, n, synth.LSST_SHAPE, with_catalog=False)

    def cpu(job):
        i, rho = job
        return oracle.process_dim(frames[i].copy(), dict(pd, houghMethod=rho), flip=True)

    jobs = [(i, r) for i in range(n) for r in rhos]
    with ThreadPoolExecutor(12) as ex:
        want = dict(zip(jobs, ex.map(cpu, jobs)))
    with _native.Context(0, 4096, 4096, n) as ctx:
        assert ctx.workspace_bytes() < n * 260e6                    # ~7.9 B/px of tables + fixed Hough storage per slot
        res = ctx.process_multiscale(frames, pd, rhos, dim=True, flip=True)          # host frames, staged by the library
        d = torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        res_d = ctx.process_multiscale(d, pd, rhos, dim=True, flip=True)             # device-resident, used in place
        assert res.tobytes() == res_d.tobytes()
        assert ctx.spill_count() == 0                               # sky frames stay inside the default capacities
        runs = ctx.get_counters(0, n)[:, 12]
        assert runs.max() <= 32768, runs                            # ... and on the per-frame LDS kernels
        one = ctx.process_dim(frames[:2], dict(pd, houghMethod=10.0), flip=True)[0]   # the single-scale entry point agrees
        assert one.tobytes() == res[1, :2].tobytes()
    found = set()
    for s, rho in enumerate(rhos):
        for i in range(n):
            assert same(res[s, i], want[(i, rho)]), (rho, i, want[(i, rho)], res[s, i])
            found.add(want[(i, rho)]["found"])
    assert found == {0, 2}                                          # bright streaks survive a 9x9 erosion, dim ones do not


def test_tiny_capacities_spill_to_the_worst_case_workspace(oracle, monkeypatch):
    """A workspace whose tables are far too small for the frames and may not grow (LFDMI_GROW=0): every overflow (run tables,
    contour keys, row slots, Hough chunk lists, peak lists) is flagged on the device before any table is indexed past its end,
    and the frame is run again through the worst-case workspace: records equal those of a default context and of the oracle."""
    from lfd_amd import _native, synth
    monkeypatch.setenv("LFDMI_GROW", "0")
    pb, pd, prs = params()
    kw = {k: v for k, v in prs.items() if k != "debug"}
    rs_g, rs_o = _native.make_rs_params("r", **kw), oracle.rs_params("r", **kw)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(6)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 6) as ref_ctx:
        ref = ref_ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        assert ref_ctx.spill_count() == 0
    for caps in ({"run_cap": 3000}, {"key_cap": 64}, {"slot_cap": 2000}, {"list_cap": 1500}, {"peak_cap": 256},
                 {"run_cap": 3000, "key_cap": 64, "slot_cap": 2000, "list_cap": 1500, "peak_cap": 256}):
        with _native.Context(0, 1489, 2048, 6, caps=caps) as ctx:
            res = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
            assert res.tobytes() == ref.tobytes(), caps
            assert ctx.spill_count() > 0, caps
            # per-pass entry points and the host-frame path take the same route
            rb, _, _ = ctx.process_bright(np.ascontiguousarray(batch[:2, ::-1]), pb)
            for i in range(2):
                assert same(rb[i], oracle.process_bright(np.ascontiguousarray(batch[i, ::-1]), pb)), caps
    for i in (0, 1):
        assert same(ref[i], oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o))
    with _native.Context(0, 1489, 2048, 2, caps="worst") as ctx:        # every table at its theoretical maximum: never spills
        res = ctx.detect_batch(batch[:2].copy(), pb, pd, {k: v[:2] for k, v in packed.items()}, rs_g)
        assert res.tobytes() == ref[:2].tobytes() and ctx.spill_count() == 0


def test_tables_grow_instead_of_spilling_every_time(oracle):
    """Round 4: a context whose tables overflow enlarges them (k_finalize leaves each frame's demands in its record) and runs
    the chunk again, instead of sending every such frame alone through the worst-case workspace on every call: same records,
    the growth happens once, later calls are plain fast-path calls."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    kw = {k: v for k, v in prs.items() if k != "debug"}
    rs_g = _native.make_rs_params("r", **kw)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(6)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 6) as ref_ctx:
        ref = ref_ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        ref_bytes = ref_ctx.workspace_bytes()
    for caps in ({"run_cap": 3000}, {"key_cap": 64}, {"slot_cap": 2000}, {"list_cap": 1500}, {"peak_cap": 256},
                 {"run_cap": 3000, "key_cap": 64, "slot_cap": 2000, "list_cap": 1500, "peak_cap": 256}):
        with _native.Context(0, 1489, 2048, 6, caps=caps) as ctx:
            small = ctx.workspace_bytes()
            res = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
            assert res.tobytes() == ref.tobytes(), caps
            st = ctx.stats()
            assert st["cap_growths"] >= 1 and st["spilled_frames"] == 0, (caps, st)
            assert small < ctx.workspace_bytes() < 4 * ref_bytes, caps          # grown to what the frames asked for, not to the worst case
            res = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)          # the second call finds its tables large enough
            st2 = ctx.stats()
            assert res.tobytes() == ref.tobytes() and st2["cap_growths"] == st["cap_growths"] and st2["chunks"] == st["chunks"] + 1, (caps, st, st2)
            rb, _, _ = ctx.process_bright(np.ascontiguousarray(batch[:2, ::-1]), pb)   # the per-pass entry points share the tables
            for i in range(2):
                assert same(rb[i], oracle.process_bright(np.ascontiguousarray(batch[i, ::-1]), pb)), caps


def test_spilled_big_endian_frames_keep_their_catalogue_entry(monkeypatch):
    """A raw big-endian frame is a read-only input: what the worst-case rerun of a spilled frame uploads is not blotted, so
    the rerun takes the frame's own catalogue entry (frames 1 .. 5 of a batch: the entry is not the first one); native host
    frames, blotted by then, go without.  Records equal those of a context that never spills."""
    from lfd_amd import _native, synth
    monkeypatch.setenv("LFDMI_GROW", "0")
    pb, pd, prs = params()
    rs_g = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(6)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 6) as ref_ctx:
        ref = ref_ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        none = ref_ctx.detect_batch(batch.copy(), pb, pd)              # (the stars matter: without the catalogue records differ)
        assert ref_ctx.spill_count() == 0 and none.tobytes() != ref.tobytes()
    be = batch.astype(">f4")
    with _native.Context(0, 1489, 2048, 6, caps={"key_cap": 64, "slot_cap": 2000}) as ctx:
        got = ctx.detect_batch(be, pb, pd, packed, rs_g)
        assert ctx.spill_count() > 0
        assert got.tobytes() == ref.tobytes()
        assert np.array_equal(be.astype(np.float32), batch)
        n0 = ctx.spill_count()
        assert ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g).tobytes() == ref.tobytes() and ctx.spill_count() > n0


def test_operator_entry_points_spill_too(oracle):
    """Canny / fit_minAreaRect / HoughLines on a dense random image through a workspace with tiny tables."""
    from lfd_amd import _native
    rng = np.random.default_rng(5)
    img = (rng.random((300, 400)) < 0.5).astype(np.uint8) * 255
    smooth = rng.integers(0, 256, (300, 400), dtype=np.uint8)
    with _native.Context(0, 300, 400, 2, caps={"run_cap": 500, "key_cap": 50, "slot_cap": 500, "list_cap": 300, "peak_cap": 64}) as ctx:
        assert np.array_equal(ctx.canny(smooth, 50, 150), oracle.canny(smooth, 50, 150))
        det, box, nb = ctx.fit_min_area_rect(smooth)
        det_o, box_o, nb_o = oracle.fit_min_area_rect(smooth)
        assert det == det_o and nb == nb_o and np.array_equal(box, box_o)
        lines, n = ctx.hough_lines(img, 20, max_lines=50)
        lines_o, n_o = oracle.hough_lines(img, 20, max_lines=50)
        assert n == n_o and np.array_equal(lines, lines_o)
        assert np.array_equal(ctx.hough_accum(img, 7.5), oracle.hough_accum(img, 7.5))
        assert ctx.spill_count() >= 4


def test_rho_finer_than_the_workspace_was_sized_for(oracle):
    """houghMethod = 2 with accumulators sized for rho >= 5 (the default): the call runs through the worst-case workspace."""
    from lfd_amd import _native, synth
    pb, pd, _ = params()
    img = synth.make_frame(0, with_catalog=False)[0][::-1].copy()
    with _native.Context(0, 1489, 2048, 2) as ctx:
        p = dict(pb, houghMethod=2, dro=40)
        res, _, _ = ctx.process_bright(img, p)
        assert same(res, oracle.process_bright(img, p))
        both = ctx.detect_batch(img[::-1].copy()[None], p, dict(pd, houghMethod=3))[0]
        assert same(both, oracle.detect_frame(img[::-1].copy(), p, dict(pd, houghMethod=3)))


def test_get_stage_checks_the_shape_of_the_last_call():
    from lfd_amd import _native
    with _native.Context(0, 256, 256, 1) as ctx:
        with pytest.raises(_native.NativeError):
            ctx.get_stage(0, _native.STAGE_EQU, 256, 256)            # nothing has run yet
        a = np.zeros((64, 128), np.uint8)
        a[20:30, 40:90] = 200
        ctx.dilate(a, np.ones((3, 3), np.uint8))
        with pytest.raises(_native.NativeError):
            ctx.get_stage(0, _native.STAGE_EQU, 256, 256)            # the last call worked on 64 x 128
        out = ctx.get_stage(0, _native.STAGE_EQU, 64, 128)
        assert out.shape == (64, 128) and out[25, 60] == 200


@pytest.mark.parametrize("fill", ["1", "0"])
def test_wide_erosion_fills_only_what_the_tile_kernel_reads(oracle, monkeypatch, fill):
    """A 9 x 9 erosion in a batch context (no stage images kept) zero-fills its output plane only around the cells that hold
    anything; LFDMI_SPARSE_ERODE_FILL=0 fills all of it.  Frames with objects of every size next to tile borders, a second
    call on the same context with other frames (stale bytes of the first call lie around), records and edge maps vs the
    oracle."""
    from lfd_amd import _native
    from lfd_amd.detecttrails import default_params
    monkeypatch.setenv("LFDMI_SPARSE_ERODE_FILL", fill)
    _, pd, _ = default_params()
    pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    h, w = 512, 1024
    rng = np.random.default_rng(11)

    def frame(seed):
        r = np.random.default_rng(seed)
        img = r.normal(0.0, 0.6, (h, w)).astype(np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(14):                                           # blobs of radius 6 .. 30, some across tile borders
            cy, cx, rad = r.integers(0, h), r.integers(0, w), r.integers(6, 31)
            img[(yy - cy) ** 2 + (xx - cx) ** 2 < rad * rad] += r.uniform(5, 200)
        a, b = r.uniform(-0.8, 0.8), r.uniform(50, h - 50)
        img[np.abs(yy - (a * xx + b)) < r.uniform(6, 12)] += 80.0     # a wide streak
        return img

    with _native.Context(0, h, w, 3) as ctx:
        ctx.set_stage_images(0)
        for seeds in ((1, 2, 3), (4, 5, 6)):
            frames = np.stack([frame(s) for s in seeds])
            res, _, _ = ctx.process_dim(frames.copy(), pd)
            for i in range(3):
                want, _, _ = oracle.process_dim(frames[i].copy(), pd, want_images=True)
                assert all(res[i][k].item() == v for k, v in want.items()), (fill, seeds, i)
                edges = ctx.get_stage(i, _native.STAGE_BOX, h, w)
                assert np.array_equal(edges != 0, oracle.process_dim(frames[i].copy(), pd, want_images=True)[2] != 0)


def test_stage_images_are_kept_only_where_asked_for(oracle):
    """lfdmi_set_stage_images: the per-pass calls keep the 8-bit stage images by default and lfdmi_detect_batch does not;
    a batch detector switches them off (get_stage then refuses instead of handing out stale bytes), mode 1 keeps them in
    detect_batch too; records and the always-available edge / box images are the same in every mode."""
    from lfd_amd import _native, synth
    from lfd_amd.batch import BatchDetector
    from lfd_amd.detecttrails import default_params
    pb, pd, _ = default_params()
    h, w = 256, 512
    rng = np.random.default_rng(5)
    img = rng.normal(0.3, 0.8, (h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    img[np.abs(yy - (0.4 * xx + 20)) < 2.0] += 60.0
    want = oracle.process_dim(img, pd)
    with _native.Context(0, h, w, 2) as ctx:
        res, _, _ = ctx.process_dim(img.copy(), pd)
        assert res["found"] == want["found"] and res["rho"] == want["rho"] and res["theta"] == want["theta"]
        equ_kept = ctx.get_stage(0, _native.STAGE_EQU, h, w)              # default: kept by the per-pass call
        edges = ctx.get_stage(0, _native.STAGE_CANNY, h, w)
        ctx.set_stage_images(0)
        res0, _, _ = ctx.process_dim(img.copy(), pd)
        assert res0.tobytes() == res.tobytes()
        with pytest.raises(_native.NativeError):
            ctx.get_stage(0, _native.STAGE_EQU, h, w)
        assert np.array_equal(ctx.get_stage(0, _native.STAGE_CANNY, h, w), edges)
        ctx.set_stage_images(-1)
        rb = ctx.detect_batch(img.copy()[None], pb, pd)
        with pytest.raises(_native.NativeError):
            ctx.get_stage(0, _native.STAGE_EQU, h, w)                     # default: not kept by detect_batch
        ctx.set_stage_images(1)
        rb1 = ctx.detect_batch(img.copy()[None], pb, pd)
        assert rb1.tobytes() == rb.tobytes()
        assert ctx.get_stage(0, _native.STAGE_EQU, h, w).shape == equ_kept.shape
    det = BatchDetector(0, (h, w), 2)
    try:
        det.multiscale(img.copy()[None], pd, [20.0, 10.0], dim=True, flip=False)
        with pytest.raises(_native.NativeError):
            det.ctx.get_stage(0, _native.STAGE_EQU, h, w)
    finally:
        det.close()


def test_default_workspace_sizes_for_the_baseline_batches():
    """Bytes per in-flight frame of the default capacities: 256 SDSS frames and 256 LSST-size frames
    both fit one GPU with room to spare (the theoretical-maximum layout took ~95 GB and ~470 GB)."""
    from lfd_amd import _native
    with _native.Context(0, 1489, 2048, 256) as ctx:
        b = ctx.workspace_bytes()
        assert b < 256 * 32e6, b
    with _native.Context(0, 4096, 4096, 256) as ctx:
        b = ctx.workspace_bytes()
        assert b < 256 * 170e6, b
