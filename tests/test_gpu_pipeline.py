"""Whole-path parity: process_field_bright / process_field_dim / the batched full pipe on the
GPU vs the CPU oracle, the committed golden records, and size-independent properties at the
BASELINE sizes (2048x1489 batch, 4096x4096 dim pass with 9x9 erosion)."""
import json
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")


def params():
    from lfd_amd.detecttrails import default_params
    return default_params()


def rs_pair(oracle, prs, flt="r"):
    from lfd_amd import _native
    kw = {k: v for k, v in prs.items() if k != "debug"}
    return _native.make_rs_params(flt, **kw), oracle.rs_params(flt, **kw)


def same(rec_gpu, rec_oracle):
    return all(rec_gpu[k].item() == v for k, v in rec_oracle.items())


def test_golden_records(gpu_ctx, oracle):
    """HIP path == committed golden records (made by tests/golden/make_pipeline_golden.py) == oracle now."""
    from lfd_amd import synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    with open(GOLD) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 12
    for c in cases:
        img, cat, truth = synth.make_portable_frame(c["k"], tuple(c["shape"]))
        assert hashlib.sha256(img.tobytes()).hexdigest() == c["image_sha256"], "portable frame differs on this machine"
        got = gpu_ctx.detect_batch(img.copy()[None], pb, pd, synth.pack_catalogs([cat]), rs_g)[0]
        for k, v in c["record"].items():
            if k in ("rho", "theta"):
                assert np.float32(got[k]) == np.float32(v), (c["k"], k)
            else:
                assert got[k].item() == v, (c["k"], k, got[k].item(), v)
        assert same(got, oracle.detect_frame(img.copy(), pb, pd, cat, rs_o))


def test_batch_vs_oracle_and_chunking(gpu_ctx, oracle):
    """12 SDSS-size frames through an 8-slot context (two chunks), catalogue included."""
    from lfd_amd import synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(12)])
    batch = np.stack(frames)
    res = gpu_ctx.detect_batch(batch.copy(), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
    kinds = set()
    for i in range(12):
        want = oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o)
        assert same(res[i], want), (i, want, res[i])
        kinds.add(want["found"])
    assert kinds == {0, 1, 2}
    res2 = gpu_ctx.detect_batch(batch.copy(), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
    assert res.tobytes() == res2.tobytes()                          # deterministic
    # endpoints within +-1 px / theta within +-0.5 deg of the CPU path is implied by equality


def test_device_resident_frames_and_inplace_blotting(gpu_ctx, oracle):
    import torch
    from lfd_amd import synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in (3, 4)])
    packed = synth.pack_catalogs(list(cats))
    dframes = torch.from_numpy(np.stack(frames)).cuda()
    dcat = {k: torch.from_numpy(v).cuda() for k, v in packed.items()}
    torch.cuda.synchronize()
    res = gpu_ctx.detect_batch(dframes, pb, pd, dcat, rs_g)
    for i in range(2):
        ref = frames[i].copy()
        want = oracle.detect_frame(ref, pb, pd, cats[i], rs_o)       # ref is blotted in place
        assert same(res[i], want)
        assert np.array_equal(dframes[i].cpu().numpy(), ref)          # remove_stars mutated the frame identically


@pytest.mark.parametrize("min_flux,add_flux", [(0.02, 0.5), (0.4, 0.3), (0.5, 1.0), (0.0, 0.0), (-0.3, 0.7), (0.02, 1.5), (2.0, 0.5)])
def test_dim_front_end_from_the_bright_sweep_for_any_flux_knobs(oracle, min_flux, add_flux):
    """lfdmi_detect_batch rebuilds the dim pass's 8-bit image from the bright one plus one bit per pixel when 0 <= addFlux <= 1
    and minFlux <= 0.5 (with a shorter conversion for minFlux > 0), and converts the float frames a second time otherwise:
    every combination gives the oracle's records."""
    from lfd_amd import _native, synth
    pb, pd, _ = params()
    pd = dict(pd, minFlux=min_flux, addFlux=add_flux)
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in (2, 9, 11, 14)])
    frames[1, 200:210, 300:900] = np.nan                                # NaN / inf inside a frame: converted to 0 / 255 either way
    frames[2, 700:704, 100:1500] = np.inf
    frames[3, 50:60, 50:600] = -np.inf
    with _native.Context(0, 1489, 2048, 4) as ctx:
        res = ctx.detect_batch(frames.copy(), pb, pd)
    for i in range(4):
        want = oracle.detect_frame(frames[i].copy(), pb, pd)
        assert same(res[i], want), (min_flux, add_flux, i, want, res[i])


@pytest.mark.parametrize("shape", [(250, 1056), (133, 544), (77, 96), (64, 2080)])
def test_batch_on_widths_that_are_not_whole_bit_row_words(oracle, shape):
    """Widths that are multiples of 32 but not of 64 (the last 64-pixel word of a bit row is half empty), heights that are not
    multiples of the 16-row cell bands: lfdmi_detect_batch (bright sweep + bit-plane erosion) against the oracle."""
    from lfd_amd import _native
    pb, pd, _ = params()
    h, w = shape
    rng = np.random.default_rng(h * 10007 + w)
    frames = rng.normal(0.2, 0.7, (3, h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    frames[0][np.abs(yy - (0.3 * xx + 10)) < 2.5] += 90.0               # bright streak
    frames[1][np.abs(yy - (h - 5 - 0.2 * xx)) < 4.0] += 6.0             # dim streak running into the right border
    frames[2][:, w - 40:] += 3.0                                        # a slab touching the last, half-empty word
    with _native.Context(0, h, w, 2) as ctx:
        res = ctx.detect_batch(frames.copy(), pb, pd)
    for i in range(3):
        want = oracle.detect_frame(frames[i].copy(), pb, pd)
        assert same(res[i], want), (shape, i, want, res[i])


@pytest.mark.parametrize("knobs", [dict(contoursMode=0), dict(contoursMode=3), dict(gaussKernel=5), dict(gaussKernel=3, gaussSigma=1.2, contoursMode=0),
                                   dict(erodeKernel=np.ones((5, 5), np.uint8)), dict(erodeKernel=np.ones((7, 3), np.uint8)),
                                   dict(dilateKernel=np.ones((5, 7), np.uint8)), dict(dilateKernel=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8))])
def test_batch_with_other_knob_values(oracle, knobs):
    """lfdmi_detect_batch with contour modes, the optional Gaussian stage and other structuring elements in the dim pass
    (its front end comes out of the bright sweep where the erosion kernel is small; the plane it leaves is then read by a
    separate dilation or blur kernel rather than the fused tile kernel): records equal the oracle's."""
    from lfd_amd import _native, synth
    pb, pd, _ = params()
    pd = dict(pd, **knobs)
    if "contoursMode" in knobs or "gaussKernel" in knobs:
        pb = dict(pb, **{k: v for k, v in knobs.items() if k in ("contoursMode", "gaussKernel", "gaussSigma")})
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in (0, 4, 8, 13, 17)])
    with _native.Context(0, 1489, 2048, 3) as ctx:
        res = ctx.detect_batch(frames.copy(), pb, pd)
    for i in range(len(frames)):
        want = oracle.detect_frame(frames[i].copy(), pb, pd)
        assert same(res[i], want), (knobs, i, want, res[i])


def test_python_api_bright_and_dim(oracle):
    from lfd_amd import synth
    from lfd_amd.detecttrails import process_field_bright, process_field_dim, dictify_hough
    pb, pd, _ = params()
    for k in (0, 1, 2, 3, 8):
        img = synth.make_frame(k)[0][::-1].copy()                     # the caller flips (detecttrails.py:124)
        a, b = img.copy(), img.copy()
        det, res = process_field_bright(a, **pb)
        want = oracle.process_bright(b, pb)
        assert det == (want["found"] == 1)
        assert (a >= 0).all()                                        # img[img < 0] = 0 happened in place
        if det:
            assert res == dictify_hough(img.shape, (np.float32(want["rho"]), np.float32(want["theta"])))
            assert res == {k2: want[k2] for k2 in ("x1", "y1", "x2", "y2")}
        det2, res2 = process_field_dim(a, **pd)                       # same (already clamped) array, like process_field
        want2 = oracle.process_dim(b, pd, after_bright=True)
        assert det2 == (want2["found"] == 2)
        if det2:
            assert res2 == {k2: want2[k2] for k2 in ("x1", "y1", "x2", "y2")}
        c = img.copy()
        c[c < 0] = 0
        c[c < 0.02] = 0
        c[c > 0] += 0.5
        assert np.array_equal(a, c)                                   # dim's in-place masking


def test_config1_uint8_frame(oracle):
    """BASELINE configs[0]: uint8 frame, one streak, bright pass through the Python API."""
    from lfd_amd import synth
    from lfd_amd.detecttrails import process_field_bright, process_field_dim
    pb, pd, _ = params()
    img = synth.make_config1_frame()
    det, res = process_field_bright(img.copy(), **pb)
    want = oracle.process_bright(img, pb)
    assert det == (want["found"] == 1) and det
    assert res == {k: want[k] for k in ("x1", "y1", "x2", "y2")}
    theta_deg = np.rad2deg(want["theta"])
    assert abs(theta_deg - 125.0) <= 1.0                              # 35-degree streak -> normal at 125 degrees
    with pytest.raises(Exception):                                    # numpy: uint8 += float is a casting error
        process_field_dim(img.copy(), **pd)


def test_knob_variations(gpu_ctx, oracle):
    from lfd_amd import synth
    pb, pd, _ = params()
    img = synth.make_frame(0)[0][::-1].copy()
    for upd in ({"houghMethod": 10}, {"nlinesInSet": 5, "dro": 40}, {"dilateKernel": np.ones((6, 3), np.uint8)},
                {"lwTresh": 50}, {"minAreaRectMinLen": 8}, {"thetaTresh": 0.01, "lineSetTresh": 0.01}):
        p = dict(pb, **upd)
        res, le, lb = gpu_ctx.process_bright(img, p)
        want = oracle.process_bright(img, p)
        assert same(res, want), (upd, want)
    for upd in ({"erodeKernel": np.ones((5, 5), np.uint8)}, {"minFlux": 0.05, "addFlux": 1.5},
                {"erodeKernel": np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)}):
        p = dict(pd, **upd)
        res, le, lb = gpu_ctx.process_dim(img, p)
        want = oracle.process_dim(img, p)
        assert same(res, want), (upd, want)


def test_lsst_scale_dim_pass():
    """BASELINE configs[4]: 4096x4096 float32, dim pass with 9x9 erosion; plus the "multi-scale
    Hough" (HoughLines at rho 20, 10, 5 on the same image; no reference call site, self-consistency
    GPU == oracle)."""
    from lfd_amd import _native, synth
    from oracle import lfd_oracle as O
    _, pd, _ = params()
    pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    img = synth.make_frame(1, shape=synth.LSST_SHAPE, with_catalog=False)[0]
    with _native.Context(0, 4096, 4096, 2) as ctx:
        res, le, lb = ctx.process_dim(img, pd, flip=True)
        want, equ, box = O.process_dim(img, pd, flip=True, want_images=True)
        assert same(res, want), want
        assert np.array_equal(ctx.get_stage(0, _native.STAGE_EQU, 4096, 4096), equ)
        assert np.array_equal(ctx.get_stage(0, _native.STAGE_BOX, 4096, 4096), box)
        for rho in (20, 10, 5):
            l_o, n_o = O.hough_lines(equ, rho, max_lines=16)
            l_g, n_g = ctx.hough_lines(equ, rho, max_lines=16)
            assert n_o == n_g and np.array_equal(l_o, l_g)


def test_full_batch_properties_at_baseline_size(gpu_ctx):
    """configs[2] size (256 frames would take the oracle minutes): run 64 device-resident frames
    and check properties that need no oracle: determinism, permutation equivariance (frames are
    independent), flip symmetry of the reported line."""
    import torch
    from lfd_amd import synth
    pb, pd, prs = params()
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(16)] * 4)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    res = gpu_ctx.detect_batch(d, pb, pd)
    assert res[:16].tobytes() == res[16:32].tobytes() == res[48:].tobytes()
    perm = np.random.default_rng(0).permutation(64)
    dp = d[torch.from_numpy(perm).cuda()].contiguous()
    torch.cuda.synchronize()          # the context runs on its own stream: inputs must be complete
    res_p = gpu_ctx.detect_batch(dp, pb, pd)
    assert res_p.tobytes() == res[perm].tobytes()
    assert (res["status"] == 0).all() and set(np.unique(res["found"])) == {0, 1, 2}


@pytest.mark.parametrize("runcap", [0, 6000])
def test_frame_kernels_fallback_path(oracle, monkeypatch, runcap):
    """The per-frame LDS connectivity kernels hand frames with more runs than their table holds to
    the multi-workgroup run kernels (k_frame.h).  LFDMI_FRAME_RUNCAP lowers the table size: 0 sends
    every frame down the general path, 6000 splits a synthetic batch between the two.  Records
    and stage images must not depend on the path taken."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(6)])
    batch = np.stack(frames)
    cat = synth.pack_catalogs(list(cats))
    ref_ctx = _native.Context(0, 1489, 2048, 6)
    ref = ref_ctx.detect_batch(batch.copy(), pb, pd, cat, rs_g)
    ref_edges = [ref_ctx.get_stage(i, _native.STAGE_CANNY, 1489, 2048) for i in range(6)]
    ref_ctx.close()
    monkeypatch.setenv("LFDMI_FRAME_RUNCAP", str(runcap))
    ctx = _native.Context(0, 1489, 2048, 6)
    res = ctx.detect_batch(batch.copy(), pb, pd, cat, rs_g)
    assert res.tobytes() == ref.tobytes()
    for i in range(6):
        assert np.array_equal(ctx.get_stage(i, _native.STAGE_CANNY, 1489, 2048), ref_edges[i])
    runs = ctx.get_counters(0, 6)[:, 12]
    if runcap:
        assert (runs > runcap).any(), runs                              # the general path was taken
    want = oracle.detect_frame(frames[0].copy(), pb, pd, cats[0], rs_o)
    assert same(res[0], want)
    ctx.close()


def test_fused_prep_erode_matches_separate_kernels(oracle, monkeypatch):
    """Dim pass front end: the band kernel that converts, histograms and erodes in one pass
    (k_prep_erode) against the separate prep / erode kernels (LFDMI_FUSE_PREP_ERODE=0): same
    records, same edge images, for a 3x3 and a 5x5 erosion (taller kernels at this width keep the
    separate kernels: the halo rows of a short band would dominate)."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in (1, 3, 5, 7)])
    for ek in (3, 5):
        pdk = dict(pd, erodeKernel=np.ones((ek, ek), np.uint8))
        out = []
        for fused in ("1", "0"):
            monkeypatch.setenv("LFDMI_FUSE_PREP_ERODE", fused)
            ctx = _native.Context(0, 1489, 2048, 4)
            res = ctx.detect_batch(frames.copy(), pb, pdk)
            edges = [ctx.get_stage(i, _native.STAGE_CANNY, 1489, 2048) for i in range(4)]
            out.append((res.tobytes(), edges))
            ctx.close()
        assert out[0][0] == out[1][0]
        for a, b in zip(out[0][1], out[1][1]):
            assert np.array_equal(a, b)
    want = oracle.detect_frame(frames[0].copy(), pb, pd)
    ctx = _native.Context(0, 1489, 2048, 4)
    assert same(ctx.detect_batch(frames.copy(), pb, pd)[0], want)
    ctx.close()


def test_cell_bitmap_and_general_run_kernels_do_not_change_results(monkeypatch):
    """LFDMI_CELLBM=0 (every tile of the fused dilate + Canny kernel loads its input), LFDMI_FRAME_CCL=0 (multi-workgroup
    run kernels instead of the per-frame LDS kernels), LFDMI_DC_TILELIST=0 (the strip-walking fused kernel instead of the
    active-tile list), another split of the tile list over waves and LFDMI_DC_SPECIALIZE=0 (run-time instead of compile-time
    structuring-element sizes in the tile kernel), LFDMI_FUSE_DUAL=1 (one band-kernel sweep over the float frames feeds both
    passes) and LFDMI_DELTA_DIM=0 (the dim pass converts the float frames again instead of rebuilding its image from the
    bright image and one bit per pixel); round 3: LFDMI_PERM=0 / LFDMI_TILE_PERM=0 (frame slot == workgroup index: no XCD
    balancing of a pass's active frames / no sorting by tiles), LFDMI_VOTE_BALANCE=0 and LFDMI_VOTE_CLASSES=0 (fixed pieces per
    image, one chunk list per image in the Hough vote), LFDMI_SKY_FAST=0 (the bright sweep without its all-sky shortcut),
    LFDMI_FRAME_LDS=16384 (smaller label tables: busy frames take the general kernels), LFDMI_RECTS_PREP=0 (the wave-per-key
    rectangle kernels scan their hulls sequentially), LFDMI_SCAN_FUSED=0 (run scans as three launches instead of one with a look-back) -- against the default fast paths:
    identical records and edge images."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(6)])
    outs = []
    switches = ("LFDMI_CELLBM", "LFDMI_FRAME_CCL", "LFDMI_DC_TILELIST", "LFDMI_DC_PARTS", "LFDMI_DC_SPECIALIZE", "LFDMI_FUSE_DUAL", "LFDMI_DELTA_DIM",
                "LFDMI_PERM", "LFDMI_TILE_PERM", "LFDMI_VOTE_BALANCE", "LFDMI_VOTE_CLASSES", "LFDMI_SKY_FAST", "LFDMI_FRAME_LDS", "LFDMI_RECTS_PREP", "LFDMI_SCAN_FUSED")
    for env in ({}, {"LFDMI_CELLBM": "0"}, {"LFDMI_FRAME_CCL": "0"}, {"LFDMI_DC_TILELIST": "0"}, {"LFDMI_DC_TILELIST": "0", "LFDMI_CELLBM": "0"},
                {"LFDMI_DC_PARTS": "7"}, {"LFDMI_DC_SPECIALIZE": "0"}, {"LFDMI_FUSE_DUAL": "1"}, {"LFDMI_DELTA_DIM": "0"},
                {"LFDMI_PERM": "0"}, {"LFDMI_TILE_PERM": "0"}, {"LFDMI_VOTE_BALANCE": "0"}, {"LFDMI_VOTE_CLASSES": "0"}, {"LFDMI_SKY_FAST": "0"},
                {"LFDMI_FRAME_LDS": "16384"}, {"LFDMI_RECTS_PREP": "0"}, {"LFDMI_SCAN_FUSED": "0"},
                {"LFDMI_VOTE_BALANCE": "0", "LFDMI_VOTE_CLASSES": "0", "LFDMI_PERM": "0", "LFDMI_SKY_FAST": "0"}):
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = _native.Context(0, 1489, 2048, 6)
        res = ctx.detect_batch(frames.copy(), pb, pd)
        outs.append((res.tobytes(), [ctx.get_stage(i, _native.STAGE_CANNY, 1489, 2048) for i in range(6)]))
        ctx.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        for a, b in zip(o[1], outs[0][1]):
            assert np.array_equal(a, b)


def test_remove_stars_masked_in_the_sweep_or_filled_before_it(oracle, monkeypatch):
    """lfdmi_detect_batch with a catalogue: by default the bright sweep masks remove_stars' squares as it loads the values and
    the zero fill of a device-resident frame runs later, on a side stream (LFDMI_RS_FILL_AT picks the stage); LFDMI_RS_FOLD=0
    fills before the sweep.  Records, the frames the caller gets back (device-resident and host frames are blotted in place,
    big-endian ones are not touched) and the oracle agree in every mode; a flipped catalogue row (objects near the frame's
    edges, squares clipped like Python slices) is part of the batch."""
    import torch
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(5)])
    cats = [dict(c) for c in cats]
    for key in ("ROWC", "COLC"):                                       # frame 4: objects pushed to the edges and beyond
        cats[4][key] = np.array(cats[4][key], copy=True)
    half = len(cats[4]["ROWC"]) // 2
    cats[4]["ROWC"][:half] *= 0.02
    cats[4]["COLC"][half:] = 2040.0 + 0.05 * cats[4]["COLC"][half:]
    batch = np.stack(frames)
    packed = synth.pack_catalogs(cats)
    outs = []
    # (LFDMI_RS_SORT_MIN=0: the path of crowded catalogues -- squares sorted by first row, band-wise fold and fill -- on this one)
    for env in ({}, {"LFDMI_RS_FOLD": "0"}, {"LFDMI_RS_FILL_AT": "1"}, {"LFDMI_RS_FILL_AT": "3"}, {"LFDMI_RS_FILL_AT": "5"}, {"LFDMI_RS_FILL_AT": "9"},
                {"LFDMI_RS_SORT_MIN": "0"}, {"LFDMI_RS_SORT_MIN": "0", "LFDMI_RS_FOLD": "0"}):
        for k in ("LFDMI_RS_FOLD", "LFDMI_RS_FILL_AT", "LFDMI_RS_SORT_MIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with _native.Context(0, 1489, 2048, 5) as ctx:
            dev = torch.from_numpy(batch.copy()).cuda()
            r_dev = ctx.detect_batch(dev, pb, pd, packed, rs_g)
            torch.cuda.synchronize()
            host = batch.copy()
            r_host = ctx.detect_batch(host, pb, pd, packed, rs_g)
            be = batch.astype(">f4")
            r_be = ctx.detect_batch(be, pb, pd, packed, rs_g)
            assert np.array_equal(be.astype(np.float32), batch)
            outs.append((r_dev.tobytes(), r_host.tobytes(), r_be.tobytes(), dev.cpu().numpy(), host))
    for o in outs:
        assert o[0] == outs[0][0] == o[1] == o[2]
        assert np.array_equal(o[3], outs[1][3]) and np.array_equal(o[4], outs[1][3])
    res = np.frombuffer(outs[0][0], _native.RESULT_DTYPE)
    for i in (0, 1, 4):
        assert same(res[i], oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o))
        f = frames[i].copy()
        oracle.remove_stars(f, cats[i], rs_o)
        assert np.array_equal(f, outs[0][3][i])                         # the oracle blots its frame the same way


def test_fused_run_scan_gives_up_gracefully(oracle, monkeypatch):
    """k_scan_fused's workgroups wait (bounded) for the totals of the frame's earlier workgroups; when several processes share
    the GPU that wait can run into its bound.  With the bound set to zero polls (LFDMI_SCAN_SPIN=0) every such wait gives up:
    the frames are flagged, the context goes back to the three-launch scan for good and runs the chunk once more -- same
    records as an ordinary context, nothing sent to the worst-case workspace, through the batch entry point and the per-pass ones."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(4)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 4) as ctx:
        want = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        wb, _, _ = ctx.process_bright(np.ascontiguousarray(batch[:2, ::-1]), pb)
    monkeypatch.setenv("LFDMI_SCAN_SPIN", "0")
    with _native.Context(0, 1489, 2048, 4) as ctx:
        assert ctx.stats()["scan_fused_on"] == 1
        for _ in range(2):
            assert ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g).tobytes() == want.tobytes()
        st = ctx.stats()                                  # the switch is visible (lfdmi_get_stats), not a silent slowdown
        assert ctx.spill_count() == 0 and st["scan_giveups"] == 1 and st["scan_fused_on"] == 0 and st["chunks"] == 2, st
    with _native.Context(0, 1489, 2048, 4) as ctx:
        rb, _, _ = ctx.process_bright(np.ascontiguousarray(batch[:2, ::-1]), pb)
        assert rb.tobytes() == wb.tobytes() and ctx.spill_count() == 0 and ctx.stats()["scan_giveups"] == 1
    with _native.Context(0, 300, 400, 2) as ctx:          # the operator entry points rerun the chunk too (round 3: one worst-case rerun per image)
        from scipy import ndimage as ndi
        rng = np.random.default_rng(3)
        smooth = np.stack([(ndi.gaussian_filter(rng.random((300, 400)), 3.0) * 900).clip(0, 255).astype(np.uint8) for _ in range(2)])
        assert np.array_equal(ctx.canny(smooth, 50, 150)[1], oracle.canny(smooth[1], 50, 150))
        det, box, nb = ctx.fit_min_area_rect(smooth[0])
        det_o, box_o, nb_o = oracle.fit_min_area_rect(smooth[0])
        assert det == det_o and nb == nb_o and np.array_equal(box, box_o)
        assert ctx.spill_count() == 0 and ctx.stats()["scan_giveups"] == 1, ctx.stats()
    # the one-launch scan is tried again after a while (here: after one quiet chunk, then two, ...) and gives up again
    monkeypatch.setenv("LFDMI_SCAN_REARM", "1")
    with _native.Context(0, 1489, 2048, 4) as ctx:
        for _ in range(5):
            assert ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g).tobytes() == want.tobytes()
        assert ctx.stats()["scan_giveups"] >= 2 and ctx.spill_count() == 0, ctx.stats()
    monkeypatch.delenv("LFDMI_SCAN_REARM")
    # a bound of a few polls: some workgroups of a launch see their predecessors in time, others give up
    for spin in ("1", "3"):
        monkeypatch.setenv("LFDMI_SCAN_SPIN", spin)
        with _native.Context(0, 1489, 2048, 4) as ctx:
            for _ in range(2):
                assert ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g).tobytes() == want.tobytes(), spin
            assert ctx.spill_count() == 0
    assert same(want[1], oracle.detect_frame(frames[1].copy(), pb, pd, cats[1], rs_o))


def test_two_contexts_of_one_process_at_the_same_time(oracle):
    """Two contexts (own streams, own workspaces) driven by two host threads at once -- what BatchDetector(lanes=2) does: the
    look-back scans of the two can hold each other's CU slots like two processes do; whatever the scans decide (wait, give up,
    three launches), the records are those of a context working alone."""
    import threading
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(8)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 8) as ctx:
        want = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
    a, b = _native.Context(0, 1489, 2048, 4), _native.Context(0, 1489, 2048, 4)
    got = [None, None]

    def work(k, ctx):
        sl = slice(4 * k, 4 * k + 4)
        for _ in range(6):
            got[k] = ctx.detect_batch(batch[sl].copy(), pb, pd, {n: v[sl] for n, v in packed.items()}, rs_g)

    th = [threading.Thread(target=work, args=(k, c)) for k, c in enumerate((a, b))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert np.concatenate(got).tobytes() == want.tobytes()
    assert a.spill_count() == 0 and b.spill_count() == 0
    a.close(); b.close()
    assert same(want[5], oracle.detect_frame(frames[5].copy(), pb, pd, cats[5], rs_o))


def test_batch_detector_with_two_calls_in_flight(oracle):
    """BatchDetector(calls_in_flight=2): two contexts launching into one stream, a host thread each, calls dealt out in turn
    (bench.py's headline loop).  Eight calls over three different device-resident batches submitted at once: every future
    holds the records of the synchronous call on its batch; host frames and the multi-scale pass go the same way."""
    import torch
    from lfd_amd import synth
    from lfd_amd.batch import BatchDetector
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    dev = torch.device("cuda", 0)
    batches = []
    for b in range(3):
        frames, cats = zip(*[synth.make_frame(10 * b + k)[:2] for k in range(4)])
        packed = synth.pack_catalogs(list(cats))
        batches.append((np.stack(frames), {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}, frames, cats))
    one = BatchDetector(0, synth.SDSS_SHAPE, 4)
    want = [one.detect(torch.from_numpy(f).to(dev), pb, pd, c, rs_g) for f, c, _, _ in batches]
    with pytest.raises(RuntimeError):
        one.detect_async(batches[0][0], pb, pd, None, rs_g)
    one.close()
    det = BatchDetector(0, synth.SDSS_SHAPE, 4, calls_in_flight=2)
    assert len(det.ctxs) == 2
    order = [0, 1, 2, 0, 2, 1, 1, 0]
    dframes = [torch.from_numpy(batches[b][0]).to(dev) for b in order]          # (remove_stars works in place: a copy per call)
    futs = [det.detect_async(dframes[i], pb, pd, batches[b][1], rs_g) for i, b in enumerate(order)]
    for f, b in zip(futs, order):
        assert f.result().tobytes() == want[b].tobytes(), b
    host = det.detect_async(batches[1][0].copy(), pb, pd, {k: v.cpu().numpy() for k, v in batches[1][1].items()}, rs_g)
    assert host.result().tobytes() == want[1].tobytes()
    plain = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(4)])
    ms = [det.multiscale_async(torch.from_numpy(plain).to(dev), pd, [20.0, 10.0]) for _ in range(3)]
    ref = det.multiscale(torch.from_numpy(plain).to(dev), pd, [20.0, 10.0])
    assert all(m.result().tobytes() == ref.tobytes() for m in ms)
    assert det.spill_count() == 0
    det.close()
    assert same(want[2][1], oracle.detect_frame(batches[2][2][1].copy(), pb, pd, batches[2][3][1], rs_o))


def test_fused_run_scan_survives_the_wrap_of_its_epoch(monkeypatch):
    """k_scan_fused marks a workgroup's published totals with the launch's 22-bit epoch; when the epoch wraps the host clears
    the words and starts over.  A context started three launches before the wrap (LFDMI_SCAN_EPOCH0) gives the records of an
    ordinary one over several calls (four scans per call)."""
    from lfd_amd import _native, synth
    pb, pd, _ = params()
    frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(4)])
    with _native.Context(0, 1489, 2048, 4) as ctx:
        want = ctx.detect_batch(frames.copy(), pb, pd)
    monkeypatch.setenv("LFDMI_SCAN_EPOCH0", str((1 << 22) - 3))
    with _native.Context(0, 1489, 2048, 4) as ctx:
        for _ in range(3):
            assert ctx.detect_batch(frames.copy(), pb, pd).tobytes() == want.tobytes()
        assert ctx.spill_count() == 0


@pytest.mark.parametrize("caps", ["worst", None])
def test_dense_noise_frames_take_the_general_kernels_and_match_the_oracle(oracle, caps):
    """Frames the per-frame LDS kernels cannot hold (noise everywhere: far more than 32 768 runs) are
    flagged on the device, the chunk is run again with the general multi-workgroup kernels, and the records
    still equal the oracle's; later chunks of the same context launch the general kernels right away.  With the default
    (compact) capacities such frames exceed the run tables first: the context enlarges them and runs the chunk again."""
    from lfd_amd import _native
    pb, pd, prs = params()
    rng = np.random.default_rng(7)
    h, w = 600, 768
    frames = []
    for k in range(3):
        f = rng.uniform(1.0, 200.0, (h, w)).astype(np.float32)   # texture everywhere: ~60-100 k candidate runs per pass
        f[100 + 40 * k:108 + 40 * k, :] = 250.0                  # a bright bar so that something elongated exists
        frames.append(f)
    frames.append(np.zeros((h, w), np.float32))           # and an empty frame in the same chunk
    batch = np.stack(frames)
    ctx = _native.Context(0, h, w, 4, caps=caps)
    res = ctx.detect_batch(batch.copy(), pb, pd)
    if caps == "worst":
        runs = ctx.get_counters(0, 4)[:, 12]
        assert runs[:3].min() > 32768, runs               # really beyond the LDS tables
        assert ctx.spill_count() == 0
    else:                                                 # beyond the default run tables (N / 16): the tables grow (round 4; before: three
        st = ctx.stats()                                  # frames through the worst-case workspace on every call)
        assert st["cap_growths"] >= 1 and st["spilled_frames"] == 0, st
    for i in range(4):
        want = oracle.detect_frame(frames[i].copy(), pb, pd)
        assert same(res[i], want), (i, want, res[i])
    res2 = ctx.detect_batch(batch.copy(), pb, pd)         # general kernels now launched up front
    assert res2.tobytes() == res.tobytes()
    ctx.close()


@pytest.mark.parametrize("shape", [(16, 16), (33, 48), (64, 64), (100, 272), (130, 1040), (97, 131), (40, 2048)])
def test_small_and_odd_shapes_full_pipe(oracle, shape):
    """The whole pipe on small frames: widths that are / are not multiples of 16 and 64 (fused tile kernels vs
    the generic ones), a single tile row, frames narrower than a tile; bright, dim and empty streak kinds."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    h, w = shape
    ctx = _native.Context(0, h, w, 3)
    frames, cats = zip(*[synth.make_portable_frame(k, shape, n_star=max(2, h * w // 20000))[:2] for k in (0, 1, 2)])
    res = ctx.detect_batch(np.stack(frames).copy(), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
    for i in range(3):
        want = oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o)
        assert same(res[i], want), (shape, i, want, res[i])
    ctx.close()


def test_host_frames_are_blotted_in_place_without_a_copy_back(oracle):
    """Frames handed over as a host array: remove_stars must leave the caller's array exactly as the
    reference would (the library zero-fills it on the host from the squares the device computed; the
    frames themselves cross PCIe once), over two chunks and with a ragged last chunk."""
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    ks = list(range(40, 47))
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in ks])
    batch = np.stack(frames)
    ctx = _native.Context(0, 1489, 2048, 4)                     # 7 frames through 4 slots: chunks of 4 and 3
    res = ctx.detect_batch(batch, pb, pd, synth.pack_catalogs(list(cats)), rs_g)
    for i in range(len(ks)):
        ref = frames[i].copy()
        want = oracle.detect_frame(ref, pb, pd, cats[i], rs_o)  # ref is blotted in place
        assert same(res[i], want)
        assert np.array_equal(batch[i], ref)
    ctx.close()


def test_host_feed_ring_keeps_rows_order_and_blotting(oracle, monkeypatch):
    """Host frames go through the pinned double buffer of lfdmi_detect_batch (chunk k+1 uploads while chunk k is
    processed).  With LFDMI_FEED_MB=30 a chunk is two SDSS frames, so nine frames take five chunks and every buffer is
    reused: records, their order and the in-place blotting equal the device-resident run and the oracle."""
    import torch
    from lfd_amd import _native, synth
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    ks = list(range(60, 69))
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in ks])
    packed = synth.pack_catalogs(list(cats))
    d = torch.from_numpy(np.stack(frames)).cuda()
    dcat = {k: torch.from_numpy(v).cuda() for k, v in packed.items()}
    torch.cuda.synchronize()
    with _native.Context(0, 1489, 2048, 9) as ctx:
        ref = ctx.detect_batch(d, pb, pd, dcat, rs_g)
    blotted = d.cpu().numpy()
    for feed_mb in ("30", "0", None):
        if feed_mb is None:
            monkeypatch.delenv("LFDMI_FEED_MB", raising=False)
        else:
            monkeypatch.setenv("LFDMI_FEED_MB", feed_mb)
        batch = np.stack(frames)
        with _native.Context(0, 1489, 2048, 4) as ctx:
            res = ctx.detect_batch(batch, pb, pd, packed, rs_g)
            res2 = ctx.detect_batch(np.stack(frames), pb, pd, packed, rs_g)      # buffers reused by a second call
        assert res.tobytes() == ref.tobytes() == res2.tobytes(), feed_mb
        assert np.array_equal(batch, blotted), feed_mb
    for i in (0, 4, 8):
        assert same(ref[i], oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o))


@pytest.mark.parametrize("add_flux", [0.5, 0.999, 0.9995, 1.0])
def test_exact_ties_through_the_bit_plane_dim_front_end(oracle, add_flux):
    """Integer and half-integer pixel values (BZERO/BSCALE-like data): x = k + 0.5 with k even rounds to k in the bright pass
    and, for addFlux = 1, to k + 2 in the dim pass -- a difference the one-bit-per-pixel front end of lfdmi_detect_batch cannot
    hold, so such knob values must take the second float sweep.  The ERODED stage image (the plane rebuilt from bright byte +
    bit) is compared with the oracle's, not only the records."""
    from lfd_amd import _native
    pb, pd, _ = params()
    pd = dict(pd, addFlux=add_flux, minFlux=0.02)
    h, w = 256, 512
    rng = np.random.default_rng(17)
    base = rng.integers(0, 3, (h, w)).astype(np.float32)                 # 0, 1, 2
    base += np.where(rng.random((h, w)) < 0.5, np.float32(0.5), np.float32(0.0))   # ... and k + 0.5 ties
    yy, xx = np.mgrid[0:h, 0:w]
    frames = np.stack([base, base * np.float32(0.5), base + np.float32(253.0)])     # (last: ties next to the saturation at 255)
    frames[0][np.abs(yy - (0.25 * xx + 40)) < 3.0] = np.float32(6.5)
    frames[1][np.abs(yy - (200 - 0.3 * xx)) < 3.0] = np.float32(2.5)
    pb_never = dict(pb, lwTresh=1e9)                                     # the bright pass finds nothing: every frame reaches the dim pass
    with _native.Context(0, h, w, 4) as ctx:
        res = ctx.detect_batch(frames.copy(), pb_never, pd)
        for i in range(3):
            flipped = frames[i][::-1].copy()
            flipped[flipped < 0] = 0
            gray = oracle.prep(flipped, oracle.PREP_DIM, minFlux=pd["minFlux"], addFlux=pd["addFlux"])
            want_eroded = oracle.erode(oracle.equalize_hist(gray), pd["erodeKernel"])
            got = ctx.get_stage(i, _native.STAGE_ERODED, h, w)
            assert np.array_equal(got, want_eroded), (add_flux, i, int((got != want_eroded).sum()))
    for i in range(3):
        want = oracle.detect_frame(frames[i].copy(), pb_never, pd)
        assert same(res[i], want), (add_flux, i, want, res[i])


def test_frame_must_fit_the_context_in_both_dimensions():
    """A taller, narrower frame of smaller area than the context's: the band / tile tables are sized by max_h and max_w
    separately, so the call is refused (LFDMI_ERR_CAPACITY) instead of running past them; the context stays usable."""
    from lfd_amd import _native
    pb, pd, _ = params()
    with _native.Context(0, 256, 512, 2) as ctx:
        tall = np.zeros((1, 320, 256), np.float32)
        for call in (lambda: ctx.detect_batch(tall, pb, pd), lambda: ctx.process_bright(tall[0], pb),
                     lambda: ctx.canny(np.zeros((320, 256), np.uint8)), lambda: ctx.prep_u8(tall[0], _native.PREP_BRIGHT)):
            with pytest.raises(_native.NativeError) as e:
                call()
            assert e.value.code == _native.ERR_CAPACITY
        wide = np.zeros((1, 128, 576), np.float32)
        with pytest.raises(_native.NativeError) as e:
            ctx.detect_batch(wide, pb, pd)
        assert e.value.code == _native.ERR_CAPACITY
        ok = ctx.detect_batch(np.zeros((1, 256, 512), np.float32), pb, pd)
        assert ok[0]["status"] == 0 and ok[0]["found"] == 0


def test_host_feed_error_path_returns_promptly_and_leaves_the_context_usable(oracle, monkeypatch):
    """A host-frame batch large enough for the pinned double-buffer feed (>= 64 MB, several chunks) whose second chunk fails:
    the call returns its error without uploading the rest of the batch, and the same context then processes the batch
    correctly."""
    import time
    from lfd_amd import _native, synth
    monkeypatch.setenv("LFDMI_FEED_MB", "32")                           # chunks of two SDSS-size frames
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(8)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 4) as ctx:
        good = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)     # warm: buffers, streams
        ctx.debug_fail_chunk(1)
        t0 = time.perf_counter()
        with pytest.raises(_native.NativeError) as e:
            ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        dt = time.perf_counter() - t0
        assert e.value.code == _native.ERR_ARG and "injected" in str(e.value)
        assert dt < 2.0, dt
        again = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        assert again.tobytes() == good.tobytes()
    for i in (0, 1, 5):
        assert same(good[i], oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o))


def test_big_endian_and_pinned_frames(oracle, monkeypatch):
    """lfdmi_detect_batch_raw: the raw big-endian data unit of a FITS image (LFDMI_F32_BE), as ordinary host memory (staged:
    the small-batch path and the pinned double-buffer feed) and in memory from lfdmi_host_alloc (LFDMI_HOST_PINNED: uploaded
    in place, several chunks), gives the records of the native float32 frames; big-endian frames are a read-only input (only
    the device copy is blotted), native frames in pinned memory are blotted like any host frames."""
    from lfd_amd import _native, synth
    monkeypatch.setenv("LFDMI_FEED_MB", "40")                           # chunks of three SDSS-size frames
    pb, pd, prs = params()
    rs_g, rs_o = rs_pair(oracle, prs)
    frames, cats = zip(*[synth.make_frame(k)[:2] for k in range(8)])
    batch = np.stack(frames)
    packed = synth.pack_catalogs(list(cats))
    with _native.Context(0, 1489, 2048, 4) as ctx:
        want = ctx.detect_batch(batch.copy(), pb, pd, packed, rs_g)
        blotted = batch.copy()
        ctx.detect_batch(blotted, pb, pd, packed, rs_g)
        be = batch.astype(">f4")
        got = ctx.detect_batch(be, pb, pd, packed, rs_g)                # 97 MB: the staged feed
        assert got.tobytes() == want.tobytes()
        assert np.array_equal(be.astype(np.float32), batch)             # raw file bytes stay as they are
        small = batch[:2].astype(">f4")
        assert ctx.detect_batch(small, pb, pd, {k: v[:2] for k, v in packed.items()}, rs_g).tobytes() == want[:2].tobytes()
        pin = ctx.pinned_buffer(be.nbytes)
        try:
            view = pin.array.view(">f4").reshape(be.shape)
            view[...] = batch
            got = ctx.detect_batch(view, pb, pd, packed, rs_g, pinned=True)
            assert got.tobytes() == want.tobytes()
            assert np.array_equal(view.astype(np.float32), batch)
            nat = pin.array.view(np.float32).reshape(be.shape)          # native floats in pinned memory
            nat[...] = batch
            assert ctx.detect_batch(nat, pb, pd, packed, rs_g, pinned=True).tobytes() == want.tobytes()
            assert np.array_equal(nat, blotted)                         # ... are blotted in place
            one = ctx.detect_batch(nat[5], pb, pd, None, None, pinned=True)    # a single pinned (blotted) frame, no catalogue
            del view, nat
        finally:
            pin.close()
    assert same(one[0], oracle.detect_frame(np.ascontiguousarray(blotted[5]), pb, pd))
    assert same(want[3], oracle.detect_frame(frames[3].copy(), pb, pd, cats[3], rs_o))


def test_dim_pass_on_an_arbitrary_subset_of_the_slots(oracle):
    """The dim pass works on the frames the bright pass left undecided, whatever slots they sit in; the launches whose frame ->
    XCD mapping is fixed index frames through a list with the active slots first (k_active_perm / k_tile_perm).  Batches
    whose undecided frames are a few scattered slots, all even slots, none and all of them: records equal the oracle's and
    do not depend on where a frame sits in the batch."""
    from lfd_amd import _native, synth
    pb, pd, _ = params()
    bright, _, _ = synth.make_frame(0, with_catalog=False)               # found by the bright pass
    dim, _, _ = synth.make_frame(1, with_catalog=False)                  # needs the dim pass
    none, _, _ = synth.make_frame(7, with_catalog=False)
    want = {id(f): oracle.detect_frame(f.copy(), pb, pd) for f in (bright, dim, none)}
    assert want[id(bright)]["found"] == 1 and want[id(dim)]["found"] == 2
    n = 19
    with _native.Context(0, 1489, 2048, n) as ctx:
        for pattern in ([1, 2, 17], list(range(0, n, 2)), [], list(range(n))):
            batch = [bright] * n
            for j, i in enumerate(pattern):
                batch[i] = dim if j % 2 == 0 else none
            res = ctx.detect_batch(np.stack(batch), pb, pd)
            for i in range(n):
                assert same(res[i], want[id(batch[i])]), (pattern, i)
