"""debug=True: the reference's stage images (processfield.py:349-378, :459-496; docs/source/detecttrails/detparams.rst:39-54)
written from the device buffers -- file names, and pixel content against the oracle's stage images."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _overlay_ok(png, base, line, color_rgb):
    """every pixel is either the grey base image or the line colour; coloured pixels lie on the top Hough line"""
    from lfd_amd.detecttrails import dictify_hough
    assert png.shape == base.shape + (3,)
    col = (png == np.array(color_rgb, np.uint8)).all(axis=2) & ~((png[..., 0] == png[..., 1]) & (png[..., 1] == png[..., 2]))
    grey = np.repeat(base[:, :, None], 3, axis=2)
    assert np.array_equal(png[~col], grey[~col])
    assert col.sum() > 100
    rho, theta = float(line[0]), float(line[1])
    ys, xs = np.nonzero(col)
    d = np.abs(xs * np.cos(theta) + ys * np.sin(theta) - rho)
    assert (d < 3.0).mean() > 0.3                                  # (up to three lines are drawn; the first one is among them)
    assert dictify_hough(base.shape, (np.float32(rho), np.float32(theta)))


def test_debug_dumps_bright_and_dim(tmp_path, monkeypatch, oracle):
    from lfd_amd import synth
    from lfd_amd.detecttrails import default_params, process_field_bright, process_field_dim, processfield
    from lfd_amd.detecttrails import debugio
    monkeypatch.setenv("DEBUG_PATH", str(tmp_path))
    processfield.setup_debug()
    pb, pd, _ = default_params()
    pb["debug"] = pd["debug"] = True
    img = synth.make_frame(0, with_catalog=False)[0][::-1].copy()          # a bright streak
    det, res = process_field_bright(img.copy(), **pb)
    want, equ, box = oracle.process_bright(img.copy(), pb, want_images=True)
    assert det == (want["found"] == 1) and det
    gray = oracle.prep(img, oracle.PREP_BRIGHT)
    names = sorted(os.listdir(tmp_path))
    assert names == ["1equBRIGHT.png", "2dilateBRIGHT.png", "3contoursBRIGHT.png", "4boxhoughBRIGHT.png", "5equhoughBRIGHT.png"]
    assert np.array_equal(debugio.read_png(tmp_path / "1equBRIGHT.png"), oracle.equalize_hist(gray))
    assert np.array_equal(debugio.read_png(tmp_path / "2dilateBRIGHT.png"), equ)
    assert np.array_equal(debugio.read_png(tmp_path / "3contoursBRIGHT.png"), box)
    _overlay_ok(debugio.read_png(tmp_path / "5equhoughBRIGHT.png"), equ, (want["rho"], want["theta"]), (0, 0, 255))
    lb, _ = oracle.hough_lines(box, 20, max_lines=3)
    _overlay_ok(debugio.read_png(tmp_path / "4boxhoughBRIGHT.png"), box, lb[0][0], (0, 0, 255))

    for f in names:
        os.remove(tmp_path / f)
    img = synth.make_frame(1, with_catalog=False)[0][::-1].copy()          # a dim streak
    det, res = process_field_dim(img.copy(), **pd)
    want, equ, box = oracle.process_dim(img.copy(), pd, want_images=True)
    assert det == (want["found"] == 2) and det
    gray = oracle.prep(img, oracle.PREP_DIM, minFlux=pd["minFlux"], addFlux=pd["addFlux"])
    eq = oracle.equalize_hist(gray)
    assert sorted(os.listdir(tmp_path)) == ["10equhoughDIM.png", "11boxhoughDIM.png", "6equDIM.png", "7erodedDIM.png",
                                            "8openedDIM.png", "9contoursDIM.png"]
    assert np.array_equal(debugio.read_png(tmp_path / "6equDIM.png"), eq)
    assert np.array_equal(debugio.read_png(tmp_path / "7erodedDIM.png"), oracle.erode(eq, pd["erodeKernel"]))
    assert np.array_equal(debugio.read_png(tmp_path / "8openedDIM.png"), equ)
    assert np.array_equal(debugio.read_png(tmp_path / "9contoursDIM.png"), box)
    _overlay_ok(debugio.read_png(tmp_path / "10equhoughDIM.png"), equ, (want["rho"], want["theta"]), (0, 0, 255))
    lb, _ = oracle.hough_lines(box, 20, max_lines=3)
    _overlay_ok(debugio.read_png(tmp_path / "11boxhoughDIM.png"), box, lb[0][0], (255, 0, 0))

    # no rectangle: the three (four) stage images only
    for f in os.listdir(tmp_path):
        os.remove(tmp_path / f)
    det, res = process_field_bright(np.zeros((64, 128), np.float32), **pb)
    assert det is False and sorted(os.listdir(tmp_path)) == ["1equBRIGHT.png", "2dilateBRIGHT.png", "3contoursBRIGHT.png"]


def test_retr_external(gpu_ctx, oracle):
    """contoursMode = cv2.RETR_EXTERNAL (processfield.py:226-230 passes the knob through): outer borders of components no
    other component encloses -- nested rings, components inside holes, frame-touching components, random texture."""
    from lfd_amd import _native
    rng = np.random.default_rng(3)
    imgs = []
    a = np.zeros((96, 160), np.uint8)
    a[8:60, 10:120] = 255; a[12:56, 14:116] = 0; a[20:24, 30:100] = 255          # a bar inside a ring
    a[70:73, 5:150] = 255                                                        # a bar outside
    a[30:50, 40:90] = 255; a[33:47, 43:87] = 0; a[38:41, 50:80] = 255            # ring in ring, bar inside
    a[0:3, 130:160] = 255                                                        # touches the frame
    imgs.append(a)
    imgs.append((rng.random((80, 144)) < 0.4).astype(np.uint8) * 255)
    for img in imgs:
        for lw in (5, 0.5):
            det_l, box_l, nb_l = gpu_ctx.fit_min_area_rect(img, _native_mode("LIST"), 1, 1, lw)
            det, box, nb = gpu_ctx.fit_min_area_rect(img, _native_mode("EXTERNAL"), 1, 1, lw)
            det_o, box_o, nb_o = oracle.fit_min_area_rect(img, oracle.RETR_EXTERNAL, 1, 1, lw)
            assert (det, nb) == (det_o, nb_o) and np.array_equal(box, box_o)
            assert nb <= nb_l
    from lfd_amd import synth
    from lfd_amd.detecttrails import default_params
    pb, pd, _ = default_params()
    img = synth.make_frame(0, with_catalog=False)[0][::-1].copy()
    p = dict(pb, contoursMode=0)
    res, _, _ = gpu_ctx.process_bright(img, p)
    want = oracle.process_bright(img, p)
    assert all(res[k].item() == v for k, v in want.items())


def _native_mode(name):
    return {"EXTERNAL": 0, "LIST": 1}[name]
