"""Per-image trail detection: the bright and the dim pass, on the GPU.

Host-side mirror of the reference module ``lfd/detecttrails/processfield.py``: same function
names, same arguments (the keys of ``params_bright`` / ``params_dim`` are splatted into them,
detecttrails.py:125,129), same return values ``(bool, dict-or-None)`` and the same in-place
side effects on ``img``.  Every image operation (convertScaleAbs, equalizeHist, erode/dilate,
Canny, contours -> minAreaRect -> fillPoly, HoughLines) runs in liblfdmi.so on the MI355X;
only the two scalar tails the reference itself evaluates in numpy, ``check_theta``
(processfield.py:36-150) and ``dictify_hough`` (:266-288), are evaluated here in numpy.
There is no CPU fallback for the image operations.
"""
import contextlib
import os
import threading

import numpy as np

from .. import _native

__all__ = ["process_field_bright", "process_field_dim", "check_theta", "dictify_hough",
           "fit_minAreaRect", "setup_debug", "pathBright", "pathDim"]

pathBright = None
pathDim = None

_ctx = None
_ctx_lock = threading.RLock()   # the process-wide context below is shared by every single-image entry point of this package


def setup_debug():
    """Point the debug image dumps at $DEBUG_PATH (reference: processfield.py:22-33)."""
    global pathBright, pathDim
    where = os.environ.get("DEBUG_PATH")
    if where is not None:
        pathBright = where
        pathDim = where


def _device_index():
    for key in ("LFD_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            return int(os.environ[key])
    return 0


def get_context(h, w, inflight=None):
    """Process-wide GPU context, grown (never shrunk) when a larger image or batch arrives.

    A ``lfdmi_ctx`` is not thread-safe (include/lfdmi.h) and growing replaces it, so callers that may run on several
    threads go through ``use_context``, which holds the module lock for the duration of their device calls; this
    function alone only serialises the creation."""
    global _ctx
    with _ctx_lock:
        want = int(inflight or os.environ.get("LFD_INFLIGHT", 4))
        if _ctx is None or _ctx.max_h < h or _ctx.max_w < w or _ctx.max_inflight < want:
            if _ctx is not None:
                want = max(want, _ctx.max_inflight)
                h, w = max(h, _ctx.max_h), max(w, _ctx.max_w)
                _ctx.close()
            _ctx = _native.Context(_device_index(), h, w, want)
        return _ctx


@contextlib.contextmanager
def use_context(h, w, inflight=None):
    """``with use_context(h, w) as ctx:`` -- the shared context, exclusively, until the block ends."""
    with _ctx_lock:
        yield get_context(h, w, inflight)


def check_theta(hough1, hough2, navg, dro, thetaTresh, lineSetTresh, debug):
    """Colinearity test between two sets of Hough lines; **True means "not colinear"**.

    Follows processfield.py:89-150: four zero-initialised float64 columns of length ``navg``
    are filled from the first ``navg`` lines while both sets still have a line ``i`` (a missing
    line leaves zeros behind); the sets are rejected when the mean rho differs by more than
    ``dro``, when either set's theta spread exceeds ``thetaTresh``, or when the mean absolute
    theta difference exceeds ``lineSetTresh``.  Returns ``None`` (falsy) when all tests pass.
    """
    cols = np.zeros((4, navg, 1))
    ro1, ro2, theta1, theta2 = cols
    for i in range(navg):
        try:
            ro1[i] = hough1[i][0][0]
            ro2[i] = hough2[i][0][0]
            theta1[i] = hough1[i][0][1]
            theta2[i] = hough2[i][0][1]
        except IndexError:
            continue
    if debug:
        print(f"rho test: |{np.average(ro1)} - {np.average(ro2)}| vs dro={dro}")
    if abs(np.average(ro1) - np.average(ro2)) > dro:
        return True
    spread1 = abs(theta1.max() - theta1.min())
    spread2 = abs(theta2.max() - theta2.min())
    if debug:
        print(f"theta spreads {spread1} {spread2} vs {thetaTresh}; "
              f"set difference {np.average(abs(theta1 - theta2))} vs {lineSetTresh}")
    if spread1 > thetaTresh or spread2 > thetaTresh:
        return True
    if np.average(abs(theta1 - theta2)) > lineSetTresh:
        return True
    return None


def dictify_hough(shape, houghVals):
    """(rho, theta) -> two points far along the line, as the reference does (processfield.py:266-288)."""
    rho, theta = houghVals
    reach = shape[0] + shape[1]
    c, s = np.cos(theta), np.sin(theta)
    x0, y0 = c * rho, s * rho
    return {"x1": int(x0 - reach * s), "y1": int(y0 + reach * c),
            "x2": int(x0 + reach * s), "y2": int(y0 - reach * c)}


def _as_native_image(img):
    """Array of a dtype the kernels read directly (u8 / f32 / f64); torch CUDA tensors pass through."""
    if _native._is_dev(img):
        _native._dtype_code(img)  # raises TypeError for unsupported dtypes
        return img.contiguous()
    if img.dtype in (np.uint8, np.float32, np.float64):
        return np.ascontiguousarray(img)
    if img.dtype.kind in "iub":
        return np.ascontiguousarray(img, dtype=np.float64)  # exact for |x| < 2**53
    if img.dtype == np.float16:
        return np.ascontiguousarray(img, dtype=np.float32)
    raise TypeError(f"unsupported image dtype {img.dtype}")


def fit_minAreaRect(img, contoursMode, contoursMethod, minAreaRectMinLen, lwTresh, debug):
    """Canny(0, 255) -> contours -> minimum-area rectangles -> (detection, box image)
    (reference: processfield.py:201-263)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    with use_context(*img.shape) as ctx:
        detection, box_img, _ = ctx.fit_min_area_rect(img, contoursMode, contoursMethod,
                                                      minAreaRectMinLen, lwTresh)
    return detection, box_img


def _lines_as_cv(lines, n_total):
    k = min(int(n_total), len(lines))
    if k == 0:
        return None
    return np.ascontiguousarray(lines[:k]).reshape(k, 1, 2)


def _finish(ctx, res, lines_equ, lines_box, shape, nlinesInSet, dro, thetaTresh, lineSetTresh,
            debug, tag, path):
    status = int(res["status"])
    if status not in (0, _native.ERR_NOLINES):
        raise _native.NativeError(status, "frame failed on the device")
    equhough = _lines_as_cv(lines_equ, res["n_lines_equ"]) if res["detection"] else None
    boxhough = _lines_as_cv(lines_box, res["n_lines_box"]) if res["detection"] else None
    if debug:
        _dump_debug(ctx, shape, tag, path, bool(res["detection"]), equhough, boxhough, nlinesInSet)
    if not res["detection"]:
        if debug:
            print(f"{tag}: no boxes found")
        return (False, None)
    if equhough is None or boxhough is None:
        # cv2.HoughLines returned None: the reference dies inside check_theta with this error
        raise TypeError("'NoneType' object is not subscriptable")
    if check_theta(equhough, boxhough, nlinesInSet, dro, thetaTresh, lineSetTresh, debug):
        return (False, None)
    return (True, dictify_hough(shape, equhough[0][0]))


def _dump_debug(ctx, shape, tag, path, detection, equhough, boxhough, nlines):
    """The reference's debug images (processfield.py:349-378 bright, :459-496 dim; list in
    docs/source/detecttrails/detparams.rst:39-54), fetched from the device buffers of slot 0:
    bright 1equ, 2dilate, 3contours and -- when a rectangle was found -- 4boxhough, 5equhough;
    dim 6equ, 7eroded, 8opened, 9contours, 10equhough, 11boxhough."""
    from . import debugio
    if path is None:
        raise TypeError("expected str, bytes or os.PathLike object, not NoneType")  # os.path.join(None, ..)
    h, w = shape
    equalized = ctx.get_stage(0, _native.STAGE_EQUALIZED, h, w)
    equ = ctx.get_stage(0, _native.STAGE_EQU, h, w)
    box = ctx.get_stage(0, _native.STAGE_BOX, h, w)
    if tag == "BRIGHT":
        debugio.write_png(os.path.join(path, "1equBRIGHT.png"), equalized, 3)
        debugio.write_png(os.path.join(path, "2dilateBRIGHT.png"), equ, 3)
        debugio.write_png(os.path.join(path, "3contoursBRIGHT.png"), box, 3)
        if detection:
            debugio.draw_lines(equhough, equ, nlines, "5equhoughBRIGHT", path)
            debugio.draw_lines(boxhough, box, nlines, "4boxhoughBRIGHT", path)
    else:
        debugio.write_png(os.path.join(path, "6equDIM.png"), equalized, 0)
        debugio.write_png(os.path.join(path, "7erodedDIM.png"), ctx.get_stage(0, _native.STAGE_ERODED, h, w), 0)
        debugio.write_png(os.path.join(path, "8openedDIM.png"), equ, 0)
        debugio.write_png(os.path.join(path, "9contoursDIM.png"), box, 0)
        if detection:
            debugio.draw_lines(equhough, equ, nlines, "10equhoughDIM", path, compression=4)
            debugio.draw_lines(boxhough, box, nlines, "11boxhoughDIM", path, compression=4, color=(0, 0, 255))


def process_field_bright(img, lwTresh, thetaTresh, dilateKernel, contoursMode, contoursMethod,
                         minAreaRectMinLen, houghMethod, nlinesInSet, lineSetTresh, dro, debug,
                         gaussKernel=0, gaussSigma=0.0):
    """Bright-trail pass (reference: processfield.py:291-388).

    ``img`` is clamped at zero IN PLACE (``img[img < 0] = 0``), converted to 8 bit, equalised,
    dilated, screened with minimum-area rectangles and, if any rectangle qualifies, fitted with
    Hough lines on the dilated image and on the rectangle image; colinear line sets give
    ``(True, {"x1":..,"y1":..,"x2":..,"y2":..})``, anything else ``(False, None)``.

    ``gaussKernel`` / ``gaussSigma`` are not reference parameters: an optional Gaussian smoothing of Canny's input
    (odd kernel size, 0 = off = the reference's behaviour; cv2.Canny has no smoothing stage).

    ``contoursMode``: ``RETR_EXTERNAL`` / ``RETR_LIST`` (default) / ``RETR_CCOMP`` / ``RETR_TREE`` (the last three yield
    the same contour set; only the hierarchy, which the reference discards, differs).  ``contoursMethod``:
    ``CHAIN_APPROX_NONE`` (default) and ``CHAIN_APPROX_SIMPLE`` (drops collinear interior points: same convex hull, same
    rectangles).  ``CHAIN_APPROX_TC89_L1`` / ``CHAIN_APPROX_TC89_KCOS`` are NOT supported: the Teh-Chin approximations
    drop curvature-dependent subsets of the border that need not keep the hull's vertices, so minAreaRect of the
    approximated contour can differ from that of the full one; they raise ``NativeError`` (``LFDMI_ERR_UNSUPPORTED``)
    instead of returning rectangles the reference would not produce.
    """
    img[img < 0] = 0
    dev_img = _as_native_image(img)
    params = dict(lwTresh=lwTresh, thetaTresh=thetaTresh, dilateKernel=dilateKernel,
                  contoursMode=contoursMode, contoursMethod=contoursMethod,
                  minAreaRectMinLen=minAreaRectMinLen, houghMethod=houghMethod,
                  nlinesInSet=nlinesInSet, lineSetTresh=lineSetTresh, dro=dro,
                  gaussKernel=gaussKernel, gaussSigma=gaussSigma)
    with use_context(*tuple(dev_img.shape)) as ctx:   # (the debug dumps of _finish read this call's device buffers)
        res, le, lb = ctx.process_bright(dev_img, params)
        return _finish(ctx, res, le, lb, tuple(dev_img.shape), nlinesInSet, dro, thetaTresh, lineSetTresh, debug,
                       "BRIGHT", pathBright)


def process_field_dim(img, minFlux, addFlux, lwTresh, thetaTresh, erodeKernel, dilateKernel,
                      contoursMode, contoursMethod, minAreaRectMinLen, houghMethod, nlinesInSet,
                      dro, lineSetTresh, debug, gaussKernel=0, gaussSigma=0.0):
    """Dim-trail pass (reference: processfield.py:391-506).

    IN PLACE: ``img[img < minFlux] = 0; img[img > 0] += addFlux``.  Then 8-bit conversion,
    equalisation, erosion, dilation and the same rectangle / Hough screening as the bright pass.
    Integer images raise like numpy does for ``uint8 += float``.
    """
    if _native._is_dev(img):
        if not img.is_floating_point():
            raise TypeError("dim pass needs a floating-point image")
        gpu_src = img.contiguous().clone()
    else:
        gpu_src = _as_native_image(img).copy() if img.dtype.kind == "f" else None
    img[img < minFlux] = 0
    img[img > 0] += addFlux  # raises numpy's casting error for integer images, as the reference
    if gpu_src is None:  # pragma: no cover - unreachable: the line above raised
        raise TypeError("dim pass needs a floating-point image")
    params = dict(minFlux=minFlux, addFlux=addFlux, lwTresh=lwTresh, thetaTresh=thetaTresh,
                  erodeKernel=erodeKernel, dilateKernel=dilateKernel, contoursMode=contoursMode,
                  contoursMethod=contoursMethod, minAreaRectMinLen=minAreaRectMinLen,
                  houghMethod=houghMethod, nlinesInSet=nlinesInSet, lineSetTresh=lineSetTresh, dro=dro,
                  gaussKernel=gaussKernel, gaussSigma=gaussSigma)
    # the device applies the same masking to the untouched copy (PREP_DIM)
    with use_context(*tuple(gpu_src.shape)) as ctx:
        res, le, lb = ctx.process_dim(gpu_src, params, after_bright=False)
        return _finish(ctx, res, le, lb, tuple(gpu_src.shape), nlinesInSet, dro, thetaTresh, lineSetTresh, debug,
                       "DIM", pathDim)
