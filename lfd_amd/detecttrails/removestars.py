"""Blot catalogued sources out of a frame before line detection, on the GPU.

Host-side mirror of ``lfd/detecttrails/removestars.py``: ``read_photoObj`` and ``remove_stars``
keep the reference's signatures and return layouts.  The per-object tests and the zero-fill
(removestars.py:212-231, including its axis-swapped ``img[x-d:x+d, y-d:y+d]`` indexing with
x = COLC and Python slice wrap/clip semantics) run in liblfdmi.so, one workgroup per object.
"""
import math

import numpy as np

from .. import _native
from . import fitslite, sdssfiles
from .processfield import use_context

__all__ = ["read_photoObj", "read_photoObj_arrays", "remove_stars", "remove_stars_arrays"]

_COLUMNS = ("OBJC_TYPE", "TYPE", "ROWC", "COLC", "PETROTH90", "PSFMAG", "NOBSERVE", "NDETECT")


def read_photoObj_arrays(path_to_photoOBJ):
    """The eight photoObj columns removestars.py:97-104 uses, as numpy arrays."""
    return fitslite.read_table(path_to_photoOBJ, _COLUMNS, ext=1)


def read_photoObj(path_to_photoOBJ):
    """Same return tuple as the reference (removestars.py:63-132): lists of per-filter dicts of
    ``math.ceil``-ed ROWC, COLC, PSFMAG, PETROTH90, then OBJC_TYPE, TYPE, NOBSERVE, NDETECT arrays."""
    t = read_photoObj_arrays(path_to_photoOBJ)

    def dicts(col):
        return [{f: math.ceil(v) for f, v in zip("ugriz", row)} for row in col]

    return (dicts(t["ROWC"]), dicts(t["COLC"]), dicts(t["PSFMAG"]), dicts(t["PETROTH90"]),
            t["OBJC_TYPE"], t["TYPE"], t["NOBSERVE"], t["NDETECT"])


def _check_finite(cat):
    # math.ceil raises on NaN / inf (removestars.py:113-130); keep that a frame error
    for key in ("ROWC", "COLC", "PSFMAG", "PETROTH90"):
        a = np.asarray(cat[key])
        if np.isnan(a).any():
            raise ValueError("cannot convert float NaN to integer")
        if np.isinf(a).any():
            raise OverflowError("cannot convert float infinity to integer")


def remove_stars_arrays(img, cat, _filter, defaultxy, filter_caps, maxxy, pixscale, magcount,
                        maxmagdiff, debug=False):
    """remove_stars on catalogue arrays (dict with ROWC/COLC/PSFMAG/PETROTH90 [n,5], NOBSERVE/NDETECT [n])."""
    if getattr(img, "ndim", 2) != 2:
        raise ValueError("remove_stars expects one 2-d frame")
    _check_finite(cat)
    n = len(cat["NOBSERVE"])
    if n == 0:
        return img
    packed = {"count": np.array([n], np.int32)}
    for key in ("ROWC", "COLC", "PSFMAG", "PETROTH90"):
        packed[key] = np.ascontiguousarray(cat[key], np.float32).reshape(1, n, 5)
    for key in ("NOBSERVE", "NDETECT"):
        packed[key] = np.ascontiguousarray(cat[key], np.int32).reshape(1, n)
    rs = _native.make_rs_params(_filter, defaultxy, filter_caps, maxxy, pixscale, magcount, maxmagdiff)
    if _native._is_dev(img):
        with use_context(*tuple(img.shape)) as ctx:
            ctx.remove_stars(img, packed, rs)
        return img
    if img.dtype != np.float32 or not img.flags.c_contiguous:
        # other dtypes / strided views: let the device blot a float32 plane of ones and zero
        # the same pixels here (ndarray.fill(0.0) is dtype-agnostic in the reference)
        work = np.ones(img.shape, np.float32)
        with use_context(*work.shape) as ctx:
            ctx.remove_stars(work, packed, rs)
        img[work == 0] = 0
        return img
    with use_context(*img.shape) as ctx:
        ctx.remove_stars(img, packed, rs)
    return img


def remove_stars(img, _run, _camcol, _filter, _field, defaultxy, filter_caps, maxxy, pixscale,
                 magcount, maxmagdiff, debug):
    """Reference signature (removestars.py:148-149): reads the field's photoObj file and blots
    the selected objects out of ``img`` in place; returns ``img``."""
    cat = read_photoObj_arrays(sdssfiles.filename("photoObj", run=_run, camcol=_camcol, field=_field))
    return remove_stars_arrays(img, cat, _filter, defaultxy, filter_caps, maxxy, pixscale, magcount,
                               maxmagdiff, debug)
