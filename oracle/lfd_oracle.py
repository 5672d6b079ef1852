"""ctypes view of the CPU oracle (oracle/liblfd_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``lfd_amd/`` imports this module.
Parity at the OpenCV boundary is unpinned (see lfd_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblfd_oracle.so")

U8, F32, F64 = 0, 1, 2
PREP_NONE, PREP_BRIGHT, PREP_DIM, PREP_BRIGHT_THEN_DIM = 0, 1, 2, 3
DILATE, ERODE = 0, 1
RETR_EXTERNAL, RETR_LIST, RETR_CCOMP, RETR_TREE = 0, 1, 2, 3
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 1, 2


def build(force=False):
    src = os.path.join(_HERE, "lfd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Params(C.Structure):
    _fields_ = [("lwTresh", C.c_double), ("thetaTresh", C.c_double), ("lineSetTresh", C.c_double),
                ("dro", C.c_double), ("minAreaRectMinLen", C.c_double), ("houghMethod", C.c_double),
                ("nlinesInSet", C.c_int), ("contoursMode", C.c_int), ("contoursMethod", C.c_int),
                ("dilate_kh", C.c_int), ("dilate_kw", C.c_int), ("dilateKernel", C.c_void_p),
                ("erode_kh", C.c_int), ("erode_kw", C.c_int), ("erodeKernel", C.c_void_p),
                ("minFlux", C.c_double), ("addFlux", C.c_double),
                ("gaussKernel", C.c_int), ("gaussSigma", C.c_double)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("rho", C.c_float), ("theta", C.c_float),
                ("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32), ("y2", C.c_int32),
                ("n_lines_equ", C.c_int32), ("n_lines_box", C.c_int32), ("detection", C.c_int32),
                ("rejected_by_theta", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RsParams(C.Structure):
    _fields_ = [("defaultxy", C.c_int), ("maxxy", C.c_int), ("magcount", C.c_int),
                ("pixscale", C.c_double), ("maxmagdiff", C.c_double), ("filter_cap", C.c_double),
                ("filter_index", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get("LFD_ORACLE_LIB")   # an instrumented build of the same source (tools/oracle_sanitize.sh)
        if not alt:
            build()
        _lib = C.CDLL(alt or _SO)
        _lib.lfo_free.argtypes = [C.c_void_p]
        _lib.lfo_free.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dt(a):
    if a.dtype == np.uint8:
        return U8
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.float64:
        return F64
    raise TypeError(a.dtype)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def prep(img, mode, flip=False, minFlux=0.0, addFlux=0.0):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().lfo_prep(_p(img), _dt(img), h, w, int(flip), mode, C.c_double(minFlux),
                        C.c_double(addFlux), _p(out))
    if rc:
        raise RuntimeError(f"lfo_prep rc={rc}")
    return out


def equalize_hist(src):
    src = _u8(src)
    out = np.empty_like(src)
    rc = lib().lfo_equalize_hist(_p(src), src.shape[0], src.shape[1], _p(out))
    if rc:
        raise RuntimeError(f"lfo_equalize_hist rc={rc}")
    return out


def equalize_lut(hist, total):
    hist = np.ascontiguousarray(hist, np.int32)
    lut = np.zeros(256, np.uint8)
    first, const = C.c_int(), C.c_int()
    rc = lib().lfo_equalize_lut(_p(hist), int(total), _p(lut), C.byref(first), C.byref(const))
    if rc:
        raise RuntimeError("empty histogram")
    return lut, first.value, const.value


def morph(src, kernel, op):
    src = _u8(src)
    kernel = _u8(kernel)
    out = np.empty_like(src)
    rc = lib().lfo_morph(_p(src), src.shape[0], src.shape[1], _p(kernel), kernel.shape[0],
                         kernel.shape[1], op, _p(out))
    if rc:
        raise RuntimeError(f"lfo_morph rc={rc}")
    return out


def dilate(src, kernel):
    return morph(src, kernel, DILATE)


def erode(src, kernel):
    return morph(src, kernel, ERODE)


def sobel_mag(src):
    src = _u8(src)
    h, w = src.shape
    dx = np.empty((h, w), np.int16)
    dy = np.empty((h, w), np.int16)
    mag = np.empty((h, w), np.int32)
    lib().lfo_sobel_mag(_p(src), h, w, _p(dx), _p(dy), _p(mag))
    return dx, dy, mag


def canny(src, low=0.0, high=255.0):
    src = _u8(src)
    out = np.empty_like(src)
    rc = lib().lfo_canny(_p(src), src.shape[0], src.shape[1], C.c_double(low), C.c_double(high),
                         _p(out))
    if rc:
        raise RuntimeError(f"lfo_canny rc={rc}")
    return out


def gaussian_blur(src, ksize, sigma=0.0):
    """The optional smoothing stage as this build defines it (see lfd_oracle.h; not a cv2 parity claim)."""
    src = _u8(src)
    out = np.empty_like(src)
    rc = lib().lfo_gaussian_blur(_p(src), src.shape[0], src.shape[1], int(ksize), C.c_double(sigma), _p(out))
    if rc:
        raise RuntimeError(f"lfo_gaussian_blur rc={rc}")
    return out


def gaussian_kernel(ksize, sigma=0.0):
    taps = np.zeros(32, np.float32)
    if lib().lfo_gaussian_kernel(int(ksize), C.c_double(sigma), _p(taps)):
        raise ValueError("ksize must be odd, 1..31")
    return taps[:ksize].copy()


def find_contours(img, mode=RETR_LIST):
    """Returns (list of (n,2) int32 arrays of (x,y) points, list of is_hole flags)."""
    img = _u8(img)
    pts = C.POINTER(C.c_int32)()
    offs = C.POINTER(C.c_int32)()
    holes = C.POINTER(C.c_int32)()
    n = C.c_int32()
    rc = lib().lfo_find_contours(_p(img), img.shape[0], img.shape[1], mode, C.byref(pts),
                                 C.byref(offs), C.byref(holes), C.byref(n))
    if rc:
        raise RuntimeError(f"lfo_find_contours rc={rc}")
    nc = n.value
    o = np.ctypeslib.as_array(offs, shape=(nc + 1,)).copy()
    total = int(o[-1])
    p = (np.ctypeslib.as_array(pts, shape=(max(total, 1) * 2,)).copy()[: total * 2]).reshape(-1, 2)
    hl = np.ctypeslib.as_array(holes, shape=(max(nc, 1),)).copy()[:nc]
    lib().lfo_free(pts)
    lib().lfo_free(offs)
    lib().lfo_free(holes)
    return [p[o[i]:o[i + 1]] for i in range(nc)], [int(v) for v in hl]


def convex_hull(points):
    points = np.ascontiguousarray(points, np.int32).reshape(-1, 2)
    n = len(points)
    hull = np.empty((max(n, 1), 2), np.int32)
    nh = C.c_int()
    lib().lfo_convex_hull(_p(points), n, _p(hull), C.byref(nh))
    return hull[: nh.value].copy()


def min_area_rect(points):
    """((cx, cy), (w, h), angle_deg) as float32 values."""
    points = np.ascontiguousarray(points, np.int32).reshape(-1, 2)
    rect = np.zeros(5, np.float32)
    rc = lib().lfo_min_area_rect(_p(points), len(points), _p(rect))
    if rc:
        raise RuntimeError("lfo_min_area_rect")
    return rect


def box_points(rect):
    rect = np.ascontiguousarray(rect, np.float32)
    box = np.zeros(8, np.float32)
    lib().lfo_box_points(_p(rect), _p(box))
    return box.reshape(4, 2)


def debug_trig(y, x):
    """(angle_deg, cos/2, sin/2) float32 of minAreaRect / boxPoints for atan2(y, x), with the host libm."""
    y = np.ascontiguousarray(y, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    out = [np.zeros(y.size, np.float32) for _ in range(3)]
    lib().lfo_debug_trig(int(y.size), _p(y), _p(x), *[_p(o) for o in out])
    return out


def fill_poly(img, pts, color=255):
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    lib().lfo_fill_poly(_p(img), img.shape[0], img.shape[1], _p(pts), len(pts), color)
    return img


def fit_min_area_rect(img, contoursMode=RETR_LIST, contoursMethod=CHAIN_APPROX_NONE,
                      minAreaRectMinLen=1, lwTresh=5):
    img = _u8(img)
    box = np.zeros_like(img)
    det, nb = C.c_int32(), C.c_int32()
    rc = lib().lfo_fit_min_area_rect(_p(img), img.shape[0], img.shape[1], contoursMode,
                                     contoursMethod, C.c_double(minAreaRectMinLen),
                                     C.c_double(lwTresh), _p(box), C.byref(det), C.byref(nb))
    if rc:
        raise RuntimeError(f"lfo_fit_min_area_rect rc={rc}")
    return bool(det.value), box, nb.value


def hough_dims(h, w, rho, theta=np.pi / 180):
    na, nr = C.c_int(), C.c_int()
    lib().lfo_hough_dims(h, w, C.c_double(rho), C.c_double(theta), C.byref(na), C.byref(nr))
    return na.value, nr.value


def hough_accum(img, rho, theta=np.pi / 180):
    img = _u8(img)
    na, nr = hough_dims(img.shape[0], img.shape[1], rho, theta)
    acc = np.zeros((na + 2, nr + 2), np.int32)
    a, r = C.c_int(), C.c_int()
    lib().lfo_hough_accum(_p(img), img.shape[0], img.shape[1], C.c_double(rho), C.c_double(theta),
                          _p(acc), C.byref(a), C.byref(r))
    return acc


def hough_lines(img, rho, theta=np.pi / 180, threshold=1, max_lines=1 << 20):
    """cv2.HoughLines layout: (n,1,2) float32 or None; second value = total number of lines."""
    img = _u8(img)
    na, nr = hough_dims(img.shape[0], img.shape[1], rho, theta)
    cap = min(max_lines, na * nr)
    lines = np.zeros((max(cap, 1), 2), np.float32)
    n = C.c_int32()
    rc = lib().lfo_hough_lines(_p(img), img.shape[0], img.shape[1], C.c_double(rho),
                               C.c_double(theta), int(threshold), cap, _p(lines), C.byref(n))
    if rc:
        raise RuntimeError("lfo_hough_lines")
    k = min(n.value, cap)
    if k == 0:
        return None, 0
    return lines[:k].reshape(k, 1, 2).copy(), n.value


def check_theta(h1, h2, navg, dro, thetaTresh, lineSetTresh):
    """True (reject) or None (accept), like processfield.py:36-150."""
    a = np.ascontiguousarray(np.asarray(h1, np.float32).reshape(-1, 2))
    b = np.ascontiguousarray(np.asarray(h2, np.float32).reshape(-1, 2))
    r = lib().lfo_check_theta(_p(a), len(a), _p(b), len(b), int(navg), C.c_double(dro),
                              C.c_double(thetaTresh), C.c_double(lineSetTresh))
    return True if r else None


def dictify_hough(shape, rho, theta):
    out = np.zeros(4, np.int32)
    lib().lfo_dictify_hough(int(shape[0]), int(shape[1]), C.c_float(rho), C.c_float(theta), _p(out))
    return {"x1": int(out[0]), "y1": int(out[1]), "x2": int(out[2]), "y2": int(out[3])}


def rs_params(filter="r", defaultxy=20, maxxy=60, pixscale=0.396, magcount=3, maxmagdiff=3,
              filter_caps=None, **_):
    caps = filter_caps or {'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5}
    return RsParams(int(defaultxy), int(maxxy), int(magcount), float(pixscale), float(maxmagdiff),
                    float(caps[filter]), "ugriz".index(filter))


def remove_stars(img, cat, rs):
    """img float32 (h,w) mutated in place; cat: dict of ROWC/COLC/PSFMAG/PETROTH90 [n,5] f32 and
    NOBSERVE/NDETECT [n] i32."""
    assert img.dtype == np.float32 and img.flags.c_contiguous
    rowc = np.ascontiguousarray(cat["ROWC"], np.float32)
    colc = np.ascontiguousarray(cat["COLC"], np.float32)
    mag = np.ascontiguousarray(cat["PSFMAG"], np.float32)
    pet = np.ascontiguousarray(cat["PETROTH90"], np.float32)
    nob = np.ascontiguousarray(cat["NOBSERVE"], np.int32)
    nde = np.ascontiguousarray(cat["NDETECT"], np.int32)
    rc = lib().lfo_remove_stars(_p(img), img.shape[0], img.shape[1], len(nob), _p(rowc), _p(colc),
                                _p(mag), _p(pet), _p(nob), _p(nde), C.byref(rs))
    if rc:
        raise RuntimeError("lfo_remove_stars")
    return img


class _Keep:
    """Params struct plus the numpy kernels it points at."""

    def __init__(self, struct, keep):
        self.struct = struct
        self.keep = keep


def make_params(d, dim=False):
    dk = _u8(d["dilateKernel"])
    keep = [dk]
    p = Params()
    p.lwTresh = d["lwTresh"]
    p.thetaTresh = d["thetaTresh"]
    p.lineSetTresh = d["lineSetTresh"]
    p.dro = d["dro"]
    p.minAreaRectMinLen = d["minAreaRectMinLen"]
    p.houghMethod = d["houghMethod"]
    p.nlinesInSet = d["nlinesInSet"]
    p.contoursMode = d["contoursMode"]
    p.contoursMethod = d["contoursMethod"]
    p.dilate_kh, p.dilate_kw = dk.shape
    p.dilateKernel = dk.ctypes.data
    if dim:
        ek = _u8(d["erodeKernel"])
        keep.append(ek)
        p.erode_kh, p.erode_kw = ek.shape
        p.erodeKernel = ek.ctypes.data
        p.minFlux = d["minFlux"]
        p.addFlux = d["addFlux"]
    p.gaussKernel = int(d.get("gaussKernel", 0) or 0)
    p.gaussSigma = float(d.get("gaussSigma", 0.0) or 0.0)
    return _Keep(p, keep)


def process_bright(img, params, flip=False, want_images=False):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    kp = make_params(params)
    res = Result()
    equ = np.zeros((h, w), np.uint8) if want_images else None
    box = np.zeros((h, w), np.uint8) if want_images else None
    lib().lfo_process_bright(_p(img), _dt(img), h, w, int(flip), C.byref(kp.struct), C.byref(res),
                             _p(equ) if want_images else None, _p(box) if want_images else None)
    return (res.as_dict(), equ, box) if want_images else res.as_dict()


def process_dim(img, params, flip=False, after_bright=False, want_images=False):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    kp = make_params(params, dim=True)
    res = Result()
    equ = np.zeros((h, w), np.uint8) if want_images else None
    box = np.zeros((h, w), np.uint8) if want_images else None
    lib().lfo_process_dim(_p(img), _dt(img), h, w, int(flip), int(after_bright), C.byref(kp.struct),
                          C.byref(res), _p(equ) if want_images else None,
                          _p(box) if want_images else None)
    return (res.as_dict(), equ, box) if want_images else res.as_dict()


def detect_frame(img, params_bright, params_dim, cat=None, rs=None):
    """removestars -> flip -> bright -> dim on a float32 frame (mutated)."""
    assert img.dtype == np.float32 and img.flags.c_contiguous
    h, w = img.shape
    kb = make_params(params_bright)
    kd = make_params(params_dim, dim=True)
    res = Result()
    if cat is not None:
        rowc = np.ascontiguousarray(cat["ROWC"], np.float32)
        colc = np.ascontiguousarray(cat["COLC"], np.float32)
        mag = np.ascontiguousarray(cat["PSFMAG"], np.float32)
        pet = np.ascontiguousarray(cat["PETROTH90"], np.float32)
        nob = np.ascontiguousarray(cat["NOBSERVE"], np.int32)
        nde = np.ascontiguousarray(cat["NDETECT"], np.int32)
        lib().lfo_detect_frame(_p(img), h, w, C.byref(kb.struct), C.byref(kd.struct), len(nob),
                               _p(rowc), _p(colc), _p(mag), _p(pet), _p(nob), _p(nde), C.byref(rs),
                               C.byref(res))
    else:
        lib().lfo_detect_frame(_p(img), h, w, C.byref(kb.struct), C.byref(kd.struct), 0, None, None,
                               None, None, None, None, None, C.byref(res))
    return res.as_dict()
