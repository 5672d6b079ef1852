"""TEST / BASELINE INFRASTRUCTURE ONLY (like everything under oracle/): the reference's per-frame path written out as
the direct ``cv2.*`` call sequence, for machines where ``import cv2`` works.

Nothing here comes from the reference's files: the sequence is restated from its call sites --
detecttrails.py:119-131 (remove_stars -> cv2.flip -> bright -> dim), processfield.py:342-384 (bright),
:453-502 (dim), :236-261 (fit_minAreaRect).  Uses: tests/test_oracle_vs_cv2.py (pins the C oracle against the real
library where it exists) and bench.py's ``cpu_baseline.opencv`` leg ("the reference OpenCV CPU
path timed on the node's own host cores").  The product (lfd_amd/) never imports this module.
"""
import numpy as np


def load():
    """The cv2 module, or None where OpenCV is not installed (this build's container and GPU boxes so far)."""
    try:
        import cv2
    except Exception:  # noqa: BLE001 - ImportError, or a broken binary wheel
        return None
    return cv2


def contours(cv2, img, mode, method):
    return cv2.findContours(img.copy(), mode, method)[-2]             # OpenCV 3 returns (image, contours, hierarchy)


def fit_min_area_rect(cv2, img, contoursMode, contoursMethod, minAreaRectMinLen, lwTresh):
    edges = cv2.Canny(img, 0, 255)
    box = np.zeros_like(img)
    det = False
    for c in contours(cv2, edges, contoursMode, contoursMethod):
        (_, (w, h), _) = rect = cv2.minAreaRect(c)
        if min(w, h) > minAreaRectMinLen and max(w, h) / min(w, h) > lwTresh:
            det = True
            cv2.fillPoly(box, [np.int32(cv2.boxPoints(rect))], 255)
    return det, box


def run_pass(cv2, img, p, dim, check_theta):
    """One pass on ``img`` (masked IN PLACE like the reference); returns (found, rho, theta)."""
    if dim:
        img[img < p["minFlux"]] = 0
        img[img > 0] += p["addFlux"]
    else:
        img[img < 0] = 0
    equ = cv2.equalizeHist(cv2.convertScaleAbs(img))
    if dim:
        equ = cv2.erode(equ, p["erodeKernel"])
    equ = cv2.dilate(equ, p["dilateKernel"])
    det, box = fit_min_area_rect(cv2, equ, p["contoursMode"], p["contoursMethod"], p["minAreaRectMinLen"], p["lwTresh"])
    if not det:
        return False, 0.0, 0.0
    l1 = cv2.HoughLines(equ, p["houghMethod"], np.pi / 180, 1)
    l2 = cv2.HoughLines(box, p["houghMethod"], np.pi / 180, 1)
    if check_theta(l1, l2, p["nlinesInSet"], p["dro"], p["thetaTresh"], p["lineSetTresh"], False):
        return False, 0.0, 0.0
    return True, float(l1[0][0][0]), float(l1[0][0][1])


def detect_frame(cv2, img, params_bright, params_dim, cat, rs, remove_stars, check_theta):
    """remove_stars (the caller's function: numpy-only in the reference) -> flip -> bright -> dim; (found 0/1/2, rho, theta)."""
    if cat is not None:
        remove_stars(img, cat, rs)
    work = cv2.flip(img, 0)
    ok, rho, theta = run_pass(cv2, work, params_bright, False, check_theta)
    if ok:
        return 1, rho, theta
    ok, rho, theta = run_pass(cv2, work, params_dim, True, check_theta)
    return (2, rho, theta) if ok else (0, 0.0, 0.0)
