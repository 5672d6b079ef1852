"""The GPU path does not trace borders: it labels runs and takes, per contour, the convex hull
of (a) all pixels of an 8-connected component (outer border) and (b) the pixels of the
surrounding component that are 4-adjacent to a hole (hole border), see lfd_amd/csrc/k_ccl.h.
This test proves on the CPU that those hulls equal the hulls of the Suzuki-Abe borders that
cv2.findContours(RETR_LIST) returns (oracle: lfo_find_contours)."""
import numpy as np
from scipy import ndimage as ndi


def ccl_hulls(O, E):
    E = E != 0
    fl, nf = ndi.label(E, structure=np.ones((3, 3)))
    P = np.pad(~E, 1, constant_values=True)           # the virtual zero frame joins the outside
    bl, nb = ndi.label(P, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outside = bl[0, 0]
    flp = np.pad(fl, 1)
    hulls = []
    for a in range(1, nf + 1):
        ys, xs = np.nonzero(fl == a)
        hulls.append(tuple(map(tuple, O.convex_hull(np.stack([xs, ys], 1)))))
    for b in range(1, nb + 1):
        if b == outside:
            continue
        by, bx = np.nonzero(bl == b)
        k = np.lexsort((bx, by))[0]
        A = flp[by[k] - 1, bx[k]]                     # component of the pixel above the hole's first pixel
        assert A > 0
        pts = set()
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            yy, xx = by + dy, bx + dx
            m = flp[yy, xx] == A
            pts.update(zip((xx[m] - 1).tolist(), (yy[m] - 1).tolist()))
        hulls.append(tuple(map(tuple, O.convex_hull(np.array(sorted(pts), np.int32)))))
    return sorted(hulls)


def suzuki_hulls(O, E):
    cs, _ = O.find_contours(E)
    return sorted(tuple(map(tuple, O.convex_hull(c))) for c in cs)


def test_hulls_equal_on_random_images(oracle):
    rng = np.random.default_rng(5)
    total = 0
    for trial in range(40):
        h, w = int(rng.integers(4, 48)), int(rng.integers(4, 56))
        dens = float(rng.choice([0.1, 0.3, 0.5, 0.6, 0.8]))
        E = (rng.random((h, w)) < dens).astype(np.uint8) * 255
        if trial % 3 == 0:
            E = (ndi.gaussian_filter(rng.random((h, w)), 1.5) > 0.5).astype(np.uint8) * 255
        a, b = ccl_hulls(oracle, E), suzuki_hulls(oracle, E)
        assert a == b, (trial, h, w, dens)
        total += len(b)
    assert total > 1000


def test_hulls_equal_on_nested_structures(oracle):
    E = np.zeros((15, 15), np.uint8)
    E[1:14, 1:14] = 1; E[2:13, 2:13] = 0            # ring
    E[4:11, 4:11] = 1; E[5:10, 5:10] = 0            # ring inside the hole
    E[7, 7] = 1                                      # dot inside the inner hole
    assert ccl_hulls(oracle, E) == suzuki_hulls(oracle, E)
    assert len(suzuki_hulls(oracle, E)) == 5
