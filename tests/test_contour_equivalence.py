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


def external_by_topology(E):
    """RETR_EXTERNAL without border following (k_rect.h: k_filter_external): 8-connected components whose raster-first
    pixel has the frame-connected 4-background as its left neighbour (or sits in column 0) -> their first pixels."""
    fg, n = ndi.label(E != 0, structure=np.ones((3, 3), int))
    pad = np.pad(E != 0, 1)
    bg, _ = ndi.label(~pad)                      # 4-connectivity, the 1-px zero frame included
    outside = bg[0, 0]
    ext = set()
    for c in range(1, n + 1):
        ys, xs = np.nonzero(fg == c)
        y0 = int(ys.min())
        x0 = int(xs[ys == y0].min())
        if bg[y0 + 1, x0] == outside:            # padded coordinates of (y0, x0 - 1)
            ext.add((x0, y0))
    return ext


def test_retr_external_is_the_outside_background_rule(oracle):
    """Suzuki-Abe's RETR_EXTERNAL test ("the last border pixel met on this row is positive", lfo_find_contours) against
    the topological rule the GPU uses: random noise, nested rectangles / rings, thinned blobs, Canny edge maps."""
    rng = np.random.default_rng(1)
    total = 0
    for trial in range(600):
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        kind = trial % 4
        if kind == 0:
            img = (rng.random((h, w)) < rng.uniform(0.2, 0.8)).astype(np.uint8)
        elif kind == 1:
            img = np.zeros((h, w), np.uint8)
            for _ in range(int(rng.integers(1, 6))):
                y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                y1, x1 = int(rng.integers(y0, h)) + 1, int(rng.integers(x0, w)) + 1
                img[y0:y1, x0:x1] = 1
                if y1 - y0 > 2 and x1 - x0 > 2:
                    img[y0 + 1:y1 - 1, x0 + 1:x1 - 1] = 0
        elif kind == 2:
            img = ndi.binary_dilation(rng.random((h, w)) < 0.5).astype(np.uint8) & (rng.random((h, w)) < 0.9)
        else:
            img = oracle.canny((ndi.gaussian_filter(rng.random((h, w)), 1.5) * 255).astype(np.uint8), 0, 255)
        img = np.ascontiguousarray((img != 0) * 255, np.uint8)
        cs, holes = oracle.find_contours(img, oracle.RETR_EXTERNAL)
        assert not any(holes)
        assert {tuple(int(v) for v in c[0]) for c in cs} == external_by_topology(img), (trial, kind)
        total += len(cs)
    assert total > 1500
