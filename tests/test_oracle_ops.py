"""Known answers and independent cross-checks (scipy.ndimage, brute force) for the CPU oracle.
The reference holds no tests for these operators: parity with OpenCV is
unpinned; these tests pin the oracle to SURVEY.md Appendix A by hand-derived cases."""
import numpy as np
import pytest
from scipy import ndimage as ndi


def test_convert_scale_abs_known_answers(oracle):
    x = np.array([[0.5, 1.5, 2.5, -1.5, 255.4999, 255.5, 1e9, -1e9, np.nan, np.inf, -0.49, 254.5]], np.float32)
    want = np.array([[0, 2, 2, 2, 255, 255, 255, 255, 0, 255, 0, 254]], np.uint8)
    assert np.array_equal(oracle.prep(x, oracle.PREP_NONE), want)
    assert np.array_equal(oracle.prep(x.astype(np.float64), oracle.PREP_NONE), want)
    u = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(oracle.prep(u, oracle.PREP_NONE), u)


def test_prep_masks_and_flip(oracle):
    x = np.array([[-3.0, 0.01, 0.02, 0.6], [1.4, 2.6, -0.7, 300.0]], np.float32)
    assert np.array_equal(oracle.prep(x, oracle.PREP_BRIGHT), [[0, 0, 0, 1], [1, 3, 0, 255]])
    # dim: < 0.02 -> 0 ; > 0 -> +0.5   (0.02 survives: float32(0.02) < float32(0.02) is False)
    assert np.array_equal(oracle.prep(x, oracle.PREP_DIM, minFlux=0.02, addFlux=0.5), [[0, 0, 1, 1], [2, 3, 0, 255]])
    assert np.array_equal(oracle.prep(x, oracle.PREP_BRIGHT, flip=True), [[1, 3, 0, 255], [0, 0, 0, 1]])
    with pytest.raises(RuntimeError):
        oracle.prep(np.zeros((2, 2), np.uint8), oracle.PREP_DIM, minFlux=0.02, addFlux=0.5)


def test_equalize_hist_known_answer(oracle):
    # values 0 x6, 1 x2, 3 x1, 7 x1: first bin 0 (6), scale = 255/4
    img = np.array([[0, 0, 0, 0, 0], [0, 1, 1, 3, 7]], np.uint8)
    out = oracle.equalize_hist(img)
    # lut[1] = round(2*63.75)=128 (127.5 half-even -> 128), lut[3] = round(3*63.75)=191, lut[7] = 255
    assert np.array_equal(out, [[0, 0, 0, 0, 0], [0, 128, 128, 191, 255]])
    const = np.full((3, 4), 9, np.uint8)
    assert np.array_equal(oracle.equalize_hist(const), const)


def test_equalize_lut_is_monotone(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        img = rng.integers(0, 256, (40, 50), dtype=np.uint8) >> rng.integers(0, 6)
        hist = np.bincount(img.ravel(), minlength=256)
        lut, first, const = oracle.equalize_lut(hist, img.size)
        assert np.all(np.diff(lut[first:].astype(int)) >= 0)


@pytest.mark.parametrize("k", [(3, 3), (4, 4), (9, 9), (2, 5), (1, 7)])
def test_morphology_matches_scipy(oracle, k):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    kh, kw = k
    # scipy origin: window centre floor(k/2) by default equals OpenCV's anchor k//2 for odd k;
    # for even k scipy's default centre is k//2 too (origin 0 puts it at index k//2)
    d = ndi.maximum_filter(img, size=k, mode="constant", cval=0)
    e = ndi.minimum_filter(img, size=k, mode="constant", cval=255)
    assert np.array_equal(oracle.dilate(img, np.ones(k, np.uint8)), d)
    assert np.array_equal(oracle.erode(img, np.ones(k, np.uint8)), e)


def test_morphology_arbitrary_kernel(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (20, 31), dtype=np.uint8)
    ker = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    d = ndi.maximum_filter(img, footprint=ker, mode="constant", cval=0)
    assert np.array_equal(oracle.dilate(img, ker), d)
    assert np.array_equal(oracle.erode(img, ker), ndi.minimum_filter(img, footprint=ker, mode="constant", cval=255))


def test_sobel_matches_scipy(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (25, 33), dtype=np.uint8)
    dx, dy, mag = oracle.sobel_mag(img)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    rx = ndi.correlate(img.astype(np.int32), kx, mode="nearest")
    ry = ndi.correlate(img.astype(np.int32), kx.T, mode="nearest")
    assert np.array_equal(dx, rx) and np.array_equal(dy, ry)
    assert np.array_equal(mag, np.abs(rx) + np.abs(ry))


def test_canny_step_edge_known_answer(oracle):
    # vertical step 0 | 200: dx = 800 at the two columns next to the step, NMS keeps the LEFT one
    # (m > left neighbour, m >= right neighbour), magnitude 800 > 255 -> strong
    img = np.zeros((8, 10), np.uint8)
    img[:, 5:] = 200
    out = oracle.canny(img)
    want = np.zeros_like(img)
    want[:, 4] = 255
    assert np.array_equal(out, want)


def test_canny_hysteresis_is_component_selection(oracle):
    """Canny == NMS candidates whose 8-connected component (among candidates) holds a strong pixel."""
    rng = np.random.default_rng(4)
    blobs = (ndi.gaussian_filter(rng.random((60, 80)), 2.0) > 0.5) * 220.0
    img = (ndi.gaussian_filter(blobs, 0.8) + rng.integers(0, 12, (60, 80))).clip(0, 255).astype(np.uint8)
    full = oracle.canny(img, 0, 255)
    cand = oracle.canny(img, 0, 0) != 0      # high = 0: every candidate is strong
    lab, n = ndi.label(cand, structure=np.ones((3, 3)))
    dx, dy, mag = oracle.sobel_mag(img)
    strong = cand & (mag > 255)
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    assert np.array_equal(full != 0, keep[lab])
    assert (full != 0).any()


def test_find_contours_known_answers(oracle):
    img = np.zeros((7, 9), np.uint8)
    img[1:6, 1:8] = 1
    img[3, 3:6] = 0                       # a 1x3 hole
    cs, holes = oracle.find_contours(img)
    assert holes == [0, 1]
    outer = {tuple(p) for p in cs[0]}
    assert outer == {(x, y) for x in range(1, 8) for y in range(1, 6) if x in (1, 7) or y in (1, 5)}
    hole = {tuple(p) for p in cs[1]}
    # hole border = the 1-pixels 4-adjacent to the hole (Suzuki-Abe, 8-connected 1-components)
    assert hole == {(2, 3), (6, 3), (3, 2), (4, 2), (5, 2), (3, 4), (4, 4), (5, 4)}
    single = np.zeros((3, 3), np.uint8)
    single[1, 1] = 255
    cs, holes = oracle.find_contours(single)
    assert len(cs) == 1 and cs[0].tolist() == [[1, 1]]
    border = np.ones((2, 3), np.uint8)   # touches every image edge: still one outer contour
    cs, holes = oracle.find_contours(border)
    assert len(cs) == 1 and len({tuple(p) for p in cs[0]}) == 6


def test_find_contours_external_mode(oracle):
    img = np.zeros((9, 9), np.uint8)
    img[1:8, 1:8] = 1
    img[2:7, 2:7] = 0
    img[4, 4] = 1                          # a component nested inside the ring's hole
    cs_list, _ = oracle.find_contours(img, oracle.RETR_LIST)
    cs_ext, _ = oracle.find_contours(img, oracle.RETR_EXTERNAL)
    assert len(cs_list) == 3 and len(cs_ext) == 1


def test_convex_hull_order_and_strictness(oracle):
    pts = np.array([[0, 0], [4, 0], [4, 3], [0, 3], [2, 0], [2, 3], [1, 1], [0, 1]], np.int32)
    hull = oracle.convex_hull(pts)
    # start (min x, min y), walk the large-y side first
    assert hull.tolist() == [[0, 0], [0, 3], [4, 3], [4, 0]]
    assert oracle.convex_hull(np.array([[3, 3], [1, 1], [2, 2]])).tolist() == [[1, 1], [3, 3]]
    assert oracle.convex_hull(np.array([[5, 5], [5, 5]])).tolist() == [[5, 5]]


def test_min_area_rect_known_answers(oracle):
    t = np.zeros((50, 60), np.uint8)
    t[10:20, 5:45] = 255
    cs, _ = oracle.find_contours(t)
    r = oracle.min_area_rect(cs[0])
    assert r[:2].tolist() == [24.5, 14.5] and sorted(r[2:4].tolist()) == [9.0, 39.0]
    box = oracle.box_points(r)
    assert {tuple(p) for p in box.tolist()} == {(5.0, 10.0), (44.0, 10.0), (44.0, 19.0), (5.0, 19.0)}
    # 45-degree square: vertices (5,0) (10,5) (5,10) (0,5): sides 5*sqrt(2)
    sq = np.array([[5, 0], [10, 5], [5, 10], [0, 5], [5, 5]], np.int32)
    r = oracle.min_area_rect(sq)
    assert np.allclose(r[2:4], 5 * np.sqrt(2), rtol=1e-6) and np.allclose(r[:2], [5, 5], atol=1e-5)
    # two points: width = distance, height 0
    r = oracle.min_area_rect(np.array([[0, 0], [3, 4]], np.int32))
    assert r[2] == 5.0 and r[3] == 0.0


def test_min_area_rect_is_minimal_over_hull_edges(oracle):
    rng = np.random.default_rng(5)
    for _ in range(30):
        pts = rng.integers(0, 200, (rng.integers(3, 40), 2)).astype(np.int32)
        hull = oracle.convex_hull(pts).astype(np.float64)
        if len(hull) < 3:
            continue
        r = oracle.min_area_rect(pts)
        best = np.inf
        for i in range(len(hull)):
            e = hull[(i + 1) % len(hull)] - hull[i]
            e /= np.hypot(*e)
            nrm = np.array([-e[1], e[0]])
            a = (np.ptp(hull @ e)) * (np.ptp(hull @ nrm))
            best = min(best, a)
        assert abs(float(r[2]) * float(r[3]) - best) <= 1e-3 * max(1.0, best)


def test_fill_poly_known_answers(oracle):
    img = np.zeros((12, 14), np.uint8)
    oracle.fill_poly(img, [[2, 3], [9, 3], [9, 8], [2, 8]])
    want = np.zeros_like(img)
    want[3:9, 2:10] = 255
    assert np.array_equal(img, want)
    # clipped: polygon partly outside on all sides
    img = np.zeros((6, 6), np.uint8)
    oracle.fill_poly(img, [[-3, -2], [8, -2], [8, 9], [-3, 9]])
    assert (img == 255).all()
    # diamond: every pixel strictly inside is filled, nothing outside the bounding box
    img = np.zeros((21, 21), np.uint8)
    oracle.fill_poly(img, [[10, 2], [18, 10], [10, 18], [2, 10]])
    yy, xx = np.mgrid[0:21, 0:21]
    inside = np.abs(xx - 10) + np.abs(yy - 10) <= 8
    assert np.array_equal(img != 0, inside)


def test_hough_analytic_bins(oracle):
    h, w = 300, 400
    na, nr = oracle.hough_dims(h, w, 20)
    assert (na, nr) == (180, 70)
    assert oracle.hough_dims(1489, 2048, 20) == (180, 354)
    assert oracle.hough_dims(4096, 4096, 20) == (180, 819)
    img = np.zeros((h, w), np.uint8)
    img[:, 150] = 255                          # vertical line x = 150: theta = 0, rho = 150
    lines, n = oracle.hough_lines(img, 20)
    # r = round(150/20) = 8 (7.5 rounds half-even to 8); rho = (8 + (nr-1)//2 - (nr-1)/2) * 20
    r = 8
    assert lines[0, 0, 1] == 0.0
    assert lines[0, 0, 0] == np.float32((r + (nr - 1) // 2 - (nr - 1) * 0.5) * 20)
    acc = oracle.hough_accum(img, 20)
    assert acc[1, r + (nr - 1) // 2 + 1] == h                  # all 300 pixels vote there at theta 0
    assert (acc[1:-1].sum(axis=1) == h).all()                   # every angle receives one vote per pixel
    assert acc[0].sum() == 0 and acc[-1].sum() == 0 and acc[:, 0].sum() == 0 and acc[:, -1].sum() == 0


def test_hough_sort_order(oracle):
    rng = np.random.default_rng(6)
    img = (rng.random((80, 90)) < 0.02).astype(np.uint8)
    lines, n = oracle.hough_lines(img, 5, threshold=1)
    acc = oracle.hough_accum(img, 5)
    na, nr = oracle.hough_dims(80, 90, 5)
    votes = []
    for rho, theta in lines[:, 0]:
        nidx = int(round(theta / np.float32(np.pi / 180)))
        r = int(round(rho / 5 + (nr - 1) * 0.5))
        votes.append(acc[nidx + 1, r + 1])
    assert all(votes[i] >= votes[i + 1] for i in range(len(votes) - 1)) and min(votes) > 1 and n == len(lines)


def test_remove_stars_python_slice_semantics(oracle):
    rng = np.random.default_rng(7)
    h, w = 60, 90
    n = 40
    cat = {"ROWC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32), "COLC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32),
           "PSFMAG": rng.uniform(14, 24, (n, 5)).astype(np.float32), "PETROTH90": rng.uniform(-2, 30, (n, 5)).astype(np.float32),
           "NOBSERVE": rng.integers(1, 3, n).astype(np.int32), "NDETECT": rng.integers(1, 3, n).astype(np.int32)}
    cat["PSFMAG"][::7, 2] = -9999
    img = np.ones((h, w), np.float32)
    got = oracle.remove_stars(img.copy(), cat, oracle.rs_params("r", defaultxy=4, maxxy=12))
    # literal restatement of removestars.py:212-231 with numpy slicing
    import math
    ref = img.copy()
    for i in range(n):
        x = int(math.ceil(cat["COLC"][i][2])); y = int(math.ceil(cat["ROWC"][i][2]))
        mags = [math.ceil(v) for v in cat["PSFMAG"][i]]
        if mags[2] < 22.2:
            diffs = np.absolute([mags[j] - mags[k] for j in range(5) for k in range(j + 1, 5)])
            if 3 >= np.count_nonzero(diffs > 3):
                dxy = 4
                pet = math.ceil(cat["PETROTH90"][i][2])
                if pet > 0:
                    dxy = int(pet / 0.396) + 10
                if dxy > 12:
                    dxy = 4
                if cat["NOBSERVE"][i] == cat["NDETECT"][i]:
                    ref[x - dxy:x + dxy, y - dxy:y + dxy].fill(0.0)
    assert np.array_equal(got, ref) and (ref == 0).any()
