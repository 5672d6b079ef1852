"""Per-operator parity: HIP kernels (through the C-ABI) vs the CPU oracle, bit-exact.
Seeded inputs at sizes the oracle finishes in seconds; edge cases the domain has (ragged
shapes, one-pixel images, constant / empty images, NaN / inf pixels, arbitrary structuring
elements, dense random edge maps with nested contours)."""
import numpy as np
import pytest
from scipy import ndimage as ndi

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (3, 5), (64, 64), (65, 129), (131, 333), (257, 640)]


def rand_u8(rng, shape, sparse=False):
    if sparse:
        return ((rng.random(shape) < 0.05) * rng.integers(1, 256, shape)).astype(np.uint8)
    return rng.integers(0, 256, shape, dtype=np.uint8)


def blobs(rng, shape, amp=230):
    b = (ndi.gaussian_filter(rng.random(shape), 2.0) > 0.5) * float(amp)
    return (ndi.gaussian_filter(b, 0.8) + rng.integers(0, 10, shape)).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8])
def test_prep(gpu_ctx, oracle, shape, dtype):
    rng = np.random.default_rng(11)
    if dtype == np.uint8:
        img = rand_u8(rng, shape)
        modes = [(oracle.PREP_NONE, False), (oracle.PREP_BRIGHT, True)]
    else:
        img = (rng.normal(0, 2, shape) * rng.choice([0.01, 1, 100], shape)).astype(dtype)
        flat = img.reshape(-1)
        flat[:: 7] = np.round(flat[:: 7]) + 0.5          # ties: round-half-even matters
        if flat.size > 4:
            flat[1], flat[2], flat[3] = np.nan, np.inf, -np.inf
        modes = [(m, f) for m in (0, 1, 2, 3) for f in (False, True)]
    for mode, flip in modes:
        want = oracle.prep(img, mode, flip=flip, minFlux=0.02, addFlux=0.5)
        got, hist = gpu_ctx.prep_u8(img, mode, flip=flip, minFlux=0.02, addFlux=0.5, want_hist=True)
        assert np.array_equal(got, want), (mode, flip)
        assert np.array_equal(hist, np.bincount(want.ravel(), minlength=256))


def test_prep_dim_on_uint8_is_a_dtype_error(gpu_ctx):
    from lfd_amd import _native
    with pytest.raises(_native.NativeError) as e:
        gpu_ctx.prep_u8(np.zeros((4, 4), np.uint8), _native.PREP_DIM, minFlux=0.02, addFlux=0.5)
    assert e.value.code == _native.ERR_DTYPE


@pytest.mark.parametrize("shape", SHAPES)
def test_equalize_hist(gpu_ctx, oracle, shape):
    rng = np.random.default_rng(12)
    for img in (rand_u8(rng, shape), rand_u8(rng, shape, sparse=True), np.full(shape, 7, np.uint8),
                np.zeros(shape, np.uint8), (rand_u8(rng, shape) >> 6) + 100):
        assert np.array_equal(gpu_ctx.equalize_hist(img), oracle.equalize_hist(img))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("k", [(1, 1), (3, 3), (4, 4), (9, 9), (2, 7), (31, 31)])
def test_rect_morphology(gpu_ctx, oracle, shape, k):
    rng = np.random.default_rng(13)
    img = rand_u8(rng, shape)
    ker = np.ones(k, np.uint8)
    assert np.array_equal(gpu_ctx.dilate(img, ker), oracle.dilate(img, ker))
    assert np.array_equal(gpu_ctx.erode(img, ker), oracle.erode(img, ker))


def test_arbitrary_structuring_elements(gpu_ctx, oracle):
    rng = np.random.default_rng(14)
    img = rand_u8(rng, (90, 150))
    for ker in (np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8), (rng.random((5, 8)) < 0.5).astype(np.uint8),
                np.eye(7, dtype=np.uint8)):
        assert np.array_equal(gpu_ctx.dilate(img, ker), oracle.dilate(img, ker))
        assert np.array_equal(gpu_ctx.erode(img, ker), oracle.erode(img, ker))
    from lfd_amd import _native
    with pytest.raises(_native.NativeError):
        gpu_ctx.dilate(img, np.ones((40, 3), np.uint8))


def test_batched_morphology_matches_single(gpu_ctx, oracle):
    rng = np.random.default_rng(15)
    batch = rng.integers(0, 256, (11, 70, 100), dtype=np.uint8)      # more images than slots: chunking
    got = gpu_ctx.dilate(batch, np.ones((4, 4), np.uint8))
    for i in range(len(batch)):
        assert np.array_equal(got[i], oracle.dilate(batch[i], np.ones((4, 4), np.uint8)))


@pytest.mark.parametrize("shape", SHAPES)
def test_canny(gpu_ctx, oracle, shape):
    rng = np.random.default_rng(16)
    for img in (rand_u8(rng, shape), blobs(rng, shape) if min(shape) > 8 else rand_u8(rng, shape),
                np.zeros(shape, np.uint8), np.full(shape, 200, np.uint8)):
        for lo, hi in ((0, 255), (50, 150), (300, 100)):
            assert np.array_equal(gpu_ctx.canny(img, lo, hi), oracle.canny(img, lo, hi)), (shape, lo, hi)


@pytest.mark.parametrize("shape", SHAPES[2:])
def test_fit_min_area_rect(gpu_ctx, oracle, shape):
    rng = np.random.default_rng(17)
    imgs = [blobs(rng, shape), rand_u8(rng, shape), rand_u8(rng, shape, sparse=True)]
    bar = np.zeros(shape, np.uint8)
    bar[shape[0] // 3: shape[0] // 3 + 4, 2: shape[1] - 2] = 220          # an elongated bar: accepted
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    diag = ((np.abs((xx - 5) * 2 - (yy - 3) * 3) < 9) * 210).astype(np.uint8)   # diagonal streak crossing the borders
    imgs += [bar, diag]
    for img in imgs:
        for min_len, lw in ((1, 5), (0, 2), (3, 1.5)):
            det_o, box_o, nb_o = oracle.fit_min_area_rect(img, 1, 1, min_len, lw)
            det_g, box_g, nb_g = gpu_ctx.fit_min_area_rect(img, 1, 1, min_len, lw)
            assert (det_g, nb_g) == (det_o, nb_o)
            assert np.array_equal(box_g, box_o)
    det, box, nb = gpu_ctx.fit_min_area_rect(bar)
    assert det and nb >= 1


def test_fit_min_area_rect_dense_nested_contours(gpu_ctx, oracle):
    """Checkerboards, rings in rings, single pixels: many tiny keys and holes."""
    rng = np.random.default_rng(18)
    yy, xx = np.mgrid[0:96, 0:160]
    imgs = [(((yy // 2 + xx // 2) % 2) * 255).astype(np.uint8), (((yy + xx) % 2) * 255).astype(np.uint8)]
    rings = np.zeros((96, 160), np.uint8)
    for r in range(4, 44, 6):
        rings[48 - r:48 + r, 80 - r:80 + r] = 255 if (r // 6) % 2 == 0 else 0
    imgs.append(rings)
    imgs.append(((rng.random((96, 160)) < 0.5) * 255).astype(np.uint8))
    for img in imgs:
        det_o, box_o, nb_o = oracle.fit_min_area_rect(img, 1, 1, 0, 1.01)
        det_g, box_g, nb_g = gpu_ctx.fit_min_area_rect(img, 1, 1, 0, 1.01)
        assert (det_g, nb_g) == (det_o, nb_o) and np.array_equal(box_g, box_o)


def test_fit_min_area_rect_key_heights_around_the_launch_boundaries(gpu_ctx, oracle):
    """Keys go to one of four launches by their height in rows (k_rect.h: a lane per key up to 16 rows, a wave with a 64-row
    footprint up to 64, a wave with a 320-row footprint, a wave with the full-height footprint): slanted bars whose contours are
    15..18, 62..67, 317..324 and 600 rows tall, several per image, with and without a hole."""
    heights = list(range(13, 19)) + list(range(60, 68)) + list(range(315, 325)) + [600]
    for k0 in range(0, len(heights), 3):
        img = np.zeros((700, 640), np.uint8)
        for j, hh in enumerate(heights[k0:k0 + 3]):
            x0 = 20 + 200 * j
            for r in range(hh):
                x = x0 + (r * 60) // max(hh, 1)                 # a bar leaning to the right, 9 px wide
                img[40 + r, x:x + 9] = 230
                if j == 1 and 3 < r < hh - 3:
                    img[40 + r, x + 3:x + 6] = 0                 # ... the middle one hollow: a hole border of nearly the same height
        for min_len, lw in ((1, 5), (0, 1.5)):
            det_o, box_o, nb_o = oracle.fit_min_area_rect(img, 1, 1, min_len, lw)
            det_g, box_g, nb_g = gpu_ctx.fit_min_area_rect(img, 1, 1, min_len, lw)
            assert (det_g, nb_g) == (det_o, nb_o), heights[k0:k0 + 3]
            assert np.array_equal(box_g, box_o), heights[k0:k0 + 3]
        assert nb_g >= 1


def test_device_libm_agrees_with_the_host_on_the_rectangle_path(gpu_ctx, oracle):
    """minAreaRect's angle (atan2 in double, rounded to float32, in degrees) and boxPoints' cos / sin of it are the only
    libm calls on the accept / reject path; the device evaluates them with its own libm, the oracle with glibc.  Both are
    accurate to about an ulp in double, so their float32 roundings can only differ where the true value sits on a float32
    rounding boundary.  4 million operand pairs of the kind the path produces (edge vectors of integer-coordinate hulls,
    float32 calipers vectors, the axes) and none differs."""
    rng = np.random.default_rng(99)
    n = 1_000_000
    ys = [rng.integers(-4096, 4097, n).astype(np.float64), rng.normal(0, 300, n).astype(np.float32).astype(np.float64),
          (rng.integers(-64, 65, n) / 2.0), rng.normal(0, 3, n).astype(np.float32).astype(np.float64)]
    xs = [rng.integers(-4096, 4097, n).astype(np.float64), rng.normal(0, 300, n).astype(np.float32).astype(np.float64),
          (rng.integers(-64, 65, n) / 2.0), rng.normal(0, 3, n).astype(np.float32).astype(np.float64)]
    axes = np.array([[0, 1], [1, 0], [0, -1], [-1, 0], [0, 0], [1, 1], [-1, 1], [1, -1], [-1, -1]], np.float64)
    ys.append(axes[:, 0]); xs.append(axes[:, 1])
    bad = 0
    for y, x in zip(ys, xs):
        dev = gpu_ctx.debug_trig(y, x)
        host = oracle.debug_trig(y, x)
        for d, h_ in zip(dev, host):
            bad += int(np.count_nonzero(d.view(np.uint32) != h_.view(np.uint32)))
    assert bad == 0


def test_unsupported_contour_knobs_raise(gpu_ctx):
    from lfd_amd import _native
    img = np.zeros((32, 32), np.uint8)
    for mode, method in ((7, 1), (1, 3), (1, 4)):          # the TC89 approximations change the point set: not restated
        with pytest.raises(_native.NativeError) as e:
            gpu_ctx.fit_min_area_rect(img, mode, method)
        assert e.value.code == _native.ERR_UNSUPPORTED


@pytest.mark.parametrize("shape", SHAPES[2:])
@pytest.mark.parametrize("rho", [20, 10, 5, 1, 7.5])
def test_hough(gpu_ctx, oracle, shape, rho):
    rng = np.random.default_rng(19)
    line = np.zeros(shape, np.uint8)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    line[np.abs((xx - shape[1] / 2) * 0.6 - (yy - shape[0] / 2) * 0.8) < 1.5] = 9
    for img in (line, rand_u8(rng, shape, sparse=True), np.zeros(shape, np.uint8), blobs(rng, shape)):
        acc_o = oracle.hough_accum(img, rho)
        assert np.array_equal(gpu_ctx.hough_accum(img, rho), acc_o)
        for thr in (1, 30):
            l_o, n_o = oracle.hough_lines(img, rho, threshold=thr)
            l_g, n_g = gpu_ctx.hough_lines(img, rho, threshold=thr)
            assert n_g == n_o
            assert (l_o is None and l_g is None) or np.array_equal(l_g, l_o)
        l_g3, n_g3 = gpu_ctx.hough_lines(img, rho, max_lines=3)
        l_o3, _ = oracle.hough_lines(img, rho, max_lines=3)
        assert (l_o3 is None and l_g3 is None) or np.array_equal(l_g3, l_o3)


def test_hough_other_theta(gpu_ctx, oracle):
    rng = np.random.default_rng(20)
    img = rand_u8(rng, (120, 90), sparse=True)
    for theta in (np.pi / 90, np.pi / 180, np.pi / 360):
        assert np.array_equal(gpu_ctx.hough_accum(img, 4, theta), oracle.hough_accum(img, 4, theta))


def test_hough_size_independent_properties_at_full_size(gpu_ctx):
    """BASELINE-size frame: linearity of the accumulator in the pixel set, one vote per pixel and
    angle, empty guard rows/columns -- no oracle needed."""
    rng = np.random.default_rng(21)
    a = ((rng.random((1489, 2048)) < 0.01) * 255).astype(np.uint8)
    b = ((rng.random((1489, 2048)) < 0.01) * 255).astype(np.uint8)
    b[a != 0] = 0
    acc_a, acc_b, acc_ab = (gpu_ctx.hough_accum(x, 20) for x in (a, b, a | b))
    assert acc_ab.shape == (182, 356) and np.array_equal(acc_ab, acc_a + acc_b)
    assert (acc_a[1:-1].sum(axis=1) == np.count_nonzero(a)).all()
    assert acc_a[0].sum() == 0 and acc_a[-1].sum() == 0 and acc_a[:, 0].sum() == 0 and acc_a[:, -1].sum() == 0
    # determinism: same bytes on a second run
    assert np.array_equal(gpu_ctx.hough_accum(a, 20), acc_a)


def test_remove_stars(gpu_ctx, oracle):
    from lfd_amd import _native, synth
    rng = np.random.default_rng(22)
    h, w = 120, 170
    frames, cats = [], []
    for _ in range(3):
        n = 60
        cats.append({"ROWC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32),
                     "COLC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32),
                     "PSFMAG": rng.uniform(14, 24, (n, 5)).astype(np.float32),
                     "PETROTH90": rng.uniform(-2, 30, (n, 5)).astype(np.float32),
                     "NOBSERVE": rng.integers(1, 3, n).astype(np.int32), "NDETECT": rng.integers(1, 3, n).astype(np.int32)})
        cats[-1]["PSFMAG"][::7, 1] = -9999
        frames.append(rng.normal(1, 1, (h, w)).astype(np.float32))
    for flt in "ugriz":
        kw = dict(defaultxy=6, maxxy=25, pixscale=0.396, magcount=3, maxmagdiff=3,
                  filter_caps={'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5})
        batch = np.stack(frames).copy()
        gpu_ctx.remove_stars(batch, synth.pack_catalogs(cats), _native.make_rs_params(flt, **kw))
        for i in range(3):
            want = oracle.remove_stars(frames[i].copy(), cats[i], oracle.rs_params(flt, **kw))
            assert np.array_equal(batch[i], want)
        again = batch.copy()
        gpu_ctx.remove_stars(again, synth.pack_catalogs(cats), _native.make_rs_params(flt, **kw))
        assert np.array_equal(again, batch)                  # idempotent


def test_remove_stars_crowded_catalogue(oracle, monkeypatch):
    """Thousands of objects per frame take another route (squares sorted by first row, every band of rows blotted once through
    an LDS bit plane: k_rs_sort / k_rs_fill_bands): same frames as the oracle, overlapping squares, squares clipped at every
    edge, objects that stay; and the same route forced on a small catalogue (LFDMI_RS_SORT_MIN=0)."""
    from lfd_amd import _native, synth
    rng = np.random.default_rng(23)
    kw = dict(defaultxy=6, maxxy=25, pixscale=0.396, magcount=3, maxmagdiff=3,
              filter_caps={'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5})
    for (h, w, n), env in (((200, 256, 3000), None), ((97, 160, 50), "0"), ((64, 96, 0), "0")):
        if env is not None:
            monkeypatch.setenv("LFDMI_RS_SORT_MIN", env)
        frames, cats = [], []
        for _ in range(2):
            cats.append({"ROWC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32),
                         "COLC": rng.uniform(-5, w + 5, (n, 5)).astype(np.float32),
                         "PSFMAG": rng.uniform(14, 24, (n, 5)).astype(np.float32),
                         "PETROTH90": rng.uniform(-2, 30, (n, 5)).astype(np.float32),
                         "NOBSERVE": rng.integers(1, 3, n).astype(np.int32), "NDETECT": rng.integers(1, 3, n).astype(np.int32)})
            frames.append(rng.normal(1, 1, (h, w)).astype(np.float32))
        if n == 0:
            continue                                         # (an empty catalogue packs to nothing: covered by the pipeline tests)
        with _native.Context(0, h, w, 2) as ctx:
            for flt in "ri":
                batch = np.stack(frames).copy()
                ctx.remove_stars(batch, synth.pack_catalogs(cats), _native.make_rs_params(flt, **kw))
                for i in range(2):
                    want = oracle.remove_stars(frames[i].copy(), cats[i], oracle.rs_params(flt, **kw))
                    assert np.array_equal(batch[i], want), (h, w, n, flt, i)


def test_optional_gaussian_stage(gpu_ctx, oracle):
    """The Gaussian smoothing north_star lists inside Canny -- off by default, because cv2.Canny has none
    (processfield.py:236): the operator against the oracle's definition, and a pass with gaussKernel set."""
    rng = np.random.default_rng(11)
    for shape in ((40, 50), (97, 131), (64, 256), (5, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for ksize, sigma in ((1, 0), (3, 0), (5, 0), (7, 0), (9, 0), (5, 1.7), (15, 3.0), (31, 0)):
            assert np.array_equal(gpu_ctx.gaussian_blur(img, ksize, sigma), oracle.gaussian_blur(img, ksize, sigma)), (shape, ksize, sigma)
    batch = rng.integers(0, 256, (3, 33, 70), dtype=np.uint8)
    out = gpu_ctx.gaussian_blur(batch, 5)
    for i in range(3):
        assert np.array_equal(out[i], oracle.gaussian_blur(batch[i], 5))
    from lfd_amd import synth
    from lfd_amd.detecttrails import default_params
    pb, pd, _ = default_params()
    img = synth.make_frame(0, with_catalog=False)[0][::-1].copy()
    for p, fn_g, fn_o in ((dict(pb, gaussKernel=5), gpu_ctx.process_bright, oracle.process_bright),
                          (dict(pd, gaussKernel=3, gaussSigma=1.0), gpu_ctx.process_dim, oracle.process_dim)):
        res = fn_g(img, p)[0]
        want = fn_o(img, p)
        assert all(res[k].item() == v for k, v in want.items()), (want, res)
    off = gpu_ctx.process_bright(img, pb)[0]
    assert off.tobytes() == gpu_ctx.process_bright(img, dict(pb, gaussKernel=0))[0].tobytes()


def test_device_check_theta_and_host_dictify_against_the_reference_vectors():
    """The only arithmetic the reference itself pins (tests/golden/tail_fixtures.json: 309 check_theta and 206 dictify_hough
    vectors produced by processfield.py:36-150 / :266-288): fed straight to k_finalize (the device's check_theta, float64,
    numpy's summation order, the zero fill of :89-102) and to the library's host-side dictify -- until round 4 these two were
    only checked through the oracle."""
    import json
    import os
    from lfd_amd import _native as N
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tail_fixtures.json")))
    with N.Context(0, 64, 64, 32) as ctx:                             # (several chunks of 32)
        # check_theta: one call per parameter set
        groups = {}
        for c in fx["check_theta"]:
            groups.setdefault((c["navg"], c["dro"], c["thetaTresh"], c["lineSetTresh"]), []).append(c)
        checked = 0
        for (navg, dro, tt, lt), cases in groups.items():
            kmax = max(max(len(c["h1"]), len(c["h2"])) for c in cases)
            h1 = np.zeros((len(cases), kmax, 2), np.float32)
            h2 = np.zeros_like(h1)
            n1 = np.array([len(c["h1"]) for c in cases], np.int32)
            n2 = np.array([len(c["h2"]) for c in cases], np.int32)
            for i, c in enumerate(cases):
                h1[i, :n1[i]] = np.asarray(c["h1"], np.float32).reshape(-1, 2)
                h2[i, :n2[i]] = np.asarray(c["h2"], np.float32).reshape(-1, 2)
            for which in (1, 2):
                rec = ctx.debug_tail(h1, n1, h2, n2, navg, dro, tt, lt, which, (1489, 2048))
                for i, c in enumerate(cases):
                    rejected = c["out"] is True                      # check_theta: True = reject, None = accept
                    assert bool(rec["rejected_by_theta"][i]) == rejected, c
                    assert rec["found"][i] == (0 if rejected else which), c
                    if not rejected:
                        assert rec["rho"][i] == np.float32(c["h1"][0][0]) and rec["theta"][i] == np.float32(c["h1"][0][1])
                    checked += 1
        assert checked == 2 * len(fx["check_theta"]) == 618
        # dictify_hough: a single accepted line per record, per image shape
        by_shape = {}
        for c in fx["dictify_hough"]:
            by_shape.setdefault(tuple(c["shape"]), []).append(c)
        done = 0
        for shape, cases in by_shape.items():
            h = np.array([[[c["rho"], c["theta"]]] for c in cases], np.float32)
            ones = np.ones(len(cases), np.int32)
            rec = ctx.debug_tail(h, ones, h, ones, 1, 25.0, 0.15, 0.15, 1, shape)
            for i, c in enumerate(cases):
                assert rec["found"][i] == 1
                assert {k: int(rec[k][i]) for k in ("x1", "y1", "x2", "y2")} == c["out"], c
                done += 1
        assert done == len(fx["dictify_hough"]) == 206
