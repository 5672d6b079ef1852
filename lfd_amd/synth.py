"""Deterministic synthetic SDSS-like frames and photoObj-like catalogues.

Frame ``k`` is a pure function of ``k`` (numpy ``default_rng(PCG64)``, seed 20240000 + k), so
the GPU box and the build container produce identical inputs.  There is no network for real
SDSS data; shapes and statistics follow the reference's input contract: a float32
(1489, 2048) nanomaggie image (detecttrails.py:113) and the eight photoObj columns that
removestars.py:96-104 reads.
"""
import numpy as np

SDSS_SHAPE = (1489, 2048)
LSST_SHAPE = (4096, 4096)
SEED0 = 20240000

BRIGHT_PEAK = 5.0
DIM_PEAK = 0.15


def _add_stars(img, ys, xs, flux, sigma):
    h, w = img.shape
    rad = int(np.ceil(6 * sigma))
    ax = np.arange(-rad, rad + 1, dtype=np.float64)
    norm = 1.0 / (2 * np.pi * sigma * sigma)
    for y, x, f in zip(ys, xs, flux):
        iy, ix = int(np.floor(y)), int(np.floor(x))
        gy = np.exp(-((ax + iy - y) ** 2) / (2 * sigma * sigma))
        gx = np.exp(-((ax + ix - x) ** 2) / (2 * sigma * sigma))
        y0, y1 = max(iy - rad, 0), min(iy + rad + 1, h)
        x0, x1 = max(ix - rad, 0), min(ix + rad + 1, w)
        if y0 >= y1 or x0 >= x1:
            continue
        stamp = (f * norm) * np.outer(gy[y0 - iy + rad:y1 - iy + rad], gx[x0 - ix + rad:x1 - ix + rad])
        img[y0:y1, x0:x1] += stamp.astype(np.float32)


def _add_streak(img, y0, x0, angle_deg, peak, sigma):
    h, w = img.shape
    phi = np.deg2rad(angle_deg)
    yy = np.arange(h, dtype=np.float32)[:, None]
    xx = np.arange(w, dtype=np.float32)[None, :]
    d = (xx - np.float32(x0)) * np.float32(np.sin(phi)) - (yy - np.float32(y0)) * np.float32(np.cos(phi))
    img += (np.float32(peak) * np.exp(-(d * d) / np.float32(2 * sigma * sigma))).astype(np.float32)


def make_frame(k, shape=SDSS_SHAPE, n_star=None, with_catalog=True, sky_sigma=0.025, bleed=False):
    """Returns (float32 image, catalogue dict or None, truth dict).

    ``n_star``, ``sky_sigma`` and ``bleed`` are the knobs of the stress workloads (bench.py ``stress`` leg, tests): crowded fields,
    noisier sky, a saturated star with a full-height bleed column.  The defaults are the benchmark recipe of SURVEY.md 8(d); the
    extra draws of ``bleed`` come after every other draw, so the default frames are unchanged."""
    h, w = shape
    if n_star is None:
        n_star = 400 if shape == SDSS_SHAPE else int(round(400 * (h * w) / (1489 * 2048)))
    rng = np.random.default_rng(np.random.PCG64(SEED0 + int(k)))
    img = rng.normal(0.0, sky_sigma, (h, w)).astype(np.float32)
    ys = rng.uniform(0, h, n_star)
    xs = rng.uniform(0, w, n_star)
    flux = np.exp(rng.uniform(np.log(1.0), np.log(2000.0), n_star))
    _add_stars(img, ys, xs, flux, 1.5)
    has_streak = bool(rng.random() < 0.75)
    sy = rng.uniform(0.2 * h, 0.8 * h)
    sx = rng.uniform(0.2 * w, 0.8 * w)
    ang = rng.uniform(5.0, 85.0)
    if rng.random() < 0.5:
        ang += 90.0
    kind = "none"
    if has_streak:
        kind = "bright" if k % 2 == 0 else "dim"
        _add_streak(img, sy, sx, ang, BRIGHT_PEAK if kind == "bright" else DIM_PEAK, 2.0)
    truth = {"k": int(k), "streak": kind, "y0": float(sy), "x0": float(sx), "angle_deg": float(ang)}
    cat = None
    if with_catalog:
        n_decoy = max(1, n_star // 20)
        n = n_star + n_decoy
        rowc = np.empty((n, 5), np.float32)
        colc = np.empty((n, 5), np.float32)
        rowc[:n_star] = ys[:, None]
        colc[:n_star] = xs[:, None]
        mag = (22.5 - 2.5 * np.log10(flux))[:, None] + rng.uniform(-0.2, 0.2, (n_star, 5))
        psf = np.empty((n, 5), np.float32)
        psf[:n_star] = mag
        pet = rng.uniform(1.0, 8.0, (n, 5)).astype(np.float32)
        nob = np.ones(n, np.int32)
        nde = np.ones(n, np.int32)
        # decoys: must NOT be masked (NOBSERVE != NDETECT, or one band at -9999)
        dy = rng.uniform(0, h, n_decoy)
        dx = rng.uniform(0, w, n_decoy)
        rowc[n_star:] = dy[:, None]
        colc[n_star:] = dx[:, None]
        psf[n_star:] = rng.uniform(15.0, 21.0, (n_decoy, 1)) + rng.uniform(-0.2, 0.2, (n_decoy, 5))
        for j in range(n_decoy):
            if j % 2 == 0:
                nob[n_star + j] = 2
            else:
                psf[n_star + j, int(rng.integers(0, 5))] = -9999.0
        cat = {"ROWC": rowc, "COLC": colc, "PSFMAG": psf, "PETROTH90": pet, "NOBSERVE": nob,
               "NDETECT": nde}
    if bleed:
        # a saturated star: flat-topped core of 12 px radius at the CCD's full well and a 3-px bleed trail down its whole
        # column (what a 6th-magnitude star does to an SDSS frame); it is not in the catalogue (saturated objects are flagged
        # out of photoObj's clean sample), so remove_stars leaves it alone
        by, bx = float(rng.uniform(0.1 * h, 0.9 * h)), int(rng.integers(16, w - 16))
        yy = np.arange(h, dtype=np.float32)[:, None]
        xx = np.arange(w, dtype=np.float32)[None, :]
        core = (yy - np.float32(by)) ** 2 + (xx - np.float32(bx)) ** 2 <= np.float32(144.0)
        img[core] = np.float32(1200.0)
        img[:, bx - 1:bx + 2] = np.float32(1200.0)
        truth["bleed_x"] = bx
    return img, cat, truth


def make_config1_frame(shape=SDSS_SHAPE):
    """BASELINE config 1: uint8 zeros + one 5-px-wide streak of value 200 + 50 salt pixels (255)."""
    h, w = shape
    rng = np.random.default_rng(np.random.PCG64(1))
    img = np.zeros((h, w), np.uint8)
    yy = np.arange(h, dtype=np.float32)[:, None]
    xx = np.arange(w, dtype=np.float32)[None, :]
    phi = np.deg2rad(35.0)
    d = (xx - np.float32(w / 2)) * np.float32(np.sin(phi)) - (yy - np.float32(h / 2)) * np.float32(np.cos(phi))
    img[np.abs(d) <= 2.5] = 200
    sy = rng.integers(0, h, 50)
    sx = rng.integers(0, w, 50)
    img[sy, sx] = 255
    return img


from .catalogs import pack_catalogs  # noqa: E402,F401  (kept importable from here: tests and tools use synth.pack_catalogs)


def make_portable_frame(k, shape=(512, 768), n_star=40, with_catalog=True):
    """A frame built ONLY from integer RNG draws and IEEE +,-,*,/,sqrt, so that every machine
    produces the same bits (``make_frame`` uses exp/normal, whose last bit may depend on the
    CPU's vector math library).  Used for golden fixtures that travel to the GPU box."""
    h, w = shape
    rng = np.random.default_rng(np.random.PCG64(7700000 + int(k)))
    u = rng.integers(0, 1 << 16, (h, w), dtype=np.int64)
    img = ((u - 32768).astype(np.float64) / 65536.0 * 0.1).astype(np.float32)       # sky in [-0.05, 0.05)
    ys = rng.integers(0, h, n_star)
    xs = rng.integers(0, w, n_star)
    amp = rng.integers(1, 400, n_star)
    rad = rng.integers(2, 7, n_star)
    yy = np.arange(h, dtype=np.int64)[:, None]
    xx = np.arange(w, dtype=np.int64)[None, :]
    for y, x, a, r in zip(ys, xs, amp, rad):
        y0, y1, x0, x1 = max(y - r, 0), min(y + r + 1, h), max(x - r, 0), min(x + r + 1, w)
        d = np.abs(yy[y0:y1] - y) + np.abs(xx[:, x0:x1] - x)
        prof = np.maximum(0, (r + 1) - d).astype(np.float64) * (float(a) / float(r + 1))
        img[y0:y1, x0:x1] += prof.astype(np.float32)
    kind = ("bright", "dim", "none")[k % 3]
    dirs = [(1, 2), (2, 1), (1, 1), (3, -1), (-1, 3), (5, 2), (-2, 5), (1, -4)]
    a, b = dirs[int(rng.integers(0, len(dirs)))]
    y0 = int(rng.integers(h // 4, 3 * h // 4))
    x0 = int(rng.integers(w // 4, 3 * w // 4))
    if kind != "none":
        num = np.abs((xx - x0) * a - (yy - y0) * b).astype(np.float64)          # |cross| = distance * norm
        dist = num / np.sqrt(float(a * a + b * b))
        peak = 6.0 if kind == "bright" else 0.2
        prof = np.maximum(0.0, 1.0 - dist / 5.0) * peak
        img += prof.astype(np.float32)
    truth = {"k": int(k), "streak": kind, "y0": y0, "x0": x0, "dir": [a, b]}
    cat = None
    if with_catalog:
        n = n_star
        rowc = np.repeat(ys.astype(np.float32)[:, None], 5, 1) + np.float32(0.25)
        colc = np.repeat(xs.astype(np.float32)[:, None], 5, 1) + np.float32(0.25)
        mag = (rng.integers(140, 240, (n, 5)).astype(np.float32)) / np.float32(10.0)
        pet = (rng.integers(-10, 120, (n, 5)).astype(np.float32)) / np.float32(8.0)
        nob = rng.integers(1, 3, n).astype(np.int32)
        nde = np.where(rng.integers(0, 10, n) < 8, nob, nob + 1).astype(np.int32)
        mag[::9, int(k) % 5] = -9999.0
        cat = {"ROWC": rowc, "COLC": colc, "PSFMAG": mag, "PETROTH90": pet, "NOBSERVE": nob, "NDETECT": nde}
    return img, cat, truth


# ---- a synthetic $BOSS tree (tests, bench.py's dropin leg) ----------------------------------------------------------
BOSS_HEADER = {"TAI": 4649973000.5, "CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.5, "CRVAL2": -1.25,
               "CD1_1": 1e-4, "CD1_2": 2e-5, "CD2_1": -2e-5, "CD2_2": 1e-4}


def write_boss_tree(root, frames, cats, run=94, camcol=1, filter="r", field0=100, bz2_all=False, bz2_fields=(), threads=None,
                    rerun=301, link_to=None):
    """Write ``frames[i]`` / ``cats[i]`` as field ``field0 + i`` of (run, camcol, filter) in the directory layout the
    reference reads (docs/source/lfd/setup.rst:9-22: $BOSS_PHOTOOBJ/frames/<rerun>/<run>/<camcol>/frame-*.fits[.bz2],
    $BOSS_PHOTOOBJ/<rerun>/<run>/<camcol>/photoObj-*.fits, $PHOTO_REDUX/runList.par), point the two environment variables
    at it and return the header values every frame file carries.  ``cats[i] is None``: no photoObj file for that field.
    ``link_to`` = M > len(frames): fields field0 + len(frames) .. field0 + M - 1 are hard links of the written files (field
    field0 + i repeats frame i % len(frames)): a long run for throughput measurements without M distinct frames."""
    import bz2
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .detecttrails import fitslite, sdssfiles
    root = str(root)
    redux = os.path.join(root, "photo", "redux")
    os.makedirs(redux, exist_ok=True)
    n = len(frames)
    total = max(n, int(link_to or 0))
    with open(os.path.join(redux, "runList.par"), "w") as f:
        f.write("typedef struct {\n int run;\n char rerun[];\n int exist;\n int done;\n int calib;\n int startfield;\n"
                " int endfield;\n char machine[];\n char disk[];\n} RUNDATA;\n\n"
                f"RUNDATA {run} {rerun} 1 1 1 {field0} {field0 + total} m d\n")
    os.environ["PHOTO_REDUX"] = redux
    os.environ["BOSS_PHOTOOBJ"] = os.path.join(root, "photoObj")
    sdssfiles._runlist_cache.clear()
    fdir = os.path.dirname(sdssfiles.filename("frame", run, camcol, field0, filter, rerun=rerun))
    pdir = os.path.dirname(sdssfiles.filename("photoObj", run, camcol, field0, rerun=rerun))
    os.makedirs(fdir, exist_ok=True)
    os.makedirs(pdir, exist_ok=True)
    packed = set(bz2_fields)

    def one(i):
        field = field0 + i
        fpath = sdssfiles.filename("frame", run, camcol, field, filter, rerun=rerun)
        if bz2_all or field in packed:
            tmp = fpath + ".tmp"
            fitslite.write_image(tmp, frames[i], BOSS_HEADER)
            with open(tmp, "rb") as f, open(fpath + ".bz2", "wb") as g:
                g.write(bz2.compress(f.read()))
            os.remove(tmp)
        else:
            fitslite.write_image(fpath, frames[i], BOSS_HEADER)
        if cats[i] is not None:
            cols = dict(cats[i])
            m = len(cols["NOBSERVE"])
            cols["OBJC_TYPE"] = np.zeros(m, np.int32)
            cols["TYPE"] = np.zeros((m, 5), np.int32)
            fitslite.write_table(sdssfiles.filename("photoObj", run, camcol, field, rerun=rerun), cols)

    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    with ThreadPoolExecutor(max(1, min(threads or cores, 32))) as ex:
        list(ex.map(one, range(n)))
    for j in range(n, total):
        i = j % n
        for kind, flt in (("frame", filter), ("photoObj", None)):
            src = sdssfiles.filename(kind, run, camcol, field0 + i, flt, rerun=rerun)
            dst = sdssfiles.filename(kind, run, camcol, field0 + j, flt, rerun=rerun)
            for ext in ("", ".bz2"):
                if os.path.exists(src + ext):
                    os.link(src + ext, dst + ext)
    return dict(BOSS_HEADER)


# ---- many frames at once (bench.py, tools/) -------------------------------------------------------
_TOOL_ENV = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_LIBRARY")


def _worker_main(argv):
    """``python -m lfd_amd.synth <shared file> <n> <h> <w> <i0> <k_first> <count> <with_catalog>``: fills rows
    i0 .. i0+count-1 of the shared [n, h, w] float32 array with frames k_first .., pickles the catalogues to stdout."""
    import pickle
    import sys
    import json
    name, n, h, w, i0, k_first, count, with_cat = argv[0], *[int(x) for x in argv[1:8]]
    recipes = json.loads(argv[8]) if len(argv) > 8 else None      # per-frame keyword arguments of make_frame (stress workloads)
    out = np.memmap(name, np.float32, "r+", shape=(n, h, w))
    cats = []
    for j in range(count):
        img, cat, truth = make_frame(k_first + j, (h, w), with_catalog=bool(with_cat), **(recipes[j] if recipes else {}))
        out[i0 + j] = img
        cats.append((cat, truth))
    out.flush()
    del out
    sys.stdout.buffer.write(pickle.dumps(cats))
    sys.stdout.buffer.flush()


# The stress workloads of bench.py's `stress` leg and tests/test_gpu_stress.py: what the benchmark's sky (400 stars, sigma 0.025)
# does not exercise -- crowded fields, noisier sky (half of it survives the dim pass's minFlux at sigma 0.1), a saturated star
# with a full-height bleed column, one crowded frame (more candidate runs than the per-frame kernels' LDS table holds) in an
# otherwise quiet chunk.  name -> per-frame keyword arguments of make_frame (a function of the frame's position in the batch).
STRESS = {
    "stars4000": lambda i: {"n_star": 4000},
    "stars20000": lambda i: {"n_star": 20000},
    "sky0.05": lambda i: {"sky_sigma": 0.05},
    "sky0.1": lambda i: {"sky_sigma": 0.1},
    "bleed": lambda i: {"bleed": True},
    "one_crowded": lambda i: {"n_star": 8000} if i % 16 == 5 else {},
}


def stress_recipes(name, n):
    return [STRESS[name](i) for i in range(n)]


def make_frames(k0, n, shape=SDSS_SHAPE, workers=None, with_catalog=True, recipes=None, with_truth=False):
    """Frames k0 .. k0+n-1 as one float32 array [n, h, w] plus their catalogues, generated by ``workers`` child
    processes that write into shared memory.  ``recipes``: per-frame keyword arguments of make_frame (stress workloads).
    ``with_truth``: also return the frames' truth dicts (what was injected where), as a third value.

    The workers are separate ``python -m lfd_amd.synth`` programs started as child processes, never forked copies of
    the caller: the caller may already have initialised the GPU -- under rocprofv3 the profiler's preloaded tool
    library does that before ``main`` runs -- and a forked copy of such a process can hang.  The tool-library
    variables are removed from the children's environment and the GPU is hidden from them, so they neither load the
    profiler nor touch the card.
    """
    import os
    import pickle
    import subprocess
    import sys
    h, w = shape
    workers = workers or min(16, os.cpu_count() or 1)
    workers = max(1, min(workers, n // 2))
    if workers <= 1:
        out = np.empty((n, h, w), np.float32)
        cats, truths = [], []
        for i in range(n):
            img, cat, truth = make_frame(k0 + i, shape, with_catalog=with_catalog, **(recipes[i] if recipes else {}))
            out[i] = img
            cats.append(cat)
            truths.append(truth)
        return (out, cats, truths) if with_truth else (out, cats)
    import tempfile
    env = {k: v for k, v in os.environ.items()
           if k not in _TOOL_ENV and not k.startswith("ROCPROF") and not k.startswith("ROCTX")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["ROCR_VISIBLE_DEVICES"] = ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    # a file in /dev/shm (memory backed) that parent and children map; removed before returning
    shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    fd, path = tempfile.mkstemp(prefix="lfd_synth_", suffix=".f32", dir=shm_dir)
    procs = []
    try:
        os.ftruncate(fd, n * h * w * 4)
        os.close(fd)
        for r in range(workers):
            a, b = n * r // workers, n * (r + 1) // workers
            if b > a:
                cmd = [sys.executable, "-m", "lfd_amd.synth", path, str(n), str(h), str(w), str(a), str(k0 + a),
                       str(b - a), str(int(with_catalog))]
                if recipes:
                    import json
                    cmd.append(json.dumps(recipes[a:b]))
                procs.append((a, b, subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, cwd=root)))
        cats = [None] * n
        for a, b, p in procs:
            blob, _ = p.communicate()
            if p.returncode:
                raise RuntimeError(f"frame generator worker failed with exit code {p.returncode}")
            cats[a:b] = pickle.loads(blob)
        truths = [c[1] for c in cats]
        cats = [c[0] for c in cats]
        out = np.fromfile(path, np.float32).reshape(n, h, w)
    finally:
        for _, _, p in procs:
            if p.poll() is None:
                p.kill()
        os.unlink(path)
    return (out, cats, truths) if with_truth else (out, cats)


if __name__ == "__main__":
    import sys
    _worker_main(sys.argv[1:])
