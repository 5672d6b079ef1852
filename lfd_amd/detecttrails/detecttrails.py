"""SDSS-facing driver: which frames to process, in what order, and what gets written.

Host-side mirror of ``lfd/detecttrails/detecttrails.py``.  ``DetectTrails(**kwargs).process()``
keeps the reference's keyword interface, its three parameter dictionaries
(``params_bright`` / ``params_dim`` / ``params_removestars``, same keys and defaults,
detecttrails.py:202-239) and its frame-selection rules (detecttrails.py:290-342); per frame,
``process_field`` runs remove_stars -> vertical flip -> bright pass -> dim pass on the GPU
through one ``lfdmi_detect_batch`` call and appends a results row / an errors entry in the
reference's text formats.

Deliberate fixes of reference bugs: C8 ``params_dim=`` /
``params_removestars=`` keyword arguments land in their own dictionaries; C9 the results row
carries the 13 header values instead of literal ``{h['CRPIX2']}`` text.
"""
import bz2
import os
import traceback

import numpy as _np

from .. import _native
from . import fitslite, sdssfiles
from .processfield import get_context, use_context, setup_debug, check_theta, dictify_hough  # noqa: F401
from .removestars import read_photoObj_arrays

# values of the cv2 constants the reference re-exports (detecttrails.py:14-18)
RETR_EXTERNAL, RETR_LIST, RETR_CCOMP, RETR_TREE = 0, 1, 2, 3
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE, CHAIN_APPROX_TC89_L1, CHAIN_APPROX_TC89_KCOS = 1, 2, 3, 4

__all__ = ["DetectTrails", "process_field", "process_fields_batched", "process_frame_arrays", "default_params"]

_HEADER_KEYS = ("TAI", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CD1_1", "CD1_2", "CD2_1", "CD2_2")


def default_params():
    """Fresh copies of the three default dictionaries (detecttrails.py:202-239)."""
    bright = {"lwTresh": 5, "thetaTresh": 0.15, "dilateKernel": _np.ones((4, 4), _np.uint8),
              "contoursMode": RETR_LIST, "contoursMethod": CHAIN_APPROX_NONE, "minAreaRectMinLen": 1,
              "houghMethod": 20, "nlinesInSet": 3, "lineSetTresh": 0.15, "dro": 25, "debug": False}
    dim = {"minFlux": 0.02, "addFlux": 0.5, "lwTresh": 5, "thetaTresh": 0.15,
           "erodeKernel": _np.ones((3, 3), _np.uint8), "dilateKernel": _np.ones((9, 9), _np.uint8),
           "contoursMode": RETR_LIST, "contoursMethod": CHAIN_APPROX_NONE, "minAreaRectMinLen": 1,
           "houghMethod": 20, "nlinesInSet": 3, "lineSetTresh": 0.15, "dro": 20, "debug": False}
    removestars = {"pixscale": 0.396, "defaultxy": 20, "maxxy": 60,
                   "filter_caps": {'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5},
                   "magcount": 3, "maxmagdiff": 3, "debug": False}
    return bright, dim, removestars


def _rs_struct(filter, params_removestars):
    p = {k: v for k, v in params_removestars.items() if k != "debug"}
    return _native.make_rs_params(filter, **p)


def process_frame_arrays(img, cat, filter, params_bright, params_dim, params_removestars):
    """The hot part of process_field (detecttrails.py:119-131) on arrays.

    ``img``: float32 (h, w) frame, blotted in place by remove_stars like the reference;
    ``cat``: photoObj columns (dict) or None.  Returns ``(detection, res_dict_or_None, record)``.
    """
    if img.dtype != _np.float32 or not img.flags.c_contiguous:
        raise TypeError("process_frame_arrays needs a C-contiguous float32 frame")
    packed = rs = None
    if cat is not None and len(cat["NOBSERVE"]):
        from .removestars import _check_finite
        _check_finite(cat)
        n = len(cat["NOBSERVE"])
        packed = {"count": _np.array([n], _np.int32)}
        for key in ("ROWC", "COLC", "PSFMAG", "PETROTH90"):
            packed[key] = _np.ascontiguousarray(cat[key], _np.float32).reshape(1, n, 5)
        for key in ("NOBSERVE", "NDETECT"):
            packed[key] = _np.ascontiguousarray(cat[key], _np.int32).reshape(1, n)
        rs = _rs_struct(filter, params_removestars)
    with use_context(*img.shape) as ctx:
        rec = ctx.detect_batch(img, params_bright, params_dim, packed, rs)[0]
    status = int(rec["status"])
    if status == _native.ERR_NOLINES:
        raise TypeError("'NoneType' object is not subscriptable")  # HoughLines gave None
    if status:
        raise _native.NativeError(status, "frame failed on the device")
    if rec["found"]:
        # coordinates the way the reference computes them: numpy float32 scalars
        res = dictify_hough(img.shape, (_np.float32(rec["rho"]), _np.float32(rec["theta"])))
        return True, res, rec
    return False, None, rec


def _load_frame(run, camcol, filter, field):
    """Frame image (float32, C-contiguous), results-row head, photoObj columns; raises like the
    reference when neither the .fits nor the .fits.bz2 exists (detecttrails.py:81-87)."""
    path = sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=filter)
    if not os.path.exists(path):
        if not os.path.exists(path + ".bz2"):
            raise FileNotFoundError(("File {0} or its bz2 compressed version not found. "
                                     "Are you sure they exist?").format(path))
        path = path + ".bz2"  # decompressed in memory; no $FITS_DUMP round trip needed
    img, h = fitslite.read_image(path)
    img = _np.ascontiguousarray(img, dtype=_np.float32)
    head = " ".join(str(x) for x in (run, camcol, filter, field, *(h[k] for k in _HEADER_KEYS)))
    cat = read_photoObj_arrays(sdssfiles.filename("photoObj", run=run, camcol=camcol, field=field))
    return img, head, cat


def _log_error(errors, ids, exc, debug):
    """errors.txt entry of the reference (detecttrails.py:133-139): ids, 3-frame traceback, message."""
    if debug:
        traceback.print_exception(type(exc), exc, exc.__traceback__, limit=3)
    errors.write("{} {} {} {}\n".format(*ids))
    traceback.print_exception(type(exc), exc, exc.__traceback__, limit=3, file=errors)
    errors.write(str(exc) + "\n\n")


def process_field(results, errors, run, camcol, filter, field, params_bright, params_dim,
                  params_removestars):
    """One frame end to end (reference: detecttrails.py:30-143): locate the frame (or its .bz2),
    read image + header + photoObj, detect, append ``run camcol filter field tai crpix1 crpix2
    crval1 crval2 cd11 cd12 cd21 cd22 x1 y1 x2 y2`` to ``results``; every exception is logged
    to ``errors`` (ids, 3-frame traceback, message) and swallowed."""
    try:
        img, head, cat = _load_frame(run, camcol, filter, field)
        detection, res, _ = process_frame_arrays(img, cat, filter, params_bright, params_dim,
                                                 params_removestars)
        if detection:
            results.write(f"{head} {res['x1']} {res['y1']} {res['x2']} {res['y2']}\n")
    except Exception as e:  # noqa: BLE001 - the reference swallows everything per frame
        _log_error(errors, (run, camcol, filter, field), e, params_bright.get("debug") or params_dim.get("debug"))


def process_fields_batched(results, errors, ids, params_bright, params_dim, params_removestars, loaded=None):
    """Same outcome as calling process_field for every (run, camcol, filter, field) in ``ids``, in order -- the same
    results rows, the same errors entries, a bad frame costs only itself (detecttrails.py:119-139) -- but all frames
    that load go through ONE lfdmi_detect_batch call per (filter, shape) group (frames with different filters use
    different magnitude caps).  ``loaded``: what ``_load_many(ids)`` returned, if the caller read the files already."""
    from .removestars import _check_finite
    if loaded is None:
        loaded = _load_many(ids)
    rows = {}
    debug = params_bright.get("debug") or params_dim.get("debug")
    groups = {}
    for item in loaded:
        if len(item) != 4:
            continue                                  # did not load: its exception is logged below
        key, img, _, cat = item
        try:
            if cat is not None and len(cat["NOBSERVE"]):
                _check_finite(cat)                    # math.ceil(nan) in the reference: this frame's error alone
        except Exception as e:  # noqa: BLE001
            rows[key] = e
            continue
        groups.setdefault((key[2], img.shape), []).append(item)
    from ..catalogs import pack_catalogs
    for (flt, shape), group in groups.items():
        try:
            frames = _np.stack([it[1] for it in group])
            packed = pack_catalogs([it[3] for it in group])
            with use_context(*shape, inflight=min(32, len(group))) as ctx:
                recs = ctx.detect_batch(frames, params_bright, params_dim, packed, _rs_struct(flt, params_removestars))
            for it, rec in zip(group, recs):
                rows[it[0]] = rec
        except Exception:  # noqa: BLE001 - a call-level failure: every frame of the group on its own, under its own try
            for it in group:
                try:
                    rows[it[0]] = process_frame_arrays(it[1], it[3], flt, params_bright, params_dim, params_removestars)[2]
                except Exception as e:  # noqa: BLE001
                    rows[it[0]] = e
    for item in loaded:
        key = item[0]
        try:
            if len(item) == 2:
                raise item[1]
            rec = rows[key]
            if isinstance(rec, Exception):
                raise rec
            status = int(rec["status"])
            if status == _native.ERR_NOLINES:
                raise TypeError("'NoneType' object is not subscriptable")
            if status:
                raise _native.NativeError(status, "frame failed on the device")
            if rec["found"]:
                res = dictify_hough(item[1].shape, (_np.float32(rec["rho"]), _np.float32(rec["theta"])))
                results.write(f"{item[2]} {res['x1']} {res['y1']} {res['x2']} {res['y2']}\n")
        except Exception as e:  # noqa: BLE001
            _log_error(errors, key, e, debug)


def _load_many(ids):
    """[(key, img, head, cat) or (key, exception)] for every key, in order (FITS / bz2 decoding: host work that
    DetectTrails.process overlaps with the GPU passes of the previous chunk)."""
    out = []
    for key in ids:
        try:
            out.append((key,) + _load_frame(*key))
        except Exception as e:  # noqa: BLE001
            out.append((key, e))
    return out


def _frame_shape(keys):
    """(h, w) of the first frame of the selection that can be opened (SDSS frames are all 1489 x 2048; a selection none of
    whose files exists gets the SDSS shape and one errors entry per frame)."""
    from .loader import card_value, header_end
    for key in keys[:64]:
        run, camcol, flt, field = key
        try:
            path = sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=flt)
            if os.path.exists(path):
                with open(path, "rb") as f:
                    head = f.read(16 * fitslite.BLOCK)
            elif os.path.exists(path + ".bz2"):
                with bz2.open(path + ".bz2", "rb") as f:
                    head = f.read(16 * fitslite.BLOCK)
            else:
                continue
            end = header_end(head)
            if end > 0 and card_value(head[:end], b"NAXIS") == 2:
                return int(card_value(head[:end], b"NAXIS2")), int(card_value(head[:end], b"NAXIS1"))
        except Exception:  # noqa: BLE001 - the frame's own error is logged when its turn comes
            continue
    return 1489, 2048


def process_loaded(results, errors, loaded, params_bright, params_dim, params_removestars):
    """process_fields_batched for a chunk the loader has read (``loader.Loaded``): the frames that sit in pinned memory go to
    the GPU as contiguous same-filter slices of that memory with the matching rows of the padded catalogue arrays (no copy
    of a frame on the host, no per-frame Python), the others take the per-frame path; rows and errors entries come out in
    the caller's order, each frame under its own try (detecttrails.py:119-139)."""
    import time
    from .loader import header_values
    trace = os.environ.get("LFD_LOADER_TRACE") == "1"
    t_in = time.perf_counter()
    t_gpu = 0.0
    debug = params_bright.get("debug") or params_dim.get("debug")
    n = len(loaded.keys)
    rows = [None] * n
    by_slot = {}
    for i in range(n):
        if loaded.error[i] is not None:
            rows[i] = loaded.error[i]
        elif loaded.slot[i] >= 0:
            by_slot[loaded.slot[i]] = i
    # maximal runs of neighbouring slots with one filter
    slots = sorted(by_slot)
    runs, start = [], 0
    for j in range(1, len(slots) + 1):
        if j == len(slots) or slots[j] != slots[j - 1] + 1 or loaded.keys[by_slot[slots[j]]][2] != loaded.keys[by_slot[slots[start]]][2]:
            runs.append(slots[start:j])
            start = j
    h, w = loaded.shape if getattr(loaded, "shape", None) else (loaded.buffer.shape[1:] if loaded.buffer is not None else (0, 0))
    for run_slots in runs:
        idx = [by_slot[sl] for sl in run_slots]
        flt = loaded.keys[idx[0]][2]
        a, b = run_slots[0], run_slots[-1] + 1
        try:
            cats = loaded.cats
            m = max(1, int(cats["count"][a:b].max()))
            packed = {k: _np.ascontiguousarray(v[a:b, :m]) for k, v in cats.items() if k != "count"}
            packed["count"] = cats["count"][a:b]
            with use_context(h, w, inflight=min(256, len(idx))) as ctx:
                t_g = time.perf_counter()
                if loaded.device is not None:                 # decompressed on the GPU and still there
                    recs = ctx.detect_batch(loaded.device.slice(a, b), params_bright, params_dim, packed, _rs_struct(flt, params_removestars))
                else:
                    recs = ctx.detect_batch(loaded.buffer[a:b], params_bright, params_dim, packed, _rs_struct(flt, params_removestars), pinned=True)
                t_gpu += time.perf_counter() - t_g
            for i, rec in zip(idx, recs):
                rows[i] = rec
        except Exception:  # noqa: BLE001 - a call-level failure: every frame of the slice on its own, under its own try
            for sl, i in zip(run_slots, idx):
                try:
                    if loaded.device is not None:             # (the frames are wherever the failed call left them: no second source)
                        raise
                    img = loaded.buffer[sl].astype(_np.float32)
                    rows[i] = process_frame_arrays(img, loaded.cat_of(i), flt, params_bright, params_dim, params_removestars)[2]
                except Exception as e:  # noqa: BLE001
                    rows[i] = e
    for i in range(n):
        if rows[i] is None and loaded.array[i] is not None:      # not a plain float32 image of the chunk's shape, or an oversized catalogue
            try:
                rows[i] = process_frame_arrays(loaded.array[i], loaded.cat_of(i), loaded.keys[i][2], params_bright, params_dim,
                                               params_removestars)[2]
            except Exception as e:  # noqa: BLE001
                rows[i] = e
    for i, key in enumerate(loaded.keys):
        try:
            rec = rows[i]
            if isinstance(rec, Exception):
                raise rec
            status = int(rec["status"])
            if status == _native.ERR_NOLINES:
                raise TypeError("'NoneType' object is not subscriptable")
            if status:
                raise _native.NativeError(status, "frame failed on the device")
            if rec["found"]:
                shape = loaded.array[i].shape if loaded.array[i] is not None else (h, w)
                res = dictify_hough(shape, (_np.float32(rec["rho"]), _np.float32(rec["theta"])))
                head = " ".join(str(x) for x in (*key, *header_values(loaded.hdr[i], _HEADER_KEYS)))
                results.write(f"{head} {res['x1']} {res['y1']} {res['x2']} {res['y2']}\n")
        except Exception as e:  # noqa: BLE001
            _log_error(errors, key, e, debug)
    if trace:
        print("[loader]   process_loaded: %.1f ms in all, %.1f ms inside lfdmi_detect_batch_raw" %
              (1e3 * (time.perf_counter() - t_in), 1e3 * t_gpu), flush=True)


class DetectTrails:
    """Process a selection of SDSS frames.

    ``DetectTrails(run=2888)``, ``DetectTrails(run=2888, camcol=1, filter='i')``,
    ``DetectTrails(run=2888, camcol=1, filter='i', field=139).process()`` ... at least one of
    run / camcol / filter / field (or frame) must be given.  ``results`` / ``errors`` /
    ``savepath`` choose the output files (appended to), ``debug`` switches all three parameter
    dictionaries to debug mode, ``params_bright`` / ``params_dim`` / ``params_removestars``
    replace the defaults; the dictionaries can also be edited on the instance afterwards.
    """

    _FILTERS = ('u', 'g', 'r', 'i', 'z')
    _CAMCOLS = (1, 2, 3, 4, 5, 6)

    def __init__(self, **kwargs):
        save = kwargs.get("savepath", ".")
        self.kwargs = kwargs
        self.params_bright, self.params_dim, self.params_removestars = default_params()
        self.results = kwargs.get("results", os.path.join(save, "results.txt"))
        self.errors = kwargs.get("errors", os.path.join(save, "errors.txt"))
        for name in ("params_bright", "params_dim", "params_removestars"):
            if name in kwargs:
                setattr(self, name, kwargs[name])
        if "debug" in kwargs:
            self.debug = kwargs.pop("debug")
            for d in (self.params_bright, self.params_dim, self.params_removestars):
                d["debug"] = self.debug
        if any(d["debug"] for d in (self.params_removestars, self.params_bright, self.params_dim)):
            setup_debug()
        self._load()

    def _runInfo(self):
        rl = sdssfiles.runlist()
        w = _np.nonzero(rl["run"] == self._run)[0]
        if len(w) == 0:
            raise ValueError("Run %s not found in runList.par" % self._run)
        return int(rl["startfield"][w[0]]), int(rl["endfield"][w[0]])

    def _getRuns(self):
        return [int(r) for r in sdssfiles.runlist()["run"]]

    def _load(self):
        """Selection mode from the keywords given (detecttrails.py:290-342): run, run-camcol,
        run-filter, run-camcol-filter, camcol-filter, camcol-frame, field."""
        kw = self.kwargs
        self._run = self._camcol = self._field = 0
        self._filter = self._pick = "0"
        if "run" in kw:
            self._run, self._pick = kw["run"], "run"
        if "camcol" in kw:
            if kw["camcol"] not in self._CAMCOLS:
                raise ValueError("Nonexisting camcol")
            self._camcol, self._pick = kw["camcol"], "run-camcol"
        if "field" in kw or "frame" in kw:
            if self._camcol == 0:
                raise ValueError("send camcol= ")
            self._field = kw["field"] if "field" in kw else kw["frame"]
        if "filter" in kw:
            if kw["filter"] not in self._FILTERS:
                raise ValueError("Nonexistting filter")
            self._filter = kw["filter"]
            if self._camcol != 0:
                self._pick = "camcol-filter"
            if self._run != 0:
                self._pick = "run-filter"
            if self._camcol != 0 and self._run != 0:
                self._pick = "run-camcol-filter"
        elif self._field != 0 and self._camcol != 0:
            self._pick = "camcol-frame"
        if self._field != 0 and self._camcol != 0 and self._filter != "0":
            self._pick = "field"

    def _frames(self):
        """Yield (run, camcol, filter, field) in the reference's loop order (detecttrails.py:350-407)."""
        pick = self._pick
        if pick == "camcol-filter":
            for run in self._getRuns():
                self._run = run
                start, end = self._runInfo()
                for field in range(start, end):
                    yield run, self._camcol, self._filter, field
            self._run = 0
        elif pick == "run":
            start, end = self._runInfo()
            for camcol in self._CAMCOLS:
                for flt in self._FILTERS:
                    for field in range(start, end):
                        yield self._run, camcol, flt, field
        elif pick == "run-filter":
            start, end = self._runInfo()
            for camcol in self._CAMCOLS:
                for field in range(start, end):
                    yield self._run, camcol, self._filter, field
        elif pick == "run-camcol":
            start, end = self._runInfo()
            for flt in self._FILTERS:
                for field in range(start, end, 50):  # the reference samples every 50th field here
                    yield self._run, self._camcol, flt, field
        elif pick == "run-camcol-filter":
            start, end = self._runInfo()
            for field in range(start, end):
                yield self._run, self._camcol, self._filter, field
        elif pick == "camcol-frame":
            for flt in self._FILTERS:
                yield self._run, self._camcol, flt, self._field
        elif pick == "field":
            yield self._run, self._camcol, self._filter, self._field

    def process(self, batch=32, rank=None, world_size=None, loader_threads=None, resume=False):
        """Run the selection; results and errors files are opened in append mode.

        At most ``batch`` frames go to the GPU per call (same rows, same order as frame by frame; ``batch=1`` is the
        reference's frame-at-a-time loop).  A pool of ``loader_threads`` reader threads (default: one per core, at most 32;
        $LFD_LOADER_THREADS) reads the FITS / .fits.bz2 files of the next chunk straight into page-locked staging memory
        while the GPU works on the current one (``loader.FrameLoader``; chunks of min(batch, $LFD_LOADER_SLOTS = 64, or 256 for a selection of .fits.bz2 files) frames:
        two 0.8 GB staging buffers keep the link busy, larger ones only cost set-up time); the big-endian floats are swapped
        on the device.  With ``world_size`` > 1 (default: $RANK / $WORLD_SIZE, i.e. one process per GPU under torchrun) every
        rank processes one contiguous block of the selection (``lfd_amd.batch.shard_bounds``: ceil(n / world_size) frames
        each, the same rule the batch detector and bench.py use) and appends to ``<results>.rank<r>`` / ``<errors>.rank<r>``
        -- the replacement for splitting runs into PBS jobs (lfd/createjobs/createjobs.py:173-202).

        ``resume=True``: frames listed in ``<results>[.rank<r>].progress`` (appended to, chunk by chunk, after the chunk's rows
        and error entries have been flushed) are skipped, so a run that was interrupted continues where it stopped instead
        of appending its rows twice (the reference restarts a PBS job from its first frame).

        ``self.last_stats`` afterwards: frames, total seconds, set-up seconds (context + staging buffers) and the seconds
        after which every chunk was done."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        from ..batch import shard_range
        from .loader import FrameLoader
        t_start = time.perf_counter()
        rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
        suffix = f".rank{rank}" if world_size > 1 else ""
        keys = list(self._frames())
        n_selection = len(keys)
        if world_size > 1:
            a, b = shard_range(len(keys), rank, world_size)
            keys = keys[a:b]
        progress_path = self.results + suffix + ".progress"
        # first line of the progress file: what the marks below it belong to.  A resume only trusts marks written for the same
        # selection, shard and world size; anything else (an older run of another selection into the same savepath, a change
        # of world_size: the marks would be compared against a different shard) is refused, not silently applied.
        header = "# lfd-progress v1 pick=%s run=%s camcol=%s filter=%s field=%s rank=%d world_size=%d selection=%d" % (
            self._pick, self.kwargs.get("run", 0), self._camcol, self._filter, self._field, rank, world_size, n_selection)
        skipped = 0
        fresh = True
        if resume and os.path.exists(progress_path):
            with open(progress_path) as f:
                lines = [ln.strip() for ln in f if ln.strip()]
            if lines:
                if lines[0] != header:
                    raise ValueError("resume=True: %s was written for another selection / shard (%r, this run: %r); "
                                     "delete it or run with resume=False" % (progress_path, lines[0], header))
                done = {tuple(ln.split()) for ln in lines[1:]}
                before = len(keys)
                keys = [k for k in keys if tuple(str(x) for x in k) not in done]
                skipped = before - len(keys)
                fresh = False
        self.last_stats = {"frames": len(keys), "chunk_frames": 0, "setup_s": 0.0, "chunk_done_s": [], "seconds": 0.0,
                           "skipped_by_resume": skipped}
        # resume=False starts a new record of marks (an earlier run's marks must never make a later resume skip frames this
        # run did not process); rows and errors are appended to, as in the reference
        with open(self.results + suffix, "a") as results, open(self.errors + suffix, "a") as errors, \
                open(progress_path, "w" if fresh else "a") as progress:
            if fresh:
                progress.write(header + "\n")
                progress.flush()

            def mark(done_keys):                     # rows first, then the marks: a crash in between repeats a chunk, never loses one
                results.flush()
                errors.flush()
                progress.write("".join("%s %s %s %s\n" % tuple(k) for k in done_keys))
                progress.flush()

            if batch <= 1:
                for key in keys:
                    process_field(results, errors, *key, self.params_bright, self.params_dim, self.params_removestars)
                    mark([key])
                self.last_stats["seconds"] = time.perf_counter() - t_start
                return
            # a chunk = one GPU call: 64 frames keep the link and the GPU busy for plain files; a selection that exists only as
            # .fits.bz2 is decompressed on the GPU a chunk at a time, and that decoder wants thousands of 900 kB blocks at once
            # (~14 per frame): 256 frames per chunk
            if not keys:
                return
            first = sdssfiles.filename("frame", run=keys[0][0], camcol=keys[0][1], field=keys[0][3], filter=keys[0][2])
            compressed = not os.path.exists(first) and os.path.exists(first + ".bz2") and os.environ.get("LFD_BZ2_DEVICE", "1") != "0"
            slots = max(1, min(batch, int(os.environ.get("LFD_LOADER_SLOTS", 256 if compressed else 64)), len(keys)))
            chunks = [keys[i:i + slots] for i in range(0, len(keys), slots)]
            if not chunks:
                return
            shape = _frame_shape(keys)
            # chunks loading at once: one for plain files (the link is the limit), two for selections decompressed on the GPU (two
            # decoders side by side: one chunk's Huffman stage overlaps the other's inverse BWT, loader.FrameLoader)
            depth = max(1, min(int(os.environ.get("LFD_LOADER_DEPTH", 2 if compressed else 1)), len(chunks)))
            with use_context(*shape, inflight=slots) as ctx:
                loader = FrameLoader(ctx, shape, slots, loader_threads, depth=depth, expect_bz2=compressed)
            self.last_stats.update(chunk_frames=slots, setup_s=time.perf_counter() - t_start)
            try:
                trace = os.environ.get("LFD_LOADER_TRACE") == "1"
                from collections import deque
                with ThreadPoolExecutor(depth, thread_name_prefix="lfd-chunk") as coord:
                    pending, nxt_i = deque(), 0

                    def submit():
                        nonlocal nxt_i
                        j = nxt_i
                        nxt_i += 1
                        pending.append(coord.submit(loader.load, chunks[j], j % (depth + 1), chunks[j + 1] if j + 1 < len(chunks) else None, j))
                    for _ in range(depth):
                        submit()
                    for i, chunk in enumerate(chunks):
                        t0 = time.perf_counter()
                        loaded = pending.popleft().result()
                        t1 = time.perf_counter()
                        # (chunk i sits in buffer i % (depth + 1); the loads in flight fill the other `depth` buffers)
                        if nxt_i < len(chunks):
                            submit()
                        process_loaded(results, errors, loaded, self.params_bright, self.params_dim, self.params_removestars)
                        mark(chunk)
                        self.last_stats["chunk_done_s"].append(time.perf_counter() - t_start)
                        if trace:
                            print("[loader] chunk %d: waited %.1f ms for its files, GPU call + rows %.1f ms" %
                                  (i, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)), flush=True)
            finally:
                self.last_stats["bz2"] = dict(loader.bz2_stats)
                loader.close()
                self.last_stats["seconds"] = time.perf_counter() - t_start
