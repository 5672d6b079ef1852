"""Host-side logic that needs no GPU: DetectTrails keyword handling and frame selection
(reference: detecttrails.py:199-267, :290-407), FITS / SDSS-path plumbing, results and errors
text formats, sharding arithmetic, synthetic-data determinism."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from lfd_amd import batch, synth
from lfd_amd.detecttrails import DetectTrails, default_params, detecttrails, fitslite, sdssfiles


def test_default_params_match_reference_defaults():
    pb, pd, prs = default_params()
    assert pb["dilateKernel"].shape == (4, 4) and pb["houghMethod"] == 20 and pb["dro"] == 25
    assert pb["contoursMode"] == 1 and pb["contoursMethod"] == 1 and pb["nlinesInSet"] == 3
    assert pd["erodeKernel"].shape == (3, 3) and pd["dilateKernel"].shape == (9, 9) and pd["dro"] == 20
    assert pd["minFlux"] == 0.02 and pd["addFlux"] == 0.5
    assert prs["filter_caps"] == {'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5}
    assert set(pb) == {"lwTresh", "thetaTresh", "dilateKernel", "contoursMode", "contoursMethod",
                       "minAreaRectMinLen", "houghMethod", "nlinesInSet", "lineSetTresh", "dro", "debug"}


@pytest.mark.parametrize("kw,pick", [
    (dict(run=94), "run"), (dict(run=94, camcol=1), "run-camcol"), (dict(run=94, filter="i"), "run-filter"),
    (dict(run=94, camcol=1, filter="i"), "run-camcol-filter"), (dict(camcol=1, filter="i"), "camcol-filter"),
    (dict(run=94, camcol=1, field=12), "camcol-frame"), (dict(run=94, camcol=1, filter="i", field=12), "field"),
    (dict(run=94, camcol=1, filter="i", frame=12), "field")])
def test_selection_modes(kw, pick):
    assert DetectTrails(**kw)._pick == pick


def test_bad_keywords_raise():
    with pytest.raises(ValueError):
        DetectTrails(run=1, camcol=7)
    with pytest.raises(ValueError):
        DetectTrails(run=1, filter="x")
    with pytest.raises(ValueError):
        DetectTrails(run=1, field=5)


def test_param_kwargs_land_in_their_own_dicts_and_debug_fans_out():
    _, pd, prs = default_params()
    pd["minFlux"] = 0.03
    prs["maxxy"] = 50
    d = DetectTrails(run=1, params_dim=pd, params_removestars=prs, debug=False)
    assert d.params_dim["minFlux"] == 0.03 and d.params_removestars["maxxy"] == 50
    assert d.params_bright["dro"] == 25          # not overwritten (reference bug C8 fixed)
    assert d.results == os.path.join(".", "results.txt")
    d2 = DetectTrails(run=1, savepath="/tmp/x", results="/tmp/r.txt")
    assert d2.results == "/tmp/r.txt" and d2.errors == "/tmp/x/errors.txt"


def _write_runlist(tmp_path):
    redux = tmp_path / "photo" / "redux"
    redux.mkdir(parents=True)
    (redux / "runList.par").write_text(
        "typedef struct {\n int run;\n char rerun[];\n int exist;\n int done;\n int calib;\n"
        " int startfield;\n int endfield;\n char machine[];\n char disk[];\n} RUNDATA;\n\n"
        "RUNDATA 94 301 1 1 1 100 104 m d\nRUNDATA 125 301 1 1 1 11 13 m d\n"
        "RUNDATA 5194 157 1 1 1 1 2 m d\nRUNDATA 5194 301 1 1 1 30 33 m d\n")
    os.environ["PHOTO_REDUX"] = str(redux)
    os.environ["BOSS_PHOTOOBJ"] = str(tmp_path / "photoObj")
    sdssfiles._runlist_cache.clear()


def test_runlist_and_filenames(tmp_path):
    _write_runlist(tmp_path)
    rl = sdssfiles.runlist()
    assert rl["run"].tolist() == [94, 125, 5194] and sdssfiles.find_rerun(5194) == "301"
    f = sdssfiles.filename("frame", 94, 3, 101, "r")
    assert f.endswith("photoObj/frames/301/94/3/frame-r-000094-3-0101.fits")
    p = sdssfiles.filename("photoObj", 94, 3, 101)
    assert p.endswith("photoObj/301/94/3/photoObj-000094-3-0101.fits")
    with pytest.raises(ValueError):
        sdssfiles.find_rerun(7)


def test_frame_iteration_order(tmp_path):
    _write_runlist(tmp_path)
    fr = list(DetectTrails(run=94, camcol=2, filter="g")._frames())
    assert fr == [(94, 2, "g", f) for f in range(100, 104)]
    fr = list(DetectTrails(run=94, camcol=2)._frames())
    assert fr == [(94, 2, flt, 100) for flt in "ugriz"]           # every 50th field (reference quirk C11)
    fr = list(DetectTrails(run=125)._frames())
    assert len(fr) == 6 * 5 * 2 and fr[0] == (125, 1, "u", 11) and fr[-1] == (125, 6, "z", 12)
    fr = list(DetectTrails(camcol=1, filter="z")._frames())
    assert [x[0] for x in fr] == [94] * 4 + [125] * 2 + [5194] * 3


def test_fits_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.normal(0, 1, (17, 23)).astype(np.float32)
    hdr = {"TAI": 4.5e9, "CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.5, "CRVAL2": -1.25,
           "CD1_1": 1e-4, "CD1_2": 2e-5, "CD2_1": -2e-5, "CD2_2": 1e-4}
    path = tmp_path / "frame.fits"
    fitslite.write_image(path, img, hdr)
    got, h = fitslite.read_image(path)
    assert got.dtype == np.float32 and np.array_equal(got, img)
    assert all(h[k] == v for k, v in hdr.items())
    cols = {"OBJC_TYPE": np.arange(5, dtype=np.int32), "ROWC": rng.random((5, 5)).astype(np.float32),
            "NOBSERVE": np.ones(5, np.int32), "PSFMAG": rng.random((5, 5)).astype(np.float32)}
    tpath = tmp_path / "t.fits"
    fitslite.write_table(tpath, cols)
    t = fitslite.read_table(tpath, ["ROWC", "NOBSERVE", "PSFMAG"])
    assert np.array_equal(t["ROWC"], cols["ROWC"]) and np.array_equal(t["NOBSERVE"], cols["NOBSERVE"])
    with pytest.raises(KeyError):
        fitslite.read_table(tpath, ["NDETECT"])
    import bz2
    (tmp_path / "frame.fits.bz2").write_bytes(bz2.compress(path.read_bytes()))
    got2, _ = fitslite.read_image(tmp_path / "frame.fits.bz2")
    assert np.array_equal(got2, img)


def test_process_field_row_and_error_formats(tmp_path, monkeypatch):
    _write_runlist(tmp_path)
    hdr = {"TAI": 4.5e9, "CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.5, "CRVAL2": -1.25,
           "CD1_1": 1e-4, "CD1_2": 2e-5, "CD2_1": -2e-5, "CD2_2": 1e-4}
    fpath = sdssfiles.filename("frame", 94, 1, 100, "r")
    os.makedirs(os.path.dirname(fpath))
    fitslite.write_image(fpath, np.zeros((8, 9), np.float32), hdr)
    ppath = sdssfiles.filename("photoObj", 94, 1, 100)
    os.makedirs(os.path.dirname(ppath))
    fitslite.write_table(ppath, {"OBJC_TYPE": np.zeros(1, np.int32), "TYPE": np.zeros((1, 5), np.int32),
                                 "ROWC": np.zeros((1, 5), np.float32), "COLC": np.zeros((1, 5), np.float32),
                                 "PETROTH90": np.zeros((1, 5), np.float32), "PSFMAG": np.zeros((1, 5), np.float32),
                                 "NOBSERVE": np.ones(1, np.int32), "NDETECT": np.ones(1, np.int32)})
    seen = {}

    def fake(img, cat, flt, pb, pd, prs):
        seen["shape"], seen["n"] = img.shape, len(cat["NOBSERVE"])
        return True, {"x1": 1, "y1": -2, "x2": 3, "y2": 4}, None

    monkeypatch.setattr(detecttrails, "process_frame_arrays", fake)
    pb, pd, prs = default_params()
    res, err = io.StringIO(), io.StringIO()
    detecttrails.process_field(res, err, 94, 1, "r", 100, pb, pd, prs)
    row = res.getvalue().split()
    assert len(row) == 17 and row[:4] == ["94", "1", "r", "100"] and row[-4:] == ["1", "-2", "3", "4"]
    assert float(row[4]) == 4.5e9 and float(row[6]) == 745.0 and err.getvalue() == ""
    assert seen == {"shape": (8, 9), "n": 1}
    detecttrails.process_field(res, err, 94, 1, "r", 101, pb, pd, prs)      # missing file -> errors entry
    e = err.getvalue()
    assert e.startswith("94 1 r 101\n") and "FileNotFoundError" in e and e.endswith("\n\n")


def test_shard_bounds():
    assert batch.shard_bounds(8192, 8) == [(1024 * g, 1024 * (g + 1)) for g in range(8)]
    assert batch.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert batch.shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert batch.shard_range(256, 0, 1) == (0, 256)


def test_synthetic_frames_are_deterministic():
    img, cat, truth = synth.make_frame(3, shape=(96, 128), n_star=12)
    img2, cat2, truth2 = synth.make_frame(3, shape=(96, 128), n_star=12)
    assert np.array_equal(img, img2) and truth == truth2 and np.array_equal(cat["PSFMAG"], cat2["PSFMAG"])
    with open(os.path.join(os.path.dirname(__file__), "golden", "synth_checksums.json")) as f:
        gold = json.load(f)
    for k in (0, 1):
        im, c, t = synth.make_frame(k)
        assert hashlib.sha256(im.tobytes()).hexdigest() == gold[str(k)]["image_sha256"]
        assert t["streak"] == gold[str(k)]["streak"] and len(c["NOBSERVE"]) == gold[str(k)]["n_obj"]
    c1 = synth.make_config1_frame()
    assert c1.dtype == np.uint8 and c1.shape == (1489, 2048) and set(np.unique(c1)) == {0, 200, 255}


def test_parallel_frame_generator_matches_the_serial_one():
    """bench.py / tools generate their frames with child processes (never forks of a possibly GPU-initialised caller);
    the frames and catalogues are those of synth.make_frame."""
    shape = (96, 128)
    frames, cats = synth.make_frames(5, 6, shape, workers=3)
    assert frames.shape == (6, 96, 128) and frames.dtype == np.float32 and len(cats) == 6
    for i in range(6):
        img, cat, _ = synth.make_frame(5 + i, shape)
        assert np.array_equal(frames[i], img)
        for k in cat:
            assert np.array_equal(cats[i][k], cat[k]), k
    one, cats1 = synth.make_frames(7, 1, shape, workers=8, with_catalog=False)
    assert np.array_equal(one[0], synth.make_frame(7, shape)[0]) and cats1 == [None]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lfd_synth_")] if os.path.isdir("/dev/shm") else True


def test_bench_finds_the_committed_traffic_counters():
    """bench.py prices `roofline.traffic` from profiles/r*_traffic*.json (PMC passes of the same command): the timing slots of
    the kernels that can dominate a step resolve to kernels of the committed files, for both workloads."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sdss = ("sdss", 256, 256, 1, [1489, 2048])
    lsst = ("lsst", 256, 256, 1, [4096, 4096])
    for name, cfg in (("k_dilate_canny", sdss), ("k_prep_hist", sdss), ("k_bits_erode", sdss), ("k_hough_vote", sdss),
                      ("k_prep_hist", lsst), ("k_morph(erode)", lsst), ("k_dilate_canny", lsst), ("k_hough_vote", lsst)):
        nbytes, src = bench.load_traffic(name, cfg)
        assert nbytes and nbytes > 1e6 and src.endswith(".json"), (name, cfg[0])
    # ... and the utilisation figures of `kernels{}` (tools/make_util.py): the newest file per workload was collected from the
    # kernels as they are now (it records the SHA-256 of lfd_amd/csrc's sources; VERDICT r03 item 3: no profile older than the
    # last kernel change is quoted by the bench line)
    import glob
    import hashlib
    import json
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "lfd_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "lfd_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(root, "lfd_amd", "csrc", "*.inc"))):
        if os.path.basename(f) in ("k_bz2.h", "bz2_core.h", "bz2dev.hip"):   # the bzip2 decoder: its own translation unit (as in tools/make_util.py)
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    for workload, slots in (("sdss", ("k_dilate_canny", "k_prep_hist", "k_frame_bg", "k_frame_fg", "k_hough_vote", "k_bits_erode", "k_scan_fused")),
                            ("lsst", ("k_dilate_canny", "k_prep_hist", "k_hough_vote", "k_morph(erode)"))):
        util, src, top = bench.load_util(workload)
        assert src and top and top["name"], workload
        doc = json.load(open(os.path.join(root, "profiles", src)))
        assert doc["csrc_sha256"] == h.hexdigest(), ("%s was collected before the last change under lfd_amd/csrc: run tools/collect_profiles.sh "
                                                     "on the GPU box and commit the new profiles" % src)
        for s_ in slots:
            u = util.get(s_)
            assert u and 0 < u["valu_pipe"] < 1 and 0 < u["waves_per_simd"] <= 8 and "hbm" in u, (workload, s_, u)



def test_resume_skips_frames_already_marked_done(tmp_path, monkeypatch):
    """process(resume=True) continues an interrupted run: the frames listed in <results>.progress are skipped, rows are not
    appended twice (frame-at-a-time loop; the chunked loop marks its chunks the same way)."""
    _write_runlist(tmp_path)
    monkeypatch.setattr(detecttrails.DetectTrails, "_runInfo", lambda self: (100, 106))
    calls = []

    def fake_field(results, errors, run, camcol, flt, field, pb, pd, prs):
        calls.append(field)
        if field == 103 and len(calls) == 4:
            raise KeyboardInterrupt                                   # the run is cut short inside its fourth frame
        results.write("%d %d %s %d row\n" % (run, camcol, flt, field))

    monkeypatch.setattr(detecttrails, "process_field", fake_field)
    dt = detecttrails.DetectTrails(run=94, camcol=1, filter="r", savepath=str(tmp_path))
    import pytest
    with pytest.raises(KeyboardInterrupt):
        dt.process(batch=1)
    marks = open(dt.results + ".progress").read().split("\n")
    assert marks[0].startswith("# lfd-progress v1 ") and "world_size=1" in marks[0]
    assert marks[1:4] == ["94 1 r 100", "94 1 r 101", "94 1 r 102"]
    dt.process(batch=1, resume=True)
    assert calls == [100, 101, 102, 103, 103, 104, 105] and dt.last_stats["skipped_by_resume"] == 3
    rows = [ln.split()[3] for ln in open(dt.results)]
    assert rows == ["100", "101", "102", "103", "104", "105"]
    dt.process(batch=1, resume=True)                                  # nothing left
    assert dt.last_stats["frames"] == 0 and len(calls) == 7


def test_progress_marks_of_an_earlier_run_never_leak_into_a_later_resume(tmp_path, monkeypatch):
    """a finished run, then an interrupted resume=False run, then resume=True must process exactly the frames the
    interrupted run did not reach (resume=False starts a new progress file); marks written for another shard layout or
    another selection are refused instead of being compared against the wrong frames."""
    import pytest
    _write_runlist(tmp_path)
    monkeypatch.setattr(detecttrails.DetectTrails, "_runInfo", lambda self: (100, 106))
    calls, stop_at = [], [None]

    def fake_field(results, errors, run, camcol, flt, field, pb, pd, prs):
        if stop_at[0] is not None and field == stop_at[0]:
            raise KeyboardInterrupt
        calls.append(field)
        results.write("%d %d %s %d row\n" % (run, camcol, flt, field))

    monkeypatch.setattr(detecttrails, "process_field", fake_field)
    dt = detecttrails.DetectTrails(run=94, camcol=1, filter="r", savepath=str(tmp_path))
    dt.process(batch=1)                                               # a full run: six marks
    assert calls == [100, 101, 102, 103, 104, 105]
    del calls[:]
    stop_at[0] = 102
    with pytest.raises(KeyboardInterrupt):
        dt.process(batch=1)                                           # a second run from scratch, cut short after two frames
    assert calls == [100, 101]
    stop_at[0] = None
    dt.process(batch=1, resume=True)                                  # ... continues with the other four, not with nothing
    assert calls == [100, 101, 102, 103, 104, 105] and dt.last_stats["skipped_by_resume"] == 2
    # another shard layout / another selection: refused
    with pytest.raises(ValueError, match="another selection"):
        detecttrails.DetectTrails(run=94, camcol=1, filter="g", savepath=str(tmp_path)).process(batch=1, resume=True)
    # the same selection sharded over two ranks keeps its own files (.rank0 / .rank1) and headers
    dt.process(batch=1, rank=0, world_size=2)
    assert "world_size=2" in open(dt.results + ".rank0.progress").readline()
    os.rename(dt.results + ".rank0.progress", dt.results + ".rank1.progress")
    with pytest.raises(ValueError, match="another selection"):
        dt.process(batch=1, rank=1, world_size=2, resume=True)        # rank 0's marks are not rank 1's


def test_jobs_merges_the_workers_files_in_selection_order(tmp_path):
    """lfd_amd.jobs: the node-level replacement of the reference's PBS fan-out (createjobs/createjobs.py:173-202): one worker per
    GPU over contiguous blocks of the selection, their files joined in rank order = selection order."""
    from lfd_amd import jobs
    res = tmp_path / "results.txt"
    (tmp_path / "results.txt.rank0").write_text("94 1 r 100 a\n94 1 r 101 b\n")
    (tmp_path / "results.txt.rank2").write_text("94 1 r 105 c\n")          # (rank 1 found nothing: no file)
    n = jobs.merge_rank_files(str(res), 3, remove=True)
    assert res.read_text() == "94 1 r 100 a\n94 1 r 101 b\n94 1 r 105 c\n" and n == len(res.read_text())
    assert not (tmp_path / "results.txt.rank0").exists() and not (tmp_path / "results.txt.rank2").exists()
    assert jobs.merge_rank_files(str(res), 1) == 0
    j = jobs.Jobs(3, devices=[4, 5, 6], run=94, camcol=1)
    cmds = j.commands("/tmp/spec")
    assert [c[1]["RANK"] for c in cmds] == ["0", "1", "2"] and [c[1]["LFD_DEVICE"] for c in cmds] == ["4", "5", "6"]
    assert all(c[1]["WORLD_SIZE"] == "3" and c[1]["LOCAL_WORLD_SIZE"] == "3" and c[0][-2:] == ["--worker", "/tmp/spec"] for c in cmds)
    assert jobs._parse_value("94") == 94 and jobs._parse_value("r") == "r" and jobs._parse_value("94,125") == [94, 125]
    import pytest
    with pytest.raises(ValueError):
        jobs.Jobs(2, devices=[0])
