"""The drop-in driver end to end on the GPU: a synthetic $BOSS tree of FITS frames and photoObj
tables -> DetectTrails(...).process() -> results.txt rows equal to the CPU oracle's, for the
per-frame and the batched code paths (reference flow: detecttrails.py:30-143, :344-407)."""
import bz2
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HDR = {"TAI": 4649973000.5, "CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.5, "CRVAL2": -1.25,
       "CD1_1": 1e-4, "CD1_2": 2e-5, "CD2_1": -2e-5, "CD2_2": 1e-4}


def build_tree(root, fields, shape=(512, 768)):
    from lfd_amd import synth
    from lfd_amd.detecttrails import fitslite, sdssfiles
    redux = root / "photo" / "redux"
    redux.mkdir(parents=True)
    (redux / "runList.par").write_text(
        "typedef struct {\n int run;\n char rerun[];\n int exist;\n int done;\n int calib;\n int startfield;\n"
        " int endfield;\n char machine[];\n char disk[];\n} RUNDATA;\n\n"
        f"RUNDATA 94 301 1 1 1 {fields[0]} {fields[-1] + 1} m d\n")
    os.environ["PHOTO_REDUX"] = str(redux)
    os.environ["BOSS_PHOTOOBJ"] = str(root / "photoObj")
    sdssfiles._runlist_cache.clear()
    truth = {}
    for i, field in enumerate(fields):
        img, cat, t = synth.make_portable_frame(field, shape)
        fpath = sdssfiles.filename("frame", 94, 1, field, "r")
        os.makedirs(os.path.dirname(fpath), exist_ok=True)
        if i == 1:      # one frame only as .bz2 (detecttrails.py:81-109)
            tmp = fpath + ".tmp"
            fitslite.write_image(tmp, img, HDR)
            with open(tmp, "rb") as f, open(fpath + ".bz2", "wb") as g:
                g.write(bz2.compress(f.read()))
            os.remove(tmp)
        else:
            fitslite.write_image(fpath, img, HDR)
        if i != 2:      # one field has no catalogue -> an errors.txt entry
            ppath = sdssfiles.filename("photoObj", 94, 1, field)
            os.makedirs(os.path.dirname(ppath), exist_ok=True)
            cols = dict(cat)
            cols["OBJC_TYPE"] = np.zeros(len(cat["NOBSERVE"]), np.int32)
            cols["TYPE"] = np.zeros((len(cat["NOBSERVE"]), 5), np.int32)
            fitslite.write_table(ppath, cols)
        truth[field] = (img, cat)
    return truth


def expected_rows(oracle, truth, skip):
    from lfd_amd import results
    from lfd_amd.detecttrails import default_params
    pb, pd, prs = default_params()
    rs = oracle.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    rows = []
    for field, (img, cat) in truth.items():
        if field in skip:
            continue
        rec = oracle.detect_frame(img.copy(), pb, pd, cat, rs)
        if rec["found"]:
            rows.append(results.format_result_row(94, 1, "r", field, HDR, rec))
    return rows


@pytest.mark.parametrize("batch", [1, 4])
def test_detecttrails_process(tmp_path, oracle, batch):
    from lfd_amd.detecttrails import DetectTrails
    fields = list(range(0, 7))
    truth = build_tree(tmp_path, fields)
    want = expected_rows(oracle, truth, skip={fields[2]})
    assert len(want) >= 2
    dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(tmp_path))
    dt.process(batch=batch)
    got = [l.strip() for l in open(dt.results) if l.strip()]
    assert got == want
    err = open(dt.errors).read()
    assert err.count("\n\n") == 1 and err.startswith(f"94 1 r {fields[2]}\n") and "FileNotFoundError" in err


def test_rank_sharding_of_the_driver(tmp_path, oracle):
    """world_size = 2 in one process: rank r takes the r-th contiguous block of the selection (batch.shard_bounds, the
    rule the batch detector and bench.py use); the two ranks' files, concatenated, are the single-process rows in order."""
    from lfd_amd.detecttrails import DetectTrails
    fields = list(range(0, 7))
    truth = build_tree(tmp_path, fields)
    want = expected_rows(oracle, truth, skip={fields[2]})
    for r in (0, 1):
        DetectTrails(run=94, camcol=1, filter="r", savepath=str(tmp_path)).process(batch=2, rank=r, world_size=2)
    got = []
    for r in (0, 1):
        rows = [l.strip() for l in open(tmp_path / f"results.txt.rank{r}") if l.strip()]
        lo, hi = (0, 4) if r == 0 else (4, 7)                  # ceil(7 / 2) = 4 frames for rank 0
        assert all(lo <= int(row.split()[3]) < hi for row in rows), (r, rows)
        got += rows
    assert got == want


def test_one_bad_frame_costs_only_itself_in_a_batch(tmp_path, oracle):
    """A NaN in one photoObj catalogue (math.ceil raises in the reference: that frame's error, detecttrails.py:133-139)
    inside a batch of good frames: rows and errors entries equal the frame-at-a-time run."""
    from lfd_amd.detecttrails import DetectTrails, fitslite, sdssfiles
    fields = list(range(0, 6))
    truth = build_tree(tmp_path, fields)
    bad = fields[4]
    ppath = sdssfiles.filename("photoObj", 94, 1, bad)
    cols = dict(truth[bad][1])
    cols["ROWC"] = cols["ROWC"].copy()
    cols["ROWC"][3, 2] = np.nan
    cols["OBJC_TYPE"] = np.zeros(len(cols["NOBSERVE"]), np.int32)
    cols["TYPE"] = np.zeros((len(cols["NOBSERVE"]), 5), np.int32)
    fitslite.write_table(ppath, cols)
    want = expected_rows(oracle, truth, skip={fields[2], bad})
    outs = {}
    for batch in (1, 6):
        sub = tmp_path / f"b{batch}"
        sub.mkdir()
        dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(sub))
        dt.process(batch=batch)
        rows = [l.strip() for l in open(dt.results) if l.strip()]
        err = open(dt.errors).read()
        heads = [blk.splitlines()[0] for blk in err.split("\n\n") if blk.strip()]
        tails = [blk.strip().splitlines()[-1] for blk in err.split("\n\n") if blk.strip()]
        outs[batch] = (rows, heads, tails)
    assert outs[1] == outs[6]
    assert outs[6][0] == want
    assert outs[6][1] == [f"94 1 r {fields[2]}", f"94 1 r {bad}"]
    assert outs[6][2][1] == "cannot convert float NaN to integer"


def test_mixed_filters_in_one_chunk_and_resume(tmp_path, oracle):
    """camcol-frame selection: one field in all five filters (detecttrails.py:400-403).  remove_stars' magnitude cap depends on
    the filter, so the loader gives the frames of a filter neighbouring slots and the GPU gets one call per filter; rows come
    out in the reference's order (u g r i z) and equal the oracle's with that filter's parameters.  Then the same selection
    with resume=True appends nothing."""
    from lfd_amd import results, synth
    from lfd_amd.detecttrails import DetectTrails, default_params, sdssfiles
    pb, pd, prs = default_params()
    img, cat, _ = synth.make_portable_frame(0, (512, 768))                   # a bright streak; which filters keep it depends on their caps
    img2, cat2, _ = synth.make_portable_frame(1, (512, 768))
    hdr = synth.write_boss_tree(tmp_path, [img, img2], [cat, cat2], field0=100, filter="r")
    for flt in "ugiz":                                                       # the same pixels under the other filters' names
        for f in (100, 101):
            os.link(sdssfiles.filename("frame", 94, 1, f, "r"), sdssfiles.filename("frame", 94, 1, f, flt))
    want = []
    for flt in "ugriz":
        rs = oracle.rs_params(flt, **{k: v for k, v in prs.items() if k != "debug"})
        rec = oracle.detect_frame(img.copy(), pb, pd, cat, rs)
        if rec["found"]:
            want.append(results.format_result_row(94, 1, flt, 100, hdr, rec))
    assert 1 <= len(want) <= 4                                              # (the filters' magnitude caps blot different objects)
    dt = DetectTrails(run=94, camcol=1, field=100, savepath=str(tmp_path))
    assert dt._pick == "camcol-frame"
    dt.process(batch=4)                                                      # two chunks: u g r i | z
    got = [ln.strip() for ln in open(dt.results) if ln.strip()]
    assert got == want and open(dt.errors).read() == ""
    dt.process(batch=4, resume=True)
    assert dt.last_stats["skipped_by_resume"] == 5 and [ln.strip() for ln in open(dt.results) if ln.strip()] == want


def test_jobs_two_workers_equal_one_process(tmp_path, oracle):
    """lfd_amd.jobs.Jobs: two worker processes (both on the test box's one GPU) over one selection -- a mix of plain and
    compressed frames, one field without a catalogue -- leave the results.txt / errors.txt a single process leaves."""
    from lfd_amd.detecttrails import DetectTrails
    from lfd_amd.jobs import Jobs
    fields = list(range(0, 9))
    truth = build_tree(tmp_path, fields)
    want = expected_rows(oracle, truth, skip={fields[2]})
    one = tmp_path / "one"
    two = tmp_path / "two"
    one.mkdir()
    two.mkdir()
    dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(one))
    dt.process(batch=4)
    results, errors = Jobs(2, devices=[0, 0], run=94, camcol=1, filter="r", savepath=str(two)).launch(batch=4, timeout=600)
    got = [l.strip() for l in open(results) if l.strip()]
    assert got == want == [l.strip() for l in open(dt.results) if l.strip()]
    assert open(errors).read() == open(dt.errors).read() != ""
    assert not os.path.exists(results + ".rank0") and not os.path.exists(results + ".rank1")


def test_jobs_resume_continues_where_a_launch_stopped(tmp_path, oracle):
    """A launch that died after worker 0 had finished (its results.rank0 and progress file are there, worker 1 never ran):
    ``launch(resume=True)`` lets worker 0 skip its block, runs worker 1's, and the joined file is the complete one."""
    from lfd_amd.detecttrails import DetectTrails
    from lfd_amd.jobs import Jobs
    fields = list(range(0, 9))
    truth = build_tree(tmp_path, fields)
    want = expected_rows(oracle, truth, skip={fields[2]})
    out = tmp_path / "out"
    out.mkdir()
    dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(out))
    dt.process(batch=4, rank=0, world_size=2)                          # what worker 0 of the interrupted launch left behind
    assert os.path.exists(dt.results + ".rank0") and not os.path.exists(dt.results + ".rank1")
    results, errors = Jobs(2, devices=[0, 0], run=94, camcol=1, filter="r", savepath=str(out)).launch(batch=4, resume=True, timeout=600)
    assert [l.strip() for l in open(results) if l.strip()] == want
    assert open(errors).read().count("\n\n") == 1
