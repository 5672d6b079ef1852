"""SDSS file naming for the two file types the detection path opens.

The reference resolves paths through a bundled copy of ``sdsspy`` (sdss/files.py:216-275 with
the templates of sdss/share/sdssFileTypes.par:43,75 and the rerun looked up in
$PHOTO_REDUX/runList.par, sdss/files.py:707-712).  Only 'frame' and 'photoObj' are needed here:

    frame     $BOSS_PHOTOOBJ/frames/$RERUN/$RUN/$CAMCOL/frame-$FILTER-$RUN6-$CAMCOL-$FIELD4.fits
    photoObj  $BOSS_PHOTOOBJ/$RERUN/$RUN/$CAMCOL/photoObj-$RUN6-$CAMCOL-$FIELD4.fits
"""
import os

import numpy as np

_runlist_cache = {}


def runlist(path=None, reload=False):
    """Structured array (run, rerun, startfield, endfield) parsed from runList.par (yanny RUNDATA rows)."""
    path = path or os.path.join(os.path.expandvars("$PHOTO_REDUX"), "runList.par")
    if path in _runlist_cache and not reload:
        return _runlist_cache[path]
    fields = None
    rows = []
    with open(path) as f:
        text = f.read()
    in_struct = False
    names = []
    for line in text.splitlines():
        s = line.split("#")[0].strip()
        if not s:
            continue
        if s.startswith("typedef struct"):
            in_struct, names = True, []
            continue
        if in_struct:
            if s.startswith("}"):
                in_struct = False
                fields = names
                continue
            tok = s.rstrip(";").split()
            if len(tok) >= 2:
                names.append(tok[1].split("[")[0])
            continue
        tok = s.split()
        if tok[0].upper() == "RUNDATA" and fields:
            rec = dict(zip(fields, tok[1:]))
            rows.append((int(rec["run"]), rec["rerun"].strip('"'), int(rec["startfield"]), int(rec["endfield"])))
    arr = np.array(rows, dtype=[("run", "i4"), ("rerun", "U16"), ("startfield", "i4"), ("endfield", "i4")])
    # the reference drops the duplicated bad entry of run 5194 (sdss/files.py:667-669)
    arr = arr[(arr["run"] != 5194) | (arr["rerun"] == "301")]
    _runlist_cache[path] = arr
    return arr


def find_rerun(run):
    rl = runlist()
    w = np.nonzero(rl["run"] == run)[0]
    if len(w) == 0:
        raise ValueError("Run %s not found in runList.par" % run)
    return str(rl["rerun"][w[0]])


def filename(ftype, run, camcol, field, filter=None, rerun=None):
    if rerun is None:
        rerun = find_rerun(run)
    root = os.path.expandvars("$BOSS_PHOTOOBJ")
    if ftype == "frame":
        if filter is None:
            raise ValueError("frame files need a filter")
        return os.path.join(root, "frames", str(rerun), str(run), str(camcol),
                            "frame-%s-%06d-%d-%04d.fits" % (filter, run, camcol, field))
    if ftype == "photoObj":
        return os.path.join(root, str(rerun), str(run), str(camcol),
                            "photoObj-%06d-%d-%04d.fits" % (run, camcol, field))
    raise ValueError("unsupported file type %r" % (ftype,))
