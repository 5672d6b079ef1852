"""fitslite against FITS files written by an independent implementation (astropy.io.fits, through the committed script
tests/golden/make_fits_fixtures.py): the layouts the reference reads with fitsio (detecttrails.py:113-114,
removestars.py:96-104)."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(__file__), "golden", "fits")


def expected():
    with open(os.path.join(HERE, "expected.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "expected.npz")), meta


def test_float32_frame_and_header_cards():
    from lfd_amd.detecttrails import fitslite
    arrs, meta = expected()
    img, h = fitslite.read_image(os.path.join(HERE, "frame_f32.fits"))
    assert img.dtype == np.float32 and img.dtype.isnative and np.array_equal(img, arrs["frame_f32"])
    for k, v in meta["frame_f32"].items():
        assert h[k] == v, (k, h[k], v)
    assert isinstance(h["RUN"], int) and isinstance(h["TAI"], float) and h["BOOLCARD"] is True
    assert fitslite.read_header(os.path.join(HERE, "frame_f32.fits"))["CRPIX2"] == 745.0


def test_scaled_integer_images():
    from lfd_amd.detecttrails import fitslite
    arrs, _ = expected()
    u16, _ = fitslite.read_image(os.path.join(HERE, "frame_u16.fits"))
    assert np.array_equal(u16.astype(np.float64), arrs["frame_u16"])
    sc, _ = fitslite.read_image(os.path.join(HERE, "frame_scaled.fits"))
    assert np.array_equal(sc.astype(np.float64), arrs["frame_scaled"])


def test_photoobj_like_table():
    from lfd_amd.detecttrails import fitslite
    from lfd_amd.detecttrails.removestars import read_photoObj, read_photoObj_arrays
    arrs, meta = expected()
    path = os.path.join(HERE, "photoobj.fits")
    assert meta["photoobj"]["pcount_positive"]                     # the heap of the variable-length column is there
    t = read_photoObj_arrays(path)
    for name in ("OBJC_TYPE", "TYPE", "ROWC", "COLC", "PETROTH90", "PSFMAG", "NOBSERVE", "NDETECT"):
        assert t[name].dtype == arrs["tab_" + name].dtype and np.array_equal(t[name], arrs["tab_" + name]), name
    other = fitslite.read_table(path, ["OBJID", "RA", "SMALLINT", "BYTECOL", "U16COL", "GOOD", "NAME"])
    for name in ("OBJID", "RA", "SMALLINT", "BYTECOL"):
        assert np.array_equal(other[name], arrs["tab_" + name]), name
    assert np.array_equal(other["U16COL"].astype(np.int64), arrs["tab_U16COL"].astype(np.int64))       # TZERO applied
    assert np.array_equal(other["GOOD"].astype(bool), arrs["tab_GOOD"])                                 # 'T' / 'F' bytes
    names = [b"".join(row).rstrip(b" \0") for row in other["NAME"]]
    assert names == [bytes(x).rstrip(b" \0") for x in arrs["tab_NAME"]]
    rows, cols, mag, pet, objc, typ, nob, nde = read_photoObj(path)                                     # the reference's tuple layout
    assert len(rows) == meta["photoobj"]["nrows"] and set(rows[0]) == set("ugriz")
    import math
    assert rows[3]["r"] == math.ceil(float(arrs["tab_ROWC"][3, 2])) and mag[5]["z"] == math.ceil(float(arrs["tab_PSFMAG"][5, 4]))
    assert fitslite.read_header(path, ext=1)["TFIELDS"] == 16
