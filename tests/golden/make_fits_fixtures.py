"""Writes the FITS fixtures of tests/golden/fits/ with astropy.io.fits -- an implementation independent of
lfd_amd/detecttrails/fitslite.py, which until now had only ever read files written by its own writer.

Run in the build container with the conda interpreter (astropy 4.3.1 is not importable from the system Python):

    /opt/conda/bin/python3.9 tests/golden/make_fits_fixtures.py

Fixtures (layouts the reference reads through fitsio: detecttrails.py:113-114, removestars.py:96-104):
  frame_f32.fits   primary HDU, float32 image, the nine header cards process_field writes out (one with a FORTRAN
                   `D` exponent), a long string split over CONTINUE cards, COMMENT / HISTORY cards, two header blocks
  frame_u16.fits   uint16 image stored as int16 with BZERO = 32768
  frame_scaled.fits int16 image with BSCALE = 0.5, BZERO = 10
  photoobj.fits    binary table in HDU 1 shaped like an SDSS photoObj file: 5E / 5J / J / K / L / nA / D / I / B columns,
                   a variable-length (`PJ`) column BEFORE the columns the path reads (heap after the table, PCOUNT > 0),
                   an unsigned 16-bit column stored with TZERO, more than one header block
  expected.npz / expected.json   the arrays and header values that went in
"""
import json
import os

import numpy as np

# astropy 4.3.1 (the build container's) predates numpy 1.23-1.25's removal of these aliases; its units package,
# which astropy.io.fits pulls in for tables, only lists them
for _name, _fn in (("asscalar", lambda a: a.item()), ("alen", len), ("msort", lambda a: np.sort(a, axis=0))):
    if not hasattr(np, _name):
        setattr(np, _name, _fn)
from astropy.io import fits  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fits")


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(12345)
    out = {}
    meta = {}

    # ---- float32 frame
    img = rng.normal(0.0, 0.05, (37, 53)).astype(np.float32)
    img[10:14, 5:40] += 3.25
    hdu = fits.PrimaryHDU(img)
    h = hdu.header
    h["COMMENT"] = "synthetic frame for the fitslite reader tests"
    h.append(fits.Card.fromstring("TAI     =  4.649973000500D+09 / FORTRAN-style exponent"))
    vals = {"CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.4953219, "CRVAL2": -1.2502, "CD1_1": 1.1e-4, "CD1_2": -2.0e-5,
            "CD2_1": 2.0e-5, "CD2_2": 1.1e-4}
    for k, v in vals.items():
        h[k] = (v, "WCS")
    h["RUN"] = (94, "integer card")
    h["FILTER"] = ("r", "string card")
    h["BOOLCARD"] = (True, "logical card")
    long_text = "a long string value that does not fit on one card " * 3 + "it's quoted too"
    h["LONGSTR"] = long_text
    for i in range(30):
        h[f"FILL{i:03d}"] = (i * 1.5, "filler so that the header spans more than one block")
    h["HISTORY"] = "written by astropy"
    hdu.writeto(os.path.join(HERE, "frame_f32.fits"), overwrite=True)
    out["frame_f32"] = img
    meta["frame_f32"] = dict(vals, TAI=4.649973000500e9, RUN=94, FILTER="r", BOOLCARD=True, LONGSTR=long_text, FILL029=43.5)

    # ---- unsigned 16 bit (BZERO) and a scaled int16 image
    u16 = rng.integers(0, 65536, (21, 30), dtype=np.uint16)
    fits.PrimaryHDU(u16).writeto(os.path.join(HERE, "frame_u16.fits"), overwrite=True)
    out["frame_u16"] = u16.astype(np.float64)
    raw = rng.integers(-2000, 2000, (12, 17)).astype(np.int16)
    hdu = fits.PrimaryHDU(raw)
    hdu.header["BSCALE"] = 0.5
    hdu.header["BZERO"] = 10.0
    hdu.writeto(os.path.join(HERE, "frame_scaled.fits"), overwrite=True, output_verify="ignore")
    out["frame_scaled"] = raw.astype(np.float64) * 0.5 + 10.0

    # ---- photoObj-like binary table
    n = 23
    cols_in = {
        "OBJID": rng.integers(1 << 40, 1 << 41, n).astype(np.int64),
        "OBJC_TYPE": rng.integers(0, 7, n).astype(np.int32),
        "TYPE": rng.integers(0, 7, (n, 5)).astype(np.int32),
        "ROWC": rng.uniform(0, 1489, (n, 5)).astype(np.float32),
        "COLC": rng.uniform(0, 2048, (n, 5)).astype(np.float32),
        "PETROTH90": rng.uniform(-2, 15, (n, 5)).astype(np.float32),
        "PSFMAG": rng.uniform(14, 25, (n, 5)).astype(np.float32),
        "NOBSERVE": rng.integers(1, 4, n).astype(np.int32),
        "NDETECT": rng.integers(1, 4, n).astype(np.int32),
        "RA": rng.uniform(0, 360, n).astype(np.float64),
        "SMALLINT": rng.integers(-300, 300, n).astype(np.int16),
        "BYTECOL": rng.integers(0, 256, n).astype(np.uint8),
        "U16COL": rng.integers(0, 65536, n).astype(np.uint16),
        "GOOD": rng.integers(0, 2, n).astype(bool),
    }
    names = np.array([("obj%02d_%s" % (i, "x" * (i % 7))) for i in range(n)])
    var = np.empty(n, dtype=object)
    for i in range(n):
        var[i] = np.arange(i % 5, dtype=np.int32)
    columns = [
        fits.Column("OBJID", "K", array=cols_in["OBJID"]),
        fits.Column("VARLEN", "PJ()", array=var),                       # variable-length column before the wanted ones
        fits.Column("NAME", "19A", array=names),
        fits.Column("OBJC_TYPE", "J", array=cols_in["OBJC_TYPE"]),
        fits.Column("TYPE", "5J", array=cols_in["TYPE"]),
        fits.Column("GOOD", "L", array=cols_in["GOOD"]),
        fits.Column("ROWC", "5E", array=cols_in["ROWC"]),
        fits.Column("COLC", "5E", array=cols_in["COLC"]),
        fits.Column("RA", "D", array=cols_in["RA"]),
        fits.Column("SMALLINT", "I", array=cols_in["SMALLINT"]),
        fits.Column("BYTECOL", "B", array=cols_in["BYTECOL"]),
        fits.Column("U16COL", "I", bzero=32768, array=cols_in["U16COL"]),
        fits.Column("PETROTH90", "5E", array=cols_in["PETROTH90"]),
        fits.Column("PSFMAG", "5E", array=cols_in["PSFMAG"]),
        fits.Column("NOBSERVE", "J", array=cols_in["NOBSERVE"]),
        fits.Column("NDETECT", "J", array=cols_in["NDETECT"]),
    ]
    tb = fits.BinTableHDU.from_columns(columns)
    for i in range(20):
        tb.header[f"PAD{i:03d}"] = (i, "filler so that the extension header spans more than one block")
    fits.HDUList([fits.PrimaryHDU(), tb]).writeto(os.path.join(HERE, "photoobj.fits"), overwrite=True)
    for k, v in cols_in.items():
        out["tab_" + k] = v
    out["tab_NAME"] = names.astype("S19")
    meta["photoobj"] = {"nrows": n, "pcount_positive": int(tb.header["PCOUNT"]) > 0}

    np.savez(os.path.join(HERE, "expected.npz"), **out)
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
