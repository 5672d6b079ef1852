"""results.txt row format and CCD-border snapping (reference: results/utils.py:185-210, results/event.py:280-353)."""
import pytest

from lfd_amd import results


def test_row_roundtrip():
    hdr = {"TAI": 4649973000.12, "CRPIX1": 1025.0, "CRPIX2": 745.0, "CRVAL1": 10.5, "CRVAL2": -1.25,
           "CD1_1": 1e-4, "CD1_2": 2e-5, "CD2_1": -2e-5, "CD2_2": 1e-4}
    row = results.format_result_row(2888, 1, "i", 139, hdr, {"x1": -1786, "y1": 3215, "x2": 3215, "y2": -1786})
    assert len(row.split(" ")) == 17
    p = results.parse_result_row(row)
    assert (p["run"], p["camcol"], p["filter"], p["field"]) == (2888, 1, "i", 139)
    assert p["tai"] == hdr["TAI"] and p["cd21"] == -2e-5 and (p["x1"], p["y2"]) == (-1786.0, -1786.0)
    with pytest.raises(ValueError):
        results.parse_result_row("1 2 r 3")


def test_snap2ccd_known_answers():
    # the reference's docstring example: the diagonal through both corners
    assert results.snap2ccd(-1000, -1000, 10000, 10000) == (0, 0, 2048.0, 2048.0)
    x1, y1, x2, y2 = results.snap2ccd(-3537, 1000, 3537, 1400)
    assert (x1, x2) == (0, 2048.0) and 1000 < y1 < y2 < 1400
    x1, y1, x2, y2 = results.snap2ccd(500, -3000, 900, 5000)      # steep: crosses y = 0 and y = 2048
    assert (y1, y2) == (0, 2048.0) and 500 < x1 < x2 < 900
    with pytest.raises(ValueError):
        results.snap2ccd(-100, -50, -10, -60)                      # never enters the CCD box
    with pytest.raises(ZeroDivisionError):
        results.snap2ccd(5, 0, 5, 10)                              # vertical: the reference divides by zero too


def test_read_results(tmp_path):
    p = tmp_path / "results.txt"
    p.write_text("94 1 r 100 4.5 1 2 3 4 5 6 7 8 -10 20 30 -40\n\n94 1 r 101 4.5 1 2 3 4 5 6 7 8 1 2 3 4\n")
    rows = results.read_results(p)
    assert [r["field"] for r in rows] == [100, 101] and rows[0]["y2"] == -40.0
