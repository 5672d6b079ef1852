#!/bin/bash
# The native FITS readers under ASan + UBSan on well-formed, truncated and hostile files (header offsets rounded
# past the end of a truncated file, 64-bit products of header sizes).  CPU only.
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -o $D/fits_asan tools/fits_reader_asan.cpp
python - "$D" <<'PY'
import sys, os
sys.path.insert(0, ".")
from lfd_amd import synth
from lfd_amd.detecttrails import loader, sdssfiles
D = sys.argv[1]
os.environ["BOSS"] = D
frames, cats = [], []
for k in range(2):
    img, cat, _ = synth.make_portable_frame(k, (64, 96), n_star=7)
    frames.append(img); cats.append(cat)
synth.write_boss_tree(D, frames, cats, field0=100)
p = sdssfiles.filename("photoObj", 94, 1, 100)
data = open(p, "rb").read()
e0 = loader.header_end(data); e1 = e0 + loader.header_end(data[e0:])
cases = {"ok_table": data, "cut_in_table_header": data[:e1 - 1000], "cut_in_primary_header": data[:e0 - 100], "cut_in_rows": data[:e1 + 10], "empty": b""}
t = bytearray(data); k = data.index(b"NAXIS1  =", e0); t[k:k + 30] = b"NAXIS1  =  9223372036854775807"; cases["row_bytes_wraps"] = bytes(t)
t = bytearray(data); k = data.index(b"NAXIS2  =", e0); t[k:k + 30] = b"NAXIS2  =  4611686018427387904"; cases["nrows_wraps"] = bytes(t)
t = bytearray(data); k = data.index(b"TFORM1  =", e0); t[k:k + 40] = (b"TFORM1  = '99999999999999999999E'" + b" " * 40)[:40]; cases["repeat_overflows"] = bytes(t)
f = open(sdssfiles.filename("frame", 94, 1, 100, "r"), "rb").read()
cases.update({"ok_frame": f, "frame_cut_in_header": f[:1000], "frame_cut_in_data": f[:len(f) // 2]})
for name, blob in cases.items():
    open(os.path.join(D, name + ".fits"), "wb").write(blob)
PY
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $D/fits_asan $D/*.fits
echo "fits readers under ASan + UBSan: no sanitizer report"
rm -rf $D
