#!/bin/bash
# The shared bzip2 block core (lfd_amd/csrc/bz2_core.h: header, tables, symbols) under AddressSanitizer + UBSan on the CPU: valid files
# at levels 1 and 9 and 400 damaged ones (bit flips, bursts, truncations, header bytes).  The device kernels cannot run under a
# sanitizer on this pool; the header / table code they share with this build can.  usage: tools/bz2_core_sanitize.sh [out.txt]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/dev/stdout}
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $T/chk tools/bz2_core_check.cpp
python3 - "$T" > $T/log.txt 2>&1 <<'PY'
import bz2, os, subprocess, sys
import numpy as np
T = sys.argv[1]
rng = np.random.default_rng(5)
plains = [rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(), bytes(rng.integers(0, 4, 200000, dtype=np.uint8)),
          b"".join(bytes([b]) * int(n) for b, n in zip(rng.integers(0, 256, 3000), rng.integers(1, 300, 3000))), b"z", bytes(1200000)]
good = bad = agree = 0
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
def run(blob):
    open(T + "/x.bz2", "wb").write(blob)
    r = subprocess.run([T + "/chk", T + "/x.bz2", T + "/x.out"], capture_output=True, text=True, env=env)
    san = "Sanitizer" in r.stderr or "runtime error" in r.stderr
    return r.returncode, san, r.stderr[-400:]
for lvl in (1, 9):
    for p in plains:
        rc, san, err = run(bz2.compress(p, lvl))
        assert rc == 0 and not san and open(T + "/x.out", "rb").read() == p, (lvl, len(p), rc, err)
        good += 1
base = [bz2.compress(p, l) for p, l in zip(plains[:3], (1, 9, 9))]
for k in range(400):
    b = bytearray(base[k % 3])
    kind = k % 4
    if kind == 0:
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        i = int(rng.integers(0, len(b) - 16)); b[i:i + 16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
    elif kind == 2:
        b = b[:int(rng.integers(8, len(b)))]
    else:
        b[int(rng.integers(4, min(300, len(b))))] = int(rng.integers(0, 256))
    rc, san, err = run(bytes(b))
    assert not san, (k, err)
    try:
        want = bz2.decompress(bytes(b))
    except (OSError, ValueError, EOFError):
        want = None
    if rc == 0:
        assert want is not None and open(T + "/x.out", "rb").read() == want, k
        agree += 1
    else:
        bad += 1
print("valid files decoded and equal: %d; damaged files: %d declined, %d decoded and equal to Python's bz2; no sanitizer report" % (good, bad, agree))
PY
tail -1 $T/log.txt > $OUT
cat $OUT
rm -rf $T
