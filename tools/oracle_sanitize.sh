#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on this pool; the oracle is the only CPU
# code that walks images with hand-written index arithmetic).  Builds oracle/_asan/liblfd_oracle_asan.so and runs the oracle
# test modules against it (LFD_ORACLE_LIB points the Python wrapper at the instrumented library).
set -e
cd "$(dirname "$0")/.."
mkdir -p oracle/_asan
gcc -O1 -g -std=c99 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -o oracle/_asan/liblfd_oracle_asan.so oracle/lfd_oracle.c -lm
ASAN_LIB=$(gcc -print-file-name=libasan.so)
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LFD_ORACLE_LIB="$PWD/oracle/_asan/liblfd_oracle_asan.so" \
  python -m pytest tests/test_oracle_ops.py tests/test_oracle_tail.py tests/test_contour_equivalence.py -q -x -p no:cacheprovider
# ... and the whole per-frame path (remove_stars -> flip -> bright -> dim) on the portable golden frames
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LFD_ORACLE_LIB="$PWD/oracle/_asan/liblfd_oracle_asan.so" python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from lfd_amd import synth
from lfd_amd.detecttrails import default_params
from oracle import lfd_oracle as O
pb, pd, prs = default_params()
rs = O.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
cases = json.load(open("tests/golden/pipeline_golden.json"))["cases"]
ok = 0
for c in cases:
    img, cat, _ = synth.make_portable_frame(c["k"], tuple(c["shape"]))
    rec = O.detect_frame(img, pb, pd, cat, rs)
    ok += all(rec[k] == v or k in ("rho", "theta") for k, v in c["record"].items())
print("whole path under ASan + UBSan: %d / %d golden records reproduced, no sanitizer report" % (ok, len(cases)))
PY
