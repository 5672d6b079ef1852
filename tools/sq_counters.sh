#!/bin/bash
# SQ counters of the step's kernels (separate --pmc passes, --kernel-trace only; the program follows `--` directly):
#   pass 1: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS     pass 2: SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
# Summary (per kernel: launches, counter sums per launch) -> $O/sq_counters.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-sq}
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p1 -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/p2.err
python3 - "$O" > $O/sq_counters.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    c = acc[k]
    launches = max(n[k].values())
    print(k[:60], "launches", launches, " ".join("%s/launch=%.3g" % (name, c[name] / n[k][name]) for name in sorted(c)))
PY
rm -rf $O/p1 $O/p2
cat $O/sq_counters.txt | head -30
