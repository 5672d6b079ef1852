#!/bin/bash
# Second set of SQ counters for the step's kernels (occupancy / stall attribution), separate --pmc passes, --kernel-trace only:
#   pass 3: SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
#   pass 4: SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
#   pass 5: GRBM_GUI_ACTIVE GRBM_COUNT
# Summary -> $O/sq_counters2.txt (per kernel: launches, counter sums per launch)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-sq2}
mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/p3 -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/p3.err || echo "pass 3 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p4 -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/p4.err || echo "pass 4 failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/p5 -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/p5.err || echo "pass 5 failed"
python3 - "$O" > $O/sq_counters2.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    c = acc[k]
    launches = max(n[k].values())
    print(k[:60], "launches", launches, " ".join("%s/launch=%.3g" % (name, c[name] / n[k][name]) for name in sorted(c)))
PY
rm -rf $O/p3 $O/p4 $O/p5
head -12 $O/sq_counters2.txt | cut -c1-500
