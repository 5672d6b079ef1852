#!/bin/bash
# SQ counters of the device bzip2 decoder's kernels (tools/bz2_probe.py, 256 frames): instructions by type, busy / wave cycles,
# waiting shares; separate --pmc passes, --kernel-trace only.  Summary -> gpurun_out/<tag>/bz2_counters.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-bz2c}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 tools/bz2_probe.py 256 > $O/p0.out 2> $O/p0.err || echo "pass 0 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $O/p1 -- python3 tools/bz2_probe.py 256 > /dev/null 2> $O/p1.err || echo "pass 1 failed"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -- python3 tools/bz2_probe.py 256 > /dev/null 2> $O/p2.err || echo "pass 2 failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU --output-format csv -d $O/p3 -- python3 tools/bz2_probe.py 256 > /dev/null 2> $O/p3.err || echo "pass 3 failed"
python3 - "$O" > $O/bz2_counters.txt <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p[123]/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    if "bz2" not in k:
        continue
    c = acc[k]
    print(k[:40], " ".join("%s=%.4g" % (name, c[name] / n[k][name]) for name in sorted(c)))
for f in glob.glob(O + "/p0/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "bz2" in r["Name"]:
            print("stats", r["Name"].split("(")[0][:40], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/p0 $O/p1 $O/p2 $O/p3
cat $O/bz2_counters.txt | cut -c1-600
