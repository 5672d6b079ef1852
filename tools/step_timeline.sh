#!/bin/bash
# One step's kernels in time order from a rocprofv3 --kernel-trace run of the bench (no per-launch HIP events: --no-kernel-timing):
# start / end / duration in microseconds from the step's first kernel, hardware queue, kernel, grid.  -> gpurun_out/<tag>_step_timeline.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-r04}
WL=${2:-sdss}          # sdss | lsst
MARK=${3:-k_rs_boxes}  # the kernel a step begins with, or `gap` (after the GPU idled > 60 us)
O=gpurun_out/${T}_tl
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --workload $WL --cpu-sample 0 --no-host-leg --no-secondary --steps 3 --warmup 2 --no-kernel-timing > $O/bench.json 2> $O/err.txt
python3 - "$O" "$MARK" "$WL" > gpurun_out/${T}_step_timeline.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
if sys.argv[2] == "gap":   # a step begins after the GPU was idle for more than 60 us (the call's return and re-entry)
    starts, last_end = [], 0
    for i, r in enumerate(rows):
        if i and r[0] - last_end > 60000:
            starts.append(i)
        last_end = max(last_end, r[1])
else:
    starts = [i for i, r in enumerate(rows) if r[3].startswith(sys.argv[2])]   # a step begins with this kernel (sdss: remove_stars' box kernel)
if len(starts) < 2:
    print("no complete step found"); sys.exit(0)
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
queues = {}
print("One 256-frame step of `bench.py --workload " + "%s" % sys.argv[3] + "` (the last but one of the run) under rocprofv3 --kernel-trace: start / end / duration in us from the step's")
print("first kernel, hardware queue (q0 = the launch stream in order of first use), kernel, grid (threads).  The next step's first kernel follows.")
print()
for r in rows[a:b + 1]:
    q = queues.setdefault(r[2], "q%d" % len(queues))
    print("%9.1f %9.1f %8.1f  %-3s %-48s %s x %s x %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, q, r[3][:48], r[4], r[5], r[6]))
PY
rm -rf $O
cat gpurun_out/${T}_step_timeline.txt | head -70
