#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun; results under gpurun_out/):
#   kernel trace + stats of the default command, FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only), same for --workload lsst.
# The program follows `--` directly (no env / sh wrappers: the profiler's preloaded library has initialised the GPU by then).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
python3 bench.py > $O/bench_sdss.json 2> $O/bench_sdss.err
python3 bench.py --workload lsst > $O/bench_lsst.json 2> $O/bench_lsst.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sdss -- python3 bench.py --cpu-sample 0 --no-host-leg --no-secondary > $O/bench_sdss_under_rocprof.json 2> $O/kt_sdss.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lsst -- python3 bench.py --workload lsst --cpu-sample 0 --no-host-leg --no-secondary > $O/bench_lsst_under_rocprof.json 2> $O/kt_lsst.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_sdss -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/pmc_fetch_sdss.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_sdss -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/pmc_write_sdss.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_lsst -- python3 bench.py --workload lsst --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/pmc_fetch_lsst.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_lsst -- python3 bench.py --workload lsst --cpu-sample 0 --steps 2 --no-host-leg --no-secondary > /dev/null 2> $O/pmc_write_lsst.err
for w in sdss lsst; do
  python3 tools/make_traffic.py $(ls $O/pmc_fetch_$w/*/*counter_collection.csv) $(ls $O/pmc_write_$w/*/*counter_collection.csv) $O/traffic_$w.json $w > $O/traffic_$w.txt
  cp $(ls $O/kt_$w/*/*kernel_stats.csv) $O/kernel_stats_$w.csv
done
# keep what is committed small: drop the raw per-dispatch traces
rm -rf $O/kt_sdss $O/kt_lsst $O/pmc_fetch_sdss $O/pmc_write_sdss $O/pmc_fetch_lsst $O/pmc_write_lsst
ls -la $O
