#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun; results under gpurun_out/<tag>/):
#   bench lines, kernel trace + stats of the profiled command, FETCH_SIZE / WRITE_SIZE passes, four SQ / GRBM counter passes
#   (all separate, --kernel-trace only), for the sdss and the lsst workload; tools/make_traffic.py and tools/make_util.py turn them
#   into <tag>_traffic_<w>.json and <tag>_util_<w>.json; with `both`, tools/bz2_counters.sh follows (-> gpurun_out/<tag>bz2/).  The program follows `--` directly (no env / sh wrappers: the profiler's
#   preloaded library has initialised the GPU by then).  usage: collect_profiles.sh <tag> [sdss|lsst|both]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r04}
WHICH=${2:-both}
O=gpurun_out/$R
mkdir -p $O
CMD="python3 bench.py --cpu-sample 0 --no-host-leg --no-secondary"
for w in sdss lsst; do
  if [ "$WHICH" != both ] && [ "$WHICH" != $w ]; then continue; fi
  WF=""; if [ $w = lsst ]; then WF="--workload lsst"; fi
  if [ $w = sdss ]; then python3 bench.py > $O/bench_sdss.json 2> $O/bench_sdss.err; else python3 bench.py --workload lsst > $O/bench_lsst.json 2> $O/bench_lsst.err; fi
  echo "bench $w done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- $CMD $WF > $O/bench_${w}_under_rocprof.json 2> $O/kt_$w.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_fetch_$w.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_write_$w.err
  echo "traffic passes $w done"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_a_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_a_$w.err
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_b_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_b_$w.err
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_c_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_c_$w.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d_$w -- $CMD $WF --steps 2 > /dev/null 2> $O/pmc_d_$w.err
  echo "counter passes $w done"
  python3 tools/make_traffic.py $(ls $O/pmc_fetch_$w/*/*counter_collection.csv) $(ls $O/pmc_write_$w/*/*counter_collection.csv) $O/traffic_$w.json $w > $O/traffic_$w.txt
  cp $(ls $O/kt_$w/*/*kernel_stats.csv) $O/kernel_stats_$w.csv
  python3 tools/make_util.py $O/util_$w.json $w $O/kernel_stats_$w.csv $O/traffic_$w.json $(ls $O/pmc_[abcd]_$w/*/*counter_collection.csv) > $O/util_$w.txt
  # keep what is committed small: drop the raw per-dispatch traces
  rm -rf $O/kt_$w $O/pmc_fetch_$w $O/pmc_write_$w $O/pmc_a_$w $O/pmc_b_$w $O/pmc_c_$w $O/pmc_d_$w
done
# the device bzip2 decoder's kernels (its own translation unit): rocprofv3 averages + SQ counters -> gpurun_out/${R}bz2/
if [ "$WHICH" = both ]; then bash tools/bz2_counters.sh ${R}bz2 > $O/bz2_counters.log 2>&1 || echo "bz2 counters failed"; fi
ls -la $O
