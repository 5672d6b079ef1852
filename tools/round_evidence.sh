#!/bin/bash
# The round's evidence beside tools/collect_profiles.sh (bench lines, rocprofv3 stats, PMC traffic), one gpurun call:
# GPU suite, SQ counters, tile-kernel stage clocks, per-frame kernel phase clocks, four-rank rehearsal, drop-in probe,
# exhaustive trig check, bulk parity.  Results under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
T=${1:-r03}
O=gpurun_out
python -m pytest tests -m gpu -q > $O/${T}_gpu_tests.log 2>&1; tail -2 $O/${T}_gpu_tests.log
bash tools/sq_counters.sh ${T}_sq > /dev/null 2>&1; cp $O/${T}_sq/sq_counters.txt $O/${T}_sq_counters.txt; head -3 $O/${T}_sq_counters.txt | cut -c1-200
python tools/dc_profile.py > $O/${T}_dc_stage_clocks.txt 2>/dev/null; cat $O/${T}_dc_stage_clocks.txt
python tools/frame_profile.py > $O/${T}_frame_profile.txt 2>/dev/null; cat $O/${T}_frame_profile.txt
bash tools/rehearse_ranks.sh 4 > $O/${T}_four_ranks.json 2> $O/${T}_four_ranks.err; tail -c 400 $O/${T}_four_ranks.json; echo
python tools/dropin_probe.py 2048 64 > $O/${T}_dropin_probe.txt 2>&1; tail -3 $O/${T}_dropin_probe.txt
python tools/trig_exhaustive.py 2048 > $O/${T}_trig_exhaustive.txt 2>&1; tail -1 $O/${T}_trig_exhaustive.txt
python tools/bulk_parity.py 2048 > $O/${T}_bulk_parity.txt 2>&1; tail -1 $O/${T}_bulk_parity.txt
