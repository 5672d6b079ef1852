#!/bin/bash
# final evidence of the round on the final code: profiles (bench lines, rocprof stats, PMC traffic), SQ counters, stage clocks
T=${1:-r03}
bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1; tail -3 gpurun_out/${T}_collect.log
bash tools/sq_counters.sh ${T}_sq > /dev/null 2>&1; cp gpurun_out/${T}_sq/sq_counters.txt gpurun_out/${T}_sq_counters.txt
python tools/dc_profile.py > gpurun_out/${T}_dc_stage_clocks.txt 2>/dev/null; cat gpurun_out/${T}_dc_stage_clocks.txt
python tools/frame_profile.py > gpurun_out/${T}_frame_profile.txt 2>/dev/null; cat gpurun_out/${T}_frame_profile.txt
