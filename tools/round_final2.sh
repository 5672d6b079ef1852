#!/bin/bash
T=${1:-r03}
O=gpurun_out
python -m pytest tests -m gpu -q > $O/${T}_gpu_tests_final.log 2>&1; tail -2 $O/${T}_gpu_tests_final.log
python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
python -c "
import json
d=json.load(open('$O/${T}_bench_default.json'))
print(d['value'], d['ms_per_step'], d['sustained'], d['dropin']['plain']['value'], d['dropin']['plain']['steady_state_frames_per_s'], d['lsst']['value'], d['roofline']['canny_hough']['frac'])"
python tools/bulk_parity.py 4096 2048 > $O/${T}_bulk_parity_more.txt 2>&1; tail -1 $O/${T}_bulk_parity_more.txt
python tools/bulk_parity.py lsst 32 > $O/${T}_bulk_parity_lsst.txt 2>&1; tail -1 $O/${T}_bulk_parity_lsst.txt
