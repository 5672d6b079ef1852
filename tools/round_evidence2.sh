#!/bin/bash
T=${1:-r03}
O=gpurun_out
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_capacity.py -m gpu -x -q 2>&1 | tail -2
bash tools/scratch_ab.sh 2>&1 | tee $O/${T}_ab_bits_erode_rep.txt
bash tools/rehearse_ranks.sh 4 > $O/${T}_four_ranks.json 2> $O/${T}_four_ranks.err; tail -c 600 $O/${T}_four_ranks.json; echo
python tools/dropin_probe.py 2048 64 > $O/${T}_dropin_probe.txt 2>&1; tail -4 $O/${T}_dropin_probe.txt
python tools/trig_exhaustive.py 2048 > $O/${T}_trig_exhaustive.txt 2>&1; tail -1 $O/${T}_trig_exhaustive.txt
python tools/bulk_parity.py 2048 > $O/${T}_bulk_parity.txt 2>&1; tail -2 $O/${T}_bulk_parity.txt
