mkdir -p gpurun_out/r02b
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02b/tests.log 2>&1 || { tail -30 gpurun_out/r02b/tests.log; exit 1; }
tail -2 gpurun_out/r02b/tests.log
timeout -k 10 200 python3 bench.py --cpu-sample 0 --no-host-leg --steps 10 > gpurun_out/r02b/s.json 2> gpurun_out/r02b/s.err || exit 1
timeout -k 10 300 python3 bench.py --workload lsst --cpu-sample 0 --no-host-leg --steps 5 > gpurun_out/r02b/l.json 2> gpurun_out/r02b/l.err || exit 1
python3 tools/show_bench.py gpurun_out/r02b/s.json | grep -E "'value'" | cut -c1-200
python3 tools/show_bench.py gpurun_out/r02b/l.json | grep -vE "^None|bound|workload" | cut -c1-200
