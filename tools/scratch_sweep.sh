mkdir -p gpurun_out/r02b
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02b/tests.log 2>&1 || { tail -30 gpurun_out/r02b/tests.log; exit 1; }
tail -2 gpurun_out/r02b/tests.log
for f in 0 1; do
LFDMI_DELTA_DIM=$f timeout -k 10 200 python3 bench.py --cpu-sample 8 --no-host-leg --steps 10 > gpurun_out/r02b/dd$f.json 2> gpurun_out/r02b/dd$f.err || exit 1
echo delta $f; python3 tools/show_bench.py gpurun_out/r02b/dd$f.json | grep -E "'value'|k_prep|identical" | cut -c1-260
done
