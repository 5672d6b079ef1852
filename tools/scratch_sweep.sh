mkdir -p gpurun_out/r02b
timeout -k 10 900 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "libm or fit_min" 2>&1 | tail -15
