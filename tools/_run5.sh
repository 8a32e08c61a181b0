set -x
python -m pytest tests/test_gpu_greeks.py tests/test_gpu_parity.py -x -q -m gpu -k "greeks" > gpurun_out/r02_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t5.log
