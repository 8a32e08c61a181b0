set -x
python -m pytest tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r02_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t8.log
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --strong-reps 0 --c-multi-seconds 0 --fp64-steps 0 > gpurun_out/r02_b8.log 2>&1
python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --strong-reps 0 --c-multi-seconds 0 --fp64-steps 0 >> gpurun_out/r02_b8.log 2>&1
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench.py > gpurun_out/r02_pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu_full.log
