set -x
export TMPDIR=/tmp
python tools/call_latency.py > gpurun_out/r02_call_latency.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_gpu_full.log
STEPS=400 bash tools/bench_all.sh > gpurun_out/r02_bench_all.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --strong-reps 0 --c-multi-seconds 0 > gpurun_out/r02_rocprofv3_bench_default.log 2>&1
