set -x
B="python bench.py --cpu-seconds 0 --fp64-steps 0 --exclusive-launches 0"
run() { echo "== $*" >> gpurun_out/r02_b3.log; "$@" >> gpurun_out/r02_b3.log 2>&1; }
for rep in 1 2; do
for src in torch context; do
for st in 2 3 4; do
  run $B --streams $st --stream-source $src --steps 1000 --warmup 100
  run $B --streams $st --stream-source $src --steps 20 --warmup 5
done
done
run env GPU_MAX_HW_QUEUES=8 $B --streams 2 --steps 1000 --warmup 100
run env GPU_MAX_HW_QUEUES=8 $B --streams 3 --steps 1000 --warmup 100
run env GPU_MAX_HW_QUEUES=8 $B --streams 4 --steps 1000 --warmup 100
run env GPU_MAX_HW_QUEUES=8 $B --streams 4 --steps 20 --warmup 5
run $B --streams 3 --finish kernel --steps 1000 --warmup 100
run $B --streams 4 --finish kernel --steps 1000 --warmup 100
run $B --streams 4 --blocks 1024 --steps 1000 --warmup 100
run $B --streams 4 --blocks 512 --steps 1000 --warmup 100
run $B --streams 8 --blocks 512 --steps 1000 --warmup 100
done
python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r02_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t3.log
drivers/multiBench --reps 5 > gpurun_out/r02_multibench.log 2>&1
