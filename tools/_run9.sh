B="python bench.py --cpu-seconds 0 --fp64-steps 0 --exclusive-launches 0 --strong-reps 0 --c-multi-seconds 0"
run() { echo "== $*" >> gpurun_out/r02_b10.log; "$@" >> gpurun_out/r02_b10.log 2>&1; }
for rep in 1 2 3 4 5 6; do
run $B --steps 20 --warmup 5 --poll 1
run $B --steps 20 --warmup 5 --poll 0
done
run $B --steps 1000 --warmup 100 --poll 1
run $B --steps 1000 --warmup 100 --poll 0
