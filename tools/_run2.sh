set -x
python -m pytest tests -x -q -m gpu > gpurun_out/r02_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t2.log
B="python bench.py --cpu-seconds 0 --fp64-steps 0"
for rep in 1 2; do
for args in "--finish fused --steps 1000 --warmup 100" "--finish kernel --steps 1000 --warmup 100" "--finish fused --steps 1000 --warmup 100 --streams 1" "--finish kernel --steps 1000 --warmup 100 --streams 1" "--finish fused --steps 20 --warmup 5" "--finish kernel --steps 20 --warmup 5" "--finish fused --steps 20 --warmup 5 --preheat-ms 0" "--finish fused --steps 20 --warmup 5 --streams 1" "--finish fused --steps 20 --warmup 5 --streams 3" "--finish fused --steps 1000 --warmup 100 --streams 3"; do
  echo "== $args" >> gpurun_out/r02_b2.log
  $B $args >> gpurun_out/r02_b2.log 2>&1
done
done
for w in basket4_f32 basket16_f64 cva256_f64 vanilla_f64; do
for args in "--finish fused" "--finish kernel"; do
  echo "== $w $args" >> gpurun_out/r02_b2.log
  $B --workload $w $args --steps 200 --warmup 20 >> gpurun_out/r02_b2.log 2>&1
done
done
for blocks in 1280 1536 1792 2048; do
  echo "== basket4_f32 --blocks $blocks" >> gpurun_out/r02_b2.log
  $B --workload basket4_f32 --blocks $blocks --steps 200 --warmup 20 >> gpurun_out/r02_b2.log 2>&1
done
