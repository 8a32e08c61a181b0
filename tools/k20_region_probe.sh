# The 20-step timed region of the driver (--steps 20 --warmup 5), 9 regions per run, two runs per variant: default | --poll 1 | --streams 3 | both.
# gpurun -- bash tools/k20_region_probe.sh   (profiles/r05_k20_region_probe.log)
for a in "" "--poll 1" "--streams 3" "--poll 1 --streams 3" "" "--poll 1"; do
  for i in 1 2; do
    python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --fp64-steps 0 --strong-reps 0 --c-multi-seconds 0 --exclusive-launches 0 --regions 9 $a --detail-file /tmp/k20.json > /dev/null 2>&1
    python -c "
import json; d=json.load(open('/tmp/k20.json')); r=sorted(d['region_ms_per_step']); print('%-24s value %.4g  ms/step median %.5f min %.5f max %.5f  enqueue %.3f close %.3f' % ('$a', d['value'], d['ms_per_step'], r[0], r[-1], d['timed_region_host']['enqueue_K_steps_ms'], d['timed_region_host']['close_region_ms']))"
  done
done
