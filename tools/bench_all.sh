#!/bin/bash
# Every bench workload once (whole-job rate, the kernel alone, price), for the table in DESIGN.md section 7.
for w in vanilla_f32 vanilla_f64 basket4_f32 basket16_f32 basket16_f64 cva256_f64 cva256_f32; do
python bench.py --workload $w --steps ${STEPS:-400} --warmup 40 --cpu-seconds 0 --fp64-steps 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-13s %-38s value %.4g paths/s   alone %.4g paths/s (%.1f us)   flop frac %.3f   price %.6f +- %.6f' % ('$w', r['kernel'], d['value'], r['exclusive']['kernel_paths_per_s'], r['exclusive']['avg_kernel_us'], r['exclusive']['frac'], d['price'], d['confidence_95']))"
done
