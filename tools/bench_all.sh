#!/bin/bash
# Every bench workload once (whole-job rate, the kernel alone, the issue ceiling, price), for the table in DESIGN.md section 7.
for w in ${WORKLOADS:-vanilla_f32 vanilla_f64 vanilla_f64_n32 basket4_f32 basket16_f32 basket16_f64 basket16_f64_n32 cva256_f64 cva256_f64_n32 cva256_f32}; do
python bench.py --workload $w --steps ${STEPS:-400} --warmup 40 --cpu-seconds 0 --fp64-steps 0 --strong-reps 0 --c-multi-seconds 0 --detail-file /tmp/bench_all_$w.json > /dev/null 2>&1; python -c "
import json,sys; d=json.load(open('/tmp/bench_all_$w.json')); r=d['roofline']; m=r.get('issue_model',{})
print('%-17s %-38s value %.4g paths/s   alone %.4g paths/s (%.1f us)   flop frac %.3f   issue ceiling %s us frac %s (step period %s, typical-cost estimate %s)   price %.6f +- %.6f' % ('$w', r['kernel'], d['value'], r['kernel_paths_per_s'], r['avg_kernel_us'], r['frac'], ('%.1f' % m['ceiling_us']) if m else 'n/a', ('%.3f' % r['issue_frac']) if 'issue_frac' in r else 'n/a', ('%.3f' % m['frac_effective']) if m else 'n/a', ('%.3f' % m['typical_frac']) if m else 'n/a', d['price'], d['confidence_95']))"
done
