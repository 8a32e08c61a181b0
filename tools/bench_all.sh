#!/bin/bash
# Every bench workload once (whole-job rate, the kernel alone, the issue ceiling at 2.4 GHz and at the measured clock, price), for the
# table in DESIGN.md section 7 and for tests/golden/bench_all_r06.json (gpurun_out/bench_all.json: the CPU test asserts ceiling < measured).
# The reference's CPU path beside C2-C5 is in the DEFAULT bench run (bench.py `configs`); here --cpu-seconds 0.
OUT=${OUT:-gpurun_out/bench_all.json}
rm -f /tmp/bench_all_*.json
for w in ${WORKLOADS:-vanilla_f32 vanilla_f64 vanilla_f64_n32 basket4_f32 basket16_f32 basket16_f64 basket16_f64_n32 cva256_f64 cva256_f64_n32 cva256_f32 vanilla_f32_anti basket16_f64_anti basket16_f64_cv}; do
python bench.py --workload $w --steps ${STEPS:-400} --warmup 40 --cpu-seconds 0 --fp64-steps 0 --strong-reps 0 --c-multi-seconds 0 --configs 0 --detail-file /tmp/bench_all_$w.json > /dev/null 2>&1; python -c "
import json,sys; d=json.load(open('/tmp/bench_all_$w.json')); r=d['roofline']; m=r.get('issue_model',{})
print('%-17s %-38s value %.4g paths/s   alone %.4g paths/s (%.1f us)   flop frac %.3f   issue ceiling %s us frac %s (step period %s, typical-cost estimate %s)   sclk %s MHz -> at measured clock %s   price %.6f +- %.6f' % ('$w', r['kernel'][:38], d['value'], r['kernel_paths_per_s'], r['avg_kernel_us'], r['frac'], ('%.1f' % m['ceiling_us']) if m else 'n/a', ('%.3f' % r['issue_frac']) if 'issue_frac' in r else 'n/a', ('%.3f' % m['frac_effective']) if m else 'n/a', ('%.3f' % m['typical_frac']) if m else 'n/a', ('%.0f' % r['sclk_mhz']) if r.get('sclk_mhz') else 'n/a', ('%.3f' % r['issue_frac_at_measured_clock']) if 'issue_frac_at_measured_clock' in r else 'n/a', d['price'], d['confidence_95']))"
done
python - "$OUT" <<'P'
import glob, json, os, sys
rows = {}
for f in sorted(glob.glob('/tmp/bench_all_*.json')):
    d = json.load(open(f)); r = d['roofline']; m = r.get('issue_model') or {}
    if not isinstance(m, dict) or 'ceiling_us' not in m:
        continue
    rows[os.path.basename(f)[len('bench_all_'):-5]] = {
        'kernel': r['kernel'], 'paths': d['config']['paths_per_gpu_per_step'], 'value': d['value'], 'kernel_us': r['avg_kernel_us'], 'frac': r['frac'],
        'ceiling_us': m['ceiling_us'], 'typical_us': m['typical_us'], 'issue_frac': r['issue_frac'], 'frac_effective': m['frac_effective'],
        'sclk_mhz': r.get('sclk_mhz'), 'issue_frac_at_measured_clock': r.get('issue_frac_at_measured_clock'), 'price': d['price'], 'confidence_95': d['confidence_95']}
json.dump({'what': 'tools/bench_all.sh: every bench workload once on one MI355X (kernel alone = exclusive launches; ceiling = tools/issue_model.py, architectural costs, at 2.4 GHz; sclk = amdgpu hwmon during the launches)', 'workloads': rows}, open(sys.argv[1], 'w'), indent=1)
print('wrote', sys.argv[1], len(rows), 'workloads')
P
