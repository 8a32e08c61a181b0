for cfg in "cva256_f64 2048" "cva256_f64 1792" "cva256_f64 1536" "cva256_f64 1024" "basket16_f64 2048" "basket16_f64 1024" "basket16_f64 3072" "vanilla_f64 2048" "vanilla_f64 1792" "vanilla_f64 1024"; do
set -- $cfg
python bench.py --workload $1 --steps 200 --warmup 20 --cpu-seconds 0 --fp64-steps 0 --strong-reps 0 --c-multi-seconds 0 --streams 1 --blocks $2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 blocks $2 (1 stream): %.4g paths/s; kernel alone %.1f us' % (d['value'], d['roofline']['avg_kernel_us']))"
done
