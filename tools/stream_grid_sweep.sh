for w in vanilla_f64 vanilla_f32; do
for cfg in "1 2048" "2 2048" "2 1024" "3 1024" "2 1536" "4 512"; do
set -- $cfg
python bench.py --workload $w --steps 600 --warmup 60 --cpu-seconds 0 --fp64-steps 0 --exclusive-launches 0 --streams $1 --blocks $2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w streams $1 blocks $2: %.4g paths/s  step %.2f us' % (d['value'], d['ms_per_step']*1e3))"
done; done
