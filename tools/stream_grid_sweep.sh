#!/bin/bash
# Whole-job rate of bench.py for a few (streams, workgroups per launch) choices, one process each, back to back.
W=${1:-vanilla_f32}
for rep in 1 2; do
for cfg in "2 2048" "2 1024" "4 512" "4 1024" "3 1024" "8 256" "1 2048"; do
set -- $cfg
python bench.py --workload $W --steps 1000 --warmup 100 --cpu-seconds 0 --fp64-steps 0 --exclusive-launches 0 --strong-reps 0 --c-multi-seconds 0 --streams $1 --blocks $2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W streams $1 blocks $2: %.4g paths/s  step %.2f us' % (d['value'], d['ms_per_step']*1e3))"
done; done
