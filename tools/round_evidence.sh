#!/bin/bash
# The round's evidence set on one MI355X box, every step writing under gpurun_out/ (copy what is to be judged into profiles/):
#   gpurun --timeout 1200 -- bash tools/round_evidence.sh r05
# default bench line, two driver-style runs, all ten workloads, rocprofv3 kernel stats of the default command and of the
# single-stream run (where every launch is alone on the device: the per-kernel duration), launch contention probe, smoke.
TAG=${1:-r06}
O=gpurun_out
export TMPDIR=/tmp
set -e -o pipefail
mkdir -p $O
step() { echo "== $1 $(date +%T)"; }
# the issue ceilings of every bench line come from profiles/issue_model.json: stale (stamp of another code object) means lines without them
python3 - <<'P' || { echo "profiles/issue_model.json or pmc_traffic.json is stale: collect_pmc_all.sh, then make asm + tools/issue_model.py in the container, THEN this"; exit 3; }
import json, sys, bench
s = bench.launch_stamp()["stamp"]
m = json.load(open("profiles/issue_model.json")); p = json.load(open("profiles/pmc_traffic.json"))
sys.exit(0 if m["vanilla_f32"]["launch_stamp"] == s and p["vanilla_f32"]["launch_stamp"] == s else 1)
P
# every bench run: the ONE stdout line (<= 4 KB) into the .log, the full record next to it (--detail-file)
step "bench default";       python3 bench.py --detail-file $O/${TAG}_bench_detail_default.json > $O/${TAG}_bench_default.log 2>&1
step "bench detail full";   python3 bench.py --detail full --detail-file $O/${TAG}_bench_detail_full.json > $O/${TAG}_bench_full.log 2>&1
step "driver style 1";      ( time python3 bench.py --steps 20 --warmup 5 --detail-file $O/${TAG}_bench_detail_driver_style_1.json ) > $O/${TAG}_bench_driver_style_1.log 2>&1
step "driver style 2";      ( time python3 bench.py --steps 20 --warmup 5 --detail-file $O/${TAG}_bench_detail_driver_style_2.json ) > $O/${TAG}_bench_driver_style_2.log 2>&1
step "all workloads";       bash tools/bench_all.sh > $O/${TAG}_bench_all_workloads.log 2>&1
step "rocprofv3 default";   rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --cpu-seconds 0 --strong-reps 0 --c-multi-seconds 0 --configs 0 > $O/${TAG}_rocprofv3_bench_default.log 2>&1
step "rocprofv3 streams 1"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_streams1 -- python3 bench.py --streams 1 --cpu-seconds 0 --strong-reps 0 --c-multi-seconds 0 --configs 0 > $O/${TAG}_rocprofv3_bench_streams1.log 2>&1
cp "$(find $O/prof_default -name '*kernel_stats.csv' | tail -1)" $O/${TAG}_rocprofv3_kernel_stats_bench_default.csv
cp "$(find $O/prof_streams1 -name '*kernel_stats.csv' | tail -1)" $O/${TAG}_rocprofv3_kernel_stats_bench_streams1.csv
rm -rf $O/prof_default $O/prof_streams1
step "CVA staircase";       [ -x tools/c/shard_clock ] && tools/c/shard_clock AD > $O/${TAG}_shard_clock_AD.log 2>&1
step "basket greeks";       python3 tools/basket_greeks_speed.py > $O/${TAG}_basket_greeks_speed.log 2>&1
step "smoke";               python3 -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
step done
grep "^{" $O/${TAG}_bench_default.log | cut -c1-600
