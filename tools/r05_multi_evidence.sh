#!/bin/bash
# Round-5 evidence for libmc_multi's fan-out: soak with percentiles and the slowest calls' per-device traces,
# at the default linger time and at a short one; the fixed-cost / fan-out medians; all under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L="-Lmontecarlocuda_amd/csrc -lmc_multi -lmc_mi355x -lm -Wl,-rpath,$PWD/montecarlocuda_amd/csrc -Wl,-rpath-link,montecarlocuda_amd/csrc:/opt/rocm/lib"
gcc -O2 -std=gnu11 -Iinclude tools/c/multi_soak.c $L -o /tmp/multi_soak || exit 1
gcc -O2 -std=gnu11 -Iinclude tools/c/multi_cost.c $L -o /tmp/multi_cost || exit 1
CALLS=${1:-300000}
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > gpurun_out/r05_multi_soak_default_linger.log
timeout -k 10 400 /tmp/multi_soak $CALLS >> gpurun_out/r05_multi_soak_default_linger.log 2>&1; echo "rc $?" >> gpurun_out/r05_multi_soak_default_linger.log
echo "MC_MULTI_LINGER_US=1000" > gpurun_out/r05_multi_soak_linger_1ms.log
MC_MULTI_LINGER_US=1000 timeout -k 10 400 /tmp/multi_soak $CALLS >> gpurun_out/r05_multi_soak_linger_1ms.log 2>&1; echo "rc $?" >> gpurun_out/r05_multi_soak_linger_1ms.log
timeout -k 10 300 /tmp/multi_cost > gpurun_out/r05_multi_fixed_cost_and_fanout.log 2>&1; echo "rc $?" >> gpurun_out/r05_multi_fixed_cost_and_fanout.log
tail -32 gpurun_out/r05_multi_soak_default_linger.log
tail -8 gpurun_out/r05_multi_fixed_cost_and_fanout.log
