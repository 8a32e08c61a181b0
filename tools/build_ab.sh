#!/bin/bash
# Builds A/B variants of the engine, one .so each, for tools/ab_f64.py and tools/ab_basket.py (both time
# every tools/ab_*.so they find, interleaved in one process).
#   tools/build_ab.sh 0_old=-DMC_AB_NO_TABLES 1_default= 2_fence4=-DMC_AB_FENCE_PERIOD=4
set -e
cd "$(dirname "$0")/../montecarlocuda_amd/csrc"
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -w -shared"
rm -f ../../tools/ab_*.so
n=0
for spec in "$@"; do
    name="${spec%%=*}"; flags="${spec#*=}"
    /opt/rocm/bin/hipcc $F ${flags//,/ } -o "../../tools/ab_${name}.so" mc_api.hip &
    n=$((n + 1)); if [ $((n % 4)) = 0 ]; then wait; fi
done
wait
ls -la ../../tools/ab_*.so
