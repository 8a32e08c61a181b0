#!/bin/bash
# tools/collect_pmc.sh + tools/summarize_pmc.py for every bench workload (or the ones named), with a progress line per pass:
#   gpurun --timeout 1200 -- bash tools/collect_pmc_all.sh [tag] [workload ...]
# Refreshes profiles/<tag>_pmc_<workload>.txt and profiles/pmc_traffic.json (stamped with the kernel headers' sha256).
TAG=${1:-r04}; shift
W=${@:-vanilla_f32 vanilla_f64 vanilla_f64_n32 basket4_f32 basket16_f32 basket16_f64 basket16_f64_n32 cva256_f64 cva256_f64_n32 cva256_f32 vanilla_f32_anti basket16_f64_anti basket16_f64_cv}
for w in $W; do
    echo "== $w $(date +%T)"
    bash tools/collect_pmc.sh "$w" 2>&1 | grep "pass" || true
    python3 tools/summarize_pmc.py "$w" "$TAG" > "gpurun_out/pmc_summary_$w.txt" 2>&1 || echo "summarize failed for $w"
    tail -4 "gpurun_out/pmc_summary_$w.txt"
done
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
cp profiles/${TAG}_pmc_*.txt gpurun_out/ 2>/dev/null
echo done
