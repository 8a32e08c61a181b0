#!/bin/bash
# Collect PMC counters for one bench workload on the GPU box, one rocprofv3 pass per counter group
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   gpurun -- bash tools/collect_pmc.sh vanilla_f32
# Results: gpurun_out/pmc_<workload>/<group>/  (then: python tools/summarize_pmc.py)
set -o pipefail
W=${1:-vanilla_f32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
OUT=$R/gpurun_out/pmc_$W
mkdir -p "$OUT"
PATHS_ARG=""
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --workload "$W" $PATHS_ARG --steps 20 --warmup 2 --regions 1 --cpu-seconds 0 --profile-every 0 --preheat-ms 0 --strong-reps 0 --c-multi-seconds 0 --fp64-steps 0 --exclusive-launches 0 --configs 0 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY &&
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
# the instruction counters once more at TWICE the paths per launch (same grid): the difference of the two runs is what the
# hot loop alone issues per path -- prologue, table staging and reduction cancel (tools/issue_model.py checks its opcode
# histogram against that slope)
case "$W" in vanilla*|basket4*) P2=200000000 ;; basket16*) P2=250000000 ;; cva256*) P2=2500000 ;; *) P2=0 ;; esac
if [ "$P2" != 0 ]; then PATHS_ARG="--paths $P2"; run sq1x2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS || true; PATHS_ARG=""; fi
# LDS behaviour of the table lookups (fp64 kernels); counter names vary between ASICs, so this pass may fail alone
run lds SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT || true
find "$OUT" -name "*counter_collection.csv" | head
