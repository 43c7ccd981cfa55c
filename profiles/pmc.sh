#!/bin/bash
# rocprofv3 PMC passes of the bench command (run on the GPU box), one pass per counter group as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass):
#   profiles/pmc.sh <out_dir> [bench args...]       then   python profiles/summarize_pmc.py <out_dir> profiles/rNN_pmc_summary.csv rNN
out=$(realpath -m "$1"); shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python "$root/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 || tail -3 "$out/$name.log"
}
BENCH_ARGS=("$@")
run sq SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find "$out" -name '*counter_collection.csv' | head
