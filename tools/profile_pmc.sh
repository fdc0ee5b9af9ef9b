#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the MI355X box from the repo root).
# One pass per counter group (rocprofv3 does not split counters over passes); counter runs
# carry no trace options.  Summaries: tools/summarize_pmc.py.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"
}
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM || exit 1
find "$OUT" -name "*counter_collection.csv" | head
