#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the MI355X box from the repo root).
# One pass per counter group (rocprofv3 does not split counters over passes; per pass at most
# 8 SQ, 4 TCC -- FETCH_SIZE takes 3, WRITE_SIZE 2 --, 4 TCP, 2 TA, 2 GRBM); counter runs carry
# no trace options.  Writes gpurun_out/pmc/<pass>/ and gpurun_out/pmc/summary.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-frame16 --no-serial-leg ${BENCH_ARGS:-}"
run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    rc=$?
    echo "$name rc=$rc"
    return $rc
}
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM || exit 1
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 1
run tcp1 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
run tcp2 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum || exit 1
python3 "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json"
echo "summary written"
