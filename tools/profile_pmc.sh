#!/bin/bash
# rocprofv3 counter passes for bench.py (run on the MI355X box from the repo root).
# One pass per counter group (rocprofv3 does not split counters over passes; per pass at most
# 8 SQ, 4 TCC -- FETCH_SIZE takes 3, WRITE_SIZE 2 --, 4 TCP, 2 TA, 2 GRBM); counter runs carry
# no trace options.  For each workload W in $WORKLOADS (default: main f16w f16p c5) writes
# gpurun_out/pmc_W/<pass>/ and gpurun_out/pmc_W/summary.json.  bench.py runs with --pmc-run:
# only the frames a plain run times are rendered, once, so every launch belongs to one of them.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for W in ${WORKLOADS:-main f16w f16p c5}; do
  OUT=$R/gpurun_out/pmc_$W
  rm -rf "$OUT"
  mkdir -p "$OUT"
  # the frames the driver's run times (bench.py --steps 20 --warmup 5; ten 16-spp frames; one configs[4] frame)
  case $W in
    main) ARGS="--only main --pmc-run --steps ${PMC_STEPS:-20} --warmup ${PMC_WARMUP:-5}";;
    c5) ARGS="--only c5 --pmc-run --c5-frames 1";;
    *) ARGS="--only $W --pmc-run --frames16 ${PMC_FRAMES16:-10}";;
  esac
  ARGS="$ARGS ${BENCH_ARGS:-}"
  echo "$ARGS" > "$OUT/bench_args.txt"
  run() {
      name=$1; shift
      timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1
      rc=$?
      echo "$W $name rc=$rc"
      return $rc
  }
  run fetch FETCH_SIZE || exit 1
  run write WRITE_SIZE || exit 1
  run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
  run sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
  run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM || exit 1
  run tcp1 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
  python3 "$R/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json"
  echo "$W summary written"
done
