#!/bin/bash
# rocprofv3 kernel traces of bench.py (run on the MI355X box from the repo root): the headline
# workload with four frames in flight (as in the timed steps) and with one frame at a time on one
# stream, and each 16-spp frame on its own; then the counter passes (tools/profile_pmc.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for name in ${KT_NAMES:-main main-serial f16w f16p f16w-serial f16p-serial}; do
  case $name in
    main) extra="--only main --steps 12 --warmup 4 --no-cpu-baseline --no-serial-leg --repeats 0";;
    main-serial) extra="--only main --steps 12 --warmup 4 --no-cpu-baseline --no-serial-leg --repeats 0 --in-flight 1 --serial-chunks";;
    f16w-serial|f16p-serial) extra="--only ${name%-serial} --frames16 4 --no-cpu-baseline --serial-chunks --frames16-in-flight 0";;
    *) extra="--only $name --frames16 4 --no-cpu-baseline";;
  esac
  rm -rf $R/gpurun_out/kt_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$name -- python3 $R/bench.py $extra > $R/gpurun_out/kt_$name.json 2> $R/gpurun_out/kt_$name.err || exit 1
  echo "kt $name ok"
done
if [ -z "${SKIP_PMC:-}" ]; then cd $R && bash tools/profile_pmc.sh || exit 1; fi
