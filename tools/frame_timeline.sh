#!/bin/bash
# Kernel trace of two frames of a 16-spp workload of bench.py and the timeline of the last one
# (run on the MI355X box from the repo root).   bash tools/frame_timeline.sh f16p [extra bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-f16p}; shift
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/tl_$W
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$W -- python3 $R/bench.py --only $W --frames16 2 --frames16-in-flight 0 --no-cpu-baseline --pmc-run "$@" > $R/gpurun_out/tl_$W.json 2> $R/gpurun_out/tl_$W.err || exit 1
cd $R
python3 tools/frame_timeline.py gpurun_out/tl_$W > gpurun_out/tl_$W.txt
tail -3 gpurun_out/tl_$W.txt
