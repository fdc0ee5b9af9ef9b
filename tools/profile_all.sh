#!/bin/bash
# rocprofv3 kernel traces of bench.py for the three walks, then the counter passes (run on the MI355X box from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for walk in bvh kd-culled reference; do
  steps=5; [ $walk = reference ] && steps=2
  rm -rf $R/gpurun_out/kt_$walk
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$walk -- python3 $R/bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-frame16 --walk $walk > $R/gpurun_out/kt_$walk.json 2> $R/gpurun_out/kt_$walk.err || exit 1
  echo "kt $walk ok"
done
cd $R && bash tools/profile_pmc.sh || exit 1
