#!/bin/bash
# rocprofv3 kernel traces of bench.py for the three walks (four frames in flight, as in the timed
# steps) and for the default walk with one frame at a time on one stream, then the counter passes
# (run on the MI355X box from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for walk in bvh kd-culled reference bvh-serial; do
  steps=12; [ $walk = reference ] && steps=4
  extra="--no-serial-leg --walk $walk"; [ $walk = bvh-serial ] && extra="--in-flight 1 --serial-chunks --walk bvh"
  rm -rf $R/gpurun_out/kt_$walk
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$walk -- python3 $R/bench.py --steps $steps --warmup 4 --no-cpu-baseline --no-frame16 $extra > $R/gpurun_out/kt_$walk.json 2> $R/gpurun_out/kt_$walk.err || exit 1
  echo "kt $walk ok"
done
cd $R && bash tools/profile_pmc.sh || exit 1
