#!/bin/bash
# A/B timing of the three bench workloads, brief (run on the MI355X box from the repo root):
#   bash tools/ab.sh <tag> [path of another build of librtx_hip.so] [VAR=value ...]
# writes gpurun_out/ab_<tag>.log: per workload the ms per step / frame of bench.py.
tag=$1; lib=${2:-}; shift; shift
mkdir -p gpurun_out
out=gpurun_out/ab_$tag.log
: > $out
for w in ${AB_WORKLOADS:-main f16w f16p}; do
  extra="--frames16 ${AB_FRAMES16:-6} --frames16-in-flight 0"
  [ $w = main ] && extra="--no-serial-leg --repeats 2 --steps 20 --warmup 5"
  echo "== $tag workload $w" >> $out
  env ${lib:+RTX_HIP_LIB=$lib} "$@" timeout -k 10 150 python bench.py --only $w --no-cpu-baseline $extra 2>&1 | python tools/brief.py >> $out || exit 1
done
cat $out
