set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err
rc=$?
tail -c 600 gpurun_out/r3_bench2.err
echo "bench rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests2.log 2>&1
rc=$?
tail -5 gpurun_out/r3_tests2.log
exit $rc
