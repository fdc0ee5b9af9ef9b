set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests1.log 2>&1
rc=$?
tail -5 gpurun_out/r3_tests1.log
[ $rc -eq 0 ] || exit $rc
KT_NAMES="f16w-serial f16p-serial" SKIP_PMC=1 bash tools/profile_all.sh
