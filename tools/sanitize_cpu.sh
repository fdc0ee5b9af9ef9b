#!/bin/bash
# CPU test-suite with AddressSanitizer + UndefinedBehaviorSanitizer builds of the host library and
# the oracle (host code only: GPU sanitizers are not available).  Restores the normal builds afterwards.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp "$R/oracle/librt_oracle.so" "$T/oracle.so"
cp "$R/raytracer_amd/host/librth_host.so" "$T/host.so"
restore() { cp "$T/oracle.so" "$R/oracle/librt_oracle.so"; cp "$T/host.so" "$R/raytracer_amd/host/librth_host.so"; rm -rf "$T"; }
trap restore EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
gcc $SAN -std=c11 -fPIC -ffp-contract=off -fopenmp -shared -o "$R/oracle/librt_oracle.so" "$R/oracle/rt_oracle.c" -lm
(cd "$R/raytracer_amd/host" && g++ $SAN -std=c++17 -fPIC -ffp-contract=off -pthread -shared -o librth_host.so \
    scene.cpp kdbuild.cpp png.cpp fbx.cpp capi.cpp -lz)
cd "$R"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/ -x -q -m "not gpu" -p no:cacheprovider
