#!/bin/bash
# a variant build of the library for A/B runs: tools/ab_build.sh <tag> [-DPM_...=...]  ->  ab/lib_<tag>.so  (use with PM_LIB=ab/lib_<tag>.so)
set -e
tag=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -shared -o ../../ab/lib_$tag.so pm_api.hip
