#!/bin/sh
# oracle/make_golden.sh -- TEST INFRASTRUCTURE. Regenerates tests/golden/*.bin by running the
# compiled reference (oracle/_ref/ref_harness) in the build container. The vectors are data
# (inputs + the reference's outputs); no reference source is copied.
set -e
cd "$(dirname "$0")"
make ref
mkdir -p ../tests/golden
./_ref/ref_harness ../tests/golden 160 90 2048 12345 | grep -v "^Building\|traced in\|kd-tree" | tee ../tests/golden/REPORT_160x90_n2048.txt
./_ref/ref_harness ../tests/golden 96 96 4096 2024 | grep -v "^Building\|traced in\|kd-tree" | tee ../tests/golden/REPORT_96x96_n4096.txt
# synthetic scenes built with the reference's constructors (see ref_harness.cpp): BASELINE config 4, config 5's
# generator at 100 spheres, 100 large overlapping spheres with glass and mirrors among them, and the same in motion
for scene in caustics spheres100 mixed100 moving100; do
    ./_ref/ref_harness ../tests/golden 96 54 2048 4242 write $scene | grep -v "^Building\|traced in\|kd-tree" | tee ../tests/golden/REPORT_${scene}_96x54_n2048.txt
done
