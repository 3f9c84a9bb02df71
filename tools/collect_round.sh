#!/bin/bash
# After tools/gpu_round.sh <tag> ran on the GPU box (gpurun merges gpurun_out/ back): copy the evidence into profiles/ and regenerate DESIGN.md's blocks.
#   tools/collect_round.sh r06
set -eu
TAG=$1
mkdir -p profiles/bench_$TAG
cp gpurun_out/profiles/${TAG}_* profiles/
cp gpurun_out/bench/*.json profiles/bench_$TAG/
cp gpurun_out/gpu_tests.log profiles/${TAG}_gpu_tests.log
cp gpurun_out/gpu_round_${TAG}.log profiles/${TAG}_round.log
python tools/gen_design_tables.py --write --round $TAG
