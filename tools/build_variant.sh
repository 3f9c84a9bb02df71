#!/bin/bash
# Build a variant of libcmx.so for same-box A/B runs (tools/ab_bench.sh):
#   [LITCOEF_EXTRA="cmx_x.o …"] tools/build_variant.sh <tag> [extra hipcc flags, e.g. -DCMX_PHASE_CONSTS=0]   →  cloudmicrophysics.jl_amd/csrc/libcmx_<tag>.so
set -e
tag=$1; shift
src=$(cd "$(dirname "$0")/../cloudmicrophysics.jl_amd/csrc" && pwd)
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
make -s -C "$src" -f "$src/Makefile" -j8 VPATH="$src" OUT="$src/libcmx_$tag.so" OBJDIR="$obj" LITCOEF_EXTRA="${LITCOEF_EXTRA:-}" NOSLP_EXTRA="${NOSLP_EXTRA:-}" VARIANT_FLAGS="$*" \
     CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $*" variant
echo "$src/libcmx_$tag.so"
