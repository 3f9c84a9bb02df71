#!/bin/bash
# Build libcmx_<tag>.so from a COMMITTED revision (for same-box A/B against the working tree):
#   tools/build_ref_variant.sh <tag> <git-rev>      →  cloudmicrophysics.jl_amd/csrc/libcmx_<tag>.so
# The sources are `git worktree add`-ed under a temporary directory, built with that revision's own Makefile, and removed again.
set -e
tag=$1; rev=$2
repo=$(cd "$(dirname "$0")/.." && pwd)
wt=$(mktemp -d)
trap 'git -C "$repo" worktree remove --force "$wt" >/dev/null 2>&1 || rm -rf "$wt"' EXIT
git -C "$repo" worktree add --detach "$wt" "$rev" >/dev/null
make -s -C "$wt/cloudmicrophysics.jl_amd/csrc" -j8
cp "$wt/cloudmicrophysics.jl_amd/csrc/libcmx.so" "$repo/cloudmicrophysics.jl_amd/csrc/libcmx_$tag.so"
echo "$repo/cloudmicrophysics.jl_amd/csrc/libcmx_$tag.so  ($(git -C "$repo" rev-parse --short "$rev"))"
