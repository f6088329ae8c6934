#!/bin/bash
# tools/build_base.sh [commit=HEAD] — build the library as of a commit into build_ab/base.so (git-ignored), for
# tools/probes/ab_lib.py <base.so> <cur.so>: two builds timed on one GPU box in alternation.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rev=${1:-HEAD}
tmp=$(mktemp -d /tmp/pt_base.XXXXXX)
git -C "$root" archive "$rev" include path_tracer_amd/csrc | tar -x -C "$tmp"
mkdir -p "$root/build_ab"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -shared -std=c++17 -Wall \
  -o "$root/build_ab/base.so" "$tmp/path_tracer_amd/csrc/pt_capi.hip"
rm -rf "$tmp"
echo "build_ab/base.so = $(git -C "$root" rev-parse --short "$rev")"
