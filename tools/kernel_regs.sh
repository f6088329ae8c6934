#!/bin/bash
# tools/kernel_regs.sh — compile pt_capi.hip for gfx950 (device code only, into /tmp) and print the register and
# scratch use of every kernel; with -s also keep the assembly in /tmp/pt_asm/pt_capi.s.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/pt_asm
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -std=c++17 --cuda-device-only -S \
  -o /tmp/pt_asm/pt_capi.s "$root/path_tracer_amd/csrc/pt_capi.hip" "$@"
awk '$1==".name:"{n=$2} $1==".private_segment_fixed_size:"{p=$2} $1==".sgpr_spill_count:"{ss=$2} $1==".vgpr_count:"{v=$2} $1==".vgpr_spill_count:"{printf "%-72s vgpr %3d  vspill %3d  sspill %3d  scratch %4d B\n", n, v, $2, ss, p}' /tmp/pt_asm/pt_capi.s
