#!/bin/bash
# Usage: tools/prof_pmc.sh <tag> <workload> -- collects kernel-trace stats and
# two PMC passes for `bench.py --workload <workload>` into gpurun_out/<tag>/.
set -u
tag=$1; wl=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-extras --no-cpu"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $B > $out/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $out/pmc1 -o p -- $B > $out/pmc1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc2 -o p -- $B > $out/pmc2.log 2>&1 || exit 1
find $out -name "*.csv" | head -20
