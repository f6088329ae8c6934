#!/bin/bash
# tools/round_profile.sh <tag> — profiles for profiles/<tag>_*: run on the GPU box via gpurun.
#   1. the default bench line (bench.py --steps 10 --warmup 2)
#   2. rocprofv3 --kernel-trace --stats of the same command with --frames-in-flight 1 (the mode the roofline object
#      describes: kernels of consecutive frames do not overlap, a launch's duration is that of a kernel running alone)
#   2b. rocprofv3 --kernel-trace of the default mode (two frames in flight): the kernels' begin and end stamps
#   3. PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE, L2 hit/miss/request counters — one pass each) for workloads c2, c3, c4
set -u
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2"
timeout -k 10 600 $BENCH > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $out/trace_bench -o t -- $BENCH --no-cpu --no-extras --frames-in-flight 1 > $out/trace_bench.log 2>&1 || { echo trace failed; exit 1; }
# the headline mode (two frames in flight): begin / end stamps of the kernels, for the overlap
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $out/trace_two_in_flight -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $out/trace_two_in_flight.log 2>&1 || { echo trace two in flight failed; exit 1; }
for wl in c2 c3 c4; do
  B="python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-extras --no-cpu --frames-in-flight 1"
  if [ $wl != c2 ]; then
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/trace_$wl -o t -- $B > $out/trace_$wl.log 2>&1 || { echo trace $wl failed; exit 1; }
  fi
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/pmc_sq_$wl -o p -- $B > $out/pmc_sq_$wl.log 2>&1 || { echo pmc sq $wl failed; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU -d $out/pmc_valu_$wl -o p -- $B > $out/pmc_valu_$wl.log 2>&1 || { echo pmc valu $wl failed; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch_$wl -o p -- $B > $out/pmc_fetch_$wl.log 2>&1 || { echo pmc fetch $wl failed; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write_$wl -o p -- $B > $out/pmc_write_$wl.log 2>&1 || { echo pmc write $wl failed; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $out/pmc_l2_$wl -o p -- $B > $out/pmc_l2_$wl.log 2>&1 || { echo pmc l2 $wl failed; exit 1; }
done
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import __graft_entry__ as g; print(g.source_hash())" > $out/source_hash.txt
echo profile done
