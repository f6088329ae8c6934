set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 0; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_c4v/sq_$t -o p -- python3 $R/tools/probes/run_frames.py c4 $t 6 > $R/gpurun_out/pmc_c4v/sq_$t.log 2>&1 || { echo fail sq $t; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_c4v/l2_$t -o p -- python3 $R/tools/probes/run_frames.py c4 $t 6 > $R/gpurun_out/pmc_c4v/l2_$t.log 2>&1 || { echo fail l2 $t; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum -d $R/gpurun_out/pmc_c4v/tcp_$t -o p -- python3 $R/tools/probes/run_frames.py c4 $t 6 > $R/gpurun_out/pmc_c4v/tcp_$t.log 2>&1 || { echo fail tcp $t; tail -3 $R/gpurun_out/pmc_c4v/tcp_$t.log; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_c4v | grep -v "^   .* 0.0$"
