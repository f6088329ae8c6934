# PMC passes over the volume frame with the nearest and the trilinear filter (run from the repository root on the GPU box)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_tri
for w in c4 c4tri; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_tri/sq_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_tri/sq_$w.log 2>&1 || { echo fail sq $w; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc_tri/act_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_tri/act_$w.log 2>&1 || { echo fail act $w; tail -3 $R/gpurun_out/pmc_tri/act_$w.log; }
  timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum -d $R/gpurun_out/pmc_tri/tcp_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_tri/tcp_$w.log 2>&1 || { echo fail tcp $w; tail -3 $R/gpurun_out/pmc_tri/tcp_$w.log; }
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_tri/l2_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_tri/l2_$w.log 2>&1 || { echo fail l2 $w; tail -3 $R/gpurun_out/pmc_tri/l2_$w.log; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_tri | grep -v "^   .* 0.0$"
