# VALU / SALU instructions and time of frames without any march: the per-tile fixed cost of the frame kernel
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_fixed
for w in c4away empty; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_fixed/sq_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_fixed/sq_$w.log 2>&1 || { echo fail $w; exit 1; }
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc_fixed/kt_$w -o p -- python3 $R/tools/probes/run_frames.py $w 0 6 > $R/gpurun_out/pmc_fixed/kt_$w.log 2>&1 || { echo fail kt $w; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_fixed | grep -v "^   .* 0.0$"
