cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 3 --warmup 1 --no-extras --no-cpu --frames-in-flight 1"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/icache -o p -- $B > $GRAFT_REPO_ROOT/gpurun_out/icache.log 2>&1
