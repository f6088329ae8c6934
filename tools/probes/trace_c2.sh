#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_quick
rm -rf $out; mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu --no-extras > $out/log.txt 2>&1 || { echo trace failed; tail -5 $out/log.txt; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out | head -12
grep "^{" $out/log.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"bench\", d[\"value\"], d[\"ms_per_step\"])"
