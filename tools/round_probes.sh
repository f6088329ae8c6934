#!/bin/bash
# tools/round_probes.sh <tag> — side measurements of a build (band shares, launch sizes of frames in flight,
# wave timelines, rank 0's exchange cost, what other libraries in the process do to the streams), one log each under
# gpurun_out/<tag>/; run on the GPU box after tools/round_profile.sh.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift; timeout -k 10 300 "$@" >> $out/$name.txt 2>&1 || { echo "$name failed"; exit 1; }; }
run band_shares python tools/probes/band_throughput.py c3 8
run band_shares python tools/probes/band_throughput.py c2 30
run band_blocks python tools/probes/band_blocks.py c3 8
run band_blocks python tools/probes/band_blocks.py c2 8 40
run band_blocks python tools/probes/band_blocks.py c4 8 40
run timeline python tools/probes/timeline.py c2
run timeline python tools/probes/timeline.py c3
run share_exchange python tools/probes/share_exchange.py c3
run share_exchange python tools/probes/share_exchange.py c2 8 64
for m in 0 1 2 3; do run torch_effect python tools/probes/torch_effect.py c3 $m; done
echo probes done
