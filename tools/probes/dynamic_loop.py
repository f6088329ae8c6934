"""python tools/probes/dynamic_loop.py [frames]: bench.py's dynamic_loop leg alone (updateScene + render per frame along the
reference's orbit), e.g. under rocprofv3 --kernel-trace --memory-copy-trace."""
import json, sys
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.dynamic_loop(0, int(sys.argv[1]) if len(sys.argv) > 1 else 120), indent=1))
