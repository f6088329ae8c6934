"""Frames per timed region: python tools/probes/throughput.py <workload> <option>=<v1,v2,..> [frames] [opt:val ...] -> ms per frame.
Trailing opt:val pairs are set on every renderer before the swept option."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; opt, vals = sys.argv[2].split('='); opt = int(opt); vals = [int(v) for v in vals.split(',')]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
fixed = [tuple(int(x) for x in a.split(':')) for a in sys.argv[4:]]
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name]()
pc = sc.push_constants()
for v in vals:
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0)
    for k, val in fixed: r.setOption(k, val)
    r.setOption(opt, v); r.setScene(sc, honour_max_depth=True)
    res = []
    for rep in range(4):
        r.timerBegin()
        for i in range(n):
            r.render(pc)
        r.timerEnd()
        res.append(r.timerElapsedMs() / n)
    print(f"{name} opt{opt}={v}: ms/frame over {n}-frame regions: " + ' '.join('%.3f' % x for x in res))
    r.shutdown()
