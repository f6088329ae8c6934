"""python tools/probes/trilinear.py [lib.so]: kernel ms of the C4 frame with the nearest and the trilinear filter, one frame at a time."""
import sys, numpy as np
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    from path_tracer_amd import _lib
    _lib.LIB_PATH = __import__('os').path.abspath(sys.argv[1])
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config4(); pc = sc.push_constants()
for filt in (0, 1):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, 1); r.setOption(5, filt); r.setScene(sc, honour_max_depth=True)
    ms = []
    for rep in range(5):
        r.timerBegin()
        for i in range(20): r.render(pc)
        r.timerEnd(); ms.append(r.timerElapsedMs() / 20)
    print(sys.argv[1] if len(sys.argv) > 1 else 'tree', 'filter', filt, 'median %.4f ms' % np.median(ms[1:]), r.lastLaunch()); r.shutdown()
