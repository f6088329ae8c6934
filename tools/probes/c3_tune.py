"""python tools/probes/c3_tune.py <lib.so> <tune>: kernel ms of the C3 frame for one build and one tuning word, one frame at a
time and two in flight (used to compare a build bound to 768 threads per workgroup, tune 0x2c0000, with the committed one)."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = __import__('os').path.abspath(sys.argv[1])
tune = int(sys.argv[2], 0)
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config3(); pc = sc.push_constants()
for fif in (1, 2):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, fif); r.setOption(100, tune); r.setScene(sc, honour_max_depth=True)
    ms = []
    for rep in range(4):
        r.timerBegin()
        for i in range(4): r.render(pc)
        r.timerEnd(); ms.append(r.timerElapsedMs() / 4)
    print(sys.argv[1], hex(tune), 'in flight', fif, 'median %.3f ms' % np.median(ms[1:]), r.lastLaunch()); r.shutdown()
