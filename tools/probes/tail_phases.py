"""python tools/probes/tail_phases.py <lib.so>: for a build whose wave log's first word carries three 20-bit sums of
100 MHz ticks taken after the tickets ran dry (object loops, shading sweep, fan-out block), the split of the last
waves' time after dry.  (A measurement build; the committed library logs the wave's start stamp there.)"""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = __import__('os').path.abspath(sys.argv[1])
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config2(); pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1, 0); r.setOption(6, 1); r.setScene(sc, honour_max_depth=True)
for _ in range(3): r.render(pc)
r.present()
for rep in range(2): raw = r.waveTimeline(pc)
w0 = raw[:, 0]
fan = (w0 & np.uint64(0xFFFFF)).astype(np.float64); sweep = ((w0 >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64)
obj = ((w0 >> np.uint64(40)) & np.uint64(0xFFFFF)).astype(np.float64)
dry, end = raw[:, 1].astype(np.float64), raw[:, 2].astype(np.float64)
passes, at_dry = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64), (raw[:, 3] >> np.uint64(32)).astype(np.float64)
order = np.argsort(end)[-12:]
print('last 12 waves: us after dry | passes after dry | us per pass: objects, sweep, fan-out block')
for i in order:
    n = max(passes[i] - at_dry[i], 1)
    print('  %7.1f | %3d | %5.2f %5.2f %5.2f' % ((end[i] - dry[i]) / 100, n, obj[i] / 100 / n, sweep[i] / 100 / n, fan[i] / 100 / n))
tot = obj.sum() + sweep.sum() + fan.sum()
print('all waves after dry: objects %.3f sweep %.3f fan-out %.3f' % (obj.sum() / tot, sweep.sum() / tot, fan.sum() / tot))
