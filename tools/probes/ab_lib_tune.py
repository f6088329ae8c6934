"""python tools/probes/ab_lib_tune.py <lib.so> <workload> <tune,tune,...> [frames]: one library, several values of the
tuning word (pt_set_option 100), interleaved, one frame at a time and two in flight; median kernel ms per frame."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = __import__('os').path.abspath(sys.argv[1])
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[2]; tunes = [int(v, 0) for v in sys.argv[3].split(',')]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
for fif in (1, 2):
    rs = {}
    for t in tunes:
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setOption(100, t); r.setScene(sc, honour_max_depth=True); rs[t] = r
    ms = {t: [] for t in tunes}
    for rep in range(5):
        for t in tunes:
            r = rs[t]; r.timerBegin()
            for i in range(n): r.render(pc)
            r.timerEnd(); ms[t].append(r.timerElapsedMs() / n)
    for t in tunes:
        print(f"{sys.argv[1]} {name} in-flight {fif} tune {t:#x}: median {np.median(ms[t][1:]):.4f} ms  launch {rs[t].lastLaunch()}")
        rs[t].shutdown()
