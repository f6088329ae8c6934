"""python tools/probes/c2_modes.py <lib.so> <tune> [tune ...]: C2 kernel ms per frame, one at a time and two in flight, per tune word."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config2(); pc = sc.push_constants()
for tune in [int(x, 0) for x in sys.argv[2:]] or [0]:
    out = []
    for fif in (1, 2):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setOption(100, tune); r.setScene(sc, honour_max_depth=True)
        res = []
        for rep in range(5):
            r.timerBegin()
            for i in range(40): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / 40)
        out.append(float(np.median(res[1:]))); r.shutdown()
    print(f'{sys.argv[1]} tune {tune:#x}: one at a time {out[0]:.4f} ms, two in flight {out[1]:.4f} ms', flush=True)
