import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from path_tracer_amd import scenes, HipRenderer, types as T
for name, sc in (("c5 (objects + volume)", scenes.config5()), ("c2 objects at 4K", scenes.config2(3840, 2160))):
    pc = sc.push_constants()
    for fif in (1, 2):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
        ms = []
        for rep in range(4):
            r.timerBegin()
            for i in range(24): r.render(pc)
            r.timerEnd(); ms.append(r.timerElapsedMs() / 24)
        print(f"{name}, {fif} in flight: {np.median(ms[1:]):.3f} ms per frame", flush=True)
        r.shutdown()
