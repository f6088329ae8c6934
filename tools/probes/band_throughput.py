"""One rank's share of a 1/2/4/8-way band split, timed alone on one GPU, with 1, 2, 3, 4, 6 and 8 frames in flight."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fixed = [tuple(int(x) for x in a.split(':')) for a in sys.argv[3:]]
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name]()
pc = sc.push_constants()
base = {}
for world in (1, 2, 4, 8):
    out = []
    for fif in (1, 2, 3, 4, 6, 8):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif)
        for k, val in fixed: r.setOption(k, val)
        r.setScene(sc, honour_max_depth=True)
        if world > 1: r.setRowBands(16, 0, world)
        res = []
        for rep in range(3):
            r.timerBegin()
            for i in range(n): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / n)
        ms = float(np.median(res)); r.shutdown()
        if world == 1: base[fif] = ms
        out.append(f"{fif}: {ms:.3f} ms ({base[fif] / ms:.2f}x)")
    print(name, 'world', world, ' | '.join(out))
