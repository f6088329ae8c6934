"""python tools/probes/band_blocks.py <workload> <world> [frames]: one rank's share of a <world>-way band split with 8 frames
in flight, for the library's own choice of the launch and for forced launches of 1, 1/2, 1/4, 1/8 of the resident workgroups (tune bits 27-29)."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; world = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
for fif in (1, 2, 4, 8):
    out = []
    for sh in (0, 1, 2, 3, 4):  # 0: automatic
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setOption(100, sh << 27); r.setScene(sc, honour_max_depth=True)
        if world > 1: r.setRowBands(16, 0, world)
        res = []
        for rep in range(4):
            r.timerBegin()
            for i in range(n): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / n)
        out.append(f"{'auto' if sh == 0 else '1/%d' % (1 << (sh - 1))}: {np.median(res[1:]):.3f} ms {r.lastLaunch()[:2]}"); r.shutdown()
    print(name, 'world', world, 'in flight', fif, ' | '.join(out))
