"""python tools/probes/passes.py <lib.so> [tune]: passes per wave (the length of the longest chain of dependent passes) on small
frames of the 64-object scene, where every wave holds one tile: what the end of a frame looks like for a wave."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
tune = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
for (w, h) in ((64, 40), (256, 144), (480, 270)):
    sc = scenes.config2(w, h); pc = sc.push_constants()
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, 1); r.setOption(100, tune); r.setScene(sc, honour_max_depth=True)
    for _ in range(2): r.render(pc)
    r.present()
    raw = r.waveTimeline(pc)
    passes = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t = raw.astype(np.float64); span = (t[:, 2].max() - t[:, 0].min()) / 100
    ms = []
    for rep in range(3):
        r.timerBegin()
        for i in range(20): r.render(pc)
        r.timerEnd(); ms.append(r.timerElapsedMs() / 20)
    print(f'{w}x{h}: waves {len(passes)} passes median {int(np.median(passes))} p90 {int(np.quantile(passes, .9))} max {passes.max()}; span {span:.1f} us; ms/frame {min(ms):.4f}')
    r.shutdown()
