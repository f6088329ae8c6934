"""python tools/probes/c5_batch.py [lib.so]: the 4K objects + volume frame, 64 samples per pixel, at several values of
PT_OPT_SAMPLE_BATCH (1 = one launch per sample), and one sample per frame with two frames in flight for comparison."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from path_tracer_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from path_tracer_amd import scenes, HipRenderer, types as T
sc = scenes.config5(); pc = sc.push_constants()
for batch in (1, 0, 8, 1, 0):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(T.OPT_SAMPLE_BATCH, batch); r.setScene(sc, honour_max_depth=True)
    r.renderSamples(pc, 0, 64); r.present()
    ms = []
    for rep in range(3):
        r.timerBegin(); r.renderSamples(pc, 0, 64); r.timerEnd(); ms.append(r.timerElapsedMs())
    print(f"batch {batch}: 64 spp {np.median(ms):.2f} ms  launch {r.lastLaunch()}", flush=True)
    r.shutdown()
for fif in (1, 2):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
    ms = []
    for rep in range(4):
        r.timerBegin()
        for i in range(32): r.render(pc)
        r.timerEnd(); ms.append(r.timerElapsedMs() / 32)
    print(f"one sample per frame, {fif} in flight: {np.median(ms[1:]):.3f} ms per frame", flush=True)
    r.shutdown()
