import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from path_tracer_amd import scenes, HipRenderer, types as T
sc = scenes.config4(); pc = sc.push_constants()
for batch in (1, 0, 1, 0):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(T.OPT_SAMPLE_BATCH, batch); r.setScene(sc, honour_max_depth=True)
    r.renderSamples(pc, 0, 64); r.present()
    ms = []
    for rep in range(5):
        r.timerBegin(); r.renderSamples(pc, 0, 64); r.timerEnd(); ms.append(r.timerElapsedMs())
    print(f"volume frame, 64 samples, batch {batch}: {np.median(ms):.3f} ms", flush=True)
    r.shutdown()
