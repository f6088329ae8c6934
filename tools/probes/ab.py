"""A/B timing: python tools/probes/ab.py <workload> <opt>=<v1,v2,..> [frames]  -> median/min ms per option value, interleaved."""
import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; opt, vals = sys.argv[2].split('='); opt=int(opt); vals=[int(v) for v in vals.split(',')]
n = int(sys.argv[3]) if len(sys.argv)>3 else 30
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
rs = {}
for v in vals:
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1,0); r.setOption(opt,v); r.setScene(sc, honour_max_depth=True); rs[v]=r
pc = sc.push_constants()
ms = {v:[] for v in vals}
for i in range(n+3):
    for v in vals:
        r=rs[v]; r.timerBegin(); r.render(pc); r.timerEnd(); t=r.timerElapsedMs()
        if i>=3: ms[v].append(t)
ref=None
for v in vals:
    f = rs[v].readbackRGBA8()
    if ref is None: ref=f
    a=np.array(ms[v]); print(f"{name} opt{opt}={v}: median {np.median(a):.3f} min {a.min():.3f} mean {a.mean():.3f} max {a.max():.3f}  same-frame {np.array_equal(ref,f)}")
