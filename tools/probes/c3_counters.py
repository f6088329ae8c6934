import sys, time, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
name = sys.argv[1] if len(sys.argv)>1 else 'c3'
sc = {'c3':scenes.config3,'c2':scenes.config2}[name]()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
import os
if os.environ.get('TUNE'): r.setOption(100,int(os.environ['TUNE']))
r.setOption(1,0); r.setScene(sc, honour_max_depth=True); pc = sc.push_constants()
cnt = r.renderCounted(pc)
print(dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt])))
ns = len(sc.spheres); groups=(ns+7)//8
wave_passes_min = cnt[0]/64
tot=sum(int(cnt[i]) for i in range(13,18)); print('phase share: refill %.3f spheres %.3f ellipsoids %.3f shade %.3f exact %.3f; cycles/pass/wave %.0f' % tuple([int(cnt[i])/tot for i in range(13,18)]+[tot/(cnt[0]/64)]))
print('rays/px', cnt[0]/(sc.width*sc.height), 'ideal trips', wave_passes_min*groups, 'actual trips', int(cnt[10]), 'lane util', wave_passes_min*groups/cnt[10], 'slow frac', cnt[11]/cnt[10], 'roots/ray', cnt[8]/cnt[0])
for i in range(3):
    r.timerBegin(); r.render(pc); r.timerEnd(); print('ms', r.timerElapsedMs())
print(r.lastLaunch())
