import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
name = sys.argv[1]; world=int(sys.argv[2])
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setScene(sc, honour_max_depth=True)
if world>1: r.setRowBands(16, 0, world)
cnt = r.renderCounted(pc)
d=dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt]))
ms=[]
for i in range(5):
    r.timerBegin(); r.render(pc); r.timerEnd(); ms.append(r.timerElapsedMs())
ms=float(np.median(ms[1:]))
tot=sum(int(cnt[i]) for i in range(13,18))
blocks,threads,lds = r.lastLaunch()
waves = blocks*threads//64
print(name,'world',world,'ms %.3f'%ms,'rays %.2fM'%(d['rays']/1e6),'waves',waves,'passes/wave min %.1f'%(d['rays']/64/waves),
      'trips/wave %.0f'%(d['wave_trips']/waves), 'lane util %.3f'%((d['rays']/64*((len(sc.spheres)+7)//8))/max(d['wave_trips'],1)),
      'exact rays %d'%d['exact_rays'],
      'alive frac @2.1GHz %.2f'%(tot/waves/(ms*1e-3*2.1e9)),
      'phase: refill %.3f spheres %.3f ell %.3f shade %.3f exact %.3f'%tuple(int(cnt[i])/tot for i in range(13,18)), 'pooled share of wave time %.3f'%(int(cnt[18])/tot), 'pooled passes per WG %.1f'%(int(cnt[19])/blocks), 'pooled rays per WG-pass %.0f'%(int(cnt[20])/max(int(cnt[19]),1)), 'pooled rays total %.2fM'%(int(cnt[20])/1e6))
