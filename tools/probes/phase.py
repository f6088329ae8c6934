import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
name=sys.argv[1]; tune=int(sys.argv[2])
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setOption(100,tune|(1<<26)); r.setScene(sc, honour_max_depth=True); pc = sc.push_constants()
cnt = r.renderCounted(pc)  # with the per-phase stamps

tot=sum(int(cnt[i]) for i in range(13,18))
print(name,'tune',tune,'phase share: refill %.3f spheres %.3f ellipsoids %.3f shade %.3f exact %.3f; cycles/pass/wave %.0f' % tuple([int(cnt[i])/tot for i in range(13,18)]+[tot/(cnt[0]/64)]))
c=dict(zip(__import__('path_tracer_amd').types.COUNTER_NAMES,[int(x) for x in cnt]))
print('tickets %d, wave-cycles per ticket %.0f, ticket share of all wave time %.3f; tail (after the tickets ran out) share %.3f; alive wave-cycles %.3e'
      % (c['tickets'], c['cyc_ticket']/max(c['tickets'],1), c['cyc_ticket']/c['cyc_alive'], c['cyc_tail']/c['cyc_alive'], c['cyc_alive']))
import time
r.setOption(6,1)
for _ in range(3): r.render(pc)
r.present(); t0=time.perf_counter()
for _ in range(20): r.render(pc)
r.present(); dt=(time.perf_counter()-t0)/20
b,t,l=r.lastLaunch()
print('ms/frame %.4f; waves %d; alive share of kernel time at 2.1 GHz: %.3f' % (dt*1e3, b*t//64, c['cyc_alive']/(b*t/64)/(dt*2.1e9)))
