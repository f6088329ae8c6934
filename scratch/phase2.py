import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
name=sys.argv[1]; tune=int(sys.argv[2]); depth=int(sys.argv[3]) if len(sys.argv)>3 else 8
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name](max_depth=depth)
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setOption(100,tune); r.setScene(sc); pc = sc.push_constants()
cnt = r.renderCounted(pc)
d=dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt]))
r.timerBegin(); r.render(pc); r.timerEnd(); ms=r.timerElapsedMs()
tot=sum(int(cnt[i]) for i in range(13,18))
blocks,threads,lds = r.lastLaunch(); waves=blocks*threads//64
print(name,'tune',tune,'depth',depth,'launch',blocks,threads,lds,'ms %.3f'%ms,'trips/wave %.0f'%(d['wave_trips']/waves),'slow %.3f'%(d['wave_slow_trips']/max(d['wave_trips'],1)),
 'cycles/wave %.2fM'%(tot/waves/1e6), 'cycles per trip per wave %.0f'%(int(cnt[14])/max(d['wave_trips'],1)),
 'phase share: refill %.3f spheres %.3f ellipsoids %.3f shade %.3f exact %.3f' % tuple([int(cnt[i])/tot for i in range(13,18)]))
