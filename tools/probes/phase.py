import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
name=sys.argv[1]; tune=int(sys.argv[2])
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setOption(100,tune); r.setScene(sc); pc = sc.push_constants()
cnt = r.renderCounted(pc)
tot=sum(int(cnt[i]) for i in range(13,18))
print(name,'tune',tune,'phase share: refill %.3f spheres %.3f ellipsoids %.3f shade %.3f exact %.3f; cycles/pass/wave %.0f' % tuple([int(cnt[i])/tot for i in range(13,18)]+[tot/(cnt[0]/64)]))
