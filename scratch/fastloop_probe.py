import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
sc = scenes.config3()
# move every sphere far behind the camera, tiny radius: no ray line touches any sphere
n=len(sc.spheres)
rng=np.random.default_rng(1)
sc.spheres['center'] = np.stack([rng.uniform(300,400,n), rng.uniform(1000,1100,n), rng.uniform(500,600,n)],1).astype(np.float32)
sc.spheres['radius'] = 0.01
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setScene(sc); pc = sc.push_constants()
cnt = r.renderCounted(pc)
d=dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt])); print(d)
for i in range(3):
    r.timerBegin(); r.render(pc); r.timerEnd(); ms=r.timerElapsedMs()
    print('ms', ms, 'cycles/trip/SIMD @2.4GHz', ms*1e-3*2.4e9*1024/d['wave_trips'], 'alg TF', d['sphere_tests']*17/ms/1e9)
