"""python tools/probes/phase_sweep.py <lib.so>: share of wave time per phase of a C2 pass, with the shading sweep split
into its parts.  For a measurement build of the library whose phase stamps are placed that way (counters 13-17, 21,
22: refill, spheres + exact routine, ellipsoids, path result, results of the lighting rays, next lighting ray, the
rest); the committed library stamps refill / spheres / ellipsoids / shading / exact routine (tools/probes/phase.py)."""
import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import _lib
_lib.LIB_PATH = __import__('os').path.abspath(sys.argv[1])
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config2()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1,0); r.setOption(100,(1<<26)); r.setScene(sc, honour_max_depth=True); pc = sc.push_constants()
cnt = r.renderCounted(pc)
names={13:'refill',14:'spheres+exact',15:'ellipsoids',17:'apply results',16:'path result',21:'next lighting ray',22:'after lighting+finish+fanout'}
tot=sum(int(cnt[i]) for i in names)
for i,n in names.items(): print('%-32s %.3f' % (n, int(cnt[i])/tot))
print('cycles per pass-wave', tot/(int(cnt[0])/64))
