"""python tools/probes/phase_small.py <lib.so> [w h] [tune]: per-phase clock share and cycles per pass on a small frame of the
64-object scene (one tile per wave, waves alone on their SIMDs): where a pass's latency goes when nothing hides it."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer, types
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 40)
tune = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
sc = scenes.config2(w, h); pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1, 0); r.setOption(6, 1); r.setOption(100, tune | (1 << 26)); r.setScene(sc, honour_max_depth=True)
cnt = r.renderCounted(pc)
c = dict(zip(types.COUNTER_NAMES, [int(x) for x in cnt]))
raw = r.waveTimeline(pc)
passes = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
names = ['cyc_refill', 'cyc_spheres', 'cyc_ellipsoids', 'cyc_shade', 'cyc_exact']
tot = sum(int(cnt[i]) for i in range(13, 18))
print(f'{w}x{h} tune {tune}: rays {c["rays"]}, wave passes {passes.sum()}, cycles per wave-pass {tot / passes.sum():.0f} (100 MHz x21 = shader clocks?)')
print(' '.join(f'{n} {int(cnt[13 + i]) / passes.sum():.0f}' for i, n in enumerate(['refill', 'spheres', 'ell', 'shade', 'exact'])))
print(c)
