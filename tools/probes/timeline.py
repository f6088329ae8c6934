"""python tools/probes/timeline.py <workload> [tune]: per-wave timeline of one frame (three clock stamps per wave):
when the tile tickets ran dry, how the waves' exits spread, how long a pass takes before and after the tickets ran dry."""
import sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; tune = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1, 0); r.setOption(6, 1); r.setOption(100, tune); r.setScene(sc, honour_max_depth=True)
for _ in range(3): r.render(pc)
r.present()
for rep in range(2):
    raw = r.waveTimeline(pc)
passes, passes_at_dry = (raw[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64), (raw[:, 3] >> np.uint64(32)).astype(np.float64)
t = raw.astype(np.float64); t0 = t[:, 0].min(); T = t[:, 2].max() - t0  # 100 MHz clock shared by all compute units
start, dry, end = (t[:, 0] - t0) / T, (t[:, 1] - t0) / T, (t[:, 2] - t0) / T
hist = np.histogram(end, bins=20, range=(0, 1))[0]

q = lambda a: ' '.join('%.3f' % v for v in np.quantile(a, [0, .1, .25, .5, .75, .9, .99, 1]))
print(f'{name}: {len(t)} waves, kernel span {T / 100:.1f} us; quantiles 0/10/25/50/75/90/99/100 % as a fraction of the span')
print(' start        ', q(start))
print(' tickets dry  ', q(dry))
print(' exit         ', q(end))
print(' alive share of span: %.3f; wave time after dry: %.3f of all wave time; passes per wave median %d max %d' %
      ((end - start).mean(), (end - dry).sum() / (end - start).sum(), np.median(passes), passes.max()))
print(' waves exiting per 5 % of the span:', ' '.join(str(v) for v in hist))
order = np.argsort(end)[-8:]
bulk_us = dry * T / 100 / np.maximum(passes_at_dry, 1); tail_n = passes - passes_at_dry
tail_us = (end - dry) * T / 100 / np.maximum(tail_n, 1)
order = np.argsort(end)[-8:]
print(' us per pass before the tickets ran dry: median %.2f; after, for the last 8 waves: %s (passes after dry: %s)' %
      (np.median(bulk_us[passes_at_dry > 0]), ' '.join('%.2f' % tail_us[i] for i in order), ' '.join('%d' % tail_n[i] for i in order)))
print(' last waves: exit', ' '.join('%.3f' % end[i] for i in order), '| dry', ' '.join('%.3f' % dry[i] for i in order))
