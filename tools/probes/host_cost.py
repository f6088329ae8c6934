"""Host time of one pt_render call (enqueue only): python tools/probes/host_cost.py
A tiny frame, many calls without waiting, wall time per call; with and without the tile-order pass."""
import sys, time
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config2(192, 108)
pc = sc.push_constants()
for order in (0, 1):
    for fif in (1, 8):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(3, order); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
        for i in range(50): r.render(pc)
        r.present()
        n = 2000
        t0 = time.perf_counter()
        for i in range(n): r.render(pc)
        t1 = time.perf_counter()
        r.present()
        t2 = time.perf_counter()
        print(f"tile order {order}, {fif} in flight: enqueue {1e6 * (t1 - t0) / n:.1f} us per call, with the wait {1e6 * (t2 - t0) / n:.1f} us per frame")
        r.shutdown()
